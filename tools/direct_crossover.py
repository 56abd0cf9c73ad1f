"""Where the one-launch single-query search (pvs_direct.hip) stops paying against the filter scan: p50 of pvs_search, k = 10 and 100,
both routes on the same index.  Usage: python tools/direct_crossover.py [out.json]"""
import json, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import panoptikon_amd as pvs
from panoptikon_amd import _lib as L
lib = pvs.lib()
rng = np.random.default_rng(1)
res = {}
D = 768
for name, dt, sizes in (("i8", pvs.I8, (1_000_000, 2_000_000, 4_000_000, 8_000_000)), ("f16", pvs.F16, (1_000_000, 2_000_000, 4_000_000)),
                        ("f32", pvs.F32, (1_000_000, 2_000_000, 4_000_000))):
    for N in sizes:
        ix = pvs.VectorIndex(dt, D, capacity_rows=N)
        if dt == pvs.I8:
            ix.set_scale(1.0 / 127 * 0.2)
        stage = pvs.DeviceBuffer(1_000_000 * D * 4)
        for off in range(0, N, 1_000_000):
            L.check(lib.pvs_synth_rows_f32(0, 1, off, 1_000_000, D, stage.ptr))
            ix.add_f32((stage, 1_000_000))
        stage.free()
        q = rng.standard_normal((64, 1, D)).astype(np.float32)
        q /= np.linalg.norm(q, axis=2, keepdims=True)
        pvs.debug_set("direct_max_mb", 1 << 20)
        for k in (10, 100):
            row = {}
            for route, nd in (("direct", 0), ("filter", 1)):
                pvs.debug_set("no_direct_topk", nd)
                for i in range(8):
                    ix.search(q[i], k, pvs.COSINE)
                ts = []
                for i in range(120):
                    t = time.perf_counter()
                    ix.search(q[i % 64], k, pvs.COSINE)
                    ts.append(time.perf_counter() - t)
                row[route] = round(float(np.sort(ts)[60]) * 1e3, 4)
            pvs.debug_set("no_direct_topk", 0)
            esz = {pvs.I8: 1, pvs.F16: 2, pvs.F32: 4}[dt]
            res[f"{name}_{N}_k{k}"] = row
            print(f"{name} N={N} ({N*D*esz/2**20:.0f} MB) k={k}: direct {row['direct']:.4f} ms  filter {row['filter']:.4f} ms", flush=True)
        pvs.debug_set("direct_max_mb", 0)
        ix.close()
print(json.dumps(res))
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write(json.dumps(res) + "\n")
