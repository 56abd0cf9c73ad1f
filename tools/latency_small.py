"""Single-query latency of the host-buffer entry point (pvs_search: H2D query, search, D2H page) at the reference's own scales
(10k ... 1M rows; its measured DB holds 690k vectors), k = 10 (the API's default page) and 100.
Usage: python tools/latency_small.py [out.json] [both]      (both: also the forced dense path, pvs_index_set_path(1))"""
import json, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import panoptikon_amd as pvs
from panoptikon_amd import _lib as L
lib = pvs.lib()
rng = np.random.default_rng(1)
res = {}
paths = (0, 1) if len(sys.argv) > 2 and sys.argv[2] == "both" else (0,)
for name, dt, D in (("i8", pvs.I8, 768), ("f32", pvs.F32, 768), ("f32", pvs.F32, 512)):
    for N in (10_000, 100_000, 690_000, 1_000_000):
        ix = pvs.VectorIndex(dt, D, capacity_rows=N)
        if dt == pvs.I8:
            ix.set_scale(1.0 / 127 * 0.2)
        stage = pvs.DeviceBuffer(N * D * 4)
        L.check(lib.pvs_synth_rows_f32(0, 1, 0, N, D, stage.ptr))
        ix.add_f32((stage, N))
        stage.free()
        q = rng.standard_normal((64, 1, D)).astype(np.float32)
        q /= np.linalg.norm(q, axis=2, keepdims=True)
        for k in (10, 100):
            for path in paths:
                ix.set_path(path)
                for i in range(10):
                    ix.search(q[i], k, pvs.COSINE)
                ts = []
                for i in range(300):
                    t = time.perf_counter()
                    ix.search(q[i % 64], k, pvs.COSINE)
                    ts.append(time.perf_counter() - t)
                ts = np.sort(np.array(ts)) * 1e3
                esz = 1 if dt == pvs.I8 else 4
                floor = N * D * esz / 8e12 * 1e3
                res[f"{name}_{D}_{N}_k{k}" + ("_dense" if path else "")] = {"p50_ms": round(float(ts[150]), 4), "p99_ms": round(float(ts[296]), 4),
                                                                               "hbm_floor_ms": round(floor, 4)}
                print(f"{name} dim {D} N={N} k={k} path={path}: p50 {ts[150]:.4f} ms p99 {ts[296]:.4f}  (HBM floor {floor:.4f})", flush=True)
        ix.set_path(0)
        ix.close()
print(json.dumps(res))
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write(json.dumps(res) + "\n")
