"""k_dense_exact, float rows: the 8-query packed instance against the 4-query form (pvs_debug_set("dense_nq4", 1)) through
pvs_score_batch into device memory.  Usage: python tools/dense_exact_bench.py [rows] [out.json]"""
import ctypes as C, json, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import panoptikon_amd as pvs
from panoptikon_amd import _lib as L
lib = pvs.lib()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
D = 768
res = {"rows": N, "dim": D}
for name, dt in (("f32", pvs.F32), ("f16", pvs.F16)):
    ix = pvs.VectorIndex(dt, D, capacity_rows=N)
    stage = pvs.DeviceBuffer(1_000_000 * D * 4)
    for off in range(0, N, 1_000_000):
        m = min(1_000_000, N - off)
        L.check(lib.pvs_synth_rows_f32(0, 1, off, m, D, stage.ptr))
        ix.add_f32((stage, m))
    stage.free()
    q = np.random.default_rng(1).standard_normal((32, D)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    out = pvs.DeviceBuffer(N * 32 * 4)
    esz = 4 if dt == pvs.F32 else 2
    for metric, mn in ((pvs.COSINE, "cosine"), (pvs.L2, "l2")):
        for nb in (8, 32):
            row = {}
            sums = {}
            for nq4 in (1, 0):
                pvs.debug_set("dense_nq4", nq4)
                for rep in range(2):
                    t = time.perf_counter()
                    for _ in range(3):
                        L.check(lib.pvs_score_batch(ix._h, q.ctypes.data, L.F32, nb, metric, C.c_void_p(out.ptr), L.DEVICE))
                    ms = (time.perf_counter() - t) / 3 * 1e3
                row["nq4_ms" if nq4 else "nq8_ms"] = round(ms, 3)
                sums[nq4] = out.to_numpy(np.uint32, (N * nb,))[: 1 << 22].copy()
            pvs.debug_set("dense_nq4", 0)
            row["same_bits"] = bool(np.array_equal(sums[0], sums[1]))
            row["nq8_GBs_per_pass"] = round(N * D * esz * (nb / 8) / (row["nq8_ms"] * 1e-3) / 1e9)
            res[f"{name}_{mn}_b{nb}"] = row
            print(name, mn, nb, row, flush=True)
    out.free()
    ix.close()
line = json.dumps(res)
print(line)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(line + "\n")
