"""k_dense_exact2 (two rows per lane) against k_dense_exact (one) for 8 float queries through pvs_score_batch into device memory.
Usage: python tools/dense2_ab.py [rows]"""
import ctypes as C, json, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import panoptikon_amd as pvs
from panoptikon_amd import _lib as L
lib = pvs.lib()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
D = 768
res = {}
for name, dt in (("f16", pvs.F16), ("f32", pvs.F32)):
    ix = pvs.VectorIndex(dt, D, capacity_rows=N)
    stage = pvs.DeviceBuffer(1_000_000 * D * 4)
    for off in range(0, N, 1_000_000):
        m = min(1_000_000, N - off)
        L.check(lib.pvs_synth_rows_f32(0, 1, off, m, D, stage.ptr))
        ix.add_f32((stage, m))
    stage.free()
    q = np.random.default_rng(1).standard_normal((8, D)).astype(np.float32)
    out = pvs.DeviceBuffer(N * 8 * 4)
    for metric, mn in ((pvs.COSINE, "cosine"), (pvs.L2, "l2")):
        row = {}
        for off_ in (1, 0, 1, 0):
            pvs.debug_set("no_dense2", off_)
            L.check(lib.pvs_score_batch(ix._h, q.ctypes.data, L.F32, 8, metric, C.c_void_p(out.ptr), L.DEVICE))
            t = time.perf_counter()
            for _ in range(5):
                L.check(lib.pvs_score_batch(ix._h, q.ctypes.data, L.F32, 8, metric, C.c_void_p(out.ptr), L.DEVICE))
            row.setdefault("one_row_ms" if off_ else "two_rows_ms", []).append(round((time.perf_counter() - t) / 5 * 1e3, 3))
        pvs.debug_set("no_dense2", 0)
        res[f"{name}_{mn}"] = row
        print(name, mn, row, flush=True)
    out.free()
    ix.close()
print(json.dumps(res))
