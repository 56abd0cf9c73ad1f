import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import panoptikon_amd as pvs
from panoptikon_amd import _lib as L
lib = pvs.lib()
N, D = 690_000, 768
rng = np.random.default_rng(1)
for name, dt in (("i8", pvs.I8), ("f32", pvs.F32)):
    ix = pvs.VectorIndex(dt, D, capacity_rows=N)
    if dt == pvs.I8:
        ix.set_scale(1.0 / 127 * 0.2)
    stage = pvs.DeviceBuffer(N * D * 4)
    L.check(lib.pvs_synth_rows_f32(0, 1, 0, N, D, stage.ptr))
    ix.add_f32((stage, N))
    stage.free()
    q = rng.standard_normal((64, 1, D)).astype(np.float32)
    for k in (256, 257, 1000, 4096):
        for i in range(5):
            ix.search(q[i], k, pvs.COSINE)
        ts = []
        for i in range(60):
            t = time.perf_counter()
            ix.search(q[i % 64], k, pvs.COSINE)
            ts.append(time.perf_counter() - t)
        st = ix.stats()
        print(f"{name} 690k k={k}: p50 {np.sort(ts)[30]*1e3:.4f} ms  (dense_queries {st.dense_queries})", flush=True)
    ix.close()
