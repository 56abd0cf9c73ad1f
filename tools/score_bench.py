import sys, time, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
import panoptikon_amd as pvs
from panoptikon_amd import _lib as L
lib = pvs.lib()
N, D = 4_000_000, 768
for name, dt in (("i8", pvs.I8), ("f16", pvs.F16), ("f32", pvs.F32)):
    ix = pvs.VectorIndex(dt, D, capacity_rows=N)
    if dt == pvs.I8:
        ix.set_scale(1.0 / 127 * 0.2)
    stage = pvs.DeviceBuffer(1_000_000 * D * 4)
    for off in range(0, N, 1_000_000):
        L.check(lib.pvs_synth_rows_f32(0, 1, off, 1_000_000, D, stage.ptr))
        ix.add_f32((stage, 1_000_000))
    q = np.random.default_rng(1).standard_normal((8, D)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    out = pvs.DeviceBuffer(N * 4)
    for metric in (pvs.COSINE, pvs.L2):
        for rep in range(2):
            t = time.perf_counter()
            for i in range(5):
                L.check(lib.pvs_score_all(ix._h, q[i].ctypes.data, L.F32, metric, C.c_void_p(out.ptr), L.DEVICE))
            dt_ = (time.perf_counter() - t) / 5
        esz = {pvs.I8: 1, pvs.F16: 2, pvs.F32: 4}[dt]
        print(f"score_all {name} metric={metric}: {dt_*1e3:.3f} ms/query  {N*D*esz/dt_/1e9:.0f} GB/s", flush=True)
    for nb in (8,):
        t = time.perf_counter()
        r = ix.score_batch(q[:nb], pvs.COSINE)
        print(f"score_batch {name} x{nb} (incl. D2H of {r.nbytes/1e6:.0f} MB): {(time.perf_counter()-t)*1e3:.1f} ms", flush=True)
    ix.close()

# host-space outputs (what a SQLite host would consume): includes the D2H of N*4 bytes per query
ix = pvs.VectorIndex(pvs.I8, D, capacity_rows=N)
ix.set_scale(1.0 / 127 * 0.2)
stage = pvs.DeviceBuffer(1_000_000 * D * 4)
for off in range(0, N, 1_000_000):
    L.check(lib.pvs_synth_rows_f32(0, 1, off, 1_000_000, D, stage.ptr))
    ix.add_f32((stage, 1_000_000))
q = np.random.default_rng(1).standard_normal((8, D)).astype(np.float32)
for rep in range(2):
    t = time.perf_counter()
    for i in range(5):
        r = ix.score_all(q[i], pvs.COSINE)
    dt_ = (time.perf_counter() - t) / 5
print(f"score_all i8 -> HOST buffer: {dt_*1e3:.2f} ms/query ({N*4/dt_/1e9:.1f} GB/s of results)", flush=True)
for rep in range(2):
    t = time.perf_counter()
    r = ix.score_batch(q, pvs.COSINE)
    dt_ = time.perf_counter() - t
print(f"score_batch i8 x8 -> HOST buffer: {dt_*1e3:.1f} ms ({r.nbytes/dt_/1e9:.1f} GB/s of results)", flush=True)
