import sys, time, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
import panoptikon_amd as pvs
from panoptikon_amd import _lib as L
lib = pvs.lib()
D = 768
for N in (100_000, 690_000, 4_000_000):
    ix = pvs.VectorIndex(pvs.I8, D, capacity_rows=N)
    ix.set_scale(1.0 / 127 * 0.2)
    ch = min(N, 1_000_000)
    stage = pvs.DeviceBuffer(ch * D * 4)
    for off in range(0, N, ch):
        L.check(lib.pvs_synth_rows_f32(0, 1, off, min(ch, N - off), D, stage.ptr))
        ix.add_f32((stage, min(ch, N - off)))
    stage.free()
    q = np.random.default_rng(1).standard_normal((1, D)).astype(np.float32)
    out = pvs.DeviceBuffer(N * 4)
    for rep in range(3):
        t = time.perf_counter()
        for i in range(20):
            L.check(lib.pvs_score_all(ix._h, q.ctypes.data, L.F32, pvs.COSINE, C.c_void_p(out.ptr), L.DEVICE))
        ms = (time.perf_counter() - t) / 20 * 1e3
    print(f"score_all i8 N={N}: {ms:.4f} ms per call (wall, one sync each) = {N*D/(ms*1e-3)/1e12:.2f} TB/s", flush=True)
    out.free()
    ix.close()
