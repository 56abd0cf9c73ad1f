"""A few 32-query pvs_score_batch calls over 1M x 768 float rows (for counters / traces of k_exact_wide).  Usage: python tools/one_wide.py [f16|f32] [batch]"""
import ctypes as C, sys
sys.path.insert(0, "/root/repo")
import numpy as np
import panoptikon_amd as pvs
from panoptikon_amd import _lib as L
lib = pvs.lib()
dt = sys.argv[1] if len(sys.argv) > 1 else "f16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
N, D = 1_000_000, 768
ix = pvs.VectorIndex(pvs.F16 if dt == "f16" else pvs.F32, D, capacity_rows=N)
stage = pvs.DeviceBuffer(N * D * 4)
L.check(lib.pvs_synth_rows_f32(0, 1, 0, N, D, stage.ptr))
ix.add_f32((stage, N))
stage.free()
q = np.random.default_rng(1).standard_normal((B, D)).astype(np.float32)
out = pvs.DeviceBuffer(N * B * 4)
for _ in range(4):
    L.check(lib.pvs_score_batch(ix._h, q.ctypes.data, L.F32, B, pvs.COSINE, C.c_void_p(out.ptr), L.DEVICE))
out.free()
ix.close()
