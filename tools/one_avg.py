import sys
sys.path.insert(0, "/root/repo")
import numpy as np
import panoptikon_amd as pvs
from panoptikon_amd import _lib as L
lib = pvs.lib()
N, D = 690_000, 768
rng = np.random.default_rng(1)
ix = pvs.VectorIndex(pvs.I8, D, capacity_rows=N)
ix.set_scale(1.0 / 127 * 0.2)
stage = pvs.DeviceBuffer(N * D * 4)
L.check(lib.pvs_synth_rows_f32(0, 1, 0, N, D, stage.ptr))
g = np.sort(rng.integers(0, N // 3 + 1, N)).astype(np.int64)
L.check(lib.pvs_index_add_f32(ix._h, stage.ptr, N, None, g.ctypes.data, L.DEVICE))
stage.free()
q = rng.standard_normal((8, 1, D)).astype(np.float32)
for i in range(8):
    ix.search_groups(q[i], 10, pvs.COSINE, pvs.AGG_AVG)
ix.close()
