"""A few masked single-query searches (690k x 768 int8, host mask allowing half the rows) for a kernel timeline."""
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
ix.add_f32((stage, N))
stage.free()
q = rng.standard_normal((8, 1, D)).astype(np.float32)
mask = (rng.random(N) < 0.5).astype(np.uint8)
for i in range(8):
    ix.search_filtered(q[i], 10, mask, pvs.COSINE)
ix.close()
