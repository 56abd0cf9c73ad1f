"""A few similar_to calls (690k x 768 int8, 8 target vectors) for a kernel timeline under rocprofv3."""
import sys
sys.path.insert(0, "/root/repo")
import numpy as np
import panoptikon_amd as pvs
from panoptikon_amd import _lib as L
lib = pvs.lib()
N, D, PER = 690_000, 768, 8
dt = pvs.I8 if len(sys.argv) < 2 or sys.argv[1] == "i8" else pvs.F32
ix = pvs.VectorIndex(dt, D, capacity_rows=N)
if dt == pvs.I8:
    ix.set_scale(0.0015)
stage = pvs.DeviceBuffer(N * D * 4)
L.check(lib.pvs_synth_rows_f32(0, 7, 0, N, D, stage.ptr))
g = np.arange(N, dtype=np.int64) // PER
L.check(lib.pvs_index_add_f32(ix._h, stage.ptr, N, None, g.ctypes.data, L.DEVICE))
stage.free()
targets = np.arange(8 * 1000, 8 * 1000 + PER, dtype=np.int64)
for _ in range(6):
    ix.similar_to(targets, 100, pvs.COSINE, pvs.AGG_AVG)
ix.close()
