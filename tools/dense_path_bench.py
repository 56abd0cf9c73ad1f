"""Time of the dense path (exact score of every row + full device sort) vs the filter scan, 10M x 768 int8."""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import panoptikon_amd as pvs
from panoptikon_amd import _lib as L
lib = pvs.lib()
N, D = 10_000_000, 768
ix = pvs.VectorIndex(pvs.I8, D, capacity_rows=N)
ix.set_scale(1.0 / 127 * 0.2)
stage = pvs.DeviceBuffer(1_000_000 * D * 4)
for off in range(0, N, 1_000_000):
    L.check(lib.pvs_synth_rows_f32(0, 1, off, 1_000_000, D, stage.ptr))
    ix.add_f32((stage, 1_000_000))
q = np.random.default_rng(1).standard_normal((4, D)).astype(np.float32)
for path, name in ((0, "filter scan"), (1, "dense score + sort")):
    ix.set_path(path)
    ix.search(q, 100, pvs.COSINE)
    t = time.perf_counter()
    for _ in range(3):
        ix.search(q, 100, pvs.COSINE)
    print(f"{name}: {(time.perf_counter()-t)/3*1e3:.2f} ms per batch of 4 (k=100)", flush=True)
