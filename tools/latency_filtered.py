import sys, time
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
q = rng.standard_normal((64, 1, D)).astype(np.float32)
for frac in (0.5, 0.05, 0.002):
    mask = (rng.random(N) < frac).astype(np.uint8)
    for i in range(5):
        ix.search_filtered(q[i], 10, mask, pvs.COSINE)
    ts = []
    for i in range(100):
        t = time.perf_counter()
        ix.search_filtered(q[i % 64], 10, mask, pvs.COSINE)
        ts.append(time.perf_counter() - t)
    print(f"i8 690k single query, host mask allowing {frac:.3f} of the rows: p50 {np.sort(ts)[50]*1e3:.4f} ms", flush=True)
ts = []
for i in range(100):
    t = time.perf_counter()
    ix.search(q[i % 64], 10, pvs.COSINE)
    ts.append(time.perf_counter() - t)
print(f"i8 690k single query, no mask: p50 {np.sort(ts)[50]*1e3:.4f} ms", flush=True)
ix.close()
