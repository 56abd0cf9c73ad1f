import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import panoptikon_amd as pvs
from panoptikon_amd import _lib as L
lib = pvs.lib()
N, D, B, K = 4_000_000, 768, 32, 50
ix = pvs.VectorIndex(pvs.I8, D, capacity_rows=N)
ix.set_scale(1.0 / 127 * 0.2)
stage = pvs.DeviceBuffer(1_000_000 * D * 4)
rng = np.random.default_rng(1)
for off in range(0, N, 1_000_000):
    L.check(lib.pvs_synth_rows_f32(0, 1, off, 1_000_000, D, stage.ptr))
    g = np.sort(rng.integers(off // 3, (off + 1_000_000) // 3, 1_000_000)).astype(np.int64)  # ~3 rows per group
    ix.add_f32((stage, 1_000_000), group_ids=g) if False else L.check(lib.pvs_index_add_f32(ix._h, stage.ptr, 1_000_000, None, g.ctypes.data, L.DEVICE))
q = rng.standard_normal((B, D)).astype(np.float32)
q /= np.linalg.norm(q, axis=1, keepdims=True)
for path, name in ((0, "filter-scan MIN"), (1, "dense MIN")):
    ix.set_path(path)
    ix.search_groups(q, K, pvs.COSINE, pvs.AGG_MIN)
    t = time.perf_counter()
    for _ in range(3):
        r = ix.search_groups(q, K, pvs.COSINE, pvs.AGG_MIN)
    print(f"search_groups {name}: {(time.perf_counter()-t)/3*1e3:.2f} ms per batch of {B} (N={N}, ~3 rows/group)", flush=True)
