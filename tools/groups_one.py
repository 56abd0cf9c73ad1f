"""One per-item configuration in a loop (for rocprofv3): python tools/groups_one.py <agg avg|max|weighted> <metric cosine|l2> <fused 0|1> [rows] [batch] [reps]"""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import panoptikon_amd as pvs
from panoptikon_amd import _lib as L
lib = pvs.lib()
aggn, metn, fused = sys.argv[1], sys.argv[2], int(sys.argv[3])
N = int(sys.argv[4]) if len(sys.argv) > 4 else 4_000_000
B = int(sys.argv[5]) if len(sys.argv) > 5 else 32
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 10
D, K = 768, 50
ix = pvs.VectorIndex(pvs.I8, D, capacity_rows=N)
ix.set_scale(1.0 / 127 * 0.2)
stage = pvs.DeviceBuffer(1_000_000 * D * 4)
rng = np.random.default_rng(1)
for off in range(0, N, 1_000_000):
    m = min(1_000_000, N - off)
    L.check(lib.pvs_synth_rows_f32(0, 1, off, m, D, stage.ptr))
    g = np.sort(rng.integers(off // 3, (off + m) // 3 + 1, m)).astype(np.int64)
    L.check(lib.pvs_index_add_f32(ix._h, stage.ptr, m, None, g.ctypes.data, L.DEVICE))
stage.free()
q = rng.standard_normal((B, D)).astype(np.float32)
q /= np.linalg.norm(q, axis=1, keepdims=True)
w = (rng.random(N) + 0.05).astype(np.float32) if aggn == "weighted" else None
agg = pvs.AGG_MAX if aggn == "max" else pvs.AGG_AVG
metric = pvs.COSINE if metn == "cosine" else pvs.L2
pvs.debug_set("no_fused_agg", 0 if fused else 1)
ix.search_groups(q, K, metric, agg, w)
t = time.perf_counter()
for _ in range(reps):
    ix.search_groups(q, K, metric, agg, w)
print(f"{aggn} {metn} fused={fused}: {(time.perf_counter() - t) / reps * 1e3:.3f} ms per batch of {B}")
ix.close()
