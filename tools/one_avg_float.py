"""Five 32-query per-item AVG calls over 4M x 768 f16 rows (for a kernel trace).  Usage: python tools/one_avg_float.py [f16|f32] [batch]"""
import sys
sys.path.insert(0, "/root/repo")
import numpy as np
import panoptikon_amd as pvs
from panoptikon_amd import _lib as L
lib = pvs.lib()
dt = sys.argv[1] if len(sys.argv) > 1 else "f16"
B = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 32
N, D = 4_000_000, 768
rng = np.random.default_rng(1)
ix = pvs.VectorIndex(pvs.F16 if dt == "f16" else pvs.F32, D, capacity_rows=N)
stage = pvs.DeviceBuffer(1_000_000 * D * 4)
for off in range(0, N, 1_000_000):
    L.check(lib.pvs_synth_rows_f32(0, 1, off, 1_000_000, D, stage.ptr))
    g = np.sort(rng.integers(off // 3, (off + 1_000_000) // 3 + 1, 1_000_000)).astype(np.int64)
    L.check(lib.pvs_index_add_f32(ix._h, stage.ptr, 1_000_000, None, g.ctypes.data, L.DEVICE))
stage.free()
q = rng.standard_normal((B, D)).astype(np.float32)
for i in range(5):
    ix.search_groups(q, 50, pvs.COSINE, pvs.AGG_AVG)
ix.close()
if "--json" in sys.argv:  # (what tools/make_traffic.py reads: the region's shape and the algorithmic bytes of its dominant kernel)
    import json
    esz = 2 if dt == "f16" else 4
    print(json.dumps({"dtype": dt, "config": {"rows": N, "dim": D, "batch": B}, "roofline": {"algorithmic_bytes_per_launch": N * D * esz}}))
