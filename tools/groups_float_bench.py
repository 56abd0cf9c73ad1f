"""Per-item AVG / MAX over a FLOAT index: 4M x 768 f16 (or f32), ~3 vectors per file, 1 / 8 / 32 queries.
Usage: python tools/groups_float_bench.py [f16|f32] [rows] [out.json]"""
import json, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import panoptikon_amd as pvs
from panoptikon_amd import _lib as L
lib = pvs.lib()
dt = sys.argv[1] if len(sys.argv) > 1 else "f16"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4_000_000
D, K = 768, 50
ix = pvs.VectorIndex(pvs.F16 if dt == "f16" else pvs.F32, D, capacity_rows=N)
stage = pvs.DeviceBuffer(1_000_000 * D * 4)
rng = np.random.default_rng(1)
for off in range(0, N, 1_000_000):
    m = min(1_000_000, N - off)
    L.check(lib.pvs_synth_rows_f32(0, 1, off, m, D, stage.ptr))
    g = np.sort(rng.integers(off // 3, (off + m) // 3 + 1, m)).astype(np.int64)
    L.check(lib.pvs_index_add_f32(ix._h, stage.ptr, m, None, g.ctypes.data, L.DEVICE))
stage.free()
q = rng.standard_normal((32, D)).astype(np.float32)
q /= np.linalg.norm(q, axis=1, keepdims=True)
esz = 2 if dt == "f16" else 4
res = {"dtype": dt, "rows": N, "dim": D, "k": K, "hbm_floor_ms": round(N * D * esz / 8e12 * 1e3, 3)}


def timed(f, reps=5):
    f()
    t = time.perf_counter()
    for _ in range(reps):
        r = f()
    return (time.perf_counter() - t) / reps * 1e3, r


for bq in (1, 8, 32):
    for name, agg in (("min", pvs.AGG_MIN), ("avg", pvs.AGG_AVG), ("max", pvs.AGG_MAX)):
        for metric, mn in ((pvs.COSINE, "cosine"), (pvs.L2, "l2")):
            ms, r = timed(lambda: ix.search_groups(q[:bq], K, metric, agg))
            res[f"{name}_{mn}_b{bq}"] = round(ms, 3)
            print(name, mn, bq, round(ms, 3), flush=True)
line = json.dumps(res)
print(line)
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write(line + "\n")
ix.close()
