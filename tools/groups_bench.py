"""Per-item search timings at 4M x 768 int8, ~3 vectors per file, 32 queries: MIN through the filter scan, AVG /
MAX / weighted through the fused one-pass scorer (k_scan MODE 2 + per-group fold) and through the round-3 route (N x B matrix +
k_group_aggregate, pvs_debug_set("no_fused_agg", 1)); both routes must return the same pages bit for bit.
Usage: python tools/groups_bench.py [rows] [batch] [out.json]"""
import json, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import panoptikon_amd as pvs
from panoptikon_amd import _lib as L
lib = pvs.lib()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
D, K = 768, 50
ix = pvs.VectorIndex(pvs.I8, D, capacity_rows=N)
ix.set_scale(1.0 / 127 * 0.2)
stage = pvs.DeviceBuffer(1_000_000 * D * 4)
rng = np.random.default_rng(1)
for off in range(0, N, 1_000_000):
    m = min(1_000_000, N - off)
    L.check(lib.pvs_synth_rows_f32(0, 1, off, m, D, stage.ptr))
    g = np.sort(rng.integers(off // 3, (off + m) // 3 + 1, m)).astype(np.int64)  # ~3 rows per group, a group's rows adjacent
    L.check(lib.pvs_index_add_f32(ix._h, stage.ptr, m, None, g.ctypes.data, L.DEVICE))
stage.free()
q = rng.standard_normal((B, D)).astype(np.float32)
q /= np.linalg.norm(q, axis=1, keepdims=True)
w = (rng.random(N) + 0.05).astype(np.float32)
res = {"rows": N, "dim": D, "batch": B, "k": K, "hbm_floor_ms": round(N * D / 8e12 * 1e3, 3)}


def timed(f, reps=5):
    f()
    t = time.perf_counter()
    for _ in range(reps):
        r = f()
    return (time.perf_counter() - t) / reps * 1e3, r


for bq in (1, 4, B):
    pvs.debug_set("no_fused_agg", 1)
    ms_s, r1 = timed(lambda: ix.search_groups(q[:bq], K, pvs.COSINE, pvs.AGG_MIN))
    pvs.debug_set("no_fused_agg", 0)
    ms_f, r2 = timed(lambda: ix.search_groups(q[:bq], K, pvs.COSINE, pvs.AGG_MIN))
    res[f"min_b{bq}"] = {"filter_scan_ms": round(ms_s, 3), "product_ms": round(ms_f, 3), "same_pages": bool(np.array_equal(r1[0], r2[0]) and np.array_equal(r1[1].view(np.uint64), r2[1].view(np.uint64)))}
    print("min", bq, res[f"min_b{bq}"], flush=True)
for name, kw in (("avg", dict(agg=pvs.AGG_AVG)), ("max", dict(agg=pvs.AGG_MAX)), ("weighted", dict(agg=pvs.AGG_AVG, row_weights=w))):
    for metric, mn in ((pvs.COSINE, "cosine"), (pvs.L2, "l2")):
        pvs.debug_set("no_fused_agg", 0)
        ix.set_profiling(True)
        ix.profile(reset=True)
        ms_f, rf = timed(lambda: ix.search_groups(q, K, metric, **kw))
        p = ix.profile()
        ix.set_profiling(False)
        pvs.debug_set("no_fused_agg", 1)
        ms_u, ru = timed(lambda: ix.search_groups(q, K, metric, **kw), reps=3)
        pvs.debug_set("no_fused_agg", 0)
        same = all(np.array_equal(a, b) if a.dtype != np.float64 else np.array_equal(a.view(np.uint64), b.view(np.uint64)) for a, b in zip(rf, ru))
        res[f"{name}_{mn}"] = {"fused_ms": round(ms_f, 3), "scorer_kernel_ms": round(p.scan_ms / max(p.scan_launches, 1), 3), "round3_route_ms": round(ms_u, 3),
                               "same_pages": bool(same)}
        print(name, mn, res[f"{name}_{mn}"], flush=True)
line = json.dumps(res)
print(line)
if len(sys.argv) > 3:
    import os; os.makedirs(os.path.dirname(os.path.abspath(sys.argv[3])), exist_ok=True); open(sys.argv[3], "w").write(line + "\n")
ix.close()
sys.exit(0 if all(v["same_pages"] for v in res.values() if isinstance(v, dict)) else 1)
