"""Per-item pages over FLOAT rows, certified route (csrc/pvs_items_float.hip) against the exact-everywhere route (pvs_debug no_float_certify):
4M x 768 rows in ~1.33M files, AVG / MAX, cosine / L2, 8 / 16 / 32 queries; candidate rows per query; similar_to at 690k x 768.
Usage: python tools/float_certify_bench.py [f16|f32] [rows] [out.json] [--quick]"""
import json, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import panoptikon_amd as pvs
from panoptikon_amd import _lib as L
lib = pvs.lib()
args = [a for a in sys.argv[1:] if not a.startswith("--")]
quick = "--quick" in sys.argv
dt = args[0] if len(args) > 0 else "f16"
N = int(args[1]) if len(args) > 1 and args[1].isdigit() else 4_000_000
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
qb = pvs.DeviceBuffer(32 * D * 4)
L.check(lib.pvs_synth_rows_f32(0, 0x5EED0000, 0, 32, D, qb.ptr))
q = qb.to_numpy(np.float32, (32, D))
esz = 2 if dt == "f16" else 4
res = {"dtype": dt, "rows": N, "dim": D, "k": K, "hbm_floor_ms": round(N * D * esz / 8e12 * 1e3, 3)}


def timed(f, reps=8):
    f(); f()
    t = time.perf_counter()
    for _ in range(reps):
        r = f()
    return (time.perf_counter() - t) / reps * 1e3, r


combos = [(32, "avg", pvs.AGG_AVG, pvs.COSINE, "cosine")] if quick else [(b, n, a, m, mn) for b in (8, 16, 32) for n, a in (("avg", pvs.AGG_AVG), ("max", pvs.AGG_MAX))
                                                                         for m, mn in ((pvs.COSINE, "cosine"), (pvs.L2, "l2"))]
for bq, name, agg, metric, mn in combos:
    c0, r0 = pvs.debug_get("float_certify_queries"), pvs.debug_get("float_certify_rows")
    ms, r = timed(lambda: ix.search_groups(q[:bq], K, metric, agg))
    nq, nr = pvs.debug_get("float_certify_queries") - c0, pvs.debug_get("float_certify_rows") - r0
    ix.set_profiling(True); ix.profile(reset=True)
    for _ in range(4):
        ix.search_groups(q[:bq], K, metric, agg)
    p = ix.profile(); ix.set_profiling(False)
    scan = p.scan_ms / max(p.scan_launches, 1)
    pvs.debug_set("no_float_certify", 1)
    ms_old, r_old = timed(lambda: ix.search_groups(q[:bq], K, metric, agg), reps=3)
    pvs.debug_set("no_float_certify", 0)
    same = all(np.array_equal(a, b) if a.dtype != np.float64 else np.array_equal(a.view(np.uint64), b.view(np.uint64)) for a, b in zip(r, r_old))
    rec = {"ms": round(ms, 3), "exact_everywhere_ms": round(ms_old, 3), "scan_ms": round(scan, 3), "scan_hbm_frac": round(N * D * esz / (scan * 1e-3) / 8e12, 3) if scan else None,
           "call_hbm_frac": round(N * D * esz / (ms * 1e-3) / 8e12, 3), "certified_queries_per_call": nq / 10, "candidate_rows_per_call": nr / 10, "same_pages": bool(same)}
    res[f"{name}_{mn}_b{bq}"] = rec
    print(name, mn, bq, rec, flush=True)
ix.close()
if not quick:
    # similar_to at the reference's scale: 8 target vectors, AVG per item
    n_sim = 690_000
    ixs = pvs.VectorIndex(pvs.F16 if dt == "f16" else pvs.F32, D, capacity_rows=n_sim)
    st = pvs.DeviceBuffer(n_sim * D * 4)
    L.check(lib.pvs_synth_rows_f32(0, 1, 0, n_sim, D, st.ptr))
    g = np.arange(n_sim, dtype=np.int64) // 8
    L.check(lib.pvs_index_add_f32(ixs._h, st.ptr, n_sim, None, g.ctypes.data, L.DEVICE))
    st.free()
    tg = np.arange(8000, 8008, dtype=np.int64)
    for metric, mn in ((pvs.COSINE, "cosine"), (pvs.L2, "l2")):
        ms, r = timed(lambda: ixs.similar_to(tg, 100, metric, pvs.AGG_AVG), reps=30)
        res[f"similar_{mn}"] = round(ms, 4)
        print("similar_to", mn, round(ms, 4), flush=True)
    ixs.close()
line = json.dumps(res)
print(line)
if len(args) > 2:
    open(args[2], "w").write(line + "\n")
