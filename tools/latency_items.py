"""Per-item (GROUP BY file) single-query latency at the reference's scale: 690k x 768, ~3 vectors per file; MIN (the default
aggregation: a row page of the one-launch search + host grouping), AVG / MAX (dense column + aggregate), similar_to."""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import panoptikon_amd as pvs
from panoptikon_amd import _lib as L
lib = pvs.lib()
N, D = 690_000, 768
rng = np.random.default_rng(1)
for name, dt in (("i8", pvs.I8), ("f32", pvs.F32)):
    ix = pvs.VectorIndex(dt, D, capacity_rows=N)
    if dt == pvs.I8:
        ix.set_scale(1.0 / 127 * 0.2)
    stage = pvs.DeviceBuffer(N * D * 4)
    L.check(lib.pvs_synth_rows_f32(0, 1, 0, N, D, stage.ptr))
    g = np.sort(rng.integers(0, N // 3 + 1, N)).astype(np.int64)
    L.check(lib.pvs_index_add_f32(ix._h, stage.ptr, N, None, g.ctypes.data, L.DEVICE))
    stage.free()
    q = rng.standard_normal((64, 1, D)).astype(np.float32)
    for agg, an in ((pvs.AGG_MIN, "MIN"), (pvs.AGG_AVG, "AVG"), (pvs.AGG_MAX, "MAX")):
        for nd in (0, 1):
            pvs.debug_set("no_direct_topk" if agg == pvs.AGG_MIN else "no_fused_agg", nd)
            for i in range(5):
                ix.search_groups(q[i], 10, pvs.COSINE, agg)
            ts = []
            for i in range(100):
                t = time.perf_counter()
                ix.search_groups(q[i % 64], 10, pvs.COSINE, agg)
                ts.append(time.perf_counter() - t)
            print(f"{name} per-item {an} k=10{(' (filter scan)' if agg == pvs.AGG_MIN else ' (dense column + k_group_aggregate)') if nd else ''}: p50 {np.sort(ts)[50]*1e3:.4f} ms", flush=True)
        pvs.debug_set("no_direct_topk", 0)
        pvs.debug_set("no_fused_agg", 0)
    ix.close()
