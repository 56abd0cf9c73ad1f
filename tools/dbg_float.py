import sys; sys.path.insert(0,'/root/repo')
import numpy as np, oracle as orc, panoptikon_amd as pvs
rng = np.random.default_rng(77)
n, dim, k, batch = 180_000, 96, 25, 20
grp = np.sort(rng.integers(0, n // 3, n)).astype(np.int64)
rows = orc.synth_rows(191, 0, n, dim)
rows[rng.integers(0, n, 40)] = 0.0
rows[5000:5040] = rows[5000]
rows[70_000:70_030] *= np.float32(1e-3)
rows[90_000:90_020] *= np.float32(300.0)
ix = pvs.VectorIndex(pvs.F16, dim); ix.add_f32(rows, group_ids=grp)
q = orc.synth_rows(192, 0, batch, dim); q[3] = 0.0; q[5] = rows[5000]
pvs.debug_set("float_certify_trace", 1)
for metric in (pvs.COSINE, pvs.L2):
    for agg in (pvs.AGG_AVG, pvs.AGG_MAX):
        print("metric", metric, "agg", agg, flush=True)
        ix.search_groups(q, k, metric, agg)
        for sub in ([0,1,2],[5],[3],[0,4,6,7]):
            print(" sub", sub, flush=True); ix.search_groups(q[sub], k, metric, agg)
