"""The dense path on a batch of tie-heavy queries (int8 L2 over real data ties massively): 10M x 768 int8 built from a
few thousand distinct vectors, 128 queries.  Times the filter scan on a tie-free corpus beside it."""
import sys, time, json
sys.path.insert(0, "/root/repo")
import numpy as np
import panoptikon_amd as pvs
from panoptikon_amd import _lib as L
lib = pvs.lib()
N, D, B, K = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000, 768, 128, 100
out = {}
stage = pvs.DeviceBuffer(1_000_000 * D * 4)
q = pvs.DeviceBuffer(B * D * 4)
L.check(lib.pvs_synth_rows_f32(0, 0x5EED0000, 0, B, D, q.ptr))
qh = q.to_numpy(np.float32, (B, D))
for name, distinct in (("tie_free", None), ("ties_4096_distinct_vectors", 4096)):
    ix = pvs.VectorIndex(pvs.I8, D, capacity_rows=N)
    ix.set_scale(0.0015)
    for off in range(0, N, 1_000_000):
        # tie-heavy corpus: every 1M-row chunk repeats the SAME 4,096 rows 244 times (row r = r mod 4096)
        if distinct is None:
            L.check(lib.pvs_synth_rows_f32(0, 1, off, 1_000_000, D, stage.ptr))
            ix.add_f32((stage, 1_000_000))
        else:
            L.check(lib.pvs_synth_rows_f32(0, 1, 0, distinct, D, stage.ptr))
            for o2 in range(0, 1_000_000, distinct):
                ix.add_f32((stage, min(distinct, 1_000_000 - o2)))
    for rep in range(2):
        t = time.perf_counter()
        ids, dist, cnt = ix.search(qh, K, pvs.L2)
        ms = (time.perf_counter() - t) * 1e3
    st = ix.stats()
    out[name] = {"ms_per_128_queries": round(ms, 2), "fast_queries": int(st.fast_queries), "dense_queries": int(st.dense_queries),
                 "first_page_ids": ids[0, :4].tolist()}
    ix.close()
print(json.dumps(out, indent=1))
