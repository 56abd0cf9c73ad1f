"""Soak of the polled host entry (pvs_search, one-launch route): many searches with changing k and batch against the pages the
event-wait form returns; reports how often a flag word was in host memory before its page (pvs_debug_get("poll_late_pages"))."""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import panoptikon_amd as pvs
import oracle as orc
n, dim = 100_000, 200
rows = orc.synth_rows(91, 0, n, dim)
ix = pvs.VectorIndex(pvs.F16, dim)
ix.add_f32(rows)
rng = np.random.default_rng(3)
qs = orc.synth_rows(92, 0, 64, dim)
cases = [(int(rng.integers(0, 60)), int(rng.choice([1, 1, 1, 2, 4])), int(rng.choice([1, 10, 60, 120, 200, 256]))) for _ in range(400)]
pvs.debug_set("no_flag_poll", 1)
ref = [ix.search(qs[q0:q0 + nb], k, pvs.COSINE) for q0, nb, k in cases]
pvs.debug_set("no_flag_poll", 0)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
bad = 0
t = time.time()
for it in range(iters):
    q0, nb, k = cases[it % len(cases)]
    gi, gd, gc = ix.search(qs[q0:q0 + nb], k, pvs.COSINE)
    ri, rd, rc = ref[it % len(cases)]
    if not (np.array_equal(gi, ri) and np.array_equal(gd.view(np.uint32), rd.view(np.uint32)) and np.array_equal(gc, rc)):
        bad += 1
        print("MISMATCH at", it, cases[it % len(cases)], flush=True)
print(f"{iters} polled searches in {time.time() - t:.1f}s: {bad} mismatches, flag-before-page events: {pvs.debug_get('poll_late_pages')}")
ix.close()
