"""configs[4] in a loop: the same composed query must return the same page every time, on both paths (race hunting)."""
import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
import panoptikon_amd as pvs
import oracle as orc
from tests.test_gpu_fullsize import _build_i8
n = int(sys.argv[1]) if len(sys.argv) > 1 else 25_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
img = _build_i8(pvs, n, 512, 11, 0.00185, groups_of=lambda r: r // 3)
txt = _build_i8(pvs, n, 1024, 12, 0.0013, groups_of=lambda r: (r // 3) * 2)
if len(sys.argv) > 3:
    img.set_streams(int(sys.argv[3]))
    txt.set_streams(int(sys.argv[3]))
qi, qt = orc.synth_rows(0x5EED0000, 0, 1, 512)[0], orc.synth_rows(0x5EED0011, 0, 1, 1024)[0]
brs = [dict(index=img, query=qi, metric=pvs.COSINE, agg=pvs.AGG_MIN, rrf_k=5, weight=1.0),
       dict(index=txt, query=qt, metric=pvs.L2, agg=pvs.AGG_MIN, rrf_k=10, weight=0.7)]
ref = None
bad = 0
for it in range(reps):
    for full in (False, True):
        pvs.debug_set("rrf_full", 1 if full else 0)
        g, s = pvs.rrf_search(brs, 100)
        if ref is None:
            ref = (g.copy(), s.copy())
        if not (np.array_equal(g, ref[0]) and np.array_equal(s.view(np.uint64), ref[1].view(np.uint64))):
            bad += 1
            d = np.nonzero(s.view(np.uint64) != ref[1].view(np.uint64))[0]
            print("iteration", it, "full" if full else "bounded", "differs at", d[:5], s[d[:5]], ref[1][d[:5]], flush=True)
print("mismatches:", bad, "of", 2 * reps)
