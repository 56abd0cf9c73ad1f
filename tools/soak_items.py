"""Soak of the polled per-item pages (rank_values reads the device ranking's pages from pinned memory when no poisoned word is left):
N calls of search_groups / similar_to against the pages of the event-synchronised route (pvs_debug_set("no_flag_poll", 1)).
Usage: python tools/soak_items.py [calls]"""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import panoptikon_amd as pvs
from panoptikon_amd import _lib as L
lib = pvs.lib()
CALLS = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
N, D = 300_000, 256
rng = np.random.default_rng(3)
ix = pvs.VectorIndex(pvs.I8, D, capacity_rows=N)
ix.set_scale(1.0 / 127 * 0.2)
stage = pvs.DeviceBuffer(N * D * 4)
L.check(lib.pvs_synth_rows_f32(0, 1, 0, N, D, stage.ptr))
g = np.sort(rng.integers(0, N // 3 + 1, N)).astype(np.int64)
L.check(lib.pvs_index_add_f32(ix._h, stage.ptr, N, None, g.ctypes.data, L.DEVICE))
stage.free()
q = rng.standard_normal((32, 1, D)).astype(np.float32)
q4 = rng.standard_normal((8, 4, D)).astype(np.float32)
tg = [np.arange(8 * i + 1000, 8 * i + 1008, dtype=np.int64) for i in range(16)]
pvs.debug_set("no_flag_poll", 1)
ref1 = [ix.search_groups(q[i], 10, pvs.COSINE, pvs.AGG_AVG) for i in range(32)]
ref4 = [ix.search_groups(q4[i], 25, pvs.L2, pvs.AGG_MAX) for i in range(8)]
refs = [ix.similar_to(tg[i], 50, pvs.COSINE, pvs.AGG_AVG) for i in range(16)]
pvs.debug_set("no_flag_poll", 0)
same = lambda a, b: all(np.array_equal(np.asarray(x).view(np.uint8), np.asarray(y).view(np.uint8)) for x, y in zip(a, b))
bad = 0
t = time.perf_counter()
for c in range(CALLS):
    m = c % 3
    if m == 0:
        bad += not same(ix.search_groups(q[c % 32], 10, pvs.COSINE, pvs.AGG_AVG), ref1[c % 32])
    elif m == 1:
        bad += not same(ix.search_groups(q4[c % 8], 25, pvs.L2, pvs.AGG_MAX), ref4[c % 8])
    else:
        bad += not same(ix.similar_to(tg[c % 16], 50, pvs.COSINE, pvs.AGG_AVG), refs[c % 16])
el = time.perf_counter() - t
print(f"{CALLS} polled per-item calls in {el:.1f} s, pages that differ from the event-synchronised route: {bad}")
ix.close()
sys.exit(1 if bad else 0)
