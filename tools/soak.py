"""Soak: many host threads hammering one index (searches of mixed batch sizes, per-item searches, dense columns,
filtered searches) on one or two streams; every result compared with a page computed up front."""
import sys, threading, time
sys.path.insert(0, "/root/repo")
import numpy as np
import panoptikon_amd as pvs
from panoptikon_amd import _lib as L
lib = pvs.lib()
N, D, T, REPS = 2_000_000, 768, 16, int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(5)
ix = pvs.VectorIndex(pvs.I8, D, capacity_rows=N)
ix.set_scale(1.0 / 127 * 0.2)
stage = pvs.DeviceBuffer(1_000_000 * D * 4)
for off in range(0, N, 1_000_000):
    L.check(lib.pvs_synth_rows_f32(0, 9, off, 1_000_000, D, stage.ptr))
    g = (np.arange(off, off + 1_000_000, dtype=np.int64) // 4)
    L.check(lib.pvs_index_add_f32(ix._h, stage.ptr, 1_000_000, None, g.ctypes.data, L.DEVICE))
stage.free()
Q = rng.standard_normal((96, D)).astype(np.float32)
Q /= np.linalg.norm(Q, axis=1, keepdims=True)
mask = (rng.random(N) < 0.3).astype(np.uint8)
ref_i, ref_d, _ = ix.search(Q, 50, pvs.COSINE)
ref_fi, ref_fd, _ = ix.search_filtered(Q[:16], 50, mask, pvs.COSINE)
ref_g, ref_v, _ = ix.search_groups(Q[:16], 20, pvs.COSINE, pvs.AGG_MIN)
ref_col = ix.score_all(Q[0], pvs.COSINE)
errors, done = [], [0]
def worker(t):
    r = np.random.default_rng(100 + t)
    try:
        for rep in range(REPS):
            kind = r.integers(0, 10)
            if kind < 6:
                nb = int(r.choice([1, 2, 7, 33]))
                sel = r.integers(0, 96, nb)
                gi, gd, _ = ix.search(Q[sel], 50, pvs.COSINE)
                ok = np.array_equal(gi, ref_i[sel]) and np.array_equal(gd.view(np.uint32), ref_d[sel].view(np.uint32))
            elif kind < 8:
                sel = r.integers(0, 16, 2)
                gi, gd, _ = ix.search_filtered(Q[sel], 50, mask, pvs.COSINE)
                ok = np.array_equal(gi, ref_fi[sel]) and np.array_equal(gd.view(np.uint32), ref_fd[sel].view(np.uint32))
            elif kind < 9:
                sel = r.integers(0, 16, 2)
                gg, gv, _ = ix.search_groups(Q[sel], 20, pvs.COSINE, pvs.AGG_MIN)
                ok = np.array_equal(gg, ref_g[sel]) and np.array_equal(gv.view(np.uint64), ref_v[sel].view(np.uint64))
            else:
                col = ix.score_all(Q[0], pvs.COSINE)
                ok = np.array_equal(col.view(np.uint32), ref_col.view(np.uint32))
            if not ok:
                errors.append((t, rep, int(kind)))
            done[0] += 1
    except Exception as e:  # noqa: BLE001
        errors.append((t, repr(e)))
for streams in (1, 2):
    ix.set_streams(streams)
    t0 = time.time()
    th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
    [x.start() for x in th]; [x.join() for x in th]
    print(f"streams={streams}: {T} threads x {REPS} calls in {time.time()-t0:.1f}s, errors: {errors[:5]} ({len(errors)})", flush=True)
st = ix.stats()
print("stats: searches", st.searches, "fast", st.fast_queries, "dense", st.dense_queries)
sys.exit(1 if errors else 0)
