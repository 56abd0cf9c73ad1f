"""BASELINE configs[4] on one GPU: a 512-d image-embedding index and a 1024-d text-embedding index (int8, 25M rows
each, ~3 vectors per file), PQL `or` of the two filters ranked by RRF -> pvs_rrf_search (every group of both
branches ranked exactly on the device)."""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import panoptikon_amd as pvs
from panoptikon_amd import _lib as L
lib = pvs.lib()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 25_000_000
CH = 1_000_000
branches = []
t0 = time.time()
for dim, metric, seed in ((512, pvs.COSINE, 11), (1024, pvs.L2, 12)):
    ix = pvs.VectorIndex(pvs.I8, dim, capacity_rows=N)
    ix.set_scale(1.0 / 127 * (0.25 if dim == 512 else 0.18))
    stage = pvs.DeviceBuffer(CH * dim * 4)
    for off in range(0, N, CH):
        m = min(CH, N - off)
        L.check(lib.pvs_synth_rows_f32(0, seed, off, m, dim, stage.ptr))
        g = (np.arange(off, off + m, dtype=np.int64) // 3) * (1 if dim == 512 else 2)  # text files: every other id -> partial overlap
        L.check(lib.pvs_index_add_f32(ix._h, stage.ptr, m, None, g.ctypes.data, L.DEVICE))
    stage.free()
    q = np.random.default_rng(seed).standard_normal(dim).astype(np.float32)
    q /= np.linalg.norm(q)
    branches.append(dict(index=ix, query=q, metric=metric, agg=pvs.AGG_MIN, rrf_k=5 if dim == 512 else 10, weight=1.0 if dim == 512 else 0.7))
print(f"built 2 x {N} rows in {time.time()-t0:.1f}s", flush=True)
pvs.rrf_search(branches, 100)
ts = []
for _ in range(5):
    t = time.perf_counter()
    g, s = pvs.rrf_search(branches, 100)
    ts.append(time.perf_counter() - t)
print(f"pvs_rrf_search 2 branches x {N} rows ({N//3} + {N//3} groups), k=100: {np.median(ts)*1e3:.1f} ms per query; top: {g[:3].tolist()} {s[:3].tolist()}", flush=True)
for b in branches:
    t = time.perf_counter()
    b["index"].search_groups(b["query"], 100, b["metric"], pvs.AGG_MIN)
    print(f"  single branch dim={b['index'].dim} page of 100 files (filter scan): {(time.perf_counter()-t)*1e3:.1f} ms", flush=True)
