"""Throughput of many host threads each asking for ONE query's page (the reference's read pool, db/connection.rs:235) with and
without request coalescing.  Usage: python tools/coalesce_bench.py [--rows 10000000] [--threads 16]  -> JSON line."""
import argparse, json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import panoptikon_amd as pvs
from panoptikon_amd import _lib as L

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=10_000_000)
ap.add_argument("--dim", type=int, default=768)
ap.add_argument("--threads", type=int, default=16)
ap.add_argument("--calls", type=int, default=60)
args = ap.parse_args()
lib = pvs.lib()
ix = pvs.VectorIndex(pvs.I8, args.dim, capacity_rows=args.rows)
ix.set_scale(1.0 / 127 * 0.2)
chunk = 1_000_000
stage = pvs.DeviceBuffer(chunk * args.dim * 4)
for off in range(0, args.rows, chunk):
    m = min(chunk, args.rows - off)
    L.check(lib.pvs_synth_rows_f32(0, 3, off, m, args.dim, stage.ptr))
    L.check(lib.pvs_index_add_f32(ix._h, stage.ptr, m, None, None, L.DEVICE))
stage.free()
rng = np.random.default_rng(2)
Q = rng.standard_normal((args.threads * args.calls, args.dim)).astype(np.float32)
Q /= np.linalg.norm(Q, axis=1, keepdims=True)
out = {"rows": args.rows, "dim": args.dim, "threads": args.threads, "calls_per_thread": args.calls, "k": 100}
for label, window in (("direct", 0), ("coalesced_200us", 200), ("coalesced_1000us", 1000)):
    ix.set_coalescing(window, 32)
    lat = []
    def worker(t):
        for r in range(args.calls):
            t0 = time.perf_counter()
            ix.search(Q[t * args.calls + r: t * args.calls + r + 1], 100, pvs.COSINE)
            lat.append(time.perf_counter() - t0)
    for warm in range(2):
        ix.search(Q[:1], 100, pvs.COSINE)
    c0 = ix.coalescing_stats()
    th = [threading.Thread(target=worker, args=(t,)) for t in range(args.threads)]
    t0 = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    dt = time.perf_counter() - t0
    c1 = ix.coalescing_stats()
    lat.sort()
    out[label] = {"queries_per_s": round(args.threads * args.calls / dt, 1), "p50_ms": round(lat[len(lat) // 2] * 1e3, 3),
                  "p99_ms": round(lat[int(len(lat) * 0.99)] * 1e3, 3), "calls": c1[0] - c0[0], "passes": c1[1] - c0[1]}
print(json.dumps(out))
