#!/usr/bin/env python3
"""bench.py — k-NN queries/sec on the BASELINE.json headline workload.

    python bench.py --gpus N --steps K --warmup W [--config 0|1|2|3|4]

Default workload (BASELINE.json `metric`, configs[2]): 10,000,000 x 768 int8-quantized embeddings, batches of 128
queries, cosine, k = 100.  One "step" = one batch of queries answered over the whole corpus (page 1 of the reference
ordering).  Synthetic data: unit-normalised pseudo-Gaussian rows generated in HBM, absmax -> scale
-> quantize_int8 on the device.  Inputs are resident in HBM when the timed region starts.

N > 1, one process per GPU (the form `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` starts;
a bare `python bench.py --gpus N` starts the N ranks itself): the corpus is row-sharded (strong scaling: the corpus
is fixed by the metric), every rank answers the same batch over its shard, the per-shard pages are exchanged by one
grouped RCCL all-gather over xGMI and merged on every rank.  Control plane (rendezvous, barriers, timing reduction):
panoptikon_amd/rendezvous.py, a Unix socket — no torch anywhere.
N > 1, `--single-process`: ONE process drives all N GPUs through a multi-device index (pvs_index_desc.n_devices):
per-shard pages travel to GPU 0 by peer copies and are merged there.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E datasheet (MI355X_MICROARCH.md)
I8_MFMA_PEAK_TOPS = 5000.0  # dense int8 MFMA, datasheet (2x the 2.5 PF bf16 dense peak)
F16_MFMA_PEAK_TFLOPS = 2500.0
NOMINAL_SCLK_MHZ = 2400.0   # the shader clock the datasheet peaks are quoted at
SEED_CORPUS = 20260928
SEED_QUERY = 0x5EED0000

CLUSTERED_QUERY_ROW0 = 1 << 40  # queries of the clustered corpus: rows of the same generator far beyond the corpus
CFG4_ROWS = 25_000_000  # BASELINE configs[4]: 50M rows = a 512-d image-embedding index + a 1024-d text-embedding index (split assumed
                        # even), int8, ~3 vectors per file; the query is the PQL `or` of the two filters fused by RRF
CONFIGS = {  # BASELINE.json configs[i] -> (rows, dim, dtype, batch, k, metric)
    0: (10_000, 512, "f32", 1, 10, "cosine"),  # the reference's own CPU-runnable case (plumbing): the same shape on the device
    1: (1_000_000, 768, "f16", 32, 100, "cosine"),
    2: (10_000_000, 768, "i8", 128, 100, "cosine"),
    3: (100_000_000, 768, "i8", 256, 100, "cosine"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, choices=sorted(CONFIGS) + [4], default=None,
                    help="BASELINE.json configs[i]: sets --rows/--dim/--dtype/--batch/--k/--metric (default: configs[2], the metric's workload)")
    ap.add_argument("--rows", type=int, default=None)
    ap.add_argument("--dim", type=int, default=None)
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--k", type=int, default=None)
    ap.add_argument("--dtype", choices=["i8", "f16", "f32"], default=None)
    ap.add_argument("--metric", choices=["cosine", "l2"], default=None)
    ap.add_argument("--single-process", action="store_true",
                    help="N > 1: one process, a multi-device index over GPUs 0..N-1 (peer-copy gather) instead of N ranks over RCCL")
    ap.add_argument("--devices", default=None,
                    help="--single-process: explicit device list, e.g. 0,0 (two shards on one GPU: exercises the path on a one-GPU box)")
    ap.add_argument("--inflight", type=int, default=0,
                    help="search batches queued ahead of the one being waited for (default: 2 on one GPU, 4 on several)")
    ap.add_argument("--streams", type=int, default=0,
                    help="1: all batches on one HIP stream (default on one GPU); >1: one stream per in-flight batch, collectives "
                         "on their own stream (default on several GPUs, where a shard's scan is too short to fill the machine alone)")
    ap.add_argument("--check-queries", type=int, default=8, help="queries verified against the CPU oracle over the full corpus (every query of the batch is also checked against the device dense path)")
    ap.add_argument("--cpu-sample-rows", type=int, default=400_000)
    ap.add_argument("--cpu-sample-queries", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-peaks", action="store_true", help="skip the on-box stream-copy / MFMA microbenchmarks")
    ap.add_argument("--chunk-rows", type=int, default=1_000_000)
    ap.add_argument("--no-kernel-events", action="store_true", help="do not bracket the kernels with HIP events (roofline fields become 0)")
    ap.add_argument("--force-comm", action="store_true", help="use the RCCL shard-merge path even with one rank (testing)")
    ap.add_argument("--allow-host-gather", action="store_true",
                    help="N ranks: when RCCL cannot be initialised (e.g. two ranks on one GPU) exchange the pages over the control "
                         "socket instead of failing (testing only; the line then says exchange=ctl-host-gather)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary timed regions of the default run (single-query 10M x 768 f16; 256 int8 queries per pass; one query over 690k x 768 int8)")
    ap.add_argument("--debug", action="append", default=[], metavar="KEY=VALUE",
                    help="pvs_debug_set(KEY, VALUE) before the run (tuning sweeps: sample_div, sample_j_div, scan_no_wide128, ...; the line records it)")
    a = ap.parse_args()
    base = CONFIGS[a.config if a.config in CONFIGS else 2]
    for name, val in zip(("rows", "dim", "dtype", "batch", "k", "metric"), base):
        if getattr(a, name) is None:
            setattr(a, name, val)
    return a


def replay_traffic(rows, dim, dtype, batch):
    """HBM bytes per launch of the dominant kernel of a (rows, dim, dtype, batch) region, REPLAYED from the committed PMC passes
    (profiles/traffic_table.json: one record per region, written by tools/make_traffic.py from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
    runs of the same command; the counters cannot be collected inside a timed run).  Returns (bytes or None, provenance or None)."""
    recs = []
    for name in ("traffic_table.json", "traffic_latest.json"):
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            try:
                j = json.load(open(path))
                recs += j if isinstance(j, list) else [j]
            except Exception:  # noqa: BLE001
                pass
    for tj in recs:
        if tj.get("rows") == rows and tj.get("dim") == dim and tj.get("dtype") == dtype and tj.get("batch") == batch:
            return tj.get("hbm_bytes_per_launch"), (f"replayed from profiles/ (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this region's command, "
                                                    f"{tj.get('collected', 'date not recorded')}, kernel {str(tj.get('kernel', ''))[:48]}); not measured in this run")
    return None, None


def summarize_secondary(secondary):
    """One short record per secondary region: t = tag, ms = ms per step (or per call), f = fraction of the region's roofline (b = which),
    ok = every parity verdict of the region true (None: the region carries none).  <= ~70 characters each."""
    out = []
    for r in secondary:
        if "error" in r:
            out.append({"t": "error", "e": str(r["error"])[:60]})
            continue
        rl = r.get("roofline", {})
        par = r.get("parity")
        ok = None if not par else all(v is True for v in par.values() if isinstance(v, bool))
        rec = {"t": r.get("tag", "?"), "ms": r.get("ms_per_step"), "b": rl.get("bound"), "f": rl.get("frac"), "ok": ok}
        if rl.get("mfma", {}).get("frac") and r.get("dtype") == "i8" and r.get("config", {}).get("batch", 0) >= 128:
            rec["mfma"] = rl["mfma"]["frac"]
        if r.get("pvs_search_latency_ms"):
            rec["p50"] = r["pvs_search_latency_ms"]["p50"]
        if rl.get("traffic") and rl.get("algorithmic_bytes_per_launch"):
            rec["tr"] = round(rl["traffic"] / rl["algorithmic_bytes_per_launch"], 3)
        if "fast_frac" in r.get("path", {}):
            rec["fast"] = r["path"]["fast_frac"]
        out.append(rec)
    return out


def replay_projected_scaling(rows, dim, dtype, batch):
    """The shard ladder (tools/shard_ladder.py: this workload at rows/N per GPU for N = 1, 2, 4, 8, each through a 1-rank RCCL
    communicator so that the all-gather + merge span is in the step) as PROJECTED whole-job q/s — replayed from
    profiles/projected_scaling_latest.json, never measured on N GPUs here (the driver's SCALE run is the measurement)."""
    path = os.path.join(ROOT, "profiles", "projected_scaling_latest.json")
    if not os.path.exists(path):
        return None
    try:
        j = json.load(open(path))
    except Exception:  # noqa: BLE001
        return None
    if (j.get("rows"), j.get("dim"), j.get("dtype"), j.get("batch")) != (rows, dim, dtype, batch):
        return None
    # (compact: the driver keeps the last 2,000 characters of the line; the full rungs are profiles/bench_r06_ladder_*.json)
    return {"label": "PROJECTED, not measured on N GPUs: one-GPU runs at rows/N per GPU with a 1-rank RCCL exchange in the step (tools/shard_ladder.py)",
            "collected": j.get("collected"),
            "by_n_gpus": {n: {"ms": r.get("ms_per_step"), "qps": r.get("projected_qps"), "scan_frac": r.get("scan_hbm_frac"), "xchg_ms": r.get("exchange_ms"),
                              "eff": r.get("efficiency")} for n, r in (j.get("by_n_gpus") or {}).items()}}


def which_config(n, d, dtype, b, k, metric):
    for i, c in CONFIGS.items():
        if (n, d, dtype, b, k, metric) == c:
            return f"BASELINE configs[{i}]"
    if (n, d, dtype, b, k) == (10_000, 512, "f32", 1, 10):
        return "BASELINE configs[0] shape, on the GPU"
    return "not a BASELINE config"


def sqlite_udf_baseline(orc, odt, omet, rows, q, k, n_total):
    """`SELECT id, vec_distance_*(embedding, ?) AS d FROM embeddings ORDER BY d LIMIT k` through stdlib SQLite,
    one scalar-function call per row like the reference (db/sql_functions.rs:105-128 registers sqlite-vec's); the function body is
    the CPU oracle's.  One thread = one read connection.  Extrapolated linearly to the corpus."""
    import sqlite3

    conn = sqlite3.connect(":memory:")
    conn.execute("CREATE TABLE embeddings (id INTEGER PRIMARY KEY, embedding BLOB)")
    conn.executemany("INSERT INTO embeddings VALUES (?, ?)", ((i, r.tobytes()) for i, r in enumerate(rows)))
    dt = rows.dtype
    name = "vec_distance_cosine" if omet == orc.COSINE else "vec_distance_L2"

    def udf(a, b):
        v = orc.vec_distance(omet, np.frombuffer(a, dt), np.frombuffer(b, dt))
        return None if v != v else float(v)

    conn.create_function(name, 2, udf, deterministic=True)
    t = time.perf_counter()
    page = conn.execute(f"SELECT id, {name}(embedding, ?) AS d FROM embeddings ORDER BY d LIMIT ?", (q.tobytes(), k)).fetchall()
    dt_s = time.perf_counter() - t
    return {"value": round(1.0 / dt_s * len(rows) / n_total, 5), "unit": "queries/s", "cores": 1, "rows_timed": len(rows),
            "seconds": round(dt_s, 2), "us_per_row": round(dt_s / len(rows) * 1e6, 2), "page_rows": len(page),
            "how": "stdlib sqlite3, per-row scalar UDF = Python trampoline -> C oracle, ORDER BY d LIMIT k; the trampoline costs several us per "
                   "row on top of the arithmetic (the reference's Rust+sqlite-vec path measures ~2-3.3 us/row, docs/vector-quant-measurements.md)"}


def run_config4(args, ctl, rank, world, device, real_stdout):
    """BASELINE configs[4]: mixed 512-d image + 1024-d text embedding indexes (int8), PQL OR-composition of the two filters ranked
    by reciprocal-rank fusion (pql/builder.rs:638-661, 757-771, 1284-1301).  One step = one composed query answered over both
    corpora: every row scored exactly, MIN per file, the bounded fusion (pvs_rrf_search; rrf_search_sharded for N ranks, rows sharded
    BY FILE)."""
    import panoptikon_amd as pvs
    from panoptikon_amd import _lib as L

    lib = pvs.lib()
    N = args.rows if args.rows not in (None, CONFIGS[2][0]) else CFG4_ROWS
    K = args.k
    per_file = 3
    files = (N + per_file - 1) // per_file
    f0, f1 = pvs.shard_range(files, world, rank)          # shard BY FILE: every vector of a file on one rank
    r0, r1 = f0 * per_file, min(f1 * per_file, N)
    n_local = r1 - r0
    specs = [(512, pvs.COSINE, 11, 5, 1.0, 1), (1024, pvs.L2, 12, 10, 0.7, 2)]  # dim, metric, seed, rrf k, weight, file-id stride
    branches, scales = [], []
    t_build = time.time()
    for dim, metric, seed, rk, wt, gstride in specs:
        chunk = min(args.chunk_rows, max(n_local, 1))
        stage = pvs.DeviceBuffer(chunk * dim * 4, device)
        amax = 0.0
        for off in range(0, n_local, chunk):
            m = min(chunk, n_local - off)
            L.check(lib.pvs_synth_rows_f32(device, seed, r0 + off, m, dim, stage.ptr))
            out = L.C.c_float()
            L.check(lib.pvs_absmax(stage.ptr, m * dim, L.DEVICE, device, L.C.byref(out)))
            amax = max(amax, float(out.value))
        scale = pvs.scale_from_absmax(ctl.max_float(amax))
        ix = pvs.VectorIndex(pvs.I8, dim, device=device, capacity_rows=n_local, id_base=r0)
        ix.set_scale(scale)
        for off in range(0, n_local, chunk):
            m = min(chunk, n_local - off)
            L.check(lib.pvs_synth_rows_f32(device, seed, r0 + off, m, dim, stage.ptr))
            g = (np.arange(r0 + off, r0 + off + m, dtype=np.int64) // per_file) * gstride  # text files: every other id -> partial overlap
            L.check(lib.pvs_index_add_f32(ix._h, stage.ptr, m, None, g.ctypes.data, L.DEVICE))
        stage.free()
        branches.append(dict(index=ix, metric=metric, agg=pvs.AGG_MIN, rrf_k=rk, weight=wt, dim=dim, seed=seed))
        scales.append(scale)
    L.check(lib.pvs_device_synchronize(device))
    if rank == 0:
        print(f"[bench] configs[4]: 2 x {N} rows ({files} files x2 branches), this rank holds rows [{r0}, {r1}) of each, built in "
              f"{time.time() - t_build:.1f}s", file=sys.stderr, flush=True)
    comm4 = make_comm(lib, L, ctl, world, rank, device, args.allow_host_gather, lambda m: print("[bench] " + m, file=sys.stderr, flush=True)) if world > 1 else None
    NQ = 8
    queries = []  # [query][branch] f32 vectors from the device generator (pvs_synth_rows_f32), as every other config's
    for i in range(NQ):
        per_branch = []
        for bi, b in enumerate(branches):
            qb = pvs.DeviceBuffer(b["dim"] * 4, device)
            L.check(lib.pvs_synth_rows_f32(device, SEED_QUERY + 17 * bi, i, 1, b["dim"], qb.ptr))
            per_branch.append(qb.to_numpy(np.float32, (b["dim"],)).copy())
            qb.free()
        queries.append(per_branch)

    def step(i):
        brs = [dict(index=b["index"], query=queries[i % NQ][j], metric=b["metric"], agg=b["agg"], rrf_k=b["rrf_k"], weight=b["weight"])
               for j, b in enumerate(branches)]
        if world == 1:
            return pvs.rrf_search(brs, K)
        if comm4 is not None:
            return pvs.rrf_search_sharded(brs, K, comm=comm4)  # pvs_rrf_search_sharded over RCCL: the round loop and its exchange inside libpvs
        return pvs.rrf_search_sharded(brs, K, gather=ctl)

    for i in range(args.warmup):
        step(i)
    L.check(lib.pvs_device_synchronize(device))
    ctl.barrier()
    for b in branches:
        b["index"].set_profiling(not args.no_kernel_events)
        b["index"].profile(reset=True)
    t0 = time.perf_counter()
    for i in range(args.steps):
        last = step(args.warmup + i)
    L.check(lib.pvs_device_synchronize(device))
    ctl.barrier()
    elapsed = ctl.max_float(time.perf_counter() - t0)
    for b in branches:
        b["index"].set_profiling(False)
    # Kernel durations: in the timed region the two branches are scored from two host threads and their kernels overlap — each one's
    # event bracket then includes the share of HBM the other takes.  Exclusive durations come from a serialized second pass over the
    # same queries (pvs_debug_set("rrf_serial", 1)), like the several-streams case of the row search.
    n2 = min(args.steps, 10)
    if not args.no_kernel_events and world == 1:
        pvs.debug_set("rrf_serial", 1)
        for b in branches:
            b["index"].set_profiling(True)
            b["index"].profile(reset=True)
        for i in range(n2):
            step(args.warmup + i)
        L.check(lib.pvs_device_synchronize(device))
        pvs.debug_set("rrf_serial", 0)
    profs = [b["index"].profile() for b in branches]
    for b in branches:
        b["index"].set_profiling(False)
    scan_ms = sum(p.scan_ms for p in profs) / max(n2 if (not args.no_kernel_events and world == 1) else args.steps, 1)  # per composed query, both branches
    bytes_per_query = sum(n_local * b["dim"] for b in branches)                        # every stored code read once per query
    achieved = bytes_per_query / (scan_ms * 1e-3) / 1e9 if scan_ms else 0.0
    result = {
        "metric": "knn_queries_per_sec", "value": round(args.steps / elapsed, 2), "unit": "queries/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "i8", "data": "synthetic",
        "config": {"workload": f"{N}x512 + {N}x1024 i8 indexes (~{per_file} vectors per file), PQL or-composition of an image (cosine) and a "
                               f"text (L2) filter, MIN per file, row_n + RRF (5/1.0, 10/0.7), k={K} (BASELINE configs[4])",
                   "rows": 2 * N, "batch": 1, "k": K, "parallelism": f"shard by file x{world}", "exchange": "single-gpu" if world == 1 else
                   ("pvs_rrf_search_sharded over RCCL (ncclAllGather of the padded pages, ncclAllReduce min / sum of thresholds and counts)" if comm4 is not None
                    else "pvs_rrf_search_sharded with the control socket as its all-gather (--allow-host-gather)")},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                     "traffic": None, "kernel": "k_score_i8_direct (exact int8 distances of every row for one query: v_dot4 straight from HBM)", "launches": int(sum(p.scan_launches for p in profs)),
                     "avg_launch_ms": round(sum(p.scan_ms for p in profs) / max(sum(p.scan_launches for p in profs), 1), 4),
                     "algorithmic_bytes_per_query": int(bytes_per_query), "scoring_ms_per_query": round(scan_ms, 3),
                     "whole_query_frac_of_peak": round(bytes_per_query / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
                     "kernel_events": ("off" if args.no_kernel_events else "serialized second pass over the same queries (in the timed region the two branches' scoring kernels "
                                       "overlap on two streams: their event brackets there are shares, not exclusive durations)" if world == 1 else "timed region")},
        "path": {"rrf_path": {1: "bounded fusion", 2: "every group ranked"}.get(int(lib.pvs_rrf_last_path()), "sharded bounded fusion")},
    }
    if not args.no_verify:
        # oracle, literally: every row of this rank scored on the CPU, MIN per file; then (N = 1) the whole composition, or (N > 1) the
        # exact global window ranks of the returned files by counting, summed over ranks
        import oracle as orc  # (the checker: verification leg only)

        t_or = time.time()
        gq = queries[(args.warmup + args.steps - 1) % NQ]
        cols = []
        for j, b in enumerate(branches):
            ix = b["index"]
            qh = orc.quantize_int8(gq[j][None, :], scales[j])[0]
            vals = np.empty(0, np.float64)
            gids = np.empty(0, np.int64)
            om = orc.COSINE if b["metric"] == pvs.COSINE else orc.L2
            dcol = np.empty(n_local, np.float32)
            for off in range(0, n_local, args.chunk_rows):
                m = min(args.chunk_rows, n_local - off)
                dcol[off: off + m] = orc.score_all(orc.I8, om, ix.read_rows(off, m), qh, threads=max(1, orc.max_threads() // world))
            grp = (np.arange(r0, r1, dtype=np.int64) // per_file) * specs[j][5]
            gids, vals = orc.aggregate(dcol, grp, orc.AGG_MIN)
            cols.append((gids, vals))
        got_g, got_s = last
        ks, ws = [b["rrf_k"] for b in branches], [b["weight"] for b in branches]
        if world == 1:
            allg = np.union1d(cols[0][0], cols[1][0])
            ranks = np.full((2, len(allg)), -1, np.int64)
            for j, (g, v) in enumerate(cols):
                ranks[j, np.searchsorted(allg, g)] = orc.row_number(v, g)
            score = pvs.rrf_fuse(ranks, ks, ws)  # (host arithmetic of the C ABI = orc.rrf_score, checked in tests/)
            order = np.lexsort((allg, -score))[:K]
            exact = bool(np.array_equal(got_g, allg[order]) and np.array_equal(got_s.view(np.uint64), score[order].view(np.uint64)))
            how = "oracle: every row scored on the CPU, MIN per file, row_number over ALL files of each branch, UNION, RRF, ORDER BY score DESC LIMIT k"
        else:
            ranks = np.full((2, len(got_g)), -1, np.int64)
            for j, (g, v) in enumerate(cols):
                pos = np.searchsorted(g, got_g)
                mine = (pos < len(g)) & (g[np.minimum(pos, len(g) - 1)] == got_g)
                val = np.where(mine, v[np.minimum(pos, len(v) - 1)], np.nan)
                allv, allm = ctl.all_gather_np(val), ctl.all_gather_np(mine.astype(np.uint8))
                have = allm.any(axis=0)
                vv = np.where(have, allv[np.argmax(allm, axis=0), np.arange(len(got_g))], np.nan)
                # window order ascending: NULL first, then value, then file id
                below = np.zeros(len(got_g), np.int64)
                vn = np.isnan(v)
                sv = np.sort(v[~vn])
                for i, (x, gid) in enumerate(zip(vv, got_g)):
                    if not have[i]:
                        continue
                    if np.isnan(x):
                        below[i] = int(np.sum(vn & (g < gid)))
                    else:
                        below[i] = int(vn.sum()) + int(np.searchsorted(sv, x, "left")) + int(np.sum((v == x) & (g < gid)))
                tot = ctl.all_gather_np(below).sum(axis=0)
                ranks[j, have] = tot[have] + 1
            score = pvs.rrf_fuse(ranks, ks, ws)
            sorted_ok = bool(np.all((score[:-1] > score[1:]) | ((score[:-1] == score[1:]) & (got_g[:-1] < got_g[1:]))))
            exact = bool(np.array_equal(got_s.view(np.uint64), score.view(np.uint64)) and sorted_ok)
            how = ("per rank: oracle distances of its rows, MIN per file; exact global window ranks of the returned files by counting, summed over "
                   "ranks; their RRF scores and order (completeness of the page rests on the fusion's bound, proven against the full oracle at N = 1)")
        if rank == 0:
            result["parity"] = {"checked_queries": 1, "groups_and_scores_bit_exact": exact, "how": how, "oracle_seconds": round(time.time() - t_or, 1)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # the oracle's literal composition on one host core over a bounded sample of both corpora (the same files in both
        # branches), extrapolated linearly in the rows: score every row, MIN per file, row_number per branch, RRF, top k
        import oracle as orc  # (the checker: the cpu_baseline leg only)

        S = min(args.cpu_sample_rows, n_local)
        gq = queries[0]
        ora = []
        for j, b in enumerate(branches):
            ora.append(dict(dtype=orc.I8, metric=orc.COSINE if b["metric"] == pvs.COSINE else orc.L2, corpus=b["index"].read_rows(0, S),
                            query=orc.quantize_int8(gq[j][None, :], scales[j])[0], groups=(np.arange(r0, r0 + S, dtype=np.int64) // per_file) * specs[j][5],
                            agg=orc.AGG_MIN, rrf_k=b["rrf_k"], weight=b["weight"]))
        t1 = time.perf_counter()
        orc.rrf_search(ora, K)
        dt1 = time.perf_counter() - t1
        result["cpu_baseline"] = {"value": round(1.0 / dt1 * S / N, 5), "unit": "queries/s", "cores": 1, "kind": "port",
                                  "sample": f"oracle composition (scores, MIN per file, row_number, RRF, top-{K}) over the first {S} rows of both corpora, one query "
                                            f"({dt1:.1f}s measured), extrapolated linearly to {N} rows per branch"}
    if rank == 0:
        real_stdout.write(json.dumps(result) + "\n")
        real_stdout.flush()
    ctl.barrier()
    for b in branches:
        b["index"].close()
    ctl.close()


def sample_clock_and_power(step, drain, seconds=1.5):
    """rocm-smi's shader clock and socket power while `step` loops (a side thread polls rocm-smi; the loop runs on this one)."""
    import re
    import shutil
    import threading

    exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    if not os.path.exists(exe):
        return {"error": "rocm-smi not found"}
    samples, stop = [], threading.Event()

    def poll():
        while not stop.is_set():
            try:
                out = subprocess.run([exe, "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
            except Exception:  # noqa: BLE001
                return
            clk = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", out)
            pw = re.search(r"Power \(W\): ([0-9.]+)", out)
            if clk and pw:
                samples.append((int(clk.group(1)), float(pw.group(1))))

    th = threading.Thread(target=poll, daemon=True)
    t_end = time.perf_counter() + seconds
    i = 0
    th.start()
    while time.perf_counter() < t_end or len(samples) < 2:
        step(i)
        i += 1
        if time.perf_counter() > t_end + 8.0:
            break
    stop.set()
    drain()
    th.join(timeout=15)
    if not samples:
        return {"error": "no rocm-smi sample"}
    samples = samples[1:] or samples  # (the first one may predate the loop)
    clk = sorted(c for c, _ in samples)[len(samples) // 2]
    pw = sorted(p for _, p in samples)[len(samples) // 2]
    return {"sclk_mhz": clk, "socket_power_w": pw, "samples": len(samples), "steps_looped": i,
            "how": "median of rocm-smi --showclocks --showpower polled while the same steps loop, after the timed region"}


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks (one process per GPU) ourselves, relay rank 0's
    JSON line, fail if any rank fails."""
    sock = os.path.join(tempfile.gettempdir(), f"pvs_ctl_{os.getpid()}.sock")
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), PVS_CTL_SOCK=sock, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0 = procs[0].communicate()[0].decode()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out0)
    sys.stdout.flush()
    if any(rcs):
        print(f"[bench] rank exit codes {rcs}", file=sys.stderr)
        return 1
    return 0


def make_comm(lib, L, ctl, world, rank, device, allow_host_gather, log):
    """The RCCL communicator of this run, or None (with --allow-host-gather: the control socket stands in).  Every rank walks the
    same sequence of control-plane collectives whatever fails locally, and the ranks agree on whether the in-library RCCL path is
    usable: a rank that gave up alone would leave the others waiting in an all-gather."""
    idb, ok, err = bytes(L.UNIQUE_ID_BYTES), 1.0, ""
    if rank == 0:
        try:
            buf = (L.C.c_uint8 * L.UNIQUE_ID_BYTES)()
            L.check(lib.pvs_comm_unique_id(buf))
            idb = bytes(buf)
        except Exception as e:  # noqa: BLE001
            ok, err = 0.0, str(e)
    idb = ctl.bcast_bytes(idb)
    ok = ctl.min_float(ok)
    h = None
    if ok > 0:
        try:
            h = L.C.c_void_p()
            idarr = (L.C.c_uint8 * L.UNIQUE_ID_BYTES).from_buffer_copy(idb)
            L.check(lib.pvs_comm_create(idarr, world, rank, device, L.C.byref(h)))
        except Exception as e:  # noqa: BLE001
            ok, err, h = 0.0, str(e), None
        ok = ctl.min_float(ok)
    if ok > 0:
        return h
    if h is not None:
        lib.pvs_comm_destroy(h)
    errs = [e.decode() for e in ctl.allgather_bytes(err.encode())]
    if not allow_host_gather:
        raise SystemExit(f"RCCL communicator could not be created on every rank: {[e for e in errs if e]}")
    log(f"in-library RCCL unavailable ({[e for e in errs if e]}); --allow-host-gather: pages go over the control socket")
    return None


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.single_process:
        raise SystemExit(launch_ranks(args.gpus))
    # stdout must carry exactly ONE JSON line, but RCCL prints banners on fd 1 when it initialises: keep a
    # private handle on the real stdout and point fd 1 at stderr for the run.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    single = args.single_process and args.gpus > 1
    if single and world > 1:
        raise SystemExit("--single-process drives every GPU from one process: do not start it under a multi-rank launcher")
    if not single and world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; refusing to report a "
                         f"line for a different GPU count than asked for")

    from panoptikon_amd.rendezvous import LocalRendezvous

    ctl = LocalRendezvous(rank, world)
    if not os.path.exists(os.path.join(ROOT, "panoptikon_amd", "libpvs.so")):  # fresh checkout: the library is git-ignored
        if rank == 0:
            subprocess.check_call([sys.executable, "-m", "panoptikon_amd.build"], cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        ctl.barrier()
    import panoptikon_amd as pvs
    from panoptikon_amd import _lib as L

    n_dev = pvs.device_count()
    if n_dev < 1:
        raise SystemExit("bench.py needs an MI355X (gfx950); libpvs has no CPU path")
    for kv in args.debug:
        key, _, val = kv.partition("=")
        pvs.debug_set(key, int(val or 1))
    if args.config == 4:
        return run_config4(args, ctl, rank, world, local_rank % n_dev if world > 1 else 0, real_stdout)
    dev_list = [int(x) for x in args.devices.split(",")] if args.devices else list(range(args.gpus))
    if single and (len(dev_list) != args.gpus or max(dev_list) >= n_dev):
        raise SystemExit(f"--single-process --gpus {args.gpus}: device list {dev_list} does not fit the {n_dev} visible device(s)")
    if world > n_dev and not args.allow_host_gather:
        raise SystemExit(f"{world} ranks but {n_dev} visible device(s): RCCL needs one GPU per rank (--allow-host-gather to test the "
                         f"control flow on fewer)")
    device = (local_rank % n_dev) if world > 1 else 0
    devices = dev_list if single else None
    n_gpus = args.gpus
    dtype = {"i8": pvs.I8, "f16": pvs.F16, "f32": pvs.F32}[args.dtype]
    metric = pvs.COSINE if args.metric == "cosine" else pvs.L2
    esz = {pvs.I8: 1, pvs.F16: 2, pvs.F32: 4}[dtype]
    N, D, B, K = args.rows, args.dim, args.batch, args.k
    r0, r1 = pvs.shard_range(N, world, rank)
    n_local = r1 - r0                      # rows this process holds
    n_per_gpu = n_local // (args.gpus if single else 1)
    lib = pvs.lib()

    def log(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    def device_sync():
        for d in (devices or [device]):
            L.check(lib.pvs_device_synchronize(d))

    # ------------------------------------------------------------ build shard
    t_build = time.time()
    chunk = min(args.chunk_rows, max(n_local, 1))
    stage = pvs.DeviceBuffer(chunk * D * 4, device)
    scale = None
    if dtype == pvs.I8:
        amax = 0.0
        for off in range(0, n_local, chunk):
            m = min(chunk, n_local - off)
            L.check(lib.pvs_synth_rows_f32(device, SEED_CORPUS, r0 + off, m, D, stage.ptr))
            out = L.C.c_float()
            L.check(lib.pvs_absmax(stage.ptr, m * D, L.DEVICE, device, L.C.byref(out)))
            amax = max(amax, float(out.value))
        local_amax = amax
        amax = ctl.max_float(amax)  # one scale per embedding space, global over all shards (the RCCL communicator does not exist
        scale = pvs.scale_from_absmax(amax)  # yet: it is created after the build; pvs_comm_allreduce_max_f32 cross-checks below)
    ix = pvs.VectorIndex(dtype, D, device=device, capacity_rows=n_local, id_base=r0, devices=devices)
    if scale is not None:
        ix.set_scale(scale)
    for off in range(0, n_local, chunk):
        m = min(chunk, n_local - off)
        L.check(lib.pvs_synth_rows_f32(device, SEED_CORPUS, r0 + off, m, D, stage.ptr))
        ix.add_f32((stage, m))
    stage.free()
    ix.sync()
    device_sync()
    log(f"rows [{r0}, {r1}) resident in HBM as {args.dtype} in {time.time() - t_build:.1f}s (scale={scale}"
        + (f", sharded over devices {devices}" if single else "") + ")")

    multi = world > 1 or args.force_comm or single
    n_streams = args.streams or (2 if multi else 1)
    n_inflight = args.inflight or (4 if multi else 2)
    if n_streams > 1:
        ix.set_streams(n_streams)

    # ---------------------------------------------------------------- queries
    NQB = 4
    qbufs = []
    for i in range(NQB):
        qb = pvs.DeviceBuffer(B * D * 4, device)
        L.check(lib.pvs_synth_rows_f32(device, SEED_QUERY, i * B, B, D, qb.ptr))
        qbufs.append(qb)
    slots = max(1, min(n_inflight, 4))
    outs = [(pvs.DeviceBuffer(B * K * 8, device), pvs.DeviceBuffer(B * K * 4, device), pvs.DeviceBuffer(B * 4, device))
            for _ in range(slots)]

    comm = None
    gather_mode = "peer-copy (one process, multi-device index)" if single else "single-gpu"
    if world > 1 or args.force_comm:
        comm = make_comm(lib, L, ctl, world, rank, device, args.allow_host_gather, log)
        gather_mode = "rccl-allgather" if comm is not None else "ctl-host-gather"
        if comm is not None and args.dtype == "i8":  # the space's absmax once more, by ncclAllReduce(max) over xGMI: must agree bit for bit
            v = L.C.c_float(local_amax)
            L.check(lib.pvs_comm_allreduce_max_f32(comm, L.C.byref(v)))
            if np.float32(v.value) != np.float32(amax):
                raise SystemExit(f"rank {rank}: ncclAllReduce(max) of the shard absmax gave {v.value}, the control plane {amax}")

    pending = []

    def step_sharded(i):
        q = qbufs[i % NQB]
        if comm is not None:  # stream-ordered: search -> RCCL all-gather -> merge, `slots` batches in flight
            oi, od, oc = outs[i % slots]
            if len(pending) >= slots:
                ix.wait(pending.pop(0))
            t = L.C.c_uint32()
            L.check(lib.pvs_search_sharded_async(ix._h, comm, q.ptr, L.F32, B, K, metric, oi.ptr, od.ptr, oc.ptr, L.C.byref(t)))
            pending.append(int(t.value))
        else:
            oi, od, oc = outs[0]
            t = ix.search_device(q, L.F32, B, K, metric, oi, od, oc)
            ix.wait(t)
            return pvs.merge_shard_pages(oi.to_numpy(np.int64, (B, K)), od.to_numpy(np.float32, (B, K)),
                                         oc.to_numpy(np.uint32, (B,)), ctl, K)
        return None

    def step_single(i):  # one index object (one GPU, or all of them behind a multi-device index)
        q = qbufs[i % NQB]
        oi, od, oc = outs[i % slots]
        if len(pending) >= slots:
            ix.wait(pending.pop(0))
        pending.append(ix.search_device(q, L.F32, B, K, metric, oi, od, oc))

    def drain():
        while pending:
            ix.wait(pending.pop(0))

    step = step_single if (world == 1 and not args.force_comm) else step_sharded

    # ----------------------------------------------------------------- timing
    # Every in-flight slot owns a search context whose buffers (and, on a multi-device index, every shard's) are allocated by its first
    # search: with fewer warm-up steps than slots the timed region pays those allocations (round 5: `--single-process --devices 0,0`
    # at --steps 5 --warmup 2 reported 4.76 ms/step where the steady state is 0.39 — profiles/r06_single_process_0_0_timeline.md).
    # The untimed priming steps below touch every slot once; the W warm-up steps follow as asked.
    priming = max(0, (slots if (world == 1 or comm is not None) else 1) + 1 - args.warmup)
    for i in range(priming):
        step(i)
    drain()
    for i in range(args.warmup):
        step(i)
    drain()
    device_sync()
    ctl.barrier()
    # Kernel durations come from HIP events around each launch on the launch stream.  When searches
    # overlap on several streams a kernel's wall time includes the kernels it shares the GPU with, so
    # in that mode the events are taken in a second, serialized pass over the same steps (below).
    overlapped = n_streams > 1
    ix.set_profiling(not args.no_kernel_events and not overlapped)
    ix.profile(reset=True)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    drain()
    device_sync()
    ctl.barrier()
    elapsed = ctl.max_float(time.perf_counter() - t0)
    ix.set_profiling(False)
    prof = ix.profile()
    if overlapped and not args.no_kernel_events:
        ix.set_streams(1)
        ix.set_profiling(True)
        ix.profile(reset=True)
        for i in range(args.steps):
            step(args.warmup + i)
            drain()
        device_sync()
        ctl.barrier()
        ix.set_profiling(False)
        prof = ix.profile()
        ix.set_streams(n_streams)
    st = ix.stats()
    qps = args.steps * B / elapsed

    # ---------------------------------------------------------------- roofline
    scan_ms = prof.scan_ms / max(prof.scan_launches, 1)
    bytes_per_launch = n_per_gpu * D * esz  # algorithmic bytes: each corpus component read once per batch (per GPU)
    achieved_gbs = bytes_per_launch / (scan_ms * 1e-3) / 1e9 if prof.scan_launches else 0.0
    ops_per_launch = 2.0 * n_per_gpu * D * B
    mfma_peak = I8_MFMA_PEAK_TOPS if dtype == pvs.I8 else F16_MFMA_PEAK_TFLOPS  # f32 rows run as f16 on the matrix core
    traffic, traffic_source = replay_traffic(n_per_gpu, D, args.dtype, B)
    roofline = {
        "bound": "hbm", "achieved": round(achieved_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved_gbs / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
        "kernel": (lambda nm: nm if nm.startswith("k_direct_topk") else nm + " (pass B, filter scan)")(ix.scan_kernel_name(B)), "launches": int(prof.scan_launches),
        "avg_launch_ms": round(scan_ms, 4), "algorithmic_bytes_per_launch": int(bytes_per_launch),
        "mfma": {"achieved": round(ops_per_launch / (scan_ms * 1e-3) / 1e12, 1) if prof.scan_launches else 0.0,
                 "peak": mfma_peak, "unit": "TOP/s" if dtype == pvs.I8 else "TFLOP/s",
                 "frac": round(ops_per_launch / (scan_ms * 1e-3) / 1e12 / mfma_peak, 4) if prof.scan_launches else 0.0},
        "kernel_events": "serialized second pass over the same steps (the timed region overlaps searches on several streams)"
                         if overlapped else "timed region",
        "sample_pass_avg_ms": round(prof.sample_ms / max(prof.sample_launches, 1), 4),
        "finalize_avg_ms": round(prof.finalize_ms / max(prof.finalize_launches, 1), 4),
    }
    if prof.exchange_launches:
        roofline["exchange_avg_ms"] = round(prof.exchange_ms / prof.exchange_launches, 4)  # grouped all-gather + merge kernel
    if not args.no_peaks and rank == 0 and world == 1 and not args.force_comm:
        # What clock and power does the part hold under this workload?  (The 128- and 256-query passes run at the board power
        # limit: DESIGN.md section 5.)  Untimed: the same steps loop for ~1.5 s while rocm-smi is sampled from a side thread.
        try:
            roofline["under_load"] = sample_clock_and_power(lambda i: step(i), drain, seconds=1.5)
        except Exception as e:  # noqa: BLE001
            roofline["under_load"] = {"error": str(e)}
        ul = roofline["under_load"]
        if ul.get("sclk_mhz") and prof.scan_launches:
            roofline["power"] = {"sclk_mhz": ul["sclk_mhz"], "socket_power_w": ul["socket_power_w"], "nominal_sclk_mhz": NOMINAL_SCLK_MHZ,
                                 "mfma_frac_of_clock_scaled_peak": round(roofline["mfma"]["frac"] * NOMINAL_SCLK_MHZ / ul["sclk_mhz"], 4)}
    if not args.no_peaks and rank == 0 and hasattr(lib, "pvs_microbench"):
        try:
            roofline["measured_peaks"] = pvs.microbench(device)
        except Exception as e:  # noqa: BLE001
            roofline["measured_peaks"] = {"error": str(e)}

    result = {
        "metric": "knn_queries_per_sec", "value": round(qps, 1), "unit": "queries/s", "n_gpus": n_gpus,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"{N}x{D} {args.dtype} corpus, batch {B}, {args.metric}, k={K} ({which_config(N, D, args.dtype, B, K, args.metric)})",
                   "rows": N, "dim": D, "batch": B, "k": K, "metric": args.metric,
                   "parallelism": f"row-shard x{n_gpus}" + (" (one process)" if single else f" ({world} rank{'s' if world > 1 else ''})"),
                   "exchange": gather_mode, "priming_steps_before_warmup": priming, "streams": n_streams, "inflight": slots if (world == 1 or comm is not None) else 1,
                   **({"debug_knobs": args.debug} if args.debug else {})},
        "roofline": roofline,
        "path": {"fast_queries": int(st.fast_queries), "dense_queries": int(st.dense_queries),
                 "scan_candidates_per_query": round(int(st.last_candidates) / max(B, 1), 1)},
    }

    # ------------------------------------------------- the north star's other two shapes (own timed regions, headline fields untouched)
    def timed_region(ixh, dt_name, b, steps, warmup, K=K, load_sample=False, clustered=False):
        """`steps` batches of b queries through pvs_search_device on index ixh (one stream), kernel durations from HIP events in the
        timed region: the same measurement as the headline's, as one self-contained record."""
        esz2 = {"i8": 1, "f16": 2, "f32": 4}[dt_name]
        qb = pvs.DeviceBuffer(b * D * 4, device)
        if clustered:  # queries of a clustered corpus: the same generator and seed at rows beyond the corpus (they fall into its clusters)
            L.check(lib.pvs_synth_rows_clustered_f32(device, SEED_CORPUS, CLUSTERED_QUERY_ROW0, b, D, qb.ptr))
        else:
            L.check(lib.pvs_synth_rows_f32(device, SEED_QUERY, 0, b, D, qb.ptr))
        st0 = ixh.stats()
        o = [(pvs.DeviceBuffer(b * K * 8, device), pvs.DeviceBuffer(b * K * 4, device), pvs.DeviceBuffer(b * 4, device)) for _ in range(2)]
        pend = []

        def one(i):
            if len(pend) >= 2:
                ixh.wait(pend.pop(0))
            pend.append(ixh.search_device(qb, L.F32, b, K, metric, *o[i % 2]))

        for i in range(warmup):
            one(i)
        while pend:
            ixh.wait(pend.pop(0))
        L.check(lib.pvs_device_synchronize(device))
        ixh.set_profiling(True)
        ixh.profile(reset=True)
        t = time.perf_counter()
        for i in range(steps):
            one(i)
        while pend:
            ixh.wait(pend.pop(0))
        L.check(lib.pvs_device_synchronize(device))
        el = time.perf_counter() - t
        ixh.set_profiling(False)
        p = ixh.profile()
        st2 = ixh.stats()
        sms = p.scan_ms / max(p.scan_launches, 1)
        nrows = int(st2.rows)
        by = nrows * D * esz2
        gbs = by / (sms * 1e-3) / 1e9 if p.scan_launches else 0.0
        ops = 2.0 * nrows * D * b
        pk = I8_MFMA_PEAK_TOPS if dt_name == "i8" else F16_MFMA_PEAK_TFLOPS
        under = None
        if load_sample and not args.no_peaks:
            # the clock and the socket power the part holds under THIS region's loop (the 256-query pass sits at the board power
            # limit): untimed, the same searches loop for ~1.5 s while rocm-smi is polled from a side thread
            def drain2():
                while pend:
                    ixh.wait(pend.pop(0))
            try:
                under = sample_clock_and_power(one, drain2, seconds=1.5)
            except Exception as e:  # noqa: BLE001
                under = {"error": str(e)}
        for bufs in o:
            for x in bufs:
                x.free()
        qb.free()
        power = None
        if under and under.get("sclk_mhz"):
            # achieved / (peak x sclk / 2400 MHz): how much of what the matrix cores (or, for an HBM-bound kernel, nothing: HBM does
            # not follow sclk) could do AT THE CLOCK THE BOARD HOLDS under this load
            power = {"sclk_mhz": under["sclk_mhz"], "socket_power_w": under["socket_power_w"], "nominal_sclk_mhz": NOMINAL_SCLK_MHZ,
                     "mfma_frac_of_clock_scaled_peak": round(ops / (sms * 1e-3) / 1e12 / (pk * under["sclk_mhz"] / NOMINAL_SCLK_MHZ), 4) if p.scan_launches else 0.0}
        tr2, tr2_src = replay_traffic(nrows, D, dt_name, b)
        return {"config": {"workload": f"{nrows}x{D} {dt_name} corpus, batch {b}, {args.metric}, k={K}", "rows": nrows, "dim": D, "batch": b, "k": K},
                "metric": "knn_queries_per_sec", "value": round(steps * b / el, 1), "unit": "queries/s", "steps": steps, "warmup": warmup,
                "ms_per_step": round(el / steps * 1e3, 4), "dtype": dt_name, "data": "synthetic",
                "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": tr2,
                             **({"traffic_source": tr2_src} if tr2_src else {}),
                             "kernel": (lambda nm: nm if nm.startswith("k_direct_topk") else nm + " (pass B, filter scan)")(ixh.scan_kernel_name(b)), "launches": int(p.scan_launches), "avg_launch_ms": round(sms, 4),
                             "algorithmic_bytes_per_launch": int(by), "kernel_events": "timed region",
                             "mfma": {"achieved": round(ops / (sms * 1e-3) / 1e12, 1) if p.scan_launches else 0.0, "peak": pk,
                                      "unit": "TOP/s" if dt_name == "i8" else "TFLOP/s", "frac": round(ops / (sms * 1e-3) / 1e12 / pk, 4) if p.scan_launches else 0.0},
                             **({"under_load": under} if under else {}), **({"power": power} if power else {})},
                "path": {"fast_queries": int(st2.fast_queries), "dense_queries": int(st2.dense_queries),
                         # of the queries of this region (warm-up included): answered by the filter scan / the one-launch search, handed to the
                         # dense path, sent through the scan twice (segment overflow); candidates the last filter scan emitted per query
                         "fast_frac": round((int(st2.fast_queries) - int(st0.fast_queries)) / max(1, (int(st2.fast_queries) - int(st0.fast_queries)) + (int(st2.dense_queries) - int(st0.dense_queries))), 4),
                         "rescanned_queries": int(st2.rescanned_queries) - int(st0.rescanned_queries),
                         "scan_candidates_per_query": round(int(st2.last_candidates) / max(b, 1), 1)}}

    secondary = []
    want_secondary = (rank == 0 and world == 1 and not single and not args.force_comm and not args.no_secondary and n_streams == 1
                      and (N, D, args.dtype, B, K, args.metric) == CONFIGS[2])
    if want_secondary:
        try:
            # (b) 256 int8 queries per pass over the same corpus: the configuration the int8 MFMA target is reachable on
            rec = timed_region(ix, "i8", 256, 20, 3, load_sample=True)
            rec["tag"] = "i8x256"
            rec["what"] = "north-star MFMA target shape: 256 int8 queries per corpus pass (k_scan_wide, one workgroup per CU)"
            secondary.append(rec)
            # (a) single query over 10M x 768 f16: the north star's >= 70 % of HBM target
            free_b, tot_b = L.C.c_uint64(), L.C.c_uint64()
            L.check(lib.pvs_device_mem_info(device, L.C.byref(free_b), L.C.byref(tot_b)))
            if free_b.value > 40 << 30:
                t_b = time.time()
                ix16 = pvs.VectorIndex(pvs.F16, D, device=device, capacity_rows=N)
                st16 = pvs.DeviceBuffer(chunk * D * 4, device)
                for off in range(0, N, chunk):
                    m = min(chunk, N - off)
                    L.check(lib.pvs_synth_rows_f32(device, SEED_CORPUS, off, m, D, st16.ptr))
                    ix16.add_f32((st16, m))
                st16.free()
                ix16.sync()
                rec = timed_region(ix16, "f16", 1, 30, 3, load_sample=True)
                rec["tag"] = "f16x1"
                rec["what"] = "north-star HBM target shape: single query over 10M x 768 f16 (>= 70 % of the HBM roofline asked)"
                rec["build_seconds"] = round(time.time() - t_b, 1)
                if not args.no_verify:  # the page of the timed query against the device's dense path (the reference's algorithm in HBM)
                    qf = np.empty((1, D), np.float32)
                    qtmp = pvs.DeviceBuffer(D * 4, device)
                    L.check(lib.pvs_synth_rows_f32(device, SEED_QUERY, 0, 1, D, qtmp.ptr))
                    qf[:] = qtmp.to_numpy(np.float32, (1, D))
                    qtmp.free()
                    fi, fd, fc = ix16.search(qf, K, metric)
                    ix16.set_path(1)
                    di2, dd2, dc2 = ix16.search(qf, K, metric)
                    ix16.set_path(0)
                    rec["parity"] = {"filter_path_equals_device_dense_path": bool(np.array_equal(fi, di2) and np.array_equal(fd.view(np.uint32), dd2.view(np.uint32)))}
                    # ... and against the CPU oracle over all 10M rows (read back from HBM chunk by chunk, widened f16 -> f32 as the
                    # oracle scores them)
                    import oracle as orc_h

                    t_o = time.time()
                    thr_h = min(os.cpu_count() or 1, 256)
                    oi = np.empty(0, np.int64)
                    od = np.empty(0, np.float32)
                    omet_h = orc_h.COSINE if metric == pvs.COSINE else orc_h.L2
                    for off in range(0, N, args.chunk_rows):
                        m = min(args.chunk_rows, N - off)
                        ci, cd = orc_h.search(orc_h.F16, omet_h, ix16.read_rows(off, m), qf, K, ids=np.arange(off, off + m, dtype=np.int64), threads=thr_h)
                        oi, od = orc_h.topk(np.concatenate([od, cd[0]]), K, ids=np.concatenate([oi, ci[0]]))
                    rec["parity"].update({"oracle_rows": N, "oracle_threads": thr_h, "oracle_seconds": round(time.time() - t_o, 1),
                                          "ids_and_distances_bit_exact": bool(np.array_equal(fi[0, :K], oi) and np.array_equal(fd[0, :K].view(np.uint32), od.view(np.uint32)))})
                secondary.append(rec)
                ix16.close()
            # (c) the reference's own shape: ONE query (its API answers one per request, api/search.rs:524-694), a page of 10 rows
            # (DEFAULT_LIMIT), over an index of its measured size (690k vectors, docs/vector-quant-measurements.md) — served by the
            # one-launch exact search (csrc/pvs_direct.hip): queries/s of the pipelined entry point + the latency of pvs_search
            n_ref, k_ref = 690_000, 10
            ixr = pvs.VectorIndex(pvs.I8, D, device=device, capacity_rows=n_ref)
            ixr.set_scale(scale)
            str_ = pvs.DeviceBuffer(n_ref * D * 4, device)
            L.check(lib.pvs_synth_rows_f32(device, SEED_CORPUS, 0, n_ref, D, str_.ptr))
            ixr.add_f32((str_, n_ref))
            str_.free()
            ixr.sync()
            rec = timed_region(ixr, "i8", 1, 200, 10, K=k_ref)
            rec["tag"] = "690k_i8x1"
            rec["what"] = "the reference's request shape: one query, page of 10, 690k x 768 int8 (one launch: exact distances + page)"
            qf = np.empty((1, D), np.float32)
            qtmp = pvs.DeviceBuffer(D * 4, device)
            L.check(lib.pvs_synth_rows_f32(device, SEED_QUERY, 0, 1, D, qtmp.ptr))
            qf[:] = qtmp.to_numpy(np.float32, (1, D))
            qtmp.free()
            for _ in range(10):
                ixr.search(qf, k_ref, metric)
            lat = []
            for _ in range(300):
                t_l = time.perf_counter()
                hi_, hd_, hc_ = ixr.search(qf, k_ref, metric)
                lat.append(time.perf_counter() - t_l)
            lat = np.sort(np.array(lat)) * 1e3
            rec["pvs_search_latency_ms"] = {"p50": round(float(lat[150]), 4), "p99": round(float(lat[296]), 4), "calls": 300,
                                            "how": "host-buffer entry point, one caller: query from and page into pinned memory, one synchronisation"}
            rec["path"]["direct_queries"] = int(pvs.debug_get("direct_queries"))
            if not args.no_verify:
                import oracle as orc_r

                rows_r = ixr.read_rows(0, n_ref)
                ei_r, ed_r = orc_r.search(orc_r.I8, orc_r.COSINE if metric == pvs.COSINE else orc_r.L2, rows_r, orc_r.quantize_int8(qf, scale), k_ref, threads=min(os.cpu_count() or 1, 64))
                rec["parity"] = {"oracle_rows": n_ref, "ids_and_distances_bit_exact": bool(np.array_equal(hi_[0, :k_ref], ei_r[0]) and np.array_equal(hd_[0, :k_ref].view(np.uint32), ed_r[0].view(np.uint32)))}
            secondary.append(rec)
            lat1 = rec["pvs_search_latency_ms"]["p50"]
            # (d) a FEW queries at that scale — a PQL `or` of vector filters over one space (pql/builder.rs:638-661), callers that
            # arrive together (16 read connections, db/connection.rs:235): four queries share the one launch (round 5)
            rec = timed_region(ixr, "i8", 4, 200, 10, K=k_ref)
            rec["tag"] = "690k_i8x4"
            rec["what"] = "a few queries at the reference's scale: four queries, pages of 10, 690k x 768 int8, ONE launch (k_direct_topk, 4-query instance)"
            qf4 = np.empty((4, D), np.float32)
            qtmp = pvs.DeviceBuffer(4 * D * 4, device)
            L.check(lib.pvs_synth_rows_f32(device, SEED_QUERY, 0, 4, D, qtmp.ptr))
            qf4[:] = qtmp.to_numpy(np.float32, (4, D))
            qtmp.free()
            for _ in range(10):
                ixr.search(qf4, k_ref, metric)
            lat = []
            for _ in range(300):
                t_l = time.perf_counter()
                hi4, hd4, hc4 = ixr.search(qf4, k_ref, metric)
                lat.append(time.perf_counter() - t_l)
            lat = np.sort(np.array(lat)) * 1e3
            rec["pvs_search_latency_ms"] = {"p50": round(float(lat[150]), 4), "p99": round(float(lat[296]), 4), "calls": 300,
                                            "vs_single_query_p50": round(float(lat[150]) / lat1, 3) if lat1 else None,
                                            "how": "host-buffer entry point, one caller, four queries per call"}
            rec["path"]["direct_queries"] = int(pvs.debug_get("direct_queries"))
            if not args.no_verify:
                ei4, ed4 = orc_r.search(orc_r.I8, orc_r.COSINE if metric == pvs.COSINE else orc_r.L2, rows_r, orc_r.quantize_int8(qf4, scale), k_ref, threads=min(os.cpu_count() or 1, 64))
                rec["parity"] = {"oracle_rows": n_ref, "ids_and_distances_bit_exact": bool(np.array_equal(hi4[:, :k_ref], ei4) and np.array_equal(hd4[:, :k_ref].view(np.uint32), ed4.view(np.uint32)))}
            secondary.append(rec)
            ixr.close()
            # (e) the reference's EXACT mode per item (f32 payloads narrowed to f16 here; filters/exact.rs:106-165: every row's distance,
            # GROUP BY file, AVG, rank): 32 float queries over 4M x 768 rows in 1.33M files — every (row, query) pair is one in-order f32
            # chain, so the scorer (k_exact_wide, round 5) is bound by the packed-f32 VALU rate, not by HBM: its roofline is that rate
            n_it, b_it, k_it = 4_000_000, 32, 50
            for it_name, it_dt, it_esz, it_orc in (("f16", pvs.F16, 2, "F16"), ("f32", pvs.F32, 4, "F32")):  # (round 6, end: the reference's exact mode IS f32 — its line is driver-visible too)
                ixg = pvs.VectorIndex(it_dt, D, device=device, capacity_rows=n_it)
                stg = pvs.DeviceBuffer(chunk * D * 4, device)
                rng_g = np.random.default_rng(7)
                grp_all = []
                for off in range(0, n_it, chunk):
                    m = min(chunk, n_it - off)
                    L.check(lib.pvs_synth_rows_f32(device, SEED_CORPUS, off, m, D, stg.ptr))
                    g = np.sort(rng_g.integers(off // 3, (off + m) // 3 + 1, m)).astype(np.int64)  # ~3 adjacent rows per file
                    grp_all.append(g)
                    L.check(lib.pvs_index_add_f32(ixg._h, stg.ptr, m, None, g.ctypes.data, L.DEVICE))
                stg.free()
                ixg.sync()
                qg = np.empty((b_it, D), np.float32)
                qtmp = pvs.DeviceBuffer(b_it * D * 4, device)
                L.check(lib.pvs_synth_rows_f32(device, SEED_QUERY, 0, b_it, D, qtmp.ptr))
                qg[:] = qtmp.to_numpy(np.float32, (b_it, D))
                qtmp.free()
                for _ in range(2):
                    ixg.search_groups(qg, k_it, metric, pvs.AGG_AVG)
                ixg.set_profiling(True)
                ixg.profile(reset=True)
                n_calls = 6
                cq0, cr0 = pvs.debug_get("float_certify_queries"), pvs.debug_get("float_certify_rows")
                t_g = time.perf_counter()
                for _ in range(n_calls):
                    gg, gv, gc = ixg.search_groups(qg, k_it, metric, pvs.AGG_AVG)
                el_g = time.perf_counter() - t_g
                cert_q, cert_r = pvs.debug_get("float_certify_queries") - cq0, pvs.debug_get("float_certify_rows") - cr0
                pg_ = ixg.profile()
                ixg.set_profiling(False)
                sms_g = pg_.scan_ms / max(pg_.scan_launches, 1)
                # Round 6: the page is CERTIFIED, not computed row by row (csrc/pvs_items_float.hip): one matrix-core pass writes the scan key of
                # every (row, query) pair, per-file brackets of the aggregate pick the files that can reach the page, and the reference's
                # in-order f32 chain runs on those files only.  The dominant kernel is the scan (k_scan MODE 4): HBM-bound, the rows once.
                # (Until round 5 every pair ran the exact chain: k_exact_wide, 4.05 ms of a 4.58-ms call, bound by the packed-f32 VALU rate.)
                by_g = n_it * D * it_esz
                gbs_g = by_g / (sms_g * 1e-3) / 1e9 if pg_.scan_launches else 0.0
                under_g = None
                if not args.no_peaks:
                    try:
                        under_g = sample_clock_and_power(lambda i: ixg.search_groups(qg, k_it, metric, pvs.AGG_AVG), lambda: None, seconds=1.5)
                    except Exception as e:  # noqa: BLE001
                        under_g = {"error": str(e)}
                # the exact-everywhere route on the same index, for the record (pvs_debug no_float_certify)
                pvs.debug_set("no_float_certify", 1)
                try:
                    ixg.search_groups(qg, k_it, metric, pvs.AGG_AVG)
                    t_x = time.perf_counter()
                    for _ in range(2):
                        xg, xv, xc = ixg.search_groups(qg, k_it, metric, pvs.AGG_AVG)
                    ms_exact_everywhere = (time.perf_counter() - t_x) / 2 * 1e3
                finally:
                    pvs.debug_set("no_float_certify", 0)
                rec = {"tag": f"items_{it_name}x32", "config": {"workload": f"per-item AVG: {n_it}x{D} {it_name} rows in ~{n_it // 3} files, batch {b_it}, {args.metric}, k={k_it}", "rows": n_it, "dim": D,
                                  "batch": b_it, "k": k_it},
                       "what": "the reference's exact mode per item over FLOAT rows (GROUP BY file, AVG, page): certified — matrix-core brackets per file, exact in-order rescan of the candidate files only",
                       "metric": "knn_queries_per_sec", "value": round(n_calls * b_it / el_g, 1), "unit": "queries/s", "steps": n_calls, "ms_per_step": round(el_g / n_calls * 1e3, 4),
                       "dtype": f"{it_name} rows; brackets from f16 MFMA keys, pages from f32 chains", "data": "synthetic",
                       "roofline": {"bound": "hbm", "achieved": round(gbs_g, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs_g / HBM_PEAK_GBS, 4),
                                    "traffic": replay_traffic(n_it, D, it_name, b_it)[0], "traffic_source": replay_traffic(n_it, D, it_name, b_it)[1],
                                    "kernel": f"k_scan<{it_name}, {D * it_esz} B, 32 queries, MODE 5> (per-row brackets of the distance folded per file in the epilogue)", "launches": int(pg_.scan_launches),
                                    "avg_launch_ms": round(sms_g, 4), "algorithmic_bytes_per_launch": int(by_g), "kernel_events": "timed region",
                                    "whole_call_frac_of_peak": round(by_g / (el_g / n_calls) / 1e9 / HBM_PEAK_GBS, 4),
                                    **({"under_load": under_g} if under_g else {})},
                       "certified": {"queries_per_call": cert_q / n_calls, "candidate_rows_per_query": round(cert_r / max(cert_q, 1), 1),
                                     "exact_everywhere_ms_per_call": round(ms_exact_everywhere, 3),
                                     "same_pages_as_exact_everywhere": bool(np.array_equal(gg, xg) and np.array_equal(gv.view(np.uint64), xv.view(np.uint64)) and np.array_equal(gc, xc))}}
                if not args.no_verify:  # one column against the oracle over ALL rows (scored chunk by chunk), aggregated and ranked as SQLite does
                    import oracle as orc_g

                    t_o = time.time()
                    thr_g = min(os.cpu_count() or 1, 256)
                    omet_g = orc_g.COSINE if metric == pvs.COSINE else orc_g.L2
                    col = b_it - 1
                    dd = np.concatenate([orc_g.score_all(getattr(orc_g, it_orc), omet_g, ixg.read_rows(off, min(args.chunk_rows, n_it - off)), qg[col], threads=thr_g)
                                         for off in range(0, n_it, args.chunk_rows)])
                    og_, ov_ = orc_g.aggregate(dd, np.concatenate(grp_all), orc_g.AGG_AVG)
                    eg_, ev_ = orc_g._rank_groups(og_, ov_, k_it)
                    rec["parity"] = {"oracle_rows": n_it, "oracle_column": col, "oracle_seconds": round(time.time() - t_o, 1),
                                     "groups_and_f64_values_bit_exact": bool(int(gc[col]) == len(eg_) and np.array_equal(gg[col, :len(eg_)], eg_)
                                                                             and np.array_equal(gv[col, :len(eg_)].view(np.uint64), np.asarray(ev_).view(np.uint64)))}
                secondary.append(rec)
                ixg.close()
            # (f) similar_to at the reference's measured scale (filters/item_similarity.rs:432-581; its worst performer: 9.5-31 s per call
            # at ~690k vectors, docs/or-composition-penalty.md:225): the 8 stored vectors of one item against every other row, AVG per
            # item, page of 100 — int8 rows (quant mode)
            n_sim, per_item, k_sim = 690_000, 8, 100
            ixs = pvs.VectorIndex(pvs.I8, D, device=device, capacity_rows=n_sim)
            ixs.set_scale(scale)
            sts = pvs.DeviceBuffer(n_sim * D * 4, device)
            L.check(lib.pvs_synth_rows_f32(device, SEED_CORPUS, 0, n_sim, D, sts.ptr))
            g_sim = np.arange(n_sim, dtype=np.int64) // per_item
            L.check(lib.pvs_index_add_f32(ixs._h, sts.ptr, n_sim, None, g_sim.ctypes.data, L.DEVICE))
            sts.free()
            ixs.sync()
            tg = np.arange(per_item * 1000, per_item * 1001, dtype=np.int64)  # every vector of item 1000 (row ids = row numbers)
            for _ in range(5):
                ixs.similar_to(tg, k_sim, metric, pvs.AGG_AVG)
            lat = []
            for _ in range(100):
                t_l = time.perf_counter()
                sg, sv = ixs.similar_to(tg, k_sim, metric, pvs.AGG_AVG)
                lat.append(time.perf_counter() - t_l)
            lat = np.sort(np.array(lat)) * 1e3
            rec = {"tag": "similar_i8", "config": {"workload": f"similar_to: {n_sim}x{D} i8 rows, {per_item} vectors per item, {per_item} target vectors, AVG per item, {args.metric}, k={k_sim}",
                              "rows": n_sim, "dim": D, "targets": per_item, "k": k_sim},
                   "what": "the reference's similar_to at its measured scale (9.5-31 s per call there): target vectors x every other row, per-item AVG, page",
                   "metric": "calls_per_sec", "value": round(1e3 / float(lat[50]), 1), "unit": "calls/s", "steps": 100, "ms_per_step": round(float(lat[50]), 4),
                   "latency_ms": {"p50": round(float(lat[50]), 4), "p99": round(float(lat[98]), 4)}, "dtype": "i8", "data": "synthetic",
                   "roofline": {"bound": "hbm", "achieved": round(n_sim * D / (float(lat[50]) * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": round(n_sim * D / (float(lat[50]) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                                "kernel": "whole call (gather targets, k_scan MODE 2 scorer, fan-out aggregate, page ranking) against the bytes of the rows once",
                                "algorithmic_bytes_per_launch": int(n_sim * D)}}
            if not args.no_verify:
                import oracle as orc_s

                eg_s, ev_s = orc_s.similar_to(orc_s.I8, orc_s.COSINE if metric == pvs.COSINE else orc_s.L2, ixs.read_rows(0, n_sim), [int(x) for x in tg], g_sim,
                                              orc_s.AGG_AVG, k_sim)
                rec["parity"] = {"oracle_rows": n_sim, "groups_and_f64_values_bit_exact": bool(np.array_equal(sg, eg_s) and np.array_equal(np.asarray(sv).view(np.uint64), np.asarray(ev_s).view(np.uint64)))}
            secondary.append(rec)
            ixs.close()
            # (g) the headline shape and the reference's request shape on a REALISTIC distribution: 2,000 anisotropic clusters with power-law
            # sizes, runs of near-duplicates, 1 % exact duplicates (pvs_synth_rows_clustered_f32; the reference measures on production
            # CLIP / mpnet embeddings, docs/vector-int8-quant.md:220-224) — where the sampled threshold of the filter scan is NOT at its
            # best: q/s, which path answered, candidates per query, parity
            import oracle as orc_c

            def build_clustered(n_rows):
                amax_c = 0.0
                stc = pvs.DeviceBuffer(chunk * D * 4, device)
                for off in range(0, n_rows, chunk):
                    m = min(chunk, n_rows - off)
                    L.check(lib.pvs_synth_rows_clustered_f32(device, SEED_CORPUS, off, m, D, stc.ptr))
                    outc = L.C.c_float()
                    L.check(lib.pvs_absmax(stc.ptr, m * D, L.DEVICE, device, L.C.byref(outc)))
                    amax_c = max(amax_c, float(outc.value))
                sc = pvs.scale_from_absmax(amax_c)
                ixc = pvs.VectorIndex(pvs.I8, D, device=device, capacity_rows=n_rows)
                ixc.set_scale(sc)
                for off in range(0, n_rows, chunk):
                    m = min(chunk, n_rows - off)
                    L.check(lib.pvs_synth_rows_clustered_f32(device, SEED_CORPUS, off, m, D, stc.ptr))
                    ixc.add_f32((stc, m))
                stc.free()
                ixc.sync()
                return ixc, sc

            def clustered_parity(ixc, sc, n_rows, b, kk, n_oracle):
                qtmp2 = pvs.DeviceBuffer(b * D * 4, device)
                L.check(lib.pvs_synth_rows_clustered_f32(device, SEED_CORPUS, CLUSTERED_QUERY_ROW0, b, D, qtmp2.ptr))
                qc = qtmp2.to_numpy(np.float32, (b, D)).copy()
                qtmp2.free()
                fi, fd, fc = ixc.search(qc, kk, metric)
                ixc.set_path(1)
                di3, dd3, dc3 = ixc.search(qc, kk, metric)
                ixc.set_path(0)
                par = {"queries_vs_device_dense_path": b,
                       "filter_path_equals_device_dense_path": bool(np.array_equal(fc, dc3) and np.array_equal(fi, di3) and np.array_equal(fd.view(np.uint32), dd3.view(np.uint32)))}
                qh3 = orc_c.quantize_int8(qc[:n_oracle], sc)
                om3 = orc_c.COSINE if metric == pvs.COSINE else orc_c.L2
                acc_i = [np.empty(0, np.int64) for _ in range(n_oracle)]
                acc_d = [np.empty(0, np.float32) for _ in range(n_oracle)]
                for off in range(0, n_rows, args.chunk_rows):
                    m = min(args.chunk_rows, n_rows - off)
                    ci, cd = orc_c.search(orc_c.I8, om3, ixc.read_rows(off, m), qh3, kk, ids=np.arange(off, off + m, dtype=np.int64), threads=min(os.cpu_count() or 1, 256))
                    for qq_ in range(n_oracle):
                        acc_i[qq_], acc_d[qq_] = orc_c.topk(np.concatenate([acc_d[qq_], cd[qq_]]), kk, ids=np.concatenate([acc_i[qq_], ci[qq_]]))
                par.update({"oracle_rows": n_rows, "oracle_queries": n_oracle,
                            "ids_and_distances_bit_exact": bool(all(np.array_equal(fi[x, :kk], acc_i[x]) and np.array_equal(fd[x, :kk].view(np.uint32), acc_d[x].view(np.uint32)) for x in range(n_oracle)))})
                return par

            t_c = time.time()
            ixc, sc = build_clustered(N)
            rec = timed_region(ixc, "i8", B, 20, 5, clustered=True)
            rec["tag"] = "clustered_i8x128"
            rec["what"] = ("the headline shape on a realistic distribution: 10M x 768 int8, 128 queries, k = 100 over 2,000 anisotropic power-law clusters with "
                           "near-duplicate runs and 1 % exact duplicates; queries drawn from the same clusters")
            rec["vs_iid_headline_qps"] = round(rec["value"] / qps, 3)
            if not args.no_verify:
                rec["parity"] = clustered_parity(ixc, sc, N, B, K, 2)
            rec["build_and_check_seconds"] = round(time.time() - t_c, 1)
            secondary.append(rec)
            ixc.close()
            ixc, sc = build_clustered(690_000)
            rec = timed_region(ixc, "i8", 1, 200, 10, K=10, clustered=True)
            rec["tag"] = "clustered_690k_i8x1"
            rec["what"] = "the reference's request shape (one query, page of 10, 690k x 768 int8) on the clustered distribution"
            if not args.no_verify:
                rec["parity"] = clustered_parity(ixc, sc, 690_000, 4, 10, 4)
            secondary.append(rec)
            ixc.close()
        except Exception as e:  # noqa: BLE001  (the headline line must not depend on the extras)
            secondary.append({"error": str(e)})
        result["secondary"] = secondary

    # ------------------------------------------------- verification (untimed)
    def oracle_page_over(ixh, row_base, n_rows, qh, odt, omet, threads):
        """the oracle's top-K over rows [0, n_rows) of index ixh (read back from HBM), ids = row_base + row"""
        import oracle as orc

        nq = len(qh)
        acc_i = [np.empty(0, np.int64) for _ in range(nq)]
        acc_d = [np.empty(0, np.float32) for _ in range(nq)]
        for off in range(0, n_rows, args.chunk_rows):
            m = min(args.chunk_rows, n_rows - off)
            rows = ixh.read_rows(off, m)
            ids = np.arange(row_base + off, row_base + off + m, dtype=np.int64)
            ci, cd = orc.search(odt, omet, rows, qh, K, ids=ids, threads=threads)
            for q in range(nq):
                acc_i[q], acc_d[q] = orc.topk(np.concatenate([acc_d[q], cd[q]]), K, ids=np.concatenate([acc_i[q], ci[q]]))
        return acc_i, acc_d

    if not args.no_verify:
        import oracle as orc

        odt = {pvs.I8: orc.I8, pvs.F16: orc.F16, pvs.F32: orc.F32}[dtype]
        omet = orc.COSINE if metric == pvs.COSINE else orc.L2
        nq = max(1, min(args.check_queries, B))
        qf32 = qbufs[0].to_numpy(np.float32, (B, D))
        qh = (orc.quantize_int8(qf32, scale) if dtype == pvs.I8 else qf32)[:nq]
        t_or = time.time()
        if world > 1:
            # every rank: oracle page over ITS shard -> gathered over the control socket -> host merge = the oracle's page
            # over the whole corpus; compared with what the GPUs returned.
            oi, od, oc = outs[0]
            if comm is not None:
                L.check(lib.pvs_search_sharded(ix._h, comm, qbufs[0].ptr, L.F32, B, K, metric, oi.ptr, od.ptr, oc.ptr))
                gi, gd = oi.to_numpy(np.int64, (B, K))[:nq], od.to_numpy(np.float32, (B, K))[:nq]
            else:
                gi, gd, _ = step_sharded(0)
                gi, gd = gi[:nq], gd[:nq]
            acc_i, acc_d = oracle_page_over(ix, r0, n_local, qh, odt, omet, max(1, orc.max_threads() // world))
            best_i = np.full((nq, K), -1, np.int64)
            best_d = np.full((nq, K), np.nan, np.float32)
            cnt = np.zeros(nq, np.uint32)
            for q in range(nq):
                c = len(acc_i[q])
                best_i[q, :c], best_d[q, :c], cnt[q] = acc_i[q], acc_d[q], c
            ei, ed, ec = pvs.merge_shard_pages(best_i, best_d, cnt, ctl, K)
            # every rank must hold the same merged page
            same = ctl.min_float(float(np.array_equal(gi, ei) and np.array_equal(gd.view(np.uint32), ed.view(np.uint32))))
            if rank == 0:
                hits = sum(len(set(gi[q].tolist()) & set(ei[q, : ec[q]].tolist())) for q in range(nq))
                result["recall_at_k"] = round(hits / (nq * min(K, N)), 6)
                result["parity"] = {"checked_queries": nq, "oracle_rows": N, "ids_and_distances_bit_exact": bool(same > 0),
                                    "checked_on": f"every one of the {world} ranks",
                                    "how": "per-rank oracle pages over each shard, gathered over the control socket and merged on the host",
                                    "oracle_seconds": round(time.time() - t_or, 1)}
        elif rank == 0:
            # The page of EVERY query of the batch, through the entry point and kernel instances the timed region used
            # (search_device at batch B), against (a) the device's dense path — the reference's algorithm in HBM: exact
            # distance of every row, full ordering — for all B queries and (b) the CPU oracle over the whole corpus for
            # the first `--check-queries` of them.
            oi, od, oc = outs[0]
            ix.wait(ix.search_device(qbufs[0], L.F32, B, K, metric, oi, od, oc))
            gi, gd, gc = oi.to_numpy(np.int64, (B, K)), od.to_numpy(np.float32, (B, K)), oc.to_numpy(np.uint32, (B,))
            t_dense = time.time()
            ix.set_path(1)
            di, dd, dc = ix.search(qf32, K, metric)
            ix.set_path(0)
            dense_same = bool(np.array_equal(gc, dc) and np.array_equal(gi, di[:, :K]) and np.array_equal(gd.view(np.uint32), dd[:, :K].view(np.uint32)))
            t_dense = time.time() - t_dense
            best_i, best_d = oracle_page_over(ix, r0, n_local, qh, odt, omet, orc.max_threads())
            hits = sum(len(set(gi[q, :K].tolist()) & set(best_i[q].tolist())) for q in range(nq))
            exact = all(np.array_equal(gi[q, : len(best_i[q])], best_i[q]) and
                        np.array_equal(gd[q, : len(best_d[q])].view(np.uint32), best_d[q].view(np.uint32)) for q in range(nq))
            result["recall_at_k"] = round(hits / (nq * min(K, n_local)), 6)
            result["parity"] = {"checked_queries": nq, "oracle_rows": n_local, "ids_and_distances_bit_exact": bool(exact),
                                "oracle_threads": orc.max_threads(), "oracle_seconds": round(time.time() - t_or - t_dense, 1),
                                "batch_vs_device_dense_path": {"queries": B, "identical_pages": dense_same, "seconds": round(t_dense, 1)},
                                "entry_point": f"pvs_search_device, batch {B} (the timed region's kernel instances)"}
        if rank == 0 and not args.no_cpu_baseline and n_gpus == 1:
            S = min(args.cpu_sample_rows, n_local)
            Q = min(args.cpu_sample_queries, B)
            rows = ix.read_rows(0, S)
            qq = (orc.quantize_int8(qf32, scale) if dtype == pvs.I8 else qf32)[:Q]
            t1 = time.perf_counter()
            orc.search(odt, omet, rows, qq, K, threads=1)
            dt1 = time.perf_counter() - t1
            allc = orc.max_threads()
            # all host cores: the queries side by side (each one scores with its share of the cores, then
            # runs its own single-threaded sort, like concurrent SQLite read connections would)
            from concurrent.futures import ThreadPoolExecutor

            per_q = max(1, allc // Q)
            reps = 0
            t2 = time.perf_counter()
            with ThreadPoolExecutor(max_workers=Q) as ex:
                while reps == 0 or time.perf_counter() - t2 < 1.5:  # at least 1.5 s of all-core work
                    list(ex.map(lambda i: orc.search(odt, omet, rows, qq[i:i + 1], K, threads=per_q), range(Q)))
                    reps += 1
            dt2 = (time.perf_counter() - t2) / reps
            result["cpu_baseline"] = {
                "value": round(Q / dt1 * S / N, 4), "unit": "queries/s", "cores": 1, "kind": "port",
                "sample": f"oracle scalar scan + top-{K} over the first {S} rows x {Q} queries of the same corpus "
                          f"({dt1:.1f}s measured), extrapolated linearly to {N} rows",
                "all_cores": {"value": round(Q / dt2 * S / N, 4), "cores": allc, "seconds": round(dt2 * reps, 2), "repetitions": reps},
            }
            try:  # the reference's SQL shape through real SQLite, per-row scalar UDF = the oracle
                result["cpu_baseline"]["sqlite_udf"] = sqlite_udf_baseline(orc, odt, omet, rows[:50_000], qq[0], K, N)
            except Exception as e:  # noqa: BLE001
                result["cpu_baseline"]["sqlite_udf"] = {"error": str(e)}
    if rank == 0:
        # The driver keeps the LAST 2,000 characters of this line: every secondary region once more, compact (tag, ms per step or per
        # call, fraction of its roofline, parity verdict), and the projected shard ladder, as the line's final keys.
        if secondary:
            result["secondary_summary"] = summarize_secondary(secondary)
        proj = replay_projected_scaling(N, D, args.dtype, B)
        if proj and n_gpus == 1 and not args.force_comm:
            result["projected_scaling"] = proj
        real_stdout.write(json.dumps(result) + "\n")
        real_stdout.flush()
    ctl.barrier()
    if comm is not None:
        lib.pvs_comm_destroy(comm)
    ix.close()
    ctl.close()


if __name__ == "__main__":
    main()
