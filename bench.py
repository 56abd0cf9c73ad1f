#!/usr/bin/env python3
"""bench.py — k-NN queries/sec on the BASELINE.json headline workload.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json `metric`, configs[2]): 10,000,000 x 768 int8-quantized
embeddings, batches of 128 queries, cosine, k = 100.  One "step" = one batch of
128 queries answered over the whole corpus (page 1 of the reference ordering).
Synthetic data: unit-normalised pseudo-Gaussian rows generated
in HBM, absmax -> scale -> quantize_int8 on the device.  Inputs are resident in
HBM when the timed region starts.

N > 1: launched by torch.distributed.run, one process per GPU; the corpus is
row-sharded (strong scaling: the corpus is fixed at --rows), every rank answers
the same batch over its shard, the per-shard pages are exchanged by one RCCL
all-gather over xGMI and merged on every rank.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md)
I8_MFMA_PEAK_TOPS = 5000.0  # dense int8 MFMA (~2x the 2.5 PF bf16 dense peak)
F16_MFMA_PEAK_TFLOPS = 2500.0
SEED_CORPUS = 20260928
SEED_QUERY = 0x5EED0000


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--dtype", choices=["i8", "f16", "f32"], default="i8")
    ap.add_argument("--metric", choices=["cosine", "l2"], default="cosine")
    ap.add_argument("--inflight", type=int, default=0,
                    help="search batches queued ahead of the one being waited for (default: 2 on one GPU, 4 on several)")
    ap.add_argument("--streams", type=int, default=0,
                    help="1: all batches on one HIP stream (default on one GPU); >1: one stream per in-flight batch, collectives "
                         "on their own stream (default on several GPUs, where a shard's scan is too short to fill the machine alone)")
    ap.add_argument("--check-queries", type=int, default=2, help="queries verified against the CPU oracle over the full corpus")
    ap.add_argument("--cpu-sample-rows", type=int, default=400_000)
    ap.add_argument("--cpu-sample-queries", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--chunk-rows", type=int, default=1_000_000)
    ap.add_argument("--no-kernel-events", action="store_true", help="do not bracket the kernels with HIP events (roofline fields become 0)")
    ap.add_argument("--force-comm", action="store_true", help="use the RCCL shard-merge path even with one rank (testing)")
    return ap.parse_args()


class Dist:
    """Control plane for N > 1 (rendezvous, barrier, tiny host reductions): torch.distributed
    over gloo.  The data path (top-k exchange) is RCCL inside libpvs."""

    def __init__(self, gpus: int):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.td = None
        if self.world > 1 or "--force-comm" in sys.argv:
            import torch  # noqa: F401  (plumbing only)
            import torch.distributed as td

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            td.init_process_group(backend="gloo", rank=self.rank, world_size=self.world)
            self.td = td
        if gpus != self.world and self.rank == 0:
            print(f"[bench] note: --gpus {gpus} but WORLD_SIZE={self.world}; using WORLD_SIZE", file=sys.stderr)

    def barrier(self):
        if self.td:
            self.td.barrier()

    def max_float(self, v: float) -> float:
        if not self.td:
            return v
        import torch

        t = torch.tensor([v], dtype=torch.float64)
        self.td.all_reduce(t, op=self.td.ReduceOp.MAX)
        return float(t.item())

    def bcast_bytes(self, b: bytes | None, n: int) -> bytes:
        if not self.td:
            return b
        import torch

        t = torch.zeros(n, dtype=torch.uint8)
        if self.rank == 0:
            t = torch.frombuffer(bytearray(b), dtype=torch.uint8).clone()
        self.td.broadcast(t, src=0)
        return bytes(t.numpy().tobytes())

    def all_gather_np(self, a: np.ndarray) -> np.ndarray:
        import torch

        a = np.ascontiguousarray(a)
        t = torch.from_numpy(a.view(np.uint8).reshape(-1))
        outs = [torch.empty_like(t) for _ in range(self.world)]
        self.td.all_gather(outs, t)
        return np.stack([o.numpy().view(a.dtype).reshape(a.shape) for o in outs])

    def close(self):
        if self.td:
            self.td.destroy_process_group()


def which_config(n, d, dtype, b, k):
    """BASELINE.json `configs` entry a workload corresponds to (the default run is configs[2])."""
    if (n, d, dtype, b, k) == (10_000_000, 768, "i8", 128, 100):
        return "BASELINE configs[2]"
    if (n, d, dtype, b, k) == (1_000_000, 768, "f16", 32, 100):
        return "BASELINE configs[1]"
    if (n, d, dtype, b, k) == (10_000, 512, "f32", 1, 10):
        return "BASELINE configs[0] shape, on the GPU"
    if (n, d, dtype, b, k) == (100_000_000, 768, "i8", 256, 100):
        return "BASELINE configs[3] corpus"
    return "not a BASELINE config"


def device_sync(pvs, device):
    # hipDeviceSynchronize through the library's own runtime; torch.cuda.synchronize too when torch is here
    from panoptikon_amd import _lib as L

    L.check(pvs.lib().pvs_device_synchronize(device))
    if "torch" in sys.modules:
        import torch

        if torch.cuda.is_available():
            torch.cuda.synchronize()


def main():
    args = parse()
    # stdout must carry exactly ONE JSON line, but gloo and RCCL print banners on fd 1 when they
    # initialise: keep a private handle on the real stdout and point fd 1 at stderr for the run.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    dist = Dist(args.gpus)
    if not os.path.exists(os.path.join(ROOT, "panoptikon_amd", "libpvs.so")):  # fresh checkout: the library is git-ignored
        if dist.rank == 0:
            import subprocess

            subprocess.check_call([sys.executable, "-m", "panoptikon_amd.build"], cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        dist.barrier()
    import panoptikon_amd as pvs
    from panoptikon_amd import _lib as L

    rank, world = dist.rank, dist.world
    if pvs.device_count() < 1:
        raise SystemExit("bench.py needs an MI355X (gfx950); libpvs has no CPU path")
    device = (dist.local_rank % pvs.device_count()) if world > 1 else 0
    dtype = {"i8": pvs.I8, "f16": pvs.F16, "f32": pvs.F32}[args.dtype]
    metric = pvs.COSINE if args.metric == "cosine" else pvs.L2
    esz = {pvs.I8: 1, pvs.F16: 2, pvs.F32: 4}[dtype]
    N, D, B, K = args.rows, args.dim, args.batch, args.k
    r0, r1 = pvs.shard_range(N, world, rank)
    n_local = r1 - r0
    lib = pvs.lib()

    def log(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

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
        amax = dist.max_float(amax)  # one scale per embedding space, global over all shards
        scale = pvs.scale_from_absmax(amax)
    ix = pvs.VectorIndex(dtype, D, device=device, capacity_rows=n_local, id_base=r0)
    if scale is not None:
        ix.set_scale(scale)
    for off in range(0, n_local, chunk):
        m = min(chunk, n_local - off)
        L.check(lib.pvs_synth_rows_f32(device, SEED_CORPUS, r0 + off, m, D, stage.ptr))
        ix.add_f32((stage, m))
    stage.free()
    ix.sync()
    device_sync(pvs, device)
    log(f"shard rows [{r0}, {r1}) resident in HBM as {args.dtype} in {time.time() - t_build:.1f}s (scale={scale})")

    multi = world > 1 or args.force_comm
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
    gather_mode = "single-gpu"
    if world > 1 or args.force_comm:
        # Every rank walks the same sequence of control-plane collectives whatever fails locally, and the
        # ranks agree (gloo all-reduce) on whether the in-library RCCL path is usable: a rank that fell
        # back alone would leave the others waiting in an all-gather.
        idb, ok, err = bytes(L.UNIQUE_ID_BYTES), 1.0, ""
        if rank == 0:
            try:
                buf = (L.C.c_uint8 * L.UNIQUE_ID_BYTES)()
                L.check(lib.pvs_comm_unique_id(buf))
                idb = bytes(buf)
            except Exception as e:  # noqa: BLE001
                ok, err = 0.0, str(e)
        idb = dist.bcast_bytes(idb, L.UNIQUE_ID_BYTES)
        ok = -dist.max_float(-ok)
        h = None
        if ok > 0:
            try:
                h = L.C.c_void_p()
                idarr = (L.C.c_uint8 * L.UNIQUE_ID_BYTES).from_buffer_copy(idb)
                L.check(lib.pvs_comm_create(idarr, world, rank, device, L.C.byref(h)))
            except Exception as e:  # noqa: BLE001
                ok, err, h = 0.0, str(e), None
            ok = -dist.max_float(-ok)
        if ok > 0:
            comm = h
            gather_mode = "rccl-allgather"
        else:
            if h is not None:
                lib.pvs_comm_destroy(h)
            log(f"in-library RCCL unavailable on some rank ({err or 'see other ranks'}); falling back to a host gather over gloo")
            gather_mode = "gloo-host-gather"

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
                                         oc.to_numpy(np.uint32, (B,)), dist.all_gather_np, K)
        return None

    def step_single(i):
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
    for i in range(args.warmup):
        step(i)
    drain()
    device_sync(pvs, device)
    dist.barrier()
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
    device_sync(pvs, device)
    dist.barrier()
    elapsed = dist.max_float(time.perf_counter() - t0)
    ix.set_profiling(False)
    prof = ix.profile()
    if overlapped and not args.no_kernel_events:
        ix.set_streams(1)
        ix.set_profiling(True)
        ix.profile(reset=True)
        for i in range(args.steps):
            step(args.warmup + i)
            drain()
        device_sync(pvs, device)
        dist.barrier()
        ix.set_profiling(False)
        prof = ix.profile()
        ix.set_streams(n_streams)
    st = ix.stats()
    qps = args.steps * B / elapsed

    # ---------------------------------------------------------------- roofline
    scan_ms = prof.scan_ms / max(prof.scan_launches, 1)
    bytes_per_launch = n_local * D * esz  # algorithmic bytes: each corpus component read once per batch
    achieved_gbs = bytes_per_launch / (scan_ms * 1e-3) / 1e9 if prof.scan_launches else 0.0
    ops_per_launch = 2.0 * n_local * D * B
    mfma_peak = I8_MFMA_PEAK_TOPS if dtype == pvs.I8 else F16_MFMA_PEAK_TFLOPS  # f32 rows run as bf16 on the matrix core
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if tj.get("rows") == n_local and tj.get("dim") == D and tj.get("dtype") == args.dtype and tj.get("batch") == B:
                traffic = tj.get("hbm_bytes_per_launch")
        except Exception:  # noqa: BLE001
            traffic = None
    roofline = {
        "bound": "hbm", "achieved": round(achieved_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved_gbs / HBM_PEAK_GBS, 4), "traffic": traffic,
        "kernel": "k_scan (pass B, filter scan)", "launches": int(prof.scan_launches),
        "avg_launch_ms": round(scan_ms, 4), "algorithmic_bytes_per_launch": int(bytes_per_launch),
        "mfma": {"achieved": round(ops_per_launch / (scan_ms * 1e-3) / 1e12, 1) if prof.scan_launches else 0.0,
                 "peak": mfma_peak, "unit": "TOP/s" if dtype == pvs.I8 else "TFLOP/s",
                 "frac": round(ops_per_launch / (scan_ms * 1e-3) / 1e12 / mfma_peak, 4) if prof.scan_launches else 0.0},
        "kernel_events": "serialized second pass over the same steps (the timed region overlaps searches on several streams)"
                         if overlapped else "timed region",
        "sample_pass_avg_ms": round(prof.sample_ms / max(prof.sample_launches, 1), 4),
        "finalize_avg_ms": round(prof.finalize_ms / max(prof.finalize_launches, 1), 4),
    }

    result = {
        "metric": "knn_queries_per_sec", "value": round(qps, 1), "unit": "queries/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"{N}x{D} {args.dtype} corpus, batch {B}, {args.metric}, k={K} ({which_config(N, D, args.dtype, B, K)})",
                   "rows": N, "dim": D, "batch": B, "k": K, "metric": args.metric,
                   "parallelism": f"row-shard x{world}", "exchange": gather_mode, "streams": n_streams, "inflight": slots if (world == 1 or comm is not None) else 1},
        "roofline": roofline,
        "path": {"fast_queries": int(st.fast_queries), "dense_queries": int(st.dense_queries)},
    }

    # ------------------------------------------------- verification (untimed)
    if world > 1 and not args.no_verify:
        # every rank: oracle page over ITS shard for a few queries -> gathered over gloo -> host merge
        # = the oracle's page over the whole corpus; compared with what the GPUs returned.
        import oracle as orc

        odt = {pvs.I8: orc.I8, pvs.F16: orc.F16, pvs.F32: orc.F32}[dtype]
        omet = orc.COSINE if metric == pvs.COSINE else orc.L2
        nq = max(1, min(args.check_queries, B))
        qf32 = qbufs[0].to_numpy(np.float32, (B, D))
        qh = (orc.quantize_int8(qf32, scale) if dtype == pvs.I8 else qf32)[:nq]
        oi, od, oc = outs[0]
        if comm is not None:
            L.check(lib.pvs_search_sharded(ix._h, comm, qbufs[0].ptr, L.F32, B, K, metric, oi.ptr, od.ptr, oc.ptr))
            gi, gd = oi.to_numpy(np.int64, (B, K))[:nq], od.to_numpy(np.float32, (B, K))[:nq]
        else:
            gi, gd, _ = step_sharded(0)
            gi, gd = gi[:nq], gd[:nq]
        threads = max(1, orc.max_threads() // world)
        best_i = np.full((nq, K), -1, np.int64)
        best_d = np.full((nq, K), np.nan, np.float32)
        cnt = np.zeros(nq, np.uint32)
        acc_i = [np.empty(0, np.int64) for _ in range(nq)]
        acc_d = [np.empty(0, np.float32) for _ in range(nq)]
        for off in range(0, n_local, args.chunk_rows):
            m = min(args.chunk_rows, n_local - off)
            rows = ix.read_rows(off, m)
            ci, cd = orc.search(odt, omet, rows, qh, K, ids=np.arange(r0 + off, r0 + off + m, dtype=np.int64), threads=threads)
            for q in range(nq):
                acc_i[q], acc_d[q] = orc.topk(np.concatenate([acc_d[q], cd[q]]), K, ids=np.concatenate([acc_i[q], ci[q]]))
        for q in range(nq):
            c = len(acc_i[q])
            best_i[q, :c], best_d[q, :c], cnt[q] = acc_i[q], acc_d[q], c
        ei, ed, ec = pvs.merge_shard_pages(best_i, best_d, cnt, dist.all_gather_np, K)
        if rank == 0:
            hits = sum(len(set(gi[q].tolist()) & set(ei[q, : ec[q]].tolist())) for q in range(nq))
            exact = bool(np.array_equal(gi, ei) and np.array_equal(gd.view(np.uint32), ed.view(np.uint32)))
            result["recall_at_k"] = round(hits / (nq * min(K, N)), 6)
            result["parity"] = {"checked_queries": nq, "oracle_rows": N, "ids_and_distances_bit_exact": exact,
                                "how": "per-rank oracle pages over each shard, gathered over gloo and merged on the host"}
    if rank == 0 and not args.no_verify and world == 1:
        import oracle as orc

        odt = {pvs.I8: orc.I8, pvs.F16: orc.F16, pvs.F32: orc.F32}[dtype]
        omet = orc.COSINE if metric == pvs.COSINE else orc.L2
        nq = max(1, min(args.check_queries, B))
        # the batch the GPU answers, as the oracle sees it
        qf32 = qbufs[0].to_numpy(np.float32, (B, D))[:nq]
        qh = orc.quantize_int8(qf32, scale) if dtype == pvs.I8 else qf32
        if True:
            gi, gd, gc = ix.search(qf32, K, metric)
            threads = orc.max_threads()
            best_i = [np.empty(0, np.int64) for _ in range(nq)]
            best_d = [np.empty(0, np.float32) for _ in range(nq)]
            t_or = time.time()
            for off in range(0, n_local, args.chunk_rows):
                m = min(args.chunk_rows, n_local - off)
                rows = ix.read_rows(off, m)
                ids = np.arange(r0 + off, r0 + off + m, dtype=np.int64)
                ci, cd = orc.search(odt, omet, rows, qh, K, ids=ids, threads=threads)
                for q in range(nq):
                    ai = np.concatenate([best_i[q], ci[q]])
                    ad = np.concatenate([best_d[q], cd[q]])
                    best_i[q], best_d[q] = orc.topk(ad, K, ids=ai)
            hits = sum(len(set(gi[q, :K].tolist()) & set(best_i[q].tolist())) for q in range(nq))
            exact = all(np.array_equal(gi[q, : len(best_i[q])], best_i[q]) and
                        np.array_equal(gd[q, : len(best_d[q])].view(np.uint32), best_d[q].view(np.uint32)) for q in range(nq))
            result["recall_at_k"] = round(hits / (nq * min(K, n_local)), 6)
            result["parity"] = {"checked_queries": nq, "oracle_rows": n_local, "ids_and_distances_bit_exact": bool(exact),
                                "oracle_threads": threads, "oracle_seconds": round(time.time() - t_or, 1)}
        if not args.no_cpu_baseline and world == 1:
            S = min(args.cpu_sample_rows, n_local)
            Q = min(args.cpu_sample_queries, B)
            rows = ix.read_rows(0, S)
            qf = qbufs[0].to_numpy(np.float32, (B, D))[:Q]
            qq = orc.quantize_int8(qf, scale) if dtype == pvs.I8 else qf
            t1 = time.perf_counter()
            orc.search(odt, omet, rows, qq, K, threads=1)
            dt1 = time.perf_counter() - t1
            allc = orc.max_threads()
            # all host cores: the queries side by side (each one scores with its share of the cores, then
            # runs its own single-threaded sort, like concurrent SQLite read connections would)
            from concurrent.futures import ThreadPoolExecutor

            per_q = max(1, allc // Q)
            t2 = time.perf_counter()
            with ThreadPoolExecutor(max_workers=Q) as ex:
                list(ex.map(lambda i: orc.search(odt, omet, rows, qq[i:i + 1], K, threads=per_q), range(Q)))
            dt2 = time.perf_counter() - t2
            result["cpu_baseline"] = {
                "value": round(Q / dt1 * S / N, 4), "unit": "queries/s", "cores": 1, "kind": "port",
                "sample": f"oracle scalar scan + top-{K} over the first {S} rows x {Q} queries of the same corpus "
                          f"({dt1:.1f}s measured), extrapolated linearly to {N} rows",
                "all_cores": {"value": round(Q / dt2 * S / N, 4), "cores": allc, "seconds": round(dt2, 2)},
            }
    if rank == 0:
        real_stdout.write(json.dumps(result) + "\n")
        real_stdout.flush()
    if comm is not None:
        lib.pvs_comm_destroy(comm)
    ix.close()
    dist.close()


if __name__ == "__main__":
    main()
