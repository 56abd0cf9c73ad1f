"""N > 1 path on CPU: two ranks over gloo.  Each rank owns a contiguous row shard;
its local page comes from the oracle here (there is no GPU in this test), the
exchange is torch.distributed all_gather over gloo (in the test: the package itself
carries no torch), the merge the package's own (the C ABI's pvs_merge_topk) and the merged page must equal the oracle's page over the
whole corpus."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    sys.path.insert(0, %r)
    import torch.distributed as dist
    import oracle as orc
    import panoptikon_amd as pvs

    import torch
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()

    def gloo_gather(a):
        a = np.ascontiguousarray(a)
        t = torch.from_numpy(a.view(np.uint8).reshape(-1).copy())  # raw bytes: gloo has no u32
        outs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(outs, t)
        return np.stack([o.numpy().view(a.dtype).reshape(a.shape) for o in outs])

    n, dim, batch, k = 3001, 64, 5, 17
    rows = orc.synth_rows(123, 0, n, dim)
    rows[5] = rows[2000]          # a cross-shard tie: id order must decide
    rows[1700] = 0.0              # a NULL-distance row on rank 1
    queries = orc.synth_rows(77, 0, batch, dim)
    scale = orc.compute_int8_scale(rows)
    codes, qcodes = orc.quantize_int8(rows, scale), orc.quantize_int8(queries, scale)
    r0, r1 = pvs.shard_range(n, world, rank)
    for metric in (orc.COSINE, orc.L2):
        li, ld = orc.search(orc.I8, metric, codes[r0:r1], qcodes, k, ids=np.arange(r0, r1))
        cnt = np.full(batch, li.shape[1], np.uint32)
        pi = np.full((batch, k), -1, np.int64); pd = np.full((batch, k), np.nan, np.float32)
        pi[:, : li.shape[1]] = li; pd[:, : ld.shape[1]] = ld
        mi, md, mc = pvs.merge_shard_pages(pi, pd, cnt, gloo_gather, k)
        ei, ed = orc.search(orc.I8, metric, codes, qcodes, k)
        assert (mc == k).all()
        assert np.array_equal(mi, ei), (rank, metric)
        assert np.array_equal(md.view(np.uint32), ed.view(np.uint32))
    # per-item search: shard BY GROUP, local aggregate per rank (oracle here), merge of (group, f64) pages
    groups = np.sort(np.random.default_rng(5).integers(0, 400, n)).astype(np.int64)
    ranges = pvs.shard_ranges_by_group(groups, world)
    g0, g1 = ranges[rank]
    assert g0 == 0 or groups[g0] != groups[g0 - 1]
    for agg in (orc.AGG_MIN, orc.AGG_AVG, orc.AGG_MAX):
        pg = np.full((batch, k), -1, np.int64); pv = np.full((batch, k), np.nan); pc = np.zeros(batch, np.uint32)
        for q in range(batch):
            lg, lv = orc.search_groups(orc.I8, orc.COSINE, codes[g0:g1], qcodes[q], groups[g0:g1], agg, k)
            pg[q, : len(lg)], pv[q, : len(lv)], pc[q] = lg, lv, len(lg)
        mg, mv, mc = pvs.merge_shard_group_pages(pg, pv, pc, gloo_gather, k)
        for q in range(batch):
            eg, ev = orc.search_groups(orc.I8, orc.COSINE, codes, qcodes[q], groups, agg, k)
            assert mc[q] == len(eg) and np.array_equal(mg[q, : len(eg)], eg), (rank, agg, q)
            assert np.array_equal(mv[q, : len(eg)].view(np.uint64), ev.view(np.uint64))
    dist.barrier()
    dist.destroy_process_group()
    sys.stdout.write("rank" + str(rank) + "-ok" + chr(10))
    sys.stdout.flush()
""") % ROOT


def test_two_rank_gloo_merge(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    import socket

    with socket.socket() as sock:  # a free rendezvous port
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "rank0-ok" in out.stdout and "rank1-ok" in out.stdout


WORKER2 = textwrap.dedent("""
    import os, sys
    import numpy as np
    sys.path.insert(0, %r)
    import oracle as orc
    import panoptikon_amd as pvs

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    rdv = pvs.LocalRendezvous(rank, world)
    assert rdv.max_float(float(rank)) == world - 1 and rdv.min_float(float(rank)) == 0.0
    assert rdv.bcast_bytes(b"id-from-rank-0" if rank == 0 else None) == b"id-from-rank-0"
    n, dim, batch, k = 2500, 48, 4, 13
    rows = orc.synth_rows(321, 0, n, dim).astype(np.float16)
    queries = orc.synth_rows(78, 0, batch, dim)
    r0, r1 = pvs.shard_range(n, world, rank)
    li, ld = orc.search(orc.F16, orc.L2, rows[r0:r1], queries, k, ids=np.arange(r0, r1))
    mi, md, mc = pvs.merge_shard_pages(li, ld, np.full(batch, k, np.uint32), rdv, k)
    ei, ed = orc.search(orc.F16, orc.L2, rows, queries, k)
    assert np.array_equal(mi, ei) and np.array_equal(md.view(np.uint32), ed.view(np.uint32)) and (mc == k).all()
    rdv.barrier()
    rdv.close()
    sys.stdout.write("rank" + str(rank) + "-ok" + chr(10))
""") % ROOT


def test_three_ranks_over_the_package_rendezvous(tmp_path):
    """The torch-free control plane bench.py uses (panoptikon_amd/rendezvous.py): 3 ranks, star over a Unix socket."""
    script = tmp_path / "worker2.py"
    script.write_text(WORKER2)
    sock = str(tmp_path / "ctl.sock")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(os.environ, RANK=str(r), WORLD_SIZE="3", PVS_CTL_SOCK=sock),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(3)]
    outs = [p.communicate(timeout=300) for p in procs]
    for r, (p, (o, e)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, e[-3000:]
        assert f"rank{r}-ok" in o
