"""Row-sharded search across ranks: contiguous row ranges per rank,
queries replicated, per-shard pages exchanged once and merged on every rank.

The GPU data path is pvs_search_sharded (RCCL all-gather + device merge inside
libpvs) or, in one process, a multi-device index (pvs_multi.hip).  This module holds
the rank arithmetic and the host-side exchange used when the pages are gathered by
other means: `gather` is any callable that all-gathers a numpy array over the ranks
(rendezvous.LocalRendezvous; torch.distributed over gloo in the CPU tests)."""
from __future__ import annotations

import numpy as np

from .host import merge_group_pages, merge_topk


def shard_range(n_rows: int, world: int, rank: int):
    """Rows [r0, r1) owned by `rank`: ceil(n/world) contiguous rows each, so that
    global row id = r0 + local row and ids stay increasing across ranks."""
    per = (n_rows + world - 1) // world
    return min(rank * per, n_rows), min((rank + 1) * per, n_rows)


def merge_shard_pages(local_ids, local_dist, local_cnt, gather, k: int):
    """Gathers every rank's page ([batch][k] ids / distances, [batch] counts) and merges
    them under the shared ordering (distance asc, id asc, NaN last)."""
    ids = gather(np.ascontiguousarray(local_ids, np.int64))
    dist = gather(np.ascontiguousarray(local_dist, np.float32))
    cnt = gather(np.ascontiguousarray(local_cnt, np.uint32))
    return merge_topk(ids, dist, cnt, k)


def shard_ranges_by_group(group_ids, world: int):
    """Row ranges [r0, r1) per rank for a corpus whose rows are clustered by group (non-decreasing group
    ids): cuts fall on group boundaries nearest to the even split, so every group lives on one rank
    and per-item aggregates stay shard-local."""
    g = np.asarray(group_ids)
    n = len(g)
    cuts = [0]
    for r in range(1, world):
        c = min(max((n * r) // world, cuts[-1]), n)
        while 0 < c < n and g[c] == g[c - 1]:
            c += 1
        cuts.append(c)
    cuts.append(n)
    return [(cuts[i], cuts[i + 1]) for i in range(world)]


def merge_shard_group_pages(local_groups, local_values, local_cnt, gather, k: int):
    """Gathers every rank's per-item page ([batch][k] group ids / f64 values, [batch] counts) and merges them
    under the group ordering (value asc, group id asc, NULL last)."""
    grp = gather(np.ascontiguousarray(local_groups, np.int64))
    val = gather(np.ascontiguousarray(local_values, np.float64))
    cnt = gather(np.ascontiguousarray(local_cnt, np.uint32))
    return merge_group_pages(grp, val, cnt, k)


def rrf_search_sharded(branches, k: int, gather, max_rounds: int = 6):
    """pvs_rrf_search over branches sharded BY GROUP across ranks (every row of a group on one rank; BASELINE configs[4] on
    several GPUs).  `branches`: this rank's dicts as for rrf_search (its shard of each branch's index); `gather(a)` all-gathers a
    numpy array over the ranks ([world, ...]; ragged data goes through fixed-size padded arrays).  Every rank returns the same
    (groups[k], scores[k]) — the reference's page, bit for bit:

      thresholds: each shard proposes a window key from a sample, the MINIMUM over shards is used, so "at or below T_b" is the
      same set whichever shard a group lives on and R_b = sum of the shards' page sizes is the number of groups ranked before
      everything outside;  candidates = union of all pages;  exact global rank of a candidate in branch b = 1 + sum over shards
      of the groups strictly before it (one counting pass per shard);  scores by pvs_rrf_fuse (SQLite's arithmetic);  a group
      outside every page scores at most U = sum_b w_b/(k_b + R_b + 1): when the k-th candidate beats U the page is exact,
      otherwise the thresholds move up.  Messages: a few thousand (id, key) pairs per round."""
    from .host import RrfCols, rrf_fuse

    nb = len(branches)
    ks = [int(b.get("rrf_k", 1)) for b in branches]
    ws = [float(b.get("weight", 1.0)) for b in branches]
    if any(not (w >= 0.0) for w in ws) or any(kk < 0 for kk in ks):
        raise ValueError("the sharded fusion needs non-negative RRF weights and k")
    cols = [RrfCols(b) for b in branches]

    def gather_ragged(a, dtype):
        a = np.ascontiguousarray(a, dtype)
        n = gather(np.array([a.size], np.int64)).reshape(-1)
        cap = max(int(n.max()), 1)
        pad = np.zeros(cap, dtype)
        pad[: a.size] = a
        allp = gather(pad)
        return [allp[r][: int(n[r])] for r in range(len(n))]

    try:
        n_tot = [int(gather(np.array([c.n_groups], np.int64)).sum()) for c in cols]
        world = len(gather(np.array([0], np.int64)))
        target = max(4 * k, 1024)
        for _ in range(max_rounds):
            R, cand_parts = [], []
            for b in range(nb):
                t_local = cols[b].threshold(max(target // world, 64)) if cols[b].n_groups else (1 << 64) - 1
                t = int(gather(np.array([t_local], np.uint64)).min())
                cap = 8 * target + 65536
                pg = cols[b].page(t, cap) if cols[b].n_groups else (np.empty(0, np.int64), np.empty(0, np.uint64))
                ok = int(gather(np.array([0 if pg is None else 1], np.int64)).min())
                if not ok:
                    raise RuntimeError("a page overflowed (massive ties at the threshold): gather the branch on one device instead")
                R.append(sum(len(x) for x in gather_ragged(pg[0], np.int64)))
                cand_parts += gather_ragged(pg[0], np.int64)
            cand = np.unique(np.concatenate(cand_parts)) if cand_parts else np.empty(0, np.int64)
            m = len(cand)
            ranks = np.full((nb, m), -1, np.int64)
            for b in range(nb):
                keys, present = cols[b].lookup(cand)
                allk, allp = gather(keys), gather(present.astype(np.uint8))
                if (allp.sum(axis=0) > 1).any():
                    raise RuntimeError("a group lives on two shards: shard the branches BY GROUP")
                have = allp.any(axis=0)
                key = np.where(have, allk[np.argmax(allp, axis=0), np.arange(m)], 0).astype(np.uint64)
                order = np.flatnonzero(have)
                order = order[np.lexsort((cand[order], key[order]))]  # window order: (key, group id)
                below = gather(cols[b].count_below(key[order], cand[order])).sum(axis=0) if len(order) else np.empty(0, np.uint64)
                ranks[b, order] = below.astype(np.int64) + 1
            score = rrf_fuse(ranks, ks, ws) if m else np.empty(0)
            top = np.lexsort((cand, -score))[:k]
            U = sum(w / (float(kk) + float(r) + 1.0) for w, kk, r in zip(ws, ks, R)) * (1.0 + 1e-12)
            if m >= k and score[top[-1]] > U:
                return cand[top], score[top]
            if all(4 * target >= n for n in n_tot if n):
                break
            target *= 4
        # everything is a candidate: rank every group (pages with the largest key)
        raise RuntimeError("bounded fusion did not converge (k close to the number of groups): fuse on one device")
    finally:
        for c in cols:
            c.close()
