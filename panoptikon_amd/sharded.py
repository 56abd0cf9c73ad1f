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
