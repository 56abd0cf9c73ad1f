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


def rrf_search_sharded(branches, k: int, gather=None, comm=None, world: int = 1):
    """pvs_rrf_search over branches sharded BY GROUP across ranks (every row of a group on one rank; BASELINE configs[4] on
    several GPUs): pvs_rrf_search_sharded — the whole round loop runs inside libpvs (csrc/pvs_rrf_sharded.hip).  `branches`: this
    rank's dicts as for rrf_search (its shard of each branch's index).  Exchange: `comm` (a pvs_comm handle: RCCL over xGMI) or
    `gather(a)`, any callable that all-gathers a numpy array over the ranks into [world, ...] (rendezvous.Rendezvous; the tests'
    threads-as-ranks).  Every rank returns the same (groups[k'], scores[k']) — the reference's page, bit for bit."""
    import ctypes as C

    from . import _lib as L
    from .host import _branch_struct, _ptr

    arr = (L.RrfBranch * len(branches))()
    keep = []
    for i, b in enumerate(branches):
        arr[i] = _branch_struct(b, keep)
    og = np.empty(k, np.int64)
    ov = np.empty(k, np.float64)
    oc = C.c_uint32()
    CB = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64)
    err = []

    def _cb(_ctx, send, recv, nbytes):
        try:
            a = np.frombuffer((C.c_uint8 * nbytes).from_address(send), np.uint8).copy()
            out = np.ascontiguousarray(gather(a), np.uint8)
            C.memmove(recv, out.ctypes.data, out.size)
            return 0
        except Exception as e:  # noqa: BLE001  (an exception must not unwind through the C frame)
            err.append(e)
            return 1

    cb = CB(_cb) if (gather is not None and comm is None) else None
    cbp = C.cast(cb, C.c_void_p) if cb is not None else None
    if gather is not None and comm is None and world <= 1:
        world = len(gather(np.zeros(1, np.uint8)))
    st = L.lib().pvs_rrf_search_sharded(arr, len(branches), k, comm, world, cbp, None, _ptr(og), _ptr(ov), C.byref(oc))
    if err:
        raise err[0]
    L.check(st)
    return og[: oc.value], ov[: oc.value]
