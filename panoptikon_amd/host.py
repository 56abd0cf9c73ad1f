"""Host-side mirror of the reference's operator helpers (names follow the Rust).

Thin wrappers over the C ABI's host entry points: scale artifact
(db/vector_quants.rs:1449-1471), per-item aggregation (filters/exact.rs:67-80),
rank / RRF (pql/builder.rs:757-771, 1284-1317), query ingestion
(pql/embedding_utils.rs) and quant resolution (pql/preprocess.rs:314-446).
"""
from __future__ import annotations

import base64
import ctypes as C

import numpy as np

from . import _lib as L


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def scale_from_absmax(absmax: float) -> float:
    return float(L.lib().pvs_scale_from_absmax(np.float32(absmax)))


def scale_artifact(scale: float) -> bytes:
    out = (C.c_uint8 * 4)()
    L.lib().pvs_scale_artifact(np.float32(scale), out)
    return bytes(out)


def artifact_scale(artifact: bytes):
    """Some(scale) / None, like the reference."""
    s = C.c_float()
    buf = (C.c_uint8 * max(len(artifact), 1)).from_buffer_copy(bytes(artifact) or b"\0")
    st = L.lib().pvs_artifact_scale(buf, len(artifact), C.byref(s))
    return float(s.value) if st == L.OK else None


def aggregate(dist, group_ids, agg: int, weights=None):
    dist = np.ascontiguousarray(dist, np.float32)
    grp = np.ascontiguousarray(group_ids, np.int64)
    w = None if weights is None else np.ascontiguousarray(weights, np.float32)
    og = np.empty(max(dist.size, 1), np.int64)
    ov = np.empty(max(dist.size, 1), np.float64)
    n = C.c_uint64()
    L.check(L.lib().pvs_aggregate(_ptr(dist), _ptr(w), _ptr(grp), dist.size, agg, _ptr(og), _ptr(ov), C.byref(n)))
    return og[: n.value].copy(), ov[: n.value].copy()


def row_number(values, ids=None, descending: bool = False) -> np.ndarray:
    """row_number() OVER (ORDER BY value ASC|DESC) with SQLite's NULL placement (first ascending, last descending)."""
    v = np.ascontiguousarray(values, np.float64)
    i = None if ids is None else np.ascontiguousarray(ids, np.int64)
    out = np.empty(max(v.size, 1), np.int64)
    L.check(L.lib().pvs_row_number_dir(_ptr(v), _ptr(i), v.size, int(descending), _ptr(out)))
    return out[: v.size]


def rrf_fuse(ranks, ks, weights) -> np.ndarray:
    """ranks: [n_branches][n] int64, < 0 = NULL."""
    r = np.ascontiguousarray(ranks, np.int64)
    if r.ndim == 1:
        r = r[:, None]
    k = np.ascontiguousarray(ks, np.int32)
    w = np.ascontiguousarray(weights, np.float64)
    out = np.empty(max(r.shape[1], 1), np.float64)
    L.check(L.lib().pvs_rrf_fuse(_ptr(r), r.shape[0], r.shape[1], _ptr(k), _ptr(w), _ptr(out)))
    return out[: r.shape[1]]


def coalesce_ranks(ranks, descending: bool = False) -> np.ndarray:
    """min/max(coalesce(rank_b, +-9223372036854775805)) over same-priority filters (builder.rs:1303-1317); ranks [nb][n], < 0 = NULL."""
    r = np.ascontiguousarray(ranks, np.int64)
    out = np.empty(max(r.shape[1], 1), np.int64)
    L.check(L.lib().pvs_coalesce_ranks(_ptr(r), r.shape[0], r.shape[1], int(descending), _ptr(out)))
    return out[: r.shape[1]]


def coalesce_values(values, descending: bool = False) -> np.ndarray:
    """the same over raw f64 aggregates (NaN = NULL)."""
    v = np.ascontiguousarray(values, np.float64)
    out = np.empty(max(v.shape[1], 1), np.float64)
    L.check(L.lib().pvs_coalesce_values(_ptr(v), v.shape[0], v.shape[1], int(descending), _ptr(out)))
    return out[: v.shape[1]]


def sort_bounds(order_rank, gt=None, lt=None) -> np.ndarray:
    """apply_sort_bounds (builder.rs:781-815): boolean keep mask for `order_rank > gt AND order_rank < lt` (NaN = NULL fails)."""
    v = np.ascontiguousarray(order_rank, np.float64)
    keep = np.zeros(max(v.size, 1), np.uint8)
    L.check(L.lib().pvs_sort_bounds(_ptr(v), v.size, int(gt is not None), float(gt or 0.0), int(lt is not None), float(lt or 0.0), _ptr(keep)))
    return keep[: v.size].astype(bool)


def _branch_struct(b, keep):
    ix = b["index"]
    q, qd = ix._queries(b["query"])
    w = None if b.get("row_weights") is None else np.ascontiguousarray(b["row_weights"], np.float32)
    keep += [q, w]
    return L.RrfBranch(ix._h.value if hasattr(ix._h, "value") else ix._h, q.ctypes.data, qd, b["metric"], b.get("agg", L.AGG_MIN),
                       None if w is None else w.ctypes.data, int(bool(b.get("descending", False))), int(b.get("rrf_k", 1)),
                       float(b.get("weight", 1.0)))


class RrfCols:
    """One branch of an OR-composition scored on one shard (pvs_rrf_cols): the pieces of the bounded fusion."""

    def __init__(self, branch: dict):
        keep = []
        st = _branch_struct(branch, keep)
        h = C.c_void_p()
        L.check(L.lib().pvs_rrf_cols_create(C.byref(st), C.byref(h)))
        self._h = h
        n = C.c_uint64()
        L.check(L.lib().pvs_rrf_cols_groups(self._h, C.byref(n)))
        self.n_groups = int(n.value)

    def close(self):
        if self._h:
            L.lib().pvs_rrf_cols_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def threshold(self, target_groups: int) -> int:
        k = C.c_uint64()
        L.check(L.lib().pvs_rrf_cols_threshold(self._h, int(target_groups), C.byref(k)))
        return int(k.value)

    def page(self, key: int, cap: int):
        """(group ids, keys) of every group at or below `key`; None when more than `cap` qualify."""
        g = np.empty(max(cap, 1), np.int64)
        kk = np.empty(max(cap, 1), np.uint64)
        n = C.c_uint32()
        L.check(L.lib().pvs_rrf_cols_page(self._h, int(key), int(cap), _ptr(g), _ptr(kk), C.byref(n)))
        if n.value > cap:
            return None
        return g[: n.value].copy(), kk[: n.value].copy()

    def lookup(self, gids):
        g = np.ascontiguousarray(gids, np.int64)
        keys = np.zeros(max(g.size, 1), np.uint64)
        present = np.zeros(max(g.size, 1), np.uint8)
        if g.size:
            L.check(L.lib().pvs_rrf_cols_lookup(self._h, _ptr(g), g.size, _ptr(keys), _ptr(present)))
        return keys[: g.size], present[: g.size].astype(bool)

    def count_below(self, keys, gids):
        k = np.ascontiguousarray(keys, np.uint64)
        g = np.ascontiguousarray(gids, np.int64)
        out = np.zeros(max(k.size, 1), np.uint64)
        if k.size:
            L.check(L.lib().pvs_rrf_cols_count_below(self._h, _ptr(k), _ptr(g), k.size, _ptr(out)))
        return out[: k.size]


def rrf_search(branches, k: int):
    """branches: dicts {index, query, metric, agg=AGG_MIN, row_weights=None, descending=False, rrf_k=1, weight=1.0}.
    OR-composition ranked by reciprocal-rank fusion on the device (pvs_rrf_search) -> (groups[k'], scores[k'])."""
    arr = (L.RrfBranch * len(branches))()
    keep = []
    for i, b in enumerate(branches):
        arr[i] = _branch_struct(b, keep)
    og = np.empty(k, np.int64)
    ov = np.empty(k, np.float64)
    oc = C.c_uint32()
    L.check(L.lib().pvs_rrf_search(C.byref(arr), len(branches), k, _ptr(og), _ptr(ov), C.byref(oc)))
    return og[: oc.value], ov[: oc.value]


def merge_group_pages(groups, values, counts, k: int, keys=None):
    """groups/values: [world][batch][k] (i64 / f64); counts: [world][batch] -> merged per-item pages.  keys ([world][batch][k]
    i64, optional): the second sort key of every entry's group (pvs_merge_group_pages_keyed)."""
    groups = np.ascontiguousarray(groups, np.int64)
    values = np.ascontiguousarray(values, np.float64)
    counts = np.ascontiguousarray(counts, np.uint32)
    world, batch = counts.shape
    og = np.empty((batch, k), np.int64)
    ov = np.empty((batch, k), np.float64)
    oc = np.empty(batch, np.uint32)
    if keys is None:
        L.check(L.lib().pvs_merge_group_pages(_ptr(groups), _ptr(values), _ptr(counts), world, batch, k, _ptr(og), _ptr(ov), _ptr(oc)))
    else:
        keys = np.ascontiguousarray(keys, np.int64)
        L.check(L.lib().pvs_merge_group_pages_keyed(_ptr(groups), _ptr(values), _ptr(keys), _ptr(counts), world, batch, k, _ptr(og), _ptr(ov), _ptr(oc)))
    return og, ov, oc


def merge_topk(ids, dist, counts, k: int, keys=None):
    """ids/dist: [world][batch][k]; counts: [world][batch] -> merged ([batch][k], ...).  keys ([world][batch][k] i64, optional): the
    second sort key of every entry's row (pvs_merge_topk_keyed: distance asc, NULL last, key DESC, id asc)."""
    ids = np.ascontiguousarray(ids, np.int64)
    dist = np.ascontiguousarray(dist, np.float32)
    counts = np.ascontiguousarray(counts, np.uint32)
    world, batch = counts.shape
    oi = np.empty((batch, k), np.int64)
    od = np.empty((batch, k), np.float32)
    oc = np.empty(batch, np.uint32)
    if keys is None:
        L.check(L.lib().pvs_merge_topk(_ptr(ids), _ptr(dist), _ptr(counts), world, batch, k, _ptr(oi), _ptr(od), _ptr(oc)))
    else:
        keys = np.ascontiguousarray(keys, np.int64)
        L.check(L.lib().pvs_merge_topk_keyed(_ptr(ids), _ptr(dist), _ptr(keys), _ptr(counts), world, batch, k, _ptr(oi), _ptr(od), _ptr(oc)))
    return oi, od, oc


def embedding_from_npy_bytes(buffer: bytes) -> bytes:
    """pql/embedding_utils.rs:10-13: .npy -> f32 little-endian bytes (Err -> PvsError(ERR_PARSE))."""
    buf = np.frombuffer(bytes(buffer), np.uint8)
    n = C.c_size_t()
    L.check(L.lib().pvs_npy_to_f32(_ptr(buf) if buf.size else None, buf.size, None, 0, C.byref(n)))
    out = np.empty(max(n.value, 1), np.float32)
    L.check(L.lib().pvs_npy_to_f32(_ptr(buf), buf.size, _ptr(out), out.size, C.byref(n)))
    return out[: n.value].astype("<f4").tobytes()


def extract_embeddings(encoded: str) -> bytes:
    """pql/embedding_utils.rs:3-8: base64 -> .npy -> f32 LE bytes."""
    try:
        decoded = base64.b64decode(encoded.encode("ascii"), validate=True)
    except Exception as err:  # noqa: BLE001
        raise L.PvsError(L.ERR_PARSE, f"Invalid base64 embeddings: {err}") from None
    return embedding_from_npy_bytes(decoded)


def resolve_vector_quant(index: int, variant, k: int, pair: L.ReadyPair | None, embedding: bytes | None):
    """pql/preprocess.rs:314-393.  Returns None (search exact) or (profile_id, query_quant | None)."""
    emb = None if embedding is None else np.frombuffer(bytes(embedding), np.uint8)
    cap = 0 if pair is None else max(int(pair.dim), 0)
    qq = np.empty(max(cap, 1), np.int8)
    out = L.QuantResolved()
    var = None if variant is None else variant.encode("utf-8")
    L.check(L.lib().pvs_resolve_vector_quant(index, var, k, None if pair is None else C.byref(pair), _ptr(emb),
                                             0 if emb is None else emb.size, _ptr(qq), cap, C.byref(out)))
    if not out.use_quant:
        return None
    return int(out.profile_id), (qq[: out.query_quant_len].copy() if out.query_quant_len else None)
