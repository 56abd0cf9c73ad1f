"""VectorIndex: Python handle over pvs_index (HBM-resident corpus shard)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L

_NP = {L.F32: np.float32, L.F16: np.float16, L.I8: np.int8}


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def device_count() -> int:
    return int(L.lib().pvs_device_count())


class DeviceBuffer:
    """A raw HBM allocation owned through the C ABI (pvs_device_malloc)."""

    def __init__(self, nbytes: int, device: int = -1):
        self.device = device
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        L.check(L.lib().pvs_device_malloc(device, self.nbytes, C.byref(p)))
        self.ptr = p.value

    @classmethod
    def from_numpy(cls, a: np.ndarray, device: int = -1) -> "DeviceBuffer":
        a = np.ascontiguousarray(a)
        b = cls(max(a.nbytes, 16), device)
        if a.nbytes:
            L.check(L.lib().pvs_memcpy(b.ptr, _ptr(a), a.nbytes, device))
        return b

    def to_numpy(self, dtype, shape) -> np.ndarray:
        out = np.empty(shape, dtype)
        if out.nbytes:
            L.check(L.lib().pvs_memcpy(_ptr(out), self.ptr, out.nbytes, self.device))
        return out

    def free(self):
        if self.ptr:
            L.lib().pvs_device_free(self.device, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def microbench(device: int = -1) -> dict:
    """On-box peaks (pvs_microbench): HBM streaming reads / LDS-DMA reads / copy in GB/s, dense int8 and f16 MFMA rates."""
    r = L.MicrobenchResult()
    r.struct_size = C.sizeof(L.MicrobenchResult)
    L.check(L.lib().pvs_microbench(device, C.byref(r)))
    return {"hbm_read_GBs": round(r.hbm_read_gbs, 1), "hbm_lds_dma_read_GBs": round(r.hbm_lds_dma_gbs, 1),
            "hbm_copy_GBs_read_plus_write": round(r.hbm_copy_gbs, 1), "mfma_i8_TOPs": round(r.mfma_i8_tops, 1),
            "mfma_f16_TFLOPs": round(r.mfma_f16_tflops, 1), "compute_units": int(r.compute_units), "clock_mhz": int(r.clock_mhz),
            "how": "pvs_microbench on this device in this run (4 GiB streams, best of 5; MFMA: 4 independent accumulators, 2 waves/SIMD)"}


def absmax(x, device: int = -1) -> float:
    """blob_absmax over all components, on the GPU (db/vector_quants.rs:1474-1483)."""
    out = C.c_float()
    if isinstance(x, DeviceBuffer):
        L.check(L.lib().pvs_absmax(x.ptr, x.nbytes // 4, L.DEVICE, device, C.byref(out)))
    else:
        x = np.ascontiguousarray(x, np.float32)
        L.check(L.lib().pvs_absmax(_ptr(x), x.size, L.HOST, device, C.byref(out)))
    return float(out.value)


def quantize_int8(x, scale: float, device: int = -1) -> np.ndarray:
    """quantize_int8 on the GPU (db/vector_quants.rs:1489-1497)."""
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty(x.shape, np.int8)
    L.check(L.lib().pvs_quantize_i8(_ptr(x), x.size, np.float32(scale), _ptr(out), L.HOST, device))
    return out


class VectorIndex:
    def __init__(self, dtype: int, dim: int, device: int = -1, capacity_rows: int = 0, id_base: int = 0, devices=None):
        """devices: list of HIP ordinals -> one index sharded over several GPUs inside this process
        (pvs_index_desc.n_devices; buffers of the device entry points then live on devices[0])."""
        self.dtype, self.dim = dtype, dim
        self.devices = None if devices is None else [int(x) for x in devices]
        self.device = device if not self.devices else self.devices[0]
        arr = (C.c_int32 * len(self.devices))(*self.devices) if self.devices else None
        d = L.IndexDesc(C.sizeof(L.IndexDesc), self.device, dtype, dim, capacity_rows, id_base,
                        len(self.devices) if self.devices else 0, arr)
        h = C.c_void_p()
        L.check(L.lib().pvs_index_create(C.byref(d), C.byref(h)))
        self._h = h

    def close(self):
        if self._h:
            L.lib().pvs_index_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------ build side
    def set_scale(self, scale: float):
        L.check(L.lib().pvs_index_set_scale(self._h, np.float32(scale)))

    def set_scale_artifact(self, artifact: bytes):
        buf = (C.c_uint8 * max(len(artifact), 1)).from_buffer_copy(bytes(artifact) or b"\0")
        L.check(L.lib().pvs_index_set_scale_artifact(self._h, buf, len(artifact)))

    def add(self, rows, row_ids=None, group_ids=None):
        """rows: [n][dim] in the index dtype (numpy) or a DeviceBuffer of that layout."""
        ids = None if row_ids is None else np.ascontiguousarray(row_ids, np.int64)
        grp = None if group_ids is None else np.ascontiguousarray(group_ids, np.int64)
        if isinstance(rows, tuple):  # (DeviceBuffer, n)
            buf, n = rows
            L.check(L.lib().pvs_index_add(self._h, buf.ptr, n, _ptr(ids), _ptr(grp), L.DEVICE))
            return
        rows = np.ascontiguousarray(rows)
        if rows.dtype != _NP[self.dtype] or rows.ndim != 2 or rows.shape[1] != self.dim:
            raise ValueError(f"rows must be [{self.dim}]-wide {_NP[self.dtype]}")
        L.check(L.lib().pvs_index_add(self._h, _ptr(rows), rows.shape[0], _ptr(ids), _ptr(grp), L.HOST))

    def add_f32(self, rows, row_ids=None, group_ids=None):
        """f32 rows converted on the device to the index dtype (backfill_chunk's codec)."""
        ids = None if row_ids is None else np.ascontiguousarray(row_ids, np.int64)
        grp = None if group_ids is None else np.ascontiguousarray(group_ids, np.int64)
        if isinstance(rows, tuple):
            buf, n = rows
            L.check(L.lib().pvs_index_add_f32(self._h, buf.ptr, n, _ptr(ids), _ptr(grp), L.DEVICE))
            return
        rows = np.ascontiguousarray(rows, np.float32)
        if rows.ndim != 2 or rows.shape[1] != self.dim:
            raise ValueError("rows must be [n][dim]")
        L.check(L.lib().pvs_index_add_f32(self._h, _ptr(rows), rows.shape[0], _ptr(ids), _ptr(grp), L.HOST))

    def remove_rows(self, row_ids) -> int:
        """pvs_index_remove_rows: the rows with these ids leave the index (compacted on the device); returns how many went."""
        ids = np.ascontiguousarray(row_ids, np.int64).ravel()
        out = C.c_uint64(0)
        L.check(L.lib().pvs_index_remove_rows(self._h, _ptr(ids) if ids.size else None, int(ids.size), C.byref(out)))
        return int(out.value)

    def replace_rows(self, rows, row_ids):
        """pvs_index_replace_rows[_f32]: new vectors for rows the index already holds (f32 rows are converted like add_f32)."""
        ids = np.ascontiguousarray(row_ids, np.int64).ravel()
        if isinstance(rows, tuple):  # (DeviceBuffer, "f32" | "native"): rows already in HBM
            buf, kind = rows
            fn = L.lib().pvs_index_replace_rows_f32 if kind == "f32" else L.lib().pvs_index_replace_rows
            L.check(fn(self._h, buf.ptr, int(ids.size), _ptr(ids), L.DEVICE))
            return
        rows = np.ascontiguousarray(rows)
        if rows.ndim != 2 or rows.shape[1] != self.dim or rows.shape[0] != ids.size:
            raise ValueError("rows must be [len(row_ids)][dim]")
        if rows.dtype == np.float32 and self.dtype != L.F32:
            L.check(L.lib().pvs_index_replace_rows_f32(self._h, _ptr(rows), rows.shape[0], _ptr(ids), L.HOST))
        else:
            rows = np.ascontiguousarray(rows, _NP[self.dtype])
            L.check(L.lib().pvs_index_replace_rows(self._h, _ptr(rows), rows.shape[0], _ptr(ids), L.HOST))

    # ----------------------------------------------------------- query side
    def _queries(self, queries):
        q = np.ascontiguousarray(queries)
        if q.ndim == 1:
            q = q[None, :]
        if q.dtype == np.int8:
            qd = L.I8
        else:
            q = np.ascontiguousarray(q, np.float32)
            qd = L.F32
        if q.shape[1] != self.dim:
            # the reference's sqlite-vec raises a SQL error for mismatched lengths (db/pql.rs:18-21)
            raise L.PvsError(L.ERR_DIM_MISMATCH, f"query dimension {q.shape[1]} != index dimension {self.dim}")
        return q, qd

    def search(self, queries, k: int, metric: int = L.COSINE):
        q, qd = self._queries(queries)
        b = q.shape[0]
        ids = np.empty((b, max(k, 1)), np.int64)
        dist = np.empty((b, max(k, 1)), np.float32)
        cnt = np.zeros(b, np.uint32)
        L.check(L.lib().pvs_search(self._h, _ptr(q), qd, b, k, metric, _ptr(ids), _ptr(dist), _ptr(cnt)))
        return ids, dist, cnt

    def search_page(self, queries, offset: int, limit: int, metric: int = L.COSINE):
        """Entries [offset, offset + limit) of the search ordering (LIMIT ? OFFSET ?, pql/builder.rs:578-582)."""
        q, qd = self._queries(queries)
        b = q.shape[0]
        ids = np.empty((b, max(limit, 1)), np.int64)
        dist = np.empty((b, max(limit, 1)), np.float32)
        cnt = np.zeros(b, np.uint32)
        L.check(L.lib().pvs_search_page(self._h, _ptr(q), qd, b, int(offset), int(limit), metric, _ptr(ids), _ptr(dist), _ptr(cnt)))
        return ids, dist, cnt

    def search_groups_page(self, queries, offset: int, limit: int, metric: int = L.COSINE, agg: int = L.AGG_MIN, row_weights=None):
        q, qd = self._queries(queries)
        b = q.shape[0]
        w = None if row_weights is None else np.ascontiguousarray(row_weights, np.float32)
        og = np.empty((b, max(limit, 1)), np.int64)
        ov = np.empty((b, max(limit, 1)), np.float64)
        oc = np.zeros(b, np.uint32)
        L.check(L.lib().pvs_search_groups_page(self._h, _ptr(q), qd, b, int(offset), int(limit), metric, agg, _ptr(w), _ptr(og), _ptr(ov), _ptr(oc)))
        return og, ov, oc

    def search_bounded(self, queries, k: int, metric: int = L.COSINE, gt=None, lt=None):
        """pvs_search restricted to rows with gt < distance < lt (apply_sort_bounds, builder.rs:781-815)."""
        q, qd = self._queries(queries)
        b = q.shape[0]
        ids = np.full((b, k), -1, np.int64)
        dist = np.full((b, k), np.nan, np.float32)
        cnt = np.zeros(b, np.uint32)
        L.check(L.lib().pvs_search_bounded(self._h, _ptr(q), qd, b, k, metric, int(gt is not None), float(gt or 0.0), int(lt is not None),
                                           float(lt or 0.0), _ptr(ids), _ptr(dist), _ptr(cnt)))
        return ids, dist, cnt

    def search_filtered(self, queries, k: int, allowed_rows, metric: int = L.COSINE):
        """pvs_search over the rows whose byte in `allowed_rows` ([rows] uint8/bool, row order) is non-zero."""
        q, qd = self._queries(queries)
        m = np.ascontiguousarray(allowed_rows).astype(np.uint8, copy=False)
        if m.shape != (self.stats().rows,):
            raise ValueError("allowed_rows must have one entry per stored row")
        b = q.shape[0]
        ids = np.full((b, k), -1, np.int64)
        dist = np.full((b, k), np.nan, np.float32)
        cnt = np.zeros(b, np.uint32)
        L.check(L.lib().pvs_search_filtered(self._h, _ptr(q), qd, b, k, metric, _ptr(m), L.HOST, _ptr(ids), _ptr(dist), _ptr(cnt)))
        return ids, dist, cnt

    def search_rows(self, queries, k: int, rows, metric: int = L.COSINE):
        """pvs_search_rows: the page over an explicit candidate set, `rows` = strictly ascending row positions (uint32) or a
        DeviceBuffer / (DeviceBuffer, count) holding them in HBM."""
        q, qd = self._queries(queries)
        b = q.shape[0]
        ids = np.full((b, k), -1, np.int64)
        dist = np.full((b, k), np.nan, np.float32)
        cnt = np.zeros(b, np.uint32)
        if isinstance(rows, tuple):
            rp, n, space = rows[0].ptr, int(rows[1]), L.DEVICE
        else:
            r = np.ascontiguousarray(rows, dtype=np.uint32)
            rp, n, space = (_ptr(r) if r.size else None), int(r.size), L.HOST
        L.check(L.lib().pvs_search_rows(self._h, _ptr(q), qd, b, k, metric, rp, n, space, _ptr(ids), _ptr(dist), _ptr(cnt)))
        return ids, dist, cnt

    def search_device(self, d_queries: DeviceBuffer, qdtype: int, batch: int, k: int, metric: int,
                      d_ids: DeviceBuffer, d_dist: DeviceBuffer, d_cnt: DeviceBuffer) -> int:
        t = C.c_uint32()
        L.check(L.lib().pvs_search_device(self._h, d_queries.ptr, qdtype, batch, k, metric, d_ids.ptr, d_dist.ptr,
                                          d_cnt.ptr, C.byref(t)))
        return int(t.value)

    def wait(self, ticket: int):
        L.check(L.lib().pvs_wait(self._h, ticket))

    def sync(self):
        L.check(L.lib().pvs_sync(self._h))

    def set_streams(self, n: int):
        L.check(L.lib().pvs_index_set_streams(self._h, n))

    def set_coalescing(self, window_us: int, max_batch: int = 0):
        """Concurrent `search` calls from several host threads within `window_us` share one corpus pass (pvs.h)."""
        L.check(L.lib().pvs_index_set_coalescing(self._h, int(window_us), int(max_batch)))

    def coalescing_stats(self):
        calls, passes = C.c_uint64(), C.c_uint64()
        L.check(L.lib().pvs_index_coalescing_stats(self._h, C.byref(calls), C.byref(passes)))
        return int(calls.value), int(passes.value)

    def set_path(self, path: int):
        L.check(L.lib().pvs_index_set_path(self._h, path))

    def set_order_keys(self, keys) -> None:
        """One int64 per stored row (e.g. last_modified): rows that tie on the distance come out by key DESC, then id.  None removes them."""
        if keys is None:
            L.check(L.lib().pvs_index_set_order_keys(self._h, None, 0, L.HOST))
            return
        k = np.ascontiguousarray(keys, np.int64)
        L.check(L.lib().pvs_index_set_order_keys(self._h, _ptr(k), k.size, L.HOST))

    def scan_kernel_name(self, batch: int) -> str:
        buf = C.create_string_buffer(128)
        L.check(L.lib().pvs_index_scan_kernel_name(self._h, batch, buf, 128))
        return buf.value.decode()

    def score_all(self, query, metric: int = L.COSINE) -> np.ndarray:
        q, qd = self._queries(query)
        out = np.empty(self.stats().rows, np.float32)
        L.check(L.lib().pvs_score_all(self._h, _ptr(q), qd, metric, _ptr(out), L.HOST))
        return out

    def score_batch(self, queries, metric: int = L.COSINE) -> np.ndarray:
        """[rows][batch] exact distances (the dist_{cte}.d column for every query of the batch)."""
        q, qd = self._queries(queries)
        out = np.empty((self.stats().rows, q.shape[0]), np.float32)
        L.check(L.lib().pvs_score_batch(self._h, _ptr(q), qd, q.shape[0], metric, _ptr(out), L.HOST))
        return out

    def search_groups(self, queries, k: int, metric: int = L.COSINE, agg: int = L.AGG_MIN, row_weights=None):
        """Per-item page: (group ids [b][k], f64 aggregate [b][k], counts [b])."""
        q, qd = self._queries(queries)
        b = q.shape[0]
        w = None if row_weights is None else np.ascontiguousarray(row_weights, np.float32)
        og = np.empty((b, k), np.int64)
        ov = np.empty((b, k), np.float64)
        oc = np.zeros(b, np.uint32)
        L.check(L.lib().pvs_search_groups(self._h, _ptr(q), qd, b, k, metric, agg, _ptr(w), _ptr(og), _ptr(ov), _ptr(oc)))
        return og, ov, oc

    def search_groups_filtered(self, queries, k: int, allowed_rows, metric: int = L.COSINE, agg: int = L.AGG_MIN, row_weights=None):
        """search_groups over the rows whose byte in `allowed_rows` is non-zero; groups without such a row are absent."""
        q, qd = self._queries(queries)
        b = q.shape[0]
        m = np.ascontiguousarray(allowed_rows).astype(np.uint8, copy=False)
        w = None if row_weights is None else np.ascontiguousarray(row_weights, np.float32)
        og = np.empty((b, k), np.int64)
        ov = np.empty((b, k), np.float64)
        oc = np.zeros(b, np.uint32)
        L.check(L.lib().pvs_search_groups_filtered(self._h, _ptr(q), qd, b, k, metric, agg, _ptr(w), _ptr(m), L.HOST, _ptr(og), _ptr(ov),
                                                   _ptr(oc)))
        return og, ov, oc

    def similar_to(self, target_row_ids, k: int, metric: int = L.L2, agg: int = L.AGG_AVG):
        """filters/item_similarity.rs: the target item's stored vectors against everything else."""
        t = np.ascontiguousarray(target_row_ids, np.int64)
        og = np.empty(k, np.int64)
        ov = np.empty(k, np.float64)
        oc = C.c_uint32()
        L.check(L.lib().pvs_similar_to(self._h, _ptr(t), t.size, k, metric, agg, _ptr(og), _ptr(ov), C.byref(oc)))
        return og[: oc.value], ov[: oc.value]

    def similar_to_ex(self, target_row_ids, k: int, metric: int = L.L2, agg: int = L.AGG_AVG, confidence=None,
                      language_confidence=None, confidence_weight: float = 0.0, language_confidence_weight: float = 0.0,
                      row_kind=None, xmodal_i2i: bool = True, xmodal_t2t: bool = True):
        """similar_to with the text source's confidence weights and the CLIP cross-modal gates
        (item_similarity.rs:473-581); confidence arrays: one f64 per stored row (NaN = NULL); row_kind: 0 = clip,
        1 = text-embedding per stored row."""
        t = np.ascontiguousarray(target_row_ids, np.int64)
        cf = None if confidence is None else np.ascontiguousarray(confidence, np.float64)
        lg = None if language_confidence is None else np.ascontiguousarray(language_confidence, np.float64)
        kd = None if row_kind is None else np.ascontiguousarray(row_kind, np.uint8)
        addr = lambda a: None if a is None else a.ctypes.data  # noqa: E731
        o = L.SimilarOpts(C.sizeof(L.SimilarOpts), agg, addr(cf), addr(lg), float(confidence_weight), float(language_confidence_weight),
                          addr(kd), int(bool(xmodal_i2i)), int(bool(xmodal_t2t)))
        og = np.empty(k, np.int64)
        ov = np.empty(k, np.float64)
        oc = C.c_uint32()
        L.check(L.lib().pvs_similar_to_ex(self._h, _ptr(t), t.size, k, metric, C.byref(o), _ptr(og), _ptr(ov), C.byref(oc)))
        return og[: oc.value], ov[: oc.value]

    def similar_to_weighted(self, target_row_ids, k: int, metric: int = L.L2, agg: int = L.AGG_AVG, confidence=None,
                            language_confidence=None, confidence_weight: float = 0.0, language_confidence_weight: float = 0.0):
        return self.similar_to_ex(target_row_ids, k, metric, agg, confidence, language_confidence, confidence_weight,
                                  language_confidence_weight)

    def read_ids(self, row0: int = 0, n: int | None = None, groups: bool = False):
        """row ids (and group ids) of rows [row0, row0+n) in row order"""
        n = self.stats().rows - row0 if n is None else n
        ids = np.empty(n, np.int64)
        grp = np.empty(n, np.int64) if groups else None
        L.check(L.lib().pvs_index_read_ids(self._h, row0, n, _ptr(ids), _ptr(grp)))
        return (ids, grp) if groups else ids

    def read_rows(self, row0: int, n: int) -> np.ndarray:
        out = np.empty((n, self.dim), _NP[self.dtype])
        L.check(L.lib().pvs_index_read_rows(self._h, row0, n, _ptr(out)))
        return out

    def set_profiling(self, enable: bool):
        L.check(L.lib().pvs_index_set_profiling(self._h, 1 if enable else 0))

    def profile(self, reset: bool = False) -> L.Profile:
        p = L.Profile()
        L.check(L.lib().pvs_index_get_profile(self._h, C.byref(p), 1 if reset else 0))
        return p

    def stats(self) -> L.Stats:
        s = L.Stats()
        L.check(L.lib().pvs_index_stats_ex(self._h, C.byref(s), C.sizeof(L.Stats)))
        return s
