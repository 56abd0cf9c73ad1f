"""ctypes binding of libpvs.so (the C ABI in include/pvs.h).

The library is the product: if it is missing or cannot be loaded this module
raises — there is no Python or CPU fallback for any distance computation.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libpvs.so")

OK = 0
ERR_INVALID_ARG, ERR_DIM_MISMATCH, ERR_DEVICE, ERR_OOM, ERR_STATE, ERR_UNSUPPORTED, ERR_PARSE, ERR_NOT_READY, ERR_COMM = range(1, 10)
F32, F16, I8 = 0, 1, 2
COSINE, L2 = 0, 1
AGG_NONE, AGG_MIN, AGG_MAX, AGG_AVG = 0, 1, 2, 3
HOST, DEVICE = 0, 1
INDEX_AUTO, INDEX_EXACT, INDEX_QUANT, INDEX_ANN = 0, 1, 2, 3
UNIQUE_ID_BYTES = 128


class IndexDesc(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("dtype", C.c_uint32), ("dim", C.c_uint32),
                ("capacity_rows", C.c_uint64), ("id_base", C.c_int64),
                ("n_devices", C.c_uint32), ("devices", C.POINTER(C.c_int32))]  # ABI v2: one process, several GPUs


class RrfBranch(C.Structure):
    _fields_ = [("idx", C.c_void_p), ("query", C.c_void_p), ("query_dtype", C.c_int32), ("metric", C.c_int32), ("agg", C.c_int32),
                ("row_weights", C.c_void_p), ("row_n_descending", C.c_int32), ("rrf_k", C.c_int32), ("weight", C.c_double)]


class SimilarOpts(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("agg", C.c_int32), ("row_confidence", C.c_void_p), ("row_language_confidence", C.c_void_p),
                ("confidence_weight", C.c_double), ("language_confidence_weight", C.c_double), ("row_kind", C.c_void_p),
                ("xmodal_i2i", C.c_uint32), ("xmodal_t2t", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("dtype", C.c_uint32), ("dim", C.c_uint32), ("rows", C.c_uint64),
                ("capacity_rows", C.c_uint64), ("row_stride_bytes", C.c_uint64), ("hbm_bytes", C.c_uint64),
                ("scale", C.c_float), ("searches", C.c_uint64), ("fast_queries", C.c_uint64),
                ("dense_queries", C.c_uint64), ("last_candidates", C.c_uint64), ("rescanned_queries", C.c_uint64),
                ("sparse_queries", C.c_uint64), ("null_tail_queries", C.c_uint64)]


class Profile(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("scan_launches", C.c_uint64), ("scan_ms", C.c_double),
                ("scan_rows", C.c_uint64), ("sample_launches", C.c_uint64), ("sample_ms", C.c_double),
                ("finalize_launches", C.c_uint64), ("finalize_ms", C.c_double),
                ("exchange_launches", C.c_uint64), ("exchange_ms", C.c_double)]


class MicrobenchResult(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("compute_units", C.c_uint32), ("clock_mhz", C.c_uint32), ("hbm_read_gbs", C.c_double),
                ("hbm_lds_dma_gbs", C.c_double), ("hbm_copy_gbs", C.c_double), ("mfma_i8_tops", C.c_double), ("mfma_f16_tflops", C.c_double)]


class ReadyPair(C.Structure):
    _fields_ = [("have_db_context", C.c_int32), ("have_default_profile", C.c_int32), ("pair_ready", C.c_int32),
                ("profile_id", C.c_int64), ("scale", C.c_float), ("dim", C.c_int64)]


class QuantResolved(C.Structure):
    _fields_ = [("use_quant", C.c_int32), ("profile_id", C.c_int64), ("query_quant_len", C.c_uint64)]


class PvsError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"pvs status {status}: {message}")
        self.status = status
        self.message = message


# every symbol include/pvs.h declares: name -> (restype, argtypes)
_vp, _sz, _u32, _u64, _i32, _i64, _f = C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint64, C.c_int32, C.c_int64, C.c_float
SYMBOLS = {
    "pvs_abi_version": (_u32, []),
    "pvs_last_error": (C.c_char_p, []),
    "pvs_device_count": (_i32, []),
    "pvs_index_create": (_i32, [C.POINTER(IndexDesc), C.POINTER(_vp)]),
    "pvs_index_destroy": (None, [_vp]),
    "pvs_index_add": (_i32, [_vp, _vp, _u64, _vp, _vp, _i32]),
    "pvs_index_add_f32": (_i32, [_vp, _vp, _u64, _vp, _vp, _i32]),
    "pvs_index_remove_rows": (_i32, [_vp, _vp, _u64, C.POINTER(_u64)]),
    "pvs_index_replace_rows": (_i32, [_vp, _vp, _u64, _vp, _i32]),
    "pvs_index_replace_rows_f32": (_i32, [_vp, _vp, _u64, _vp, _i32]),
    "pvs_index_set_scale_artifact": (_i32, [_vp, _vp, _sz]),
    "pvs_index_set_scale": (_i32, [_vp, _f]),
    "pvs_index_stats": (_i32, [_vp, C.POINTER(Stats)]),
    "pvs_index_stats_ex": (_i32, [_vp, C.POINTER(Stats), _sz]),
    "pvs_index_read_rows": (_i32, [_vp, _u64, _u64, _vp]),
    "pvs_index_read_ids": (_i32, [_vp, _u64, _u64, _vp, _vp]),
    "pvs_index_set_profiling": (_i32, [_vp, _i32]),
    "pvs_index_get_profile": (_i32, [_vp, C.POINTER(Profile), _i32]),
    "pvs_search": (_i32, [_vp, _vp, _i32, _u32, _u32, _i32, _vp, _vp, _vp]),
    "pvs_search_page": (_i32, [_vp, _vp, _i32, _u32, C.c_uint64, _u32, _i32, _vp, _vp, _vp]),
    "pvs_search_groups_page": (_i32, [_vp, _vp, _i32, _u32, C.c_uint64, _u32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "pvs_index_set_coalescing": (_i32, [_vp, _u32, _u32]),
    "pvs_index_coalescing_stats": (_i32, [_vp, _vp, _vp]),
    "pvs_search_bounded": (_i32, [_vp, _vp, _i32, _u32, _u32, _i32, _i32, C.c_double, _i32, C.c_double, _vp, _vp, _vp]),
    "pvs_search_device": (_i32, [_vp, _vp, _i32, _u32, _u32, _i32, _vp, _vp, _vp, C.POINTER(_u32)]),
    "pvs_wait": (_i32, [_vp, _u32]),
    "pvs_sync": (_i32, [_vp]),
    "pvs_index_set_streams": (_i32, [_vp, _u32]),
    "pvs_index_set_path": (_i32, [_vp, _u32]),
    "pvs_index_set_order_keys": (_i32, [_vp, _vp, _u64, _i32]),
    "pvs_score_column_create": (_i32, [_vp, _vp, _i32, _i32, C.POINTER(_vp)]),
    "pvs_score_column_rows": (_i32, [_vp, C.POINTER(_u64)]),
    "pvs_score_column_read": (_i32, [_vp, _u64, _u64, _vp]),
    "pvs_score_column_destroy": (None, [_vp]),
    "pvs_index_scan_kernel_name": (_i32, [_vp, _u32, C.c_char_p, _u32]),
    "pvs_score_all": (_i32, [_vp, _vp, _i32, _i32, _vp, _i32]),
    "pvs_score_batch": (_i32, [_vp, _vp, _i32, _u32, _i32, _vp, _i32]),
    "pvs_search_groups": (_i32, [_vp, _vp, _i32, _u32, _u32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "pvs_similar_to": (_i32, [_vp, _vp, _u32, _u32, _i32, _i32, _vp, _vp, _vp]),
    "pvs_search_groups_sharded": (_i32, [_vp, _vp, _vp, _i32, _u32, _u32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "pvs_merge_group_pages": (_i32, [_vp, _vp, _vp, _u32, _u32, _u32, _vp, _vp, _vp]),
    "pvs_merge_group_pages_keyed": (_i32, [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _vp, _vp, _vp]),
    "pvs_search_groups_filtered": (_i32, [_vp, _vp, _i32, _u32, _u32, _i32, _i32, _vp, _vp, _i32, _vp, _vp, _vp]),
    "pvs_search_filtered": (_i32, [_vp, _vp, _i32, _u32, _u32, _i32, _vp, _i32, _vp, _vp, _vp]),
    "pvs_search_rows": (_i32, [_vp, _vp, _i32, _u32, _u32, _i32, _vp, _u64, _i32, _vp, _vp, _vp]),
    "pvs_device_synchronize": (_i32, [_i32]),
    "pvs_device_mem_info": (_i32, [_i32, _vp, _vp]),
    "pvs_rrf_search": (_i32, [_vp, _u32, _u32, _vp, _vp, _vp]),
    "pvs_rrf_last_path": (_i32, []),
    "pvs_rrf_search_sharded": (_i32, [C.POINTER(RrfBranch), _u32, _u32, _vp, _u32, _vp, _vp, _vp, _vp, _vp]),
    "pvs_rrf_cols_create": (_i32, [C.POINTER(RrfBranch), C.POINTER(_vp)]),
    "pvs_rrf_cols_destroy": (None, [_vp]),
    "pvs_rrf_cols_groups": (_i32, [_vp, C.POINTER(_u64)]),
    "pvs_rrf_cols_threshold": (_i32, [_vp, _u64, C.POINTER(_u64)]),
    "pvs_rrf_cols_page": (_i32, [_vp, _u64, _u32, _vp, _vp, C.POINTER(_u32)]),
    "pvs_rrf_cols_lookup": (_i32, [_vp, _vp, _u32, _vp, _vp]),
    "pvs_rrf_cols_count_below": (_i32, [_vp, _vp, _vp, _u32, _vp]),
    "pvs_similar_to_ex": (_i32, [_vp, _vp, _u32, _u32, _i32, _vp, _vp, _vp, _vp]),
    "pvs_aggregate": (_i32, [_vp, _vp, _vp, _u64, _i32, _vp, _vp, C.POINTER(_u64)]),
    "pvs_absmax": (_i32, [_vp, _u64, _i32, _i32, C.POINTER(_f)]),
    "pvs_quantize_i8": (_i32, [_vp, _u64, _f, _vp, _i32, _i32]),
    "pvs_scale_from_absmax": (_f, [_f]),
    "pvs_scale_artifact": (None, [_f, _vp]),
    "pvs_artifact_scale": (_i32, [_vp, _sz, C.POINTER(_f)]),
    "pvs_row_number": (_i32, [_vp, _vp, _u64, _vp]),
    "pvs_row_number_dir": (_i32, [_vp, _vp, _u64, _i32, _vp]),
    "pvs_rrf_fuse": (_i32, [_vp, _u32, _u64, _vp, _vp, _vp]),
    "pvs_coalesce_ranks": (_i32, [_vp, _u32, _u64, _i32, _vp]),
    "pvs_coalesce_values": (_i32, [_vp, _u32, _u64, _i32, _vp]),
    "pvs_sort_bounds": (_i32, [_vp, _u64, _i32, C.c_double, _i32, C.c_double, _vp]),
    "pvs_npy_to_f32": (_i32, [_vp, _sz, _vp, _sz, C.POINTER(_sz)]),
    "pvs_resolve_vector_quant": (_i32, [_i32, C.c_char_p, _i64, C.POINTER(ReadyPair), _vp, _sz, _vp, _sz,
                                        C.POINTER(QuantResolved)]),
    "pvs_comm_unique_id": (_i32, [_vp]),
    "pvs_comm_create": (_i32, [_vp, _i32, _i32, _i32, C.POINTER(_vp)]),
    "pvs_comm_allreduce_max_f32": (_i32, [_vp, C.POINTER(C.c_float)]),
    "pvs_comm_destroy": (None, [_vp]),
    "pvs_search_sharded": (_i32, [_vp, _vp, _vp, _i32, _u32, _u32, _i32, _vp, _vp, _vp]),
    "pvs_search_sharded_async": (_i32, [_vp, _vp, _vp, _i32, _u32, _u32, _i32, _vp, _vp, _vp, C.POINTER(_u32)]),
    "pvs_merge_topk": (_i32, [_vp, _vp, _vp, _u32, _u32, _u32, _vp, _vp, _vp]),
    "pvs_merge_topk_device": (_i32, [_i32, _vp, _vp, _vp, _u32, _u32, _u32, _vp, _vp, _vp]),
    "pvs_merge_topk_keyed": (_i32, [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _vp, _vp, _vp]),
    "pvs_merge_topk_keyed_device": (_i32, [_i32, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _vp, _vp, _vp]),
    "pvs_device_malloc": (_i32, [_i32, _sz, C.POINTER(_vp)]),
    "pvs_device_free": (_i32, [_i32, _vp]),
    "pvs_memcpy": (_i32, [_vp, _vp, _sz, _i32]),
    "pvs_synth_rows_f32": (_i32, [_i32, _u64, _u64, _u64, _u32, _vp]),
    "pvs_synth_rows_clustered_f32": (_i32, [_i32, _u64, _u64, _u64, _u32, _vp]),
    "pvs_microbench": (_i32, [_i32, C.POINTER(MicrobenchResult)]),
    "pvs_debug_set": (_i32, [C.c_char_p, _i64]),
    "pvs_debug_get": (_i32, [C.c_char_p, C.POINTER(_i64)]),
    "pvs_debug_rrf_digests": (_i32, [_vp, C.POINTER(_u32), C.POINTER(_i32)]),
}

_lib = None


def lib() -> C.CDLL:
    """Loads libpvs.so; raises if it was not built (run `python -m panoptikon_amd.build`)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `python -m panoptikon_amd.build` "
                              "(there is no fallback implementation)")
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def debug_set(key: str, value: int = 1) -> None:
    """Test / tuning knob of the library (include/pvs.h, "test and tuning hooks"); 0 restores the product behaviour."""
    check(lib().pvs_debug_set(key.encode(), int(value)))


def debug_get(key: str) -> int:
    v = _i64()
    check(lib().pvs_debug_get(key.encode(), C.byref(v)))
    return v.value


def check(status: int) -> None:
    if status != OK:
        msg = lib().pvs_last_error()
        raise PvsError(status, msg.decode("utf-8", "replace") if msg else "")
