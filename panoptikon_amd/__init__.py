"""panoptikon_amd — MI355X-native vector-similarity scan for Panoptikon.

The product is libpvs.so (hand-written HIP for gfx950 behind the C ABI of
include/pvs.h).  This package is the thin Python host mirror used by the tests,
bench.py and smoke(): it only marshals buffers; nothing here computes a distance.
"""
from ._lib import (AGG_AVG, AGG_MAX, AGG_MIN, AGG_NONE, COSINE, DEVICE, F16, F32, HOST, I8, INDEX_ANN, INDEX_AUTO,
                   INDEX_EXACT, INDEX_QUANT, L2, PvsError, debug_get, debug_set, lib)
from .index import DeviceBuffer, VectorIndex, absmax, device_count, microbench, quantize_int8
from .host import (aggregate, artifact_scale, coalesce_ranks, coalesce_values, sort_bounds, embedding_from_npy_bytes, extract_embeddings, merge_group_pages, merge_topk,
                   resolve_vector_quant, row_number, rrf_fuse, rrf_search, scale_artifact, scale_from_absmax)

from .rendezvous import LocalRendezvous
from .host import RrfCols
from .sharded import merge_shard_group_pages, merge_shard_pages, rrf_search_sharded, shard_range, shard_ranges_by_group

__all__ = [n for n in dir() if not n.startswith("_")]
