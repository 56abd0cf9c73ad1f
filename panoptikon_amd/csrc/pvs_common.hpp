// pvs_common.hpp — shared declarations of libpvs (host + device).
// MI355X / gfx950 only: no CUDA paths, no portability macros.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "pvs.h"

#define PVS_EXPORT extern "C" __attribute__((visibility("default")))

// ------------------------------------------------------------------ errors
pvs_status pvs_fail(pvs_status code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess)                                                                \
            return pvs_fail(_e == hipErrorOutOfMemory ? PVS_ERR_OOM : PVS_ERR_DEVICE,        \
                            "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                            __LINE__);                                                       \
    } while (0)

#define PVS_TRY(expr)                 \
    do {                              \
        pvs_status _s = (expr);       \
        if (_s != PVS_OK) return _s;  \
    } while (0)

// Scratch for the host-orchestrated paths (per-item search, RRF, similar_to): blocks are kept per device and size class after
// use instead of going back to hipMalloc / hipFree every call (each costs ~0.1-0.5 ms and hipFree synchronises the device:
// 6 of configs[4]'s 13.8 ms per query were exactly that).  pvs_index_destroy returns the device's idle blocks to the runtime.
// A block that queued work may still touch goes back with pvs_scratch_free_on(p, stream): it is handed out again only after
// that stream has passed the point of the call.  pvs_scratch_free(p) is for blocks whose work the caller has waited for.  Idle
// bytes per device are capped (least recently used blocks return to the runtime); pvs_malloc_retry is hipMalloc that empties
// the cache and retries before it reports out-of-memory — every device allocation of the library goes through it.
hipError_t pvs_scratch_alloc(void **out, size_t bytes);  // on the current device
void pvs_scratch_free_on(void *p, hipStream_t s, bool pending = true);
void pvs_scratch_free(void *p);
void pvs_scratch_trim(int device);
hipError_t pvs_malloc_retry(void **out, size_t bytes);

// Test / tuning knobs (pvs_debug_set in include/pvs.h).  Every knob is 0 in a process that never calls pvs_debug_set: the
// library reads no environment variable to decide which algorithm answers (round 3 had fourteen getenv switches here).
enum PvsDbg {
    PVS_DBG_SAMPLE_DIV = 0,        // pass A samples 1/value of the corpus (0: the built-in choice)
    PVS_DBG_SAMPLE_J_DIV,          // threshold = the (k / value)-th sample value (0: 4)
    PVS_DBG_NO_LIGHT_FINALIZE,     // multi-stream indexes keep the LDS-heavy pass C
    PVS_DBG_FORCE_LIGHT_FINALIZE,  // every search uses the LDS-light pass C
    PVS_DBG_DENSE_PER_QUERY,       // dense fallback: one query per corpus pass + full sort (the round-1 form)
    PVS_DBG_NO_DIRECT_SCORE,       // 1..4 int8 queries: matrix-core MODE 2 instead of k_score_i8_direct
    PVS_DBG_NO_PAGE_RANK,          // per-item search: always sort every group
    PVS_DBG_RRF_SERIAL,            // pvs_rrf_search: branches one after the other on the calling thread
    PVS_DBG_RRF_FULL,              // pvs_rrf_search: skip the bounded fusion, rank every group
    PVS_DBG_RRF_TRACE,             // pvs_rrf_search: phase wall times on stderr
    PVS_DBG_SCAN_NO_WIDE128,       // 128-query int8 passes stay on k_scan
    PVS_DBG_SCRATCH_IDLE_CAP_MB,   // idle scratch bytes kept per device (0: 16 GiB)
    PVS_DBG_SCRATCH_BYPASS,        // scratch blocks come from hipMalloc and go back with hipFree (race hunting)
    PVS_DBG_RRF_DIGEST,            // pvs_rrf_search records per-stage digests (pvs_debug_rrf_digests)
    PVS_DBG_NO_SPARSE,             // filtered searches never take the gather-score path
    PVS_DBG_SPARSE_MAX,            // ... take it up to this many allowed rows (0: the built-in crossover)
    PVS_DBG_NO_FUSED_AGG,          // per-item MAX/AVG/weighted: dense matrix + k_group_aggregate (the round-3 form)
    PVS_DBG_NO_SIDE_FINALIZE,      // pvs_search_device: pass C stays on the search's own stream
    PVS_DBG_RRF_HOST_ROUNDS,       // pvs_rrf_search: every round of the bounded fusion in its host form (the round-3 route)
    PVS_DBG_MULTI_HOST_PAGES,      // multi-device index: per-item pages merged on the host, device-space masks split on the host (the round-3 route)
    PVS_DBG_PRELUDE_STREAM,        // pvs_search_device: query prep, pass A and the k-th select on a stream of their own (measured slower: search_enqueue)
    PVS_DBG_DENSE_NQ4,             // k_dense_exact: at most 4 float queries per pass (the form before the packed 8-query instance)
    PVS_DBG_MARKER_EVENTS,         // filter-scan searches: profiling spans and the pass-B -> pass-C dependency as hipEventRecord markers around the kernels (the round-3 form) instead of events bound to the dispatches
    PVS_DBG_NO_DIRECT_TOPK,        // single queries always take the filter scan (never the one-launch exact search, pvs_direct.hip)
    PVS_DBG_DIRECT_MAX_MB,         // ... take the one-launch search up to this many MB of rows (0: the built-in crossover)
    PVS_DBG_DIRECT_QUERIES,        // (a counter, read with pvs_debug_get) single queries answered by the one-launch search, process-wide
    PVS_DBG_DIRECT_UNIT,           // one-launch search: 64-row pairs per work unit (0: >= 48 KB of rows)
    PVS_DBG_DIRECT_STATIC_PCT,     // ... share of a wave's units that is dealt instead of dequeued, in percent (0: 50; 100: round 4's dealing; -1: one dealt unit)
    PVS_DBG_DIRECT_MAX_NQ,         // ... at most this many queries per launch (0: what the instance table takes; 1: round 4's single-query form only)
    PVS_DBG_DENSE_FULL_SORT,       // dense path: always sort every row (the form before the page-first threshold, round 5)
    PVS_DBG_DENSE_PAGE_FIRST,      // (a counter) dense pages answered by the sampled threshold + a sort of the admitted rows
    PVS_DBG_NO_FLAG_POLL,          // pvs_search (one-launch route): wait for the stream's completion event instead of polling the kernel's flag words in pinned memory
    PVS_DBG_POLL_LATE_PAGES,       // (a counter) polled searches whose flag word reached host memory before every word of its page had
    PVS_DBG_NO_EXACT_WIDE,         // dense exact path, float rows: 8 queries per pass through LDS (k_dense_exact) also for 9+ queries
    PVS_DBG_NO_AGG8,               // per-item aggregation of a distance matrix: one thread per (group, column) also when the columns are a multiple of 8
    PVS_DBG_NO_DENSE2,             // dense exact path, 8 float queries: one row per lane (k_dense_exact) instead of two (k_dense_exact2)
    PVS_DBG_COMM_TIMEOUT_S,        // bound on every wait for the other ranks (communicator creation, a shard exchange): seconds (0: 180)
    PVS_DBG_COMM_FAIL_LOCAL,       // tests: the next pvs_search_sharded_async of this process fails locally before its exchange (value = how many)
    PVS_DBG_NO_FLOAT_CERTIFY,      // per-item pages over float rows: every distance exact (k_exact_wide) instead of bound + certify + rescan (pvs_items_float.hip)
    PVS_DBG_FLOAT_CERTIFY_QUERIES, // (a counter) per-item queries answered by the certified route
    PVS_DBG_FLOAT_CERTIFY_ROWS,    // (a counter) candidate rows its exact stage rescanned, summed over chunks
    PVS_DBG_FLOAT_CERTIFY_TRACE,   // the certified route reports its decisions on stderr (candidate rows, bad queries, who answered)
    PVS_DBG_FLOAT_CERTIFY_NO_FOLD, // the certified route writes the key matrix (k_scan MODE 4) and folds in a second kernel also when the files are runs
    PVS_DBG_COUNT
};
int64_t pvs_dbg(PvsDbg key);
void pvs_dbg_add(PvsDbg key, int64_t v);  // counters among the keys

// ------------------------------------------------------------ geometry
// Rows live in HBM at a pitch that is a multiple of 256 B so that a row is a
// whole number of 16-chunk (256 B) "k-slabs": the unit the scan kernel streams
// through LDS and XOR-swizzles.
constexpr uint32_t PVS_KSLAB_BYTES = 256;
constexpr uint32_t PVS_TILE_ROWS = 32;    // one MFMA 32x32 tile of rows
constexpr uint32_t PVS_ROW_ALIGN = 128;   // capacity granularity (largest WG tile: 4 row tiles)
constexpr uint32_t PVS_MAX_BATCH = 128;   // queries per dense / group pass (4 waves x 32)
constexpr uint32_t PVS_SCAN_MAX_BATCH = 256;  // queries per filter-scan pass (int8: 4 waves x 2 groups x 32); sizes the per-search buffers
constexpr uint32_t PVS_MAX_K = 4096;      // page size served by the filter path (the reference prefetches up to 4096 rows, api/search.rs:51)
constexpr uint32_t PVS_CAND_CAP = 16384;  // candidate slots per query
constexpr uint32_t PVS_SEG_CAP = 64;       // candidate slots per (segment, query): ~3 expected at k = 100 (1,600 candidates over >= 512 segments)
constexpr uint32_t PVS_SEG_PAIRS = 131072; // (segment, query) pairs per pass: 256 queries x 512 segments ... 32 queries x 4,096 segments
                                           // (segment = one half-wave of one workgroup row stream: its candidates are written by one lane)
// the 256-query int8 kernel (pvs_scan_wide.hpp): a lane quarter c of every workgroup stream owns a segment; a (lane, query)
// supplies up to 8 group minima in pass A
constexpr uint32_t PVS_WIDE_SEG_PER_STREAM = 4;
constexpr uint32_t PVS_WIDE_SEG_CAP = 32;   // slots per (segment, query): ~1.6 expected at k = 100 over 1,024 segments
constexpr uint32_t PVS_WIDE_GMIN_MAX = 8;
constexpr uint32_t PVS_AUX_REC = 64;     // floats per 32-row tile in the scan's row-scalar stream: 32 row scalars, min, max, padding
constexpr uint32_t PVS_SURV_CAP = 8192;   // survivors reranked exactly per query (their sort records overlay the 64 KiB bound array)

// Physical row layout in HBM ("tiled"): rows are grouped in tiles of 32; a tile is
// stored k-slab major — [slab = byte/256][row in tile][256 B] — and inside each 256-B segment
// the 16-byte chunks are XOR-swizzled with (row & 15).  This is exactly the LDS image the scan
// kernel wants, so one LDS-DMA instruction moves one contiguous KiB of HBM (measured +13 %
// over a row-major pitch) and lands conflict-free for ds_read_b128.
//   byte b of row r  ->  (r/32)*(32*stride) + (b/256)*8192 + (r%32)*256 + (((b%256)/16) ^ (r&15))*16 + b%16
__host__ __device__ static inline size_t pvs_chunk_off(uint64_t r, uint32_t chunk, uint32_t stride) {
    const uint32_t i = (uint32_t)(r & 31);
    return (size_t)(r >> 5) * (32u * (size_t)stride) + (size_t)(chunk >> 4) * 8192u + (size_t)i * 256u +
           (size_t)(((chunk & 15u) ^ (i & 15u)) << 4);
}

static inline uint32_t pvs_esz(uint32_t dtype) { return dtype == PVS_F32 ? 4u : dtype == PVS_F16 ? 2u : 1u; }
static inline uint64_t pvs_round_up(uint64_t v, uint64_t m) { return (v + m - 1) / m * m; }

// Per-query constants produced by the query-prep kernel and consumed by the
// scan epilogue / finaliser.  key = monotone surrogate of the reference distance
// (cosine: -dot/|a| ; L2: |a|^2 + |q|^2 - 2 dot); err = eA + eC*|a| + eR*|a|^2 is a
// rigorous bound on |key - key_ref| (HISTORY.md §4.2), so [key-err, key+err] brackets
// the value the reference ordering is monotone in.
struct QInfo {
    float bb;      // sum q_i^2, accumulated sequentially in f32 (the reference's bMag)
    float qn;      // sqrt(bb)
    float dscale;  // multiply the MFMA dot by this (undo the f16 power-of-two prescale)
    float eA, eC, eR;
    float pad0, pad1;
};

// Does this query make EVERY row's distance NULL?  bb = sum q_i^2 in f32.  Cosine: a zero query (0/0) or one with a NaN component
// (bb NaN); L2: a NaN component.  (bb = +inf — an inf component or an overflow — is not decidable from bb alone: not claimed.)
__host__ __device__ static inline bool pvs_query_all_null(int metric, float bb) { return bb != bb || (metric == PVS_COSINE && bb == 0.f); }

// ------------------------------------------------------ device helpers
#if defined(__HIPCC__)

// Order-preserving f32 -> u32 (ascending), NaN last.
__host__ __device__ static inline uint32_t f32_sort_key(float f) {
    uint32_t b = __builtin_bit_cast(uint32_t, f);
    if ((b & 0x7fffffffu) > 0x7f800000u) return 0xffffffffu;  // NaN (either sign) sorts last
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__host__ __device__ static inline float f32_from_sort_key(uint32_t k) {
    uint32_t b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    if (k == 0xffffffffu) b = 0x7fc00000u;
    return __builtin_bit_cast(float, b);
}

__device__ static inline float ld_elem_f32(const float *p, int i) { return p[i]; }

// IEEE binary16 -> f32 (exact) via the hardware converter.
__device__ static inline float h2f(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }

// Visits the components of stored row r in order, 16 bytes per load (tiled addressing).
// f(i, v): v is int for I8 rows, float (exactly widened) for F16/F32 rows.  Bytes past
// dim*esz inside the last chunk are zero padding and are not visited.
template <int DT, typename F>
__device__ static inline void row_foreach(const uint8_t *rows, uint32_t stride, uint64_t r, int dim, F &&f) {
    constexpr int PER = DT == PVS_I8 ? 16 : DT == PVS_F16 ? 8 : 4;
    constexpr int UN = 8;  // loads in flight per lane: the visit is a dependent chain, the loads are not
    const int nchunks = (dim + PER - 1) / PER;
    auto visit = [&](int c, const uint4 &v) {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < PER; j++) {
            const int i = c * PER + j;
            if (i < dim) {
                if constexpr (DT == PVS_I8)
                    f(i, (int)(int8_t)(w[j >> 2] >> ((j & 3) * 8)));
                else if constexpr (DT == PVS_F16)
                    f(i, h2f((uint16_t)(w[j >> 1] >> ((j & 1) * 16))));
                else
                    f(i, __builtin_bit_cast(float, w[j]));
            }
        }
    };
    int c = 0;
    for (; c + UN <= nchunks; c += UN) {
        uint4 v[UN];
#pragma unroll
        for (int u = 0; u < UN; u++) v[u] = *(const uint4 *)(rows + pvs_chunk_off(r, (uint32_t)(c + u), stride));
#pragma unroll
        for (int u = 0; u < UN; u++) visit(c + u, v[u]);
    }
    for (; c < nchunks; c++) visit(c, *(const uint4 *)(rows + pvs_chunk_off(r, (uint32_t)c, stride)));
}

// Sequential-f32 restatement of sqlite-vec's scalar kernels (oracle/pvs_oracle.c):
// one rounding per multiply and one per add (__fmul_rn/__fadd_rn forbid FMA
// contraction), components in order.  DT = dtype of the stored row.
//   I8 : query is int8 codes;  F16/F32 : query is f32.
template <int DT>
__device__ static inline float seq_dot(const uint8_t *rows, uint32_t stride, uint64_t r, const void *q, int dim) {
    float dot = 0.0f;
    if constexpr (DT == PVS_I8) {
        const int8_t *b = (const int8_t *)q;
        row_foreach<DT>(rows, stride, r, dim, [&](int i, int a) { dot = __fadd_rn(dot, (float)(a * (int)b[i])); });
    } else {
        const float *b = (const float *)q;
        row_foreach<DT>(rows, stride, r, dim, [&](int i, float a) { dot = __fadd_rn(dot, __fmul_rn(a, b[i])); });
    }
    return dot;
}

template <int DT>
__device__ static inline float seq_sumsq_diff(const uint8_t *rows, uint32_t stride, uint64_t r, const void *q, int dim) {
    float res = 0.0f;
    if constexpr (DT == PVS_I8) {
        const int8_t *b = (const int8_t *)q;
        row_foreach<DT>(rows, stride, r, dim, [&](int i, int a) {
            float t = (float)(a - (int)b[i]);
            res = __fadd_rn(res, __fmul_rn(t, t));
        });
    } else {
        const float *b = (const float *)q;
        row_foreach<DT>(rows, stride, r, dim, [&](int i, float a) {
            float t = __fsub_rn(a, b[i]);
            res = __fadd_rn(res, __fmul_rn(t, t));
        });
    }
    return res;
}

template <int DT>
__device__ static inline float seq_sumsq(const uint8_t *rows, uint32_t stride, uint64_t r, int dim) {
    float aa = 0.0f;
    if constexpr (DT == PVS_I8) {
        row_foreach<DT>(rows, stride, r, dim, [&](int, int a) { aa = __fadd_rn(aa, (float)(a * a)); });
    } else {
        row_foreach<DT>(rows, stride, r, dim, [&](int, float a) { aa = __fadd_rn(aa, __fmul_rn(a, a)); });
    }
    return aa;
}

// return sqrt(res): sqrt evaluated in double (IEEE, correctly rounded), narrowed to f32.
__device__ static inline float ref_l2_finish(float res) { return (float)__dsqrt_rn((double)res); }

// return (f32)(1 - dot / (sqrt(aa) * sqrt(bb))) evaluated in double.
__device__ static inline float ref_cosine_finish(float dot, float aa, float bb) {
    double den = __dsqrt_rn((double)aa) * __dsqrt_rn((double)bb);
    return (float)(1.0 - (double)dot / den);
}

template <int DT>
__device__ static inline float exact_distance(const uint8_t *rows, uint32_t stride, uint64_t r, const void *q, int dim, int metric,
                                              float aa, float bb) {
    if (metric == PVS_L2) return ref_l2_finish(seq_sumsq_diff<DT>(rows, stride, r, q, dim));
    return ref_cosine_finish(seq_dot<DT>(rows, stride, r, q, dim), aa, bb);
}

// sum of squares of a dense (untiled) vector, in order — the query side (bMag)
template <int DT>
__device__ static inline float seq_sumsq_dense(const void *vec, int dim) {
    float acc = 0.0f;
    if constexpr (DT == PVS_I8) {
        const int8_t *a = (const int8_t *)vec;
        for (int i = 0; i < dim; i++) acc = __fadd_rn(acc, (float)((int)a[i] * (int)a[i]));
    } else {
        const float *a = (const float *)vec;
        for (int i = 0; i < dim; i++) acc = __fadd_rn(acc, __fmul_rn(a[i], a[i]));
    }
    return acc;
}

#endif  // __HIPCC__
