// pvs_kernels_scan.hip — dispatch of the filter-scan kernel (pvs_scan_kernel.hpp, instantiated
// in the pvs_scan_*.hip units), the radix k-th select, the exact finaliser (pass C) and the
// shard-page merge.  gfx950.
#include <cstdlib>

#include <hip/hip_ext.h>
#include "pvs_kernels.hpp"
#include "pvs_rerank.hpp"
#include "pvs_scan_dispatch.hpp"

bool pvs_scan_supported(int dtype, uint32_t kslabs) {
    if (dtype == PVS_I8) return (kslabs >= 1 && kslabs <= 6) || kslabs == 8 || kslabs == 12;
    if (dtype == PVS_F16)
        return kslabs == 1 || kslabs == 2 || kslabs == 3 || kslabs == 4 || kslabs == 6 || kslabs == 8 || kslabs == 9 || kslabs == 10 || kslabs == 12;
    if (dtype == PVS_F32)
        return kslabs == 1 || kslabs == 2 || kslabs == 3 || kslabs == 4 || kslabs == 6 || kslabs == 8 || kslabs == 12 || kslabs == 16 || kslabs == 18 ||
               kslabs == 20 || kslabs == 24;
    return false;
}
// Which kernel serves the filter passes (modes 0 and 1) of a shape: k_scan_wide (pvs_scan_wide.hpp: int8, 256 queries at row
// pitches up to 1 KiB, 128 queries up to 768 B) or k_scan.  pvs_debug_set("scan_no_wide128", 1) keeps the 128-query passes on k_scan (A/B
// timing of the two kernels; both are product paths, the GPU suite runs under either).
bool pvs_scan_is_wide(int dtype, uint32_t qgroups, uint32_t kslabs) {
    const bool no128 = pvs_dbg(PVS_DBG_SCAN_NO_WIDE128) != 0;
    if (dtype != PVS_I8 || (qgroups == 4 && no128)) return false;
    return pvs_scan_wide_serves(qgroups, kslabs, 1);
}
uint32_t pvs_scan_wg_rows(int dtype, uint32_t qgroups, uint32_t kslabs) {
    if (pvs_scan_is_wide(dtype, qgroups, kslabs)) return pvs_scan_wide_rows(kslabs);
    return qgroups >= 4 ? 32u : 32u * (4u / qgroups);
}
uint32_t pvs_scan_row_tiles(uint32_t qgroups) { return qgroups >= 4 ? 1u : 4u / qgroups; }
uint32_t pvs_scan_segs_per_stream(int dtype, uint32_t qgroups, uint32_t kslabs) {
    return pvs_scan_is_wide(dtype, qgroups, kslabs) ? pvs_scan_wide_segs(qgroups) : pvs_scan_row_tiles(qgroups) * 2u;
}
uint32_t pvs_scan_seg_cap(int dtype, uint32_t qgroups, uint32_t kslabs) { return pvs_scan_is_wide(dtype, qgroups, kslabs) ? PVS_WIDE_SEG_CAP : PVS_SEG_CAP; }
uint32_t pvs_scan_gmin_max(int dtype, uint32_t qgroups, uint32_t kslabs) { return pvs_scan_is_wide(dtype, qgroups, kslabs) ? PVS_WIDE_GMIN_MAX : 16u; }
uint32_t pvs_scan_wg_per_cu(int dtype, uint32_t qgroups, uint32_t kslabs) {
    if (pvs_scan_is_wide(dtype, qgroups, kslabs)) return 1u;
    return (qgroups == 1 || kslabs > 4) ? 1u : 2u;
}
uint32_t pvs_scan_max_batch(int dtype, uint32_t kslabs) { return dtype == PVS_I8 && kslabs >= 1 && kslabs <= 4 ? 256u : 128u; }  // (8-wave instances: pitch <= 1 KiB)

// MODE 5 (brackets folded in the epilogue) exists for the pitches whose query fragments leave room for 64 more registers per lane:
// f16 up to 2,048-B rows (1,024-d), f32 up to 4,096-B rows (1,024-d)
bool pvs_scan_fold5_supported(int dtype, uint32_t kslabs) {
    if (dtype == PVS_I8 || !pvs_scan_supported(dtype, kslabs)) return false;
    return kslabs * (dtype == PVS_F32 ? 4u : 8u) <= 64u;
}

hipError_t pvs_launch_scan(const ScanArgs &a, hipStream_t s) {
    ScanK k;
    k.rows = a.rows;
    k.aux = a.aux;
    k.qmat = a.qmat;
    k.qinfo = a.qinfo;
    k.thr = a.thr;
    k.gmin = a.gmin;
    k.seg = a.seg;
    k.seg_cnt = a.seg_cnt;
    k.seg_queries = a.qgroups * 32;
    k.flat = a.flat;
    k.flat_cnt = a.flat_cnt;
    k.flat_cap = a.flat_cap;
    const bool wide = (a.mode == 0 || a.mode == 1) && pvs_scan_is_wide(a.dtype, a.qgroups, a.kslabs);
    k.seg_cap = wide ? PVS_WIDE_SEG_CAP : PVS_SEG_CAP;
    k.seg_stride = a.grid * (wide ? pvs_scan_wide_segs(a.qgroups) : pvs_scan_row_tiles(a.qgroups) * 2u);
    k.n_rows = a.n_rows;
    k.stride = a.stride;
    const uint32_t wg_rows = wide ? pvs_scan_wide_rows(a.kslabs) : (a.qgroups >= 4 ? 32u : 32u * (4u / a.qgroups));
    k.n_wgtiles = (uint32_t)((a.n_rows + wg_rows - 1) / wg_rows);
    k.tile_step = a.tile_step ? a.tile_step : 1;
    k.groups_per_query = a.groups_per_query;
    k.gmin_per_lane = a.gmin_per_lane ? a.gmin_per_lane : 16;
    k.grid = a.grid;
    k.dense_out = a.dense_out;
    k.dense_flag = a.dense_flag;
    k.dense_ld = a.dense_ld;
    k.batch = a.batch;
    k.tile_grp = a.tile_grp;
    k.fold_weights = a.fold_weights;
    k.fold_mask = a.fold_mask;
    k.fold_out = a.fold_out;
    k.fold_ld = a.fold_ld;
    k.fold_agg = a.fold_agg;
    k.fold_bucket = a.fold_bucket;
    k.fold_hi_off = a.fold_hi_off;
    k.fold_bucket_lo = a.fold_bucket_lo;
    k.ev_start = a.ev_start;
    k.ev_stop = a.ev_stop;
    hipError_t e = hipErrorInvalidValue;
    if (a.dtype == PVS_I8)
        e = wide            ? pvs_scan_dispatch_i8_wide(k, a.kslabs, a.qgroups, a.metric, a.mode, s)
            : a.kslabs <= 4 ? pvs_scan_dispatch_i8(k, a.kslabs, a.qgroups, a.metric, a.mode, s)
                            : pvs_scan_dispatch_i8_large(k, a.kslabs, a.qgroups, a.metric, a.mode, s);
    else if (a.dtype == PVS_F16)
        e = a.kslabs <= 4   ? pvs_scan_dispatch_f16_small(k, a.kslabs, a.qgroups, a.metric, a.mode, s)
            : a.kslabs <= 8 ? pvs_scan_dispatch_f16_large(k, a.kslabs, a.qgroups, a.metric, a.mode, s)
                            : pvs_scan_dispatch_f16_xl(k, a.kslabs, a.qgroups, a.metric, a.mode, s);
    else if (a.dtype == PVS_F32)
        e = a.kslabs <= 4   ? pvs_scan_dispatch_f32_small(k, a.kslabs, a.qgroups, a.metric, a.mode, s)
            : a.kslabs <= 8 ? pvs_scan_dispatch_f32_mid(k, a.kslabs, a.qgroups, a.metric, a.mode, s)
            : a.kslabs <= 16 ? pvs_scan_dispatch_f32_large(k, a.kslabs, a.qgroups, a.metric, a.mode, s)
                             : pvs_scan_dispatch_f32_xl(k, a.kslabs, a.qgroups, a.metric, a.mode, s);
    return e;
}

__global__ void k_void_thresholds(float *thr, const uint32_t *need_dense, uint32_t n) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n && need_dense[q] != 2) thr[q] = -__builtin_inff();
}
hipError_t pvs_launch_void_thresholds(float *thr, const uint32_t *need_dense, uint32_t n, hipStream_t s) {
    hipLaunchKernelGGL(k_void_thresholds, dim3((n + 255) / 256), dim3(256), 0, s, thr, need_dense, n);
    return hipGetLastError();
}

// ------------------------------------------------------------- k-th select
// 4-pass MSB radix select on order-preserving u32 keys, one workgroup (256 threads) per query.
// (Measured alternatives on MI355X: a bitonic sort in LDS was 4x slower, wave-aggregated LDS
// atomics 1.7x slower; what matters is keeping the keys on chip across the four passes.)
// One radix digit: after the histogram of the matching keys is complete (and a __syncthreads), find the bin that holds rank kk.
// The inclusive prefix over the 256 bins is a shuffle scan inside each of the four waves plus their totals through LDS — two
// workgroup barriers per digit where a Hillis-Steele scan through LDS took sixteen (this select is pure latency: 22 us -> see
// DESIGN.md for a 16k-key query).  s_sel: 4 words, two per parity of `pass` (a thread may still be reading the previous digit's
// answer when thread 0 resets the next one's).  Returns the bin (256: fewer than kk keys match) and updates kk.
__device__ static inline uint32_t radix_pick(const uint32_t *hist, uint32_t *s_sel, int pass, uint32_t &kk) {
    __shared__ uint32_t s_wave_total[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t *sel = s_sel + 2 * (pass & 1);
    const uint32_t own = hist[tid];
    uint32_t v = own;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)v, off, 64);
        if (lane >= off) v += up;
    }
    if (lane == 63) s_wave_total[wave] = v;
    if (tid == 0) sel[0] = 256;
    __syncthreads();
    for (int w = 0; w < wave; w++) v += s_wave_total[w];
    const uint32_t before = v - own;
    if (before < kk && v >= kk) {  // (at most one bin satisfies this)
        sel[0] = (uint32_t)tid;
        sel[1] = kk - before;
    }
    __syncthreads();
    const uint32_t bin = sel[0];
    if (bin < 256) kk = sel[1];
    return bin;
}

// lo / hi: the smallest and the largest key when the caller knows them (it usually produced the keys a moment ago): the digits then
// start at the highest bit in which two keys differ — the keys of one query are floats of similar magnitude, their leading byte or
// two are common — as in kth_in_registers.  lo > hi: unknown, four digits from the top.
template <typename KeyAt>
__device__ static inline uint32_t radix_kth(uint32_t n, uint32_t k, uint32_t *hist, uint32_t *s_sel, KeyAt key_at, uint32_t lo = 1, uint32_t hi = 0) {
    const int tid = threadIdx.x;
    uint32_t prefix = 0, mask = 0, kk = k;
    int shift = 24;
    if (lo <= hi) {
        const uint32_t diff = lo ^ hi;
        if (diff == 0) return lo;
        const int top = 31 - __builtin_clz(diff);
        mask = top == 31 ? 0u : ~((2u << top) - 1u);
        prefix = hi & mask;
        shift = top >= 7 ? top - 7 : 0;
    }
    for (int pass = 0;; pass++) {
        hist[tid] = 0;  // (every thread has read the previous digit's histogram before radix_pick's first barrier)
        __syncthreads();
        for (uint32_t i = tid; i < n; i += 256) {
            const uint32_t key = key_at(i);
            if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        const uint32_t b = radix_pick(hist, s_sel, pass, kk);
        if (b >= 256) return 0xffffffffu;
        prefix |= b << shift;  // (a last digit that overlaps the previous one re-states bits the filter already fixed)
        mask |= 0xffu << shift;
        if (shift == 0) break;
        shift = shift >= 8 ? shift - 8 : 0;
    }
    return prefix;
}
// workgroup-wide minimum and maximum of per-thread values (256 threads); s_mm: 8 words of LDS
__device__ static inline void wg_min_max(uint32_t &lo, uint32_t &hi, uint32_t *s_mm) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t l2 = (uint32_t)__shfl_xor((int)lo, off, 64), h2 = (uint32_t)__shfl_xor((int)hi, off, 64);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
    }
    if (lane == 0) {
        s_mm[wave] = lo;
        s_mm[4 + wave] = hi;
    }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < 4; w++) {
        lo = s_mm[w] < lo ? s_mm[w] : lo;
        hi = s_mm[4 + w] > hi ? s_mm[4 + w] : hi;
    }
}

// The select with the keys of one query in NR registers per thread (per_query <= 256 * NR).  The keys of a query are floats of
// similar magnitude: their common leading bits are found first (one min/max reduction) and the 8-bit digits start at the highest
// bit in which two keys differ — usually three digit passes instead of four, and a first digit that actually spreads (a pass
// whose keys all fall into one bin is 2,048 LDS atomics on one address).
template <int NR>
__device__ static inline uint32_t kth_in_registers(const float *v, uint32_t per_query, uint32_t k, uint32_t *hist, uint32_t *s_sel) {
    __shared__ uint32_t s_mm[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t kreg[NR];
    uint32_t lo = 0xffffffffu, hi = 0u;
#pragma unroll
    for (int j = 0; j < NR; j++) {
        const uint32_t i = threadIdx.x + 256u * j;
        kreg[j] = i < per_query ? f32_sort_key(v[i]) : 0xffffffffu;
        if (i < per_query) {
            lo = kreg[j] < lo ? kreg[j] : lo;
            hi = kreg[j] > hi ? kreg[j] : hi;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t l2 = (uint32_t)__shfl_xor((int)lo, off, 64), h2 = (uint32_t)__shfl_xor((int)hi, off, 64);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
    }
    if (lane == 0) {
        s_mm[wave] = lo;
        s_mm[4 + wave] = hi;
    }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < 4; w++) {
        lo = s_mm[w] < lo ? s_mm[w] : lo;
        hi = s_mm[4 + w] > hi ? s_mm[4 + w] : hi;
    }
    const uint32_t diff = lo ^ hi;
    if (diff == 0) return lo;  // every key equal
    const int top = 31 - __builtin_clz(diff);  // highest bit in which two keys differ
    uint32_t mask = top == 31 ? 0u : ~((2u << top) - 1u), prefix = hi & mask, kk = k;
    int shift = top >= 7 ? top - 7 : 0;
    for (int pass = 0;; pass++) {
        hist[tid] = 0;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NR; j++) {
            const uint32_t i = threadIdx.x + 256u * j;
            if (i < per_query && (kreg[j] & mask) == prefix) atomicAdd(&hist[(kreg[j] >> shift) & 255u], 1u);
        }
        __syncthreads();
        const uint32_t b = radix_pick(hist, s_sel, pass, kk);
        if (b >= 256) return 0xffffffffu;
        prefix |= b << shift;  // (a last digit that overlaps the previous one re-states bits the filter already fixed)
        mask |= 0xffu << shift;
        if (shift == 0) break;
        shift = shift >= 8 ? shift - 8 : 0;
    }
    return prefix;
}

__global__ __launch_bounds__(256) void k_kth(const float *vals, uint32_t per_query, uint32_t batch, uint32_t k, float *out, const QInfo *qinfo, int metric) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t s_sel[4];
    const uint32_t q = blockIdx.x;
    if (q >= batch || (qinfo && pvs_query_all_null(metric, qinfo[q].bb))) {  // padding query / every distance NULL: never admits a candidate
        if (threadIdx.x == 0) out[q] = -__builtin_inff();
        return;
    }
    const float *v = vals + (size_t)q * per_query;
    uint32_t key = 0xffffffffu;
    if (per_query >= k) {
        // keys live in registers across the passes (the loads are the latency that matters)
        if (per_query <= 2048)
            key = kth_in_registers<8>(v, per_query, k, hist, s_sel);
        else if (per_query <= 16384)
            key = kth_in_registers<64>(v, per_query, k, hist, s_sel);
        else
            key = radix_kth(per_query, k, hist, s_sel, [&](uint32_t i) { return f32_sort_key(v[i]); });
    }
    if (threadIdx.x == 0) {
        float t = f32_from_sort_key(key);
        out[q] = (t != t) ? __builtin_inff() : t;  // too few finite minima: admit everything
    }
}

hipError_t pvs_launch_kth(const float *vals, uint32_t per_query, uint32_t batch, uint32_t k, float *out, hipStream_t s, const QInfo *qinfo, int metric) {
    const uint32_t bp = (batch + 31) / 32 * 32;
    hipLaunchKernelGGL(k_kth, dim3(bp), dim3(256), 0, s, vals, per_query, batch, k, out, qinfo, metric);
    return hipGetLastError();
}

// ---------------------------------------------------------------- finalise
struct FinK {
    const uint8_t *rows;
    const float *norm2;
    const int64_t *ids;
    const void *qexact;
    const QInfo *qinfo;
    const uint2 *seg;
    const uint32_t *seg_cnt;
    uint2 *cand;
    int64_t *out_ids;
    float *out_dist;
    uint32_t *out_count, *need_dense, *cand_seen;
    uint64_t n_rows;
    uint32_t stride, dim, cand_cap, k, n_segments, seg_queries, seg_cap;
    int metric;
    uint32_t *w_ub, *w_surv;       // LIGHT: global work area (FinalizeArgs.w_*)
    unsigned long long *w_sort;
    const uint32_t *flat_cnt;      // segment-overflow rerun (FinalizeArgs.flat_cnt)
    const float *thr;              // thresholds to certify (FinalizeArgs.thr), or nullptr
    const float *thr_all;          // the thresholds pass B ran with (FinalizeArgs.thr_all)
    int null_ok;                   // short pages may be completed from the NULL list (flag 3)
    uint32_t *h_flags, *h_seen;    // pinned host mirrors of need_dense / cand_seen (FinalizeArgs.h_flags), or nullptr
    const uint32_t *trank, *tinv;  // second sort key: tie rank of a row and its inverse (FinalizeArgs.trank)
};

constexpr int FIN_QMAX = 8192;  // bytes of LDS for the query vector the rerank reads (dims beyond that read it from global memory)
constexpr int FIN_LDS = PVS_CAND_CAP * 4 + PVS_SURV_CAP * 4 + 64 + FIN_QMAX;

__device__ static inline float cand_err(const FinK &a, const QInfo &qi, float aa) {
    return a.metric == PVS_COSINE ? qi.eA : qi.eA + qi.eR * aa;
}
// scan key of a candidate.  int8 candidates carry the exact integer dot, f16 ones the key itself.
template <int DT>
__device__ static inline float cand_key(const FinK &a, const QInfo &qi, uint32_t payload, float aa) {
    if constexpr (DT == PVS_I8) {
        const float d = (float)(int)payload;
        return a.metric == PVS_COSINE ? -d * __frcp_rn(__fsqrt_rn(aa)) : aa + (qi.bb - 2.0f * d);
    } else {
        return __builtin_bit_cast(float, payload);
    }
}

// LIGHT (FinalizeArgs.w_*; float rows keep their query — and nothing else — in dynamic LDS): bound keys, survivor list and sorts of more than 512 records live in global memory (they
// stay in L2: a few KB per query), LDS holds the histogram and a 512-record sort buffer — ~6 KB, so the workgroup fits beside
// k_scan's two workgroups on a CU and pass C of one search runs under the scan of the next one (several streams per index).
constexpr uint32_t FIN_SMALL_SORT = 512;
template <int DT, bool LIGHT>
__global__ __launch_bounds__(256) void k_finalize(FinK a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ uint32_t hist[256];
    __shared__ uint32_t s_misc_light[LIGHT ? 16 : 1];
    __shared__ unsigned long long s_sort_light[LIGHT ? FIN_SMALL_SORT : 1];
    const uint32_t q = blockIdx.x;
    uint32_t *const s_ub = LIGHT ? a.w_ub + (size_t)q * a.cand_cap : (uint32_t *)smem;                            // [CAND_CAP] sort keys of upper bounds
    uint32_t *const s_surv = LIGHT ? a.w_surv + (size_t)q * PVS_SURV_CAP : (uint32_t *)(smem + PVS_CAND_CAP * 4);  // [SURV_CAP] candidate slots
    uint32_t *const s_misc = LIGHT ? s_misc_light : (uint32_t *)(smem + PVS_CAND_CAP * 4) + PVS_SURV_CAP;          // [0]=survivor count, [2..5] select scratch
    unsigned long long *s_sort = LIGHT ? s_sort_light : (unsigned long long *)smem;  // !LIGHT: overlays s_ub after the select

    const int tid = threadIdx.x;
#ifdef PVS_FIN_PROF  // tuning build: wall clock (100 MHz s_memrealtime) at the phase boundaries of workgroup 0
    unsigned long long fp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define FIN_STAMP(i) fp[i] = __builtin_amdgcn_s_memrealtime()
#else
#define FIN_STAMP(i) do { } while (0)
#endif
    FIN_STAMP(0);
    int64_t *oid = a.out_ids + (size_t)q * a.k;
    float *od = a.out_dist + (size_t)q * a.k;
    // a query that makes every distance NULL (zero / NaN-bearing): the whole page is the head of ALL rows in tie order — the
    // NULL-tail step writes it (flag 3); nothing was emitted for it (k_kth gave it T = -inf)
    if (a.null_ok && !a.flat_cnt && pvs_query_all_null(a.metric, a.qinfo[q].bb)) {
        if (tid == 0) {
            a.need_dense[q] = 3;
            if (a.h_flags) a.h_flags[q] = 3;
            a.out_count[q] = 0;
            if (a.cand_seen) a.cand_seen[q] = 0;
            if (a.h_seen) a.h_seen[q] = 0;
        }
        return;
    }
    // ---- gather: the scan left this query's candidates in one segment per workgroup row stream (no atomics on its side);
    // prefix-sum the fill counts (offsets live in the not-yet-used bound array) and copy the segments into one flat list.
    uint2 *const flat = a.cand + (size_t)q * a.cand_cap;
    // A list of <= FIN_STAGE candidates (the usual ~1.6k) never leaves the chip: the pairs and the rows' |a|^2 are staged in the
    // parts of the LDS carve-out that nothing else uses at that size — pairs above the first FIN_STAGE sort records, |a|^2 in
    // the upper half of the survivor list (m <= cnt <= FIN_STAGE) — instead of a global list that every later phase reads back
    // (each read a round trip to L2: the gather, the bounds and the survivor pass were 12 + 6 + 5 us of a 35-us workgroup).
    constexpr uint32_t FIN_STAGE = 4096;
    static_assert(FIN_STAGE * 8 * 2 <= PVS_CAND_CAP * 4 && FIN_STAGE * 2 <= PVS_SURV_CAP, "staging areas overlap");
    uint2 *const s_cand = (uint2 *)(smem + FIN_STAGE * 8);
    float *const s_aa = (float *)(smem + PVS_CAND_CAP * 4 + FIN_STAGE * 4);
    bool staged = false;
    uint32_t cnt = 0;
    bool seg_overflow_only = false;  // a segment overflowed although the query's candidates fit one list: the scan can be rerun into flat lists
    if (a.flat_cnt) {
        // segment-overflow rerun: only the queries handed back for it; their candidates are already one flat list
        if (a.need_dense[q] != 2) return;
        cnt = a.flat_cnt[q];
        if (cnt > a.cand_cap) cnt = a.cand_cap + 1;
    } else {
        uint32_t *const s_off = s_ub;  // [n_segments + 1]
        __shared__ uint32_t s_part[256];
        __shared__ uint32_t s_over;
        const uint32_t *sc = a.seg_cnt + (size_t)q * a.n_segments;
        const uint32_t per = (a.n_segments + 255) / 256;
        // (this part of pass C is pure load latency: the counts and the first slot of every non-empty segment are requested in
        //  independent batches — at most 4,096 segments = 16 per lane — instead of one dependent load after the other)
        constexpr uint32_t PER_MAX = 16;
        uint32_t cs[PER_MAX];
        uint32_t mine = 0, mine_true = 0;
        bool over = false;
        if (per <= PER_MAX) {
#pragma unroll
            for (uint32_t i = 0; i < PER_MAX; i++) {
                const uint32_t sg = tid * per + i;
                cs[i] = (i < per && sg < a.n_segments) ? sc[sg] : 0u;
            }
#pragma unroll
            for (uint32_t i = 0; i < PER_MAX; i++) {
                over |= cs[i] > a.seg_cap;
                mine_true += cs[i] < (1u << 20) ? cs[i] : (1u << 20);
                cs[i] = cs[i] < a.seg_cap ? cs[i] : a.seg_cap;
                mine += cs[i];
            }
        } else {
            for (uint32_t i = 0; i < per; i++) {
                const uint32_t sg = tid * per + i;
                const uint32_t c = sg < a.n_segments ? sc[sg] : 0u;
                over |= c > a.seg_cap;
                mine_true += c < (1u << 20) ? c : (1u << 20);
                mine += c < a.seg_cap ? c : a.seg_cap;
            }
        }
        __shared__ uint32_t s_true;
        if (tid == 0) {
            s_over = 0;
            s_true = 0;
        }
        __syncthreads();
        if (over) s_over = 1;
        if (mine_true) atomicAdd(&s_true, mine_true < (1u << 20) ? mine_true : (1u << 20));
        // exclusive prefix of the 256 per-lane totals: shuffle scan inside each wave + the wave totals through LDS
        uint32_t v = mine;
        {
            const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t up = (uint32_t)__shfl_up((int)v, off, 64);
                if (lane >= off) v += up;
            }
            if (lane == 63) s_part[wave] = v;
            __syncthreads();
            for (int w = 0; w < wave; w++) v += s_part[w];
            cnt = s_part[0] + s_part[1] + s_part[2] + s_part[3];
        }
        uint32_t run = v - mine;
        const bool too_many = s_over != 0 || cnt > a.cand_cap;
        staged = !LIGHT && !too_many && cnt <= FIN_STAGE;
        auto put = [&](uint32_t at, uint2 v) __attribute__((always_inline)) {
            if (staged) s_cand[at] = v;
            else flat[at] = v;
        };
        if (!too_many) {
            if (per <= PER_MAX) {
                uint2 first[PER_MAX];
#pragma unroll
                for (uint32_t i = 0; i < PER_MAX; i++)
                    if (cs[i]) first[i] = a.seg[((size_t)(tid * per + i) * a.seg_queries + q) * a.seg_cap];
#pragma unroll
                for (uint32_t i = 0; i < PER_MAX; i++)
                    if (cs[i]) {
                        put(run, first[i]);
                        const uint2 *src = a.seg + ((size_t)(tid * per + i) * a.seg_queries + q) * a.seg_cap;
                        for (uint32_t e = 1; e < cs[i]; e++) put(run + e, src[e]);
                        run += cs[i];
                    }
            } else {
                for (uint32_t i = 0; i < per; i++) {
                    const uint32_t sg = tid * per + i;
                    if (sg >= a.n_segments) break;
                    const uint32_t c = sc[sg];
                    const uint2 *src = a.seg + ((size_t)sg * a.seg_queries + q) * a.seg_cap;
                    for (uint32_t e = 0; e < c; e++) put(run + e, src[e]);
                    run += c;
                }
            }
        } else {
            seg_overflow_only = s_over != 0 && s_true <= a.cand_cap;  // (s_true: what the scan emitted, segment caps ignored)
            cnt = a.cand_cap + 1;  // a segment or the list overflowed: rerun into flat lists, or the dense path
        }
        (void)s_off;
        __syncthreads();  // (workgroup-scope fence: the list is read back by other lanes below)
    }
    FIN_STAMP(1);
    if (tid == 0 && a.cand_seen) a.cand_seen[q] = cnt;
    if (tid == 0 && a.h_seen) a.h_seen[q] = cnt;
    const uint64_t want = a.k < a.n_rows ? a.k : a.n_rows;
    // a short page can be completed from the index's NULL list when every row with a comparable key was emitted (T = +inf)
    // ... and the query is an ordinary one (a finite, for cosine non-zero, norm): a query with an infinite component turns rows
    // NULL that no list knows
    const float bb_q = a.qinfo[q].bb;
    const bool can_complete = a.null_ok && a.thr_all && a.thr_all[q] == __builtin_inff() && bb_q < __builtin_inff() && (a.metric == PVS_L2 || bb_q > 0.f);
    if (cnt > a.cand_cap || (cnt < want && !can_complete)) {  // overflowed, or NULL-distance rows are needed to fill the page
        if (tid == 0) {
            a.need_dense[q] = seg_overflow_only ? 2 : 1;
            if (a.h_flags) a.h_flags[q] = seg_overflow_only ? 2 : 1;
            a.out_count[q] = 0;
        }
        return;
    }
    const QInfo qi = a.qinfo[q];
    auto cand_at = [&](uint32_t i) __attribute__((always_inline)) { return staged ? s_cand[i] : flat[i]; };
    uint32_t ub_lo = 0xffffffffu, ub_hi = 0u;  // range of the upper-bound keys: the select below starts at their first differing bit
    for (uint32_t i = tid; i < cnt; i += 256) {
        const uint2 c = cand_at(i);
        const float aa = a.norm2[c.x];
        if (staged) s_aa[i] = aa;
        const uint32_t ubk = f32_sort_key(cand_key<DT>(a, qi, c.y, aa) + cand_err(a, qi, aa));
        s_ub[i] = ubk;
        ub_lo = ubk < ub_lo ? ubk : ub_lo;
        ub_hi = ubk > ub_hi ? ubk : ub_hi;
    }
    if (tid == 0) s_misc[0] = 0;
    __shared__ uint32_t s_mm_fin[8];
    wg_min_max(ub_lo, ub_hi, s_mm_fin);  // (its barrier also publishes s_ub and s_misc[0])
    __syncthreads();
    FIN_STAMP(2);
    uint32_t kub = 0xffffffffu;
    if (cnt > a.k || (a.thr && cnt == a.k)) kub = radix_kth(cnt, a.k, hist, s_misc + 2, [&](uint32_t i) { return s_ub[i]; }, ub_lo, ub_hi);
    const float kappa = (kub == 0xffffffffu) ? __builtin_inff() : f32_from_sort_key(kub);
    __syncthreads();
    // A threshold taken below the k-th sample value (a.thr set) does not by itself guarantee k rows at or below it.  The scan emitted
    // every row whose LOWER bound is <= T; if the k-th smallest UPPER bound among them is <= T, k rows have their reference key
    // <= T, every row of the k best has its lower bound <= T, and the list is complete.  Otherwise: the dense path.
    if (a.thr && want == a.k && !(kappa <= a.thr[q])) {
        if (tid == 0) {
            a.need_dense[q] = 1;
            if (a.h_flags) a.h_flags[q] = 1;
            a.out_count[q] = 0;
        }
        return;
    }
    FIN_STAMP(3);
    // survivors: lower bound <= k-th smallest upper bound
    for (uint32_t i = tid; i < cnt; i += 256) {
        const uint2 c = cand_at(i);
        const float aa = staged ? s_aa[i] : a.norm2[c.x];
        if (cand_key<DT>(a, qi, c.y, aa) - cand_err(a, qi, aa) <= kappa) {
            const uint32_t p = atomicAdd(&s_misc[0], 1u);
            if (p < PVS_SURV_CAP) s_surv[p] = i;  // candidate slot
        }
    }
    __syncthreads();
    const uint32_t m = s_misc[0];
    if (m > PVS_SURV_CAP) {  // massive near-ties: let the dense path answer
        if (tid == 0) {
            a.need_dense[q] = 1;
            if (a.h_flags) a.h_flags[q] = 1;
            a.out_count[q] = 0;
        }
        return;
    }
    FIN_STAMP(4);
    uint32_t m2 = 1;
    while (m2 < m) m2 <<= 1;
    __syncthreads();  // s_ub is dead from here: s_sort overlays it
    if (LIGHT && m2 > FIN_SMALL_SORT) s_sort = a.w_sort + (size_t)q * PVS_SURV_CAP;  // (many near-ties or a large k: sort in global memory)
    const uint8_t *qe = (const uint8_t *)a.qexact + (size_t)q * a.dim * (DT == PVS_I8 ? 1 : 4);
    // the query of this workgroup, zero-padded to a whole 16-byte chunk, in LDS beyond everything s_sort overlays (rerank_distance)
    // (LIGHT, float rows: the query is all the dynamic LDS there is; int8 rows rerank from the exact integer sums and keep none)
    uint8_t *const s_q = LIGHT ? smem : smem + PVS_CAND_CAP * 4 + PVS_SURV_CAP * 4 + 64;
    if constexpr (!LIGHT || DT != PVS_I8) {
        const uint32_t qbytes = a.dim * (DT == PVS_I8 ? 1u : 4u), padded = (qbytes + 63u) & ~63u;
        if constexpr (DT == PVS_I8) {
            for (uint32_t i = tid; i < padded; i += 256) s_q[i] = i < qbytes ? qe[i] : (uint8_t)0;
        } else {
            for (uint32_t i = tid; i < padded / 4; i += 256) ((uint32_t *)s_q)[i] = i < qbytes / 4 ? ((const uint32_t *)qe)[i] : 0u;
        }
    }
    __syncthreads();
    for (uint32_t i = tid; i < m2; i += 256) {
        unsigned long long v = ~0ull;
        if (i < m) {
            const uint32_t slot = s_surv[i];
            const uint2 c = cand_at(slot);
            const uint32_t row = c.x;
            const float aa = staged ? s_aa[slot] : a.norm2[row];
            float d = 0.f;
            bool closed = false;
            if constexpr (DT == PVS_I8) {
                // The reference accumulates integer-valued f32 terms; while every partial sum stays
                // below 2^24 the result is a pure function of the exact integer sums (oracle:
                // orc_i8_*_from_sums), which the MFMA already produced.  Otherwise recompute in order.
                const int dot = (int)c.y;
                const float lim = 16777216.0f;
                if (a.metric == PVS_COSINE) {
                    if (aa < lim && qi.bb < lim) {
                        d = ref_cosine_finish((float)dot, aa, qi.bb);
                        closed = true;
                    }
                } else {
                    const double ss = (double)aa + (double)qi.bb - 2.0 * (double)dot;
                    if (aa < lim && qi.bb < lim && ss < (double)lim) {
                        d = ref_l2_finish((float)ss);
                        closed = true;
                    }
                }
            }
            if (!closed) {  // (LIGHT, int8 rows: no query in LDS — the generic in-order form, rare: sums beyond 2^24)
                if constexpr (LIGHT && DT == PVS_I8)
                    d = exact_distance<DT>(a.rows, a.stride, row, qe, (int)a.dim, a.metric, aa, qi.bb);
                else
                    d = rerank_distance<DT>(a.rows, a.stride, row, s_q, (int)a.dim, a.metric, aa, qi.bb);
            }
            v = ((unsigned long long)f32_sort_key(d) << 32) | (a.trank ? a.trank[row] : row);  // ties: by tie rank (key DESC, id ASC), else by row = id
        }
        s_sort[i] = v;
    }
    __syncthreads();
    FIN_STAMP(5);
    // sort ascending on (distance key, row or tie rank).  The usual ~100 survivors: a rank sort — the entries are distinct, so an
    // entry's slot is the number of smaller ones; one pass of broadcast LDS reads and one barrier where the bitonic network below
    // takes 28 barrier-separated steps for 128 entries (4.4 us of a 30-us workgroup).
    if (m2 <= 256) {
        const unsigned long long mine = (uint32_t)tid < m ? s_sort[tid] : ~0ull;
        uint32_t rank = 0;
        for (uint32_t j = 0; j < m; j++) rank += s_sort[j] < mine ? 1u : 0u;
        __syncthreads();
        if ((uint32_t)tid < m) s_sort[rank] = mine;
        __syncthreads();
    } else
    for (uint32_t sz = 2; sz <= m2; sz <<= 1) {
        for (uint32_t st = sz >> 1; st > 0; st >>= 1) {
            for (uint32_t i = tid; i < m2 / 2; i += 256) {
                const uint32_t lo = 2 * i - (i & (st - 1));
                const uint32_t hi = lo + st;
                const bool up = (lo & sz) == 0;
                const unsigned long long x = s_sort[lo], y = s_sort[hi];
                if ((x > y) == up) {
                    s_sort[lo] = y;
                    s_sort[hi] = x;
                }
            }
            __syncthreads();
        }
    }
    FIN_STAMP(6);
#ifdef PVS_FIN_PROF
    if (q == 0 && tid == 0)
        printf("finprof cand %u survivors %u: gather %llu ub-keys %llu kth %llu survive %llu rerank %llu sort %llu (x10 ns)\n", cnt, m, fp[1] - fp[0],
               fp[2] - fp[1], fp[3] - fp[2], fp[4] - fp[3], fp[5] - fp[4], fp[6] - fp[5]);
#endif
    // survivors with a NULL distance (a row with non-finite components that the scan emitted) sort last: they are rows of the NULL
    // list, not of the finite part of the page
    __shared__ uint32_t s_nfin;
    if (tid == 0) s_nfin = 0;
    __syncthreads();
    for (uint32_t i = tid; i < m; i += 256) {
        const bool fin = (uint32_t)(s_sort[i] >> 32) != 0xffffffffu;
        const bool next_fin = i + 1 < m && (uint32_t)(s_sort[i + 1] >> 32) != 0xffffffffu;
        if (fin && !next_fin) s_nfin = i + 1;
    }
    __syncthreads();
    const uint32_t nfin = s_nfin;
    const uint32_t nout = nfin < a.k ? nfin : a.k;
    // A short page: NULL rows are ordered by tie order over the WHOLE corpus and the candidate list only holds rows whose scan key
    // was comparable.  Cosine, a query-independent NULL set and T = +inf: the finite part is complete, the host appends the tail
    // from the NULL list (flag 3).  Otherwise the dense path.
    const bool tail = nout < want;
    if (tail && !can_complete) {
        if (tid == 0) {
            a.need_dense[q] = 1;
            if (a.h_flags) a.h_flags[q] = 1;
            a.out_count[q] = 0;
        }
        return;
    }
    for (uint32_t i = tid; i < a.k; i += 256) {
        if (i < nout) {
            const unsigned long long v = s_sort[i];
            oid[i] = a.ids[a.tinv ? a.tinv[(uint32_t)v] : (uint32_t)v];
            od[i] = f32_from_sort_key((uint32_t)(v >> 32));
        } else {
            oid[i] = -1;
            od[i] = __builtin_nanf("");
        }
    }
    if (tid == 0) {
        a.out_count[q] = nout;
        a.need_dense[q] = tail ? 3 : 0;
        if (a.h_flags) a.h_flags[q] = tail ? 3 : 0;
    }
}

hipError_t pvs_launch_finalize(const FinalizeArgs &f, hipStream_t s) {
    FinK k;
    k.rows = f.rows;
    k.norm2 = f.norm2;
    k.ids = f.ids;
    k.qexact = f.qexact;
    k.qinfo = f.qinfo;
    k.seg = f.seg;
    k.seg_cnt = f.seg_cnt;
    k.n_segments = f.n_segments;
    k.w_ub = f.w_ub;
    k.w_surv = f.w_surv;
    k.w_sort = f.w_sort;
    k.seg_queries = f.seg_queries;
    k.seg_cap = f.seg_cap;
    k.flat_cnt = f.flat_cnt;
    k.thr = f.thr;
    k.thr_all = f.thr_all;
    k.null_ok = f.null_ok;
    k.h_flags = f.h_flags;
    k.h_seen = f.h_seen;
    k.trank = f.trank;
    k.tinv = f.tinv;
    k.cand = f.cand;
    k.out_ids = f.out_ids;
    k.out_dist = f.out_dist;
    k.out_count = f.out_count;
    k.need_dense = f.need_dense;
    k.cand_seen = f.cand_seen;
    k.n_rows = f.n_rows;
    k.stride = f.stride;
    k.dim = f.dim;
    k.cand_cap = f.cand_cap;
    k.k = f.k;
    k.metric = f.metric;
    static std::atomic<bool> configured{false};
    if (!configured.load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void *)k_finalize<PVS_I8, false>, hipFuncAttributeMaxDynamicSharedMemorySize, FIN_LDS);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void *)k_finalize<PVS_F16, false>, hipFuncAttributeMaxDynamicSharedMemorySize, FIN_LDS);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void *)k_finalize<PVS_F32, false>, hipFuncAttributeMaxDynamicSharedMemorySize, FIN_LDS);
        if (e != hipSuccess) return e;
        configured.store(true, std::memory_order_release);
    }
    if ((((uint64_t)f.dim * (f.dtype == PVS_I8 ? 1u : 4u)) + 63u & ~63ull) > (uint64_t)FIN_QMAX) return hipErrorInvalidValue;  // (no scan instance is that wide)
    const bool light = f.w_ub && f.w_surv && f.w_sort;
    const size_t qlds = ((size_t)f.dim * 4 + 63) & ~(size_t)63;
#define PVS_FIN_LAUNCH(kernel, lds)                                                                                  \
    do {                                                                                                             \
        if (f.ev_start || f.ev_stop)                                                                                 \
            hipExtLaunchKernelGGL(kernel, dim3(f.batch), dim3(256), lds, s, f.ev_start, f.ev_stop, 0, k);            \
        else                                                                                                         \
            hipLaunchKernelGGL(kernel, dim3(f.batch), dim3(256), lds, s, k);                                         \
    } while (0)
    if (light && f.dtype == PVS_I8)
        PVS_FIN_LAUNCH((k_finalize<PVS_I8, true>), 0);
    else if (light && f.dtype == PVS_F16)
        PVS_FIN_LAUNCH((k_finalize<PVS_F16, true>), qlds);
    else if (light)
        PVS_FIN_LAUNCH((k_finalize<PVS_F32, true>), qlds);
    else if (f.dtype == PVS_I8)
        PVS_FIN_LAUNCH((k_finalize<PVS_I8, false>), FIN_LDS);
    else if (f.dtype == PVS_F16)
        PVS_FIN_LAUNCH((k_finalize<PVS_F16, false>), FIN_LDS);
    else
        PVS_FIN_LAUNCH((k_finalize<PVS_F32, false>), FIN_LDS);
#undef PVS_FIN_LAUNCH
    return hipGetLastError();
}

// ------------------------------------------------------------------- merge
// Rank-based merge of `world` sorted pages per query: the output slot of an entry is
// the number of entries (over all shards) that precede it under (NaN last, distance,
// id).  Shards hold disjoint id sets, so ranks are a permutation.
__device__ static inline bool page_less(float da, int64_t ka, int64_t ia, float db, int64_t kb, int64_t ib) {
    const uint32_t sa = f32_sort_key(da), sb = f32_sort_key(db);
    if (sa != sb) return sa < sb;
    if (ka != kb) return ka > kb;  // second sort key DESC (all zero when the pages carry none)
    return ia < ib;
}
// The pages of `world` shards, either as three arrays [world][batch][k] / [world][batch] (pvs_merge_topk_device) or as `world`
// packed records (pvs_page_record_*) of `rec` bytes each (one RCCL all-gather of one buffer per rank; one peer copy per shard).
struct MergePages {
    const uint8_t *ids, *dist, *cnt;   // rank 0's arrays
    const uint8_t *flags, *keys;       // packed records only (nullptr: no order keys)
    size_t stride_ids, stride_dist, stride_cnt, stride_flags, stride_keys;  // bytes from one rank's array to the next
    __device__ const int64_t *ids_of(uint32_t w) const { return (const int64_t *)(ids + (size_t)w * stride_ids); }
    __device__ const float *dist_of(uint32_t w) const { return (const float *)(dist + (size_t)w * stride_dist); }
    __device__ const uint32_t *cnt_of(uint32_t w) const { return (const uint32_t *)(cnt + (size_t)w * stride_cnt); }
    __device__ const uint32_t *flags_of(uint32_t w) const { return (const uint32_t *)(flags + (size_t)w * stride_flags); }
    __device__ const int64_t *keys_of(uint32_t w) const { return (const int64_t *)(keys + (size_t)w * stride_keys); }
};
__global__ __launch_bounds__(256) void k_merge(MergePages pg, uint32_t world, uint32_t batch, uint32_t k, int64_t *out_ids, float *out_dist,
                                               uint32_t *out_count, uint32_t *h_flags) {
    const uint32_t q = blockIdx.x;
    if (h_flags && pg.flags)  // every rank's verdicts [world][batch] straight into pinned host memory (no copy kernel behind the merge)
        for (uint32_t w = threadIdx.x; w < world; w += 256) h_flags[(size_t)w * batch + q] = pg.flags_of(w)[q];
    uint32_t total = 0;
    bool keyed = pg.keys != nullptr;
    for (uint32_t w = 0; w < world; w++) {
        total += pg.cnt_of(w)[q];
        if (keyed && pg.flags && pg.cnt_of(w)[q]) keyed = (pg.flags_of(w)[q] & PVS_PAGE_KEYED) != 0;  // every shard with entries must carry keys, or none is used
    }
    const uint32_t nout = total < k ? total : k;
    for (uint32_t e = threadIdx.x; e < world * k; e += 256) {
        const uint32_t w = e / k, p = e % k;
        if (p >= pg.cnt_of(w)[q]) continue;
        const size_t off = (size_t)q * k;
        const float d = pg.dist_of(w)[off + p];
        const int64_t id = pg.ids_of(w)[off + p];
        const int64_t key = keyed ? pg.keys_of(w)[off + p] : 0;
        uint32_t rank = p;
        for (uint32_t w2 = 0; w2 < world; w2++) {
            if (w2 == w) continue;
            const int64_t *i2 = pg.ids_of(w2) + off;
            const float *d2 = pg.dist_of(w2) + off;
            const int64_t *k2 = keyed ? pg.keys_of(w2) + off : nullptr;
            uint32_t lo = 0, hi = pg.cnt_of(w2)[q];
            while (lo < hi) {  // first entry of shard w2 that does not precede (d, key, id)
                const uint32_t mid = (lo + hi) >> 1;
                if (page_less(d2[mid], keyed ? k2[mid] : 0, i2[mid], d, key, id)) lo = mid + 1;
                else hi = mid;
            }
            rank += lo;
        }
        if (rank < k) {
            out_ids[(size_t)q * k + rank] = id;
            out_dist[(size_t)q * k + rank] = d;
        }
    }
    for (uint32_t i = nout + threadIdx.x; i < k; i += 256) {
        out_ids[(size_t)q * k + i] = -1;
        out_dist[(size_t)q * k + i] = __builtin_nanf("");
    }
    if (threadIdx.x == 0) out_count[q] = nout;
}
hipError_t pvs_launch_merge(const int64_t *ids, const float *dist, const uint32_t *counts, uint32_t world, uint32_t batch,
                            uint32_t k, int64_t *out_ids, float *out_dist, uint32_t *out_count, hipStream_t s, const int64_t *keys) {
    MergePages pg;
    pg.ids = (const uint8_t *)ids;
    pg.dist = (const uint8_t *)dist;
    pg.cnt = (const uint8_t *)counts;
    pg.flags = nullptr;
    pg.keys = (const uint8_t *)keys;  // [world][batch][k] order keys of the entries (nullptr: ties by id)
    pg.stride_flags = 0;
    pg.stride_keys = (size_t)batch * k * 8;
    pg.stride_ids = (size_t)batch * k * 8;
    pg.stride_dist = (size_t)batch * k * 4;
    pg.stride_cnt = (size_t)batch * 4;
    hipLaunchKernelGGL(k_merge, dim3(batch), dim3(256), 0, s, pg, world, batch, k, out_ids, out_dist, out_count, (uint32_t *)nullptr);
    return hipGetLastError();
}
// packed records (pvs_page_record_*): rank w's record starts at all_rec + w * rec_bytes
hipError_t pvs_launch_merge_packed(const uint8_t *all_rec, size_t rec_bytes, uint32_t world, uint32_t batch, uint32_t k, int64_t *out_ids,
                                   float *out_dist, uint32_t *out_count, hipStream_t s, uint32_t *h_flags) {
    MergePages pg;
    pg.ids = all_rec;
    pg.dist = all_rec + pvs_page_record_off_dist(batch, k);
    pg.cnt = all_rec + pvs_page_record_off_cnt(batch, k);
    pg.flags = all_rec + pvs_page_record_off_flags(batch, k);
    pg.keys = all_rec + pvs_page_record_off_keys(batch, k);
    pg.stride_ids = pg.stride_dist = pg.stride_cnt = pg.stride_flags = pg.stride_keys = rec_bytes;
    hipLaunchKernelGGL(k_merge, dim3(batch), dim3(256), 0, s, pg, world, batch, k, out_ids, out_dist, out_count, h_flags);
    return hipGetLastError();
}

// flags and order keys of a finished local page (pvs_launch_page_finish)
__global__ void k_page_finish(uint8_t *rec, size_t off_flags, size_t off_keys, uint32_t batch, uint32_t k, const uint32_t *need_dense,
                              const int64_t *ids_sorted, uint64_t n, const int64_t *order_keys) {
    const int64_t *pid = (const int64_t *)rec;
    uint32_t *flags = (uint32_t *)(rec + off_flags);
    int64_t *keys = (int64_t *)(rec + off_keys);
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < batch) flags[e] = (need_dense[e] & ~PVS_PAGE_KEYED) | (order_keys ? PVS_PAGE_KEYED : 0u);
    if (!order_keys || e >= batch * k) return;
    const int64_t id = pid[e];
    uint64_t lo = 0, hi = n;  // ids are strictly increasing in row order
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (ids_sorted[mid] < id) lo = mid + 1;
        else hi = mid;
    }
    keys[e] = (lo < n && ids_sorted[lo] == id) ? order_keys[lo] : 0;  // (padding entries: id -1)
}
hipError_t pvs_launch_page_finish(uint8_t *rec, uint32_t batch, uint32_t k, const uint32_t *need_dense, const int64_t *d_ids, uint64_t n,
                                  const int64_t *d_order_keys, hipStream_t s) {
    const uint32_t total = batch * k > batch ? batch * k : batch;
    hipLaunchKernelGGL(k_page_finish, dim3((total + 255) / 256), dim3(256), 0, s, rec, pvs_page_record_off_flags(batch, k), pvs_page_record_off_keys(batch, k), batch, k,
                       need_dense, d_ids, n, d_order_keys);
    return hipGetLastError();
}
