// pvs_rrf.hip — OR-composition of several vector filters ranked by reciprocal-rank fusion, on the device.
// Replaces, for the vector branches of a PQL `or` (pql/builder.rs:638-661), the chain the reference runs in SQLite:
// per branch `row_number() OVER (ORDER BY agg <dir>)` over every group (add_rank_column_expr :757-771), the UNION of
// the branches' groups, the fused score  sum_b 1.0 / (k_b + coalesce(rank_b, 9223372036854775805)) * weight_b
// (build_coalesced_expr :1284-1301) and ORDER BY score DESC ... LIMIT k.
// Exactness needs every group's exact rank in every branch (even a rank of 800,000 changes the low bits of the
// f64 sum), so each branch is ranked completely: stable radix sorts over all groups, nothing sampled.
#include <hipcub/hipcub.hpp>

#include "pvs_kernels.hpp"

namespace {
// order-preserving f64 -> u64 (ascending); -0.0 == +0.0 as in SQL comparisons
__device__ inline unsigned long long f64_key(double d) {
    d = d + 0.0;
    const unsigned long long b = __builtin_bit_cast(unsigned long long, d);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__host__ __device__ inline unsigned long long i64_key(int64_t v) { return (unsigned long long)v ^ 0x8000000000000000ull; }

// window order of one branch: ascending = NULL first, descending = NULL last (SQLite's default NULL placement)
__global__ void k_rank_keys(const double *vals, uint32_t n, int descending, unsigned long long *keys, uint32_t *idx) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const double v = vals[i];
        unsigned long long k;
        if (v != v)
            k = descending ? ~0ull : 0ull;
        else
            k = descending ? ~f64_key(v) : f64_key(v);
        // keep real values away from the two NULL codes
        if (v == v && k == 0ull) k = 1ull;
        if (v == v && k == ~0ull) k = ~0ull - 1;
        keys[i] = k;
        idx[i] = i;
    }
}
constexpr int RANK_BITS = 40;
// entry of group slot s of branch b: key = group id, payload = (branch, rank)
__global__ void k_scatter_entries(const uint32_t *idx_sorted, const int64_t *gids, uint32_t n, uint32_t branch, unsigned long long *cat_key,
                                  unsigned long long *cat_pay) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t slot = idx_sorted[i];
        cat_key[slot] = i64_key(gids[slot]);
        cat_pay[slot] = ((unsigned long long)branch << RANK_BITS) | (unsigned long long)(i + 1);
    }
}
__device__ inline double rrf_term(int32_t k, int64_t rank_or_neg, double w) {
    const int64_t BIG = 9223372036854775805LL;  // VERY_LARGE_NUMBER, builder.rs:17-18
    const int64_t rank = rank_or_neg < 0 ? BIG : rank_or_neg;
    int64_t di;
    // SQLite integer addition; falls back to REAL on i64 overflow
    const double denom = __builtin_add_overflow((int64_t)k, rank, &di) ? (double)k + (double)rank : (double)di;
    return (1.0 / denom) * w;
}
__global__ void k_rrf_score(const unsigned long long *key, const unsigned long long *pay, uint64_t total, PvsRrfParams p, double *score,
                            unsigned long long *key2, uint32_t *idx2) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const bool head = i == 0 || key[i] != key[i - 1];
        idx2[i] = (uint32_t)i;
        if (!head) {
            key2[i] = ~0ull;
            continue;
        }
        int64_t rank[PVS_RRF_MAX_BRANCHES];
#pragma unroll
        for (int b = 0; b < PVS_RRF_MAX_BRANCHES; b++) rank[b] = -1;
        for (uint64_t e = i; e < total && key[e] == key[i]; e++) {
            const uint32_t b = (uint32_t)(pay[e] >> RANK_BITS);
#pragma unroll
            for (int bb = 0; bb < PVS_RRF_MAX_BRANCHES; bb++)
                if ((uint32_t)bb == b) rank[bb] = (int64_t)(pay[e] & ((1ull << RANK_BITS) - 1));
        }
        double tot = 0.0;
#pragma unroll
        for (int b = 0; b < PVS_RRF_MAX_BRANCHES; b++)
            if ((uint32_t)b < p.n_branches) {
                const double t = rrf_term(p.k[b], rank[b], p.w[b]);
                tot = b == 0 ? t : tot + t;
            }
        score[i] = tot;
        unsigned long long k2 = ~f64_key(tot);  // score DESC; NaN scores (NaN weights) last
        if (tot != tot || k2 == ~0ull) k2 = ~0ull - 1;
        key2[i] = k2;
    }
}
__global__ void k_rrf_emit(const unsigned long long *key2_sorted, const uint32_t *idx2_sorted, const unsigned long long *key, const double *score,
                           uint64_t total, uint32_t k, int64_t *out_groups, double *out_scores, uint32_t *out_count) {
    uint32_t n = 0;
    for (uint32_t j = threadIdx.x; j < k; j += blockDim.x) {
        const bool live = j < total && key2_sorted[j] != ~0ull;
        out_groups[j] = live ? (int64_t)(key[idx2_sorted[j]] ^ 0x8000000000000000ull) : -1;
        out_scores[j] = live ? score[idx2_sorted[j]] : __builtin_nan("");
    }
    if (threadIdx.x == 0) {
        // heads sort before non-heads, so the live entries are a prefix
        uint32_t lo = 0, hi = (uint32_t)(total < k ? total : k);
        while (lo < hi) {
            const uint32_t mid = (lo + hi) / 2;
            if (key2_sorted[mid] != ~0ull)
                lo = mid + 1;
            else
                hi = mid;
        }
        n = lo;
        *out_count = n;
    }
}
inline unsigned grid_for(uint64_t n) { return (unsigned)std::min<uint64_t>(std::max<uint64_t>((n + 255) / 256, 1), 65535); }

// ---- bounded fusion (pvs_rrf_search's fast path): pages of each branch's ranking instead of ranking every group
__global__ void k_sample_keys(const unsigned long long *keys, uint32_t n, uint32_t m, unsigned long long *out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) out[i] = keys[(uint64_t)i * n / m];
}
// every group whose window key is <= thr, in any order: (slot, key)
// (hits are collected per workgroup in LDS and claim their slots with one global atomic: thousands of hits on one counter
//  serialise in L2 — see k_page_compact_cols)
__global__ __launch_bounds__(256) void k_page_compact(const unsigned long long *keys, uint32_t n, unsigned long long thr, uint32_t cap, uint32_t *count,
                                                      uint32_t *slots) {
    constexpr uint32_t LIST = 1024;
    __shared__ uint32_t s_n, s_base, s_list[LIST];
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        if (keys[i] <= thr) {
            const uint32_t p = atomicAdd(&s_n, 1u);
            if (p < LIST) {
                s_list[p] = i;
            } else {
                const uint32_t gp = atomicAdd(count, 1u);
                if (gp < cap) slots[gp] = i;
            }
        }
    __syncthreads();
    const uint32_t m = s_n < LIST ? s_n : LIST;
    if (threadIdx.x == 0 && m) s_base = atomicAdd(count, m);
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < m; j += blockDim.x)
        if (s_base + j < cap) slots[s_base + j] = s_list[j];
}
__global__ void k_gather_page(const int64_t *gids, const unsigned long long *keys, const uint32_t *slots, uint32_t m, int64_t *out_g,
                              unsigned long long *out_k) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) {
        out_g[i] = gids[slots[i]];
        out_k[i] = keys[slots[i]];
    }
}
// candidate group ids -> their slot in this branch (groups are stored in id order) and window key; absent: slot = ~0
__global__ void k_cand_lookup(const int64_t *gids, const unsigned long long *keys, uint32_t n, const int64_t *cand, uint32_t m, uint32_t *slot,
                              unsigned long long *key) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= m) return;
    const int64_t g = cand[c];
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (gids[mid] < g) lo = mid + 1;
        else hi = mid;
    }
    const bool found = lo < n && gids[lo] == g;
    slot[c] = found ? lo : 0xffffffffu;
    key[c] = found ? keys[lo] : 0ull;
}
// hist[p]++ with p = number of candidates whose (key, group id) is <= the group's: one pass over all groups, binary search in
// LDS.  Candidates arrive sorted ascending by (key, group id) — the window order with its id tie-break, comparable across
// shards.  Groups strictly before candidate j = sum_{p <= j} hist[p].
__global__ __launch_bounds__(256) void k_rank_count(const unsigned long long *keys, const int64_t *gids, uint32_t n, const unsigned long long *ckey,
                                                    const int64_t *cgid, uint32_t m, unsigned long long *hist) {
    extern __shared__ unsigned long long sm[];
    unsigned long long *sk = sm;              // [m]
    int64_t *ss = (int64_t *)(sm + m);        // [m]
    uint32_t *sh = (uint32_t *)(ss + m);      // [m + 1]
    for (uint32_t i = threadIdx.x; i < m; i += 256) {
        sk[i] = ckey[i];
        ss[i] = cgid[i];
    }
    for (uint32_t i = threadIdx.x; i <= m; i += 256) sh[i] = 0;
    __syncthreads();
    for (uint32_t g = blockIdx.x * 256 + threadIdx.x; g < n; g += gridDim.x * 256) {
        const unsigned long long k = keys[g];
        const int64_t gid = gids[g];
        uint32_t lo = 0, hi = m;  // first candidate with (key, group id) > this group's
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            const bool le = sk[mid] < k || (sk[mid] == k && ss[mid] <= gid);
            if (le) lo = mid + 1;
            else hi = mid;
        }
        if (lo < m) atomicAdd(&sh[lo], 1u);  // (groups behind every candidate — nearly all of them — need no count)
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i <= m; i += 256)
        if (sh[i]) atomicAdd(&hist[i], (unsigned long long)sh[i]);
}
// 64-bit digest of an array of 4- or 8-byte words: the wrapping sum of mix(index, word) — order-independent, so the atomics that
// build it cannot make it move; any single changed word changes it (race hunting: pvs_debug_set("rrf_digest", 1))
__device__ inline unsigned long long digest_mix(unsigned long long i, unsigned long long w) {
    unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ull ^ w * 0xD6E8FEB86659FD93ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__global__ __launch_bounds__(256) void k_digest(const void *p, uint64_t n, int word_bytes, unsigned long long *out) {
    unsigned long long acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256)
        acc += digest_mix(i, word_bytes == 4 ? (unsigned long long)((const uint32_t *)p)[i] : ((const unsigned long long *)p)[i]);
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd(out, acc);
}
}  // namespace

pvs_status pvs_digest_device(const void *d, uint64_t n, int word_bytes, uint64_t *out_host, hipStream_t s) {
    *out_host = 0;
    if (n == 0) return PVS_OK;
    unsigned long long *d_out = nullptr;
    HIP_TRY(hipMalloc((void **)&d_out, 8));  // (deliberately outside the scratch cache: the digests must not depend on what they examine)
    hipError_t e = hipMemsetAsync(d_out, 0, 8, s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_digest, dim3((unsigned)std::min<uint64_t>((n + 255) / 256, 2048)), dim3(256), 0, s, d, n, word_bytes, d_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out_host, d_out, 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d_out);
    if (e != hipSuccess) return pvs_fail(PVS_ERR_DEVICE, "digest: %s", hipGetErrorString(e));
    return PVS_OK;
}

pvs_status pvs_rrf_window_keys(const double *d_vals, uint32_t n, int descending, unsigned long long *d_keys, hipStream_t s) {
    if (n == 0) return PVS_OK;
    uint32_t *idx = nullptr;
    HIP_TRY(pvs_scratch_alloc((void **)&idx, (size_t)n * 4));  // (k_rank_keys also writes the identity permutation)
    hipLaunchKernelGGL(k_rank_keys, dim3(grid_for(n)), dim3(256), 0, s, d_vals, n, descending, d_keys, idx);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    pvs_scratch_free_on(idx, s);
    if (e != hipSuccess) return pvs_fail(PVS_ERR_DEVICE, "window keys: %s", hipGetErrorString(e));
    return PVS_OK;
}
pvs_status pvs_rrf_sample_keys(const unsigned long long *d_keys, uint32_t n, uint32_t m, unsigned long long *h_out, hipStream_t s) {
    unsigned long long *d = nullptr;
    HIP_TRY(pvs_scratch_alloc((void **)&d, (size_t)m * 8));
    hipLaunchKernelGGL(k_sample_keys, dim3((m + 255) / 256), dim3(256), 0, s, d_keys, n, m, d);
    hipError_t e = hipMemcpyAsync(h_out, d, (size_t)m * 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    pvs_scratch_free_on(d, s);
    if (e != hipSuccess) return pvs_fail(PVS_ERR_DEVICE, "sample keys: %s", hipGetErrorString(e));
    return PVS_OK;
}
// every group with key <= thr: (group id, key), any order; *out_count may exceed cap (then nothing is written)
pvs_status pvs_rrf_page(const unsigned long long *d_keys, const int64_t *d_gids, uint32_t n, unsigned long long thr, uint32_t cap,
                        int64_t *out_gids, unsigned long long *out_keys, uint32_t *out_count, hipStream_t s) {
    uint32_t *d_cnt = nullptr, *d_slots = nullptr;
    int64_t *d_g = nullptr;
    unsigned long long *d_k = nullptr;
    auto body = [&]() -> pvs_status {
        HIP_TRY(pvs_scratch_alloc((void **)&d_cnt, 4));
        HIP_TRY(pvs_scratch_alloc((void **)&d_slots, (size_t)std::max<uint32_t>(cap, 1) * 4));
        HIP_TRY(hipMemsetAsync(d_cnt, 0, 4, s));
        if (n) hipLaunchKernelGGL(k_page_compact, dim3(grid_for(n)), dim3(256), 0, s, d_keys, n, thr, cap, d_cnt, d_slots);
        HIP_TRY(hipGetLastError());
        uint32_t cnt = 0;
        HIP_TRY(hipMemcpyAsync(&cnt, d_cnt, 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        *out_count = cnt;
        if (cnt > cap || cnt == 0) return PVS_OK;
        HIP_TRY(pvs_scratch_alloc((void **)&d_g, (size_t)cnt * 8));
        HIP_TRY(pvs_scratch_alloc((void **)&d_k, (size_t)cnt * 8));
        hipLaunchKernelGGL(k_gather_page, dim3((cnt + 255) / 256), dim3(256), 0, s, d_gids, d_keys, d_slots, cnt, d_g, d_k);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(out_gids, d_g, (size_t)cnt * 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(out_keys, d_k, (size_t)cnt * 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        return PVS_OK;
    };
    pvs_status st = body();
    pvs_scratch_free_on(d_cnt, s);
    pvs_scratch_free_on(d_slots, s);
    pvs_scratch_free_on(d_g, s);
    pvs_scratch_free_on(d_k, s);
    return st;
}
constexpr uint32_t CNT_PAD = 32;  // words between the counters of two columns (one 128-byte line each)
// The page step for `ncol` key columns at once ([ncol][n] keys, one threshold each): three host round trips for the whole batch
// (samples, counts, pages) instead of three per column.  Used by the per-item search's page-first ranking (pvs_items.hip).
namespace {
__global__ void k_sample_keys_cols(const unsigned long long *keys, uint32_t n, uint32_t ncol, uint32_t m, unsigned long long *out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m * ncol) {
        const uint32_t col = i / m, j = i % m;
        out[i] = keys[(size_t)col * n + (uint64_t)j * n / m];
    }
}
__global__ __launch_bounds__(256) void k_page_compact_cols(const unsigned long long *keys, uint32_t n, const unsigned long long *thr, uint32_t cap,
                                                           uint32_t *count, uint32_t *slots) {
    // Hits are collected per workgroup in LDS and claim their slots with ONE global atomic (a few thousand hits per column on
    // one counter serialise in L2: 0.4 ms for a 60-us pass over the keys; counters of different columns sit CNT_PAD words apart).
    constexpr uint32_t LIST = 1024;
    __shared__ uint32_t s_n, s_base, s_list[LIST];
    const uint32_t col = blockIdx.y;
    const unsigned long long t = thr[col];
    const unsigned long long *kc = keys + (size_t)col * n;
    uint32_t *cnt = count + (size_t)col * CNT_PAD;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        if (kc[i] <= t) {
            const uint32_t p = atomicAdd(&s_n, 1u);
            if (p < LIST) {
                s_list[p] = i;
            } else {  // (a dense patch of hits: straight to the global counter)
                const uint32_t gp = atomicAdd(cnt, 1u);
                if (gp < cap) slots[(size_t)col * cap + gp] = i;
            }
        }
    __syncthreads();
    const uint32_t m = s_n < LIST ? s_n : LIST;
    if (threadIdx.x == 0 && m) s_base = atomicAdd(cnt, m);
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < m; j += blockDim.x)
        if (s_base + j < cap) slots[(size_t)col * cap + s_base + j] = s_list[j];
}
__global__ void k_gather_page_cols(const int64_t *gids, const unsigned long long *keys, uint32_t n, const uint32_t *slots, const uint32_t *count,
                                   uint32_t cap, int64_t *out_g, unsigned long long *out_k) {
    const uint32_t col = blockIdx.y;
    // (an overflowed column — count > cap — is not read back; count == cap filled its slots exactly and IS: the host applies the same
    //  predicate, pvs_rrf_pages_cols / rank_groups_page_first)
    const uint32_t m = count[(size_t)col * CNT_PAD] <= cap ? count[(size_t)col * CNT_PAD] : 0u;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) {
        const uint32_t sl = slots[(size_t)col * cap + i];
        out_g[(size_t)col * cap + i] = gids[sl];
        out_k[(size_t)col * cap + i] = keys[(size_t)col * n + sl];
    }
}
}  // namespace
pvs_status pvs_rrf_sample_keys_cols(const unsigned long long *d_keys, uint32_t n, uint32_t ncol, uint32_t m, unsigned long long *h_out, hipStream_t s) {
    unsigned long long *d = nullptr;
    HIP_TRY(pvs_scratch_alloc((void **)&d, (size_t)m * ncol * 8));
    hipLaunchKernelGGL(k_sample_keys_cols, dim3((m * ncol + 255) / 256), dim3(256), 0, s, d_keys, n, ncol, m, d);
    hipError_t e = hipMemcpyAsync(h_out, d, (size_t)m * ncol * 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    pvs_scratch_free_on(d, s);
    if (e != hipSuccess) return pvs_fail(PVS_ERR_DEVICE, "sample keys: %s", hipGetErrorString(e));
    return PVS_OK;
}
// per column: every group with key <= thr[col]; out_*: [ncol][cap] host arrays, out_count[col] may exceed cap (column not written)
pvs_status pvs_rrf_pages_cols(const unsigned long long *d_keys, const int64_t *d_gids, uint32_t n, uint32_t ncol, const unsigned long long *h_thr,
                              uint32_t cap, int64_t *out_gids, unsigned long long *out_keys, uint32_t *out_count, hipStream_t s) {
    uint32_t *d_cnt = nullptr, *d_slots = nullptr;
    unsigned long long *d_thr = nullptr, *d_k = nullptr;
    int64_t *d_g = nullptr;
    auto body = [&]() -> pvs_status {
        HIP_TRY(pvs_scratch_alloc((void **)&d_cnt, (size_t)ncol * CNT_PAD * 4));
        HIP_TRY(pvs_scratch_alloc((void **)&d_thr, (size_t)ncol * 8));
        HIP_TRY(pvs_scratch_alloc((void **)&d_slots, (size_t)ncol * cap * 4));
        HIP_TRY(pvs_scratch_alloc((void **)&d_g, (size_t)ncol * cap * 8));
        HIP_TRY(pvs_scratch_alloc((void **)&d_k, (size_t)ncol * cap * 8));
        HIP_TRY(hipMemsetAsync(d_cnt, 0, (size_t)ncol * CNT_PAD * 4, s));
        HIP_TRY(hipMemcpyAsync(d_thr, h_thr, (size_t)ncol * 8, hipMemcpyHostToDevice, s));
        const unsigned gx = (unsigned)std::min<uint32_t>((n + 255) / 256, 1024);
        hipLaunchKernelGGL(k_page_compact_cols, dim3(gx, ncol), dim3(256), 0, s, d_keys, n, d_thr, cap, d_cnt, d_slots);
        hipLaunchKernelGGL(k_gather_page_cols, dim3((cap + 255) / 256 > 64 ? 64 : (cap + 255) / 256, ncol), dim3(256), 0, s, d_gids, d_keys, n, d_slots,
                           d_cnt, cap, d_g, d_k);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpy2DAsync(out_count, 4, d_cnt, (size_t)CNT_PAD * 4, 4, ncol, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        // only the filled part of every column travels
        for (uint32_t col = 0; col < ncol; col++) {
            const uint32_t m = out_count[col] <= cap ? out_count[col] : 0u;
            if (!m) continue;
            HIP_TRY(hipMemcpyAsync(out_gids + (size_t)col * cap, d_g + (size_t)col * cap, (size_t)m * 8, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipMemcpyAsync(out_keys + (size_t)col * cap, d_k + (size_t)col * cap, (size_t)m * 8, hipMemcpyDeviceToHost, s));
        }
        HIP_TRY(hipStreamSynchronize(s));
        return PVS_OK;
    };
    pvs_status st = body();
    for (void *p : {(void *)d_cnt, (void *)d_thr, (void *)d_slots, (void *)d_g, (void *)d_k}) pvs_scratch_free_on(p, s);
    return st;
}
// candidate group ids -> window key in this branch; present[c] = 0 when the branch (shard) does not hold the group
pvs_status pvs_rrf_lookup(const unsigned long long *d_keys, const int64_t *d_gids, uint32_t n, const int64_t *cand, uint32_t m,
                          unsigned long long *out_keys, uint8_t *out_present, hipStream_t s) {
    if (m == 0) return PVS_OK;
    int64_t *d_cand = nullptr;
    uint32_t *d_slot = nullptr;
    unsigned long long *d_key = nullptr;
    auto body = [&]() -> pvs_status {
        HIP_TRY(pvs_scratch_alloc((void **)&d_cand, (size_t)m * 8));
        HIP_TRY(pvs_scratch_alloc((void **)&d_slot, (size_t)m * 4));
        HIP_TRY(pvs_scratch_alloc((void **)&d_key, (size_t)m * 8));
        HIP_TRY(hipMemcpyAsync(d_cand, cand, (size_t)m * 8, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_cand_lookup, dim3((m + 255) / 256), dim3(256), 0, s, d_gids, d_keys, n, d_cand, m, d_slot, d_key);
        HIP_TRY(hipGetLastError());
        std::vector<uint32_t> slot(m);
        HIP_TRY(hipMemcpyAsync(slot.data(), d_slot, (size_t)m * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(out_keys, d_key, (size_t)m * 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        for (uint32_t c = 0; c < m; c++) out_present[c] = slot[c] != 0xffffffffu;
        return PVS_OK;
    };
    pvs_status st = body();
    pvs_scratch_free_on(d_cand, s);
    pvs_scratch_free_on(d_slot, s);
    pvs_scratch_free_on(d_key, s);
    return st;
}
// candidates sorted ascending by (key, group id): out_below[j] = groups of this branch (shard) strictly before candidate j
pvs_status pvs_rrf_count_below(const unsigned long long *d_keys, const int64_t *d_gids, uint32_t n, const unsigned long long *ckeys,
                               const int64_t *cgids, uint32_t m, unsigned long long *out_below, hipStream_t s) {
    for (uint32_t j = 0; j < m; j++) out_below[j] = 0;
    if (m == 0 || n == 0) return PVS_OK;
    unsigned long long *d_ck = nullptr, *d_hist = nullptr;
    int64_t *d_cg = nullptr;
    const uint32_t CH = 2400;  // candidates per pass: 20 B each of LDS
    auto body = [&]() -> pvs_status {
        HIP_TRY(pvs_scratch_alloc((void **)&d_ck, (size_t)CH * 8));
        HIP_TRY(pvs_scratch_alloc((void **)&d_cg, (size_t)CH * 8));
        HIP_TRY(pvs_scratch_alloc((void **)&d_hist, ((size_t)CH + 1) * 8));
        unsigned long long carried = 0;  // groups before the first candidate of the chunk = groups before the last of the previous + ...
        (void)carried;
        for (uint32_t c0 = 0; c0 < m; c0 += CH) {
            const uint32_t mc = std::min(CH, m - c0);
            HIP_TRY(hipMemcpyAsync(d_ck, ckeys + c0, (size_t)mc * 8, hipMemcpyHostToDevice, s));
            HIP_TRY(hipMemcpyAsync(d_cg, cgids + c0, (size_t)mc * 8, hipMemcpyHostToDevice, s));
            HIP_TRY(hipMemsetAsync(d_hist, 0, ((size_t)mc + 1) * 8, s));
            const size_t lds = (size_t)mc * 16 + ((size_t)mc + 1) * 4 + 16;
            hipLaunchKernelGGL(k_rank_count, dim3(std::min<unsigned>(grid_for(n), 2048)), dim3(256), lds, s, d_keys, d_gids, n, d_ck, d_cg, mc, d_hist);
            HIP_TRY(hipGetLastError());
            std::vector<unsigned long long> h((size_t)mc + 1);
            HIP_TRY(hipMemcpyAsync(h.data(), d_hist, ((size_t)mc + 1) * 8, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            unsigned long long run = 0;
            for (uint32_t i = 0; i < mc; i++) {
                run += h[i];  // groups strictly before candidate c0 + i (every chunk counts from the beginning of the order)
                out_below[c0 + i] = run;
            }
        }
        return PVS_OK;
    };
    pvs_status st = body();
    pvs_scratch_free_on(d_ck, s);
    pvs_scratch_free_on(d_cg, s);
    pvs_scratch_free_on(d_hist, s);
    return st;
}

// ranks every group of one branch and writes its (group id, branch, rank) entries at cat_*[0..n)
pvs_status pvs_rrf_rank_branch(const double *d_vals, const int64_t *d_gids, uint32_t n, int descending, uint32_t branch,
                               unsigned long long *cat_key, unsigned long long *cat_pay, hipStream_t s) {
    if (n == 0) return PVS_OK;
    if ((uint64_t)n >= (1ull << RANK_BITS)) return pvs_fail(PVS_ERR_UNSUPPORTED, "too many groups in one branch");
    unsigned long long *k_in = nullptr, *k_out = nullptr;
    uint32_t *i_in = nullptr, *i_out = nullptr;
    void *temp = nullptr;
    auto body = [&]() -> pvs_status {
        HIP_TRY(pvs_malloc_retry((void **)&k_in, (size_t)n * 8));
        HIP_TRY(pvs_malloc_retry((void **)&k_out, (size_t)n * 8));
        HIP_TRY(pvs_malloc_retry((void **)&i_in, (size_t)n * 4));
        HIP_TRY(pvs_malloc_retry((void **)&i_out, (size_t)n * 4));
        size_t tb = 0;
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, k_in, k_out, i_in, i_out, (int)n));
        HIP_TRY(pvs_malloc_retry(&temp, tb ? tb : 16));
        hipLaunchKernelGGL(k_rank_keys, dim3(grid_for(n)), dim3(256), 0, s, d_vals, n, descending, k_in, i_in);
        // stable: equal values keep the input order = group id ascending (groups are stored in id order)
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(temp, tb, k_in, k_out, i_in, i_out, (int)n, 0, 64, s));
        hipLaunchKernelGGL(k_scatter_entries, dim3(grid_for(n)), dim3(256), 0, s, i_out, d_gids, n, branch, cat_key, cat_pay);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(s));
        return PVS_OK;
    };
    pvs_status st = body();
    hipFree(k_in);
    hipFree(k_out);
    hipFree(i_in);
    hipFree(i_out);
    hipFree(temp);
    return st;
}

// entries cat_*[0..total) of all branches (concatenated in branch order) -> first k groups by fused score
pvs_status pvs_rrf_fuse_device(unsigned long long *cat_key, unsigned long long *cat_pay, uint64_t total, const PvsRrfParams &p, uint32_t k,
                               int64_t *out_groups, double *out_scores, uint32_t *out_count, hipStream_t s) {
    int64_t *d_og = nullptr;
    double *d_os = nullptr, *score = nullptr;
    uint32_t *d_oc = nullptr, *i2 = nullptr, *i2s = nullptr;
    unsigned long long *key_s = nullptr, *pay_s = nullptr, *k2 = nullptr, *k2s = nullptr;
    void *temp = nullptr;
    if (total >= (1ull << 31)) return pvs_fail(PVS_ERR_UNSUPPORTED, "too many (group, branch) entries");
    auto body = [&]() -> pvs_status {
        HIP_TRY(pvs_malloc_retry((void **)&d_og, (size_t)k * 8));
        HIP_TRY(pvs_malloc_retry((void **)&d_os, (size_t)k * 8));
        HIP_TRY(pvs_malloc_retry((void **)&d_oc, 4));
        const size_t tn = std::max<uint64_t>(total, 1);
        HIP_TRY(pvs_malloc_retry((void **)&key_s, tn * 8));
        HIP_TRY(pvs_malloc_retry((void **)&pay_s, tn * 8));
        HIP_TRY(pvs_malloc_retry((void **)&score, tn * 8));
        HIP_TRY(pvs_malloc_retry((void **)&k2, tn * 8));
        HIP_TRY(pvs_malloc_retry((void **)&k2s, tn * 8));
        HIP_TRY(pvs_malloc_retry((void **)&i2, tn * 4));
        HIP_TRY(pvs_malloc_retry((void **)&i2s, tn * 4));
        if (total) {
            size_t t1 = 0, t2 = 0;
            HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, t1, cat_key, key_s, cat_pay, pay_s, (int)total));
            HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, t2, k2, k2s, i2, i2s, (int)total));
            HIP_TRY(pvs_malloc_retry(&temp, std::max<size_t>(std::max(t1, t2), 16)));
            // by group id, stable: a group's entries stay in branch order
            HIP_TRY(hipcub::DeviceRadixSort::SortPairs(temp, t1, cat_key, key_s, cat_pay, pay_s, (int)total, 0, 64, s));
            hipLaunchKernelGGL(k_rrf_score, dim3(grid_for(total)), dim3(256), 0, s, key_s, pay_s, total, p, score, k2, i2);
            // by score descending, stable: ties keep group id ascending
            HIP_TRY(hipcub::DeviceRadixSort::SortPairs(temp, t2, k2, k2s, i2, i2s, (int)total, 0, 64, s));
        }
        hipLaunchKernelGGL(k_rrf_emit, dim3(1), dim3(256), 0, s, k2s, i2s, key_s, score, total, k, d_og, d_os, d_oc);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(out_groups, d_og, (size_t)k * 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(out_scores, d_os, (size_t)k * 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(out_count, d_oc, 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        return PVS_OK;
    };
    pvs_status st = body();
    hipFree(d_og);
    hipFree(d_os);
    hipFree(d_oc);
    hipFree(key_s);
    hipFree(pay_s);
    hipFree(score);
    hipFree(k2);
    hipFree(k2s);
    hipFree(i2);
    hipFree(i2s);
    hipFree(temp);
    return st;
}
