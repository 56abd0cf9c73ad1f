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
}  // namespace

// ranks every group of one branch and writes its (group id, branch, rank) entries at cat_*[0..n)
pvs_status pvs_rrf_rank_branch(const double *d_vals, const int64_t *d_gids, uint32_t n, int descending, uint32_t branch,
                               unsigned long long *cat_key, unsigned long long *cat_pay, hipStream_t s) {
    if (n == 0) return PVS_OK;
    if ((uint64_t)n >= (1ull << RANK_BITS)) return pvs_fail(PVS_ERR_UNSUPPORTED, "too many groups in one branch");
    unsigned long long *k_in = nullptr, *k_out = nullptr;
    uint32_t *i_in = nullptr, *i_out = nullptr;
    void *temp = nullptr;
    auto body = [&]() -> pvs_status {
        HIP_TRY(hipMalloc((void **)&k_in, (size_t)n * 8));
        HIP_TRY(hipMalloc((void **)&k_out, (size_t)n * 8));
        HIP_TRY(hipMalloc((void **)&i_in, (size_t)n * 4));
        HIP_TRY(hipMalloc((void **)&i_out, (size_t)n * 4));
        size_t tb = 0;
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, k_in, k_out, i_in, i_out, (int)n));
        HIP_TRY(hipMalloc(&temp, tb ? tb : 16));
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
        HIP_TRY(hipMalloc((void **)&d_og, (size_t)k * 8));
        HIP_TRY(hipMalloc((void **)&d_os, (size_t)k * 8));
        HIP_TRY(hipMalloc((void **)&d_oc, 4));
        const size_t tn = std::max<uint64_t>(total, 1);
        HIP_TRY(hipMalloc((void **)&key_s, tn * 8));
        HIP_TRY(hipMalloc((void **)&pay_s, tn * 8));
        HIP_TRY(hipMalloc((void **)&score, tn * 8));
        HIP_TRY(hipMalloc((void **)&k2, tn * 8));
        HIP_TRY(hipMalloc((void **)&k2s, tn * 8));
        HIP_TRY(hipMalloc((void **)&i2, tn * 4));
        HIP_TRY(hipMalloc((void **)&i2s, tn * 4));
        if (total) {
            size_t t1 = 0, t2 = 0;
            HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, t1, cat_key, key_s, cat_pay, pay_s, (int)total));
            HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, t2, k2, k2s, i2, i2s, (int)total));
            HIP_TRY(hipMalloc(&temp, std::max<size_t>(std::max(t1, t2), 16)));
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
