// pvs_dense.hip — the dense path: every row scored exactly (k_score_all), then a
// full stable device radix sort — literally what the reference does (score all
// candidates, sort everything, LIMIT k; docs/vector-index-design.md:84-89), moved to
// HBM.  It serves f32 indexes, dimensions the MFMA scan has no instance for, and
// every query the filter path hands back (overflow, NULL distances needed to fill
// the page).  rocPRIM/hipCUB are header-only parts of ROCm.
#include <hipcub/hipcub.hpp>

#include "pvs_kernels.hpp"

// keys: real distances ascending, then NULL distances (0xfffffffe), then rows outside the candidate set
// (mask[i] == 0 -> 0xffffffff: sorted to the very end and never emitted)
// tinv (optional): the rows in tie order (pvs_index_set_order_keys: key DESC, id ASC).  The sort below is stable, so feeding it
// the rows in that order makes equal distances come out in that order; without it, in row (= id) order.
__global__ void k_dense_keys(const float *dist, const uint8_t *mask, const uint32_t *tinv, uint64_t n, uint32_t *keys, uint32_t *vals, DenseBounds b) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t row = tinv ? tinv[i] : (uint32_t)i;
        uint32_t k = f32_sort_key(dist[row]);
        if (k == 0xffffffffu) k = 0xfffffffeu;
        if (mask && !mask[row]) k = 0xffffffffu;
        // apply_sort_bounds (builder.rs:781-815): order_rank is the f32 distance widened to a SQL REAL; NULL fails a comparison
        const double d = (double)dist[row];
        if (b.have_gt && !(d > b.gt)) k = 0xffffffffu;
        if (b.have_lt && !(d < b.lt)) k = 0xffffffffu;
        keys[i] = k;
        vals[i] = row;
    }
}
__global__ void k_dense_emit(const uint32_t *keys, const uint32_t *vals, uint64_t n, uint32_t k, const int64_t *ids,
                             int64_t *out_ids, float *out_dist, uint32_t *out_count) {
    __shared__ uint32_t s_nout;
    if (threadIdx.x == 0) {
        uint32_t lo = 0, hi = n < k ? (uint32_t)n : k;  // live rows are a prefix of the sorted order
        while (lo < hi) {
            const uint32_t mid = (lo + hi) / 2;
            if (keys[mid] != 0xffffffffu)
                lo = mid + 1;
            else
                hi = mid;
        }
        s_nout = lo;
    }
    __syncthreads();
    const uint32_t nout = s_nout;
    for (uint32_t i = threadIdx.x; i < k; i += blockDim.x) {
        if (i < nout) {
            out_ids[i] = ids[vals[i]];
            out_dist[i] = f32_from_sort_key(keys[i]);
        } else {
            out_ids[i] = -1;
            out_dist[i] = __builtin_nanf("");
        }
    }
    if (threadIdx.x == 0) *out_count = nout;
}

void pvs_dense_release(DenseWork &w) {
    hipFree(w.d_dist);
    hipFree(w.d_keys_in);
    hipFree(w.d_keys_out);
    hipFree(w.d_vals_in);
    hipFree(w.d_vals_out);
    hipFree(w.d_temp);
    w = DenseWork();
}

pvs_status pvs_dense_reserve(DenseWork &w, uint64_t n) {
    if (n <= w.cap_rows) return PVS_OK;
    pvs_dense_release(w);
    const uint64_t cap = pvs_round_up(n, 1024);
    HIP_TRY(pvs_malloc_retry((void **)&w.d_dist, cap * 4));
    HIP_TRY(pvs_malloc_retry((void **)&w.d_keys_in, cap * 4));
    HIP_TRY(pvs_malloc_retry((void **)&w.d_keys_out, cap * 4));
    HIP_TRY(pvs_malloc_retry((void **)&w.d_vals_in, cap * 4));
    HIP_TRY(pvs_malloc_retry((void **)&w.d_vals_out, cap * 4));
    size_t tb = 0;
    HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, w.d_keys_in, w.d_keys_out, w.d_vals_in, w.d_vals_out, (int)cap));
    HIP_TRY(pvs_malloc_retry(&w.d_temp, tb ? tb : 16));
    w.temp_bytes = tb;
    w.cap_rows = cap;
    return PVS_OK;
}

pvs_status pvs_dense_topk(DenseWork &w, uint64_t n, uint32_t k, const int64_t *ids, int64_t *out_ids, float *out_dist,
                          uint32_t *out_count, hipStream_t s, const uint8_t *mask, DenseBounds bounds, const uint32_t *tinv) {
    if (n > w.cap_rows) return pvs_fail(PVS_ERR_STATE, "dense workspace too small");
    if (n > 0x7fffffffull) return pvs_fail(PVS_ERR_UNSUPPORTED, "dense path limited to 2^31-1 rows per shard");
    if (n > 0) {
        unsigned g = (unsigned)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256);
        hipLaunchKernelGGL(k_dense_keys, dim3(g), dim3(256), 0, s, w.d_dist, mask, tinv, n, w.d_keys_in, w.d_vals_in, bounds);
        size_t tb = w.temp_bytes;
        // stable LSD radix sort: equal distances keep ascending row order = ascending id
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(w.d_temp, tb, w.d_keys_in, w.d_keys_out, w.d_vals_in, w.d_vals_out, (int)n, 0,
                                                   32, s));
    }
    hipLaunchKernelGGL(k_dense_emit, dim3(1), dim3(256), 0, s, w.d_keys_out, w.d_vals_out, n, k, ids, out_ids, out_dist, out_count);
    HIP_TRY(hipGetLastError());
    return PVS_OK;
}

// ---- tie ranks of the second sort key (pvs_index_set_order_keys): rows ordered by (key DESC, row ASC)
__global__ void k_tie_sort_keys(const int64_t *keys, uint64_t n, unsigned long long *skeys, uint32_t *rows) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        skeys[i] = ~((unsigned long long)keys[i] ^ 0x8000000000000000ull);  // order-preserving image of the int64, inverted: larger key first
        rows[i] = (uint32_t)i;
    }
}
__global__ void k_tie_invert(const uint32_t *tinv, uint64_t n, uint32_t *trank) {
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (uint64_t)gridDim.x * blockDim.x) trank[tinv[t]] = (uint32_t)t;
}
pvs_status pvs_build_tie_ranks(const int64_t *d_keys, uint64_t n, uint32_t *d_trank, uint32_t *d_tinv, hipStream_t s) {
    if (n == 0) return PVS_OK;
    if (n > 0x7fffffffull) return pvs_fail(PVS_ERR_UNSUPPORTED, "order keys are limited to 2^31-1 rows per index");
    unsigned long long *k_in = nullptr, *k_out = nullptr;
    uint32_t *r_in = nullptr;
    void *tmp = nullptr;
    auto body = [&]() -> pvs_status {
        HIP_TRY(pvs_scratch_alloc((void **)&k_in, n * 8));
        HIP_TRY(pvs_scratch_alloc((void **)&k_out, n * 8));
        HIP_TRY(pvs_scratch_alloc((void **)&r_in, n * 4));
        const unsigned g = (unsigned)std::min<uint64_t>((n + 255) / 256, 8192);
        hipLaunchKernelGGL(k_tie_sort_keys, dim3(g), dim3(256), 0, s, d_keys, n, k_in, r_in);
        size_t tb = 0;
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, k_in, k_out, r_in, d_tinv, (int)n));
        HIP_TRY(pvs_scratch_alloc(&tmp, tb ? tb : 16));
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(tmp, tb, k_in, k_out, r_in, d_tinv, (int)n, 0, 64, s));  // stable: equal keys stay in row order
        hipLaunchKernelGGL(k_tie_invert, dim3(g), dim3(256), 0, s, d_tinv, n, d_trank);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(s));
        return PVS_OK;
    };
    pvs_status st = body();
    for (void *p : {(void *)k_in, (void *)k_out, (void *)r_in, tmp}) pvs_scratch_free_on(p, s);
    return st;
}
