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
__global__ void k_dense_keys(const float *dist, const uint8_t *mask, uint64_t n, uint32_t *keys, uint32_t *vals, DenseBounds b) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t k = f32_sort_key(dist[i]);
        if (k == 0xffffffffu) k = 0xfffffffeu;
        if (mask && !mask[i]) k = 0xffffffffu;
        // apply_sort_bounds (builder.rs:781-815): order_rank is the f32 distance widened to a SQL REAL; NULL fails a comparison
        const double d = (double)dist[i];
        if (b.have_gt && !(d > b.gt)) k = 0xffffffffu;
        if (b.have_lt && !(d < b.lt)) k = 0xffffffffu;
        keys[i] = k;
        vals[i] = (uint32_t)i;
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
                          uint32_t *out_count, hipStream_t s, const uint8_t *mask, DenseBounds bounds) {
    if (n > w.cap_rows) return pvs_fail(PVS_ERR_STATE, "dense workspace too small");
    if (n > 0x7fffffffull) return pvs_fail(PVS_ERR_UNSUPPORTED, "dense path limited to 2^31-1 rows per shard");
    if (n > 0) {
        unsigned g = (unsigned)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256);
        hipLaunchKernelGGL(k_dense_keys, dim3(g), dim3(256), 0, s, w.d_dist, mask, n, w.d_keys_in, w.d_vals_in, bounds);
        size_t tb = w.temp_bytes;
        // stable LSD radix sort: equal distances keep ascending row order = ascending id
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(w.d_temp, tb, w.d_keys_in, w.d_keys_out, w.d_vals_in, w.d_vals_out, (int)n, 0,
                                                   32, s));
    }
    hipLaunchKernelGGL(k_dense_emit, dim3(1), dim3(256), 0, s, w.d_keys_out, w.d_vals_out, n, k, ids, out_ids, out_dist, out_count);
    HIP_TRY(hipGetLastError());
    return PVS_OK;
}
