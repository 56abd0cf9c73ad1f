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

// ---- page first (round 5).  The full sort orders every row to return k of them: 10M rows -> 8x the filter path for one deep page
// (`LIMIT ? OFFSET ?` far into a result, pql/builder.rs:578-582; a lower sort bound deep in the ordering, builder.rs:781-815).
// Instead: the keys of 65,536 evenly spaced rows are sorted, the key at the sample rank that expects ~1.25 k rows (+ 4 sigma)
// becomes a threshold T, ONE pass over the distance column appends every live row with key <= T as (key << 32 | walk position)
// — ALL of them, so whenever at least k came back the k smallest are among them — and those few thousand 64-bit words are sorted:
// (key, walk position) is exactly the order the stable full sort produces.  Fewer than k admitted (an unlucky sample, fewer than k
// live rows) or more than the list holds (massive ties at T): the full sort answers as before.
constexpr uint32_t DENSE_SAMPLE = 65536;
__device__ static inline uint32_t dense_key(const float *dist, const uint8_t *mask, uint32_t row, const DenseBounds &b) {
    uint32_t k = f32_sort_key(dist[row]);
    if (k == 0xffffffffu) k = 0xfffffffeu;
    if (mask && !mask[row]) k = 0xffffffffu;
    const double d = (double)dist[row];
    if (b.have_gt && !(d > b.gt)) k = 0xffffffffu;
    if (b.have_lt && !(d < b.lt)) k = 0xffffffffu;
    return k;
}
__global__ void k_dense_sample(const float *dist, const uint8_t *mask, const uint32_t *tinv, uint64_t n, uint32_t *sample, DenseBounds b) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= DENSE_SAMPLE) return;
    const uint64_t i = (uint64_t)j * n / DENSE_SAMPLE;
    sample[j] = dense_key(dist, mask, tinv ? tinv[i] : (uint32_t)i, b);
}
__global__ void k_dense_threshold(const uint32_t *sorted, uint32_t idx, uint32_t *ctl) {
    uint32_t t = sorted[idx];
    if (t == 0xffffffffu) t = 0xfffffffeu;  // (rows outside the mask / the bounds are never admitted)
    ctl[0] = t;
    ctl[1] = 0;
}
__global__ __launch_bounds__(256) void k_dense_admit(const float *dist, const uint8_t *mask, const uint32_t *tinv, uint64_t n, DenseBounds b, uint32_t *ctl, unsigned long long *comp,
                                                     uint32_t cap) {
    const uint32_t T = ctl[0];
    const uint32_t lane = threadIdx.x & 63u;
    for (uint64_t i0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) & ~63ull; i0 < n; i0 += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t i = i0 + lane;
        uint32_t key = 0xffffffffu;
        if (i < n) key = dense_key(dist, mask, tinv ? tinv[i] : (uint32_t)i, b);
        const bool in = key <= T;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(in);
        if (m) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(ctl + 1, (uint32_t)__popcll(m));
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            const uint32_t at = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (in && at < cap) comp[at] = ((unsigned long long)key << 32) | (uint32_t)i;
        }
    }
}
__global__ void k_dense_emit_page(const unsigned long long *sorted, uint32_t m, uint32_t k, const uint32_t *tinv, const int64_t *ids, int64_t *out_ids, float *out_dist,
                                  uint32_t *out_count) {
    const uint32_t nout = m < k ? m : k;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < k; i += gridDim.x * blockDim.x) {
        if (i < nout) {
            const unsigned long long v = sorted[i];
            const uint32_t pos = (uint32_t)v;
            out_ids[i] = ids[tinv ? tinv[pos] : pos];
            out_dist[i] = f32_from_sort_key((uint32_t)(v >> 32));
        } else {
            out_ids[i] = -1;
            out_dist[i] = __builtin_nanf("");
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *out_count = nout;
}

void pvs_dense_release(DenseWork &w) {
    hipFree(w.d_dist);
    hipFree(w.d_keys_in);
    hipFree(w.d_keys_out);
    hipFree(w.d_vals_in);
    hipFree(w.d_vals_out);
    hipFree(w.d_temp);
    hipFree(w.d_sample);
    hipFree(w.d_sample_out);
    hipFree(w.d_ctl);
    if (w.h_ctl) hipHostFree(w.h_ctl);
    hipFree(w.d_comp);
    hipFree(w.d_comp_out);
    hipFree(w.d_temp_page);
    w = DenseWork();
}

pvs_status pvs_dense_reserve(DenseWork &w, uint64_t n) {
    if (n <= w.cap_rows) return PVS_OK;
    pvs_dense_release(w);
    const uint64_t cap = pvs_round_up(n, 1024);
    HIP_TRY(pvs_malloc_retry((void **)&w.d_dist, cap * 4));
    w.cap_rows = cap;
    return PVS_OK;
}
// the full sort's buffers: four words per row, allocated by the first page that needs them
static pvs_status dense_reserve_sort(DenseWork &w) {
    if (w.sort_rows >= w.cap_rows) return PVS_OK;
    for (void *p : {(void *)w.d_keys_in, (void *)w.d_keys_out, (void *)w.d_vals_in, (void *)w.d_vals_out, w.d_temp}) hipFree(p);
    w.d_keys_in = w.d_keys_out = w.d_vals_in = w.d_vals_out = nullptr;
    w.d_temp = nullptr;
    w.sort_rows = 0;
    const uint64_t cap = w.cap_rows;
    HIP_TRY(pvs_malloc_retry((void **)&w.d_keys_in, cap * 4));
    HIP_TRY(pvs_malloc_retry((void **)&w.d_keys_out, cap * 4));
    HIP_TRY(pvs_malloc_retry((void **)&w.d_vals_in, cap * 4));
    HIP_TRY(pvs_malloc_retry((void **)&w.d_vals_out, cap * 4));
    size_t tb = 0;
    HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, w.d_keys_in, w.d_keys_out, w.d_vals_in, w.d_vals_out, (int)cap));
    HIP_TRY(pvs_malloc_retry(&w.d_temp, tb ? tb : 16));
    w.temp_bytes = tb;
    w.sort_rows = cap;
    return PVS_OK;
}
static pvs_status dense_reserve_page(DenseWork &w, uint32_t comp_cap) {
    if (w.d_sample && w.comp_cap >= comp_cap) return PVS_OK;
    for (void *p : {(void *)w.d_comp, (void *)w.d_comp_out, w.d_temp_page}) hipFree(p);
    w.d_comp = w.d_comp_out = nullptr;
    w.d_temp_page = nullptr;
    w.comp_cap = 0;
    if (!w.d_sample) {
        HIP_TRY(pvs_malloc_retry((void **)&w.d_sample, DENSE_SAMPLE * 4));
        HIP_TRY(pvs_malloc_retry((void **)&w.d_sample_out, DENSE_SAMPLE * 4));
        HIP_TRY(pvs_malloc_retry((void **)&w.d_ctl, 64));
        HIP_TRY(hipHostMalloc((void **)&w.h_ctl, 64, hipHostMallocDefault));
    }
    HIP_TRY(pvs_malloc_retry((void **)&w.d_comp, (size_t)comp_cap * 8));
    HIP_TRY(pvs_malloc_retry((void **)&w.d_comp_out, (size_t)comp_cap * 8));
    size_t ta = 0, tb = 0;
    HIP_TRY(hipcub::DeviceRadixSort::SortKeys(nullptr, ta, w.d_sample, w.d_sample_out, (int)DENSE_SAMPLE));
    HIP_TRY(hipcub::DeviceRadixSort::SortKeys(nullptr, tb, w.d_comp, w.d_comp_out, (int)comp_cap));
    w.temp_page_bytes = std::max(ta, tb);
    HIP_TRY(pvs_malloc_retry(&w.d_temp_page, w.temp_page_bytes ? w.temp_page_bytes : 16));
    w.comp_cap = comp_cap;
    return PVS_OK;
}

pvs_status pvs_dense_topk(DenseWork &w, uint64_t n, uint32_t k, const int64_t *ids, int64_t *out_ids, float *out_dist,
                          uint32_t *out_count, hipStream_t s, const uint8_t *mask, DenseBounds bounds, const uint32_t *tinv) {
    if (n > w.cap_rows) return pvs_fail(PVS_ERR_STATE, "dense workspace too small");
    if (n > 0x7fffffffull) return pvs_fail(PVS_ERR_UNSUPPORTED, "dense path limited to 2^31-1 rows per shard");
    if (n >= 4ull * DENSE_SAMPLE && (uint64_t)k * 8 <= n && !pvs_dbg(PVS_DBG_DENSE_FULL_SORT)) {
        // expected sample rank of the k-th row + 4 sigma + 8
        const double er = (double)k * DENSE_SAMPLE / (double)n;
        const uint32_t idx = (uint32_t)std::min<double>(DENSE_SAMPLE - 1, std::ceil(er + 4.0 * std::sqrt(er) + 8.0));
        const uint32_t expect = (uint32_t)std::min<double>((double)n, (double)(idx + 1) * (double)n / DENSE_SAMPLE);
        const uint32_t cap = (uint32_t)pvs_round_up(std::max<uint64_t>(4ull * expect, 65536), 1024);
        PVS_TRY(dense_reserve_page(w, cap));
        hipLaunchKernelGGL(k_dense_sample, dim3(DENSE_SAMPLE / 256), dim3(256), 0, s, w.d_dist, mask, tinv, n, w.d_sample, bounds);
        size_t tb = w.temp_page_bytes;
        HIP_TRY(hipcub::DeviceRadixSort::SortKeys(w.d_temp_page, tb, w.d_sample, w.d_sample_out, (int)DENSE_SAMPLE, 0, 32, s));
        hipLaunchKernelGGL(k_dense_threshold, dim3(1), dim3(1), 0, s, w.d_sample_out, idx, w.d_ctl);
        hipLaunchKernelGGL(k_dense_admit, dim3((unsigned)std::min<uint64_t>((n + 255) / 256, 4096)), dim3(256), 0, s, w.d_dist, mask, tinv, n, bounds, w.d_ctl, w.d_comp, w.comp_cap);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(w.h_ctl, w.d_ctl, 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        const uint32_t admitted = w.h_ctl[1];
        if (admitted >= k && admitted <= w.comp_cap) {
            tb = w.temp_page_bytes;
            HIP_TRY(hipcub::DeviceRadixSort::SortKeys(w.d_temp_page, tb, w.d_comp, w.d_comp_out, (int)admitted, 0, 64, s));
            hipLaunchKernelGGL(k_dense_emit_page, dim3((k + 255) / 256), dim3(256), 0, s, w.d_comp_out, admitted, k, tinv, ids, out_ids, out_dist, out_count);
            HIP_TRY(hipGetLastError());
            pvs_dbg_add(PVS_DBG_DENSE_PAGE_FIRST, 1);
            return PVS_OK;
        }
    }
    PVS_TRY(dense_reserve_sort(w));
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
