// pvs_groups.hip — per-item aggregation and ranking on the device.
// Replaces the reference's  `GROUP BY file_id` + rank_aggregate (MIN / MAX / AVG / SUM(d*w)/SUM(w))
// over the materialised per-row distance (filters/exact.rs:67-134, pql/builder.rs:829-835) and the
// ORDER BY over the groups; for `similar_to` the aggregate runs over the (target vector x
// candidate vector) fan-out of the self-join (filters/item_similarity.rs:432-581).
// SQLite's SUM/AVG are Kahan-Babuska-Neumaier compensated f64 sums taken in row order; one lane
// walks one group in that fixed order, so the f64 results are bit-identical to the oracle.
#include <hipcub/hipcub.hpp>

#include "pvs_kernels.hpp"

struct Kbn {
    double s = 0.0, c = 0.0;
    __device__ inline void step(double r) {
        const double t = s + r;
        if (fabs(s) > fabs(r))
            c += (s - t) + r;
        else
            c += (r - t) + s;
        s = t;
    }
    __device__ inline double value() const { return s + c; }
};

// dist: [n_rows][ld] (query-minor).  One lane per (group, output column).
//   fanout == 0: column q aggregates dist[row][q] over the group's rows (semantic search)
//   fanout == M: a single output column aggregates dist[row][0..M) over rows and targets
//                (similar_to), skipping rows flagged in `exclude`.
//   fw.on (similar_to with confidence weights, item_similarity.rs:503-581): the pair weight
//                w = pow(coalesce(conf_t,1)*coalesce(conf_o,1), cw) * pow(coalesce(lang_o,1)*coalesce(lang_t,1), lw)
//                (a factor is dropped when its exponent is 0) and the value is SUM(d*w)/SUM(w).
__device__ static inline double coalesce1(double v) { return v != v ? 1.0 : v; }
// One value per (group, output column).  A workgroup owns a tile of TG consecutive groups x all output columns: it walks the
// tile column-fastest (adjacent lanes read adjacent floats of one row of `dist`) and hands the values to the column-major output
// through LDS, group-fastest (the direct store — adjacent lanes 8 B apart in columns that lie n_groups * 8 B apart — was one
// partial cache line per lane: 1.1 ms for 4M rows x 32 queries, the largest kernel of the dense per-item search).
// groups per workgroup tile: 32, or more when there are few columns (a tile is at least 256 values)
static inline uint32_t agg_tile_groups(uint32_t ncol_out) { return ncol_out >= 8 ? 32u : 256u / ncol_out; }
__device__ static inline double group_value(const float *dist, uint32_t ld, uint32_t n_cols, uint32_t fanout, const uint32_t *grp_off,
                                            const uint32_t *grp_rows, const float *weights, const uint8_t *exclude, int agg, const FanoutWeights &fw,
                                            uint32_t skip_when, uint32_t g, uint32_t q) {
    Kbn sum, wsum;
    double mn = __builtin_inf(), mx = -__builtin_inf();
    uint64_t cnt = 0, joined = 0;  // joined: (row, target) pairs that are part of the join at all
    const uint32_t e_begin = grp_off[g], e_end = grp_off[g + 1];
    uint32_t e_slow = e_begin;
    if (!fanout && !fw.on) {
        // per-item search (one column per query): the loads of four rows of the group go out together — the one-row-at-a-time
        // loop is a chain of two dependent loads per row (row index, then its distance), which is all this kernel waits for;
        // the sums still run in row order
        for (uint32_t e = e_begin; e < e_end; e += 4) {
            const uint32_t m = e_end - e < 4 ? e_end - e : 4;
            uint32_t rw[4];
            float df4[4], w4[4];
            uint8_t ex4[4];
#pragma unroll
            for (int i = 0; i < 4; i++) rw[i] = (uint32_t)i < m ? grp_rows[e + i] : 0u;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const bool on = (uint32_t)i < m;
                df4[i] = on ? dist[(size_t)rw[i] * ld + q] : 0.f;
                w4[i] = on && weights ? weights[rw[i]] : 1.f;
                ex4[i] = on && exclude ? exclude[rw[i]] : (uint8_t)0;
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if ((uint32_t)i >= m) break;
                if (exclude && (uint32_t)(ex4[i] != 0) == skip_when) continue;
                joined++;
                const double w = (double)w4[i];
                if (weights) wsum.step(w);
                if (df4[i] != df4[i]) continue;
                const double d = (double)df4[i];
                sum.step(weights ? d * w : d);
                mn = fmin(mn, d);
                mx = fmax(mx, d);
                cnt++;
            }
        }
        e_slow = e_end;
    }
    for (uint32_t e = e_slow; e < e_end; e++) {
        const uint32_t row = grp_rows[e];
        if (exclude && (uint32_t)(exclude[row] != 0) == skip_when) continue;  // similar_to: flagged rows; candidate mask: rows it leaves out
        const uint32_t c0 = fanout ? 0u : q, c1 = fanout ? fanout : q + 1;
        for (uint32_t c = c0; c < c1; c++) {
            double w = 1.0;
            if (fw.on && fw.kind) {  // cross-modal gates: the pair is not part of the join at all
                const uint8_t km = fw.t_kind[c], ko = fw.kind[row];
                if ((fw.skip_i2i && km == 0 && ko == 0) || (fw.skip_t2t && km == 1 && ko == 1)) continue;
            }
            joined++;
            if (fw.on && (fw.cw != 0.0 || fw.lw != 0.0)) {
                if (fw.cw != 0.0) w = pow(coalesce1(fw.t_conf[c]) * coalesce1(fw.conf[row]), fw.cw);
                if (fw.lw != 0.0) {
                    const double wl = pow(coalesce1(fw.lang[row]) * coalesce1(fw.t_lang[c]), fw.lw);
                    w = fw.cw != 0.0 ? w * wl : wl;
                }
                wsum.step(w);  // SUM(w) runs over every joined pair
            } else if (weights) {
                w = (double)weights[row];
                wsum.step(w);
            }
            const float df = dist[(size_t)row * ld + c];
            if (df != df) continue;  // SQL NULL distance: d (and d*w) is ignored by the aggregates
            const double d = (double)df;
            if (weights || (fw.on && (fw.cw != 0.0 || fw.lw != 0.0))) {
                sum.step(d * w);
            } else {
                sum.step(d);
            }
            mn = fmin(mn, d);
            mx = fmax(mx, d);
            cnt++;
        }
    }
    double v;
    if (joined == 0)
        // no candidate row under the mask / every pair of the similar_to join excluded or gated away (INNER JOIN +
        // WHERE, item_similarity.rs:445-489): the group is not part of the result at all
        v = __builtin_bit_cast(double, PVS_GROUP_ABSENT);
    else if (cnt == 0)
        v = __builtin_nan("");
    else if (weights || (fw.on && (fw.cw != 0.0 || fw.lw != 0.0)))
        v = sum.value() / wsum.value();
    else if (agg == PVS_AGG_MIN)
        v = mn;
    else if (agg == PVS_AGG_MAX)
        v = mx;
    else
        v = sum.value() / (double)cnt;
    return v;
}
__global__ __launch_bounds__(256) void k_group_aggregate(const float *dist, uint32_t ld, uint32_t n_cols, uint32_t fanout,
                                                         const uint32_t *grp_off, const uint32_t *grp_rows, uint32_t n_groups,
                                                         const float *weights, const uint8_t *exclude, int agg, FanoutWeights fw,
                                                         double *out, uint32_t skip_when, uint32_t TG) {
    extern __shared__ double s_tile[];  // [ncol_out][TG]
    const uint32_t ncol_out = fanout ? 1u : n_cols;
    const uint32_t g0 = blockIdx.x * TG;
    const uint32_t tg = n_groups - g0 < TG ? n_groups - g0 : TG;
    for (uint32_t idx = threadIdx.x; idx < tg * ncol_out; idx += 256) {
        const uint32_t gl = idx / ncol_out, q = idx % ncol_out;
        s_tile[q * TG + gl] = group_value(dist, ld, n_cols, fanout, grp_off, grp_rows, weights, exclude, agg, fw, skip_when, g0 + gl, q);
    }
    __syncthreads();
    for (uint32_t idx = threadIdx.x; idx < TG * ncol_out; idx += 256) {
        const uint32_t q = idx / TG, gl = idx % TG;
        if (gl < tg) out[(size_t)q * n_groups + g0 + gl] = s_tile[idx];
    }
}

hipError_t pvs_launch_group_aggregate(const float *dist, uint32_t ld, uint32_t n_cols, uint32_t fanout, const uint32_t *grp_off,
                                      const uint32_t *grp_rows, uint32_t n_groups, const float *weights, const uint8_t *exclude,
                                      int agg, double *out, hipStream_t s, FanoutWeights fw, uint32_t skip_when) {
    if (n_groups == 0) return hipSuccess;
    const uint32_t ncol_out = fanout ? 1u : n_cols;
    const uint32_t TG = agg_tile_groups(ncol_out);
    hipLaunchKernelGGL(k_group_aggregate, dim3((n_groups + TG - 1) / TG), dim3(256), (size_t)ncol_out * TG * 8, s, dist, ld, n_cols, fanout,
                       grp_off, grp_rows, n_groups, weights, exclude, agg, fw, out, skip_when, TG);
    return hipGetLastError();
}

// order-preserving f64 -> u64, NaN last
__device__ static inline unsigned long long f64_sort_key(double d) {
    unsigned long long b = __builtin_bit_cast(unsigned long long, d);
    if ((b & 0x7fffffffffffffffull) > 0x7ff0000000000000ull) return ~0ull;
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
// g_tinv (optional): the groups in tie order (second sort key DESC, group id ASC); the value sort is stable, so equal values come
// out in that order — without it in group id order
__global__ void k_group_keys(const double *vals, const uint32_t *g_tinv, uint32_t n, unsigned long long *keys, uint32_t *idx) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t g = g_tinv ? g_tinv[i] : i;
        unsigned long long k = f64_sort_key(vals[g]);
        if (k == ~0ull) k = ~0ull - 1;  // NULL aggregates: after every value ...
        if (__builtin_bit_cast(unsigned long long, vals[g]) == PVS_GROUP_ABSENT) k = ~0ull;  // ... absent groups: never emitted
        keys[i] = k;
        idx[i] = g;
    }
}
__global__ void k_group_emit(const uint32_t *idx_sorted, const double *vals, const int64_t *group_ids, uint32_t n, uint32_t k,
                             int64_t *out_groups, double *out_vals, uint32_t *out_count) {
    __shared__ uint32_t s_nout;
    if (threadIdx.x == 0) {  // absent groups (no candidate row under the mask) sort last: the live ones are a prefix
        uint32_t lo = 0, hi = n < k ? n : k;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) / 2;
            if (__builtin_bit_cast(unsigned long long, vals[idx_sorted[mid]]) != PVS_GROUP_ABSENT)
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
            out_groups[i] = group_ids[idx_sorted[i]];
            out_vals[i] = vals[idx_sorted[i]];
        } else {
            out_groups[i] = -1;
            out_vals[i] = __builtin_nan("");
        }
    }
    if (threadIdx.x == 0) *out_count = nout;
}

// the same keys in group order (no permutation): input of the page-first ranking (pvs_group_page_keys)
__global__ void k_group_page_keys(const double *vals, uint32_t n, unsigned long long *keys) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        unsigned long long k = f64_sort_key(vals[i]);
        if (k == ~0ull) k = ~0ull - 1;
        if (__builtin_bit_cast(unsigned long long, vals[i]) == PVS_GROUP_ABSENT) k = ~0ull;
        keys[i] = k;
    }
}
hipError_t pvs_group_page_keys(const double *d_vals, uint32_t n_groups, unsigned long long *d_keys, hipStream_t s) {
    if (n_groups == 0) return hipSuccess;
    hipLaunchKernelGGL(k_group_page_keys, dim3((n_groups + 255) / 256 > 4096 ? 4096 : (n_groups + 255) / 256), dim3(256), 0, s, d_vals, n_groups, d_keys);
    return hipGetLastError();
}

// ranks one column of group values: (value asc, group id asc — groups are stored in id order and
// the radix sort is stable), NaN last; writes the first k.
pvs_status pvs_group_rank(const double *d_vals, const int64_t *d_group_ids, uint32_t n_groups, uint32_t k, GroupWork &w,
                          int64_t *d_out_groups, double *d_out_vals, uint32_t *d_out_count, hipStream_t s, const uint32_t *g_tinv) {
    if (n_groups > w.cap) {
        hipFree(w.keys_in);
        hipFree(w.keys_out);
        hipFree(w.idx_in);
        hipFree(w.idx_out);
        hipFree(w.temp);
        w = GroupWork();
        const uint32_t cap = (uint32_t)pvs_round_up(n_groups, 1024);
        HIP_TRY(pvs_malloc_retry((void **)&w.keys_in, (size_t)cap * 8));
        HIP_TRY(pvs_malloc_retry((void **)&w.keys_out, (size_t)cap * 8));
        HIP_TRY(pvs_malloc_retry((void **)&w.idx_in, (size_t)cap * 4));
        HIP_TRY(pvs_malloc_retry((void **)&w.idx_out, (size_t)cap * 4));
        size_t tb = 0;
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, w.keys_in, w.keys_out, w.idx_in, w.idx_out, (int)cap));
        HIP_TRY(pvs_malloc_retry(&w.temp, tb ? tb : 16));
        w.temp_bytes = tb;
        w.cap = cap;
    }
    if (n_groups) {
        hipLaunchKernelGGL(k_group_keys, dim3((n_groups + 255) / 256 > 4096 ? 4096 : (n_groups + 255) / 256), dim3(256), 0, s, d_vals,
                           g_tinv, n_groups, w.keys_in, w.idx_in);
        size_t tb = w.temp_bytes;
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(w.temp, tb, w.keys_in, w.keys_out, w.idx_in, w.idx_out, (int)n_groups, 0, 64, s));
    }
    hipLaunchKernelGGL(k_group_emit, dim3(1), dim3(256), 0, s, w.idx_out, d_vals, d_group_ids, n_groups, k, d_out_groups, d_out_vals,
                       d_out_count);
    HIP_TRY(hipGetLastError());
    return PVS_OK;
}

void pvs_group_work_release(GroupWork &w) {
    hipFree(w.keys_in);
    hipFree(w.keys_out);
    hipFree(w.idx_in);
    hipFree(w.idx_out);
    hipFree(w.temp);
    w = GroupWork();
}
