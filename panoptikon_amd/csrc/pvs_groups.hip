// pvs_groups.hip — per-item aggregation and ranking on the device.
// Replaces the reference's  `GROUP BY file_id` + rank_aggregate (MIN / MAX / AVG / SUM(d*w)/SUM(w))
// over the materialised per-row distance (filters/exact.rs:67-134, pql/builder.rs:829-835) and the
// ORDER BY over the groups; for `similar_to` the aggregate runs over the (target vector x
// candidate vector) fan-out of the self-join (filters/item_similarity.rs:432-581).
// SQLite's SUM/AVG are Kahan-Babuska-Neumaier compensated f64 sums taken in row order; one lane
// walks one group in that fixed order, so the f64 results are bit-identical to the oracle.
#include <hipcub/hipcub.hpp>

#include "pvs_kernels.hpp"
#include "pvs_wg_select.hpp"

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
__device__ static inline bool in_left_out_ranges(const FanoutWeights &fw, uint32_t row) {
    bool in = false;
#pragma unroll
    for (uint32_t i = 0; i < 4; i++) in = in || (i < fw.n_ranges && row >= fw.r_lo[i] && row <= fw.r_hi[i]);
    return in;
}
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
    if (fanout && fanout <= 8 && !fw.on) {
        // similar_to without pair weights: the same batching for the fan-out join — four rows' indices, flags and up to 8 distances
        // each go out together (the row-at-a-time loop below: three dependent loads per row, 32 us for 86k groups of 8 rows);
        // the sums run in (row, target) order as before
        for (uint32_t e = e_begin; e < e_end; e += 4) {
            const uint32_t m = e_end - e < 4 ? e_end - e : 4;
            uint32_t rw[4];
            float d4[4][8], w4[4];
            uint8_t ex4[4];
#pragma unroll
            for (int i = 0; i < 4; i++) rw[i] = (uint32_t)i < m ? grp_rows[e + i] : 0u;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const bool on = (uint32_t)i < m;
                w4[i] = on && weights ? weights[rw[i]] : 1.f;
                ex4[i] = on && exclude ? exclude[rw[i]] : (uint8_t)0;
#pragma unroll
                for (int c = 0; c < 8; c++) d4[i][c] = on && (uint32_t)c < fanout ? dist[(size_t)rw[i] * ld + c] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if ((uint32_t)i >= m) break;
                if (exclude && (uint32_t)(ex4[i] != 0) == skip_when) continue;
                if (fw.n_ranges && in_left_out_ranges(fw, rw[i])) continue;
                const double w = (double)w4[i];
#pragma unroll
                for (int c = 0; c < 8; c++) {
                    if ((uint32_t)c >= fanout) break;
                    joined++;
                    if (weights) wsum.step(w);
                    const float df = d4[i][c];
                    if (df != df) continue;
                    const double d = (double)df;
                    sum.step(weights ? d * w : d);
                    mn = fmin(mn, d);
                    mx = fmax(mx, d);
                    cnt++;
                }
            }
        }
        e_slow = e_end;
    }
    for (uint32_t e = e_slow; e < e_end; e++) {
        const uint32_t row = grp_rows[e];
        if (exclude && (uint32_t)(exclude[row] != 0) == skip_when) continue;  // similar_to: flagged rows; candidate mask: rows it leaves out
        if (fw.n_ranges && in_left_out_ranges(fw, row)) continue;
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

// Many query columns (a multiple of 8), no fan-out: a thread owns one group and EIGHT columns (round 5).  k_group_aggregate gives
// every (group, column) its own thread — 42M threads for 1.3M files x 32 queries, each a chain of three dependent loads (group
// offsets, row index, distance): 0.65 ms for a 512 MB matrix, all of it latency.  Here the row indices, weights and flags of a
// group are read once per 8 columns and a row's 8 distances are two 16-byte loads, up to four rows in flight; lanes of a wave are
// 64 adjacent groups (their rows lie side by side: the lines are shared), the 4 waves of a workgroup are 4 column blocks, and
// the column-major output is written 512 bytes per wave and column.  The arithmetic per (group, column) is group_value()'s:
// SQLite's KBN sums in row order.
template <int AGG_KIND>  // 0: AVG / weighted (sums), 1: MIN / MAX
__global__ __launch_bounds__(256) void k_group_aggregate8(const float *dist, uint32_t ld, uint32_t n_cols, const uint32_t *grp_off, const uint32_t *grp_rows,
                                                          uint32_t n_groups, const float *weights, const uint8_t *exclude, int agg, double *out,
                                                          uint32_t skip_when) {
    const uint32_t n_cb = n_cols / 8, cb_per_wg = 4;
    // four neighbouring lanes take the four column blocks of ONE group: a row's 128 bytes are one request of the four of them (a wave
    // per column block — 64 lanes reading 32 bytes each of 64 x 3 rows — was bound by the address unit: 16-byte lane requests)
    const uint32_t g = blockIdx.x * 64 + (threadIdx.x >> 2);
    const uint32_t cb = blockIdx.y * cb_per_wg + (threadIdx.x & 3u);
    if (g >= n_groups || cb >= n_cb) return;
    Kbn sum[8], wsum;
    double ext[8];
    uint32_t cnt[8];
#pragma unroll
    for (int c = 0; c < 8; c++) {
        ext[c] = agg == PVS_AGG_MIN ? __builtin_inf() : -__builtin_inf();
        cnt[c] = 0;
    }
    uint64_t joined = 0;
    const uint32_t e_begin = grp_off[g], e_end = grp_off[g + 1];
    for (uint32_t e = e_begin; e < e_end; e += 4) {
        const uint32_t m = e_end - e < 4 ? e_end - e : 4;
        uint32_t rw[4];
        float4 d4[4][2];
        float w4[4];
        uint8_t ex4[4];
#pragma unroll
        for (int i = 0; i < 4; i++) rw[i] = (uint32_t)i < m ? (grp_rows ? grp_rows[e + i] : e + (uint32_t)i) : 0u;  // (grp_rows == nullptr: groups are runs of rows, CSR order = row order)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const bool on = (uint32_t)i < m;
            const float4 *p = (const float4 *)(dist + (size_t)rw[i] * ld + cb * 8);
            d4[i][0] = on ? p[0] : float4{0.f, 0.f, 0.f, 0.f};
            d4[i][1] = on ? p[1] : float4{0.f, 0.f, 0.f, 0.f};
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
            const float df[8] = {d4[i][0].x, d4[i][0].y, d4[i][0].z, d4[i][0].w, d4[i][1].x, d4[i][1].y, d4[i][1].z, d4[i][1].w};
#pragma unroll
            for (int c = 0; c < 8; c++) {
                if (df[c] != df[c]) continue;  // SQL NULL distance
                const double d = (double)df[c];
                if constexpr (AGG_KIND == 0)
                    sum[c].step(weights ? d * w : d);
                else
                    ext[c] = agg == PVS_AGG_MIN ? fmin(ext[c], d) : fmax(ext[c], d);
                cnt[c]++;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 8; c++) {
        double v;
        if (joined == 0)
            v = __builtin_bit_cast(double, PVS_GROUP_ABSENT);
        else if (cnt[c] == 0)
            v = __builtin_nan("");
        else if (weights)
            v = sum[c].value() / wsum.value();
        else if (AGG_KIND == 1)
            v = ext[c];
        else
            v = sum[c].value() / (double)cnt[c];
        out[(size_t)(cb * 8 + c) * n_groups + g] = v;
    }
}

hipError_t pvs_launch_group_aggregate(const float *dist, uint32_t ld, uint32_t n_cols, uint32_t fanout, const uint32_t *grp_off,
                                      const uint32_t *grp_rows, uint32_t n_groups, const float *weights, const uint8_t *exclude,
                                      int agg, double *out, hipStream_t s, FanoutWeights fw, uint32_t skip_when, bool rows_are_runs) {
    if (n_groups == 0) return hipSuccess;
    if (!fanout && !fw.on && !fw.n_ranges && n_cols % 8 == 0 && ld % 4 == 0 && ((uintptr_t)dist & 15) == 0 && !pvs_dbg(PVS_DBG_NO_AGG8)) {
        const dim3 grid((n_groups + 63) / 64, (n_cols / 8 + 3) / 4);
        if (weights || (agg != PVS_AGG_MIN && agg != PVS_AGG_MAX))
            hipLaunchKernelGGL(k_group_aggregate8<0>, grid, dim3(256), 0, s, dist, ld, n_cols, grp_off, rows_are_runs ? nullptr : grp_rows, n_groups, weights, exclude, agg, out, skip_when);
        else
            hipLaunchKernelGGL(k_group_aggregate8<1>, grid, dim3(256), 0, s, dist, ld, n_cols, grp_off, rows_are_runs ? nullptr : grp_rows, n_groups, weights, exclude, agg, out, skip_when);
        return hipGetLastError();
    }
    const uint32_t ncol_out = fanout ? 1u : n_cols;
    const uint32_t TG = agg_tile_groups(ncol_out);
    hipLaunchKernelGGL(k_group_aggregate, dim3((n_groups + TG - 1) / TG), dim3(256), (size_t)ncol_out * TG * 8, s, dist, ld, n_cols, fanout,
                       grp_off, grp_rows, n_groups, weights, exclude, agg, fw, out, skip_when, TG);
    return hipGetLastError();
}

// one lane per (listed group, query): the lanes of a group are adjacent, so the group-major output row is written contiguously
__global__ __launch_bounds__(256) void k_group_aggregate_list(const float *dist, uint32_t ld, uint32_t n_cols, const uint32_t *grp_off, const uint32_t *grp_rows,
                                                              const uint32_t *list, uint32_t n_list, const float *weights, const uint8_t *exclude, int agg,
                                                              double *out_t, uint32_t ld_out, uint32_t skip_when) {
    const uint64_t idx = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (uint64_t)n_list * n_cols) return;
    const uint32_t li = (uint32_t)(idx / n_cols), q = (uint32_t)(idx % n_cols);
    const uint32_t g = list[li];
    out_t[(size_t)g * ld_out + q] = group_value(dist, ld, n_cols, 0, grp_off, grp_rows, weights, exclude, agg, FanoutWeights(), skip_when, g, q);
}
hipError_t pvs_launch_group_aggregate_list(const float *dist, uint32_t ld, uint32_t n_cols, const uint32_t *grp_off, const uint32_t *grp_rows,
                                           const uint32_t *list, uint32_t n_list, const float *weights, const uint8_t *exclude, int agg, double *out_t,
                                           uint32_t ld_out, hipStream_t s, uint32_t skip_when) {
    if (n_list == 0 || n_cols == 0) return hipSuccess;
    const uint64_t total = (uint64_t)n_list * n_cols;
    hipLaunchKernelGGL(k_group_aggregate_list, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, dist, ld, n_cols, grp_off, grp_rows, list, n_list, weights,
                       exclude, agg, out_t, ld_out, skip_when);
    return hipGetLastError();
}
// [n_groups][ncol] -> [ncol][n_groups] through a 32 x 32 LDS tile (both sides coalesced)
__global__ __launch_bounds__(256) void k_group_transpose(const double *in, uint32_t n_groups, uint32_t ncol, double *out) {
    __shared__ double t[32][33];
    const uint32_t g0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const uint32_t tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (uint32_t r = ty; r < 32; r += 8)
        if (g0 + r < n_groups && c0 + tx < ncol) t[r][tx] = in[(size_t)(g0 + r) * ncol + c0 + tx];
    __syncthreads();
    for (uint32_t r = ty; r < 32; r += 8)
        if (c0 + r < ncol && g0 + tx < n_groups) out[(size_t)(c0 + r) * n_groups + g0 + tx] = t[tx][r];
}
hipError_t pvs_launch_group_transpose(const double *vals_t, uint32_t n_groups, uint32_t ncol, double *vals, hipStream_t s) {
    if (n_groups == 0 || ncol == 0) return hipSuccess;
    hipLaunchKernelGGL(k_group_transpose, dim3((n_groups + 31) / 32, (ncol + 31) / 32), dim3(256), 0, s, vals_t, n_groups, ncol, vals);
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
// the same from group-major values [n_groups][ncol] into column-major keys [ncol][n_groups] (32 x 32 tiles through LDS)
__global__ __launch_bounds__(256) void k_group_page_keys_t(const double *in, uint32_t n_groups, uint32_t ncol, unsigned long long *keys) {
    __shared__ unsigned long long t[32][33];
    const uint32_t g0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const uint32_t tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (uint32_t r = ty; r < 32; r += 8)
        if (g0 + r < n_groups && c0 + tx < ncol) {
            const double v = in[(size_t)(g0 + r) * ncol + c0 + tx];
            unsigned long long k = f64_sort_key(v);
            if (k == ~0ull) k = ~0ull - 1;
            if (__builtin_bit_cast(unsigned long long, v) == PVS_GROUP_ABSENT) k = ~0ull;
            t[r][tx] = k;
        }
    __syncthreads();
    for (uint32_t r = ty; r < 32; r += 8)
        if (c0 + r < ncol && g0 + tx < n_groups) keys[(size_t)(c0 + r) * n_groups + g0 + tx] = t[tx][r];
}
hipError_t pvs_group_page_keys_t(const double *d_vals_t, uint32_t n_groups, uint32_t ncol, unsigned long long *d_keys, hipStream_t s) {
    if (n_groups == 0 || ncol == 0) return hipSuccess;
    hipLaunchKernelGGL(k_group_page_keys_t, dim3((n_groups + 31) / 32, (ncol + 31) / 32), dim3(256), 0, s, d_vals_t, n_groups, ncol, d_keys);
    return hipGetLastError();
}
hipError_t pvs_group_page_keys(const double *d_vals, uint32_t n_groups, unsigned long long *d_keys, hipStream_t s) {
    if (n_groups == 0) return hipSuccess;
    hipLaunchKernelGGL(k_group_page_keys, dim3((n_groups + 255) / 256 > 4096 ? 4096 : (n_groups + 255) / 256), dim3(256), 0, s, d_vals, n_groups, d_keys);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// Page-first ranking of GROUP-MAJOR values [n_groups][ncol] (the fused per-item scorer's layout), entirely on the device: no
// transposition, no host round trip between its steps.
//   k_gm_thresholds  per column: 4,096 sampled values -> keys, sorted in LDS, the j-th smallest is the column's threshold
//   k_gm_compact     one pass over the values (a wave reads whole 256-byte group rows): every (key, group slot) at or below its
//                    column's threshold, staged per column in LDS, appended with one global atomic per workgroup and column
//   k_gm_topk        per column: the few thousand admitted entries sorted by (key, tie order) in LDS, the first k written out
// A column whose page came out short (fewer than k admitted: an unlucky sample, NULL-heavy columns) or too long (> GM_CAP:
// massive ties at the threshold) is flagged and ranked by the full sort instead.
constexpr uint32_t GM_M = 8192, GM_CAP = 8192, GM_LIST = 48, GM_SORT_THREADS = 1024;
// first page: ~2 x target groups per column (the threshold is the j-th of GM_M samples: sigma ~ 1/sqrt(j) of that)
static inline uint64_t gm_target(uint32_t k) { return std::max<uint64_t>(8ull * k, 1024); }
__device__ static inline unsigned long long gm_key(double v) {
    unsigned long long k = f64_sort_key(v);
    if (k == ~0ull) k = ~0ull - 1;                                               // NULL aggregates: after every value ...
    if (__builtin_bit_cast(unsigned long long, v) == PVS_GROUP_ABSENT) k = ~0ull;  // ... absent groups: never emitted
    return k;
}
// (value of group g in column col: vals[g * gs + col * cs] — group-major gs = ncol, cs = 1; column-major gs = 1, cs = n_groups)
// m: samples taken (2,048 .. GM_M, a power of two: the fewest that put the threshold's rank j at 32 or more — a quarter of the
// gathers and of the select for 230k files and a page of 10, 20 -> 10 us)
__global__ __launch_bounds__(GM_SORT_THREADS) void k_gm_thresholds(const double *vals_t, uint32_t n_groups, size_t gs, size_t cs, uint32_t j, uint32_t m,
                                                                   unsigned long long *thr, uint32_t *count) {
    __shared__ unsigned long long s[GM_M];
    __shared__ uint32_t hist[256], misc[4];
    const uint32_t col = blockIdx.x, tid = threadIdx.x;
    for (uint32_t i = tid; i < m; i += GM_SORT_THREADS) s[i] = gm_key(vals_t[(size_t)((uint64_t)i * n_groups / m) * gs + col * cs]);
    __syncthreads();
    const unsigned long long t = wg_radix_kth_u64(s, m, j + 1, hist, misc, max(1u, j / 8));  // the sample's j-th smallest (0-based), give or take an eighth
    if (tid == 0) {
        thr[col] = t >= ~0ull - 1 ? 0ull : t;  // a threshold among the NULL / absent groups: a page of nothing -> full ranking
        count[(size_t)col * 32] = 0;            // (the column's page counter for the compaction behind this launch: one fill less per ranking)
    }
}
// grid (row blocks, ceil(ncol / 32)); a workgroup walks `per_wg` consecutive groups, lane = column of its 32-column chunk
__global__ __launch_bounds__(256) void k_gm_compact(const double *vals_t, uint32_t n_groups, uint32_t ncol, const unsigned long long *thr, uint32_t per_wg,
                                                    uint32_t *count, unsigned long long *out_key, uint32_t *out_slot) {
    __shared__ unsigned long long s_key[32][GM_LIST];
    __shared__ uint32_t s_slot[32][GM_LIST];
    __shared__ uint32_t s_n[32], s_base[32];
    const uint32_t tid = threadIdx.x, cl = tid & 31u, col = blockIdx.y * 32 + cl;
    if (tid < 32) s_n[tid] = 0;
    __syncthreads();
    const unsigned long long t = col < ncol ? thr[col] : 0ull;
    const uint32_t g0 = blockIdx.x * per_wg, g1 = min(n_groups, g0 + per_wg);
    if (col < ncol)
        for (uint32_t gb = g0 + (tid >> 5); gb < g1; gb += 32) {
            double v4[4];  // four independent loads in flight per lane (one per iteration left the pass latency-bound: 1.9 TB/s)
#pragma unroll
            for (int u = 0; u < 4; u++) v4[u] = gb + 8 * u < g1 ? vals_t[(size_t)(gb + 8 * u) * ncol + col] : __builtin_bit_cast(double, PVS_GROUP_ABSENT);
#pragma unroll
            for (int u = 0; u < 4; u++) {
            const uint32_t g = gb + 8 * u;
            const unsigned long long k = gm_key(v4[u]);
            if (k <= t && g < g1) {
                const uint32_t p = atomicAdd(&s_n[cl], 1u);
                if (p < GM_LIST) {
                    s_key[cl][p] = k;
                    s_slot[cl][p] = g;
                } else {  // (a dense patch of hits: straight to the global list)
                    const uint32_t gp = atomicAdd(&count[(size_t)col * 32], 1u);
                    if (gp < GM_CAP) {
                        out_key[(size_t)col * GM_CAP + gp] = k;
                        out_slot[(size_t)col * GM_CAP + gp] = g;
                    }
                }
            }
            }
        }
    __syncthreads();
    if (tid < 32 && blockIdx.y * 32 + tid < ncol) {
        const uint32_t m = s_n[tid] < GM_LIST ? s_n[tid] : GM_LIST;
        s_base[tid] = m ? atomicAdd(&count[(size_t)(blockIdx.y * 32 + tid) * 32], m) : 0u;
    }
    __syncthreads();
    for (uint32_t i = tid; i < 32 * GM_LIST; i += 256) {
        const uint32_t c = i / GM_LIST, e = i % GM_LIST, cc = blockIdx.y * 32 + c;
        const uint32_t m = s_n[c] < GM_LIST ? s_n[c] : GM_LIST;
        if (cc < ncol && e < m && s_base[c] + e < GM_CAP) {
            out_key[(size_t)cc * GM_CAP + s_base[c] + e] = s_key[c][e];
            out_slot[(size_t)cc * GM_CAP + s_base[c] + e] = s_slot[c][e];
        }
    }
}
// ONE column (similar_to's fan-out aggregate; every column of the column-major route, ranked one by one): k_gm_compact gives a lane
// to each of 32 columns, so a single column walked its groups with 8 threads per workgroup (33 us for 86k groups).  Here a thread
// takes a group, a wave appends its hits with one atomic.
__global__ __launch_bounds__(256) void k_gm_compact1(const double *vals_cm, uint32_t n_groups, const unsigned long long *thr_all, uint32_t *count_all,
                                                     unsigned long long *out_key_all, uint32_t *out_slot_all) {
    // grid.y = column of COLUMN-MAJOR values [ncol][n_groups]
    const double *vals = vals_cm + (size_t)blockIdx.y * n_groups;
    const unsigned long long *thr = thr_all + blockIdx.y;
    uint32_t *count = count_all + (size_t)blockIdx.y * 32;
    unsigned long long *out_key = out_key_all + (size_t)blockIdx.y * GM_CAP;
    uint32_t *out_slot = out_slot_all + (size_t)blockIdx.y * GM_CAP;
    // a workgroup takes 1,024 consecutive groups (four coalesced loads per thread in flight); its hits — a percent or two of them —
    // are staged in LDS and appended with ONE global atomic (an atomic per wave on the one counter serialised: 13 us for 86k groups)
    __shared__ unsigned long long s_key[1024];
    __shared__ uint32_t s_slot[1024], s_n, s_base;
    const uint32_t tid = threadIdx.x;
    const unsigned long long t = thr[0];
    if (tid == 0) s_n = 0;
    __syncthreads();
    const uint32_t g0 = blockIdx.x * 1024;
    double v4[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const uint32_t g = g0 + u * 256 + tid;
        v4[u] = g < n_groups ? vals[g] : __builtin_bit_cast(double, PVS_GROUP_ABSENT);
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const unsigned long long k = gm_key(v4[u]);
        if (k <= t && k != ~0ull) {
            const uint32_t p = atomicAdd(&s_n, 1u);
            s_key[p] = k;
            s_slot[p] = g0 + u * 256 + tid;
        }
    }
    __syncthreads();
    const uint32_t m = s_n;
    if (m == 0) return;
    if (tid == 0) s_base = atomicAdd(count, m);
    __syncthreads();
    const uint32_t base = s_base;
    for (uint32_t i = tid; i < m; i += 256)
        if (base + i < GM_CAP) {
            out_key[base + i] = s_key[i];
            out_slot[base + i] = s_slot[i];
        }
}
// one workgroup per column; grp_trank / grp_tinv: the groups' tie order (second sort key) or nullptr (group id order = slot order).
// The k-th smallest key of the column's few thousand admitted entries by radix select; the entries below it and its ties — usually
// k of them and a handful — are sorted by (key, tie order) in LDS; the first k are the page.
constexpr uint32_t GM_SORT = 2048;
__global__ __launch_bounds__(GM_SORT_THREADS) void k_gm_topk(const uint32_t *count, const unsigned long long *in_key, const uint32_t *in_slot, const int64_t *gids,
                                                 const uint32_t *grp_trank, const uint32_t *grp_tinv, uint32_t k, int64_t *out_groups, double *out_values,
                                                 uint32_t *out_flag, uint32_t *out_cnt, int allow_short) {
    extern __shared__ unsigned long long s_k[];  // [GM_CAP] keys | [GM_SORT] selected keys | [GM_CAP] u32 ties | [GM_SORT] u32 selected ties
    __shared__ uint32_t hist[256], misc[4], s_n;
    unsigned long long *s_sk = s_k + GM_CAP;
    uint32_t *s_t = (uint32_t *)(s_sk + GM_SORT), *s_st = s_t + GM_CAP;
    const uint32_t col = blockIdx.x, tid = threadIdx.x;
    const uint32_t m = count[(size_t)col * 32];
    // (allow_short: the column holds EVERY group of a short candidate list — fewer than k is the whole answer, not an unlucky page)
    if ((m < k && !allow_short) || m > GM_CAP || (allow_short && m > GM_SORT && k >= m)) {
        if (tid == 0) out_flag[col] = 0;
        return;
    }
    for (uint32_t i = tid; i < m; i += GM_SORT_THREADS) {
        const uint32_t slot = in_slot[(size_t)col * GM_CAP + i];
        s_k[i] = in_key[(size_t)col * GM_CAP + i];
        s_t[i] = grp_trank ? grp_trank[slot] : slot;
    }
    if (tid == 0) s_n = 0;
    __syncthreads();
    const unsigned long long kth = k < m ? wg_radix_kth_u64(s_k, m, k, hist, misc, 64) : ~0ull;  // (an upper bound of the k-th: the sort below is what is exact)
    for (uint32_t i = tid; i < m; i += GM_SORT_THREADS)
        if (s_k[i] <= kth) {
            const uint32_t p = atomicAdd(&s_n, 1u);
            if (p < GM_SORT) {
                s_sk[p] = s_k[i];
                s_st[p] = s_t[i];
            }
        }
    __syncthreads();
    const uint32_t ms = s_n;
    if (ms > GM_SORT) {  // massive ties at the k-th value: the full sort orders them
        if (tid == 0) out_flag[col] = 0;
        return;
    }
    uint32_t m2 = 1;
    while (m2 < ms) m2 <<= 1;
    for (uint32_t i = ms + tid; i < m2; i += GM_SORT_THREADS) {
        s_sk[i] = ~0ull;
        s_st[i] = 0xffffffffu;
    }
    __syncthreads();
    for (uint32_t sz = 2; sz <= m2; sz <<= 1)
        for (uint32_t st = sz >> 1; st > 0; st >>= 1) {
            for (uint32_t i = tid; i < m2 / 2; i += GM_SORT_THREADS) {
                const uint32_t lo = 2 * i - (i & (st - 1)), hi = lo + st;
                const bool up = (lo & sz) == 0;
                const unsigned long long x = s_sk[lo], y = s_sk[hi];
                const uint32_t tx = s_st[lo], ty = s_st[hi];
                const bool gt = x > y || (x == y && tx > ty);
                if (gt == up) {
                    s_sk[lo] = y;
                    s_sk[hi] = x;
                    s_st[lo] = ty;
                    s_st[hi] = tx;
                }
            }
            __syncthreads();
        }
    const uint32_t nout = ms < k ? ms : k;
    for (uint32_t i = tid; i < k; i += GM_SORT_THREADS) {
        if (i >= nout) {
            out_groups[(size_t)col * k + i] = -1;
            out_values[(size_t)col * k + i] = __builtin_nan("");
            continue;
        }
        const uint32_t slot = grp_tinv ? grp_tinv[s_st[i]] : s_st[i];
        out_groups[(size_t)col * k + i] = gids[slot];
        const unsigned long long key = s_sk[i];
        double v;
        if (key >= ~0ull - 1) {
            v = __builtin_nan("");
        } else {
            const unsigned long long b = (key >> 63) ? (key & 0x7fffffffffffffffull) : ~key;
            v = __builtin_bit_cast(double, b);
        }
        out_values[(size_t)col * k + i] = v;
    }
    if (tid == 0) {
        out_flag[col] = 1;
        if (out_cnt) out_cnt[col] = nout;
    }
}
// column-major values [ncol][n_sub] of the sub-groups of a candidate list -> k_gm_topk's per-column entry lists (every sub-group)
__global__ void k_sub_entries(const double *vals, uint32_t n_sub, uint32_t ncol, const uint32_t *sub_slot, uint32_t *count, unsigned long long *out_key,
                              uint32_t *out_slot) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, col = blockIdx.y;
    if (i == 0) count[(size_t)col * 32] = n_sub;
    if (i >= n_sub) return;
    out_key[(size_t)col * GM_CAP + i] = gm_key(vals[(size_t)col * n_sub + i]);
    out_slot[(size_t)col * GM_CAP + i] = sub_slot[i];
}
// d_work: >= pvs_gm_rank_work_bytes(ncol) of device scratch; out_*: device-accessible (pinned) [ncol][k] / [ncol]; stream-ordered
size_t pvs_gm_rank_work_bytes(uint32_t ncol) { return (size_t)ncol * (8 + 128 + (size_t)GM_CAP * 12) + 256; }
bool pvs_gm_rank_supported(uint32_t n_groups, uint32_t ncol, uint32_t k) {
    const uint64_t target = gm_target(k);
    // (one query over 230k files: the device page costs five short kernels and one synchronisation where the full radix sort of the
    //  column is nineteen launches, 0.11 ms: no lower bound on the number of values any more)
    (void)ncol;
    return n_groups >= 65536 && 2 * target <= GM_CAP / 2 && target * 8 < n_groups;
}
hipError_t pvs_gm_rank(const double *d_vals_t, uint32_t n_groups, uint32_t ncol, uint32_t k, const int64_t *d_gids, const uint32_t *d_grp_trank,
                       const uint32_t *d_grp_tinv, void *d_work, int64_t *out_groups, double *out_values, uint32_t *out_flag, hipStream_t s, bool column_major) {
    uint8_t *w = (uint8_t *)d_work;
    unsigned long long *thr = (unsigned long long *)w;
    uint32_t *count = (uint32_t *)(w + (((size_t)ncol * 8 + 127) & ~(size_t)127));
    unsigned long long *keys = (unsigned long long *)((uint8_t *)count + (size_t)ncol * 128);
    uint32_t *slots = (uint32_t *)((uint8_t *)keys + (size_t)ncol * GM_CAP * 8);
    const uint64_t target = gm_target(k);
    uint32_t m = 2048;
    while (m < GM_M && (double)m * 2.0 * (double)target / (double)n_groups < 32.0) m <<= 1;
    const uint32_t j = (uint32_t)std::min<uint64_t>(m - 1, (uint64_t)((double)m * 2.0 * (double)target / (double)n_groups) + 4);
    hipError_t e = hipSuccess;
    if (ncol == 1) column_major = true;  // (the same thing)
    hipLaunchKernelGGL(k_gm_thresholds, dim3(ncol), dim3(GM_SORT_THREADS), 0, s, d_vals_t, n_groups, column_major ? (size_t)1 : (size_t)ncol,
                       column_major ? (size_t)n_groups : (size_t)1, j, m, thr, count);
    const uint32_t per_wg = 1024;
    if (column_major)
        hipLaunchKernelGGL(k_gm_compact1, dim3((n_groups + 1023) / 1024, ncol), dim3(256), 0, s, d_vals_t, n_groups, thr, count, keys, slots);
    else
        hipLaunchKernelGGL(k_gm_compact, dim3((n_groups + per_wg - 1) / per_wg, (ncol + 31) / 32), dim3(256), 0, s, d_vals_t, n_groups, ncol, thr, per_wg, count, keys,
                           slots);
    static std::atomic<bool> configured{false};
    if (!configured.load(std::memory_order_acquire)) {
        e = hipFuncSetAttribute((const void *)k_gm_topk, hipFuncAttributeMaxDynamicSharedMemorySize, (GM_CAP + GM_SORT) * 12);
        if (e != hipSuccess) return e;
        configured.store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(k_gm_topk, dim3(ncol), dim3(GM_SORT_THREADS), (size_t)(GM_CAP + GM_SORT) * 12, s, count, keys, slots, d_gids, d_grp_trank, d_grp_tinv, k, out_groups, out_values,
                       out_flag, (uint32_t *)nullptr, 0);
    return hipGetLastError();
}
// The pages of the <= GM_CAP sub-groups of a candidate list (pvs_sparse.hip): d_vals [ncol][n_sub] column-major, sub_slot[j] = the
// group slot of sub-group j.  out_flag[col] = 0: the column could not be sorted in LDS (more than 2,048 groups tie at the page's edge).
bool pvs_sub_rank_supported(uint32_t n_sub) { return n_sub <= GM_CAP; }
hipError_t pvs_sub_rank(const double *d_vals, uint32_t n_sub, uint32_t ncol, uint32_t k, const uint32_t *d_sub_slot, const int64_t *d_gids, const uint32_t *d_grp_trank,
                        const uint32_t *d_grp_tinv, void *d_work, int64_t *out_groups, double *out_values, uint32_t *out_flag, uint32_t *out_cnt, hipStream_t s) {
    uint8_t *w = (uint8_t *)d_work;
    uint32_t *count = (uint32_t *)(w + (((size_t)ncol * 8 + 127) & ~(size_t)127));
    unsigned long long *keys = (unsigned long long *)((uint8_t *)count + (size_t)ncol * 128);
    uint32_t *slots = (uint32_t *)((uint8_t *)keys + (size_t)ncol * GM_CAP * 8);
    hipLaunchKernelGGL(k_sub_entries, dim3((std::max<uint32_t>(n_sub, 1) + 255) / 256, ncol), dim3(256), 0, s, d_vals, n_sub, ncol, d_sub_slot, count, keys, slots);
    static std::atomic<bool> configured{false};
    if (!configured.load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void *)k_gm_topk, hipFuncAttributeMaxDynamicSharedMemorySize, (GM_CAP + GM_SORT) * 12);
        if (e != hipSuccess) return e;
        configured.store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(k_gm_topk, dim3(ncol), dim3(GM_SORT_THREADS), (size_t)(GM_CAP + GM_SORT) * 12, s, count, keys, slots, d_gids, d_grp_trank, d_grp_tinv, k, out_groups, out_values,
                       out_flag, out_cnt, 1);
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

// ---- per-item pages of a multi-device index, merged on devices[0] (pvs_multi.hip: multi_search_groups / multi_similar_to) ----------
// The second sort key of every entry of a shard's page [batch][k]: the group's slot by binary search in the shard's id-ordered
// group list, its key from the per-group key array (pvs_index_set_order_keys; a group's key is its first row's).
__global__ __launch_bounds__(256) void k_page_group_keys(const int64_t *page_groups, const uint32_t *counts, uint32_t batch, uint32_t k, const int64_t *grp_ids,
                                                         uint32_t G, const int64_t *grp_key, int64_t *out_keys) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= batch * k) return;
    const uint32_t q = t / k, i = t - q * k;
    int64_t key = 0;
    if (i < counts[q] && G) {
        const int64_t g = page_groups[t];
        uint32_t lo = 0, hi = G;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (grp_ids[mid] < g) lo = mid + 1; else hi = mid;
        }
        if (lo < G && grp_ids[lo] == g) key = grp_key[lo];
    }
    out_keys[t] = key;
}
hipError_t pvs_launch_page_group_keys(const int64_t *page_groups, const uint32_t *counts, uint32_t batch, uint32_t k, const int64_t *grp_ids, uint32_t G,
                                      const int64_t *grp_key, int64_t *out_keys, hipStream_t s) {
    if (batch * k == 0) return hipSuccess;
    hipLaunchKernelGGL(k_page_group_keys, dim3((batch * k + 255) / 256), dim3(256), 0, s, page_groups, counts, batch, k, grp_ids, G, grp_key, out_keys);
    return hipGetLastError();
}
// S shard pages [S][batch][k] (groups, f64 values, keys or nullptr, counts [S][batch]) -> the first k of their union per query under
// the page order (value ascending, NULL last, key DESCENDING, group id ascending).  The shards hold disjoint groups (placement by
// group), so there is nothing to fold.  One workgroup per query, one LDS sort: S * k <= PVS_GROUP_MERGE_MAX entries.
constexpr uint32_t GMERGE_THREADS = 256;
__global__ __launch_bounds__(256) void k_merge_group_pages(const int64_t *g, const double *v, const int64_t *key, const uint32_t *cnt, uint32_t S, uint32_t batch,
                                                           uint32_t k, uint32_t cap, int64_t *out_g, double *out_v, uint32_t *out_c) {
    extern __shared__ __attribute__((aligned(16))) uint8_t gm_smem[];
    uint64_t *const s_v = (uint64_t *)gm_smem;   // [cap] sortable value bits
    uint64_t *const s_k = s_v + cap;              // [cap] ~key (so that ascending = key descending)
    int64_t *const s_g = (int64_t *)(s_k + cap);  // [cap]
    uint32_t *const s_at = (uint32_t *)(s_g + cap);  // [cap] where the entry came from (its value is written back as it was stored)
    const uint32_t q = blockIdx.x, tid = threadIdx.x;
    const size_t elems = (size_t)batch * k;
    uint32_t total = 0;
    for (uint32_t s = 0; s < S; s++) total += min(cnt[(size_t)s * batch + q], k);
    for (uint32_t e = tid; e < cap; e += GMERGE_THREADS) {
        // entry e of the concatenation of the shards' pages
        uint32_t s = 0, base = 0;
        bool have = false;
        for (; s < S; s++) {
            const uint32_t c = min(cnt[(size_t)s * batch + q], k);
            if (e < base + c) {
                have = true;
                break;
            }
            base += c;
        }
        if (have) {
            const size_t at = s * elems + (size_t)q * k + (e - base);
            const double val = v[at] == 0.0 ? 0.0 : v[at];  // (-0 and +0 tie, as they do under the host's comparison)
            uint64_t b = (uint64_t)__double_as_longlong(val);
            b = val != val ? ~0ull - 1 : ((b >> 63) ? ~b : (b | 0x8000000000000000ull));  // NULL: behind every value, in front of the padding
            s_v[e] = b;
            s_k[e] = key ? ~((uint64_t)key[at] ^ 0x8000000000000000ull) : 0;
            s_g[e] = g[at];
            s_at[e] = (uint32_t)at;
        } else {
            s_v[e] = ~0ull;
            s_k[e] = ~0ull;
            s_g[e] = 0x7fffffffffffffffll;
            s_at[e] = 0;
        }
    }
    __syncthreads();
    for (uint32_t sz = 2; sz <= cap; sz <<= 1)
        for (uint32_t st = sz >> 1; st > 0; st >>= 1) {
            for (uint32_t i = tid; i < cap; i += GMERGE_THREADS) {
                const uint32_t j = i ^ st;
                if (j > i) {
                    const bool up = (i & sz) == 0;
                    const bool gt = s_v[i] != s_v[j] ? s_v[i] > s_v[j] : s_k[i] != s_k[j] ? s_k[i] > s_k[j] : s_g[i] > s_g[j];
                    if (gt == up) {
                        const uint64_t a = s_v[i], b = s_k[i];
                        const int64_t c = s_g[i];
                        const uint32_t d = s_at[i];
                        s_v[i] = s_v[j]; s_k[i] = s_k[j]; s_g[i] = s_g[j]; s_at[i] = s_at[j];
                        s_v[j] = a; s_k[j] = b; s_g[j] = c; s_at[j] = d;
                    }
                }
            }
            __syncthreads();
        }
    const uint32_t nout = min(total, k);
    for (uint32_t i = tid; i < k; i += GMERGE_THREADS) {
        if (i < nout) {
            out_g[(size_t)q * k + i] = s_g[i];
            out_v[(size_t)q * k + i] = v[s_at[i]];
        } else {
            out_g[(size_t)q * k + i] = -1;
            out_v[(size_t)q * k + i] = __builtin_nan("");
        }
    }
    if (tid == 0) out_c[q] = nout;
}
// Larger unions (4,096 < S * k <= 32,768: pages of thousands of files from 8 shards): every shard's page is already in page order,
// so an entry's place in the union is its place in its own page plus, for every other shard, the number of that shard's entries in
// front of it — one binary search per other shard (round 5; until then these pages went to the host).  The pages are staged from the
// pinned block into device memory first (sortable value bits, inverted key, group, raw value bits): the searches then run in L2.
struct GPEntry {
    unsigned long long vb, kb;
    long long g;
    unsigned long long raw;
};
__global__ __launch_bounds__(256) void k_stage_group_pages(const int64_t *g, const double *v, const int64_t *key, const uint32_t *cnt, uint32_t S, uint32_t batch,
                                                           uint32_t k, GPEntry *st, uint32_t *cnt_dev) {
    const uint32_t q = blockIdx.y, e = blockIdx.x * 256 + threadIdx.x;
    if (e < S) cnt_dev[(size_t)e * batch + q] = cnt[(size_t)e * batch + q];  // (the merge reads the counts many times: not from pinned host memory)
    if (e >= S * k) return;
    const uint32_t s = e / k, i = e - s * k;
    GPEntry o{~0ull, ~0ull, 0x7fffffffffffffffll, 0ull};
    if (i < min(cnt[(size_t)s * batch + q], k)) {
        const size_t at = (size_t)s * batch * k + (size_t)q * k + i;
        const double raw = v[at], val = raw == 0.0 ? 0.0 : raw;  // (-0 and +0 tie)
        unsigned long long b = (unsigned long long)__double_as_longlong(val);
        b = val != val ? ~0ull - 1 : ((b >> 63) ? ~b : (b | 0x8000000000000000ull));
        o.vb = b;
        o.kb = key ? ~((unsigned long long)key[at] ^ 0x8000000000000000ull) : 0ull;
        o.g = g[at];
        o.raw = (unsigned long long)__double_as_longlong(raw);
    }
    st[((size_t)q * S + s) * k + i] = o;
}
__device__ static inline bool gp_less(const GPEntry &a, const GPEntry &b) { return a.vb != b.vb ? a.vb < b.vb : a.kb != b.kb ? a.kb < b.kb : a.g < b.g; }
__global__ __launch_bounds__(256) void k_rankmerge_group_pages(const GPEntry *st, const uint32_t *cnt, uint32_t S, uint32_t batch, uint32_t k, int64_t *out_g,
                                                               double *out_v, uint32_t *out_c) {
    const uint32_t q = blockIdx.y, e = blockIdx.x * 256 + threadIdx.x;
    uint32_t total = 0;
    for (uint32_t s = 0; s < S; s++) total += min(cnt[(size_t)s * batch + q], k);
    const uint32_t nout = min(total, k);
    if (e == 0) out_c[q] = nout;
    if (e < k && e >= nout) {  // padding behind a short union
        out_g[(size_t)q * k + e] = -1;
        out_v[(size_t)q * k + e] = __builtin_nan("");
    }
    if (e >= S * k) return;
    const uint32_t s = e / k, i = e - s * k;
    if (i >= min(cnt[(size_t)s * batch + q], k)) return;
    const GPEntry *base = st + (size_t)q * S * k;
    const GPEntry me = base[(size_t)s * k + i];
    uint32_t rank = i;
    for (uint32_t o = 0; o < S && rank < k; o++) {
        if (o == s) continue;
        const GPEntry *pg = base + (size_t)o * k;
        uint32_t lo = 0, hi = min(cnt[(size_t)o * batch + q], k);
        while (lo < hi) {  // entries of shard o in front of me (groups are disjoint across shards: no ties)
            const uint32_t mid = (lo + hi) >> 1;
            if (gp_less(pg[mid], me)) lo = mid + 1; else hi = mid;
        }
        rank += lo;
    }
    if (rank < k) {
        out_g[(size_t)q * k + rank] = me.g;
        out_v[(size_t)q * k + rank] = __longlong_as_double((long long)me.raw);
    }
}
constexpr uint32_t PVS_GROUP_MERGE_LDS = 4096, PVS_GROUP_MERGE_MAX = 32768;
bool pvs_merge_group_pages_supported(uint32_t S, uint32_t k) { return (uint64_t)S * k <= PVS_GROUP_MERGE_MAX; }
hipError_t pvs_launch_merge_group_pages(const int64_t *g, const double *v, const int64_t *key, const uint32_t *cnt, uint32_t S, uint32_t batch, uint32_t k,
                                        int64_t *out_g, double *out_v, uint32_t *out_c, hipStream_t s) {
    if (batch == 0) return hipSuccess;
    if ((uint64_t)S * k > PVS_GROUP_MERGE_LDS) {
        GPEntry *st = nullptr;
        const size_t st_bytes = (size_t)batch * S * k * sizeof(GPEntry);
        hipError_t e = pvs_scratch_alloc((void **)&st, st_bytes + (size_t)S * batch * 4);
        if (e != hipSuccess) return e;
        uint32_t *cnt_dev = (uint32_t *)((uint8_t *)st + st_bytes);
        const dim3 grid((S * k + 255) / 256, batch);
        hipLaunchKernelGGL(k_stage_group_pages, grid, dim3(256), 0, s, g, v, key, cnt, S, batch, k, st, cnt_dev);
        hipLaunchKernelGGL(k_rankmerge_group_pages, grid, dim3(256), 0, s, st, cnt_dev, S, batch, k, out_g, out_v, out_c);
        e = hipGetLastError();
        pvs_scratch_free_on(st, s);
        return e;
    }
    uint32_t cap = 64;
    while (cap < S * k) cap <<= 1;
    const size_t lds = (size_t)cap * 28;
    static std::atomic<bool> configured{false};
    if (!configured.load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void *)k_merge_group_pages, hipFuncAttributeMaxDynamicSharedMemorySize, 4096 * 28);
        if (e != hipSuccess) return e;
        configured.store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(k_merge_group_pages, dim3(batch), dim3(GMERGE_THREADS), lds, s, g, v, key, cnt, S, batch, k, cap, out_g, out_v, out_c);
    return hipGetLastError();
}
