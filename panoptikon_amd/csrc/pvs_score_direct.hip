// pvs_score_direct.hip — exact int8 distances of EVERY row for 1..4 queries (the `d` column of dist_{cte}, filters/exact.rs:106-165,
// for one query; the per-branch scoring of pvs_rrf_search, configs[4]) without the matrix cores.  With a handful of queries the
// pass is a pure HBM stream: the MFMA scan (k_scan MODE 2) pads them to 32, pays LDS staging and a workgroup barrier per tile,
// and streamed 5.9-6.0 TB/s; here every wave reads its own layout tiles straight into registers, multiplies with v_dot4_i32_i8
// against the query chunk it needs (LDS, 16 distinct chunks per wave-read: conflict free) and never synchronises with another wave.
//
// Layout (pvs_common.hpp): a tile = 32 rows, k-slab major; one slab of a tile is 8 KiB = 32 rows x 256 B and a row's sixteen
// 16-byte chunks are stored XOR-swizzled by (row & 15).  A wave-load of 1 KiB = rows 4j .. 4j+3 of the slab: lane l holds stored
// chunk (l & 15) of row 4j + (l >> 4), i.e. logical chunk (l & 15) ^ (row & 15).  Eight loads cover the slab; lane l keeps one
// partial sum per j (and per query); after the last slab the sixteen lanes of a row are added up with four DPP steps.
//
// Result: the closed form of the exact integer sums (orc_i8_cosine_from_sums / orc_i8_l2_from_sums), identical to MODE 2 and to
// the oracle while every sum stays below 2^24 (the caller checks dim * 127^2; an L2 sum beyond it raises *flag and the caller
// falls back to the in-order scorer).
#include "pvs_kernels.hpp"

namespace {
typedef int v4i __attribute__((ext_vector_type(4)));

__device__ static inline int row16_sum(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);  // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false);  // row_mirror
    return v;
}

// The per-item fold (FOLD, round 5): files that are runs of rows are aggregated in the tile's epilogue — the same contract as k_scan
// MODE 3 (ScanK.tile_grp: per 32-row tile the group of its first row, the rows that end a group, the rows of groups that cross a
// tile boundary; those go to the matrix `out` and k_group_aggregate_list folds them) for the ONE to four queries this kernel
// serves: there the fold walks a tile's 32 rows serially in every query lane (built for 32 queries; for one query 31 lanes idle
// behind a chain of ~400 dependent f64 operations per tile: 125 us for 690k x 768 rows), here a wave is one tile, lane r takes row
// r, and the first row of every group walks ITS group's rows in row order (SQLite's SUM / AVG are compensated sums, order matters)
// — as many steps as the longest group of the tile has rows.
struct FoldArgs {
    const uint4 *tile_grp;
    const float *weights;  // per row or nullptr
    const uint8_t *mask;   // candidate mask (a byte per row) or nullptr
    double *out;           // [n_groups][ld] group-major
    uint32_t ld;
    int agg;
};
struct KbnD {
    double s = 0.0, c = 0.0;
    __device__ inline void step_if(bool on, double r) {  // the state of a skipped row stays untouched bit for bit
        const double t = s + r;
        const double x = (s - t) + r, y = (r - t) + s;
        const double c2 = c + (fabs(s) > fabs(r) ? x : y);
        c = on ? c2 : c;
        s = on ? t : s;
    }
};

template <int NQ, bool COS, bool FOLD>
__global__ __launch_bounds__(256) void k_score_i8_direct(const uint8_t *__restrict__ rows, uint32_t stride, uint64_t n_rows, uint32_t n_tiles,
                                                          const float *__restrict__ norm2, const int8_t *__restrict__ qexact, uint32_t dim,
                                                          const QInfo *__restrict__ qinfo, uint32_t nb, float *__restrict__ out, uint32_t ld,
                                                          uint32_t *flag, FoldArgs fa) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_q[];  // [NQ][stride] query codes, zero padded
    const uint32_t kslabs = stride / PVS_KSLAB_BYTES;
    for (uint32_t i = threadIdx.x; i < NQ * stride; i += 256) {
        const uint32_t q = i / stride, b = i % stride;
        s_q[i] = (q < nb && b < dim) ? (uint8_t)qexact[(size_t)q * dim + b] : (uint8_t)0;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = gridDim.x * 4;
    const uint32_t t_end = n_tiles, t_step = n_waves;  // wave w walks tiles w, w + n_waves, ... (a contiguous range per wave measured the same)
    float bb[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) bb[q] = q < (int)nb ? qinfo[q].bb : 0.f;
    const uint32_t pc = lane & 15, rsub = lane >> 4;
    // query chunk this lane needs for load j of a slab: logical chunk pc ^ ((4j + rsub) & 15) = pc ^ (4 (j & 3) + rsub)
    uint32_t qoff[4];
#pragma unroll
    for (int j = 0; j < 4; j++) qoff[j] = (pc ^ (uint32_t)(4 * j + rsub)) * 16u;
    const size_t tile_bytes = 32u * (size_t)stride;
    auto load_slab = [&](uint32_t tile, uint32_t s, v4i (&dst)[8]) __attribute__((always_inline)) {
        const uint8_t *p = rows + (size_t)tile * tile_bytes + (size_t)s * 8192u + (size_t)lane * 16u;
#pragma unroll
        for (int j = 0; j < 8; j++) dst[j] = __builtin_nontemporal_load((const v4i *)(p + j * 1024));
    };
    uint32_t tile = wave;
    if (tile >= t_end) return;
    v4i cur[8], nxt[8];
    load_slab(tile, 0, cur);
    while (tile < t_end) {
        int acc[8][NQ];
#pragma unroll
        for (int j = 0; j < 8; j++)
#pragma unroll
            for (int q = 0; q < NQ; q++) acc[j][q] = 0;
        for (uint32_t s = 0; s < kslabs; s++) {
            // request the next slab (of this tile, or the first of this wave's next tile) before consuming the current one
            const bool last = s + 1 == kslabs;
            const uint32_t nt = last ? tile + t_step : tile, ns = last ? 0u : s + 1;
            const bool more = nt < t_end;
            if (more) load_slab(nt, ns, nxt);
#pragma unroll
            for (int j = 0; j < 8; j++) {
#pragma unroll
                for (int q = 0; q < NQ; q++) {
                    const v4i qv = *(const v4i *)(s_q + (size_t)q * stride + s * 256u + qoff[j & 3]);
                    int a = acc[j][q];
                    a = __builtin_amdgcn_sdot4(cur[j].x, qv.x, a, false);
                    a = __builtin_amdgcn_sdot4(cur[j].y, qv.y, a, false);
                    a = __builtin_amdgcn_sdot4(cur[j].z, qv.z, a, false);
                    a = __builtin_amdgcn_sdot4(cur[j].w, qv.w, a, false);
                    acc[j][q] = a;
                }
            }
            if (more) {
#pragma unroll
                for (int j = 0; j < 8; j++) cur[j] = nxt[j];
            }
        }
        // row totals: lane (pc == j, j < 8) keeps row 4j + rsub
        float dq[NQ];
        {
            const uint64_t row = (uint64_t)tile * 32u + 4u * pc + rsub;
            const bool have = pc < 8 && row < n_rows;
            const float aa = have ? norm2[row] : 0.f;
#pragma unroll
            for (int q = 0; q < NQ; q++) {
                int mine = 0;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int t = row16_sum(acc[j][q]);
                    mine = (int)pc == j ? t : mine;
                }
                float d = 0.f;
                if (have && q < (int)nb) {
                    if (COS) {
                        d = ref_cosine_finish((float)mine, aa, bb[q]);
                    } else {
                        const double ss = (double)aa + (double)bb[q] - 2.0 * (double)mine;
                        if (!(ss < 16777216.0)) *flag = 1u;  // (a plain store: the word may live in pinned host memory; 1 is the only value ever written)
                        d = ref_l2_finish((float)ss);
                    }
                    if (!FOLD) out[(size_t)row * ld + q] = d;
                }
                dq[q] = d;
            }
        }
        if constexpr (FOLD) {
            // lane r (< 32) takes row r of the tile: its distances come from lane (r >> 2) + 16 (r & 3)
            const uint32_t r = (uint32_t)lane & 31u;
            const int src = (int)((r >> 2) + 16u * (r & 3u));
            float dr[NQ];
#pragma unroll
            for (int q = 0; q < NQ; q++) dr[q] = __shfl(dq[q], src, 64);
            const uint4 rec = fa.tile_grp[tile];
            const uint64_t row = (uint64_t)tile * 32u + r;
            const uint32_t n_here = n_rows - (uint64_t)tile * 32u >= 32u ? 32u : (uint32_t)(n_rows - (uint64_t)tile * 32u);
            const uint32_t m_rows = n_here == 32u ? 0xffffffffu : ((1u << n_here) - 1u);
            const uint32_t m_end = rec.y & m_rows, m_sp = rec.z & m_rows;
            const bool in_tile = lane < 32 && r < n_here;
            const bool spilled = (m_sp >> r) & 1u;
            if (in_tile && spilled) {  // rows of tile-crossing groups: through the matrix
#pragma unroll
                for (int q = 0; q < NQ; q++)
                    if (q < (int)nb) out[(size_t)row * ld + q] = dr[q];
            }
            bool use = in_tile && !spilled;
            if (fa.mask) use = use && (in_tile ? fa.mask[row] != 0 : false);
            const float wf = fa.weights && in_tile ? fa.weights[row] : 1.f;
            // a group's first row in the tile: row 0, or the row behind one that ends a group
            const bool leader = in_tile && !spilled && (r == 0 || ((m_end >> (r - 1)) & 1u));
            // rows from r to the end of its group (the end bit exists: groups without one in this tile are spilled)
            const uint32_t ends_from = m_end >> r;
            const uint32_t len = leader && ends_from ? (uint32_t)__builtin_ctz(ends_from) + 1u : 0u;
            const uint32_t g = rec.x + (uint32_t)__builtin_popcount(m_end & ((1u << r) - 1u));  // group slot of row r
            uint32_t max_len = len;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) max_len = max(max_len, (uint32_t)__shfl_xor((int)max_len, off, 64));
            KbnD sum[NQ], wsum;
            double ext[NQ];
            uint32_t cnt[NQ], joined = 0;
#pragma unroll
            for (int q = 0; q < NQ; q++) {
                ext[q] = fa.agg == PVS_AGG_MIN ? __builtin_inf() : -__builtin_inf();
                cnt[q] = 0;
            }
            for (uint32_t t = 0; t < max_len; t++) {  // (wave-uniform trip count; a leader's members are lanes r .. r + len - 1)
                const int from = (int)((r + t) & 31u);
                const bool mem = t < len;
                const bool u = __shfl((int)use, from, 64) != 0 && mem;
                const double w = (double)__shfl(wf, from, 64);
                joined += u ? 1u : 0u;
                if (fa.weights) wsum.step_if(u, w);
#pragma unroll
                for (int q = 0; q < NQ; q++) {
                    const float df = __shfl(dr[q], from, 64);
                    const bool val = u && df == df;
                    const double d = (double)df;
                    if (fa.weights || fa.agg == PVS_AGG_AVG)
                        sum[q].step_if(val, fa.weights ? d * w : d);
                    else
                        ext[q] = val ? (fa.agg == PVS_AGG_MIN ? fmin(ext[q], d) : fmax(ext[q], d)) : ext[q];
                    cnt[q] += val ? 1u : 0u;
                }
            }
            if (len) {
#pragma unroll
                for (int q = 0; q < NQ; q++)
                    if (q < (int)nb) {
                        double v;
                        if (joined == 0)
                            v = __builtin_bit_cast(double, PVS_GROUP_ABSENT);
                        else if (cnt[q] == 0)
                            v = __builtin_nan("");
                        else if (fa.weights)
                            v = (sum[q].s + sum[q].c) / (wsum.s + wsum.c);
                        else if (fa.agg == PVS_AGG_AVG)
                            v = (sum[q].s + sum[q].c) / (double)cnt[q];
                        else
                            v = ext[q];
                        fa.out[(size_t)g * fa.ld + q] = v;
                    }
            }
        }
        tile += t_step;
    }
}
}  // namespace

// out[row * ld + q] for q < nb <= 4.  qexact: [nb][dim] int8 codes (dense), qinfo[q].bb = |q|^2.
static hipError_t launch_score_i8(int metric, const uint8_t *rows, uint32_t stride, uint32_t dim, uint64_t n_rows, const float *norm2, const void *qexact,
                                  const QInfo *qinfo, uint32_t nb, float *out, uint32_t ld, uint32_t *flag, const FoldArgs *fold, uint32_t n_cu, hipStream_t s) {
    if (nb == 0 || nb > 4 || n_rows == 0) return nb == 0 || n_rows == 0 ? hipSuccess : hipErrorInvalidValue;
    const uint32_t n_tiles = (uint32_t)((n_rows + 31) / 32);
    const int nq = nb == 1 ? 1 : nb == 2 ? 2 : 4;
    // 8 waves per CU, 16 KiB in flight each.  Measured on MI355X, 25M rows, one query (rocprofv3): 512-B rows 1.94 ms = 6.6 TB/s
    // (k_scan MODE 2: 2.43 ms; 1 / 4 / 8 workgroups per CU: 2.37 / 2.10 / 2.05 ms), 1-KiB rows 4.06-4.08 ms = 6.3 TB/s whatever
    // the occupancy (MODE 2: 3.99 ms) — 25.6 GB streams no faster than that on this part.
    const uint32_t wg_per_cu = 2;
    const uint32_t grid = std::min<uint32_t>((n_tiles + 3) / 4, n_cu * wg_per_cu);
    const size_t lds = (size_t)nq * stride;
    const FoldArgs fa = fold ? *fold : FoldArgs{nullptr, nullptr, nullptr, nullptr, 0, 0};
#define PVS_SD_LAUNCH(NQ, COS, FOLD)                                                                                                             \
    hipLaunchKernelGGL((k_score_i8_direct<NQ, COS, FOLD>), dim3(grid), dim3(256), lds, s, rows, stride, n_rows, n_tiles, norm2, (const int8_t *)qexact, \
                       dim, qinfo, nb, out, ld, flag, fa)
#define PVS_SD_PICK(NQ)                                                    \
    do {                                                                   \
        if (cos && fold) PVS_SD_LAUNCH(NQ, true, true);                    \
        else if (cos) PVS_SD_LAUNCH(NQ, true, false);                      \
        else if (fold) PVS_SD_LAUNCH(NQ, false, true);                     \
        else PVS_SD_LAUNCH(NQ, false, false);                              \
    } while (0)
    const bool cos = metric == PVS_COSINE;
    if (nq == 1)
        PVS_SD_PICK(1);
    else if (nq == 2)
        PVS_SD_PICK(2);
    else
        PVS_SD_PICK(4);
#undef PVS_SD_PICK
#undef PVS_SD_LAUNCH
    return hipGetLastError();
}
hipError_t pvs_launch_score_i8_direct(int metric, const uint8_t *rows, uint32_t stride, uint32_t dim, uint64_t n_rows, const float *norm2,
                                      const void *qexact, const QInfo *qinfo, uint32_t nb, float *out, uint32_t ld, uint32_t *flag,
                                      uint32_t n_cu, hipStream_t s) {
    return launch_score_i8(metric, rows, stride, dim, n_rows, norm2, qexact, qinfo, nb, out, ld, flag, nullptr, n_cu, s);
}
// the same with the per-item fold in the tile's epilogue (k_scan MODE 3's contract for 1..4 queries): fold_out [n_groups][fold_ld]
// receives every group that lies inside one tile, `out` [n][ld] the rows of the groups that cross tiles
hipError_t pvs_launch_score_i8_fold(int metric, const uint8_t *rows, uint32_t stride, uint32_t dim, uint64_t n_rows, const float *norm2, const void *qexact,
                                    const QInfo *qinfo, uint32_t nb, float *out, uint32_t ld, uint32_t *flag, const uint4 *tile_grp, const float *weights,
                                    const uint8_t *mask, double *fold_out, uint32_t fold_ld, int agg, uint32_t n_cu, hipStream_t s) {
    const FoldArgs fa{tile_grp, weights, mask, fold_out, fold_ld, agg};
    return launch_score_i8(metric, rows, stride, dim, n_rows, norm2, qexact, qinfo, nb, out, ld, flag, &fa, n_cu, s);
}
