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

template <int NQ, bool COS>
__global__ __launch_bounds__(256) void k_score_i8_direct(const uint8_t *__restrict__ rows, uint32_t stride, uint64_t n_rows, uint32_t n_tiles,
                                                          const float *__restrict__ norm2, const int8_t *__restrict__ qexact, uint32_t dim,
                                                          const QInfo *__restrict__ qinfo, uint32_t nb, float *__restrict__ out, uint32_t ld,
                                                          uint32_t *flag) {
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
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            int mine = 0;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int t = row16_sum(acc[j][q]);
                mine = (int)pc == j ? t : mine;
            }
            const uint64_t row = (uint64_t)tile * 32u + 4u * pc + rsub;
            if (pc < 8 && row < n_rows && q < (int)nb) {
                const float aa = norm2[row];
                float d;
                if (COS) {
                    d = ref_cosine_finish((float)mine, aa, bb[q]);
                } else {
                    const double ss = (double)aa + (double)bb[q] - 2.0 * (double)mine;
                    if (!(ss < 16777216.0)) *flag = 1u;  // (a plain store: the word may live in pinned host memory; 1 is the only value ever written)
                    d = ref_l2_finish((float)ss);
                }
                out[(size_t)row * ld + q] = d;
            }
        }
        tile += t_step;
    }
}
}  // namespace

// out[row * ld + q] for q < nb <= 4.  qexact: [nb][dim] int8 codes (dense), qinfo[q].bb = |q|^2.
hipError_t pvs_launch_score_i8_direct(int metric, const uint8_t *rows, uint32_t stride, uint32_t dim, uint64_t n_rows, const float *norm2,
                                      const void *qexact, const QInfo *qinfo, uint32_t nb, float *out, uint32_t ld, uint32_t *flag,
                                      uint32_t n_cu, hipStream_t s) {
    if (nb == 0 || nb > 4 || n_rows == 0) return nb == 0 || n_rows == 0 ? hipSuccess : hipErrorInvalidValue;
    const uint32_t n_tiles = (uint32_t)((n_rows + 31) / 32);
    const int nq = nb == 1 ? 1 : nb == 2 ? 2 : 4;
    // 8 waves per CU, 16 KiB in flight each.  Measured on MI355X, 25M rows, one query (rocprofv3): 512-B rows 1.94 ms = 6.6 TB/s
    // (k_scan MODE 2: 2.43 ms; 1 / 4 / 8 workgroups per CU: 2.37 / 2.10 / 2.05 ms), 1-KiB rows 4.06-4.08 ms = 6.3 TB/s whatever
    // the occupancy (MODE 2: 3.99 ms) — 25.6 GB streams no faster than that on this part.
    const uint32_t wg_per_cu = 2;
    const uint32_t grid = std::min<uint32_t>((n_tiles + 3) / 4, n_cu * wg_per_cu);
    const size_t lds = (size_t)nq * stride;
#define PVS_SD_LAUNCH(NQ, COS)                                                                                                          \
    hipLaunchKernelGGL((k_score_i8_direct<NQ, COS>), dim3(grid), dim3(256), lds, s, rows, stride, n_rows, n_tiles, norm2, (const int8_t *)qexact, \
                       dim, qinfo, nb, out, ld, flag)
    const bool cos = metric == PVS_COSINE;
    if (nq == 1) {
        if (cos) PVS_SD_LAUNCH(1, true); else PVS_SD_LAUNCH(1, false);
    } else if (nq == 2) {
        if (cos) PVS_SD_LAUNCH(2, true); else PVS_SD_LAUNCH(2, false);
    } else {
        if (cos) PVS_SD_LAUNCH(4, true); else PVS_SD_LAUNCH(4, false);
    }
#undef PVS_SD_LAUNCH
    return hipGetLastError();
}
