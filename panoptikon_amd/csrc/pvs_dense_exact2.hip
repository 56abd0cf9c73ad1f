// pvs_dense_exact2.hip — k_dense_exact for EIGHT float queries with TWO rows per lane (round 5): the `d` column of the
// MATERIALIZED dist_{cte} (filters/exact.rs:106-134; vec_distance_cosine / vec_distance_L2 of sqlite-vec 0.1.9, restated in
// oracle/pvs_oracle.c orc_vec_distance_*) for similar_to's target vectors and a handful of exact-mode queries.
//
// k_dense_exact at 8 queries is bound by the LDS pipe: per 16-byte row chunk a lane issues 1 read of its row and 8 (f32 rows) or
// 16 (f16) broadcast reads of query components — 4M x 768 f16: 1.9 ms per pass where HBM needs 0.8 and the packed VALU 0.7.
// A query read can serve two rows if a lane owns two: a wave then covers 128 rows (four 32-row tiles) and a 16 KiB stage holds
// HALF a k-slab of them (128 bytes per row: one cache line) instead of a whole slab of 64 rows.  The half-slab is not contiguous
// in the tiled layout, so the LDS-DMA gathers it: 8 lanes fetch the 8 chunks of one row's line, one instruction moves 8 full
// lines.  The LDS image is row-major [128 rows][8 slots of 16 B] with the slot XOR-swizzled by row & 7 (the lane that fetches
// slot k of row r asks for logical chunk k ^ (r & 7)), so the 8 rows a ds_read_b128 serves per cycle hit 8 different slots.
// LDS reads per pair of row chunks: 2 + 8 (f32) / 2 + 16 (f16) instead of 18 / 34.
// Everything else is k_dense_exact's: ring of two stages per wave, no workgroup barrier in the loop, units dequeued by the waves,
// the row norms loaded asynchronously in front of a stage, chains of query pairs on v_pk_mul_f32 / v_pk_add_f32.
//
// Roofline: HBM (f32 rows) / LDS + packed VALU (f16 rows).  Algorithmic bytes per launch = rows x row pitch.
#include "pvs_kernels.hpp"
#include "pvs_lds_dma.hpp"

namespace {

struct Dense2K {
    const uint8_t *rows;
    const float *norm2;
    const float *__restrict__ qpad;  // [8][qpad_ld] f32, zero padded to the row pitch
    const QInfo *qinfo;
    float *out;  // out[row * out_ld + out_col + q]
    uint64_t n_rows;
    uint32_t stride, kslabs, qpad_ld, out_ld, out_col, n_quads, n_waves, n_tiles;
    uint32_t *ctr;  // four dequeue counters 256 B apart (zeroed by k_pad_queries in front of the launch)
    uint32_t unit, n_units, static_rounds, dyn;
};

constexpr int D2_STAGE = 16384, D2_WAVE_LDS = 2 * D2_STAGE, D2_RING_LDS = 4 * D2_WAVE_LDS;
constexpr int NQ2 = 8;
typedef float v2f __attribute__((ext_vector_type(2)));

// the second row's norm beside the first (row_scalars_wait reads a250 / a251 / a252 back)
__device__ static inline void norm2_pair_async(const float *a, const float *b) {
    asm volatile("global_load_dword a250, %0, off" ::"v"(a) : "memory", "a250");
    asm volatile("global_load_dword a252, %0, off" ::"v"(b) : "memory", "a252");
}

template <int DT, int METRIC>
__global__ __launch_bounds__(256, 1) void k_dense_exact2(Dense2K a) {
    constexpr int PER = DT == PVS_F16 ? 8 : 4;  // components per 16-B chunk
    constexpr int EPH = 8 * PER;                // components per half-slab row (128 B)
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float *const qlds = (const float *)(smem + D2_RING_LDS);
    {
        float *w = (float *)(smem + D2_RING_LDS);  // [pair][component][2]
        for (uint32_t i = threadIdx.x; i < (uint32_t)NQ2 * a.qpad_ld; i += 256) {
            const uint32_t q = i / a.qpad_ld, x = i - q * a.qpad_ld;
            w[((size_t)(q >> 1) * a.qpad_ld + x) * 2 + (q & 1)] = a.qpad[i];
        }
        __syncthreads();  // the only workgroup barrier
    }
    const uint32_t gw = blockIdx.x * 4 + wave;
    if (gw >= a.n_units) return;
    uint8_t *const wbuf = smem + wave * D2_WAVE_LDS;
    const uint32_t wlds = lds_addr(wbuf);
    auto uni = [](const uint8_t *p) {  // keep the DMA base in SGPRs
        const uint64_t v = (uint64_t)(uintptr_t)p;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        return (const uint8_t *)(uintptr_t)(((uint64_t)hi << 32) | lo);
    };
    // the lane's part in a DMA instruction: row (lane >> 3) of the instruction's 8 rows, slot lane & 7 of that row's line.
    // vo[e4]: offset inside the tile's k-slab of what it fetches for instruction e4 (rows 8 e4 .. 8 e4 + 7 of a tile) in the first half
    // of the slab; the second half is the other 128 bytes of the same row: the offset with bit 7 flipped
    uint32_t vo[4];
#pragma unroll
    for (int e4 = 0; e4 < 4; e4++) {
        const uint32_t r32 = (uint32_t)e4 * 8 + ((uint32_t)lane >> 3), k = (uint32_t)lane & 7u;
        vo[e4] = r32 * 256u + (((k ^ (r32 & 7u)) ^ (r32 & 15u)) << 4);
    }
    uint32_t iquad = 0, iend = 0, istep = 0, round = 0, nxt_unit = 0, issued = 0, consumed = 0, last_quad = 0;
    bool nxt_ok = false, pending = false, more_dyn = a.dyn != 0;
    const uint32_t n_steps = a.kslabs * 2;  // half-slabs per row
    auto unit_range = [&](uint32_t u) {
        iquad = u * a.unit;
        iend = min(iquad + a.unit, a.n_quads);
        istep = 0;
    };
    auto fetch_next = [&]() {
        round++;
        if (round < a.static_rounds) {
            nxt_unit = gw + round * a.n_waves;
            nxt_ok = nxt_unit < a.n_units;
        } else if (more_dyn) {
            if (lane == 0) dequeue_async(a.ctr + 64 * wave);
            pending = true;
            nxt_ok = false;
        } else {
            nxt_ok = false;
        }
    };
    auto issue_one = [&]() -> bool {
        if (iquad == iend) {
            if (!nxt_ok) return false;
            unit_range(nxt_unit);
            fetch_next();
        }
        const uint32_t j = istep >> 1, h = istep & 1u;
        const uint32_t dst = wlds + (issued & 1u) * (uint32_t)D2_STAGE;
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const uint32_t tile = min(iquad * 4 + (uint32_t)t, a.n_tiles - 1);  // (a quad past the last tile re-reads it; nothing is written for it)
            const uint8_t *b = uni(a.rows + (uint64_t)tile * 32 * a.stride + (uint64_t)j * 8192);
#pragma unroll
            for (int e4 = 0; e4 < 4; e4++) dma16(b, vo[e4] ^ (h << 7), dst + (uint32_t)(t * 4 + e4) * 1024u);
        }
        last_quad = iquad;
        issued++;
        if (++istep == n_steps) {
            istep = 0;
            iquad++;
        }
        return true;
    };
    unit_range(gw);
    fetch_next();
    (void)issue_one();
    uint32_t cquad = last_quad;

    float bbv[NQ2];
#pragma unroll
    for (int q = 0; q < NQ2; q++) bbv[q] = a.qinfo[q].bb;
    v2f acc[2][NQ2 / 2];
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int p = 0; p < NQ2 / 2; p++) acc[r][p] = v2f{0.0f, 0.0f};
    const uint32_t sw = (uint32_t)lane & 7u;                    // the rows' slot swizzle (row-in-quad & 7 = lane & 7 for both rows)
    const uint32_t row_in[2] = {(uint32_t)lane * 128u, ((uint32_t)lane + 64u) * 128u};
    uint32_t cs = 0;
    while (consumed < issued) {
        const uint32_t deq = wait_all_and_dequeued();  // the stage has landed (and nothing else is outstanding)
        if (pending) {
            const uint32_t u = a.static_rounds * a.n_waves + 4u * (uint32_t)__builtin_amdgcn_readfirstlane((int)deq) + (uint32_t)wave;
            pending = false;
            nxt_unit = u;
            nxt_ok = u < a.n_units;
            more_dyn = nxt_ok;
        }
        const bool last_step = cs + 1 == n_steps;
        const uint64_t row0 = (uint64_t)cquad * 128 + (uint32_t)lane, row1 = row0 + 64;
        if (METRIC == PVS_COSINE && last_step) norm2_pair_async(a.norm2 + min(row0, a.n_rows - 1), a.norm2 + min(row1, a.n_rows - 1));  // (in front of the next stage's DMA)
        const bool fed = issue_one();  // streams in while this stage is consumed
        const uint8_t *st = wbuf + (consumed & 1u) * (uint32_t)D2_STAGE;
        const float *q0p = qlds + ((size_t)(cs >> 1) * (2 * EPH) + (size_t)(cs & 1u) * EPH) * 2;  // [pair][component][2]: component offset of this half
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const uint32_t slot = (((uint32_t)c) ^ sw) << 4;
            const uint4 v0 = *(const uint4 *)(st + row_in[0] + slot), v1 = *(const uint4 *)(st + row_in[1] + slot);
            float4 qv4[NQ2 / 2][PER / 2];  // pair p, components 2x, 2x+1: (q0 c0, q1 c0, q0 c1, q1 c1)
#pragma unroll
            for (int p = 0; p < NQ2 / 2; p++)
#pragma unroll
                for (int x = 0; x < PER / 2; x++) qv4[p][x] = *(const float4 *)(q0p + ((size_t)p * a.qpad_ld + c * PER + 2 * x) * 2);  // broadcast
#pragma unroll
            for (int e = 0; e < PER; e++) {
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    const uint4 &v = r ? v1 : v0;
                    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
                    float av;
                    if constexpr (DT == PVS_F16)
                        av = h2f((uint16_t)(w[e >> 1] >> ((e & 1) * 16)));
                    else
                        av = __builtin_bit_cast(float, w[e]);
                    const v2f av2 = v2f{av, av};
#pragma unroll
                    for (int p = 0; p < NQ2 / 2; p++) {
                        const float4 &t4 = qv4[p][e >> 1];
                        const v2f qv = (e & 1) == 0 ? v2f{t4.x, t4.y} : v2f{t4.z, t4.w};
                        if (METRIC == PVS_COSINE) {
                            acc[r][p] = acc[r][p] + av2 * qv;  // (-ffp-contract=off: one rounding per multiply, one per add)
                        } else {
                            const v2f t = av2 - qv;
                            acc[r][p] = acc[r][p] + t * t;
                        }
                    }
                }
            }
        }
        consumed++;
        if (++cs == n_steps) {
            uint32_t r_aa0 = 0, r_1, r_aa1 = 0;
            if (METRIC == PVS_COSINE) {
                if (fed)
                    row_scalars_wait<16>(r_aa0, r_1, r_aa1);
                else
                    row_scalars_wait<0>(r_aa0, r_1, r_aa1);
            }
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const uint64_t row = r ? row1 : row0;
                if (row < a.n_rows) {
                    const float aa = METRIC == PVS_COSINE ? __builtin_bit_cast(float, r ? r_aa1 : r_aa0) : 0.f;
#pragma unroll
                    for (int q = 0; q < NQ2; q++) {
                        const float sum = acc[r][q >> 1][q & 1];
                        a.out[row * a.out_ld + a.out_col + q] = METRIC == PVS_COSINE ? ref_cosine_finish(sum, aa, bbv[q]) : ref_l2_finish(sum);
                    }
                }
#pragma unroll
                for (int p = 0; p < NQ2 / 2; p++) acc[r][p] = v2f{0.0f, 0.0f};
            }
            cs = 0;
            cquad = last_quad;
        }
    }
    wait_vm<0>();
}

template <int DT, int METRIC>
hipError_t launch_one(const Dense2K &k, uint32_t grid, size_t lds, hipStream_t s) {
    static std::atomic<bool> configured{false};
    if (!configured.load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void *)k_dense_exact2<DT, METRIC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        configured.store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((k_dense_exact2<DT, METRIC>), dim3(grid), dim3(256), lds, s, k);
    return hipGetLastError();
}
}  // namespace

// do 8 padded f32 queries of this pitch fit beside the ring?
bool pvs_dense_exact2_fits(uint32_t stride, uint32_t esz) { return (uint64_t)NQ2 * (stride / esz) * 4 + D2_RING_LDS <= 160 * 1024; }

// exactly 8 float queries, already padded to [8][stride / esz] f32 in `qpad` (k_pad_queries, which also zeroes `ctr`)
hipError_t pvs_launch_dense_exact2(int dtype, int metric, const uint8_t *rows, uint32_t stride, uint64_t n, const float *norm2, const float *qpad, uint32_t *ctr,
                                   const QInfo *qinfo, float *out, uint32_t out_ld, uint32_t out_col, uint32_t n_cu, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const uint32_t esz = pvs_esz((uint32_t)dtype);
    if (dtype == PVS_I8 || !pvs_dense_exact2_fits(stride, esz)) return hipErrorInvalidValue;
    Dense2K k;
    k.rows = rows;
    k.norm2 = norm2;
    k.qpad = qpad;
    k.qinfo = qinfo;
    k.out = out;
    k.n_rows = n;
    k.stride = stride;
    k.kslabs = stride / PVS_KSLAB_BYTES;
    k.qpad_ld = stride / esz;
    k.out_ld = out_ld;
    k.out_col = out_col;
    k.n_quads = (uint32_t)((n + 127) / 128);
    k.n_tiles = (uint32_t)((n + 31) / 32);
    const uint32_t grid = std::min<uint32_t>((k.n_quads + 3) / 4, std::max<uint32_t>(n_cu, 1));
    k.n_waves = grid * 4;
    k.ctr = ctr;
    k.unit = std::max<uint32_t>(1, (49152u + 128u * stride - 1) / (128u * stride));  // >= 48 KB of rows per unit
    k.n_units = (k.n_quads + k.unit - 1) / k.unit;
    k.static_rounds = std::max<uint32_t>(1, k.n_units / k.n_waves / 2);
    k.dyn = (uint64_t)k.static_rounds * k.n_waves < k.n_units ? 1u : 0u;
    const size_t lds = (size_t)D2_RING_LDS + (size_t)NQ2 * k.qpad_ld * 4;
    if (dtype == PVS_F16) return metric == PVS_COSINE ? launch_one<PVS_F16, PVS_COSINE>(k, grid, lds, s) : launch_one<PVS_F16, PVS_L2>(k, grid, lds, s);
    return metric == PVS_COSINE ? launch_one<PVS_F32, PVS_COSINE>(k, grid, lds, s) : launch_one<PVS_F32, PVS_L2>(k, grid, lds, s);
}
