// pvs_dense_exact.hip — exact distance of every stored row to a few queries, in the
// reference's arithmetic: the `d` column of the MATERIALIZED dist_{cte}
// (filters/exact.rs:106-134; vec_distance_cosine / vec_distance_L2 of sqlite-vec 0.1.9,
// restated in oracle/pvs_oracle.c orc_vec_distance_*).
//
// The reference accumulates component by component in f32 (one rounding per multiply, one
// per add), so a row's sum is one dependent chain and cannot go to the matrix core (int8
// rows below 2^24 have a closed form and do: k_scan MODE 2).  What can be fixed is the memory
// side.  One lane owns one row (the chain), a wave owns 64 rows = two 32-row tiles of the
// tiled layout, and the corpus streams HBM -> LDS by LDS-DMA in k-slabs (64 rows x 256 B =
// 16 KiB per wave, double buffered, no workgroup barrier: a wave consumes only what it
// loaded).  Lanes then read their own row with conflict-free ds_read_b128 (the XOR swizzle
// of the layout) and multiply by query components that are wave-uniform: the zero-padded
// queries sit in LDS too and come back as broadcast reads.  (They must not be VMEM loads: a
// wave's loads return in order, so a query load issued behind the prefetch DMA would wait
// for the whole next slab and serialise the pipeline.)  Up to four queries share one pass
// over the rows (four independent chains per lane); float rows take EIGHT per pass (round 4): the chains of a pair of
// queries are the two halves of v_pk_mul_f32 / v_pk_add_f32 (one IEEE rounding per half and instruction, the same chain
// as the scalar form), the queries sit in LDS interleaved by pairs so that one broadcast ds_read_b128 delivers the
// packed operands of two components.  At 8 queries the pass stays HBM-bound for f32 rows (9 LDS reads and 32 packed
// VALU per 16-B chunk and lane); f16 rows become LDS-bound (17 reads per chunk) and gain less.
//
// Roofline: HBM.  Algorithmic bytes per launch = rows x row pitch.  VALU work per 16 KiB
// slab and wave, f16 rows: 128 components x (1 convert + 2 per query) instructions.
#include "pvs_kernels.hpp"
#include "pvs_lds_dma.hpp"

namespace {

struct DenseK {
    const uint8_t *rows;
    const float *norm2;
    const float *__restrict__ qpad;  // [nq][qpad_ld] f32, zero padded to the row pitch
    const QInfo *qinfo;
    float *out;  // out[row * out_ld + out_col + q]
    uint64_t n_rows;
    uint32_t stride, kslabs, qpad_ld, out_ld, out_col, n_pairs, n_waves;
    // work distribution (round 5, as k_direct_topk): units of >= 48 KB of rows, the first static_rounds per wave dealt, the rest
    // dequeued from four counters (one per wave slot, 256 B apart; zeroed by k_pad_queries in front of every launch)
    uint32_t *ctr;
    uint32_t unit, n_units, static_rounds, dyn;
};

constexpr int DENSE_WAVE_LDS = 2 * 16384;
constexpr int DENSE_RING_LDS = 4 * DENSE_WAVE_LDS;
constexpr int DENSE_Q_LDS = 160 * 1024 - DENSE_RING_LDS;  // what is left of the CU's LDS holds the queries

template <int DT>
__device__ static inline float elem_f32(const uint4 &v, int e) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    if constexpr (DT == PVS_I8)
        return (float)(int)(int8_t)(w[e >> 2] >> ((e & 3) * 8));  // |a*b| <= 2^14: the float product is the integer product
    else if constexpr (DT == PVS_F16)
        return h2f((uint16_t)(w[e >> 1] >> ((e & 1) * 16)));
    else
        return __builtin_bit_cast(float, w[e]);
}

typedef float v2f __attribute__((ext_vector_type(2)));

template <int DT, int NQ, int METRIC>
__global__ __launch_bounds__(256, 1) void k_dense_exact(DenseK a) {
    constexpr bool PK = NQ >= 8;  // query pairs on the packed f32 pipe; LDS copy of the queries interleaved by pairs
    constexpr int PER = DT == PVS_I8 ? 16 : DT == PVS_F16 ? 8 : 4;  // components per 16-B chunk
    constexpr int EPS = 16 * PER;                                    // components per 256-B slab row
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float *const qlds = (const float *)(smem + DENSE_RING_LDS);
    {
        float *w = (float *)(smem + DENSE_RING_LDS);
        if constexpr (PK) {  // [pair][component][2]
            for (uint32_t i = threadIdx.x; i < (uint32_t)NQ * a.qpad_ld; i += 256) {
                const uint32_t q = i / a.qpad_ld, x = i - q * a.qpad_ld;
                w[((size_t)(q >> 1) * a.qpad_ld + x) * 2 + (q & 1)] = a.qpad[i];
            }
        } else {
            for (uint32_t i = threadIdx.x; i < (uint32_t)NQ * a.qpad_ld; i += 256) w[i] = a.qpad[i];
        }
        __syncthreads();  // the only workgroup barrier
    }
    // Work is dequeued, not dealt (round 5): equal shares end at very different times (690k x 768 int8: 65 / 80 / 117 us min / mean /
    // max over workgroups for a stream whose bytes take 80) — see pvs_direct_kernel.hpp for the measurements and the reasons behind
    // the named accumulation registers the asynchronous values live in.
    const uint32_t gw = blockIdx.x * 4 + wave;
    if (gw >= a.n_units) return;
    uint8_t *const wbuf = smem + wave * DENSE_WAVE_LDS;
    const uint32_t wlds = lds_addr(wbuf);
    const uint32_t voff = (uint32_t)lane * 16u;

    auto uni = [](const uint8_t *p) {  // keep the DMA base in SGPRs
        const uint64_t v = (uint64_t)(uintptr_t)p;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        return (const uint8_t *)(uintptr_t)(((uint64_t)hi << 32) | lo);
    };
    uint32_t ipair = 0, iend = 0, islab = 0, round = 0, nxt_unit = 0, issued = 0, consumed = 0, last_pair = 0;
    bool nxt_ok = false, pending = false, more_dyn = a.dyn != 0;
    auto unit_range = [&](uint32_t u) {
        ipair = u * a.unit;
        iend = min(ipair + a.unit, a.n_pairs);
        islab = 0;
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
        if (ipair == iend) {
            if (!nxt_ok) return false;
            unit_range(nxt_unit);
            fetch_next();
        }
        const uint8_t *bA = uni(a.rows + (uint64_t)ipair * 64 * a.stride + (uint64_t)islab * 8192);  // k-slab of tile 2*pair: 8 KiB contiguous
        const uint8_t *bB = uni(bA + 32ull * a.stride);                                                // ... of tile 2*pair+1
        const uint32_t dst = wlds + (issued & 1u) * 16384u;
#pragma unroll
        for (int e = 0; e < 8; e++) dma16(bA + e * 1024, voff, dst + e * 1024);
#pragma unroll
        for (int e = 0; e < 8; e++) dma16(bB + e * 1024, voff, dst + 8192 + e * 1024);
        last_pair = ipair;
        issued++;
        if (++islab == a.kslabs) {
            islab = 0;
            ipair++;
        }
        return true;
    };
    unit_range(gw);
    fetch_next();
    (void)issue_one();
    uint32_t cpair = last_pair;

    float acc[NQ], bbv[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        acc[q] = 0.0f;
        bbv[q] = a.qinfo[q].bb;  // (read once, in front of the loop: inside it the load waited for the whole prefetched stage, NQ times per pair)
    }
    v2f acc2[NQ / 2 + 1];
#pragma unroll
    for (int p = 0; p < NQ / 2; p++) acc2[p] = v2f{0.0f, 0.0f};
    const uint32_t row_in = (uint32_t)(lane >> 5) * 8192u + (uint32_t)(lane & 31) * 256u;
    const uint32_t jx = (uint32_t)lane & 15u;
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
        const bool last_slab = cs + 1 == a.kslabs;
        const uint64_t row = (uint64_t)cpair * 64 + (uint32_t)lane;
        if (METRIC == PVS_COSINE && last_slab && row < a.n_rows) row_scalars_async(a.norm2 + row, nullptr, nullptr);  // (in front of the next stage's DMA)
        const bool fed = issue_one();  // streams in while this item is consumed
        const uint8_t *tile = wbuf + (consumed & 1u) * 16384u + row_in;
        const float *q0 = qlds + (size_t)cs * EPS;
        if constexpr (PK) {
            const float *q0p = qlds + (size_t)cs * EPS * 2;
#pragma unroll
            for (int c = 0; c < 16; c++) {
                const uint4 v = *(const uint4 *)(tile + ((((uint32_t)c) ^ jx) << 4));
                float4 qv4[NQ / 2][PER / 2];  // pair p, components 2x, 2x+1: (q0 c0, q1 c0, q0 c1, q1 c1)
#pragma unroll
                for (int p = 0; p < NQ / 2; p++)
#pragma unroll
                    for (int x = 0; x < PER / 2; x++) qv4[p][x] = *(const float4 *)(q0p + ((size_t)p * a.qpad_ld + c * PER + 2 * x) * 2);  // broadcast
#pragma unroll
                for (int e = 0; e < PER; e++) {
                    const float av = elem_f32<DT>(v, e);
                    const v2f av2 = v2f{av, av};
#pragma unroll
                    for (int p = 0; p < NQ / 2; p++) {
                        const float4 &t4 = qv4[p][e >> 1];
                        const v2f qv = (e & 1) == 0 ? v2f{t4.x, t4.y} : v2f{t4.z, t4.w};
                        if (METRIC == PVS_COSINE) {
                            acc2[p] = acc2[p] + av2 * qv;  // (-ffp-contract=off: one rounding per multiply, one per add)
                        } else {
                            const v2f t = av2 - qv;
                            acc2[p] = acc2[p] + t * t;
                        }
                    }
                }
            }
        } else {
    #pragma unroll
            for (int c = 0; c < 16; c++) {
                const uint4 v = *(const uint4 *)(tile + ((((uint32_t)c) ^ jx) << 4));
                float4 qv4[NQ][PER / 4];
    #pragma unroll
                for (int q = 0; q < NQ; q++)
    #pragma unroll
                    for (int x = 0; x < PER / 4; x++) qv4[q][x] = *(const float4 *)(q0 + (size_t)q * a.qpad_ld + c * PER + 4 * x);  // broadcast
    #pragma unroll
                for (int e = 0; e < PER; e++) {
                    const float av = elem_f32<DT>(v, e);
    #pragma unroll
                    for (int q = 0; q < NQ; q++) {
                        const float4 &t4 = qv4[q][e >> 2];
                        const float qv = (e & 3) == 0 ? t4.x : (e & 3) == 1 ? t4.y : (e & 3) == 2 ? t4.z : t4.w;
                        if (METRIC == PVS_COSINE) {
                            acc[q] = __fadd_rn(acc[q], __fmul_rn(av, qv));
                        } else {
                            const float t = __fsub_rn(av, qv);
                            acc[q] = __fadd_rn(acc[q], __fmul_rn(t, t));
                        }
                    }
                }
            }
        }
        consumed++;
        if (++cs == a.kslabs) {
            uint32_t r_aa = 0, r_1, r_2;
            if (METRIC == PVS_COSINE) {
                if (fed)
                    row_scalars_wait<16>(r_aa, r_1, r_2);
                else
                    row_scalars_wait<0>(r_aa, r_1, r_2);
            }
            if (row < a.n_rows) {
                const float aa = METRIC == PVS_COSINE ? __builtin_bit_cast(float, r_aa) : 0.f;
#pragma unroll
                for (int q = 0; q < NQ; q++) {
                    const float sum = PK ? acc2[q >> 1][q & 1] : acc[q];
                    const float d = METRIC == PVS_COSINE ? ref_cosine_finish(sum, aa, bbv[q]) : ref_l2_finish(sum);
                    a.out[row * a.out_ld + a.out_col + q] = d;
                }
            }
#pragma unroll
            for (int q = 0; q < NQ; q++) acc[q] = 0.0f;
#pragma unroll
            for (int p = 0; p < NQ / 2; p++) acc2[p] = v2f{0.0f, 0.0f};
            cs = 0;
            cpair = last_pair;
        }
    }
    wait_vm<0>();
}

// [nq][dim] int8 codes or f32 -> [nq][ld] f32, zero padded
__global__ __launch_bounds__(256) void k_pad_queries(const void *qexact, int is_i8, uint32_t dim, uint32_t ld, float *qpad, uint32_t *ctr) {
    const uint32_t q = blockIdx.x;
    if (q == 0 && threadIdx.x < 4) ctr[64 * threadIdx.x] = 0;  // the dequeue counters of the launch behind this one
    for (uint32_t i = threadIdx.x; i < ld; i += 256) {
        float v = 0.f;
        if (i < dim) v = is_i8 ? (float)((const int8_t *)qexact)[(size_t)q * dim + i] : ((const float *)qexact)[(size_t)q * dim + i];
        qpad[(size_t)q * ld + i] = v;
    }
}

template <int DT, int NQ, int METRIC>
hipError_t launch_one(const DenseK &k, uint32_t grid, hipStream_t s) {
    static std::atomic<bool> configured{false};
    if (!configured.load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void *)k_dense_exact<DT, NQ, METRIC>, hipFuncAttributeMaxDynamicSharedMemorySize, DENSE_RING_LDS + DENSE_Q_LDS);
        if (e != hipSuccess) return e;
        configured.store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((k_dense_exact<DT, NQ, METRIC>), dim3(grid), dim3(256), DENSE_RING_LDS + DENSE_Q_LDS, s, k);
    return hipGetLastError();
}
template <int DT, int NQ>
hipError_t launch_metric(const DenseK &k, int metric, uint32_t grid, hipStream_t s) {
    return metric == PVS_COSINE ? launch_one<DT, NQ, PVS_COSINE>(k, grid, s) : launch_one<DT, NQ, PVS_L2>(k, grid, s);
}
template <int DT>
hipError_t launch_nq(const DenseK &k, uint32_t nq, int metric, uint32_t grid, hipStream_t s) {
    switch (nq) {
        case 1: return launch_metric<DT, 1>(k, metric, grid, s);
        case 2: return launch_metric<DT, 2>(k, metric, grid, s);
        case 4: return launch_metric<DT, 4>(k, metric, grid, s);
        case 8:
            if constexpr (DT != PVS_I8) return launch_metric<DT, 8>(k, metric, grid, s);
    }
    return hipErrorInvalidValue;
}
}  // namespace

static inline uint64_t dense_q_bytes(uint32_t stride, uint32_t esz) { return pvs_round_up((uint64_t)PVS_DENSE_NQ * (stride / esz) * 4, 256); }
// the padded queries, then four dequeue counters 256 B apart, then the wide pass's transposed queries (float rows)
uint64_t pvs_dense_exact_scratch_bytes(uint32_t stride, uint32_t esz) { return dense_q_bytes(stride, esz) + 1024 + pvs_exact_wide_scratch_bytes(stride, esz); }

hipError_t pvs_launch_dense_exact(int dtype, int metric, const uint8_t *rows, uint32_t stride, uint32_t dim, uint64_t n,
                                  const float *norm2, const void *qexact, const QInfo *qinfo, uint32_t nq, float *qpad_scratch,
                                  float *out, uint32_t out_ld, uint32_t out_col, uint32_t n_cu, hipStream_t s) {
    if (n == 0 || nq == 0) return hipSuccess;
    const uint32_t esz = pvs_esz((uint32_t)dtype);
    DenseK k;
    k.rows = rows;
    k.norm2 = norm2;
    k.qpad = qpad_scratch;
    k.out = out;
    k.n_rows = n;
    k.stride = stride;
    k.kslabs = stride / PVS_KSLAB_BYTES;
    k.qpad_ld = stride / esz;
    k.out_ld = out_ld;
    k.n_pairs = (uint32_t)((n + 63) / 64);
    const uint32_t grid = std::min<uint32_t>((k.n_pairs + 3) / 4, std::max<uint32_t>(n_cu, 1));
    k.n_waves = grid * 4;
    k.ctr = (uint32_t *)((uint8_t *)qpad_scratch + dense_q_bytes(stride, esz));
    k.unit = std::max<uint32_t>(1, (49152u + 64u * stride - 1) / (64u * stride));
    k.n_units = (k.n_pairs + k.unit - 1) / k.unit;
    k.static_rounds = std::max<uint32_t>(1, k.n_units / k.n_waves / 2);
    k.dyn = (uint64_t)k.static_rounds * k.n_waves < k.n_units ? 1u : 0u;
    const size_t qsz = dtype == PVS_I8 ? 1 : 4;
    for (uint32_t q0 = 0; q0 < nq;) {
        const uint32_t wide = dtype != PVS_I8 && !pvs_dbg(PVS_DBG_NO_EXACT_WIDE) ? pvs_exact_wide_fit(stride, esz) : 0u;
        if (wide && nq - q0 > PVS_DENSE_NQ) {
            // nine and more queries over float rows: 16 or 32 per pass, four rows per lane (pvs_exact_wide.hip)
            const uint32_t g = std::min<uint32_t>(nq - q0, wide);
            hipError_t e = pvs_launch_exact_wide(dtype, metric, rows, stride, dim, n, norm2, (const float *)qexact + (size_t)q0 * dim, qinfo + q0, g,
                                                 (float *)((uint8_t *)qpad_scratch + dense_q_bytes(stride, esz) + 1024), out, out_ld, out_col + q0, n_cu, s);
            if (e != hipSuccess) return e;
            q0 += g;
            continue;
        }
        uint32_t g = nq - q0 >= 4 ? 4 : nq - q0 >= 2 ? 2 : 1;
        if (dtype != PVS_I8 && nq - q0 >= 8 && !pvs_dbg(PVS_DBG_DENSE_NQ4)) g = 8;
        while (g > 1 && (uint64_t)g * k.qpad_ld * 4 > (uint64_t)DENSE_Q_LDS) g >>= 1;  // queries must fit beside the ring
        if ((uint64_t)g * k.qpad_ld * 4 > (uint64_t)DENSE_Q_LDS) return hipErrorInvalidValue;  // row pitch > 32 KiB (dim > 8192 f32)
        hipLaunchKernelGGL(k_pad_queries, dim3(g), dim3(256), 0, s, (const void *)((const uint8_t *)qexact + (size_t)q0 * dim * qsz),
                           dtype == PVS_I8 ? 1 : 0, dim, k.qpad_ld, qpad_scratch, k.ctr);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        k.qinfo = qinfo + q0;
        k.out_col = out_col + q0;
        if (g == 8 && dtype != PVS_I8 && pvs_dense_exact2_fits(stride, esz) && !pvs_dbg(PVS_DBG_NO_DENSE2)) {
            // eight float queries: two rows per lane share every query read (pvs_dense_exact2.hip)
            e = pvs_launch_dense_exact2(dtype, metric, rows, stride, n, norm2, qpad_scratch, k.ctr, k.qinfo, out, out_ld, k.out_col, n_cu, s);
            if (e != hipSuccess) return e;
            q0 += g;
            continue;
        }
        e = dtype == PVS_I8    ? launch_nq<PVS_I8>(k, g, metric, grid, s)
            : dtype == PVS_F16 ? launch_nq<PVS_F16>(k, g, metric, grid, s)
                               : launch_nq<PVS_F32>(k, g, metric, grid, s);
        if (e != hipSuccess) return e;
        q0 += g;
    }
    return hipSuccess;
}
