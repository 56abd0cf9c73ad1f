// pvs_rerank.hpp — the exact distance of ONE (row, query) pair in the reference's order, for a lane that is alone with a cold row:
// pass C's rerank of the filter scan's survivors (pvs_kernels_scan.hip) and the gather-and-score path of sparse candidate sets
// (pvs_sparse.hip).  Device code only.
#pragma once
#include "pvs_common.hpp"

// Pass C's exact distance of one survivor, in the reference's order (sqlite-vec's scalar kernels: one rounding per multiply and
// per add, components in sequence — oracle/pvs_oracle.c), written for a lane that is alone with a cold row: the row streams from
// global memory in groups of 8 sixteen-byte chunks with the next group requested before the current one is consumed, the query
// comes from LDS (s_q, zero-padded to a whole chunk) one vector read per chunk, and whole chunks are processed without a bounds
// test per component — the padding of row and query is zero, and adding +0 products changes nothing a distance can show (at
// most the sign of a zero dot product, which `1 - dot/den` does not see).  The generic form (exact_distance<DT> with the query in
// global memory) compiled to a flat load of the query plus `s_waitcnt vmcnt(0)` per component: ~125 cycles per component,
// 50-85 us of a 90-140 us finaliser for 768-d f16 rows.
typedef unsigned int fin_u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) fin_u32x4 *gchunk_ptr;
template <int DT>
__device__ static inline float rerank_distance(const uint8_t *rows, uint32_t stride, uint64_t r, const uint8_t *s_q, int dim, int metric, float aa,
                                               float bb) {
    constexpr int PER = DT == PVS_I8 ? 16 : DT == PVS_F16 ? 8 : 4;
    constexpr int UN = 8;  // 2 x 8 loads in flight per lane (16, 24 and 48 measured the same or worse: the lanes of a wave touch 64 different lines per load)
    const int nchunks = (dim + PER - 1) / PER;
    const bool l2 = metric == PVS_L2;
    float acc = 0.0f;
    auto step = [&](float av, float qv) {
        if (l2) {
            const float t = __fsub_rn(av, qv);
            acc = __fadd_rn(acc, __fmul_rn(t, t));
        } else {
            acc = __fadd_rn(acc, __fmul_rn(av, qv));
        }
    };
    auto visit = [&](int c, const uint4 &v) {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        if constexpr (DT == PVS_I8) {
            const uint4 qv = ((const uint4 *)s_q)[c];
            const uint32_t qw[4] = {qv.x, qv.y, qv.z, qv.w};
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const int ai = (int)(int8_t)(w[j >> 2] >> ((j & 3) * 8)), qi = (int)(int8_t)(qw[j >> 2] >> ((j & 3) * 8));
                if (l2) {  // (integers below 2^17: the f32 images and the product are exact, as in the reference)
                    const float t = (float)(ai - qi);
                    acc = __fadd_rn(acc, __fmul_rn(t, t));
                } else {
                    acc = __fadd_rn(acc, (float)(ai * qi));
                }
            }
        } else if constexpr (DT == PVS_F16) {
            const float4 q0 = ((const float4 *)s_q)[2 * c], q1 = ((const float4 *)s_q)[2 * c + 1];
            const float qf[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
            for (int j = 0; j < 8; j++) step(h2f((uint16_t)(w[j >> 1] >> ((j & 1) * 16))), qf[j]);
        } else {
            const float4 q0 = ((const float4 *)s_q)[c];
            const float qf[4] = {q0.x, q0.y, q0.z, q0.w};
#pragma unroll
            for (int j = 0; j < 4; j++) step(__builtin_bit_cast(float, w[j]), qf[j]);
        }
    };
    auto load = [&](int c) {  // (an explicit global-memory load: pointers that arrive inside a by-value kernel argument struct compile to flat loads)
        const fin_u32x4 v = *(gchunk_ptr)(uintptr_t)(rows + pvs_chunk_off(r, (uint32_t)c, stride));
        return make_uint4(v.x, v.y, v.z, v.w);
    };
    int c = 0;
    if (nchunks >= UN) {
        uint4 cur[UN], nxt[UN];
#pragma unroll
        for (int u = 0; u < UN; u++) cur[u] = load(u);
        for (; c + UN <= nchunks; c += UN) {
            const bool more = c + 2 * UN <= nchunks;
            if (more) {
#pragma unroll
                for (int u = 0; u < UN; u++) nxt[u] = load(c + UN + u);
            }
#pragma unroll
            for (int u = 0; u < UN; u++) visit(c + u, cur[u]);
            if (more) {
#pragma unroll
                for (int u = 0; u < UN; u++) cur[u] = nxt[u];
            }
        }
    }
    for (; c < nchunks; c++) visit(c, load(c));
    return l2 ? ref_l2_finish(acc) : ref_cosine_finish(acc, aa, bb);
}

