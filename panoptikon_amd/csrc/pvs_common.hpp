// pvs_common.hpp — shared declarations of libpvs (host + device).
// MI355X / gfx950 only: no CUDA paths, no portability macros.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "pvs.h"

#define PVS_EXPORT extern "C" __attribute__((visibility("default")))

// ------------------------------------------------------------------ errors
pvs_status pvs_fail(pvs_status code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess)                                                                \
            return pvs_fail(_e == hipErrorOutOfMemory ? PVS_ERR_OOM : PVS_ERR_DEVICE,        \
                            "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                            __LINE__);                                                       \
    } while (0)

#define PVS_TRY(expr)                 \
    do {                              \
        pvs_status _s = (expr);       \
        if (_s != PVS_OK) return _s;  \
    } while (0)

// ------------------------------------------------------------ geometry
// Rows live in HBM at a pitch that is a multiple of 256 B so that a row is a
// whole number of 16-chunk (256 B) "k-slabs": the unit the scan kernel streams
// through LDS and XOR-swizzles.
constexpr uint32_t PVS_KSLAB_BYTES = 256;
constexpr uint32_t PVS_TILE_ROWS = 32;    // one MFMA 32x32 tile of rows
constexpr uint32_t PVS_ROW_ALIGN = 128;   // capacity granularity (largest WG tile: 4 row tiles)
constexpr uint32_t PVS_MAX_BATCH = 128;   // queries per scan pass (4 waves x 32)
constexpr uint32_t PVS_MAX_K = 2048;      // page size served by the filter path
constexpr uint32_t PVS_CAND_CAP = 16384;  // candidate slots per query
constexpr uint32_t PVS_SURV_CAP = 4096;   // survivors reranked exactly per query

static inline uint32_t pvs_esz(uint32_t dtype) { return dtype == PVS_F32 ? 4u : dtype == PVS_F16 ? 2u : 1u; }
static inline uint64_t pvs_round_up(uint64_t v, uint64_t m) { return (v + m - 1) / m * m; }

// Per-query constants produced by the query-prep kernel and consumed by the
// scan epilogue / finaliser.  key = monotone surrogate of the reference distance
// (cosine: -dot/|a| ; L2: |a|^2 + |q|^2 - 2 dot); err = eA + eC*|a| + eR*|a|^2 is a
// rigorous bound on |key - key_ref|, so [key-err, key+err] brackets
// the value the reference ordering is monotone in.
struct QInfo {
    float bb;      // sum q_i^2, accumulated sequentially in f32 (the reference's bMag)
    float qn;      // sqrt(bb)
    float dscale;  // multiply the MFMA dot by this (undo the f16 power-of-two prescale)
    float eA, eC, eR;
    float pad0, pad1;
};

// ------------------------------------------------------ device helpers
#if defined(__HIPCC__)

// Order-preserving f32 -> u32 (ascending), NaN last.
__host__ __device__ static inline uint32_t f32_sort_key(float f) {
    uint32_t b = __builtin_bit_cast(uint32_t, f);
    if ((b & 0x7fffffffu) > 0x7f800000u) return 0xffffffffu;  // NaN (either sign) sorts last
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__host__ __device__ static inline float f32_from_sort_key(uint32_t k) {
    uint32_t b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    if (k == 0xffffffffu) b = 0x7fc00000u;
    return __builtin_bit_cast(float, b);
}

__device__ static inline float ld_elem_f32(const float *p, int i) { return p[i]; }

// IEEE binary16 -> f32 (exact) via the hardware converter.
__device__ static inline float h2f(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }

// Visits the components of one stored row in order, 16 bytes per load.
// f(i, v): v is int for I8 rows, float (exactly widened) for F16/F32 rows.
template <int DT, typename F>
__device__ static inline void row_foreach(const uint8_t *row, int dim, F &&f) {
    constexpr int PER = DT == PVS_I8 ? 16 : DT == PVS_F16 ? 8 : 4;
    const int full = dim / PER;
    const uint4 *p = (const uint4 *)row;
    for (int c = 0; c < full; c++) {
        const uint4 v = p[c];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < PER; j++) {
            if constexpr (DT == PVS_I8)
                f(c * PER + j, (int)(int8_t)(w[j >> 2] >> ((j & 3) * 8)));
            else if constexpr (DT == PVS_F16)
                f(c * PER + j, h2f((uint16_t)(w[j >> 1] >> ((j & 1) * 16))));
            else
                f(c * PER + j, __builtin_bit_cast(float, w[j]));
        }
    }
    for (int i = full * PER; i < dim; i++) {  // ragged tail (dim not a multiple of 16 bytes)
        if constexpr (DT == PVS_I8)
            f(i, (int)((const int8_t *)row)[i]);
        else if constexpr (DT == PVS_F16)
            f(i, h2f(((const uint16_t *)row)[i]));
        else
            f(i, ((const float *)row)[i]);
    }
}

// Sequential-f32 restatement of sqlite-vec's scalar kernels (oracle/pvs_oracle.c):
// one rounding per multiply and one per add (__fmul_rn/__fadd_rn forbid FMA
// contraction), components in order.  DT = dtype of the stored row.
//   I8 : query is int8 codes;  F16/F32 : query is f32.
template <int DT>
__device__ static inline float seq_dot(const uint8_t *row, const void *q, int dim) {
    float dot = 0.0f;
    if constexpr (DT == PVS_I8) {
        const int8_t *b = (const int8_t *)q;
        row_foreach<DT>(row, dim, [&](int i, int a) { dot = __fadd_rn(dot, (float)(a * (int)b[i])); });
    } else {
        const float *b = (const float *)q;
        row_foreach<DT>(row, dim, [&](int i, float a) { dot = __fadd_rn(dot, __fmul_rn(a, b[i])); });
    }
    return dot;
}

template <int DT>
__device__ static inline float seq_sumsq_diff(const uint8_t *row, const void *q, int dim) {
    float res = 0.0f;
    if constexpr (DT == PVS_I8) {
        const int8_t *b = (const int8_t *)q;
        row_foreach<DT>(row, dim, [&](int i, int a) {
            float t = (float)(a - (int)b[i]);
            res = __fadd_rn(res, __fmul_rn(t, t));
        });
    } else {
        const float *b = (const float *)q;
        row_foreach<DT>(row, dim, [&](int i, float a) {
            float t = __fsub_rn(a, b[i]);
            res = __fadd_rn(res, __fmul_rn(t, t));
        });
    }
    return res;
}

template <int DT>
__device__ static inline float seq_sumsq(const uint8_t *row, int dim) {
    float aa = 0.0f;
    if constexpr (DT == PVS_I8) {
        row_foreach<DT>(row, dim, [&](int, int a) { aa = __fadd_rn(aa, (float)(a * a)); });
    } else {
        row_foreach<DT>(row, dim, [&](int, float a) { aa = __fadd_rn(aa, __fmul_rn(a, a)); });
    }
    return aa;
}

// return sqrt(res): sqrt evaluated in double (IEEE, correctly rounded), narrowed to f32.
__device__ static inline float ref_l2_finish(float res) { return (float)__dsqrt_rn((double)res); }

// return (f32)(1 - dot / (sqrt(aa) * sqrt(bb))) evaluated in double.
__device__ static inline float ref_cosine_finish(float dot, float aa, float bb) {
    double den = __dsqrt_rn((double)aa) * __dsqrt_rn((double)bb);
    return (float)(1.0 - (double)dot / den);
}

template <int DT>
__device__ static inline float exact_distance(const uint8_t *row, const void *q, int dim, int metric,
                                              float aa, float bb) {
    if (metric == PVS_L2) return ref_l2_finish(seq_sumsq_diff<DT>(row, q, dim));
    return ref_cosine_finish(seq_dot<DT>(row, q, dim), aa, bb);
}

#endif  // __HIPCC__
