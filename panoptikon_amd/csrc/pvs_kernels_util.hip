// pvs_kernels_util.hip — codec, norms, query preparation, exact per-row scoring,
// synthetic rows.  gfx950.
#include "pvs_kernels.hpp"

// ------------------------------------------------------------------ norms
// norm2[r] = the reference's aMag for row r: sum a_i^2 accumulated sequentially in
// f32 (oracle: orc_vec_distance_cosine_*).  One lane per row; runs once per add.
// rnorm[r] = 1/sqrt(norm2[r]) (two correctly rounded steps): the scan's cosine filter key.
template <int DT>
__global__ __launch_bounds__(256) void k_norm2(const uint8_t *rows, uint32_t stride, int dim, uint64_t row0, uint64_t n,
                                               float *norm2, float *rnorm) {
    uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    const float aa = seq_sumsq<DT>(rows, stride, row0 + r, dim);
    norm2[row0 + r] = aa;
    rnorm[row0 + r] = __frcp_rn(__fsqrt_rn(aa));
}

hipError_t pvs_launch_norm2(int dtype, const uint8_t *rows, uint32_t stride, uint32_t dim, uint64_t row0, uint64_t n,
                            float *norm2, float *rnorm, hipStream_t s) {
    if (n == 0) return hipSuccess;
    dim3 g((unsigned)((n + 255) / 256)), b(256);
    if (dtype == PVS_I8)
        hipLaunchKernelGGL(k_norm2<PVS_I8>, g, b, 0, s, rows, stride, (int)dim, row0, n, norm2, rnorm);
    else if (dtype == PVS_F16)
        hipLaunchKernelGGL(k_norm2<PVS_F16>, g, b, 0, s, rows, stride, (int)dim, row0, n, norm2, rnorm);
    else
        hipLaunchKernelGGL(k_norm2<PVS_F32>, g, b, 0, s, rows, stride, (int)dim, row0, n, norm2, rnorm);
    return hipGetLastError();
}

// The scan's row-scalar stream: one record of PVS_AUX_REC floats per 32-row tile — the 32 row scalars (cosine: 1/|a|,
// L2: |a|^2; NaN on padding rows), then the tile's extremes for the pass-B pre-test (pvs_scan_kernel.hpp):
//   cosine  [32] = min |a|, [33] = max |a| over the rows with a finite 1/|a| (|a| = 1/(1/|a|), clamped to FLT_MAX);
//   L2      [32] = min |a|^2, [33] = max |a|^2 over the rows whose |a|^2 is not NaN (clamped to FLT_MAX);
//   no such row: min = +inf, max = -inf (every bound derived from them compares false).
// One wave per tile.
__global__ __launch_bounds__(256) void k_scan_aux(const float *norm2, const float *rnorm, uint64_t tile0, uint64_t ntiles, float *scan_cos,
                                                  float *scan_l2) {
    const uint64_t t = tile0 + (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (t >= tile0 + ntiles) return;
    const uint64_t r = t * 32 + (uint64_t)(lane & 31);
    const float rn = rnorm[r], n2 = norm2[r];
    const float FMAX = 3.402823466e38f, INF = __builtin_inff();
    const bool okc = rn == rn && rn < INF;
    const float w = fminf(__fdiv_rn(1.0f, rn), FMAX);
    float cmn = okc ? w : INF, cmx = okc ? w : -INF;
    const bool okl = n2 == n2;
    const float v = fminf(n2, FMAX);
    float lmn = okl ? v : INF, lmx = okl ? v : -INF;
    for (int off = 16; off > 0; off >>= 1) {
        cmn = fminf(cmn, __shfl_xor(cmn, off));
        cmx = fmaxf(cmx, __shfl_xor(cmx, off));
        lmn = fminf(lmn, __shfl_xor(lmn, off));
        lmx = fmaxf(lmx, __shfl_xor(lmx, off));
    }
    scan_cos[t * PVS_AUX_REC + lane] = lane < 32 ? rn : lane == 32 ? cmn : lane == 33 ? cmx : 0.f;
    scan_l2[t * PVS_AUX_REC + lane] = lane < 32 ? n2 : lane == 32 ? lmn : lane == 33 ? lmx : 0.f;
}
hipError_t pvs_launch_scan_aux(const float *norm2, const float *rnorm, uint64_t row0, uint64_t n, float *scan_cos, float *scan_l2,
                               hipStream_t s) {
    if (n == 0) return hipSuccess;
    const uint64_t t0 = row0 / 32, t1 = (row0 + n + 31) / 32;
    hipLaunchKernelGGL(k_scan_aux, dim3((unsigned)((t1 - t0 + 3) / 4)), dim3(256), 0, s, norm2, rnorm, t0, t1 - t0, scan_cos, scan_l2);
    return hipGetLastError();
}

__global__ void k_fill_f32(float *p, uint64_t n, float v) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        p[i] = v;
}
hipError_t pvs_launch_fill_f32(float *p, uint64_t n, float v, hipStream_t s) {
    if (n == 0) return hipSuccess;
    unsigned g = (unsigned)((n + 255) / 256);
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(k_fill_f32, dim3(g), dim3(256), 0, s, p, n, v);
    return hipGetLastError();
}

__global__ void k_iota_ids(int64_t *ids, uint64_t n, int64_t base) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        ids[i] = base + (int64_t)i;
}
hipError_t pvs_launch_iota_ids(int64_t *ids, uint64_t n, int64_t base, hipStream_t s) {
    if (n == 0) return hipSuccess;
    unsigned g = (unsigned)((n + 255) / 256);
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(k_iota_ids, dim3(g), dim3(256), 0, s, ids, n, base);
    return hipGetLastError();
}

// ------------------------------------------------------------------ codec
// quantize_int8 (db/vector_quants.rs:1489-1497): IEEE f32 division (never a
// reciprocal multiply), round-half-to-even (v_rndne_f32), clamp, NaN -> 0.
__device__ static inline int8_t quant_one(float x, float scale) {
    float q = rintf(__fdiv_rn(x, scale));
    q = fminf(fmaxf(q, -128.0f), 127.0f);  // fmin/fmax drop NaN, so test it explicitly
    float raw = __fdiv_rn(x, scale);
    return (raw != raw) ? (int8_t)0 : (int8_t)(int)q;
}

// strided destination (index rows): one thread per (row, 4 components)
// Ingest kernels: dense [n][dim] source rows -> tiled index rows row0.. (one thread per 16-byte
// destination chunk; source reads stay coalesced along the row).  MODE 0: int8 codes from f32
// (quantize_int8), 1: f16 from f32 (round-to-nearest-even), 2: copy of rows already in the index dtype.
template <int MODE>
__global__ __launch_bounds__(256) void k_rows_ingest(const void *src, uint32_t dim, uint32_t esz, uint64_t row0, uint64_t n, float scale,
                                                     uint8_t *rows, uint32_t stride) {
    const uint32_t per = 16u / esz;                        // elements per destination chunk
    const uint32_t cpr = (dim + per - 1) / per;            // chunks per row that hold data
    const uint64_t total = n * (uint64_t)cpr;
    for (uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (uint64_t)gridDim.x * 256) {
        const uint64_t r = t / cpr;
        const uint32_t c = (uint32_t)(t - r * cpr);
        uint32_t w[4] = {0, 0, 0, 0};
        for (uint32_t j = 0; j < per; j++) {
            const uint32_t e = c * per + j;
            if (e >= dim) break;
            if (MODE == 0) {
                const uint32_t v = (uint8_t)quant_one(((const float *)src)[r * dim + e], scale);
                w[j >> 2] |= v << ((j & 3) * 8);
            } else if (MODE == 1) {
                const _Float16 hv = (_Float16)((const float *)src)[r * dim + e];  // v_cvt_f16_f32: RNE
                w[j >> 1] |= (uint32_t)__builtin_bit_cast(uint16_t, hv) << ((j & 1) * 16);
            } else if (esz == 4) {
                w[j] = ((const uint32_t *)src)[r * dim + e];
            } else if (esz == 2) {
                w[j >> 1] |= (uint32_t)((const uint16_t *)src)[r * dim + e] << ((j & 1) * 16);
            } else {
                w[j >> 2] |= (uint32_t)((const uint8_t *)src)[r * dim + e] << ((j & 3) * 8);
            }
        }
        *(uint4 *)(rows + pvs_chunk_off(row0 + r, c, stride)) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}
// quantize_int8 without the IEEE division on the common path: t = x * (1/s) is within 2 ulp of x/s, so rint(t) = rint(fl(x/s))
// unless t sits within a few ulp of a half-integer (where the rounding of the quotient itself decides: the exact division then)
__device__ static inline int8_t quant_fast(float x, float scale, float inv_scale) {
    const float t = x * inv_scale;
    const float a = fabsf(t);
    const float f = a - floorf(a);
    if (__builtin_expect(!(fabsf(f - 0.5f) > a * 6.0e-7f + 1.0e-30f), 0)) return quant_one(x, scale);  // (also NaN)
    const float q = fminf(fmaxf(rintf(t), -128.0f), 127.0f);
    return (int8_t)(int)q;
}
// The build-side codec at streaming speed: one lane = four consecutive f32 components (16-byte loads, a wave reads 1 KiB
// contiguous) -> 4 int8 codes (one dword) or 4 f16 (two dwords) into the tiled layout.  dim % 4 == 0, 16-byte aligned source.
template <int MODE>
__global__ __launch_bounds__(256) void k_rows_ingest4(const float4 *src, uint32_t dim4, uint64_t row0, uint64_t n, float scale, float inv_scale,
                                                      uint8_t *rows, uint32_t stride) {
    const uint64_t total = n * (uint64_t)dim4;
    for (uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (uint64_t)gridDim.x * 256) {
        const uint64_t r = t / dim4;
        const uint32_t g = (uint32_t)(t - r * dim4);
        typedef float v4f_t __attribute__((ext_vector_type(4)));
        const v4f_t vv = __builtin_nontemporal_load((const v4f_t *)(src + t));  // read once
        const float4 v = make_float4(vv[0], vv[1], vv[2], vv[3]);
        if (MODE == 0) {
            const uint32_t w = (uint32_t)(uint8_t)quant_fast(v.x, scale, inv_scale) | ((uint32_t)(uint8_t)quant_fast(v.y, scale, inv_scale) << 8) |
                               ((uint32_t)(uint8_t)quant_fast(v.z, scale, inv_scale) << 16) | ((uint32_t)(uint8_t)quant_fast(v.w, scale, inv_scale) << 24);
            *(uint32_t *)(rows + pvs_chunk_off(row0 + r, g >> 2, stride) + (g & 3u) * 4u) = w;
        } else {
            const uint32_t lo = (uint32_t)__builtin_bit_cast(uint16_t, (_Float16)v.x) | ((uint32_t)__builtin_bit_cast(uint16_t, (_Float16)v.y) << 16);
            const uint32_t hi = (uint32_t)__builtin_bit_cast(uint16_t, (_Float16)v.z) | ((uint32_t)__builtin_bit_cast(uint16_t, (_Float16)v.w) << 16);
            *(uint2 *)(rows + pvs_chunk_off(row0 + r, g >> 1, stride) + (g & 1u) * 8u) = make_uint2(lo, hi);
        }
    }
}
hipError_t pvs_launch_rows_ingest(int mode, const void *src, uint32_t dim, uint32_t esz, uint64_t row0, uint64_t n, float scale,
                                  uint8_t *rows, uint32_t stride, hipStream_t s) {
    if (n == 0) return hipSuccess;
    if ((mode == 0 || mode == 1) && dim % 4 == 0 && ((uintptr_t)src & 15) == 0) {
        const uint64_t total = n * (uint64_t)(dim / 4);
        const unsigned g = (unsigned)std::min<uint64_t>((total + 255) / 256, 65536);
        if (mode == 0)
            hipLaunchKernelGGL(k_rows_ingest4<0>, dim3(g), dim3(256), 0, s, (const float4 *)src, dim / 4, row0, n, scale, 1.0f / scale, rows, stride);
        else
            hipLaunchKernelGGL(k_rows_ingest4<1>, dim3(g), dim3(256), 0, s, (const float4 *)src, dim / 4, row0, n, scale, 1.0f, rows, stride);
        return hipGetLastError();
    }
    const uint32_t per = 16u / esz;
    const uint64_t total = n * (uint64_t)((dim + per - 1) / per);
    unsigned g = (unsigned)((total + 255) / 256 > 32768 ? 32768 : (total + 255) / 256);
    if (mode == 0)
        hipLaunchKernelGGL(k_rows_ingest<0>, dim3(g), dim3(256), 0, s, src, dim, esz, row0, n, scale, rows, stride);
    else if (mode == 1)
        hipLaunchKernelGGL(k_rows_ingest<1>, dim3(g), dim3(256), 0, s, src, dim, esz, row0, n, scale, rows, stride);
    else
        hipLaunchKernelGGL(k_rows_ingest<2>, dim3(g), dim3(256), 0, s, src, dim, esz, row0, n, scale, rows, stride);
    return hipGetLastError();
}
// tiled rows row0..row0+n -> dense [n][dim*esz] (pvs_index_read_rows)
__global__ __launch_bounds__(256) void k_rows_gather(const uint8_t *rows, uint32_t stride, uint32_t row_bytes, uint64_t row0, uint64_t n,
                                                     uint8_t *dst) {
    const uint64_t total = n * (uint64_t)row_bytes;
    for (uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (uint64_t)gridDim.x * 256) {
        const uint64_t r = t / row_bytes;
        const uint32_t b = (uint32_t)(t - r * row_bytes);
        dst[t] = rows[pvs_chunk_off(row0 + r, b >> 4, stride) + (b & 15u)];
    }
}
hipError_t pvs_launch_rows_gather(const uint8_t *rows, uint32_t stride, uint32_t row_bytes, uint64_t row0, uint64_t n, uint8_t *dst,
                                  hipStream_t s) {
    if (n == 0) return hipSuccess;
    const uint64_t total = n * (uint64_t)row_bytes;
    unsigned g = (unsigned)((total + 255) / 256 > 32768 ? 32768 : (total + 255) / 256);
    hipLaunchKernelGGL(k_rows_gather, dim3(g), dim3(256), 0, s, rows, stride, row_bytes, row0, n, dst);
    return hipGetLastError();
}

// flat: 4 components per thread per step, 16-byte loads, 4-byte stores
__global__ __launch_bounds__(256) void k_quantize_flat(const float *src, uint64_t n, float scale, int8_t *dst) {
    const uint64_t n4 = n / 4;
    const bool aligned = (((uintptr_t)src & 15) == 0) && (((uintptr_t)dst & 3) == 0);
    uint64_t start = 0;
    if (aligned) {
        for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (uint64_t)gridDim.x * 256) {
            float4 v = ((const float4 *)src)[i];
            uint32_t o = (uint32_t)(uint8_t)quant_one(v.x, scale) | ((uint32_t)(uint8_t)quant_one(v.y, scale) << 8) |
                         ((uint32_t)(uint8_t)quant_one(v.z, scale) << 16) |
                         ((uint32_t)(uint8_t)quant_one(v.w, scale) << 24);
            ((uint32_t *)dst)[i] = o;
        }
        start = n4 * 4;
    }
    for (uint64_t i = start + (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256)
        dst[i] = quant_one(src[i], scale);
}
hipError_t pvs_launch_quantize_flat(const float *src, uint64_t n, float scale, int8_t *dst, hipStream_t s) {
    if (n == 0) return hipSuccess;
    uint64_t w = (n + 1023) / 1024;
    unsigned g = (unsigned)(w > 8192 ? 8192 : (w ? w : 1));
    hipLaunchKernelGGL(k_quantize_flat, dim3(g), dim3(256), 0, s, src, n, scale, dst);
    return hipGetLastError();
}

// blob_absmax over a flat array: `v > absmax` keeps NaN out, +-inf wins.  Non-
// negative floats order like their bit patterns, so the cross-workgroup reduce is
// one atomicMax on the u32 view.
__global__ __launch_bounds__(256) void k_absmax(const float *src, uint64_t n, uint32_t *out_bits) {
    float m = 0.0f;
    const uint64_t n4 = (((uintptr_t)src & 15) == 0) ? n / 4 : 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (uint64_t)gridDim.x * 256) {
        float4 v = ((const float4 *)src)[i];
        float a = fabsf(v.x), b = fabsf(v.y), c = fabsf(v.z), d = fabsf(v.w);
        if (a > m) m = a;
        if (b > m) m = b;
        if (c > m) m = c;
        if (d > m) m = d;
    }
    for (uint64_t i = n4 * 4 + (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        float a = fabsf(src[i]);
        if (a > m) m = a;
    }
    for (int off = 32; off > 0; off >>= 1) {
        float o = __shfl_xor(m, off);
        if (o > m) m = o;
    }
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++)
            if (part[w] > m) m = part[w];
        atomicMax(out_bits, __builtin_bit_cast(uint32_t, m));
    }
}
hipError_t pvs_launch_absmax(const float *src, uint64_t n, float *d_out_bits, hipStream_t s) {
    hipError_t e = hipMemsetAsync(d_out_bits, 0, 4, s);
    if (e != hipSuccess || n == 0) return e;
    uint64_t w = (n + 4095) / 4096;
    unsigned g = (unsigned)(w > 2048 ? 2048 : (w ? w : 1));
    hipLaunchKernelGGL(k_absmax, dim3(g), dim3(256), 0, s, src, n, (uint32_t *)d_out_bits);
    return hipGetLastError();
}

// ---------------------------------------------------------------- synthetic
__device__ static inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}
__device__ static inline int32_t synth_raw(uint64_t seed, uint64_t row, uint32_t col) {
    uint64_t key = seed * 0xD1342543DE82EF95ULL + row * 0x9E3779B97F4A7C15ULL + (uint64_t)col * 0xC2B2AE3D27D4EB4FULL;
    uint64_t a = splitmix64(key), b = splitmix64(key ^ 0xA5A5A5A5A5A5A5A5ULL), c = splitmix64(key + 0x1234567ULL);
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) s += (uint32_t)((a >> (16 * i)) & 0xffff);
#pragma unroll
    for (int i = 0; i < 4; i++) s += (uint32_t)((b >> (16 * i)) & 0xffff);
#pragma unroll
    for (int i = 0; i < 4; i++) s += (uint32_t)((c >> (16 * i)) & 0xffff);
    return (int32_t)s - 393210;
}
// One wave per row: exact integer sum of squares (order independent), so the
// bytes equal oracle/pvs_oracle.c orc_synth_rows.
__global__ __launch_bounds__(256) void k_synth(uint64_t seed, uint64_t row0, uint64_t n, uint32_t dim, float *out) {
    const int lane = threadIdx.x & 63;
    for (uint64_t r = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < n; r += (uint64_t)gridDim.x * 4) {
        long long ss = 0;
        for (uint32_t c = lane; c < dim; c += 64) {
            long long v = synth_raw(seed, row0 + r, c);
            ss += v * v;
        }
        for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
        float nrm = (float)__dsqrt_rn((double)ss * (1.0 / 4294967296.0));
        if (!(nrm > 0.0f)) nrm = 1.0f;
        for (uint32_t c = lane; c < dim; c += 64) {
            float g = (float)synth_raw(seed, row0 + r, c) * (1.0f / 65536.0f);
            out[r * dim + c] = __fdiv_rn(g, nrm);
        }
    }
}
hipError_t pvs_launch_synth(uint64_t seed, uint64_t row0, uint64_t n, uint32_t dim, float *out, hipStream_t s) {
    if (n == 0) return hipSuccess;
    uint64_t w = (n + 3) / 4;
    unsigned g = (unsigned)(w > 65536 ? 65536 : w);
    hipLaunchKernelGGL(k_synth, dim3(g), dim3(256), 0, s, seed, row0, n, dim, out);
    return hipGetLastError();
}

// ---- a corpus that looks like production embeddings instead of iid Gaussian directions (every timed number was on a
// distribution where distances concentrate at 1 +- 0.036 and the sampled threshold is at its best).  CLIP / mpnet embeddings are
// clustered, anisotropic and full of near-duplicates (docs/vector-int8-quant.md:220-224, docs/vector-quant-measurements.md:102-117):
//   * 2,000 clusters with power-law sizes: cluster = floor(2000 u^3), u uniform — the largest holds ~8 % of the rows;
//   * row = 2 x centre + noise, the noise ANISOTROPIC: per (cluster, block of 8 dimensions) divided by 1, 2, 4 or 8;
//   * one block of 8 consecutive rows in ten is a run of NEAR-duplicates (a shared vector + 1/16 of the noise): frames of a video,
//     crops of an image;
//   * one row in a hundred is an EXACT duplicate of a row up to 16 positions before it (re-imported files);
//   * unit-normalised.  Integer arithmetic up to the final IEEE operations, like k_synth: the oracle produces identical bytes.
__device__ static inline uint64_t synth_row_hash(uint64_t seed, uint64_t row) { return splitmix64(seed * 0xA24BAED4963EE407ULL + row * 0x9FB21C651E98DF25ULL + 0x51ED27ULL); }
__device__ static inline bool synth_is_dup(uint64_t seed, uint64_t row) { return row >= 17 && synth_row_hash(seed, row) % 100 == 0; }
// the row whose bytes row r carries (itself, or the earlier row it duplicates — one level: a duplicate of a duplicate is an original)
__device__ static inline uint64_t synth_source_row(uint64_t seed, uint64_t row) {
    if (!synth_is_dup(seed, row)) return row;
    const uint64_t src = row - 1 - ((synth_row_hash(seed, row) >> 8) & 15);
    return synth_is_dup(seed, src) ? row : src;
}
__device__ static inline uint32_t synth_cluster(uint64_t seed, uint64_t row) {
    const uint64_t u = synth_row_hash(seed ^ 0xC1u, row) >> 32;           // 32 uniform bits
    const uint64_t u3 = (((u * u) >> 32) * u) >> 32;                       // ~ u^3 / 2^64, < 2^32
    return (uint32_t)((u3 * 2000u) >> 32);
}
__device__ static inline int32_t synth_clustered_raw(uint64_t seed, uint64_t row, uint32_t col) {
    const uint32_t cl = synth_cluster(seed, row);
    const int sh = (int)(splitmix64(seed + 0x77u + (uint64_t)cl * 0x100000001B3ULL + (uint64_t)(col >> 3)) & 3);
    const int32_t centre = synth_raw(seed ^ 0xCE47E5ULL, cl, col);
    const uint64_t block = row >> 3;
    const bool near = synth_row_hash(seed ^ 0xB10Cu, block) % 10 == 0;
    int32_t v = 2 * centre;
    if (near) {
        // (the block's rows share the cluster of its first row: a run of near-duplicates lies in one cluster)
        const uint32_t cb = synth_cluster(seed, block << 3);
        const int shb = (int)(splitmix64(seed + 0x77u + (uint64_t)cb * 0x100000001B3ULL + (uint64_t)(col >> 3)) & 3);
        v = 2 * synth_raw(seed ^ 0xCE47E5ULL, cb, col) + (synth_raw(seed ^ 0x5A5AULL, block, col) >> shb) + (synth_raw(seed, row, col) >> (shb + 4));
    } else {
        v += synth_raw(seed, row, col) >> sh;
    }
    return v;
}
__global__ __launch_bounds__(256) void k_synth_clustered(uint64_t seed, uint64_t row0, uint64_t n, uint32_t dim, float *out) {
    const int lane = threadIdx.x & 63;
    for (uint64_t r = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < n; r += (uint64_t)gridDim.x * 4) {
        const uint64_t src = synth_source_row(seed, row0 + r);
        long long ss = 0;
        for (uint32_t c = lane; c < dim; c += 64) {
            long long v = synth_clustered_raw(seed, src, c);
            ss += v * v;
        }
        for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
        float nrm = (float)__dsqrt_rn((double)ss * (1.0 / 4294967296.0));
        if (!(nrm > 0.0f)) nrm = 1.0f;
        for (uint32_t c = lane; c < dim; c += 64) {
            float g = (float)synth_clustered_raw(seed, src, c) * (1.0f / 65536.0f);
            out[r * dim + c] = __fdiv_rn(g, nrm);
        }
    }
}
hipError_t pvs_launch_synth_clustered(uint64_t seed, uint64_t row0, uint64_t n, uint32_t dim, float *out, hipStream_t s) {
    if (n == 0) return hipSuccess;
    uint64_t w = (n + 3) / 4;
    unsigned g = (unsigned)(w > 65536 ? 65536 : w);
    hipLaunchKernelGGL(k_synth_clustered, dim3(g), dim3(256), 0, s, seed, row0, n, dim, out);
    return hipGetLastError();
}

// ------------------------------------------------------------ query prep
// One workgroup per (padded) query.
__global__ __launch_bounds__(256) void k_prep_queries(int index_dtype, int qdtype, const void *queries,
                                                      uint32_t batch, uint32_t dim, uint32_t stride, float scale,
                                                      int metric, uint8_t *qmat, void *qexact, QInfo *qinfo, uint32_t *need_dense) {
    const uint32_t b = blockIdx.x;
    const int tid = threadIdx.x;
    if (tid == 0 && b < batch) need_dense[b] = 0;  // per-search state (saves a memset launch)
    uint8_t *mrow = qmat + (uint64_t)b * stride;
    // zero the scan operand row (padding bytes and padding queries contribute 0 to every dot)
    for (uint32_t i = tid * 16; i < stride; i += 256 * 16) *(uint4 *)(mrow + i) = make_uint4(0, 0, 0, 0);
    __syncthreads();
    if (b >= batch) {
        if (tid == 0) {
            QInfo z;
            z.bb = 0.f; z.qn = 0.f; z.dscale = 0.f; z.eA = 0.f; z.eC = 0.f; z.eR = 0.f; z.pad0 = 0.f; z.pad1 = 0.f;
            qinfo[b] = z;
        }
        return;
    }
    __shared__ float red[4];
    __shared__ float s_amax;
    float dscale = 1.0f;
    if (index_dtype == PVS_I8) {
        int8_t *qe = (int8_t *)qexact + (uint64_t)b * dim;
        if (qdtype == PVS_I8) {
            const int8_t *q = (const int8_t *)queries + (uint64_t)b * dim;
            for (uint32_t i = tid; i < dim; i += 256) {
                qe[i] = q[i];
                mrow[i] = (uint8_t)q[i];
            }
        } else {
            const float *q = (const float *)queries + (uint64_t)b * dim;
            for (uint32_t i = tid; i < dim; i += 256) {
                int8_t c = quant_one(q[i], scale);  // compute_query_quant: the write-side codec
                qe[i] = c;
                mrow[i] = (uint8_t)c;
            }
        }
    } else {
        const float *q = (const float *)queries + (uint64_t)b * dim;
        float *qe = (float *)qexact + (uint64_t)b * dim;
        float m = 0.f;
        for (uint32_t i = tid; i < dim; i += 256) {
            float v = q[i];
            qe[i] = v;
            float a = fabsf(v);
            if (a > m && a < __builtin_inff()) m = a;
        }
        for (int off = 32; off > 0; off >>= 1) {
            float o = __shfl_xor(m, off);
            if (o > m) m = o;
        }
        if ((tid & 63) == 0) red[tid >> 6] = m;
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < 4; w++)
                if (red[w] > m) m = red[w];
            s_amax = m;
        }
        __syncthreads();
        // exact power-of-two prescale of the scan operand: the largest |q_i| lands in [2^13, 2^14) of its f16 image
        // (away from overflow and from the subnormal range)
        int e = 0;
        if (s_amax > 0.f) {
            int x;
            (void)frexpf(s_amax, &x);
            e = 14 - x;
            if (e > 100) e = 100;
            if (e < -100) e = -100;
        }
        dscale = ldexpf(1.0f, -e);
        // f16 and f32 indexes both take the f16 image of the prescaled query (f32 rows are narrowed to f16 in the kernel)
        for (uint32_t i = tid; i < dim; i += 256) {
            _Float16 h = (_Float16)ldexpf(q[i], e);
            *(uint16_t *)(mrow + 2 * (uint64_t)i) = __builtin_bit_cast(uint16_t, h);
        }
    }
    __syncthreads();
    // the reference's bMag: sequential f32 sum of squares of what it scores.  int8 codes: integer
    // terms, exact in any order while < 2^24 -> parallel integer sum; otherwise (and for floats)
    // one lane walks the vector in order, from a coalesced LDS copy.
    __shared__ __attribute__((aligned(16))) float s_q[4096];
    __shared__ long long s_isum[4];
    float bb = 0.f;
    bool have_bb = false;
    if (index_dtype == PVS_I8) {
        const int8_t *qe = (const int8_t *)qexact + (uint64_t)b * dim;
        long long part = 0;
        for (uint32_t i = tid; i < dim; i += 256) part += (long long)qe[i] * qe[i];
        for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
        if ((tid & 63) == 0) s_isum[tid >> 6] = part;
        __syncthreads();
        const long long tot = s_isum[0] + s_isum[1] + s_isum[2] + s_isum[3];
        if (tot < 16777216) {
            bb = (float)tot;
            have_bb = true;
        }
    } else if (dim <= 4096) {
        // the products (one rounding each) by every thread, the chain of additions — the only sequential part — by one lane from
        // 16-byte LDS reads: the same roundings in the same order as the reference's loop, in a third of the time
        const float *qe = (const float *)qexact + (uint64_t)b * dim;
        const uint32_t dim4 = (dim + 3u) & ~3u;
        for (uint32_t i = tid; i < dim4; i += 256) s_q[i] = i < dim ? __fmul_rn(qe[i], qe[i]) : 0.f;
        __syncthreads();
        if (tid == 0) {
            float acc = 0.f;
            const uint32_t full = dim & ~3u;
            for (uint32_t i = 0; i < full; i += 4) {
                const float4 v = *(const float4 *)(s_q + i);
                acc = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(acc, v.x), v.y), v.z), v.w);
            }
            for (uint32_t i = full; i < dim; i++) acc = __fadd_rn(acc, s_q[i]);  // (never the zero padding: -0 + 0 would lose a sign)
            bb = acc;
        }
        have_bb = true;
    }
    if (tid == 0) {
        if (!have_bb) {
            if (index_dtype == PVS_I8)
                bb = seq_sumsq_dense<PVS_I8>((const int8_t *)qexact + (uint64_t)b * dim, (int)dim);
            else
                bb = seq_sumsq_dense<PVS_F32>((const float *)qexact + (uint64_t)b * dim, (int)dim);
        }
        QInfo qi;
        qi.bb = bb;
        qi.qn = sqrtf(bb);
        qi.dscale = dscale;
        qi.pad0 = 0.f;
        qi.pad1 = 0.f;
        // Error budget of the scan key (HISTORY.md §4.2): f32 accumulation of K terms is within
        // K*2^-24 of sum|terms| in either evaluation order; an f16 query image adds 2^-11 |a||q|;
        // f32 index: rows narrowed to f16 toward zero after a per-row power-of-two scaling (2^-10, plus sqrt(D) 2^-26 for
        // the components the scaling leaves below the f16 normal range) and the f16 query image (2^-11): 1.5e-3 |a||q|.
        const float acc = (float)dim * 6.0e-8f;
        const float qround = (index_dtype == PVS_F16) ? 4.9e-4f : (index_dtype == PVS_F32) ? 1.5e-3f : 0.0f;
        if (metric == PVS_COSINE) {
            qi.eA = (qround + 4.0f * acc + 4.0e-6f) * qi.qn;  // (the matrix core's internal adds may chop: twice the IEEE budget)
            qi.eC = 0.f;
            qi.eR = 0.f;
        } else {
            const float coef = qround + 6.0f * acc + 1.0e-6f;
            qi.eA = coef * bb + (index_dtype == PVS_I8 ? 2.0f : 0.0f);
            qi.eC = 0.f;
            qi.eR = coef;
        }
        qinfo[b] = qi;
    }
}

hipError_t pvs_launch_prep_queries(int index_dtype, int qdtype, const void *queries, uint32_t batch,
                                   uint32_t batch_pad, uint32_t dim, uint32_t stride, float scale, int metric,
                                   uint8_t *qmat, void *qexact, QInfo *qinfo, uint32_t *need_dense, hipStream_t s) {
    hipLaunchKernelGGL(k_prep_queries, dim3(batch_pad), dim3(256), 0, s, index_dtype, qdtype, queries, batch, dim,
                       stride, scale, metric, qmat, qexact, qinfo, need_dense);
    return hipGetLastError();
}

// ------------------------------------------------------------ candidate mask
// The scan's row-scalar stream (records of PVS_AUX_REC floats per 32-row tile, k_scan_aux) with NaN in place of the
// scalar of every row the mask leaves out; the tile extremes stay those of the unmasked tile (still necessary
// conditions: a subset's extremes lie inside them).
__global__ __launch_bounds__(256) void k_mask_aux(const float *aux, const uint8_t *mask, uint64_t n, uint64_t cap, float *out) {
    const uint64_t total = cap / 32 * PVS_AUX_REC;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (uint64_t)gridDim.x * 256) {
        const uint32_t e = (uint32_t)(i % PVS_AUX_REC);
        const uint64_t r = i / PVS_AUX_REC * 32 + e;
        out[i] = (e >= 32 || (r < n && mask[r])) ? aux[i] : __builtin_nanf("");
    }
}
hipError_t pvs_launch_mask_aux(const float *aux, const uint8_t *mask, uint64_t n, uint64_t cap, float *out, hipStream_t s) {
    if (cap == 0) return hipSuccess;
    const unsigned g = (unsigned)std::min<uint64_t>((cap * 2 + 255) / 256, 16384);
    hipLaunchKernelGGL(k_mask_aux, dim3(g), dim3(256), 0, s, aux, mask, n, cap, out);
    return hipGetLastError();
}

// ---- multi-device index, by-group placement of device-resident rows (pvs_multi.hip): the rows of one add that belong to one shard,
// picked by index out of the caller's buffer (row-major, row_bytes each) into a dense block
__global__ __launch_bounds__(256) void k_pick_rows(const uint8_t *src, uint32_t row_bytes, const uint32_t *idx, uint64_t m, uint8_t *dst) {
    if ((row_bytes & 15u) == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15u) == 0) {
        const uint32_t per = row_bytes >> 4;
        const uint64_t total = m * per;
        for (uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (uint64_t)gridDim.x * 256) {
            const uint64_t r = t / per;
            const uint32_t c = (uint32_t)(t - r * per);
            ((uint4 *)dst)[t] = ((const uint4 *)(src + (uint64_t)idx[r] * row_bytes))[c];
        }
    } else {
        const uint64_t total = m * row_bytes;
        for (uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (uint64_t)gridDim.x * 256) {
            const uint64_t r = t / row_bytes;
            dst[t] = src[(uint64_t)idx[r] * row_bytes + (t - r * row_bytes)];
        }
    }
}
template <int DT>
__global__ __launch_bounds__(256) void k_rows_to_queries(const uint8_t *rows, uint32_t stride, uint32_t dim, const uint32_t *idx, void *out) {
    const uint32_t i = blockIdx.x;
    const uint64_t r = idx[i];
    constexpr uint32_t ESZ = DT == PVS_I8 ? 1 : DT == PVS_F16 ? 2 : 4;
    for (uint32_t e = threadIdx.x; e < dim; e += 256) {
        const uint32_t b = e * ESZ;
        const uint8_t *p = rows + pvs_chunk_off(r, b >> 4, stride) + (b & 15u);
        if constexpr (DT == PVS_I8)
            ((int8_t *)out)[(size_t)i * dim + e] = (int8_t)*p;
        else if constexpr (DT == PVS_F16)
            ((float *)out)[(size_t)i * dim + e] = h2f(*(const uint16_t *)p);
        else
            ((float *)out)[(size_t)i * dim + e] = *(const float *)p;
    }
}
hipError_t pvs_launch_rows_to_queries(int dtype, const uint8_t *rows, uint32_t stride, uint32_t dim, const uint32_t *idx, uint32_t m, void *out, hipStream_t s) {
    if (m == 0) return hipSuccess;
    if (dtype == PVS_I8)
        hipLaunchKernelGGL(k_rows_to_queries<PVS_I8>, dim3(m), dim3(256), 0, s, rows, stride, dim, idx, out);
    else if (dtype == PVS_F16)
        hipLaunchKernelGGL(k_rows_to_queries<PVS_F16>, dim3(m), dim3(256), 0, s, rows, stride, dim, idx, out);
    else
        hipLaunchKernelGGL(k_rows_to_queries<PVS_F32>, dim3(m), dim3(256), 0, s, rows, stride, dim, idx, out);
    return hipGetLastError();
}
hipError_t pvs_launch_pick_rows(const void *src, uint32_t row_bytes, const uint32_t *idx, uint64_t m, void *dst, hipStream_t s) {
    if (m == 0) return hipSuccess;
    const uint64_t total = m * (uint64_t)row_bytes / 16 + 1;
    const unsigned g = (unsigned)std::min<uint64_t>((total + 255) / 256, 16384);
    hipLaunchKernelGGL(k_pick_rows, dim3(g), dim3(256), 0, s, (const uint8_t *)src, row_bytes, idx, m, (uint8_t *)dst);
    return hipGetLastError();
}
// a per-row array over the GLOBAL rows of a multi-device index -> one shard's local row order: out[l] = in[global_row[l]]
// (global_row: the shard's rows of the segment table, expanded once per index state)
template <typename T>
__global__ __launch_bounds__(256) void k_take_rows(const T *in, const uint32_t *global_row, uint64_t n_local, T *out) {
    for (uint64_t l = (uint64_t)blockIdx.x * 256 + threadIdx.x; l < n_local; l += (uint64_t)gridDim.x * 256) out[l] = in[global_row[l]];
}
hipError_t pvs_launch_take_rows(const void *in, uint32_t elem_bytes, const uint32_t *global_row, uint64_t n_local, void *out, hipStream_t s) {
    if (n_local == 0) return hipSuccess;
    const unsigned g = (unsigned)std::min<uint64_t>((n_local + 255) / 256, 16384);
    if (elem_bytes == 1)
        hipLaunchKernelGGL(k_take_rows<uint8_t>, dim3(g), dim3(256), 0, s, (const uint8_t *)in, global_row, n_local, (uint8_t *)out);
    else if (elem_bytes == 4)
        hipLaunchKernelGGL(k_take_rows<uint32_t>, dim3(g), dim3(256), 0, s, (const uint32_t *)in, global_row, n_local, (uint32_t *)out);
    else if (elem_bytes == 8)
        hipLaunchKernelGGL(k_take_rows<uint64_t>, dim3(g), dim3(256), 0, s, (const uint64_t *)in, global_row, n_local, (uint64_t *)out);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}
