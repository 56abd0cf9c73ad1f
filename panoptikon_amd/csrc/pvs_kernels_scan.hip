// pvs_kernels_scan.hip — the hot path: a bandwidth-bound filter scan of the corpus
// on the CDNA4 matrix cores, a radix k-th select, and the exact finaliser.
//
// Replaces, for a batch of queries, the reference's per-row
//   vec_distance_{cosine,L2}(payload, ?)          (image_embeddings.rs:321-362,
//   ... ORDER BY order_rank ASC ... LIMIT k          text_embeddings.rs:386-418, builder.rs:578-582)
// The reference scores every row and sorts everything.  Here:
//   pass A  scan a strided sample of row tiles, keep per-lane minima of an UPPER bound of
//           the key -> the k-th smallest of those group minima is a valid upper bound T of
//           the k-th best key of the whole corpus;
//   pass B  scan every row once, emit (row, key) for rows whose LOWER bound is <= T;
//   pass C  per query: k-th smallest upper bound among the candidates -> survivors
//           (lower bound <= that) -> EXACT sequential-f32 distance (bit-identical to the
//           oracle) -> sort by (distance, row) -> first k.
// The corpus is read from HBM exactly once in pass B (plus the sample in pass A).
//
// Scan kernel geometry (gfx950: 64-wide waves, 4 SIMDs/CU, 160 KiB LDS/CU):
//   workgroup = 4 waves; wave (qg, rt) owns query group qg (32 queries, held in VGPRs
//   for the whole kernel as MFMA B fragments) and row sub-tile rt (32 rows);
//   QG = batch_pad/32 in {1,2,4}, RT = 4/QG, workgroup tile = 32*RT rows.
//   The corpus streams HBM -> LDS by LDS-DMA (global_load_lds_dwordx4, no VGPR round
//   trip) in "slabs" of (32*RT rows x 256 B), NS-deep ring, P = NS-1 slabs in flight,
//   one s_barrier per slab, counted s_waitcnt vmcnt (never 0 in steady state).
//   A fragments are ds_read_b128 from an XOR-swizzled slab image (chunk ^= row & 15,
//   applied on the DMA *source* address; the LDS destination is lane-linear), which is
//   bank-conflict free for the 16-lane groups ds_read_b128 is serviced in.
//   v_mfma_i32_32x32x32_i8 / v_mfma_f32_32x32x16_f16: A = 32 corpus rows, B = 32 queries,
//   so each lane ends up with ONE query (lane & 31) and 16 rows: the per-query threshold
//   is a lane-private register and the epilogue is branch-free until a row passes.
#include <cstdlib>

#include "pvs_kernels.hpp"

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

#define PVS_LDS(p) ((__attribute__((address_space(3))) void *)(p))
#define PVS_GLB(p) ((const __attribute__((address_space(1))) void *)(p))

// LDS-DMA issued from inline asm: hipcc models the builtin form as an LDS store that may
// alias every later ds_read and drains vmcnt(0) in front of them (two full pipeline
// drains per tile in the first build of this kernel).  An asm statement is invisible to
// its waitcnt insertion, so the counted s_waitcnt vmcnt(N) below are the only waits.
// M0 carries the wave-uniform LDS destination; each lane lands at M0 + lane*size.
__device__ static inline void dma16(const void *gsrc, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
}
__device__ static inline void dma4(const void *gsrc, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
}
__device__ static inline uint32_t lds_addr(const void *p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t *)p;
}

template <int N>
__device__ static inline void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ static inline void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int LCAP = 512;      // LDS candidate staging entries per workgroup
constexpr int FLUSH_AT = 192;  // flush to HBM at a tile boundary once this many are staged

template <int DT>
struct Acc;
template <>
struct Acc<PVS_I8> {
    using type = v16i;
    __device__ static inline type mfma(v4i a, v4i b, type c) { return __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0); }
    __device__ static inline float tof(int v, float) { return (float)v; }
};
template <>
struct Acc<PVS_F16> {
    using type = v16f;
    __device__ static inline type mfma(v4i a, v4i b, type c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, a), __builtin_bit_cast(v8h, b), c, 0, 0, 0);
    }
    __device__ static inline float tof(float v, float dscale) { return v * dscale; }
};

struct ScanK {
    const uint8_t *rows;
    const float *norm2;
    const uint8_t *qmat;
    const QInfo *qinfo;
    const float *thr;
    float *gmin;
    uint32_t *cand_cnt;
    uint2 *cand;
    uint64_t n_rows;
    uint32_t stride, n_wgtiles, tile_step, groups_per_query, cand_cap;
    int metric, mode;
    int debug;  // profiling ablations (PVS_SCAN_DEBUG): 1 = no MFMA, 2 = no epilogue, 4 = no DMA in the loop
};

template <int QG>
struct Geo {
    static constexpr int RT = 4 / QG;
    static constexpr int SLAB_ROWS = 32 * RT;
    static constexpr int SLAB_BYTES = SLAB_ROWS * 256;
    static constexpr int NS = QG == 4 ? 8 : 4;
    static constexpr int P = NS - 1;
    static constexpr int VM_PER_SLAB = 2 * RT + 1;  // per wave: 2*RT row DMAs + 1 norm DMA
    static constexpr int LDS_BYTES = NS * SLAB_BYTES + NS * 1024 + 16 + LCAP * 12;
};

template <int DT, int KSLABS, int QG>
__global__ __launch_bounds__(256, (QG == 1 || KSLABS > 4) ? 1 : 2) void k_scan(ScanK a) {
    using G = Geo<QG>;
    using A = Acc<DT>;
    constexpr int RT = G::RT, SLAB_ROWS = G::SLAB_ROWS, SLAB_BYTES = G::SLAB_BYTES, NS = G::NS, P = G::P;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *const ring = smem;
    uint8_t *const normring = smem + NS * SLAB_BYTES;  // [NS][4 waves][256 B]
    uint32_t *const st_cnt = (uint32_t *)(normring + NS * 1024);
    uint32_t *const st_row = st_cnt + 4;
    uint32_t *const st_key = st_row + LCAP;
    uint32_t *const st_q = st_key + LCAP;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qg = wave % QG, rt = wave / QG;
    const int j = lane & 31, h = lane >> 5;  // j: query (B operand / C column) and row (A operand)
    const int myq = qg * 32 + j;

    if (tid == 0) *st_cnt = 0;
    const uint32_t ring_lds = lds_addr(ring), norm_lds = lds_addr(normring);

    // tiles of this workgroup: (blockIdx.x + it*gridDim.x) * tile_step
    const uint32_t n_samp = (a.n_wgtiles + a.tile_step - 1) / a.tile_step;
    const int n_my = blockIdx.x < n_samp ? (int)((n_samp - blockIdx.x + gridDim.x - 1) / gridDim.x) : 0;

    float mins[16];
#pragma unroll
    for (int r = 0; r < 16; r++) mins[r] = __builtin_inff();

    if (n_my > 0) {
        // ---- query fragments: resident in registers for the whole kernel
        v4i qf[KSLABS * 8];
        {
            const uint8_t *qrow = a.qmat + (size_t)myq * a.stride;
#pragma unroll
            for (int x = 0; x < KSLABS * 8; x++) qf[x] = *(const v4i *)(qrow + (x >> 3) * 256 + ((x & 7) * 2 + h) * 16);
        }
        QInfo qi = a.qinfo[myq];
        float thr = a.mode ? a.thr[myq] : 0.f;
        // Pin every value loaded above as an asm operand: hipcc must retire its own loads
        // HERE (it cannot see the asm waits), otherwise it re-emits partial vmcnt waits for
        // them inside the main loop and throttles the DMA prefetch depth.
#pragma unroll
        for (int x = 0; x < KSLABS * 8; x++) asm volatile("" : "+v"(qf[x]));
        asm volatile("" : "+v"(qi.bb), "+v"(qi.dscale), "+v"(qi.eA), "+v"(qi.eR), "+v"(thr));
        wait_vm<0>();

        // ---- DMA issue state (runs P slabs ahead of the consumer)
        int i_tl = 0, i_ks = 0, i_slot = 0;
        auto issue = [&]() {
            const int tl = i_tl < n_my ? i_tl : n_my - 1;  // past the end: harmless re-read keeps vmcnt uniform
            const uint64_t row0 = (uint64_t)(blockIdx.x + (uint32_t)tl * gridDim.x) * a.tile_step * SLAB_ROWS;
            const uint32_t sl = __builtin_amdgcn_readfirstlane(ring_lds + i_slot * SLAB_BYTES);
#pragma unroll
            for (int e = 0; e < 2 * RT; e++) {
                const int bidx = wave * 2 * RT + e;  // 1 KiB block = 4 rows x 16 chunks
                const int r = 4 * bidx + (lane >> 4);
                const int c = (lane & 15) ^ (r & 15);
                const uint8_t *src = a.rows + (row0 + r) * a.stride + i_ks * 256 + c * 16;
                dma16(src, sl + bidx * 1024);
            }
            const float *nsrc = a.norm2 + row0 + rt * 32 + j;
            dma4(nsrc, __builtin_amdgcn_readfirstlane(norm_lds + i_slot * 1024 + wave * 256));
            if (++i_ks == KSLABS) {
                i_ks = 0;
                i_tl++;
            }
            if (++i_slot == NS) i_slot = 0;
        };
#pragma unroll
        for (int p = 0; p < P; p++) issue();

        int c_slot = 0;
        for (int tl = 0; tl < n_my; tl++) {
            typename A::type acc;
#pragma unroll
            for (int r = 0; r < 16; r++) acc[r] = 0;
            int norm_slot = 0;
#pragma unroll
            for (int ks = 0; ks < KSLABS; ks++) {
                if (!(a.debug & 4)) wait_vm<(P - 1) * G::VM_PER_SLAB>();  // this wave's share of slab (tl,ks) has landed
                wg_barrier();                          // ... and everyone else's; slab g-1 is fully consumed
                if (!(a.debug & 4)) issue();           // refill the slot slab g-1 occupied
                const uint8_t *sl = ring + c_slot * SLAB_BYTES + (rt * 32 + j) * 256;
                if (!(a.debug & 1)) {
#pragma unroll
                    for (int t = 0; t < 8; t++) {
                        const int c = (2 * t + h) ^ (j & 15);
                        const v4i af = *(const v4i *)(sl + c * 16);
                        acc = A::mfma(af, qf[ks * 8 + t], acc);
                    }
                }
                norm_slot = c_slot;
                if (++c_slot == NS) c_slot = 0;
            }

            if (a.debug & 2) {
                asm volatile("" ::"v"(acc[0]), "v"(acc[15]));
                continue;
            }
            // ---- epilogue: lane = one query, 16 rows: i(reg) = (reg&3) + 8*(reg>>2) + 4*h
            const float *nl = (const float *)(normring + norm_slot * 1024 + wave * 256);
            float nr[16];
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++) {
                const float4 v = *(const float4 *)(nl + 8 * g4 + 4 * h);
                nr[4 * g4 + 0] = v.x;
                nr[4 * g4 + 1] = v.y;
                nr[4 * g4 + 2] = v.z;
                nr[4 * g4 + 3] = v.w;
            }
            float key[16];
            uint32_t pass = 0;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const float dotf = A::tof(acc[r], qi.dscale);
                const float aa = nr[r];  // NaN for padding rows beyond n_rows: every compare below fails
                float err;
                if (a.metric == PVS_COSINE) {
                    key[r] = -dotf * __builtin_amdgcn_rsqf(aa);
                    err = qi.eA;
                } else {
                    key[r] = aa + (qi.bb - 2.0f * dotf);
                    err = qi.eA + qi.eR * aa;
                }
                if (a.mode == 0)
                    mins[r] = fminf(mins[r], key[r] + err);
                else if (key[r] - err <= thr)
                    pass |= 1u << r;
            }
            if (a.mode != 0) {
                if (__builtin_amdgcn_ballot_w64(pass != 0) != 0) {
                    const uint32_t row_base =
                        (uint32_t)((uint64_t)(blockIdx.x + (uint32_t)tl * gridDim.x) * a.tile_step * SLAB_ROWS) + rt * 32 + 4 * h;
                    bool direct = false;
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        if (pass & (1u << r)) {
                            const uint32_t row = row_base + (r & 3) + 8 * (r >> 2);
                            const uint32_t pos = atomicAdd(st_cnt, 1u);
                            if (pos < (uint32_t)LCAP) {
                                st_row[pos] = row;
                                st_key[pos] = __builtin_bit_cast(uint32_t, key[r]);
                                st_q[pos] = (uint32_t)myq;
                            } else {  // staging full (very loose threshold): go to HBM directly
                                const uint32_t gp = atomicAdd(&a.cand_cnt[myq], 1u);
                                if (gp < a.cand_cap)
                                    a.cand[(size_t)myq * a.cand_cap + gp] = make_uint2(row, __builtin_bit_cast(uint32_t, key[r]));
                                direct = true;
                            }
                        }
                    }
                    if (__builtin_amdgcn_ballot_w64(direct) != 0) wait_vm<0>();  // stores are unordered vs loads: drain
                }
                // flush decision: uniform because no wave appends between this barrier and the
                // next epilogue
                wg_barrier();
                const uint32_t staged = *(volatile uint32_t *)st_cnt;
                if (staged >= (uint32_t)FLUSH_AT) {
                    const uint32_t nst = staged < (uint32_t)LCAP ? staged : (uint32_t)LCAP;
                    for (uint32_t e = tid; e < nst; e += 256) {
                        const uint32_t q = st_q[e];
                        const uint32_t gp = atomicAdd(&a.cand_cnt[q], 1u);
                        if (gp < a.cand_cap) a.cand[(size_t)q * a.cand_cap + gp] = make_uint2(st_row[e], st_key[e]);
                    }
                    wg_barrier();
                    if (tid == 0) *st_cnt = 0;
                    wait_vm<0>();
                }
            }
        }
        wait_vm<0>();  // retire the dummy tail DMAs before LDS is reused / the wave exits
    }

    if (a.mode == 0) {
        float *o = a.gmin + (size_t)myq * a.groups_per_query + (size_t)((blockIdx.x * RT + rt) * 2 + h) * 16;
#pragma unroll
        for (int r = 0; r < 16; r++) o[r] = mins[r];
    } else {
        wg_barrier();
        const uint32_t staged = *(volatile uint32_t *)st_cnt;
        const uint32_t nst = staged < (uint32_t)LCAP ? staged : (uint32_t)LCAP;
        for (uint32_t e = tid; e < nst; e += 256) {
            const uint32_t q = st_q[e];
            const uint32_t gp = atomicAdd(&a.cand_cnt[q], 1u);
            if (gp < a.cand_cap) a.cand[(size_t)q * a.cand_cap + gp] = make_uint2(st_row[e], st_key[e]);
        }
    }
}

// ------------------------------------------------------------------ dispatch
bool pvs_scan_supported(int dtype, uint32_t kslabs) {
    if (dtype == PVS_I8) return kslabs >= 1 && kslabs <= 4;
    if (dtype == PVS_F16) return kslabs == 1 || kslabs == 2 || kslabs == 3 || kslabs == 4 || kslabs == 6 || kslabs == 8;
    return false;
}
uint32_t pvs_scan_wg_rows(uint32_t qgroups) { return 32u * (4u / qgroups); }

template <int DT, int KS, int QG>
static hipError_t launch_one(const ScanK &k, uint32_t grid, hipStream_t s) {
    static std::atomic<bool> configured{false};
    constexpr int lds = Geo<QG>::LDS_BYTES;
    if (!configured.load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void *)k_scan<DT, KS, QG>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        configured.store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((k_scan<DT, KS, QG>), dim3(grid), dim3(256), lds, s, k);
    return hipGetLastError();
}
template <int DT, int KS>
static hipError_t launch_qg(const ScanK &k, uint32_t qg, uint32_t grid, hipStream_t s) {
    switch (qg) {
        case 1: return launch_one<DT, KS, 1>(k, grid, s);
        case 2: return launch_one<DT, KS, 2>(k, grid, s);
        case 4: return launch_one<DT, KS, 4>(k, grid, s);
    }
    return hipErrorInvalidValue;
}

hipError_t pvs_launch_scan(const ScanArgs &a, hipStream_t s) {
    ScanK k;
    k.rows = a.rows;
    k.norm2 = a.norm2;
    k.qmat = a.qmat;
    k.qinfo = a.qinfo;
    k.thr = a.thr;
    k.gmin = a.gmin;
    k.cand_cnt = a.cand_cnt;
    k.cand = a.cand;
    k.n_rows = a.n_rows;
    k.stride = a.stride;
    const uint32_t wg_rows = pvs_scan_wg_rows(a.qgroups);
    k.n_wgtiles = (uint32_t)((a.n_rows + wg_rows - 1) / wg_rows);
    k.tile_step = a.tile_step ? a.tile_step : 1;
    k.groups_per_query = a.groups_per_query;
    k.cand_cap = a.cand_cap;
    k.metric = a.metric;
    k.mode = a.mode;
    static const int dbg = getenv("PVS_SCAN_DEBUG") ? atoi(getenv("PVS_SCAN_DEBUG")) : 0;
    k.debug = a.mode == 1 ? dbg : 0;
    if (a.dtype == PVS_I8) {
        switch (a.kslabs) {
            case 1: return launch_qg<PVS_I8, 1>(k, a.qgroups, a.grid, s);
            case 2: return launch_qg<PVS_I8, 2>(k, a.qgroups, a.grid, s);
            case 3: return launch_qg<PVS_I8, 3>(k, a.qgroups, a.grid, s);
            case 4: return launch_qg<PVS_I8, 4>(k, a.qgroups, a.grid, s);
        }
    } else if (a.dtype == PVS_F16) {
        switch (a.kslabs) {
            case 1: return launch_qg<PVS_F16, 1>(k, a.qgroups, a.grid, s);
            case 2: return launch_qg<PVS_F16, 2>(k, a.qgroups, a.grid, s);
            case 3: return launch_qg<PVS_F16, 3>(k, a.qgroups, a.grid, s);
            case 4: return launch_qg<PVS_F16, 4>(k, a.qgroups, a.grid, s);
            case 6: return launch_qg<PVS_F16, 6>(k, a.qgroups, a.grid, s);
            case 8: return launch_qg<PVS_F16, 8>(k, a.qgroups, a.grid, s);
        }
    }
    return hipErrorInvalidValue;
}

// ------------------------------------------------------------- k-th select
// 4-pass MSB radix select on order-preserving keys.  One workgroup per query.
template <typename KeyAt>
__device__ static inline uint32_t radix_kth(uint32_t n, uint32_t k, uint32_t *hist, uint32_t *s_sel, KeyAt key_at) {
    const int tid = threadIdx.x;
    uint32_t prefix = 0, mask = 0, kk = k;
    for (int pass = 0; pass < 4; pass++) {
        const int shift = 24 - 8 * pass;
        hist[tid] = 0;
        __syncthreads();
        for (uint32_t i = tid; i < n; i += 256) {
            const uint32_t key = key_at(i);
            if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t cum = 0, b = 0;
            for (; b < 256; b++) {
                if (cum + hist[b] >= kk) break;
                cum += hist[b];
            }
            s_sel[0] = b;          // 256 => fewer than kk keys match
            s_sel[1] = kk - cum;
        }
        __syncthreads();
        const uint32_t b = s_sel[0];
        if (b >= 256) return 0xffffffffu;
        kk = s_sel[1];
        prefix |= b << shift;
        mask |= 0xffu << shift;
        __syncthreads();
    }
    return prefix;
}

__global__ __launch_bounds__(256) void k_kth(const float *vals, uint32_t per_query, uint32_t batch, uint32_t k, float *out) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t s_sel[2];
    const uint32_t q = blockIdx.x;
    if (q >= batch) {  // padding query: never admits a candidate
        if (threadIdx.x == 0) out[q] = -__builtin_inff();
        return;
    }
    const float *v = vals + (size_t)q * per_query;
    uint32_t key = 0xffffffffu;
    if (per_query >= k) key = radix_kth(per_query, k, hist, s_sel, [&](uint32_t i) { return f32_sort_key(v[i]); });
    if (threadIdx.x == 0) {
        float t = f32_from_sort_key(key);
        out[q] = (t != t) ? __builtin_inff() : t;  // too few finite minima: admit everything
    }
}

hipError_t pvs_launch_kth(const float *vals, uint32_t per_query, uint32_t batch, uint32_t k, float *out, hipStream_t s) {
    const uint32_t bp = (batch + 31) / 32 * 32;
    hipLaunchKernelGGL(k_kth, dim3(bp), dim3(256), 0, s, vals, per_query, batch, k, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------- finalise
struct FinK {
    const uint8_t *rows;
    const float *norm2;
    const int64_t *ids;
    const void *qexact;
    const QInfo *qinfo;
    const uint32_t *cand_cnt;
    const uint2 *cand;
    int64_t *out_ids;
    float *out_dist;
    uint32_t *out_count, *need_dense;
    uint64_t n_rows;
    uint32_t stride, dim, cand_cap, k;
    int metric;
};

constexpr int FIN_LDS = PVS_CAND_CAP * 4 + PVS_SURV_CAP * 4 + 64;

__device__ static inline float cand_err(const FinK &a, const QInfo &qi, float aa) {
    return a.metric == PVS_COSINE ? qi.eA : qi.eA + qi.eR * aa;
}

template <int DT>
__global__ __launch_bounds__(256) void k_finalize(FinK a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t *const s_ub = (uint32_t *)smem;                        // [CAND_CAP] sort keys of upper bounds
    uint32_t *const s_surv = (uint32_t *)(smem + PVS_CAND_CAP * 4);  // [SURV_CAP] candidate slots
    uint32_t *const s_misc = s_surv + PVS_SURV_CAP;                  // [0]=survivor count, [2..3] select scratch
    unsigned long long *const s_sort = (unsigned long long *)smem;   // overlays s_ub after the select
    __shared__ uint32_t hist[256];

    const uint32_t q = blockIdx.x;
    const int tid = threadIdx.x;
    int64_t *oid = a.out_ids + (size_t)q * a.k;
    float *od = a.out_dist + (size_t)q * a.k;
    const uint32_t cnt = a.cand_cnt[q];
    const uint64_t want = a.k < a.n_rows ? a.k : a.n_rows;
    if (cnt > a.cand_cap || cnt < want) {  // overflowed, or NULL-distance rows are needed to fill the page
        if (tid == 0) {
            a.need_dense[q] = 1;
            a.out_count[q] = 0;
        }
        return;
    }
    const QInfo qi = a.qinfo[q];
    const uint2 *cand = a.cand + (size_t)q * a.cand_cap;
    for (uint32_t i = tid; i < cnt; i += 256) {
        const uint2 c = cand[i];
        const float key = __builtin_bit_cast(float, c.y);
        s_ub[i] = f32_sort_key(key + cand_err(a, qi, a.norm2[c.x]));
    }
    if (tid == 0) s_misc[0] = 0;
    __syncthreads();
    uint32_t kub = 0xffffffffu;
    if (cnt > a.k) kub = radix_kth(cnt, a.k, hist, s_misc + 2, [&](uint32_t i) { return s_ub[i]; });
    const float kappa = (kub == 0xffffffffu) ? __builtin_inff() : f32_from_sort_key(kub);
    __syncthreads();
    // survivors: lower bound <= k-th smallest upper bound
    for (uint32_t i = tid; i < cnt; i += 256) {
        const uint2 c = cand[i];
        const float key = __builtin_bit_cast(float, c.y);
        if (key - cand_err(a, qi, a.norm2[c.x]) <= kappa) {
            const uint32_t p = atomicAdd(&s_misc[0], 1u);
            if (p < PVS_SURV_CAP) s_surv[p] = c.x;
        }
    }
    __syncthreads();
    const uint32_t m = s_misc[0];
    if (m > PVS_SURV_CAP) {  // massive near-ties: let the dense path answer
        if (tid == 0) {
            a.need_dense[q] = 1;
            a.out_count[q] = 0;
        }
        return;
    }
    uint32_t m2 = 1;
    while (m2 < m) m2 <<= 1;
    __syncthreads();  // s_ub is dead from here: s_sort overlays it
    const uint8_t *qe = (const uint8_t *)a.qexact + (size_t)q * a.dim * (DT == PVS_I8 ? 1 : 4);
    for (uint32_t i = tid; i < m2; i += 256) {
        unsigned long long v = ~0ull;
        if (i < m) {
            const uint32_t row = s_surv[i];
            const float d = exact_distance<DT>(a.rows + (size_t)row * a.stride, qe, (int)a.dim, a.metric, a.norm2[row], qi.bb);
            v = ((unsigned long long)f32_sort_key(d) << 32) | row;
        }
        s_sort[i] = v;
    }
    __syncthreads();
    // bitonic sort ascending on (distance key, row)
    for (uint32_t sz = 2; sz <= m2; sz <<= 1) {
        for (uint32_t st = sz >> 1; st > 0; st >>= 1) {
            for (uint32_t i = tid; i < m2 / 2; i += 256) {
                const uint32_t lo = 2 * i - (i & (st - 1));
                const uint32_t hi = lo + st;
                const bool up = (lo & sz) == 0;
                const unsigned long long x = s_sort[lo], y = s_sort[hi];
                if ((x > y) == up) {
                    s_sort[lo] = y;
                    s_sort[hi] = x;
                }
            }
            __syncthreads();
        }
    }
    const uint32_t nout = m < a.k ? m : a.k;
    for (uint32_t i = tid; i < a.k; i += 256) {
        if (i < nout) {
            const unsigned long long v = s_sort[i];
            oid[i] = a.ids[(uint32_t)v];
            od[i] = f32_from_sort_key((uint32_t)(v >> 32));
        } else {
            oid[i] = -1;
            od[i] = __builtin_nanf("");
        }
    }
    if (tid == 0) {
        a.out_count[q] = nout;
        a.need_dense[q] = 0;
    }
}

hipError_t pvs_launch_finalize(const FinalizeArgs &f, hipStream_t s) {
    FinK k;
    k.rows = f.rows;
    k.norm2 = f.norm2;
    k.ids = f.ids;
    k.qexact = f.qexact;
    k.qinfo = f.qinfo;
    k.cand_cnt = f.cand_cnt;
    k.cand = f.cand;
    k.out_ids = f.out_ids;
    k.out_dist = f.out_dist;
    k.out_count = f.out_count;
    k.need_dense = f.need_dense;
    k.n_rows = f.n_rows;
    k.stride = f.stride;
    k.dim = f.dim;
    k.cand_cap = f.cand_cap;
    k.k = f.k;
    k.metric = f.metric;
    static std::atomic<bool> configured{false};
    if (!configured.load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void *)k_finalize<PVS_I8>, hipFuncAttributeMaxDynamicSharedMemorySize, FIN_LDS);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void *)k_finalize<PVS_F16>, hipFuncAttributeMaxDynamicSharedMemorySize, FIN_LDS);
        if (e != hipSuccess) return e;
        configured.store(true, std::memory_order_release);
    }
    if (f.dtype == PVS_I8)
        hipLaunchKernelGGL(k_finalize<PVS_I8>, dim3(f.batch), dim3(256), FIN_LDS, s, k);
    else if (f.dtype == PVS_F16)
        hipLaunchKernelGGL(k_finalize<PVS_F16>, dim3(f.batch), dim3(256), FIN_LDS, s, k);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

// ------------------------------------------------------------------- merge
// Rank-based merge of `world` sorted pages per query: the output slot of an entry is
// the number of entries (over all shards) that precede it under (NaN last, distance,
// id).  Shards hold disjoint id sets, so ranks are a permutation.
__device__ static inline bool page_less(float da, int64_t ia, float db, int64_t ib) {
    const uint32_t ka = f32_sort_key(da), kb = f32_sort_key(db);
    if (ka != kb) return ka < kb;
    return ia < ib;
}
__global__ __launch_bounds__(256) void k_merge(const int64_t *ids, const float *dist, const uint32_t *counts, uint32_t world,
                                               uint32_t batch, uint32_t k, int64_t *out_ids, float *out_dist, uint32_t *out_count) {
    const uint32_t q = blockIdx.x;
    uint32_t total = 0;
    for (uint32_t w = 0; w < world; w++) total += counts[(size_t)w * batch + q];
    const uint32_t nout = total < k ? total : k;
    for (uint32_t e = threadIdx.x; e < world * k; e += 256) {
        const uint32_t w = e / k, p = e % k;
        if (p >= counts[(size_t)w * batch + q]) continue;
        const size_t off = ((size_t)w * batch + q) * k;
        const float d = dist[off + p];
        const int64_t id = ids[off + p];
        uint32_t rank = p;
        for (uint32_t w2 = 0; w2 < world; w2++) {
            if (w2 == w) continue;
            const size_t o2 = ((size_t)w2 * batch + q) * k;
            uint32_t lo = 0, hi = counts[(size_t)w2 * batch + q];
            while (lo < hi) {  // first entry of shard w2 that does not precede (d, id)
                const uint32_t mid = (lo + hi) >> 1;
                if (page_less(dist[o2 + mid], ids[o2 + mid], d, id)) lo = mid + 1;
                else hi = mid;
            }
            rank += lo;
        }
        if (rank < k) {
            out_ids[(size_t)q * k + rank] = id;
            out_dist[(size_t)q * k + rank] = d;
        }
    }
    for (uint32_t i = nout + threadIdx.x; i < k; i += 256) {
        out_ids[(size_t)q * k + i] = -1;
        out_dist[(size_t)q * k + i] = __builtin_nanf("");
    }
    if (threadIdx.x == 0) out_count[q] = nout;
}
hipError_t pvs_launch_merge(const int64_t *ids, const float *dist, const uint32_t *counts, uint32_t world, uint32_t batch,
                            uint32_t k, int64_t *out_ids, float *out_dist, uint32_t *out_count, hipStream_t s) {
    hipLaunchKernelGGL(k_merge, dim3(batch), dim3(256), 0, s, ids, dist, counts, world, batch, k, out_ids, out_dist, out_count);
    return hipGetLastError();
}
