// pvs_scan_kernel.hpp — the hot kernel: a bandwidth-bound filter scan of the corpus on
// the CDNA4 matrix cores (gfx950).  Included by the pvs_scan_*.hip translation units,
// which instantiate it per (dtype, k-slabs, query groups, metric, mode).
//
// Replaces, for a batch of queries, the reference's per-row
//   vec_distance_{cosine,L2}(payload, ?)          (image_embeddings.rs:321-362,
//   ... ORDER BY order_rank ASC ... LIMIT k          text_embeddings.rs:386-418, builder.rs:578-582)
// The reference scores every row and sorts everything.  Here:
//   pass A (MODE 0)  scan a strided sample of row tiles, keep per-lane minima of an UPPER
//           bound of the key -> the k-th smallest of those group minima is a valid upper
//           bound T of the k-th best key of the whole corpus;
//   pass B (MODE 1)  scan every row once, emit (row, key) for rows whose LOWER bound <= T;
//   pass C  (pvs_kernels_scan.hip) exact rerank of the few survivors.
//
// Geometry (64-wide waves, 4 SIMDs/CU, 160 KiB LDS/CU):
//   workgroup = 4 waves; wave (qg, rt) owns query group qg (32 queries, held in VGPRs for
//   the whole kernel as MFMA B fragments) and row sub-tile rt (32 rows);
//   QG = batch_pad/32 in {1,2,4}, RT = 4/QG, workgroup tile = 32*RT rows.
//   The corpus streams HBM -> LDS by LDS-DMA (global_load_lds_dwordx4, no VGPR round trip)
//   in "slabs" of (32*RT rows x 256 B) grouped into chunks of SPB slabs (the whole 24 KiB tile for
//   768-B rows and 128 queries): NC-chunk ring, NC-1 chunks in flight, one s_barrier and one counted
//   s_waitcnt vmcnt per chunk (never 0 in steady state).  MODE 2 = dense exact int8 distances.
//   QG = 8 (256 queries, int8): 8 waves, one query group each, on one 32-row tile (Geo below).
//   A fragments are ds_read_b128 from an XOR-swizzled slab image (chunk ^= row & 15).  The corpus
//   is STORED in that image (tiled layout, pvs_common.hpp), so a DMA piece is one contiguous KiB
//   of HBM landing lane-linear in LDS; reads are conflict-free for ds_read_b128's 16-lane groups.
//   v_mfma_i32_32x32x32_i8 / v_mfma_f32_32x32x16_f16 (f32 rows: scaled per row and narrowed to f16 on the
//   way from LDS, see Acc<PVS_F32>) with A = 32 corpus rows, B = 32 queries:
//   each lane ends up with ONE query (lane & 31) and 16 rows, so the per-query threshold is
//   a lane-private register and the epilogue is 3-4 VALU per score until a row passes.
#pragma once
#include <cstdlib>

#include "pvs_lds_dma.hpp"
#include "pvs_scan_dispatch.hpp"

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float v2f __attribute__((ext_vector_type(2)));

constexpr int WCAP = 96;          // candidate staging entries per WAVE (wave-private LDS region)

template <int DT>
struct Acc;
template <>
struct Acc<PVS_I8> {
    using type = v16i;
    __device__ static inline type mfma(v4i a, v4i b, type c) { return __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0); }
    __device__ static inline int sum2(const type &a, const type &b, int r) { return a[r] + b[r]; }  // exact
};
template <>
struct Acc<PVS_F16> {
    using type = v16f;
    __device__ static inline type mfma(v4i a, v4i b, type c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, a), __builtin_bit_cast(v8h, b), c, 0, 0, 0);
    }
    __device__ static inline float sum2(const type &a, const type &b, int r) { return a[r] + b[r]; }
};

// f32 rows go to the matrix core as f16: on their way from LDS a lane scales its row by a power of two chosen from
// the row's own norm (so the largest component lands below 2^13 whatever the row's magnitude: no overflow, nothing
// above 2^-26 |a| is flushed) and narrows it with v_cvt_pkrtz_f16_f32 (round toward zero: 2^-10 relative).  The
// filter only needs an interval around the key — |dot error| <= (2^-10 + 2^-11 + 2^-21 + sqrt(D) 2^-26) |a||q| with
// the f16 image of the query, QInfo.eA/eR — and the survivors are rescored from the f32 rows in the reference's
// order.  (bf16 needs no scaling but is 5x coarser: 2^-7 for the pair.)
template <>
struct Acc<PVS_F32> {
    using type = v16f;
    __device__ static inline type mfma(v4i a, v4i b, type c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, a), __builtin_bit_cast(v8h, b), c, 0, 0, 0);
    }
    __device__ static inline float sum2(const type &a, const type &b, int r) { return a[r] + b[r]; }
};
typedef _Float16 v2h __attribute__((ext_vector_type(2)));
__device__ static inline int cvt_pkrtz_f16(float lo, float hi) {
    return __builtin_bit_cast(int, __builtin_amdgcn_cvt_pkrtz(lo, hi));
}
// power-of-two exponent a row is scaled by, from the per-row scalar the scan streams (cosine: 1/|a|, L2: |a|^2):
// |a| * 2^e lies in (2^11, 2^13].  The same function serves the A side (scale) and the C side (undo).
template <bool COSINE>
__device__ static inline int f32_row_exp(float aux) {
    const int x = __builtin_amdgcn_frexp_expf(aux);  // aux = m * 2^x, m in [0.5, 1)
    int e = COSINE ? 12 + x : 13 - ((x + 1) >> 1);
    e = e < -100 ? -100 : (e > 100 ? 100 : e);       // zero / non-finite norms: any value, those rows never pass
    return e;
}
// MFMA steps per 256-B row slab: 8 for 1- and 2-byte elements (32 B of k per step and half-wave),
// 4 for f32 (two 16-B pieces per step and half-wave)
template <int DT>
constexpr int steps_per_slab() { return DT == PVS_F32 ? 4 : 8; }

// Pipeline unit = "chunk" of SPB consecutive k-slabs of one workgroup tile: one counted
// vmcnt wait + one s_barrier per chunk.  For the headline shape (768-B rows, 128 queries) a
// chunk is the whole 24 KiB tile: 24 MFMAs run back to back between barriers.
// QG = 8 (256 queries per pass) runs EIGHT waves per workgroup, one query group each, all on the same
// 32-row tile: every stored byte is used by twice as many queries per HBM read.  Its 8 waves fill the
// CU's wave slots at this register budget (2 per SIMD), so there is one workgroup per CU and its ring
// takes most of the LDS.
template <int QG, int KSLABS>
struct Geo {
    static constexpr int WAVES = QG == 8 ? 8 : 4;
    static constexpr int RT = WAVES / QG;
    static constexpr int SLAB_ROWS = 32 * RT;
    static constexpr int SLAB_BYTES = SLAB_ROWS * 256;
    static constexpr int PPW = 8 * RT / WAVES;  // 1-KiB DMA pieces per wave and slab
    static constexpr int SPB = RT > 1 ? 1 : (KSLABS % 3 == 0 ? 3 : (KSLABS % 2 == 0 ? 2 : 1));
    static constexpr int NC_FIT = 150000 / (SPB * SLAB_BYTES + WAVES * 256);  // ring + row scalars beside ~9 KiB of staging
    static constexpr int NC_BIG = NC_FIT > 16 ? 16 : NC_FIT;
    static constexpr int NC = QG == 8 ? NC_BIG : (RT > 1 ? 4 : (SPB == 3 ? 3 : (SPB == 2 ? 4 : 8)));  // chunks in the ring
    static constexpr int NS = NC * SPB;                                            // slabs in the ring
    static constexpr int PC = NC - 1;                                              // chunks in flight
    static constexpr int CPT = KSLABS / SPB;                                       // chunks per tile
    static constexpr int VM_PER_CHUNK = PPW * SPB + 1;  // per wave: row DMAs + 1 norm DMA
    static constexpr int LCAP = WAVES * WCAP;           // candidate staging entries per workgroup
    static constexpr int LDS_BYTES = NS * SLAB_BYTES + NC * WAVES * 256 + 16 + LCAP * 12;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS per CU");
    static_assert((PC - 1) * VM_PER_CHUNK <= 63, "vmcnt is a 6-bit counter");
};

constexpr int scan_waves_per_simd(int QG, int KSLABS) { return (QG == 1 || KSLABS > 4) ? 1 : 2; }
constexpr int scan_threads(int QG) { return QG == 8 ? 512 : 256; }

template <int DT, int KSLABS, int QG, int METRIC, int MODE>
__global__ __launch_bounds__(scan_threads(QG), scan_waves_per_simd(QG, KSLABS)) void k_scan(ScanK a) {
    using G = Geo<QG, KSLABS>;
    using A = Acc<DT>;
    constexpr int RT = G::RT, SLAB_ROWS = G::SLAB_ROWS, SLAB_BYTES = G::SLAB_BYTES, NS = G::NS, NC = G::NC, PC = G::PC,
                  SPB = G::SPB, CPT = G::CPT, WAVES = G::WAVES, PPW = G::PPW, LCAP = G::LCAP;
    constexpr bool COS = METRIC == PVS_COSINE;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *const ring = smem;
    uint8_t *const normring = smem + NS * SLAB_BYTES;  // [NC][WAVES][256 B]
    uint32_t *const st_cnt = (uint32_t *)(normring + NC * WAVES * 256);
    uint32_t *const st_row = st_cnt + 4;
    uint32_t *const st_key = st_row + LCAP;
    uint32_t *const st_q = st_key + LCAP;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qg = wave % QG, rt = wave / QG;
    const int j = lane & 31, h = lane >> 5;  // j: query (B operand / C column) and row (A operand)
    const uint32_t sid = blockIdx.x, nstreams = a.grid;  // this workgroup's tile stream
    const int myq = qg * 32 + j;

    const uint32_t ring_lds = lds_addr(ring), norm_lds = lds_addr(normring);

    // tiles of this workgroup: (sid + it*nstreams) * tile_step
    const uint32_t n_samp = (a.n_wgtiles + a.tile_step - 1) / a.tile_step;
    const int n_my = sid < n_samp ? (int)((n_samp - sid + nstreams - 1) / nstreams) : 0;
    const uint64_t tile_bytes = (uint64_t)SLAB_ROWS * a.stride;

    float mins[MODE == 0 ? 16 : 1];
#pragma unroll
    for (int r = 0; r < (MODE == 0 ? 16 : 1); r++) mins[r] = __builtin_inff();

    if (n_my > 0) {
        // ---- query fragments: resident in registers for the whole kernel
        // (f32 index: the operand row holds the bf16 image of the query in its first half)
        constexpr int SPS = steps_per_slab<DT>();
        constexpr int NQF = KSLABS * SPS;
        v4i qf[NQF];
        {
            const uint8_t *qrow = a.qmat + (size_t)myq * a.stride;
#pragma unroll
            for (int x = 0; x < NQF; x++) qf[x] = *(const v4i *)(qrow + (x * 2 + h) * 16);
        }
        QInfo qi = a.qinfo[myq];
        float thr = MODE == 1 ? a.thr[myq] : 0.f;
        // Pin every value loaded above as an asm operand: hipcc must retire its own loads HERE
        // (it cannot see the asm waits), otherwise it re-emits partial vmcnt waits for them
        // inside the main loop and throttles the DMA prefetch depth.
#pragma unroll
        for (int x = 0; x < NQF; x++) asm volatile("" : "+v"(qf[x]));
        asm volatile("" : "+v"(qi.bb), "+v"(qi.dscale), "+v"(qi.eA), "+v"(qi.eR), "+v"(thr));
        wait_vm<0>();
        // Filter tests folded into one per-lane constant (key/err algebra of DESIGN.md §5):
        //   cosine  key = -dscale*acc/|a|, err = eA
        //           pass: key-err <= thr  <=>  acc/|a| >= -(thr+eA)/dscale
        //   L2      key = |a|^2 + bb - 2 dscale acc, err = eA + eR|a|^2
        //           pass: (1-eR)|a|^2 - 2 dscale acc <= thr + eA - bb
        const float c1 = 1.0f - qi.eR;
        const float m2d = -2.0f * qi.dscale;
        float tS;
        if (COS)
            tS = qi.dscale > 0.f ? -(thr + qi.eA) / qi.dscale : __builtin_inff();  // padding query: never passes
        else
            tS = qi.dscale > 0.f ? thr + qi.eA - qi.bb : -__builtin_inff();

        // ---- per-lane DMA source offsets inside a slab (row/chunk swizzle), computed once
        uint32_t voff[PPW];
#pragma unroll
        for (int e = 0; e < PPW; e++) {
            // piece = 4 slab rows x 256 B = one contiguous KiB of the tiled HBM layout, already swizzled
            const int r = 4 * (wave * PPW + e);                          // first slab row of the piece
            voff[e] = (uint32_t)(r >> 5) * (32u * a.stride) + (uint32_t)(r & 31) * 256u + (uint32_t)lane * 16u;
        }
        const uint32_t nvoff = (uint32_t)(rt * 32 + j) * 4u;
        // A-fragment LDS byte offsets of this lane inside a slab
        const uint32_t frag_row = (uint32_t)(rt * 32 + j) * 256u;
        const uint32_t jx = (uint32_t)(j & 15);
        // the 8 swizzled 16-B chunk positions this lane reads in every slab (k order): one base address per
        // position and chunk, the slab steps ride in the ds_read immediate offset
        uint32_t swz[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint32_t c = DT == PVS_F32 ? (uint32_t)(4 * (i >> 1) + 2 * h + (i & 1)) : (uint32_t)(2 * i + h);
            swz[i] = (c ^ jx) << 4;
        }

        // ---- DMA issue state (runs PC chunks ahead of the consumer)
        int i_tl = 0, i_ck = 0, i_slot = 0;  // tile, chunk within tile, ring chunk slot
        constexpr int DMA_PARTS = SPB * PPW + 1;  // row pieces + the per-row scalars
        const uint8_t *is_base = nullptr;
        const float *is_aux = nullptr;
        uint32_t is_lds = 0, is_norm = 0;
        auto issue_begin = [&]() {
            const int tl = i_tl < n_my ? i_tl : n_my - 1;  // past the end: harmless re-read keeps vmcnt uniform
            const uint64_t wt = (uint64_t)(sid + (uint32_t)tl * nstreams) * a.tile_step;
            is_base = a.rows + wt * tile_bytes + (uint32_t)i_ck * (SPB * 8192u);  // k-slab = 8 KiB per 32-row tile
            is_aux = a.aux + wt * SLAB_ROWS;
            if constexpr (DT == PVS_F32 || QG == 8) {
                // the 16-chunk unroll of the widest f32 instance makes hipcc lose track of the uniformity of
                // these two and hand VGPRs to the asm's SGPR operands; pin them scalar
                auto uni = [](const void *p) {
                    const uint64_t v = (uint64_t)(uintptr_t)p;
                    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
                    return (const void *)(uintptr_t)(((uint64_t)hi << 32) | lo);
                };
                is_base = (const uint8_t *)uni(is_base);
                is_aux = (const float *)uni(is_aux);
            }
            is_lds = ring_lds + (uint32_t)i_slot * (SPB * SLAB_BYTES) + (uint32_t)wave * (PPW * 1024);
            is_norm = norm_lds + (uint32_t)i_slot * (WAVES * 256) + (uint32_t)wave * 256;
            if (++i_ck == CPT) {
                i_ck = 0;
                i_tl++;
            }
            if (++i_slot == NC) i_slot = 0;
        };
        auto issue_part = [&](int part) {  // part is a compile-time constant at every call site
            if (part < DMA_PARTS - 1) {
                const int sb = part / PPW, e = part % PPW;
                dma16(is_base + sb * 8192, voff[e], is_lds + sb * SLAB_BYTES + e * 1024);
            } else {
                dma4(is_aux, nvoff, is_norm);
            }
        };
        auto issue = [&]() {
            issue_begin();
#pragma unroll
            for (int part = 0; part < DMA_PARTS; part++) issue_part(part);
        };
#pragma unroll
        for (int p = 0; p < PC; p++) issue();
        // ---- main loop, software-pipelined inside each wave: the MFMAs of tile t are issued while
        // the VALU works through the epilogue of tile t-1 (MFMA and VALU are separate pipes; a wave
        // issues in order, so the two instruction streams must sit in one basic block for the
        // scheduler to interleave them).  `hold` carries tile t-1's 16 dot products and `xh` its
        // per-row scalars across the iteration boundary.
        using acc_t = typename A::type;
        int c_slot = 0;
        float xh[16];
        decltype(A::sum2(acc_t{}, acc_t{}, 0)) hold[16];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            xh[r] = __builtin_nanf("");  // tile "-1": every test fails
            hold[r] = 0;
        }
        uint32_t prev_row_base = 0;
        bool prev_valid = false;  // MODE 2: the dummy tile "-1" writes nothing

        // Fast part of an epilogue, cut into 24 micro-steps so the main loop can drop them between
        // MFMAs:  m < 16: sv[m] = score of row m;  m >= 16: fold two scores into the running best.
        //   sv[r]: cosine  dot * (1/|a|)                 (pass iff sv >= tS)
        //          L2      (1-eR)*|a|^2 - 2*dscale*dot   (pass iff sv <= tS)
        // (two rows per step: v_pk_mul_f32 / v_pk_fma_f32)
        auto epi_micro = [&](int m, float(&sv)[16], float &best, auto &&pv) {
            if (m < 8) {
                v2f d = {(float)pv(2 * m), (float)pv(2 * m + 1)};
                if constexpr (DT == PVS_F32) {  // undo the per-row power-of-two scaling (exact)
                    d[0] = __builtin_ldexpf(d[0], -f32_row_exp<COS>(xh[2 * m]));
                    d[1] = __builtin_ldexpf(d[1], -f32_row_exp<COS>(xh[2 * m + 1]));
                }
                const v2f x = {xh[2 * m], xh[2 * m + 1]};
                const v2f r = COS ? d * x : __builtin_elementwise_fma(d, (v2f){m2d, m2d}, (v2f){c1, c1} * x);
                sv[2 * m] = r[0];
                sv[2 * m + 1] = r[1];
            } else {
                const int r = (m - 8) * 2;
                // NaN (padding / zero-norm rows) never wins a fmax/fmin
                best = COS ? fmaxf(best, fmaxf(sv[r], sv[r + 1])) : fminf(best, fminf(sv[r], sv[r + 1]));
            }
        };
        constexpr int EPI_STEPS = 16;
        // wave-private candidate staging: fill count (wave-uniform) and the flush to HBM.  The flush
        // issues global atomics/stores, which are unordered against the DMA loads the counted
        // vmcnt waits rely on, so it ends with a full drain of this wave's VMEM queue.
        uint32_t wcnt = 0;
        auto flush_wave = [&]() {
            for (uint32_t e = (uint32_t)lane; e < wcnt; e += 64) {
                const uint32_t slot = (uint32_t)wave * WCAP + e;
                const uint32_t q = st_q[slot];
                const uint32_t gp = atomicAdd(&a.cand_cnt[q], 1u);
                if (gp < a.cand_cap) a.cand[(size_t)q * a.cand_cap + gp] = make_uint2(st_row[slot], st_key[slot]);
            }
            wcnt = 0;
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        };
        // the rest: group minima (pass A) or candidate emission (pass B)
        auto epi_rest = [&](const float(&sv)[16], float best, auto &&pv) {
            if constexpr (MODE == 2) {
                // dense exact int8 distances (the reference's dist_{cte}.d for a batch of queries):
                // closed form of the exact integer sums, valid while they stay below 2^24
                // (oracle: orc_i8_cosine_from_sums / orc_i8_l2_from_sums).  xh = |a|^2 here.
                if (myq < (int)a.batch) {
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const uint32_t row = prev_row_base + (r & 3) + 8 * (r >> 2);
                        if (row < a.n_rows && prev_valid) {
                            float d;
                            if (COS) {
                                d = ref_cosine_finish((float)pv(r), xh[r], qi.bb);
                            } else {
                                const double ss = (double)xh[r] + (double)qi.bb - 2.0 * (double)pv(r);
                                if (!(ss < 16777216.0)) atomicOr(a.dense_flag, 1u);
                                d = ref_l2_finish((float)ss);
                            }
                            a.dense_out[(size_t)row * a.dense_ld + myq] = d;
                        }
                    }
                }
            } else if (MODE == 0) {
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    // upper bound of this row's key: key + err
                    const float ub = COS ? __builtin_fmaf(-sv[r], qi.dscale, qi.eA) : sv[r] + (qi.bb + qi.eA) + 2.0f * qi.eR * xh[r];
                    mins[r] = fminf(mins[r], ub);
                }
            } else {
                const bool lane_pass = COS ? (best >= tS) : (best <= tS);
                if (__builtin_amdgcn_ballot_w64(lane_pass) != 0) {
                    // Rare path (some lane of the wave has a passing row).  Staging is wave-private and its
                    // fill count lives in a scalar register: no LDS atomics, no barrier.  One ballot per
                    // tile row r (compile-time r: no dynamic register indexing, no select chains); rows
                    // nobody passes cost one compare and one not-taken scalar branch.
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const bool p = COS ? (sv[r] >= tS) : (sv[r] <= tS);
                        const unsigned long long bal = __builtin_amdgcn_ballot_w64(p);
                        if (bal != 0) {
                            const uint32_t npass = (uint32_t)__builtin_popcountll(bal);
                            if (__builtin_expect(wcnt + npass > (uint32_t)WCAP, 0)) flush_wave();
                            uint32_t payload;
                            if constexpr (DT == PVS_I8)
                                payload = (uint32_t)pv(r);  // exact integer dot
                            else
                                payload = __builtin_bit_cast(uint32_t, COS ? -sv[r] * qi.dscale : sv[r] + qi.bb + qi.eR * xh[r]);
                            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                            if (p) {
                                const uint32_t slot = (uint32_t)wave * WCAP + wcnt + rank;
                                st_row[slot] = prev_row_base + (uint32_t)((r & 3) + 8 * (r >> 2));
                                st_key[slot] = payload;
                                st_q[slot] = (uint32_t)myq;
                            }
                            wcnt += npass;
                        }
                    }
                }
            }
        };

        // One tile: MFMA burst into (acc, acc1) with the previous tile's epilogue slices in its shadow; `pv(r)` = the
        // previous tile's r-th dot product.  PARITY (the 8-wave geometry): accumulators alternate between two
        // register sets, the previous tile's sums are read where the matrix core left them — no hand-off copy or
        // add, and one accumulation chain is enough because the SIMD's second wave fills the dependent-issue gap.
        constexpr bool PARITY = QG == 8;
        auto run_tile = [&](int tl, acc_t &acc, acc_t &acc1, auto &&pv) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                acc[r] = 0;
                if (!PARITY) acc1[r] = 0;
            }
            int norm_slot = 0;
            float sv[16];
            float best = COS ? -__builtin_inff() : __builtin_inff();
#pragma unroll
            for (int ck = 0; ck < CPT; ck++) {
                wait_vm<(PC - 1) * G::VM_PER_CHUNK>();  // this wave's share of the chunk has landed
                wg_barrier();                           // ... and everyone else's; the previous chunk is consumed
                issue_begin();                          // the slot the previous chunk occupied is refilled below
                const uint8_t *cb = ring + c_slot * (SPB * SLAB_BYTES) + frag_row;
                float row_scale = 1.0f;  // f32 rows: 2^e of this lane's A row (its scalar sits in this chunk's slot)
                if constexpr (DT == PVS_F32) {
                    const float ax = ((const float *)(normring + c_slot * (WAVES * 256) + wave * 256))[lane];
                    row_scale = __builtin_ldexpf(1.0f, f32_row_exp<COS>(ax));
                }
                (void)row_scale;
                const uint8_t *fb[8];
#pragma unroll
                for (int i = 0; i < 8; i++) fb[i] = cb + swz[i];
                // Explicit software pipeline, fenced with sched_barrier(0) so hipcc keeps the order:
                //   step t:  LDS read of fragment t+PF | MFMA t | one DMA piece of the chunk PC ahead |
                //            a slice of the previous tile's epilogue
                // A wave issues in order, so only VALU placed BETWEEN MFMAs runs in their shadow.
                constexpr int NF = SPB * SPS, PF = 4;
                v4i af[NF];
                v4i raw[DT == PVS_F32 ? NF : 1][2];  // f32: the two 16-B pieces of a step, before narrowing
                (void)raw;
                auto frag = [&](int t) {
                    if constexpr (DT == PVS_F32) {
                        raw[t][0] = *(const v4i *)(fb[2 * (t & 3)] + (t >> 2) * SLAB_BYTES);
                        raw[t][1] = *(const v4i *)(fb[2 * (t & 3) + 1] + (t >> 2) * SLAB_BYTES);
                    } else {
                        af[t] = *(const v4i *)(fb[t & 7] + (t >> 3) * SLAB_BYTES);
                    }
                };
                auto narrow = [&](int t) {  // VALU work placed behind MFMA t-1
                    if constexpr (DT == PVS_F32) {
                        // scale (v_pk_mul_f32 on register pairs) and narrow, two components at a time
                        typedef float v4ff __attribute__((ext_vector_type(4)));
                        const v4ff r0 = __builtin_bit_cast(v4ff, raw[t][0]), r1 = __builtin_bit_cast(v4ff, raw[t][1]);
                        const v2f s2 = {row_scale, row_scale};
                        const v2f p0 = __builtin_shufflevector(r0, r0, 0, 1) * s2, p1 = __builtin_shufflevector(r0, r0, 2, 3) * s2;
                        const v2f p2 = __builtin_shufflevector(r1, r1, 0, 1) * s2, p3 = __builtin_shufflevector(r1, r1, 2, 3) * s2;
                        af[t][0] = cvt_pkrtz_f16(p0[0], p0[1]);
                        af[t][1] = cvt_pkrtz_f16(p1[0], p1[1]);
                        af[t][2] = cvt_pkrtz_f16(p2[0], p2[1]);
                        af[t][3] = cvt_pkrtz_f16(p3[0], p3[1]);
                    }
                };
#pragma unroll
                for (int t = 0; t < PF && t < NF; t++) frag(t);
                narrow(0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < NF; t++) {
                    if (t + PF < NF) frag(t + PF);
                    if (!PARITY && (t & 1))
                        acc1 = A::mfma(af[t], qf[ck * NF + t], acc1);
                    else
                        acc = A::mfma(af[t], qf[ck * NF + t], acc);
                    if (t + 1 < NF) narrow(t + 1);
#pragma unroll
                    for (int part = t * DMA_PARTS / NF; part < (t + 1) * DMA_PARTS / NF; part++) issue_part(part);
                    if (ck == 0) {
#pragma unroll
                        for (int m = t * EPI_STEPS / NF; m < (t + 1) * EPI_STEPS / NF; m++) epi_micro(m, sv, best, pv);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                norm_slot = c_slot;
                if (++c_slot == NC) c_slot = 0;
            }
            epi_rest(sv, best, pv);
            // ---- hand this tile's results to the next iteration (its row scalars leave LDS now: the
            // slot is refilled by the DMA issued after the next barrier)
            {
                const float *nl = (const float *)(normring + norm_slot * (WAVES * 256) + wave * 256);
#pragma unroll
                for (int g4 = 0; g4 < 4; g4++) {
                    const float4 v = *(const float4 *)(nl + 8 * g4 + 4 * h);
                    xh[4 * g4 + 0] = v.x;
                    xh[4 * g4 + 1] = v.y;
                    xh[4 * g4 + 2] = v.z;
                    xh[4 * g4 + 3] = v.w;
                }
                if constexpr (!PARITY) {
#pragma unroll
                    for (int r = 0; r < 16; r++) hold[r] = A::sum2(acc, acc1, r);  // i8: exact integers; floats: within the error budget
                }
                prev_row_base = (uint32_t)((sid + (uint32_t)tl * nstreams) * a.tile_step * SLAB_ROWS) + rt * 32 + 4 * h;
                prev_valid = true;
            }
        };
        auto drain = [&](auto &&pv) {  // the last tile's epilogue
            float sv[16];
            float best = COS ? -__builtin_inff() : __builtin_inff();
#pragma unroll
            for (int m = 0; m < EPI_STEPS; m++) epi_micro(m, sv, best, pv);
            epi_rest(sv, best, pv);
        };
        if constexpr (PARITY) {
            acc_t accA, accB, unused;
#pragma unroll
            for (int r = 0; r < 16; r++) accB[r] = 0;  // tile "-1"
            auto pa = [&](int r) { return accA[r]; };
            auto pb = [&](int r) { return accB[r]; };
            int tl = 0;
            for (; tl + 1 < n_my; tl += 2) {
                run_tile(tl, accA, unused, pb);
                run_tile(tl + 1, accB, unused, pa);
            }
            if (tl < n_my) {
                run_tile(tl, accA, unused, pb);
                drain(pa);
            } else {
                drain(pb);
            }
        } else {
            auto ph = [&](int r) { return hold[r]; };
            for (int tl = 0; tl < n_my; tl++) {
                acc_t acc, acc1;
                run_tile(tl, acc, acc1, ph);
            }
            drain(ph);
        }
        if (MODE == 1) flush_wave();
        wait_vm<0>();  // retire the dummy tail DMAs before LDS is reused / the wave exits
    }

    if (MODE == 2) {
    } else if (MODE == 0) {
        // Each lane holds 16 minima (one per accumulator row slot) = 16 disjoint row groups of its query.
        // The threshold only needs a few times k groups per query; folding to a.gmin_per_lane (a power of two)
        // keeps the k-th select that follows short.
        const uint32_t gr = a.gmin_per_lane;
#pragma unroll
        for (int sft = 8; sft >= 1; sft >>= 1)
            if (gr <= (uint32_t)sft) {
#pragma unroll
                for (int r = 0; r < sft; r++) mins[r] = fminf(mins[r], mins[r + sft]);
            }
        float *o = a.gmin + (size_t)myq * a.groups_per_query + (size_t)((blockIdx.x * RT + rt) * 2 + h) * gr;
#pragma unroll
        for (int r = 0; r < 16; r++)
            if ((uint32_t)r < gr) o[r] = mins[r];
    }
}

// ---- per-TU dispatch helpers
template <int DT, int KS, int QG, int METRIC, int MODE>
static hipError_t scan_launch_one(const ScanK &k, hipStream_t s) {
    static std::atomic<bool> configured{false};
    constexpr int lds = Geo<QG, KS>::LDS_BYTES;
    if (!configured.load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void *)k_scan<DT, KS, QG, METRIC, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        configured.store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((k_scan<DT, KS, QG, METRIC, MODE>), dim3(k.grid), dim3(Geo<QG, KS>::WAVES * 64), lds, s, k);
    return hipGetLastError();
}
template <int DT, int KS, int QG>
static hipError_t scan_launch_mm(const ScanK &k, int metric, int mode, hipStream_t s) {
    if constexpr (DT == PVS_I8) {
        if (mode == 2)
            return metric == PVS_COSINE ? scan_launch_one<DT, KS, QG, PVS_COSINE, 2>(k, s) : scan_launch_one<DT, KS, QG, PVS_L2, 2>(k, s);
    } else {
        if (mode == 2) return hipErrorInvalidValue;  // float order matters: no closed form
    }
    if (metric == PVS_COSINE) return mode == 0 ? scan_launch_one<DT, KS, QG, PVS_COSINE, 0>(k, s) : scan_launch_one<DT, KS, QG, PVS_COSINE, 1>(k, s);
    return mode == 0 ? scan_launch_one<DT, KS, QG, PVS_L2, 0>(k, s) : scan_launch_one<DT, KS, QG, PVS_L2, 1>(k, s);
}
// 256 queries per pass: 8 waves x 32 queries (filter passes only)
template <int DT, int KS>
static hipError_t scan_launch_wide(const ScanK &k, int metric, int mode, hipStream_t s) {
    if (mode != 0 && mode != 1) return hipErrorInvalidValue;
    if (metric == PVS_COSINE) return mode == 0 ? scan_launch_one<DT, KS, 8, PVS_COSINE, 0>(k, s) : scan_launch_one<DT, KS, 8, PVS_COSINE, 1>(k, s);
    return mode == 0 ? scan_launch_one<DT, KS, 8, PVS_L2, 0>(k, s) : scan_launch_one<DT, KS, 8, PVS_L2, 1>(k, s);
}
template <int DT, int KS>
static hipError_t scan_launch_qg(const ScanK &k, uint32_t qg, int metric, int mode, hipStream_t s) {
    switch (qg) {
        case 1: return scan_launch_mm<DT, KS, 1>(k, metric, mode, s);
        case 2: return scan_launch_mm<DT, KS, 2>(k, metric, mode, s);
        case 4: return scan_launch_mm<DT, KS, 4>(k, metric, mode, s);
    }
    return hipErrorInvalidValue;
}
