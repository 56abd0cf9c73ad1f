// pvs_scan_kernel.hpp — the hot kernel: a bandwidth-bound filter scan of the corpus on
// the CDNA4 matrix cores (gfx950).  Included by the pvs_scan_*.hip translation units,
// which instantiate it per (dtype, k-slabs, query groups, metric, mode).
//
// Replaces, for a batch of queries, the reference's per-row
//   vec_distance_{cosine,L2}(payload, ?)          (image_embeddings.rs:321-362,
//   ... ORDER BY order_rank ASC ... LIMIT k          text_embeddings.rs:386-418, builder.rs:578-582)
// The reference scores every row and sorts everything.  Here (HISTORY.md §4.1-4.2):
//   pass A (MODE 0)  scan a strided sample of row tiles, keep per-lane minima of an UPPER
//           bound of the key -> the k-th smallest of those group minima is a valid upper
//           bound T of the k-th best key of the whole corpus;
//   pass B (MODE 1)  scan every row once, emit (row, key) for rows whose LOWER bound <= T;
//   pass C  (pvs_kernels_scan.hip) exact rerank of the few survivors.
//
// Geometry (64-wide waves, 4 SIMDs/CU, 160 KiB LDS/CU):
//   workgroup = 4 waves; wave (qw, rt) owns one query group of 32 queries (held in VGPRs for the whole kernel as MFMA B
//   fragments) and row sub-tile rt (32 rows); QG = batch_pad/32 in {1,2,4}; RT = 4/QG, tile = 32*RT rows.
//   The corpus streams HBM -> LDS by LDS-DMA (global_load_lds_dwordx4, no VGPR round trip)
//   in "slabs" of (32*RT rows x 256 B) grouped into chunks of SPB slabs (the whole 24 KiB tile for
//   768-B rows and >= 128 queries): NC-chunk ring, NC-1 chunks in flight, one s_barrier and one counted
//   s_waitcnt vmcnt per chunk (never 0 in steady state).  MODE 2 = dense exact int8 distances.
//   A fragments are ds_read_b128 from an XOR-swizzled slab image (chunk ^= row & 15).  The corpus
//   is STORED in that image (tiled layout, pvs_common.hpp), so a DMA piece is one contiguous KiB
//   of HBM landing lane-linear in LDS; reads are conflict-free for ds_read_b128's 16-lane groups.
//   v_mfma_i32_32x32x32_i8 / v_mfma_f32_32x32x16_f16 (f32 rows: scaled per row and narrowed to f16 on the
//   way from LDS, see Acc<PVS_F32>) with A = 32 corpus rows, B = 32 queries:
//   each lane ends up with ONE query (lane & 31) and 16 rows, so the per-query threshold is a lane-private register.
//
// Pass-B epilogue (MODE 1, int8 and f16 rows): "could any of this lane's 16 rows pass?" is answered without
// touching the per-row scalars: fold the 16 accumulators with v_max3 (8 VALU) and compare with a per-(query, tile)
// bound derived from the tile's extreme row scalars (k_scan_aux: min/max of |a| resp. |a|^2 over the 32 rows, built at
// add time, streamed behind the row scalars) — a NECESSARY condition for the exact per-row test, so the emitted
// candidate set is exactly what the per-row test alone would emit.  Only wave-tiles where some lane passes run the
// per-row test, with the row scalars read back from LDS (their ring keeps a tile's scalars one tile longer than its rows).
//
// What the measurements say (10M x 768 int8 on MI355X; HISTORY.md §4.1b, §5):
//   * 128 queries: 1.25-1.31 ms = 5.9-6.1 TB/s; a pure read stream (plain loads or LDS-DMA, pvs_microbench) reaches 7.1 TB/s on the
//     box and the 32-query instance 6.7.  The part runs this pass at its 1,400 W limit with the clock lowered: ring depth, prefetch
//     depth, DMA placement, workgroups per CU and a barrier-free rewrite all measured the same (profiles/r02_tile_phase_profile_b256.txt),
//     and so did the 16x16x64 instruction shape that pays at 256 queries (profiles/r03_wide_ablation.md).
//   * 256 queries (int8, row pitch <= 1 KiB) run on k_scan_wide (pvs_scan_wide.hpp), not here.
#pragma once
#include <cstdlib>
#include <type_traits>

#include "pvs_lds_dma.hpp"
#include "pvs_scan_dispatch.hpp"

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float v2f __attribute__((ext_vector_type(2)));


template <int DT>
struct Acc;
template <>
struct Acc<PVS_I8> {
    using type = v16i;
    using elem = int;
    __device__ static inline type mfma(v4i a, v4i b, type c) { return __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0); }
    __device__ static inline int sum2(const type &a, const type &b, int r) { return a[r] + b[r]; }  // exact
    __device__ static inline int lowest() { return (int)0x80000000; }
    __device__ static inline int max3(int a, int b, int c) { return max(a, max(b, c)); }  // v_max3_i32
};
template <>
struct Acc<PVS_F16> {
    using type = v16f;
    using elem = float;
    __device__ static inline type mfma(v4i a, v4i b, type c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, a), __builtin_bit_cast(v8h, b), c, 0, 0, 0);
    }
    __device__ static inline float sum2(const type &a, const type &b, int r) { return a[r] + b[r]; }
    __device__ static inline float lowest() { return -__builtin_inff(); }
    __device__ static inline float max3(float a, float b, float c) { return fmaxf(a, fmaxf(b, c)); }  // v_max3_f32 (drops NaN)
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
    using elem = float;
    __device__ static inline type mfma(v4i a, v4i b, type c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, a), __builtin_bit_cast(v8h, b), c, 0, 0, 0);
    }
    __device__ static inline float sum2(const type &a, const type &b, int r) { return a[r] + b[r]; }
    __device__ static inline float lowest() { return -__builtin_inff(); }
    __device__ static inline float max3(float a, float b, float c) { return fmaxf(a, fmaxf(b, c)); }
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
// MODE 2 at 32 queries (QG == 1: four 32-row sub-tiles per workgroup) runs with a TWO-chunk ring instead of four: 67 KiB of LDS, so
// two workgroups share a CU and one wave's epilogue — the closed-form distances in f64 and, for the per-item search, the sequential
// per-group fold: thousands of cycles of dependent VALU work per tile — runs while the other workgroup's waves issue MFMAs and
// wait for their DMA.  (One wave per SIMD paid that epilogue in full: 0.94 ms for 4M x 768 x 32 AVG against 0.46 without it.)
constexpr bool scan_fold2(int QG, int KSLABS, int MODE) { return (MODE == 2 || MODE == 3) && QG == 1 && KSLABS <= 3; }
template <int QG, int KSLABS, int MODE = 0>
struct Geo {
    static constexpr int WAVES = 4;
    static constexpr int RT = WAVES / QG;        // row sub-tiles per workgroup
    static constexpr int SLAB_ROWS = 32 * RT;
    static constexpr int SLAB_BYTES = SLAB_ROWS * 256;
    static constexpr int PPW = 8 * RT / WAVES;   // 1-KiB DMA pieces per wave and slab
    static constexpr int SPB = RT > 1 ? 1 : (KSLABS % 3 == 0 ? 3 : (KSLABS % 2 == 0 ? 2 : 1));  // k-slabs per chunk
    static constexpr int CPT = KSLABS / SPB;                                       // chunks per tile
    // chunks in the ring (measured at 128 queries x 768 B: 2-chunk rings with 2 or 3 workgroups per CU 1.295 / 1.398 ms against 1.28;
    // one k-slab per chunk with 6 / 8 / 9 chunks 1.35)
    static constexpr int NC = scan_fold2(QG, KSLABS, MODE) ? 2 : RT > 1 ? 4 : (SPB == 3 ? 3 : (SPB == 2 ? 4 : 8));
    static constexpr int NS = NC * SPB;                                            // slabs in the ring
    static constexpr int PC = NC - 1;                                              // chunks in flight
    static constexpr int NCN = 2 + (PC + CPT - 1) / CPT;  // row-scalar ring slots, one per TILE (every chunk of a tile re-lands the
                                                          // same record): tiles in flight + the previous tile, kept for its epilogue
    static constexpr int VM_PER_CHUNK = PPW * SPB + 1;  // per wave: row DMAs + 1 row-scalar DMA
    static constexpr int LDS_BYTES = NS * SLAB_BYTES + NCN * WAVES * 256;  // ring, row-scalar records
    static_assert(QG == 1 || QG == 2 || QG == 4, "32, 64 or 128 queries per pass");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS per CU");
    static_assert((PC - 1) * VM_PER_CHUNK <= 63, "vmcnt is a 6-bit counter");
};

// (MODE 3 — the per-group fold — holds 32 distances and the f64 sum state on top of the query fragments: one wave per SIMD beyond 768-B rows)
constexpr int scan_waves_per_simd(int QG, int KSLABS, int MODE) {
    return scan_fold2(QG, KSLABS, MODE) ? 2 : (QG == 1 || KSLABS > 4 || (MODE == 3 && KSLABS > 3) || MODE == 5) ? 1 : 2;  // (MODE 5 holds 64 bracket ends per lane on top of the query fragments)
}

template <int DT, int KSLABS, int QG, int METRIC, int MODE>
__global__ __launch_bounds__(256, scan_waves_per_simd(QG, KSLABS, MODE)) void k_scan(ScanK a) {
    using G = Geo<QG, KSLABS, MODE>;
    using A = Acc<DT>;
    using elem_t = typename A::elem;
    constexpr int RT = G::RT, SLAB_ROWS = G::SLAB_ROWS, SLAB_BYTES = G::SLAB_BYTES, NS = G::NS, NC = G::NC, PC = G::PC,
                  SPB = G::SPB, CPT = G::CPT, WAVES = G::WAVES, PPW = G::PPW, NCN = G::NCN;
    constexpr int GPW = 1, QW = QG;  // one query group per wave, QG waves side by side along the queries
    constexpr bool COS = METRIC == PVS_COSINE;
    constexpr bool PRETEST = MODE == 1 && DT != PVS_F32;  // (f32 rows carry a per-row power-of-two scale in their sums)
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *const ring = smem;
    uint8_t *const normring = smem + NS * SLAB_BYTES;  // [NCN][WAVES][256 B]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qw = wave % QW, rt = wave / QW;
    const int j = lane & 31, h = lane >> 5;  // j: query (B operand / C column) and row (A operand)
    const uint32_t sid = blockIdx.x, qoff = 0;  // this workgroup's tile stream, and its first query in the batch
    const uint32_t nstreams = a.grid;
    int myq[GPW];
#pragma unroll
    for (int g = 0; g < GPW; g++) myq[g] = (int)qoff + (qw * GPW + g) * 32 + j;

    const uint32_t ring_lds = lds_addr(ring), norm_lds = lds_addr(normring);

    // tiles of this workgroup: (sid + it*nstreams) * tile_step
    const uint32_t n_samp = (a.n_wgtiles + a.tile_step - 1) / a.tile_step;
    const int n_my = (sid < n_samp && sid < nstreams) ? (int)((n_samp - sid + nstreams - 1) / nstreams) : 0;
    const uint64_t tile_bytes = (uint64_t)SLAB_ROWS * a.stride;

    float mins[MODE == 0 ? GPW : 1][MODE == 0 ? 16 : 1];
#pragma unroll
    for (int g = 0; g < (MODE == 0 ? GPW : 1); g++)
#pragma unroll
        for (int r = 0; r < (MODE == 0 ? 16 : 1); r++) mins[g][r] = __builtin_inff();

    // MODE 5: minima of the files' upper bounds (lanes of half 1) / lower bounds (half 0) this lane has seen, over FOLD_SLOTS disjoint sets
    // of files (the files that end in this wave's tiles number s, s + FOLD_SLOTS, ...):
    // its share of the per-query buckets the threshold select reads — registers, written once at the end of the kernel
    constexpr int FOLD_SLOTS = 8;
    float xmin[MODE == 5 ? FOLD_SLOTS : 1];
#pragma unroll
    for (int i = 0; i < (MODE == 5 ? FOLD_SLOTS : 1); i++) xmin[i] = __builtin_inff();
    // Candidate emission (MODE 1).  Every (workgroup row stream, half-wave, query) triple owns a SEGMENT of a.seg_cap slots in
    // HBM, written by exactly one LANE (lane (j, h) holds query column j and the rows of half h), so the fill count is a
    // register of that lane: no staging list, no flush, no atomic of any kind.  A segment that overflows is reported through
    // its count (pass C then hands the query to the dense path).
    const uint32_t seg = (sid * RT + rt) * 2 + h;  // this lane's segment index (the QW waves side by side hold different queries)
    uint32_t mycnt[GPW];
#pragma unroll
    for (int g = 0; g < GPW; g++) mycnt[g] = 0;
    if (n_my > 0) {
        // ---- query fragments: resident in registers for the whole kernel
        constexpr int SPS = steps_per_slab<DT>();
        constexpr int NQF = KSLABS * SPS;
        v4i qf[GPW][NQF];
#pragma unroll
        for (int g = 0; g < GPW; g++) {
            const uint8_t *qrow = a.qmat + (size_t)myq[g] * a.stride;
#pragma unroll
            for (int x = 0; x < NQF; x++) qf[g][x] = *(const v4i *)(qrow + (x * 2 + h) * 16);
        }
        QInfo qi[GPW];
        float thr[GPW];
#pragma unroll
        for (int g = 0; g < GPW; g++) {
            qi[g] = a.qinfo[myq[g]];
            thr[g] = MODE == 1 ? a.thr[myq[g]] : 0.f;
        }
        // Pin every value loaded above as an asm operand: hipcc must retire its own loads HERE
        // (it cannot see the asm waits), otherwise it re-emits partial vmcnt waits for them
        // inside the main loop and throttles the DMA prefetch depth.
#pragma unroll
        for (int g = 0; g < GPW; g++) {
#pragma unroll
            for (int x = 0; x < NQF; x++) asm volatile("" : "+v"(qf[g][x]));
            asm volatile("" : "+v"(qi[g].bb), "+v"(qi[g].dscale), "+v"(qi[g].eA), "+v"(qi[g].eR), "+v"(thr[g]));
        }
        wait_vm<0>();
        // Filter tests folded into one per-lane constant (key/err algebra of HISTORY.md §4.2):
        //   cosine  key = -dscale*acc/|a|, err = eA
        //           pass: key-err <= thr  <=>  acc/|a| >= -(thr+eA)/dscale
        //   L2      key = |a|^2 + bb - 2 dscale acc, err = eA + eR|a|^2
        //           pass: (1-eR)|a|^2 - 2 dscale acc <= thr + eA - bb
        float c1[GPW], m2d[GPW], tS[GPW], hd[GPW];
#pragma unroll
        for (int g = 0; g < GPW; g++) {
            c1[g] = 1.0f - qi[g].eR;
            m2d[g] = -2.0f * qi[g].dscale;
            hd[g] = qi[g].dscale > 0.f ? 0.5f / qi[g].dscale : 0.f;
            if (COS)
                tS[g] = qi[g].dscale > 0.f ? -(thr[g] + qi[g].eA) / qi[g].dscale : __builtin_inff();  // padding query: never passes
            else
                tS[g] = qi[g].dscale > 0.f ? thr[g] + qi[g].eA - qi[g].bb : -__builtin_inff();
        }

        // ---- per-lane DMA source offsets inside a slab (row/chunk swizzle), computed once
        uint32_t voff[PPW];
#pragma unroll
        for (int e = 0; e < PPW; e++) {
            // piece = 4 slab rows x 256 B = one contiguous KiB of the tiled HBM layout, already swizzled
            const int r = 4 * (wave * PPW + e);                          // first slab row of the piece
            voff[e] = (uint32_t)(r >> 5) * (32u * a.stride) + (uint32_t)(r & 31) * 256u + (uint32_t)lane * 16u;
        }
        const uint32_t nvoff = (uint32_t)(rt * PVS_AUX_REC + lane) * 4u;  // this wave's tile record: 32 row scalars, then the tile's extremes
        const int rec_wave = wave;  // every wave fetches its own copy of its tile record
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
        int i_tl = 0, i_ck = 0, i_slot = 0, i_nslot = 0;  // tile, chunk within tile, ring chunk slot, row-scalar ring slot
        constexpr int DMA_PARTS = SPB * PPW + 1;  // row pieces + the per-row scalars
        const uint8_t *is_base = nullptr;
        const float *is_aux = nullptr;
        uint32_t is_lds = 0, is_norm = 0;
        auto uni = [](const void *p) {  // pin a wave-uniform pointer in SGPRs (the asm's "s" operands)
            const uint64_t v = (uint64_t)(uintptr_t)p;
            const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
            return (const void *)(uintptr_t)(((uint64_t)hi << 32) | lo);
        };
        auto issue_begin = [&]() {
            const int tl = i_tl < n_my ? i_tl : n_my - 1;  // past the end: harmless re-read keeps vmcnt uniform
            const uint64_t wt = (uint64_t)(sid + (uint32_t)tl * nstreams) * a.tile_step;
            is_base = a.rows + wt * tile_bytes + (uint32_t)i_ck * (SPB * 8192u);  // k-slab = 8 KiB per 32-row tile
            is_aux = a.aux + wt * (RT * PVS_AUX_REC);  // one 64-float record per 32-row tile
            if constexpr (DT == PVS_F32) {
                // the long unrolls of these instances make hipcc lose track of the uniformity of these two
                is_base = (const uint8_t *)uni(is_base);
                is_aux = (const float *)uni(is_aux);
            }
            is_lds = ring_lds + (uint32_t)i_slot * (SPB * SLAB_BYTES) + (uint32_t)wave * (PPW * 1024);
            is_norm = norm_lds + (uint32_t)i_nslot * (WAVES * 256) + (uint32_t)wave * 256;
            if (++i_ck == CPT) {
                i_ck = 0;
                i_tl++;
                if (++i_nslot == NCN) i_nslot = 0;
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
        // scheduler to interleave them).
        using acc_t = typename A::type;
        int c_slot = 0, c_nslot = 0;   // consumer: ring chunk slot, row-scalar slot of the chunk being consumed
        int p_nslot = -1;              // row-scalar slot of the PREVIOUS tile (its last chunk); -1: there is none yet
        elem_t hold[GPW][16];          // the previous tile's 16 dot products per group
#pragma unroll
        for (int g = 0; g < GPW; g++)
#pragma unroll
            for (int r = 0; r < 16; r++) hold[g][r] = 0;
        uint32_t prev_row_base = 0;

        // the previous tile's 16 row scalars of this lane (rows 4h + (r&3) + 8(r>>2)), from its slot of the scalar ring
        auto load_xh = [&](float(&xh)[16]) {
            if (p_nslot < 0) {  // tile "-1": every test fails
#pragma unroll
                for (int r = 0; r < 16; r++) xh[r] = __builtin_nanf("");
                return;
            }
            const float *nl = (const float *)(normring + p_nslot * (WAVES * 256) + rec_wave * 256);
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++) {
                const float4 v = *(const float4 *)(nl + 8 * g4 + 4 * h);
                xh[4 * g4 + 0] = v.x;
                xh[4 * g4 + 1] = v.y;
                xh[4 * g4 + 2] = v.z;
                xh[4 * g4 + 3] = v.w;
            }
        };
        // score of row slot r from its dot product d and row scalar x:
        //   cosine  d * (1/|a|)                  (pass iff sv >= tS)
        //   L2      (1-eR)*|a|^2 - 2*dscale*d    (pass iff sv <= tS)
        auto score = [&](int g, float d, float x) { return COS ? d * x : __builtin_fmaf(d, m2d[g], c1[g] * x); };
        auto undo_f32 = [&](float d, float x) {  // f32 rows: undo the per-row power-of-two scaling (exact)
            if constexpr (DT == PVS_F32) return __builtin_ldexpf(d, -f32_row_exp<COS>(x));
            return d;
        };
        // per-row emission for one group: rows whose exact filter test passes are appended to the lane's segment — with
        // SCALAR stores (s_store_dwordx2, one per candidate, lane by lane off the ballot), fire and forget.  Why not a vector
        // store: gfx9's vmcnt counts stores too, so one issued here makes the next counted LDS-DMA wait cover the store's
        // acknowledgement as well — a partial drain of the prefetch ring.  Scalar stores travel on lgkmcnt, which the LDS-DMA
        // stream never touches; their operands are read at issue (tools/probe/sstore_nowait_test.hip: 84M back-to-back stores
        // with the SGPRs rewritten right behind them), so nothing waits per candidate; s_dcache_wb at the end of the kernel
        // writes the scalar cache back for pass C.  What a candidate costs matters eight-fold: the emitting wave's extra
        // cycles are what the other waves of the workgroup wait for at the next per-tile barrier (§4.1b of HISTORY.md).
        // Two stages, so that the common case — one passing row in one lane — costs one pipelined sweep and one branch chain on
        // SCALAR masks instead of 16 dependent (convert, scale, compare, branch-on-VCC) sequences: stage 1 compares all 16
        // sums with the lane's pre-test bound `eb` (no row scalar, v_cmp straight into an SGPR pair per row); stage 2 runs the
        // exact per-row test only for the rows whose mask is non-empty.  Without a pre-test (f32 rows) stage 1 IS the exact test.
        auto emit_rows = [&](int g, elem_t eb, const float(&xh)[16], auto &&pv) {
            const uint32_t q0 = qoff + (uint32_t)(qw * GPW + g) * 32u;  // first query of the group (wave-uniform)
            const uint32_t seg0 = (sid * RT + rt) * 2;    // segment of half 0 (wave-uniform)
            auto exact = [&](int r) { return score(g, undo_f32((float)pv(g, r), xh[r]), xh[r]); };
            unsigned long long mr[16];
#pragma unroll
            for (int r = 0; r < 16; r++) {
                bool c;
                if constexpr (PRETEST) {
                    c = pv(g, r) >= eb;
                } else {
                    const float sv = exact(r);
                    c = COS ? (sv >= tS[g]) : (sv <= tS[g]);
                }
                mr[r] = __builtin_amdgcn_ballot_w64(c);
            }
#pragma unroll
            for (int r = 0; r < 16; r++) {
                if (mr[r] == 0) continue;
                const float sv = exact(r);
                const bool p = COS ? (sv >= tS[g]) : (sv <= tS[g]);
                unsigned long long m = PRETEST ? __builtin_amdgcn_ballot_w64(p) : mr[r];
                if (m != 0) {
                    uint32_t payload;
                    if constexpr (DT == PVS_I8)
                        payload = (uint32_t)pv(g, r);  // exact integer dot
                    else
                        payload = __builtin_bit_cast(uint32_t, COS ? -sv * qi[g].dscale : sv + qi[g].bb + qi[g].eR * xh[r]);
                    const uint32_t rowv = prev_row_base + (uint32_t)((r & 3) + 8 * (r >> 2));
                    if (a.flat) {  // rerun after a segment overflow: one list per query, slots handed out by an atomic counter
                        if (p) {
                            const uint32_t qg = q0 + (uint32_t)j;
                            const uint32_t pos = atomicAdd(a.flat_cnt + qg, 1u);
                            if (pos < a.flat_cap) a.flat[(size_t)qg * a.flat_cap + pos] = make_uint2(rowv, payload);
                        }
                        continue;
                    }
                    do {
                        const int l = __builtin_ctzll(m);
                        m &= m - 1;
                        const uint32_t pos_s = (uint32_t)__builtin_amdgcn_readlane((int)mycnt[g], l);
                        if (pos_s < a.seg_cap) {
                            const uint32_t row_s = (uint32_t)__builtin_amdgcn_readlane((int)rowv, l);
                            const uint32_t key_s = (uint32_t)__builtin_amdgcn_readlane((int)payload, l);
                            const uint2 *dst = a.seg + ((size_t)(seg0 + ((uint32_t)l >> 5)) * a.seg_queries + (q0 + ((uint32_t)l & 31u))) * a.seg_cap + pos_s;
                            const uint64_t data = ((uint64_t)key_s << 32) | row_s;
                            // s_nop: a readlane result (VALU-written SGPR) may not feed an SMEM instruction within 4 wait states,
                            // and the hazard recognizer does not look inside inline asm
                            asm volatile("s_nop 4\n\ts_store_dwordx2 %0, %1, 0x0" ::"s"(data), "s"(dst) : "memory");
                        }
                    } while (m != 0);
                    if (p) mycnt[g]++;
                }
            }
        };
        // ---- epilogue of the previous tile, cut into micro-steps the main loop drops between MFMAs, plus a rest.
        //  PRETEST: 8 steps per group (v_max3 fold of two accumulators each) — no row scalar is touched;
        //  otherwise 16 steps per group (8 score pairs, 8 folds), row scalars in xh.
        // MODE 5: per-query constants of the bracket arithmetic, and the NEXT fold's tile record (+ candidate-mask bits), loaded through
        // the scalar cache at the end of the tile before — the record of the tile a fold works on was asked for one whole tile
        // earlier (asked for at the fold itself, its ~1 us of latency stalled the one wave per SIMD once per tile)
        const bool f5_ok = qi[0].bb == qi[0].bb && qi[0].bb < __builtin_inff() && (!COS || qi[0].bb > 0.f) && qi[0].dscale > 0.f;
        const float f5_inv_sb = COS && f5_ok ? 1.0f / sqrtf(qi[0].bb) : 0.f;
        uint32_t f5_g = 0, f5_last = 0, f5_spill = 0, f5_allow = 0xffffffffu, f5_seq = 0;
        auto fold_prefetch = [&](uint32_t row_base) {
            if constexpr (MODE == 5) {
                typedef const __attribute__((address_space(4))) uint32_t *cptr32;
                const uint32_t tile_u = (uint32_t)__builtin_amdgcn_readfirstlane((int)(row_base >> 5));
                if ((uint64_t)tile_u * 32u >= a.n_rows) return;
                cptr32 tg = (cptr32)(uintptr_t)(a.tile_grp + tile_u);
                f5_g = tg[0];
                f5_last = tg[1];
                f5_spill = tg[2];
                f5_allow = 0xffffffffu;
                if (a.fold_mask) {
                    cptr32 mk = (cptr32)(uintptr_t)(a.fold_mask + (uint64_t)tile_u * 32u);
                    uint32_t m = 0;
#pragma unroll
                    for (int w8 = 0; w8 < 8; w8++) {
                        const uint32_t v = mk[w8];
#pragma unroll
                        for (int b = 0; b < 4; b++) m |= ((v >> (8 * b)) & 0xffu) ? (1u << (4 * w8 + b)) : 0u;
                    }
                    f5_allow = m;
                }
            }
        };
        struct Epi {
            float xh[16];
            float lo16[MODE == 5 ? 16 : 1], hi16[MODE == 5 ? 16 : 1];  // MODE 5: the lane's 16 row brackets (computed in the MFMAs' shadow)
            float sv[GPW][16];
            float best[GPW];
            elem_t mx[GPW];
            elem_t eb[GPW];  // PRETEST: the lane's bound for this tile
            float t0, t1;    // PRETEST: the previous tile's extreme row scalars
        };
        constexpr int EPI_STEPS = PRETEST ? 8 * GPW : 16 * GPW;
        auto epi_begin = [&](Epi &e) {
            if constexpr (!PRETEST) load_xh(e.xh);
#pragma unroll
            for (int g = 0; g < GPW; g++) {
                e.best[g] = COS ? -__builtin_inff() : __builtin_inff();
                e.mx[g] = A::lowest();
            }
        };
        auto epi_micro = [&](int m, Epi &e, auto &&pv) {
            if constexpr (PRETEST) {
                // Every slice ends in an empty volatile asm on its result: without it the optimizer reassociates the fold and
                // sinks it, the bound and the LDS read of the extremes behind the last MFMA of the tile, where nothing hides
                // them (~250 cycles per wave and tile, in front of the barrier the other waves are waiting at).
                const int g = m / 8, i = m % 8;
                if (m == 0) {
                    // Necessary condition for "some row of this lane passes", from the tile's extreme row scalars, which
                    // ride behind the 32 row scalars in the tile's record (k_scan_aux: t0 = min |a| resp. min |a|^2 over
                    // the tile's rows with a usable norm, t1 = the max).  Tile "-1": NaN bounds, nothing passes.
                    e.t0 = e.t1 = __builtin_nanf("");
                    if (p_nslot >= 0) {
                        const float2 tmm = *(const float2 *)((const float *)(normring + p_nslot * (WAVES * 256) + rec_wave * 256) + 32);
                        e.t0 = tmm.x;
                        e.t1 = tmm.y;
                    }
                    asm volatile("" : "+v"(e.t0), "+v"(e.t1));
                }
                if (i == 4) {
                    //   cosine  d/|a| >= tS            =>  d >= tS * (tS > 0 ? min|a| : max|a|)
                    //   L2      c1|a|^2 - 2 ds d <= tS =>  d >= (c1 * min|a|^2 - tS) / (2 ds)
                    // minus a slack that covers the f32 roundings of the exact test (2^-18 relative is 30x what they add up
                    // to).  NaN bounds (no usable row) compare false: nothing passes, which is right.
                    float b, mag;
                    if (COS) {
                        b = tS[g] * (tS[g] > 0.f ? e.t0 : e.t1);
                        mag = fabsf(b);
                    } else {
                        const float x = c1[g] * e.t0;
                        b = (x - tS[g]) * hd[g];
                        mag = (fabsf(x) + fabsf(tS[g])) * hd[g];
                    }
                    b = b - mag * 3.8147e-6f;
                    if constexpr (DT == PVS_I8) {
                        // integer dots (|d| < 2^24, exact in f32): d >= b  <=>  d >= ceil(b); one more of slack for the floor
                        b -= 1.0f;
                        e.eb[g] = b == b ? (int)fminf(fmaxf(ceilf(b), -1.0e9f), 1.0e9f) : 0x7fffffff;
                    } else {
                        e.eb[g] = b;
                    }
                    asm volatile("" : "+v"(e.eb[g]));
                }
                e.mx[g] = A::max3(e.mx[g], pv(g, 2 * i), pv(g, 2 * i + 1));
                asm volatile("" : "+v"(e.mx[g]));
            } else {
                const int g = m / 16, i = m % 16;
                if (i < 8) {
                    e.sv[g][2 * i] = score(g, undo_f32((float)pv(g, 2 * i), e.xh[2 * i]), e.xh[2 * i]);
                    e.sv[g][2 * i + 1] = score(g, undo_f32((float)pv(g, 2 * i + 1), e.xh[2 * i + 1]), e.xh[2 * i + 1]);
                } else if constexpr (MODE == 5) {
                    // the brackets of two rows: [D(key - err), D(key + err)] widened by 1e-6 (1 + |d|); [-inf, +inf] for a row whose distance
                    // may be NULL (norm zero / not finite, key not a number): it poisons its file's sums (forced candidate)
#pragma unroll
                    for (int rr = (i - 8) * 2; rr < (i - 8) * 2 + 2; rr++) {
                        const float x = e.xh[rr];  // cosine: 1/|a|, L2: |a|^2
                        const float key = COS ? -e.sv[0][rr] * qi[0].dscale : e.sv[0][rr] + qi[0].bb + qi[0].eR * x;
                        const bool valid = (COS ? (x > 1e-15f && x < 1e15f) : (x >= 0.f && x < 1e30f)) && fabsf(key) <= 1e30f;
                        float lo, hi;
                        if (COS) {
                            lo = 1.0f + (key - qi[0].eA) * f5_inv_sb;
                            hi = 1.0f + (key + qi[0].eA) * f5_inv_sb;
                        } else {
                            const float err = qi[0].eA + qi[0].eR * x;
                            lo = sqrtf(fmaxf(key - err, 0.f));
                            hi = sqrtf(fmaxf(key + err, 0.f));
                        }
                        lo -= 1e-6f * (1.0f + fabsf(lo));
                        hi += 1e-6f * (1.0f + fabsf(hi));
                        e.lo16[rr] = valid ? lo : -__builtin_inff();
                        e.hi16[rr] = valid ? hi : __builtin_inff();
                    }
                } else {
                    const int r = (i - 8) * 2;
                    // NaN (padding / zero-norm rows) never wins a fmax/fmin
                    e.best[g] = COS ? fmaxf(e.best[g], fmaxf(e.sv[g][r], e.sv[g][r + 1])) : fminf(e.best[g], fminf(e.sv[g][r], e.sv[g][r + 1]));
                }
            }
        };
        bool prev_valid = false;  // MODE 2: the dummy tile "-1" writes nothing
        // MODE 2 with the per-group fold (ScanK.tile_grp): the previous tile's 32 rows x 32 queries of this wave, folded per group.
        // Lane (j, h) holds query j and 16 of the 32 rows; the halves exchange theirs (one cross-lane move per value), then the
        // lanes of half 0 walk the 32 rows IN ROW ORDER — SQLite's SUM / AVG are order-dependent compensated sums — under
        // wave-uniform control flow: the tile record (which rows end a group, which belong to a group that crosses the tile
        // boundary) and the optional per-row weights / candidate mask arrive through the scalar cache (s_load: lgkmcnt, so the
        // counted vmcnt waits of the LDS-DMA stream never see them).  The value of a group that ends here goes to
        // fold_out[group][query]: 32 lanes, 256 contiguous bytes.
        auto fold_groups = [&](Epi &e, auto &&pv) {
            if constexpr (MODE == 3) {
                typedef const __attribute__((address_space(4))) uint32_t *cptr32;
                const uint32_t tile_u = (uint32_t)__builtin_amdgcn_readfirstlane((int)(prev_row_base >> 5));  // 32-row tile of this wave (uniform)
                const uint64_t row0 = (uint64_t)tile_u * 32u;
                if (row0 >= a.n_rows) return;
                float dm[16];
                // cosine: sqrt(|a|^2) in f64 once per ROW, not once per (row, query) — lane L takes row L & 31 of the tile (its |a|^2
                // sits in the tile record the previous tile left in LDS) and the sixteen a lane needs come over the cross-lane
                // network; the same correctly rounded value ref_cosine_finish computes, a third of its instructions gone
                double sa_mine = 0.0, sb = 0.0;
                if (COS) {
                    sa_mine = __dsqrt_rn((double)((const float *)(normring + p_nslot * (WAVES * 256) + rec_wave * 256))[j]);
                    sb = __dsqrt_rn((double)qi[0].bb);
                }
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    if (COS) {
                        const double sa = __shfl(sa_mine, (r & 3) + 8 * (r >> 2) + 4 * h, 64);
                        dm[r] = (float)(1.0 - (double)(float)pv(0, r) / (sa * sb));
                    } else {
                        const double ss = (double)e.xh[r] + (double)qi[0].bb - 2.0 * (double)pv(0, r);
                        // (padding rows behind the last stored row carry a NaN norm: they are no reason to leave the closed form)
                        if (!(ss < 16777216.0) && row0 + (uint64_t)((r & 3) + 8 * (r >> 2) + 4 * h) < a.n_rows && myq[0] < (int)a.batch) *a.dense_flag = 1u;  // (plain store: the word may be pinned host memory; 1 is the only value written)
                        dm[r] = ref_l2_finish((float)ss);
                    }
                }
#if defined(PVS_FOLD_ABL) && PVS_FOLD_ABL == 1  // (tuning builds: what does each part of the fold cost — tools/r4_fold_ablation.sh)
                {
                    float acc_ = 0.f;
                    for (int r = 0; r < 16; r++) acc_ += dm[r];
                    if (acc_ == 12345.678f) a.dense_out[0] = acc_;
                    return;
                }
#endif
                float D[32];
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const float other = __shfl_xor(dm[r], 32, 64);
                    const int t = (r & 3) + 8 * (r >> 2);
                    D[t] = h ? other : dm[r];       // rows of half 0
                    D[t + 4] = h ? dm[r] : other;   // rows of half 1
                }
                cptr32 tg = (cptr32)(uintptr_t)(a.tile_grp + tile_u);
                uint32_t g_run = tg[0];
                const uint32_t m_last = tg[1], m_spill = tg[2];
                uint32_t m_allow = 0xffffffffu;
                if (a.fold_mask) {
                    cptr32 mk = (cptr32)(uintptr_t)(a.fold_mask + row0);
                    m_allow = 0;
#pragma unroll
                    for (int w8 = 0; w8 < 8; w8++) {
                        const uint32_t v = mk[w8];
#pragma unroll
                        for (int b = 0; b < 4; b++) m_allow |= ((v >> (8 * b)) & 0xffu) ? (1u << (4 * w8 + b)) : 0u;
                    }
                }
                const uint32_t n_here = a.n_rows - row0 >= 32u ? 32u : (uint32_t)(a.n_rows - row0);
                const uint32_t m_rows = n_here == 32u ? 0xffffffffu : ((1u << n_here) - 1u);
                const uint32_t m_use = m_allow & ~m_spill & m_rows;  // rows folded here
                const uint32_t m_sp = m_spill & m_rows;              // rows that go through the matrix
                const uint32_t m_end = m_last & m_rows;
#if defined(PVS_FOLD_ABL) && PVS_FOLD_ABL == 2
                {
                    float acc_ = 0.f;
                    for (int r = 0; r < 32; r++) acc_ += D[r];
                    if (acc_ == 12345.678f + (float)(g_run + m_end + m_sp + m_use)) a.dense_out[0] = acc_;
                    return;
                }
#endif
                const int q = myq[0];
                const bool mine = h == 0 && q < (int)a.batch;
                // rows of tile-crossing groups: into the matrix (k_group_aggregate_list folds them)
                if (m_sp != 0) {
#pragma unroll
                    for (int i = 0; i < 32; i++)
                        if (((m_sp >> i) & 1u) && mine) a.dense_out[(row0 + (uint64_t)i) * a.dense_ld + (uint32_t)q] = D[i];
                }
                // The fold proper is STRAIGHT-LINE code per row (one wave per SIMD runs this: every taken branch and every dependent
                // f64 operation is paid in full).  Per row: the candidate next state of the compensated sums (SQLite's
                // Kahan-Babuska-Neumaier step, filters/exact.rs:67-80 -> SUM / AVG) is computed unconditionally and SELECTED into the
                // state when the row takes part (a candidate, not spilled, distance not NULL) — the state of a skipped row is
                // untouched bit for bit; only the serial chains s -> s' and c -> c' link consecutive rows, the corrections are
                // independent work the scheduler overlaps.  The one uniform branch per row is "a group ends here".
                auto emit = [&](double v, bool any_joined, bool any_cnt) {
                    if (!any_joined)
                        v = __builtin_bit_cast(double, PVS_GROUP_ABSENT);
                    else if (!any_cnt)
                        v = __builtin_nan("");
                    if (mine) a.fold_out[(size_t)g_run * a.fold_ld + (uint32_t)q] = v;
                    g_run++;
                };
                auto kbn_next = [](double s, double c, double r, double &s2, double &c2) {
                    const double t = s + r;
                    const double x = (s - t) + r, y = (r - t) + s;
                    c2 = c + (fabs(s) > fabs(r) ? x : y);
                    s2 = t;
                };
                if (a.fold_weights) {
                    cptr32 wp = (cptr32)(uintptr_t)(a.fold_weights + row0);
                    uint32_t wb[32];
#pragma unroll
                    for (int i = 0; i < 32; i++) wb[i] = wp[i];  // (one batch of scalar loads, not one round trip per row)
                    double s_s = 0.0, s_c = 0.0, w_s = 0.0, w_c = 0.0;
                    uint32_t cnt = 0, joined = 0;
#pragma unroll
                    for (int i = 0; i < 32; i++) {
                        const bool use = (m_use >> i) & 1u;  // uniform
                        const double w = (double)__builtin_bit_cast(float, wb[i]);
                        double n_s, n_c;
                        kbn_next(w_s, w_c, w, n_s, n_c);   // SUM(w) runs over every joined row, NULL distance or not
                        w_s = use ? n_s : w_s;
                        w_c = use ? n_c : w_c;
                        joined += use ? 1u : 0u;
                        const float df = D[i];
                        const bool val = use && df == df;
                        kbn_next(s_s, s_c, (double)df * w, n_s, n_c);
                        s_s = val ? n_s : s_s;
                        s_c = val ? n_c : s_c;
                        cnt += val ? 1u : 0u;
                        if ((m_end >> i) & 1u) {
                            if (!((m_sp >> i) & 1u)) emit((s_s + s_c) / (w_s + w_c), joined != 0, cnt != 0);
                            else g_run++;
                            s_s = s_c = w_s = w_c = 0.0;
                            cnt = joined = 0;
                        }
                    }
                } else if (a.fold_agg == PVS_AGG_AVG) {
                    double s_s = 0.0, s_c = 0.0;
                    uint32_t cnt = 0, joined = 0;
#pragma unroll
                    for (int i = 0; i < 32; i++) {
                        const bool use = (m_use >> i) & 1u;
                        joined += use ? 1u : 0u;
                        const float df = D[i];
                        const bool val = use && df == df;
                        double n_s, n_c;
                        kbn_next(s_s, s_c, (double)df, n_s, n_c);
                        s_s = val ? n_s : s_s;
                        s_c = val ? n_c : s_c;
                        cnt += val ? 1u : 0u;
                        if ((m_end >> i) & 1u) {
                            if (!((m_sp >> i) & 1u)) emit((s_s + s_c) / (double)cnt, joined != 0, cnt != 0);
                            else g_run++;
                            s_s = s_c = 0.0;
                            cnt = joined = 0;
                        }
                    }
                } else {
                    const bool want_min = a.fold_agg == PVS_AGG_MIN;
                    double ext = want_min ? __builtin_inf() : -__builtin_inf();
                    uint32_t cnt = 0, joined = 0;
#pragma unroll
                    for (int i = 0; i < 32; i++) {
                        const bool use = (m_use >> i) & 1u;
                        joined += use ? 1u : 0u;
                        const float df = D[i];
                        const bool val = use && df == df;
                        const double d = (double)df;
                        const double nx = want_min ? fmin(ext, d) : fmax(ext, d);
                        ext = val ? nx : ext;
                        cnt += val ? 1u : 0u;
                        if ((m_end >> i) & 1u) {
                            if (!((m_sp >> i) & 1u)) emit(ext, joined != 0, cnt != 0);
                            else g_run++;
                            ext = want_min ? __builtin_inf() : -__builtin_inf();
                            cnt = joined = 0;
                        }
                    }
                }
            }
        };
        // MODE 5 — the certified per-item search of FLOAT indexes (pvs_items_float.hip) folded into the scan's epilogue: the previous
        // tile's 32 rows x 32 queries of this wave become per-row BRACKETS of the reference's distance (from the scan key and its
        // rigorous error eA + eR |a|^2, HISTORY.md section 4.2), the brackets of every file that lies inside the tile are folded (AVG: sums,
        // MIN / MAX: extremes, positive weights: weighted sums — f32: at most 32 rows, the roundings are inside the margin) and
        // the file's bracket [L, U] leaves as L -> fold_out[file][query] and U -> an atomic minimum in its bucket.  No key matrix is
        // written or read (MODE 4 + k_run_bounds: 0.5 GB out, 0.5 GB back, 0.5 ms at 4M x 768 x 32).  Same control structure as
        // MODE 3's fold: tile record, weights and mask through the scalar cache, wave-uniform branches; the row brackets of files
        // that CROSS a tile boundary go to two sparse matrices (dense_out: lower ends, dense_out + fold_hi_off: upper ends) and are
        // folded by k_spill_bounds afterwards.  A row that may be NULL (norm zero / not finite, key not a number) poisons its file
        // (NaN bracket -> forced candidate).
        auto fold_brackets = [&](Epi &e) {
            if constexpr (MODE == 5) {
                typedef const __attribute__((address_space(4))) uint32_t *cptr32;
                const uint32_t tile_u = (uint32_t)__builtin_amdgcn_readfirstlane((int)(prev_row_base >> 5));
                const uint64_t row0 = (uint64_t)tile_u * 32u;
                const uint32_t t_slot = (f5_seq++) & (uint32_t)(FOLD_SLOTS - 1);  // the bucket of this tile's files: (this wave, tile number mod FOLD_SLOTS)
                if (row0 >= a.n_rows) return;
                const int q = myq[0];
                if (q >= (int)a.batch) return;  // (padding queries: nothing of theirs leaves the kernel)
                // The halves of the wave split the work: after ONE v_permlane32_swap per pair of rows the lanes of half 0 hold the LOWER ends
                // of all 32 rows of their query, the lanes of half 1 the UPPER ends — the same instruction stream folds both (a sum, an
                // extreme), widens away from the value (sgn) and keeps one running minimum: of the lower bounds in half 0 (which buckets
                // can hold a candidate), of the upper bounds in half 1 (the threshold).  (Both ends in every lane, exchanged by 32
                // shuffles and 64 selects, every minimum through fminf's canonicalisation, an IEEE division per file: ~1,600 instructions
                // per tile on the one wave per SIMD — 0.39 of the scan's 1.30 ms at 4M x 768 x 32.)
                float X[32];
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    // (inline asm: with __builtin_amdgcn_permlane32_swap hipcc 7.2 used the FIRST result for both — rows t + 4 got the
                    //  values of rows t (tools/probe/permlane32_swap_test.hip shows the instruction itself does what the ISA says); the
                    //  nops are the VALU-write -> permlane hazard the compiler would have covered)
                    float xa = e.lo16[r], xb = e.hi16[r];
                    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(xa), "+v"(xb));
                    const int t = (r & 3) + 8 * (r >> 2);
                    X[t] = xa;      // half 0: lo of row t (its own),        half 1: hi of row t (half 0's)
                    X[t + 4] = xb;  // half 0: lo of row t + 4 (half 1's),   half 1: hi of row t + 4 (its own)
                }
                const float sgn = h ? 1.0f : -1.0f;
                const float far = h ? __builtin_inff() : -__builtin_inff();  // the end of a bracket that says nothing
                const float bad = f5_ok ? 0.f : __builtin_nanf("");           // nothing of this query can be bracketed: NaN ends (compare false everywhere)
                uint32_t g_run = f5_g;  // (the record asked for at the end of the tile before: fold_prefetch)
                const uint32_t m_last = f5_last, m_spill = f5_spill, m_allow = f5_allow;
                const uint32_t n_here = a.n_rows - row0 >= 32u ? 32u : (uint32_t)(a.n_rows - row0);
                const uint32_t m_rows = n_here == 32u ? 0xffffffffu : ((1u << n_here) - 1u);
                const uint32_t m_use = m_allow & ~m_spill & m_rows;
                const uint32_t m_sp = m_spill & m_rows;
                const uint32_t m_end = m_last & m_rows;
                if (m_sp != 0) {
                    float *sp_out = a.dense_out + (h ? a.fold_hi_off : 0) + row0 * a.dense_ld + (uint32_t)q;
#pragma unroll
                    for (int i = 0; i < 32; i++)
                        if ((m_sp >> i) & 1u) sp_out[(size_t)i * a.dense_ld] = X[i];
                }
                float *lo_out = (float *)a.fold_out + (uint32_t)q;
                float t_m = __builtin_inff();  // this tile's minimum of the ends this lane folds
                // val: the file's aggregate of this lane's ends; cnt rows took part; forced: a weight that says nothing
                auto emit = [&](float val, uint32_t cnt, bool forced) {
                    float v;
                    if (cnt == 0) {
                        v = __builtin_inff();  // no candidate row: the file is not part of the result at all
                    } else if (forced) {
                        v = far;
                    } else {
                        const float mg = sgn * (1e-6f + 1.5e-7f * (float)cnt);
                        v = __builtin_fmaf(mg, 1.0f + __builtin_fabsf(val), val);  // (an infinite end stays where it is)
                    }
                    v += bad;
                    if (h == 0) lo_out[(size_t)g_run * a.fold_ld] = v;
                    t_m = v < t_m ? v : t_m;
                    g_run++;
                };
                if (a.fold_weights) {
                    cptr32 wp = (cptr32)(uintptr_t)(a.fold_weights + row0);
                    uint32_t wb[32];
#pragma unroll
                    for (int i = 0; i < 32; i++) wb[i] = wp[i];
                    float s = 0.f, s_w = 0.f;
                    uint32_t cnt = 0;
                    bool forced = false;
#pragma unroll
                    for (int i = 0; i < 32; i++) {
                        const bool use = (m_use >> i) & 1u;  // uniform
                        const float w = __builtin_bit_cast(float, wb[i]);
                        if (use) {
                            cnt++;
                            forced |= !(w > 0.f && w < 1e30f);
                            s = __builtin_fmaf(X[i], w, s);
                            s_w += w;
                        }
                        if ((m_end >> i) & 1u) {
                            if (!((m_sp >> i) & 1u))
                                emit(s * __builtin_amdgcn_rcpf(s_w), cnt, forced);  // (1 ulp: inside the margin)
                            else
                                g_run++;
                            s = s_w = 0.f;
                            cnt = 0;
                            forced = false;
                        }
                    }
                } else if (a.fold_agg == PVS_AGG_AVG) {
                    // (ALL: no candidate mask, a full tile — the rows of tile-crossing files may be added like any other, what they are added
                    //  to is never emitted: one scalar branch per row instead of two)
                    auto rows = [&](auto all_rows) {
                        float s = 0.f;
                        uint32_t cnt = 0;
#pragma unroll
                        for (int i = 0; i < 32; i++) {
                            if (decltype(all_rows)::value || ((m_use >> i) & 1u)) {
                                cnt++;
                                s += X[i];  // (a row whose distance may be NULL carries -inf / +inf: the sum says nothing, as it must)
                            }
                            if ((m_end >> i) & 1u) {
                                if (!((m_sp >> i) & 1u))
                                    emit(s * __builtin_amdgcn_rcpf((float)(cnt ? cnt : 1u)), cnt, false);
                                else
                                    g_run++;
                                s = 0.f;
                                cnt = 0;
                            }
                        }
                    };
                    if ((m_allow & m_rows) == 0xffffffffu)
                        rows(std::true_type{});
                    else
                        rows(std::false_type{});
                } else {
                    // MIN: the smallest lower / upper end (a NULL-able row: -inf below, ignored above — MIN skips a NULL, and a row that
                    // is not NULL can only lower it); MAX: the mirror image
                    auto rows = [&](auto all_rows, auto want_min) {
                        constexpr bool MN = decltype(want_min)::value;
                        float x = MN ? __builtin_inff() : -__builtin_inff();
                        uint32_t cnt = 0;
#pragma unroll
                        for (int i = 0; i < 32; i++) {
                            if (decltype(all_rows)::value || ((m_use >> i) & 1u)) {
                                cnt++;
                                const bool take = MN ? X[i] < x : X[i] > x;
                                x = take ? X[i] : x;
                            }
                            if ((m_end >> i) & 1u) {
                                if (!((m_sp >> i) & 1u))
                                    emit(x, cnt, false);
                                else
                                    g_run++;
                                x = MN ? __builtin_inff() : -__builtin_inff();
                                cnt = 0;
                            }
                        }
                    };
                    const bool all = (m_allow & m_rows) == 0xffffffffu;
                    if (a.fold_agg == PVS_AGG_MIN) {
                        if (all)
                            rows(std::true_type{}, std::true_type{});
                        else
                            rows(std::false_type{}, std::true_type{});
                    } else {
                        if (all)
                            rows(std::true_type{}, std::false_type{});
                        else
                            rows(std::false_type{}, std::false_type{});
                    }
                }
#pragma unroll
                for (int i = 0; i < FOLD_SLOTS; i++) xmin[i] = (t_slot == (uint32_t)i && t_m < xmin[i]) ? t_m : xmin[i];
            }
        };
        auto epi_rest = [&](Epi &e, auto &&pv) {
            if constexpr (MODE == 3) {
                if (prev_valid) fold_groups(e, pv);
            } else if constexpr (MODE == 2) {
                // dense exact int8 distances (the reference's dist_{cte}.d for a batch of queries):
                // closed form of the exact integer sums, valid while they stay below 2^24
                // (oracle: orc_i8_cosine_from_sums / orc_i8_l2_from_sums).  xh = |a|^2 here.
#pragma unroll
                for (int g = 0; g < GPW; g++)
                    if (myq[g] < (int)a.batch) {
#pragma unroll
                        for (int r = 0; r < 16; r++) {
                            const uint32_t row = prev_row_base + (r & 3) + 8 * (r >> 2);
                            if (row < a.n_rows && prev_valid) {
                                float d;
                                if (COS) {
                                    d = ref_cosine_finish((float)pv(g, r), e.xh[r], qi[g].bb);
                                } else {
                                    const double ss = (double)e.xh[r] + (double)qi[g].bb - 2.0 * (double)pv(g, r);
                                    if (!(ss < 16777216.0)) *a.dense_flag = 1u;  // (plain store: the word may be pinned host memory; 1 is the only value written)
                                    d = ref_l2_finish((float)ss);
                                }
                                a.dense_out[(size_t)row * a.dense_ld + myq[g]] = d;
                            }
                        }
                    }
            } else if constexpr (MODE == 5) {
                if (prev_valid) fold_brackets(e);
            } else if constexpr (MODE == 4) {
                // the scan KEY of every (row, query) pair, dense: dense_out[row][query] = key with |key - kappa| <= eA + eR |a|^2
                // (kappa = the reference key its distance is a monotone function of, HISTORY.md section 4.2) — what passes A / B
                // test against thresholds, written out instead: the certified per-item search of float indexes (pvs_items_float.hip)
                // brackets every row's distance with it, folds the brackets per file, and rescans exactly only the files that can
                // reach the page.  cosine: key = -dscale acc / |a|; L2: key = |a|^2 + bb - 2 dscale acc (the payload form of MODE 1).
#pragma unroll
                for (int g = 0; g < GPW; g++)
                    if (myq[g] < (int)a.batch && prev_valid) {
#pragma unroll
                        for (int r = 0; r < 16; r++) {
                            const uint32_t row = prev_row_base + (r & 3) + 8 * (r >> 2);
                            if (row < a.n_rows)
                                a.dense_out[(size_t)row * a.dense_ld + myq[g]] = COS ? -e.sv[g][r] * qi[g].dscale : e.sv[g][r] + qi[g].bb + qi[g].eR * e.xh[r];
                        }
                    }
            } else if constexpr (MODE == 0) {
#pragma unroll
                for (int g = 0; g < GPW; g++)
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        // upper bound of this row's key: key + err
                        const float ub = COS ? __builtin_fmaf(-e.sv[g][r], qi[g].dscale, qi[g].eA)
                                             : e.sv[g][r] + (qi[g].bb + qi[g].eA) + 2.0f * qi[g].eR * e.xh[r];
                        mins[g][r] = fminf(mins[g][r], ub);
                    }
            } else if constexpr (PRETEST) {
                // the bound was computed in the shadow of the MFMAs (epi_micro); what is left here is one compare per group
                bool any = false;
                bool lp[GPW];
#pragma unroll
                for (int g = 0; g < GPW; g++) {
                    lp[g] = e.mx[g] >= e.eb[g];
                    any |= lp[g];
                }
                if (__builtin_amdgcn_ballot_w64(any) != 0) {
                    // Some lane of the wave may hold a passing row (~15 % of the wave-tiles at k=100 over 10M rows): now
                    // the per-row scalars are needed.  Staging is wave-private and its fill count lives in a scalar
                    // register: no LDS atomics, no barrier.  One ballot per tile row r (compile-time r).
                    load_xh(e.xh);
#pragma unroll
                    for (int g = 0; g < GPW; g++)
                        if (__builtin_amdgcn_ballot_w64(lp[g]) != 0) emit_rows(g, e.eb[g], e.xh, pv);
                }
            } else {
#pragma unroll
                for (int g = 0; g < GPW; g++) {
                    const bool lane_pass = COS ? (e.best[g] >= tS[g]) : (e.best[g] <= tS[g]);
                    if (__builtin_amdgcn_ballot_w64(lane_pass) != 0) emit_rows(g, elem_t{}, e.xh, pv);
                }
            }
        };

        // One tile: MFMA burst into two accumulation chains (acc, acc1) with the previous tile's epilogue slices in its shadow;
        // `pv(g, r)` = the previous tile's r-th dot product of group g, summed into `hold` at the end of its tile.
        auto run_tile = [&](int tl, acc_t(&acc)[GPW], acc_t &acc1, auto &&pv) {
#pragma unroll
            for (int g = 0; g < GPW; g++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[g][r] = 0;
#pragma unroll
            for (int r = 0; r < 16; r++) acc1[r] = 0;
            Epi e;
#pragma unroll
            for (int ck = 0; ck < CPT; ck++) {
                wait_vm<(PC - 1) * G::VM_PER_CHUNK>();  // this wave's share of the chunk has landed
                wg_barrier();                           // ... and everyone else's; the previous chunk is consumed
                issue_begin();                          // the slot the previous chunk occupied is refilled below
                if (ck == 0) epi_begin(e);
                const uint8_t *cb = ring + c_slot * (SPB * SLAB_BYTES) + frag_row;
                float row_scale = 1.0f;  // f32 rows: 2^e of this lane's A row (its scalar sits in this chunk's slot)
                if constexpr (DT == PVS_F32) {
                    const float ax = ((const float *)(normring + c_nslot * (WAVES * 256) + rec_wave * 256))[j];
                    row_scale = __builtin_ldexpf(1.0f, f32_row_exp<COS>(ax));
                }
                (void)row_scale;
                const uint8_t *fb[8];
#pragma unroll
                for (int i = 0; i < 8; i++) fb[i] = cb + swz[i];
                // Explicit software pipeline, fenced with sched_barrier(0) so hipcc keeps the order:
                //   step t:  LDS read of fragment t+PF | MFMA t (one per group) | one DMA piece of the chunk PC ahead |
                //            a slice of the previous tile's epilogue
                // A wave issues in order, so only VALU placed BETWEEN MFMAs runs in their shadow.
                constexpr int NF = SPB * SPS, PF = 4;  // (prefetch depth 2 / 6 / 8 / 12 at 128 queries: 1.297 / 1.304 / 1.296 / 1.305 ms against 1.302)
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
                    if (t & 1)
                        acc1 = A::mfma(af[t], qf[0][ck * NF + t], acc1);
                    else
                        acc[0] = A::mfma(af[t], qf[0][ck * NF + t], acc[0]);
                    if (t + 1 < NF) narrow(t + 1);
#pragma unroll
                    for (int part = t * DMA_PARTS / NF; part < (t + 1) * DMA_PARTS / NF; part++) issue_part(part);
                    if (ck == 0) {
#pragma unroll
                        for (int m = t * EPI_STEPS / NF; m < (t + 1) * EPI_STEPS / NF; m++) epi_micro(m, e, pv);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (++c_slot == NC) c_slot = 0;
                if (ck == CPT - 1) {  // this tile's row scalars: kept in their slot until the end of the NEXT tile
                    epi_rest(e, pv);  // (the previous tile's, reading p_nslot)
                    p_nslot = c_nslot;
                    if (++c_nslot == NCN) c_nslot = 0;
                }
            }
            // ---- hand this tile's results to the next iteration
#pragma unroll
            for (int r = 0; r < 16; r++) hold[0][r] = A::sum2(acc[0], acc1, r);  // i8: exact integers; floats: within the error budget
            prev_row_base = (uint32_t)((sid + (uint32_t)tl * nstreams) * a.tile_step * SLAB_ROWS) + rt * 32 + 4 * h;
            prev_valid = true;
            fold_prefetch(prev_row_base);
        };
        // The last tile's epilogue runs inside one extra "ghost" tile (the DMA stream already re-reads the last tile past
        // the end to keep vmcnt uniform; its sums are never looked at): 1/n_my more work, but the epilogue — the bulk of the
        // kernel's code — exists once instead of three more times in a drain path.
        {
            auto ph = [&](int g, int r) { return hold[g][r]; };
            for (int tl = 0; tl < n_my + 1; tl++) {
                acc_t acc[GPW], acc1;
                run_tile(tl, acc, acc1, ph);
            }
        }
        wait_vm<0>();  // retire the dummy tail DMAs before the wave exits
        if (MODE == 1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory");  // candidates: scalar cache -> L2
    }
    if (MODE == 1 && sid < nstreams) {   // every lane's fill count: seg_cnt[query][segment] (above seg_cap: overflowed -> dense path)
#pragma unroll
        for (int g = 0; g < GPW; g++) a.seg_cnt[(size_t)myq[g] * a.seg_stride + seg] = mycnt[g];
    }

    if (MODE == 5 && sid < nstreams && myq[0] < (int)a.batch) {
        // this lane's FOLD_SLOTS bucket minima: buckets [(sid * RT + rt) * FOLD_SLOTS, +FOLD_SLOTS) of its query, query-minor rows — half 1:
        // of the upper bounds (raised to 0: raising an upper bound keeps it one; non-negative floats order like their bit patterns),
        // half 0: of the lower bounds
        float *out = h ? (float *)a.fold_bucket : a.fold_bucket_lo;
#pragma unroll
        for (int i = 0; i < FOLD_SLOTS; i++) {
            const float v = xmin[i];
            out[(size_t)((sid * RT + rt) * FOLD_SLOTS + i) * a.fold_ld + (uint32_t)myq[0]] = (h && v < 0.f) ? 0.f : v;
        }
    }
    if (MODE == 0 && sid < nstreams) {
        // Each lane holds 16 minima per query (one per accumulator row slot) = 16 disjoint row groups of its query.
        // The threshold only needs a few times k groups per query; folding to a.gmin_per_lane (a power of two)
        // keeps the k-th select that follows short.
        const uint32_t gr = a.gmin_per_lane;
#pragma unroll
        for (int g = 0; g < (MODE == 0 ? GPW : 1); g++) {
#pragma unroll
            for (int sft = 8; sft >= 1; sft >>= 1)
                if (gr <= (uint32_t)sft) {
#pragma unroll
                    for (int r = 0; r < sft; r++) mins[g][r] = fminf(mins[g][r], mins[g][r + sft]);
                }
            float *o = a.gmin + (size_t)myq[g] * a.groups_per_query + (size_t)((sid * RT + rt) * 2 + h) * gr;
#pragma unroll
            for (int r = 0; r < 16; r++)
                if ((uint32_t)r < gr) o[r] = mins[g][r];
        }
    }
}

// ---- per-TU dispatch helpers
template <int DT, int KS, int QG, int METRIC, int MODE>
static hipError_t scan_launch_one(const ScanK &k, hipStream_t s) {
    static std::atomic<bool> configured{false};
    constexpr int lds = Geo<QG, KS, MODE>::LDS_BYTES;
    if (!configured.load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void *)k_scan<DT, KS, QG, METRIC, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        configured.store(true, std::memory_order_release);
    }
    PVS_SCAN_LAUNCH((k_scan<DT, KS, QG, METRIC, MODE>), dim3(k.grid), dim3(Geo<QG, KS, MODE>::WAVES * 64), lds, s, k);
    return hipGetLastError();
}
template <int DT, int KS, int QG>
static hipError_t scan_launch_mm(const ScanK &k, int metric, int mode, hipStream_t s) {
    if constexpr (DT == PVS_I8) {
        if (mode == 4 || mode == 5) return hipErrorInvalidValue;  // (int8 distances have a closed form: MODE 2 / 3)
        if (mode == 2)
            return metric == PVS_COSINE ? scan_launch_one<DT, KS, QG, PVS_COSINE, 2>(k, s) : scan_launch_one<DT, KS, QG, PVS_L2, 2>(k, s);
        if (mode == 3)
            return metric == PVS_COSINE ? scan_launch_one<DT, KS, QG, PVS_COSINE, 3>(k, s) : scan_launch_one<DT, KS, QG, PVS_L2, 3>(k, s);
    } else {
        if (mode == 2 || mode == 3) return hipErrorInvalidValue;  // float order matters: no closed form
        if (mode == 4) return metric == PVS_COSINE ? scan_launch_one<DT, KS, QG, PVS_COSINE, 4>(k, s) : scan_launch_one<DT, KS, QG, PVS_L2, 4>(k, s);
        if (mode == 5) {
            // (the widest pitches keep 384 VGPRs of query fragments: 64 bracket ends on top of them spill — pvs_scan_fold5_supported
            //  sends those shapes through MODE 4 and the second kernel; tools/check_scratch.py)
            if constexpr (KS * steps_per_slab<DT>() <= 64)
                return metric == PVS_COSINE ? scan_launch_one<DT, KS, QG, PVS_COSINE, 5>(k, s) : scan_launch_one<DT, KS, QG, PVS_L2, 5>(k, s);
            else
                return hipErrorInvalidValue;
        }
    }
    if (metric == PVS_COSINE) return mode == 0 ? scan_launch_one<DT, KS, QG, PVS_COSINE, 0>(k, s) : scan_launch_one<DT, KS, QG, PVS_COSINE, 1>(k, s);
    return mode == 0 ? scan_launch_one<DT, KS, QG, PVS_L2, 0>(k, s) : scan_launch_one<DT, KS, QG, PVS_L2, 1>(k, s);
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
