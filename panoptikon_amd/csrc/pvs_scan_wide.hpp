// pvs_scan_wide.hpp — the 256- and 128-query int8 filter scans (gfx950): passes A and B of DESIGN.md §4.1 for row pitches up to
// 1 KiB (256 queries) / 768 B (128 queries).
//
// Replaces, for 256 queries at once, the reference's per-row vec_distance_{cosine,L2}(payload, ?) + ORDER BY ... LIMIT k
// (filters/image_embeddings.rs:321-362, text_embeddings.rs:386-418, pql/builder.rs:578-582) — the same contract as k_scan
// (pvs_scan_kernel.hpp): pass A = group minima of an upper bound of the key over a strided sample of tiles, pass B = every
// row whose lower bound is at or below the threshold goes to a candidate segment owned by one lane.  Same HBM layout, same
// ScanK arguments; pass C does not know which kernel produced its candidates.
//
// What bounds this pass is ENERGY.  At 256 queries x 10M x 768 the part sits at its 1,400 W limit and lowers its clock until
// the work fits; ablations of this kernel add up instead of overlapping (tools/probe/wide_probe.hip, profiles/r03_*):
// matrix cores 0.92 ms-equivalents, HBM + LDS-DMA transport 0.52, A-fragment reads from LDS 0.19, epilogue VALU 0.16 — sum
// 1.78 against 1.75 for the full kernel (the first form of this kernel, with v_mfma_i32_32x32x32_i8: 1.14 + 0.46 + 0.17 + 0.15
// = 1.91).  Stalls are free (the clock rises), instructions are not.  So:
//   * v_mfma_i32_16x16x64_i8, not 32x32x32: on int8 codes the matrix pipe sustains 4.1 POP/s through the 16x16 shape and 3.4
//     through the 32x32 shape at the same board power (tools/probe/mfma_rate.hip) — K = 64 per instruction means a quarter of
//     the accumulator traffic per operation;
//   * a wave owns 32 queries (two 16-query B fragments per 64 bytes of k, resident in registers: 96 VGPRs at 768-B rows) and
//     ALL 64 rows of the workgroup tile (four 16-row A fragments per k step, each feeding two MFMAs): 96 MFMAs, 48 fragment
//     reads and ONE barrier per 64-row tile (two layout tiles = one contiguous 48 KiB of HBM, ring of 3);
//   * the pass-B test of the previous tile runs in the shadow of the current tile's MFMAs and costs ~40 VALU per wave-tile:
//     per (query, 32-row layout tile) a float bound from the tile's extreme row scalars (1-4 VALU), a v_max3 fold of the
//     lane's 8 sums (4 VALU), one compare.  Only wave-tiles where some lane passes (a necessary condition of the exact test)
//     look at row scalars, and a passing row is written with one predicated vector store: no lane loop, no scalar stores;
//   * one copy of the tile record per workgroup (wave 0 fetches it), incremental tile addresses.
// Stores and the counted vmcnt waits: gfx9 counts stores on vmcnt too and retires a wave's VMEM operations in order, so a
// store issued between LDS-DMA pieces makes a later counted wait cover at most that many pieces more than it needs — pieces
// issued a whole tile earlier.  The waits stay correct (never too few), the prefetch ring is not drained.
//
// Lane geometry (n = lane & 15, c = lane >> 4): A fragment of row group rg = row 16 rg + n, bytes 16 c .. 16 c + 15 of the
// k step; B fragment of query group q = query 16 q + n, same bytes; D block (rg, q): register v = row 16 rg + 4 c + v,
// query 16 q + n.  A lane therefore holds two queries x 16 rows per tile, its thresholds are lane-private registers, and
// candidate segment (stream, c) x query has exactly one writer.  A dot product is invariant under a common permutation of
// k, so only "A and B use the same 16-byte chunk per (c, k step)" matters, not the instruction's internal k order.
#pragma once
#include <cstdlib>
#include <type_traits>

#include "pvs_lds_dma.hpp"
#include "pvs_scan_dispatch.hpp"

typedef int wv4i __attribute__((ext_vector_type(4)));

// compile-time loop: f(std::integral_constant<int, I>) for I in [B, E) — the tile body below must be straight-line code with
// every index a constant (a "#pragma unroll" the optimizer declines turns register arrays into scratch)
template <int B, int E, typename F>
__device__ static inline __attribute__((always_inline)) void static_for(F &&f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}
#define PVS_CI(x) (decltype(x)::value)

// Eight waves = QW query blocks (16 NQ queries each) x RW row blocks.  RW = 1: every wave works on all rows of the tile, 256
// queries per pass.  RW = 2: a wave works on one of the tile's two 32-row layout tiles, 128 queries per pass — every A
// fragment is then read by four waves instead of eight.  (128 queries at 10M x 768, same probe: this form 1.1x ms; 8 waves x 16
// queries on all 64 rows 1.229; 4 waves x 32 queries, two workgroups per CU, 32-row tiles 1.268; k_scan 1.29-1.31.  With the
// matrix cores at 0.47 ms-equivalents the stream itself, 1.08 ms at 7.1 TB/s, is the floor: profiles/r03_wide_ablation.md.)
template <int KSLABS, int NQ, int RW>
struct WideGeo {
    static constexpr int WAVES = 8;
    static constexpr int QW = WAVES / RW;                 // query blocks; QW * 16 * NQ queries per pass
    static constexpr int RPW = KSLABS <= 3 ? 2 : 1;       // 32-row layout tiles per workgroup tile
    static constexpr int LTW = RPW / RW;                  // layout tiles per wave
    static constexpr int TILE_ROWS = 32 * RPW;
    static constexpr int RG = 2 * LTW;                    // 16-row A fragments per wave and k step
    static_assert(RPW % RW == 0 && LTW >= 1, "a wave works on whole layout tiles");
    static constexpr int SUB_BYTES = KSLABS * 8192;       // one 32-row layout tile
    static constexpr int TILE_BYTES = RPW * SUB_BYTES;    // contiguous in HBM and, byte for byte, in LDS
    static constexpr int PIECES = TILE_BYTES / 1024;      // 1-KiB LDS-DMA pieces per tile
    static constexpr int PPW = PIECES / WAVES;            // per wave (RPW * KSLABS)
    static constexpr int NC = (152 * 1024) / TILE_BYTES > 6 ? 6 : (152 * 1024) / TILE_BYTES;  // ring tiles
    static constexpr int PC = NC - 1;                     // tiles in flight
    static constexpr int NCN = PC + 2;                    // tile-record slots: in flight + consumed + previous (its epilogue)
    static constexpr int REC_SLOT = 1024;                 // one DMA piece: RPW records of 256 B (the rest is a duplicate)
    static constexpr int LDS_BYTES = NC * TILE_BYTES + NCN * REC_SLOT;
    static_assert(PIECES % WAVES == 0, "pieces divide over the waves");
    static_assert(NC >= 3, "two tiles in flight at least");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS per CU");
    static_assert((PC - 1) * (PPW + 1) <= 63, "vmcnt is a 6-bit counter");
};

template <int NQ>
struct WideCnt {  // fill counts of a lane's (segment, query) lists
    uint32_t k[NQ];
};

// MODE 0 = pass A (group minima), MODE 1 = pass B (candidates)
template <int KSLABS, int NQ, int RW, int METRIC, int MODE>
__global__ __launch_bounds__(512, 2) void k_scan_wide(ScanK a) {
    using G = WideGeo<KSLABS, NQ, RW>;
    constexpr int QPW = 16 * NQ;         // queries per wave
    constexpr int RPW = G::RPW, LTW = G::LTW, RG = G::RG, QW = G::QW, NC = G::NC, PC = G::PC, NCN = G::NCN, PPW = G::PPW;
    constexpr int NK = KSLABS * 4;       // k steps of 64 bytes
    constexpr int NG = NK * RG * NQ;     // MFMAs (= filler gaps) per tile
    constexpr bool COS = METRIC == PVS_COSINE;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *const ring = smem;                           // [NC][TILE_BYTES]
    uint8_t *const recring = smem + NC * G::TILE_BYTES;   // [NCN][REC_SLOT]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qw = wave % QW, rw = wave / QW;  // this wave's query block and row block
    const int n = lane & 15, c = lane >> 4;
    const uint32_t sid = blockIdx.x, nstreams = a.grid;
    const uint32_t ring_lds = lds_addr(ring), rec_lds = lds_addr(recring);
    if (tid < 256) ((float *)(recring + (NCN - 1) * G::REC_SLOT))[tid] = __builtin_nanf("");  // the record of tile "-1" (see p_nslot)

    // tiles of this workgroup: (sid + it * nstreams) * tile_step, in units of workgroup tiles
    const uint32_t n_samp = (a.n_wgtiles + a.tile_step - 1) / a.tile_step;
    const int n_my = (sid < n_samp && sid < nstreams) ? (int)((n_samp - sid + nstreams - 1) / nstreams) : 0;

    constexpr int NMIN = MODE == 0 ? 8 : 1;
    float mins[NQ][NMIN];  // pass A: per query 8 minima, one per (row group parity, v): 8 disjoint row groups
#pragma unroll
    for (int q = 0; q < NQ; q++)
#pragma unroll
        for (int r = 0; r < NMIN; r++) mins[q][r] = __builtin_inff();
    const uint32_t seg = (sid * RW + (uint32_t)rw) * PVS_WIDE_SEG_PER_STREAM + (uint32_t)c;  // this lane's segment (all of its queries; no other writer)
    WideCnt<NQ> cnt;
#pragma unroll
    for (int q = 0; q < NQ; q++) cnt.k[q] = 0;

    if (n_my > 0) {
        // ---- query fragments: resident for the whole kernel
        wv4i qf[NK][NQ];
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const uint8_t *qrow = a.qmat + (size_t)(qw * QPW + 16 * q + n) * a.stride;
#pragma unroll
            for (int i = 0; i < NK; i++) qf[i][q] = *(const wv4i *)(qrow + (4 * i + c) * 16);
        }
        QInfo qi[NQ];
        float thr[NQ];
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            qi[q] = a.qinfo[qw * QPW + 16 * q + n];
            thr[q] = MODE == 1 ? a.thr[qw * QPW + 16 * q + n] : 0.f;
        }
        // retire the compiler's own loads here (it cannot see the asm waits below)
#pragma unroll
        for (int i = 0; i < NK; i++)
#pragma unroll
            for (int q = 0; q < NQ; q++) asm volatile("" : "+v"(qf[i][q]));
#pragma unroll
        for (int q = 0; q < NQ; q++) asm volatile("" : "+v"(qi[q].bb), "+v"(qi[q].dscale), "+v"(qi[q].eA), "+v"(qi[q].eR), "+v"(thr[q]));
        wait_vm<0>();
        // Filter test folded into per-lane constants (key / err algebra of HISTORY.md §4.2, as in k_scan):
        //   cosine  pass iff d * (1/|a|) >= tS              tS = -(thr + eA) / dscale
        //   L2      pass iff c1 |a|^2 + m2d d <= tS         tS = thr + eA - bb, c1 = 1 - eR, m2d = -2 dscale
        // and the necessary condition on d alone from the tile's extreme row scalars (t0 = min, t1 = max of |a| resp. |a|^2):
        //   cosine  d >= tS * (tS > 0 ? t0 : t1)            L2   d >= (c1 t0 - tS) / (2 dscale)
        // minus a slack of 2^-18 relative (30x the f32 roundings of the exact test) and 1:
        //   cosine  bound = tSe * t - 1,  tSe = tS (1 -+ 2^-18), t chosen by the sign of tS
        //   L2      bound = (x hd - tShd) - (|x| hde + tSae) - 1,  x = c1 t0
        // A NaN bound (no usable row in the tile, padding query) compares false: nothing passes.
        float c1[NQ], m2d[NQ], tS[NQ], tSe[NQ], hd[NQ], tShd[NQ], hde[NQ], tSae[NQ];
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const bool live = qi[q].dscale > 0.f;  // padding queries: dscale = 0, never pass
            c1[q] = 1.0f - qi[q].eR;
            m2d[q] = -2.0f * qi[q].dscale;
            hd[q] = live ? 0.5f / qi[q].dscale : 0.f;
            if (COS) {
                tS[q] = live ? -(thr[q] + qi[q].eA) / qi[q].dscale : __builtin_inff();
                tSe[q] = tS[q] - fabsf(tS[q]) * 3.8147e-6f;  // (an infinite tS gives NaN: nothing passes)
            } else {
                tS[q] = live ? thr[q] + qi[q].eA - qi[q].bb : -__builtin_inff();
                tSe[q] = 0.f;
            }
            tShd[q] = live ? tS[q] * hd[q] : -__builtin_inff();
            hde[q] = hd[q] * 3.8147e-6f;
            tSae[q] = live ? fabsf(tS[q]) * hde[q] : 0.f;
        }
        auto score = [&](int q, float d, float x) __attribute__((always_inline)) { return COS ? d * x : __builtin_fmaf(d, m2d[q], c1[q] * x); };
        auto passes = [&](int q, float sv) __attribute__((always_inline)) { return COS ? sv >= tS[q] : sv <= tS[q]; };
        // MODE 1: this lane's slots for its first query; query group q (= +16 q queries) sits 16 q seg_cap slots further
        uint2 *const seg_lane = a.seg + ((size_t)seg * a.seg_queries + (uint32_t)(qw * QPW + n)) * a.seg_cap;
        const uint32_t seg_q1 = 16u * a.seg_cap;

        // ---- LDS-DMA producer state: PC tiles ahead of the consumer
        const uint32_t voff = (uint32_t)lane * 16u;
        const uint32_t recvoff = (uint32_t)(lane & 31) * 16u;  // 512 B of records per tile at most; the upper lanes re-read them
        const uint64_t tile_stride = (uint64_t)nstreams * a.tile_step * G::TILE_BYTES;
        const uint8_t *src = a.rows + (uint64_t)sid * a.tile_step * G::TILE_BYTES + (uint32_t)wave * (PPW * 1024u);
        const uint8_t *srec = (const uint8_t *)(a.aux + (uint64_t)sid * a.tile_step * (RPW * PVS_AUX_REC));
        const uint64_t rec_stride = (uint64_t)nstreams * a.tile_step * (RPW * PVS_AUX_REC * 4);
        int i_tl = 0;
        uint32_t i_dst = ring_lds + (uint32_t)wave * (PPW * 1024u), i_rec = rec_lds;
        int i_slot = 0, i_nslot = 0;
        const uint8_t *is_src = nullptr, *is_srec = nullptr;
        uint32_t is_dst = 0, is_rec = 0;
        auto uni64 = [](const void *p) __attribute__((always_inline)) {  // pin a wave-uniform pointer in SGPRs (the asm's "s" operands)
            const uint64_t v = (uint64_t)(uintptr_t)p;
            const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
            return (const uint8_t *)(uintptr_t)(((uint64_t)hi << 32) | lo);
        };
        auto issue_begin = [&]() __attribute__((always_inline)) {
            is_src = uni64(src);
            is_srec = uni64(srec);
            is_dst = i_dst;
            is_rec = i_rec;
            if (i_tl + 1 < n_my) {  // past the end: the last tile again (keeps vmcnt uniform; its sums are never looked at)
                src += tile_stride;
                srec += rec_stride;
            }
            i_tl++;
            if (++i_slot == NC) {
                i_slot = 0;
                i_dst = ring_lds + (uint32_t)wave * (PPW * 1024u);
            } else {
                i_dst += G::TILE_BYTES;
            }
            if (++i_nslot == NCN) {
                i_nslot = 0;
                i_rec = rec_lds;
            } else {
                i_rec += G::REC_SLOT;
            }
        };
        auto issue_part = [&](int part) __attribute__((always_inline)) {  // compile-time part: 0..PPW-1 = this wave's row pieces, PPW = the tile record (wave 0)
            if (part < PPW) {  // the wave's pieces are contiguous at both ends: one M0 write per 4 KiB (the immediate offset moves source and destination)
                if (part % 4 == 0) {
                    if (PPW - part >= 4)
                        dma16_group<4>(is_src + part * 1024, voff, is_dst + part * 1024);
                    else if (PPW - part == 3)
                        dma16_group<3>(is_src + part * 1024, voff, is_dst + part * 1024);
                    else if (PPW - part == 2)
                        dma16_group<2>(is_src + part * 1024, voff, is_dst + part * 1024);
                    else
                        dma16_group<1>(is_src + part * 1024, voff, is_dst + part * 1024);
                }
            } else if (wave == 0)
                dma16(is_srec, recvoff, is_rec);
        };
#pragma unroll
        for (int p = 0; p < PC; p++) {
            issue_begin();
#pragma unroll
            for (int part = 0; part <= PPW; part++) issue_part(part);
        }

        // ---- consumer state
        int c_slot = 0, c_nslot = 0, p_nslot = NCN - 1;  // ring slot / record slot of the tile being consumed; record slot of the previous
                                                        // tile (tile "-1": the last slot, preset to NaN — nothing passes, no minimum moves)
        uint32_t prev_row_base = 0;                 // row 4 c of the previous tile (the lane's first row in row group 0)
        uint32_t wt_cur = sid * a.tile_step;        // workgroup tile being consumed
        const uint32_t wt_step = nstreams * a.tile_step;
        // A-fragment LDS addresses: row 16 (rg & 1) + n of layout tile rg >> 1, chunk (4 (i & 3) + c) ^ n of k-slab i >> 2: four
        // swizzled bases; row group, layout tile and k-slab ride in the ds_read immediate offset
        uint32_t swz[4];
#pragma unroll
        for (int i = 0; i < 4; i++) swz[i] = (uint32_t)n * 256u + ((((uint32_t)(4 * i + c)) ^ (uint32_t)n) << 4);

        // the lane's 8 row scalars of layout tile s (rows 16 r2 + 4 c + v), from the previous tile's record
        auto load_xs = [&](int s, float(&xs)[8]) __attribute__((always_inline)) {
            const float *rec = (const float *)(recring + p_nslot * G::REC_SLOT) + (rw * LTW + s) * PVS_AUX_REC;
#pragma unroll
            for (int r2 = 0; r2 < 2; r2++) {
                const float4 v = *(const float4 *)(rec + 16 * r2 + 4 * c);
                xs[4 * r2 + 0] = v.x;
                xs[4 * r2 + 1] = v.y;
                xs[4 * r2 + 2] = v.z;
                xs[4 * r2 + 3] = v.w;
            }
        };
        // per-row path of pass B for (query q, layout tile s), taken by ~8 % of the (wave, q, s) combinations at k = 100: each of
        // the lane's 8 sums against the tile bound first (one integer compare, no row scalar), the exact test only where that
        // passes, a passing row -> one predicated store
        auto emit_rows = [&](int q, int s, float bnd, uint32_t k, auto &&pv) __attribute__((always_inline)) {
            float xs[8];
            load_xs(s, xs);
            // d >= bnd  <=>  d >= ceil(bnd) for an integer d; sums are below 2^24 in magnitude, a NaN bound passes nothing
            const int ebi = bnd == bnd ? (int)ceilf(fminf(fmaxf(bnd, -1.0e9f), 1.0e9f)) : 0x7fffffff;
            uint2 *const dst = seg_lane + (uint32_t)q * seg_q1;
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int d = pv(2 * s + (r >> 2), q, r & 3);
                if (__builtin_amdgcn_ballot_w64(d >= ebi) == 0) continue;
                const bool p = passes(q, score(q, (float)d, xs[r]));
                if (p) {
                    const uint2 cand = make_uint2(prev_row_base + (uint32_t)(32 * s + 16 * (r >> 2) + (r & 3)), (uint32_t)d);
                    if (a.flat) {  // rerun after a segment overflow: one list per query, slots handed out by an atomic counter
                        const uint32_t qg = (uint32_t)(qw * QPW + 16 * q + n);
                        const uint32_t pos = atomicAdd(a.flat_cnt + qg, 1u);
                        if (pos < a.flat_cap) a.flat[(size_t)qg * a.flat_cap + pos] = cand;
                    } else {
                        if (k < a.seg_cap) dst[k] = cand;
                        k++;
                    }
                }
            }
            return k;
        };

        struct Epi {
            float t[LTW][2];                   // per layout tile of the wave: min / max row scalar
            float bnd[NQ][LTW];                // per (query, layout tile): the bound on d
            int mx[NQ][LTW];                   // running v_max3 fold
            unsigned long long hit[NQ][LTW];   // lanes whose fold clears the bound
            float xs[8];                       // pass A: row scalars of the layout tile being scored
        };
        // Epilogue slices of the previous tile (compile-time m); pv(rg, q, v) = its sum for row 16 rg + 4 c + v, query 16 q + n.
        //   pass B: 0 = read the extremes; combo x = LTW q + s at 1 + 5 x: bound + first fold, +1..+3 = folds, +4 = compare;
        //           M_DEC = decisions.  pass A: per layout tile s: 10 s = read row scalars, 10 s + 1 .. + 8 = one row x NQ queries each.
        constexpr int M_DEC = 1 + 5 * NQ * LTW;
        constexpr int EPI_STEPS = MODE == 1 ? M_DEC + 1 : 10 * LTW;
        auto epi_slice = [&](auto mc, Epi &e, WideCnt<NQ> k, auto &&pv) __attribute__((always_inline)) {
            constexpr int m = PVS_CI(mc);
            if constexpr (MODE == 1) {
                if constexpr (m == 0) {
#pragma unroll
                    for (int s = 0; s < LTW; s++) {
                        const float2 tmm = *(const float2 *)((const float *)(recring + p_nslot * G::REC_SLOT) + (rw * LTW + s) * PVS_AUX_REC + 32);
                        e.t[s][0] = tmm.x;
                        e.t[s][1] = tmm.y;
                    }
                } else if constexpr (m == M_DEC) {
                    // wave-uniform decisions: per (query, layout tile), only if some lane's fold cleared its bound
                    unsigned long long any = 0;
#pragma unroll
                    for (int q = 0; q < NQ; q++)
#pragma unroll
                        for (int s = 0; s < LTW; s++) any |= e.hit[q][s];
                    if (any != 0) {
#pragma unroll
                        for (int s = 0; s < LTW; s++)
#pragma unroll
                            for (int q = 0; q < NQ; q++)
                                if (e.hit[q][s] != 0) k.k[q] = emit_rows(q, s, e.bnd[q][s], k.k[q], pv);
                    }
                } else {
                    constexpr int x = (m - 1) / 5, st = (m - 1) % 5, q = x / LTW, s = x % LTW;
                    {
                        if constexpr (st == 0) {
                            if constexpr (x == 0) {
#pragma unroll
                                for (int ss = 0; ss < LTW; ss++) asm volatile("" : "+v"(e.t[ss][0]), "+v"(e.t[ss][1]));  // (slice 0's LDS reads are waited for here)
                            }
                            if (COS) {
                                e.bnd[q][s] = __builtin_fmaf(tSe[q], tS[q] > 0.f ? e.t[s][0] : e.t[s][1], -1.0f);
                            } else {
                                const float xx = c1[q] * e.t[s][0];
                                e.bnd[q][s] = (__builtin_fmaf(xx, hd[q], -tShd[q]) - __builtin_fmaf(fabsf(xx), hde[q], tSae[q])) - 1.0f;
                            }
                            e.mx[q][s] = max(pv(2 * s, q, 0), max(pv(2 * s, q, 1), pv(2 * s, q, 2)));
                        } else if constexpr (st == 1) {
                            e.mx[q][s] = max(e.mx[q][s], max(pv(2 * s, q, 3), pv(2 * s + 1, q, 0)));
                        } else if constexpr (st == 2) {
                            e.mx[q][s] = max(e.mx[q][s], max(pv(2 * s + 1, q, 1), pv(2 * s + 1, q, 2)));
                        } else if constexpr (st == 3) {
                            e.mx[q][s] = max(e.mx[q][s], pv(2 * s + 1, q, 3));
                        } else {
                            e.hit[q][s] = __builtin_amdgcn_ballot_w64((float)e.mx[q][s] >= e.bnd[q][s]);
                        }
                        if constexpr (st < 4) asm volatile("" : "+v"(e.mx[q][s]));  // keep the slice where it is (the optimizer would sink the fold behind the last MFMA)
                    }
                }
            } else {
                constexpr int s = m / 10, st = m % 10;
                if constexpr (st == 0) {
                    load_xs(s, e.xs);
                } else if constexpr (st <= 8) {
                    // upper bound of the key of one row for the lane's queries: key + err (NaN — padding, zero norm, masked — never wins a fmin)
                    constexpr int r = st - 1;  // row 16 (r >> 2) + 4 c + (r & 3) of layout tile s
#pragma unroll
                    for (int q = 0; q < NQ; q++) {
                        const float sv = score(q, (float)pv(2 * s + (r >> 2), q, r & 3), e.xs[r]);
                        const float ub = COS ? __builtin_fmaf(-sv, qi[q].dscale, qi[q].eA) : sv + (qi[q].bb + qi[q].eA) + 2.0f * qi[q].eR * e.xs[r];
                        mins[q][MODE == 0 ? r : 0] = fminf(mins[q][MODE == 0 ? r : 0], ub);
                    }
                }
            }
            return k;
        };

        // One tile: NG MFMAs with the LDS-DMA pieces of the tile PC ahead and the previous tile's epilogue slices between them.
        // Accumulators alternate between two register sets; the previous tile's sums are read where the matrix core left them.
        auto run_tile = [&](wv4i(&acc)[RG][NQ], WideCnt<NQ> k, auto &&pv) __attribute__((always_inline)) {
            if (wave == 0)
                wait_vm<(PC - 1) * (PPW + 1)>();
            else
                wait_vm<(PC - 1) * PPW>();
            wg_barrier();
            issue_begin();  // refills the slot the previous tile occupied
            const uint8_t *cb = ring + c_slot * G::TILE_BYTES + rw * (LTW * G::SUB_BYTES);  // this wave's layout tile(s) of the slot
            const uint8_t *fb[4];
#pragma unroll
            for (int i = 0; i < 4; i++) fb[i] = cb + swz[i];
            Epi e;
#pragma unroll
            for (int q = 0; q < NQ; q++)
#pragma unroll
                for (int s = 0; s < LTW; s++) e.hit[q][s] = 0;
            wv4i af[NK][RG];
            auto frag = [&](int i, int rg) __attribute__((always_inline)) {  // k step i, row group rg
                af[i][rg] = *(const wv4i *)(fb[i & 3] + (rg >> 1) * G::SUB_BYTES + (i >> 2) * 8192 + (rg & 1) * 4096);
            };
#pragma unroll
            for (int rg = 0; rg < RG; rg++) frag(0, rg);
            __builtin_amdgcn_sched_barrier(0);
            const wv4i zero = {0, 0, 0, 0};
            static_for<0, NK>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = PVS_CI(ic);
                static_for<0, RG>([&](auto rc) __attribute__((always_inline)) {
                    constexpr int rg = PVS_CI(rc);
                    static_for<0, NQ>([&](auto qc) __attribute__((always_inline)) {
                        constexpr int q = PVS_CI(qc);
                        constexpr int g = (i * RG + rg) * NQ + PVS_CI(qc);  // MFMA gap index, 0 .. NG-1
                        if constexpr (i == 0)
                            acc[rg][q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[i][rg], qf[i][q], zero, 0, 0, 0);
                        else
                            acc[rg][q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[i][rg], qf[i][q], acc[rg][q], 0, 0, 0);
                        if constexpr (PVS_CI(qc) == 0 && i + 1 < NK) frag(i + 1, rg);  // the next k step's fragment of this row group
                        static_for<g * (PPW + 1) / NG, (g + 1) * (PPW + 1) / NG>([&](auto pc) __attribute__((always_inline)) { issue_part(PVS_CI(pc)); });
                        static_for<g * EPI_STEPS / NG, (g + 1) * EPI_STEPS / NG>(
                            [&](auto mc) __attribute__((always_inline)) { k = epi_slice(mc, e, k, pv); });
                        __builtin_amdgcn_sched_barrier(0);
                    });
                });
            });
            if (++c_slot == NC) c_slot = 0;
            p_nslot = c_nslot;
            if (++c_nslot == NCN) c_nslot = 0;
            prev_row_base = wt_cur * (uint32_t)G::TILE_ROWS + (uint32_t)(rw * (32 * LTW)) + 4u * (uint32_t)c;
            wt_cur += wt_step;
            return k;
        };
        // The last tile's epilogue runs inside one extra "ghost" tile (the DMA stream re-reads the last tile past the end).
        {
            wv4i accA[RG][NQ], accB[RG][NQ];
#pragma unroll
            for (int rg = 0; rg < RG; rg++)
#pragma unroll
                for (int q = 0; q < NQ; q++) accB[rg][q] = wv4i{0, 0, 0, 0};  // tile "-1"
            auto pa = [&](int rg, int q, int v) __attribute__((always_inline)) { return accA[rg][q][v]; };
            auto pb = [&](int rg, int q, int v) __attribute__((always_inline)) { return accB[rg][q][v]; };
            for (int tl = 0; tl < n_my + 1; tl += 2) {
                cnt = run_tile(accA, cnt, pb);
                if (tl + 1 < n_my + 1) cnt = run_tile(accB, cnt, pa);
            }
        }
        wait_vm<0>();  // retire the tail DMAs (and the candidate stores) before the wave exits
    }
    if constexpr (MODE == 1) {
        if (sid < nstreams) {  // every lane's fill counts (above seg_cap: overflowed)
#pragma unroll
            for (int q = 0; q < NQ; q++) a.seg_cnt[(size_t)(qw * QPW + 16 * q + n) * a.seg_stride + seg] = cnt.k[q];
        }
    } else {
        if (sid < nstreams) {
            const uint32_t gr = a.gmin_per_lane;  // fold the 8 minima of a (lane, query) to gmin_per_lane (a power of two <= 8)
#pragma unroll
            for (int q = 0; q < NQ; q++) {
#pragma unroll
                for (int sft = 4; sft >= 1; sft >>= 1)
                    if (gr <= (uint32_t)sft) {
#pragma unroll
                        for (int r = 0; r < sft; r++) mins[q][r] = fminf(mins[q][r], mins[q][r + sft]);
                    }
                float *o = a.gmin + (size_t)(qw * QPW + 16 * q + n) * a.groups_per_query + (size_t)seg * gr;
#pragma unroll
                for (int r = 0; r < 8; r++)
                    if ((uint32_t)r < gr) o[r] = mins[q][r];
            }
        }
    }
}

// NQ = 2: 32 queries per wave, two waves per SIMD.  (NQ = 4 — four waves x 64 queries, one per SIMD, ~400 registers, half the
// A-fragment reads — measured 1.86 ms against 1.70 at 256 queries: the fragment reads do drop from 0.19 to 0.10 ms-equivalents,
// but the per-row path, now alone on its SIMD, doubles; profiles/r03_wide_ablation.md.)
constexpr int PVS_WIDE_NQ = 2;
template <int KS, int RW, int METRIC, int MODE>
static hipError_t scan_wide_launch_one(const ScanK &k, hipStream_t s) {
    constexpr int NQ = PVS_WIDE_NQ;
    static std::atomic<bool> configured{false};
    constexpr int lds = WideGeo<KS, NQ, RW>::LDS_BYTES;
    if (!configured.load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void *)k_scan_wide<KS, NQ, RW, METRIC, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        configured.store(true, std::memory_order_release);
    }
    PVS_SCAN_LAUNCH((k_scan_wide<KS, NQ, RW, METRIC, MODE>), dim3(k.grid), dim3(512), lds, s, k);
    return hipGetLastError();
}
template <int KS, int RW>
static hipError_t scan_wide_launch(const ScanK &k, int metric, int mode, hipStream_t s) {
    if (mode != 0 && mode != 1) return hipErrorInvalidValue;
    if (metric == PVS_COSINE) return mode == 0 ? scan_wide_launch_one<KS, RW, PVS_COSINE, 0>(k, s) : scan_wide_launch_one<KS, RW, PVS_COSINE, 1>(k, s);
    return mode == 0 ? scan_wide_launch_one<KS, RW, PVS_L2, 0>(k, s) : scan_wide_launch_one<KS, RW, PVS_L2, 1>(k, s);
}
