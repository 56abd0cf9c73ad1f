// pvs_scan_wide.hpp — the 256-query int8 filter scan (gfx950): passes A and B of DESIGN.md §4.1 for row pitches up to 1 KiB.
//
// Replaces, for 256 queries at once, the reference's per-row vec_distance_{cosine,L2}(payload, ?) + ORDER BY ... LIMIT k
// (filters/image_embeddings.rs:321-362, text_embeddings.rs:386-418, pql/builder.rs:578-582) — the same contract as k_scan
// (pvs_scan_kernel.hpp): pass A = group minima of an upper bound of the key over a strided sample of tiles, pass B = every
// row whose lower bound is at or below the threshold goes to its (stream, half-wave, query) segment.  Same HBM layout, same
// ScanK arguments, same outputs; pass C does not know which kernel produced its candidates.
//
// Why a kernel of its own.  At 256 queries the pass is matrix-pipe AND HBM bound at once (10M x 768: 1,536 matrix-pipe cycles
// per SIMD and 32-row tile against ~1,580 cycles of HBM time), so everything a wave does alone — loop top, barrier skew,
// epilogue rest, a candidate to emit — is paid by the seven waves waiting for it at the per-tile barrier.  Round 2's form (8
// waves x 32 queries on 32-row tiles inside k_scan) spent 2,320-2,870 cycles per tile.  This kernel changes three things:
//   * 64-row workgroup tiles (two layout tiles = one contiguous 48 KiB of HBM at 768-B rows, ring of 3): every wave keeps its
//     32 queries' B fragments in registers and runs TWO independent accumulation chains (rows 0-31, rows 32-63) per barrier —
//     48 MFMAs between barriers instead of 24, and a wave that has the matrix pipe to itself can issue back to back;
//   * the pass-B test of the previous tile runs entirely in the shadow of the current tile's MFMAs and is branch-free up to the
//     (wave-uniform) decision to store: the 16 sums of a lane are packed as (sum << 4 | slot) and folded to their TOP TWO with
//     v_max3 / v_med3; the best row is tested exactly (its row scalar comes from the tile record in LDS) and, when it passes,
//     written with ONE predicated vector store; only when the second best also clears the tile bound — two candidates in one
//     lane and tile: ties, clustered data — does the wave take the per-row path.  No lane loop, no scalar stores, nothing left
//     behind the last MFMA of a tile;
//   * one copy of the tile record per workgroup (wave 0 fetches it) and incremental tile addresses: the loop top is a counted
//     wait and the barrier.
// Stores and the counted vmcnt waits: gfx9 counts stores on vmcnt too and retires VMEM operations of a wave in order, so a
// store issued between LDS-DMA pieces makes a later counted wait cover at most that many pieces more than it needs — pieces
// issued a whole tile earlier.  The waits stay correct (never too few), the prefetch ring is not drained.
#pragma once
#include <cstdlib>

#include "pvs_lds_dma.hpp"
#include "pvs_scan_dispatch.hpp"

#include <type_traits>

typedef int wv4i __attribute__((ext_vector_type(4)));
typedef int wv16i __attribute__((ext_vector_type(16)));

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

template <int KSLABS>
struct WideGeo {
    static constexpr int WAVES = 8;                       // one query group of 32 per wave
    static constexpr int RPW = KSLABS <= 3 ? 2 : 1;       // 32-row sub-tiles per wave and barrier
    static constexpr int TILE_ROWS = 32 * RPW;
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

// MODE 0 = pass A (group minima), MODE 1 = pass B (candidates)
template <int KSLABS, int METRIC, int MODE>
__global__ __launch_bounds__(512, 2) void k_scan_wide(ScanK a) {
    using G = WideGeo<KSLABS>;
    constexpr int RPW = G::RPW, NC = G::NC, PC = G::PC, NCN = G::NCN, PPW = G::PPW, NF = KSLABS * 8, NG = NF * RPW;
    constexpr bool COS = METRIC == PVS_COSINE;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *const ring = smem;                           // [NC][TILE_BYTES]
    uint8_t *const recring = smem + NC * G::TILE_BYTES;   // [NCN][REC_SLOT]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;  // j: query column of the wave's group (B operand / C column) and A row; h: k half / row half
    const uint32_t sid = blockIdx.x, nstreams = a.grid;
    const int myq = wave * 32 + j;
    const uint32_t ring_lds = lds_addr(ring), rec_lds = lds_addr(recring);
    if (tid < 256) ((float *)(recring + (NCN - 1) * G::REC_SLOT))[tid] = __builtin_nanf("");  // the record of tile "-1" (see p_nslot)

    // tiles of this workgroup: (sid + it * nstreams) * tile_step, in units of 64-row (RPW = 2) workgroup tiles
    const uint32_t n_samp = (a.n_wgtiles + a.tile_step - 1) / a.tile_step;
    const int n_my = (sid < n_samp && sid < nstreams) ? (int)((n_samp - sid + nstreams - 1) / nstreams) : 0;

    float mins[MODE == 0 ? 16 : 1];
#pragma unroll
    for (int r = 0; r < (MODE == 0 ? 16 : 1); r++) mins[r] = __builtin_inff();
    const uint32_t seg = sid * 2 + h;  // this lane's segment (written by no other lane)
    uint32_t mycnt = 0;

    if (n_my > 0) {
        // ---- query fragments: resident for the whole kernel
        wv4i qf[NF];
        {
            const uint8_t *qrow = a.qmat + (size_t)myq * a.stride;
#pragma unroll
            for (int x = 0; x < NF; x++) qf[x] = *(const wv4i *)(qrow + (x * 2 + h) * 16);
        }
        QInfo qi = a.qinfo[myq];
        float thr = MODE == 1 ? a.thr[myq] : 0.f;
        // retire the compiler's own loads here (it cannot see the asm waits below)
#pragma unroll
        for (int x = 0; x < NF; x++) asm volatile("" : "+v"(qf[x]));
        asm volatile("" : "+v"(qi.bb), "+v"(qi.dscale), "+v"(qi.eA), "+v"(qi.eR), "+v"(thr));
        wait_vm<0>();
        // filter test folded into per-lane constants (key / err algebra of DESIGN.md §4.2, as in k_scan)
        const float c1 = 1.0f - qi.eR, m2d = -2.0f * qi.dscale, hd = qi.dscale > 0.f ? 0.5f / qi.dscale : 0.f;
        const float tS = COS ? (qi.dscale > 0.f ? -(thr + qi.eA) / qi.dscale : __builtin_inff())
                             : (qi.dscale > 0.f ? thr + qi.eA - qi.bb : -__builtin_inff());
        auto score = [&](float d, float x) __attribute__((always_inline)) { return COS ? d * x : __builtin_fmaf(d, m2d, c1 * x); };
        auto passes = [&](float sv) __attribute__((always_inline)) { return COS ? sv >= tS : sv <= tS; };
        uint2 *const seg_lane = a.seg + ((size_t)seg * a.seg_queries + (uint32_t)myq) * a.seg_cap;  // MODE 1: this lane's slots

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
            if (part < PPW)
                dma16(is_src + part * 1024, voff, is_dst + part * 1024);
            else if (wave == 0)
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
        uint32_t prev_row_base = 0;                 // first row of the previous tile's rows of this lane (sub-tile 0)
        uint32_t wt_cur = sid * a.tile_step;        // workgroup tile being consumed
        const uint32_t wt_step = nstreams * a.tile_step;
        // A-fragment LDS addresses: 8 swizzled chunk positions of row j, the k-slab and the sub-tile ride in the immediate offset
        uint32_t swz[8];
#pragma unroll
        for (int i = 0; i < 8; i++) swz[i] = (uint32_t)j * 256u + ((((uint32_t)(2 * i + h)) ^ (uint32_t)(j & 15)) << 4);

        // per-row path (pass B: rare): every row of sub-tile s whose sum clears the tile bound gets the exact test
        auto emit_rows = [&](int s, int eb, uint32_t cnt, auto &&pv) __attribute__((always_inline)) {
            const float *rec = (const float *)(recring + p_nslot * G::REC_SLOT) + s * PVS_AUX_REC;
            float xh[16];
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++) {
                const float4 v = *(const float4 *)(rec + 8 * g4 + 4 * h);
                xh[4 * g4 + 0] = v.x;
                xh[4 * g4 + 1] = v.y;
                xh[4 * g4 + 2] = v.z;
                xh[4 * g4 + 3] = v.w;
            }
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int d = pv(s, r);
                const bool c = d >= eb;
                if (__builtin_amdgcn_ballot_w64(c) == 0) continue;
                const bool p = c && passes(score((float)d, xh[r]));
                if (p) {
                    if (cnt < a.seg_cap) seg_lane[cnt] = make_uint2(prev_row_base + (uint32_t)(s * 32 + (r & 3) + 8 * (r >> 2)), (uint32_t)d);
                    cnt++;
                }
            }
            return cnt;
        };

        struct Epi {   // pass B, per sub-tile
            float t0, t1, x;
            int eb, ebk, m1, m2;
        };
        struct EpiA {  // pass A
            float xh[8];
        };
        constexpr int EPI_STEPS = MODE == 1 ? 13 : 10;  // per sub-tile
        // slice m of the previous tile's epilogue for sub-tile s (compile-time m, s); pv(s, r) = its r-th sum
        auto epi_slice = [&](auto sc, auto mc, Epi &e, EpiA &ea, uint32_t cnt, auto &&pv) __attribute__((always_inline)) {
            constexpr int s = PVS_CI(sc), m = PVS_CI(mc);
            if constexpr (MODE == 1) {
                if constexpr (m == 0) {  // the tile's extreme row scalars (k_scan_aux: min / max over the rows with a usable norm); tile "-1": NaN
                    const float2 tmm = *(const float2 *)((const float *)(recring + p_nslot * G::REC_SLOT) + s * PVS_AUX_REC + 32);
                    e.t0 = tmm.x;
                    e.t1 = tmm.y;
                } else if constexpr (m == 1) {
                    // necessary condition for "row passes", from the tile's extremes (same algebra and slack as k_scan):
                    //   cosine  d/|a| >= tS            =>  d >= tS * (tS > 0 ? min|a| : max|a|)
                    //   L2      c1|a|^2 - 2 ds d <= tS =>  d >= (c1 * min|a|^2 - tS) / (2 ds)
                    asm volatile("" : "+v"(e.t0), "+v"(e.t1));  // (the LDS read of slice 0 is waited for here, one MFMA later)
                    float b, mag;
                    if (COS) {
                        b = tS * (tS > 0.f ? e.t0 : e.t1);
                        mag = fabsf(b);
                    } else {
                        const float x = c1 * e.t0;
                        b = (x - tS) * hd;
                        mag = (fabsf(x) + fabsf(tS)) * hd;
                    }
                    b = b - mag * 3.8147e-6f - 1.0f;
                    e.eb = b == b ? (int)fminf(fmaxf(ceilf(b), -1.0e9f), 1.0e9f) : 0x7fffffff;
                    // sums are below 2^24 in magnitude: beyond +-2^26 the bound decides for every row, and the packed form fits
                    const int ebc = e.eb < -(1 << 26) ? -(1 << 26) : (e.eb > (1 << 26) ? (1 << 26) : e.eb);
                    e.ebk = ebc * 16;
                    asm volatile("" : "+v"(e.eb), "+v"(e.ebk));
                } else if constexpr (m < 10) {
                    // top two of the packed sums (sum << 4 | slot): d >= eb  <=>  packed >= 16 eb
                    constexpr int i = m - 2;
                    const int ka = (int)(((uint32_t)pv(s, 2 * i) << 4) | (uint32_t)(2 * i));
                    const int kb = (int)(((uint32_t)pv(s, 2 * i + 1) << 4) | (uint32_t)(2 * i + 1));
                    if constexpr (i == 0) {
                        e.m1 = max(ka, kb);
                        e.m2 = min(ka, kb);
                    } else {
                        int md;  // second largest of (m1, ka, kb); hipcc has no pattern for v_med3_i32 on three variables
                        asm("v_med3_i32 %0, %1, %2, %3" : "=v"(md) : "v"(e.m1), "v"(ka), "v"(kb));
                        e.m2 = max(e.m2, md);
                        e.m1 = max(e.m1, max(ka, kb));  // v_max3_i32
                    }
                    asm volatile("" : "+v"(e.m1), "+v"(e.m2));
                } else if constexpr (m == 10) {
                    // the best row's scalar, from the tile record (kept in LDS one tile longer than the rows)
                    const int r1 = e.m1 & 15;
                    const int ri = 4 * h + (r1 & 3) + 8 * (r1 >> 2);
                    e.x = ((const float *)(recring + p_nslot * G::REC_SLOT))[s * PVS_AUX_REC + ri];
                } else if constexpr (m == 11) {
                    // wave-uniform decisions: two rows of one lane at or above the bound -> per-row path; else the best row alone
                    asm volatile("" : "+v"(e.x));  // (slice 10's LDS read is waited for here)
                    const bool hit2 = e.m2 >= e.ebk;
                    if (__builtin_amdgcn_ballot_w64(hit2) != 0) {
                        cnt = emit_rows(s, e.eb, cnt, pv);
                    } else {
                        const int d = e.m1 >> 4;
                        const bool p = e.m1 >= e.ebk && passes(score((float)d, e.x));
                        if (__builtin_amdgcn_ballot_w64(p) != 0) {
                            if (p) {
                                const int r1 = e.m1 & 15;
                                if (cnt < a.seg_cap)
                                    seg_lane[cnt] = make_uint2(prev_row_base + (uint32_t)(s * 32 + (r1 & 3) + 8 * (r1 >> 2)), (uint32_t)d);
                                cnt++;
                            }
                        }
                    }
                }
            } else {
                // rows 0-7 of the lane's 16 in slices 0-4, rows 8-15 in slices 5-9 (eight row scalars live at a time)
                constexpr int half = m / 5, i = m % 5;
                if constexpr (i == 0) {
                    const float *rec = (const float *)(recring + p_nslot * G::REC_SLOT) + s * PVS_AUX_REC;
#pragma unroll
                    for (int g4 = 0; g4 < 2; g4++) {
                        const float4 v = *(const float4 *)(rec + 8 * (2 * half + g4) + 4 * h);
                        ea.xh[4 * g4 + 0] = v.x;
                        ea.xh[4 * g4 + 1] = v.y;
                        ea.xh[4 * g4 + 2] = v.z;
                        ea.xh[4 * g4 + 3] = v.w;
                    }
                } else {
                    // upper bound of the key of two rows: key + err (NaN — padding, zero norm, masked — never wins a fmin)
#pragma unroll
                    for (int rr = 2 * (i - 1); rr < 2 * i; rr++) {
                        constexpr int r0 = 8 * half;
                        const float sv = score((float)pv(s, r0 + rr), ea.xh[rr]);
                        const float ub = COS ? __builtin_fmaf(-sv, qi.dscale, qi.eA) : sv + (qi.bb + qi.eA) + 2.0f * qi.eR * ea.xh[rr];
                        mins[MODE == 0 ? r0 + rr : 0] = fminf(mins[MODE == 0 ? r0 + rr : 0], ub);
                    }
                }
            }
            return cnt;
        };

        // One tile: 2 * NF MFMAs (two accumulation chains) with the LDS-DMA pieces of the tile PC ahead and the previous tile's
        // epilogue slices between them.  Accumulators alternate between two register sets; the previous tile's sums are read where
        // the matrix core left them.
        auto run_tile = [&](wv16i(&acc)[RPW], uint32_t tcnt, auto &&pv) __attribute__((always_inline)) {
            if (wave == 0)
                wait_vm<(PC - 1) * (PPW + 1)>();
            else
                wait_vm<(PC - 1) * PPW>();
            wg_barrier();
            issue_begin();  // refills the slot the previous tile occupied
            const uint8_t *cb = ring + c_slot * G::TILE_BYTES;
            const uint8_t *fb[8];
#pragma unroll
            for (int i = 0; i < 8; i++) fb[i] = cb + swz[i];
            Epi e[RPW];
            EpiA ea;
            constexpr int PF = MODE == 0 ? 1 : 2;  // k-steps of A fragments read ahead (pass A holds 16 minima and 8 row scalars more per lane)
            wv4i af[RPW][NF];
            auto frag = [&](int t) __attribute__((always_inline)) {
#pragma unroll
                for (int s = 0; s < RPW; s++) af[s][t] = *(const wv4i *)(fb[t & 7] + s * G::SUB_BYTES + (t >> 3) * 8192);
            };
#pragma unroll
            for (int t = 0; t < PF && t < NF; t++) frag(t);
            __builtin_amdgcn_sched_barrier(0);
            const wv16i zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            static_for<0, NF>([&](auto tc) __attribute__((always_inline)) {
                constexpr int t = PVS_CI(tc);
                if constexpr (t + PF < NF) frag(t + PF);
                static_for<0, RPW>([&](auto sc) __attribute__((always_inline)) {
                    constexpr int s = PVS_CI(sc);
                    constexpr int g = t * RPW + s;  // MFMA gap index, 0 .. NG-1
                    if constexpr (t == 0)
                        acc[s] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[s][t], qf[t], zero, 0, 0, 0);
                    else
                        acc[s] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[s][t], qf[t], acc[s], 0, 0, 0);
                    static_for<g * (PPW + 1) / NG, (g + 1) * (PPW + 1) / NG>([&](auto pc) __attribute__((always_inline)) { issue_part(PVS_CI(pc)); });
                    // epilogue slices: sub-tile 0's over the first half of the gaps, sub-tile 1's over the second (RPW = 1: all gaps)
                    constexpr int GPS = NG / RPW;  // gaps per sub-tile
                    constexpr int es = g / GPS, eg = g % GPS;
                    static_for<eg * EPI_STEPS / GPS, (eg + 1) * EPI_STEPS / GPS>(
                        [&](auto mc) __attribute__((always_inline)) { tcnt = epi_slice(std::integral_constant<int, es>{}, mc, e[es], ea, tcnt, pv); });
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
            if (++c_slot == NC) c_slot = 0;
            p_nslot = c_nslot;
            if (++c_nslot == NCN) c_nslot = 0;
            prev_row_base = wt_cur * (uint32_t)G::TILE_ROWS + 4u * (uint32_t)h;
            wt_cur += wt_step;
            return tcnt;
        };
        // The last tile's epilogue runs inside one extra "ghost" tile (the DMA stream re-reads the last tile past the end).
        {
            wv16i accA[RPW], accB[RPW];
#pragma unroll
            for (int s = 0; s < RPW; s++)
#pragma unroll
                for (int r = 0; r < 16; r++) accB[s][r] = 0;  // tile "-1"
            auto pa = [&](int s, int r) __attribute__((always_inline)) { return accA[s][r]; };
            auto pb = [&](int s, int r) __attribute__((always_inline)) { return accB[s][r]; };
            for (int tl = 0; tl < n_my + 1; tl += 2) {
                mycnt = run_tile(accA, mycnt, pb);
                if (tl + 1 < n_my + 1) mycnt = run_tile(accB, mycnt, pa);
            }
        }
        wait_vm<0>();  // retire the tail DMAs (and the candidate stores) before the wave exits
    }
    if constexpr (MODE == 1) {
        if (sid < nstreams) a.seg_cnt[(size_t)myq * a.seg_stride + seg] = mycnt;  // every lane's fill count (above seg_cap: overflowed)
    } else {
        if (sid < nstreams) {
            const uint32_t gr = a.gmin_per_lane;  // fold the 16 minima of a lane to gmin_per_lane (a power of two)
#pragma unroll
            for (int sft = 8; sft >= 1; sft >>= 1)
                if (gr <= (uint32_t)sft) {
#pragma unroll
                    for (int r = 0; r < sft; r++) mins[r] = fminf(mins[r], mins[r + sft]);
                }
            float *o = a.gmin + (size_t)myq * a.groups_per_query + (size_t)(sid * 2 + h) * gr;
#pragma unroll
            for (int r = 0; r < 16; r++)
                if ((uint32_t)r < gr) o[r] = mins[r];
        }
    }
}

template <int KS, int METRIC, int MODE>
static hipError_t scan_wide_launch_one(const ScanK &k, hipStream_t s) {
    static std::atomic<bool> configured{false};
    constexpr int lds = WideGeo<KS>::LDS_BYTES;
    if (!configured.load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void *)k_scan_wide<KS, METRIC, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        configured.store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((k_scan_wide<KS, METRIC, MODE>), dim3(k.grid), dim3(512), lds, s, k);
    return hipGetLastError();
}
template <int KS>
static hipError_t scan_wide_launch(const ScanK &k, int metric, int mode, hipStream_t s) {
    if (mode != 0 && mode != 1) return hipErrorInvalidValue;
    if (metric == PVS_COSINE) return mode == 0 ? scan_wide_launch_one<KS, PVS_COSINE, 0>(k, s) : scan_wide_launch_one<KS, PVS_COSINE, 1>(k, s);
    return mode == 0 ? scan_wide_launch_one<KS, PVS_L2, 0>(k, s) : scan_wide_launch_one<KS, PVS_L2, 1>(k, s);
}
