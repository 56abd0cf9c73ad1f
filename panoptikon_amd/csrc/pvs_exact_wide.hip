// pvs_exact_wide.hip — the reference's component-by-component f32 distance (filters/exact.rs:106-134; sqlite-vec's
// vec_distance_cosine / vec_distance_L2, restated in oracle/pvs_oracle.c orc_vec_distance_*) of every stored FLOAT row to 16 or 32
// queries in one pass over the rows (round 5).
//
// k_dense_exact (pvs_dense_exact.hip) holds 8 queries per pass, one row per lane, and reads the query components from LDS: every
// packed multiply needs 8 bytes of query per lane, a broadcast ds_read_b128 feeds two of them and costs 8 LDS cycles for the
// whole CU, so the four SIMDs wait for the one LDS pipe — 4M x 768 f16 rows x 8 queries: 1.9 ms where the packed VALU work is
// 0.7, and 32 queries are four such passes.  What lowers the LDS traffic per multiply is reuse: here a lane owns R rows (2 at 32
// queries, 3 at 16: R x NQ / 2 accumulator pairs + the rows' data must fit 256 registers without a spill), so a query component read
// from LDS once feeds R packed multiplies and the kernel is bound by the packed f32 pipe, which is the chain the reference computes:
// per row, component and query pair one multiply and one add (cosine) or a subtract, a multiply and an add (L2), each with its own
// IEEE rounding, in component order.  Rows come straight from HBM: a lane reads one 128-byte line of each of its rows per step, as
// two 64-byte halves that are reloaded the moment they are consumed (inline assembly, counted waits: see the loop); LDS holds only
// the transposed, zero-padded queries qT[component][NQ]; workgroups are persistent and their waves dequeue blocks of 64 R rows.
// (Tried first: the query components as wave-uniform scalar operands — s_load into SGPR pairs that v_pk_mul_f32 takes directly,
//  no LDS at all, 76 VGPRs.  The instruction stream was ideal, but a scalar load that misses the 16 KB scalar cache takes ~1,000
//  cycles, every line of the 96 KB of queries is used once per wave and SMEM returns out of order (lgkmcnt(0) only), so nothing
//  hides it: 32 queries x 4M x 768 f16 took 6.8 ms, the same as four LDS passes.)
//
// Roofline: VALU (packed f32).  4M x 768 x 32 queries, cosine: 98.3 G component-queries = 1.54 G wave instructions of 4 cycles
// = 2.5 ms at 2.4 GHz on 1,024 SIMDs; HBM traffic is the rows once (6.1 GB for f16: 0.8 ms).
#include "pvs_kernels.hpp"

namespace {

struct WideK {
    const uint8_t *rows;
    const float *norm2;
    const float *qT;  // [qld][NQ] f32: component-major, zero padded in both directions
    const QInfo *qinfo;
    float *out;  // out[row * out_ld + out_col + q]
    uint64_t n_rows;
    uint32_t stride, kslabs, out_ld, out_col, nq, n_tiles, qld;
    uint32_t *ctr;  // the next row block to hand out (zeroed by k_transpose_queries in front of the launch)
    uint32_t n_blocks, n_waves;
};

typedef float v2f __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int WIDE_WAVES = 8;  // per workgroup: they share the LDS copy of the queries

template <int DT, int NQ, int METRIC, int R, int LC>  // LC: 16-byte chunks of a row per load batch (4: half a 128-byte line)
__global__ __launch_bounds__(64 * WIDE_WAVES) void k_exact_wide(WideK a) {
    constexpr int PER = DT == PVS_F16 ? 8 : 4;  // components per 16-byte chunk
    constexpr int GRP = (DT == PVS_F32 && NQ == 32) ? 4 : 8;  // independent products in front of their adds (the f32 x 32 instance sits at 256 registers)
    extern __shared__ __attribute__((aligned(16))) float qs[];  // [qld][NQ]
    for (uint32_t i = threadIdx.x; i < a.qld * (NQ / 4); i += 64 * WIDE_WAVES) ((float4 *)qs)[i] = ((const float4 *)a.qT)[i];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    // A workgroup stays (its LDS copy of the queries is filled once); its waves take row blocks of 64 R rows independently: the
    // first one dealt, the rest from a counter, the next block's number asked for while the current one is computed.  (One
    // workgroup per 8 blocks, the first form: the CU's only LDS slot was held until the slowest of 8 waves had finished and the
    // next workgroup filled its LDS again — 1.4 of 2 waves per SIMD resident on average, SQ_WAVE_CYCLES.)
    uint32_t blk = blockIdx.x * WIDE_WAVES + wave;
    while (blk < a.n_blocks) {
    uint32_t nxt = 0;
    if (lane == 0) nxt = atomicAdd(a.ctr, 1u);
    const uint64_t row0 = (uint64_t)blk * (64 * R) + lane;  // the lane's rows: row0 + 64 r
    const uint32_t rr = lane & 31u, jx = rr & 15u;
    // addresses: one wave-uniform base (the wave's first tile) + a 32-bit lane offset per row (a wave's 2R tiles span < 4 GiB)
    const uint32_t tile0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)((row0 - lane) >> 5));
    const uint8_t *wbase = a.rows + (uint64_t)tile0 * 32 * a.stride;
    uint32_t voff[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const uint32_t tile = min(tile0 + 2 * r + (lane >> 5), a.n_tiles - 1);  // (lanes past the last tile re-read it and write nothing)
        voff[r] = (tile - tile0) * 32 * a.stride + rr * 256u;
    }
    v2f acc[R][NQ / 2];
#pragma unroll
    for (int r = 0; r < R; r++)
#pragma unroll
        for (int p = 0; p < NQ / 2; p++) acc[r][p] = v2f{0.0f, 0.0f};
    // A step = one 128-byte line of each of the lane's rows.  (16 bytes per row and step — the first form — touched each line 8
    // times, thousands of instructions apart, with 2,048 lines per CU in flight against a 32 KB L1: every touch was a new fetch,
    // 1.3 TB/s of useful bytes.)  The line is held as two halves of LC chunks that are reloaded as soon as they are consumed: the
    // loads of one half fly under the arithmetic of the other.  (Loading the whole line at the top of a step left every wave of
    // the chip loading at the same time and then computing at the same time — waves that share a SIMD fall into step — and HBM
    // idle in between: 64 % of the packed-f32 rate.)
    static_assert(LC == 4, "two halves of a line, ping-pong");
    const uint32_t n_steps = a.kslabs * 2;
    u32x4 v[2][R][LC];
    // The loads are inline assembly and so are their waits: written as plain loads, the compiler hoisted every batch to the top of
    // the loop body and waited for it there (it schedules by data dependence only, and the reload of a register it can rename
    // depends on nothing).  A batch is 4R loads; vmcnt counts loads in issue order, so "at most 4R outstanding" behind the issue of
    // the other half means this half has landed.  The destination registers are read only through the wait statement.
    auto load_half = [&](int h, uint32_t step) __attribute__((always_inline)) {
        const uint32_t off = (step >> 1) * 8192u, c0 = (step & 1u) * 8u + (uint32_t)h * LC;
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int c = 0; c < LC; c++) {
                const uint32_t o = voff[r] + off + (((c0 + (uint32_t)c) ^ jx) << 4);
                asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v[h][r][c]) : "v"(o), "s"(wbase));
            }
    };
    auto wait_half = [&](int h, auto all) __attribute__((always_inline)) {  // all but the 4R youngest loads have landed (all: every load)
        constexpr bool ALL = decltype(all)::value;
        static_assert(R == 2 || R == 3, "operand lists below");
#define PVS_W4(r) "+v"(v[h][r][0]), "+v"(v[h][r][1]), "+v"(v[h][r][2]), "+v"(v[h][r][3])
        if constexpr (R == 2 && ALL)
            asm volatile("s_waitcnt vmcnt(0)" : PVS_W4(0), PVS_W4(1));
        else if constexpr (R == 2)
            asm volatile("s_waitcnt vmcnt(8)" : PVS_W4(0), PVS_W4(1));
        else if constexpr (ALL)
            asm volatile("s_waitcnt vmcnt(0)" : PVS_W4(0), PVS_W4(1), PVS_W4(2));
        else
            asm volatile("s_waitcnt vmcnt(12)" : PVS_W4(0), PVS_W4(1), PVS_W4(2));
#undef PVS_W4
    };
    auto compute_half = [&](int h, uint32_t step) __attribute__((always_inline)) {
        const float *qh = qs + ((size_t)step * 8 + (size_t)h * LC) * PER * NQ;
        // one component of all queries per LDS batch: (q0, q1 | q2, q3) per read, the same address in every lane — read one component
        // AHEAD of the arithmetic (a wave that waits for its reads at the top of every component leaves the SIMD to the other wave,
        // which does the same: ~a fifth of the time both wait)
        float4 q4[2][NQ / 4];
#pragma unroll
        for (int x = 0; x < NQ / 4; x++) q4[0][x] = *(const float4 *)(qh + 4 * x);
#pragma unroll
        for (int i = 0; i < LC * PER; i++) {
            const int c = i / PER, e = i % PER, cur = i & 1;
            if (i + 1 < LC * PER) {
#pragma unroll
                for (int x = 0; x < NQ / 4; x++) q4[cur ^ 1][x] = *(const float4 *)(qh + (i + 1) * NQ + 4 * x);
            }
#pragma unroll
            for (int r = 0; r < R; r++) {
                const uint32_t w[4] = {v[h][r][c][0], v[h][r][c][1], v[h][r][c][2], v[h][r][c][3]};
                float av;
                if constexpr (DT == PVS_F16)
                    av = h2f((uint16_t)(w[e >> 1] >> ((e & 1) * 16)));
                else
                    av = __builtin_bit_cast(float, w[e]);
                const v2f av2 = v2f{av, av};
                // (-ffp-contract=off: one rounding per multiply, one per add.)  Groups of GRP independent products, then their adds.
#pragma unroll
                for (int p0 = 0; p0 < NQ / 2; p0 += GRP) {
                    v2f t[GRP];
#pragma unroll
                    for (int j = 0; j < GRP; j++) {
                        const float4 &t4 = q4[cur][(p0 + j) >> 1];
                        const v2f qv = (j & 1) == 0 ? v2f{t4.x, t4.y} : v2f{t4.z, t4.w};
                        if (METRIC == PVS_COSINE) {
                            t[j] = av2 * qv;
                        } else {
                            const v2f d = av2 - qv;
                            t[j] = d * d;
                        }
                    }
#pragma unroll
                    for (int j = 0; j < GRP; j++) acc[r][p0 + j] = acc[r][p0 + j] + t[j];
                }
            }
        }
    };
    load_half(0, 0);
#pragma unroll 1
    for (uint32_t step = 0; step < n_steps; step++) {
        load_half(1, step);
        wait_half(0, std::false_type{});
        compute_half(0, step);
        load_half(0, min(step + 1, n_steps - 1));
        wait_half(1, std::false_type{});
        compute_half(1, step);
    }
    // (the last, redundant reload: its registers must stay reserved until it has landed — the compiler does not know a load is in
    //  flight into them, and handed them to the arithmetic above as temporaries when nothing named them after the loop: a late
    //  arrival overwrote an address, found as a memory fault in the f32 instance)
    wait_half(0, std::true_type{});
    blk = a.n_waves + (uint32_t)__builtin_amdgcn_readfirstlane((int)nxt);
#pragma unroll
    for (int r = 0; r < R; r++) {
        const uint64_t row = row0 + 64 * r;
        if (row < a.n_rows) {
            const float aa = METRIC == PVS_COSINE ? a.norm2[row] : 0.f;
            float *o = a.out + row * a.out_ld + a.out_col;
            const bool full = a.nq == (uint32_t)NQ && (((uintptr_t)o) & 15u) == 0;  // a full pass: 16-byte stores (a dword per lane and query was 4x the write requests)
#pragma unroll
            for (int q4 = 0; q4 < NQ; q4 += 4) {
                float d[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int q = q4 + i;
                    const float sum = acc[r][q >> 1][q & 1];
                    d[i] = METRIC == PVS_COSINE ? ref_cosine_finish(sum, aa, a.qinfo[(uint32_t)q < a.nq ? q : 0].bb) : ref_l2_finish(sum);
                }
                if (full) {
                    *(float4 *)(o + q4) = float4{d[0], d[1], d[2], d[3]};
                } else {
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        if ((uint32_t)(q4 + i) < a.nq) o[q4 + i] = d[i];
                }
            }
        }
    }
    }
}

// [nq][dim] f32 queries -> qT[ld][NQ] f32 (component-major), zero padded
__global__ __launch_bounds__(256) void k_transpose_queries(const float *q, uint32_t nq, uint32_t dim, uint32_t ld, uint32_t NQ, float *qT, uint32_t *ctr) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) *ctr = 0;
    if (i >= ld * NQ) return;
    const uint32_t x = i / NQ, qq = i - x * NQ;
    qT[i] = qq < nq && x < dim ? q[(size_t)qq * dim + x] : 0.f;
}

template <int DT, int NQ, int METRIC>
hipError_t launch_wide_one(const WideK &k, hipStream_t s) {
    constexpr int WIDE_R = NQ == 32 ? 2 : 3, WIDE_LC = 4;  // rows per lane (R x NQ / 2 accumulator pairs + 8 R registers of row data: four rows at 16 queries spill); chunks per load batch
    const size_t lds = (size_t)k.qld * NQ * 4;
    static std::atomic<bool> configured{false};
    if (!configured.load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void *)k_exact_wide<DT, NQ, METRIC, WIDE_R, WIDE_LC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        configured.store(true, std::memory_order_release);
    }
    WideK kk = k;
    kk.n_blocks = (uint32_t)((k.n_rows + 64 * WIDE_R - 1) / (64 * WIDE_R));
    const uint32_t grid = std::min<uint32_t>((kk.n_blocks + WIDE_WAVES - 1) / WIDE_WAVES, std::max<uint32_t>(k.n_waves, 1));  // (k.n_waves: the CU count on entry)
    kk.n_waves = grid * WIDE_WAVES;
    hipLaunchKernelGGL((k_exact_wide<DT, NQ, METRIC, WIDE_R, WIDE_LC>), dim3(grid), dim3(64 * WIDE_WAVES), lds, s, kk);
    return hipGetLastError();
}
template <int DT, int NQ>
hipError_t launch_wide(const WideK &k, int metric, hipStream_t s) {
    return metric == PVS_COSINE ? launch_wide_one<DT, NQ, PVS_COSINE>(k, s) : launch_wide_one<DT, NQ, PVS_L2>(k, s);
}

}  // namespace

// queries per pass the LDS copy allows for this row pitch: 32, 16 or 0 (rows wider than 2,560 components: k_dense_exact's passes)
uint32_t pvs_exact_wide_fit(uint32_t stride, uint32_t esz) {
    const uint64_t ld = stride / esz;
    return ld * 32 * 4 <= 160 * 1024 ? 32u : ld * 16 * 4 <= 160 * 1024 ? 16u : 0u;
}
uint64_t pvs_exact_wide_scratch_bytes(uint32_t stride, uint32_t esz) { return pvs_round_up((uint64_t)PVS_EXACT_WIDE_NQ * (stride / esz) * 4, 256) + 256; }  // qT, then the block counter

// nq <= PVS_EXACT_WIDE_NQ queries (f32, [nq][dim]) against every row; qT_scratch: pvs_exact_wide_scratch_bytes() of device memory
hipError_t pvs_launch_exact_wide(int dtype, int metric, const uint8_t *rows, uint32_t stride, uint32_t dim, uint64_t n, const float *norm2,
                                 const float *queries, const QInfo *qinfo, uint32_t nq, float *qT_scratch, float *out, uint32_t out_ld, uint32_t out_col,
                                 uint32_t n_cu, hipStream_t s) {
    if (n == 0 || nq == 0) return hipSuccess;
    if (dtype == PVS_I8 || nq > PVS_EXACT_WIDE_NQ) return hipErrorInvalidValue;
    const uint32_t esz = pvs_esz((uint32_t)dtype), ld = stride / esz, NQ = nq <= 16 ? 16 : 32;
    uint32_t *ctr = (uint32_t *)((uint8_t *)qT_scratch + pvs_round_up((uint64_t)PVS_EXACT_WIDE_NQ * ld * 4, 256));
    hipLaunchKernelGGL(k_transpose_queries, dim3((ld * NQ + 255) / 256), dim3(256), 0, s, queries, nq, dim, ld, NQ, qT_scratch, ctr);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    WideK k;
    k.rows = rows;
    k.norm2 = norm2;
    k.qT = qT_scratch;
    k.qinfo = qinfo;
    k.out = out;
    k.n_rows = n;
    k.stride = stride;
    k.kslabs = stride / PVS_KSLAB_BYTES;
    k.out_ld = out_ld;
    k.out_col = out_col;
    k.nq = nq;
    k.n_tiles = (uint32_t)((n + 31) / 32);
    k.qld = ld;
    k.ctr = ctr;
    k.n_waves = n_cu;
    k.n_blocks = 0;
    if ((uint64_t)ld * NQ * 4 > 160 * 1024) return hipErrorInvalidValue;  // (pvs_exact_wide_fits: the caller asked first)
    if (dtype == PVS_F16) return NQ == 16 ? launch_wide<PVS_F16, 16>(k, metric, s) : launch_wide<PVS_F16, 32>(k, metric, s);
    return NQ == 16 ? launch_wide<PVS_F32, 16>(k, metric, s) : launch_wide<PVS_F32, 32>(k, metric, s);
}
