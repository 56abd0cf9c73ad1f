// pvs_direct.hip — host side of the one-launch exact search for 1..8 queries (pvs_direct_kernel.hpp): which instance serves a
// (element type, batch, k, row pitch), how the rows are dealt to the waves, the launch.  Reference: one request = one query and a
// page of 10..320 rows (api/search.rs:46, 524-694), a PQL `or` of a handful of vector filters (pql/builder.rs:638-661), 16 read
// connections asking at once (db/connection.rs:235).
#include "pvs_direct_kernel.hpp"

using namespace pvs_direct;

namespace {
// LDS plan of a (row pitch, element size, k, instance width): the wave lists' capacity, 0 when it does not fit.  A list must
// take the 64 keys of one row pair behind a cut to k; twice the page (>= 128) where LDS allows it halves the number of cuts.
uint32_t plan_capw(uint32_t stride, uint32_t esz, uint32_t k, uint32_t nq_inst) {
    const uint32_t qbytes = nq_inst * (esz == 1 ? stride : stride / esz * 4u);
    if (qbytes + 1024 > (uint32_t)DIR_QSEL_LDS) return 0;
    uint32_t kp = 16;
    while (kp < k) kp <<= 1;
    const uint32_t want = std::max<uint32_t>(2 * kp, 128);
    const uint32_t fit = ((uint32_t)DIR_QSEL_LDS - qbytes) / (32u * nq_inst) / 32u * 32u;  // 4 waves x nq lists x 8 B, a multiple of 32 keys
    const uint32_t capw = std::min<uint32_t>({want, fit, 512u});
    return capw >= k + 64 && nq_inst * k <= 1024 ? capw : 0;
}
uint32_t inst_width(int dtype, uint32_t nq) {
    const uint32_t w = nq <= 1 ? 1 : nq <= 2 ? 2 : nq <= 4 ? 4 : 8;
    return (dtype != PVS_I8 && w > 4) ? 0 : w;  // float rows: up to 4 queries (their f32 copies fill the LDS)
}
}  // namespace

bool pvs_direct_supported(int dtype, uint32_t stride, uint32_t esz, uint32_t k, uint32_t nq) {
    if (k < 1 || k > PVS_DIRECT_MAX_K || nq < 1 || nq > PVS_DIRECT_MAX_NQ) return false;
    const uint32_t w = inst_width(dtype, nq);
    return w && plan_capw(stride, esz, k, w) != 0;
}
// [nq][grid][k] keys (nq k <= 1,024) + [8][grid] counts + control words
uint64_t pvs_direct_work_bytes(uint32_t n_cu) {
    const uint64_t g = std::max<uint32_t>(n_cu, 1);
    return g * 1024 * 8 + g * 8 * 4 + (uint64_t)CTL_WORDS * 4 + 256;
}

hipError_t pvs_launch_direct_topk(const DirectArgs &d, hipStream_t s) {
    const uint32_t esz = pvs_esz((uint32_t)d.dtype);
    if (d.n_rows == 0 || !pvs_direct_supported(d.dtype, d.stride, esz, d.k, d.nq)) return hipErrorInvalidValue;
    const uint32_t nq_inst = inst_width(d.dtype, d.nq);
    DirectK k;
    k.rows = d.rows;
    k.norm2 = d.norm2;
    k.qexact = d.qexact;
    k.qinfo = d.qinfo;
    k.trank = d.trank;
    k.tinv = d.tinv;
    k.ids = d.ids;
    k.mask = d.mask;
    const uint32_t n_cu = std::max<uint32_t>(d.n_cu, 1);
    const uint32_t n_pairs = (uint32_t)((d.n_rows + 63) / 64);
    const uint32_t grid = std::min<uint32_t>({(n_pairs + 3) / 4, n_cu, 256u});
    uint8_t *w = (uint8_t *)d.work;
    k.wg_keys = (unsigned long long *)w;
    k.wg_cnt = (uint32_t *)(w + (size_t)n_cu * 1024 * 8);
    k.ctl = (uint32_t *)(((uintptr_t)(k.wg_cnt + (size_t)n_cu * 8) + 255) & ~(uintptr_t)255);
    k.out_ids = d.out_ids;
    k.out_dist = d.out_dist;
    k.out_count = d.out_count;
    k.need_dense = d.need_dense;
    k.h_flags = d.h_flags;
    k.h_seen = d.h_seen;
    k.h_out_ids = d.h_out_ids;
    k.h_out_dist = d.h_out_dist;
    k.h_out_count = d.h_out_count;
    k.h_out_rows = d.h_out_rows;
    k.n_rows = d.n_rows;
    k.stride = d.stride;
    k.kslabs = d.stride / PVS_KSLAB_BYTES;
    k.dim = d.dim;
    k.qpad_ld = d.stride / esz;
    k.n_pairs = n_pairs;
    k.n_waves = grid * 4;
    k.k = d.k;
    k.capw = plan_capw(d.stride, esz, d.k, nq_inst);
    k.nq = d.nq;
    k.null_ok = d.null_ok;
    // Work distribution.  A unit is >= 48 KB of rows (one 64-row pair at a 768-B pitch: 7 us of a wave's share of the HBM stream):
    // the first half of a wave's share is dealt (unit u of round r = wave + r x waves), the rest is dequeued — workgroups stream at
    // rates that differ by +-25 % (measured: equal shares finished 65 .. 117 us into a search whose bytes take 80), and with ten
    // units per wave nothing but taking units as they are needed evens that out.
    const int64_t unit_dbg = pvs_dbg(PVS_DBG_DIRECT_UNIT), static_pct = pvs_dbg(PVS_DBG_DIRECT_STATIC_PCT);
    k.unit = unit_dbg > 0 ? (uint32_t)unit_dbg : std::max<uint32_t>(1, (49152u + 64u * d.stride - 1) / (64u * d.stride));
    k.n_units = (n_pairs + k.unit - 1) / k.unit;
    const uint32_t per_wave = k.n_units / k.n_waves;  // whole rounds
    const uint32_t pct = static_pct > 0 ? (uint32_t)std::min<int64_t>(static_pct, 100) : 50u;
    k.static_rounds = static_pct < 0 ? 1u : std::max<uint32_t>(1, (uint32_t)((uint64_t)per_wave * pct / 100));
    if (pct >= 100) k.static_rounds = (k.n_units + k.n_waves - 1) / k.n_waves;  // (round 4's dealing: every unit dealt)
    k.dyn = (uint64_t)k.static_rounds * k.n_waves < k.n_units ? 1u : 0u;
    return d.dtype == PVS_I8    ? launch_i8(k, d.metric, nq_inst, grid, s, d.ev_start, d.ev_stop)
           : d.dtype == PVS_F16 ? launch_f16(k, d.metric, nq_inst, grid, s, d.ev_start, d.ev_stop)
                                : launch_f32(k, d.metric, nq_inst, grid, s, d.ev_start, d.ev_stop);
}
