// pvs_items_float.hip — per-item pages over FLOAT rows without computing every distance exactly: bound, certify, rescan.
//
// The reference's exact mode is f32 (filters/exact.rs:106-165, image_embeddings.rs:412-438) and similar_to defaults to AVG
// (item_similarity.rs:432-581): `GROUP BY file_id` + rank_aggregate over the `d` column, then ORDER BY ... LIMIT k — a page of k
// files out of millions.  Until round 6 the build answered that for f16 / f32 indexes by running the reference's in-order f32
// chain on EVERY (row, query) pair (k_exact_wide: 0.18 of the HBM rate the rows stream at, bound by the packed-f32 VALU rate:
// 98 M exact chains to return 32 pages of 100 files).  The row search never did that: it filters on the matrix cores with
// rigorous error bounds and rescans only survivors.  This file does the same for per-item pages:
//
//   1. one corpus pass on the matrix cores (pvs_scan_kernel.hpp) yields the scan KEY of every (row, query) pair,
//      |key - kappa| <= err = eA + eR |a|^2, kappa = the reference key the distance D is a monotone function of (cosine:
//      D = fl32(1 + kappa / sqrt(bb)); L2: D = fl32(sqrt(kappa)); HISTORY.md section 4.2 — the algebra passes A/B/C rest on).
//      Files that are runs of rows: k_scan MODE 5 folds steps 1 + 2 in its epilogue (no key matrix), only the
//      rows of tile-crossing files leave it (k_spill_bounds); otherwise MODE 4 writes the keys and k_run_bounds / k_group_bounds fold;
//   2. per (file, query) the bracket [lo_i, hi_i] of every row's distance is folded into a bracket [L, U] of the
//      file's aggregate — AVG: means of the brackets; MIN / MAX: min / max of the ends; SUM(d w)/SUM(w) with positive weights: the
//      weighted means — widened for every rounding on the way.  A file with a row whose distance may be NULL (zero / non-finite
//      norm, non-finite key) or a non-positive weight is FORCED: L = -inf, it never lowers the threshold.  U goes into one of
//      16,384 per-query buckets (a minimum per bucket: registers of the scan's lanes in MODE 5, atomics otherwise); L is stored, and
//      MODE 5 keeps the buckets' minima of L as well;
//   3. k_kth (pass A's select): T = the k-th smallest bucket minimum >= the k-th smallest U >= the k-th smallest exact value;
//   4. k_candidates_tiles (after MODE 5: only the buckets whose smallest L is at or below T are looked into) / k_candidates (the dense
//      form) / k_union: every file with L <= T for a query is that query's candidate (every file of its true page is one:
//      exact value <= the k-th exact value <= T, and L <= exact); a query that names more than 1,024 ties with everything and is
//      set aside; the union of the others' candidates is the file list of the chunk;
//   5. pvs_sparse_groups_of_files (pvs_sparse.hip): the exact in-order chain on the rows of those files only, SQLite's KBN sums per
//      file, ranking under the page order (value, [order key,] file id, NULL last) — for every query over the UNION of the
//      candidates, a superset of each query's own, all of it exact: the page is the reference's, bit for bit.
//
// Whatever cannot be certified — too many candidates (ties, a page deeper than the files with a finite bracket, a NULL query,
// non-finite components), k beyond what the buckets resolve — is handed back (*handled = false) and the exact-everywhere route
// (k_exact_wide + k_group_aggregate8) answers as before.  pvs_score_all / pvs_score_batch (the SQL seam's full `d` column) keep
// the exact kernels: they return every distance.
#include "pvs_index.hpp"

namespace {
constexpr uint32_t BUCKETS = PVS_FLOAT_BUCKETS;  // per query: minima of U over disjoint sets of files (file index mod BUCKETS) -> k_kth

struct BoundsK {
    const float *keys;       // [n][ld]
    uint32_t ld, nb;         // ld = nb rounded up to a multiple of 4 (16-byte lines); lo and bucket_min share it
    const QInfo *qinfo;
    const float *norm2;      // [n] |a|^2 (sequential f32: the reference's aMag)
    const uint32_t *grp_off, *grp_rows;
    uint32_t n_groups;
    const float *weights;    // optional [n]
    const uint8_t *mask;     // optional [n]: 0 = the row takes part in nothing
    bool rows_are_runs;      // grp_rows[e] == e: every file is one run of consecutive rows (no indirection)
    int metric, agg;
    float *lo;               // [n_groups][ld]
    uint32_t *bucket_min;    // [BUCKETS][ld] bit patterns of non-negative floats (query-minor: the lanes of a file touch one line; k_kth reads the transpose)
    uint32_t *bad_query;     // [nb] 1: nothing of this query can be bracketed (every distance NULL, a non-finite norm): it names no candidate
                             // and the caller answers it through the exact-everywhere route
};

constexpr uint32_t QCAP = 1024;   // candidate files one query may name before it counts as "cannot be certified" (ties with everything)
constexpr uint32_t UCAP = 8192;   // files in the union of a chunk's candidates (one LDS ranking per query column: pvs_sub_rank)
constexpr uint32_t QCS = 32;      // words between two queries' candidate counters: a 128-byte line each
constexpr int RUN_ROWS = 16;      // rows a thread of k_run_bounds loads at once

// The bracket of one row's distance from its key (f32: every rounding — five operations of 6e-8, the half ulp of the reference's own
// f32 distance — is inside the 1e-6 (1 + |d|) the bracket is widened by) folded into a file's accumulators.
struct FileAcc {
    double s_lo, s_hi, s_w;
    float x_lo, x_hi;
    uint32_t cnt;
    bool forced;
    __device__ void reset(bool want_min) {
        s_lo = s_hi = s_w = 0.0;
        x_lo = x_hi = want_min ? __builtin_inff() : -__builtin_inff();
        cnt = 0;
        forced = false;
    }
};
struct QConst {
    float eA, eR, inv_sb;
    bool ok, cosine, weighted, want_min, want_max;
};
__device__ static inline QConst q_const(const BoundsK &a, uint32_t q) {
    const QInfo qi = a.qinfo[q];
    QConst c;
    c.cosine = a.metric == PVS_COSINE;
    // a query whose distances are all NULL, or whose norm is not finite: nothing can be bracketed — it names no candidate here and is
    // answered by the exact-everywhere route (its page is the first k files in (order key, id) order, all NULL, or worse)
    c.ok = qi.bb == qi.bb && qi.bb < __builtin_inff() && (!c.cosine || qi.bb > 0.f) && qi.dscale > 0.f;
    c.eA = qi.eA;
    c.eR = qi.eR;
    c.inv_sb = c.cosine && c.ok ? 1.0f / sqrtf(qi.bb) : 0.f;  // (two roundings: inside the margin)
    c.weighted = a.weights != nullptr;
    c.want_min = a.agg == PVS_AGG_MIN && !c.weighted;
    c.want_max = a.agg == PVS_AGG_MAX && !c.weighted;
    return c;
}
__device__ static inline void acc_row(FileAcc &f, const QConst &c, float key, float aa, float w) {
    f.cnt++;
    // a row whose distance may be NULL or infinite (cosine: zero vector, |a|^2 under / overflow; NaN / inf components), or whose key
    // is not a number: no bracket.  (L2: a zero vector is an ordinary row at distance |q|.)
    if (!((c.cosine ? aa > 1e-30f : aa >= 0.f) && aa < 1e30f) || !(fabsf(key) <= 1e30f)) {
        f.forced = true;
        return;
    }
    float lo, hi;
    if (c.cosine) {
        lo = 1.0f + (key - c.eA) * c.inv_sb;
        hi = 1.0f + (key + c.eA) * c.inv_sb;
    } else {
        const float err = c.eA + c.eR * aa;
        lo = sqrtf(fmaxf(key - err, 0.f));
        hi = sqrtf(fmaxf(key + err, 0.f));
    }
    lo -= 1e-6f * (1.0f + fabsf(lo));
    hi += 1e-6f * (1.0f + fabsf(hi));
    if (c.weighted) {
        if (!(w > 0.f && w < 1e30f)) {
            f.forced = true;
            return;
        }
        f.s_lo += (double)lo * (double)w;
        f.s_hi += (double)hi * (double)w;
        f.s_w += (double)w;
    } else if (c.want_min) {
        f.x_lo = fminf(f.x_lo, lo);
        f.x_hi = fminf(f.x_hi, hi);
    } else if (c.want_max) {
        f.x_lo = fmaxf(f.x_lo, lo);
        f.x_hi = fmaxf(f.x_hi, hi);
    } else {
        f.s_lo += (double)lo;
        f.s_hi += (double)hi;
    }
}
// a file's bracket [L, U] -> lo[file][q], and U into its bucket
__device__ static inline void emit_file(const BoundsK &a, const QConst &c, const FileAcc &f, uint64_t file, uint32_t q) {
    float L, U;
    if (!c.ok) {
        L = __builtin_nanf("");  // (compares false against any threshold, +inf included)
        U = __builtin_inff();
    } else if (f.cnt == 0) {  // no candidate row: the file is not part of the result at all
        L = U = __builtin_inff();
    } else if (f.forced) {
        L = -__builtin_inff();
        U = __builtin_inff();
    } else {
        if (c.want_min || c.want_max) {
            L = f.x_lo;
            U = f.x_hi;
        } else {
            // (the f64 sums are exact to 1e-16 relative per term, SQLite's compensated sum within an ulp of the true one; the f32
            //  reciprocal, product and conversion: three roundings — 1e-6 relative covers all of it many times over)
            const float inv = 1.0f / (c.weighted ? (float)f.s_w : (float)f.cnt);
            L = (float)f.s_lo * inv;
            U = (float)f.s_hi * inv;
            L -= 1e-6f * (1.0f + fabsf(L));
            U += 1e-6f * (1.0f + fabsf(U));
        }
        if (!(L == L) || !(U == U)) {
            L = -__builtin_inff();
            U = __builtin_inff();
        }
    }
    a.lo[file * a.ld + q] = L;
    if (U < __builtin_inff()) {
        if (U < 0.f) U = 0.f;  // (raising an upper bound keeps it one; non-negative floats order like their bit patterns)
        uint32_t *slot = a.bucket_min + (size_t)(file % BUCKETS) * a.ld + q;
        const uint32_t ub = __builtin_bit_cast(uint32_t, U);
        // (the minimum settles after a few files per bucket: most threads only look — with a device-scope load the L2 serves; a
        //  `volatile` read compiles to a system-scope load that goes out to the fabric: 42 M of them were most of this kernel's time)
        if (ub < __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(slot, ub);
    }
}

// Files that are RUNS of consecutive rows (the reference's loader streams ORDER BY item_data.id: a file's vectors are adjacent): a
// thread takes RUN_ROWS consecutive rows of one query — file slot, |a|^2, key (+ weight, mask) of all of them loaded before any is
// used: ~50 independent loads in flight per thread, the keys of a row one 128-byte line across the 32 query lanes — and folds them
// file by file.  A file belongs to the thread that holds its FIRST row: rows at the start of the block that continue a file from
// the block before are skipped, a file that runs past the block's end is followed to its end.  (A thread per (file, query) walked
// file -> row -> key one dependent load at a time: 0.44-0.95 ms for 0.7 GB in three variants — latency, not bandwidth; a thread
// per file with the queries in registers: 1.5 ms — 16 bytes per lane and line.)
__global__ __launch_bounds__(256) void k_run_bounds(BoundsK a, const uint32_t *row_gidx, uint64_t n_rows, uint32_t nbp, uint32_t nbp_log2) {
    const uint32_t q = threadIdx.x & (nbp - 1u);
    if (q >= a.nb) return;
    const QConst c = q_const(a, q);
    if (blockIdx.x == 0 && threadIdx.x < nbp) a.bad_query[q] = c.ok ? 0u : 1u;
    const uint64_t r0 = ((uint64_t)blockIdx.x * (256u / nbp) + (threadIdx.x >> nbp_log2)) * RUN_ROWS;
    if (r0 >= n_rows) return;
    uint32_t g[RUN_ROWS];
    float key[RUN_ROWS], aa[RUN_ROWS], w[RUN_ROWS];
    uint8_t use[RUN_ROWS];
    const uint32_t g_before = r0 ? row_gidx[r0 - 1] : 0xffffffffu;
#pragma unroll
    for (int i = 0; i < RUN_ROWS; i++) {
        const uint64_t r = r0 + i < n_rows ? r0 + i : n_rows - 1;
        g[i] = row_gidx[r];
        aa[i] = a.norm2[r];
        key[i] = a.keys[r * a.ld + q];
        w[i] = c.weighted ? a.weights[r] : 1.0f;
        use[i] = a.mask ? a.mask[r] : (uint8_t)1;
    }
    FileAcc f;
    f.reset(c.want_min);
    uint32_t cur = 0xffffffffu;
#pragma unroll
    for (int i = 0; i < RUN_ROWS; i++) {
        if (r0 + i >= n_rows || g[i] == g_before) continue;  // (beyond the index / the tail of a file an earlier thread owns)
        if (g[i] != cur) {
            if (cur != 0xffffffffu) emit_file(a, c, f, cur, q);
            cur = g[i];
            f.reset(c.want_min);
        }
        if (use[i]) acc_row(f, c, key[i], aa[i], w[i]);
    }
    if (cur == 0xffffffffu) return;
    for (uint64_t r = r0 + RUN_ROWS; r < n_rows && row_gidx[r] == cur; r++)  // the last file runs on into the next block(s)
        if (!a.mask || a.mask[r]) acc_row(f, c, a.keys[r * a.ld + q], a.norm2[r], c.weighted ? a.weights[r] : 1.0f);
    emit_file(a, c, f, cur, q);
}

// Files whose rows lie anywhere: one thread per (file, query) walks the file's rows through the CSR (the scattered case: dependent
// loads, several times slower than the run form above)
__global__ __launch_bounds__(256) void k_group_bounds(BoundsK a, uint32_t nbp, uint32_t nbp_log2) {
    const uint32_t q = threadIdx.x & (nbp - 1u);
    if (q >= a.nb) return;
    const QConst c = q_const(a, q);
    if (blockIdx.x == 0 && threadIdx.x < nbp) a.bad_query[q] = c.ok ? 0u : 1u;
    const uint64_t file = (uint64_t)blockIdx.x * (256u / nbp) + (threadIdx.x >> nbp_log2);
    if (file >= a.n_groups) return;
    FileAcc f;
    f.reset(c.want_min);
    const uint32_t e0 = a.grp_off[file], e1 = a.grp_off[file + 1];
    for (uint32_t e = e0; e < e1; e += 4) {
        uint32_t row[4];
        bool in[4];
        float aa4[4], key4[4], w4[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            in[i] = e + i < e1;
            row[i] = in[i] ? a.grp_rows[e + i] : 0u;
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            aa4[i] = a.norm2[row[i]];
            key4[i] = a.keys[(size_t)row[i] * a.ld + q];
            w4[i] = c.weighted ? a.weights[row[i]] : 1.0f;
            if (a.mask) in[i] = in[i] && a.mask[row[i]] != 0;
        }
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (in[i]) acc_row(f, c, key4[i], aa4[i], w4[i]);
    }
    emit_file(a, c, f, file, q);
}

// MODE 5 leaves the row brackets of the files that cross a 32-row tile boundary in two sparse matrices: one thread per (such
// file, query) folds them (a few percent of the files)
__global__ __launch_bounds__(256) void k_spill_bounds(BoundsK a, const uint32_t *list, uint32_t n_list, const float *lo_rows, const float *hi_rows, uint32_t nbp, uint32_t nbp_log2,
                                                      float *lo_list) {
    const uint32_t q = threadIdx.x & (nbp - 1u);
    if (q >= a.nb) return;
    const QConst c = q_const(a, q);
    const uint64_t li = (uint64_t)blockIdx.x * (256u / nbp) + (threadIdx.x >> nbp_log2);
    if (li >= n_list) return;
    const uint32_t file = list[li];
    const uint32_t e0 = a.grp_off[file], e1 = a.grp_off[file + 1];
    float s_lo = 0.f, s_hi = 0.f, s_w = 0.f;
    float x_lo = c.want_min ? __builtin_inff() : -__builtin_inff(), x_hi = x_lo;
    uint32_t cnt = 0;
    bool forced = false;
    // (four rows at a time, their loads issued together: a row after the other was one memory round trip per row and end)
    for (uint32_t e = e0; e < e1; e += 4) {
        uint32_t row[4];
        bool in[4];
        float lo4[4], hi4[4], w4[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            in[i] = e + i < e1;
            row[i] = in[i] ? (a.rows_are_runs ? e + i : a.grp_rows[e + i]) : 0u;
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            lo4[i] = in[i] ? lo_rows[(size_t)row[i] * a.ld + q] : 0.f;
            hi4[i] = in[i] ? hi_rows[(size_t)row[i] * a.ld + q] : 0.f;
            w4[i] = (in[i] && c.weighted) ? a.weights[row[i]] : 1.0f;
            if (a.mask) in[i] = in[i] && a.mask[row[i]] != 0;
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (!in[i]) continue;
            cnt++;
            const float lo = lo4[i], hi = hi4[i];
            if (!(lo == lo) || !(hi == hi)) forced = true;
            if (c.weighted) {
                const float w = w4[i];
                if (!(w > 0.f && w < 1e30f)) forced = true;
                s_lo += lo * w;
                s_hi += hi * w;
                s_w += w;
            } else if (c.want_min) {
                x_lo = fminf(x_lo, lo);
                x_hi = fminf(x_hi, hi);
            } else if (c.want_max) {
                x_lo = fmaxf(x_lo, lo);
                x_hi = fmaxf(x_hi, hi);
            } else {
                s_lo += lo;
                s_hi += hi;
            }
        }
    }
    float L, U;
    if (!c.ok) {
        L = __builtin_nanf("");
        U = __builtin_inff();
    } else if (cnt == 0) {
        L = U = __builtin_inff();
    } else if (forced) {
        L = -__builtin_inff();
        U = __builtin_inff();
    } else {
        float l, u;
        if (c.want_min || c.want_max) {
            l = x_lo;
            u = x_hi;
        } else {
            const float inv = 1.0f / (c.weighted ? s_w : (float)cnt);
            l = s_lo * inv;
            u = s_hi * inv;
        }
        // (f32 sums over the rows of one file: cnt roundings of 6e-8 each — a file of more than a few thousand rows is not worth a
        //  bracket: forced)
        const float mg = 1e-6f + 1.5e-7f * (float)cnt;
        L = l - mg * (1.0f + fabsf(l));
        U = u + mg * (1.0f + fabsf(u));
        if (!(L == L) || !(U == U) || cnt > 4096) {
            L = -__builtin_inff();
            U = __builtin_inff();
        }
    }
    lo_list[li * a.ld + q] = L;  // (compact, in list order: k_candidates reads these files' lower bounds as one dense matrix)
    if (U < __builtin_inff()) {
        if (U < 0.f) U = 0.f;
        // (the scan's lanes own buckets [0, BUCKETS / 2): k_scan MODE 5 writes grid x RT x 8 <= 8,192 of them; these files take the rest)
        uint32_t *slot = a.bucket_min + (size_t)(BUCKETS / 2 + file % (BUCKETS / 2)) * a.ld + q;
        const uint32_t ub = __builtin_bit_cast(uint32_t, U);
        if (ub < __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(slot, ub);
    }
}
__global__ void k_query_flags(BoundsK a) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < a.nb) a.bad_query[q] = q_const(a, q).ok ? 0u : 1u;
}

// one thread per (file, query), 8 files per thread (their loads issued before the first compare): a candidate (L <= T) is appended
// to its query's list.  (170 MB of lower bounds in 0.09 ms = 1.8 TB/s; 16-byte loads per lane measured no better: 0.107 ms.)
__global__ __launch_bounds__(256) void k_candidates(const float *lo, uint32_t nb, uint32_t ld, uint32_t nbp, uint32_t nbp_log2, uint32_t n_groups, const float *thr,
                                                    uint32_t *qcnt, uint32_t *qlist, const uint32_t *file_of) {
    const uint32_t q = threadIdx.x & (nbp - 1u);
    if (q >= nb) return;
    const float t = thr[q];
    const uint32_t fpb = 256u / nbp;
    float v[8];
#pragma unroll
    for (uint32_t it = 0; it < 8; it++) {
        const uint64_t f = ((uint64_t)blockIdx.x * 8 + it) * fpb + (threadIdx.x >> nbp_log2);
        v[it] = f < n_groups ? lo[f * ld + q] : __builtin_nanf("");
    }
#pragma unroll
    for (uint32_t it = 0; it < 8; it++) {
        if (!(v[it] <= t)) continue;
        const uint64_t f = ((uint64_t)blockIdx.x * 8 + it) * fpb + (threadIdx.x >> nbp_log2);
        if (__hip_atomic_load(qcnt + q * QCS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 4 * QCAP) return;  // (already beyond saving: the count only has to say so)
        const uint32_t slot = atomicAdd(qcnt + q * QCS, 1u);
        if (slot < QCAP) qlist[(size_t)q * QCAP + slot] = file_of ? file_of[f] : (uint32_t)f;
    }
}
// After k_scan MODE 5: the lower bounds of the files folded in the scan are never read as a matrix.  The scan's lanes kept, beside
// the minima of the upper bounds, the minima of the LOWER bounds of the same buckets (bucket = (wave of the scan, number of the
// wave's tile mod 8): the files that end in those tiles); a bucket whose minimum is above the query's threshold holds no candidate —
// all but ~60 of 8,192 per query.  One workgroup per scan wave: its (slot, query) pairs at or below the threshold are collected, then
// its threads visit the tiles of those slots (the tile records name the files that end in a tile, in order) and look at the lower
// bounds of their files.
// (170 MB of lower bounds read at 1.8 TB/s: 0.09 ms; this: a 1-MB matrix of minima and a few thousand 4-byte reads.)
__global__ __launch_bounds__(256) void k_candidates_tiles(const float *lmin, const float *lo, uint32_t nb, uint32_t ld, const float *thr, const uint4 *tile_grp, uint64_t n_rows,
                                                          uint32_t RT, uint32_t nstreams, uint32_t n_wgtiles, uint32_t *qcnt, uint32_t *qlist, uint32_t *visits) {
    __shared__ uint32_t s_pairs[8 * PVS_MAX_BATCH];
    __shared__ uint32_t s_n;
    __shared__ uint32_t s_dead[PVS_MAX_BATCH];  // the query already names more files than it may (ties with everything): stop adding to it
    const uint32_t w = blockIdx.x, sid = w / RT, rt = w % RT;
    if (threadIdx.x == 0) s_n = 0;
    if (threadIdx.x < nb) s_dead[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t p = threadIdx.x; p < 8u * nb; p += 256u) {
        const uint32_t slot = p / nb, q = p % nb;
        if (lmin[(size_t)(w * 8u + slot) * ld + q] <= thr[q]) s_pairs[atomicAdd(&s_n, 1u)] = p;
    }
    __syncthreads();
    const uint32_t np = s_n;
    if (np == 0 || sid >= n_wgtiles) return;
    if (visits && threadIdx.x == 0) atomicAdd(visits, np);
    const uint32_t n_my = (n_wgtiles - sid + nstreams - 1) / nstreams, n8 = (n_my + 7u) / 8u;
    // one item per (pair, tile of the pair's slot: this wave's tiles number slot, slot + 8, ...): the record, then the lower bounds of
    // the files that end in the tile, eight loads at a time.  (A thread per tile that walked the pairs and the files one dependent
    // load after the other: 45 us on the slowest workgroup's chain.)
    for (uint32_t it = threadIdx.x; it < np * n8; it += 256u) {
        const uint32_t ip = it / n8, j = s_pairs[ip] / nb + 8u * (it - ip * n8);
        const uint32_t q = s_pairs[ip] % nb;
        if (j >= n_my || *(volatile uint32_t *)&s_dead[q]) continue;
        const uint64_t u = (uint64_t)(sid + j * nstreams) * RT + rt, row0 = u * 32u;
        if (row0 >= n_rows) continue;
        const uint4 rec = tile_grp[u];
        const uint32_t n_here = n_rows - row0 >= 32u ? 32u : (uint32_t)(n_rows - row0);
        const uint32_t m_rows = n_here == 32u ? 0xffffffffu : ((1u << n_here) - 1u);
        const uint32_t m_end = rec.y & m_rows;
        // bit o of em: the o-th file that ends in this tile was folded by the scan (does not cross a tile boundary)
        uint32_t em = 0, o = 0, m = m_end;
        while (m) {
            const uint32_t i = (uint32_t)__builtin_ctz(m);
            m &= m - 1u;
            em |= ((rec.z >> i) & 1u) ? 0u : (1u << o);
            o++;
        }
        const float t = thr[q];
        while (em) {
            float v[8];
            uint32_t off[8];
#pragma unroll
            for (int k8 = 0; k8 < 8; k8++) {
                off[k8] = em ? (uint32_t)__builtin_ctz(em) : 0u;
                v[k8] = em ? lo[(size_t)(rec.x + off[k8]) * ld + q] : __builtin_nanf("");
                em &= em - 1u;  // (0 stays 0)
            }
#pragma unroll
            for (int k8 = 0; k8 < 8; k8++) {
                if (!(v[k8] <= t)) continue;
                const uint32_t at = atomicAdd(qcnt + q * QCS, 1u);
                if (at < QCAP) qlist[(size_t)q * QCAP + at] = rec.x + off[k8];
                if (at > 4 * QCAP) s_dead[q] = 1;
            }
        }
    }
}
// what a chunk starts from, in one launch (three fills cost five: ~26 us in front of every scan): bucket minima +inf, counters and
// the union's bits 0
__global__ __launch_bounds__(256) void k_certify_init(uint32_t *bmin, uint32_t n_bmin, uint32_t *small, uint32_t n_small, uint32_t *bits, uint32_t n_bits) {
    const uint32_t stride = gridDim.x * 256;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n_bmin; i += stride) bmin[i] = 0x7f800000u;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n_bits; i += stride) bits[i] = 0;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n_small; i += stride) small[i] = 0;
}
// bucket minima [BUCKETS][ld] -> [nb][BUCKETS / merge] (what k_kth reads): `merge` neighbouring buckets become one (the union of
// disjoint sets of files is one; a page of k <= 256 files does not need 16,384 buckets to resolve its k-th smallest upper bound, and
// the select over 4,096 values takes a third of the time)
__global__ __launch_bounds__(256) void k_bucket_transpose(const uint32_t *in, uint32_t nb, uint32_t ld, uint32_t *out, uint32_t merge) {
    const uint32_t per = BUCKETS / merge;
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nb * per) return;
    const uint32_t q = i / per, b = i % per;
    uint32_t m = 0xffffffffu;  // (bit patterns of non-negative floats, +inf included: unsigned order)
    for (uint32_t j = 0; j < merge; j++) m = min(m, in[(size_t)(b * merge + j) * ld + q]);
    out[i] = m;
}
// the union of the lists of the queries that stayed below QCAP: a file enters once (a bit per file), with its allowed rows counted
__global__ __launch_bounds__(256) void k_union(const uint32_t *qcnt, const uint32_t *qlist, uint32_t nb, uint32_t *bits, const uint32_t *grp_off, const uint32_t *grp_rows,
                                               const uint8_t *mask, uint32_t *ucnt, uint32_t *ufiles, uint32_t *qtot) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x, q = blockIdx.y;
    const uint32_t c = qcnt[q * QCS];
    if (i == 0) qtot[q] = c;  // (what the host reads: the padded counters stay on the device)
    if (c > QCAP || i >= c) return;
    const uint32_t f = qlist[(size_t)q * QCAP + i];
    const uint32_t bit = 1u << (f & 31u);
    if (atomicOr(bits + (f >> 5), bit) & bit) return;
    const uint32_t slot = atomicAdd(ucnt, 1u);
    if (slot < UCAP) ufiles[slot] = f;
    uint32_t rows = 0;
    for (uint32_t e = grp_off[f]; e < grp_off[f + 1]; e++) rows += (!mask || mask[grp_rows[e]]) ? 1u : 0u;
    atomicAdd(ucnt + 1, rows);
}
}  // namespace

bool pvs_float_certify_applies(const pvs_index *ix, uint32_t nb, uint32_t k) {
    if (pvs_dbg(PVS_DBG_NO_FLOAT_CERTIFY) || ix->forced_path == 1) return false;
    if (ix->dtype == PVS_I8 || ix->n == 0 || ix->n_groups == 0 || !ix->d_row_gidx) return false;
    if (!pvs_scan_supported((int)ix->dtype, ix->stride / PVS_KSLAB_BYTES)) return false;
    if (nb > 128) return false;
    // the buckets must resolve the k-th smallest upper bound: many more buckets (and files) than k
    if ((uint64_t)k * 16 > BUCKETS || (uint64_t)k * 64 > ix->n_groups) return false;
    // a handful of rows: the exact kernels are cheaper than five launches
    return ix->n >= 16384;
}

// The queries [q0, q0 + nb) of d_queries were prepared in c (prep_chunk with batch_pad).
// *handled: the pages of the chunk are in out_*; (*redo)[q] != 0: except this query's, which the caller answers through the
// exact-everywhere route (nothing of it can be bracketed, or it ties with more than QCAP files).
pvs_status pvs_float_groups_certified(pvs_index *ix, SearchCtx &c, const void *d_queries, int qdtype, uint32_t q0, uint32_t nb, uint32_t batch_pad, uint32_t k,
                                      int metric, int agg, const float *d_w, const uint8_t *d_mask, int64_t *out_groups, double *out_values,
                                      uint32_t *out_count, bool *handled, std::vector<uint8_t> *redo) {
    *handled = false;
    redo->assign(nb, 0);
    hipStream_t s = c.stream;
    const uint32_t G = ix->n_groups;
    const uint32_t ld = (nb + 3u) & ~3u;  // keys, lower bounds and bucket minima in 16-byte lines
    float *d_keys = nullptr, *d_lo = nullptr, *d_thr = nullptr, *d_lo_list = nullptr, *d_lmin = nullptr;
    uint32_t *d_bmin = nullptr, *d_small = nullptr, *d_qlist = nullptr, *d_bits = nullptr, *d_ufiles = nullptr;
    // d_small: [bad query flags nb | candidate counts nb | union files, union rows] — one copy to the host
    const size_t n_small = 2 * (size_t)nb + 3;  // bad-query flags, candidate totals, union files, union rows, (trace) bucket visits; behind them the padded counters
    const size_t n_pad = (size_t)nb * QCS;
    if (!c.h_cert) HIP_TRY(hipHostMalloc((void **)&c.h_cert, (2 * (size_t)PVS_SCAN_MAX_BATCH + 3) * 4, hipHostMallocDefault));
    const uint32_t *h_small = c.h_cert;  // (pinned: the copy is one DMA, not a staged one behind a synchronisation of its own)
    const size_t bits_bytes = ((size_t)G + 31) / 32 * 4;
    auto body = [&]() -> pvs_status {
        HIP_TRY(pvs_scratch_alloc((void **)&d_lo, (size_t)G * ld * 4));
        // (query-minor minima of the upper bounds, their transpose, the scan's minima of the lower bounds [BUCKETS / 2][ld])
        HIP_TRY(pvs_scratch_alloc((void **)&d_bmin, (((size_t)ld + nb) * BUCKETS + (size_t)ld * (BUCKETS / 2)) * 4));
        HIP_TRY(pvs_scratch_alloc((void **)&d_thr, (size_t)nb * 4));
        HIP_TRY(pvs_scratch_alloc((void **)&d_small, (n_small + n_pad) * 4));
        HIP_TRY(pvs_scratch_alloc((void **)&d_qlist, (size_t)nb * QCAP * 4));
        HIP_TRY(pvs_scratch_alloc((void **)&d_bits, bits_bytes));
        HIP_TRY(pvs_scratch_alloc((void **)&d_ufiles, (size_t)UCAP * 4));
        d_lmin = (float *)(d_bmin + ((size_t)ld + nb) * BUCKETS);
        uint32_t *d_badq = d_small, *d_qtot = d_small + nb, *d_ucnt = d_small + 2 * (size_t)nb, *d_qcnt = d_small + n_small;
        // 1. + 2. one corpus pass on the matrix cores.  Files that are runs of rows (the loader's order): the brackets are folded per
        // file in the scan's epilogue (k_scan MODE 5) and only the rows of tile-crossing files leave it; otherwise the scan writes
        // the key of every (row, query) pair (MODE 4) and a second kernel walks the files.
        BoundsK b;
        b.keys = nullptr;
        b.ld = ld;
        b.nb = nb;
        b.qinfo = c.d_qinfo;
        b.norm2 = ix->d_norm2;
        b.grp_off = ix->d_grp_off;
        b.grp_rows = ix->d_grp_rows;
        b.n_groups = G;
        b.weights = d_w;
        b.mask = d_mask;
        b.rows_are_runs = ix->groups_are_runs;
        b.metric = metric;
        b.agg = agg;
        b.lo = d_lo;
        b.bucket_min = d_bmin;
        b.bad_query = d_badq;
        uint32_t nbp = 1, nbp_log2 = 0;
        while (nbp < nb) nbp <<= 1, nbp_log2++;
        const uint32_t lanes_rows = 256u / nbp;  // row blocks (run form), files (CSR form) or listed files per workgroup
        const bool fold_in_scan = ix->groups_are_runs && ix->d_tile_grp && pvs_scan_fold5_supported((int)ix->dtype, ix->stride / PVS_KSLAB_BYTES) && !(((uintptr_t)d_mask & 3u) || ((uintptr_t)d_w & 3u)) && !pvs_dbg(PVS_DBG_FLOAT_CERTIFY_NO_FOLD);
        ScanArgs a;
        a.dtype = (int)ix->dtype;
        a.metric = metric;
        a.kslabs = ix->stride / PVS_KSLAB_BYTES;
        a.qgroups = batch_pad / 32;
        a.rows = ix->d_rows;
        a.aux = metric == PVS_COSINE ? ix->d_scan_cos : ix->d_scan_l2;
        a.stride = ix->stride;
        a.n_rows = ix->n;
        a.qmat = c.d_qmat;
        a.qinfo = c.d_qinfo;
        a.thr = c.d_thr;
        a.gmin = c.d_gmin;
        a.groups_per_query = 0;
        a.mode = fold_in_scan ? 5 : 4;
        a.tile_step = 1;
        const uint32_t rt_scan = pvs_scan_row_tiles(a.qgroups);
        const uint32_t wg_rows = 32u * rt_scan;
        const uint32_t n_wgtiles = (uint32_t)((ix->n + wg_rows - 1) / wg_rows);
        // (MODE 5: one workgroup per CU; its lanes own 8 buckets each: grid x RT x 8 <= BUCKETS / 2)
        a.grid = std::min<uint32_t>(n_wgtiles, fold_in_scan ? std::min<uint32_t>((uint32_t)ix->n_cu, BUCKETS / 2 / (8u * pvs_scan_row_tiles(a.qgroups)))
                                                            : (uint32_t)ix->n_cu * pvs_scan_wg_per_cu(a.dtype, a.qgroups, a.kslabs));
        const size_t mat = (size_t)pvs_round_up(ix->n, 32) * ld;  // floats of one [rows][ld] matrix
        HIP_TRY(pvs_scratch_alloc((void **)&d_keys, mat * 4 * (fold_in_scan ? 2 : 1)));
        a.dense_out = d_keys;
        a.dense_ld = ld;
        a.batch = nb;
        if (fold_in_scan) {
            a.tile_grp = ix->d_tile_grp;
            a.fold_weights = d_w;
            a.fold_mask = d_mask;
            a.fold_out = (double *)d_lo;  // (MODE 5: a float matrix)
            a.fold_ld = ld;
            a.fold_agg = agg;
            a.fold_bucket = d_bmin;
            a.fold_hi_off = mat;
            a.fold_bucket_lo = d_lmin;
            if (ix->n_straddlers) HIP_TRY(pvs_scratch_alloc((void **)&d_lo_list, (size_t)ix->n_straddlers * ld * 4));
        }
        hipLaunchKernelGGL(k_certify_init, dim3(512), dim3(256), 0, s, d_bmin, (ld + nb) * BUCKETS + ld * (BUCKETS / 2), d_small, (uint32_t)(n_small + n_pad), d_bits, (uint32_t)(bits_bytes / 4));
        HIP_TRY(hipGetLastError());
        if (!span_bound(ix, c, 1, ix->n, &a.ev_start, &a.ev_stop)) a.ev_start = a.ev_stop = nullptr;
        HIP_TRY(pvs_launch_scan(a, s));
        b.keys = d_keys;
        if (fold_in_scan) {
            hipLaunchKernelGGL(k_query_flags, dim3(1), dim3(128), 0, s, b);
            if (ix->n_straddlers)
                hipLaunchKernelGGL(k_spill_bounds, dim3((ix->n_straddlers + lanes_rows - 1) / lanes_rows), dim3(256), 0, s, b, ix->d_straddlers, ix->n_straddlers, d_keys,
                                   d_keys + mat, nbp, nbp_log2, d_lo_list);
        } else if (ix->groups_are_runs && ix->d_row_gidx) {
            const uint64_t blocks = (ix->n + RUN_ROWS - 1) / RUN_ROWS;
            hipLaunchKernelGGL(k_run_bounds, dim3((unsigned)((blocks + lanes_rows - 1) / lanes_rows)), dim3(256), 0, s, b, ix->d_row_gidx, ix->n, nbp, nbp_log2);
        } else {
            hipLaunchKernelGGL(k_group_bounds, dim3((unsigned)(((uint64_t)G + lanes_rows - 1) / lanes_rows)), dim3(256), 0, s, b, nbp, nbp_log2);
        }
        HIP_TRY(hipGetLastError());
        // 3. the thresholds, 4. every query's candidate files and their union
        uint32_t *d_bmin_t = d_bmin + (size_t)ld * BUCKETS;
        const uint32_t merge = k <= 256 ? 4u : (k <= 512 ? 2u : 1u);  // (>= 16 buckets per page entry stay)
        hipLaunchKernelGGL(k_bucket_transpose, dim3((nb * (BUCKETS / merge) + 255) / 256), dim3(256), 0, s, d_bmin, nb, ld, d_bmin_t, merge);
        HIP_TRY(pvs_launch_kth((const float *)d_bmin_t, BUCKETS / merge, nb, k, d_thr, s));
        if (fold_in_scan) {
            // (the files folded in the scan: only the buckets whose smallest lower bound is at or below the threshold; the tile-crossing
            //  files: their compact matrix)
            hipLaunchKernelGGL(k_candidates_tiles, dim3(a.grid * rt_scan), dim3(256), 0, s, d_lmin, d_lo, nb, ld, d_thr, ix->d_tile_grp, ix->n, rt_scan, a.grid, n_wgtiles, d_qcnt,
                               d_qlist, pvs_dbg(PVS_DBG_FLOAT_CERTIFY_TRACE) ? d_ucnt + 2 : (uint32_t *)nullptr);
            if (ix->n_straddlers)
                hipLaunchKernelGGL(k_candidates, dim3((unsigned)(((uint64_t)ix->n_straddlers + lanes_rows * 8 - 1) / (lanes_rows * 8))), dim3(256), 0, s, d_lo_list, nb, ld, nbp,
                                   nbp_log2, ix->n_straddlers, d_thr, d_qcnt, d_qlist, ix->d_straddlers);
        } else {
            hipLaunchKernelGGL(k_candidates, dim3((unsigned)(((uint64_t)G + lanes_rows * 8 - 1) / (lanes_rows * 8))), dim3(256), 0, s, d_lo, nb, ld, nbp, nbp_log2, G, d_thr, d_qcnt,
                               d_qlist, (const uint32_t *)nullptr);
        }
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(k_union, dim3(QCAP / 256, nb), dim3(256), 0, s, d_qcnt, d_qlist, nb, d_bits, ix->d_grp_off, ix->d_grp_rows, d_mask, d_ucnt, d_ufiles, d_qtot);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(c.h_cert, d_small, n_small * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        spans_collect(ix, c);
        uint32_t n_redo = 0;
        std::vector<uint8_t> skip(nb, 0);
        for (uint32_t q = 0; q < nb; q++) {
            skip[q] = (h_small[q] || h_small[nb + q] > QCAP) ? 1 : 0;
            n_redo += skip[q];
        }
        const uint32_t *h_u = h_small + 2 * (size_t)nb;
        const uint32_t m_f = h_u[0], m_rows = h_u[1];
        const bool trace = pvs_dbg(PVS_DBG_FLOAT_CERTIFY_TRACE) != 0;
        if (trace) {
            std::vector<float> ht(nb);
            (void)hipMemcpy(ht.data(), d_thr, (size_t)nb * 4, hipMemcpyDeviceToHost);
            fprintf(stderr, "[float certify] n=%llu files=%u nb=%u k=%u metric=%d agg=%d: union of %u candidate files (%u rows), %u queries not certifiable, candidates of q0 %u, thresholds %g %g ..., %u buckets visited\n",
                    (unsigned long long)ix->n, G, nb, k, metric, agg, m_f, m_rows, n_redo, h_small[nb], ht[0], ht[nb > 1 ? 1 : 0], h_u[2]);
        }
        // Not worth it, or not possible: every query uncertifiable; more than a couple of them (each one is a corpus pass of its own
        // through the exact route: the whole chunk through k_exact_wide is cheaper); more files than one LDS ranking takes; so many
        // rows that the rescan is no bargain
        if (n_redo == nb || n_redo > 2 || m_f == 0 || m_f > UCAP || (uint64_t)m_rows * 8 > ix->n) return PVS_OK;
        pvs_dbg_add(PVS_DBG_FLOAT_CERTIFY_ROWS, m_rows);
        // 5. the candidates, exactly
        const size_t qbytes = (size_t)ix->dim * (qdtype == PVS_I8 ? 1 : 4);
        bool done = false;
        PVS_TRY(pvs_sparse_groups_of_files(ix, c, (const uint8_t *)d_queries + (size_t)q0 * qbytes, qdtype, nb, k, metric, agg, d_w, d_mask, d_ufiles, m_f, m_rows, skip.data(),
                                           out_groups, out_values, out_count, &done));
        if (trace) fprintf(stderr, "[float certify] exact stage over %u files, %u rows: %s\n", m_f, m_rows, done ? "answered" : "handed back");
        if (done) {
            pvs_dbg_add(PVS_DBG_FLOAT_CERTIFY_QUERIES, nb - n_redo);
            *redo = skip;  // (their slots hold pages over the wrong files: the caller overwrites them)
            *handled = true;
        }
        return PVS_OK;
    };
    pvs_status st = body();
    if (st != PVS_OK) (void)hipStreamSynchronize(s);
    for (void *p : {(void *)d_keys, (void *)d_lo, (void *)d_bmin, (void *)d_thr, (void *)d_small, (void *)d_qlist, (void *)d_bits, (void *)d_ufiles, (void *)d_lo_list}) pvs_scratch_free_on(p, s);
    return st;
}
