// pvs_kernels.hpp — host-callable launchers of the HIP kernels (gfx950).
#pragma once
#include <cstring>
#include "pvs_common.hpp"

// ---- utility kernels (pvs_kernels_util.hip)
hipError_t pvs_launch_norm2(int dtype, const uint8_t *rows, uint32_t stride, uint32_t dim, uint64_t row0, uint64_t n,
                            float *norm2, float *rnorm, hipStream_t s);
hipError_t pvs_launch_fill_f32(float *p, uint64_t n, float v, hipStream_t s);
// (re)builds the scan's row-scalar records (PVS_AUX_REC floats per 32-row tile) of every tile that overlaps rows [row0, row0+n)
hipError_t pvs_launch_scan_aux(const float *norm2, const float *rnorm, uint64_t row0, uint64_t n, float *scan_cos, float *scan_l2,
                               hipStream_t s);
// dense [n][dim] rows -> tiled index rows row0..: mode 0 = quantize_int8 from f32, 1 = f16 from f32, 2 = copy
hipError_t pvs_launch_rows_ingest(int mode, const void *src, uint32_t dim, uint32_t esz, uint64_t row0, uint64_t n, float scale,
                                  uint8_t *rows, uint32_t stride, hipStream_t s);
hipError_t pvs_launch_rows_gather(const uint8_t *rows, uint32_t stride, uint32_t row_bytes, uint64_t row0, uint64_t n, uint8_t *dst,
                                  hipStream_t s);
hipError_t pvs_launch_pick_rows(const void *src, uint32_t row_bytes, const uint32_t *idx, uint64_t m, void *dst, hipStream_t s);
// stored rows rows[idx[i]] (tiled layout) -> a dense query batch out[i][dim]: int8 codes as they are, f16 / f32 rows as f32 (similar_to:
// the target item's vectors never leave the device); idx may live in pinned host memory
hipError_t pvs_launch_rows_to_queries(int dtype, const uint8_t *rows, uint32_t stride, uint32_t dim, const uint32_t *idx, uint32_t m, void *out, hipStream_t s);
hipError_t pvs_launch_take_rows(const void *in, uint32_t elem_bytes, const uint32_t *global_row, uint64_t n_local, void *out, hipStream_t s);
hipError_t pvs_launch_quantize_flat(const float *src, uint64_t n, float scale, int8_t *dst, hipStream_t s);
hipError_t pvs_launch_absmax(const float *src, uint64_t n, float *d_out_bits, hipStream_t s);
hipError_t pvs_launch_synth(uint64_t seed, uint64_t row0, uint64_t n, uint32_t dim, float *out, hipStream_t s);
hipError_t pvs_launch_synth_clustered(uint64_t seed, uint64_t row0, uint64_t n, uint32_t dim, float *out, hipStream_t s);  // clustered, anisotropic, duplicate-rich rows
hipError_t pvs_launch_iota_ids(int64_t *ids, uint64_t n, int64_t base, hipStream_t s);

// queries: [batch][dim] of qdtype (PVS_F32 or PVS_I8).  Produces, per query:
//   qmat  [batch_pad][stride] scan operand in the index dtype (zero padded),
//   qexact[batch][dim]        what the exact rerank reads (i8 codes or f32),
//   qinfo [batch_pad]         per-query constants (padding queries get NaN-proof zeros).
hipError_t pvs_launch_prep_queries(int index_dtype, int qdtype, const void *queries, uint32_t batch,
                                   uint32_t batch_pad, uint32_t dim, uint32_t stride, float scale,
                                   int metric, uint8_t *qmat, void *qexact, QInfo *qinfo, uint32_t *need_dense, hipStream_t s);

// ---- exact per-row distances (pvs_dense_exact.hip): the reference's dist_{cte}.d for `nq` prepared
// queries (qexact [nq][dim] int8 codes or f32, qinfo [nq]) -> out[row * out_ld + out_col + q].
// Sequential f32 accumulation per row, rows streamed HBM -> LDS; up to PVS_DENSE_NQ queries share a pass
// (pairs of queries on the packed f32 pipe at 8).
// qpad_scratch: pvs_dense_exact_scratch_bytes() of device memory, reused launch after launch on `s`.
constexpr uint32_t PVS_DENSE_NQ = 8;  // float rows; int8 rows (the out-of-range fallback) take 4
uint64_t pvs_dense_exact_scratch_bytes(uint32_t stride, uint32_t esz);
// exactly 8 float queries, two rows per lane (pvs_dense_exact2.hip); qpad / ctr: pvs_launch_dense_exact's scratch, prepared by it
bool pvs_dense_exact2_fits(uint32_t stride, uint32_t esz);
hipError_t pvs_launch_dense_exact2(int dtype, int metric, const uint8_t *rows, uint32_t stride, uint64_t n, const float *norm2, const float *qpad, uint32_t *ctr,
                                   const QInfo *qinfo, float *out, uint32_t out_ld, uint32_t out_col, uint32_t n_cu, hipStream_t s);
// 9..32 queries over float rows in one pass, four rows per lane (pvs_exact_wide.hip)
constexpr uint32_t PVS_EXACT_WIDE_NQ = 32;
uint64_t pvs_exact_wide_scratch_bytes(uint32_t stride, uint32_t esz);
uint32_t pvs_exact_wide_fit(uint32_t stride, uint32_t esz);  // queries per pass for this row pitch: 32, 16 or 0 (too wide for LDS)
hipError_t pvs_launch_exact_wide(int dtype, int metric, const uint8_t *rows, uint32_t stride, uint32_t dim, uint64_t n, const float *norm2,
                                 const float *queries, const QInfo *qinfo, uint32_t nq, float *qT_scratch, float *out, uint32_t out_ld, uint32_t out_col,
                                 uint32_t n_cu, hipStream_t s);
hipError_t pvs_launch_dense_exact(int dtype, int metric, const uint8_t *rows, uint32_t stride, uint32_t dim, uint64_t n,
                                  const float *norm2, const void *qexact, const QInfo *qinfo, uint32_t nq, float *qpad_scratch,
                                  float *out, uint32_t out_ld, uint32_t out_col, uint32_t n_cu, hipStream_t s);

// ---- one launch for 1..8 queries (pvs_direct.hip): exact distance of every row in the reference's in-order arithmetic + the pages,
// selected while the rows stream; flags as pass C writes them (0 done, 3 page ends in NULL rows, 1 dense path)
constexpr uint32_t PVS_DIRECT_MAX_K = 256;
constexpr uint32_t PVS_DIRECT_MAX_NQ = 8;            // queries per launch (int8 rows; float rows: 4)
constexpr uint64_t PVS_DIRECT_CROSSOVER_MB = 8192;  // rows x pitch up to which one query takes the one-launch search (search_enqueue)
struct DirectArgs {
    int dtype, metric;
    const uint8_t *rows;
    const float *norm2;
    const int64_t *ids;
    uint32_t stride, dim;
    uint64_t n_rows;
    const void *qexact;  // [nq][dim] int8 codes (int8 index) or f32: SearchCtx::d_qexact after prep_chunk
    const QInfo *qinfo;  // [nq]
    const uint32_t *trank = nullptr, *tinv = nullptr;
    const uint8_t *mask = nullptr;  // optional candidate mask [n_rows]: a row whose byte is 0 takes part in nothing
    uint32_t k;
    uint32_t nq = 1;     // queries (pvs_direct_supported says how many a (dtype, pitch, k) takes)
    void *work;          // pvs_direct_work_bytes(n_cu), zeroed once
    int64_t *out_ids;    // [nq][k]
    float *out_dist;
    uint32_t *out_count, *need_dense, *h_flags, *h_seen;  // [nq]
    // optional mirror of the pages in pinned host memory (a page is written when it is complete: flag 0), so that a host caller
    // needs no copy back
    int64_t *h_out_ids = nullptr;  // [nq][k]
    float *h_out_dist = nullptr;
    uint32_t *h_out_count = nullptr;  // [nq]
    uint32_t *h_out_rows = nullptr;  // [nq][k] ... and the stored-row number of every entry (per-item callers look their groups up by row)
    int null_ok = 0;
    uint32_t n_cu;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
};
bool pvs_direct_supported(int dtype, uint32_t stride, uint32_t esz, uint32_t k, uint32_t nq);
uint64_t pvs_direct_work_bytes(uint32_t n_cu);
hipError_t pvs_launch_direct_topk(const DirectArgs &d, hipStream_t s);

// ---- filter scan (pvs_kernels_scan.hip)
constexpr uint32_t PVS_FLOAT_BUCKETS = 16384;  // per query: minima of the files' upper bounds over disjoint sets of files (file slot mod this) -> k_kth
struct ScanArgs {
    int dtype, metric;
    uint32_t kslabs;        // stride / 256
    uint32_t qgroups;       // batch_pad / 32 in {1,2,4} (and 8 where pvs_scan_max_batch() is 256: k_scan_wide)
    const uint8_t *rows;
    const float *aux;       // the metric's row-scalar stream in tile records (k_scan_aux): 1/|a| (cosine) or |a|^2 (L2)
    uint32_t stride;
    uint64_t n_rows;        // valid rows
    const uint8_t *qmat;
    const QInfo *qinfo;
    int mode;               // 0 = group minima of the upper bound (threshold pass), 1 = filter, 2 = dense exact (int8), 3 = dense exact folded per group, 4 = dense scan KEYS (float rows: dense_out[row][query] = key, error eA + eR |a|^2), 5 = brackets of the distances folded per file (float rows, files that are runs)
    float *dense_out = nullptr;       // mode 2: [n_rows][dense_ld]
    uint32_t *dense_flag = nullptr;   // mode 2
    uint32_t dense_ld = 0, batch = 0;
    uint32_t tile_step;     // process WG tiles 0, step, 2*step, ...
    uint32_t grid;          // workgroups
    float *gmin;            // mode 0: [batch_pad][groups_per_query]
    uint32_t groups_per_query;
    uint32_t gmin_per_lane = 16;  // mode 0: 1..pvs_scan_gmin_max (power of two); groups_per_query = grid * pvs_scan_segs_per_stream * gmin_per_lane
    const float *thr;       // mode 1: [batch_pad]
    // mode 1: candidates go to per-(segment, query) lists: segment = one wave row of one workgroup stream (grid * RT of them)
    uint2 *seg = nullptr;          // [n_segments][batch_pad][PVS_SEG_CAP] = (row, key bits)
    uint32_t *seg_cnt = nullptr;   // [batch_pad][n_segments] fill counts, written by the scan (above PVS_SEG_CAP = overflowed)
    uint32_t n_segments = 0;       // grid * pvs_scan_segs_per_stream
    uint2 *flat = nullptr;         // mode 1, segment-overflow rerun: candidates appended to [batch_pad][flat_cap] lists (ScanK.flat)
    uint32_t *flat_cnt = nullptr;  // [batch_pad], zeroed by the caller
    uint32_t flat_cap = 0;
    // mode 3: the per-group fold (ScanK.tile_grp, pvs_scan_dispatch.hpp): groups inside a 32-row tile are aggregated in the
    // scorer's epilogue into fold_out[group][fold_ld]; rows of groups that cross a tile boundary go to dense_out as usual
    const uint4 *tile_grp = nullptr;
    const float *fold_weights = nullptr;
    const uint8_t *fold_mask = nullptr;
    double *fold_out = nullptr;
    uint32_t fold_ld = 0;
    int fold_agg = 0;
    uint32_t *fold_bucket = nullptr;  // mode 5 (see ScanK)
    uint64_t fold_hi_off = 0;
    float *fold_bucket_lo = nullptr;  // mode 5: the minima of the LOWER bounds over the same buckets [PVS_FLOAT_BUCKETS / 2][fold_ld] (which buckets can hold a candidate)
    // optional events bound to the dispatch itself (hipExtLaunchKernelGGL: start / stop timestamps of THIS kernel, and something a
    // second stream can wait for, without a marker packet in the queue)
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
};
bool pvs_scan_supported(int dtype, uint32_t kslabs);
bool pvs_scan_fold5_supported(int dtype, uint32_t kslabs);  // mode 5 has an instance for this pitch
// geometry of the filter passes (modes 0 / 1) of a shape — it depends on which kernel serves them (pvs_scan_is_wide)
bool pvs_scan_is_wide(int dtype, uint32_t qgroups, uint32_t kslabs);
uint32_t pvs_scan_wg_rows(int dtype, uint32_t qgroups, uint32_t kslabs);  // rows per workgroup tile
uint32_t pvs_scan_row_tiles(uint32_t qgroups);  // k_scan's RT: 32-row sub-tiles per workgroup
uint32_t pvs_scan_segs_per_stream(int dtype, uint32_t qgroups, uint32_t kslabs);  // candidate segments (= lanes per query holding group minima) per workgroup stream
uint32_t pvs_scan_seg_cap(int dtype, uint32_t qgroups, uint32_t kslabs);          // slots per (segment, query)
uint32_t pvs_scan_gmin_max(int dtype, uint32_t qgroups, uint32_t kslabs);         // pass A: minima one lane can supply per query
uint32_t pvs_scan_wg_per_cu(int dtype, uint32_t qgroups, uint32_t kslabs);        // resident workgroups per CU
uint32_t pvs_scan_max_batch(int dtype, uint32_t kslabs);  // queries one pass can hold: 256 (int8, two groups per wave) or 128
hipError_t pvs_launch_scan(const ScanArgs &a, hipStream_t s);

// thr[q] = -inf (no row passes any filter test) for every query whose flag is not 2: the segment-overflow rerun of a chunk emits
// candidates only for the queries pass C handed back for it
hipError_t pvs_launch_void_thresholds(float *thr, const uint32_t *need_dense, uint32_t n, hipStream_t s);

// k-th smallest (1-based) of vals[q][0..per_query), +inf when fewer than k finite values
// qinfo / metric (optional): a query that makes every distance NULL (pvs_query_all_null) gets -inf — pass B emits nothing for it
// and pass C hands it straight to the NULL-tail step
hipError_t pvs_launch_kth(const float *vals, uint32_t per_query, uint32_t batch, uint32_t k, float *out,
                          hipStream_t s, const QInfo *qinfo = nullptr, int metric = 0);

struct FinalizeArgs {
    int dtype, metric;
    const uint8_t *rows;
    const float *norm2;
    const int64_t *ids;
    uint32_t stride, dim;
    uint64_t n_rows;
    const void *qexact;     // [batch][dim] i8 or f32
    const QInfo *qinfo;
    const uint2 *seg;       // the scan's candidate segments and their fill counts (ScanArgs)
    const uint32_t *seg_cnt;
    uint32_t n_segments, seg_queries;
    uint32_t seg_cap = PVS_SEG_CAP;  // slots per (segment, query) as the scan wrote them (pvs_scan_seg_cap)
    uint2 *cand;            // [batch][cand_cap] scratch: each query's segments gathered into one list
    uint32_t cand_cap;
    uint32_t batch, k;
    int64_t *out_ids;       // [batch][k]
    float *out_dist;        // [batch][k]
    uint32_t *out_count;    // [batch]
    uint32_t *need_dense;   // [batch] 1 = this query must be answered by the dense path
    uint32_t *cand_seen = nullptr;  // [batch] (optional) candidates the scan emitted for the query (pvs_stats.last_candidates)
    // segment-overflow rerun: the scan already wrote flat lists into `cand` (ScanArgs.flat); only queries whose need_dense is 2
    // are finalised, the others keep the page they have
    const uint32_t *flat_cnt = nullptr;
    // [batch] the thresholds pass B ran with, when they were taken BELOW the k-th sample value (search_enqueue: j-th of a smaller
    // sample): pass C then proves that k rows lie at or below T (k-th smallest upper bound <= T) or hands the query back
    const float *thr = nullptr;
    // A page that ends in NULL rows (cosine): when the index's NULL set is query-independent (null_ok) and the threshold pass B ran
    // with was +inf (thr_all[q]: every row with a comparable key was emitted), pass C writes the finite part of the page and hands
    // the query back with flag 3 — the host appends the tail from the NULL list (pvs_launch_null_tails) — instead of flag 1 (dense)
    const float *thr_all = nullptr;
    int null_ok = 0;
    // pinned host mirrors of need_dense / cand_seen: the kernel writes its verdicts straight into host memory, so that no copy
    // kernel (4-5 us each, plus a launch gap) follows every search just to fetch two words per query
    uint32_t *h_flags = nullptr, *h_seen = nullptr;
    // second sort key (pvs_index_set_order_keys): ties on the distance are ordered by tie rank instead of by row
    const uint32_t *trank = nullptr, *tinv = nullptr;
    // Optional global-memory work area: with it (int8 rows) pass C keeps its bound keys, survivor list and large sorts in HBM/L2
    // and needs ~6 KB of LDS instead of ~104 KB, so it can run NEXT TO the scan of another search (k_scan's two workgroups per
    // CU leave 7.8 KB of LDS free).  Used when an index runs its searches on several streams.
    uint32_t *w_ub = nullptr;             // [batch][cand_cap]
    uint32_t *w_surv = nullptr;           // [batch][PVS_SURV_CAP]
    unsigned long long *w_sort = nullptr; // [batch][PVS_SURV_CAP]
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;  // bound to the dispatch (ScanArgs.ev_start)
};
hipError_t pvs_launch_finalize(const FinalizeArgs &a, hipStream_t s);

// ---- dense score + sort (pvs_dense.hip)
struct DenseWork {
    float *d_dist = nullptr;       // [n]
    uint32_t *d_keys_in = nullptr, *d_keys_out = nullptr, *d_vals_in = nullptr, *d_vals_out = nullptr;  // the full sort's buffers (allocated when a page needs it)
    void *d_temp = nullptr;
    size_t temp_bytes = 0;
    uint64_t cap_rows = 0, sort_rows = 0;
    // page first (round 5): a sampled threshold admits ~1.3 k rows, only those are sorted
    uint32_t *d_sample = nullptr, *d_sample_out = nullptr, *d_ctl = nullptr, *h_ctl = nullptr;  // ctl: [0] threshold key, [1] admitted rows
    unsigned long long *d_comp = nullptr, *d_comp_out = nullptr;
    void *d_temp_page = nullptr;
    size_t temp_page_bytes = 0;
    uint32_t comp_cap = 0;
};
pvs_status pvs_dense_reserve(DenseWork &w, uint64_t n);
void pvs_dense_release(DenseWork &w);
// sorts d_dist[0..n) by (distance, row) and writes the first k (ids via ids[]) for one query
// mask (optional, [n] bytes on the device): rows with mask == 0 are not candidates at all
struct DenseBounds {  // apply_sort_bounds on the distance column: rows outside (gt, lt) are not candidates at all
    int have_gt = 0, have_lt = 0;
    double gt = 0.0, lt = 0.0;
};
pvs_status pvs_dense_topk(DenseWork &w, uint64_t n, uint32_t k, const int64_t *ids, int64_t *out_ids,
                          float *out_dist, uint32_t *out_count, hipStream_t s, const uint8_t *mask = nullptr,
                          DenseBounds bounds = DenseBounds(), const uint32_t *tinv = nullptr);
// tie ranks of the second sort key: d_tinv[t] = the row at position t of (key DESC, row ASC), d_trank its inverse (synchronous)
pvs_status pvs_build_tie_ranks(const int64_t *d_keys, uint64_t n, uint32_t *d_trank, uint32_t *d_tinv, hipStream_t s);
// aux / out: the scan's row-scalar stream in tile records (cap/32 * PVS_AUX_REC floats); rows outside the mask get a NaN
// scalar (a NaN scalar makes every filter comparison of the row false)
hipError_t pvs_launch_mask_aux(const float *aux, const uint8_t *mask, uint64_t n, uint64_t cap, float *out, hipStream_t s);

// ---- page 1 of many dense columns at once (pvs_select.hip): m [n][ld] f32, column j -> query slot qmap[j] (or j)
bool pvs_select_supported(uint32_t k);
// tinv (optional): tie order of the rows (second sort key); ties are then cut and ordered by position in it instead of by row
pvs_status pvs_select_topk(const float *m, uint64_t n, uint32_t ld, uint32_t nq, uint32_t k, const uint8_t *mask, const int64_t *ids,
                           const uint32_t *d_qmap, int64_t *out_ids, float *out_dist, uint32_t *out_count, hipStream_t s,
                           const uint32_t *tinv = nullptr);

// exact int8 distances of every row for 1..4 queries, straight from HBM with v_dot4 (pvs_score_direct.hip): out[row * ld + q]
hipError_t pvs_launch_score_i8_direct(int metric, const uint8_t *rows, uint32_t stride, uint32_t dim, uint64_t n_rows, const float *norm2,
                                      const void *qexact, const QInfo *qinfo, uint32_t nb, float *out, uint32_t ld, uint32_t *flag,
                                      uint32_t n_cu, hipStream_t s);
// ... with the per-item fold in the tile epilogue (the contract of k_scan MODE 3, for 1..4 queries)
hipError_t pvs_launch_score_i8_fold(int metric, const uint8_t *rows, uint32_t stride, uint32_t dim, uint64_t n_rows, const float *norm2, const void *qexact,
                                    const QInfo *qinfo, uint32_t nb, float *out, uint32_t ld, uint32_t *flag, const uint4 *tile_grp, const float *weights,
                                    const uint8_t *mask, double *fold_out, uint32_t fold_ld, int agg, uint32_t n_cu, hipStream_t s);
// merge of per-shard pages on the device: in [world][batch][k] -> out [batch][k]
hipError_t pvs_launch_merge(const int64_t *ids, const float *dist, const uint32_t *counts, uint32_t world,
                            uint32_t batch, uint32_t k, int64_t *out_ids, float *out_dist,
                            uint32_t *out_count, hipStream_t s, const int64_t *keys = nullptr);
// A rank's page as ONE buffer, so that the shard exchange is one all-gather (and one peer copy inside a multi-device index):
// [ids i64 x batch*k | dist f32 x batch*k | counts u32 x batch | flags u32 x batch | order keys i64 x batch*k], padded to 16 bytes.
// flags[q]: the query's hand-back code (pass C) in the low bits; bit 31 = the keys section is valid (the shard carries
// pvs_index_set_order_keys keys for all its rows).  The merge breaks distance ties by key DESC when EVERY shard's page says so.
constexpr uint32_t PVS_PAGE_KEYED = 0x80000000u;
// a rank that failed locally before the exchange still sends its record, with this bit in every flag word: all ranks fail the
// search together (0x40404040 is what a byte fill of the flag words with 0x40 leaves; need_dense values are 0..3)
constexpr uint32_t PVS_PAGE_FAILED = 0x40000000u;
inline size_t pvs_page_record_off_dist(uint32_t batch, uint32_t k) { return (size_t)batch * k * 8; }
inline size_t pvs_page_record_off_cnt(uint32_t batch, uint32_t k) { return (size_t)batch * k * 12; }
inline size_t pvs_page_record_off_flags(uint32_t batch, uint32_t k) { return (size_t)batch * k * 12 + (size_t)batch * 4; }
inline size_t pvs_page_record_off_keys(uint32_t batch, uint32_t k) { return ((size_t)batch * k * 12 + (size_t)batch * 8 + 15) / 16 * 16; }
inline size_t pvs_page_record_bytes(uint32_t batch, uint32_t k) { return pvs_page_record_off_keys(batch, k) + ((size_t)batch * k * 8 + 15) / 16 * 16; }
// h_flags (optional, pinned host memory [world][batch]): the records' flag words, written by the merge itself
hipError_t pvs_launch_merge_packed(const uint8_t *all_rec, size_t rec_bytes, uint32_t world, uint32_t batch, uint32_t k, int64_t *out_ids,
                                   float *out_dist, uint32_t *out_count, hipStream_t s, uint32_t *h_flags = nullptr);
// Completes a page record whose ids / dist / counts are written: flags[q] = need_dense[q] (| PVS_PAGE_KEYED), and with order keys
// (d_order_keys != nullptr: one per row, rows in ascending id order d_ids[0..n)) the key of every page entry, found by its id.
hipError_t pvs_launch_page_finish(uint8_t *rec, uint32_t batch, uint32_t k, const uint32_t *need_dense, const int64_t *d_ids, uint64_t n,
                                  const int64_t *d_order_keys, hipStream_t s);

// ---- per-item aggregation and ranking (pvs_groups.hip)
struct GroupWork {
    unsigned long long *keys_in = nullptr, *keys_out = nullptr;
    uint32_t *idx_in = nullptr, *idx_out = nullptr;
    void *temp = nullptr;
    size_t temp_bytes = 0;
    uint32_t cap = 0;
};
// similar_to confidence weights and cross-modal gates (device pointers; on == 0: plain fan-out).  The targets' own values travel
// as [fanout] arrays, so the target rows need not be rows of the index being aggregated (a multi-device index: another shard's).
struct FanoutWeights {
    uint32_t on = 0;
    const double *t_conf = nullptr, *t_lang = nullptr;  // [fanout] confidence / language_confidence of each target vector, NaN = NULL
    const uint8_t *t_kind = nullptr;                     // [fanout] PVS_KIND_* of each target vector
    const double *conf = nullptr;     // [rows] confidence, NaN = NULL
    const double *lang = nullptr;     // [rows] language_confidence, NaN = NULL
    double cw = 0.0, lw = 0.0;
    const uint8_t *kind = nullptr;    // [rows] PVS_KIND_*; gates below apply when set
    uint32_t skip_i2i = 0, skip_t2t = 0;
    // rows left out of the join (the target item's own rows) as up to four ranges [lo, hi] instead of a byte per row: an item's
    // vectors are stored side by side, and zeroing + marking a byte map of the whole index cost two fills per similar_to
    uint32_t n_ranges = 0, r_lo[4] = {0, 0, 0, 0}, r_hi[4] = {0, 0, 0, 0};
};
// `exclude` [rows] with skip_when == 1: rows whose byte is non-zero are left out (similar_to: the target's own rows);
// with skip_when == 0 it is a candidate mask: rows whose byte is zero are left out, and a group without any candidate
// row gets the value PVS_GROUP_ABSENT, which pvs_group_rank never emits.
constexpr unsigned long long PVS_GROUP_ABSENT = 0x7ff8a5a5a5a5a5a5ull;  // a NaN payload no arithmetic produces
hipError_t pvs_launch_group_aggregate(const float *dist, uint32_t ld, uint32_t n_cols, uint32_t fanout, const uint32_t *grp_off,
                                      const uint32_t *grp_rows, uint32_t n_groups, const float *weights, const uint8_t *exclude,
                                      int agg, double *out, hipStream_t s, FanoutWeights fw = FanoutWeights(), uint32_t skip_when = 1,
                                      bool rows_are_runs = false);  // rows_are_runs: grp_rows[e] == e (the 8-column kernel skips the indirection)
pvs_status pvs_group_rank(const double *d_vals, const int64_t *d_group_ids, uint32_t n_groups, uint32_t k, GroupWork &w,
                          int64_t *d_out_groups, double *d_out_vals, uint32_t *d_out_count, hipStream_t s, const uint32_t *g_tinv = nullptr);
// The groups of `list` (indices into the group CSR) only, from the dense matrix, into the group-major output out_t[group][ld_out]:
// the finishing step of the fused per-item scorer (groups that cross a 32-row tile boundary)
hipError_t pvs_launch_group_aggregate_list(const float *dist, uint32_t ld, uint32_t n_cols, const uint32_t *grp_off, const uint32_t *grp_rows,
                                           const uint32_t *list, uint32_t n_list, const float *weights, const uint8_t *exclude, int agg, double *out_t,
                                           uint32_t ld_out, hipStream_t s, uint32_t skip_when);
// page-first ranking of group-major values on the device (pvs_groups.hip): out_flag[col] = 1 -> out_groups / out_values [col][k] hold the
// column's first k groups; 0 -> rank that column by the full sort.  d_grp_trank / d_grp_tinv: the groups' tie order or nullptr.
size_t pvs_gm_rank_work_bytes(uint32_t ncol);
bool pvs_gm_rank_supported(uint32_t n_groups, uint32_t ncol, uint32_t k);
hipError_t pvs_gm_rank(const double *d_vals_t, uint32_t n_groups, uint32_t ncol, uint32_t k, const int64_t *d_gids, const uint32_t *d_grp_trank,
                       const uint32_t *d_grp_tinv, void *d_work, int64_t *out_groups, double *out_values, uint32_t *out_flag, hipStream_t s,
                       bool column_major = false);  // column_major: d_vals_t is [ncol][n_groups]
// the same for the few sub-groups of a candidate list, all of them in one LDS sort per column (column-major values [ncol][n_sub])
bool pvs_sub_rank_supported(uint32_t n_sub);
hipError_t pvs_sub_rank(const double *d_vals, uint32_t n_sub, uint32_t ncol, uint32_t k, const uint32_t *d_sub_slot, const int64_t *d_gids, const uint32_t *d_grp_trank,
                        const uint32_t *d_grp_tinv, void *d_work, int64_t *out_groups, double *out_values, uint32_t *out_flag, uint32_t *out_cnt, hipStream_t s);
// group-major values [n_groups][ncol] -> column-major [ncol][n_groups] (what pvs_group_rank and the page keys index)
hipError_t pvs_launch_page_group_keys(const int64_t *page_groups, const uint32_t *counts, uint32_t batch, uint32_t k, const int64_t *grp_ids, uint32_t G,
                                      const int64_t *grp_key, int64_t *out_keys, hipStream_t s);
bool pvs_merge_group_pages_supported(uint32_t S, uint32_t k);
hipError_t pvs_launch_merge_group_pages(const int64_t *g, const double *v, const int64_t *key, const uint32_t *cnt, uint32_t S, uint32_t batch, uint32_t k,
                                        int64_t *out_g, double *out_v, uint32_t *out_c, hipStream_t s);
hipError_t pvs_launch_group_transpose(const double *vals_t, uint32_t n_groups, uint32_t ncol, double *vals, hipStream_t s);
void pvs_group_work_release(GroupWork &w);
// order-preserving u64 keys of one column of group values, in group order: value asc, NULL aggregates (~0 - 1) after every value,
// absent groups (~0) last.  The value is recovered from its key (pvs_group_value_of_key).
hipError_t pvs_group_page_keys(const double *d_vals, uint32_t n_groups, unsigned long long *d_keys, hipStream_t s);
hipError_t pvs_group_page_keys_t(const double *d_vals_t, uint32_t n_groups, uint32_t ncol, unsigned long long *d_keys, hipStream_t s);  // group-major in, column-major keys out
inline double pvs_group_value_of_key(unsigned long long k) {
    if (k >= ~0ull - 1) return __builtin_nan("");
    const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    double d;
    memcpy(&d, &b, 8);
    return d;
}

// ---- reciprocal-rank fusion of several ranked branches (pvs_rrf.hip)
constexpr int PVS_RRF_MAX_BRANCHES = 8;
struct PvsRrfParams {
    uint32_t n_branches;
    int32_t k[PVS_RRF_MAX_BRANCHES];
    double w[PVS_RRF_MAX_BRANCHES];
};
// one round of the bounded fusion on the device (pvs_rrf_device.hip): everything behind the window keys, one synchronisation
bool pvs_rrf_round_device_supported(uint32_t nb, uint64_t target, uint32_t k);
size_t pvs_rrf_round_device_work_bytes(uint32_t nb);
size_t pvs_rrf_round_device_out_bytes(uint32_t nb, uint32_t k);
hipError_t pvs_rrf_round_device(const unsigned long long *const *d_keys, const int64_t *const *d_gids, const uint32_t *n, uint32_t nb, const PvsRrfParams &p,
                                uint32_t target, uint32_t k, void *d_work, uint8_t *h_out, hipStream_t s);
void pvs_rrf_round_device_result(const uint8_t *h_out, uint32_t nb, uint32_t k, uint32_t *flags, uint32_t *R, uint32_t *m, uint32_t *n_out, const int64_t **groups,
                                 const double **scores);
// order-independent 64-bit digest of n words of 4 or 8 bytes (synchronous; race hunting, pvs_debug_rrf_digests)
pvs_status pvs_digest_device(const void *d, uint64_t n, int word_bytes, uint64_t *out_host, hipStream_t s);
// bounded fusion: window keys of every group of a branch, a sample of them, the page of groups at or below a key, the keys of
// given groups, and how many groups stand strictly before each of a handful of candidates (one counting pass over the keys)
pvs_status pvs_rrf_window_keys(const double *d_vals, uint32_t n, int descending, unsigned long long *d_keys, hipStream_t s);
pvs_status pvs_rrf_sample_keys(const unsigned long long *d_keys, uint32_t n, uint32_t m, unsigned long long *h_out, hipStream_t s);
// the same for ncol key columns [ncol][n] at once (one round trip each)
pvs_status pvs_rrf_sample_keys_cols(const unsigned long long *d_keys, uint32_t n, uint32_t ncol, uint32_t m, unsigned long long *h_out, hipStream_t s);
pvs_status pvs_rrf_pages_cols(const unsigned long long *d_keys, const int64_t *d_gids, uint32_t n, uint32_t ncol, const unsigned long long *h_thr,
                              uint32_t cap, int64_t *out_gids, unsigned long long *out_keys, uint32_t *out_count, hipStream_t s);
pvs_status pvs_rrf_page(const unsigned long long *d_keys, const int64_t *d_gids, uint32_t n, unsigned long long thr, uint32_t cap,
                        int64_t *out_gids, unsigned long long *out_keys, uint32_t *out_count, hipStream_t s);
pvs_status pvs_rrf_lookup(const unsigned long long *d_keys, const int64_t *d_gids, uint32_t n, const int64_t *cand, uint32_t m,
                          unsigned long long *out_keys, uint8_t *out_present, hipStream_t s);
pvs_status pvs_rrf_count_below(const unsigned long long *d_keys, const int64_t *d_gids, uint32_t n, const unsigned long long *ckeys,
                               const int64_t *cgids, uint32_t m, unsigned long long *out_below, hipStream_t s);
pvs_status pvs_rrf_rank_branch(const double *d_vals, const int64_t *d_gids, uint32_t n, int descending, uint32_t branch,
                               unsigned long long *cat_key, unsigned long long *cat_pay, hipStream_t s);
pvs_status pvs_rrf_fuse_device(unsigned long long *cat_key, unsigned long long *cat_pay, uint64_t total, const PvsRrfParams &p, uint32_t k,
                               int64_t *out_groups, double *out_scores, uint32_t *out_count, hipStream_t s);
