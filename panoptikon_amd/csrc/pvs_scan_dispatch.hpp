// pvs_scan_dispatch.hpp — entry points of the scan-kernel translation units.
#pragma once
#include <hip/hip_ext.h>
#include "pvs_kernels.hpp"
struct ScanK {
    const uint8_t *rows;
    const float *aux;  // row scalars streamed with the tiles, in records of PVS_AUX_REC floats per 32-row tile (k_scan_aux):
                       // 1/|a| (cosine) or |a|^2 (L2), NaN on padding rows, then the tile's min / max
    const uint8_t *qmat;
    const QInfo *qinfo;
    const float *thr;
    float *gmin;
    uint2 *seg;          // MODE 1: candidate segments [n_segments][seg_queries][seg_cap] = (row, key bits)
    uint32_t *seg_cnt;   // MODE 1: fill counts [seg_queries][seg_stride] (a count above seg_cap = the segment overflowed)
    uint32_t seg_queries, seg_cap, seg_stride;
    // MODE 1, rerun after a segment overflowed (candidates clustered in a few tile streams): when `flat` is set every candidate
    // is appended to its query's flat list [query][flat_cap] through an atomic counter instead of going to a segment
    uint2 *flat;
    uint32_t *flat_cnt;
    uint32_t flat_cap;
    uint64_t n_rows;
    uint32_t stride, n_wgtiles, tile_step, groups_per_query, grid;
    uint32_t gmin_per_lane;  // MODE 0: minima written per lane (1, 2, 4, 8 or 16); groups_per_query = grid * RT * 2 * gmin_per_lane
    float *dense_out;        // MODE 2: exact distances, [n_rows][dense_ld] (query-minor)
    uint32_t *dense_flag;    // MODE 2: set when an int8 L2 sum left the exact range (host reruns that batch in order)
    uint32_t dense_ld, batch;
    // MODE 3 = MODE 2 with the per-group fold (per-item search: GROUP BY file_id + rank_aggregate, filters/exact.rs:67-134, in the scorer's
    // epilogue instead of a second pass over an N x batch matrix).  Needs the rows of every group
    // to be one run of consecutive rows.  Per 32-row tile one record {group index of the tile's first row, bit i: row i is the
    // LAST row of its group, bit i: row i's group crosses a tile boundary, unused}.  A group inside one tile is folded in row
    // order by the lane that holds its query (SQLite's KBN sums, bit for bit) and its value written to fold_out[group][query];
    // the rows of a group that crosses a tile boundary are written to dense_out as before (a few percent of the rows) and
    // folded by k_group_aggregate_list afterwards.
    const uint4 *tile_grp;
    const float *fold_weights;  // optional [n_rows]: SUM(d*w)/SUM(w)
    const uint8_t *fold_mask;   // optional [n_rows]: 0 = the row is not a candidate (takes part in nothing)
    double *fold_out;           // [n_groups][fold_ld]
    uint32_t fold_ld;
    int fold_agg;               // PVS_AGG_MIN / MAX / AVG (ignored with weights)
    // MODE 5 (float rows, brackets folded per file: pvs_items_float.hip): fold_out is a float matrix [n_groups][fold_ld] of lower
    // bounds, fold_bucket the minima of the upper bounds [PVS_FLOAT_BUCKETS][fold_ld], fold_hi_off the distance in floats from
    // dense_out (lower ends of the rows of tile-crossing files) to the matrix of their upper ends
    uint32_t *fold_bucket;
    uint64_t fold_hi_off;
    float *fold_bucket_lo;  // minima of the lower bounds over the same buckets: k_candidates_tiles visits only buckets with one at or below the threshold
    // host side only (16 bytes of kernel argument nobody reads): events bound to THIS dispatch by hipExtLaunchKernelGGL — the kernel's
    // start / stop timestamps come from its own completion signal, no marker packet goes into the queue (two hipEventRecord around
    // every kernel of a search cost 30-45 us of a 1.29-ms step at configs[2])
    hipEvent_t ev_start, ev_stop;
};

// one launch site for both forms
#define PVS_SCAN_LAUNCH(kernel, grid, block, lds, stream, karg)                                                           \
    do {                                                                                                                  \
        if ((karg).ev_start || (karg).ev_stop)                                                                            \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, (karg).ev_start, (karg).ev_stop, 0, karg);           \
        else                                                                                                              \
            hipLaunchKernelGGL(kernel, grid, block, lds, stream, karg);                                                   \
    } while (0)

hipError_t pvs_scan_dispatch_i8(const ScanK &k, uint32_t kslabs, uint32_t qg, int metric, int mode, hipStream_t s);
hipError_t pvs_scan_dispatch_f16_small(const ScanK &k, uint32_t kslabs, uint32_t qg, int metric, int mode, hipStream_t s);
hipError_t pvs_scan_dispatch_f16_large(const ScanK &k, uint32_t kslabs, uint32_t qg, int metric, int mode, hipStream_t s);
hipError_t pvs_scan_dispatch_f32_small(const ScanK &k, uint32_t kslabs, uint32_t qg, int metric, int mode, hipStream_t s);
hipError_t pvs_scan_dispatch_f32_mid(const ScanK &k, uint32_t kslabs, uint32_t qg, int metric, int mode, hipStream_t s);
hipError_t pvs_scan_dispatch_f32_large(const ScanK &k, uint32_t kslabs, uint32_t qg, int metric, int mode, hipStream_t s);
hipError_t pvs_scan_dispatch_i8_wide(const ScanK &k, uint32_t kslabs, uint32_t qgroups, int metric, int mode, hipStream_t s);  // pvs_scan_wide.hpp
uint32_t pvs_scan_wide_segs(uint32_t qgroups);  // candidate segments per workgroup stream of k_scan_wide
bool pvs_scan_wide_serves(uint32_t qgroups, uint32_t kslabs, int mode);  // int8: does k_scan_wide serve this pass (else k_scan)
uint32_t pvs_scan_wide_rows(uint32_t kslabs);  // rows per workgroup tile of k_scan_wide
hipError_t pvs_scan_dispatch_i8_large(const ScanK &k, uint32_t kslabs, uint32_t qg, int metric, int mode, hipStream_t s);
hipError_t pvs_scan_dispatch_f16_xl(const ScanK &k, uint32_t kslabs, uint32_t qg, int metric, int mode, hipStream_t s);
hipError_t pvs_scan_dispatch_f32_xl(const ScanK &k, uint32_t kslabs, uint32_t qg, int metric, int mode, hipStream_t s);
