/*
 * pvs.h — C ABI of the MI355X-native vector-similarity scan for Panoptikon
 * ("Panoptikon Vector Scan").  This is the drop-in boundary:
 * plain pointers and sizes, no C++ or torch types, no exceptions across it.
 * A Rust host binds it with an `extern "C"` block.
 *
 * What each entry point replaces in the reference (paths under
 * reasv/panoptikon, panoptikon/src/...):
 *
 *   pvs_search / pvs_score_all   the per-row `vec_distance_cosine` / `vec_distance_L2`
 *                                scalar SQL functions of sqlite-vec 0.1.9, registered at
 *                                db/sql_functions.rs:105-128 and emitted by
 *                                pql/builder/filters/image_embeddings.rs:321-362,
 *                                text_embeddings.rs:386-418, item_similarity.rs:503-521,
 *                                i.e. the `d` column of the MATERIALIZED dist_{cte}
 *                                (filters/exact.rs:106-165) and, for pvs_search, everything
 *                                down to `ORDER BY order_rank ASC ... LIMIT k`
 *                                (pql/builder.rs:578-582, 1043-1223).
 *   pvs_quantize_i8              quantize_int8 / compute_query_quant, db/vector_quants.rs:1489-1503
 *   pvs_absmax / pvs_scale_from_absmax
 *                                blob_absmax / scale_from_absmax / compute_int8_scale_artifact,
 *                                db/vector_quants.rs:1465-1483, 1513-1554
 *   pvs_artifact_scale / pvs_scale_artifact
 *                                artifact_scale / scale_artifact, db/vector_quants.rs:1449-1460
 *   pvs_aggregate                rank_aggregate + GROUP BY file_id, filters/exact.rs:67-80,
 *                                pql/builder.rs:829-835
 *   pvs_search_groups[_sharded]  `GROUP BY file_id` + rank_aggregate over dist_{cte} and the page order,
 *                                filters/exact.rs:67-165, pql/builder.rs:578-582
 *   pvs_similar_to[_ex]          SimilarTo: self-join fan-out, confidence weights, CLIP cross-modal gates,
 *                                filters/item_similarity.rs:84-142, 432-581
 *   pvs_row_number[_dir] / pvs_rrf_fuse / pvs_coalesce_ranks / pvs_coalesce_values
 *                                add_rank_column_expr pql/builder.rs:757-771 and
 *                                build_coalesced_expr pql/builder.rs:1284-1317 (RRF arm and min/max(coalesce) arm)
 *   pvs_sort_bounds / pvs_search_bounded
 *                                apply_sort_bounds (gt / lt on order_rank), pql/builder.rs:781-815
 *   pvs_rrf_search               the OR arm over vector filters: UNION of the branches (pql/builder.rs:638-661),
 *                                per-branch row_number(), RRF score, ORDER BY ... LIMIT k
 *   pvs_index_read_ids / pvs_index_read_rows
 *                                the key / payload columns of dist_{cte} (item_data.id, embeddings.embedding,
 *                                embedding_quants.quant: migrations/index/20250117193000_init.sql:29-33,
 *                                20260730150000_embedding_quants_rowid.sql:32-49)
 *   pvs_search_sharded / pvs_merge_topk / pvs_merge_group_pages
 *                                no reference counterpart (the reference is single-process): SURVEY.md §8e
 *   pvs_npy_to_f32               embedding_from_npy_bytes, pql/embedding_utils.rs:10-76,229-350
 *   pvs_resolve_vector_quant     resolve_vector_quant policy, pql/preprocess.rs:314-446
 *
 * Conventions
 *   - every function returns pvs_status (0 = OK); pvs_last_error() returns a
 *     thread-local message for the last failing call on this thread.
 *   - all buffers are caller-owned; `*_space` says whether a pointer is host or
 *     device (HBM) memory.  Device pointers must belong to the index's device.
 *   - vectors are dense little-endian arrays in component order: f32 (the
 *     reference's storage format, extraction_write.rs:574-616), IEEE f16, or int8
 *     codes (db/vector_quants.rs:1489-1497).
 *   - ordering: (distance ascending, row id ascending); NaN distances (SQL NULL in
 *     the reference) sort last.  The id tie-break is the build's addition
 *     (SURVEY.md §8c); the reference leaves ties unspecified.
 *   - threading: every entry point may be called concurrently from many host threads on
 *     one index, searches AND mutations (the reference runs up to 16 read connections,
 *     db/connection.rs:235,320-357, beside one writer actor, db/index_writer.rs, under
 *     SQLite's snapshot isolation).  The library holds a reader / writer gate per index
 *     (csrc/pvs_gate.hip): pvs_index_add* / remove_rows / replace_rows* / set_order_keys /
 *     set_streams wait for the calls that read the index, complete the stream-ordered
 *     searches still in flight on their owners' behalf (the owner's pvs_wait returns the
 *     parked result), keep new searches out until they are done and cannot be starved by
 *     them.  A search observes the index before a mutation or after it, never a mix; a
 *     stream-ordered search enqueued before a mutation answers over the rows as they were.
 *     Per-row arrays the CALLER keeps (masks, weights) must match the snapshot the caller
 *     means: serialise those calls with the mutations yourself.  Only pvs_index_destroy
 *     (and pvs_index_set_scale before the first add) remain the caller's to order.  With
 *     several ranks (pvs_search_sharded*) mutate collectively: every rank at the same
 *     point of its program.
 */
#ifndef PVS_H
#define PVS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PVS_ABI_VERSION 5

typedef int32_t pvs_status;
enum {
    PVS_OK = 0,
    PVS_ERR_INVALID_ARG = 1,  /* null pointer, bad enum, k < 1 (preprocess.rs:436-446) ... */
    PVS_ERR_DIM_MISMATCH = 2, /* the reference's sqlite-vec SQL error -> 500 (db/pql.rs:18-21) */
    PVS_ERR_DEVICE = 3,       /* HIP runtime failure or no gfx950 device */
    PVS_ERR_OOM = 4,
    PVS_ERR_STATE = 5,        /* e.g. f32 query on an int8 index whose scale is not set */
    PVS_ERR_UNSUPPORTED = 6,
    PVS_ERR_PARSE = 7,        /* malformed .npy (embedding_utils.rs error strings) */
    PVS_ERR_NOT_READY = 8,    /* strict quant selection that cannot be served (preprocess.rs:327-383) */
    PVS_ERR_COMM = 9          /* RCCL failure */
};

typedef enum { PVS_F32 = 0, PVS_F16 = 1, PVS_I8 = 2 } pvs_dtype;
typedef enum { PVS_COSINE = 0, PVS_L2 = 1 } pvs_metric;
/* embedding_types.rs:4-18 DistanceAggregation (+ NONE = per-row results) */
typedef enum { PVS_AGG_NONE = 0, PVS_AGG_MIN = 1, PVS_AGG_MAX = 2, PVS_AGG_AVG = 3 } pvs_agg;
typedef enum { PVS_HOST = 0, PVS_DEVICE = 1 } pvs_space;

typedef struct pvs_index pvs_index;

#define PVS_MAX_DEVICES 16
typedef struct pvs_index_desc {
    uint32_t struct_size;   /* sizeof(pvs_index_desc); the 32-byte ABI v1 prefix (up to id_base) is still accepted */
    int32_t device;         /* HIP device ordinal, -1 = current device (ignored when n_devices > 1) */
    uint32_t dtype;         /* pvs_dtype of the rows resident in HBM */
    uint32_t dim;           /* components per vector (dim = blob_len/4 in the reference) */
    uint64_t capacity_rows; /* rows to reserve up front, over all devices (0 = grow on demand) */
    int64_t id_base;        /* row id of row 0 when pvs_index_add is given row_ids == NULL */
    /* ABI v2 — one host process, several GPUs (the reference host is ONE process with a pool of read
     * connections, db/connection.rs:320-357): the rows shard across `devices`, searches fan out to all shards,
     * per-shard pages travel to devices[0] by peer copies over xGMI and are merged there.
     * n_devices 0 or 1 = single device.  An ordinal may repeat (several shards on one GPU: used by the tests on
     * one-GPU machines).  Row placement is decided by the first pvs_index_add:
     *  - WITH group_ids (then every add must give them): BY GROUP — every row of a group lives on the shard
     *    mix(group id) % n_devices, whatever call it arrives in.  Every entry point of a single-device index is
     *    served and answers the same, bit for bit: pvs_search / pvs_search_device + pvs_wait, pvs_search_filtered,
     *    pvs_score_all / pvs_score_batch (host outputs, global row order), pvs_search_groups[_filtered] with MIN /
     *    MAX / AVG / row weights / candidate masks (row_weights and masks index the GLOBAL rows, in add order),
     *    pvs_similar_to[_ex], pvs_rrf_search (every branch a multi-device index over the same number of devices),
     *    read back, stats.  Device-space rows are staged through the host once (placement is a scatter);
     *  - WITHOUT group_ids (then no add may give them): every add splits into n_devices contiguous pieces.  The
     *    index holds no groups, so the per-item entry points fail as they do on a single-device index without
     *    group ids; everything row-wise above is served.
     * Not served on any multi-device index: the
     * `_sharded` (multi-process) entry points, pvs_rrf_cols.  A pvs_index_add that fails after some shards took
     * their rows leaves the index unusable (PVS_ERR_STATE from every later call): destroy and rebuild it. */
    uint32_t n_devices;
    const int32_t *devices; /* [n_devices] HIP ordinals */
} pvs_index_desc;

typedef struct pvs_stats {
    uint32_t struct_size;
    uint32_t dtype, dim;
    uint64_t rows, capacity_rows;
    uint64_t row_stride_bytes;  /* padded row pitch in HBM */
    uint64_t hbm_bytes;         /* bytes resident for this index */
    float scale;                /* int8 scale artifact (0 when unset) */
    uint64_t searches;          /* pvs_search* calls served */
    uint64_t fast_queries;      /* queries answered by the filter-scan path */
    uint64_t dense_queries;     /* queries answered by the dense score+sort path */
    uint64_t last_candidates;   /* candidates emitted by the last filter scan (all queries) */
    /* since ABI v3: written by pvs_index_stats_ex only (pvs_index_stats writes the fields above and never reads *out) */
    uint64_t rescanned_queries; /* fast-path queries whose chunk went through the scan twice: a candidate segment overflowed
                                 * (ties clustered in a few tile streams) and pass B was rerun into flat per-query lists */
    uint64_t sparse_queries;    /* filtered / row-list queries answered by gather-and-score over the allowed rows only */
    uint64_t null_tail_queries; /* cosine queries whose page ended in NULL rows, completed from the zero-norm row list (no dense pass) */
} pvs_stats;

/* ------------------------------------------------------------------ library */
uint32_t pvs_abi_version(void);
const char *pvs_last_error(void);
/* number of visible gfx950 devices (0 when there is no usable GPU) */
int32_t pvs_device_count(void);

/* -------------------------------------------------------------------- index */
pvs_status pvs_index_create(const pvs_index_desc *desc, pvs_index **out);
void pvs_index_destroy(pvs_index *idx);

/* Appends n rows of the index dtype (f32 / f16 / int8 codes).  row_ids are the
 * reference's item_data.id (embeddings.id); they must be strictly increasing
 * across the whole index (the reference streams rows ORDER BY item_data.id,
 * db/vector_quants.rs:1085-1099) so that "row id ascending" and "row order"
 * coincide.  row_ids == NULL assigns id_base + row index.  group_ids (file_id /
 * item_id, optional) feed pvs_aggregate; rows of one group need not be adjacent. */
pvs_status pvs_index_add(pvs_index *idx, const void *rows, uint64_t n, const int64_t *row_ids,
                         const int64_t *group_ids, pvs_space rows_space);

/* Write side of the codec (backfill_chunk, db/vector_quants.rs:1119-1163): rows
 * arrive as f32 and are converted on the device to the index dtype — int8 via
 * quantize_int8 with the index scale (which must be set), f16 via round-to-
 * nearest-even, f32 unchanged. */
pvs_status pvs_index_add_f32(pvs_index *idx, const float *rows, uint64_t n, const int64_t *row_ids,
                             const int64_t *group_ids, pvs_space rows_space);

/* ABI v4 — rows leave and change without a rebuild.  The reference deletes vectors whenever a file disappears
 * (`embeddings ... ON DELETE CASCADE`, migrations/index/20250117193000_init.sql:29-33; db/files.rs:175-192,
 * db/file_scans.rs:480) and upserts quant codes per item_data id (db/vector_quants.rs:1109-1111, 1347-1438).
 *
 * pvs_index_remove_rows: removes the rows with the listed ids (any order, duplicates allowed; ids the index does not
 * hold are ignored; *out_removed, optional, counts the rows that went).  The index is COMPACTED on the device: row i
 * afterwards is its i-th surviving row — order kept, ids still strictly increasing, later adds append as before.
 * Group ids and order keys the index holds move with their rows; per-row arrays the CALLER keeps (candidate masks,
 * row weights, pvs_similar_opts arrays) follow the same renumbering.  Every search afterwards sees an ordinary index
 * (nothing on a search path tests a row for being alive).  Safe beside searches (the gate above): calls in flight
 * are completed first, later ones see the compacted index.  After a removal the next pvs_index_add ascends from the
 * LAST SURVIVING id (ids that left may come back: item_data.id is not AUTOINCREMENT in the reference).  A removal that
 * fails after the rows started to move (a HIP error half way) leaves the index unusable: every later call returns
 * PVS_ERR_STATE; destroy and rebuild it.  Multi-device indexes: every shard compacts its rows, the global order is the surviving rows' order.
 *
 * pvs_index_replace_rows[_f32]: overwrites the vectors of rows the index already holds (same ids, same positions,
 * same groups and keys): rows = dense [n][dim] of the index dtype (or f32, converted like pvs_index_add_f32),
 * row_ids strictly increasing, every id must be in the index (PVS_ERR_INVALID_ARG otherwise).  Multi-device
 * indexes stage device-space rows through the host once (every shard picks the rows it holds). */
pvs_status pvs_index_remove_rows(pvs_index *idx, const int64_t *row_ids, uint64_t n, uint64_t *out_removed);
pvs_status pvs_index_replace_rows(pvs_index *idx, const void *rows, uint64_t n, const int64_t *row_ids, pvs_space rows_space);
pvs_status pvs_index_replace_rows_f32(pvs_index *idx, const float *rows, uint64_t n, const int64_t *row_ids, pvs_space rows_space);

/* int8 only: the 4-byte scale artifact.  Same rejections as artifact_scale
 * (db/vector_quants.rs:1456-1460): len != 4, non-finite, <= 0 -> INVALID_ARG. */
pvs_status pvs_index_set_scale_artifact(pvs_index *idx, const uint8_t *artifact, size_t len);
pvs_status pvs_index_set_scale(pvs_index *idx, float scale);

/* Writes the ABI-v2 fields (up to last_candidates); *out is never read, so it need not be initialised. */
pvs_status pvs_index_stats(pvs_index *idx, pvs_stats *out);
/* Writes the first min(out_bytes, sizeof(pvs_stats)) bytes of the current struct (out_bytes >= the v2 size) and sets
 * out->struct_size to that number: pass sizeof(pvs_stats) of the header you were compiled against. */
pvs_status pvs_index_stats_ex(pvs_index *idx, pvs_stats *out, size_t out_bytes);

/* Reads rows [row0, row0+n) back to the host as dense [n][dim] of the index dtype
 * (the stored payload: embeddings.embedding / embedding_quants.quant). */
pvs_status pvs_index_read_rows(pvs_index *idx, uint64_t row0, uint64_t n, void *out_host);

/* Row ids (item_data.id) and, when out_group_ids != NULL, group ids of rows [row0, row0+n), in row
 * order — the key columns a host joins pvs_score_all's `d` column back to SQL with (dist_{cte}:
 * filters/exact.rs:106-134).  Identity groups (no group ids were added) come back as the row ids. */
pvs_status pvs_index_read_ids(pvs_index *idx, uint64_t row0, uint64_t n, int64_t *out_row_ids, int64_t *out_group_ids);

/* Per-kernel timing with HIP events recorded on the stream each kernel is launched
 * on (bench.py's roofline figure).  Off by default. */
typedef struct pvs_profile {
    uint32_t struct_size;
    uint64_t scan_launches;     /* pass B: the full filter scan (dominant kernel) */
    double scan_ms;
    uint64_t scan_rows;         /* rows streamed by those launches */
    uint64_t sample_launches;   /* pass A: threshold scan over the tile sample */
    double sample_ms;
    uint64_t finalize_launches; /* pass C */
    double finalize_ms;
    uint64_t exchange_launches; /* sharded search: the grouped RCCL all-gather + the merge kernel (ABI v2) */
    double exchange_ms;
} pvs_profile;
pvs_status pvs_index_set_profiling(pvs_index *idx, int32_t enable);
pvs_status pvs_index_get_profile(pvs_index *idx, pvs_profile *out, int32_t reset);

/* ------------------------------------------------------------------- search */

/* Page 1 of size k of the reference ordering, for `batch` queries.
 *   queries: [batch][dim] of query_dtype.  f32 queries are accepted by every index
 *     (an int8 index quantizes them with its scale, like compute_query_quant; an
 *     f16/f32 index scores them as f32).  int8 queries (QuantResolved.query_quant)
 *     are accepted by int8 indexes only.  f16 queries are not accepted.
 *   out_ids / out_dist: [batch][k]; out_count[batch] = rows written (min(k, rows)).
 *     Unwritten tail slots are set to id -1 / distance NaN.
 * Distances are the f32 value sqlite-vec would return (bit-exact for int8; for
 * f32/f16 the build reproduces the scalar sequential-f32 evaluation).
 * Routes (same results whichever answers): batch == 1, k <= 256 and at most 8 GB of rows — the reference's request shape — is ONE
 * kernel launch that scores every row exactly and selects the page while the rows stream (csrc/pvs_direct.hip: 0.05-0.06 ms at 10k
 * rows, 0.16 ms at 690k x 768 int8); everything else is the filter scan (sample, threshold, one corpus pass, exact rerank). */
pvs_status pvs_search(pvs_index *idx, const void *queries, pvs_dtype query_dtype, uint32_t batch,
                      uint32_t k, pvs_metric metric, int64_t *out_ids, float *out_dist,
                      uint32_t *out_count);

/* pvs_search over a subset of the rows: the candidate skeleton after the query's other filters (the context
 * CTE every vector filter is joined to, `WHERE begin_cte.item_id IS NOT NULL`,
 * filters/image_embeddings.rs:140-199).  allowed_rows: one byte per stored row in row order (host or
 * device memory), 0 = the row is not a candidate.  Rows outside the mask are never returned, not even as
 * NULL-distance filler; out_count[q] = min(k, allowed rows).
 * Cost: the mask is counted first; when few rows are allowed (allowed x batch <= max(rows / 32, 16384)) the page comes from
 * gather-and-score over the allowed rows only — exact distances in the reference's order, sorted; no corpus pass, NULL rows where
 * the reference puts them — so the cost follows the candidate set as it does in the reference's join; otherwise the filter scan
 * streams the corpus with the masked rows switched off.  pvs_stats.sparse_queries counts the former. */
pvs_status pvs_search_filtered(pvs_index *idx, const void *queries, pvs_dtype query_dtype, uint32_t batch, uint32_t k,
                               pvs_metric metric, const uint8_t *allowed_rows, pvs_space mask_space,
                               int64_t *out_ids, float *out_dist, uint32_t *out_count);
/* The same with the candidate set as a LIST: rows = strictly ascending row positions (the index of a mask byte; add order),
 * host or device memory.  A host that already holds the context CTE's item_data ids sends a few KB instead of one byte per
 * stored row: a 1,000-row candidate set over 10M rows costs ~0.1 ms whatever k is.  A long list is turned into a mask for the
 * filter scan; an unsorted list, duplicates or a position beyond the index is PVS_ERR_INVALID_ARG. */
pvs_status pvs_search_rows(pvs_index *idx, const void *queries, pvs_dtype query_dtype, uint32_t batch, uint32_t k,
                           pvs_metric metric, const uint32_t *rows, uint64_t n_rows_listed, pvs_space rows_space,
                           int64_t *out_ids, float *out_dist, uint32_t *out_count);

/* Pagination (pql/builder.rs:578-582: `LIMIT ? OFFSET ?` behind the final ORDER BY; api/search.rs:51,777-783 executes
 * LIMIT = max(page_size, prefetch_rows <= 4096) at OFFSET (page-1)*page_size): entries [offset, offset+limit) of the same
 * ordering pvs_search / pvs_search_groups return — (distance asc, id asc, NULL last) resp. (value asc, group id asc, NULL
 * last).  out_* hold `limit` entries per query; out_count[q] = entries actually present (0 when the offset is past the end).
 * One pass at k = offset + limit (the filter scan serves k <= 4096, the dense path anything larger). */
pvs_status pvs_search_page(pvs_index *idx, const void *queries, pvs_dtype query_dtype, uint32_t batch, uint64_t offset,
                           uint32_t limit, pvs_metric metric, int64_t *out_ids, float *out_dist, uint32_t *out_count);
pvs_status pvs_search_groups_page(pvs_index *idx, const void *queries, pvs_dtype query_dtype, uint32_t batch, uint64_t offset,
                                  uint32_t limit, pvs_metric metric, pvs_agg agg, const float *row_weights,
                                  int64_t *out_groups, double *out_values, uint32_t *out_count);

/* Request coalescing for the host-buffer entry point (pvs_search).  The reference host answers one query per SQL statement
 * from a pool of up to 16 read connections (db/connection.rs:235,320-357); a corpus pass costs the same for one query as for
 * 32, so callers that arrive within `window_us` of each other are answered by ONE pass: the first one waits out the window
 * (or until `max_batch` queries are waiting; 0 = 32, at most 128), runs one search for every waiting request with its metric
 * and query dtype at the largest k among them, and each caller gets the head of its page — results are exactly those of
 * separate calls.  pvs_search_groups calls without row weights are coalesced the same way, among callers with the same
 * aggregate.  window_us = 0 (the default) switches it off; calls with more than max_batch/2 queries are never held
 * back.  pvs_index_coalescing_stats: pvs_search calls that went through the queue, and corpus passes that served them. */
pvs_status pvs_index_set_coalescing(pvs_index *idx, uint32_t window_us, uint32_t max_batch);
pvs_status pvs_index_coalescing_stats(pvs_index *idx, uint64_t *out_calls, uint64_t *out_passes);

/* pvs_search under apply_sort_bounds (pql/builder.rs:781-815: `WHERE order_rank > gt AND order_rank < lt` with order_rank = the
 * distance, a SQL REAL): page 1 of the rows with gt < d < lt (have_gt / have_lt select the bounds; NULL distances satisfy
 * neither); out_count[q] = min(k, rows inside the bounds).  lt alone: the plain page cut where d reaches lt.  gt (a cursor: the last distance of an earlier
 * page): the rows beyond it are a suffix of the plain ordering — growing pages of the filter scan until k of them are on the
 * page; bounds more than 4,096 rows deep take the dense path (a multi-device index keeps growing the page). */
pvs_status pvs_search_bounded(pvs_index *idx, const void *queries, pvs_dtype query_dtype, uint32_t batch, uint32_t k,
                              pvs_metric metric, int32_t have_gt, double gt, int32_t have_lt, double lt, int64_t *out_ids,
                              float *out_dist, uint32_t *out_count);

/* Same, with every buffer resident in HBM (queries, out_*).  Enqueues on one of
 * the index's streams and returns without synchronising; *out_ticket identifies
 * the stream to wait on with pvs_wait (or pvs_sync for all of them).  At most 16 searches (the size of the
 * reference's read pool) can be in flight on one index: with none free the stream-ordered entry points
 * (this one, pvs_search_sharded_async) return PVS_ERR_STATE — pvs_wait one first — while the synchronous
 * ones block until a slot frees.  On a multi-device index the buffers live on devices[0]. */
pvs_status pvs_search_device(pvs_index *idx, const void *d_queries, pvs_dtype query_dtype,
                             uint32_t batch, uint32_t k, pvs_metric metric, int64_t *d_out_ids,
                             float *d_out_dist, uint32_t *d_out_count, uint32_t *out_ticket);
pvs_status pvs_wait(pvs_index *idx, uint32_t ticket);
pvs_status pvs_sync(pvs_index *idx);

/* 1 (default): every search is queued on one HIP stream — batches run back to back, their
 * scans never compete for CUs; > 1: each in-flight search gets its own stream. */
pvs_status pvs_index_set_streams(pvs_index *idx, uint32_t n_streams);

/* The reference's second sort key.  Its final order is `ORDER BY order_rank ASC NULLS LAST, last_modified DESC`
 * (pql/model.rs:547-553, pql/builder.rs:1188-1223) and, with row_n off (the default, model.rs:232), order_rank IS the
 * distance: rows that tie on it are ordered by last_modified, newest first.  One int64 key per stored row (whatever the host
 * orders by: last_modified as a Unix time, say); afterwards every row page — pvs_search*, pvs_search_device, pvs_search_page,
 * pvs_search_bounded, the dense fallbacks — is ordered (distance asc, NULL last, key DESC, row id asc), and a tie at the k-th
 * distance takes the rows with the largest keys.  The per-item pages follow the same rule: pvs_search_groups[_filtered/_page]
 * order by (value asc, NULL last, key DESC, group id asc) and pvs_rrf_search by (score desc, key DESC, group id asc), a group's
 * key being the key of its first row (the rows of a file share files.last_modified) — for pvs_rrf_search taken from the first
 * branch that carries keys and holds the group.  `n` must equal the index's row count; rows appended later drop the keys
 * (set them again).  keys == NULL removes them.  Across shards: a multi-device index takes the keys in global row order and
 * honours them everywhere (every shard's page record carries the keys of its entries; the merges compare distance, key DESC,
 * id); pvs_search_sharded and pvs_search_groups_sharded do the same when EVERY rank's index carries keys (the page records
 * carry the keys of their entries); pvs_rrf_search_sharded exchanges the candidates' keys (from the lowest branch that carries
 * keys and holds the group, on whichever rank).  The stand-alone merges take the keys of their entries:
 * pvs_merge_group_pages_keyed, pvs_merge_topk_keyed[_device] (the unkeyed forms break ties by id). */
pvs_status pvs_index_set_order_keys(pvs_index *idx, const int64_t *keys, uint64_t n, pvs_space space);

/* Forces the execution path of pvs_search*: 0 = automatic, 1 = dense score + sort
 * (every row scored exactly, full device sort), 2 = filter scan only (error
 * instead of falling back).  For tests and profiling. */
pvs_status pvs_index_set_path(pvs_index *idx, uint32_t path);

/* Which filter-scan kernel serves a pass of `batch` queries on this index (profiling reports): writes a NUL-terminated
 * name such as "k_scan_wide<i8, 768 B, 128 queries>" or "k_scan<f16, 1536 B, 32 queries>", or "dense path" when the shape
 * has no filter-scan instance. */
pvs_status pvs_index_scan_kernel_name(pvs_index *idx, uint32_t batch, char *out, uint32_t out_len);

/* The `d` column of dist_{cte}: one distance per row, in row order. */
pvs_status pvs_score_all(pvs_index *idx, const void *query, pvs_dtype query_dtype, pvs_metric metric,
                         float *out_dist, pvs_space out_space);

/* The same column kept by the library and read in windows (the SQLite scalar drop-ins look rows up one at a time and must not
 * hold 4 bytes per row per statement on the host): one device pass at creation; single-device indexes keep the column in HBM,
 * multi-device ones on the host.  Rows appended to the index afterwards are not part of the column. */
typedef struct pvs_column pvs_column;
pvs_status pvs_score_column_create(pvs_index *idx, const void *query, pvs_dtype query_dtype, pvs_metric metric, pvs_column **out);
pvs_status pvs_score_column_rows(const pvs_column *col, uint64_t *out_rows);
pvs_status pvs_score_column_read(pvs_column *col, uint64_t row0, uint64_t n, float *out_host);
void pvs_score_column_destroy(pvs_column *col);

/* Dense exact distances for a batch: out[row * batch + q] = the reference's
 * vec_distance_*(row payload, query q) — the `d` column of dist_{cte} for `batch`
 * queries at once (query-minor layout).  int8 indexes run on the matrix cores
 * (exact closed form of the integer sums); float indexes score in order. */
pvs_status pvs_score_batch(pvs_index *idx, const void *queries, pvs_dtype query_dtype, uint32_t batch,
                           pvs_metric metric, float *out_dist, pvs_space out_space);

/* Per-item results, on the device: score every row, aggregate per group
 * (GROUP BY file_id: MIN / MAX / AVG, or SUM(d*w)/SUM(w) when row_weights != NULL —
 * filters/exact.rs:67-80), rank the groups (value asc, group id asc, NULL last) and
 * return page 1 of size k.  Groups are the group_ids given to pvs_index_add (identity
 * when none were given).  out_groups/out_values: [batch][k] host buffers; values are the
 * f64 SQLite would produce (Kahan-Babuska-Neumaier sums in row order).
 * "Score every row" is the contract, not always the work: over f16 / f32 rows the page is CERTIFIED (matrix-core brackets of every
 * file's aggregate, the exact in-order chain on the files that can reach the page only; whatever cannot be certified is scored
 * exactly as before) — the same page and values bit for bit; pvs_debug_set("no_float_certify", 1) forces the exact-everywhere route. */
pvs_status pvs_search_groups(pvs_index *idx, const void *queries, pvs_dtype query_dtype, uint32_t batch,
                             uint32_t k, pvs_metric metric, pvs_agg agg, const float *row_weights,
                             int64_t *out_groups, double *out_values, uint32_t *out_count);

/* pvs_search_groups over the rows a candidate mask allows (see pvs_search_filtered): rows outside the mask take no
 * part in any aggregate, and a group without a single candidate row is not part of the result at all. */
pvs_status pvs_search_groups_filtered(pvs_index *idx, const void *queries, pvs_dtype query_dtype, uint32_t batch,
                                      uint32_t k, pvs_metric metric, pvs_agg agg, const float *row_weights,
                                      const uint8_t *allowed_rows, pvs_space mask_space, int64_t *out_groups,
                                      double *out_values, uint32_t *out_count);

/* similar_to (filters/item_similarity.rs:432-581): the target item's stored vectors
 * (rows named by their row ids) against every other row; per group aggregate over the
 * (target vector x group row) fan-out; the target's own rows are excluded
 * (`other.sha256 != target`).  AVG is the reference's default aggregation. */
pvs_status pvs_similar_to(pvs_index *idx, const int64_t *target_row_ids, uint32_t n_targets, uint32_t k,
                          pvs_metric metric, pvs_agg agg, int64_t *out_groups, double *out_values,
                          uint32_t *out_count);

/* similar_to with everything the reference's filter takes (filters/item_similarity.rs:84-142, 432-581):
 *  - the text source's confidence weights: per joined pair
 *      w = pow(coalesce(conf_main,1)*coalesce(conf_other,1), confidence_weight)
 *        * pow(coalesce(lang_other,1)*coalesce(lang_main,1), language_confidence_weight)
 *    (a factor is dropped when its exponent is 0) and the group value is SUM(d*w)/SUM(w); with both
 *    exponents 0 it is the plain `agg`.  row_confidence / row_language_confidence: host arrays, one f64
 *    per stored row in row order, NaN = SQL NULL, NULL pointer = all NULL;
 *  - the CLIP cross-modal gates (:473-489): row_kind[row] = PVS_KIND_CLIP or PVS_KIND_TEXT (data_type
 *    'clip' / 'text-embedding'); with xmodal_i2i == 0 pairs of two clip rows are left out of the
 *    join, with xmodal_t2t == 0 pairs of two text rows.  row_kind == NULL: no gating.
 * pow() is the device math library's (within 1 ulp of the C library SQLite calls): weighted values
 * agree with the reference to ~1e-15 relative, not bit for bit; unweighted ones are bit-exact. */
#define PVS_KIND_CLIP 0
#define PVS_KIND_TEXT 1
typedef struct pvs_similar_opts {
    uint32_t struct_size; /* sizeof(pvs_similar_opts) */
    pvs_agg agg;          /* MIN / MAX / AVG (the reference's default for similar_to is AVG) */
    const double *row_confidence;
    const double *row_language_confidence;
    double confidence_weight;
    double language_confidence_weight;
    const uint8_t *row_kind;
    uint32_t xmodal_i2i; /* default 1 */
    uint32_t xmodal_t2t; /* default 1 */
} pvs_similar_opts;
pvs_status pvs_similar_to_ex(pvs_index *idx, const int64_t *target_row_ids, uint32_t n_targets, uint32_t k,
                             pvs_metric metric, const pvs_similar_opts *opts, int64_t *out_groups,
                             double *out_values, uint32_t *out_count);

/* Per-group aggregate of per-row distances (GROUP BY file_id; MIN/MAX/AVG, or
 * SUM(d*w)/SUM(w) when weights != NULL — `agg` is ignored then, exact.rs:67-80; SUM(w) runs over
 * every row of the group, rows with a NULL distance only drop out of SUM(d*w)).
 * dist / weights / group_ids: [n] host arrays, group_ids non-decreasing.
 * Outputs one (group, f64 aggregate) per distinct group; NaN distance = SQL NULL. */
pvs_status pvs_aggregate(const float *dist, const float *weights, const int64_t *group_ids, uint64_t n,
                         pvs_agg agg, int64_t *out_groups, double *out_values, uint64_t *out_n);

/* ------------------------------------------------------- codec (device side) */
/* max |x_i| over n floats, NaN ignored (blob_absmax).  Runs on the GPU. */
pvs_status pvs_absmax(const float *x, uint64_t n, pvs_space space, int32_t device, float *out_absmax);
/* clamp(round_ties_even(x / scale), -128, 127) as i8, NaN -> 0.  Runs on the GPU. */
pvs_status pvs_quantize_i8(const float *x, uint64_t n, float scale, int8_t *out, pvs_space space,
                           int32_t device);

/* OR-composition of vector filters ranked by reciprocal-rank fusion, entirely on the device
 * (pql/builder.rs:638-661 UNION of the branches' groups; :757-771 per-branch
 * `row_number() OVER (ORDER BY agg <row_n_direction>)` over EVERY group of the branch — NULL aggregates
 * first ascending, last descending; :1284-1301 score = sum_b 1.0/(k_b + coalesce(rank_b, 9223372036854775805)) * weight_b,
 * a branch that does not hold the group contributing its NULL term; ORDER BY score DESC, ties by group id
 * ascending — the reference breaks them by last_modified, which the index does not hold).
 * Each branch: one index (all on the same device), one query, its metric and per-group aggregate (row_weights as in
 * pvs_search_groups).  1..8 branches.  Outputs: host buffers [k]; scores are the f64 SQLite would produce. */
typedef struct pvs_rrf_branch {
    pvs_index *idx;
    const void *query;
    pvs_dtype query_dtype;
    pvs_metric metric;
    pvs_agg agg;
    const float *row_weights; /* NULL, or one weight per row of idx: SUM(d*w)/SUM(w) */
    int32_t row_n_descending; /* row_n_direction: 0 = asc (default, pql/model.rs:233) */
    int32_t rrf_k;            /* Rrf.k, default 1 (pql/model.rs:112-133) */
    double weight;            /* Rrf.weight, default 1.0 */
} pvs_rrf_branch;
pvs_status pvs_rrf_search(const pvs_rrf_branch *branches, uint32_t n_branches, uint32_t k, int64_t *out_groups,
                          double *out_scores, uint32_t *out_count);
/* The steps of pvs_rrf_search's bounded fusion, for hosts that shard a branch BY GROUP over several GPUs or ranks (every row of a
 * group on one shard; configs[4] of BASELINE.json).  One pvs_rrf_cols = one branch scored on one shard: every row's exact
 * distance, the per-group aggregate (f64) and its window key — an order-preserving u64 with the window's NULL placement and
 * direction folded in, comparable across shards; ties are broken by group id everywhere.  Protocol (panoptikon_amd/sharded.py
 * rrf_search_sharded): threshold proposals -> minimum over shards -> pages -> union of candidates -> their keys (lookup) ->
 * groups strictly before each candidate on every shard (count_below) -> sum = exact global rank -> pvs_rrf_fuse. */
typedef struct pvs_rrf_cols pvs_rrf_cols;
pvs_status pvs_rrf_cols_create(const pvs_rrf_branch *branch, pvs_rrf_cols **out);
void pvs_rrf_cols_destroy(pvs_rrf_cols *cols);
pvs_status pvs_rrf_cols_groups(pvs_rrf_cols *cols, uint64_t *out_n_groups);
/* a key at or below which ~1.5 x target_groups of this shard's groups lie (all ones: take everything) */
pvs_status pvs_rrf_cols_threshold(pvs_rrf_cols *cols, uint64_t target_groups, uint64_t *out_key);
/* every group with window key <= key (any order); *out_count > cap means nothing was written */
pvs_status pvs_rrf_cols_page(pvs_rrf_cols *cols, uint64_t key, uint32_t cap, int64_t *out_gids, uint64_t *out_keys, uint32_t *out_count);
pvs_status pvs_rrf_cols_lookup(pvs_rrf_cols *cols, const int64_t *gids, uint32_t m, uint64_t *out_keys, uint8_t *out_present);
/* candidates strictly increasing in (key, group id): out_below[j] = groups of this shard strictly before candidate j */
pvs_status pvs_rrf_cols_count_below(pvs_rrf_cols *cols, const uint64_t *keys, const int64_t *gids, uint32_t m, uint64_t *out_below);

/* The same page for branches sharded BY GROUP over several ranks (every row of a group on one rank; BASELINE configs[4] on 8
 * GPUs): the whole round loop of that protocol behind one call.  Collective: every rank calls it with its shard of each branch
 * (same order, same k, rrf_k, weight, direction) and gets the same page — the reference's, bit for bit.  The exchange runs over
 * RCCL when `comm` is given (ncclAllGather of the padded pages, ncclAllReduce min / sum of thresholds and counts; `world` and
 * `gather` are ignored), otherwise through `gather` — the host's own all-gather of `bytes` bytes per rank into recv[world][bytes],
 * returning 0 on success — for hosts on another transport (and for ranks that share one GPU, which RCCL refuses).  world = 1
 * needs neither.  RRF weights and k must be non-negative (the bound on the groups outside the pages needs it). */
typedef struct pvs_comm pvs_comm; /* a communicator of the multi-GPU section below */
typedef int32_t (*pvs_allgather_fn)(void *ctx, const void *send, void *recv, uint64_t bytes);
pvs_status pvs_rrf_search_sharded(const pvs_rrf_branch *branches, uint32_t n_branches, uint32_t k, pvs_comm *comm, uint32_t world,
                                  pvs_allgather_fn gather, void *gather_ctx, int64_t *out_groups, double *out_scores,
                                  uint32_t *out_count);

/* Which way the last pvs_rrf_search of this thread went: 1 = bounded fusion (pages of each branch's ranking + exact ranks of
 * the candidates; the usual case), 2 = every group of every branch ranked (small inputs, negative weights, massive ties). */
int32_t pvs_rrf_last_path(void);

/* ------------------------------------------------------- codec (host scalars) */
float pvs_scale_from_absmax(float absmax);
void pvs_scale_artifact(float scale, uint8_t out[4]);
/* returns PVS_OK and *scale, or PVS_ERR_INVALID_ARG for an unusable artifact */
pvs_status pvs_artifact_scale(const uint8_t *artifact, size_t len, float *scale);

/* ------------------------------------------------------------- rank and RRF */
/* row_number() OVER (ORDER BY value <row_n_direction>) — pql/builder.rs:757-771.  The reference's window has no
 * NULLS clause, so SQLite's default holds: NaN (NULL) ranks FIRST ascending and LAST descending; ties by
 * id ascending (the build's deterministic tie-break).  pvs_row_number = ascending (the default direction,
 * pql/model.rs:233). */
pvs_status pvs_row_number(const double *values, const int64_t *ids, uint64_t n, int64_t *out_rank);
pvs_status pvs_row_number_dir(const double *values, const int64_t *ids, uint64_t n, int32_t descending, int64_t *out_rank);
/* fused[i] = sum_b weight_b * 1.0 / (k_b + coalesce(rank[b][i], 9223372036854775805));
 * ranks: [n_branches][n], rank < 0 = NULL (branch did not return the row). */
pvs_status pvs_rrf_fuse(const int64_t *ranks, uint32_t n_branches, uint64_t n, const int32_t *ks,
                        const double *weights, double *out_fused);

/* Same-priority order filters WITHOUT rrf (pql/builder.rs:1303-1317): the combined order key is
 *   min(coalesce(rank_b, 9223372036854775805), ...) ascending / max(coalesce(rank_b, -9223372036854775805), ...) descending.
 * pvs_coalesce_ranks: integer ranks [n_filters][n], rank < 0 = NULL.  pvs_coalesce_values: raw f64 aggregates (order_rank
 * without row_n), NaN = NULL; a row no filter returned comes out as the fallback's nearest double. */
pvs_status pvs_coalesce_ranks(const int64_t *ranks, uint32_t n_filters, uint64_t n, int32_t descending, int64_t *out);
pvs_status pvs_coalesce_values(const double *values, uint32_t n_filters, uint64_t n, int32_t descending, double *out);
/* apply_sort_bounds (pql/builder.rs:781-815): keep[i] = order_rank[i] > gt AND order_rank[i] < lt, each bound optional
 * (have_gt / have_lt); NaN = NULL fails any comparison.  The device form for a distance column is pvs_search_bounded. */
pvs_status pvs_sort_bounds(const double *order_rank, uint64_t n, int32_t have_gt, double gt, int32_t have_lt, double lt,
                           uint8_t *keep);

/* -------------------------------------------------- query ingestion / policy */
/* .npy (v1/2/3; f2 f4 f8 i1-8 u1-8 bool; LE/BE; C/Fortran; 1-D or first row of
 * 2-D) -> f32.  Returns the component count in *out_n; when out == NULL only
 * reports the count.  Error strings match embedding_utils.rs. */
pvs_status pvs_npy_to_f32(const uint8_t *npy, size_t len, float *out, size_t out_cap, size_t *out_n);

typedef enum { PVS_INDEX_AUTO = 0, PVS_INDEX_EXACT = 1, PVS_INDEX_QUANT = 2, PVS_INDEX_ANN = 3 } pvs_index_mode;

/* The state resolve_ready_pair (db/vector_quants.rs:1795-1869) would return for
 * (profile, setters), supplied by the caller that owns the database. */
typedef struct pvs_ready_pair {
    int32_t have_db_context;     /* 0: no quant connection available */
    int32_t have_default_profile;/* 0: no default profile configured and no variant named */
    int32_t pair_ready;          /* 0: profile missing / setter not ready / unusable scale */
    int64_t profile_id;
    float scale;
    int64_t dim;
} pvs_ready_pair;

typedef struct pvs_quant_resolved {
    int32_t use_quant;   /* 0 = search exact (Ok(None) in the reference) */
    int64_t profile_id;
    uint64_t query_quant_len; /* 0 when no embedding was given */
} pvs_quant_resolved;

/* resolve_vector_quant (pql/preprocess.rs:314-393).  embedding: f32 LE query bytes
 * (may be NULL); query_quant_out receives dim int8 codes when use_quant and an
 * embedding was given.  Strict selections (index=quant or a non-blank variant)
 * turn every fallback condition into PVS_ERR_NOT_READY / PVS_ERR_DIM_MISMATCH. */
pvs_status pvs_resolve_vector_quant(pvs_index_mode index, const char *variant, int64_t k,
                                    const pvs_ready_pair *pair, const uint8_t *embedding,
                                    size_t embedding_len, int8_t *query_quant_out,
                                    size_t query_quant_cap, pvs_quant_resolved *out);

/* ------------------------------------------------------------- multi-GPU (RCCL) */
#define PVS_UNIQUE_ID_BYTES 128
/* rank 0 creates the id and ships the 128 bytes to the other ranks out of band.  pvs_comm_create is collective and BOUNDED
 * ("comm_timeout_s"): it returns once every rank has joined AND a one-word all-reduce — the communicator's first collective —
 * has come back with the right sum, or PVS_ERR_COMM on this rank when the others do not show up in time (hosts agree on the
 * outcome out of band, as bench.py's make_comm does, before relying on it).  RCCL's diagnostics: NCCL_DEBUG defaults to WARN
 * (set by the library when the host has not chosen a level). */
pvs_status pvs_comm_unique_id(uint8_t id[PVS_UNIQUE_ID_BYTES]);
pvs_status pvs_comm_create(const uint8_t id[PVS_UNIQUE_ID_BYTES], int32_t world, int32_t rank,
                           int32_t device, pvs_comm **out);
void pvs_comm_destroy(pvs_comm *comm);
/* max over the ranks of one float per rank (ncclAllReduce): the absmax of a sharded space from the absmax of every shard —
 * compute_int8_scale_artifact (db/vector_quants.rs:1513-1554) over rows that no single rank holds.  Collective: every rank calls. */
pvs_status pvs_comm_allreduce_max_f32(pvs_comm *comm, float *inout);
/* Row-sharded search: idx holds this rank's shard (its row ids are global).
 * Each rank scores its shard, one ncclAllGather moves the per-shard top-k
 * (batch*k*(f32,i64)) over xGMI, every rank merges.  Outputs are device buffers
 * [batch][k] valid on every rank. */
pvs_status pvs_search_sharded(pvs_index *idx, pvs_comm *comm, const void *d_queries,
                              pvs_dtype query_dtype, uint32_t batch, uint32_t k, pvs_metric metric,
                              int64_t *d_out_ids, float *d_out_dist, uint32_t *d_out_count);
/* Stream-ordered form: local search, all-gather and merge are enqueued on one of the index's
 * streams with no host synchronisation in between; pvs_wait(idx, ticket) completes it (and,
 * when any shard handed a query to the dense path, redoes the exchange for the batch).
 * Failure semantics across ranks (round 6): a rank that fails LOCALLY before its exchange (out of memory, a HIP error of its
 * scan) still takes part in the all-gather with a failure record, gets a ticket, and every rank's pvs_wait fails that search —
 * the failing rank with its own error, the others with PVS_ERR_COMM naming it; nobody is left inside the collective.  A rank
 * that never arrives at all is bounded by "comm_timeout_s" (pvs_debug_set): PVS_ERR_COMM, communicator aborted. */
pvs_status pvs_search_sharded_async(pvs_index *idx, pvs_comm *comm, const void *d_queries,
                                    pvs_dtype query_dtype, uint32_t batch, uint32_t k, pvs_metric metric,
                                    int64_t *d_out_ids, float *d_out_dist, uint32_t *d_out_count,
                                    uint32_t *out_ticket);
/* k-way merge of `world` per-shard pages (host buffers, [world][batch][k] with
 * counts [world][batch]) under the shared ordering; the same routine the device
 * merge kernel implements, exposed for hosts that gather by other means. */
pvs_status pvs_merge_topk(const int64_t *ids, const float *dist, const uint32_t *counts,
                          uint32_t world, uint32_t batch, uint32_t k, int64_t *out_ids,
                          float *out_dist, uint32_t *out_count);
/* The same with the second sort key of pvs_index_set_order_keys: keys [world][batch][k] = the key of every page entry's row
 * (NULL: none) -> (distance asc, NULL last, key DESC, id asc), the order of every other route of a keyed index. */
pvs_status pvs_merge_topk_keyed(const int64_t *ids, const float *dist, const int64_t *keys, const uint32_t *counts,
                                uint32_t world, uint32_t batch, uint32_t k, int64_t *out_ids,
                                float *out_dist, uint32_t *out_count);

/* Per-item search over row shards: shard BY GROUP — every row of a file/item on one rank —
 * so MIN/MAX/AVG and the weighted average stay shard-local; each rank runs pvs_search_groups on its
 * shard, the pages [batch][k] of (group id, f64 value) are exchanged with one grouped all-gather over
 * the communicator and merged on every rank under the group ordering (value asc, group id asc, NULL
 * last).  Host buffers in and out, like pvs_search_groups; row_weights indexes this rank's rows. */
pvs_status pvs_search_groups_sharded(pvs_index *idx, pvs_comm *comm, const void *queries, pvs_dtype query_dtype,
                                     uint32_t batch, uint32_t k, pvs_metric metric, pvs_agg agg,
                                     const float *row_weights, int64_t *out_groups, double *out_values,
                                     uint32_t *out_count);
/* The merge step alone, for hosts that gather by other means: pages [world][batch][k], counts
 * [world][batch] (host buffers).  Groups must not repeat across shards. */
pvs_status pvs_merge_group_pages(const int64_t *groups, const double *values, const uint32_t *counts,
                                 uint32_t world, uint32_t batch, uint32_t k, int64_t *out_groups,
                                 double *out_values, uint32_t *out_count);
/* The same with the second sort key of pvs_index_set_order_keys: keys [world][batch][k] = the key of every page entry's group
 * (NULL: none) -> (value asc, NULL last, key DESC, group id asc).  What pvs_search_groups_sharded runs when every rank's
 * index carries keys. */
pvs_status pvs_merge_group_pages_keyed(const int64_t *groups, const double *values, const int64_t *keys, const uint32_t *counts,
                                       uint32_t world, uint32_t batch, uint32_t k, int64_t *out_groups,
                                       double *out_values, uint32_t *out_count);

/* The device form of pvs_merge_topk (the kernel pvs_search_sharded runs after the all-gather):
 * every pointer is an HBM buffer on `device`; synchronous. */
pvs_status pvs_merge_topk_device(int32_t device, const int64_t *d_ids, const float *d_dist,
                                 const uint32_t *d_counts, uint32_t world, uint32_t batch, uint32_t k,
                                 int64_t *d_out_ids, float *d_out_dist, uint32_t *d_out_count);
/* ... with order keys d_keys [world][batch][k] (NULL: none), like pvs_merge_topk_keyed. */
pvs_status pvs_merge_topk_keyed_device(int32_t device, const int64_t *d_ids, const float *d_dist, const int64_t *d_keys,
                                       const uint32_t *d_counts, uint32_t world, uint32_t batch, uint32_t k,
                                       int64_t *d_out_ids, float *d_out_dist, uint32_t *d_out_count);

/* ------------------------------------------------- device memory + synthetic */
/* Free and total HBM of `device` in bytes (hipMemGetInfo): capacity planning for shard sizes. */
pvs_status pvs_device_mem_info(int32_t device, uint64_t *free_bytes, uint64_t *total_bytes);
/* hipDeviceSynchronize on `device` (-1: current): every stream of the process, the library's included. */
pvs_status pvs_device_synchronize(int32_t device);
pvs_status pvs_device_malloc(int32_t device, size_t bytes, void **out);
pvs_status pvs_device_free(int32_t device, void *ptr);
pvs_status pvs_memcpy(void *dst, const void *src, size_t bytes, int32_t device);
/* Unit-normalised pseudo-Gaussian rows, generated in HBM:
 * row r, component c is a pure function of (seed, row0 + r, c). */
pvs_status pvs_synth_rows_f32(int32_t device, uint64_t seed, uint64_t row0, uint64_t n, uint32_t dim,
                              float *d_out);
/* ABI v5 — the same contract over a distribution shaped like production embeddings (the reference measures on CLIP / mpnet
 * vectors, docs/vector-int8-quant.md:220-224, docs/vector-quant-measurements.md:102-117): 2,000 clusters with power-law sizes,
 * anisotropic noise inside a cluster, one block of 8 consecutive rows in ten a run of near-duplicates, one row in a hundred an
 * exact duplicate of a row shortly before it; unit-normalised.  Queries for such a corpus: the same seed at rows beyond the
 * corpus (they fall into the same clusters). */
pvs_status pvs_synth_rows_clustered_f32(int32_t device, uint64_t seed, uint64_t row0, uint64_t n, uint32_t dim,
                                        float *d_out);

/* On-box peaks for the roofline report (SURVEY.md 8d asks for datasheet AND measured peaks): streaming HBM reads with plain
 * 16-byte loads and with the LDS-DMA transport the scan uses, a device-to-device copy (read + write bytes), and the dense
 * int8 / f16 matrix-core rate at the clock the chip sustains.  Diagnostic only; allocates 8 GiB for a second or two. */
typedef struct pvs_microbench_result {
    uint32_t struct_size;
    uint32_t compute_units, clock_mhz;
    double hbm_read_gbs, hbm_lds_dma_gbs, hbm_copy_gbs;
    double mfma_i8_tops, mfma_f16_tflops;
} pvs_microbench_result;
pvs_status pvs_microbench(int32_t device, pvs_microbench_result *out);

/* ------------------------------------------------------- test and tuning hooks */
/* Not part of the product contract: the test suite and the measurement scripts use these to force a code path (and so compare
 * two product paths on the same input) or to sweep a tuning parameter.  Every knob is 0 in a process that never calls
 * pvs_debug_set, and the library reads NO environment variable to decide which algorithm answers (the only variable it looks at
 * is PVS_RCCL_PATH, a loader path for librccl).  Process-wide; set a knob while no search is in flight.
 *   "sample_div" N            pass A samples 1/N of the corpus            "sample_j_div" N     threshold = (k/N)-th sample value
 *   "no_light_finalize"       multi-stream: keep the LDS-heavy pass C      "force_light_finalize"  every int8 search: LDS-light pass C
 *   "dense_per_query"         dense fallback one query per pass            "no_direct_score"    1..4 int8 queries on the matrix cores
 *   "no_page_rank"            per-item search sorts every group            "scan_no_wide128"    128-query int8 passes on k_scan
 *   "rrf_serial" / "rrf_full" / "rrf_trace"   pvs_rrf_search: branches on one thread / full ranking only / phase times on stderr
 *   "rrf_digest"              pvs_rrf_search records stage digests (below)
 *   "scratch_idle_cap_mb" N   idle scratch kept per device (default 16 GiB)  "scratch_bypass"   scratch straight from hipMalloc/hipFree
 *   "no_sparse" / "sparse_max" N   filtered searches: never / up to N allowed rows on the gather-score path
 *   "no_fused_agg"            per-item MAX/AVG/weighted through the dense matrix + k_group_aggregate
 *   "no_side_finalize"        pvs_search_device: pass C on the search's own stream instead of the index's side stream
 *   "rrf_host_rounds" / "multi_host_pages" / "prelude_stream"   round-3 host forms of the RRF rounds / the multi-device per-item merge; the prelude-stream experiment
 *   "dense_nq4"               k_dense_exact: at most 4 float queries per pass (not the packed 8-query instance)
 *   "marker_events"           profiling spans as hipEventRecord markers around the kernels instead of events bound to the dispatches
 *   "no_direct_topk"          a single query always takes the filter scan, never the one-launch exact search (pvs_direct.hip)
 *   "direct_max_mb" N         ... takes the one-launch search up to N MB of rows (default 8192)
 *   "direct_queries"          (read-only counter) single queries answered by the one-launch search, process-wide
 *   "no_float_certify"        per-item pages over f16 / f32 rows: every (row, query) distance exact (k_exact_wide) instead of the matrix-core
 *                             brackets + exact rescan of the candidate files (csrc/pvs_items_float.hip; same pages either way)
 *   "float_certify_queries" / "float_certify_rows"   (read-only counters) queries that route answered; candidate rows it rescanned
 *   "comm_timeout_s" N        bound on every wait for the other ranks: communicator creation (ncclCommInitRank + a one-word
 *                             all-reduce, the communicator's first collective), the shard exchange behind pvs_wait, the control
 *                             messages of the sharded fusion (default 180 s).  On expiry the communicator is aborted and the call
 *                             returns PVS_ERR_COMM naming the rank: create a new communicator
 *   "comm_fail_local" N       tests: the next N pvs_search_sharded_async calls of this process fail before their exchange */
pvs_status pvs_debug_set(const char *key, int64_t value);
pvs_status pvs_debug_get(const char *key, int64_t *out_value);
/* Stage digests of the process' last single-device pvs_rrf_search run under "rrf_digest": out[branch * 4 + stage], stage 0 = the
 * `d` column, 1 = per-group aggregates, 2 = window keys, 3 = ranks (bounded fusion: the candidates' counted ranks; full ranking:
 * every group's rank).  out: 8 * 4 words.  *out_path: 1 = bounded fusion, 2 = full ranking.  tools/rrf_stage_digest.py. */
pvs_status pvs_debug_rrf_digests(uint64_t *out, uint32_t *out_branches, int32_t *out_path);

#ifdef __cplusplus
}
#endif
#endif /* PVS_H */
