/*
 * pvs_sqlite.h — the SQLite side of the drop-in boundary: libpvs_sqlite.so registers, through SQLite's own extension
 * ABI, the functions that feed the reference's `dist_{cte}` from a device index.
 *
 * Replaces `sqlite3_auto_extension(sqlite3_vec_init)` (db/sql_functions.rs:105-128) for the vector path: the host calls
 * `sqlite3_auto_extension(sqlite3_pvs_init)` (or loads libpvs_sqlite.so per connection) and binds its device indexes by
 * name; the filter compilers then emit
 *     FROM pvs_dist(<index>, ?, 'cosine') AS p JOIN item_data ON item_data.id = p.id ...            -- whole `d` column
 *     pvs_distance_cosine(<index>, embeddings.id, ?) AS d                                           -- per-row drop-in
 * where they emit `vec_distance_cosine(embeddings.embedding, ?) AS d` today (filters/image_embeddings.rs:321-362,
 * text_embeddings.rs:386-418, exact.rs:106-165).  Everything else in the generated SQL is untouched.
 *
 * SQL surface
 *   pvs_dist(index TEXT, query BLOB [, metric TEXT = 'cosine' [, k INTEGER]]) -> rows (id INTEGER, d REAL)
 *       without k: one row per stored vector, `d` = the f32 distance sqlite-vec would return, widened; NULL where it
 *       yields NaN.  With k: page 1 of size k of the (distance, id) ordering (the filter scan).
 *   pvs_distance_cosine(index TEXT, id INTEGER, query BLOB) -> REAL     the same column, looked up per row: the first call
 *   pvs_distance_l2(index TEXT, id INTEGER, query BLOB) -> REAL         of a statement runs the device pass; an id the
 *                                                                       index does not hold gives NULL.
 *   query: dim*4 bytes f32 little-endian, or dim int8 codes for an int8 index (QuantResolved.query_quant).
 *   pvs_load(index TEXT, sql TEXT [, param ...]) -> INTEGER            index lifecycle: runs `sql` on this
 *       connection with the parameters bound in order and appends every row it yields — (row id INTEGER, group id INTEGER
 *       or NULL, payload BLOB), in the order the index shall hold them, i.e. the loaders' ORDER BY item_data.id
 *       (db/vector_quants.rs:1085-1099) — to the bound index, in chunks, without the rows ever becoming SQL values of the
 *       caller.  payload = dim*4 bytes (f32 LE: embeddings.embedding, converted on the device to the index dtype) or dim
 *       bytes of int8 codes for an int8 index (embedding_quants.quant); NULLs and blobs of any other length are skipped —
 *       the `length(embedding) = dim*4` guard of the reference's backfill.  Returns the number of rows appended.
 *   pvs_load_info(index TEXT) -> TEXT    JSON of that index's last pvs_load: rows, skipped, last_id, sum_id, sum_group
 *       (sums of the 32-bit residues of the ids: what a host keeps to recognise its loaded prefix after an epoch bump).
 */
#ifndef PVS_SQLITE_H
#define PVS_SQLITE_H

#include <stdint.h>

#include "pvs.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sqlite3 sqlite3;
typedef struct sqlite3_context sqlite3_context;
typedef struct sqlite3_value sqlite3_value;
struct sqlite3_module;

/* The SQLite entry points the extension needs, as pointers.  A host that links SQLite statically (sqlx / libsqlite3-sys
 * "bundled") fills this from its own copy and calls pvs_sqlite_register; a process that carries libsqlite3.so (Python,
 * the sqlite3 shell) needs nothing: the loadable-extension entry points find the functions themselves. */
typedef struct pvs_sqlite_api {
    uint32_t struct_size;
    int (*create_function_v2)(sqlite3 *, const char *, int, int, void *, void (*)(sqlite3_context *, int, sqlite3_value **),
                              void (*)(sqlite3_context *, int, sqlite3_value **), void (*)(sqlite3_context *), void (*)(void *));
    int (*create_module_v2)(sqlite3 *, const char *, const struct sqlite3_module *, void *, void (*)(void *));
    int (*declare_vtab)(sqlite3 *, const char *);
    int (*value_type)(sqlite3_value *);
    int (*value_bytes)(sqlite3_value *);
    const void *(*value_blob)(sqlite3_value *);
    const unsigned char *(*value_text)(sqlite3_value *);
    long long (*value_int64)(sqlite3_value *);
    void (*result_double)(sqlite3_context *, double);
    void (*result_int64)(sqlite3_context *, long long);
    void (*result_null)(sqlite3_context *);
    void (*result_error)(sqlite3_context *, const char *, int);
    void *(*user_data)(sqlite3_context *);
    void *(*get_auxdata)(sqlite3_context *, int);
    void (*set_auxdata)(sqlite3_context *, int, void *, void (*)(void *));
    char *(*mprintf)(const char *, ...);
    void (*free)(void *);
    /* since ABI v2 of this struct — the statement interface pvs_load streams rows through.  A host that passes the shorter
     * v1 struct (struct_size up to `free`) gets everything except pvs_load. */
    int (*prepare_v2)(sqlite3 *, const char *, int, void **, const char **);
    int (*step)(void *);
    int (*finalize)(void *);
    int (*column_type)(void *, int);
    const void *(*column_blob)(void *, int);
    int (*column_bytes)(void *, int);
    long long (*column_int64)(void *, int);
    int (*bind_value)(void *, int, const sqlite3_value *);
    sqlite3 *(*context_db_handle)(sqlite3_context *);
    const char *(*errmsg)(sqlite3 *);
    void (*result_text)(sqlite3_context *, const char *, int, void (*)(void *));
    /* since ABI v3 — what pvs_backfill writes codes with.  A host that passes a shorter struct gets everything except
     * pvs_backfill and pvs_ready_pair. */
    int (*bind_blob)(void *, int, const void *, int, void (*)(void *));
    int (*bind_int64)(void *, int, long long);
    int (*reset)(void *);
} pvs_sqlite_api;

/* SQL function pvs_ready_pair(profile_name, setter_name, ...) -> NULL | '{"profile_id": p, "scale": s, "dim": d}': the reference's
 * resolve_ready_pair (db/vector_quants.rs:1795-1869) for hosts that are not its Rust — the profile must be active, every
 * setter that exists must have a `ready` (profile, setter) pair with a usable scale artifact and a dimension, and all pairs must
 * share one (scale, dim); unknown setter names are skipped.  No GPU involved. */

/* Registers pvs_dist / pvs_distance_cosine / pvs_distance_l2 / pvs_load / pvs_backfill / pvs_ready_pair on one connection.  api == NULL: use
 * the table a previous call (or a loadable-extension entry point) installed, or — explicit opt-in of a process that knows it
 * carries exactly one SQLite, exported — resolve the entry points by name from the process image.  Returns an SQLite
 * result code (0 = SQLITE_OK). */
int32_t pvs_sqlite_register(void *db, const pvs_sqlite_api *api);

/* Diagnostics: copies the table of SQLite entry points currently installed (by an extension entry point, by
 * pvs_sqlite_register, or by the by-name lookup) into *out, at most out->struct_size bytes; returns 0, or 1 when none is
 * installed yet.  A host can compare it with its own functions' addresses to see which SQLite the extension talks to. */
int32_t pvs_sqlite_api_snapshot(pvs_sqlite_api *out);

/* Loadable-extension entry points (int xEntryPoint(sqlite3*, char **pzErrMsg, const sqlite3_api_routines*)): the symbol
 * SQLite derives from the file name, the generic one, and the one to hand to sqlite3_auto_extension — exactly where the
 * reference hands sqlite3_vec_init (db/sql_functions.rs:105-128).  They take every SQLite entry point from the
 * sqlite3_api_routines table SQLite passes in (the calling SQLite's OWN functions, whether it is a shared library or linked
 * statically into the host with hidden symbols), never from a name lookup; SQLite >= 3.8.0. */
int sqlite3_pvs_init(void *db, char **pzErrMsg, const void *pApi);
int sqlite3_extension_init(void *db, char **pzErrMsg, const void *pApi);
int sqlite3_pvssqlite_init(void *db, char **pzErrMsg, const void *pApi);

/* The row streamer behind pvs_load, for a host that holds the connection itself: prepares `sql` on `db` (no parameters),
 * streams its (row id, group id, payload) rows into `idx` in chunks of `chunk_rows` (0 = 65536).  `out` may be NULL.
 * Returns pvs_status; PVS_ERR_STATE when the registered SQLite entry points lack the statement interface. */
typedef struct pvs_sqlite_load_result {
    uint64_t rows;      /* appended */
    uint64_t skipped;   /* NULL payloads and blobs of another length */
    int64_t last_id;    /* row id of the last appended row (-1: none) */
    uint64_t sum_id;    /* sum over appended rows of (row id & 0xffffffff) */
    uint64_t sum_group; /* sum over appended rows of (group id & 0xffffffff) */
} pvs_sqlite_load_result;
int32_t pvs_sqlite_load(void *db, const char *sql, pvs_index *idx, uint32_t chunk_rows, pvs_sqlite_load_result *out);

/* Write side of the lifecycle (the reference's backfill_chunk, db/vector_quants.rs:1119-1163): runs
 * `select_sql` on `db` to completion — rows (id INTEGER, embedding BLOB f32 LE, artifact BLOB, artifact_rev INTEGER), the shape
 * of BACKFILL_CHUNK_SQL (:1085-1099), with the caller's own LIMIT — then quantizes all embeddings in one device pass
 * (quantize_int8 with the scale the artifact holds, :1489-1503) and runs `upsert_sql` once per row with
 * (?1 id, ?2 profile_id, ?3 artifact_rev, ?4 codes BLOB) — the shape of INSERT_QUANT_SQL (:1101-1117).  Like the reference,
 * the select is read completely before the first write (it reads embedding_quants itself), nothing is written when a row's
 * artifact is not a valid scale, and a blob whose length is not a multiple of 4 or differs from the first row's is an error
 * (the SQL guards `length(e.embedding) = c.dim * 4`).  `device`: HIP ordinal, -1 = current.  Also reachable from SQL:
 *     SELECT pvs_backfill(select_sql, upsert_sql, profile_id, device [, select parameter ...])   -> rows written
 *     SELECT pvs_backfill_cursor()                 -> largest id written by this connection's last pvs_backfill (NULL: none)
 *     SELECT pvs_backfill_phases()                 -> 'select_ms,quantize_ms,upsert_ms' of that call (where the time went)
 * Returns pvs_status; PVS_ERR_STATE when the registered SQLite entry points lack bind_blob / bind_int64 / reset. */
typedef struct pvs_sqlite_backfill_result {
    uint64_t written; /* rows upserted */
    int64_t cursor;   /* max(id) over them, or the value passed in `after_id` semantics of the caller: -1 when none */
} pvs_sqlite_backfill_result;
int32_t pvs_sqlite_backfill(void *db, const char *select_sql, const char *upsert_sql, int64_t profile_id, int32_t device,
                            pvs_sqlite_backfill_result *out);

/* Names the SQL functions resolve: bind every device index the host wants reachable from SQL (process-wide registry;
 * rebinding a name replaces it; unbind before pvs_index_destroy).  Return pvs_status. */
int32_t pvs_sqlite_bind_index(const char *name, pvs_index *idx);
int32_t pvs_sqlite_unbind_index(const char *name);

#ifdef __cplusplus
}
#endif
#endif /* PVS_SQLITE_H */
