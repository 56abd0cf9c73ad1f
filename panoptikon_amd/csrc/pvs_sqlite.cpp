// pvs_sqlite.cpp — the SQLite side of the drop-in boundary: a loadable extension that feeds the
// reference's `dist_{cte}` from the device index.
//
// The reference's only plugin seam on this path is SQLite's extension ABI: `sqlite3_auto_extension(sqlite3_vec_init)`
// registers sqlite-vec's scalar functions on every connection (db/sql_functions.rs:83-128) and the filter compilers
// emit `vec_distance_cosine(payload, ?) AS d` into the MATERIALIZED dist CTE (pql/builder/filters/exact.rs:106-165,
// image_embeddings.rs:321-362).  This file registers, through the same ABI,
//   pvs_dist(index, query [, metric [, k]])        table-valued function (eponymous virtual table) yielding (id, d):
//                                                  one row per stored vector — the whole `d` column of dist_{cte},
//                                                  computed by ONE device pass (pvs_score_all) — or, when k is given,
//                                                  page 1 of size k from the filter scan (pvs_search);
//   pvs_distance_cosine(index, id, query)          scalar drop-ins for `vec_distance_cosine(embeddings.embedding, ?)` /
//   pvs_distance_l2(index, id, query)              `vec_distance_L2(...)`: same place in the SQL, the payload column
//                                                  replaced by its key (`embeddings.id`).  The first call of a statement
//                                                  runs the device pass for the bound query and parks the column in the
//                                                  statement's auxiliary data; every further row is a binary search.
// so that the rest of the generated SQL (joins, GROUP BY file_id, row_number(), RRF, ORDER BY ... LIMIT) runs unchanged.
// A per-row UDF over two blobs (sqlite-vec's own signature) cannot be batched onto a device — SQLite hands it one row
// at a time — which is why the payload argument becomes the row key.
//
// `index` is a name bound to a pvs_index* with pvs_sqlite_bind_index (a host process registers its indexes once).
// `query` is a BLOB: dim*4 bytes = f32 little-endian (QuantResolved._embedding, db/pql.rs:76-85), or dim bytes = int8
// codes (query_quant) for an int8 index.  Distances are the f32 value widened to a REAL, NULL where sqlite-vec yields NaN.
//
// A Rust host links its own copy of SQLite (libsqlite3-sys "bundled", symbols not exported), so nothing here includes
// sqlite3.h or links libsqlite3: the ~30 entry points used are declared below with their documented prototypes and reach
// the library through a table of function pointers.  The loadable-extension entry points fill it from the
// sqlite3_api_routines table SQLite hands them — the calling SQLite's own functions, by their slot in that append-only
// struct (sqlite3ext.h; slots below) — which is what makes `sqlite3_auto_extension(sqlite3_pvs_init)` work in a host with a
// statically linked SQLite, and what rules out binding to a different copy of SQLite that merely happens to be loaded.
// pvs_sqlite_register(db, &table) serves a host that fills the table itself; the by-name lookup (dlsym) is left only behind
// the explicit pvs_sqlite_register(db, NULL) / pvs_sqlite_load of a process that never went through an entry point.
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <ctime>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "pvs.h"

#define PVS_EXPORT extern "C" __attribute__((visibility("default")))

// ---- the slice of the SQLite C API this file uses (sqlite.org/c3ref; layouts are part of SQLite's stable ABI)
extern "C" {
typedef struct sqlite3 sqlite3;
typedef struct sqlite3_context sqlite3_context;
typedef struct sqlite3_value sqlite3_value;
typedef long long sqlite3_int64;
typedef unsigned long long sqlite3_uint64;
struct sqlite3_module;
struct sqlite3_vtab {
    const sqlite3_module *pModule;
    int nRef;
    char *zErrMsg;
};
struct sqlite3_vtab_cursor {
    sqlite3_vtab *pVtab;
};
struct sqlite3_index_info {
    int nConstraint;
    struct sqlite3_index_constraint {
        int iColumn;
        unsigned char op;
        unsigned char usable;
        int iTermOffset;
    } *aConstraint;
    int nOrderBy;
    struct sqlite3_index_orderby {
        int iColumn;
        unsigned char desc;
    } *aOrderBy;
    struct sqlite3_index_constraint_usage {
        int argvIndex;
        unsigned char omit;
    } *aConstraintUsage;
    int idxNum;
    char *idxStr;
    int needToFreeIdxStr;
    int orderByConsumed;
    double estimatedCost;
    sqlite3_int64 estimatedRows;
    int idxFlags;
    sqlite3_uint64 colUsed;
};
struct sqlite3_module {
    int iVersion;
    int (*xCreate)(sqlite3 *, void *, int, const char *const *, sqlite3_vtab **, char **);
    int (*xConnect)(sqlite3 *, void *, int, const char *const *, sqlite3_vtab **, char **);
    int (*xBestIndex)(sqlite3_vtab *, sqlite3_index_info *);
    int (*xDisconnect)(sqlite3_vtab *);
    int (*xDestroy)(sqlite3_vtab *);
    int (*xOpen)(sqlite3_vtab *, sqlite3_vtab_cursor **);
    int (*xClose)(sqlite3_vtab_cursor *);
    int (*xFilter)(sqlite3_vtab_cursor *, int, const char *, int, sqlite3_value **);
    int (*xNext)(sqlite3_vtab_cursor *);
    int (*xEof)(sqlite3_vtab_cursor *);
    int (*xColumn)(sqlite3_vtab_cursor *, sqlite3_context *, int);
    int (*xRowid)(sqlite3_vtab_cursor *, sqlite3_int64 *);
    int (*xUpdate)(sqlite3_vtab *, int, sqlite3_value **, sqlite3_int64 *);
    int (*xBegin)(sqlite3_vtab *);
    int (*xSync)(sqlite3_vtab *);
    int (*xCommit)(sqlite3_vtab *);
    int (*xRollback)(sqlite3_vtab *);
    int (*xFindFunction)(sqlite3_vtab *, int, const char *, void (**)(sqlite3_context *, int, sqlite3_value **), void **);
    int (*xRename)(sqlite3_vtab *, const char *);
    int (*xSavepoint)(sqlite3_vtab *, int);
    int (*xRelease)(sqlite3_vtab *, int);
    int (*xRollbackTo)(sqlite3_vtab *, int);
    int (*xShadowName)(const char *);
};
}
enum { SQLITE_OK = 0, SQLITE_ERROR = 1, SQLITE_NOMEM = 7, SQLITE_CONSTRAINT = 19, SQLITE_ROW = 100, SQLITE_DONE = 101 };
enum { SQLITE_INTEGER = 1, SQLITE_FLOAT = 2, SQLITE_TEXT = 3, SQLITE_BLOB = 4, SQLITE_NULL = 5 };
enum { SQLITE_UTF8 = 1, SQLITE_DETERMINISTIC = 0x800, SQLITE_INDEX_CONSTRAINT_EQ = 2 };

// every SQLite entry point the extension calls, as pointers (pvs_sqlite.h documents the struct for hosts that fill it)
#include "pvs_sqlite.h"

namespace {
pvs_sqlite_api g_api;
bool g_api_set = false;
std::mutex g_mu;

struct Bound {
    pvs_index *ix = nullptr;
    pvs_sqlite_load_result last_load = {0, 0, -1, 0, 0};
    std::shared_ptr<const std::vector<int64_t>> ids;  // host copy of the row ids (strictly increasing), refreshed when the row
    uint64_t ids_rows = UINT64_MAX;                   // count moves; statements share it (8 bytes per row per INDEX, not per statement)
};
std::map<std::string, Bound> g_indexes;

pvs_status lookup(const std::string &name, pvs_index **ix, uint32_t *dtype, uint32_t *dim, uint64_t *rows) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_indexes.find(name);
    if (it == g_indexes.end()) return PVS_ERR_INVALID_ARG;
    pvs_stats st;
    st.struct_size = sizeof st;
    pvs_status s = pvs_index_stats(it->second.ix, &st);
    if (s != PVS_OK) return s;
    *ix = it->second.ix;
    *dtype = st.dtype;
    *dim = st.dim;
    *rows = st.rows;
    return PVS_OK;
}
// the index's row ids (cached per binding, shared by the statements that use them)
pvs_status row_ids(const std::string &name, uint64_t rows, std::shared_ptr<const std::vector<int64_t>> *out) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_indexes.find(name);
    if (it == g_indexes.end()) return PVS_ERR_INVALID_ARG;
    Bound &b = it->second;
    if (b.ids_rows != rows || !b.ids) {
        auto v = std::make_shared<std::vector<int64_t>>(rows);
        if (rows) {
            pvs_status s = pvs_index_read_ids(b.ix, 0, rows, v->data(), nullptr);
            if (s != PVS_OK) return s;
        }
        b.ids = v;
        b.ids_rows = rows;
    }
    *out = b.ids;
    return PVS_OK;
}

int parse_metric(sqlite3_value *v, pvs_metric *m) {
    *m = PVS_COSINE;
    if (!v || g_api.value_type(v) == SQLITE_NULL) return SQLITE_OK;
    const unsigned char *t = g_api.value_text(v);
    if (!t) return SQLITE_ERROR;
    std::string s((const char *)t);
    for (auto &c : s) c = (char)tolower(c);
    if (s == "cosine") return SQLITE_OK;
    if (s == "l2") {
        *m = PVS_L2;
        return SQLITE_OK;
    }
    return SQLITE_ERROR;
}

// query BLOB -> (pointer, dtype) for the index; dimension / element-type mismatch is the reference's SQL error
// (sqlite-vec raises on mismatched lengths -> db/pql.rs:18-21)
const char *query_of(sqlite3_value *v, uint32_t index_dtype, uint32_t dim, const void **q, pvs_dtype *qd) {
    if (g_api.value_type(v) != SQLITE_BLOB) return "query must be a BLOB (f32 little-endian, or int8 codes for an int8 index)";
    const int n = g_api.value_bytes(v);
    *q = g_api.value_blob(v);
    if ((uint64_t)n == (uint64_t)dim * 4) {
        *qd = PVS_F32;
        return nullptr;
    }
    if ((uint64_t)n == dim && index_dtype == PVS_I8) {
        *qd = PVS_I8;
        return nullptr;
    }
    return "query length does not match the index dimension";
}

// ------------------------------------------------------------------ pvs_dist(index, query, metric, k)
struct DistVtab {
    sqlite3_vtab base;
};
struct DistCursor {
    sqlite3_vtab_cursor base;
    uint64_t pos = 0, n = 0;
    // page mode (k given): the page itself
    std::vector<int64_t> ids;
    std::vector<float> d;
    // column mode: the ids shared with the binding, the `d` column kept by libpvs and read through a window
    std::shared_ptr<const std::vector<int64_t>> all_ids;
    pvs_column *col = nullptr;
    static constexpr uint64_t WIN = 1u << 18;
    uint64_t win0 = UINT64_MAX;
    std::vector<float> win;
    void drop_column() {
        pvs_score_column_destroy(col);
        col = nullptr;
        all_ids.reset();
        win0 = UINT64_MAX;
    }
    ~DistCursor() { drop_column(); }
};
enum { COL_ID = 0, COL_D = 1, COL_INDEX = 2, COL_QUERY = 3, COL_METRIC = 4, COL_K = 5 };

int dist_connect(sqlite3 *db, void *, int, const char *const *, sqlite3_vtab **out, char **) {
    int rc = g_api.declare_vtab(db, "CREATE TABLE x(id INTEGER, d REAL, index_name HIDDEN, query HIDDEN, metric HIDDEN, k HIDDEN)");
    if (rc != SQLITE_OK) return rc;
    DistVtab *v = new (std::nothrow) DistVtab();
    if (!v) return SQLITE_NOMEM;
    memset(&v->base, 0, sizeof v->base);
    *out = &v->base;
    return SQLITE_OK;
}
int dist_disconnect(sqlite3_vtab *v) {
    delete (DistVtab *)v;
    return SQLITE_OK;
}
// idxNum = bit mask of the hidden columns given; argv order = index, query, metric, k (the ones present)
int dist_best_index(sqlite3_vtab *, sqlite3_index_info *info) {
    int slot[4] = {-1, -1, -1, -1};
    for (int i = 0; i < info->nConstraint; i++) {
        const auto &c = info->aConstraint[i];
        if (c.iColumn < COL_INDEX || c.iColumn > COL_K) continue;
        if (c.op != SQLITE_INDEX_CONSTRAINT_EQ) continue;
        if (!c.usable) return SQLITE_CONSTRAINT;  // an argument that depends on a later table: ask for another plan
        slot[c.iColumn - COL_INDEX] = i;
    }
    if (slot[0] < 0 || slot[1] < 0) return SQLITE_CONSTRAINT;  // pvs_dist needs at least (index, query)
    int argv = 1, mask = 0;
    for (int a = 0; a < 4; a++)
        if (slot[a] >= 0) {
            info->aConstraintUsage[slot[a]].argvIndex = argv++;
            info->aConstraintUsage[slot[a]].omit = 1;
            mask |= 1 << a;
        }
    info->idxNum = mask;
    info->estimatedCost = (mask & 8) ? 1000.0 : 1e7;
    info->estimatedRows = (mask & 8) ? 100 : 1000000;
    return SQLITE_OK;
}
int dist_open(sqlite3_vtab *, sqlite3_vtab_cursor **out) {
    DistCursor *c = new (std::nothrow) DistCursor();
    if (!c) return SQLITE_NOMEM;
    memset(&c->base, 0, sizeof c->base);
    *out = &c->base;
    return SQLITE_OK;
}
int dist_close(sqlite3_vtab_cursor *c) {
    delete (DistCursor *)c;
    return SQLITE_OK;
}
int vtab_error(sqlite3_vtab *v, const char *fmt, const char *detail) {
    if (v->zErrMsg) g_api.free(v->zErrMsg);
    v->zErrMsg = g_api.mprintf(fmt, detail);
    return SQLITE_ERROR;
}
int dist_filter(sqlite3_vtab_cursor *cur, int idxNum, const char *, int argc, sqlite3_value **argv) {
    DistCursor *c = (DistCursor *)cur;
    c->pos = c->n = 0;
    c->drop_column();
    int a = 0;
    sqlite3_value *v_index = (idxNum & 1) ? argv[a++] : nullptr, *v_query = (idxNum & 2) ? argv[a++] : nullptr,
                  *v_metric = (idxNum & 4) ? argv[a++] : nullptr, *v_k = (idxNum & 8) ? argv[a++] : nullptr;
    (void)argc;
    if (!v_index || !v_query) return vtab_error(cur->pVtab, "pvs_dist(index, query[, metric[, k]]): %s", "index and query are required");
    const unsigned char *nm = g_api.value_text(v_index);
    if (!nm) return vtab_error(cur->pVtab, "pvs_dist: %s", "index name must be text");
    pvs_index *ix = nullptr;
    uint32_t dtype = 0, dim = 0;
    uint64_t rows = 0;
    if (lookup((const char *)nm, &ix, &dtype, &dim, &rows) != PVS_OK)
        return vtab_error(cur->pVtab, "pvs_dist: no index is bound to the name '%s'", (const char *)nm);
    pvs_metric metric;
    if (parse_metric(v_metric, &metric) != SQLITE_OK) return vtab_error(cur->pVtab, "pvs_dist: %s", "metric must be 'cosine' or 'l2'");
    const void *q = nullptr;
    pvs_dtype qd = PVS_F32;
    if (const char *e = query_of(v_query, dtype, dim, &q, &qd)) return vtab_error(cur->pVtab, "pvs_dist: %s", e);
    if (v_k && g_api.value_type(v_k) != SQLITE_NULL) {
        const sqlite3_int64 k = g_api.value_int64(v_k);
        if (k < 1) return vtab_error(cur->pVtab, "pvs_dist: %s", "k must be a positive integer");  // preprocess.rs:436-446
        const uint32_t kk = (uint32_t)std::min<sqlite3_int64>(k, (sqlite3_int64)std::max<uint64_t>(rows, 1));
        c->ids.assign(kk, -1);
        c->d.assign(kk, 0.f);
        uint32_t cnt = 0;
        if (pvs_search(ix, q, qd, 1, kk, metric, c->ids.data(), c->d.data(), &cnt) != PVS_OK)
            return vtab_error(cur->pVtab, "pvs_dist: %s", pvs_last_error());
        c->n = cnt;
    } else {
        if (row_ids((const char *)nm, rows, &c->all_ids) != PVS_OK || pvs_score_column_create(ix, q, qd, metric, &c->col) != PVS_OK)
            return vtab_error(cur->pVtab, "pvs_dist: %s", pvs_last_error());
        uint64_t crows = 0;
        (void)pvs_score_column_rows(c->col, &crows);
        c->n = std::min<uint64_t>(crows, c->all_ids->size());
    }
    return SQLITE_OK;
}
int dist_next(sqlite3_vtab_cursor *c) {
    ((DistCursor *)c)->pos++;
    return SQLITE_OK;
}
int dist_eof(sqlite3_vtab_cursor *c) { return ((DistCursor *)c)->pos >= ((DistCursor *)c)->n; }
int dist_column(sqlite3_vtab_cursor *cur, sqlite3_context *ctx, int col) {
    DistCursor *c = (DistCursor *)cur;
    if (col == COL_ID)
        g_api.result_int64(ctx, c->col ? (*c->all_ids)[c->pos] : c->ids[c->pos]);
    else if (col == COL_D) {
        float d;
        if (c->col) {
            const uint64_t w0 = c->pos / DistCursor::WIN * DistCursor::WIN;
            if (w0 != c->win0) {
                const uint64_t n = std::min<uint64_t>(DistCursor::WIN, c->n - w0);
                c->win.resize(n);
                if (pvs_score_column_read(c->col, w0, n, c->win.data()) != PVS_OK) {
                    g_api.result_error(ctx, pvs_last_error(), -1);
                    return SQLITE_ERROR;
                }
                c->win0 = w0;
            }
            d = c->win[(size_t)(c->pos - w0)];
        } else {
            d = c->d[c->pos];
        }
        if (d != d)
            g_api.result_null(ctx);  // sqlite3_result_double(NaN) stores NULL; say so directly
        else
            g_api.result_double(ctx, (double)d);
    } else
        g_api.result_null(ctx);
    return SQLITE_OK;
}
int dist_rowid(sqlite3_vtab_cursor *c, sqlite3_int64 *out) {
    *out = (sqlite3_int64)((DistCursor *)c)->pos;
    return SQLITE_OK;
}
sqlite3_module g_dist_module = {
    /* iVersion */ 1,
    /* xCreate */ nullptr,  // eponymous-only: usable as a table-valued function, no CREATE VIRTUAL TABLE
    dist_connect, dist_best_index, dist_disconnect, /* xDestroy */ nullptr, dist_open, dist_close, dist_filter, dist_next, dist_eof,
    dist_column, dist_rowid, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};

// ------------------------------------------------------------------ pvs_distance_{cosine,l2}(index, id, query)
struct Column {  // one device pass, parked on the statement (auxiliary data of the query argument)
    std::shared_ptr<const std::vector<int64_t>> ids;  // shared with the binding
    pvs_column *col = nullptr;                         // the `d` column, kept by libpvs (in HBM for a single-device index)
    static constexpr uint64_t WIN = 1u << 18;          // rows per host window (1 MiB): SQL walks the ids in order, so a
    uint64_t win0 = UINT64_MAX;                        // statement reads each window once
    std::vector<float> win;
    ~Column() { pvs_score_column_destroy(col); }
};
void column_free(void *p) { delete (Column *)p; }

void distance_udf(sqlite3_context *ctx, int argc, sqlite3_value **argv) {
    (void)argc;
    const pvs_metric metric = (pvs_metric)(intptr_t)g_api.user_data(ctx);
    if (g_api.value_type(argv[1]) == SQLITE_NULL) {
        g_api.result_null(ctx);
        return;
    }
    Column *col = (Column *)g_api.get_auxdata(ctx, 2);
    if (!col) {
        const unsigned char *nm = g_api.value_text(argv[0]);
        pvs_index *ix = nullptr;
        uint32_t dtype = 0, dim = 0;
        uint64_t rows = 0;
        if (!nm || lookup((const char *)nm, &ix, &dtype, &dim, &rows) != PVS_OK) {
            g_api.result_error(ctx, "pvs_distance: no index is bound to that name", -1);
            return;
        }
        const void *q = nullptr;
        pvs_dtype qd = PVS_F32;
        if (const char *e = query_of(argv[2], dtype, dim, &q, &qd)) {
            g_api.result_error(ctx, e, -1);
            return;
        }
        col = new (std::nothrow) Column();
        if (!col) {
            g_api.result_error(ctx, "out of memory", -1);
            return;
        }
        if (row_ids((const char *)nm, rows, &col->ids) != PVS_OK || pvs_score_column_create(ix, q, qd, metric, &col->col) != PVS_OK) {
            g_api.result_error(ctx, pvs_last_error(), -1);
            delete col;
            return;
        }
        g_api.set_auxdata(ctx, 2, col, column_free);
        col = (Column *)g_api.get_auxdata(ctx, 2);  // SQLite may have dropped it right away (non-constant argument)
        if (!col) {
            g_api.result_error(ctx, "pvs_distance: the query argument must be constant within the statement (bind it as a parameter)", -1);
            return;
        }
    }
    const sqlite3_int64 id = g_api.value_int64(argv[1]);
    const std::vector<int64_t> &ids = *col->ids;
    uint64_t crows = 0;
    (void)pvs_score_column_rows(col->col, &crows);
    auto it = std::lower_bound(ids.begin(), ids.end(), (int64_t)id);
    const uint64_t row = (uint64_t)(it - ids.begin());
    if (it == ids.end() || *it != id || row >= crows) {
        g_api.result_null(ctx);  // the row is not in the index (the reference would have had no payload row to score either)
        return;
    }
    const uint64_t w0 = row / Column::WIN * Column::WIN;
    if (w0 != col->win0) {
        const uint64_t n = std::min<uint64_t>(Column::WIN, crows - w0);
        col->win.resize(n);
        if (pvs_score_column_read(col->col, w0, n, col->win.data()) != PVS_OK) {
            g_api.result_error(ctx, pvs_last_error(), -1);
            return;
        }
        col->win0 = w0;
    }
    const float d = col->win[(size_t)(row - w0)];
    if (d != d)
        g_api.result_null(ctx);
    else
        g_api.result_double(ctx, (double)d);
}

// ------------------------------------------------------------------ pvs_load(index, sql, params...)
// Index lifecycle glue in C: the rows go from SQLite's page cache into a staging buffer and from there to the
// device (pvs_index_add[_f32], which converts f32 rows to the index dtype on the device) without becoming values of the host
// language.  The statement is the loader's own — `SELECT d.id, d.item_id, e.embedding FROM item_data d JOIN embeddings e ...
// ORDER BY d.id` (panoptikon_amd/loader.py; the order BACKFILL_CHUNK_SQL streams, db/vector_quants.rs:1085-1099).
bool have_stmt_api() { return g_api.struct_size >= offsetof(pvs_sqlite_api, bind_blob) && g_api.prepare_v2 && g_api.step && g_api.finalize && g_api.column_blob; }
bool have_write_api() { return have_stmt_api() && g_api.struct_size >= sizeof(pvs_sqlite_api) && g_api.bind_blob && g_api.bind_int64 && g_api.reset; }

pvs_status stream_rows(void *stmt, pvs_index *ix, uint32_t chunk_rows, pvs_sqlite_load_result *res, std::string *err) {
    pvs_stats st;
    st.struct_size = sizeof st;
    pvs_status s = pvs_index_stats(ix, &st);
    if (s != PVS_OK) return s;
    const uint64_t dim = st.dim;
    if (!chunk_rows) chunk_rows = 65536;
    std::vector<int64_t> ids, groups;
    std::vector<uint8_t> payload;
    int kind = 0;  // bytes per component of the rows staged so far: 4 (f32) or 1 (int8 codes)
    bool any_group = false, any_null_group = false;
    ids.reserve(chunk_rows);
    groups.reserve(chunk_rows);
    auto flush = [&]() -> pvs_status {
        if (ids.empty()) return PVS_OK;
        if (any_group && any_null_group) {
            *err = "group ids must be given for every row or for none";
            return PVS_ERR_INVALID_ARG;
        }
        const int64_t *g = any_group ? groups.data() : nullptr;
        pvs_status r = kind == 4 ? pvs_index_add_f32(ix, (const float *)payload.data(), ids.size(), ids.data(), g, PVS_HOST)
                                 : pvs_index_add(ix, payload.data(), ids.size(), ids.data(), g, PVS_HOST);
        if (r != PVS_OK) return r;
        for (size_t i = 0; i < ids.size(); i++) {
            res->sum_id += (uint64_t)ids[i] & 0xffffffffull;
            res->sum_group += (uint64_t)groups[i] & 0xffffffffull;
        }
        res->rows += ids.size();
        res->last_id = ids.back();
        ids.clear();
        groups.clear();
        payload.clear();
        return PVS_OK;
    };
    for (;;) {
        const int rc = g_api.step(stmt);
        if (rc == SQLITE_DONE) break;
        if (rc != SQLITE_ROW) {
            *err = "the statement failed while streaming";
            return PVS_ERR_INVALID_ARG;
        }
        if (g_api.column_type(stmt, 2) != SQLITE_BLOB) {
            res->skipped++;
            continue;
        }
        const void *blob = g_api.column_blob(stmt, 2);  // (before column_bytes: sqlite.org/c3ref/column_blob.html)
        const uint64_t n = (uint64_t)g_api.column_bytes(stmt, 2);
        const int k = n == dim * 4 ? 4 : (n == dim && st.dtype == PVS_I8 ? 1 : 0);
        if (!k || !blob) {
            res->skipped++;
            continue;
        }
        if (kind && k != kind) {  // f32 rows after int8 codes or the other way round: separate add calls
            if ((s = flush()) != PVS_OK) return s;
        }
        kind = k;
        ids.push_back((int64_t)g_api.column_int64(stmt, 0));
        if (g_api.column_type(stmt, 1) == SQLITE_NULL) {
            any_null_group = true;
            groups.push_back(0);
        } else {
            any_group = true;
            groups.push_back((int64_t)g_api.column_int64(stmt, 1));
        }
        payload.insert(payload.end(), (const uint8_t *)blob, (const uint8_t *)blob + n);
        if ((ids.size() >= chunk_rows || payload.size() >= (64u << 20)) && (s = flush()) != PVS_OK) return s;  // (chunks of <= 64 MiB)
    }
    return flush();
}

void load_udf(sqlite3_context *ctx, int argc, sqlite3_value **argv) {
    if (argc < 2) {
        g_api.result_error(ctx, "pvs_load(index, sql[, param ...])", -1);
        return;
    }
    if (!have_stmt_api()) {
        g_api.result_error(ctx, "pvs_load: the host registered SQLite entry points without the statement interface", -1);
        return;
    }
    const unsigned char *nm = g_api.value_text(argv[0]);
    const unsigned char *sql = g_api.value_text(argv[1]);
    pvs_index *ix = nullptr;
    uint32_t dtype = 0, dim = 0;
    uint64_t rows = 0;
    if (!nm || !sql || lookup((const char *)nm, &ix, &dtype, &dim, &rows) != PVS_OK) {
        g_api.result_error(ctx, "pvs_load: no index is bound to that name", -1);
        return;
    }
    const std::string name((const char *)nm);
    sqlite3 *db = g_api.context_db_handle(ctx);
    void *stmt = nullptr;
    if (g_api.prepare_v2(db, (const char *)sql, -1, &stmt, nullptr) != SQLITE_OK || !stmt) {
        char *m = g_api.mprintf("pvs_load: %s", g_api.errmsg(db));
        g_api.result_error(ctx, m ? m : "pvs_load: prepare failed", -1);
        if (m) g_api.free(m);
        return;
    }
    for (int i = 2; i < argc; i++)
        if (g_api.bind_value(stmt, i - 1, argv[i]) != SQLITE_OK) {
            g_api.finalize(stmt);
            g_api.result_error(ctx, "pvs_load: more parameters than the statement has", -1);
            return;
        }
    pvs_sqlite_load_result res = {0, 0, -1, 0, 0};
    std::string err;
    const pvs_status s = stream_rows(stmt, ix, 0, &res, &err);
    g_api.finalize(stmt);
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_indexes.find(name);
        if (it != g_indexes.end()) it->second.last_load = res;
    }
    if (s != PVS_OK) {
        char *m = g_api.mprintf("pvs_load: %s (after %lld rows)", err.empty() ? pvs_last_error() : err.c_str(), (long long)res.rows);
        g_api.result_error(ctx, m ? m : "pvs_load failed", -1);
        if (m) g_api.free(m);
        return;
    }
    g_api.result_int64(ctx, (long long)res.rows);
}
void load_info_udf(sqlite3_context *ctx, int, sqlite3_value **argv) {
    const unsigned char *nm = g_api.value_text(argv[0]);
    pvs_sqlite_load_result r = {0, 0, -1, 0, 0};
    bool found = false;
    if (nm) {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_indexes.find((const char *)nm);
        if (it != g_indexes.end()) {
            r = it->second.last_load;
            found = true;
        }
    }
    if (!found || !g_api.result_text) {
        g_api.result_null(ctx);
        return;
    }
    char *m = g_api.mprintf("{\"rows\": %llu, \"skipped\": %llu, \"last_id\": %lld, \"sum_id\": %llu, \"sum_group\": %llu}", (unsigned long long)r.rows,
                            (unsigned long long)r.skipped, (long long)r.last_id, (unsigned long long)r.sum_id, (unsigned long long)r.sum_group);
    if (!m) {
        g_api.result_null(ctx);
        return;
    }
    g_api.result_text(ctx, m, -1, (void (*)(void *))(intptr_t)-1);  // SQLITE_TRANSIENT: SQLite copies
    g_api.free(m);
}

// ------------------------------------------------------------------ pvs_backfill(select_sql, upsert_sql, profile_id, device, params...)
// The write side of the lifecycle: the reference's backfill_chunk (db/vector_quants.rs:1119-1163) with quantize_int8 on the
// device.  `sel` is prepared and bound by the caller; it is stepped to completion BEFORE the first write (it reads
// embedding_quants itself — NOT EXISTS — and the reference fetches the whole chunk first too).
thread_local pvs_sqlite_backfill_result t_last_backfill = {0, -1};
// where the last pvs_backfill of this thread spent its time: stepping the host's SELECT (SQLite page reads + one copy per blob),
// the device codec (upload, quantize_int8, download), the host's UPSERT (three b-tree inserts per row in the reference's schema)
thread_local double t_backfill_phase_ms[3] = {0, 0, 0};
static inline double now_ms() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

pvs_status backfill_run(sqlite3 *db, void *sel, const char *upsert_sql, int64_t profile_id, int32_t device, pvs_sqlite_backfill_result *res,
                        std::string *err) {
    std::vector<int64_t> ids, revs;
    std::vector<float> scales;
    std::vector<uint8_t> payload;
    uint64_t row_bytes = 0;
    const double t0 = now_ms();
    t_backfill_phase_ms[0] = t_backfill_phase_ms[1] = t_backfill_phase_ms[2] = 0;
    for (;;) {
        const int rc = g_api.step(sel);
        if (rc == SQLITE_DONE) break;
        if (rc != SQLITE_ROW) {
            *err = std::string("the select failed: ") + g_api.errmsg(db);
            return PVS_ERR_INVALID_ARG;
        }
        if (g_api.column_type(sel, 1) != SQLITE_BLOB || g_api.column_type(sel, 2) != SQLITE_BLOB) {
            *err = "pvs_backfill: the select must yield (id, embedding BLOB, artifact BLOB, artifact_rev)";
            return PVS_ERR_INVALID_ARG;
        }
        const void *emb = g_api.column_blob(sel, 1);
        const uint64_t nb = (uint64_t)g_api.column_bytes(sel, 1);
        if (!emb || nb == 0 || nb % 4 != 0 || (row_bytes && nb != row_bytes)) {
            *err = "pvs_backfill: embeddings must be non-empty f32 blobs of one length (guard the select with length(e.embedding) = c.dim * 4)";
            return PVS_ERR_DIM_MISMATCH;
        }
        row_bytes = nb;
        payload.insert(payload.end(), (const uint8_t *)emb, (const uint8_t *)emb + nb);
        const void *art = g_api.column_blob(sel, 2);
        const uint64_t na = (uint64_t)g_api.column_bytes(sel, 2);
        float scale = 0.f;
        if (!art || pvs_artifact_scale((const uint8_t *)art, (size_t)na, &scale) != PVS_OK) {
            *err = "Invalid vector quant scale artifact";  // (the reference's message, :1141-1148: refuse to backfill)
            return PVS_ERR_INVALID_ARG;
        }
        scales.push_back(scale);
        ids.push_back((int64_t)g_api.column_int64(sel, 0));
        revs.push_back((int64_t)g_api.column_int64(sel, 3));
    }
    res->written = 0;
    res->cursor = -1;
    const double t1 = now_ms();
    t_backfill_phase_ms[0] = t1 - t0;
    if (ids.empty()) return PVS_OK;
    const uint64_t dim = row_bytes / 4, n = ids.size();
    std::vector<int8_t> codes(n * dim);
    for (uint64_t i = 0; i < n;) {  // one device pass per run of rows sharing a scale (one pair: one run)
        uint64_t j = i + 1;
        while (j < n && memcmp(&scales[j], &scales[i], 4) == 0) j++;
        const pvs_status qs = pvs_quantize_i8((const float *)payload.data() + i * dim, (j - i) * dim, scales[i], codes.data() + i * dim, PVS_HOST, device);
        if (qs != PVS_OK) {
            *err = pvs_last_error();
            return qs;
        }
        i = j;
    }
    const double t2 = now_ms();
    t_backfill_phase_ms[1] = t2 - t1;
    void *up = nullptr;
    if (g_api.prepare_v2(db, upsert_sql, -1, &up, nullptr) != SQLITE_OK || !up) {
        *err = std::string("the upsert does not prepare: ") + g_api.errmsg(db);
        return PVS_ERR_INVALID_ARG;
    }
    pvs_status st = PVS_OK;
    for (uint64_t i = 0; i < n; i++) {
        g_api.reset(up);
        if (g_api.bind_int64(up, 1, ids[i]) != SQLITE_OK || g_api.bind_int64(up, 2, profile_id) != SQLITE_OK || g_api.bind_int64(up, 3, revs[i]) != SQLITE_OK ||
            g_api.bind_blob(up, 4, codes.data() + i * dim, (int)dim, nullptr /* SQLITE_STATIC: codes outlives the step */) != SQLITE_OK ||
            g_api.step(up) != SQLITE_DONE) {
            *err = std::string("the upsert failed: ") + g_api.errmsg(db);
            st = PVS_ERR_INVALID_ARG;
            break;
        }
        res->written++;
        res->cursor = std::max(res->cursor, ids[i]);
    }
    g_api.finalize(up);
    t_backfill_phase_ms[2] = now_ms() - t2;
    return st;
}
// SELECT pvs_backfill_phases() -> 'select_ms,quantize_ms,upsert_ms' of this connection thread's last pvs_backfill (measurement aid)
void backfill_phases_udf(sqlite3_context *ctx, int, sqlite3_value **) {
    char buf[96];
    snprintf(buf, sizeof buf, "%.3f,%.3f,%.3f", t_backfill_phase_ms[0], t_backfill_phase_ms[1], t_backfill_phase_ms[2]);
    g_api.result_text(ctx, buf, -1, (void (*)(void *))(intptr_t)-1);  // SQLITE_TRANSIENT
}

void backfill_udf(sqlite3_context *ctx, int argc, sqlite3_value **argv) {
    if (argc < 4) {
        g_api.result_error(ctx, "pvs_backfill(select_sql, upsert_sql, profile_id, device[, select parameter ...])", -1);
        return;
    }
    const unsigned char *sel_sql = g_api.value_text(argv[0]);
    const unsigned char *up_sql = g_api.value_text(argv[1]);
    if (!sel_sql || !up_sql) {
        g_api.result_error(ctx, "pvs_backfill: NULL statement text", -1);
        return;
    }
    const std::string up_copy((const char *)up_sql);  // (value_text pointers do not survive other calls on the same value set)
    sqlite3 *db = g_api.context_db_handle(ctx);
    void *sel = nullptr;
    if (g_api.prepare_v2(db, (const char *)sel_sql, -1, &sel, nullptr) != SQLITE_OK || !sel) {
        char *m = g_api.mprintf("pvs_backfill: %s", g_api.errmsg(db));
        g_api.result_error(ctx, m ? m : "pvs_backfill: prepare failed", -1);
        if (m) g_api.free(m);
        return;
    }
    const long long profile_id = g_api.value_int64(argv[2]);
    const int device = (int)g_api.value_int64(argv[3]);
    for (int i = 4; i < argc; i++)
        if (g_api.bind_value(sel, i - 3, argv[i]) != SQLITE_OK) {
            g_api.finalize(sel);
            g_api.result_error(ctx, "pvs_backfill: more parameters than the select has", -1);
            return;
        }
    pvs_sqlite_backfill_result res = {0, -1};
    std::string err;
    const pvs_status s = backfill_run(db, sel, up_copy.c_str(), profile_id, device, &res, &err);
    g_api.finalize(sel);
    t_last_backfill = res;
    if (s != PVS_OK) {
        char *m = g_api.mprintf("pvs_backfill: %s (after %lld rows)", err.empty() ? pvs_last_error() : err.c_str(), (long long)res.written);
        g_api.result_error(ctx, m ? m : "pvs_backfill failed", -1);
        if (m) g_api.free(m);
        return;
    }
    g_api.result_int64(ctx, (long long)res.written);
}
void backfill_cursor_udf(sqlite3_context *ctx, int, sqlite3_value **) {
    if (t_last_backfill.cursor < 0 && t_last_backfill.written == 0)
        g_api.result_null(ctx);
    else
        g_api.result_int64(ctx, (long long)t_last_backfill.cursor);
}

// pvs_ready_pair(profile_name, setter_name, ...): resolve_ready_pair (db/vector_quants.rs:1795-1869) for hosts that are not the
// reference's Rust — NULL unless the profile is active and every existing setter's (profile, setter) pair is `ready` with a usable
// scale artifact and a dimension, all pairs sharing one (scale, dim); setter names without a `setters` row are skipped.  Returns
// {"profile_id": .., "scale": .., "dim": ..} (scale printed with 9 significant digits: it round-trips the f32).  The probes are
// the reference's; they run as nested statements on the calling connection, the arguments bound as the sqlite3_values they are.
struct Stmt {
    void *s = nullptr;  // (statements are opaque pointers in pvs_sqlite_api)
    ~Stmt() {
        if (s) g_api.finalize(s);
    }
};
void ready_pair_udf(sqlite3_context *ctx, int argc, sqlite3_value **argv) {
    if (argc < 2) {
        g_api.result_error(ctx, "pvs_ready_pair(profile_name, setter_name, ...)", -1);
        return;
    }
    sqlite3 *db = g_api.context_db_handle(ctx);
    auto fail = [&](const char *what) {
        char *m = g_api.mprintf("pvs_ready_pair: %s: %s", what, g_api.errmsg(db));
        g_api.result_error(ctx, m ? m : "pvs_ready_pair failed", -1);
        if (m) g_api.free(m);
    };
    Stmt prof, setter, cov;
    if (g_api.prepare_v2(db, "SELECT id FROM vector_quant_profiles WHERE name = ?1 AND state = 'active'", -1, &prof.s, nullptr) != SQLITE_OK ||
        g_api.prepare_v2(db, "SELECT id FROM setters WHERE name = ?1", -1, &setter.s, nullptr) != SQLITE_OK ||
        g_api.prepare_v2(db, "SELECT artifact, dim FROM vector_quant_coverage WHERE profile_id = ?1 AND setter_id = ?2 AND state = 'ready'", -1,
                         &cov.s, nullptr) != SQLITE_OK)
        return fail("prepare");
    if (g_api.bind_value(prof.s, 1, argv[0]) != SQLITE_OK) return fail("bind");
    int rc = g_api.step(prof.s);
    if (rc == SQLITE_DONE) return g_api.result_null(ctx);  // no active profile of that name
    if (rc != SQLITE_ROW) return fail("profile probe");
    const long long profile_id = g_api.column_int64(prof.s, 0);
    bool have = false;
    float scale = 0.f;
    long long dim = 0;
    for (int a = 1; a < argc; a++) {
        if (g_api.reset(setter.s) != SQLITE_OK || g_api.bind_value(setter.s, 1, argv[a]) != SQLITE_OK) return fail("bind");
        rc = g_api.step(setter.s);
        if (rc == SQLITE_DONE) continue;  // unknown setter: skipped
        if (rc != SQLITE_ROW) return fail("setter probe");
        const long long sid = g_api.column_int64(setter.s, 0);
        if (g_api.reset(cov.s) != SQLITE_OK || g_api.bind_int64(cov.s, 1, profile_id) != SQLITE_OK || g_api.bind_int64(cov.s, 2, sid) != SQLITE_OK)
            return fail("bind");
        rc = g_api.step(cov.s);
        if (rc == SQLITE_DONE) return g_api.result_null(ctx);  // this pair is not ready
        if (rc != SQLITE_ROW) return fail("coverage probe");
        float sc = 0.f;
        if (g_api.column_type(cov.s, 0) != SQLITE_BLOB || g_api.column_type(cov.s, 1) == SQLITE_NULL ||
            pvs_artifact_scale((const uint8_t *)g_api.column_blob(cov.s, 0), (size_t)g_api.column_bytes(cov.s, 0), &sc) != PVS_OK)
            return g_api.result_null(ctx);  // no dimension or no usable scale: not queryable whatever the state column says
        const long long d = g_api.column_int64(cov.s, 1);
        if (!have) {
            have = true;
            scale = sc;
            dim = d;
        } else if (scale != sc || dim != d) {
            return g_api.result_null(ctx);  // siblings must share one artifact: a rebuild is pending
        }
    }
    if (!have) return g_api.result_null(ctx);
    char *m = g_api.mprintf("{\"profile_id\": %lld, \"scale\": %.9g, \"dim\": %lld}", profile_id, (double)scale, dim);
    if (!m) return g_api.result_null(ctx);
    g_api.result_text(ctx, m, -1, (void (*)(void *))(intptr_t)-1);
    g_api.free(m);
}

int register_all(sqlite3 *db) {
    int rc = g_api.create_module_v2(db, "pvs_dist", &g_dist_module, nullptr, nullptr);
    if (rc != SQLITE_OK) return rc;
    if (have_stmt_api()) {
        rc = g_api.create_function_v2(db, "pvs_load", -1, SQLITE_UTF8, nullptr, load_udf, nullptr, nullptr, nullptr);
        if (rc != SQLITE_OK) return rc;
        rc = g_api.create_function_v2(db, "pvs_load_info", 1, SQLITE_UTF8, nullptr, load_info_udf, nullptr, nullptr, nullptr);
        if (rc != SQLITE_OK) return rc;
    }
    if (have_write_api()) {
        rc = g_api.create_function_v2(db, "pvs_backfill", -1, SQLITE_UTF8, nullptr, backfill_udf, nullptr, nullptr, nullptr);
        if (rc != SQLITE_OK) return rc;
        rc = g_api.create_function_v2(db, "pvs_backfill_cursor", 0, SQLITE_UTF8, nullptr, backfill_cursor_udf, nullptr, nullptr, nullptr);
    if (rc == SQLITE_OK)
        rc = g_api.create_function_v2(db, "pvs_backfill_phases", 0, SQLITE_UTF8, nullptr, backfill_phases_udf, nullptr, nullptr, nullptr);
        if (rc != SQLITE_OK) return rc;
        rc = g_api.create_function_v2(db, "pvs_ready_pair", -1, SQLITE_UTF8, nullptr, ready_pair_udf, nullptr, nullptr, nullptr);
        if (rc != SQLITE_OK) return rc;
    }
    // not SQLITE_DETERMINISTIC: the result depends on the bound index's contents, which SQLite cannot see
    rc = g_api.create_function_v2(db, "pvs_distance_cosine", 3, SQLITE_UTF8, (void *)(intptr_t)PVS_COSINE, distance_udf, nullptr, nullptr, nullptr);
    if (rc != SQLITE_OK) return rc;
    return g_api.create_function_v2(db, "pvs_distance_l2", 3, SQLITE_UTF8, (void *)(intptr_t)PVS_L2, distance_udf, nullptr, nullptr, nullptr);
}

// Slots of sqlite3_api_routines (sqlite3ext.h), an append-only struct of function pointers: the index of each entry point
// this file uses.  create_function_v2 (3.7.3) is the youngest; libversion_number gates the read.
enum ApiSlot {
    SLOT_bind_blob = 2, SLOT_bind_int64 = 5, SLOT_bind_value = 12, SLOT_column_blob = 19, SLOT_column_bytes = 20, SLOT_column_int64 = 29,
    SLOT_column_type = 38, SLOT_declare_vtab = 50, SLOT_errmsg = 53, SLOT_finalize = 57, SLOT_free = 58, SLOT_get_auxdata = 61,
    SLOT_libversion_number = 67, SLOT_mprintf = 69, SLOT_reset = 77, SLOT_result_double = 79, SLOT_result_error = 80,
    SLOT_result_int64 = 83, SLOT_result_null = 84, SLOT_result_text = 85, SLOT_set_auxdata = 92, SLOT_step = 94, SLOT_user_data = 101,
    SLOT_value_blob = 102, SLOT_value_bytes = 103, SLOT_value_int64 = 107, SLOT_value_text = 109, SLOT_value_type = 113,
    SLOT_prepare_v2 = 116, SLOT_create_module_v2 = 119, SLOT_context_db_handle = 149, SLOT_create_function_v2 = 162
};
bool fill_api_from_routines(const void *pApi, const char **why) {
    void *const *slot = (void *const *)pApi;
    int (*libversion_number)(void) = (int (*)(void))slot[SLOT_libversion_number];
    if (!libversion_number || libversion_number() < 3008000) {
        *why = "libpvs_sqlite needs SQLite >= 3.8.0";
        return false;
    }
    pvs_sqlite_api a;
    memset(&a, 0, sizeof a);
    a.struct_size = sizeof a;
#define SLOT(field) a.field = (decltype(a.field))slot[SLOT_##field];
    SLOT(create_function_v2) SLOT(create_module_v2) SLOT(declare_vtab) SLOT(value_type) SLOT(value_bytes) SLOT(value_blob)
    SLOT(value_text) SLOT(value_int64) SLOT(result_double) SLOT(result_int64) SLOT(result_null) SLOT(result_error) SLOT(user_data)
    SLOT(get_auxdata) SLOT(set_auxdata) SLOT(mprintf) SLOT(free) SLOT(prepare_v2) SLOT(step) SLOT(finalize) SLOT(column_type)
    SLOT(column_blob) SLOT(column_bytes) SLOT(column_int64) SLOT(bind_value) SLOT(context_db_handle) SLOT(errmsg) SLOT(result_text)
    SLOT(bind_blob) SLOT(bind_int64) SLOT(reset)
#undef SLOT
    g_api = a;
    g_api_set = true;
    return true;
}

// By-name lookup in the process image: only behind the explicit pvs_sqlite_register(db, NULL) / pvs_sqlite_load of a process
// that never went through an extension entry point (the Python test harness's ctypes route).  The process's own image first;
// a loaded libsqlite3.so.0 only when the image exports nothing.
bool fill_api_from_process(std::string *missing) {
    void *h = dlopen(nullptr, RTLD_NOW);
    if (h && !dlsym(h, "sqlite3_create_function_v2")) h = nullptr;
    if (!h) h = dlopen("libsqlite3.so.0", RTLD_NOW | RTLD_NOLOAD);
    if (!h) return false;
    pvs_sqlite_api a;
    memset(&a, 0, sizeof a);
    a.struct_size = sizeof a;
#define SYM(field, name)                                   \
    a.field = (decltype(a.field))dlsym(h, name);           \
    if (!a.field) {                                        \
        *missing = name;                                   \
        return false;                                      \
    }
    SYM(create_function_v2, "sqlite3_create_function_v2")
    SYM(create_module_v2, "sqlite3_create_module_v2")
    SYM(declare_vtab, "sqlite3_declare_vtab")
    SYM(value_type, "sqlite3_value_type")
    SYM(value_bytes, "sqlite3_value_bytes")
    SYM(value_blob, "sqlite3_value_blob")
    SYM(value_text, "sqlite3_value_text")
    SYM(value_int64, "sqlite3_value_int64")
    SYM(result_double, "sqlite3_result_double")
    SYM(result_int64, "sqlite3_result_int64")
    SYM(result_null, "sqlite3_result_null")
    SYM(result_error, "sqlite3_result_error")
    SYM(user_data, "sqlite3_user_data")
    SYM(get_auxdata, "sqlite3_get_auxdata")
    SYM(set_auxdata, "sqlite3_set_auxdata")
    SYM(mprintf, "sqlite3_mprintf")
    SYM(free, "sqlite3_free")
    SYM(prepare_v2, "sqlite3_prepare_v2")
    SYM(step, "sqlite3_step")
    SYM(finalize, "sqlite3_finalize")
    SYM(column_type, "sqlite3_column_type")
    SYM(column_blob, "sqlite3_column_blob")
    SYM(column_bytes, "sqlite3_column_bytes")
    SYM(column_int64, "sqlite3_column_int64")
    SYM(bind_value, "sqlite3_bind_value")
    SYM(context_db_handle, "sqlite3_context_db_handle")
    SYM(errmsg, "sqlite3_errmsg")
    SYM(result_text, "sqlite3_result_text")
    SYM(bind_blob, "sqlite3_bind_blob")
    SYM(bind_int64, "sqlite3_bind_int64")
    SYM(reset, "sqlite3_reset")
#undef SYM
    g_api = a;
    g_api_set = true;
    return true;
}
}  // namespace

// ---- registration
PVS_EXPORT int32_t pvs_sqlite_register(void *db, const pvs_sqlite_api *api) {
    if (!db) return SQLITE_ERROR;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (api) {
            const size_t v1 = offsetof(pvs_sqlite_api, prepare_v2);  // the struct before the statement interface was added
            if (api->struct_size < v1) return SQLITE_ERROR;
            memset(&g_api, 0, sizeof g_api);
            memcpy(&g_api, api, std::min<size_t>(api->struct_size, sizeof g_api));
            g_api_set = true;
        } else if (!g_api_set) {
            std::string missing;
            if (!fill_api_from_process(&missing)) return SQLITE_ERROR;
        }
    }
    return register_all((sqlite3 *)db);
}
// loadable-extension entry points: `.load libpvs_sqlite`, sqlite3_load_extension, Python's Connection.load_extension,
// or sqlite3_auto_extension(sqlite3_pvs_init) exactly where the reference registers sqlite3_vec_init
// (db/sql_functions.rs:105-128).  Every SQLite entry point comes from the third argument.
PVS_EXPORT int sqlite3_pvs_init(void *db, char **pzErrMsg, const void *pApi) {
    if (!db || !pApi) return SQLITE_ERROR;  // (an entry point is only ever called by SQLite, which always passes its table)
    const char *why = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (!fill_api_from_routines(pApi, &why)) {
            char *(*mprintf)(const char *, ...) = (char *(*)(const char *, ...))((void *const *)pApi)[SLOT_mprintf];
            if (pzErrMsg && mprintf) *pzErrMsg = mprintf("%s", why);
            return SQLITE_ERROR;
        }
    }
    const int rc = register_all((sqlite3 *)db);
    if (rc != SQLITE_OK && pzErrMsg) *pzErrMsg = g_api.mprintf("%s", "libpvs_sqlite: registration failed");
    return rc;
}
PVS_EXPORT int sqlite3_extension_init(void *db, char **pzErrMsg, const void *pApi) { return sqlite3_pvs_init(db, pzErrMsg, pApi); }
PVS_EXPORT int sqlite3_pvssqlite_init(void *db, char **pzErrMsg, const void *pApi) { return sqlite3_pvs_init(db, pzErrMsg, pApi); }

// ---- the row streamer for a host that holds the connection itself
PVS_EXPORT int32_t pvs_sqlite_load(void *db, const char *sql, pvs_index *idx, uint32_t chunk_rows, pvs_sqlite_load_result *out) {
    if (!db || !sql || !idx) return PVS_ERR_INVALID_ARG;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (!g_api_set) {
            std::string missing;
            if (!fill_api_from_process(&missing)) return PVS_ERR_STATE;
        }
    }
    if (!have_stmt_api()) return PVS_ERR_STATE;
    void *stmt = nullptr;
    if (g_api.prepare_v2((sqlite3 *)db, sql, -1, &stmt, nullptr) != SQLITE_OK || !stmt) return PVS_ERR_INVALID_ARG;
    pvs_sqlite_load_result res = {0, 0, -1, 0, 0};
    std::string err;
    const pvs_status s = stream_rows(stmt, idx, chunk_rows, &res, &err);
    g_api.finalize(stmt);
    if (out) *out = res;
    return s;
}

PVS_EXPORT int32_t pvs_sqlite_api_snapshot(pvs_sqlite_api *out) {
    if (!out) return 1;
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_api_set) return 1;
    const uint32_t want = out->struct_size;
    memcpy(out, &g_api, std::min<size_t>(want, sizeof g_api));
    out->struct_size = (uint32_t)std::min<size_t>(want, g_api.struct_size);
    return 0;
}

// ---- the backfill for a host that holds the connection itself (select without parameters)
PVS_EXPORT int32_t pvs_sqlite_backfill(void *db, const char *select_sql, const char *upsert_sql, int64_t profile_id, int32_t device,
                                       pvs_sqlite_backfill_result *out) {
    if (!db || !select_sql || !upsert_sql) return PVS_ERR_INVALID_ARG;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (!g_api_set) {
            std::string missing;
            if (!fill_api_from_process(&missing)) return PVS_ERR_STATE;
        }
    }
    if (!have_write_api()) return PVS_ERR_STATE;
    void *sel = nullptr;
    if (g_api.prepare_v2((sqlite3 *)db, select_sql, -1, &sel, nullptr) != SQLITE_OK || !sel) return PVS_ERR_INVALID_ARG;
    pvs_sqlite_backfill_result res = {0, -1};
    std::string err;
    const pvs_status s = backfill_run((sqlite3 *)db, sel, upsert_sql, profile_id, device, &res, &err);
    g_api.finalize(sel);
    if (out) *out = res;
    return s;
}

// ---- index registry
PVS_EXPORT int32_t pvs_sqlite_bind_index(const char *name, pvs_index *idx) {
    if (!name || !*name || !idx) return PVS_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(g_mu);
    Bound b;
    b.ix = idx;
    g_indexes[name] = b;
    return PVS_OK;
}
PVS_EXPORT int32_t pvs_sqlite_unbind_index(const char *name) {
    if (!name) return PVS_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(g_mu);
    return g_indexes.erase(name) ? PVS_OK : PVS_ERR_INVALID_ARG;
}
