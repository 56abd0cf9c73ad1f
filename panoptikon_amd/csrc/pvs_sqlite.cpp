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
// No SQLite headers exist on this image (and a Rust host links its own copy, libsqlite3-sys), so nothing here includes
// sqlite3.h or links libsqlite3: the ~20 entry points used are declared below with their documented prototypes and reach
// the library through a table of function pointers — filled by the host (pvs_sqlite_register, for a statically linked
// SQLite) or, for a loadable extension, from the SQLite the calling process already carries (dlsym).
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <map>
#include <mutex>
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
    std::vector<int64_t> ids;  // host copy of the row ids (strictly increasing), refreshed when the row count moves
    uint64_t ids_rows = UINT64_MAX;
};
std::map<std::string, Bound> g_indexes;

pvs_status lookup(const std::string &name, pvs_index **ix, uint32_t *dtype, uint32_t *dim, uint64_t *rows) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_indexes.find(name);
    if (it == g_indexes.end()) return PVS_ERR_INVALID_ARG;
    pvs_stats st;
    pvs_status s = pvs_index_stats(it->second.ix, &st);
    if (s != PVS_OK) return s;
    *ix = it->second.ix;
    *dtype = st.dtype;
    *dim = st.dim;
    *rows = st.rows;
    return PVS_OK;
}
// a copy of the index's row ids (cached per binding)
pvs_status row_ids(const std::string &name, uint64_t rows, std::vector<int64_t> *out) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_indexes.find(name);
    if (it == g_indexes.end()) return PVS_ERR_INVALID_ARG;
    Bound &b = it->second;
    if (b.ids_rows != rows) {
        b.ids.resize(rows);
        if (rows) {
            pvs_status s = pvs_index_read_ids(b.ix, 0, rows, b.ids.data(), nullptr);
            if (s != PVS_OK) return s;
        }
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
    std::vector<int64_t> ids;
    std::vector<float> d;
    size_t pos = 0, n = 0;
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
        if (row_ids((const char *)nm, rows, &c->ids) != PVS_OK) return vtab_error(cur->pVtab, "pvs_dist: %s", pvs_last_error());
        c->d.assign(rows, 0.f);
        if (rows && pvs_score_all(ix, q, qd, metric, c->d.data(), PVS_HOST) != PVS_OK)
            return vtab_error(cur->pVtab, "pvs_dist: %s", pvs_last_error());
        c->n = rows;
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
        g_api.result_int64(ctx, c->ids[c->pos]);
    else if (col == COL_D) {
        const float d = c->d[c->pos];
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
    std::vector<int64_t> ids;
    std::vector<float> d;
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
        col->d.assign(rows, 0.f);
        if (row_ids((const char *)nm, rows, &col->ids) != PVS_OK || (rows && pvs_score_all(ix, q, qd, metric, col->d.data(), PVS_HOST) != PVS_OK)) {
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
    auto it = std::lower_bound(col->ids.begin(), col->ids.end(), (int64_t)id);
    if (it == col->ids.end() || *it != id) {
        g_api.result_null(ctx);  // the row is not in the index (the reference would have had no payload row to score either)
        return;
    }
    const float d = col->d[(size_t)(it - col->ids.begin())];
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
bool have_stmt_api() { return g_api.struct_size >= sizeof(pvs_sqlite_api) && g_api.prepare_v2 && g_api.step && g_api.finalize && g_api.column_blob; }

pvs_status stream_rows(void *stmt, pvs_index *ix, uint32_t chunk_rows, pvs_sqlite_load_result *res, std::string *err) {
    pvs_stats st;
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

int register_all(sqlite3 *db) {
    int rc = g_api.create_module_v2(db, "pvs_dist", &g_dist_module, nullptr, nullptr);
    if (rc != SQLITE_OK) return rc;
    if (have_stmt_api()) {
        rc = g_api.create_function_v2(db, "pvs_load", -1, SQLITE_UTF8, nullptr, load_udf, nullptr, nullptr, nullptr);
        if (rc != SQLITE_OK) return rc;
        rc = g_api.create_function_v2(db, "pvs_load_info", 1, SQLITE_UTF8, nullptr, load_info_udf, nullptr, nullptr, nullptr);
        if (rc != SQLITE_OK) return rc;
    }
    // not SQLITE_DETERMINISTIC: the result depends on the bound index's contents, which SQLite cannot see
    rc = g_api.create_function_v2(db, "pvs_distance_cosine", 3, SQLITE_UTF8, (void *)(intptr_t)PVS_COSINE, distance_udf, nullptr, nullptr, nullptr);
    if (rc != SQLITE_OK) return rc;
    return g_api.create_function_v2(db, "pvs_distance_l2", 3, SQLITE_UTF8, (void *)(intptr_t)PVS_L2, distance_udf, nullptr, nullptr, nullptr);
}

bool fill_api_from_process(std::string *missing) {
    void *h = dlopen("libsqlite3.so.0", RTLD_NOW | RTLD_NOLOAD);  // the copy the host process already loaded
    if (!h) h = dlopen(nullptr, RTLD_NOW);                        // or the host binary's own (statically linked) SQLite
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
// (db/sql_functions.rs:105-128).  The third argument (sqlite3_api_routines*) is not used: see the file comment.
PVS_EXPORT int sqlite3_pvs_init(void *db, char **pzErrMsg, const void *pApi) {
    (void)pApi;
    const int rc = pvs_sqlite_register(db, nullptr);
    if (rc != SQLITE_OK && pzErrMsg && g_api_set) *pzErrMsg = g_api.mprintf("%s", "libpvs_sqlite: registration failed");
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
