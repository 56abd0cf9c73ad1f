// pvs_host.cpp — host-side mirror of the reference's operator interface around the scan:
// scale artifact, per-item aggregation, rank / RRF, .npy query ingestion and the quant
// resolution policy.  These are the pieces the reference also runs on the host (Rust or
// SQL); none of them scores a corpus row.  Plain C++17, no HIP calls.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#include "pvs.h"

#define PVS_EXPORT extern "C" __attribute__((visibility("default")))
pvs_status pvs_fail(pvs_status code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

// --------------------------------------------------------- scale artifact
// db/vector_quants.rs:1465-1471
PVS_EXPORT float pvs_scale_from_absmax(float absmax) {
    if (absmax > 0.0f && std::isfinite(absmax)) return absmax / 127.0f;
    return 1.0f;
}
// db/vector_quants.rs:1449-1451
PVS_EXPORT void pvs_scale_artifact(float scale, uint8_t out[4]) { memcpy(out, &scale, 4); }
// db/vector_quants.rs:1456-1460
PVS_EXPORT pvs_status pvs_artifact_scale(const uint8_t *artifact, size_t len, float *scale) {
    if (!scale) return pvs_fail(PVS_ERR_INVALID_ARG, "null argument");
    if (!artifact || len != 4) return pvs_fail(PVS_ERR_INVALID_ARG, "scale artifact must be exactly 4 bytes");
    float s;
    memcpy(&s, artifact, 4);
    if (!(std::isfinite(s) && s > 0.0f)) return pvs_fail(PVS_ERR_INVALID_ARG, "scale artifact is not a positive finite f32");
    *scale = s;
    return PVS_OK;
}

// ------------------------------------------------------------- aggregation
// SQLite SUM()/AVG(): Kahan-Babuska-Neumaier compensated f64 sums.
namespace {
struct Kbn {
    double s = 0, c = 0;
    void step(double r) {
        const double t = s + r;
        if (std::fabs(s) > std::fabs(r))
            c += (s - t) + r;
        else
            c += (r - t) + s;
        s = t;
    }
    double value() const { return s + c; }
};
}  // namespace

// filters/exact.rs:67-80 rank_aggregate over GROUP BY file_id (builder.rs:829-835)
PVS_EXPORT pvs_status pvs_aggregate(const float *dist, const float *weights, const int64_t *group_ids, uint64_t n, pvs_agg agg,
                                    int64_t *out_groups, double *out_values, uint64_t *out_n) {
    if (!out_n || (n && (!dist || !group_ids || !out_groups || !out_values))) return pvs_fail(PVS_ERR_INVALID_ARG, "null argument");
    if (!weights && agg != PVS_AGG_MIN && agg != PVS_AGG_MAX && agg != PVS_AGG_AVG)
        return pvs_fail(PVS_ERR_INVALID_ARG, "aggregation must be MIN, MAX or AVG");
    uint64_t g = 0, i = 0;
    while (i < n) {
        uint64_t j = i;
        Kbn sum, wsum;
        double mn = INFINITY, mx = -INFINITY;
        uint64_t cnt = 0;
        for (; j < n && group_ids[j] == group_ids[i]; j++) {
            if (weights) wsum.step((double)weights[j]);  // SUM(w) runs over every row of the group
            if (std::isnan(dist[j])) continue;           // SQL NULL distance: d (and d*w) is ignored by the aggregates
            const double d = (double)dist[j];
            if (weights) {
                sum.step(d * (double)weights[j]);
            } else {
                sum.step(d);
            }
            mn = std::min(mn, d);
            mx = std::max(mx, d);
            cnt++;
        }
        if (j < n && group_ids[j] < group_ids[i]) return pvs_fail(PVS_ERR_INVALID_ARG, "group_ids must be non-decreasing");
        double v;
        if (cnt == 0)
            v = NAN;
        else if (weights)
            v = sum.value() / wsum.value();
        else if (agg == PVS_AGG_MIN)
            v = mn;
        else if (agg == PVS_AGG_MAX)
            v = mx;
        else
            v = sum.value() / (double)cnt;
        out_groups[g] = group_ids[i];
        out_values[g] = v;
        g++;
        i = j;
    }
    *out_n = g;
    return PVS_OK;
}

// ------------------------------------------------------------- rank / RRF
// builder.rs:757-771: row_number() OVER (ORDER BY value ASC); NULLs last; ties by id
PVS_EXPORT pvs_status pvs_row_number_dir(const double *values, const int64_t *ids, uint64_t n, int32_t descending, int64_t *out_rank) {
    if (n && (!values || !out_rank)) return pvs_fail(PVS_ERR_INVALID_ARG, "null argument");
    std::vector<uint64_t> idx(n);
    std::iota(idx.begin(), idx.end(), 0);
    // The reference's window carries no NULLS clause, so SQLite's default holds: NULL is the smallest value —
    // first ascending, last descending.  Ties by id ascending (the build's deterministic tie-break).
    std::sort(idx.begin(), idx.end(), [&](uint64_t a, uint64_t b) {
        const bool na = std::isnan(values[a]), nb = std::isnan(values[b]);
        if (na != nb) return descending ? nb : na;
        if (!na && values[a] != values[b]) return descending ? values[a] > values[b] : values[a] < values[b];
        const int64_t ia = ids ? ids[a] : (int64_t)a, ib = ids ? ids[b] : (int64_t)b;
        return ia < ib;
    });
    for (uint64_t r = 0; r < n; r++) out_rank[idx[r]] = (int64_t)r + 1;
    return PVS_OK;
}
PVS_EXPORT pvs_status pvs_row_number(const double *values, const int64_t *ids, uint64_t n, int64_t *out_rank) {
    return pvs_row_number_dir(values, ids, n, 0, out_rank);
}

// builder.rs:17-18, 1284-1301
PVS_EXPORT pvs_status pvs_rrf_fuse(const int64_t *ranks, uint32_t n_branches, uint64_t n, const int32_t *ks, const double *weights,
                                   double *out_fused) {
    if (n && n_branches && (!ranks || !ks || !weights || !out_fused)) return pvs_fail(PVS_ERR_INVALID_ARG, "null argument");
    const int64_t BIG = 9223372036854775805LL;
    for (uint64_t i = 0; i < n; i++) {
        double total = 0.0;
        for (uint32_t b = 0; b < n_branches; b++) {
            const int64_t r = ranks[(uint64_t)b * n + i];
            const int64_t rank = r < 0 ? BIG : r;
            int64_t di;
            // SQLite integer addition; falls back to REAL on i64 overflow
            const double denom = __builtin_add_overflow((int64_t)ks[b], rank, &di) ? (double)ks[b] + (double)rank : (double)di;
            const double term = (1.0 / denom) * weights[b];
            total = b == 0 ? term : total + term;
        }
        out_fused[i] = total;
    }
    return PVS_OK;
}

// builder.rs:1303-1317: filters of the same order priority WITHOUT rrf coalesce into
//   min(coalesce(rank_1, 9223372036854775805), ...)   ascending
//   max(coalesce(rank_1, -9223372036854775805), ...)  descending
// (SQLite's multi-argument min()/max(); a filter that did not return the row contributes its NULL).
static const int64_t PVS_VERY_LARGE = 9223372036854775805LL;  // builder.rs:17-18
PVS_EXPORT pvs_status pvs_coalesce_ranks(const int64_t *ranks, uint32_t n_filters, uint64_t n, int32_t descending, int64_t *out) {
    if (n && n_filters && (!ranks || !out)) return pvs_fail(PVS_ERR_INVALID_ARG, "null argument");
    if (n_filters == 0) return pvs_fail(PVS_ERR_INVALID_ARG, "at least one filter");
    for (uint64_t i = 0; i < n; i++) {
        int64_t best = 0;
        for (uint32_t b = 0; b < n_filters; b++) {
            const int64_t r = ranks[(uint64_t)b * n + i];
            const int64_t v = r < 0 ? (descending ? -PVS_VERY_LARGE : PVS_VERY_LARGE) : r;
            best = b == 0 ? v : (descending ? std::max(best, v) : std::min(best, v));
        }
        out[i] = best;
    }
    return PVS_OK;
}
// the same over raw aggregates (order_rank without row_n is the f64 aggregate itself): NaN = NULL.  In SQL the fallback is
// an INTEGER and SQLite compares integers with reals exactly; as an f64 result that is indistinguishable from comparing
// with the fallback's nearest double (+-2^63): a real on the other side of it is on the other side of 2^63 too.
PVS_EXPORT pvs_status pvs_coalesce_values(const double *values, uint32_t n_filters, uint64_t n, int32_t descending, double *out) {
    if (n && n_filters && (!values || !out)) return pvs_fail(PVS_ERR_INVALID_ARG, "null argument");
    if (n_filters == 0) return pvs_fail(PVS_ERR_INVALID_ARG, "at least one filter");
    const double big = descending ? -9223372036854775808.0 : 9223372036854775808.0;
    for (uint64_t i = 0; i < n; i++) {
        double best = 0.0;
        for (uint32_t b = 0; b < n_filters; b++) {
            double v = values[(uint64_t)b * n + i];
            if (std::isnan(v)) v = big;
            best = b == 0 ? v : (descending ? std::max(best, v) : std::min(best, v));
        }
        out[i] = best;
    }
    return PVS_OK;
}

// builder.rs:781-815 apply_sort_bounds: `WHERE order_rank > gt AND order_rank < lt` on the wrapped filter CTE.  keep[i] = 1 for
// rows that stay; a NULL order_rank fails both comparisons (SQL three-valued logic).  have_gt / have_lt select the bounds.
PVS_EXPORT pvs_status pvs_sort_bounds(const double *order_rank, uint64_t n, int32_t have_gt, double gt, int32_t have_lt, double lt,
                                      uint8_t *keep) {
    if (n && (!order_rank || !keep)) return pvs_fail(PVS_ERR_INVALID_ARG, "null argument");
    for (uint64_t i = 0; i < n; i++) {
        const double v = order_rank[i];
        bool ok = true;
        if (have_gt) ok = ok && v > gt;  // false for NaN
        if (have_lt) ok = ok && v < lt;
        if (!have_gt && !have_lt) ok = true;  // no bounds: the reference does not wrap the query at all (NULL rows stay)
        keep[i] = ok ? 1 : 0;
    }
    return PVS_OK;
}

// ------------------------------------------------------------ host k-way merge
PVS_EXPORT pvs_status pvs_merge_topk(const int64_t *ids, const float *dist, const uint32_t *counts, uint32_t world, uint32_t batch,
                                     uint32_t k, int64_t *out_ids, float *out_dist, uint32_t *out_count) {
    return pvs_merge_topk_keyed(ids, dist, nullptr, counts, world, batch, k, out_ids, out_dist, out_count);
}
// keys ([world][batch][k], optional): the second sort key of every entry (pvs_index_set_order_keys): (distance asc, NULL last,
// key DESC, id asc) — the order every other route produces when the index carries keys
PVS_EXPORT pvs_status pvs_merge_topk_keyed(const int64_t *ids, const float *dist, const int64_t *keys, const uint32_t *counts, uint32_t world,
                                           uint32_t batch, uint32_t k, int64_t *out_ids, float *out_dist, uint32_t *out_count) {
    if (!ids || !dist || !counts || !out_ids || !out_dist || !out_count) return pvs_fail(PVS_ERR_INVALID_ARG, "null argument");
    struct E {
        float d;
        int64_t id, key;
    };
    auto less = [](const E &a, const E &b) {
        const bool na = std::isnan(a.d), nb = std::isnan(b.d);
        if (na != nb) return nb;
        if (!na && a.d != b.d) return a.d < b.d;
        if (a.key != b.key) return a.key > b.key;
        return a.id < b.id;
    };
    std::vector<E> all;
    for (uint32_t q = 0; q < batch; q++) {
        all.clear();
        for (uint32_t w = 0; w < world; w++) {
            const size_t off = ((size_t)w * batch + q) * k;
            const uint32_t c = std::min(counts[(size_t)w * batch + q], k);
            for (uint32_t p = 0; p < c; p++) all.push_back({dist[off + p], ids[off + p], keys ? keys[off + p] : 0});
        }
        std::sort(all.begin(), all.end(), less);
        const uint32_t nout = (uint32_t)std::min<size_t>(all.size(), k);
        for (uint32_t i = 0; i < k; i++) {
            out_ids[(size_t)q * k + i] = i < nout ? all[i].id : -1;
            out_dist[(size_t)q * k + i] = i < nout ? all[i].d : NAN;
        }
        out_count[q] = nout;
    }
    return PVS_OK;
}

PVS_EXPORT pvs_status pvs_merge_group_pages_keyed(const int64_t *groups, const double *values, const int64_t *keys, const uint32_t *counts,
                                                  uint32_t world, uint32_t batch, uint32_t k, int64_t *out_groups, double *out_values,
                                                  uint32_t *out_count) {
    if (!groups || !values || !counts || !out_groups || !out_values || !out_count) return pvs_fail(PVS_ERR_INVALID_ARG, "null argument");
    struct E {
        double v;
        int64_t key, g;
    };
    auto less = [](const E &a, const E &b) {  // value asc, NULL last, then the second sort key DESC (all zero without keys), then group id
        const bool na = std::isnan(a.v), nb = std::isnan(b.v);
        if (na != nb) return nb;
        if (!na && a.v != b.v) return a.v < b.v;
        if (a.key != b.key) return a.key > b.key;
        return a.g < b.g;
    };
    std::vector<E> all;
    std::vector<int64_t> seen;
    for (uint32_t q = 0; q < batch; q++) {
        all.clear();
        for (uint32_t w = 0; w < world; w++) {
            const size_t off = ((size_t)w * batch + q) * k;
            const uint32_t c = std::min(counts[(size_t)w * batch + q], k);
            for (uint32_t p = 0; p < c; p++) all.push_back({values[off + p], keys ? keys[off + p] : 0, groups[off + p]});
        }
        seen.resize(all.size());
        for (size_t i = 0; i < all.size(); i++) seen[i] = all[i].g;
        std::sort(seen.begin(), seen.end());
        for (size_t i = 1; i < seen.size(); i++)
            if (seen[i] == seen[i - 1]) return pvs_fail(PVS_ERR_INVALID_ARG, "group %lld appears on two shards: shard by group", (long long)seen[i]);
        std::sort(all.begin(), all.end(), less);
        const uint32_t nout = (uint32_t)std::min<size_t>(all.size(), k);
        for (uint32_t i = 0; i < k; i++) {
            out_groups[(size_t)q * k + i] = i < nout ? all[i].g : -1;
            out_values[(size_t)q * k + i] = i < nout ? all[i].v : NAN;
        }
        out_count[q] = nout;
    }
    return PVS_OK;
}
PVS_EXPORT pvs_status pvs_merge_group_pages(const int64_t *groups, const double *values, const uint32_t *counts, uint32_t world,
                                            uint32_t batch, uint32_t k, int64_t *out_groups, double *out_values, uint32_t *out_count) {
    return pvs_merge_group_pages_keyed(groups, values, nullptr, counts, world, batch, k, out_groups, out_values, out_count);
}

// --------------------------------------------------------------- .npy ingestion
// pql/embedding_utils.rs:37-350.  Error strings follow the reference.
namespace {
struct NpyDtype {
    char kind;  // 'f','i','u','b'
    size_t size;
};

bool find_key(const std::string &h, const char *key, size_t *after) {
    const std::string ks = std::string("'") + key + "'", kd = std::string("\"") + key + "\"";
    size_t p = h.find(ks);
    if (p == std::string::npos) p = h.find(kd);
    if (p == std::string::npos) return false;
    *after = p + ks.size();
    return true;
}
std::string ltrim(const std::string &s) {
    size_t i = 0;
    while (i < s.size() && isspace((unsigned char)s[i])) i++;
    return s.substr(i);
}
bool parse_str_value(const std::string &h, const char *key, std::string *out) {
    size_t a;
    if (!find_key(h, key, &a)) return false;
    std::string rest = h.substr(a);
    size_t colon = rest.find(':');
    if (colon == std::string::npos) return false;
    std::string v = ltrim(rest.substr(colon + 1));
    if (v.empty() || (v[0] != '\'' && v[0] != '"')) return false;
    const char quote = v[0];
    v = v.substr(1);
    size_t end = v.find(quote);
    if (end == std::string::npos) return false;
    *out = v.substr(0, end);
    return true;
}
bool parse_bool_value(const std::string &h, const char *key, bool *out) {
    size_t a;
    if (!find_key(h, key, &a)) return false;
    std::string rest = h.substr(a);
    size_t colon = rest.find(':');
    if (colon == std::string::npos) return false;
    std::string v = ltrim(rest.substr(colon + 1));
    if (v.rfind("True", 0) == 0) {
        *out = true;
        return true;
    }
    if (v.rfind("False", 0) == 0) {
        *out = false;
        return true;
    }
    return false;
}
bool parse_shape(const std::string &h, std::vector<size_t> *shape) {
    size_t a;
    if (!find_key(h, "shape", &a)) return false;
    std::string rest = h.substr(a);
    size_t colon = rest.find(':');
    if (colon == std::string::npos) return false;
    std::string v = ltrim(rest.substr(colon + 1));
    size_t s = v.find('('), e = v.find(')');
    if (s == std::string::npos || e == std::string::npos || e < s) return false;
    std::string body = v.substr(s + 1, e - s - 1);
    size_t pos = 0;
    while (pos <= body.size()) {
        size_t comma = body.find(',', pos);
        std::string part = body.substr(pos, comma == std::string::npos ? std::string::npos : comma - pos);
        size_t b0 = 0, b1 = part.size();
        while (b0 < b1 && isspace((unsigned char)part[b0])) b0++;
        while (b1 > b0 && isspace((unsigned char)part[b1 - 1])) b1--;
        part = part.substr(b0, b1 - b0);
        if (!part.empty()) {
            for (char ch : part)
                if (!isdigit((unsigned char)ch)) return false;
            shape->push_back((size_t)strtoull(part.c_str(), nullptr, 10));
        }
        if (comma == std::string::npos) break;
        pos = comma + 1;
    }
    return true;
}

// the reference's f16_to_f32 (embedding_utils.rs:323-350), INCLUDING its treatment of
// subnormal halves: the exponent starts at -1 and loses one per shift, then gains
// (1 + 127 - 15), which yields half the IEEE value.  Parity = same bytes out.
float ref_f16_to_f32(uint16_t bits) {
    const uint32_t sign = (bits >> 15) & 1u, exp = (bits >> 10) & 0x1fu;
    uint32_t mant = bits & 0x3ffu, out;
    if (exp == 0) {
        if (mant == 0) {
            out = sign << 31;
        } else {
            int e = -1;
            while ((mant & 0x400u) == 0) {
                mant <<= 1;
                e -= 1;
            }
            mant &= 0x3ffu;
            out = (sign << 31) | ((uint32_t)(e + 1 + 127 - 15) << 23) | (mant << 13);
        }
    } else if (exp == 0x1f) {
        out = (sign << 31) | (0xffu << 23) | (mant ? (mant << 13) : 0u);
    } else {
        out = (sign << 31) | ((exp + 127 - 15) << 23) | (mant << 13);
    }
    float f;
    memcpy(&f, &out, 4);
    return f;
}

template <typename T>
T read_uint(const uint8_t *p, bool le) {
    T v = 0;
    for (size_t i = 0; i < sizeof(T); i++) v |= (T)p[le ? i : sizeof(T) - 1 - i] << (8 * i);
    return v;
}
}  // namespace

PVS_EXPORT pvs_status pvs_npy_to_f32(const uint8_t *buf, size_t len, float *out, size_t out_cap, size_t *out_n) {
    if (!buf || !out_n) return pvs_fail(PVS_ERR_INVALID_ARG, "null argument");
    if (len < 10) return pvs_fail(PVS_ERR_PARSE, "Numpy buffer too small");
    if (memcmp(buf, "\x93NUMPY", 6) != 0) return pvs_fail(PVS_ERR_PARSE, "Invalid numpy magic header");
    const uint8_t major = buf[6], minor = buf[7];
    size_t header_len, header_start;
    if (major == 1) {
        header_len = (size_t)buf[8] | ((size_t)buf[9] << 8);
        header_start = 10;
    } else if (major == 2 || major == 3) {
        if (len < 12) return pvs_fail(PVS_ERR_PARSE, "Numpy buffer too small");
        header_len = (size_t)read_uint<uint32_t>(buf + 8, true);
        header_start = 12;
    } else {
        return pvs_fail(PVS_ERR_PARSE, "Unsupported numpy version %u.%u", major, minor);
    }
    const size_t header_end = header_start + header_len;
    if (header_end < header_start || header_end > len) return pvs_fail(PVS_ERR_PARSE, "Numpy header truncated");
    const std::string header((const char *)buf + header_start, header_len);
    std::string descr;
    bool fortran = false;
    std::vector<size_t> shape;
    if (!parse_str_value(header, "descr", &descr)) return pvs_fail(PVS_ERR_PARSE, "Numpy header missing descr");
    if (!parse_bool_value(header, "fortran_order", &fortran)) return pvs_fail(PVS_ERR_PARSE, "Numpy header missing fortran_order");
    if (!parse_shape(header, &shape)) return pvs_fail(PVS_ERR_PARSE, "Numpy header missing shape");

    NpyDtype dt;
    bool le = true;
    if (descr == "?") {
        dt = {'b', 1};
    } else {
        if (descr.size() < 2) return pvs_fail(PVS_ERR_PARSE, "Invalid numpy descr");
        const char endian = descr[0];
        if (endian == '<' || endian == '|' || endian == '=')
            le = true;  // '=' is native; gfx950 hosts are little-endian
        else if (endian == '>')
            le = false;
        else
            return pvs_fail(PVS_ERR_PARSE, "Unsupported numpy endian in descr: %s", descr.c_str());
        const char kind = descr[1];
        const std::string num = descr.substr(2);
        if (num.empty() || !std::all_of(num.begin(), num.end(), [](char c) { return isdigit((unsigned char)c); }))
            return pvs_fail(PVS_ERR_PARSE, "Unsupported numpy dtype: %s", descr.c_str());
        if (kind != 'f' && kind != 'i' && kind != 'u' && kind != 'b') return pvs_fail(PVS_ERR_PARSE, "Unsupported numpy dtype: %s", descr.c_str());
        dt = {kind, (size_t)strtoull(num.c_str(), nullptr, 10)};
    }
    if (shape.empty()) return pvs_fail(PVS_ERR_PARSE, "Numpy array has empty shape");
    if (shape.size() > 2) return pvs_fail(PVS_ERR_PARSE, "Only 1D or 2D embeddings are supported");
    size_t total = 1;
    for (size_t s : shape) {
        if (s && total > SIZE_MAX / s) return pvs_fail(PVS_ERR_PARSE, "Embedding size overflow");
        total *= s;
    }
    if (dt.size && total > SIZE_MAX / dt.size) return pvs_fail(PVS_ERR_PARSE, "Embedding size overflow");
    const size_t total_bytes = total * dt.size;
    if (header_end + total_bytes < header_end) return pvs_fail(PVS_ERR_PARSE, "Embedding data overflow");
    if (header_end + total_bytes > len) return pvs_fail(PVS_ERR_PARSE, "Numpy data truncated");
    const uint8_t *data = buf + header_end;
    const size_t row_len = shape.size() == 1 ? shape[0] : shape[1];
    *out_n = row_len;
    // validate the scalar type even when only the length is requested
    const bool ok_size = (dt.kind == 'f' && (dt.size == 2 || dt.size == 4 || dt.size == 8)) ||
                         ((dt.kind == 'i' || dt.kind == 'u') && (dt.size == 1 || dt.size == 2 || dt.size == 4 || dt.size == 8)) ||
                         (dt.kind == 'b' && dt.size == 1);
    if (!ok_size && row_len > 0) {
        const char *what = dt.kind == 'f' ? "float" : dt.kind == 'i' ? "int" : dt.kind == 'u' ? "uint" : "bool";
        return pvs_fail(PVS_ERR_PARSE, "Unsupported %s size: %zu", what, dt.size);
    }
    if (!out) return PVS_OK;
    if (out_cap < row_len) return pvs_fail(PVS_ERR_INVALID_ARG, "output buffer holds %zu components, need %zu", out_cap, row_len);
    for (size_t idx = 0; idx < row_len; idx++) {
        // first row of a 2-D array: C order -> idx, Fortran order -> idx * shape[0]
        const size_t elem = (shape.size() == 1 || !fortran) ? idx : idx * shape[0];
        const size_t start = elem * dt.size;
        if (start + dt.size > total_bytes) return pvs_fail(PVS_ERR_PARSE, "Numpy data truncated");
        const uint8_t *p = data + start;
        float v = 0.f;
        if (dt.kind == 'f') {
            if (dt.size == 2) {
                v = ref_f16_to_f32(read_uint<uint16_t>(p, le));
            } else if (dt.size == 4) {
                const uint32_t b = read_uint<uint32_t>(p, le);
                memcpy(&v, &b, 4);
            } else {
                const uint64_t b = read_uint<uint64_t>(p, le);
                double d;
                memcpy(&d, &b, 8);
                v = (float)d;
            }
        } else if (dt.kind == 'i') {
            switch (dt.size) {
                case 1: v = (float)(int8_t)p[0]; break;
                case 2: v = (float)(int16_t)read_uint<uint16_t>(p, le); break;
                case 4: v = (float)(int32_t)read_uint<uint32_t>(p, le); break;
                default: v = (float)(int64_t)read_uint<uint64_t>(p, le); break;
            }
        } else if (dt.kind == 'u') {
            switch (dt.size) {
                case 1: v = (float)p[0]; break;
                case 2: v = (float)read_uint<uint16_t>(p, le); break;
                case 4: v = (float)read_uint<uint32_t>(p, le); break;
                default: v = (float)read_uint<uint64_t>(p, le); break;
            }
        } else {
            v = p[0] == 0 ? 0.0f : 1.0f;
        }
        out[idx] = v;
    }
    return PVS_OK;
}

// --------------------------------------------------------- quant resolution
// quantize_int8 for ONE query vector on the host, exactly where the reference runs it
// (compute_query_quant inside resolve_vector_quant, pql/preprocess.rs:370-386).  The corpus
// side of the codec runs on the GPU (pvs_quantize_i8 / pvs_index_add_f32).
static void host_query_quant(const float *x, size_t n, float scale, int8_t *out) {
    for (size_t i = 0; i < n; i++) {
        float q = rintf(x[i] / scale);  // FE_TONEAREST: round-half-to-even
        if (q < -128.0f) q = -128.0f;
        if (q > 127.0f) q = 127.0f;
        out[i] = std::isnan(q) ? (int8_t)0 : (int8_t)q;
    }
}

PVS_EXPORT pvs_status pvs_resolve_vector_quant(pvs_index_mode index, const char *variant, int64_t k, const pvs_ready_pair *pair,
                                               const uint8_t *embedding, size_t embedding_len, int8_t *query_quant_out,
                                               size_t query_quant_cap, pvs_quant_resolved *out) {
    if (!out) return pvs_fail(PVS_ERR_INVALID_ARG, "null argument");
    out->use_quant = 0;
    out->profile_id = 0;
    out->query_quant_len = 0;
    // validate_quant_args (preprocess.rs:436-446)
    if (index == PVS_INDEX_ANN) return pvs_fail(PVS_ERR_INVALID_ARG, "index \"ann\" is reserved and not yet available");
    if (index != PVS_INDEX_AUTO && index != PVS_INDEX_EXACT && index != PVS_INDEX_QUANT) return pvs_fail(PVS_ERR_INVALID_ARG, "unknown index mode");
    if (k < 1) return pvs_fail(PVS_ERR_INVALID_ARG, "k must be a positive integer");
    // quant_requested (:413-421)
    if (index == PVS_INDEX_EXACT) return PVS_OK;
    // normalize_variant (:425): blank / whitespace = unset
    bool named = false;
    if (variant)
        for (const char *p = variant; *p; p++)
            if (!isspace((unsigned char)*p)) named = true;
    const bool strict = index == PVS_INDEX_QUANT || named;
    if (!pair || !pair->have_db_context) {
        if (strict) return pvs_fail(PVS_ERR_NOT_READY, "vector quant profiles are unavailable in this context");
        return PVS_OK;
    }
    if (!named && !pair->have_default_profile) {
        if (strict) return pvs_fail(PVS_ERR_NOT_READY, "no default vector quant profile is configured");
        return PVS_OK;
    }
    if (!pair->pair_ready) {
        if (strict) return pvs_fail(PVS_ERR_NOT_READY, "vector quant profile does not exist or is not ready for this model");
        return PVS_OK;
    }
    if (embedding) {
        if ((int64_t)embedding_len != pair->dim * 4) {
            if (strict)
                return pvs_fail(PVS_ERR_DIM_MISMATCH, "query embedding dimension mismatch (expected %lld, got %zu)", (long long)pair->dim,
                                embedding_len / 4);
            return PVS_OK;  // auto: silently fall back to exact
        }
        const size_t dim = (size_t)pair->dim;
        if (!query_quant_out || query_quant_cap < dim) return pvs_fail(PVS_ERR_INVALID_ARG, "query_quant_out too small");
        std::vector<float> q(dim);
        memcpy(q.data(), embedding, dim * 4);
        host_query_quant(q.data(), dim, pair->scale, query_quant_out);
        out->query_quant_len = dim;
    }
    out->use_quant = 1;
    out->profile_id = pair->profile_id;
    return PVS_OK;
}
