/*
 * pvs_oracle.c — CPU restatement of Panoptikon's vector-similarity hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, bench.py's
 * `cpu_baseline` leg and __graft_entry__.smoke() may load it.  The product
 * library (libpvs.so, panoptikon_amd/csrc) never links, loads or calls it.
 *
 * What it restates (citations are paths under the reference tree,
 * /root/reference/, studied read-only; no reference source is copied):
 *
 *   codec          panoptikon/src/db/vector_quants.rs:1440-1503
 *   scale/artifact panoptikon/src/db/vector_quants.rs:1449-1471, 1474-1483, 1513-1554
 *   distances      sqlite-vec 0.1.9 (crates.io; pinned by panoptikon/Cargo.toml:84,
 *                  Cargo.lock:6122-6128).  ITS SOURCE IS NOT IN THE REFERENCE TREE
 *                  and is not on this machine.  The four kernels below restate the
 *                  published scalar algorithm of sqlite-vec's
 *                  vec_distance_L2 / vec_distance_cosine for float32 and int8
 *                  vectors (sequential f32 accumulation, sqrt in double, result
 *                  narrowed to f32, returned to SQL as a double).
 *   call sites     panoptikon/src/pql/builder/filters/image_embeddings.rs:321-362,
 *                  text_embeddings.rs:386-418, item_similarity.rs:503-521
 *   aggregate      panoptikon/src/pql/builder/filters/exact.rs:67-80 (MIN/MAX/AVG,
 *                  SUM(d*w)/SUM(w)), executed by SQLite in f64
 *   rank / RRF     panoptikon/src/pql/builder.rs:757-771, 1284-1317
 *
 * PINNING STATUS
 *   codec, absmax, scale, artifact : pinned bit-exact by the reference's own
 *       known-answer test int8_codec_rounds_ties_to_even_and_clamps
 *       (db/vector_quants.rs:3588-3626) and build_uses_absmax_scale_artifact
 *       (:2194-2244); see tests/test_oracle_golden.py.
 *   int8 distances : pinned to 1e-4 by the reference's
 *       sqlite_vec_int8_distances_match_a_rust_reference (:3632-3687), the only
 *       numeric test the reference holds at the sqlite-vec boundary.  Below 2^24
 *       every partial sum is an exactly representable integer, so the result is a
 *       pure function of the integer sums and is order independent.
 *   f32 distances : PARITY UNPINNED at the bit level — the reference holds no
 *       numeric test for them (only tools/pql-equivalence at rtol 1e-4/atol 1e-6,
 *       which cannot run here).  The scalar two-rounding (mul, then add) x86-64
 *       baseline evaluation order is assumed.
 *   aggregate / row_number / RRF : pinned against SQLite itself (python's stdlib sqlite3 is the engine) for the
 *       SQL-level semantics — NULL placement in the window, the RRF term incl. the k + BIG integer overflow,
 *       MIN/MAX/AVG/SUM(d*w)/SUM(w) over NULL distances (tests/test_oracle_golden.py).
 *
 * Tie-break added by the build (not in the reference):
 *   (distance ascending, row id ascending); NaN distances (SQL NULL) sort last.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

enum { ORC_F32 = 0, ORC_F16 = 1, ORC_I8 = 2 };
enum { ORC_COSINE = 0, ORC_L2 = 1 };
enum { ORC_AGG_NONE = 0, ORC_AGG_MIN = 1, ORC_AGG_MAX = 2, ORC_AGG_AVG = 3 };

/* ------------------------------------------------------------------ codec */

/* db/vector_quants.rs:1446 INT8_MAX_CODE */
static const float INT8_MAX_CODE = 127.0f;

/* db/vector_quants.rs:1465-1471 scale_from_absmax */
ORC_API float orc_scale_from_absmax(float absmax) {
    if (absmax > 0.0f && isfinite(absmax)) return absmax / INT8_MAX_CODE;
    return 1.0f;
}

/* db/vector_quants.rs:1474-1483 blob_absmax: NaN components are ignored because
 * `value > absmax` is false for NaN; +-inf propagates (then scale_from_absmax
 * returns 1.0). */
ORC_API float orc_blob_absmax(const float *x, size_t n) {
    float absmax = 0.0f;
    for (size_t i = 0; i < n; i++) {
        float v = fabsf(x[i]);
        if (v > absmax) absmax = v;
    }
    return absmax;
}

/* db/vector_quants.rs:1449-1451 scale_artifact */
ORC_API void orc_scale_artifact(float scale, uint8_t out[4]) { memcpy(out, &scale, 4); }

/* db/vector_quants.rs:1456-1460 artifact_scale; returns 1 and writes *scale when
 * usable, 0 for "None" (wrong length, non-finite, <= 0). */
ORC_API int orc_artifact_scale(const uint8_t *artifact, size_t len, float *scale) {
    if (len != 4) return 0;
    float s;
    memcpy(&s, artifact, 4);
    if (isfinite(s) && s > 0.0f) {
        *scale = s;
        return 1;
    }
    return 0;
}

/* Rust `f32::round_ties_even` */
static float round_ties_even_f32(float v) {
    /* rintf under the default FE_TONEAREST mode is round-half-to-even */
    return rintf(v);
}

/* db/vector_quants.rs:1489-1497 quantize_int8:
 *   code = (x / s).round_ties_even().clamp(-128, 127) as i8
 * Rust `as i8` on a float saturates and maps NaN to 0; clamp() propagates NaN. */
ORC_API void orc_quantize_int8(const float *x, size_t n, float scale, int8_t *out) {
    for (size_t i = 0; i < n; i++) {
        float q = round_ties_even_f32(x[i] / scale);
        if (q < -128.0f) q = -128.0f;
        if (q > INT8_MAX_CODE) q = INT8_MAX_CODE;
        out[i] = isnan(q) ? (int8_t)0 : (int8_t)q;
    }
}

/* db/vector_quants.rs:1513-1554 compute_int8_scale_artifact over a dense
 * [n][dim] block: returns 0 ("None") when n == 0. */
ORC_API int orc_compute_int8_scale(const float *rows, size_t n, size_t dim, float *scale) {
    if (n == 0) return 0;
    float absmax = 0.0f;
    for (size_t r = 0; r < n; r++) {
        float m = orc_blob_absmax(rows + r * dim, dim);
        absmax = fmaxf(absmax, m); /* f32::max: NaN-ignoring, like fmaxf */
    }
    *scale = orc_scale_from_absmax(absmax);
    return 1;
}

/* ------------------------------------------------------------- f16 helpers */

/* IEEE binary16 -> binary32 widening (exact).  This is the build's definition of
 * "the value of an f16 corpus element" (SURVEY §7 fp16 parity definition). */
ORC_API float orc_f16_to_f32(uint16_t bits) {
    uint32_t sign = (bits >> 15) & 1u, exp = (bits >> 10) & 0x1fu, mant = bits & 0x3ffu, out;
    if (exp == 0) {
        if (mant == 0) {
            out = sign << 31;
        } else {
            int s = 0;
            while ((mant & 0x400u) == 0) {
                mant <<= 1;
                s++;
            }
            mant &= 0x3ffu;
            out = (sign << 31) | ((uint32_t)(127 - 14 - s) << 23) | (mant << 13);
        }
    } else if (exp == 0x1f) {
        out = (sign << 31) | (0xffu << 23) | (mant << 13);
    } else {
        out = (sign << 31) | ((exp + 127 - 15) << 23) | (mant << 13);
    }
    float f;
    memcpy(&f, &out, 4);
    return f;
}

/* pql/embedding_utils.rs:323-350 f16_to_f32 AS THE REFERENCE WRITES IT, used only
 * by NPY query ingestion (embedding_utils.rs:229-265).  For subnormal halves the
 * reference starts its exponent at -1 and subtracts one per normalising shift,
 * then adds (1 + 127 - 15): the result is HALF the IEEE value (0x0001 -> 2^-25
 * instead of 2^-24).  Normal, zero, inf and NaN inputs are IEEE-exact.  Parity
 * means reproducing this, so the host NPY path mirrors it. */
ORC_API float orc_npy_f16_to_f32(uint16_t bits) {
    uint32_t sign = (bits >> 15) & 1u, exp = (bits >> 10) & 0x1fu, mant = bits & 0x3ffu, out;
    if (exp == 0 && mant != 0) {
        int e = -1;
        while ((mant & 0x400u) == 0) {
            mant <<= 1;
            e -= 1;
        }
        mant &= 0x3ffu;
        out = (sign << 31) | ((uint32_t)(e + 1 + 127 - 15) << 23) | (mant << 13);
        float f;
        memcpy(&f, &out, 4);
        return f;
    }
    return orc_f16_to_f32(bits);
}

/* f32 -> f16 round-to-nearest-even (the build's device corpus format) */
ORC_API uint16_t orc_f32_to_f16(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t absx = x & 0x7fffffffu;
    if (absx >= 0x7f800000u) { /* inf / nan */
        return (uint16_t)(sign | 0x7c00u | ((absx > 0x7f800000u) ? (0x200u | ((absx >> 13) & 0x3ffu)) : 0));
    }
    if (absx >= 0x477ff000u) { /* rounds to >= 65520 -> inf */
        return (uint16_t)(sign | 0x7c00u);
    }
    if (absx < 0x33000001u) { /* < 2^-25 (or exactly 2^-25 ties to even 0) */
        return (uint16_t)sign;
    }
    int32_t e = (int32_t)(absx >> 23) - 127;
    uint32_t m = (absx & 0x7fffffu) | 0x800000u;
    uint32_t shift, half;
    if (e < -14) { /* subnormal half */
        shift = (uint32_t)(13 + (-14 - e));
        half = 0;
    } else {
        shift = 13;
        half = (uint32_t)(e + 15) << 10;
    }
    uint32_t q = m >> shift;
    uint32_t rem = m & ((1u << shift) - 1u);
    uint32_t halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (q & 1u))) q++;
    if (e < -14) return (uint16_t)(sign | q);             /* q may carry into exp=1: fine */
    return (uint16_t)(sign | (half + (q - 0x400u)));       /* mantissa carry bumps exponent */
}

/* -------------------------------------------------- sqlite-vec distances */

/* vec_distance_L2(float32, float32): f32 res=0; for i: t=a-b; res+=t*t; return sqrt(res)
 * (sqrt evaluated in double, narrowed to f32; SQL result is that f32 as double). */
ORC_API double orc_vec_distance_l2_f32(const float *a, const float *b, size_t d) {
    float res = 0.0f; /* built with -ffp-contract=off, no -ffast-math: order and roundings kept */
    for (size_t i = 0; i < d; i++) {
        float t = a[i] - b[i];
        float p = t * t;
        res = res + p;
    }
    return (double)(float)sqrt((double)res);
}

/* vec_distance_cosine(float32, float32) */
ORC_API double orc_vec_distance_cosine_f32(const float *a, const float *b, size_t d) {
    float dot = 0.0f, aa = 0.0f, bb = 0.0f;
    for (size_t i = 0; i < d; i++) {
        float p0 = a[i] * b[i];
        float p1 = a[i] * a[i];
        float p2 = b[i] * b[i];
        dot = dot + p0;
        aa = aa + p1;
        bb = bb + p2;
    }
    return (double)(float)(1.0 - ((double)dot / (sqrt((double)aa) * sqrt((double)bb))));
}

/* vec_distance_L2(int8, int8): i8 loads promoted to int, difference converted to
 * f32, squares accumulated in f32. */
ORC_API double orc_vec_distance_l2_i8(const int8_t *a, const int8_t *b, size_t d) {
    float res = 0.0f;
    for (size_t i = 0; i < d; i++) {
        float t = (float)((int)a[i] - (int)b[i]);
        float p = t * t;
        res = res + p;
    }
    return (double)(float)sqrt((double)res);
}

/* vec_distance_cosine(int8, int8) */
ORC_API double orc_vec_distance_cosine_i8(const int8_t *a, const int8_t *b, size_t d) {
    float dot = 0.0f, aa = 0.0f, bb = 0.0f;
    for (size_t i = 0; i < d; i++) {
        float p0 = (float)((int)a[i] * (int)b[i]);
        float p1 = (float)((int)a[i] * (int)a[i]);
        float p2 = (float)((int)b[i] * (int)b[i]);
        dot = dot + p0;
        aa = aa + p1;
        bb = bb + p2;
    }
    return (double)(float)(1.0 - ((double)dot / (sqrt((double)aa) * sqrt((double)bb))));
}

/* The closed form the int8 kernels reduce to while every partial sum < 2^24:
 * a pure function of the exact integer sums.  Used by tests to prove the
 * order-independence claim, and by the HIP rerank kernel's specification. */
ORC_API double orc_i8_l2_from_sums(int64_t sumsq) {
    return (double)(float)sqrt((double)(float)sumsq);
}
ORC_API double orc_i8_cosine_from_sums(int64_t dot, int64_t aa, int64_t bb) {
    return (double)(float)(1.0 - ((double)(float)dot / (sqrt((double)(float)aa) * sqrt((double)(float)bb))));
}

/* -------------------------------------------------------------- scans */

/* score_all: the `d` column of the reference's dist_{cte}
 * (filters/exact.rs:106-134): one distance per corpus row, in row order.
 * corpus: [n][dim] of dtype; query: f32[dim] for F32/F16 corpora (an F16 corpus
 * is scored as its values widened to f32, SURVEY §7 "fp16 parity definition"),
 * i8[dim] for I8.  threads <= 1 => single thread (the reference's model: one
 * SQLite connection = one thread, db/connection.rs:320-357). */
ORC_API void orc_score_all(int dtype, int metric, const void *corpus, size_t n, size_t dim,
                           const void *query, float *out, int threads) {
    (void)threads;
#ifdef _OPENMP
    int nt = threads > 1 ? threads : 1;
#pragma omp parallel num_threads(nt)
#endif
    {
        float *tmp = (dtype == ORC_F16) ? (float *)malloc(dim * sizeof(float)) : NULL;
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (long long r = 0; r < (long long)n; r++) {
            double dist;
            if (dtype == ORC_I8) {
                const int8_t *row = (const int8_t *)corpus + (size_t)r * dim;
                dist = metric == ORC_L2 ? orc_vec_distance_l2_i8(row, (const int8_t *)query, dim)
                                        : orc_vec_distance_cosine_i8(row, (const int8_t *)query, dim);
            } else {
                const float *row;
                if (dtype == ORC_F16) {
                    const uint16_t *h = (const uint16_t *)corpus + (size_t)r * dim;
                    for (size_t i = 0; i < dim; i++) tmp[i] = orc_f16_to_f32(h[i]);
                    row = tmp;
                } else {
                    row = (const float *)corpus + (size_t)r * dim;
                }
                dist = metric == ORC_L2 ? orc_vec_distance_l2_f32(row, (const float *)query, dim)
                                        : orc_vec_distance_cosine_f32(row, (const float *)query, dim);
            }
            out[r] = (float)dist; /* exact: dist is an f32 value widened */
        }
        free(tmp);
    }
}

typedef struct {
    float d;
    int64_t id;
} orc_pair;

/* (isnan, distance, id) ascending — NULLS LAST, builder.rs:1201-1205 + the
 * build's id tie-break. */
static int pair_less(const orc_pair *x, const orc_pair *y) {
    int nx = isnan(x->d), ny = isnan(y->d);
    if (nx != ny) return nx < ny;
    if (!nx && x->d != y->d) return x->d < y->d;
    return x->id < y->id;
}
static int pair_cmp(const void *p, const void *q) {
    const orc_pair *x = (const orc_pair *)p, *y = (const orc_pair *)q;
    if (pair_less(x, y)) return -1;
    if (pair_less(y, x)) return 1;
    return 0;
}

/* max-heap of the current best k under pair_less */
static void heap_sift_down(orc_pair *h, size_t n, size_t i) {
    for (;;) {
        size_t l = 2 * i + 1, r = l + 1, m = i;
        if (l < n && pair_less(&h[m], &h[l])) m = l;
        if (r < n && pair_less(&h[m], &h[r])) m = r;
        if (m == i) return;
        orc_pair t = h[i];
        h[i] = h[m];
        h[m] = t;
        i = m;
    }
}

/* First k rows of the reference's full sort (ORDER BY order_rank ASC NULLS LAST
 * ... LIMIT k, builder.rs:578-582) under the shared tie-break.  ids == NULL
 * means id = row index.  Returns the number of rows written (min(k, n)). */
ORC_API uint32_t orc_topk(const float *dist, const int64_t *ids, size_t n, uint32_t k,
                          int64_t *out_ids, float *out_dist) {
    if (k == 0 || n == 0) return 0;
    size_t cap = k < n ? k : n;
    orc_pair *h = (orc_pair *)malloc(cap * sizeof(orc_pair));
    size_t cnt = 0;
    for (size_t i = 0; i < n; i++) {
        orc_pair p = {dist[i], ids ? ids[i] : (int64_t)i};
        if (cnt < cap) {
            h[cnt++] = p;
            if (cnt == cap)
                for (size_t j = cap / 2; j-- > 0;) heap_sift_down(h, cap, j);
        } else if (pair_less(&p, &h[0])) {
            h[0] = p;
            heap_sift_down(h, cap, 0);
        }
    }
    qsort(h, cnt, sizeof(orc_pair), pair_cmp);
    for (size_t i = 0; i < cnt; i++) {
        out_ids[i] = h[i].id;
        out_dist[i] = h[i].d;
    }
    free(h);
    return (uint32_t)cnt;
}

/* One query end to end: score every row, keep page 1 of size k. */
ORC_API uint32_t orc_search(int dtype, int metric, const void *corpus, size_t n, size_t dim,
                            const void *query, const int64_t *ids, uint32_t k, int64_t *out_ids,
                            float *out_dist, int threads) {
    float *d = (float *)malloc((n ? n : 1) * sizeof(float));
    orc_score_all(dtype, metric, corpus, n, dim, query, d, threads);
    uint32_t c = orc_topk(d, ids, n, k, out_ids, out_dist);
    free(d);
    return c;
}

/* ---------------------------------------------------------- aggregation */

/* SQLite's SUM()/AVG() accumulate doubles with Kahan-Babuska-Neumaier
 * compensation (sqlite3 func.c kahanBabuskaNeumaierStep, 3.44+; the reference
 * bundles libsqlite3-sys 0.37). */
typedef struct {
    double s, c;
} kbn;
static void kbn_step(kbn *p, double r) {
    double s = p->s, t = s + r;
    if (fabs(s) > fabs(r))
        p->c += (s - t) + r;
    else
        p->c += (r - t) + s;
    p->s = t;
}
static double kbn_value(const kbn *p) { return p->s + p->c; }

/* filters/exact.rs:67-80 rank_aggregate over rows grouped by group id
 * (builder.rs:829-835 GROUP BY file_id).  `dist` holds per-row distances (NaN =
 * SQL NULL, ignored by MIN/MAX/AVG/SUM); `w` (optional) the per-row confidence
 * weight of exact.rs:37-61, in which case the aggregate is SUM(d*w)/SUM(w) and
 * `agg` is ignored.  group[] must be non-decreasing (rows clustered by group).
 * Writes one (group id, aggregate f64) per distinct group, in group order;
 * returns the number of groups.  A group whose rows are all NULL yields NaN. */
ORC_API size_t orc_aggregate(const float *dist, const float *w, const int64_t *group, size_t n,
                             int agg, int64_t *out_group, double *out_val) {
    size_t g = 0, i = 0;
    while (i < n) {
        size_t j = i;
        kbn sum = {0, 0}, wsum = {0, 0};
        double mn = INFINITY, mx = -INFINITY;
        size_t cnt = 0;
        for (; j < n && group[j] == group[i]; j++) {
            if (w) kbn_step(&wsum, (double)w[j]); /* SUM(w) runs over every row; only d*w is NULL for a NULL d */
            if (isnan(dist[j])) continue;
            double d = (double)dist[j];
            if (w) {
                kbn_step(&sum, d * (double)w[j]);
            } else {
                kbn_step(&sum, d);
            }
            if (d < mn) mn = d;
            if (d > mx) mx = d;
            cnt++;
        }
        double v;
        if (cnt == 0)
            v = NAN;
        else if (w)
            v = kbn_value(&sum) / kbn_value(&wsum);
        else if (agg == ORC_AGG_MIN)
            v = mn;
        else if (agg == ORC_AGG_MAX)
            v = mx;
        else
            v = kbn_value(&sum) / (double)cnt;
        out_group[g] = group[i];
        out_val[g] = v;
        g++;
        i = j;
    }
    return g;
}

/* The similar_to self-join aggregate (filters/item_similarity.rs:432-581): dist is
 * [n][m] (row-major: other row o, target vector i) = vec_distance(main_i, other_o);
 * rows flagged in `exclude` (the target item's own rows, `other.sha256 != target`) are
 * skipped; per group the aggregate runs over every remaining (o, i) pair, o ascending then
 * i ascending.  group[] non-decreasing.  Returns the number of groups. */
ORC_API size_t orc_aggregate_fanout(const float *dist, size_t m, const uint8_t *exclude, const int64_t *group,
                                    size_t n, int agg, int64_t *out_group, double *out_val) {
    size_t g = 0, i = 0;
    while (i < n) {
        size_t j = i;
        kbn sum = {0, 0};
        double mn = INFINITY, mx = -INFINITY;
        size_t cnt = 0, joined = 0;
        for (; j < n && group[j] == group[i]; j++) {
            if (exclude && exclude[j]) continue;
            for (size_t t = 0; t < m; t++) {
                joined++;
                float df = dist[j * m + t];
                if (isnan(df)) continue;
                double d = (double)df;
                kbn_step(&sum, d);
                if (d < mn) mn = d;
                if (d > mx) mx = d;
                cnt++;
            }
        }
        if (joined == 0) { /* INNER JOIN + WHERE other.sha256 != target: no pair, no output row (item_similarity.rs:445-468) */
            i = j;
            continue;
        }
        double v = cnt == 0 ? NAN : agg == ORC_AGG_MIN ? mn : agg == ORC_AGG_MAX ? mx : kbn_value(&sum) / (double)cnt;
        out_group[g] = group[i];
        out_val[g] = v;
        g++;
        i = j;
    }
    return g;
}

/* Confidence-weighted similar_to (filters/item_similarity.rs:503-581): per (main vector t, other row o)
 *   w = pow(coalesce(conf_t,1) * coalesce(conf_o,1), cw)        [factor dropped when cw == 0]
 *     * pow(coalesce(lang_o,1) * coalesce(lang_t,1), lw)        [factor dropped when lw == 0]
 * and the rank is SUM(d*w) / SUM(w) over the group's fan-out (SUM(w) over every joined pair, d*w
 * skipped where d is NULL).  conf/lang: per row, NaN = SQL NULL.  target[t] = row index of main vector t. */
static double coalesce1(double v) { return isnan(v) ? 1.0 : v; }
ORC_API size_t orc_aggregate_fanout_weighted(const float *dist, size_t m, const uint8_t *exclude, const int64_t *group, size_t n,
                                             const size_t *target, const double *conf, const double *lang, double cw, double lw,
                                             int64_t *out_group, double *out_val) {
    size_t g = 0, i = 0;
    while (i < n) {
        size_t j = i;
        kbn sum = {0, 0}, wsum = {0, 0};
        size_t cnt = 0, joined = 0;
        for (; j < n && group[j] == group[i]; j++) {
            if (exclude && exclude[j]) continue;
            for (size_t t = 0; t < m; t++) {
                joined++;
                double w = 1.0;
                if (cw != 0.0 && lw != 0.0)
                    w = pow(coalesce1(conf[target[t]]) * coalesce1(conf[j]), cw) * pow(coalesce1(lang[j]) * coalesce1(lang[target[t]]), lw);
                else if (cw != 0.0)
                    w = pow(coalesce1(conf[target[t]]) * coalesce1(conf[j]), cw);
                else if (lw != 0.0)
                    w = pow(coalesce1(lang[j]) * coalesce1(lang[target[t]]), lw);
                kbn_step(&wsum, w);
                float df = dist[j * m + t];
                if (isnan(df)) continue;
                kbn_step(&sum, (double)df * w);
                cnt++;
            }
        }
        if (joined == 0) {
            i = j;
            continue;
        }
        out_group[g] = group[i];
        out_val[g] = cnt == 0 ? NAN : kbn_value(&sum) / kbn_value(&wsum);
        g++;
        i = j;
    }
    return g;
}

/* similar_to with every option of the reference's filter (item_similarity.rs:432-581): the weighted form above
 * plus the CLIP cross-modal gates (:473-489) — kind[row] 0 = 'clip', 1 = 'text-embedding'; with skip_i2i pairs of
 * two clip rows are not part of the join, with skip_t2t pairs of two text rows — and, when both exponents are 0,
 * the plain MIN/MAX/AVG aggregate. */
ORC_API size_t orc_aggregate_fanout_ex(const float *dist, size_t m, const uint8_t *exclude, const int64_t *group, size_t n,
                                       const size_t *target, const double *conf, const double *lang, double cw, double lw,
                                       const uint8_t *kind, int skip_i2i, int skip_t2t, int agg, int64_t *out_group,
                                       double *out_val) {
    const int weighted = cw != 0.0 || lw != 0.0;
    size_t g = 0, i = 0;
    while (i < n) {
        size_t j = i;
        kbn sum = {0, 0}, wsum = {0, 0};
        double mn = INFINITY, mx = -INFINITY;
        size_t cnt = 0, joined = 0;
        for (; j < n && group[j] == group[i]; j++) {
            if (exclude && exclude[j]) continue;
            for (size_t t = 0; t < m; t++) {
                if (kind) {
                    const uint8_t km = kind[target[t]], ko = kind[j];
                    if ((skip_i2i && km == 0 && ko == 0) || (skip_t2t && km == 1 && ko == 1)) continue;
                }
                joined++;
                double w = 1.0;
                if (cw != 0.0 && lw != 0.0)
                    w = pow(coalesce1(conf[target[t]]) * coalesce1(conf[j]), cw) * pow(coalesce1(lang[j]) * coalesce1(lang[target[t]]), lw);
                else if (cw != 0.0)
                    w = pow(coalesce1(conf[target[t]]) * coalesce1(conf[j]), cw);
                else if (lw != 0.0)
                    w = pow(coalesce1(lang[j]) * coalesce1(lang[target[t]]), lw);
                if (weighted) kbn_step(&wsum, w);
                float df = dist[j * m + t];
                if (isnan(df)) continue;
                double d = (double)df;
                kbn_step(&sum, weighted ? d * w : d);
                if (d < mn) mn = d;
                if (d > mx) mx = d;
                cnt++;
            }
        }
        if (joined == 0) { /* every pair gated away or excluded: the group is not in the join at all */
            i = j;
            continue;
        }
        double v;
        if (cnt == 0)
            v = NAN;
        else if (weighted)
            v = kbn_value(&sum) / kbn_value(&wsum);
        else
            v = agg == ORC_AGG_MIN ? mn : agg == ORC_AGG_MAX ? mx : kbn_value(&sum) / (double)cnt;
        out_group[g] = group[i];
        out_val[g] = v;
        g++;
        i = j;
    }
    return g;
}

/* ------------------------------------------------------------ rank / RRF */

typedef struct {
    double v;
    int64_t id;
} orc_dpair;

/* builder.rs:757-771: `row_number() OVER (ORDER BY agg <row_n_direction>)` — 1-based ranks.  The window has no
 * NULLS clause, so SQLite's default applies: NULL sorts as the smallest value — FIRST ascending, LAST
 * descending.  Ties (SQLite leaves them to the scan order) are broken by id ascending: the build's
 * deterministic tie-break.  rank_out[i] is the rank of element i. */
typedef struct {
    orc_dpair k; /* must stay first: the comparators read it */
    size_t pos;
} orc_rank_rec;
static int rank_cmp_asc(const void *p, const void *q) {
    const orc_dpair *x = (const orc_dpair *)p, *y = (const orc_dpair *)q;
    int nx = isnan(x->v), ny = isnan(y->v);
    if (nx != ny) return ny - nx; /* NULL first */
    if (!nx && x->v != y->v) return x->v < y->v ? -1 : 1;
    return x->id < y->id ? -1 : (x->id > y->id ? 1 : 0);
}
static int rank_cmp_desc(const void *p, const void *q) {
    const orc_dpair *x = (const orc_dpair *)p, *y = (const orc_dpair *)q;
    int nx = isnan(x->v), ny = isnan(y->v);
    if (nx != ny) return nx - ny; /* NULL last */
    if (!nx && x->v != y->v) return x->v > y->v ? -1 : 1;
    return x->id < y->id ? -1 : (x->id > y->id ? 1 : 0);
}
ORC_API void orc_row_number_dir(const double *val, const int64_t *ids, size_t n, int descending, int64_t *rank_out) {
    orc_rank_rec *r = (orc_rank_rec *)malloc((n ? n : 1) * sizeof(orc_rank_rec));
    for (size_t i = 0; i < n; i++) {
        r[i].k.v = val[i];
        r[i].k.id = ids ? ids[i] : (int64_t)i;
        r[i].pos = i;
    }
    qsort(r, n, sizeof(orc_rank_rec), descending ? rank_cmp_desc : rank_cmp_asc);
    for (size_t i = 0; i < n; i++) rank_out[r[i].pos] = (int64_t)(i + 1);
    free(r);
}
ORC_API void orc_row_number(const double *val, const int64_t *ids, size_t n, int64_t *rank_out) {
    orc_row_number_dir(val, ids, n, 0, rank_out);
}

/* builder.rs:17-18 */
#define VERY_LARGE_NUMBER 9223372036854775805LL

/* builder.rs:1284-1301: fused = sum_i (1.0 / (k_i + coalesce(rank_i, BIG))) * weight_i,
 * left to right in f64.  rank < 0 encodes SQL NULL (branch did not return the
 * file).  `k_i + rank` is SQLite integer addition, which falls back to floating
 * point on i64 overflow. */
ORC_API double orc_rrf_score(const int64_t *ranks, const int32_t *ks, const double *weights, size_t nb) {
    double total = 0.0;
    for (size_t i = 0; i < nb; i++) {
        int64_t rank = ranks[i] < 0 ? VERY_LARGE_NUMBER : ranks[i];
        int64_t denom_i;
        double denom;
        if (__builtin_add_overflow((int64_t)ks[i], rank, &denom_i))
            denom = (double)ks[i] + (double)rank;
        else
            denom = (double)denom_i;
        double term = (1.0 / denom) * weights[i];
        total = (i == 0) ? term : total + term;
    }
    return total;
}

/* builder.rs:1303-1317: same-priority order filters without rrf: min(coalesce(rank_i, BIG), ...) ascending,
 * max(coalesce(rank_i, -BIG), ...) descending; rank < 0 encodes SQL NULL. */
ORC_API int64_t orc_coalesce_rank(const int64_t *ranks, size_t nb, int descending) {
    int64_t best = 0;
    for (size_t i = 0; i < nb; i++) {
        int64_t v = ranks[i] < 0 ? (descending ? -VERY_LARGE_NUMBER : VERY_LARGE_NUMBER) : ranks[i];
        if (i == 0 || (descending ? v > best : v < best)) best = v;
    }
    return best;
}

/* builder.rs:781-815: WHERE order_rank > gt AND order_rank < lt (each optional); NaN = NULL fails a comparison */
ORC_API int orc_sort_bounds_keep(double order_rank, int have_gt, double gt, int have_lt, double lt) {
    if (have_gt && !(order_rank > gt)) return 0;
    if (have_lt && !(order_rank < lt)) return 0;
    return 1;
}

/* -------------------------------------------------- synthetic generator */

/* SURVEY §8d synthetic inputs: unit-normalised standard-normal rows, the
 * reference's own convention (tools/pql-equivalence/run_suite.py:532-542).
 * The bit stream is the build's own counter-based generator so that host and
 * device produce IDENTICAL bytes with integer-only arithmetic up to the final
 * IEEE operations: an Irwin-Hall(12) sum of 16-bit uniforms from splitmix64
 * keyed by (seed, row, col), centred and scaled (exact in f32), then the row
 * is divided by its f64-accumulated norm rounded to f32. */
static uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}
static int32_t synth_raw(uint64_t seed, uint64_t row, uint32_t col) {
    uint64_t key = seed * 0xD1342543DE82EF95ULL + row * 0x9E3779B97F4A7C15ULL + (uint64_t)col * 0xC2B2AE3D27D4EB4FULL;
    uint64_t a = splitmix64(key), b = splitmix64(key ^ 0xA5A5A5A5A5A5A5A5ULL), c = splitmix64(key + 0x1234567ULL);
    uint32_t s = 0;
    for (int i = 0; i < 4; i++) s += (uint32_t)((a >> (16 * i)) & 0xffff);
    for (int i = 0; i < 4; i++) s += (uint32_t)((b >> (16 * i)) & 0xffff);
    for (int i = 0; i < 4; i++) s += (uint32_t)((c >> (16 * i)) & 0xffff);
    /* sum of 12 U16: mean 12*32767.5 = 393210, variance 12*(65536^2-1)/12 ~ 65536^2 */
    return (int32_t)s - 393210;
}
ORC_API float orc_synth_gauss(uint64_t seed, uint64_t row, uint32_t col) {
    return (float)synth_raw(seed, row, col) * (1.0f / 65536.0f); /* exact: |raw| < 2^19 */
}
/* The squared norm is an exact integer sum (|raw|^2 < 2^38, dim <= 2^14 keeps the
 * sum < 2^53), so it is order independent and the device may reduce it in
 * parallel and still produce identical bytes. */
ORC_API void orc_synth_rows(uint64_t seed, uint64_t row0, size_t n, uint32_t dim, float *out) {
    for (size_t r = 0; r < n; r++) {
        int64_t ss = 0;
        float *o = out + r * dim;
        for (uint32_t c = 0; c < dim; c++) {
            int32_t v = synth_raw(seed, row0 + r, c);
            o[c] = (float)v * (1.0f / 65536.0f);
            ss += (int64_t)v * (int64_t)v;
        }
        float nrm = (float)sqrt((double)ss * (1.0 / 4294967296.0));
        if (!(nrm > 0.0f)) nrm = 1.0f;
        for (uint32_t c = 0; c < dim; c++) o[c] = o[c] / nrm;
    }
}

/* The clustered generator (csrc/pvs_kernels_util.hip: k_synth_clustered — see there for what the distribution models): the same
 * integer arithmetic, the same final IEEE operations, identical bytes. */
static uint64_t synth_row_hash(uint64_t seed, uint64_t row) { return splitmix64(seed * 0xA24BAED4963EE407ULL + row * 0x9FB21C651E98DF25ULL + 0x51ED27ULL); }
static int synth_is_dup(uint64_t seed, uint64_t row) { return row >= 17 && synth_row_hash(seed, row) % 100 == 0; }
static uint64_t synth_source_row(uint64_t seed, uint64_t row) {
    if (!synth_is_dup(seed, row)) return row;
    uint64_t src = row - 1 - ((synth_row_hash(seed, row) >> 8) & 15);
    return synth_is_dup(seed, src) ? row : src;
}
static uint32_t synth_cluster(uint64_t seed, uint64_t row) {
    uint64_t u = synth_row_hash(seed ^ 0xC1u, row) >> 32;
    uint64_t u3 = (((u * u) >> 32) * u) >> 32;
    return (uint32_t)((u3 * 2000u) >> 32);
}
static int32_t synth_clustered_raw(uint64_t seed, uint64_t row, uint32_t col) {
    uint32_t cl = synth_cluster(seed, row);
    int sh = (int)(splitmix64(seed + 0x77u + (uint64_t)cl * 0x100000001B3ULL + (uint64_t)(col >> 3)) & 3);
    int32_t centre = synth_raw(seed ^ 0xCE47E5ULL, cl, col);
    uint64_t block = row >> 3;
    int near = synth_row_hash(seed ^ 0xB10Cu, block) % 10 == 0;
    int32_t v = 2 * centre;
    if (near) {
        uint32_t cb = synth_cluster(seed, block << 3);
        int shb = (int)(splitmix64(seed + 0x77u + (uint64_t)cb * 0x100000001B3ULL + (uint64_t)(col >> 3)) & 3);
        v = 2 * synth_raw(seed ^ 0xCE47E5ULL, cb, col) + (synth_raw(seed ^ 0x5A5AULL, block, col) >> shb) + (synth_raw(seed, row, col) >> (shb + 4));
    } else {
        v += synth_raw(seed, row, col) >> sh;
    }
    return v;
}
ORC_API void orc_synth_rows_clustered(uint64_t seed, uint64_t row0, size_t n, uint32_t dim, float *out) {
    for (size_t r = 0; r < n; r++) {
        uint64_t src = synth_source_row(seed, row0 + r);
        int64_t ss = 0;
        float *o = out + r * dim;
        for (uint32_t c = 0; c < dim; c++) {
            int32_t v = synth_clustered_raw(seed, src, c);
            o[c] = (float)v * (1.0f / 65536.0f);
            ss += (int64_t)v * (int64_t)v;
        }
        float nrm = (float)sqrt((double)ss * (1.0 / 4294967296.0));
        if (!(nrm > 0.0f)) nrm = 1.0f;
        for (uint32_t c = 0; c < dim; c++) o[c] = o[c] / nrm;
    }
}
ORC_API uint32_t orc_synth_cluster_of(uint64_t seed, uint64_t row) { return synth_cluster(seed, synth_source_row(seed, row)); }

ORC_API int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
