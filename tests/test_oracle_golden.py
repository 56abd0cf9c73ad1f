"""Pins the CPU oracle against every known-answer test the reference holds for
the vector-similarity path.  Values are restated from the
reference's #[test] bodies; file:line under /root/reference/panoptikon/src/."""
import math

import os

import numpy as np
import pytest

import oracle as orc


def f32(*v):
    return np.array(v, np.float32)


# db/vector_quants.rs:3588-3626 int8_codec_rounds_ties_to_even_and_clamps
def test_codec_rounds_ties_to_even_and_clamps():
    codes = orc.quantize_int8(f32(0.5, 1.5, 2.5, -0.5, -1.5, -2.5, 2.4999, -2.4999), 1.0)
    assert codes.tolist() == [0, 2, 2, 0, -2, -2, 2, -2]
    scale = orc.scale_from_absmax(11.0)
    assert orc.quantize_int8(f32(11.0, -11.0, 1000.0, -1000.0), scale).tolist() == [127, -127, 127, -128]
    assert orc.quantize_int8(f32(1.0, 2.0, 3.0), 1.0).view(np.uint8).tolist() == [1, 2, 3]
    assert orc.scale_from_absmax(0.0) == 1.0
    assert orc.scale_from_absmax(float("nan")) == 1.0
    assert orc.scale_from_absmax(float("inf")) == 1.0
    assert orc.quantize_int8(f32(0.0, 0.0), 1.0).tolist() == [0, 0]
    scale = orc.scale_from_absmax(3.5)
    assert orc.artifact_scale(orc.scale_artifact(scale)) == scale
    assert orc.artifact_scale(b"") is None
    assert orc.artifact_scale(bytes(5)) is None
    assert orc.artifact_scale(np.float32(0).tobytes()) is None
    assert orc.artifact_scale(np.float32(-1).tobytes()) is None
    assert orc.artifact_scale(np.float32("nan").tobytes()) is None


def test_codec_nan_and_inf_components_follow_rust_as_cast():
    # Rust `as i8`: NaN -> 0, +-inf saturate (after clamp they are +-127/-128)
    codes = orc.quantize_int8(f32(float("nan"), float("inf"), float("-inf")), 0.5)
    assert codes.tolist() == [0, 127, -128]


# db/vector_quants.rs:1947-1949 vec8; :2194-2244 build_uses_absmax_scale_artifact
def vec8(a, b):
    return [a, b, 1.0, -1.0, 2.0, -2.0, 3.0, -3.0]


def test_absmax_scale_artifact_fixture():
    rows = np.array([vec8(1.0 + (i % 10), -2.0 - (i % 10)) for i in range(1024)], np.float32)
    scale = orc.compute_int8_scale(rows)
    assert abs(scale - 11.0 / 127.0) < 1e-6
    assert scale == float(np.float32(11.0) / np.float32(127.0))
    codes = orc.quantize_int8(rows, scale)
    # numpy restatement of the codec (independent of the C oracle)
    ref = np.clip(np.rint(rows / np.float32(scale)), -128, 127).astype(np.int8)
    assert np.array_equal(codes, ref)
    assert len({c.tobytes() for c in codes}) > 1
    assert orc.compute_int8_scale(np.zeros((0, 8), np.float32)) is None
    assert orc.blob_absmax(f32(1.0, float("nan"), -4.0)) == 4.0  # NaN ignored (:1474-1483)


# db/vector_quants.rs:3632-3687 sqlite_vec_int8_distances_match_a_rust_reference
KAT = [
    ([127, -128, 0, 3, -3, 64, -64, 1], [1, 2, 3, 4, 5, 6, 7, 8]),
    ([1, 1, 1, 1, 1, 1, 1, 1], [1, 1, 1, 1, 1, 1, 1, 1]),
    ([-5, 20, -33, 44, 0, 12, -7, 100], [100, -7, 12, 0, 44, -33, 20, -5]),
]
# SURVEY.md §8c computed expectations under the sequential-f32 model
KAT_EXPECT = [
    (203.23385620117188, 1.0652254819869995),
    (0.0, 2.220446049250313e-16),
    (177.2850799560547, 1.1518727540969849),
]


@pytest.mark.parametrize("case", range(3))
def test_int8_distance_kat(case):
    a = np.array(KAT[case][0], np.int8)
    b = np.array(KAT[case][1], np.int8)
    l2 = orc.vec_distance(orc.L2, a, b)
    cos = orc.vec_distance(orc.COSINE, a, b)
    af, bf = a.astype(np.float64), b.astype(np.float64)
    exp_l2 = math.sqrt(((af - bf) ** 2).sum())
    exp_cos = 1.0 - (af * bf).sum() / (math.sqrt((af * af).sum()) * math.sqrt((bf * bf).sum()))
    # the reference's own tolerances
    assert abs(l2 - exp_l2) <= 1e-4 * max(exp_l2, 1.0)
    assert abs(cos - exp_cos) <= 1e-4
    # the model's exact values (f32 result widened; case 2's cosine is the f64 residue
    # 1 - 8/(sqrt(8)*sqrt(8)) before narrowing, so compare after narrowing to f32)
    assert l2 == KAT_EXPECT[case][0]
    assert np.float32(cos) == np.float32(KAT_EXPECT[case][1])
    # closed forms over exact integer sums (order-independence below 2^24)
    ai, bi = a.astype(np.int64), b.astype(np.int64)
    assert l2 == orc.lib().orc_i8_l2_from_sums(int(((ai - bi) ** 2).sum()))
    assert cos == orc.lib().orc_i8_cosine_from_sums(int((ai * bi).sum()), int((ai * ai).sum()), int((bi * bi).sum()))


def test_int8_closed_form_matches_sequential_on_random_codes():
    rng = np.random.default_rng(7)
    for d in (8, 512, 768, 1024):
        a = rng.integers(-60, 61, d).astype(np.int8)
        b = rng.integers(-60, 61, d).astype(np.int8)
        ai, bi = a.astype(np.int64), b.astype(np.int64)
        assert ((ai - bi) ** 2).sum() < 2**24
        assert orc.vec_distance(orc.L2, a, b) == orc.lib().orc_i8_l2_from_sums(int(((ai - bi) ** 2).sum()))
        assert orc.vec_distance(orc.COSINE, a, b) == orc.lib().orc_i8_cosine_from_sums(
            int((ai * bi).sum()), int((ai * ai).sum()), int((bi * bi).sum()))


def test_f32_distances_close_to_f64_formula():
    rng = np.random.default_rng(3)
    a = rng.standard_normal(768).astype(np.float32)
    b = rng.standard_normal(768).astype(np.float32)
    a64, b64 = a.astype(np.float64), b.astype(np.float64)
    l2 = math.sqrt(((a64 - b64) ** 2).sum())
    cos = 1.0 - (a64 * b64).sum() / (np.linalg.norm(a64) * np.linalg.norm(b64))
    # tools/pql-equivalence/run_suite.py:74-75 tolerances
    assert abs(orc.vec_distance(orc.L2, a, b) - l2) <= 1e-4 * l2 + 1e-6
    assert abs(orc.vec_distance(orc.COSINE, a, b) - cos) <= 1e-4 * abs(cos) + 1e-6
    # sequential-f32 model restated in numpy (independent of the C code)
    acc = np.float32(0)
    for x, y in zip(a, b):
        t = np.float32(x - y)
        acc = np.float32(acc + np.float32(t * t))
    assert orc.vec_distance(orc.L2, a, b) == float(np.float32(math.sqrt(float(acc))))


def test_f16_conversions_match_numpy():
    rng = np.random.default_rng(11)
    x = np.concatenate([
        rng.standard_normal(4000).astype(np.float32) * np.float32(0.05),
        f32(0.0, -0.0, 65504.0, 65519.9, 65520.0, 1e-8, 6e-8, 5.96e-8, 2.98e-8, 2.9802322e-8, 6.1e-5, 6.0e-5,
            float("inf"), float("-inf")),
        rng.standard_normal(1000).astype(np.float32) * np.float32(1e-6),
    ])
    bits = orc.f32_to_f16_bits(x)
    assert np.array_equal(bits, x.astype(np.float16).view(np.uint16))
    allbits = np.arange(0, 65536, dtype=np.uint32).astype(np.uint16)
    wid = orc.f16_bits_to_f32(allbits)
    ref = allbits.view(np.float16).astype(np.float32)
    nan = np.isnan(ref)  # NaN payload bits do not survive the ctypes float round trip
    assert np.array_equal(np.isnan(wid), nan)
    assert np.array_equal(wid[~nan].view(np.uint32), ref[~nan].view(np.uint32))
    # the reference's NPY-ingestion widening halves subnormals (reference quirk Q1,
    # pql/embedding_utils.rs:323-350) and is IEEE elsewhere
    quirk = orc.npy_f16_bits_to_f32(allbits)
    sub = ((allbits & 0x7C00) == 0) & ((allbits & 0x3FF) != 0)
    assert np.array_equal(quirk[~sub & ~nan].view(np.uint32), ref[~sub & ~nan].view(np.uint32))
    assert np.array_equal(quirk[sub], ref[sub] * np.float32(0.5))


# db/vector_quants.rs:3254-3276 disagreeing_vectors, QUERY_VECTOR = [1.0; 8]
def disagreeing_vectors():
    vs = []
    for idx in range(6):
        v = [0.02] * 8
        v[idx] = 11.0
        v[(idx + 1) % 8] = 0.6 + 0.5 * idx
        vs.append(v)
    for idx in range(6):
        v = [4.0] * 8
        for flip in range(idx % 3 + 1):
            v[7 - flip] = -0.5 - 0.2 * idx
        vs.append(v)
    return np.array(vs, np.float32)


# :3324-3382 quant_query_matches_exact_and_is_deterministic, at the scorer level
def test_quant_order_matches_exact_on_disagreeing_fixture():
    vecs = disagreeing_vectors()
    assert vecs.dtype == np.float32
    query = np.ones(8, np.float32)
    scale = orc.compute_int8_scale(vecs)
    assert scale == float(np.float32(11.0) / np.float32(127.0))
    codes = orc.quantize_int8(vecs, scale)
    qcodes = orc.quantize_int8(query, scale)
    ids_exact, _ = orc.search(orc.F32, orc.COSINE, vecs, query, 100)
    ids_quant, _ = orc.search(orc.I8, orc.COSINE, codes, qcodes, 100)
    assert ids_exact.shape == (1, 12)
    assert ids_exact.tolist() == ids_quant.tolist()
    # :3386-3415 page walk == single shot (pages of 4 are prefixes of the full sort)
    for k in (1, 4, 8, 12, 10_000):
        ids_k, _ = orc.search(orc.I8, orc.COSINE, codes, qcodes, k)
        assert ids_k[0].tolist() == ids_quant[0][: min(k, 12)].tolist()


def test_topk_ties_and_nulls_last():
    d = f32(0.5, float("nan"), 0.25, 0.5, 0.25, float("nan"))
    ids, dist = orc.topk(d, 10)
    assert ids.tolist() == [2, 4, 0, 3, 1, 5]
    assert np.isnan(dist[-2:]).all()
    ids, _ = orc.topk(d, 3, ids=[60, 50, 40, 30, 20, 10])
    assert ids.tolist() == [20, 40, 30]
    assert orc.topk(np.zeros(0, np.float32), 5)[0].size == 0


# filters/exact.rs:67-80 rank_aggregate
def test_aggregate_min_max_avg_weighted():
    dist = f32(0.3, 0.1, 0.2, 0.9, float("nan"), 0.4)
    grp = [7, 7, 7, 9, 9, 11]
    g, v = orc.aggregate(dist, grp, orc.AGG_MIN)
    assert g.tolist() == [7, 9, 11]
    assert v.tolist() == [float(np.float32(0.1)), float(np.float32(0.9)), float(np.float32(0.4))]
    _, v = orc.aggregate(dist, grp, orc.AGG_MAX)
    assert v.tolist() == [float(np.float32(0.3)), float(np.float32(0.9)), float(np.float32(0.4))]
    _, v = orc.aggregate(dist, grp, orc.AGG_AVG)
    d64 = dist.astype(np.float64)
    assert v[0] == pytest.approx((d64[0] + d64[1] + d64[2]) / 3, rel=1e-15)
    assert v[1] == d64[3]  # NULL ignored by AVG
    w = f32(1.0, 3.0, 0.5, 2.0, 2.0, 1.0)
    _, v = orc.aggregate(dist, grp, orc.AGG_MIN, w=w)
    w64 = w.astype(np.float64)
    assert v[0] == pytest.approx((d64[:3] * w64[:3]).sum() / w64[:3].sum(), rel=1e-15)
    _, v = orc.aggregate(f32(float("nan")), [1], orc.AGG_AVG)
    assert math.isnan(v[0])


# builder.rs:757-771 row_number; builder.rs:1284-1301 RRF; weights of quant_ab.rs:233-246
def test_row_number_and_rrf():
    # the window has no NULLS clause (builder.rs:757-771): SQLite sorts NULL first ascending, last descending
    ranks = orc.row_number([0.5, 0.1, float("nan"), 0.1], ids=[4, 9, 1, 3])
    assert ranks.tolist() == [4, 3, 1, 2]
    assert orc.row_number([0.5, 0.1, float("nan"), 0.1], ids=[4, 9, 1, 3], descending=True).tolist() == [1, 3, 4, 2]
    # pinned against SQLite itself (the stdlib module IS the engine whose window semantics the reference relies on)
    import sqlite3

    rng = np.random.default_rng(12)
    vals = np.round(rng.random(300), 2)  # many ties
    vals[rng.integers(0, 300, 25)] = np.nan
    ids = rng.permutation(300).astype(np.int64)
    conn = sqlite3.connect(":memory:")
    conn.execute("CREATE TABLE t (pos INTEGER, id INTEGER, v REAL)")
    conn.executemany("INSERT INTO t VALUES (?, ?, ?)", [(i, int(ids[i]), None if np.isnan(vals[i]) else float(vals[i])) for i in range(300)])
    for desc in (False, True):
        direction = "DESC" if desc else "ASC"
        sql = conn.execute(f"SELECT pos, row_number() OVER (ORDER BY v {direction}, id ASC) FROM t ORDER BY pos").fetchall()
        assert orc.row_number(vals, ids, descending=desc).tolist() == [r for _, r in sql]
    ks, ws = [5, 5, 10], [1.0, 1.0, 0.7]
    got = orc.rrf_score([1, 3, -1], ks, ws)
    big = 9223372036854775805
    exp = (1.0 / (5 + 1)) * 1.0 + (1.0 / (5 + 3)) * 1.0 + (1.0 / (float(10) + float(big))) * 0.7
    assert got == exp
    # k=1 does not overflow i64: integer addition then conversion
    assert orc.rrf_score([-1], [1], [1.0]) == 1.0 / float(big + 1)
    # default Rrf{k=1, weight=1.0} (pql/model.rs:129-133)
    assert orc.rrf_score([1], [1], [1.0]) == 0.5


def test_rrf_expression_and_aggregate_null_semantics_pinned_against_sqlite():
    """The SQL the reference emits around the distance column, evaluated by SQLite itself (stdlib module) on
    small inputs: the RRF term `1.0/(k + coalesce(rank, 9223372036854775805)) * weight` incl. the integer
    overflow of k + BIG (builder.rs:17-18,1284-1301), and MIN/MAX/AVG/SUM(d*w)/SUM(w) with NULL distances
    (exact.rs:67-80).  Values are dyadic, so summation order and compensation do not matter."""
    import sqlite3

    conn = sqlite3.connect(":memory:")
    for ranks, ks, ws in (([1, 3, None], [5, 5, 10], [1.0, 1.0, 0.7]), ([None], [1], [1.0]), ([None, 7], [3, 60], [0.25, 2.0]),
                          ([2, None], [1, 2], [1.0, 1.0])):
        expr = " + ".join(f"1.0 / ({k} + coalesce(?, 9223372036854775805)) * {w!r}" for k, w in zip(ks, ws))
        sql = conn.execute(f"SELECT {expr}", ranks).fetchone()[0]
        assert orc.rrf_score([-1 if r is None else r for r in ranks], ks, ws) == sql, (ranks, ks, ws)
    conn.execute("CREATE TABLE d (g INTEGER, d REAL, w REAL)")
    rows = [(1, 0.5, 2.0), (1, None, 4.0), (1, 0.25, 0.5), (2, None, 1.0), (2, None, 3.0), (3, 1.5, 1.0), (3, 0.75, 1.0), (3, 0.125, 2.0)]
    conn.executemany("INSERT INTO d VALUES (?, ?, ?)", rows)
    dist = np.array([np.nan if r[1] is None else r[1] for r in rows], np.float32)
    grp = np.array([r[0] for r in rows], np.int64)
    w = np.array([r[2] for r in rows], np.float32)
    for agg, fn in ((orc.AGG_MIN, "MIN(d)"), (orc.AGG_MAX, "MAX(d)"), (orc.AGG_AVG, "AVG(d)")):
        sql = conn.execute(f"SELECT g, {fn} FROM d GROUP BY g ORDER BY g").fetchall()
        g, v = orc.aggregate(dist, grp, agg)
        assert g.tolist() == [r[0] for r in sql]
        assert [None if np.isnan(x) else float(x) for x in v] == [r[1] for r in sql], fn
    sql = conn.execute("SELECT g, SUM(d * w) / SUM(w) FROM d GROUP BY g ORDER BY g").fetchall()
    g, v = orc.aggregate(dist, grp, orc.AGG_AVG, w=w)
    assert [None if np.isnan(x) else float(x) for x in v] == [r[1] for r in sql]


def test_coalesce_arm_and_sort_bounds_pinned_against_sqlite():
    """The non-RRF arm of build_coalesced_expr (builder.rs:1303-1317) and apply_sort_bounds (:781-815), evaluated by
    SQLite itself on the expressions the builder renders, against the oracle and the C ABI's host functions."""
    import sqlite3

    import panoptikon_amd as pvs

    BIG = 9223372036854775805
    rng = np.random.default_rng(8)
    n = 400
    ranks = rng.integers(1, 5000, (3, n)).astype(np.int64)
    ranks[rng.random((3, n)) < 0.3] = -1  # NULL: the filter did not return the row
    ranks[:, 0] = -1                      # a row no filter returned
    conn = sqlite3.connect(":memory:")
    conn.execute("CREATE TABLE t (i INTEGER PRIMARY KEY, a, b, c)")
    conn.executemany("INSERT INTO t VALUES (?, ?, ?, ?)",
                     [(i, *[None if ranks[b, i] < 0 else int(ranks[b, i]) for b in range(3)]) for i in range(n)])
    for desc, fn, fb in ((False, "min", BIG), (True, "max", -BIG)):
        got = [r[0] for r in conn.execute(f"SELECT {fn}(coalesce(a, {fb}), coalesce(b, {fb}), coalesce(c, {fb})) FROM t ORDER BY i")]
        exp_o = [orc.coalesce_rank(ranks[:, i], desc) for i in range(n)]
        assert got == exp_o, "oracle restatement differs from SQLite"
        assert pvs.coalesce_ranks(ranks, desc).tolist() == got
        assert got[0] == fb
    # raw aggregates (order_rank without row_n): reals against the integer fallback
    vals = rng.standard_normal((2, n)) * 3
    vals[rng.random((2, n)) < 0.3] = np.nan
    vals[:, 1] = np.nan
    vals[0, 2], vals[1, 2] = 1e19, np.nan     # a real beyond the fallback: the INTEGER fallback wins ascending
    conn.execute("CREATE TABLE v (i INTEGER PRIMARY KEY, a REAL, b REAL)")
    conn.executemany("INSERT INTO v VALUES (?, ?, ?)", [(i, *[None if np.isnan(vals[b, i]) else float(vals[b, i]) for b in range(2)]) for i in range(n)])
    for desc, fn, fb in ((False, "min", BIG), (True, "max", -BIG)):
        got = np.array([float(r[0]) for r in conn.execute(f"SELECT {fn}(coalesce(a, {fb}), coalesce(b, {fb})) FROM v ORDER BY i")])
        assert np.array_equal(pvs.coalesce_values(vals, desc), got)
    # sort bounds on order_rank (REAL aggregate or INTEGER rank), NULL rows drop out as soon as a bound is given
    col = vals[0].copy()
    for gt, lt in ((None, None), (-0.5, None), (None, 1.25), (-1.0, 2.0), (3.0, -3.0)):
        conds = " AND ".join(x for x in [None if gt is None else f"a > {gt!r}", None if lt is None else f"a < {lt!r}"] if x)
        rows = {r[0] for r in conn.execute(f"SELECT i FROM v {'WHERE ' + conds if conds else ''}")}
        keep = pvs.sort_bounds(col, gt, lt)
        assert set(np.flatnonzero(keep).tolist()) == rows, (gt, lt)
        assert [orc.sort_bounds_keep(x, gt, lt) for x in col] == keep.tolist()


def test_synth_rows_are_unit_vectors_and_deterministic():
    a = orc.synth_rows(20260928, 0, 64, 768)
    b = orc.synth_rows(20260928, 32, 32, 768)
    assert np.array_equal(a[32:], b)
    n = np.linalg.norm(a.astype(np.float64), axis=1)
    assert np.allclose(n, 1.0, atol=1e-6)
    comp = a.ravel() * np.sqrt(768)
    assert abs(comp.mean()) < 0.02 and abs(comp.std() - 1.0) < 0.02


def _fixture_vi():
    import zlib

    f = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fixture_vi.npz"))
    rows = orc.synth_rows(int(f["seed_rows"]), 0, int(f["n"]), int(f["dim"]))
    queries = orc.synth_rows(int(f["seed_queries"]), 0, 8, int(f["dim"]))
    assert zlib.crc32(rows.tobytes()) == int(f["rows_crc32"]) and zlib.crc32(queries.tobytes()) == int(f["queries_crc32"]), "the synthetic generator drifted"
    return f, rows, queries


def test_oracle_reproduces_the_frozen_fixture_vi():
    """SURVEY §8c fixture (vi): 10k x 512 seeded corpus, 8 queries, top-10 for {f32, f16, i8} x {cosine, L2} — frozen bytes
    (tests/golden/fixture_vi.npz, made by tests/golden/make_fixture_vi.py).  The oracle must keep producing them; the GPU test
    test_device_reproduces_the_frozen_fixture_vi checks the kernels against the same bytes."""
    f, rows, queries = _fixture_vi()
    scale = float(f["scale"])
    assert np.float32(orc.compute_int8_scale(rows)) == np.float32(scale)
    corp = {"f32": (orc.F32, rows, queries), "f16": (orc.F16, rows.astype(np.float16), queries),
            "i8": (orc.I8, orc.quantize_int8(rows, scale), orc.quantize_int8(queries, scale))}
    for name, (dt, c, q) in corp.items():
        for mname, m in (("cosine", orc.COSINE), ("l2", orc.L2)):
            ids, dist = orc.search(dt, m, c, q, int(f["k"]))
            assert np.array_equal(ids, f[f"{name}_{mname}_ids"]), (name, mname)
            assert np.array_equal(dist.view(np.uint32), f[f"{name}_{mname}_dist"].view(np.uint32)), (name, mname)
            # and the fixture itself agrees with the f64 formulas within the north star's 1e-5 relative (floats) / the reference's 1e-4 (int8)
            cf = c.astype(np.float64)
            qf = q.astype(np.float64)
            for qi in range(len(q)):
                a = cf[ids[qi]]
                if m == orc.COSINE:
                    exact = 1.0 - (a @ qf[qi]) / (np.linalg.norm(a, axis=1) * np.linalg.norm(qf[qi]))
                else:
                    exact = np.linalg.norm(a - qf[qi], axis=1)
                tol = 1e-4 if name == "i8" else 1e-5
                assert np.all(np.abs(dist[qi] - exact) <= tol * np.maximum(np.abs(exact), 1e-3) + 1e-6), (name, mname, qi)


def test_clustered_generator_has_the_shape_it_claims():
    """orc_synth_rows_clustered (the realistic-distribution corpus of bench.py's round-6 secondaries; identical bytes on the device,
    tests/test_gpu_parity.py): unit rows, ~1 % exact duplicates of a row shortly before, power-law cluster sizes, near-duplicate
    runs, and a pure function of (seed, row): chunked generation equals one call."""
    n, dim = 30_000, 96
    r = orc.synth_rows_clustered(99, 0, n, dim)
    assert np.allclose(np.linalg.norm(r.astype(np.float64), axis=1), 1.0, atol=1e-6)
    parts = np.concatenate([orc.synth_rows_clustered(99, off, min(7001, n - off), dim) for off in range(0, n, 7001)])
    assert np.array_equal(parts.view(np.uint32), r.view(np.uint32))
    uniq = len(np.unique(r, axis=0))
    assert 0.005 * n < n - uniq < 0.02 * n, (n, uniq)
    cl = np.array([orc.synth_cluster_of(99, i) for i in range(n)])
    counts = np.sort(np.bincount(cl, minlength=2000))[::-1]
    assert counts[0] > 0.05 * n and counts[:20].sum() > 0.15 * n and (counts > 0).sum() > 1500  # a few big clusters, a long tail
    # rows of one cluster are close, rows of different clusters are not; near-duplicate runs are closer still
    same = cl[:4000, None] == cl[None, :4000]
    d = 1.0 - r[:4000].astype(np.float64) @ r[:4000].astype(np.float64).T
    assert np.median(d[same & ~np.eye(4000, dtype=bool)]) < 0.5 < np.median(d[~same])
    assert (d[np.triu(np.ones((4000, 4000), bool), 1)] < 0.02).sum() > 100


def test_oracle_distances_equal_an_independent_numpy_restatement():
    """A second, independent restatement of sqlite-vec 0.1.9's scalar loops (the oracle's C is the first; neither is the reference's
    binary — its source is not in the image): sequential f32 accumulation in component order, written with numpy
    primitives whose rounding is defined — an f32 multiply per element, `np.cumsum` over float32 (a strictly sequential f32 running
    sum, no pairwise tree) — then sqrt / divide in double and one narrowing to f32.  Bit for bit equal to the oracle for f32, f16
    (widened) and int8 rows, cosine and L2, over dimensions that are and are not multiples of anything."""
    rng = np.random.default_rng(123)

    def seq_sum(x):  # sequential f32 sum in order
        return np.float32(0.0) if x.size == 0 else np.cumsum(x.astype(np.float32), dtype=np.float32)[-1]

    def np_cosine(a, b):
        a, b = a.astype(np.float32), b.astype(np.float32)
        dot, aa, bb = seq_sum(a * b), seq_sum(a * a), seq_sum(b * b)
        with np.errstate(all="ignore"):
            return np.float32(1.0 - np.float64(dot) / (np.sqrt(np.float64(aa)) * np.sqrt(np.float64(bb))))

    def np_l2(a, b):
        a, b = a.astype(np.float32), b.astype(np.float32)
        d = a - b
        return np.float32(np.sqrt(np.float64(seq_sum(d * d))))

    for dim in (1, 3, 8, 100, 512, 768, 1027):
        rows = (rng.standard_normal((40, dim)) * rng.choice([1e-3, 1.0, 50.0], (40, 1))).astype(np.float32)
        q = rng.standard_normal(dim).astype(np.float32)
        for dt, data, query in ((orc.F32, rows, q), (orc.F16, rows.astype(np.float16), q),
                                (orc.I8, rng.integers(-128, 128, (40, dim)).astype(np.int8), rng.integers(-128, 128, dim).astype(np.int8))):
            wide = data.astype(np.float32)
            qw = query.astype(np.float32)
            for metric, fn in ((orc.COSINE, np_cosine), (orc.L2, np_l2)):
                got = orc.score_all(dt, metric, data, query)
                exp = np.array([fn(wide[i], qw) for i in range(len(wide))], np.float32)
                assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), (dim, dt, metric)
