"""Per-item pages over FLOAT rows by bound + certify + exact rescan (csrc/pvs_items_float.hip, round 6).  The reference's exact mode
is f32 (filters/exact.rs:106-165) and similar_to defaults to AVG (item_similarity.rs:432-581): GROUP BY file + rank_aggregate over
every row's distance, a page of k files.  The certified route brackets every file's aggregate from the matrix-core scan keys
(k_scan MODE 4) and runs the reference's in-order chain only on the files that can reach the page; its pages must be, bit for bit,
the exact-everywhere route's (pvs_debug_set("no_float_certify")) and the oracle's — near-duplicate files, NULL rows, NULL queries,
weights, candidate masks, scattered files, order keys, both metrics, f16 and f32, and a seeded sweep."""
import numpy as np
import pytest

import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pvs():
    import panoptikon_amd as p

    if p.device_count() < 1:
        pytest.fail("no gfx950 device visible: the gpu tests need an MI355X")
    return p


def _same(a, b):
    return all(np.array_equal(x, y) if x.dtype != np.float64 else np.array_equal(x.view(np.uint64), y.view(np.uint64)) for x, y in zip(a, b))


def _check_groups(got, exp, tag):
    og, ov, oc = got
    eg, ev = exp
    assert oc == len(eg), (tag, oc, len(eg))
    assert np.array_equal(og[:oc], eg), tag
    a = ov[:oc]
    assert np.array_equal(np.isnan(a), np.isnan(ev)), tag
    assert np.array_equal(a[~np.isnan(a)].view(np.uint64), ev[~np.isnan(ev)].view(np.uint64)), tag


def _both_routes(pvs, fn):
    """fn() through the certified route (counted) and through the exact-everywhere route; returns (certified result, queries it
    certified, candidate rows it rescanned, exact-everywhere result)."""
    q0, r0 = pvs.debug_get("float_certify_queries"), pvs.debug_get("float_certify_rows")
    got = fn()
    nq, nr = pvs.debug_get("float_certify_queries") - q0, pvs.debug_get("float_certify_rows") - r0
    pvs.debug_set("no_float_certify", 1)
    try:
        old = fn()
    finally:
        pvs.debug_set("no_float_certify", 0)
    return got, nq, nr, old


@pytest.mark.parametrize("dtype", ["f16", "f32"])
@pytest.mark.parametrize("scattered", [False, True])
def test_certified_pages_equal_exact_everywhere_and_the_oracle(pvs, dtype, scattered):
    dt = {"f16": pvs.F16, "f32": pvs.F32}[dtype]
    rng = np.random.default_rng(77 + scattered)
    n, dim, k, batch = 180_000, 96, 25, 20
    if scattered:
        grp = rng.integers(0, n // 3, n).astype(np.int64) * 5 + 7
    else:
        grp = np.sort(rng.integers(0, n // 3, n)).astype(np.int64)
    rows = orc.synth_rows(191, 0, n, dim)
    rows[rng.integers(0, n, 40)] = 0.0                      # NULL cosine distances inside files
    rows[5000:5040] = rows[5000]                            # a run of identical rows: files that tie exactly
    rows[70_000:70_030] *= np.float32(1e-3)                 # small norms
    rows[90_000:90_020] *= np.float32(300.0)                # large norms (L2 brackets scale with |a|^2)
    ix = pvs.VectorIndex(dt, dim)
    ix.add_f32(rows, group_ids=grp)
    hc = rows.astype(np.float16) if dt == pvs.F16 else rows
    q = orc.synth_rows(192, 0, batch, dim)
    q[3] = 0.0                                              # every cosine distance of this column NULL: answered outside the certified route
    q[5] = rows[5000]                                       # lands on the tied files
    w = (rng.random(n) + 0.05).astype(np.float32)
    mask = (rng.random(n) < 0.6).astype(np.uint8)
    allowed = np.nonzero(mask)[0]
    for metric in (pvs.COSINE, pvs.L2):
        for agg, oagg, weights in ((pvs.AGG_AVG, orc.AGG_AVG, None), (pvs.AGG_MAX, orc.AGG_MAX, None), (pvs.AGG_AVG, orc.AGG_AVG, w), (pvs.AGG_MIN, orc.AGG_MIN, None)):
            got, nq, nr, old = _both_routes(pvs, lambda: ix.search_groups(q, k, metric, agg, row_weights=weights))
            tag = (dtype, scattered, metric, agg, weights is not None)
            assert _same(got, old), tag
            # the brackets folded in the scan's epilogue (k_scan MODE 5: files that are runs) against the key matrix + a second kernel
            # (MODE 4 + k_run_bounds): the same pages
            pvs.debug_set("float_certify_no_fold", 1)
            try:
                two_pass = ix.search_groups(q, k, metric, agg, row_weights=weights)
            finally:
                pvs.debug_set("float_certify_no_fold", 0)
            assert _same(got, two_pass), (tag, "fold in the scan vs key matrix")
            if agg != pvs.AGG_MIN:  # (MIN pages come from row pages of the filter scan first)
                # (q[3] = 0: under cosine nothing of it can be bracketed; under L2 every unit row is at distance 1 from it — a MAX that ties
                #  with every file is set aside too; each is answered by the exact-everywhere route, the rest of the chunk certified)
                assert batch - 1 <= nq <= batch, (tag, nq)
                assert 0 < nr < n // 8, (tag, nr)
            for j in (0, 3, 5, batch - 1):
                exp = orc.search_groups(dt, metric, hc, q[j], grp, oagg, k, weights=weights)
                _check_groups((got[0][j], got[1][j], got[2][j]), exp, (tag, j))
        got, nq, nr, old = _both_routes(pvs, lambda: ix.search_groups_filtered(q, k, mask, metric, pvs.AGG_AVG))
        assert _same(got, old), (dtype, scattered, metric, "mask")
        assert nq > 0
        for j in (1, 5, batch - 1):
            exp = orc.search_groups(dt, metric, hc[allowed], q[j], grp[allowed], orc.AGG_AVG, k)
            _check_groups((got[0][j], got[1][j], got[2][j]), exp, (dtype, scattered, metric, "mask", j))
    # the second sort key: files that tie on the value are ordered by key DESC, then id
    keys = (rng.integers(0, 4, n) + 1_700_000_000).astype(np.int64)
    first = {}
    for r, g in enumerate(grp):
        first.setdefault(int(g), keys[r])
    keys = np.array([first[int(g)] for g in grp], np.int64)  # (a file's rows share its key)
    ix.set_order_keys(keys)
    got, nq, nr, old = _both_routes(pvs, lambda: ix.search_groups(q, k, pvs.COSINE, pvs.AGG_AVG))
    assert _same(got, old) and nq > 0
    for j in (5, 7):
        exp = orc.search_groups(dt, pvs.COSINE, hc, q[j], grp, orc.AGG_AVG, k, order_keys=keys)
        _check_groups((got[0][j], got[1][j], got[2][j]), exp, (dtype, scattered, "keys", j))
    ix.close()


def test_whatever_cannot_be_certified_is_handed_back(pvs):
    """Massive ties (every file the same vector), a page deeper than the bracketed files, k beyond what the threshold buckets resolve:
    the exact-everywhere route answers, the pages are the oracle's."""
    dim, n, k = 64, 60_000, 30
    rows = np.repeat(orc.synth_rows(5, 0, 1, dim), n, axis=0)
    rows[::7] = orc.synth_rows(6, 0, len(rows[::7]), dim)
    grp = np.arange(n, dtype=np.int64) // 2
    ix = pvs.VectorIndex(pvs.F32, dim)
    ix.add_f32(rows, group_ids=grp)
    q = orc.synth_rows(7, 0, 12, dim)
    q[0] = rows[1]
    got, nq, nr, old = _both_routes(pvs, lambda: ix.search_groups(q, k, pvs.COSINE, pvs.AGG_AVG))
    assert _same(got, old)
    for j in (0, 1, 11):
        _check_groups((got[0][j], got[1][j], got[2][j]), orc.search_groups(orc.F32, orc.COSINE, rows, q[j], grp, orc.AGG_AVG, k), ("ties", j))
    # k = 2,000: beyond the buckets' resolution -> never the certified route
    got, nq, nr, old = _both_routes(pvs, lambda: ix.search_groups(q[:9], 2000, pvs.L2, pvs.AGG_MAX))
    assert nq == 0 and _same(got, old)
    ix.close()


def test_seeded_sweep_against_the_exact_route(pvs):
    """300 seeded shapes: dims, file sizes (1..40 rows, runs or scattered), k, metric, aggregate, weights, masks, batch sizes that end
    in 32- / 64- / 128-query scans — certified == exact everywhere, and one column per shape against the oracle."""
    total_certified = 0
    for seed in range(300):
        rng = np.random.default_rng(10_000 + seed)
        dtype = pvs.F16 if rng.random() < 0.5 else pvs.F32
        dim = int(rng.choice([24, 64, 100, 128, 200, 384]))
        n = int(rng.integers(20_000, 70_000))
        per = int(rng.choice([1, 2, 3, 5, 12, 40]))
        grp = (np.arange(n, dtype=np.int64) // per) if rng.random() < 0.6 else rng.integers(0, max(n // per, 1), n).astype(np.int64)
        if len(np.unique(grp)) < 3000:
            continue
        batch = int(rng.choice([9, 16, 33, 70]))
        k = int(rng.choice([5, 10, 40]))
        metric = pvs.COSINE if rng.random() < 0.5 else pvs.L2
        agg, oagg = [(pvs.AGG_AVG, orc.AGG_AVG), (pvs.AGG_MAX, orc.AGG_MAX), (pvs.AGG_MIN, orc.AGG_MIN)][int(rng.integers(0, 3))]
        base = orc.synth_rows(seed + 1, 0, 200, dim)
        rows = (base[rng.integers(0, 200, n)] + float(rng.choice([0.05, 0.3, 1.0])) * orc.synth_rows(seed + 2, 0, n, dim)).astype(np.float32)
        if rng.random() < 0.3:
            rows[rng.integers(0, n, 5)] = 0.0
        weights = (rng.random(n) + 0.01).astype(np.float32) if rng.random() < 0.25 else None
        mask = (rng.random(n) < 0.7).astype(np.uint8) if rng.random() < 0.25 else None
        ix = pvs.VectorIndex(dtype, dim)
        ix.add_f32(rows, group_ids=grp)
        q = (base[rng.integers(0, 200, batch)] + 0.2 * orc.synth_rows(seed + 3, 0, batch, dim)).astype(np.float32)

        def run():
            if mask is not None:
                return ix.search_groups_filtered(q, k, mask, metric, agg, row_weights=weights)
            return ix.search_groups(q, k, metric, agg, row_weights=weights)

        got, nq, nr, old = _both_routes(pvs, run)
        tag = (seed, dim, n, per, batch, k, metric, agg, weights is not None, mask is not None)
        assert _same(got, old), tag
        total_certified += nq
        j = int(rng.integers(0, batch))
        hc = rows.astype(np.float16) if dtype == pvs.F16 else rows
        sel = np.nonzero(mask)[0] if mask is not None else np.arange(n)
        exp = orc.search_groups(dtype, metric, hc[sel], q[j], grp[sel], oagg, k, weights=None if weights is None else weights[sel])
        _check_groups((got[0][j], got[1][j], got[2][j]), exp, tag)
        ix.close()
    assert total_certified > 2000, total_certified
