"""The one-pass per-item scorer (k_scan MODE 3: GROUP BY file_id + rank_aggregate folded into the distance kernel's epilogue,
filters/exact.rs:67-134) against the oracle's literal composition, and against the round-3 route (N x B matrix + k_group_aggregate,
pvs_debug_set("no_fused_agg", 1)): groups of 1 to 90 rows (inside a 32-row tile, across one boundary, across several tiles), NULL
rows, candidate masks (host and device), row weights, the second sort key, batches that pad to 32 / 64 / 128 queries and split into
chunks, row pitches of 256 B to 1 KiB, and a layout whose groups are NOT runs of rows (served by the two-pass route)."""
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


def _check(got, exp, tag):
    og, ov, oc = got
    eg, ev = exp
    assert oc == len(eg), (tag, oc, len(eg))
    assert np.array_equal(og[:oc], eg), tag
    a = ov[:oc]
    assert np.array_equal(np.isnan(a), np.isnan(ev)), tag
    assert np.array_equal(a[~np.isnan(a)].view(np.uint64), ev[~np.isnan(ev)].view(np.uint64)), tag


@pytest.mark.parametrize("dim", [64, 300, 768, 1000])
def test_fused_per_item_scorer_equals_the_oracle(pvs, dim):
    rng = np.random.default_rng(1000 + dim)
    sizes = rng.choice([1, 1, 2, 3, 3, 4, 7, 31, 32, 33, 64, 90], 900)
    grp = np.repeat(np.arange(len(sizes), dtype=np.int64) * 2 + 5, sizes)
    n = len(grp)
    base = orc.synth_rows(50 + dim, 0, 400, dim)
    rows = base[rng.integers(0, 400, n)] + (0.05 * orc.synth_rows(60 + dim, 0, n, dim) if dim != 300 else 0)  # dim 300: exact ties across files
    rows = rows.astype(np.float32)
    rows[grp == grp[n // 3]] = 0.0  # a whole file of zero vectors: NULL cosine aggregate
    rows[7] = 0.0                   # and a single NULL row inside a file
    scale = orc.compute_int8_scale(rows)
    codes = orc.quantize_int8(rows, scale)
    ix = pvs.VectorIndex(pvs.I8, dim)
    ix.set_scale(scale)
    ix.add_f32(rows, group_ids=grp)
    w = (rng.random(n) + 0.1).astype(np.float32)
    mask = (rng.random(n) < 0.5).astype(np.uint8)
    allowed = np.nonzero(mask)[0]
    keys_g = rng.integers(0, 4, len(sizes)).astype(np.int64)
    keys = np.repeat(keys_g, sizes)
    for nb in (1, 2, 4, 5, 40, 130):  # (1..4 queries: the v_dot4 scorer with the fold in its tile epilogue; from 5 the matrix-core scorer)
        q = orc.synth_rows(70 + nb, 0, nb, dim)
        hq = orc.quantize_int8(q, scale)
        check = sorted({0, nb // 2, nb - 1})
        for keyed in (False, True):
            ix.set_order_keys(keys if keyed else None)
            ok = keys if keyed else None
            for metric, om in ((pvs.COSINE, orc.COSINE), (pvs.L2, orc.L2)):
                for agg, oagg, ww in ((pvs.AGG_MIN, orc.AGG_MIN, None), (pvs.AGG_MAX, orc.AGG_MAX, None), (pvs.AGG_AVG, orc.AGG_AVG, None), (pvs.AGG_AVG, orc.AGG_AVG, w)):
                    if nb == 130 and (keyed or agg == pvs.AGG_MAX):
                        continue  # (the chunked batch: a subset is enough)
                    k = 17
                    before = ix.stats().dense_queries
                    got = ix.search_groups(hq, k, metric, agg, row_weights=ww)
                    assert ix.stats().dense_queries == before, "the one-pass scorer writes no N x B matrix"
                    for j in check:
                        exp = orc.search_groups(orc.I8, om, codes, hq[j], grp, oagg, k, weights=ww, order_keys=ok)
                        _check((got[0][j], got[1][j], got[2][j]), exp, (dim, nb, keyed, metric, agg, ww is not None, j))
                    if nb in (1, 4, 40):  # the round-3 route returns the same pages, bit for bit
                        pvs.debug_set("no_fused_agg", 1)
                        try:
                            old = ix.search_groups(hq, k, metric, agg, row_weights=ww)
                        finally:
                            pvs.debug_set("no_fused_agg", 0)
                        assert np.array_equal(old[0], got[0]) and np.array_equal(old[2], got[2])
                        assert np.array_equal(old[1].view(np.uint64), got[1].view(np.uint64))
                    if nb in (2, 5):  # candidate masks, from host memory and from HBM
                        mg = ix.search_groups_filtered(hq, k, mask, metric, agg, row_weights=ww)
                        for j in check:
                            exp = orc.search_groups(orc.I8, om, codes[allowed], hq[j], grp[allowed], oagg, k, weights=None if ww is None else ww[allowed],
                                                    order_keys=None if ok is None else ok[allowed])
                            _check((mg[0][j], mg[1][j], mg[2][j]), exp, (dim, "mask", keyed, metric, agg, ww is not None, j))
    ix.close()


def test_groups_that_are_not_runs_of_rows_take_the_two_pass_route(pvs):
    rng = np.random.default_rng(5)
    n, dim, k = 5000, 128, 12
    grp = rng.integers(0, 700, n).astype(np.int64)  # interleaved: a file's vectors are scattered over the rows
    rows = orc.synth_rows(9, 0, n, dim)
    scale = orc.compute_int8_scale(rows)
    codes = orc.quantize_int8(rows, scale)
    ix = pvs.VectorIndex(pvs.I8, dim)
    ix.set_scale(scale)
    ix.add_f32(rows, group_ids=grp)
    hq = orc.quantize_int8(orc.synth_rows(10, 0, 6, dim), scale)
    before = ix.stats().dense_queries
    got = ix.search_groups(hq, k, pvs.COSINE, pvs.AGG_AVG)
    assert ix.stats().dense_queries == before + 6
    for j in range(6):
        _check((got[0][j], got[1][j], got[2][j]), orc.search_groups(orc.I8, orc.COSINE, codes, hq[j], grp, orc.AGG_AVG, k), j)
    ix.close()
