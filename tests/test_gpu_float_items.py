"""Per-item search over FLOAT rows with many queries (round 5): k_exact_wide (16 / 32 exact chains per pass), k_group_aggregate8
(a thread per group and eight columns) and the page ranking of all columns of a column-major value matrix in three launches —
against the oracle and against the routes they replace (pvs_debug_set no_exact_wide / no_agg8 / no_page_rank), bit for bit."""
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


@pytest.mark.parametrize("dtype,batch,scattered", [("f16", 32, False), ("f32", 16, False), ("f16", 24, True), ("f32", 40, True)])
def test_float_per_item_pages_many_queries(pvs, dtype, batch, scattered):
    dt = {"f16": pvs.F16, "f32": pvs.F32}[dtype]
    rng = np.random.default_rng(500 + batch)
    n, dim, k = 210_000, 48, 20
    if scattered:
        grp = rng.integers(0, n // 3, n).astype(np.int64) * 5 + 7  # a file's rows anywhere
    else:
        grp = np.sort(rng.integers(0, n // 3, n)).astype(np.int64)  # ~3 adjacent rows per file
    rows = orc.synth_rows(91, 0, n, dim)
    rows[1234] = 0.0  # a NULL cosine distance inside a group
    ix = pvs.VectorIndex(dt, dim)
    ix.add_f32(rows, group_ids=grp)
    hc = rows.astype(np.float16) if dt == pvs.F16 else rows
    q = orc.synth_rows(92, 0, batch, dim)
    q[3] = 0.0  # every cosine distance of this column NULL
    w = (rng.random(n) + 0.05).astype(np.float32)
    mask = (rng.random(n) < 0.6).astype(np.uint8)
    allowed = np.nonzero(mask)[0]
    for metric in (pvs.COSINE, pvs.L2):
        for agg, oagg, weights in ((pvs.AGG_AVG, orc.AGG_AVG, None), (pvs.AGG_MAX, orc.AGG_MAX, None), (pvs.AGG_AVG, orc.AGG_AVG, w)):
            got = ix.search_groups(q, k, metric, agg, row_weights=weights)
            for key in ("no_agg8", "no_exact_wide", "no_page_rank"):
                pvs.debug_set(key, 1)
                try:
                    old = ix.search_groups(q, k, metric, agg, row_weights=weights)
                finally:
                    pvs.debug_set(key, 0)
                assert _same(got, old), (dtype, metric, agg, weights is not None, key)
            for j in (0, 3, batch - 1):
                exp = orc.search_groups(dt, metric, hc, q[j], grp, oagg, k, weights=weights)
                _check_groups((got[0][j], got[1][j], got[2][j]), exp, (dtype, metric, agg, weights is not None, j))
        # a candidate mask: groups without an allowed row are absent, the others aggregate their allowed rows only
        got = ix.search_groups_filtered(q, k, mask, metric, pvs.AGG_AVG)
        pvs.debug_set("no_agg8", 1)
        try:
            old = ix.search_groups_filtered(q, k, mask, metric, pvs.AGG_AVG)
        finally:
            pvs.debug_set("no_agg8", 0)
        assert _same(got, old), (dtype, metric, "mask")
        for j in (1, batch - 1):
            exp = orc.search_groups(dt, metric, hc[allowed], q[j], grp[allowed], orc.AGG_AVG, k)
            _check_groups((got[0][j], got[1][j], got[2][j]), exp, (dtype, metric, "mask", j))
    ix.close()
