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
            for key in ("no_agg8", "no_exact_wide", "no_page_rank", "no_flag_poll"):  # (no_flag_poll: the pages waited for on the stream, not polled in pinned memory)
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


@pytest.mark.parametrize("dtype,dim,n,b", [("f16", 1280, 700, 40), ("f16", 1288, 333, 33), ("f32", 1280, 450, 64), ("f32", 1536, 257, 35), ("f16", 2600, 131, 20),
                                              ("f32", 5, 1000, 32), ("f16", 96, 127, 9), ("f32", 768, 64, 17), ("f16", 768, 700, 8), ("f32", 300, 129, 8),
                                              ("f16", 1024, 333, 8), ("f32", 5, 1000, 8), ("f16", 1030, 200, 8), ("f32", 768, 4097, 48), ("f16", 40, 70000, 8)])
def test_exact_wide_pitch_boundaries_and_ragged_ends(pvs, dtype, dim, n, b):
    """k_exact_wide keeps the transposed queries in LDS: 32 per pass up to 1,280 components, 16 up to 2,560, wider rows go back to
    k_dense_exact's 8 per pass; row counts that are not multiples of a wave's 128 / 192 rows (lanes past the last tile re-read it and
    write nothing), one-slab rows, batches that end in a 16-query or an 8-query pass.  Eight queries take k_dense_exact2 (two rows
    per lane, half-slab stages gathered by the LDS-DMA) while 8 padded queries fit beside its ring (pitch <= 1,024 components), the
    one-row form beyond; both must agree.  Every distance against the oracle."""
    dt = {"f16": pvs.F16, "f32": pvs.F32}[dtype]
    rows = orc.synth_rows(300 + dim, 0, n, dim)
    rows[n // 3] = 0.0
    q = orc.synth_rows(301 + dim, 0, b, dim)
    ix = pvs.VectorIndex(dt, dim)
    ix.add_f32(rows)
    hc = rows.astype(np.float16) if dt == pvs.F16 else rows
    for metric in (pvs.COSINE, pvs.L2):
        got = ix.score_batch(q, metric)
        assert got.shape == (n, b)
        if b % 8 == 0:
            pvs.debug_set("no_dense2", 1)
            try:
                old = ix.score_batch(q, metric)
            finally:
                pvs.debug_set("no_dense2", 0)
            assert np.array_equal(got.view(np.uint32), old.view(np.uint32)), (dtype, dim, metric, "one row per lane")
        for j in range(b):
            exp = orc.score_all(dt, metric, hc, q[j])
            assert np.array_equal(np.isnan(got[:, j]), np.isnan(exp)), (dtype, dim, metric, j)
            ok = ~np.isnan(exp)
            assert np.array_equal(got[ok, j].view(np.uint32), exp[ok].view(np.uint32)), (dtype, dim, metric, j)
    ix.close()
