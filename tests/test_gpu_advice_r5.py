"""Round-5 regression tests for the advisor's findings."""
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


def _check_groups(got, exp, tag):
    og, ov, oc = got
    eg, ev = exp
    assert oc == len(eg), (tag, oc, len(eg))
    assert np.array_equal(og[:oc], eg), tag
    a = ov[:oc]
    assert np.array_equal(np.isnan(a), np.isnan(ev)), tag
    assert np.array_equal(a[~np.isnan(a)].view(np.uint64), ev[~np.isnan(ev)].view(np.uint64)), tag


@pytest.mark.parametrize("batch", [32, 48, 128])
def test_sparse_per_item_fallback_keeps_the_callers_queries(pvs, batch):
    """pvs_sparse_search_groups wrote its per-column counts over the start of query 0 in the context's pinned block (the pages
    were sized with 4 bytes per column, it writes 8): with >= 32 columns and one column handed back (a zero cosine query over more
    candidate groups than one LDS sort takes: every value NULL, all tied at the page edge) the corpus-pass fallback re-read a
    corrupted query 0.  Every column against the oracle over the allowed rows."""
    rng = np.random.default_rng(50 + batch)
    n, dim, k = 60_000, 64, 10
    grp = np.repeat(np.arange(n // 3, dtype=np.int64), 3)
    rows = orc.synth_rows(77, 0, n, dim)
    scale = orc.compute_int8_scale(rows)
    ix = pvs.VectorIndex(pvs.I8, dim)
    ix.set_scale(scale)
    ix.add_f32(rows, group_ids=grp)
    hc = orc.quantize_int8(rows, scale)
    q = orc.synth_rows(78, 0, batch, dim)
    q[batch // 2] = 0.0  # every cosine distance NULL for this column: > 2,048 sub-groups tie at its page edge
    hq = orc.quantize_int8(q, scale)
    mask = np.zeros(n, np.uint8)
    mask[rng.choice(n, 9000, replace=False)] = 1  # ~8,000 candidate groups: sparse-eligible, more than one LDS sort ranks
    allowed = np.nonzero(mask)[0]
    for agg, oagg in ((pvs.AGG_AVG, orc.AGG_AVG), (pvs.AGG_MAX, orc.AGG_MAX)):
        got = ix.search_groups_filtered(hq, k, mask, pvs.COSINE, agg)
        for j in (0, 1, batch // 2, batch - 1):
            exp = orc.search_groups(pvs.I8, pvs.COSINE, hc[allowed], hq[j], grp[allowed], oagg, k)
            _check_groups((got[0][j], got[1][j], got[2][j]), exp, (batch, agg, j))
    ix.close()


def test_list_validity_flags_are_plain_words(pvs):
    """k_sparse_score reports a bad candidate list through two words of the pinned block (plain stores, no PCIe atomics):
    an unsorted list and a row beyond the index are still rejected, each with its own message, and a good list right after
    them is served."""
    n, dim = 5000, 32
    rows = orc.synth_rows(5, 0, n, dim)
    ix = pvs.VectorIndex(pvs.F32, dim)
    ix.add_f32(rows)
    q = orc.synth_rows(6, 0, 1, dim)
    good = np.array([3, 9, 100, 4000], np.uint32)
    for bad, word in ((np.array([3, 100, 9, 4000], np.uint32), "ascending"), (np.array([3, 9, 100, 5000], np.uint32), "below")):
        with pytest.raises(pvs.PvsError) as e:
            ix.search_rows(q, 3, bad, pvs.COSINE)
        assert word in str(e.value)
        gi, gd, gc = ix.search_rows(q, 3, good, pvs.COSINE)
        ei, ed = orc.search(pvs.F32, pvs.COSINE, rows[good], q, 3)
        assert gc[0] == 3 and np.array_equal(gi[0], good[ei[0]].astype(np.int64))
    ix.close()
