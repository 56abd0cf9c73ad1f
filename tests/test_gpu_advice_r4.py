"""Round-4 regression tests for the advisor's findings."""
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


def test_page_first_ranking_with_a_column_that_fills_its_page_exactly(pvs):
    """pvs_rrf.hip k_gather_page_cols treated count == cap as an overflow while the host read cap entries back: a column that
    admits EXACTLY cap groups returned uninitialised scratch.  262,144 groups, 8 query columns; every 8th group holds the same
    vector (32,768 = cap exact ties at the smallest value), every sample point of the threshold estimate lands on one of them."""
    n, dim, k, ncol = 262_144, 32, 50, 8
    rows = orc.synth_rows(91, 0, n, dim)
    near = orc.synth_rows(92, 0, 1, dim)[0]
    rows[::8] = near
    scale = orc.compute_int8_scale(rows)
    ix = pvs.VectorIndex(pvs.I8, dim)
    ix.set_scale(scale)
    ix.add_f32(rows)
    q = np.tile(near, (ncol, 1)) + 0.01 * orc.synth_rows(93, 0, ncol, dim)  # all eight columns closest to the tied vector
    hq = orc.quantize_int8(q.astype(np.float32), scale)
    hc = orc.quantize_int8(rows, scale)
    for agg, oagg in ((pvs.AGG_AVG, orc.AGG_AVG), (pvs.AGG_MAX, orc.AGG_MAX)):
        gg, gv, gc = ix.search_groups(hq, k, pvs.COSINE, agg)
        for j in range(ncol):
            d = orc.score_all(pvs.I8, pvs.COSINE, hc, hq[j])
            eg, ev = orc.aggregate(d, np.arange(n, dtype=np.int64), oagg)
            order = np.lexsort((eg, ev))[:k]
            assert gc[j] == k
            assert np.array_equal(gg[j], eg[order]), (agg, j)
            assert np.array_equal(gv[j].view(np.uint64), ev[order].view(np.uint64)), (agg, j)
    ix.close()


def test_bounded_search_with_a_deep_lower_bound_on_a_multi_device_index(pvs):
    """pvs_search_bounded on a multi-device index used to grow its page towards k = n and fail ('k too large') past 2^20 rows; now
    every shard answers the bounded search (its own dense path for a deep bound) and the pages merge.  A bound 1.3M rows deep."""
    n, dim, k = 1_400_000, 32, 25
    rows = orc.synth_rows(17, 0, n, dim)
    scale = orc.compute_int8_scale(rows)
    hc = orc.quantize_int8(rows, scale)
    hq = orc.quantize_int8(orc.synth_rows(18, 0, 2, dim), scale)
    ix = pvs.VectorIndex(pvs.I8, dim, devices=[0, 0])
    ix.set_scale(scale)
    ix.add_f32(rows)
    one = pvs.VectorIndex(pvs.I8, dim)
    one.set_scale(scale)
    one.add_f32(rows)
    for j in range(2):
        d = orc.score_all(pvs.I8, pvs.L2, hc, hq[j])
        gt = float(np.sort(d)[1_300_000])
        lt = float(np.sort(d)[1_300_000 + 4000])
        for bounds in ((gt, None), (gt, lt)):
            gi, gd, gc = ix.search_bounded(hq[j: j + 1], k, pvs.L2, gt=bounds[0], lt=bounds[1])
            si, sd, sc = one.search_bounded(hq[j: j + 1], k, pvs.L2, gt=bounds[0], lt=bounds[1])
            ok = d > bounds[0]
            if bounds[1] is not None:
                ok &= d < bounds[1]
            ei, ed = orc.topk(d[ok], k, ids=np.nonzero(ok)[0].astype(np.int64))
            for got in ((gi, gd, gc), (si, sd, sc)):
                assert got[2][0] == len(ei)
                assert np.array_equal(got[0][0, : len(ei)], ei) and np.array_equal(got[1][0, : len(ei)].view(np.uint32), ed.view(np.uint32))
    # fewer rows than k on a single device: the page is everything, no dense pass
    small = pvs.VectorIndex(pvs.I8, dim)
    small.set_scale(scale)
    small.add_f32(rows[:10])
    gi, gd, gc = small.search_bounded(hq[:1], k, pvs.L2, gt=-1.0, lt=None)
    assert gc[0] == 10 and small.stats().dense_queries == 0
    small.close()
    one.close()
    ix.close()


def test_stats_entry_points_never_read_the_callers_struct(pvs):
    import ctypes as C

    from panoptikon_amd import _lib as L

    ix = pvs.VectorIndex(pvs.F32, 8)
    ix.add_f32(orc.synth_rows(1, 0, 10, 8))
    s = L.Stats()
    C.memset(C.byref(s), 0xAB, C.sizeof(s))  # garbage, as an ABI v2 caller's uninitialised struct would be
    L.check(pvs.lib().pvs_index_stats(ix._h, C.byref(s)))
    assert s.rows == 10 and s.struct_size == L.Stats.rescanned_queries.offset
    assert s.rescanned_queries == 0xABABABABABABABAB, "the v2 entry point must not write past the v2 fields"
    L.check(pvs.lib().pvs_index_stats_ex(ix._h, C.byref(s), C.sizeof(s)))
    assert s.struct_size == C.sizeof(s) and s.rescanned_queries == 0 and s.sparse_queries == 0
    assert pvs.lib().pvs_index_stats_ex(ix._h, C.byref(s), 8) != 0
    ix.close()


def test_keyed_merge_of_row_pages_host_and_device(pvs):
    """pvs_merge_topk_keyed[_device]: (distance asc, NULL last, key DESC, id asc) — the order every keyed route produces."""
    rng = np.random.default_rng(4)
    world, batch, k = 3, 5, 16
    ids = np.full((world, batch, k), -1, np.int64)
    dist = np.full((world, batch, k), np.nan, np.float32)
    keys = np.zeros((world, batch, k), np.int64)
    cnt = np.zeros((world, batch), np.uint32)
    exp = []
    for q in range(batch):
        allv = []
        for w in range(world):
            c = int(rng.integers(0, k + 1))
            d = np.sort(rng.integers(0, 4, c).astype(np.float32))
            if c and q % 2:
                d[-1] = np.nan
            ky = rng.integers(0, 3, c).astype(np.int64)
            idv = (np.arange(c) * world + w + 100 * q).astype(np.int64)
            # each shard's page is ordered by the same rule
            isn = np.isnan(d)
            o = np.lexsort((idv, -ky, np.where(isn, 0, d), isn))
            ids[w, q, :c], dist[w, q, :c], keys[w, q, :c], cnt[w, q] = idv[o], d[o], ky[o], c
            allv += [(bool(np.isnan(x)), 0.0 if np.isnan(x) else float(x), -int(kk), int(i)) for x, kk, i in zip(d, ky, idv)]
        allv.sort()
        exp.append([t[3] for t in allv[:k]])
    oi, od, oc = pvs.merge_topk(ids, dist, cnt, k, keys=keys)
    for q in range(batch):
        assert oi[q, : oc[q]].tolist() == exp[q]
    from panoptikon_amd import _lib as L

    d_in = [pvs.DeviceBuffer.from_numpy(a) for a in (ids, dist, keys, cnt)]
    d_out = [pvs.DeviceBuffer(batch * k * 8), pvs.DeviceBuffer(batch * k * 4), pvs.DeviceBuffer(batch * 4)]
    L.check(pvs.lib().pvs_merge_topk_keyed_device(0, d_in[0].ptr, d_in[1].ptr, d_in[2].ptr, d_in[3].ptr, world, batch, k, d_out[0].ptr, d_out[1].ptr, d_out[2].ptr))
    di = d_out[0].to_numpy(np.int64, (batch, k))
    dc = d_out[2].to_numpy(np.uint32, (batch,))
    assert np.array_equal(dc, oc)
    for q in range(batch):
        assert di[q, : dc[q]].tolist() == exp[q]
    for b in d_in + d_out:
        b.free()
