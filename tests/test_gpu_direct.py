"""The one-launch single-query search (csrc/pvs_direct.hip): exact in-order distances of every row + the page selected while the
rows stream, against the CPU oracle and against the filter scan (pvs_debug_set("no_direct_topk", 1)) — bit for bit."""
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


def _index(pvs, dt, rows, scale, ids=None):
    ix = pvs.VectorIndex(dt, rows.shape[1])
    if dt == pvs.I8:
        ix.set_scale(scale)
    ix.add_f32(rows, row_ids=ids)
    return ix


def _host(dt, rows, scale):
    if dt == orc.I8:
        return orc.quantize_int8(rows, scale)
    return rows.astype(np.float16) if dt == orc.F16 else rows


def _direct_searches(pvs):
    return pvs.debug_get("direct_queries")


def _same(got, exp_ids, exp_dist, k_eff):
    gi, gd, gc = got
    assert gc[0] == k_eff, (gc, k_eff)
    assert np.array_equal(gi[0, :k_eff], exp_ids[:k_eff])
    assert np.array_equal(gd[0, :k_eff].view(np.uint32), exp_dist[:k_eff].view(np.uint32))


SHAPES = [
    # dtype, metric, n, dim, k
    ("f32", "cosine", 10000, 512, 10),  # BASELINE configs[0]
    ("f32", "l2", 4097, 768, 100),
    ("f32", "cosine", 1, 5, 1),
    ("f32", "l2", 63, 300, 7),
    ("f32", "cosine", 65, 1000, 64),
    ("f32", "l2", 30000, 384, 256),
    ("f32", "cosine", 3000, 3072, 33),
    ("f16", "cosine", 20000, 768, 100),
    ("f16", "l2", 129, 64, 128),
    ("f16", "cosine", 9000, 1152, 17),
    ("f16", "l2", 50000, 100, 200),
    ("i8", "cosine", 30011, 768, 100),
    ("i8", "l2", 20000, 768, 10),
    ("i8", "cosine", 256, 512, 256),
    ("i8", "l2", 7000, 1536, 50),
    ("i8", "cosine", 64, 1024, 1),
    ("i8", "l2", 100003, 96, 129),
]


@pytest.mark.parametrize("dtype,metric,n,dim,k", SHAPES)
def test_direct_search_matches_oracle_and_filter_scan(pvs, dtype, metric, n, dim, k):
    dt = {"i8": pvs.I8, "f16": pvs.F16, "f32": pvs.F32}[dtype]
    m = pvs.COSINE if metric == "cosine" else pvs.L2
    rows = orc.synth_rows(17 + dim, 0, n, dim)
    queries = orc.synth_rows(0x5EED0100, 0, 3, dim)
    scale = orc.compute_int8_scale(rows)
    ids = np.arange(n, dtype=np.int64) * 3 + 11
    ix = _index(pvs, dt, rows, scale, ids)
    hc = _host(dt, rows, scale)
    k_eff = min(k, n)
    for qi in range(3):
        hq = orc.quantize_int8(queries[qi], scale) if dt == pvs.I8 else queries[qi]
        ei, ed = orc.search(dt, m, hc, hq, k, ids=ids, threads=4)
        before, dense_before = _direct_searches(pvs), ix.stats().dense_queries
        got = ix.search(queries[qi], k, m)
        assert _direct_searches(pvs) == before + 1, "a single query over a small corpus takes the one-launch search"
        assert ix.stats().dense_queries == dense_before
        _same(got, ei[0], ed[0], k_eff)
        pvs.debug_set("no_direct_topk", 1)
        try:
            ref = ix.search(queries[qi], k, m)
        finally:
            pvs.debug_set("no_direct_topk", 0)
        assert _direct_searches(pvs) == before + 1
        _same(ref, ei[0], ed[0], k_eff)  # (the filter scan, or the dense path where the row pitch has no scan instance)
    ix.close()


@pytest.mark.parametrize("dtype", ["i8", "f32"])
def test_direct_search_over_a_run_of_near_duplicates(pvs, dtype):
    """The k best rows stored side by side (frames of one video, copies of one picture): they sit in ONE workgroup's rows, so its
    list hands over more than the first 64 keys in the final merge; exact duplicates tie on the distance and come out by id."""
    dt = {"i8": pvs.I8, "f32": pvs.F32}[dtype]
    n, dim = 40000, 256
    rows = orc.synth_rows(5, 0, n, dim)
    q = orc.synth_rows(6, 0, 1, dim)[0]
    rng = np.random.default_rng(3)
    # rows 12,000 .. 12,399: the query plus a little noise (the first 150 of them exact copies of each other)
    near = q[None, :] + 0.01 * rng.standard_normal((400, dim)).astype(np.float32)
    near[:150] = near[0]
    rows[12000:12400] = near
    rows[30000:30050] = q  # and 50 exact copies of the query elsewhere
    scale = orc.compute_int8_scale(rows)
    ix = _index(pvs, dt, rows, scale)
    hc = _host(dt, rows, scale)
    hq = orc.quantize_int8(q, scale) if dt == pvs.I8 else q
    for m in (pvs.COSINE, pvs.L2):
        for k in (1, 64, 65, 200, 256):
            ei, ed = orc.search(dt, m, hc, hq, k, threads=4)
            before = _direct_searches(pvs)
            got = ix.search(q, k, m)
            assert _direct_searches(pvs) == before + 1
            _same(got, ei[0], ed[0], k)
    assert ix.stats().dense_queries == 0
    ix.close()


def test_direct_search_pages_that_end_in_null_rows(pvs):
    """Zero vectors have a NULL cosine distance: they sort last, in id order.  A page the finite distances cannot fill is
    completed from the index's NULL list (no dense query); k > rows returns every row; a zero query makes every distance NULL."""
    n, dim = 300, 64
    rows = orc.synth_rows(9, 0, n, dim)
    rows[[3, 50, 51, 299]] = 0.0
    ix = _index(pvs, pvs.F32, rows, None)
    q = orc.synth_rows(10, 0, 1, dim)[0]
    for k in (10, 296, 297, 300, 256):
        ei, ed = orc.search(orc.F32, orc.COSINE, rows, q, k)
        gi, gd, gc = ix.search(q, k, pvs.COSINE)
        kk = min(k, n)
        assert gc[0] == kk
        assert np.array_equal(gi[0, :kk], ei[0, :kk]), k
        assert np.array_equal(np.isnan(gd[0, :kk]), np.isnan(ed[0, :kk]))
        fin = ~np.isnan(ed[0, :kk])
        assert np.array_equal(gd[0, :kk][fin].view(np.uint32), ed[0, :kk][fin].view(np.uint32))
    zi, zd, zc = ix.search(np.zeros(dim, np.float32), 5, pvs.COSINE)
    assert zc[0] == 5 and zi[0, :5].tolist() == [0, 1, 2, 3, 4] and np.isnan(zd[0, :5]).all()
    assert ix.stats().dense_queries == 0
    # L2 has no NULL rows here: the same index answers in full
    ei, ed = orc.search(orc.F32, orc.L2, rows, q, 300)
    gi, gd, gc = ix.search(q, 256, pvs.L2)
    assert gc[0] == 256 and np.array_equal(gi[0], ei[0, :256]) and np.array_equal(gd[0].view(np.uint32), ed[0, :256].view(np.uint32))
    ix.close()


def test_direct_search_honours_the_second_sort_key(pvs):
    """Ties on the distance come out by order key DESC, then id (pql/model.rs:547-553), a tie at the k-th distance takes the
    newest rows: int8 L2 over few distinct vectors."""
    rng = np.random.default_rng(21)
    dim, distinct, copies = 96, 200, 30
    base = orc.synth_rows(77, 0, distinct, dim)
    rows = np.tile(base, (copies, 1))[rng.permutation(distinct * copies)]
    n = len(rows)
    ids = np.arange(n, dtype=np.int64) * 2 + 5
    keys = rng.integers(0, 25, n).astype(np.int64) + 1_700_000_000
    scale = orc.compute_int8_scale(rows)
    ix = _index(pvs, pvs.I8, rows, scale, ids)
    ix.set_order_keys(keys)
    corpus = orc.quantize_int8(rows, scale)
    q = base[5] + 0.02 * orc.synth_rows(78, 0, 1, dim)[0]
    hq = orc.quantize_int8(q, scale)
    for metric, om in ((pvs.L2, orc.L2), (pvs.COSINE, orc.COSINE)):
        d = orc.score_all(orc.I8, om, corpus, hq)
        for k in (1, 29, 30, 31, 100, 256):
            ei, ed = orc.topk_ordered(d, k, ids, keys)
            before = _direct_searches(pvs)
            gi, gd, gc = ix.search(q, k, metric)
            assert _direct_searches(pvs) == before + 1
            assert gc[0] == k and np.array_equal(gi[0, :k], ei), (metric, k)
            assert np.array_equal(gd[0, :k].view(np.uint32), ed.view(np.uint32))
    ix.close()


def test_direct_search_is_not_taken_where_it_does_not_apply(pvs):
    """More than eight queries, pages beyond 256 rows, batches whose lists do not fit the LDS and a corpus above the crossover stay
    on the filter scan."""
    n, dim = 5000, 128
    rows = orc.synth_rows(31, 0, n, dim)
    ix = _index(pvs, pvs.F32, rows, None)
    q = orc.synth_rows(32, 0, 9, dim)
    before = _direct_searches(pvs)
    ix.search(q, 10, pvs.COSINE)                     # nine queries
    ix.search(q[:5], 10, pvs.COSINE)                 # five f32 queries: float rows take four per launch
    ix.search(q[:4], 200, pvs.COSINE)                # four pages of 200 rows: the wave lists do not fit
    ix.search(q[0], 257, pvs.COSINE)                 # k > 256
    pvs.debug_set("direct_max_mb", 1)                # crossover below this corpus (2.5 MB)
    try:
        ix.search(q[0], 10, pvs.COSINE)
    finally:
        pvs.debug_set("direct_max_mb", 0)
    pvs.debug_set("direct_max_nq", 1)                # round 4's form: single queries only
    try:
        ix.search(q[:2], 10, pvs.COSINE)
    finally:
        pvs.debug_set("direct_max_nq", 0)
    assert _direct_searches(pvs) == before
    ei, ed = orc.search(orc.F32, orc.COSINE, rows, q[0], 10)
    gi, gd, gc = ix.search(q[0], 10, pvs.COSINE)
    assert _direct_searches(pvs) == before + 1 and np.array_equal(gi[0], ei[0])
    ix.search(q[:2], 10, pvs.COSINE)
    assert _direct_searches(pvs) == before + 3
    ix.close()


from routes import lists_fit as _lists_fit  # noqa: E402  (mirror of plan_capw, csrc/pvs_direct.hip)


BATCH_SHAPES = [
    # dtype, metric, n, dim, k, batch
    ("i8", "cosine", 30011, 768, 10, 2),
    ("i8", "l2", 30011, 768, 100, 3),
    ("i8", "cosine", 69001, 768, 10, 4),
    ("i8", "l2", 20000, 768, 30, 8),
    ("i8", "cosine", 5000, 512, 32, 7),
    ("i8", "cosine", 100, 96, 10, 8),       # fewer workgroups than queries: finalisers take several queries each
    ("i8", "l2", 1, 33, 1, 5),
    ("i8", "cosine", 4099, 1536, 64, 4),
    ("f16", "cosine", 20000, 768, 10, 4),
    ("f16", "l2", 9000, 1152, 17, 2),
    ("f16", "cosine", 129, 64, 60, 3),
    ("f32", "cosine", 10000, 512, 10, 4),   # BASELINE configs[0] x 4
    ("f32", "l2", 4097, 768, 64, 4),
    ("f32", "cosine", 3000, 1536, 33, 2),
    ("f32", "l2", 63, 300, 7, 3),
    ("f32", "cosine", 30000, 384, 200, 2),
]


@pytest.mark.parametrize("dtype,metric,n,dim,k,batch", BATCH_SHAPES)
def test_few_query_direct_search_matches_oracle_and_filter_scan(pvs, dtype, metric, n, dim, k, batch):
    """2..8 queries share ONE launch (a PQL `or` of a few vector filters, pql/builder.rs:638-661; coalesced callers): every page
    bit for bit the oracle's and the filter scan's, through the host entry point and the stream-ordered one."""
    dt = {"i8": pvs.I8, "f16": pvs.F16, "f32": pvs.F32}[dtype]
    m = pvs.COSINE if metric == "cosine" else pvs.L2
    assert _lists_fit(dtype, dim, k, batch)
    rows = orc.synth_rows(170 + dim, 0, n, dim)
    queries = orc.synth_rows(0x5EED0200, 0, batch, dim)
    if n > 100:
        queries[batch - 1] = rows[n // 2]  # a stored row as a query: distance 0 at the top
    scale = orc.compute_int8_scale(rows)
    ids = np.arange(n, dtype=np.int64) * 3 + 11
    ix = _index(pvs, dt, rows, scale, ids)
    hc = _host(dt, rows, scale)
    hq = orc.quantize_int8(queries, scale) if dt == pvs.I8 else queries
    k_eff = min(k, n)
    ei, ed = orc.search(dt, m, hc, hq, k, ids=ids, threads=4)
    before, dense_before = _direct_searches(pvs), ix.stats().dense_queries
    gi, gd, gc = ix.search(queries, k, m)
    assert _direct_searches(pvs) == before + batch, "a few queries over a small corpus share the one-launch search"
    assert ix.stats().dense_queries == dense_before
    for j in range(batch):
        _same((gi[j:j + 1], gd[j:j + 1], gc[j:j + 1]), ei[j], ed[j], k_eff)
    # the stream-ordered entry point (device buffers in and out)
    dq = pvs.DeviceBuffer.from_numpy(np.ascontiguousarray(hq))
    d_out = [pvs.DeviceBuffer(batch * k * 8), pvs.DeviceBuffer(batch * k * 4), pvs.DeviceBuffer(batch * 4)]
    ix.wait(ix.search_device(dq, pvs.I8 if dt == pvs.I8 else pvs.F32, batch, k, m, *d_out))
    si, sd, sc = d_out[0].to_numpy(np.int64, (batch, k)), d_out[1].to_numpy(np.float32, (batch, k)), d_out[2].to_numpy(np.uint32, (batch,))
    assert _direct_searches(pvs) == before + 2 * batch
    for j in range(batch):
        _same((si[j:j + 1], sd[j:j + 1], sc[j:j + 1]), ei[j], ed[j], k_eff)
    for b in [dq] + d_out:
        b.free()
    pvs.debug_set("no_direct_topk", 1)
    try:
        ri, rd, rc = ix.search(queries, k, m)
    finally:
        pvs.debug_set("no_direct_topk", 0)
    assert _direct_searches(pvs) == before + 2 * batch
    for j in range(batch):
        _same((ri[j:j + 1], rd[j:j + 1], rc[j:j + 1]), ei[j], ed[j], k_eff)
    ix.close()


def test_few_query_direct_search_with_null_pages_keys_masks_and_a_zero_query(pvs):
    """A batch in which one query is zero (every cosine distance NULL), pages that end in NULL rows, the second sort key, a
    candidate mask — per query what the single-query search says."""
    rng = np.random.default_rng(77)
    n, dim = 40000, 200
    rows = orc.synth_rows(410, 0, n, dim)
    rows[[7, 8, 9000]] = 0.0
    rows[20000:20040] = rows[19999]
    ids = np.arange(n, dtype=np.int64) * 5 + 1
    keys = rng.integers(0, 6, n).astype(np.int64) + 1_700_000_000
    q = orc.synth_rows(420, 0, 4, dim)
    q[2] = 0.0
    q[3] = rows[19999]
    for dtype in ("i8", "f32"):
        dt = {"i8": pvs.I8, "f32": pvs.F32}[dtype]
        scale = orc.compute_int8_scale(rows)
        ix = _index(pvs, dt, rows, scale, ids)
        for keyed in (False, True):
            ix.set_order_keys(keys if keyed else None)
            for m in (pvs.COSINE, pvs.L2):
                for k in (10, 64):
                    before, dense_before = _direct_searches(pvs), ix.stats().dense_queries
                    gi, gd, gc = ix.search(q, k, m)
                    assert _direct_searches(pvs) == before + 4
                    for j in range(4):
                        oi, od, oc = ix.search(q[j], k, m)
                        assert gc[j] == oc[0] and np.array_equal(gi[j], oi[0]), (dtype, keyed, m, k, j)
                        assert np.array_equal(gd[j].view(np.uint32), od[0].view(np.uint32))
                    mask = (rng.random(n) < 0.6).astype(np.uint8)
                    mask[[7, 8]] = 1
                    fi, fd, fc = ix.search_filtered(q, k, mask, m)
                    for j in range(4):
                        oi, od, oc = ix.search_filtered(q[j], k, mask, m)
                        assert fc[j] == oc[0] and np.array_equal(fi[j], oi[0]), (dtype, keyed, m, k, j, "masked")
                        assert np.array_equal(fd[j].view(np.uint32), od[0].view(np.uint32))
                    assert ix.stats().dense_queries == dense_before
        # a mask that leaves fewer rows than the page: the page ends where the allowed rows end, no dense pass
        few = np.zeros(n, np.uint8)
        few[rng.choice(n, 17000, replace=False)] = 1
        few[[7, 8, 9000]] = 1
        dense_before, tails_before = ix.stats().dense_queries, ix.stats().null_tail_queries
        short = np.zeros(n, np.uint8)
        short[100:130] = 1  # 30 allowed rows, k = 64 (sparse route) ... and through the scan:
        pvs.debug_set("no_sparse", 1)
        try:
            si, sd, sc = ix.search_filtered(q[0], 64, short, pvs.L2)
        finally:
            pvs.debug_set("no_sparse", 0)
        hc = _host(dt, rows, scale)
        hq0 = orc.quantize_int8(q[0], scale) if dt == pvs.I8 else q[0]
        allowed = np.nonzero(short)[0]
        ei, ed = orc.search(dt, orc.L2, hc[allowed], hq0, 64, ids=ids[allowed])
        assert sc[0] == 30 and np.array_equal(si[0, :30], ei[0]) and np.array_equal(sd[0, :30].view(np.uint32), ed[0].view(np.uint32))
        assert ix.stats().dense_queries == dense_before and ix.stats().null_tail_queries == tails_before, "fewer allowed rows than k is a complete page, not a NULL tail"
        ix.close()


def test_few_query_direct_search_one_query_beyond_the_closed_form(pvs):
    """int8: a batch in which ONE query's sums leave the closed form's range goes to the dense path for that query only."""
    rng = np.random.default_rng(1101)
    n, dim = 900, 1100
    hc = rng.integers(-20, 21, size=(n, dim)).astype(np.int8)
    hq = rng.integers(-20, 21, size=(3, dim)).astype(np.int8)
    hq[1] = rng.choice(np.array([-128, 127], np.int8), size=dim)  # |q|^2 = 1,100 x 127^2 > 2^24
    ix = pvs.VectorIndex(pvs.I8, dim)
    ix.set_scale(1.0)
    ix.add(hc)
    for m in (pvs.COSINE, pvs.L2):
        ei, ed = orc.search(orc.I8, m, hc, hq, 10)
        before, dense_before = _direct_searches(pvs), ix.stats().dense_queries
        gi, gd, gc = ix.search(hq, 10, m)
        assert _direct_searches(pvs) == before + 3 and ix.stats().dense_queries == dense_before + 1
        for j in range(3):
            _same((gi[j:j + 1], gd[j:j + 1], gc[j:j + 1]), ei[j], ed[j], 10)
    ix.close()


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("PVS_FUZZ_SEEDS", "30"))))
def test_randomized_few_query_batches_against_the_oracle(pvs, seed):
    """Seeded sweep over 2..8-query batches: element type / metric / rows / dim / k / batch, ties, zero rows, an order key on every
    other seed; whatever route the library picks (one launch where the lists fit, the filter scan otherwise), every page is the
    oracle's."""
    rng = np.random.default_rng(51000 + seed)
    dtype = ["i8", "f16", "f32"][seed % 3]
    dt = {"i8": pvs.I8, "f16": pvs.F16, "f32": pvs.F32}[dtype]
    m = [pvs.COSINE, pvs.L2][(seed // 3) % 2]
    n = int(rng.choice([2, 63, 64, 65, 129, 1000, 4097, 20000, 70001, 300000]))
    dim = int(rng.choice([3, 17, 64, 100, 257, 384, 768, 1000, 1536]))
    k = int(rng.choice([1, 2, 10, 32, 33, 64, 100]))
    batch = int(rng.integers(2, 9))
    rows = orc.synth_rows(52000 + seed, 0, n, dim)
    if n > 8:
        rows[rng.integers(0, n, 5)] = rows[rng.integers(0, n, 5)]
        rows[int(rng.integers(0, n))] = 0.0
        run = int(rng.integers(0, n - 4))
        rows[run:run + 4] = rows[run]
    q = orc.synth_rows(53000 + seed, 0, batch, dim)
    if n > 8:
        q[int(rng.integers(0, batch))] = rows[int(rng.integers(0, n))]
    scale = orc.compute_int8_scale(rows)
    ids = np.cumsum(rng.integers(1, 4, n)).astype(np.int64)
    ix = _index(pvs, dt, rows, scale, ids)
    keys = None
    if seed % 2 == 1:
        keys = rng.integers(0, 6, n).astype(np.int64) + 1_700_000_000
        ix.set_order_keys(keys)
    hc = _host(dt, rows, scale)
    hq = orc.quantize_int8(q, scale) if dt == pvs.I8 else q
    om = orc.COSINE if m == pvs.COSINE else orc.L2
    before = _direct_searches(pvs)
    gi, gd, gc = ix.search(q, k, m)
    assert _direct_searches(pvs) == before + (batch if _lists_fit(dtype, dim, k, batch) else 0), (dtype, dim, k, batch)
    kk = min(k, n)
    for j in range(batch):
        d = orc.score_all(dt, om, hc, hq[j])
        ei, ed = orc.topk_ordered(d, k, ids, keys if keys is not None else np.zeros(n, np.int64))
        assert gc[j] == kk, (seed, j, gc[j], kk)
        assert np.array_equal(gi[j, :kk], ei[:kk]), (seed, n, dim, k, batch, j)
        assert np.array_equal(np.isnan(gd[j, :kk]), np.isnan(ed[:kk]))
        fin = ~np.isnan(ed[:kk])
        assert np.array_equal(gd[j, :kk][fin].view(np.uint32), ed[:kk][fin].view(np.uint32)), (seed, n, dim, k, batch, j)
    ix.close()


def test_direct_search_int8_sums_beyond_the_closed_form(pvs):
    """int8 rows are scored by integer dot products and the closed form of the reference's f32 chain, valid while the sums stay
    below 2^24 (dim * 127^2 at most): saturated codes at 1,100 dimensions leave that range — the kernel says so and the dense
    path (in-order f32 chains) answers, bit for bit the oracle."""
    rng = np.random.default_rng(1100)
    n, dim = 900, 1100
    hc = rng.choice(np.array([-128, -127, 126, 127], np.int8), size=(n, dim))
    hq = rng.choice(np.array([-128, 127], np.int8), size=(1, dim))
    ix = pvs.VectorIndex(pvs.I8, dim)
    ix.set_scale(1.0)
    ix.add(hc)
    for m in (pvs.COSINE, pvs.L2):
        ei, ed = orc.search(orc.I8, m, hc, hq, 10)
        before, dense_before = _direct_searches(pvs), ix.stats().dense_queries
        got = ix.search(hq, 10, m)
        assert _direct_searches(pvs) == before + 1 and ix.stats().dense_queries == dense_before + 1
        _same(got, ei[0], ed[0], 10)
    # ordinary codes at the same width stay on the one-launch search
    rows = orc.synth_rows(3, 0, n, dim)
    scale = orc.compute_int8_scale(rows)
    ix2 = _index(pvs, pvs.I8, rows, scale)
    q = orc.synth_rows(4, 0, 1, dim)[0]
    ei, ed = orc.search(orc.I8, orc.L2, orc.quantize_int8(rows, scale), orc.quantize_int8(q, scale), 10)
    got = ix2.search(q, 10, pvs.L2)
    _same(got, ei[0], ed[0], 10)
    assert ix2.stats().dense_queries == 0
    ix.close()
    ix2.close()


@pytest.mark.parametrize("dtype", ["i8", "f16", "f32"])
def test_direct_search_with_a_candidate_mask(pvs, dtype):
    """pvs_search_filtered with a mask too dense for the gather path: the one-launch search skips the rows outside the mask; a page
    the allowed rows cannot fill ends where they end (NULL rows of the mask first, in id order)."""
    dt = {"i8": pvs.I8, "f16": pvs.F16, "f32": pvs.F32}[dtype]
    n, dim = 60000, 200  # (masks that leave more than 16,384 rows: fewer go to the gather path, pvs_sparse.hip)
    rows = orc.synth_rows(41, 0, n, dim)
    rows[[7, 8, 9000]] = 0.0
    scale = orc.compute_int8_scale(rows)
    ids = np.arange(n, dtype=np.int64) * 5 + 1
    ix = _index(pvs, dt, rows, scale, ids)
    hc = _host(dt, rows, scale)
    q = orc.synth_rows(42, 0, 1, dim)[0]
    hq = orc.quantize_int8(q, scale) if dt == pvs.I8 else q
    rng = np.random.default_rng(8)
    for frac, k in ((0.4, 10), (0.4, 256), (0.9, 100)):
        mask = (rng.random(n) < frac).astype(np.uint8)
        mask[7] = 1
        allowed = np.nonzero(mask)[0]
        for m in (pvs.COSINE, pvs.L2):
            ei, ed = orc.search(dt, m, hc[allowed], hq, k, ids=ids[allowed], threads=4)
            before = _direct_searches(pvs)
            gi, gd, gc = ix.search_filtered(q, k, mask, m)
            assert _direct_searches(pvs) == before + 1
            assert gc[0] == k and np.array_equal(gi[0, :k], ei[0])
            assert np.array_equal(gd[0, :k].view(np.uint32), ed[0].view(np.uint32))
    # the zero vectors (NULL cosine distance) are allowed and the page reaches them only if it is long enough: same page as the oracle's either way
    mask = np.zeros(n, np.uint8)
    mask[rng.choice(n, 17000, replace=False)] = 1  # (dense enough to stay off the gather path)
    mask[[7, 8, 9000]] = 1
    allowed = np.nonzero(mask)[0]
    ei, ed = orc.search(dt, orc.COSINE, hc[allowed], hq, 256, ids=ids[allowed], threads=4)
    gi, gd, gc = ix.search_filtered(q, 256, mask, pvs.COSINE)
    assert gc[0] == 256 and np.array_equal(gi[0], ei[0]) and np.array_equal(gd[0].view(np.uint32), ed[0].view(np.uint32))
    ix.close()


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("PVS_FUZZ_SEEDS", "30"))))
def test_randomized_single_queries_against_the_oracle(pvs, seed):
    """Seeded sweep over what the one-launch search sees: element type / metric / rows / dim / k, duplicated rows (ties), zero rows
    (NULL cosine distance), huge and tiny components, an order key on every other seed, a candidate mask on every third."""
    rng = np.random.default_rng(31000 + seed)
    dt = [pvs.I8, pvs.F16, pvs.F32][seed % 3]
    m = [pvs.COSINE, pvs.L2][(seed // 3) % 2]
    n = int(rng.choice([1, 2, 63, 64, 65, 129, 1000, 4097, 20000, 70001]))
    dim = int(rng.choice([1, 3, 17, 64, 65, 100, 257, 384, 768, 1000, 1100, 1536]))
    k = int(rng.choice([1, 2, 10, 64, 65, 100, 256]))
    rows = orc.synth_rows(32000 + seed, 0, n, dim) if dim > 1 else rng.standard_normal((n, 1)).astype(np.float32)
    if n > 8:
        src = rng.integers(0, n, 5)
        rows[rng.integers(0, n, 5)] = rows[src]
        rows[int(rng.integers(0, n))] = 0.0
        run = int(rng.integers(0, n - 4))
        rows[run:run + 4] = rows[run]  # a short run of exact copies stored side by side
    if dt != pvs.I8 and n > 4:
        rows[int(rng.integers(0, n))] *= np.float32(300.0 if dt == pvs.F16 else 1e18)
        rows[int(rng.integers(0, n))] *= np.float32(1e-3 if dt == pvs.F16 else 1e-18)
    q = orc.synth_rows(33000 + seed, 0, 1, dim)[0] if dim > 1 else rng.standard_normal(1).astype(np.float32)
    if n > 8 and seed % 4 == 1:
        q = rows[int(rng.integers(0, n))].copy()  # the query is a stored row: distance 0 (or a tiny negative one) at the top
    scale = orc.compute_int8_scale(rows)
    ids = np.cumsum(rng.integers(1, 4, n)).astype(np.int64)
    ix = _index(pvs, dt, rows, scale, ids)
    keys = None
    if seed % 2 == 1:
        keys = rng.integers(0, 6, n).astype(np.int64) + 1_700_000_000
        ix.set_order_keys(keys)
    hc = _host(dt, rows, scale)
    hq = orc.quantize_int8(q, scale) if dt == pvs.I8 else q
    om = orc.COSINE if m == pvs.COSINE else orc.L2
    d = orc.score_all(dt, om, hc, hq)
    allowed = np.arange(n)
    mask = None
    if seed % 3 == 2 and n > 100:
        mask = (rng.random(n) < 0.7).astype(np.uint8)
        allowed = np.nonzero(mask)[0]
    ei, ed = orc.topk_ordered(d[allowed], k, ids[allowed], keys[allowed] if keys is not None else np.zeros(len(allowed), np.int64))
    kk = min(k, len(allowed))
    gi, gd, gc = ix.search(q, k, m) if mask is None else ix.search_filtered(q, k, mask, m)
    assert gc[0] == kk, (gc, kk)
    assert np.array_equal(gi[0, :kk], ei[:kk]), (seed, n, dim, k)
    assert np.array_equal(np.isnan(gd[0, :kk]), np.isnan(ed[:kk]))
    fin = ~np.isnan(ed[:kk])
    assert np.array_equal(gd[0, :kk][fin].view(np.uint32), ed[:kk][fin].view(np.uint32)), (seed, n, dim, k)
    ix.close()
