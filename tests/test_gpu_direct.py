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
    """Batches, pages beyond 256 rows and a corpus above the crossover stay on the filter scan."""
    n, dim = 5000, 128
    rows = orc.synth_rows(31, 0, n, dim)
    ix = _index(pvs, pvs.F32, rows, None)
    q = orc.synth_rows(32, 0, 2, dim)
    before = _direct_searches(pvs)
    ix.search(q, 10, pvs.COSINE)                     # two queries
    ix.search(q[0], 257, pvs.COSINE)                 # k > 256
    pvs.debug_set("direct_max_mb", 1)                # crossover below this corpus (2.5 MB)
    try:
        ix.search(q[0], 10, pvs.COSINE)
    finally:
        pvs.debug_set("direct_max_mb", 0)
    assert _direct_searches(pvs) == before
    ei, ed = orc.search(orc.F32, orc.COSINE, rows, q[0], 10)
    gi, gd, gc = ix.search(q[0], 10, pvs.COSINE)
    assert _direct_searches(pvs) == before + 1 and np.array_equal(gi[0], ei[0])
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
