"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same
seeded inputs.  Bit-exact for int8 distances and ids; the float paths reproduce the
oracle's sequential-f32 evaluation, so they are compared bit-exact as well (the
north-star tolerance of 1e-5 relative is asserted as the fallback bar)."""
import os

import numpy as np
import pytest

import oracle as orc

pytestmark = pytest.mark.gpu

REL_TOL = 1e-5  # BASELINE.json north_star: fp32/fp16 cosine within 1e-5 relative


@pytest.fixture(scope="module")
def pvs():
    import panoptikon_amd as p

    if p.device_count() < 1:
        pytest.fail("no gfx950 device visible: the gpu tests need an MI355X")
    return p


def unit_rows(seed, n, dim):
    return orc.synth_rows(seed, 0, n, dim)


def make_index(pvs, dtype, rows_f32, scale=None):
    n, dim = rows_f32.shape
    ix = pvs.VectorIndex(dtype, dim)
    if dtype == pvs.I8:
        ix.set_scale(scale)
    ix.add_f32(rows_f32)
    return ix


def host_corpus(dtype, rows_f32, scale=None):
    if dtype == orc.I8:
        return orc.quantize_int8(rows_f32, scale)
    if dtype == orc.F16:
        return rows_f32.astype(np.float16)
    return rows_f32


def assert_same_page(got, exp, exact=True):
    gi, gd, gc = got
    ei, ed = exp
    assert gc.tolist() == [ei.shape[1]] * ei.shape[0]
    assert np.array_equal(gi[:, : ei.shape[1]], ei), "returned item indices differ from the reference ordering"
    g = gd[:, : ei.shape[1]]
    if exact:
        assert np.array_equal(g.view(np.uint32), ed.view(np.uint32)), "distances are not bit-identical"
    else:
        assert np.all(np.abs(g - ed) <= REL_TOL * np.abs(ed) + 1e-7)


# ------------------------------------------------------------------- codec
def test_codec_on_device_matches_reference_kats(pvs):
    q = pvs.quantize_int8(np.array([0.5, 1.5, 2.5, -0.5, -1.5, -2.5, 2.4999, -2.4999], np.float32), 1.0)
    assert q.tolist() == [0, 2, 2, 0, -2, -2, 2, -2]
    s = pvs.scale_from_absmax(11.0)
    assert pvs.quantize_int8(np.array([11.0, -11.0, 1000.0, -1000.0], np.float32), s).tolist() == [127, -127, 127, -128]
    assert pvs.quantize_int8(np.array([np.nan, np.inf, -np.inf], np.float32), 0.5).tolist() == [0, 127, -128]
    rng = np.random.default_rng(5)
    x = (rng.standard_normal(100_003) * 0.05).astype(np.float32)
    x[17] = np.nan
    am = pvs.absmax(x)
    assert am == orc.blob_absmax(x)
    sc = pvs.scale_from_absmax(am)
    assert np.array_equal(pvs.quantize_int8(x, sc), orc.quantize_int8(x, sc))
    assert pvs.absmax(np.array([1.0, -np.inf], np.float32)) == np.inf
    assert pvs.absmax(np.zeros(0, np.float32)) == 0.0


def test_synth_rows_device_bytes_equal_oracle(pvs):
    n, dim = 257, 768
    buf = pvs.DeviceBuffer(n * dim * 4)
    pvs._lib.check(pvs.lib().pvs_synth_rows_f32(-1, 20260928, 1000, n, dim, buf.ptr))
    dev = buf.to_numpy(np.float32, (n, dim))
    assert np.array_equal(dev.view(np.uint32), orc.synth_rows(20260928, 1000, n, dim).view(np.uint32))


def test_clustered_rows_device_bytes_equal_oracle_and_pages_stay_exact(pvs):
    """The realistic-distribution generator (round 6: clusters with power-law sizes, anisotropic noise, near-duplicate runs, exact
    duplicates): device bytes == oracle bytes at several offsets (duplicates reach back across chunk boundaries), and an int8
    index over such rows answers batches and single queries like the oracle — the exact duplicates tie to the last bit and are
    ordered by id."""
    dim = 768
    for row0, n in ((0, 300), (12_345, 257), ((1 << 40) + 5, 64)):
        buf = pvs.DeviceBuffer(n * dim * 4)
        pvs._lib.check(pvs.lib().pvs_synth_rows_clustered_f32(-1, 20260928, row0, n, dim, buf.ptr))
        assert np.array_equal(buf.to_numpy(np.float32, (n, dim)).view(np.uint32), orc.synth_rows_clustered(20260928, row0, n, dim).view(np.uint32)), row0
    n, dim = 300_000, 128
    rows = orc.synth_rows_clustered(7, 0, n, dim)
    assert len(np.unique(rows[:20_000], axis=0)) < 20_000  # exact duplicates are there
    scale = orc.compute_int8_scale(rows)
    ix = pvs.VectorIndex(pvs.I8, dim)
    ix.set_scale(scale)
    ix.add_f32(rows)
    codes = orc.quantize_int8(rows, scale)
    q = orc.synth_rows_clustered(7, 1 << 40, 40, dim)
    q[1] = rows[4242]
    qc = orc.quantize_int8(q, scale)
    for metric, om in ((pvs.COSINE, orc.COSINE), (pvs.L2, orc.L2)):
        gi, gd, gc = ix.search(q, 100, metric)
        ei, ed = orc.search(orc.I8, om, codes, qc, 100, threads=8)
        assert np.array_equal(gi[:, :100], ei) and np.array_equal(gd[:, :100].view(np.uint32), ed.view(np.uint32)), metric
        g1, d1, c1 = ix.search(q[1], 10, metric)
        assert np.array_equal(g1[0, :10], ei[1, :10]) and np.array_equal(d1[0, :10].view(np.uint32), ed[1, :10].view(np.uint32))
    ix.close()


# --------------------------------------------------------------- score_all
@pytest.mark.parametrize("dtype", ["i8", "f16", "f32"])
@pytest.mark.parametrize("metric", ["cosine", "l2"])
@pytest.mark.parametrize("dim", [8, 100, 768])
def test_score_all_bit_exact(pvs, dtype, metric, dim):
    dt = {"i8": pvs.I8, "f16": pvs.F16, "f32": pvs.F32}[dtype]
    m = pvs.COSINE if metric == "cosine" else pvs.L2
    rows = unit_rows(11, 1500, dim)
    rows[7] = 0.0  # zero-norm row: cosine NaN (SQL NULL)
    q = unit_rows(99, 1, dim)[0]
    scale = orc.compute_int8_scale(rows)
    ix = make_index(pvs, dt, rows, scale)
    hc = host_corpus(dt, rows, scale)
    hq = orc.quantize_int8(q, scale) if dt == pvs.I8 else q
    exp = orc.score_all(dt, m, hc, hq)
    got = ix.score_all(hq, m)
    assert np.array_equal(np.isnan(got), np.isnan(exp))
    ok = ~np.isnan(exp)
    assert np.array_equal(got[ok].view(np.uint32), exp[ok].view(np.uint32))
    if dt == pvs.I8:  # f32 query quantized on the device == host-side compute_query_quant
        got2 = ix.score_all(q, m)
        assert np.array_equal(got2[ok].view(np.uint32), exp[ok].view(np.uint32))
    ix.close()


# ------------------------------------------------------------------ search
CASES = [
    # dtype, metric, n, dim, batch, k
    ("i8", "cosine", 20000, 768, 5, 10),
    ("i8", "cosine", 30011, 768, 128, 100),
    ("i8", "l2", 20000, 768, 40, 100),
    ("i8", "cosine", 9000, 512, 64, 10),
    ("i8", "l2", 5000, 1024, 33, 50),
    ("i8", "cosine", 30011, 768, 256, 100),  # 256-query pass: two query groups per wave
    ("i8", "l2", 9000, 1024, 200, 20),
    ("i8", "cosine", 7000, 100, 129, 10),
    ("i8", "l2", 6000, 512, 600, 5),  # 256 + 256 + 88
    ("i8", "cosine", 30011, 768, 3, 4096),  # the reference's largest prefetch (api/search.rs:51) on the filter path
    ("f16", "l2", 20000, 512, 2, 2500),
    ("f16", "cosine", 20000, 768, 32, 100),
    ("f16", "l2", 12000, 768, 7, 10),
    ("f16", "cosine", 8000, 512, 70, 100),
    ("f16", "cosine", 6000, 1024, 128, 20),
    ("f32", "cosine", 10000, 512, 1, 10),  # BASELINE configs[0]
    ("f32", "l2", 4000, 768, 3, 100),
    ("f32", "cosine", 20000, 768, 128, 100),
    ("f32", "l2", 9000, 1024, 40, 10),
    ("f32", "cosine", 7000, 256, 64, 50),
    ("f32", "l2", 6000, 384, 33, 20),
    ("f32", "cosine", 5000, 128, 5, 10),
    # the wider models: SigLIP so400m (1152), OpenCLIP bigG (1280), 1536-d text embeddings
    ("i8", "cosine", 9000, 1152, 128, 50),
    ("i8", "l2", 7000, 1536, 33, 20),
    ("i8", "cosine", 6000, 1280, 200, 10),
    ("f16", "cosine", 8000, 1152, 32, 50),
    ("f16", "l2", 6000, 1280, 128, 20),
    ("f16", "cosine", 5000, 1536, 5, 10),
    ("f32", "cosine", 6000, 1152, 32, 50),
    ("f32", "l2", 5000, 1280, 8, 20),
    ("f32", "cosine", 4000, 1536, 128, 10),
    # pitches without an instance of their own are padded up to the next one (640-d f16 -> 1536 B, 320-d f32 -> 1536 B)
    ("f16", "cosine", 6000, 640, 16, 20),
    ("f32", "l2", 5000, 320, 16, 20),
    ("f16", "l2", 4000, 1100, 8, 10),
    ("i8", "cosine", 5000, 2048, 64, 20),
    ("i8", "l2", 4000, 3072, 16, 10),
    ("i8", "cosine", 3000, 1700, 5, 10),  # 1792 B -> 2048 B
]


@pytest.mark.parametrize("dtype,metric,n,dim,batch,k", CASES)
@pytest.mark.parametrize("path", ["auto", "dense"])
def test_search_matches_oracle(pvs, dtype, metric, n, dim, batch, k, path):
    if path == "dense" and (n > 12000 or batch > 8):
        pytest.skip("dense path covered on the smaller cases")
    dt = {"i8": pvs.I8, "f16": pvs.F16, "f32": pvs.F32}[dtype]
    m = pvs.COSINE if metric == "cosine" else pvs.L2
    rows = unit_rows(3, n, dim)
    queries = orc.synth_rows(0x5EED0000, 0, batch, dim)
    scale = orc.compute_int8_scale(rows)
    ix = make_index(pvs, dt, rows, scale)
    ix.set_path(1 if path == "dense" else 0)
    hc = host_corpus(dt, rows, scale)
    hq = orc.quantize_int8(queries, scale) if dt == pvs.I8 else queries
    exp = orc.search(dt, m, hc, hq, k, threads=8)
    got = ix.search(queries, k, m)  # f32 queries: the int8 index quantizes them on the device
    assert_same_page(got, exp)
    st = ix.stats()
    if path == "auto":
        assert st.fast_queries == batch and st.dense_queries == 0, "filter-scan path must serve these shapes"
    if dt == pvs.I8:  # pre-quantized codes (QuantResolved.query_quant) give the same page
        assert_same_page(ix.search(hq, k, m), exp)
    if path == "auto" and batch == 1:  # a single query took the one-launch exact search (pvs_direct.hip): the filter scan as well
        pvs.debug_set("no_direct_topk", 1)
        try:
            assert_same_page(ix.search(queries, k, m), exp)
        finally:
            pvs.debug_set("no_direct_topk", 0)
        assert ix.stats().dense_queries == 0
    if path == "auto" and batch <= 128:
        # the LDS-light pass C (what a pipelined caller's search runs beside the next search's scan; every element type since round 4)
        pvs.debug_set("force_light_finalize", 1)
        try:
            assert_same_page(ix.search(queries, k, m), exp)
        finally:
            pvs.debug_set("force_light_finalize", 0)
    ix.close()


def test_disagreeing_vectors_fixture_order(pvs):
    # db/vector_quants.rs:3254-3276, :3324-3382: int8 reproduces the exact ordering
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
    vecs = np.array(vs, np.float32)
    query = np.ones(8, np.float32)
    scale = pvs.scale_from_absmax(pvs.absmax(vecs))
    assert scale == orc.compute_int8_scale(vecs)
    exact = make_index(pvs, pvs.F32, vecs)
    quant = make_index(pvs, pvs.I8, vecs, scale)
    ei, ed, ec = exact.search(query, 100, pvs.COSINE)
    qi, qd, qc = quant.search(query, 100, pvs.COSINE)
    assert ec[0] == 12 and qc[0] == 12
    assert ei[0, :12].tolist() == qi[0, :12].tolist()
    oi, od = orc.search(orc.F32, orc.COSINE, vecs, query, 100)
    assert ei[0, :12].tolist() == oi[0].tolist()
    assert np.array_equal(ed[0, :12].view(np.uint32), od[0].view(np.uint32))
    assert (qi[0, 12:] == -1).all() and np.isnan(qd[0, 12:]).all()
    # page walk (:3386-3415): k = 4, 8, 12 are prefixes of the single shot
    for k in (1, 4, 8, 12):
        ki, _, kc = quant.search(query, k, pvs.COSINE)
        assert kc[0] == k and ki[0, :k].tolist() == qi[0, :k].tolist()
    exact.close()
    quant.close()


def test_ties_duplicates_and_null_rows(pvs):
    # heavy ties (duplicated rows) and zero-norm rows (NULL distance, sorted last)
    base = unit_rows(21, 300, 768)
    rows = np.concatenate([base, base[:150], np.zeros((5, 768), np.float32), base[:50]])
    q = base[10] * 0.5 + base[11] * 0.5
    scale = orc.compute_int8_scale(rows)
    for dt in (pvs.I8, pvs.F16, pvs.F32):
        hc = host_corpus(dt, rows, scale)
        hq = orc.quantize_int8(q, scale) if dt == pvs.I8 else q
        ix = make_index(pvs, dt, rows, scale)
        for metric in (pvs.COSINE, pvs.L2):
            for k in (10, 505):
                exp = orc.search(dt, metric, hc, hq, k)
                gi, gd, gc = ix.search(hq, k, metric)
                assert gc[0] == exp[0].shape[1]
                assert np.array_equal(gi[0, : gc[0]], exp[0][0])
                a, b = gd[0, : gc[0]], exp[1][0]
                assert np.array_equal(np.isnan(a), np.isnan(b))
                assert np.array_equal(a[~np.isnan(a)].view(np.uint32), b[~np.isnan(b)].view(np.uint32))
        ix.close()


def test_row_ids_groups_and_errors(pvs):
    rows = unit_rows(5, 2000, 512)
    ids = np.arange(2000, dtype=np.int64) * 7 + 1000
    ix = pvs.VectorIndex(pvs.F16, 512)
    ix.add(rows[:1200].astype(np.float16), row_ids=ids[:1200])
    ix.add(rows[1200:].astype(np.float16), row_ids=ids[1200:])
    q = unit_rows(6, 2, 512)
    exp = orc.search(orc.F16, orc.COSINE, rows.astype(np.float16), q, 25, ids=ids)
    assert_same_page(ix.search(q, 25, pvs.COSINE), exp)
    with pytest.raises(pvs.PvsError) as e:
        ix.add(rows[:3].astype(np.float16), row_ids=[5, 6, 7])  # not increasing
    assert e.value.status == pvs._lib.ERR_INVALID_ARG
    with pytest.raises(pvs.PvsError) as e:
        ix.search(np.zeros(100, np.float32), 5)
    assert e.value.status == pvs._lib.ERR_DIM_MISMATCH
    with pytest.raises(pvs.PvsError) as e:
        ix.search(np.zeros((1, 512), np.int8), 5)  # element type mismatch
    assert e.value.status == pvs._lib.ERR_DIM_MISMATCH
    with pytest.raises(pvs.PvsError):
        ix.search(q, 0)  # k must be a positive integer (preprocess.rs:441-444)
    ix.close()
    i8 = pvs.VectorIndex(pvs.I8, 512)
    with pytest.raises(pvs.PvsError) as e:
        i8.add_f32(rows[:10])
    assert e.value.status == pvs._lib.ERR_STATE
    for bad in (b"", bytes(5), np.float32(0).tobytes(), np.float32(-1).tobytes(), np.float32("nan").tobytes()):
        with pytest.raises(pvs.PvsError):
            i8.set_scale_artifact(bad)
    i8.set_scale_artifact(pvs.scale_artifact(0.01))
    gi, gd, gc = i8.search(q, 5)  # empty index
    assert gc.tolist() == [0, 0] and (gi == -1).all()
    i8.close()


def test_device_merge_and_sharded_search(pvs):
    """The N>1 data path on one GPU: (a) the device merge kernel against the host merge and the
    oracle on synthetic multi-shard pages with ties and NULLs; (b) pvs_search_sharded through a
    1-rank RCCL communicator equals pvs_search."""
    import ctypes as C

    from panoptikon_amd import _lib as L

    rng = np.random.default_rng(9)
    world, batch, k = 8, 33, 100
    ids = np.full((world, batch, k), -1, np.int64)
    dist = np.full((world, batch, k), np.nan, np.float32)
    cnt = np.zeros((world, batch), np.uint32)
    for w in range(world):
        for q in range(batch):
            c = int(rng.integers(0, k + 1))
            d = np.round(rng.random(c), 2).astype(np.float32)
            if c > 3:
                d[-2:] = np.nan
            i = np.sort(rng.choice(50_000, c, replace=False)) + w * 50_000
            pi, pd = orc.topk(d, k, ids=i)
            ids[w, q, : len(pi)], dist[w, q, : len(pi)], cnt[w, q] = pi, pd, len(pi)
    hi, hd, hc = pvs.merge_topk(ids, dist, cnt, k)
    d_ids, d_dist, d_cnt = (pvs.DeviceBuffer.from_numpy(a) for a in (ids, dist, cnt))
    o_ids, o_dist, o_cnt = pvs.DeviceBuffer(batch * k * 8), pvs.DeviceBuffer(batch * k * 4), pvs.DeviceBuffer(batch * 4)
    L.check(pvs.lib().pvs_merge_topk_device(-1, d_ids.ptr, d_dist.ptr, d_cnt.ptr, world, batch, k, o_ids.ptr, o_dist.ptr, o_cnt.ptr))
    gi, gd, gc = o_ids.to_numpy(np.int64, (batch, k)), o_dist.to_numpy(np.float32, (batch, k)), o_cnt.to_numpy(np.uint32, (batch,))
    assert np.array_equal(gc, hc) and np.array_equal(gi, hi)
    assert np.array_equal(np.isnan(gd), np.isnan(hd)) and np.array_equal(gd[~np.isnan(gd)], hd[~np.isnan(hd)])

    rows = unit_rows(31, 9000, 768)
    scale = orc.compute_int8_scale(rows)
    ix = make_index(pvs, pvs.I8, rows, scale)
    queries = orc.synth_rows(0x5EED0000, 0, 40, 768)
    exp = ix.search(queries, 50, pvs.COSINE)
    uid = (C.c_uint8 * L.UNIQUE_ID_BYTES)()
    L.check(pvs.lib().pvs_comm_unique_id(uid))
    comm = C.c_void_p()
    L.check(pvs.lib().pvs_comm_create(uid, 1, 0, -1, C.byref(comm)))
    dq = pvs.DeviceBuffer.from_numpy(queries)
    oi, od, oc = pvs.DeviceBuffer(40 * 50 * 8), pvs.DeviceBuffer(40 * 50 * 4), pvs.DeviceBuffer(40 * 4)
    L.check(pvs.lib().pvs_search_sharded(ix._h, comm, dq.ptr, L.F32, 40, 50, pvs.COSINE, oi.ptr, od.ptr, oc.ptr))
    assert np.array_equal(oi.to_numpy(np.int64, (40, 50)), exp[0])
    assert np.array_equal(od.to_numpy(np.float32, (40, 50)).view(np.uint32), exp[1].view(np.uint32))
    assert np.array_equal(oc.to_numpy(np.uint32, (40,)), exp[2])
    # (c) four searches in flight on per-context streams, their collectives on the one comm stream;
    ix.set_streams(2)
    qsets = [orc.synth_rows(0x5EED0000 + 7 * i, 0, 40, 768) for i in range(4)]
    exps = [ix.search(qs, 50, pvs.COSINE) for qs in qsets]
    dqs = [pvs.DeviceBuffer.from_numpy(qs) for qs in qsets]
    outs = [(pvs.DeviceBuffer(40 * 50 * 8), pvs.DeviceBuffer(40 * 50 * 4), pvs.DeviceBuffer(40 * 4)) for _ in range(4)]
    for rep in range(3):
        tickets = []
        for i in range(4):
            t = C.c_uint32()
            L.check(pvs.lib().pvs_search_sharded_async(ix._h, comm, dqs[i].ptr, L.F32, 40, 50, pvs.COSINE, outs[i][0].ptr,
                                                       outs[i][1].ptr, outs[i][2].ptr, C.byref(t)))
            tickets.append(int(t.value))
        for i in range(4):
            ix.wait(tickets[i])
            assert np.array_equal(outs[i][0].to_numpy(np.int64, (40, 50)), exps[i][0])
            assert np.array_equal(outs[i][1].to_numpy(np.float32, (40, 50)).view(np.uint32), exps[i][1].view(np.uint32))
            assert np.array_equal(outs[i][2].to_numpy(np.uint32, (40,)), exps[i][2])
    # a batch in which some queries are handed to the dense path (zero query: every cosine distance is NULL): the
    # sharded search redoes the exchange for that batch after the fallback
    qz = qsets[0].copy()
    qz[3] = 0.0
    qz[17] = 0.0
    expz = ix.search(qz, 50, pvs.COSINE)
    assert np.isnan(expz[1][3]).all() and expz[0][3].tolist() == list(range(50))
    dqz = pvs.DeviceBuffer.from_numpy(qz)
    for _ in range(2):
        L.check(pvs.lib().pvs_search_sharded(ix._h, comm, dqz.ptr, L.F32, 40, 50, pvs.COSINE, outs[0][0].ptr, outs[0][1].ptr, outs[0][2].ptr))
        assert np.array_equal(outs[0][0].to_numpy(np.int64, (40, 50)), expz[0])
        gdz = outs[0][1].to_numpy(np.float32, (40, 50))
        assert np.array_equal(np.isnan(gdz), np.isnan(expz[1])) and np.array_equal(gdz[~np.isnan(gdz)], expz[1][~np.isnan(expz[1])])
    ix.set_streams(1)
    # (d) per-item search through the communicator (1 rank): gather + merge must be the identity
    grp = np.sort(rng.integers(0, 2000, len(rows))).astype(np.int64)
    ixg = pvs.VectorIndex(pvs.I8, 768)
    ixg.set_scale(scale)
    ixg.add_f32(rows, group_ids=grp)
    hq8 = orc.quantize_int8(queries[:6], scale)
    for agg in (pvs.AGG_MIN, pvs.AGG_AVG):
        eg, ev, ec = ixg.search_groups(hq8, 30, pvs.COSINE, agg)
        og, ov, ocn = np.empty((6, 30), np.int64), np.empty((6, 30), np.float64), np.empty(6, np.uint32)
        L.check(pvs.lib().pvs_search_groups_sharded(ixg._h, comm, hq8.ctypes.data_as(C.c_void_p), L.I8, 6, 30, pvs.COSINE, agg, None,
                                                    og.ctypes.data_as(C.c_void_p), ov.ctypes.data_as(C.c_void_p),
                                                    ocn.ctypes.data_as(C.c_void_p)))
        assert np.array_equal(ocn, ec) and np.array_equal(og, eg)
        assert np.array_equal(np.isnan(ov), np.isnan(ev)) and np.array_equal(ov[~np.isnan(ov)], ev[~np.isnan(ev)])
    ixg.close()
    pvs.lib().pvs_comm_destroy(comm)
    ix.close()


def _group_ids(rng, n, n_groups):
    g = rng.integers(0, n_groups, n).astype(np.int64) * 3 + 100  # unsorted: rows of a group are scattered
    return g


@pytest.mark.parametrize("dtype", ["i8", "f16", "f32"])
def test_score_batch_dense_matrix(pvs, dtype):
    dt = {"i8": pvs.I8, "f16": pvs.F16, "f32": pvs.F32}[dtype]
    n, dim, b = 3000, 768, 37
    rows = unit_rows(41, n, dim)
    rows[11] = 0.0
    queries = orc.synth_rows(0x5EED0000, 0, b, dim)
    scale = orc.compute_int8_scale(rows)
    ix = make_index(pvs, dt, rows, scale)
    hc = host_corpus(dt, rows, scale)
    hq = orc.quantize_int8(queries, scale) if dt == pvs.I8 else queries
    for metric in (pvs.COSINE, pvs.L2):
        got = ix.score_batch(hq, metric)
        assert got.shape == (n, b)
        for q in (0, 5, b - 1):
            exp = orc.score_all(dt, metric, hc, hq[q])
            assert np.array_equal(np.isnan(got[:, q]), np.isnan(exp))
            ok = ~np.isnan(exp)
            assert np.array_equal(got[ok, q].view(np.uint32), exp[ok].view(np.uint32)), (dtype, metric, q)
    ix.close()


@pytest.mark.parametrize("dtype,dim,n,b", [("f32", 3000, 700, 7), ("f16", 1536, 1000, 3), ("i8", 1100, 900, 6), ("f32", 5, 65, 1),
                                              ("f16", 4099, 129, 5), ("f32", 768, 1500, 19), ("f16", 1000, 800, 13), ("f32", 1024, 333, 8),
                                              ("f32", 1030, 200, 9), ("f16", 2050, 131, 16)])
def test_dense_exact_wide_rows_and_query_groups(pvs, dtype, dim, n, b):
    """Every row against b queries through the streaming exact kernel: query groups of 8 (float rows: pairs of queries on the
    packed f32 pipe) / 4 / 2 / 1 per pass,
    row pitches whose padded queries no longer fit four at a time beside the LDS ring, int8 rows whose
    sums leave the closed-form range (dim * 127^2 >= 2^24), odd dims and a ragged last tile pair."""
    dt = {"i8": pvs.I8, "f16": pvs.F16, "f32": pvs.F32}[dtype]
    rows = unit_rows(43 + dim, n, dim)
    rows[n // 2] = 0.0
    queries = orc.synth_rows(0x5EED0001, 0, b, dim)
    if dt == pvs.I8:
        rng = np.random.default_rng(dim)
        hc = rng.choice(np.array([-128, -127, 126, 127], np.int8), size=(n, dim))
        hq = rng.choice(np.array([-128, 127], np.int8), size=(b, dim))
        ix = pvs.VectorIndex(pvs.I8, dim)
        ix.set_scale(1.0)
        ix.add(hc)
    else:
        ix = make_index(pvs, dt, rows, None)
        hc, hq = host_corpus(dt, rows, None), queries
    for metric in (pvs.COSINE, pvs.L2):
        got = ix.score_batch(hq, metric)
        assert got.shape == (n, b)
        for q in range(b):
            exp = orc.score_all(dt, metric, hc, hq[q])
            assert np.array_equal(np.isnan(got[:, q]), np.isnan(exp))
            ok = ~np.isnan(exp)
            assert np.array_equal(got[ok, q].view(np.uint32), exp[ok].view(np.uint32)), (dtype, metric, q)
        one = ix.score_all(hq[0], metric)
        ok = ~np.isnan(one)
        assert np.array_equal(one[ok].view(np.uint32), got[ok, 0].view(np.uint32))
    ix.close()


@pytest.mark.parametrize("dtype", ["f16", "f32"])
def test_dense_exact_packed_pairs_equal_the_scalar_chains(pvs, dtype):
    """The 8-query instance of k_dense_exact runs two queries' chains in the halves of v_pk_mul_f32 / v_pk_add_f32: every
    distance must equal the 4-query form's (pvs_debug_set("dense_nq4", 1)) and the oracle's bit for bit, including products and
    partial sums that are subnormal, rows of huge components (inf sums) and NaN components."""
    dt = {"f16": pvs.F16, "f32": pvs.F32}[dtype]
    n, dim, b = 2500, 300, 24
    rng = np.random.default_rng(77)
    rows = unit_rows(91, n, dim)
    tiny = 6.0e-8 if dt == pvs.F16 else 1.0e-22  # (f16: its smallest subnormal)
    rows[5] *= tiny / max(np.abs(rows[5]).max(), 1e-30)
    rows[6] = tiny
    rows[7] = 3.0e4 if dt == pvs.F16 else 3.0e19
    rows[8, 3] = np.nan
    rows[9] = 0.0
    queries = orc.synth_rows(0x5EED0007, 0, b, dim)
    queries[1] *= 1.0e-20
    queries[2] = 1.0e-23
    queries[3] *= 1.0e19
    queries[4] = 0.0
    ix = make_index(pvs, dt, rows, None)
    hc = host_corpus(dt, rows, None)
    for metric in (pvs.COSINE, pvs.L2):
        got = ix.score_batch(queries, metric)
        pvs.debug_set("dense_nq4", 1)
        try:
            old = ix.score_batch(queries, metric)
        finally:
            pvs.debug_set("dense_nq4", 0)
        assert np.array_equal(got.view(np.uint32), old.view(np.uint32)), (dtype, metric)
        # (b = 24 > 8: `got` came from k_exact_wide — 32 chains per pass, query components as scalar operands; the 8-per-pass
        #  LDS form must agree with it bit for bit too)
        pvs.debug_set("no_exact_wide", 1)
        try:
            old8 = ix.score_batch(queries, metric)
        finally:
            pvs.debug_set("no_exact_wide", 0)
        assert np.array_equal(got.view(np.uint32), old8.view(np.uint32)), (dtype, metric)
        for q in range(b):
            exp = orc.score_all(dt, metric, hc, queries[q])
            assert np.array_equal(np.isnan(got[:, q]), np.isnan(exp)), (dtype, metric, q)
            ok = ~np.isnan(exp)
            assert np.array_equal(got[ok, q].view(np.uint32), exp[ok].view(np.uint32)), (dtype, metric, q)
    ix.close()


@pytest.mark.parametrize("dtype", ["i8", "f16"])
def test_search_groups_matches_sqlite_aggregate_semantics(pvs, dtype):
    # filters/exact.rs:67-80: MIN / MAX / AVG per file, weighted average when weights apply
    dt = pvs.I8 if dtype == "i8" else pvs.F16
    rng = np.random.default_rng(17)
    n, dim, b, k = 4000, 512, 3, 60
    rows = unit_rows(43, n, dim)
    rows[100] = 0.0  # a NULL distance inside some group
    groups = _group_ids(rng, n, 700)
    queries = orc.synth_rows(0x5EED0000, 7, b, dim)
    scale = orc.compute_int8_scale(rows)
    ix = pvs.VectorIndex(dt, dim)
    if dt == pvs.I8:
        ix.set_scale(scale)
    ix.add_f32(rows, group_ids=groups)
    hc = host_corpus(dt, rows, scale)
    hq = orc.quantize_int8(queries, scale) if dt == pvs.I8 else queries
    w = (rng.random(n) + 0.05).astype(np.float32)
    for metric in (pvs.COSINE, pvs.L2):
        for agg, oagg in ((pvs.AGG_MIN, orc.AGG_MIN), (pvs.AGG_MAX, orc.AGG_MAX), (pvs.AGG_AVG, orc.AGG_AVG)):
            gg, gv, gc = ix.search_groups(hq, k, metric, agg)
            for q in range(b):
                eg, ev = orc.search_groups(dt, metric, hc, hq[q], groups, oagg, k)
                assert gc[q] == len(eg) and np.array_equal(gg[q, : len(eg)], eg)
                assert np.array_equal(gv[q, : len(eg)].view(np.uint64), ev.view(np.uint64)), (dtype, metric, agg)
        gg, gv, gc = ix.search_groups(hq, k, metric, pvs.AGG_MIN, row_weights=w)
        for q in range(b):
            eg, ev = orc.search_groups(dt, metric, hc, hq[q], groups, orc.AGG_MIN, k, weights=w)
            assert np.array_equal(gg[q, : len(eg)], eg)
            assert np.array_equal(gv[q, : len(eg)].view(np.uint64), ev.view(np.uint64))
    ix.close()
    # no group ids: identity grouping == row search
    ix2 = make_index(pvs, dt, rows, scale)
    gg, gv, gc = ix2.search_groups(hq[:1], 10, pvs.COSINE, pvs.AGG_MIN)
    ri, rd, rc = ix2.search(hq[:1], 10, pvs.COSINE)
    assert np.array_equal(gg[0], ri[0]) and np.array_equal(gv[0], rd[0].astype(np.float64))
    ix2.close()


def test_min_groups_through_the_filter_scan_equals_dense(pvs):
    """MIN per group served from a row page of the filter scan (no dense matrix) must equal the dense
    GROUP BY, including: distance ties across groups at the page boundary (duplicated vectors), groups
    with many rows (page must grow), k >= number of groups, NULL-only groups."""
    rng = np.random.default_rng(31)
    n, dim = 6000, 256
    rows = unit_rows(59, n, dim)
    rows[2000:2400] = rows[1000:1400]  # 400 exact duplicates living in other groups: equal distances
    rows[5000] = 0.0
    groups = np.concatenate([np.repeat(np.arange(100, dtype=np.int64), 30),  # 100 groups x 30 rows
                             3000 + rng.integers(0, 800, n - 3000)]).astype(np.int64)
    groups[5000] = 99_999  # a group whose only row has a NULL cosine distance
    scale = orc.compute_int8_scale(rows)
    ix = pvs.VectorIndex(pvs.I8, dim)
    ix.set_scale(scale)
    ix.add_f32(rows, group_ids=groups)
    hq = orc.quantize_int8(np.concatenate([orc.synth_rows(0x5EED0003, 0, 5, dim), rows[1000:1002]]), scale)  # two queries ARE stored rows
    n_groups = len(np.unique(groups))
    for metric in (pvs.COSINE, pvs.L2):
        for k in (1, 10, 150, n_groups, n_groups + 5):
            ix.set_path(0)
            fg, fv, fc = ix.search_groups(hq, k, metric, pvs.AGG_MIN)
            ix.set_path(1)
            dg, dv, dc = ix.search_groups(hq, k, metric, pvs.AGG_MIN)
            assert np.array_equal(fc, dc), (metric, k)
            for q in range(len(hq)):
                assert np.array_equal(fg[q, : fc[q]], dg[q, : dc[q]]), (metric, k, q)
                a, b = fv[q, : fc[q]], dv[q, : dc[q]]
                assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)], b[~np.isnan(b)])
    ix.set_path(0)
    before = ix.stats().dense_queries
    ix.search_groups(hq, 10, pvs.COSINE, pvs.AGG_MIN)
    assert ix.stats().dense_queries == before, "MIN over groups must not score the dense matrix"
    ix.close()


@pytest.mark.parametrize("dtype", ["i8", "f16", "f32"])
def test_similar_to_matches_self_join(pvs, dtype):
    # filters/item_similarity.rs:432-581; :3532-3582 similar_to_quant_matches_exact is the ordering-level test
    dt = {"i8": pvs.I8, "f16": pvs.F16, "f32": pvs.F32}[dtype]
    rng = np.random.default_rng(23)
    n, dim, k = 2500, 512, 40
    rows = unit_rows(47, n, dim)
    groups = np.sort(_group_ids(rng, n, 600))  # files: a few vectors each
    ids = np.arange(n, dtype=np.int64) * 2 + 5
    targets = [int(t) for t in np.nonzero(groups == groups[1234])[0]]  # every vector of one item
    scale = orc.compute_int8_scale(rows)
    ix = pvs.VectorIndex(dt, dim)
    if dt == pvs.I8:
        ix.set_scale(scale)
    ix.add_f32(rows, row_ids=ids, group_ids=groups)
    hc = host_corpus(dt, rows, scale)
    for metric in (pvs.L2, pvs.COSINE):
        for agg, oagg in ((pvs.AGG_AVG, orc.AGG_AVG), (pvs.AGG_MIN, orc.AGG_MIN), (pvs.AGG_MAX, orc.AGG_MAX)):
            gg, gv = ix.similar_to(ids[targets], k, metric, agg)
            eg, ev = orc.similar_to(dt, metric, hc, targets, groups, oagg, k)
            assert np.array_equal(gg, eg), (dtype, metric, agg)
            assert np.array_equal(gv.view(np.uint64), ev.view(np.uint64))
    assert groups[1234] not in gg  # the target's own group has no non-target rows left
    with pytest.raises(pvs.PvsError):
        ix.similar_to([4], 5)  # not a row id of this index
    ix.close()


def test_similar_to_int8_sums_beyond_the_closed_form(pvs):
    """similar_to scores int8 rows by integer sums and the closed form of the reference's f32 chain and does not wait for the
    scorer's out-of-range flag (a pinned word, read after the ranking): saturated codes at 1,100 dimensions raise it, the call is
    redone with the in-order chains — same groups and f64 values as the oracle; targets stored side by side and apart (the
    exclusion mask is filled run by run)."""
    rng = np.random.default_rng(1101)
    n, dim, k = 1200, 1100, 25
    hc = rng.choice(np.array([-128, -127, 126, 127], np.int8), size=(n, dim))
    groups = (np.arange(n, dtype=np.int64) // 3) * 5 + 2
    ids = np.arange(n, dtype=np.int64) * 3 + 1
    ix = pvs.VectorIndex(pvs.I8, dim)
    ix.set_scale(1.0)
    ix.add(hc, row_ids=ids, group_ids=groups)
    for targets in ([300, 301, 302], [9, 10, 11, 600, 602, 1199]):
        for metric in (pvs.L2, pvs.COSINE):
            gg, gv = ix.similar_to(ids[targets], k, metric, pvs.AGG_AVG)
            eg, ev = orc.similar_to(orc.I8, metric, hc, targets, groups, orc.AGG_AVG, k)
            assert np.array_equal(gg, eg), (targets, metric)
            assert np.array_equal(gv.view(np.uint64), ev.view(np.uint64))
    ix.close()


def test_similar_to_confidence_weighted(pvs):
    """item_similarity.rs:503-581: SUM(d*w)/SUM(w) over the fan-out with w from the two rows' confidences.
    pow() runs in the device math library: values agree to 1e-13 relative, the ranking exactly."""
    rng = np.random.default_rng(29)
    n, dim, k = 1800, 384, 30
    rows = unit_rows(53, n, dim)
    rows[77] = 0.0  # a NULL cosine distance inside some group: drops out of SUM(d*w), stays in SUM(w)
    groups = np.sort(_group_ids(rng, n, 400))
    ids = np.arange(n, dtype=np.int64) + 10
    targets = [int(t) for t in np.nonzero(groups == groups[900])[0]]
    conf = rng.uniform(0.05, 1.0, n)
    lang = rng.uniform(0.3, 1.0, n)
    conf[rng.random(n) < 0.2] = np.nan  # NULL -> coalesce(…, 1)
    lang[rng.random(n) < 0.2] = np.nan
    ix = pvs.VectorIndex(pvs.F16, dim)
    ix.add_f32(rows, row_ids=ids, group_ids=groups)
    hc = rows.astype(np.float16)
    for metric in (pvs.L2, pvs.COSINE):
        for cw, lw in ((1.5, 0.0), (0.0, 2.0), (0.7, 1.3)):
            gg, gv = ix.similar_to_weighted(ids[targets], k, metric, pvs.AGG_AVG, conf, lang, cw, lw)
            eg, ev = orc.similar_to_weighted(orc.F16, metric, hc, targets, groups, k, conf, lang, cw, lw)
            assert np.array_equal(gg, eg), (metric, cw, lw)
            assert np.allclose(gv, ev, rtol=1e-13, atol=0.0), (metric, cw, lw, np.max(np.abs(gv - ev) / np.abs(ev)))
    # both exponents zero: the plain aggregate, bit for bit
    gg, gv = ix.similar_to_weighted(ids[targets], k, pvs.L2, pvs.AGG_MIN, conf, lang, 0.0, 0.0)
    eg, ev = orc.similar_to(orc.F16, orc.L2, hc, targets, groups, orc.AGG_MIN, k)
    assert np.array_equal(gg, eg) and np.array_equal(gv.view(np.uint64), ev.view(np.uint64))
    # no confidence arrays at all: every weight is pow(1, x) = 1 -> the AVG
    gg, gv = ix.similar_to_weighted(ids[targets], k, pvs.L2, pvs.AGG_MAX, None, None, 2.0, 0.5)
    eg, ev = orc.similar_to(orc.F16, orc.L2, hc, targets, groups, orc.AGG_AVG, k)
    assert np.array_equal(gg, eg) and np.allclose(gv, ev, rtol=1e-14)
    ix.close()


def test_similar_to_cross_modal_gates(pvs):
    """item_similarity.rs:473-489: with clip_xmodal the collection holds image ('clip') and text
    ('text-embedding') vectors of one CLIP space; xmodal_i2i = false drops clip x clip pairs from the self-join,
    xmodal_t2t = false the text x text pairs.  Unweighted aggregates stay bit-exact."""
    rng = np.random.default_rng(37)
    n, dim, k = 2000, 256, 25
    rows = unit_rows(67, n, dim)
    groups = np.sort(_group_ids(rng, n, 450))
    ids = np.arange(n, dtype=np.int64) * 3 + 1
    kind = (rng.random(n) < 0.4).astype(np.uint8)  # 1 = text-embedding
    tg = groups[777]
    targets = [int(t) for t in np.nonzero(groups == tg)[0]]
    kind[targets[0]] = 0
    if len(targets) > 1:
        kind[targets[1]] = 1  # the target item has both modalities
    scale = orc.compute_int8_scale(rows)
    ix = pvs.VectorIndex(pvs.I8, dim)
    ix.set_scale(scale)
    ix.add_f32(rows, row_ids=ids, group_ids=groups)
    hc = orc.quantize_int8(rows, scale)
    conf = rng.uniform(0.1, 1.0, n)
    for i2i, t2t in ((True, True), (False, True), (True, False), (False, False)):
        for agg, oagg in ((pvs.AGG_AVG, orc.AGG_AVG), (pvs.AGG_MIN, orc.AGG_MIN), (pvs.AGG_MAX, orc.AGG_MAX)):
            gg, gv = ix.similar_to_ex(ids[targets], k, pvs.COSINE, agg, row_kind=kind, xmodal_i2i=i2i, xmodal_t2t=t2t)
            eg, ev = orc.similar_to_ex(orc.I8, orc.COSINE, hc, targets, groups, oagg, k, kind=kind, xmodal_i2i=i2i, xmodal_t2t=t2t)
            assert np.array_equal(gg, eg), (i2i, t2t, agg)
            assert np.array_equal(gv.view(np.uint64), ev.view(np.uint64)), (i2i, t2t, agg)
        gg, gv = ix.similar_to_ex(ids[targets], k, pvs.L2, pvs.AGG_AVG, confidence=conf, confidence_weight=1.25, row_kind=kind,
                                  xmodal_i2i=i2i, xmodal_t2t=t2t)
        eg, ev = orc.similar_to_ex(orc.I8, orc.L2, hc, targets, groups, orc.AGG_AVG, k, conf=conf, cw=1.25, kind=kind, xmodal_i2i=i2i,
                                   xmodal_t2t=t2t)
        assert np.array_equal(gg, eg) and np.allclose(gv, ev, rtol=1e-13, atol=0.0)
    # gates off + no kinds == plain similar_to
    a = ix.similar_to_ex(ids[targets], k, pvs.COSINE, pvs.AGG_AVG, row_kind=kind)
    b = ix.similar_to(ids[targets], k, pvs.COSINE, pvs.AGG_AVG)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint64), b[1].view(np.uint64))
    ix.close()


# ------------------------------------------------------------ fallback / edge paths
def _check(pvs, ix, dt, metric, hc, hq, k, ids=None):
    exp = orc.search(dt, metric, hc, hq, k, ids=ids, threads=8)
    gi, gd, gc = ix.search(hq, k, metric)
    n = exp[0].shape[1]
    assert gc.tolist() == [n] * exp[0].shape[0]
    assert np.array_equal(gi[:, :n], exp[0])
    a, b = gd[:, :n], exp[1]
    assert np.array_equal(np.isnan(a), np.isnan(b))
    assert np.array_equal(a[~np.isnan(a)].view(np.uint32), b[~np.isnan(b)].view(np.uint32))


def test_massive_ties_fall_back_to_dense(pvs):
    # thousands of identical rows: more survivors than the exact-rerank stage holds -> dense path
    base = unit_rows(51, 64, 768)
    rows = np.concatenate([np.repeat(base[:1], 9000, axis=0), base, np.repeat(base[1:2], 3000, axis=0)])  # > 8192 survivors
    scale = orc.compute_int8_scale(rows)
    for dt in (pvs.I8, pvs.F16):
        ix = make_index(pvs, dt, rows, scale)
        hc = host_corpus(dt, rows, scale)
        hq = orc.quantize_int8(base[:1] * 0.7 + base[2:3] * 0.3, scale) if dt == pvs.I8 else (base[:1] * 0.7 + base[2:3] * 0.3)
        # a single query takes the one-launch exact search (pvs_direct.hip): ties cost it nothing
        _check(pvs, ix, dt, pvs.COSINE, hc, hq, 10)
        _check(pvs, ix, dt, pvs.L2, hc, hq, 100)
        assert ix.stats().dense_queries == 0
        pvs.debug_set("no_direct_topk", 1)  # the filter scan's own behaviour
        try:
            _check(pvs, ix, dt, pvs.COSINE, hc, hq, 10)
            _check(pvs, ix, dt, pvs.L2, hc, hq, 100)
        finally:
            pvs.debug_set("no_direct_topk", 0)
        assert ix.stats().dense_queries >= 1, "ties beyond the survivor capacity must be answered by the dense path"
        ix.close()


def test_unrepresentative_sample_overflows_to_dense(pvs):
    # pass A samples a strided subset of workgroup tiles (for one query: 128-row tiles, stride
    # n_tiles // 256); put every near neighbour in the unsampled tiles so the threshold is far too
    # loose and the candidate lists overflow -> dense path, still exact
    n, dim = 700_000, 64
    rows = orc.synth_rows(5, 0, n, dim)
    q = orc.synth_rows(6, 0, 1, dim)[0]
    n_wgtiles = (n + 127) // 128
    step = max(1, n_wgtiles // 256)
    near = ((np.arange(n) // 128) % step) != 0
    noise = orc.synth_rows(7, 0, n, dim)
    rows[near] = (q[None, :] + 0.05 * noise[near]).astype(np.float32)
    rows[~near] = -rows[~near] * np.sign(rows[~near] @ q)[:, None]  # sampled tiles: anti-correlated rows
    scale = orc.compute_int8_scale(rows)
    ix = make_index(pvs, pvs.I8, rows, scale)
    hc = orc.quantize_int8(rows, scale)
    hq = orc.quantize_int8(q[None, :], scale)
    _check(pvs, ix, pvs.I8, pvs.COSINE, hc, hq, 100)  # (a single query: the one-launch exact search, no sample involved)
    assert ix.stats().dense_queries == 0
    pvs.debug_set("no_direct_topk", 1)  # the filter scan's own behaviour
    try:
        _check(pvs, ix, pvs.I8, pvs.COSINE, hc, hq, 100)
    finally:
        pvs.debug_set("no_direct_topk", 0)
    st = ix.stats()
    assert st.dense_queries == 1, "candidate overflow must hand the query to the dense path"
    ix.close()


def test_large_k_many_chunks_odd_shapes(pvs):
    rng = np.random.default_rng(3)
    # k beyond the filter path's page size (4096) -> dense path; k = 3000 of 5000 rows: nearly every row is a survivor
    rows = unit_rows(61, 5000, 512)
    scale = orc.compute_int8_scale(rows)
    q = orc.synth_rows(62, 0, 2, 512)
    ix = make_index(pvs, pvs.I8, rows, scale)
    _check(pvs, ix, pvs.I8, pvs.COSINE, orc.quantize_int8(rows, scale), orc.quantize_int8(q, scale), 3000)
    before = ix.stats().dense_queries
    _check(pvs, ix, pvs.I8, pvs.COSINE, orc.quantize_int8(rows, scale), orc.quantize_int8(q, scale), 4500)
    assert ix.stats().dense_queries == before + 2
    _check(pvs, ix, pvs.I8, pvs.L2, orc.quantize_int8(rows, scale), orc.quantize_int8(q, scale), 1)
    ix.close()
    # 300 queries = three scan chunks (128 + 128 + 44)
    rows = unit_rows(63, 6000, 768)
    q = orc.synth_rows(64, 0, 300, 768)
    ix = make_index(pvs, pvs.F16, rows)
    _check(pvs, ix, pvs.F16, pvs.COSINE, rows.astype(np.float16), q, 20)
    ix.close()
    # dims that are not multiples of the k-slab / 16-byte chunk; tiny and ragged row counts
    for dt, dim in ((pvs.I8, 100), (pvs.I8, 1000), (pvs.I8, 1024), (pvs.F16, 200), (pvs.F16, 384), (pvs.F16, 1024), (pvs.F32, 77), (pvs.F32, 200), (pvs.F32, 1000)):
        for n in (1, 31, 33, 2500):
            rows = unit_rows(65 + dim, n, dim)
            scale = orc.compute_int8_scale(rows)
            q = orc.synth_rows(66, 0, 3, dim)
            ix = make_index(pvs, dt, rows, scale)
            hc = host_corpus(dt, rows, scale)
            hq = orc.quantize_int8(q, scale) if dt == pvs.I8 else q
            _check(pvs, ix, dt, pvs.COSINE, hc, hq, 7)
            _check(pvs, ix, dt, pvs.L2, hc, hq, 50)
            ix.close()


def test_zero_query_and_saturated_codes(pvs):
    # zero query: every cosine distance is NULL -> rows come back in id order, NaN distances
    rows = unit_rows(71, 400, 768)
    scale = orc.compute_int8_scale(rows)
    ix = make_index(pvs, pvs.I8, rows, scale)
    gi, gd, gc = ix.search(np.zeros((1, 768), np.int8), 10, pvs.COSINE)
    assert gc[0] == 10 and gi[0].tolist() == list(range(10)) and np.isnan(gd[0]).all()
    ix.close()
    # saturated +-127/-128 codes at dim 1024: the L2 sum of squares leaves the exactly representable
    # range (> 2^24), so the reference's sequential f32 rounding matters -> in-order recompute
    rng = np.random.default_rng(8)
    codes = rng.choice(np.array([-128, -127, 127], np.int8), size=(3000, 1024))
    qcodes = rng.choice(np.array([-128, 127], np.int8), size=(4, 1024))
    ix = pvs.VectorIndex(pvs.I8, 1024)
    ix.set_scale(1.0)
    ix.add(codes)
    assert int(((codes[0].astype(np.int64) - qcodes[0].astype(np.int64)) ** 2).sum()) > 2**24
    _check(pvs, ix, pvs.I8, pvs.L2, codes, qcodes, 25)
    _check(pvs, ix, pvs.I8, pvs.COSINE, codes, qcodes, 25)
    got = ix.score_batch(qcodes, pvs.L2)
    for qq in range(4):
        assert np.array_equal(got[:, qq].view(np.uint32), orc.score_all(orc.I8, orc.L2, codes, qcodes[qq]).view(np.uint32))
    ix.close()


def test_sixteen_host_threads_share_one_index(pvs):
    """Boundary contract: pvs_search is callable concurrently from the 16 read-pool threads
    of the reference (db/connection.rs:235,320-357).  Every thread must get its own query's page."""
    import threading

    rows = unit_rows(81, 20000, 768)
    scale = orc.compute_int8_scale(rows)
    ix = make_index(pvs, pvs.I8, rows, scale)
    hc = orc.quantize_int8(rows, scale)
    nthreads, per = 16, 6
    queries = orc.synth_rows(0x5EED0002, 0, nthreads * per, 768)
    hq = orc.quantize_int8(queries, scale)
    exp_i, exp_d = orc.search(orc.I8, orc.COSINE, hc, hq, 20, threads=8)
    errors = []

    def worker(t):
        try:
            for rep in range(per):
                q = t * per + rep
                nb = 1 + (q % 3)  # mixed batch sizes across threads
                sel = [(q + i) % (nthreads * per) for i in range(nb)]
                gi, gd, gc = ix.search(hq[sel], 20, pvs.COSINE)
                for i, qq in enumerate(sel):
                    if not (np.array_equal(gi[i], exp_i[qq]) and np.array_equal(gd[i].view(np.uint32), exp_d[qq].view(np.uint32))):
                        errors.append((t, rep, qq))
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(nthreads)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errors, errors[:5]
    ix.close()


def test_or_composition_of_image_and_text_filters_rrf(pvs):
    """BASELINE configs[4] in miniature: a 512-d image-embedding index (cosine) and a 1024-d text-embedding
    index (L2, text_embeddings.rs:386-393) over the same files, each filter aggregated per file (MIN,
    exact.rs:67-80) and ranked with row_number() (builder.rs:757-771); the OR arm is the UNION of the two
    branches (builder.rs:638-661) ordered by the RRF score (builder.rs:1284-1317).  Device results per
    branch + host fusion must equal the oracle's composition exactly (f64)."""
    rng = np.random.default_rng(5)
    n_files, k_page = 700, 50
    # image branch: ~3 vectors per file; text branch: ~5 vectors, only 60 % of the files have text
    img_files = np.sort(rng.integers(0, n_files, 2100)).astype(np.int64)
    txt_files = np.sort(rng.choice(np.arange(0, n_files, dtype=np.int64)[rng.random(n_files) < 0.6], 3000))
    img_rows, txt_rows = unit_rows(91, len(img_files), 512), unit_rows(92, len(txt_files), 1024)
    s_img, s_txt = orc.compute_int8_scale(img_rows), orc.compute_int8_scale(txt_rows)
    ix_img, ix_txt = pvs.VectorIndex(pvs.I8, 512), pvs.VectorIndex(pvs.I8, 1024)
    ix_img.set_scale(s_img)
    ix_txt.set_scale(s_txt)
    ix_img.add_f32(img_rows, group_ids=img_files)
    ix_txt.add_f32(txt_rows, group_ids=txt_files)
    q_img = orc.quantize_int8(orc.synth_rows(93, 0, 1, 512), s_img)
    q_txt = orc.quantize_int8(orc.synth_rows(94, 0, 1, 1024), s_txt)

    # device: every file of each branch, ranked (k = all groups)
    gi, vi, ci = ix_img.search_groups(q_img, n_files, pvs.COSINE, pvs.AGG_MIN)
    gt, vt, ct = ix_txt.search_groups(q_txt, n_files, pvs.L2, pvs.AGG_MIN)
    gi, vi, gt, vt = gi[0, : ci[0]], vi[0, : ci[0]], gt[0, : ct[0]], vt[0, : ct[0]]
    # oracle branches
    ogi, ovi = orc.search_groups(orc.I8, orc.COSINE, orc.quantize_int8(img_rows, s_img), q_img[0], img_files, orc.AGG_MIN, n_files)
    ogt, ovt = orc.search_groups(orc.I8, orc.L2, orc.quantize_int8(txt_rows, s_txt), q_txt[0], txt_files, orc.AGG_MIN, n_files)
    assert np.array_equal(gi, ogi) and np.array_equal(vi.view(np.uint64), ovi.view(np.uint64))
    assert np.array_equal(gt, ogt) and np.array_equal(vt.view(np.uint64), ovt.view(np.uint64))

    def fuse(rrf, rownum, g_a, v_a, g_b, v_b):
        files = np.union1d(g_a, g_b)  # UNION of the branches' files
        ranks = np.full((2, len(files)), -1, np.int64)  # NULL where a branch has no row for the file
        for b, (g, v) in enumerate(((g_a, v_a), (g_b, v_b))):
            ranks[b, np.searchsorted(files, g)] = rownum(v, g)
        score = rrf(ranks)
        order = np.lexsort((files, -score))  # ORDER BY score DESC (ties: file id, for determinism)
        return files[order][:k_page], score[order][:k_page]

    ks, ws = np.array([1, 60], np.int32), np.array([1.0, 0.5])
    got_f, got_s = fuse(lambda r: pvs.rrf_fuse(r, ks, ws), pvs.row_number, gi, vi, gt, vt)
    exp_f, exp_s = fuse(lambda r: np.array([orc.rrf_score(r[:, i], ks, ws) for i in range(r.shape[1])]), orc.row_number, ogi, ovi, ogt, ovt)
    assert np.array_equal(got_f, exp_f) and np.array_equal(got_s.view(np.uint64), exp_s.view(np.uint64))
    ix_img.close()
    ix_txt.close()


@pytest.mark.parametrize("dtype", ["i8", "f16", "f32"])
def test_clustered_and_skewed_corpora(pvs, dtype):
    """Real embeddings are clustered and (before normalisation) of very different lengths; the sampled
    threshold and the error intervals must stay valid there: tight clusters (many near-identical rows, a
    query inside a cluster), a sample stride that resonates with the cluster layout, rows whose norms span
    six orders of magnitude (L2 and cosine), and a query far outside the data.  Whatever path answers,
    the page equals the oracle's."""
    dt = {"i8": pvs.I8, "f16": pvs.F16, "f32": pvs.F32}[dtype]
    rng = np.random.default_rng(41)
    n, dim, k = 24000, 256, 50
    centers = rng.standard_normal((40, dim)).astype(np.float32)
    centers /= np.linalg.norm(centers, axis=1, keepdims=True)
    assign = np.arange(n) % 40  # cluster id cycles with the row index: periodic in the tile order
    rows = centers[assign] + 0.02 * rng.standard_normal((n, dim)).astype(np.float32)
    rows /= np.linalg.norm(rows, axis=1, keepdims=True)
    queries = np.concatenate([centers[:3] + 0.01 * rng.standard_normal((3, dim)).astype(np.float32),  # inside clusters
                              rng.standard_normal((2, dim)).astype(np.float32),                        # generic
                              -centers[5:6]]).astype(np.float32)                                       # opposite of a cluster
    scale = orc.compute_int8_scale(rows)
    ix = make_index(pvs, dt, rows, scale)
    hc = host_corpus(dt, rows, scale)
    hq = orc.quantize_int8(queries, scale) if dt == pvs.I8 else queries
    for metric in (pvs.COSINE, pvs.L2):
        _check(pvs, ix, dt, metric, hc, hq, k)
    ix.close()
    if dt == pvs.I8:
        return  # one scale per space: int8 spaces are built from normalised vectors
    # wildly different row norms
    lens = np.exp(rng.uniform(np.log(1e-3), np.log(1e3), n)).astype(np.float32)
    if dt == pvs.F16:
        lens = np.clip(lens, 1e-2, 1e2)  # stay inside binary16's range
    rows2 = (rows * lens[:, None]).astype(np.float32)
    q2 = np.concatenate([rows2[[11, 4097]], 1e2 * queries[:2], 1e-2 * queries[3:4]]).astype(np.float32)
    ix = make_index(pvs, dt, rows2, None)
    hc = host_corpus(dt, rows2, None)
    for metric in (pvs.COSINE, pvs.L2):
        _check(pvs, ix, dt, metric, hc, q2, k)
    ix.close()


@pytest.mark.parametrize("dtype", ["i8", "f16", "f32"])
def test_many_small_appends_keep_the_tiled_layout_consistent(pvs, dtype):
    """The index grows by appends of arbitrary size (inline quantization writes one vector at a time,
    write_inline_quants db/vector_quants.rs:1347-1438): capacity reallocations copy whole 32-row tiles and
    appends start in the middle of a tile.  After every few appends rows read back, dense scores and the
    search page must equal the oracle over the rows added so far."""
    dt = {"i8": pvs.I8, "f16": pvs.F16, "f32": pvs.F32}[dtype]
    dim = 200
    sizes = [1, 31, 33, 100, 7, 1000, 64, 5, 129, 2047, 3]
    total = sum(sizes)
    rows = unit_rows(97, total, dim)
    scale = orc.compute_int8_scale(rows)
    hc_all = host_corpus(dt, rows, scale)
    ids_all = np.cumsum(np.random.default_rng(3).integers(1, 5, total)).astype(np.int64)
    groups_all = (ids_all // 7).astype(np.int64)
    q = orc.synth_rows(98, 0, 3, dim)
    hq = orc.quantize_int8(q, scale) if dt == pvs.I8 else q
    ix = pvs.VectorIndex(dt, dim)  # no capacity hint: every growth step reallocates
    if dt == pvs.I8:
        ix.set_scale(scale)
    done = 0
    for step, m in enumerate(sizes):
        if step % 2 == 0:
            ix.add_f32(rows[done:done + m], row_ids=ids_all[done:done + m], group_ids=groups_all[done:done + m])
        else:  # rows already in the index dtype
            ix.add(hc_all[done:done + m], row_ids=ids_all[done:done + m], group_ids=groups_all[done:done + m])
        done += m
        assert ix.stats().rows == done
        if step % 3 == 2 or step == len(sizes) - 1:
            got = ix.read_rows(0, done)
            assert np.array_equal(got.view(np.uint8), hc_all[:done].view(np.uint8)), (dtype, step)
            for metric in (pvs.COSINE, pvs.L2):
                _check(pvs, ix, dt, metric, hc_all[:done], hq, min(20, done), ids=ids_all[:done])
                d = ix.score_all(hq[0], metric)
                e = orc.score_all(dt, metric, hc_all[:done], hq[0])
                assert np.array_equal(d.view(np.uint32), e.view(np.uint32))
            gg, gv, gc = ix.search_groups(hq[:1], 5, pvs.L2, pvs.AGG_AVG)
            eg, ev = orc.search_groups(dt, orc.L2, hc_all[:done], hq[0], groups_all[:done], orc.AGG_AVG, 5)
            assert np.array_equal(gg[0, : gc[0]], eg) and np.array_equal(gv[0, : gc[0]].view(np.uint64), ev.view(np.uint64))
    with pytest.raises(pvs.PvsError):
        ix.add_f32(rows[:2], row_ids=[ids_all[-1], ids_all[-1] + 1])  # ids must keep increasing
    ix.close()


def test_rrf_search_on_device_equals_the_sql_composition(pvs):
    """pvs_rrf_search: three branches over different indexes (512-d int8 cosine MIN, 1024-d f16 L2 AVG, 256-d f32
    cosine weighted, descending window) whose group sets only partly overlap; NULL aggregates (zero rows under
    cosine) rank FIRST in an ascending window; production RRF weights.  Groups, order and f64 scores equal the
    oracle's literal composition."""
    rng = np.random.default_rng(43)
    n_files = 900
    specs = [(pvs.I8, orc.I8, 512, 2600, pvs.COSINE, orc.COSINE), (pvs.F16, orc.F16, 1024, 1900, pvs.L2, orc.L2),
             (pvs.F32, orc.F32, 256, 1500, pvs.COSINE, orc.COSINE)]
    dev, ora = [], []
    for i, (dt, odt, dim, n, m, om) in enumerate(specs):
        rows = unit_rows(101 + i, n, dim)
        pool = np.arange(0, n_files, dtype=np.int64)[rng.random(n_files) < (0.9, 0.6, 0.4)[i]]
        groups = np.sort(rng.choice(pool, n)).astype(np.int64)
        if m == pvs.COSINE:
            z = np.nonzero(groups == groups[n // 3])[0]
            rows[z] = 0.0  # a whole file of zero vectors: NULL aggregate
        scale = orc.compute_int8_scale(rows)
        ix = pvs.VectorIndex(dt, dim)
        if dt == pvs.I8:
            ix.set_scale(scale)
        ix.add_f32(rows, group_ids=groups)
        q = orc.synth_rows(200 + i, 0, 1, dim)[0]
        hq = orc.quantize_int8(q[None, :], scale)[0] if dt == pvs.I8 else q
        w = (rng.random(n) + 0.1).astype(np.float32) if i == 2 else None
        agg = (pvs.AGG_MIN, pvs.AGG_AVG, pvs.AGG_MAX)[i]
        desc = i == 2
        rk, wt = ((5, 1.0), (5, 1.0), (10, 0.7))[i]  # quant_ab.rs:233-246
        dev.append(dict(index=ix, query=hq, metric=m, agg=agg, row_weights=w, descending=desc, rrf_k=rk, weight=wt))
        ora.append(dict(dtype=odt, metric=om, corpus=host_corpus(dt, rows, scale), query=hq, groups=groups, agg=agg, weights=w,
                        descending=desc, rrf_k=rk, weight=wt))
    for k in (1, 40, 2000):
        for nb in (1, 2, 3):
            gg, gs = pvs.rrf_search(dev[:nb], k)
            eg, es = orc.rrf_search(ora[:nb], k)
            assert np.array_equal(gg, eg), (k, nb)
            assert np.array_equal(gs.view(np.uint64), es.view(np.uint64)), (k, nb)
    for b in dev:
        b["index"].close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("PVS_FUZZ_SEEDS", "24"))))
def test_randomized_shapes_against_the_oracle(pvs, seed):
    """Seeded sweep over dtype / metric / rows / dim / batch / k (dims that are not multiples of any tile size,
    single rows, k > rows, batches across the 128/256 pass boundaries), with a sprinkling of duplicated rows,
    zero rows and, for float indexes, huge and tiny components."""
    rng = np.random.default_rng(1000 + seed)
    dt = [pvs.I8, pvs.F16, pvs.F32][seed % 3]
    metric = [pvs.COSINE, pvs.L2][(seed // 3) % 2]
    n = int(rng.choice([1, 2, 31, 32, 33, 127, 129, 500, 1999, 4097, 9000]))
    dim = int(rng.choice([1, 3, 16, 17, 63, 64, 65, 100, 255, 257, 384, 511, 768, 1000, 1024, 1100]))
    batch = int(rng.choice([1, 2, 31, 33, 127, 129, 257]))
    k = int(rng.choice([1, 2, 10, 100, 333]))
    rows = orc.synth_rows(5000 + seed, 0, n, dim) if dim > 1 else rng.standard_normal((n, 1)).astype(np.float32)
    if n > 8:
        rows[rng.integers(0, n, 3)] = rows[rng.integers(0, n, 3)]  # duplicates
        rows[int(rng.integers(0, n))] = 0.0                         # NULL cosine distance
    if dt != pvs.I8 and n > 4:
        rows[int(rng.integers(0, n))] *= np.float32(300.0 if dt == pvs.F16 else 1e18)
        rows[int(rng.integers(0, n))] *= np.float32(1e-3 if dt == pvs.F16 else 1e-18)
    queries = orc.synth_rows(6000 + seed, 0, batch, dim) if dim > 1 else rng.standard_normal((batch, 1)).astype(np.float32)
    scale = orc.compute_int8_scale(rows)
    ix = make_index(pvs, dt, rows, scale)
    hc = host_corpus(dt, rows, scale)
    hq = orc.quantize_int8(queries, scale) if dt == pvs.I8 else queries
    _check(pvs, ix, dt, metric, hc, hq, k)
    got = ix.score_batch(hq[:3], metric)
    for qq in range(min(3, batch)):
        exp = orc.score_all(dt, metric, hc, hq[qq])
        assert np.array_equal(np.isnan(got[:, qq]), np.isnan(exp))
        ok = ~np.isnan(exp)
        assert np.array_equal(got[ok, qq].view(np.uint32), exp[ok].view(np.uint32))
    ix.close()


@pytest.mark.parametrize("dtype", ["f16", "f32"])
def test_non_finite_components(pvs, dtype):
    """NaN and infinite components in stored rows and in queries: whatever IEEE arithmetic makes of them in the
    reference's loops (NaN distances = SQL NULL sort last) must come out of the device path too."""
    dt = pvs.F16 if dtype == "f16" else pvs.F32
    n, dim = 3000, 96
    rows = unit_rows(111, n, dim)
    rows[5, 7] = np.nan
    rows[17, 0] = np.inf
    rows[33, 95] = -np.inf
    rows[40, :] = np.inf
    q = orc.synth_rows(112, 0, 4, dim)
    q[1, 3] = np.inf
    q[2, 10] = np.nan
    ix = make_index(pvs, dt, rows, None)
    hc = host_corpus(dt, rows, None)
    for metric in (pvs.COSINE, pvs.L2):
        _check(pvs, ix, dt, metric, hc, q, 20)
        _check(pvs, ix, dt, metric, hc, q[:1], n)  # the whole corpus: NULL rows fill the tail in id order
    ix.close()


@pytest.mark.parametrize("dtype,n,b", [("i8", 10_000_000, 128), ("f16", 1_000_000, 32), ("f32", 1_000_000, 8), ("i8", 10_000_000, 256),
                                       ("f16", 10_000_000, 1), ("f32", 10_000_000, 128)])
def test_full_size_properties(pvs, dtype, n, b):
    """BASELINE configs[2] (10M x 768 int8, 128 queries; also 256 queries = one pass of the 8-wave instance), configs[1]
    (1M x 768 f16, 32 queries; also as f32, the reference's exact mode) and the north star's single-query 10M x 768 f16
    shape at full size, k = 100, checked through size-independent
    properties: pages sorted by (distance, id), ids unique and in range, full pages, two runs bit-identical, the
    filter path and the dense path (two different algorithms on the device) agree on a few queries, the page is a
    prefix of the k = 400 page, every returned distance equals the dense `d` column at that row, and row shards
    merged with pvs_merge_topk reproduce the whole-corpus page; for the int8 shapes the first and the last query of the batch
    also against the CPU oracle over all 10M rows."""
    from panoptikon_amd import _lib as L

    lib = pvs.lib()
    dim, k = 768, 100
    dt = {"i8": pvs.I8, "f16": pvs.F16, "f32": pvs.F32}[dtype]
    ix = pvs.VectorIndex(dt, dim, capacity_rows=n)
    stage = pvs.DeviceBuffer(1_000_000 * dim * 4)
    if dt == pvs.I8:
        L.check(lib.pvs_synth_rows_f32(0, 20260928, 0, 1_000_000, dim, stage.ptr))
        amax = L.C.c_float()
        L.check(lib.pvs_absmax(stage.ptr, 1_000_000 * dim, L.DEVICE, 0, L.C.byref(amax)))
        ix.set_scale(pvs.scale_from_absmax(float(amax.value) * 1.05))
    for off in range(0, n, 1_000_000):
        L.check(lib.pvs_synth_rows_f32(0, 20260928, off, 1_000_000, dim, stage.ptr))
        ix.add_f32((stage, 1_000_000))
    stage.free()
    q = orc.synth_rows(0x5EED0000, 0, b, dim)
    for metric in (pvs.COSINE, pvs.L2):
        dense_before = ix.stats().dense_queries
        gi, gd, gc = ix.search(q, k, metric)
        assert (gc == k).all() and gi.min() >= 0 and gi.max() < n
        for r in range(b):
            assert len(set(gi[r].tolist())) == k
            key = list(zip(gd[r].tolist(), gi[r].tolist()))
            assert key == sorted(key), r
        gi2, gd2, _ = ix.search(q, k, metric)
        assert np.array_equal(gi, gi2) and np.array_equal(gd.view(np.uint32), gd2.view(np.uint32))
        wi, wd, _ = ix.search(q[:8], 400, metric)
        assert np.array_equal(wi[:, :k], gi[:8]) and np.array_equal(wd[:, :k].view(np.uint32), gd[:8].view(np.uint32))
        assert ix.stats().dense_queries == dense_before, "the filter path must serve this shape"
        ix.set_path(1)
        di, dd, _ = ix.search(q[:3], k, metric)
        ix.set_path(0)
        assert np.array_equal(di, gi[:3]) and np.array_equal(dd.view(np.uint32), gd[:3].view(np.uint32))
        col = ix.score_all(q[0], metric)
        assert np.array_equal(col[gi[0]].view(np.uint32), gd[0].view(np.uint32))
        assert (col >= gd[0, -1]).sum() >= n - k  # nothing outside the page beats its last entry
        if dt == pvs.I8 and metric == pvs.COSINE:
            # the CPU oracle over the WHOLE corpus for EIGHT queries spread over the batch — one per wave of the 128- / 256-query
            # kernel, the first and the last of the batch among them (round 6; two until then): ids and int8 distances bit for bit
            scale = ix.stats().scale
            which = sorted({0, b - 1, *[(t * b) // 7 + (t % 3) for t in range(1, 7)]})[:8]
            which = [min(x, b - 1) for x in which]
            qc = orc.quantize_int8(q[which], scale)
            acc_i = [np.empty(0, np.int64) for _ in which]
            acc_d = [np.empty(0, np.float32) for _ in which]
            for off in range(0, n, 1_000_000):
                slab = ix.read_rows(off, 1_000_000)
                ci, cd = orc.search(orc.I8, orc.COSINE, slab, qc, k, ids=np.arange(off, off + 1_000_000, dtype=np.int64), threads=orc.max_threads())
                for t in range(len(which)):
                    acc_i[t], acc_d[t] = orc.topk(np.concatenate([acc_d[t], cd[t]]), k, ids=np.concatenate([acc_i[t], ci[t]]))
            for t, qi in enumerate(which):
                assert np.array_equal(gi[qi], acc_i[t]) and np.array_equal(gd[qi].view(np.uint32), acc_d[t].view(np.uint32)), f"query {qi} vs the oracle over {n} rows"
        if dt != pvs.I8 and n == 1_000_000:
            # configs[1] (1M x 768 f16 x 32) and its f32 form: the CPU oracle over the WHOLE corpus (sequential-f32 restatement of
            # sqlite-vec's scalar loops, ~3 s), both metrics, every query of the f32 batch / 8 queries spread over the f16 one
            which = list(range(b)) if b <= 8 else [0, 5, 11, 16, 21, 26, 30, b - 1]
            slab = ix.read_rows(0, n)
            odt = orc.F16 if dt == pvs.F16 else orc.F32
            omet = orc.COSINE if metric == pvs.COSINE else orc.L2
            ei, ed = orc.search(odt, omet, slab, q[which], k, threads=orc.max_threads())
            for t, qi in enumerate(which):
                assert np.array_equal(gi[qi], ei[t]) and np.array_equal(gd[qi].view(np.uint32), ed[t].view(np.uint32)), f"{dtype} query {qi} vs the oracle over {n} rows"
        if dt == pvs.F16 and n == 10_000_000 and metric == pvs.COSINE:
            # the north star's HBM shape (one query over 10M x 768 f16) against the CPU oracle over the WHOLE corpus, chunk by chunk
            # (round 5: until then this shape was checked against the device's dense path only)
            acc_i, acc_d = np.empty(0, np.int64), np.empty(0, np.float32)
            for off in range(0, n, 1_000_000):
                ci, cd = orc.search(orc.F16, orc.COSINE, ix.read_rows(off, 1_000_000), q[:1], k, ids=np.arange(off, off + 1_000_000, dtype=np.int64),
                                    threads=orc.max_threads())
                acc_i, acc_d = orc.topk(np.concatenate([acc_d, cd[0]]), k, ids=np.concatenate([acc_i, ci[0]]))
            assert np.array_equal(gi[0], acc_i) and np.array_equal(gd[0].view(np.uint32), acc_d.view(np.uint32)), f"f16 single query vs the oracle over {n} rows"
    # row shards of the same corpus, merged: equals the whole-corpus page (ids are global row indexes)
    nq = min(16, b)
    gi, gd, gc = ix.search(q[:nq], k, pvs.COSINE)
    pages_i, pages_d, pages_c = [], [], []
    for r0, r1 in (pvs.shard_range(n, 4, r) for r in range(4)):
        sh = pvs.VectorIndex(dt, dim, capacity_rows=r1 - r0, id_base=r0)
        if dt == pvs.I8:
            sh.set_scale(ix.stats().scale)
        for off in range(r0, r1, 500_000):
            m = min(500_000, r1 - off)
            sh.add(ix.read_rows(off, m))
        si, sd, sc = sh.search(q[:nq], k, pvs.COSINE)
        pages_i.append(si), pages_d.append(sd), pages_c.append(sc)
        sh.close()
    mi, md, mc = pvs.merge_topk(np.stack(pages_i), np.stack(pages_d), np.stack(pages_c), k)
    assert np.array_equal(mi, gi) and np.array_equal(md.view(np.uint32), gd.view(np.uint32))
    ix.close()


@pytest.mark.parametrize("dtype", ["i8", "f16", "f32"])
def test_search_filtered_by_a_candidate_mask(pvs, dtype):
    """pvs_search_filtered = the vector filter joined to a context CTE (`WHERE begin_cte.item_id IS NOT NULL`,
    image_embeddings.rs:140-199): only rows the query's other filters left are candidates.  Dense masks, sparse
    masks (fewer candidates than k: the page is short and never padded with rows outside the mask), an empty mask,
    masks that exclude the true nearest neighbours; pages equal the oracle's over the allowed rows only."""
    dt = {"i8": pvs.I8, "f16": pvs.F16, "f32": pvs.F32}[dtype]
    rng = np.random.default_rng(47)
    n, dim, k = 30000, 384, 50
    rows = unit_rows(121, n, dim)
    rows[123] = 0.0  # a NULL cosine distance among the candidates
    scale = orc.compute_int8_scale(rows)
    ix = make_index(pvs, dt, rows, scale)
    hc = host_corpus(dt, rows, scale)
    q = orc.synth_rows(122, 0, 6, dim)
    hq = orc.quantize_int8(q, scale) if dt == pvs.I8 else q
    full = orc.search(dt, pvs.COSINE, hc, hq, k)
    masks = {"half": rng.random(n) < 0.5, "1pct": rng.random(n) < 0.01, "30rows": np.zeros(n, bool), "none": np.zeros(n, bool),
             "all": np.ones(n, bool), "no_top": np.ones(n, bool), "tile": (np.arange(n) // 32) % 7 == 3}
    masks["30rows"][rng.choice(n, 30, replace=False)] = True
    masks["30rows"][123] = True
    masks["half"][123] = True
    masks["no_top"][full[0][:, :10].ravel()] = False  # every query loses its 10 best rows
    for name, m in masks.items():
        allowed = np.nonzero(m)[0]
        for metric in (pvs.COSINE, pvs.L2):
            gi, gd, gc = ix.search_filtered(hq, k, m, metric)
            if len(allowed) == 0:
                assert (gc == 0).all() and (gi == -1).all()
                continue
            ei, ed = orc.search(dt, metric, hc[allowed], hq, k, ids=allowed.astype(np.int64))
            w = ei.shape[1]
            assert gc.tolist() == [w] * len(hq), (name, metric)
            assert np.array_equal(gi[:, :w], ei), (name, metric)
            a, b = gd[:, :w], ed
            assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)].view(np.uint32), b[~np.isnan(b)].view(np.uint32))
            assert (gi[:, w:] == -1).all()
    # the unfiltered entry point is untouched by a previous filtered search on the same context
    gi, gd, gc = ix.search(hq, k, pvs.COSINE)
    assert np.array_equal(gi, full[0])
    ix.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("PVS_FUZZ_SEEDS", "18"))))
def test_randomized_item_level_operations(pvs, seed):
    """Seeded sweep over the per-item entry points: GROUP BY aggregates (MIN through the filter scan, MAX / AVG /
    weighted dense), filtered search, similar_to with random options, device RRF over two random branches."""
    rng = np.random.default_rng(7000 + seed)
    dt = [pvs.I8, pvs.F16, pvs.F32][seed % 3]
    metric = [pvs.COSINE, pvs.L2][(seed // 3) % 2]
    n = int(rng.choice([40, 333, 2049, 6000]))
    dim = int(rng.choice([24, 100, 256, 513, 768]))
    n_groups = int(rng.choice([1, 7, n // 3 + 1, n]))
    k = int(rng.choice([1, 5, 60]))
    rows = orc.synth_rows(8000 + seed, 0, n, dim)
    rows[int(rng.integers(0, n))] = 0.0
    groups = np.sort(rng.integers(0, n_groups, n)).astype(np.int64) * 5 + 2
    ids = np.cumsum(rng.integers(1, 4, n)).astype(np.int64)
    scale = orc.compute_int8_scale(rows)
    ix = pvs.VectorIndex(dt, dim)
    if dt == pvs.I8:
        ix.set_scale(scale)
    ix.add_f32(rows, row_ids=ids, group_ids=groups)
    hc = host_corpus(dt, rows, scale)
    q = orc.synth_rows(9000 + seed, 0, 3, dim)
    hq = orc.quantize_int8(q, scale) if dt == pvs.I8 else q
    w = (rng.random(n) + 0.05).astype(np.float32)
    for agg, oagg, ww in ((pvs.AGG_MIN, orc.AGG_MIN, None), (pvs.AGG_MAX, orc.AGG_MAX, None), (pvs.AGG_AVG, orc.AGG_AVG, None),
                          (pvs.AGG_AVG, orc.AGG_AVG, w)):
        gg, gv, gc = ix.search_groups(hq, k, metric, agg, row_weights=ww)
        for b in range(3):
            eg, ev = orc.search_groups(dt, metric, hc, hq[b], groups, oagg, k, weights=ww)
            assert gc[b] == len(eg) and np.array_equal(gg[b, : len(eg)], eg), (seed, agg, b)
            a, e = gv[b, : len(eg)], ev
            assert np.array_equal(np.isnan(a), np.isnan(e)) and np.array_equal(a[~np.isnan(a)], e[~np.isnan(e)]), (seed, agg, b)
    m = rng.random(n) < rng.choice([0.02, 0.3, 0.9])
    gi, gd, gc = ix.search_filtered(hq, k, m, metric)
    allowed = np.nonzero(m)[0]
    if len(allowed):
        ei, ed = orc.search(dt, metric, hc[allowed], hq, k, ids=ids[allowed])
        wdt = ei.shape[1]
        assert gc.tolist() == [wdt] * 3 and np.array_equal(gi[:, :wdt], ei)
        assert np.array_equal(np.isnan(gd[:, :wdt]), np.isnan(ed))
    else:
        assert (gc == 0).all()
    # per-item pages under the same candidate mask: groups without a candidate row are absent, the others aggregate
    # their candidate rows only
    if len(allowed):
        for agg, oagg, ww in ((pvs.AGG_MIN, orc.AGG_MIN, None), (pvs.AGG_AVG, orc.AGG_AVG, None), (pvs.AGG_MAX, orc.AGG_MAX, w)):
            gg, gv, gc = ix.search_groups_filtered(hq, k, m, metric, agg, row_weights=ww)
            for b in range(3):
                eg, ev = orc.search_groups(dt, metric, hc[allowed], hq[b], groups[allowed], oagg, k, weights=None if ww is None else ww[allowed])
                assert gc[b] == len(eg) and np.array_equal(gg[b, : len(eg)], eg), (seed, "filtered groups", agg, b)
                a_, e_ = gv[b, : len(eg)], ev
                assert np.array_equal(np.isnan(a_), np.isnan(e_)) and np.array_equal(a_[~np.isnan(a_)], e_[~np.isnan(e_)])
    tg = groups[int(rng.integers(0, n))]
    targets = [int(t) for t in np.nonzero(groups == tg)[0]][:8]
    kind = (rng.random(n) < 0.5).astype(np.uint8)
    i2i, t2t = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    oagg = [orc.AGG_MIN, orc.AGG_MAX, orc.AGG_AVG][seed % 3]
    gg, gv = ix.similar_to_ex(ids[targets], k, metric, [pvs.AGG_MIN, pvs.AGG_MAX, pvs.AGG_AVG][seed % 3], row_kind=kind, xmodal_i2i=i2i, xmodal_t2t=t2t)
    eg, ev = orc.similar_to_ex(dt, metric, hc, targets, groups, oagg, k, kind=kind, xmodal_i2i=i2i, xmodal_t2t=t2t)
    assert np.array_equal(gg, eg), seed
    assert np.array_equal(np.isnan(gv), np.isnan(ev)) and np.array_equal(gv[~np.isnan(gv)], ev[~np.isnan(ev)])
    # RRF over this index under both metrics as two branches
    br = [dict(index=ix, query=hq[0], metric=pvs.COSINE, agg=pvs.AGG_MIN, rrf_k=int(rng.integers(1, 70)), weight=1.0),
          dict(index=ix, query=hq[1], metric=pvs.L2, agg=pvs.AGG_AVG, descending=bool(rng.integers(0, 2)), rrf_k=3, weight=0.5)]
    ob = [dict(dtype=dt, metric=orc.COSINE, corpus=hc, query=hq[0], groups=groups, agg=orc.AGG_MIN, rrf_k=br[0]["rrf_k"], weight=1.0),
          dict(dtype=dt, metric=orc.L2, corpus=hc, query=hq[1], groups=groups, agg=orc.AGG_AVG, descending=br[1]["descending"], rrf_k=3, weight=0.5)]
    gg, gs = pvs.rrf_search(br, k)
    eg, es = orc.rrf_search(ob, k)
    assert np.array_equal(gg, eg) and np.array_equal(gs.view(np.uint64), es.view(np.uint64)), seed
    ix.close()


@pytest.mark.parametrize("dtype", ["f16", "f32"])
def test_near_tied_scores_with_coherent_rounding(pvs, dtype):
    """The filter scores float rows through narrowed images (f16 query image; f32 rows truncated to f16): here every
    component and every query component is positive, so the narrowing errors of a row all push its dot product the
    same way, and thousands of rows score within 1e-4 of each other — a page is exact only if the interval the
    filter puts around its key really contains the error (checked by hand: with the f32 bound set 30x too small this
    test fails; the factor-of-two questions are settled by the analysis in HISTORY.md §4.2, not by tests)."""
    dt = pvs.F16 if dtype == "f16" else pvs.F32
    rng = np.random.default_rng(53)
    n, dim, k = 12000, 384, 60  # (a row pitch both dtypes have a scan instance for)
    base = (np.abs(rng.standard_normal(dim)) + 0.5).astype(np.float32)
    rows = np.abs(rng.standard_normal((n, dim))).astype(np.float32) + 0.1           # far from the queries ...
    near = rng.choice(n, 2000, replace=False)                                        # ... except 2000 near-tied rows
    rows[near] = base[None, :] * (1.0 + 2e-4 * rng.standard_normal((2000, dim))).astype(np.float32)
    rows *= (2.0 ** rng.integers(-6, 7, n)).astype(np.float32)[:, None]  # norms over 12 octaves (cosine ignores them)
    # odd rows: mantissas just below the next f16 grid point (truncation loses almost a full unit in the last place,
    # their scan keys come out ~1e-3 pessimistic); even rows: exactly representable (no error at all).  The true
    # neighbours are spread over both kinds, so rows must not be dropped for looking 1e-3 worse than their peers.
    bits = rows.view(np.uint32).copy()
    bits[1::2] |= np.uint32(0x1FFF)
    bits[0::2] &= np.uint32(0xFFFFE000)
    rows = bits.view(np.float32)
    q = (base[None, :] * (1.0 + 1e-3 * rng.standard_normal((5, dim)))).astype(np.float32)
    ix = make_index(pvs, dt, rows, None)
    hc = host_corpus(dt, rows, None)
    _check(pvs, ix, dt, pvs.COSINE, hc, q, k)
    assert ix.stats().dense_queries == 0, "the filter path itself must get this right"
    _check(pvs, ix, dt, pvs.L2, hc, q, k)
    ix.close()


@pytest.mark.parametrize("dtype", ["i8", "f16", "f32"])
def test_search_bounded_applies_sort_bounds_on_the_distance(pvs, dtype):
    """apply_sort_bounds (pql/builder.rs:781-815): `WHERE order_rank > gt AND order_rank < lt` over the distance column, then
    the page.  Oracle: every row's exact distance, the SQL predicate in f64, (distance, id) order."""
    pdt = {"i8": pvs.I8, "f16": pvs.F16, "f32": pvs.F32}[dtype]
    odt = {"i8": orc.I8, "f16": orc.F16, "f32": orc.F32}[dtype]
    n, dim, k = 5000, 200, 60
    rows = unit_rows(91, n, dim)
    rows[17] = 0.0  # NULL cosine distance: never inside a bound
    scale = orc.compute_int8_scale(rows) if dtype == "i8" else None
    ix = make_index(pvs, pdt, rows, scale)
    corpus = host_corpus(odt, rows, scale)
    qs = orc.synth_rows(92, 0, 3, dim)
    hq = orc.quantize_int8(qs, scale) if dtype == "i8" else qs
    for metric, om in ((pvs.COSINE, orc.COSINE), (pvs.L2, orc.L2)):
        for qi in range(len(qs)):
            d = orc.score_all(odt, om, corpus, hq[qi])
            lo, hi = np.nanquantile(d, 0.01), np.nanquantile(d, 0.03)
            for gt, lt in ((None, float(hi)), (float(lo), None), (float(lo), float(hi)), (float(hi), float(lo)), (None, None)):
                keep = np.array([orc.sort_bounds_keep(float(x), gt, lt) for x in d.astype(np.float64)]) if (gt is not None or lt is not None) \
                    else np.ones(n, bool)
                ids = np.flatnonzero(keep)
                ei, ed = orc.topk(d[ids], k, ids=ids.astype(np.int64))
                gi, gd, gc = ix.search_bounded(qs[qi], k, metric, gt=gt, lt=lt)
                assert gc[0] == len(ei), (dtype, metric, gt, lt)
                assert np.array_equal(gi[0, : len(ei)], ei)
                got = gd[0, : len(ei)]
                assert np.array_equal(np.isnan(got), np.isnan(ed)) and np.array_equal(got[~np.isnan(got)].view(np.uint32), ed[~np.isnan(ed)].view(np.uint32))
                assert (gi[0, len(ei):] == -1).all()
    ix.close()


def test_search_bounded_with_a_lower_bound_pages_through_the_filter_scan(pvs):
    """`order_rank > gt` as a cursor (the last distance of an earlier page): the rows beyond it are a suffix of the plain
    ordering, so growing pages of the filter scan answer it; only a bound deeper than 4,096 rows needs the dense path.  A batch
    with bounds at depth 0, 7, 150, 3,000 and 20,000 rows (one per query), with and without `lt`, ties on the bound itself
    (int8 L2), and the same on a multi-device index."""
    rng = np.random.default_rng(8)
    n, dim, k = 300_000, 64, 40
    rows = unit_rows(95, n, dim)
    rows[:64] = rows[64:128]                      # duplicates: distances tie, also exactly on a bound
    scale = orc.compute_int8_scale(rows)
    codes = orc.quantize_int8(rows, scale)
    qs = orc.synth_rows(96, 0, 5, dim)
    hq = orc.quantize_int8(qs, scale)
    depth = [0, 7, 150, 3000, 20000]
    for devices in (None, [0, 0]):
        ix = pvs.VectorIndex(pvs.I8, dim, devices=devices) if devices else pvs.VectorIndex(pvs.I8, dim)
        ix.set_scale(scale)
        ix.add_f32(rows)
        for metric, om in ((pvs.L2, orc.L2), (pvs.COSINE, orc.COSINE)):
            ds = [orc.score_all(orc.I8, om, codes, hq[j]) for j in range(5)]
            # one call per query (each with its own bound), then the whole batch with one common bound
            before = ix.stats().dense_queries
            for j in range(5):
                srt = np.sort(ds[j])
                gt = float(srt[depth[j]]) if depth[j] else float(srt[0]) - 1.0
                for lt in (None, float(srt[depth[j] + 25])):
                    keep = np.array([orc.sort_bounds_keep(float(x), gt, lt) for x in ds[j].astype(np.float64)])
                    ids = np.flatnonzero(keep)
                    ei, ed = orc.topk(ds[j][ids], k, ids=ids.astype(np.int64))
                    gi, gd, gc = ix.search_bounded(qs[j], k, metric, gt=gt, lt=lt)
                    assert gc[0] == len(ei) and np.array_equal(gi[0, : len(ei)], ei), (devices, metric, j, lt)
                    assert np.array_equal(gd[0, : len(ei)].view(np.uint32), ed.view(np.uint32))
                    assert (gi[0, len(ei):] == -1).all()
            if devices is None:
                assert ix.stats().dense_queries - before <= 2, "only the 20,000-deep bound (without lt) may need the dense path"
            gt = float(np.sort(ds[2])[150])
            gi, gd, gc = ix.search_bounded(qs, k, metric, gt=gt)
            for j in range(5):
                ids = np.flatnonzero(ds[j].astype(np.float64) > gt)
                ei, ed = orc.topk(ds[j][ids], k, ids=ids.astype(np.int64))
                assert gc[j] == len(ei) and np.array_equal(gi[j, : len(ei)], ei), (devices, metric, "batch", j)
        ix.close()


def test_rrf_bounded_fusion_equals_the_full_ranking(pvs):
    """pvs_rrf_search's bounded path (pages of each branch's window ranking, exact ranks of the candidates by one counting pass,
    the bound sum_b w_b/(k_b + R_b + 1) on everything outside the pages) must return exactly what ranking every group returns:
    the oracle's literal composition.  Three branches over >= 65,536 groups, partial overlap, NULL aggregates first in the
    ascending window, a descending window, production weights."""
    rng = np.random.default_rng(77)
    n_files = 120_000
    specs = [(pvs.I8, orc.I8, 64, 150_000, pvs.COSINE, orc.COSINE), (pvs.F16, orc.F16, 96, 110_000, pvs.L2, orc.L2),
             (pvs.F32, orc.F32, 32, 90_000, pvs.COSINE, orc.COSINE)]
    dev, ora = [], []
    for i, (dt, odt, dim, n, m, om) in enumerate(specs):
        rows = unit_rows(301 + i, n, dim)
        pool = np.arange(0, n_files, dtype=np.int64)[rng.random(n_files) < (0.9, 0.7, 0.5)[i]]
        groups = np.sort(rng.choice(pool, n)).astype(np.int64)
        if m == pvs.COSINE:
            for gz in (groups[n // 3], groups[n // 2]):
                rows[np.nonzero(groups == gz)[0]] = 0.0  # whole files of zero vectors: NULL aggregates, ranked FIRST ascending
        scale = orc.compute_int8_scale(rows)
        ix = pvs.VectorIndex(dt, dim)
        if dt == pvs.I8:
            ix.set_scale(scale)
        ix.add_f32(rows, group_ids=groups)
        q = orc.synth_rows(400 + i, 0, 1, dim)[0]
        hq = orc.quantize_int8(q[None, :], scale)[0] if dt == pvs.I8 else q
        agg = (pvs.AGG_MIN, pvs.AGG_AVG, pvs.AGG_MIN)[i]
        desc = i == 2
        rk, wt = ((5, 1.0), (5, 1.0), (10, 0.7))[i]
        dev.append(dict(index=ix, query=hq, metric=m, agg=agg, descending=desc, rrf_k=rk, weight=wt))
        ora.append(dict(dtype=odt, metric=om, corpus=host_corpus(dt, rows, scale), query=hq, groups=groups, agg=agg, descending=desc,
                        rrf_k=rk, weight=wt))
    for k, nb, path in ((1, 2, 1), (100, 2, 1), (100, 3, 1), (1000, 3, 1), (100, 1, 1)):
        gg, gs = pvs.rrf_search(dev[:nb], k)
        assert pvs.lib().pvs_rrf_last_path() == path, (k, nb, pvs.lib().pvs_rrf_last_path())
        eg, es = orc.rrf_search(ora[:nb], k)
        assert np.array_equal(gg, eg), (k, nb)
        assert np.array_equal(gs.view(np.uint64), es.view(np.uint64)), (k, nb)
        # round 4: the first round runs on the device (pvs_rrf_device.hip); the host form of the rounds returns the same page
        pvs.debug_set("rrf_host_rounds", 1)
        try:
            hg, hs = pvs.rrf_search(dev[:nb], k)
        finally:
            pvs.debug_set("rrf_host_rounds", 0)
        assert pvs.lib().pvs_rrf_last_path() == path
        assert np.array_equal(hg, gg) and np.array_equal(hs.view(np.uint64), gs.view(np.uint64)), (k, nb, "host rounds")
    # a negative weight voids the bound: the full ranking answers, same contract
    dev[1]["weight"], ora[1]["weight"] = -0.25, -0.25
    gg, gs = pvs.rrf_search(dev[:2], 50)
    assert pvs.lib().pvs_rrf_last_path() == 2
    eg, es = orc.rrf_search(ora[:2], 50)
    assert np.array_equal(gg, eg) and np.array_equal(gs.view(np.uint64), es.view(np.uint64))
    for b in dev:
        b["index"].close()


def test_device_reproduces_the_frozen_fixture_vi(pvs):
    """SURVEY §8c fixture (vi) — frozen bytes in tests/golden/fixture_vi.npz, not computed from the oracle at test time: the filter
    scan and the dense path must both return those ids and those f32 distances for {f32, f16, i8} x {cosine, L2}."""
    import zlib

    f = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fixture_vi.npz"))
    n, dim, k = int(f["n"]), int(f["dim"]), int(f["k"])
    rows = orc.synth_rows(int(f["seed_rows"]), 0, n, dim)
    queries = orc.synth_rows(int(f["seed_queries"]), 0, 8, dim)
    assert zlib.crc32(rows.tobytes()) == int(f["rows_crc32"])
    scale = float(f["scale"])
    assert pvs.scale_from_absmax(pvs.absmax(rows)) == np.float32(scale)
    for name, dt in (("f32", pvs.F32), ("f16", pvs.F16), ("i8", pvs.I8)):
        ix = make_index(pvs, dt, rows, scale)
        for mname, m in (("cosine", pvs.COSINE), ("l2", pvs.L2)):
            for path in (0, 1):
                ix.set_path(path)
                ids, dist, cnt = ix.search(queries, k, m)  # f32 queries: an int8 index quantizes them on the device (compute_query_quant)
                assert np.array_equal(ids, f[f"{name}_{mname}_ids"]), (name, mname, path)
                assert np.array_equal(dist.view(np.uint32), f[f"{name}_{mname}_dist"].view(np.uint32)), (name, mname, path)
        ix.close()


@pytest.mark.parametrize("dtype", ["f16", "f32"])
def test_float_distances_within_1e5_of_the_f64_formula(pvs, dtype):
    """BASELINE.json north_star: fp32 / fp16 cosine within 1e-5 relative of the reference.  Asserted here WITHOUT the oracle:
    returned distances against numpy f64 evaluations of 1 - a.q/(|a||q|) and |a - q| over the stored values, and the returned
    ids against the f64 ranking wherever the f64 gap to the next row exceeds the tolerance."""
    pdt = pvs.F16 if dtype == "f16" else pvs.F32
    n, dim, k = 60_000, 768, 50
    rows = unit_rows(123, n, dim)
    stored = (rows.astype(np.float16) if dtype == "f16" else rows).astype(np.float64)
    ix = make_index(pvs, pdt, rows)
    qs = orc.synth_rows(124, 0, 6, dim)
    qf = qs.astype(np.float64)
    for m in (pvs.COSINE, pvs.L2):
        ids, dist, cnt = ix.search(qs, k, m)
        for qi in range(len(qs)):
            if m == pvs.COSINE:
                full = 1.0 - (stored @ qf[qi]) / (np.linalg.norm(stored, axis=1) * np.linalg.norm(qf[qi]))
            else:
                full = np.linalg.norm(stored - qf[qi], axis=1)
            exact = full[ids[qi]]
            assert np.all(np.abs(dist[qi] - exact) <= REL_TOL * np.abs(exact)), (dtype, m, qi, float(np.max(np.abs(dist[qi] - exact) / np.abs(exact))))
            order = np.argsort(full, kind="stable")[: k + 1]
            gaps = np.diff(full[order])
            for j in range(k):
                if gaps[j] > 2 * REL_TOL * full[order[j]] and (j == 0 or gaps[j - 1] > 2 * REL_TOL * full[order[j]]):
                    assert ids[qi, j] == order[j], (dtype, m, qi, j)
    ix.close()


def test_ingest_codec_is_bit_exact_on_rounding_boundaries(pvs):
    """The build-side codec (k_rows_ingest4: multiply by 1/s on the common path, IEEE division only near half-integers) against
    quantize_int8's definition clamp(round_ties_even(x / s)) (db/vector_quants.rs:1489-1497): quotients on, just below and just
    above every half-integer and integer in the code range and beyond, for awkward scales, plus NaN / inf / huge values."""
    rng = np.random.default_rng(17)
    dim = 256
    for scale in (np.float32(3.0 / 127), np.float32(11.0 / 127), np.float32(0.0015056864), np.float32(1.0), np.float32(7.3e-5), np.float32(123.456)):
        m = np.arange(-135, 136, dtype=np.float64)
        targets = np.concatenate([m, m + 0.5, m + 0.25])
        base = (targets * np.float64(scale)).astype(np.float32)
        vals = [base]
        for step in (1, 2, 3, 8):
            up, dn = base.copy(), base.copy()
            for _ in range(step):
                up = np.nextafter(up, np.float32(np.inf))
                dn = np.nextafter(dn, np.float32(-np.inf))
            vals += [up, dn]
        x = np.concatenate(vals + [np.array([np.nan, np.inf, -np.inf, 3e38, -3e38, 0.0, -0.0, 1e-45, -1e-45], np.float32),
                                   (rng.standard_normal(5000) * 40 * scale).astype(np.float32)])
        n = (len(x) + dim - 1) // dim
        x = np.concatenate([x, np.zeros(n * dim - len(x), np.float32)]).reshape(n, dim)
        ix = pvs.VectorIndex(pvs.I8, dim)
        ix.set_scale(float(scale))
        ix.add_f32(x)
        got = ix.read_rows(0, n)
        exp = orc.quantize_int8(x, float(scale))
        bad = np.argwhere(got != exp)
        assert len(bad) == 0, (float(scale), x[tuple(bad[0])], got[tuple(bad[0])], exp[tuple(bad[0])])
        ix.close()
    # f16 ingest (round to nearest even) through the same kernel shape
    x = (rng.standard_normal((300, dim)) * 10).astype(np.float32)
    x[0, :8] = [65504.0, 65520.0, 1e-8, 6.1e-5, np.nan, np.inf, -0.0, 5.96e-8]
    ix = pvs.VectorIndex(pvs.F16, dim)
    ix.add_f32(x)
    with np.errstate(over="ignore"):
        exp = x.astype(np.float16)
    got = ix.read_rows(0, 300)
    assert np.array_equal(got.view(np.uint16)[~np.isnan(exp)], exp.view(np.uint16)[~np.isnan(exp)]) and np.isnan(got[np.isnan(exp)]).all()
    ix.close()


@pytest.mark.parametrize("dtype", ["i8", "f16", "f32"])
def test_batched_dense_select_on_massive_ties_and_nulls(pvs, dtype):
    """The dense path for many queries at once (pvs_select.hip): every query scored per corpus pass, pages by an exact radix
    select with ties cut by row.  Worst cases for a select: all distances equal (duplicated corpus), two-valued distances, NULL
    rows needed to fill the page, fewer rows than k, a candidate mask — all against the oracle; and the round-1 per-query form
    must agree (same kernel inputs, full sort)."""
    pdt = {"i8": pvs.I8, "f16": pvs.F16, "f32": pvs.F32}[dtype]
    odt = {"i8": orc.I8, "f16": orc.F16, "f32": orc.F32}[dtype]
    rng = np.random.default_rng(3)
    dim, n = 64, 30_000
    base = unit_rows(211, 7, dim)
    rows = base[rng.integers(0, 7, n)]          # 7 distinct vectors: every query sees 7 tie classes of ~4,300 rows
    rows[1000:1100] = 0.0                       # NULL cosine distances
    rows[5] = unit_rows(212, 1, dim)[0]         # one unique row
    scale = orc.compute_int8_scale(rows) if dtype == "i8" else None
    corpus = host_corpus(odt, rows, scale)
    qs = np.concatenate([base[:3], unit_rows(213, 9, dim)])
    hq = orc.quantize_int8(qs, scale) if dtype == "i8" else qs
    ix = make_index(pvs, pdt, rows, scale)
    for metric, om in ((pvs.COSINE, orc.COSINE), (pvs.L2, orc.L2)):
        for k in (1, 50, 5000):
            ei, ed = orc.search(odt, om, corpus, hq, k)
            for path in (0, 1):   # 0: the filter scan hands the tie-heavy queries back; 1: dense for everyone
                ix.set_path(path)
                gi, gd, gc = ix.search(qs, k, metric)
                assert (gc == min(k, n)).all()
                assert np.array_equal(gi[:, : ei.shape[1]], ei), (dtype, metric, k, path)
                g = gd[:, : ei.shape[1]]
                assert np.array_equal(np.isnan(g), np.isnan(ed)) and np.array_equal(g[~np.isnan(g)].view(np.uint32), ed[~np.isnan(ed)].view(np.uint32))
            ix.set_path(0)
    st = ix.stats()
    assert st.dense_queries > 0
    # a candidate mask that leaves fewer rows than k, and one that cuts through a tie class
    mask = np.zeros(n, np.uint8)
    mask[2000:2040] = 1
    mask[1050] = 1  # a NULL row among the candidates
    ix.set_path(1)
    gi, gd, gc = ix.search_filtered(qs, 64, mask, pvs.COSINE)
    d = np.stack([orc.score_all(odt, orc.COSINE, corpus, hq[i]) for i in range(len(qs))])
    cand = np.flatnonzero(mask)
    for i in range(len(qs)):
        ei, ed = orc.topk(d[i][cand], 64, ids=cand.astype(np.int64))
        assert gc[i] == len(cand) and np.array_equal(gi[i, : len(ei)], ei)
    ix.close()


def test_pagination_is_a_window_of_the_same_ordering(pvs):
    """pvs_search_page / pvs_search_groups_page (LIMIT ? OFFSET ?, pql/builder.rs:578-582): every page is the matching window of
    the oracle's ordering over the whole corpus — across the filter-path / dense-path boundary (k > 4096), past the end, with a
    tie block straddling a page boundary."""
    dim = 192
    rows = unit_rows(21, 9000, dim)
    rows[100:140] = rows[50]  # 41 equal rows: ties broken by id across page boundaries
    ix = make_index(pvs, pvs.F32, rows, None)
    groups = np.arange(len(rows), dtype=np.int64) // 4
    ixg = pvs.VectorIndex(pvs.F32, dim)
    ixg.add(rows, group_ids=groups)
    q = orc.synth_rows(0x5EED0123, 0, 3, dim)
    q[1] = rows[50]
    try:
        ei, ed = orc.search(orc.F32, orc.COSINE, rows, q, len(rows))
        for offset, limit in ((0, 25), (25, 25), (30, 20), (4090, 50), (8990, 25), (9000, 10), (20000, 5)):
            gi, gd, gc = ix.search_page(q, offset, limit, pvs.COSINE)
            have = max(0, min(len(rows) - offset, limit))
            assert gc.tolist() == [have] * 3, (offset, limit, gc)
            assert np.array_equal(gi[:, :have], ei[:, offset:offset + have]), (offset, limit)
            assert np.array_equal(gd[:, :have].view(np.uint32), ed[:, offset:offset + have].view(np.uint32))
            assert (gi[:, have:] == -1).all() and np.isnan(gd[:, have:]).all()
        n_groups = len(rows) // 4
        for qi in range(2):
            eg, ev = orc.search_groups(orc.F32, orc.L2, rows, q[qi], groups, orc.AGG_MIN, n_groups)
            for offset, limit in ((0, 10), (10, 10), (2240, 20)):
                gg, gv, gn = ixg.search_groups_page(q[qi:qi + 1], offset, limit, pvs.L2, pvs.AGG_MIN)
                have = max(0, min(n_groups - offset, limit))
                assert gn[0] == have and np.array_equal(gg[0, :have], eg[offset:offset + have])
                assert np.array_equal(gv[0, :have].view(np.uint64), ev[offset:offset + have].view(np.uint64))
        with pytest.raises(Exception):
            ix.search_page(q, 5, 0, pvs.COSINE)
    finally:
        ix.close()
        ixg.close()


@pytest.mark.parametrize("batch", [128, 256, 40])
def test_ties_clustered_in_a_few_tile_streams_are_rescanned_not_sent_to_the_dense_path(pvs, batch):
    """A corpus of few distinct vectors repeated with a period that is a multiple of the tile-stream count puts every copy of
    the best vector into the same handful of candidate segments: they overflow although the query's candidates fit one list.
    Such queries are handed back with need_dense = 2 and their chunk goes through the scan once more with pass B appending to
    flat per-query lists (pvs_stats.rescanned_queries), not through the dense path; the page is the oracle's — the copies with
    the lowest ids."""
    # period 16,384 rows = 256 tiles of 64 rows: one or two streams hold every copy.  (k is small on purpose: the copies also share ONE
    # of pass A's row groups, so the threshold is the k-th best SAMPLED vector — about the 16 k-th best overall — and everything
    # closer than that, times 100 copies, has to fit the 16,384-entry list; beyond that the dense path is the right answer.)
    dim, distinct, copies, k = 64, 16384, 100, 5
    base = unit_rows(777, distinct, dim)
    rows = np.tile(base, (copies, 1))
    scale = orc.compute_int8_scale(rows)
    ix = make_index(pvs, pvs.I8, rows, scale)
    corpus = host_corpus(orc.I8, rows, scale)
    # each query sits next to one stored vector, whose copies tie at the top of its page; the vectors are picked from tiles the
    # sample pass visits (tile index = 0 mod 16), so the threshold is the tie itself and the candidates are just the copies
    pick = 1024 * (np.arange(batch) % 16) + np.arange(batch) // 16
    qs = base[pick] + 0.01 * orc.synth_rows(778, 0, batch, dim)
    hq = orc.quantize_int8(qs, scale)
    for metric, om in ((pvs.L2, orc.L2), (pvs.COSINE, orc.COSINE)):
        before = ix.stats()
        gi, gd, gc = ix.search(qs, k, metric)
        after = ix.stats()
        ei, ed = orc.search(orc.I8, om, corpus, hq, k)
        assert (gc == k).all()
        assert np.array_equal(gi[:, :k], ei) and np.array_equal(gd[:, :k].view(np.uint32), ed.view(np.uint32))
        assert after.dense_queries == before.dense_queries, "no query may need the dense path"
        if batch >= 128:  # (k_scan's 64-slot segments at 512 streams see 50 copies each: no overflow at 40 queries)
            assert after.rescanned_queries - before.rescanned_queries == batch, "every query's ties must have overflowed a segment"
    ix.close()


def test_second_sort_key_orders_ties_like_the_reference(pvs):
    """`ORDER BY order_rank ASC NULLS LAST, last_modified DESC` (pql/model.rs:547-553; order_rank is the distance when row_n is
    off, model.rs:232): rows that tie on the distance come out newest first, then by id, and a tie at the k-th distance takes the
    newest rows.  int8 L2 over a corpus of few distinct vectors ties massively.  Every route to a row page must
    agree with the oracle's ordering: the filter scan (pass C), the per-query dense sort, the batched dense select; NULL
    distances stay last; appending rows drops the keys."""
    rng = np.random.default_rng(99)
    dim, distinct, copies = 96, 300, 40
    base = unit_rows(555, distinct, dim)
    rows = np.tile(base, (copies, 1))[rng.permutation(distinct * copies)]
    rows[123] = 0.0  # NULL cosine distance
    n = len(rows)
    ids = np.arange(n, dtype=np.int64) * 2 + 5
    keys = rng.integers(-5, 40, n).astype(np.int64) * 86_400 + 1_700_000_000  # "last_modified" days: plenty of equal keys too
    scale = orc.compute_int8_scale(rows)
    ix = pvs.VectorIndex(pvs.I8, dim)
    ix.set_scale(scale)
    ix.add_f32(rows, row_ids=ids)
    ix.set_order_keys(keys)
    corpus = orc.quantize_int8(rows, scale)
    qs = base[[3, 77, 150, 299, 8]] + 0.02 * orc.synth_rows(556, 0, 5, dim)
    hq = orc.quantize_int8(qs, scale)
    for metric, om in ((pvs.L2, orc.L2), (pvs.COSINE, orc.COSINE)):
        want = {}
        for k in (1, 7, 40, 41, 130):
            for qi in range(len(qs)):
                d = orc.score_all(orc.I8, om, corpus, hq[qi])
                want[(k, qi)] = orc.topk_ordered(d, k, ids, keys)
        for path in (0, 1):  # automatic (filter scan) / dense
            ix.set_path(path)
            for k in (1, 7, 40, 41, 130):
                gi, gd, gc = ix.search(qs, k, metric)                 # batch: pass C, or the batched dense select
                g1 = [ix.search(qs[qi], k, metric) for qi in range(2)]  # single queries: pass C, or the per-query dense sort
                for qi in range(len(qs)):
                    ei, ed = want[(k, qi)]
                    assert gc[qi] == k and np.array_equal(gi[qi, :k], ei), (metric, path, k, qi)
                    assert np.array_equal(gd[qi, :k].view(np.uint32), ed.view(np.uint32))
                for qi in range(2):
                    assert np.array_equal(g1[qi][0][0, :k], want[(k, qi)][0]), (metric, path, k, qi, "single")
        ix.set_path(0)
    # the keys really matter here (the id order alone gives another page), and removing them restores the id order
    d = orc.score_all(orc.I8, orc.L2, corpus, hq[0])
    plain_i, _ = orc.topk(d, 40, ids=ids)
    assert not np.array_equal(plain_i, want[(40, 0)][0]), "the test corpus must tie at the top"
    ix.set_order_keys(None)
    gi, gd, gc = ix.search(qs[0], 40, pvs.L2)
    assert np.array_equal(gi[0, :40], plain_i)
    # appended rows drop the keys until they are set again for all rows
    ix.set_order_keys(keys)
    ix.add_f32(rows[:3], row_ids=np.array([10**9, 10**9 + 1, 10**9 + 2], np.int64))
    gi, gd, gc = ix.search(qs[0], 5, pvs.L2)
    d2 = orc.score_all(orc.I8, orc.L2, np.concatenate([corpus, corpus[:3]]), hq[0])
    ei, _ = orc.topk(d2, 5, ids=np.concatenate([ids, [10**9, 10**9 + 1, 10**9 + 2]]))
    assert np.array_equal(gi[0, :5], ei)
    with pytest.raises(Exception):
        ix.set_order_keys(keys)  # one key per stored row
    ix.close()


def test_second_sort_key_orders_tied_items_like_the_reference(pvs):
    """The per-item page under the same final ordering: files whose aggregate ties come out newest first (the file's
    last_modified = the key its rows carry), then by file id — through the page-first MIN path and the dense aggregate + group
    rank, for MIN / MAX / AVG, with a candidate mask, and on a page cut inside a run of ties."""
    rng = np.random.default_rng(7)
    dim, distinct, files = 64, 40, 900
    base = unit_rows(321, distinct, dim)
    per_file = rng.integers(1, 4, files)
    grp = np.repeat(np.arange(files, dtype=np.int64) * 7 + 3, per_file)
    n = len(grp)
    # every row of a file is a copy of ONE base vector: aggregates of different files tie exactly
    file_vec = rng.integers(0, distinct, files)
    rows = base[np.repeat(file_vec, per_file)]
    fkey = rng.integers(0, 12, files).astype(np.int64) + 1_700_000_000
    keys = np.repeat(fkey, per_file)
    scale = orc.compute_int8_scale(rows)
    ix = pvs.VectorIndex(pvs.I8, dim)
    ix.set_scale(scale)
    ix.add_f32(rows, group_ids=grp)
    ix.set_order_keys(keys)
    corpus = orc.quantize_int8(rows, scale)
    qs = base[[1, 20, 39]] + 0.02 * orc.synth_rows(322, 0, 3, dim)
    hq = orc.quantize_int8(qs, scale)
    mask = (rng.random(n) < 0.6).astype(np.uint8)
    allowed = np.nonzero(mask)[0]
    differs = 0
    for path in (0, 1):
        ix.set_path(path)
        for metric, om in ((pvs.L2, orc.L2), (pvs.COSINE, orc.COSINE)):
            for agg, oagg in ((pvs.AGG_MIN, orc.AGG_MIN), (pvs.AGG_MAX, orc.AGG_MAX), (pvs.AGG_AVG, orc.AGG_AVG)):
                for k in (1, 13, 60):
                    og, ov, oc = ix.search_groups(qs, k, metric, agg)
                    for j in range(len(qs)):
                        eg, ev = orc.search_groups(orc.I8, om, corpus, hq[j], grp, oagg, k, order_keys=keys)
                        assert oc[j] == len(eg) and np.array_equal(og[j, : oc[j]], eg), (path, metric, agg, k, j)
                        assert np.array_equal(ov[j, : oc[j]].view(np.uint64), ev.view(np.uint64))
                        pg, _ = orc.search_groups(orc.I8, om, corpus, hq[j], grp, oagg, k)
                        differs += int(not np.array_equal(pg, eg))
            og, ov, oc = ix.search_groups_filtered(qs, 25, mask, metric, pvs.AGG_MIN)
            for j in range(len(qs)):
                eg, ev = orc.search_groups(orc.I8, om, corpus[allowed], hq[j], grp[allowed], orc.AGG_MIN, 25, order_keys=keys[allowed])
                assert oc[j] == len(eg) and np.array_equal(og[j, : oc[j]], eg), (path, metric, "masked", j)
    assert differs > 20, "the corpus must tie between files, or the test shows nothing"
    ix.set_path(0)
    # the OR arm: fused scores tie whenever two files swap places between the branches (and massively on this corpus); with the
    # keys the ties come out newest first — through the small-input full ranking and, on a larger corpus, the bounded fusion
    rows2 = base[np.repeat(rng.integers(0, distinct, files), per_file)]
    ix2 = pvs.VectorIndex(pvs.I8, dim)
    ix2.set_scale(scale)
    ix2.add_f32(rows2, group_ids=grp)
    corpus2 = orc.quantize_int8(rows2, scale)
    for k in (1, 10, 200):
        br = [dict(index=ix, query=hq[0], metric=pvs.L2, agg=pvs.AGG_MIN, rrf_k=1, weight=1.0),
              dict(index=ix2, query=hq[1], metric=pvs.L2, agg=pvs.AGG_MIN, rrf_k=1, weight=1.0)]
        fg, fs = pvs.rrf_search(br, k)
        ora = [dict(dtype=orc.I8, metric=orc.L2, corpus=corpus, query=hq[0], groups=grp, agg=orc.AGG_MIN, rrf_k=1, weight=1.0, order_keys=keys),
               dict(dtype=orc.I8, metric=orc.L2, corpus=corpus2, query=hq[1], groups=grp, agg=orc.AGG_MIN, rrf_k=1, weight=1.0)]
        eg, es = orc.rrf_search(ora, k)
        assert np.array_equal(fg, eg) and np.array_equal(fs.view(np.uint64), es.view(np.uint64)), k
    ix2.close()
    ix.close()


def test_second_sort_key_in_the_bounded_fusion(pvs):
    """The same tie-break where pvs_rrf_search takes its bounded path (>= 65536 groups): two branches over the same files whose
    rankings are mirror images on a stretch of files make pairs of files tie on the fused score exactly."""
    rng = np.random.default_rng(3)
    files, dim, k = 70_000, 32, 60
    rows = orc.synth_rows(77, 0, files, dim)
    grp = np.arange(files, dtype=np.int64) * 2 + 1
    keys = rng.integers(0, 5, files).astype(np.int64)
    ixa = pvs.VectorIndex(pvs.F32, dim)
    ixa.add(rows, group_ids=grp)
    ixa.set_order_keys(keys)
    ixb = pvs.VectorIndex(pvs.F32, dim)
    ixb.add(rows, group_ids=grp)
    qa = rows[5] + 0.01
    d = orc.score_all(orc.F32, orc.L2, rows, qa)
    top = np.argsort(d, kind="stable")[:40]
    # branch b ranks the same 40 files in reverse order: its query is irrelevant, the row weights decide (SUM(d*w)/SUM(w) = d)
    # -> use a second corpus whose distances to qb mirror the first: simply permute the rows of those files
    rows_b = rows.copy()
    rows_b[top] = rows[top[::-1]]
    ixb.close()
    ixb = pvs.VectorIndex(pvs.F32, dim)
    ixb.add(rows_b, group_ids=grp)
    br = [dict(index=ixa, query=qa, metric=pvs.L2, agg=pvs.AGG_MIN, rrf_k=1, weight=1.0),
          dict(index=ixb, query=qa, metric=pvs.L2, agg=pvs.AGG_MIN, rrf_k=1, weight=1.0)]
    fg, fs = pvs.rrf_search(br, k)
    assert pvs.lib().pvs_rrf_last_path() == 1
    ora = [dict(dtype=orc.F32, metric=orc.L2, corpus=rows, query=qa, groups=grp, agg=orc.AGG_MIN, rrf_k=1, weight=1.0, order_keys=keys),
           dict(dtype=orc.F32, metric=orc.L2, corpus=rows_b, query=qa, groups=grp, agg=orc.AGG_MIN, rrf_k=1, weight=1.0)]
    eg, es = orc.rrf_search(ora, k)
    assert np.array_equal(fg, eg) and np.array_equal(fs.view(np.uint64), es.view(np.uint64))
    pg, _ = orc.rrf_search([{**o, "order_keys": None} for o in ora], k)
    assert not np.array_equal(pg, eg), "the fused scores must tie"
    ixa.close()
    ixb.close()


@pytest.mark.parametrize("dtype", ["i8", "f16"])
def test_threshold_below_the_kth_sample_value_is_certified_or_handed_back(pvs, dtype):
    """Batches over >= 2^20 rows take pass B's threshold from the j-th (j = k/4) value of a sample a quarter the size.  That
    threshold does not guarantee k rows below it; pass C certifies it (k-th smallest upper bound among the candidates <= T) or
    hands the query to the dense path.  Adversarial corpus: the 25 best rows of query 0 all sit in sampled tiles, one per row
    group, so its threshold lies far below the corpus' 100-th value: the filter scan sees 25 candidates (int8) or,
    with 100 more rows inside the error band of the threshold (f16), 125 candidates whose 100-th upper bound exceeds T.  Either
    way the page must be the oracle's; the other queries of the batch stay on the filter path."""
    dt, odt = (pvs.I8, orc.I8) if dtype == "i8" else (pvs.F16, orc.F16)
    rng = np.random.default_rng(2024)
    n, dim, k = (1 << 20) + 4096, 64, 100
    rows = unit_rows(900, n, dim)
    qs = unit_rows(901, 8, dim)
    v1 = qs[0] / np.linalg.norm(qs[0])
    planted = 4096 * np.arange(25)                              # first rows of 25 sampled tiles (the sample takes every 32nd workgroup tile
    rows[planted] = v1 * 3.0                                    # of 32, 64 or 128 rows): 25 copies of the query's direction in 25 row groups
    ortho = rng.standard_normal(dim).astype(np.float32)
    ortho -= ortho.dot(v1) * v1
    ortho /= np.linalg.norm(ortho)
    band = rng.choice(np.setdiff1d(np.arange(5000, n), planted), 100, replace=False)  # 100 rows a small angle away: inside the f16 error band of T
    rows[band] = (v1 + 0.02 * ortho) * 3.0
    far = rng.choice(np.setdiff1d(np.arange(5000, n), np.concatenate([band, planted])), 300, replace=False)
    rows[far] = (v1 + 0.5 * ortho) * 3.0                        # the rest of the page comes from here
    scale = orc.compute_int8_scale(rows)
    ix = make_index(pvs, dt, rows, scale)
    hc = host_corpus(odt, rows, scale)
    hq = orc.quantize_int8(qs, scale) if dt == pvs.I8 else qs
    before = ix.stats()
    got = ix.search(qs, k, pvs.COSINE)
    after = ix.stats()
    exp = orc.search(odt, orc.COSINE, hc, hq, k)
    assert_same_page(got, exp)
    handed_back = after.dense_queries - before.dense_queries
    assert 1 <= handed_back <= 2, handed_back                  # query 0 (and nothing like the whole batch)
    assert after.fast_queries - before.fast_queries >= 6
    ix.close()


def test_per_item_page_from_many_groups_without_sorting_them_all(pvs):
    """MAX / AVG / weighted per-item search over >= 65,536 files ranks through a page of the files at or below a sampled
    threshold (aggregate_and_rank: rank_groups_page_first) instead of sorting every file per query: the page must be the
    oracle's — values tying across files (int8 L2), NULL aggregates, a candidate mask, the second sort key, k from 1 to 3,000."""
    rng = np.random.default_rng(17)
    files, dim = 90_000, 32
    per_file = rng.integers(1, 4, files)
    grp = np.repeat(np.arange(files, dtype=np.int64) * 3 + 1, per_file)
    n = len(grp)
    base = unit_rows(411, 5000, dim)
    rows = base[rng.integers(0, 5000, n)]                     # few distinct vectors: aggregates tie across files
    rows[np.nonzero(grp == grp[n // 2])[0]] = 0.0               # a file of zero vectors: NULL cosine aggregate
    scale = orc.compute_int8_scale(rows)
    codes = orc.quantize_int8(rows, scale)
    keys = np.repeat(rng.integers(0, 7, files).astype(np.int64), per_file)
    w = (rng.random(n) + 0.2).astype(np.float32)
    mask = (rng.random(n) < 0.4).astype(np.uint8)
    allowed = np.nonzero(mask)[0]
    ix = pvs.VectorIndex(pvs.I8, dim)
    ix.set_scale(scale)
    ix.add_f32(rows, group_ids=grp)
    qs = base[np.arange(7, 7 + 24)] + 0.01 * orc.synth_rows(412, 0, 24, dim)   # 24 columns x 90k files: the page-first ranking
    hq = orc.quantize_int8(qs, scale)
    for with_keys in (False, True):
        ix.set_order_keys(keys if with_keys else None)
        ok = keys if with_keys else None
        for k in (1, 50, 3000):
            for metric, om in ((pvs.L2, orc.L2), (pvs.COSINE, orc.COSINE)):
                for agg, oagg, ww in ((pvs.AGG_MAX, orc.AGG_MAX, None), (pvs.AGG_AVG, orc.AGG_AVG, None), (pvs.AGG_AVG, orc.AGG_AVG, w)):
                    og, ov, oc = ix.search_groups(qs, k, metric, agg, row_weights=ww)
                    for j in (0, 11, 23):
                        eg, ev = orc.search_groups(orc.I8, om, codes, hq[j], grp, oagg, k, weights=ww, order_keys=ok)
                        assert oc[j] == len(eg) and np.array_equal(og[j, : oc[j]], eg), (with_keys, k, metric, agg, ww is not None, j)
                        a, e = ov[j, : oc[j]], ev
                        assert np.array_equal(np.isnan(a), np.isnan(e)) and np.array_equal(a[~np.isnan(a)].view(np.uint64), e[~np.isnan(e)].view(np.uint64))
            og, ov, oc = ix.search_groups_filtered(qs, k, mask, pvs.L2, pvs.AGG_MAX)
            for j in (0, 23):
                eg, ev = orc.search_groups(orc.I8, orc.L2, codes[allowed], hq[j], grp[allowed], orc.AGG_MAX, k, order_keys=None if ok is None else ok[allowed])
                assert oc[j] == len(eg) and np.array_equal(og[j, : oc[j]], eg), (with_keys, k, "masked", j)
    ix.close()


def test_direct_int8_scorer_hands_sums_beyond_2_24_to_the_in_order_scorer(pvs):
    """k_score_i8_direct (1..4 queries: pvs_score_all, small pvs_score_batch, per-branch RRF scoring) finishes with the closed form
    of the exact integer sums, valid while they stay below 2^24.  Saturated codes at dim 1024 push an L2 sum to 4 x 1024 x 127^2:
    the kernel raises its flag and the in-order scorer answers — the column must be the oracle's either way, bit for bit."""
    dim, n = 1024, 4096
    rng = np.random.default_rng(5)
    codes = rng.integers(-30, 31, (n, dim)).astype(np.int8)
    codes[7] = 127
    codes[8] = -128
    codes[9, ::2] = 127
    ix = pvs.VectorIndex(pvs.I8, dim)
    ix.set_scale(0.01)
    ix.add(codes)
    qs = np.stack([np.full(dim, -127, np.int8), np.full(dim, 127, np.int8), rng.integers(-20, 21, dim).astype(np.int8)])
    for metric, om in ((pvs.L2, orc.L2), (pvs.COSINE, orc.COSINE)):
        exp = [orc.score_all(orc.I8, om, codes, qs[j]) for j in range(3)]
        for j in range(3):
            got = ix.score_all(qs[j], metric)
            assert np.array_equal(got.view(np.uint32), exp[j].view(np.uint32)), (metric, j, "score_all")
        m = ix.score_batch(qs, metric)
        for j in range(3):
            assert np.array_equal(m[:, j].view(np.uint32), exp[j].view(np.uint32)), (metric, j, "score_batch")
    assert float(orc.score_all(orc.I8, orc.L2, codes, qs[0])[7]) ** 2 > 2 ** 24, "the corpus must leave the exact range"
    ix.close()

