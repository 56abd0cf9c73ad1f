"""Several shards behind one search, on real multi-shard data.

* one process, several GPUs: a multi-device pvs_index (pvs_index_desc.n_devices).  On a one-GPU machine the device
  list repeats ordinal 0, which still exercises everything that can go wrong in the sharded path — the row split,
  per-shard ids, peer copies into the root's [shards][batch][k] gather buffers, k_merge's indexing, the dense-path
  redo — against the oracle over the WHOLE corpus (not against pvs_search on the same index);
* one process per GPU (RCCL): two ranks as two host threads when at least two devices are visible (skipped, not
  passed, otherwise: RCCL refuses two ranks on one GPU);
* concurrency regressions: searches in flight never share sort scratch; the in-flight limit is an error, not a hang.
"""
import ctypes as C
import os
import threading

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


def _layouts(pvs):
    n = pvs.device_count()
    out = [[0, 0], [0, 0, 0]]
    if n >= 2:
        out.append(list(range(min(n, 8))))
    return out


def _host_corpus(dtype, rows, scale):
    if dtype == "i8":
        return orc.quantize_int8(rows, scale)
    if dtype == "f16":
        return rows.astype(np.float16)
    return rows


@pytest.mark.parametrize("dtype", ["i8", "f16", "f32"])
@pytest.mark.parametrize("metric", ["cosine", "l2"])
def test_multi_device_index_equals_oracle_over_whole_corpus(pvs, dtype, metric):
    from panoptikon_amd import _lib as L

    pdt = {"i8": pvs.I8, "f16": pvs.F16, "f32": pvs.F32}[dtype]
    odt = {"i8": orc.I8, "f16": orc.F16, "f32": orc.F32}[dtype]
    pm, om = (pvs.COSINE, orc.COSINE) if metric == "cosine" else (pvs.L2, orc.L2)
    dim, k = 768, 40
    rows = orc.synth_rows(77, 0, 21_003, dim)
    rows[5000] = rows[100]  # a duplicate vector that lands in another shard: tie broken by id across shards
    scale = orc.compute_int8_scale(rows) if dtype == "i8" else None
    corpus = _host_corpus(dtype, rows, scale)
    queries = orc.synth_rows(0x5EED0000, 0, 37, dim)
    queries[0] = rows[100]
    hq = orc.quantize_int8(queries, scale) if dtype == "i8" else queries
    ids = np.cumsum(np.random.default_rng(5).integers(1, 4, len(rows))).astype(np.int64) + 1000  # increasing, with gaps
    exp = orc.search(odt, om, corpus, hq, k, ids=ids)
    for devices in _layouts(pvs):
        ix = pvs.VectorIndex(pdt, dim, devices=devices)
        if scale is not None:
            ix.set_scale(scale)
        # ragged add calls, one smaller than the number of shards
        cuts = [0, 1, 7000, 7001, 7003, 15_000, len(rows)]
        for a, b in zip(cuts[:-1], cuts[1:]):
            ix.add_f32(rows[a:b], row_ids=ids[a:b])
        st = ix.stats()
        assert st.rows == len(rows) and st.dim == dim
        assert np.array_equal(ix.read_ids(), ids), "global row order = order of the add calls"
        assert np.array_equal(ix.read_rows(6990, 40).view(np.uint8), np.ascontiguousarray(corpus[6990:7030]).view(np.uint8))
        gi, gd, gc = ix.search(queries, k, pm)
        assert gc.tolist() == [k] * len(queries)
        assert np.array_equal(gi, exp[0]), f"devices={devices}: ids differ from the oracle over the whole corpus"
        assert np.array_equal(gd.view(np.uint32), exp[1].view(np.uint32)), f"devices={devices}: distances not bit-exact"
        assert gi[0, 0] == ids[100] and gi[0, 1] == ids[5000], "equal distances across shards: smaller id first"
        # the `d` column in global row order
        d_all = ix.score_all(queries[3], pm)
        e_all = orc.score_all(odt, om, corpus, hq[3])
        assert np.array_equal(d_all.view(np.uint32), e_all.view(np.uint32))
        # stream-ordered form, several searches in flight, buffers on devices[0]
        dq = pvs.DeviceBuffer.from_numpy(queries, devices[0])
        outs = [(pvs.DeviceBuffer(37 * k * 8, devices[0]), pvs.DeviceBuffer(37 * k * 4, devices[0]), pvs.DeviceBuffer(37 * 4, devices[0]))
                for _ in range(3)]
        for streams in (1, 2):
            ix.set_streams(streams)
            tickets = [ix.search_device(dq, L.F32, 37, k, pm, *o) for o in outs]
            for t, o in zip(tickets, outs):
                ix.wait(t)
                assert np.array_equal(o[0].to_numpy(np.int64, (37, k)), exp[0])
                assert np.array_equal(o[1].to_numpy(np.float32, (37, k)).view(np.uint32), exp[1].view(np.uint32))
        ix.set_streams(1)
        if metric == "cosine":
            # a zero query: every cosine distance is NULL, every shard hands it to its dense path, the root merges again
            qz = queries.copy()
            qz[2] = 0.0
            zi, zd, zc = ix.search(qz, k, pm)
            assert np.isnan(zd[2]).all() and zi[2].tolist() == ids[:k].tolist(), "NULL distances: id order over the whole corpus"
            assert np.array_equal(zi[3:], exp[0][3:]) and np.array_equal(zd[3:].view(np.uint32), exp[1][3:].view(np.uint32))
        # forced dense path on every shard
        ix.set_path(1)
        gi2, gd2, _ = ix.search(queries[:5], k, pm)
        assert np.array_equal(gi2, exp[0][:5]) and np.array_equal(gd2.view(np.uint32), exp[1][:5].view(np.uint32))
        ix.close()


def test_multi_device_short_pages_and_implicit_ids(pvs):
    """Fewer rows than k, fewer rows than shards, implicit ids (id_base + global row index)."""
    dim = 64
    rows = orc.synth_rows(3, 0, 11, dim)
    q = orc.synth_rows(4, 0, 3, dim)
    ix = pvs.VectorIndex(pvs.F32, dim, devices=[0, 0, 0, 0], id_base=500)
    ix.add(rows[:2])   # two rows over four shards
    ix.add(rows[2:])
    gi, gd, gc = ix.search(q, 20, pvs.L2)
    ei, ed = orc.search(orc.F32, orc.L2, rows, q, 20, ids=np.arange(500, 511, dtype=np.int64))
    assert gc.tolist() == [11] * 3 and np.array_equal(gi[:, :11], ei) and np.array_equal(gd[:, :11].view(np.uint32), ed.view(np.uint32))
    assert (gi[:, 11:] == -1).all() and np.isnan(gd[:, 11:]).all()
    assert np.array_equal(ix.read_ids(), np.arange(500, 511))
    with pytest.raises(pvs.PvsError):
        ix.add(rows[:1], row_ids=np.array([505], np.int64))  # not increasing over the whole index
    ix.close()


def test_multi_device_min_groups_spanning_shards(pvs):
    """pvs_search_groups(MIN) on a multi-device index: groups span shards; the global MIN is the min of shard minima."""
    rng = np.random.default_rng(11)
    dim, n, k = 256, 9000, 25
    rows = orc.synth_rows(19, 0, n, dim)
    grp = rng.integers(0, 1500, n).astype(np.int64) * 5 + 7
    scale = orc.compute_int8_scale(rows)
    codes = orc.quantize_int8(rows, scale)
    q = orc.synth_rows(23, 0, 6, dim)
    hq = orc.quantize_int8(q, scale)
    for devices in _layouts(pvs):
        ix = pvs.VectorIndex(pvs.I8, dim, devices=devices)
        ix.set_scale(scale)
        ix.add_f32(rows[:4000], group_ids=grp[:4000])
        ix.add_f32(rows[4000:], group_ids=grp[4000:])
        og, ov, oc = ix.search_groups(hq, k, pvs.COSINE, pvs.AGG_MIN)
        for j in range(len(hq)):
            eg, ev = orc.search_groups(orc.I8, orc.COSINE, codes, hq[j], grp, orc.AGG_MIN, k)
            assert oc[j] == len(eg) and np.array_equal(og[j, : oc[j]], eg), f"devices={devices} query {j}"
            assert np.array_equal(ov[j, : oc[j]].view(np.uint64), ev.view(np.uint64))
        ix.close()
    # rows added WITHOUT group ids are placed in contiguous pieces: the per-item operators that need a group on one device say so
    ix = pvs.VectorIndex(pvs.I8, dim, devices=_layouts(pvs)[0])
    ix.set_scale(scale)
    ix.add_f32(rows[:100])
    with pytest.raises(pvs.PvsError) as e:
        ix.add_f32(rows[100:200], group_ids=grp[100:200])   # group ids must come from the first add
    assert e.value.status == 5  # PVS_ERR_STATE
    ix.close()


def _same_f32(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)].view(np.uint32), b[~np.isnan(b)].view(np.uint32))


def test_multi_device_per_item_operators_placed_by_group(pvs):
    """with group ids at add time a multi-device index places every group on one shard, and MAX / AVG /
    weighted per-item search, candidate masks, pvs_score_batch, similar_to(_ex) and pvs_rrf_search answer like one device —
    each compared with the oracle over the WHOLE corpus, bit for bit."""
    rng = np.random.default_rng(77)
    dim, n, k = 128, 12_000, 40
    rows = orc.synth_rows(91, 0, n, dim)
    grp = np.sort(rng.integers(0, 2500, n)).astype(np.int64) * 3 + 1     # ~5 rows per file, files adjacent as in the reference
    rows[np.nonzero(grp == grp[n // 3])[0]] = 0.0                          # one file with NULL cosine distances only
    scale = orc.compute_int8_scale(rows)
    codes = orc.quantize_int8(rows, scale)
    q = orc.synth_rows(93, 0, 5, dim)
    hq = orc.quantize_int8(q, scale)
    w = (rng.random(n) + 0.1).astype(np.float32)
    mask = (rng.random(n) < 0.3).astype(np.uint8)
    allowed = np.nonzero(mask)[0]
    ids = np.arange(n, dtype=np.int64) * 2 + 10
    conf = rng.random(n)
    conf[rng.random(n) < 0.2] = np.nan
    lang = rng.random(n)
    kind = (rng.random(n) < 0.4).astype(np.uint8)
    for devices in _layouts(pvs):
        ix = pvs.VectorIndex(pvs.I8, dim, devices=devices)
        ix.set_scale(scale)
        for a in range(0, n, 5000):   # several adds; a file may straddle two calls
            ix.add_f32(rows[a:a + 5000], row_ids=ids[a:a + 5000], group_ids=grp[a:a + 5000])
        tag = f"devices={devices}"
        # the global row order survives the placement
        ri, rg = ix.read_ids(0, n, groups=True)
        assert np.array_equal(ri, ids) and np.array_equal(rg, grp), tag
        assert np.array_equal(ix.read_rows(4990, 30), codes[4990:5020]), tag
        # plain search still sees every row
        gi, gd, gc = ix.search(hq, k, pvs.COSINE)
        ei, ed = orc.search(orc.I8, orc.COSINE, codes, hq, k, ids=ids)
        assert (gc == k).all() and np.array_equal(gi, ei) and _same_f32(gd, ed), tag
        # per-item search: every aggregate, weights, masks
        for agg, oagg in ((pvs.AGG_MIN, orc.AGG_MIN), (pvs.AGG_MAX, orc.AGG_MAX), (pvs.AGG_AVG, orc.AGG_AVG)):
            og, ov, oc = ix.search_groups(hq, k, pvs.COSINE, agg)
            for j in range(len(hq)):
                eg, ev = orc.search_groups(orc.I8, orc.COSINE, codes, hq[j], grp, oagg, k)
                assert oc[j] == len(eg) and np.array_equal(og[j, : oc[j]], eg), f"{tag} agg {agg} query {j}"
                assert np.array_equal(ov[j, : oc[j]].view(np.uint64), ev.view(np.uint64)), f"{tag} agg {agg} query {j}"
        og, ov, oc = ix.search_groups(hq, k, pvs.L2, pvs.AGG_AVG, row_weights=w)
        for j in range(len(hq)):
            eg, ev = orc.search_groups(orc.I8, orc.L2, codes, hq[j], grp, orc.AGG_AVG, k, weights=w)
            assert oc[j] == len(eg) and np.array_equal(og[j, : oc[j]], eg) and np.array_equal(ov[j, : oc[j]].view(np.uint64), ev.view(np.uint64)), tag
        og, ov, oc = ix.search_groups_filtered(hq, k, mask, pvs.COSINE, pvs.AGG_MAX)
        for j in range(len(hq)):
            eg, ev = orc.search_groups(orc.I8, orc.COSINE, codes[allowed], hq[j], grp[allowed], orc.AGG_MAX, k)
            assert oc[j] == len(eg) and np.array_equal(og[j, : oc[j]], eg) and np.array_equal(ov[j, : oc[j]].view(np.uint64), ev.view(np.uint64)), tag
        fi, fd, fc = ix.search_filtered(hq, k, mask, pvs.COSINE)
        ei, ed = orc.search(orc.I8, orc.COSINE, codes[allowed], hq, k, ids=ids[allowed])
        assert (fc == k).all() and np.array_equal(fi, ei) and _same_f32(fd, ed), tag
        # the dense matrix in global row order
        m = ix.score_batch(hq, pvs.L2)
        for j in range(len(hq)):
            assert np.array_equal(m[:, j].view(np.uint32), orc.score_all(orc.I8, orc.L2, codes, hq[j]).view(np.uint32)), tag
        # similar_to: a whole file as target (its rows live on one shard), plain and with weights + gates
        tfile = grp[n // 2]
        targets = ids[grp == tfile]
        trows = np.nonzero(grp == tfile)[0].tolist()
        sg, sv = ix.similar_to(targets, k, pvs.L2, pvs.AGG_AVG)
        eg, ev = orc.similar_to(orc.I8, orc.L2, codes, trows, grp, orc.AGG_AVG, k)
        assert np.array_equal(sg, eg) and np.array_equal(sv.view(np.uint64), ev.view(np.uint64)), tag
        sg, sv = ix.similar_to_ex(targets, k, pvs.COSINE, pvs.AGG_MIN, confidence=conf, language_confidence=lang, confidence_weight=1.5,
                                  language_confidence_weight=0.5, row_kind=kind, xmodal_t2t=False)
        eg, ev = orc.similar_to_ex(orc.I8, orc.COSINE, codes, trows, grp, orc.AGG_MIN, k, conf=conf, lang=lang, cw=1.5, lw=0.5, kind=kind,
                                   xmodal_t2t=False)
        assert np.array_equal(sg, eg), tag
        assert np.allclose(sv, ev, rtol=1e-12, atol=0, equal_nan=True), tag   # device pow(): within 1 ulp of libm (pvs.h)
        # OR arm: two multi-device branches over the same devices
        rows2 = orc.synth_rows(95, 0, 7000, 64)
        grp2 = np.sort(rng.integers(0, 2500, 7000)).astype(np.int64) * 3 + 1
        ix2 = pvs.VectorIndex(pvs.F16, 64, devices=devices)
        ix2.add_f32(rows2, group_ids=grp2)
        q2 = orc.synth_rows(97, 0, 1, 64)[0]
        br = [dict(index=ix, query=hq[0], metric=pvs.COSINE, agg=pvs.AGG_MIN, rrf_k=5, weight=1.0),
              dict(index=ix2, query=q2, metric=pvs.L2, agg=pvs.AGG_AVG, rrf_k=10, weight=0.7, row_weights=None)]
        fg, fs = pvs.rrf_search(br, 100)
        ora = [dict(dtype=orc.I8, metric=orc.COSINE, corpus=codes, query=hq[0], groups=grp, agg=orc.AGG_MIN, rrf_k=5, weight=1.0),
               dict(dtype=orc.F16, metric=orc.L2, corpus=rows2.astype(np.float16), query=q2, groups=grp2, agg=orc.AGG_AVG, rrf_k=10, weight=0.7)]
        eg, es = orc.rrf_search(ora, 100)
        assert np.array_equal(fg, eg) and np.array_equal(fs.view(np.uint64), es.view(np.uint64)), tag
        ix2.close()
        ix.close()


def test_rccl_two_ranks_as_two_threads(pvs):
    """One process per GPU in miniature: two ranks (two host threads, one device each) over a real RCCL communicator.
    Needs two devices — RCCL refuses two ranks on one GPU."""
    from panoptikon_amd import _lib as L

    if pvs.device_count() < 2:
        pytest.skip("needs >= 2 visible devices (RCCL: 'Duplicate GPU detected' with two ranks on one)")
    world, dim, k, b = 2, 768, 50, 64
    n = 40_000
    rows = orc.synth_rows(41, 0, n, dim)
    scale = orc.compute_int8_scale(rows)
    codes = orc.quantize_int8(rows, scale)
    q = orc.synth_rows(43, 0, b, dim)
    exp = orc.search(orc.I8, orc.COSINE, codes, orc.quantize_int8(q, scale), k)
    uid = (C.c_uint8 * L.UNIQUE_ID_BYTES)()
    L.check(pvs.lib().pvs_comm_unique_id(uid))
    res, errs, fused = [None] * world, [], [None] * world

    def rank_main(r):
        try:
            r0, r1 = pvs.shard_range(n, world, r)
            comm = C.c_void_p()
            L.check(pvs.lib().pvs_comm_create(uid, world, r, r, C.byref(comm)))
            ix = pvs.VectorIndex(pvs.I8, dim, device=r, id_base=r0)
            ix.set_scale(scale)
            ix.add_f32(rows[r0:r1])
            dq = pvs.DeviceBuffer.from_numpy(q, r)
            oi, od, oc = pvs.DeviceBuffer(b * k * 8, r), pvs.DeviceBuffer(b * k * 4, r), pvs.DeviceBuffer(b * 4, r)
            for _ in range(3):
                L.check(pvs.lib().pvs_search_sharded(ix._h, comm, dq.ptr, L.F32, b, k, pvs.COSINE, oi.ptr, od.ptr, oc.ptr))
            res[r] = (oi.to_numpy(np.int64, (b, k)), od.to_numpy(np.float32, (b, k)), oc.to_numpy(np.uint32, (b,)))
            # ncclAllReduce(max): the space's absmax from the shards'
            v = C.c_float(float(np.abs(rows[r0:r1]).max()))
            L.check(pvs.lib().pvs_comm_allreduce_max_f32(comm, C.byref(v)))
            assert np.float32(v.value) == np.float32(np.abs(rows).max())
            # the sharded OR/RRF round loop over the same communicator: rows sharded by group (3 rows per file), one branch
            r0g = r0 - r0 % 3 if r == 0 else r0 + (-r0) % 3  # cut on a group boundary
            r1g = r1 + (-r1) % 3 if r == 0 else r1
            fx = pvs.VectorIndex(pvs.I8, dim, device=r, id_base=r0g)
            fx.set_scale(scale)
            fx.add_f32(rows[r0g:r1g], group_ids=np.arange(r0g, r1g, dtype=np.int64) // 3)
            hq = orc.quantize_int8(q[:1], scale)[0]
            fused[r] = pvs.rrf_search_sharded([dict(index=fx, query=hq, metric=pvs.COSINE, agg=pvs.AGG_MIN, rrf_k=5, weight=1.0)], 20, comm=comm)
            fx.close()
            pvs.lib().pvs_comm_destroy(comm)
            ix.close()
        except Exception as e:  # noqa: BLE001
            errs.append((r, repr(e)))

    th = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(world)]  # (daemon: a rank stuck in a collective fails the test, it does not hang the run)
    [t.start() for t in th]
    [t.join(180) for t in th]
    assert not any(t.is_alive() for t in th), "a rank did not come back from the RCCL exchange within 180 s"
    assert not errs, errs
    for r in range(world):
        assert np.array_equal(res[r][0], exp[0]) and np.array_equal(res[r][1].view(np.uint32), exp[1].view(np.uint32)), f"rank {r}"
    eg, es = orc.rrf_search([dict(dtype=orc.I8, metric=orc.COSINE, corpus=codes, query=orc.quantize_int8(q[:1], scale)[0],
                                  groups=np.arange(n, dtype=np.int64) // 3, agg=orc.AGG_MIN, rrf_k=5, weight=1.0)], 20)
    for r in range(world):
        assert np.array_equal(fused[r][0], eg) and np.array_equal(fused[r][1].view(np.uint64), es.view(np.uint64)), f"rank {r}: sharded fusion over RCCL"


def test_rccl_one_rank_collectives_and_the_sharded_fusion_entry_point(pvs):
    """What a one-GPU box can run of the RCCL-backed entry points: a 1-rank communicator through pvs_comm_allreduce_max_f32 and
    pvs_rrf_search_sharded(comm) — the page must be pvs_rrf_search's, which the other tests pin against the oracle."""
    from panoptikon_amd import _lib as L

    dim, n = 96, 30_000
    rows = orc.synth_rows(71, 0, n, dim)
    grp = np.arange(n, dtype=np.int64) // 3
    scale = orc.compute_int8_scale(rows)
    uid = (C.c_uint8 * L.UNIQUE_ID_BYTES)()
    L.check(pvs.lib().pvs_comm_unique_id(uid))
    comm = C.c_void_p()
    L.check(pvs.lib().pvs_comm_create(uid, 1, 0, 0, C.byref(comm)))
    try:
        v = C.c_float(1.25)
        L.check(pvs.lib().pvs_comm_allreduce_max_f32(comm, C.byref(v)))
        assert v.value == 1.25
        ix = pvs.VectorIndex(pvs.I8, dim)
        ix.set_scale(scale)
        ix.add_f32(rows, group_ids=grp)
        ix2 = pvs.VectorIndex(pvs.F16, dim)
        ix2.add_f32(rows[::-1].copy(), group_ids=grp)
        q = orc.synth_rows(72, 0, 1, dim)[0]
        brs = [dict(index=ix, query=orc.quantize_int8(q[None, :], scale)[0], metric=pvs.COSINE, agg=pvs.AGG_MIN, rrf_k=5, weight=1.0),
               dict(index=ix2, query=q, metric=pvs.L2, agg=pvs.AGG_AVG, rrf_k=10, weight=0.7)]
        g1, s1 = pvs.rrf_search(brs, 50)
        g2, s2 = pvs.rrf_search_sharded(brs, 50, comm=comm)
        g3, s3 = pvs.rrf_search_sharded(brs, 50)  # world = 1 without a communicator
        assert np.array_equal(g1, g2) and np.array_equal(s1.view(np.uint64), s2.view(np.uint64))
        assert np.array_equal(g1, g3) and np.array_equal(s1.view(np.uint64), s3.view(np.uint64))
        with pytest.raises(pvs.PvsError):
            pvs.rrf_search_sharded([dict(brs[0], weight=-1.0)], 5, comm=comm)  # the bound needs non-negative weights
        ix.close()
        ix2.close()
    finally:
        pvs.lib().pvs_comm_destroy(comm)


def test_in_flight_limit_is_an_error_not_a_hang(pvs):
    from panoptikon_amd import _lib as L

    rows = orc.synth_rows(7, 0, 4096, 128)
    ix = pvs.VectorIndex(pvs.F16, 128)
    ix.add_f32(rows)
    q = pvs.DeviceBuffer.from_numpy(orc.synth_rows(8, 0, 4, 128))
    outs = [(pvs.DeviceBuffer(4 * 5 * 8), pvs.DeviceBuffer(4 * 5 * 4), pvs.DeviceBuffer(4 * 4)) for _ in range(17)]
    tickets = [ix.search_device(q, L.F32, 4, 5, pvs.COSINE, *outs[i]) for i in range(16)]
    with pytest.raises(pvs.PvsError) as e:
        ix.search_device(q, L.F32, 4, 5, pvs.COSINE, *outs[16])
    assert e.value.status == 5 and "in flight" in e.value.message  # PVS_ERR_STATE
    for t in tickets:
        ix.wait(t)
    t = ix.search_device(q, L.F32, 4, 5, pvs.COSINE, *outs[16])
    ix.wait(t)
    ref = outs[0][0].to_numpy(np.int64, (4, 5))
    assert all(np.array_equal(o[0].to_numpy(np.int64, (4, 5)), ref) for o in outs)
    ix.close()


def test_concurrent_per_item_searches_do_not_share_sort_scratch(pvs):
    """MAX / AVG / weighted group searches and similar_to rank through a radix sort; searches in flight on one index must
    each own that scratch (16 read connections in the reference, db/connection.rs:235)."""
    rng = np.random.default_rng(2)
    dim, n = 128, 6000
    rows = orc.synth_rows(51, 0, n, dim)
    grp = rng.integers(0, 900, n).astype(np.int64)
    ix = pvs.VectorIndex(pvs.F16, dim)
    ix.add_f32(rows, group_ids=grp)
    qs = orc.synth_rows(53, 0, 12, dim)
    w = rng.random(n).astype(np.float32) + 0.1
    targets = np.flatnonzero(grp == grp[17])[:3].astype(np.int64)

    def work(i):
        q = qs[i % len(qs)][None, :]
        kind = i % 4
        if kind == 0:
            return ix.search_groups(q, 40, pvs.COSINE, pvs.AGG_AVG)
        if kind == 1:
            return ix.search_groups(q, 40, pvs.L2, pvs.AGG_MAX)
        if kind == 2:
            return ix.search_groups(q, 40, pvs.COSINE, pvs.AGG_AVG, row_weights=w)
        return ix.similar_to(targets, 40, pvs.L2, pvs.AGG_AVG)

    ref = [work(i) for i in range(24)]
    got = [None] * 96
    errs = []

    def runner(t):
        try:
            for i in range(t, 96, 8):
                got[i] = work(i % 24)
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    th = [threading.Thread(target=runner, args=(t,)) for t in range(8)]
    [t.start() for t in th]
    [t.join(300) for t in th]
    assert not errs, errs
    for i in range(96):
        for a, b in zip(got[i], ref[i % 24]):
            a, b = np.asarray(a), np.asarray(b)
            assert a.shape == b.shape and np.array_equal(a.view(np.uint8), b.view(np.uint8)), f"call {i} differs under concurrency"
    ix.close()


def test_similar_to_leaves_out_groups_without_a_joined_pair(pvs):
    """INNER JOIN + `other.sha256 != target` + the cross-modal gates (item_similarity.rs:445-489): the target's own
    group and groups whose every pair is gated away are not in the result, even when k exceeds the number of groups."""
    dim, n = 64, 40
    rows = orc.synth_rows(61, 0, n, dim)
    grp = (np.arange(n) // 4).astype(np.int64)          # 10 groups of 4 rows
    kind = np.zeros(n, np.uint8)
    kind[grp == 3] = 1                                   # group 3: text rows only
    ix = pvs.VectorIndex(pvs.F32, dim)
    ix.add(rows, group_ids=grp)
    targets = np.array([0, 1, 2, 3], np.int64)           # every row of group 0 (all clip)
    g, v = ix.similar_to(targets, 50, pvs.L2, pvs.AGG_AVG)
    eg, ev = orc.similar_to(orc.F32, orc.L2, rows, targets.tolist(), grp, orc.AGG_AVG, 50)
    assert 0 not in g.tolist() and len(g) == 9 and np.array_equal(g, eg) and np.array_equal(v.view(np.uint64), ev.view(np.uint64))
    # i2i off: clip x clip pairs leave the join -> only the text group remains
    g2, v2 = ix.similar_to_ex(targets, 50, pvs.L2, pvs.AGG_AVG, row_kind=kind, xmodal_i2i=False)
    eg2, ev2 = orc.similar_to_ex(orc.F32, orc.L2, rows, targets.tolist(), grp, orc.AGG_AVG, 50, kind=kind, xmodal_i2i=False)
    assert g2.tolist() == [3] and np.array_equal(g2, eg2) and np.array_equal(v2.view(np.uint64), ev2.view(np.uint64))
    ix.close()


class _ThreadGather:
    """all-gather between host threads standing in for ranks (each rank calls with the same sequence of shapes)."""

    def __init__(self, world):
        self.world = world
        self.slots = [None] * world
        self.bar = threading.Barrier(world)

    def for_rank(self, r):
        def gather(a):
            self.slots[r] = np.array(a, copy=True)
            self.bar.wait()
            out = np.stack(self.slots)
            self.bar.wait()
            return out

        return gather


def test_rrf_sharded_by_group_equals_the_whole_corpus(pvs):
    """BASELINE configs[4] on several GPUs, in miniature: an image branch (int8 cosine, MIN) and a text branch (f16 L2, MIN) whose
    rows are sharded BY GROUP over three shards (here three indexes per branch, three host threads as ranks); the sharded bounded
    fusion (pvs_rrf_cols_* + panoptikon_amd.sharded.rrf_search_sharded) must return the oracle's page over the WHOLE corpus."""
    rng = np.random.default_rng(5)
    world, n_files, k = 3, 90_000, 100
    specs = [(pvs.I8, orc.I8, 64, 140_000, pvs.COSINE, orc.COSINE, 5, 1.0), (pvs.F16, orc.F16, 48, 100_000, pvs.L2, orc.L2, 10, 0.7)]
    shards = [[] for _ in range(world)]
    ora = []
    for i, (dt, odt, dim, n, m, om, rk, wt) in enumerate(specs):
        rows = orc.synth_rows(500 + i, 0, n, dim)
        pool = np.arange(0, n_files, dtype=np.int64)[rng.random(n_files) < (0.9, 0.6)[i]]
        groups = np.sort(rng.choice(pool, n)).astype(np.int64)
        if m == pvs.COSINE:
            rows[np.nonzero(groups == groups[n // 2])[0]] = 0.0  # a NULL aggregate: first in the ascending window
        scale = orc.compute_int8_scale(rows)
        q = orc.synth_rows(600 + i, 0, 1, dim)[0]
        hq = orc.quantize_int8(q[None, :], scale)[0] if dt == pvs.I8 else q
        ranges = pvs.shard_ranges_by_group(groups, world)
        for r, (a, b) in enumerate(ranges):
            ix = pvs.VectorIndex(dt, dim, id_base=a)
            if dt == pvs.I8:
                ix.set_scale(scale)
            ix.add_f32(rows[a:b], group_ids=groups[a:b])
            shards[r].append(dict(index=ix, query=hq, metric=m, agg=pvs.AGG_MIN, rrf_k=rk, weight=wt))
        corpus = orc.quantize_int8(rows, scale) if dt == pvs.I8 else rows.astype(np.float16)
        ora.append(dict(dtype=odt, metric=om, corpus=corpus, query=hq, groups=groups, agg=orc.AGG_MIN, rrf_k=rk, weight=wt))
    eg, es = orc.rrf_search(ora, k)
    tg = _ThreadGather(world)
    res, errs = [None] * world, []

    def rank_main(r):
        try:
            res[r] = pvs.rrf_search_sharded(shards[r], k, tg.for_rank(r))
        except Exception as e:  # noqa: BLE001
            errs.append((r, repr(e)))
            tg.bar.abort()

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join(600) for t in th]
    assert not errs, errs
    for r in range(world):
        assert np.array_equal(res[r][0], eg), f"rank {r}: groups differ"
        assert np.array_equal(res[r][1].view(np.uint64), es.view(np.uint64)), f"rank {r}: scores not bit-exact"
    for sh in shards:
        for b in sh:
            b["index"].close()


def test_coalesced_single_query_callers_get_their_own_pages(pvs):
    """pvs_index_set_coalescing: sixteen host threads, one query (sometimes two or three) per call, different k and both metrics
    in the mix — every caller gets exactly the page a lone call returns, and the calls were served by fewer corpus passes."""
    import threading

    dim = 256
    rows = orc.synth_rows(31, 0, 30_000, dim)
    rows[777] = rows[12]  # a duplicate: ties broken by id inside every page
    scale = orc.compute_int8_scale(rows)
    codes = orc.quantize_int8(rows, scale)
    ix = pvs.VectorIndex(pvs.I8, dim)
    ix.set_scale(scale)
    groups = (np.arange(len(rows), dtype=np.int64) // 3) * 7 + 1
    ix.add_f32(rows, group_ids=groups)
    nthreads, per = 16, 8
    queries = orc.synth_rows(0x5EED0044, 0, nthreads * per + 3, dim)
    queries[5] = rows[12]
    hq = orc.quantize_int8(queries, scale)
    exp = {(m, k): orc.search(orc.I8, om, codes, hq, k, threads=8) for m, om in ((pvs.COSINE, orc.COSINE), (pvs.L2, orc.L2)) for k in (7, 20, 33)}
    errors = []
    start = threading.Barrier(nthreads)

    def worker(t):
        try:
            start.wait()
            for rep in range(per):
                q = t * per + rep
                if t % 4 in (1, 2) and rep % 2 == 1:  # per-item searches share passes among themselves (same aggregate)
                    agg, oagg = (pvs.AGG_MIN, orc.AGG_MIN) if t % 4 == 1 else (pvs.AGG_MAX, orc.AGG_MAX)
                    kg = 5 + rep
                    gg, gv, gn = ix.search_groups(hq[q:q + 1], kg, pvs.COSINE, agg)
                    eg, ev = orc.search_groups(orc.I8, orc.COSINE, codes, hq[q], groups, oagg, kg)
                    if not (gn[0] == len(eg) and np.array_equal(gg[0, :gn[0]], eg) and np.array_equal(gv[0, :gn[0]].view(np.uint64), ev.view(np.uint64))):
                        errors.append((t, rep, q, "groups", agg))
                    continue
                nb = 1 + (q % 5 == 0) + (q % 7 == 0)
                k = (7, 20, 33)[(t + rep) % 3]
                metric = pvs.L2 if t % 4 == 3 else pvs.COSINE
                as_f32 = t % 2 == 0  # f32 queries are quantized on the device with the frozen scale; int8 codes pass through
                gi, gd, gc = ix.search(queries[q:q + nb] if as_f32 else hq[q:q + nb], k, metric)
                ei, ed = exp[(metric, k)]
                for i in range(nb):
                    if not (gc[i] == k and np.array_equal(gi[i], ei[q + i]) and np.array_equal(gd[i].view(np.uint32), ed[q + i].view(np.uint32))):
                        errors.append((t, rep, q + i, k, metric))
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    try:
        ix.set_coalescing(2000, 32)
        th = [threading.Thread(target=worker, args=(t,)) for t in range(nthreads)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        assert not errors, errors[:5]
        calls, passes = ix.coalescing_stats()
        assert calls == nthreads * per and passes < calls, (calls, passes)
        # a bad call fails alone and at once; big batches are not held back; switching it off restores the direct path
        with pytest.raises(Exception):
            ix.search(hq[:1], 0, pvs.COSINE)
        gi, gd, gc = ix.search(hq[:40], 7, pvs.COSINE)
        assert np.array_equal(gi, exp[(pvs.COSINE, 7)][0][:40]) and ix.coalescing_stats()[0] == calls
        ix.set_coalescing(0)
        gi, gd, gc = ix.search(hq[:1], 7, pvs.COSINE)
        assert np.array_equal(gi, exp[(pvs.COSINE, 7)][0][:1]) and ix.coalescing_stats()[0] == calls
    finally:
        ix.close()


def test_pass_c_in_global_memory_on_several_streams(pvs):
    """With several streams per index, int8 indexes run pass C out of a global-memory work area (~6 KB of LDS, so that it fits
    beside another search's scan): small and large k (the 512-record LDS sort and the global-memory sort), duplicates (ties by
    id), both metrics, several searches in flight — pages bit-identical to the one-stream form and to the oracle."""
    dim = 384
    rows = orc.synth_rows(91, 0, 60_000, dim)
    rows[40_000:40_050] = rows[7]  # 51 equal rows: a tie block inside every page
    scale = orc.compute_int8_scale(rows)
    codes = orc.quantize_int8(rows, scale)
    queries = orc.synth_rows(0x5EED0077, 0, 40, dim)
    queries[3] = rows[7]
    hq = orc.quantize_int8(queries, scale)
    ix = pvs.VectorIndex(pvs.I8, dim)
    ix.set_scale(scale)
    ix.add_f32(rows)
    try:
        for k in (10, 100, 700, 2000):
            for metric, om in ((pvs.COSINE, orc.COSINE), (pvs.L2, orc.L2)):
                ei, ed = orc.search(orc.I8, om, codes, hq, k, threads=8)
                ix.set_streams(1)
                a = ix.search(hq, k, metric)
                ix.set_streams(2)
                b = ix.search(hq, k, metric)
                for got in (a, b):
                    assert np.array_equal(got[0], ei) and np.array_equal(got[1].view(np.uint32), ed.view(np.uint32)), (k, metric)
        # four searches in flight on their own streams
        dq = pvs.DeviceBuffer.from_numpy(hq, 0)
        k = 100
        ei, ed = orc.search(orc.I8, orc.COSINE, codes, hq, k, threads=8)
        outs = [(pvs.DeviceBuffer(40 * k * 8, 0), pvs.DeviceBuffer(40 * k * 4, 0), pvs.DeviceBuffer(40 * 4, 0)) for _ in range(4)]
        from panoptikon_amd import _lib as L
        tickets = [ix.search_device(dq, L.I8, 40, k, pvs.COSINE, *o) for o in outs]
        for t, o in zip(tickets, outs):
            ix.wait(t)
            assert np.array_equal(o[0].to_numpy(np.int64, (40, k)), ei)
            assert np.array_equal(o[1].to_numpy(np.float32, (40, k)).view(np.uint32), ed.view(np.uint32))
    finally:
        ix.set_streams(1)
        ix.close()


def test_second_sort_key_across_the_shards_of_a_multi_device_index(pvs):
    """pvs_index_set_order_keys on a multi-device index: every shard orders its pages with its rows' keys and ships the keys of the
    page entries with the page record; the root's merge compares (distance, key DESC, id).  Tie-heavy int8 L2 corpus (few
    distinct vectors): the row page, the masked row page, the per-item pages, similar_to and the RRF page must equal the oracle's
    ordering over the whole corpus — and differ from the id-only ordering, or the test shows nothing."""
    rng = np.random.default_rng(12)
    dim, distinct, files, k = 64, 40, 1500, 60
    base = orc.synth_rows(201, 0, distinct, dim)
    per_file = rng.integers(1, 4, files)
    grp = np.repeat(np.arange(files, dtype=np.int64) * 3 + 2, per_file)
    rows = base[np.repeat(rng.integers(0, distinct, files), per_file)]
    n = len(rows)
    ids = np.arange(n, dtype=np.int64) * 2 + 7
    fkey = rng.integers(0, 10, files).astype(np.int64) + 1_700_000_000
    keys = np.repeat(fkey, per_file)
    scale = orc.compute_int8_scale(rows)
    codes = orc.quantize_int8(rows, scale)
    qs = base[[2, 17, 33]] + 0.02 * orc.synth_rows(202, 0, 3, dim)
    hq = orc.quantize_int8(qs, scale)
    mask = (rng.random(n) < 0.5).astype(np.uint8)
    allowed = np.nonzero(mask)[0]
    for devices in _layouts(pvs):
        tag = f"devices={devices}"
        ix = pvs.VectorIndex(pvs.I8, dim, devices=devices)
        ix.set_scale(scale)
        ix.add_f32(rows, row_ids=ids, group_ids=grp)
        ix.set_order_keys(keys)
        differs = 0
        for path in (0, 1):
            ix.set_path(path)
            gi, gd, gc = ix.search(qs, k, pvs.L2)
            fi, fd, fc = ix.search_filtered(qs, k, mask, pvs.L2)
            for j in range(len(qs)):
                d = orc.score_all(orc.I8, orc.L2, codes, hq[j])
                ei, ed = orc.topk_ordered(d, k, ids, keys)
                assert gc[j] == k and np.array_equal(gi[j], ei) and np.array_equal(gd[j].view(np.uint32), ed.view(np.uint32)), (tag, path, j)
                pi, _ = orc.topk(d, k, ids=ids)
                differs += int(not np.array_equal(pi, ei))
                ei, ed = orc.topk_ordered(d[allowed], k, ids[allowed], keys[allowed])
                assert fc[j] == k and np.array_equal(fi[j], ei), (tag, path, j, "masked")
        ix.set_path(0)
        assert differs >= 3, "the corpus must tie across shards"
        for agg, oagg in ((pvs.AGG_MIN, orc.AGG_MIN), (pvs.AGG_AVG, orc.AGG_AVG)):
            og, ov, oc = ix.search_groups(qs, k, pvs.L2, agg)
            for j in range(len(qs)):
                eg, ev = orc.search_groups(orc.I8, orc.L2, codes, hq[j], grp, oagg, k, order_keys=keys)
                assert oc[j] == len(eg) and np.array_equal(og[j, : oc[j]], eg) and np.array_equal(ov[j, : oc[j]].view(np.uint64), ev.view(np.uint64)), (tag, agg, j)
        # OR arm over two multi-device branches: score ties by key
        rows2 = base[np.repeat(rng.integers(0, distinct, files), per_file)]
        ix2 = pvs.VectorIndex(pvs.I8, dim, devices=devices)
        ix2.set_scale(scale)
        ix2.add_f32(rows2, row_ids=ids, group_ids=grp)
        codes2 = orc.quantize_int8(rows2, scale)
        br = [dict(index=ix, query=hq[0], metric=pvs.L2, agg=pvs.AGG_MIN, rrf_k=1, weight=1.0),
              dict(index=ix2, query=hq[1], metric=pvs.L2, agg=pvs.AGG_MIN, rrf_k=1, weight=1.0)]
        fg, fs = pvs.rrf_search(br, 80)
        ora = [dict(dtype=orc.I8, metric=orc.L2, corpus=codes, query=hq[0], groups=grp, agg=orc.AGG_MIN, rrf_k=1, weight=1.0, order_keys=keys),
               dict(dtype=orc.I8, metric=orc.L2, corpus=codes2, query=hq[1], groups=grp, agg=orc.AGG_MIN, rrf_k=1, weight=1.0)]
        eg, es = orc.rrf_search(ora, 80)
        assert np.array_equal(fg, eg) and np.array_equal(fs.view(np.uint64), es.view(np.uint64)), tag
        # removing the keys restores the id order; adding rows drops them
        ix.set_order_keys(None)
        gi, gd, gc = ix.search(qs[:1], k, pvs.L2)
        pi, _ = orc.topk(orc.score_all(orc.I8, orc.L2, codes, hq[0]), k, ids=ids)
        assert np.array_equal(gi[0], pi), tag
        ix2.close()
        ix.close()


def test_sharded_row_search_ships_order_keys_in_the_page_record(pvs):
    """pvs_search_sharded with order keys: the page record carries the keys of its entries and the keyed flag (one rank here: the
    merge is the identity, but the record layout, the key lookup by id and the flag handling on the redo path all run)."""
    rng = np.random.default_rng(13)
    dim, distinct, copies, k = 64, 30, 80, 50
    base = orc.synth_rows(211, 0, distinct, dim)
    rows = np.tile(base, (copies, 1))[rng.permutation(distinct * copies)]
    rows[77] = 0.0
    n = len(rows)
    ids = np.arange(n, dtype=np.int64) * 3 + 1
    keys = rng.integers(0, 30, n).astype(np.int64)
    scale = orc.compute_int8_scale(rows)
    codes = orc.quantize_int8(rows, scale)
    ix = pvs.VectorIndex(pvs.I8, dim)
    ix.set_scale(scale)
    ix.add_f32(rows, row_ids=ids)
    ix.set_order_keys(keys)
    from panoptikon_amd import _lib as L

    uid = (C.c_uint8 * L.UNIQUE_ID_BYTES)()
    L.check(pvs.lib().pvs_comm_unique_id(uid))
    comm = C.c_void_p()
    L.check(pvs.lib().pvs_comm_create(uid, 1, 0, 0, C.byref(comm)))
    qs = (base[[1, 9, 22]] + 0.02 * orc.synth_rows(212, 0, 3, dim)).astype(np.float32)
    hq = orc.quantize_int8(qs, scale)
    try:
        dq = pvs.DeviceBuffer.from_numpy(qs, 0)
        # cosine with k = 2400 = every row: the NULL row is inside the page, so the queries go through the dense redo + second exchange
        for metric, om, kq in ((pvs.L2, orc.L2, k), (pvs.COSINE, orc.COSINE, k), (pvs.COSINE, orc.COSINE, n)):
            b = len(qs)
            oi, od, oc = pvs.DeviceBuffer(b * kq * 8, 0), pvs.DeviceBuffer(b * kq * 4, 0), pvs.DeviceBuffer(b * 4, 0)
            L.check(pvs.lib().pvs_search_sharded(ix._h, comm, dq.ptr, L.F32, b, kq, metric, oi.ptr, od.ptr, oc.ptr))
            gi, gc = oi.to_numpy(np.int64, (b, kq)), oc.to_numpy(np.uint32, (b,))
            for j in range(b):
                d = orc.score_all(orc.I8, om, codes, hq[j])
                ei, ed = orc.topk_ordered(d, kq, ids, keys)
                assert gc[j] == len(ei) and np.array_equal(gi[j, : gc[j]], ei), (metric, kq, j)
    finally:
        pvs.lib().pvs_comm_destroy(comm)
        ix.close()


def test_sharded_fusion_orders_score_ties_by_order_key(pvs):
    """pvs_rrf_search_sharded with order keys: branches sharded BY GROUP over three ranks (host threads over the all-gather
    callback), tie-heavy int8 corpora, keys on the first branch's shards only; the fused page must be the oracle's under
    (score DESC, key DESC, group id) — a candidate's key travels from whichever rank holds its group in the lowest keyed branch."""
    rng = np.random.default_rng(31)
    world, files, dim, k = 3, 3000, 64, 30
    rows_a = orc.synth_rows(401, 0, files, dim)
    grp = np.arange(files, dtype=np.int64) * 2 + 1          # one row per file
    keys = rng.integers(0, 6, files).astype(np.int64)
    scale = orc.compute_int8_scale(rows_a)
    qv = orc.quantize_int8((rows_a[5] + 0.01)[None, :], scale)[0]
    d = orc.score_all(orc.I8, orc.L2, orc.quantize_int8(rows_a, scale), qv)
    top = np.argsort(d, kind="stable")[:40]
    rows_b = rows_a.copy()
    rows_b[top] = rows_a[top[::-1]]                          # the second branch ranks the same 40 files in reverse: pairs tie on the fused score
    specs = [rows_a, rows_b]
    q = [qv, qv]
    ranges = pvs.shard_ranges_by_group(grp, world)
    shards = [[] for _ in range(world)]
    for i, rows in enumerate(specs):
        for r, (a, b) in enumerate(ranges):
            ix = pvs.VectorIndex(pvs.I8, dim, id_base=a)
            ix.set_scale(scale)
            ix.add_f32(rows[a:b], group_ids=grp[a:b])
            if i == 0:
                ix.set_order_keys(keys[a:b])
            shards[r].append(dict(index=ix, query=q[i], metric=pvs.L2, agg=pvs.AGG_MIN, rrf_k=1, weight=1.0))
    ora = [dict(dtype=orc.I8, metric=orc.L2, corpus=orc.quantize_int8(specs[i], scale), query=q[i], groups=grp, agg=orc.AGG_MIN, rrf_k=1, weight=1.0,
                order_keys=keys if i == 0 else None) for i in range(2)]
    eg, es = orc.rrf_search(ora, k)
    pg, _ = orc.rrf_search([{**o, "order_keys": None} for o in ora], k)
    assert not np.array_equal(pg, eg), "the fused scores must tie"
    tg = _ThreadGather(world)
    res, errs = [None] * world, []

    def rank_main(r):
        try:
            res[r] = pvs.rrf_search_sharded(shards[r], k, tg.for_rank(r))
        except Exception as e:  # noqa: BLE001
            errs.append((r, repr(e)))
            tg.bar.abort()

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join(600) for t in th]
    assert not errs, errs
    for r in range(world):
        assert np.array_equal(res[r][0], eg), f"rank {r}: groups differ"
        assert np.array_equal(res[r][1].view(np.uint64), es.view(np.uint64)), f"rank {r}: scores not bit-exact"
    for sh in shards:
        for b in sh:
            b["index"].close()



def test_per_item_work_of_a_multi_device_index_stays_on_the_devices(pvs):
    """rows handed over in HBM are placed by group without visiting the host (a pick kernel per shard + a peer
    copy), a candidate mask resident on devices[0] is split there by one gather per shard, and the shards' per-item pages are
    merged by a kernel on devices[0] (with the second sort key looked up on the shards' devices) — all bit-identical to the
    oracle over the whole corpus and to the host route (pvs_debug_set("multi_host_pages", 1))."""
    from panoptikon_amd import _lib as L

    rng = np.random.default_rng(31)
    dim, files, k = 96, 3000, 50
    per_file = rng.integers(1, 6, files)
    grp = np.repeat(np.arange(files, dtype=np.int64) * 5 + 3, per_file)
    n = len(grp)
    base = orc.synth_rows(401, 0, 60, dim)
    rows = (base[rng.integers(0, 60, n)] + 0.01 * orc.synth_rows(402, 0, n, dim) * (rng.random((n, 1)) < 0.5)).astype(np.float32)  # ties across files
    ids = np.arange(n, dtype=np.int64) * 3 + 1
    keys = np.repeat(rng.integers(0, 6, files).astype(np.int64), per_file)
    scale = orc.compute_int8_scale(rows)
    codes = orc.quantize_int8(rows, scale)
    q = base[[1, 7, 30, 44]] + 0.02 * orc.synth_rows(403, 0, 4, dim)
    hq = orc.quantize_int8(q, scale)
    mask = (rng.random(n) < 0.4).astype(np.uint8)
    allowed = np.nonzero(mask)[0]
    w = (rng.random(n) + 0.2).astype(np.float32)
    for devices in _layouts(pvs):
        tag = f"devices={devices}"
        ix = pvs.VectorIndex(pvs.I8, dim, devices=devices)
        ix.set_scale(scale)
        # rows resident on devices[0], in three adds (a file may straddle two)
        for a in range(0, n, 4000):
            b = min(n, a + 4000)
            dev = pvs.DeviceBuffer.from_numpy(rows[a:b], devices[0])
            L.check(pvs.lib().pvs_index_add_f32(ix._h, dev.ptr, b - a, ids[a:b].ctypes.data_as(C.c_void_p), grp[a:b].ctypes.data_as(C.c_void_p), L.DEVICE))
            dev.free()
        ri, rg = ix.read_ids(0, n, groups=True)
        assert np.array_equal(ri, ids) and np.array_equal(rg, grp), tag
        assert np.array_equal(ix.read_rows(3990, 40), codes[3990:4030]), tag
        dmask = pvs.DeviceBuffer.from_numpy(mask, devices[0])
        for keyed in (False, True):
            ix.set_order_keys(keys if keyed else None)
            ok = keys if keyed else None
            for agg, oagg, ww in ((pvs.AGG_MIN, orc.AGG_MIN, None), (pvs.AGG_MAX, orc.AGG_MAX, None), (pvs.AGG_AVG, orc.AGG_AVG, w)):
                res = {}
                for host_route in (0, 1):
                    pvs.debug_set("multi_host_pages", host_route)
                    try:
                        og = np.empty((len(hq), k), np.int64)
                        ov = np.empty((len(hq), k), np.float64)
                        oc = np.zeros(len(hq), np.uint32)
                        wp = None if ww is None else ww.ctypes.data_as(C.c_void_p)
                        L.check(pvs.lib().pvs_search_groups_filtered(ix._h, hq.ctypes.data_as(C.c_void_p), pvs.I8, len(hq), k, pvs.L2, agg, wp, dmask.ptr, L.DEVICE,
                                                                     og.ctypes.data_as(C.c_void_p), ov.ctypes.data_as(C.c_void_p), oc.ctypes.data_as(C.c_void_p)))
                        full = ix.search_groups(hq, k, pvs.L2, agg, row_weights=ww)
                    finally:
                        pvs.debug_set("multi_host_pages", 0)
                    res[host_route] = (og, ov, oc, full)
                    for j in range(len(hq)):
                        eg, ev = orc.search_groups(orc.I8, orc.L2, codes[allowed], hq[j], grp[allowed], oagg, k, weights=None if ww is None else ww[allowed],
                                                   order_keys=None if ok is None else ok[allowed])
                        assert oc[j] == len(eg) and np.array_equal(og[j, : oc[j]], eg), (tag, keyed, agg, host_route, j)
                        assert np.array_equal(ov[j, : oc[j]].view(np.uint64), ev.view(np.uint64)), (tag, keyed, agg, host_route, j)
                        eg, ev = orc.search_groups(orc.I8, orc.L2, codes, hq[j], grp, oagg, k, weights=ww, order_keys=ok)
                        assert full[2][j] == len(eg) and np.array_equal(full[0][j, : len(eg)], eg), (tag, keyed, agg, host_route, j, "unmasked")
                        assert np.array_equal(full[1][j, : len(eg)].view(np.uint64), ev.view(np.uint64)), (tag, keyed, agg, host_route, j, "unmasked")
                for x, y in zip(res[0][:3], res[1][:3]):
                    assert np.array_equal(x.view(np.uint8), y.view(np.uint8)), (tag, keyed, agg, "device route == host route")
            # the row page under the device-resident mask
            fi = np.empty((len(hq), k), np.int64)
            fd = np.empty((len(hq), k), np.float32)
            fc = np.zeros(len(hq), np.uint32)
            L.check(pvs.lib().pvs_search_filtered(ix._h, hq.ctypes.data_as(C.c_void_p), pvs.I8, len(hq), k, pvs.L2, dmask.ptr, L.DEVICE,
                                                  fi.ctypes.data_as(C.c_void_p), fd.ctypes.data_as(C.c_void_p), fc.ctypes.data_as(C.c_void_p)))
            for j in range(len(hq)):
                d = orc.score_all(orc.I8, orc.L2, codes, hq[j])
                if keyed:
                    ei, ed = orc.topk_ordered(d[allowed], k, ids[allowed], keys[allowed])
                else:
                    ei, ed = orc.topk(d[allowed], k, ids=ids[allowed])
                assert fc[j] == k and np.array_equal(fi[j], ei) and np.array_equal(fd[j].view(np.uint32), ed.view(np.uint32)), (tag, keyed, j, "row page")
        # k beyond one LDS sort of the merge (4,096 < S * k <= 32,768): the shards' pages are merged by rank on devices[0] (round 5:
        # every entry's place = its place in its own page + the entries of the other pages in front of it), same contract, and the
        # same bits as the host merge; a page longer than the union (kk > files) comes back short
        for kk, agg, oagg in ((2100, pvs.AGG_AVG, orc.AGG_AVG), (2999, pvs.AGG_MAX, orc.AGG_MAX), (4000, pvs.AGG_AVG, orc.AGG_AVG)):
            got = {}
            for host_route in (0, 1):
                pvs.debug_set("multi_host_pages", host_route)
                try:
                    got[host_route] = ix.search_groups(hq[:2], kk, pvs.L2, agg)
                finally:
                    pvs.debug_set("multi_host_pages", 0)
            for x, y in zip(got[0], got[1]):
                assert np.array_equal(x.view(np.uint8), y.view(np.uint8)), (tag, kk, agg, "rank merge == host merge")
            og, ov, oc = got[0]
            for j in range(2):
                eg, ev = orc.search_groups(orc.I8, orc.L2, codes, hq[j], grp, oagg, kk, order_keys=keys)
                assert oc[j] == len(eg) and np.array_equal(og[j, : oc[j]], eg) and np.array_equal(ov[j, : oc[j]].view(np.uint64), ev.view(np.uint64)), (tag, kk, agg, j)
        dmask.free()
        ix.close()


def test_a_rank_that_fails_before_the_exchange_fails_the_search_on_every_rank_not_the_job(pvs):
    """Round 6, RCCL first-run hygiene: a rank that fails LOCALLY before the all-gather of a sharded search still sends a record —
    every flag word carries PVS_PAGE_FAILED — so the collective completes and every rank's pvs_wait returns an error for THAT
    search; the communicator and the index keep working.  On a one-GPU box the communicator has one rank (the path is the
    same: enqueue fails -> failure record -> all-gather -> merge -> pvs_wait reports)."""
    from panoptikon_amd import _lib as L

    dim, n, k = 64, 20_000, 10
    rows = orc.synth_rows(81, 0, n, dim)
    ix = pvs.VectorIndex(pvs.F32, dim)
    ix.add_f32(rows)
    uid = (C.c_uint8 * L.UNIQUE_ID_BYTES)()
    L.check(pvs.lib().pvs_comm_unique_id(uid))
    comm = C.c_void_p()
    L.check(pvs.lib().pvs_comm_create(uid, 1, 0, 0, C.byref(comm)))  # (ends with the communicator's first collective: a one-word all-reduce)
    try:
        q = orc.synth_rows(82, 0, 4, dim)
        dq = pvs.DeviceBuffer.from_numpy(q)
        oi, od, oc = pvs.DeviceBuffer(4 * k * 8), pvs.DeviceBuffer(4 * k * 4), pvs.DeviceBuffer(4 * 4)
        ei, ed = orc.search(orc.F32, orc.COSINE, rows, q, k)

        def run():
            L.check(pvs.lib().pvs_search_sharded(ix._h, comm, dq.ptr, L.F32, 4, k, pvs.COSINE, oi.ptr, od.ptr, oc.ptr))
            return oi.to_numpy(np.int64, (4, k)), od.to_numpy(np.float32, (4, k))

        gi, gd = run()
        assert np.array_equal(gi, ei) and np.array_equal(gd.view(np.uint32), ed.view(np.uint32))
        pvs.debug_set("comm_fail_local", 1)
        # stream-ordered form: the enqueue hands out a ticket (the failure record is on its way), pvs_wait reports
        t = C.c_uint32()
        L.check(pvs.lib().pvs_search_sharded_async(ix._h, comm, dq.ptr, L.F32, 4, k, pvs.COSINE, oi.ptr, od.ptr, oc.ptr, C.byref(t)))
        with pytest.raises(pvs.PvsError, match="injected local failure"):
            ix.wait(int(t.value))
        assert pvs.debug_get("comm_fail_local") == 0
        gi, gd = run()  # the communicator, the index and the context are all still good
        assert np.array_equal(gi, ei) and np.array_equal(gd.view(np.uint32), ed.view(np.uint32))
    finally:
        pvs.debug_set("comm_fail_local", 0)
        pvs.lib().pvs_comm_destroy(comm)
        ix.close()


_NO_SHOW = r"""
import ctypes as C, sys, time, os
import panoptikon_amd as pvs
from panoptikon_amd import _lib as L
pvs.debug_set("comm_timeout_s", 4)
uid = (C.c_uint8 * L.UNIQUE_ID_BYTES)()
L.check(pvs.lib().pvs_comm_unique_id(uid))
comm = C.c_void_p()
t0 = time.time()
rc = pvs.lib().pvs_comm_create(uid, 2, 0, 0, C.byref(comm))   # rank 1 never calls
print("RESULT", rc, round(time.time() - t0, 1), pvs.lib().pvs_last_error().decode(), flush=True)
os._exit(0)   # (the helper thread is still inside ncclCommInitRank: leave without running destructors under it)
"""


def test_a_rank_that_never_arrives_is_an_error_not_a_hang(pvs):
    """pvs_comm_create for a 2-rank communicator whose other rank never calls: PVS_ERR_COMM after the deadline
    (pvs_debug "comm_timeout_s"), naming the rank — not an 1,800-s driver timeout.  Own process: ncclCommInitRank cannot be
    cancelled, its helper thread stays behind."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _NO_SHOW], cwd=root, env=dict(os.environ, PYTHONPATH=root, HSA_ENABLE_IPC_MODE_LEGACY="0"),
                       capture_output=True, text=True, timeout=120)
    line = [ln for ln in (r.stdout + r.stderr).splitlines() if ln.startswith("RESULT")]
    assert line, (r.stdout[-1500:], r.stderr[-1500:])
    _, rc, secs, msg = line[0].split(" ", 3)
    assert int(rc) == 9, line[0]  # PVS_ERR_COMM
    assert 3.0 <= float(secs) <= 30.0, line[0]
    assert "rank 0 of 2" in msg and "waited" in msg, msg
