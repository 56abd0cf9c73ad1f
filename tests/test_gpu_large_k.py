"""Deep pages and deep sort bounds (`LIMIT ? OFFSET ?` far into a result, pql/builder.rs:578-582; `gt` / `lt` on order_rank,
builder.rs:781-815): the dense path no longer sorts every row for them — a sampled threshold admits ~1.3 k rows, only those are
sorted (csrc/pvs_dense.hip, round 5).  Same pages as the full sort and as the oracle, bit for bit, and a k = 16,384 page over
10M x 768 int8 within 2x of a k = 100 page."""
import time

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


@pytest.mark.parametrize("dtype", ["i8", "f32"])
def test_deep_pages_come_from_the_page_first_dense_path(pvs, dtype):
    dt = {"i8": pvs.I8, "f32": pvs.F32}[dtype]
    rng = np.random.default_rng(77)
    n, dim = 400_000, 64
    rows = orc.synth_rows(31, 0, n, dim)
    rows[rng.integers(0, n, 200)] = 0.0                    # NULL cosine distances
    rows[1000:1400] = rows[999]                            # a run of exact ties
    rows[rng.integers(0, n, 5000)] = rows[rng.integers(0, n, 5000)]  # scattered duplicates
    scale = orc.compute_int8_scale(rows)
    ids = np.cumsum(rng.integers(1, 4, n)).astype(np.int64)
    keys = rng.integers(0, 7, n).astype(np.int64)
    ix = pvs.VectorIndex(dt, dim)
    if dt == pvs.I8:
        ix.set_scale(scale)
    ix.add_f32(rows, row_ids=ids)
    hc = orc.quantize_int8(rows, scale) if dt == pvs.I8 else rows
    q = orc.synth_rows(32, 0, 2, dim)
    q[1] = rows[999]
    hq = orc.quantize_int8(q, scale) if dt == pvs.I8 else q
    for keyed in (False, True):
        ix.set_order_keys(keys if keyed else None)
        for metric in (pvs.COSINE, pvs.L2):
            for j in range(2):
                d = orc.score_all(dt, metric, hc, hq[j])
                for k in (5000, 16384, 40000):
                    before = pvs.debug_get("dense_page_first")
                    gi, gd, gc = ix.search(q[j], k, metric)
                    assert pvs.debug_get("dense_page_first") == before + 1, "a deep page takes the threshold + short sort"
                    ei, ed = orc.topk_ordered(d, k, ids, keys if keyed else np.zeros(n, np.int64))
                    assert gc[0] == k and np.array_equal(gi[0], ei), (dtype, keyed, metric, j, k)
                    fin = ~np.isnan(ed)
                    assert np.array_equal(np.isnan(gd[0]), ~fin) and np.array_equal(gd[0][fin].view(np.uint32), ed[fin].view(np.uint32))
                # the full sort says the same
                pvs.debug_set("dense_full_sort", 1)
                try:
                    fi, fd, fc = ix.search(q[j], 16384, metric)
                finally:
                    pvs.debug_set("dense_full_sort", 0)
                gi, gd, gc = ix.search(q[j], 16384, metric)
                assert np.array_equal(fi, gi) and np.array_equal(fd.view(np.uint32), gd.view(np.uint32))
                # a lower sort bound 30,000 rows deep (the growing pages stop at 4,096 rows: the dense path answers)
                order = np.sort(d[~np.isnan(d)])
                gt, lt = float(order[30_000]), float(order[30_000 + 9000])
                for bounds in ((gt, None), (gt, lt)):
                    bi, bd, bc = ix.search_bounded(q[j], 50, metric, gt=bounds[0], lt=bounds[1])
                    ok = d > bounds[0]
                    if bounds[1] is not None:
                        ok &= d < bounds[1]
                    sel = np.nonzero(ok)[0]
                    ei, ed = orc.topk_ordered(d[sel], 50, ids[sel], keys[sel] if keyed else np.zeros(len(sel), np.int64))
                    assert bc[0] == len(ei) and np.array_equal(bi[0, :bc[0]], ei) and np.array_equal(bd[0, :bc[0]].view(np.uint32), ed.view(np.uint32)), (dtype, keyed, metric, j, bounds)
    # a mask that leaves fewer live rows than k: the full sort answers (every allowed row, then nothing)
    mask = np.zeros(n, np.uint8)
    mask[rng.choice(n, 3000, replace=False)] = 1
    gi, gd, gc = ix.search_filtered(q[0], 5000, mask, pvs.L2)
    assert gc[0] == 3000
    ix.close()


def test_a_page_of_sixteen_thousand_rows_of_ten_million_within_twice_a_page_of_a_hundred(pvs):
    from panoptikon_amd import _lib as L

    lib = pvs.lib()
    n, dim = 10_000_000, 768
    ix = pvs.VectorIndex(pvs.I8, dim, capacity_rows=n)
    ix.set_scale(0.2 / 127)
    stage = pvs.DeviceBuffer(1_000_000 * dim * 4)
    for off in range(0, n, 1_000_000):
        L.check(lib.pvs_synth_rows_f32(0, 20260928, off, 1_000_000, dim, stage.ptr))
        ix.add_f32((stage, 1_000_000))
    stage.free()
    q = orc.synth_rows(0x5EED0000, 0, 8, dim)

    def p50(k):
        for i in range(4):
            ix.search(q[i], k, pvs.COSINE)
        ts = []
        for i in range(24):
            t = time.perf_counter()
            ix.search(q[i % 8], k, pvs.COSINE)
            ts.append(time.perf_counter() - t)
        return float(np.sort(ts)[12])

    t100, t16k = p50(100), p50(16384)
    pvs.debug_set("dense_full_sort", 1)
    try:
        tfull = p50(16384)
    finally:
        pvs.debug_set("dense_full_sort", 0)
    print(f"10M x 768 int8, one query: k=100 {t100 * 1e3:.3f} ms, k=16384 {t16k * 1e3:.3f} ms (full sort: {tfull * 1e3:.3f} ms)")
    assert t16k <= 2.0 * t100, (t100, t16k)
    # the deep page extends the shallow one
    a = ix.search(q[0], 100, pvs.COSINE)
    b = ix.search(q[0], 16384, pvs.COSINE)
    assert np.array_equal(a[0][0], b[0][0, :100]) and np.array_equal(a[1][0].view(np.uint32), b[1][0, :100].view(np.uint32))
    ix.close()
