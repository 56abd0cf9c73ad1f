"""Mutation beside searches, inside the library (csrc/pvs_gate.hip).  The reference mutates while it serves: one writer actor
(db/index_writer.rs; db/extraction_write.rs:574-616; db/vector_quants.rs:1347-1438; ON DELETE CASCADE) beside up to 16 read
connections (db/connection.rs:235,320-357) under SQLite's snapshot isolation — a statement sees the tables before or after a
transaction, never in between.  Here: 16 searching threads (row pages, batches, masked pages, per-item pages, similar_to,
stream-ordered tickets left in flight) while a writer thread interleaves add / remove / replace; EVERY returned page must be
the oracle's page over the index as it stood after some whole number of mutations inside the call's window.

Plus the round-5 advisor's findings on pvs_lifecycle.hip: ids that left may come back (tail removal + re-add), the multi-device
parent's id -> row map after remove x + add x."""
import threading
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


def _dt(pvs, name):
    return {"i8": pvs.I8, "f16": pvs.F16, "f32": pvs.F32}[name]


def _host(dt, rows, scale):
    if dt == orc.I8:
        return orc.quantize_int8(rows, scale)
    return rows.astype(np.float16) if dt == orc.F16 else rows


# ------------------------------------------------------------------ ADVICE r5
@pytest.mark.parametrize("devices", [None, [0, 0]])
def test_ids_that_left_may_come_back(pvs, devices):
    """item_data.id is not AUTOINCREMENT: after the newest rows are deleted SQLite hands their ids out again, and
    loader.reconcile_deletions re-adds them.  The index must ascend from its last SURVIVING id."""
    dim, n = 64, 3000
    rows = orc.synth_rows(11, 0, n + 500, dim)
    ids = np.arange(100, 100 + n, dtype=np.int64)
    grp = ids // 5
    ix = pvs.VectorIndex(pvs.F32, dim, devices=devices)
    ix.add_f32(rows[:n], row_ids=ids, group_ids=grp)
    assert ix.remove_rows(ids[-400:]) == 400
    # the same ids again, other vectors; and one id in the middle of what is left must still be refused
    new_ids = ids[-400:].copy()
    with pytest.raises(pvs.PvsError):
        ix.add_f32(rows[n:n + 1], row_ids=ids[n - 401:n - 400], group_ids=grp[n - 401:n - 400])
    ix.add_f32(rows[n:n + 400], row_ids=new_ids, group_ids=new_ids // 5)
    cur = np.concatenate([rows[:n - 400], rows[n:n + 400]])
    assert np.array_equal(ix.read_ids(0, n), ids)
    q = orc.synth_rows(5, 0, 1, dim)
    gi, gd, gc = ix.search(q, 20, pvs.COSINE)
    ei, ed = orc.search(orc.F32, orc.COSINE, cur, q, 20, ids=ids)
    assert np.array_equal(gi[0, :20], ei[0]) and np.array_equal(gd[0, :20].view(np.uint32), ed[0].view(np.uint32))
    # everything removed: any id may come next
    assert ix.remove_rows(ids) == n
    ix.add_f32(rows[:10], row_ids=np.arange(5, 15, dtype=np.int64), group_ids=np.arange(5, 15, dtype=np.int64))
    assert np.array_equal(ix.read_ids(0, 10), np.arange(5, 15))
    ix.close()


def test_multi_device_id_map_after_equal_remove_and_add(pvs):
    """remove x rows + add x rows: the parent's row count is what it was, its id -> row map is not (similar_to resolves its
    targets through it; the keyed host merge looks order keys up through it)."""
    dim, n = 64, 2400
    rows = orc.synth_rows(12, 0, n + 300, dim)
    ids = np.arange(1000, 1000 + n, dtype=np.int64)
    grp = ids // 6
    ix = pvs.VectorIndex(pvs.F32, dim, devices=[0, 0, 0])
    ix.add_f32(rows[:n], row_ids=ids, group_ids=grp)
    tg = ids[600:606].copy()
    ix.similar_to(tg, 10, pvs.COSINE, pvs.AGG_AVG)  # (warms the id cache)
    gone = ids[100:400].copy()
    assert ix.remove_rows(gone) == 300
    new_ids = np.arange(5000, 5300, dtype=np.int64)
    ix.add_f32(rows[n:n + 300], row_ids=new_ids, group_ids=new_ids // 6)
    keep = ~np.isin(ids, gone)
    cur_rows = np.concatenate([rows[:n][keep], rows[n:n + 300]])
    cur_ids = np.concatenate([ids[keep], new_ids])
    cur_grp = cur_ids // 6
    assert ix.stats().rows == n
    sg, sv = ix.similar_to(tg, 10, pvs.COSINE, pvs.AGG_AVG)
    trow = [int(np.searchsorted(cur_ids, t)) for t in tg]
    eg, ev = orc.similar_to(orc.F32, orc.COSINE, cur_rows, trow, cur_grp, orc.AGG_AVG, 10)
    assert np.array_equal(sg, eg)
    assert np.array_equal(np.asarray(sv).view(np.uint64), np.asarray(ev).view(np.uint64))
    # a target that left is refused, one that arrived is found
    with pytest.raises(pvs.PvsError):
        ix.similar_to(gone[:2], 5, pvs.COSINE, pvs.AGG_AVG)
    ix.similar_to(new_ids[:6], 5, pvs.COSINE, pvs.AGG_AVG)
    ix.close()


# ------------------------------------------------------------------ the gate
class World:
    """Every row version that ever existed (f32 originals) and the states the index went through: state j = after j mutations."""

    def __init__(self, dim):
        self.dim = dim
        self.vecs = []            # list of [m][dim] blocks; version v = row v of their concatenation
        self.n_vecs = 0
        self.states = []          # (ids, ver) per state
        self.started = 0          # mutations started   (written by the writer only)
        self.completed = 0        # mutations completed

    def new_versions(self, rows):
        v0 = self.n_vecs
        self.vecs.append(rows)
        self.n_vecs += len(rows)
        return np.arange(v0, v0 + len(rows), dtype=np.int64)


PROTECTED = 1200  # the first rows of the index are never removed or replaced: masks / targets over them mean the same in every state


def _writer(pvs, ix, w, rng, stop, errors, dt, scale):
    try:
        ids, ver = w.states[0]
        ids, ver = ids.copy(), ver.copy()
        next_id = int(ids[-1]) + 1
        base = np.concatenate(w.vecs)[:400]
        while not stop.is_set():
            op = rng.choice(["add", "remove", "replace", "remove_tail"], p=[0.4, 0.3, 0.2, 0.1])
            n = len(ids)
            if op == "add" or n < PROTECTED + 600:
                cnt = int(rng.integers(8, 60))
                rows = (base[rng.integers(0, 400, cnt)] + 0.25 * orc.synth_rows(int(rng.integers(1, 1 << 30)), 0, cnt, w.dim)).astype(np.float32)
                nid = next_id + np.cumsum(rng.integers(1, 3, cnt)).astype(np.int64)
                nver = w.new_versions(rows)
                w.started += 1
                ix.add_f32(rows, row_ids=nid, group_ids=nid // 4)
                ids, ver = np.concatenate([ids, nid]), np.concatenate([ver, nver])
                next_id = int(nid[-1]) + 1
            elif op == "remove":
                pos = PROTECTED + rng.choice(n - PROTECTED, int(rng.integers(5, 50)), replace=False)
                w.started += 1
                assert ix.remove_rows(ids[pos].copy()) == len(pos)
                keep = np.ones(n, bool)
                keep[pos] = False
                ids, ver = ids[keep], ver[keep]
            elif op == "remove_tail":
                cnt = int(rng.integers(1, 30))
                w.started += 1
                assert ix.remove_rows(ids[-cnt:].copy()) == cnt
                ids, ver = ids[:-cnt], ver[:-cnt]
                next_id = int(ids[-1]) + 1  # (ids that left come back)
            else:
                pos = np.sort(PROTECTED + rng.choice(n - PROTECTED, int(rng.integers(4, 30)), replace=False))
                rows = orc.synth_rows(int(rng.integers(1, 1 << 30)), 0, len(pos), w.dim)
                nver = w.new_versions(rows)
                w.started += 1
                ix.replace_rows(rows, ids[pos].copy())
                ver = ver.copy()
                ver[pos] = nver
            w.states.append((ids.copy(), ver.copy()))
            w.completed += 1
            time.sleep(0.0004)
    except Exception as e:  # noqa: BLE001
        errors.append(("writer", repr(e)))
        stop.set()


def _reader(pvs, ix, w, kind, qs, mask, targets, stop, log, errors, device):
    try:
        from panoptikon_amd import _lib as L

        i = 0
        dq = None
        if kind == "tickets":
            dq = [pvs.DeviceBuffer.from_numpy(qs[j][None, :].astype(np.float32), device) for j in range(len(qs))]
            outs = [(pvs.DeviceBuffer(10 * 8, device), pvs.DeviceBuffer(10 * 4, device), pvs.DeviceBuffer(4, device)) for _ in range(3)]
        while not stop.is_set():
            j = i % len(qs)
            lo = w.completed
            if kind == "row":
                gi, gd, gc = ix.search(qs[j], 10, pvs.COSINE)
                res = (gi[0, :gc[0]].copy(), gd[0, :gc[0]].copy())
            elif kind == "batch":
                gi, gd, gc = ix.search(qs, 10, pvs.L2)
                res = (gi[j, :gc[j]].copy(), gd[j, :gc[j]].copy())
            elif kind == "masked":
                # (through the C ABI directly: the mask is longer than the index — the Python wrapper insists on one byte per row,
                # which no caller can promise while a writer is at work; rows beyond PROTECTED are never allowed)
                gi, gd, gc = np.full((1, 10), -1, np.int64), np.full((1, 10), np.nan, np.float32), np.zeros(1, np.uint32)
                qq = np.ascontiguousarray(qs[j][None, :], np.float32)
                L.check(L.lib().pvs_search_filtered(ix._h, qq.ctypes.data, L.F32, 1, 10, pvs.COSINE, mask.ctypes.data, L.HOST, gi.ctypes.data,
                                                    gd.ctypes.data, gc.ctypes.data))
                res = (gi[0, :gc[0]].copy(), gd[0, :gc[0]].copy())
            elif kind == "items":
                og, ov, oc = ix.search_groups(qs[j], 8, pvs.COSINE, pvs.AGG_AVG)
                res = (og[0, :oc[0]].copy(), ov[0, :oc[0]].copy())
            elif kind == "items_min":
                og, ov, oc = ix.search_groups(qs[j], 8, pvs.COSINE, pvs.AGG_MIN)
                res = (og[0, :oc[0]].copy(), ov[0, :oc[0]].copy())
            elif kind == "similar":
                sg, sv = ix.similar_to(targets, 8, pvs.COSINE, pvs.AGG_AVG)
                res = (np.asarray(sg).copy(), np.asarray(sv).copy())
            else:  # "tickets": three stream-ordered searches left in flight for a while, then waited for
                ts = []
                for s in range(3):
                    lo_s = w.completed
                    try:
                        ts.append((ix.search_device(dq[(j + s) % len(qs)], L.F32, 1, 10, pvs.COSINE, *outs[s]), (j + s) % len(qs), lo_s))
                    except pvs.PvsError as e:
                        # every one of the 16 contexts is taken (the reference's pool size): the stream-ordered entry says so instead of
                        # blocking — its caller may be the thread that has to pvs_wait one.  Wait for what we hold and go on.
                        if e.status != 5:
                            raise
                        break
                time.sleep(0.0005)
                for s, (t, jj, lo_s) in enumerate(ts):
                    ix.wait(t)
                    c = int(outs[s][2].to_numpy(np.uint32, (1,))[0])
                    log.append(("row", jj, lo_s, w.started, (outs[s][0].to_numpy(np.int64, (10,))[:c], outs[s][1].to_numpy(np.float32, (10,))[:c])))
                i += 1
                continue
            hi = w.started
            log.append(("row" if kind == "row" else kind, j, lo, hi, res))
            i += 1
    except Exception as e:  # noqa: BLE001
        errors.append((kind, repr(e)))
        stop.set()


@pytest.mark.parametrize("dtype,devices", [("i8", None), ("f16", None), ("f32", None), ("f32", [0, 0, 0])])
def test_sixteen_searching_threads_beside_a_writer(pvs, dtype, devices):
    dt = _dt(pvs, dtype)
    dim = 64
    seconds = 5.0
    rng = np.random.default_rng(4242 + len(dtype) + (3 if devices else 0))
    n0 = 4000
    rows0 = orc.synth_rows(900, 0, n0, dim)
    scale = orc.compute_int8_scale(rows0) * 1.3
    ids0 = np.arange(10, 10 + n0, dtype=np.int64)
    w = World(dim)
    ver0 = w.new_versions(rows0)
    w.states.append((ids0.copy(), ver0.copy()))
    ix = pvs.VectorIndex(dt, dim, devices=devices)
    if dt == pvs.I8:
        ix.set_scale(scale)
    ix.add_f32(rows0, row_ids=ids0, group_ids=ids0 // 4)
    qs = orc.synth_rows(77, 0, 6, dim)
    qs[0] = rows0[50]
    mask = np.zeros(1 << 16, np.uint8)  # (longer than the index ever gets: rows beyond PROTECTED are never allowed)
    mask[:PROTECTED] = rng.random(PROTECTED) < 0.4
    targets = ids0[400:404].copy()      # one whole file inside the protected prefix
    kinds = ["row"] * 4 + ["batch"] * 2 + ["masked"] * 2 + ["items"] * 2 + ["items_min"] * 2 + ["similar"] * 2 + ["tickets"] * 2
    assert len(kinds) == 16
    if devices:  # (a multi-device index serves stream-ordered searches from devices[0]: same test)
        pass
    stop, errors = threading.Event(), []
    logs = [[] for _ in kinds]
    device = devices[0] if devices else 0
    threads = [threading.Thread(target=_reader, args=(pvs, ix, w, kd, qs, mask, targets, stop, logs[t], errors, device), daemon=True) for t, kd in enumerate(kinds)]
    wt = threading.Thread(target=_writer, args=(pvs, ix, w, np.random.default_rng(1), stop, errors, dt, scale), daemon=True)  # (daemon: a deadlock fails the test instead of hanging the interpreter at exit)
    for t in threads:
        t.start()
    wt.start()
    time.sleep(seconds)
    stop.set()
    for t in threads + [wt]:
        t.join(timeout=30)
    stuck = [kd for t, kd in zip(threads + [wt], kinds + ["writer"]) if t.is_alive()]
    assert not stuck, f"threads that did not come back (deadlock at the gate?): {stuck}; mutations started {w.started}, completed {w.completed}"
    assert not errors, errors
    n_mut = w.completed
    n_calls = sum(len(lg) for lg in logs)
    assert n_mut >= 50, f"only {n_mut} mutations in {seconds} s: the writer is being starved"
    assert n_calls >= 400, f"only {n_calls} searches in {seconds} s"
    for t, kd in enumerate(kinds):
        assert len(logs[t]) >= 5, f"thread {t} ({kd}) completed {len(logs[t])} calls: readers are being starved"

    # ---- verification against the oracle: every logged page equals the page over SOME state of its window
    allv = np.concatenate(w.vecs)
    hv = _host(dt, allv, scale)
    hq = orc.quantize_int8(qs, scale) if dt == pvs.I8 else qs
    dist_cache = {}

    def col(metric, j):  # distance of every row version to query j
        key = (metric, j)
        if key not in dist_cache:
            dist_cache[key] = orc.score_all(dt, metric, hv, hq[j], threads=8)
        return dist_cache[key]

    page_cache = {}

    def expected(kind, j, s):
        key = (kind, j, s)
        if key in page_cache:
            return page_cache[key]
        ids, ver = w.states[s]
        if kind in ("row", "batch", "masked"):
            d = col(pvs.L2 if kind == "batch" else pvs.COSINE, j)[ver]
            if kind == "masked":
                al = np.nonzero(mask[:len(ids)])[0]
                out = orc.topk_ordered(d[al], 10, ids[al], np.zeros(len(al), np.int64))
            else:
                out = orc.topk_ordered(d, 10, ids, np.zeros(len(ids), np.int64))
        elif kind in ("items", "items_min"):
            d = col(pvs.COSINE, j)[ver]
            g, v = orc.aggregate(d, ids // 4, orc.AGG_AVG if kind == "items" else orc.AGG_MIN)
            out = orc._rank_groups(g, v, 8)
        else:  # similar
            trow = [int(np.searchsorted(ids, t)) for t in targets]
            out = orc.similar_to(dt, pvs.COSINE, hv[ver], trow, ids // 4, orc.AGG_AVG, 8)
        page_cache[key] = out
        return out

    def same(a, b):
        if len(a[0]) != len(b[0]) or not np.array_equal(a[0], b[0]):
            return False
        x, y = np.asarray(a[1]), np.asarray(b[1])
        if x.dtype == np.float32:
            return np.array_equal(x.view(np.uint32), np.asarray(y, np.float32).view(np.uint32))
        return np.array_equal(x.view(np.uint64), np.asarray(y, np.float64).view(np.uint64))

    checked = overlapped = 0
    vr = np.random.default_rng(5)
    for t, kd in enumerate(kinds):
        lg = logs[t]
        # every call whose window holds a mutation (the interesting ones, capped) + a sample of the quiet ones
        hot = [r for r in lg if r[3] > r[2]]
        quiet = [r for r in lg if r[3] == r[2]]
        take = [hot[i] for i in vr.permutation(len(hot))[:120]] + [quiet[i] for i in vr.permutation(len(quiet))[:60]]
        for kind, j, lo, hi, res in take:
            hi = min(hi, n_mut)
            ok = any(same(res, expected(kind, j, s)) for s in range(lo, hi + 1))
            assert ok, (dtype, devices, kd, j, lo, hi, res[0][:5], expected(kind, j, lo)[0][:5], expected(kind, j, hi)[0][:5])
            checked += 1
            overlapped += hi > lo
    assert overlapped >= 20, f"only {overlapped} of the checked calls overlapped a mutation: the test did not exercise the gate"
    print(f"[gate] {dtype} devices={devices}: {n_mut} mutations, {n_calls} searches in {seconds} s; {checked} pages checked against the oracle, "
          f"{overlapped} of them overlapping a mutation")
    ix.close()
