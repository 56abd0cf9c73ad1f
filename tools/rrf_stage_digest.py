"""configs[4] race hunt: per-stage digests of pvs_rrf_search over many back-to-back iterations.

    python tools/rrf_stage_digest.py [--rows 25000000] [--iters 2000] [--mode racing|streams2|serial|bypass] [--out FILE]

One composed query (the PQL `or` of a 512-d cosine and a 1024-d L2 int8 filter, MIN per file, RRF 5/1.0 + 10/0.7 — BASELINE
configs[4]) is answered `iters` times through the bounded fusion and `iters` times through the full ranking.  With
pvs_debug_set("rrf_digest", 1) the library records, per branch, a 64-bit digest of (0) the `d` column, (1) the per-file aggregates,
(2) the window keys, (3) the ranks (bounded: the candidates' counted ranks; full: every file's rank).  All of them are pure
functions of the input, so every iteration must reproduce iteration 0's digests, and the two paths must return the same page.
A digest that moves names the stage (and so the kernel) that is not deterministic; the final page is compared too.

Modes: racing   = the product default (one host thread per branch, shared scratch cache);
       streams2 = pvs_index_set_streams(2) on both indexes (every search context on its own stream);
       serial   = pvs_debug_set("rrf_serial", 1) (branches one after the other on the calling thread);
       bypass   = pvs_debug_set("scratch_bypass", 1) (no scratch block is ever handed out twice).
VERDICT r3 item 1; the reference composition: pql/builder.rs:757-771, 1284-1301.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import oracle as orc
import panoptikon_amd as pvs
from tests.test_gpu_fullsize import _build_i8

STAGES = ("d_column", "file_aggregates", "window_keys", "ranks")


def digests():
    out = (C.c_uint64 * 32)()
    nb, path = C.c_uint32(), C.c_int32()
    pvs._lib.check(pvs.lib().pvs_debug_rrf_digests(out, C.byref(nb), C.byref(path)))
    return np.array(out, dtype=np.uint64).reshape(8, 4)[: nb.value].copy(), int(path.value)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=25_000_000)
    ap.add_argument("--iters", type=int, default=2000)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--mode", default="racing", choices=["racing", "streams2", "serial", "bypass"])
    ap.add_argument("--out", default=None)
    ap.add_argument("--testlike", action="store_true",
                    help="as tests/test_gpu_fullsize.py saw the one event: k alternates between 100 and 1000, and other per-item operators run "
                         "between the fusions (they cycle scratch blocks of other size classes through the cache)")
    a = ap.parse_args()
    n = a.rows
    img = _build_i8(pvs, n, 512, 11, 0.00185, groups_of=lambda r: r // 3)
    txt = _build_i8(pvs, n, 1024, 12, 0.0013, groups_of=lambda r: (r // 3) * 2)
    if a.mode == "streams2":
        img.set_streams(2)
        txt.set_streams(2)
    if a.mode == "serial":
        pvs.debug_set("rrf_serial", 1)
    if a.mode == "bypass":
        pvs.debug_set("scratch_bypass", 1)
    pvs.debug_set("rrf_digest", 1)
    qi, qt = orc.synth_rows(0x5EED0000, 0, 1, 512)[0], orc.synth_rows(0x5EED0011, 0, 1, 1024)[0]
    brs = [dict(index=img, query=qi, metric=pvs.COSINE, agg=pvs.AGG_MIN, rrf_k=5, weight=1.0),
           dict(index=txt, query=qt, metric=pvs.L2, agg=pvs.AGG_MIN, rrf_k=10, weight=0.7)]
    ref = {}
    moved = []
    t0 = time.time()
    for it in range(a.iters):
        kk = a.k if not a.testlike or it % 2 == 0 else 10 * a.k
        if a.testlike and it % 16 == 0:
            pvs.debug_set("rrf_full", 0)
            img.search_groups(qi[None, :], 50, pvs.COSINE, pvs.AGG_MIN)
            txt.search_groups(np.stack([qt, qt]), 20, pvs.L2, pvs.AGG_AVG)
            pvs.rrf_search(brs[:1], 50)
        for full in (0, 1):
            pvs.debug_set("rrf_full", full)
            g, s = pvs.rrf_search(brs, kk)
            dg, path = digests()
            assert path == (2 if full else 1), (path, full)
            key = ("full" if full else "bounded") + (f"_k{kk}" if a.testlike else "")
            if key not in ref:
                ref[key] = (dg, g.copy(), s.copy())
                continue
            rd, rg, rs = ref[key]
            for b in range(dg.shape[0]):
                for st in range(4):
                    if dg[b, st] != rd[b, st]:
                        moved.append({"iteration": it, "path": key, "branch": b, "stage": STAGES[st], "got": hex(int(dg[b, st])), "first": hex(int(rd[b, st]))})
            if not (np.array_equal(g, rg) and np.array_equal(s.view(np.uint64), rs.view(np.uint64))):
                moved.append({"iteration": it, "path": key, "stage": "page", "differs_at": np.nonzero(s.view(np.uint64) != rs.view(np.uint64))[0][:5].tolist()})
        if len(moved) > 50:
            break
        if it % 200 == 0:
            print(f"[digest] {a.mode}: iteration {it}, {len(moved)} moved, {time.time() - t0:.0f}s", file=sys.stderr, flush=True)
    pvs.debug_set("rrf_full", 0)
    # across the paths: the first three stages are the same computation, the page must be identical
    cross = []
    same_page = True
    sfx = sorted({k_[len("bounded"):] for k_ in ref if k_.startswith("bounded")})
    for sx in sfx:
        rb, rf = ref["bounded" + sx], ref["full" + sx]
        for b in range(rb[0].shape[0]):
            for st in range(3):
                if rb[0][b, st] != rf[0][b, st]:
                    cross.append({"branch": b, "stage": STAGES[st], "k": sx})
        same_page &= bool(np.array_equal(rb[1], rf[1]) and np.array_equal(rb[2].view(np.uint64), rf[2].view(np.uint64)))
    res = {"mode": a.mode + ("+testlike" if a.testlike else ""), "rows_per_branch": n, "k": a.k, "iterations": it + 1, "searches": 2 * (it + 1),
           "digests_moved": len(moved), "moved": moved[:50],
           "bounded_vs_full_stage_digests_differ": cross, "bounded_page_equals_full_page": same_page,
           "digests_bounded": [[hex(int(x)) for x in row] for row in ref["bounded" + sfx[0]][0]],
           "digests_full": [[hex(int(x)) for x in row] for row in ref["full" + sfx[0]][0]], "seconds": round(time.time() - t0, 1)}
    line = json.dumps(res)
    print(line)
    if a.out:
        with open(a.out, "w") as f:
            f.write(line + "\n")
    img.close()
    txt.close()
    return 1 if (moved or cross or not same_page) else 0


if __name__ == "__main__":
    sys.exit(main())
