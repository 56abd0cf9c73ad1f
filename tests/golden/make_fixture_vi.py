"""Generates tests/golden/fixture_vi.npz — SURVEY.md §8c golden fixture (vi): a 10,000 x 512 seeded synthetic corpus and 8 queries
-> top-10 (ids, distances) for cosine-f32, L2-f32, cosine-i8, L2-i8 (plus the f16 pair), computed by the CPU oracle
(oracle/pvs_oracle.c).  The corpus is not stored: it is a pure function of the seed (orc.synth_rows); its CRC32 is, so a drift of
the generator is caught too.  Frozen expectations: the kernels AND the oracle are checked against these bytes, so the two cannot
drift together on the float paths.

    python tests/golden/make_fixture_vi.py      (re-run only when the contract itself changes; commit the .npz)
"""
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle as orc  # noqa: E402

N, D, NQ, K = 10_000, 512, 8, 10
SEED_ROWS, SEED_Q = 20260928, 0x5EED0000


def build():
    rows = orc.synth_rows(SEED_ROWS, 0, N, D)
    queries = orc.synth_rows(SEED_Q, 0, NQ, D)
    scale = orc.compute_int8_scale(rows)
    out = {"n": N, "dim": D, "k": K, "seed_rows": SEED_ROWS, "seed_queries": SEED_Q, "scale": np.float32(scale),
           "rows_crc32": np.uint32(zlib.crc32(rows.tobytes())), "queries_crc32": np.uint32(zlib.crc32(queries.tobytes()))}
    corp = {"f32": (orc.F32, rows, queries), "f16": (orc.F16, rows.astype(np.float16), queries),
            "i8": (orc.I8, orc.quantize_int8(rows, scale), orc.quantize_int8(queries, scale))}
    for name, (dt, c, q) in corp.items():
        for mname, m in (("cosine", orc.COSINE), ("l2", orc.L2)):
            ids, dist = orc.search(dt, m, c, q, K)
            out[f"{name}_{mname}_ids"] = ids.astype(np.int64)
            out[f"{name}_{mname}_dist"] = dist.astype(np.float32)
    return out


if __name__ == "__main__":
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "fixture_vi.npz"), **build())
    print("wrote fixture_vi.npz")
