"""Times the index lifecycle glue on a synthetic Panoptikon index database: rows/s from SQLite to a device-resident index,
Python chunk loader (panoptikon_amd/loader.py) against the C streamer of libpvs_sqlite.so (pvs_load).
Usage: python tools/loader_bench.py [--rows 300000] [--dim 768]  -> one JSON line."""
import argparse
import json
import os
import sqlite3
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=300_000)
    ap.add_argument("--dim", type=int, default=768)
    args = ap.parse_args()
    import panoptikon_amd as pvs
    from panoptikon_amd import loader
    from test_loader import DDL

    path = os.path.join(tempfile.mkdtemp(prefix="pvs_loader_bench_"), "index.db")
    conn = sqlite3.connect(path, isolation_level=None)
    conn.execute('BEGIN')
    conn.executescript(DDL)
    conn.execute("INSERT INTO setters (id, name) VALUES (1, 'clip/m')")
    rng = np.random.default_rng(1)
    t0 = time.time()
    step = 20_000
    for off in range(0, args.rows, step):
        m = min(step, args.rows - off)
        mat = rng.standard_normal((m, args.dim), dtype=np.float32)
        conn.executemany("INSERT INTO item_data (id, item_id, setter_id, data_type, idx) VALUES (?, ?, 1, 'clip', 0)",
                         [(off + i + 1, (off + i) // 3 + 1) for i in range(m)])
        conn.executemany("INSERT INTO embeddings (id, embedding) VALUES (?, ?)", [(off + i + 1, mat[i].tobytes()) for i in range(m)])
    conn.commit()
    build_s = time.time() - t0
    out = {"rows": args.rows, "dim": args.dim, "db_bytes": os.path.getsize(path), "db_build_s": round(build_s, 1)}
    for label, native in (("python_chunks", False), ("c_streamer", True), ("python_chunks_again", False), ("c_streamer_again", True)):
        t0 = time.time()
        li = loader.load_exact_index(conn, ["clip/m"], dtype=pvs.F16, native=native)
        pvs.lib().pvs_device_synchronize(0)
        dt = time.time() - t0
        assert li.rows == args.rows
        out[label] = {"seconds": round(dt, 3), "rows_per_s": round(args.rows / dt), "MB_per_s": round(args.rows * args.dim * 4 / dt / 1e6)}
        li.index.close()
    # write side: the backfill of one (profile, setter) pair — the reference measured 49.8 s for 1.45M vectors
    # (docs/vector-quant-measurements.md: quantize_int8 in Rust + one INSERT per row), i.e. ~29k rows/s
    import struct

    from panoptikon_amd import sqlite_seam
    from test_loader import _BACKFILL_SELECT, _BACKFILL_UPSERT

    conn.execute("INSERT INTO vector_quant_profiles (id, name, quantizer, state, is_default) VALUES (5, 'int8', 'int8', 'active', 1)")
    conn.execute("INSERT INTO vector_quant_coverage (profile_id, setter_id, artifact, artifact_rev, dim, state) VALUES (5, 1, ?, 1, ?, 'building')",
                 (struct.pack("<f", 4.5 / 127.0), args.dim))
    conn.commit()
    sqlite_seam.load(conn)
    for label, chunk in (("backfill_chunks_of_8192", 8192), ("backfill_rebuild_chunks_of_65536", 65536)):
        if label.startswith("backfill_rebuild"):
            conn.execute("UPDATE vector_quant_coverage SET artifact_rev = artifact_rev + 1 WHERE profile_id = 5")
            conn.commit()
        t0 = time.time()
        cursor, total = 0, 0
        phases = [0.0, 0.0, 0.0]
        while True:
            conn.execute("BEGIN IMMEDIATE")
            n = conn.execute("SELECT pvs_backfill(?, ?, 5, 0, 5, 1, ?, ?)", (_BACKFILL_SELECT, _BACKFILL_UPSERT, cursor, chunk)).fetchone()[0]
            conn.execute("COMMIT")
            if n == 0:
                break
            cursor = conn.execute("SELECT pvs_backfill_cursor()").fetchone()[0]
            for i, v in enumerate(conn.execute("SELECT pvs_backfill_phases()").fetchone()[0].split(",")):
                phases[i] += float(v)
            total += n
        dt = time.time() - t0
        assert total == args.rows
        out[label] = {"seconds": round(dt, 3), "rows_per_s": round(args.rows / dt), "vs_reference_29k_rows_per_s": round(args.rows / dt / 29100, 1),
                      "inside_pvs_backfill_ms": {"stepping_the_select": round(phases[0], 1), "device_codec_incl_copies": round(phases[1], 1),
                                                  "upserts": round(phases[2], 1)}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
