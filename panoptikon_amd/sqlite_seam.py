"""The `dist_{cte}` seam, concretely: fill a TEMP table with the device-computed distance column so that the
rest of the SQL the reference generates (GROUP BY file_id aggregate, row_number(), RRF ORDER BY, LIMIT/OFFSET
— filters/exact.rs:106-165, pql/builder.rs:757-771,1284-1317) runs unchanged on top of it.

The reference's CTE body is `SELECT item_id, file_id, vec_distance_*(payload, ?) AS d FROM <candidate skeleton>`;
here `d` comes from `pvs_score_all` (one exact distance per stored row, in `item_data.id` order) and is joined
back by `item_data.id`.  NaN distances become SQL NULL, as `sqlite3_result_double(NaN)` does in the reference.
Host-side glue (stdlib `sqlite3`); the compute is the library's.
"""
from __future__ import annotations

import math
import sqlite3

import numpy as np

from . import _lib as L


def fill_distance_table(conn: sqlite3.Connection, table: str, index, query, metric: int = L.COSINE) -> int:
    """CREATE TEMP TABLE <table>(id INTEGER PRIMARY KEY, d REAL) holding vec_distance(row, query) for every row of
    `index`; `id` is the row id given at pvs_index_add (item_data.id).  Returns the number of rows written."""
    if not table.replace("_", "").isalnum():
        raise ValueError("table name must be a plain identifier")
    d = index.score_all(query, metric)  # f32, reference arithmetic
    ids = index.read_ids()
    conn.execute(f"DROP TABLE IF EXISTS temp.{table}")
    conn.execute(f"CREATE TEMP TABLE {table} (id INTEGER PRIMARY KEY, d REAL)")
    dd = d.astype(np.float64)  # the f32 result widened, like sqlite3_result_double((double)f32)
    conn.executemany(f"INSERT INTO {table} (id, d) VALUES (?, ?)",
                     ((int(i), None if math.isnan(v) else float(v)) for i, v in zip(ids.tolist(), dd.tolist())))
    return int(ids.size)
