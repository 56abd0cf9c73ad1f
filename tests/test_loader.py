"""Index lifecycle glue (panoptikon_amd/loader.py) against a miniature Panoptikon index database.

The DDL below restates the columns of the reference's migrations that the path touches
(migrations/index/20250117193000_init.sql: items, setters, item_data, embeddings;
20260720130000_vector_quants.sql + 20260730150000_embedding_quants_rowid.sql: vector_quant_profiles,
vector_quant_coverage, embedding_quants).  Readiness cases follow resolve_ready_pair
(db/vector_quants.rs:1795-1869)."""
import os
import sqlite3
import struct

import numpy as np
import pytest

import oracle as orc

DDL = """
CREATE TABLE items (id INTEGER PRIMARY KEY, sha256 TEXT UNIQUE NOT NULL);
CREATE TABLE setters (id INTEGER PRIMARY KEY, name TEXT NOT NULL UNIQUE);
CREATE TABLE item_data (id INTEGER PRIMARY KEY, item_id INTEGER NOT NULL, setter_id INTEGER NOT NULL,
                        data_type TEXT NOT NULL, idx INTEGER NOT NULL);
CREATE TABLE embeddings (id INTEGER PRIMARY KEY, embedding float[]);
CREATE TABLE vector_quant_profiles (id INTEGER PRIMARY KEY, name TEXT UNIQUE NOT NULL, quantizer TEXT NOT NULL, options TEXT,
                                    state TEXT NOT NULL, is_default INTEGER NOT NULL DEFAULT 0);
CREATE TABLE vector_quant_coverage (profile_id INTEGER NOT NULL, setter_id INTEGER NOT NULL, needs_artifact INTEGER NOT NULL DEFAULT 1,
                                    artifact BLOB, artifact_rev INTEGER NOT NULL DEFAULT 0, n_at_artifact INTEGER, dim INTEGER,
                                    metric TEXT, state TEXT NOT NULL DEFAULT 'pending', PRIMARY KEY (profile_id, setter_id));
CREATE TABLE embedding_quants (id INTEGER NOT NULL, profile_id INTEGER NOT NULL, rev INTEGER NOT NULL, quant BLOB NOT NULL,
                               UNIQUE (id, profile_id));
"""


def build_db(n_items=300, dim=96, seed=3, ragged=True):
    """Two setters of one embedding space (an image model and its 't'-prefixed text sibling), interleaved
    item_data ids, one active default int8 profile with a ready pair per setter at artifact_rev 2."""
    rng = np.random.default_rng(seed)
    conn = sqlite3.connect(":memory:")
    conn.executescript(DDL)
    conn.executemany("INSERT INTO setters (id, name) VALUES (?, ?)", [(1, "clip/m"), (2, "tclip/m"), (3, "other")])
    conn.executemany("INSERT INTO items (id, sha256) VALUES (?, ?)", [(i, f"sha{i}") for i in range(1, n_items + 1)])
    rows = orc.synth_rows(1234 + seed, 0, 2 * n_items, dim)
    did, meta = 0, []
    for i in range(1, n_items + 1):
        for setter in (1, 2):
            if setter == 2 and rng.random() < 0.3:
                continue  # not every item has text
            did += 1
            conn.execute("INSERT INTO item_data (id, item_id, setter_id, data_type, idx) VALUES (?, ?, ?, ?, 0)",
                         (did, i, setter, "clip" if setter == 1 else "text-embedding"))
            vec = rows[did - 1]
            blob = vec.astype("<f4").tobytes()
            if ragged and did == 7:
                blob = blob[:-4]  # a stray row of another length: skipped by the loaders
            conn.execute("INSERT INTO embeddings (id, embedding) VALUES (?, ?)", (did, blob))
            meta.append((did, i, setter, vec, len(blob) == dim * 4))
    good = [m for m in meta if m[4]]
    scale = orc.compute_int8_scale(np.stack([m[3] for m in good]))
    art = struct.pack("<f", scale)
    conn.execute("INSERT INTO vector_quant_profiles (id, name, quantizer, state, is_default) VALUES (5, 'int8', 'int8', 'active', 1)")
    for setter in (1, 2):
        conn.execute("INSERT INTO vector_quant_coverage (profile_id, setter_id, artifact, artifact_rev, dim, state) VALUES (5, ?, ?, 2, ?, 'ready')",
                     (setter, art, dim))
    codes = orc.quantize_int8(np.stack([m[3] for m in good]), scale)
    for m, c in zip(good, codes):
        conn.execute("INSERT INTO embedding_quants (id, profile_id, rev, quant) VALUES (?, 5, 2, ?)", (m[0], c.tobytes()))
    conn.execute("INSERT INTO embedding_quants (id, profile_id, rev, quant) VALUES (?, 5, 1, ?)", (10_000, codes[0].tobytes()))  # stale rev, no item_data
    conn.commit()
    return conn, good, scale, codes


def test_resolve_ready_pair_matrix():
    """resolve_ready_pair (db/vector_quants.rs:1795-1869) through both host forms: the Python probes of loader.py and the SQL
    function pvs_ready_pair of the C extension must agree on every case (no GPU involved)."""
    import json

    from panoptikon_amd import loader as py_loader, sqlite_seam

    conn, good, scale, _ = build_db()
    sqlite_seam.load(conn)

    class loader:  # every probe below runs twice
        default_profile_name = staticmethod(py_loader.default_profile_name)
        active_profile_id = staticmethod(py_loader.active_profile_id)

        @staticmethod
        def resolve_ready_pair(c, profile, setter_names):
            a = py_loader.resolve_ready_pair(c, profile, setter_names)
            marks = ", ".join("?" * (1 + len(setter_names)))
            row = c.execute(f"SELECT pvs_ready_pair({marks})", [profile, *setter_names]).fetchone()[0]
            if a is None:
                assert row is None, (profile, setter_names, row)
            else:
                b = json.loads(row)
                assert b["profile_id"] == a.profile_id and b["dim"] == a.dim and np.float32(b["scale"]) == np.float32(a.scale), (a, row)
            return a

    names = ["clip/m", "tclip/m"]
    p = loader.resolve_ready_pair(conn, "int8", names)
    assert p is not None and p.profile_id == 5 and p.dim == 96 and np.float32(p.scale) == np.float32(scale)
    assert loader.default_profile_name(conn) == "int8" and loader.active_profile_id(conn, "int8") == 5
    # a setter name without a setters row is skipped; only unknown names -> None (nothing to query)
    assert loader.resolve_ready_pair(conn, "int8", ["clip/m", "no-such-setter"]) is not None
    assert loader.resolve_ready_pair(conn, "int8", ["no-such-setter"]) is None
    # existing setter without a ready pair -> None
    assert loader.resolve_ready_pair(conn, "int8", ["clip/m", "other"]) is None
    assert loader.resolve_ready_pair(conn, "nope", names) is None
    # sibling with a different artifact -> rebuild pending -> None
    conn.execute("UPDATE vector_quant_coverage SET artifact = ? WHERE setter_id = 2", (struct.pack("<f", scale * 2),))
    assert loader.resolve_ready_pair(conn, "int8", names) is None
    assert loader.resolve_ready_pair(conn, "int8", ["clip/m"]) is not None
    # unusable artifacts (artifact_scale rejections, db/vector_quants.rs:1456-1460) and missing dim
    for bad in (b"", b"\x00\x00\x00", struct.pack("<f", 0.0), struct.pack("<f", -1.0), struct.pack("<f", float("nan")), struct.pack("<f", float("inf")), None):
        conn.execute("UPDATE vector_quant_coverage SET artifact = ? WHERE setter_id = 1", (bad,))
        assert loader.resolve_ready_pair(conn, "int8", ["clip/m"]) is None, bad
    conn.execute("UPDATE vector_quant_coverage SET artifact = ?, dim = NULL WHERE setter_id = 1", (struct.pack("<f", scale),))
    assert loader.resolve_ready_pair(conn, "int8", ["clip/m"]) is None
    conn.execute("UPDATE vector_quant_coverage SET dim = 96, state = 'building' WHERE setter_id = 1")
    assert loader.resolve_ready_pair(conn, "int8", ["clip/m"]) is None
    conn.execute("UPDATE vector_quant_coverage SET state = 'ready' WHERE setter_id = 1")
    conn.execute("UPDATE vector_quant_profiles SET state = 'removing'")
    assert loader.resolve_ready_pair(conn, "int8", ["clip/m"]) is None and loader.default_profile_name(conn) is None


def test_row_streams_are_in_item_data_id_order_and_skip_ragged_rows():
    from panoptikon_amd import loader

    conn, good, scale, codes = build_db()
    ids, items, mats = zip(*loader.iter_exact_rows(conn, ["clip/m", "tclip/m"], chunk_rows=64))
    ids, items, mat = np.concatenate(ids), np.concatenate(items), np.concatenate(mats)
    assert np.all(np.diff(ids) > 0) and 7 not in ids
    assert ids.tolist() == [m[0] for m in good] and items.tolist() == [m[1] for m in good]
    assert np.array_equal(mat.view(np.uint32), np.stack([m[3] for m in good]).view(np.uint32))
    only_img = np.concatenate([c[0] for c in loader.iter_exact_rows(conn, ["clip/m"])])
    assert only_img.tolist() == [m[0] for m in good if m[2] == 1]
    assert list(loader.iter_exact_rows(conn, ["no-such-setter"])) == []
    qi, qit, qm = zip(*loader.iter_quant_rows(conn, 5, ["clip/m", "tclip/m"], 96, chunk_rows=50))
    assert np.concatenate(qi).tolist() == [m[0] for m in good] and np.array_equal(np.concatenate(qm), codes)
    # codes of another revision are not served
    conn.execute("UPDATE vector_quant_coverage SET artifact_rev = 3 WHERE setter_id = 2")
    qi2 = np.concatenate([c[0] for c in loader.iter_quant_rows(conn, 5, ["clip/m", "tclip/m"], 96)])
    assert qi2.tolist() == [m[0] for m in good if m[2] == 1]
    assert loader.coverage_revs(conn, 5, ["clip/m", "tclip/m", "zzz"]) == ((1, 2), (2, 3))


@pytest.mark.gpu
def test_loaded_indexes_answer_like_the_oracle_and_follow_the_epoch():
    import panoptikon_amd as pvs
    from panoptikon_amd import loader

    conn, good, scale, codes = build_db(n_items=900, dim=128, ragged=True)
    names = ["clip/m", "tclip/m"]
    rows = np.stack([m[3] for m in good])
    ids = np.array([m[0] for m in good], np.int64)
    q = orc.synth_rows(77, 0, 4, 128)
    cache = loader.IndexCache()
    ex = cache.get(conn, "idx", 0, names)  # exact mode: f32 rows
    assert ex.kind == "exact" and ex.rows == len(good) and ex.dim == 128
    gi, gd, gc = ex.index.search(q, 10, pvs.L2)  # text filters are always L2 (text_embeddings.rs:386-393)
    ei, ed = orc.search(orc.F32, orc.L2, rows, q, 10, ids=ids)
    assert np.array_equal(gi, ei) and np.array_equal(gd.view(np.uint32), ed.view(np.uint32))
    qu = cache.get(conn, "idx", 0, names, profile_name="int8")
    assert qu.kind == "quant" and qu.rows == len(good) and np.float32(qu.scale) == np.float32(scale)
    gi, gd, gc = qu.index.search(q, 10, pvs.COSINE)  # f32 query quantized with the frozen scale on the device
    ei, ed = orc.search(orc.I8, orc.COSINE, codes, orc.quantize_int8(q, scale), 10, ids=ids)
    assert np.array_equal(gi, ei) and np.array_equal(gd.view(np.uint32), ed.view(np.uint32))
    # per-item aggregation over the loaded group ids (item_id)
    gg, gv, gn = qu.index.search_groups(q[:1], 5, pvs.COSINE, pvs.AGG_MIN)
    eg, ev = orc.search_groups(orc.I8, orc.COSINE, codes, orc.quantize_int8(q, scale)[0], np.array([m[1] for m in good], np.int64), orc.AGG_MIN, 5)
    assert np.array_equal(gg[0, : gn[0]], eg) and np.array_equal(gv[0, : gn[0]].view(np.uint64), ev.view(np.uint64))
    # same epoch -> same object; bumped epoch (an index-DB write happened) -> rebuilt, sees the new row
    assert cache.get(conn, "idx", 0, names) is ex
    conn.execute("INSERT INTO item_data (id, item_id, setter_id, data_type, idx) VALUES (99999, 1, 1, 'clip', 1)")
    conn.execute("INSERT INTO embeddings (id, embedding) VALUES (99999, ?)", (q[0].astype('<f4').tobytes(),))
    ex2 = cache.get(conn, "idx", 1, names)
    assert ex2 is ex and ex2.rows == len(good) + 1, "an intact prefix is appended to, not rebuilt"
    gi, gd, gc = ex2.index.search(q[:1], 1, pvs.L2)
    assert gi[0, 0] == 99999 and gd[0, 0] == 0.0
    # a row the index holds disappears (ON DELETE CASCADE of an item): exactly that row leaves the device index (round 5:
    # pvs_index_remove_rows; until then: a rebuild from SQLite)
    victim = int(ids[5])
    conn.execute("DELETE FROM embeddings WHERE id = ?", (victim,))
    conn.execute("DELETE FROM item_data WHERE id = ?", (victim,))
    ex3 = cache.get(conn, "idx", 2, names)
    assert ex3 is ex and ex3.rows == len(good), "a deletion is applied in place"
    gi, gd, gc = ex3.index.search(rows[5:6], 3, pvs.L2)
    assert victim not in gi[0]
    keep = ids != victim
    ei, ed = orc.search(orc.F32, orc.L2, np.concatenate([rows[keep], q[:1]]), q, 10, ids=np.concatenate([ids[keep], [99999]]))
    gi, gd, gc = ex3.index.search(q, 10, pvs.L2)
    assert np.array_equal(gi, ei) and np.array_equal(gd.view(np.uint32), ed.view(np.uint32))
    assert cache.get(conn, "idx", 2, names) is ex3 and loader._prefix_intact(conn, ex3, names)
    # the quant index follows: one more code row at the current revision is appended
    conn.execute("INSERT INTO embedding_quants (id, profile_id, rev, quant) VALUES (99999, 5, 2, ?)",
                 (orc.quantize_int8(q[:1], scale)[0].tobytes(),))
    conn.execute("DELETE FROM embedding_quants WHERE id = ?", (victim,))
    qu2 = cache.get(conn, "idx", 2, names, profile_name="int8")
    assert qu2 is qu and qu2.rows == len(good)  # victim gone (removed in place), new row present (appended)
    gi, gd, gc = qu2.index.search(q, 10, pvs.COSINE)
    ei, ed = orc.search(orc.I8, orc.COSINE, np.concatenate([codes[keep], orc.quantize_int8(q[:1], scale)]), orc.quantize_int8(q, scale), 10,
                        ids=np.concatenate([ids[keep], [99999]]))
    assert np.array_equal(gi, ei) and np.array_equal(gd.view(np.uint32), ed.view(np.uint32))
    conn.execute("INSERT INTO item_data (id, item_id, setter_id, data_type, idx) VALUES (100001, 2, 1, 'clip', 2)")
    conn.execute("INSERT INTO embeddings (id, embedding) VALUES (100001, ?)", (q[1].astype('<f4').tobytes(),))
    conn.execute("INSERT INTO embedding_quants (id, profile_id, rev, quant) VALUES (100001, 5, 2, ?)",
                 (orc.quantize_int8(q[1:2], scale)[0].tobytes(),))
    qu3 = cache.get(conn, "idx", 3, names, profile_name="int8")
    assert qu3 is qu2 and qu3.rows == len(good) + 1
    gi, gd, gc = qu3.index.search(q[1:2], 1, pvs.COSINE)
    assert gi[0, 0] == 100001
    # not-ready pair -> None (caller falls back to exact / raises under strict selection)
    conn.execute("UPDATE vector_quant_coverage SET state = 'building' WHERE setter_id = 2")
    assert cache.get(conn, "idx", 4, names, profile_name="int8") is None
    cache.clear()


def codes_now(conn, good):
    """the int8 codes currently stored for the good rows, in id order"""
    rows = dict(conn.execute("SELECT id, quant FROM embedding_quants WHERE profile_id = 5 AND rev = 2"))
    return np.stack([np.frombuffer(rows[m[0]], np.int8) for m in good])


def test_sqlite_extension_loads_and_registers_without_a_gpu():
    """libpvs_sqlite.so through SQLite's loadable-extension ABI (the reference's seam, db/sql_functions.rs:83-128): the module and
    the scalar functions register on a stdlib connection; with no index bound every call is a clean SQL error."""
    import re
    import subprocess

    from panoptikon_amd import sqlite_seam

    header = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "pvs_sqlite.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b((?:pvs_sqlite|sqlite3_pvs|sqlite3_extension)[a-z0-9_]*)\s*\(", header))
    out = subprocess.check_output(["nm", "-D", "--defined-only", sqlite_seam.EXT_PATH], text=True)
    exported = set(re.findall(r"\bT ([a-z0-9_]+)", out))
    assert declared and declared <= exported, sorted(declared - exported)
    deps = subprocess.check_output(["ldd", sqlite_seam.EXT_PATH], text=True)
    assert "libpvs.so" in deps and "libsqlite3" not in deps and "oracle" not in deps
    conn = sqlite3.connect(":memory:")
    sqlite_seam.load(conn)
    assert [r[0] for r in conn.execute("SELECT name FROM pragma_module_list WHERE name = 'pvs_dist'")] == ["pvs_dist"]
    assert {r[0] for r in conn.execute("SELECT name FROM pragma_function_list WHERE name LIKE 'pvs_distance%'")} == {"pvs_distance_cosine", "pvs_distance_l2"}
    assert {r[0] for r in conn.execute("SELECT name FROM pragma_function_list WHERE name LIKE 'pvs_load%'")} == {"pvs_load", "pvs_load_info"}
    assert {r[0] for r in conn.execute("SELECT name FROM pragma_function_list WHERE name LIKE 'pvs_backfill%'")} == {"pvs_backfill", "pvs_backfill_cursor", "pvs_backfill_phases"}
    assert conn.execute("SELECT pvs_load_info('nope')").fetchone()[0] is None
    assert conn.execute("SELECT pvs_backfill_cursor()").fetchone()[0] is None
    # backfill: errors that need no device (the select is read, and refused, before anything is quantized)
    for sql in ("SELECT pvs_backfill('SELECT 1, 2, 3, 4', 'SELECT 1', 5, 0)",                       # embedding / artifact are not blobs
                "SELECT pvs_backfill('SELECT 1, zeroblob(6), zeroblob(4), 2', 'SELECT 1', 5, 0)",   # not a whole number of floats
                "SELECT pvs_backfill('SELECT 1, zeroblob(8), zeroblob(3), 2', 'SELECT 1', 5, 0)",   # artifact is not a scale
                "SELECT pvs_backfill('SELECT nope', 'SELECT 1', 5, 0)", "SELECT pvs_backfill('SELECT 1')"):
        with pytest.raises(sqlite3.OperationalError):
            conn.execute(sql).fetchall()
    assert conn.execute("SELECT pvs_backfill('SELECT 1, zeroblob(8), zeroblob(4), 2 WHERE 0', 'SELECT 1', 5, 0)").fetchone()[0] == 0  # nothing to do
    for sql, args in (("SELECT * FROM pvs_dist('nope', ?)", (b"\0" * 16,)), ("SELECT pvs_distance_l2('nope', 1, ?)", (b"\0" * 16,)), ("SELECT * FROM pvs_dist('x')", ()),
                      ("SELECT pvs_load('nope', 'SELECT 1, 1, zeroblob(4)')", ()), ("SELECT pvs_load('nope')", ())):
        with pytest.raises(sqlite3.OperationalError):
            conn.execute(sql, args).fetchall()


_PAPI_SCRIPT = r"""
import ctypes as C, sqlite3
from panoptikon_amd import sqlite_seam
NAMES = ["create_function_v2", "create_module_v2", "declare_vtab", "value_type", "value_bytes", "value_blob", "value_text", "value_int64",
         "result_double", "result_int64", "result_null", "result_error", "user_data", "get_auxdata", "set_auxdata", "mprintf", "free",
         "prepare_v2", "step", "finalize", "column_type", "column_blob", "column_bytes", "column_int64", "bind_value", "context_db_handle",
         "errmsg", "result_text", "bind_blob", "bind_int64", "reset"]
class Api(C.Structure):
    _fields_ = [("struct_size", C.c_uint32)] + [(n, C.c_void_p) for n in NAMES]
ext = sqlite_seam.ext()
ext.pvs_sqlite_api_snapshot.restype = C.c_int32
ext.pvs_sqlite_api_snapshot.argtypes = [C.c_void_p]
a = Api(); a.struct_size = C.sizeof(Api)
assert ext.pvs_sqlite_api_snapshot(C.byref(a)) == 1, "nothing installed before an entry point ran"
conn = sqlite3.connect(":memory:")
sqlite_seam.load(conn)   # sqlite3_load_extension -> sqlite3_extension_init(db, &err, pApi): SQLite's own table
a = Api(); a.struct_size = C.sizeof(Api)
assert ext.pvs_sqlite_api_snapshot(C.byref(a)) == 0 and a.struct_size == C.sizeof(Api)
s3 = C.CDLL("libsqlite3.so.0")  # the SQLite the stdlib module runs on
bad = [n for n in NAMES if getattr(a, n) != C.cast(getattr(s3, "sqlite3_" + n), C.c_void_p).value]
print("mismatched slots:", bad)
"""


def test_extension_entry_point_takes_sqlite_from_the_api_routines_table():
    """sqlite3_auto_extension(sqlite3_pvs_init) in a host with a statically linked SQLite (the reference: libsqlite3-sys bundled,
    db/sql_functions.rs:105-128) only works when the entry point uses the sqlite3_api_routines table SQLite passes it.  Every slot
    index the extension reads (csrc/pvs_sqlite.cpp: ApiSlot) is checked here against the real library: after the stdlib module
    loaded the extension, the installed table must hold exactly the addresses of libsqlite3's own functions.  (Own process: the
    table is process-wide, and nothing may have installed one before.)"""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.check_output([sys.executable, "-c", _PAPI_SCRIPT], cwd=root, text=True, env=dict(os.environ, PYTHONPATH=root))
    assert out.strip().splitlines()[-1] == "mismatched slots: []", out


class _StubIndex:
    """Stands in for VectorIndex in the CPU tests of the cache logic: records what was appended."""

    def __init__(self, *a, **k):
        self.ids = []

    def set_scale(self, s):
        pass

    def add_f32(self, mat, row_ids=None, group_ids=None):
        # the library's rule (check_ids, csrc/pvs_api.hip): strictly increasing, ascending from the LAST SURVIVING id — ids that
        # left with a removal may come back (round 6; ADVICE r5: the stub used to accept anything, which hid that the real
        # library refused re-added rowids after a tail removal)
        from panoptikon_amd._lib import PvsError

        new = list(map(int, row_ids))
        prev = self.ids[-1] if self.ids else None
        for r in new:
            if prev is not None and r <= prev:
                raise PvsError(1, f"row_ids must be strictly increasing ({r} after {prev})")
            prev = r
        self.ids += new
        self.grp = getattr(self, "grp", []) + list(map(int, group_ids))

    add = add_f32

    def read_ids(self, row0=0, n=None, groups=False):
        ids, grp = np.array(self.ids, np.int64), np.array(self.grp, np.int64)
        return (ids, grp) if groups else ids

    def remove_rows(self, row_ids):
        gone = set(map(int, row_ids))
        keep = [i for i, r in enumerate(self.ids) if r not in gone]
        removed = len(self.ids) - len(keep)
        self.ids, self.grp = [self.ids[i] for i in keep], [self.grp[i] for i in keep]
        return removed

    def close(self):
        pass


def test_reused_rowids_force_a_rebuild(monkeypatch):
    """item_data.id is INTEGER PRIMARY KEY without AUTOINCREMENT (init.sql:94): deleting the newest extraction and
    inserting another one reuses its id.  COUNT(prefix) is then unchanged; the loader must still notice."""
    from panoptikon_amd import index as pvs_index
    from panoptikon_amd import loader

    monkeypatch.setattr(pvs_index, "VectorIndex", _StubIndex)
    conn, good, scale, codes = build_db(ragged=False)
    names = ["clip/m", "tclip/m"]
    for kind in ("exact", "quant"):
        li = loader.load_exact_index(conn, names) if kind == "exact" else loader.load_quant_index(conn, "int8", names)
        assert li.rows == len(good) and li.last_id == good[-1][0] and len(li.tail) == min(len(good), loader.TAIL_WINDOW)
        assert loader.append_new_rows(conn, li, names) == 0  # untouched database: nothing to append, prefix intact
        conn.execute("SAVEPOINT t")
        last_id, last_item, last_setter = good[-1][0], good[-1][1], good[-1][2]
        conn.execute("DELETE FROM embeddings WHERE id = ?", (last_id,))
        conn.execute("DELETE FROM embedding_quants WHERE id = ?", (last_id,))
        conn.execute("DELETE FROM item_data WHERE id = ?", (last_id,))
        # a new extraction of the SAME item: SQLite hands out the same id again, item_id is the same too
        cur = conn.execute("INSERT INTO item_data (item_id, setter_id, data_type, idx) VALUES (?, ?, 'clip', 0)", (last_item, last_setter))
        assert cur.lastrowid == last_id
        vec = -good[-1][3]
        conn.execute("INSERT INTO embeddings (id, embedding) VALUES (?, ?)", (last_id, vec.astype("<f4").tobytes()))
        conn.execute("INSERT INTO embedding_quants (id, profile_id, rev, quant) VALUES (?, 5, 2, ?)",
                     (last_id, orc.quantize_int8(vec[None, :], scale)[0].tobytes()))
        assert loader.append_new_rows(conn, li, names) is None, f"{kind}: same count, same ids, new payload -> rebuild"
        conn.execute("ROLLBACK TO t")
        # a different item takes the id: caught by the integer sums even outside the tail window
        conn.execute("DELETE FROM embeddings WHERE id = ?", (last_id,))
        conn.execute("DELETE FROM embedding_quants WHERE id = ?", (last_id,))
        conn.execute("DELETE FROM item_data WHERE id = ?", (last_id,))
        other_item = last_item - 1
        conn.execute("INSERT INTO item_data (id, item_id, setter_id, data_type, idx) VALUES (?, ?, ?, 'clip', 7)", (last_id, other_item, last_setter))
        conn.execute("INSERT INTO embeddings (id, embedding) VALUES (?, ?)", (last_id, good[-1][3].astype("<f4").tobytes()))
        conn.execute("INSERT INTO embedding_quants (id, profile_id, rev, quant) VALUES (?, 5, 2, ?)", (last_id, codes[-1].tobytes()))
        monkeypatch.setattr(loader, "TAIL_WINDOW", 0)
        li0 = type(li)(li.index, li.kind, li.rows, li.dim, profile_id=li.profile_id, scale=li.scale, last_id=li.last_id,
                       sum_id=li.sum_id, sum_item=li.sum_item, tail=[])
        assert loader.append_new_rows(conn, li0, names) is None, f"{kind}: another item under a reused id"
        monkeypatch.undo()
        monkeypatch.setattr(pvs_index, "VectorIndex", _StubIndex)
        conn.execute("ROLLBACK TO t")
        conn.execute("RELEASE t")


def test_deletions_are_reconciled_in_place_and_reused_ids_come_back_as_new_rows(monkeypatch):
    """loader.reconcile_deletions (round 5): rows deleted from the database leave the loaded index one by one; an id that was deleted
    and handed out again (item_data.id is not AUTOINCREMENT) sits ABOVE the anchor — the newest loaded row that is still there
    unchanged — so its old row is dropped and append_new_rows brings the new content in; no anchor in the tail window: rebuild."""
    from panoptikon_amd import index as pvs_index

    monkeypatch.setattr(pvs_index, "VectorIndex", _StubIndex)
    _reconcile_scenario(lambda li: list(li.index.ids))


@pytest.mark.gpu
def test_deletions_are_reconciled_in_place_with_the_real_library():
    """The same scenario against libpvs itself (the stub accepted re-added rowids the library refused — the index
    now ascends from its last SURVIVING id, csrc/pvs_lifecycle.hip)."""
    import panoptikon_amd as pvs

    if pvs.device_count() < 1:
        pytest.fail("no gfx950 device visible: the gpu tests need an MI355X")
    _reconcile_scenario(lambda li: [int(x) for x in li.index.read_ids(0, int(li.index.stats().rows))])


def _reconcile_scenario(ids_of):
    from panoptikon_amd import loader

    conn, good, scale, codes = build_db(ragged=False)
    names = ["clip/m", "tclip/m"]
    for kind in ("exact", "quant"):
        li = loader.load_exact_index(conn, names) if kind == "exact" else loader.load_quant_index(conn, "int8", names)
        all_ids = [m[0] for m in good]
        conn.execute("SAVEPOINT t")
        # three scattered rows and the newest row go; the newest id is handed out again with another payload
        victims = [all_ids[3], all_ids[40], all_ids[41], all_ids[-1]]
        for v in victims:
            conn.execute("DELETE FROM embeddings WHERE id = ?", (v,))
            conn.execute("DELETE FROM embedding_quants WHERE id = ?", (v,))
            conn.execute("DELETE FROM item_data WHERE id = ?", (v,))
        cur = conn.execute("INSERT INTO item_data (item_id, setter_id, data_type, idx) VALUES (?, ?, 'clip', 0)", (good[-1][1], good[-1][2]))
        assert cur.lastrowid == all_ids[-1]
        vec = -good[-1][3]
        conn.execute("INSERT INTO embeddings (id, embedding) VALUES (?, ?)", (all_ids[-1], vec.astype("<f4").tobytes()))
        conn.execute("INSERT INTO embedding_quants (id, profile_id, rev, quant) VALUES (?, 5, 2, ?)",
                     (all_ids[-1], orc.quantize_int8(vec[None, :], scale)[0].tobytes()))
        assert loader.append_new_rows(conn, li, names) is None
        assert loader.reconcile_deletions(conn, li, names) == 4, kind  # (the reused id's OLD row goes too)
        assert ids_of(li) == [i for i in all_ids if i not in victims] and li.last_id == all_ids[-2]
        assert loader.append_new_rows(conn, li, names) == 1 and ids_of(li)[-1] == all_ids[-1]
        assert loader._prefix_intact(conn, li, names) and li.rows == len(good) - 3
        # every row of the tail window gone: no anchor, the caller rebuilds
        for v in [t[0] for t in li.tail]:
            conn.execute("DELETE FROM embeddings WHERE id = ?", (v,))
            conn.execute("DELETE FROM embedding_quants WHERE id = ?", (v,))
            conn.execute("DELETE FROM item_data WHERE id = ?", (v,))
        assert loader.reconcile_deletions(conn, li, names) is None
        conn.execute("ROLLBACK TO t")
        conn.execute("RELEASE t")


@pytest.mark.gpu
def test_dist_cte_seam_sql_runs_unchanged_over_device_distances():
    """The SQL shape the reference generates around the distance column — MATERIALIZED dist CTE, GROUP BY file_id
    with MIN, row_number() rank, ORDER BY … NULLS LAST, LIMIT (filters/exact.rs:106-165, builder.rs:757-771,
    578-582) — run twice on a miniature index DB: (A) with sqlite-vec's scalar functions emulated IN THE TEST by
    the CPU oracle, as the reference runs it; (B) with the distance column read from a temp table filled by
    pvs_score_all.  Same pages, same f64 values."""
    import panoptikon_amd as pvs
    from panoptikon_amd import loader, sqlite_seam

    conn, good, scale, codes = build_db(n_items=700, dim=128, ragged=False)
    conn.execute("CREATE TABLE files (id INTEGER PRIMARY KEY, item_id INTEGER NOT NULL, last_modified TEXT NOT NULL)")
    fid = 0
    for item in range(1, 701):  # one or two files per item
        for rep in range(1 + item % 2):
            fid += 1
            conn.execute("INSERT INTO files (id, item_id, last_modified) VALUES (?, ?, ?)", (fid, item, f"2024-01-{1 + fid % 28:02d}"))
    # one all-zero code row: its cosine distance is NULL in SQL
    zero_id = good[10][0]
    conn.execute("UPDATE embedding_quants SET quant = ? WHERE id = ? AND rev = 2", (bytes(128), zero_id))
    q = orc.quantize_int8(orc.synth_rows(4242, 0, 1, 128), scale)[0]

    conn.create_function("vec_int8", 1, lambda b: b, deterministic=True)

    def vdc(a, b):
        v = orc.vec_distance(orc.COSINE, np.frombuffer(a, np.int8), np.frombuffer(b, np.int8))
        return None if v != v else float(v)

    conn.create_function("vec_distance_cosine", 2, vdc, deterministic=True)
    tail = """, agg AS (SELECT file_id, MIN(d) AS order_rank FROM dist GROUP BY file_id),
              ranked AS (SELECT file_id, order_rank, row_number() OVER (ORDER BY order_rank ASC NULLS LAST, file_id) AS rn FROM agg)
         SELECT r.file_id, r.order_rank, r.rn FROM ranked r JOIN files f ON f.id = r.file_id
         ORDER BY r.order_rank ASC NULLS LAST, f.last_modified DESC, r.file_id LIMIT 40 OFFSET 20"""
    sql_ref = """WITH dist AS MATERIALIZED (
             SELECT d.item_id AS item_id, f.id AS file_id, vec_distance_cosine(vec_int8(qq.quant), vec_int8(?)) AS d
             FROM item_data d JOIN setters s ON s.id = d.setter_id
             JOIN embedding_quants qq ON qq.id = d.id AND qq.profile_id = 5 AND qq.rev = 2
             JOIN files f ON f.item_id = d.item_id
             WHERE s.name IN ('clip/m', 'tclip/m'))""" + tail
    # (B) the device column through the table-valued function registered by libpvs_sqlite.so (the SQLite extension ABI,
    # the reference's own seam: db/sql_functions.rs:83-128); (C) the scalar drop-in in the place of vec_distance_cosine
    sql_tvf = """WITH dist AS MATERIALIZED (
             SELECT d.item_id AS item_id, f.id AS file_id, p.d AS d
             FROM item_data d JOIN setters s ON s.id = d.setter_id
             JOIN pvs_dist('clip-int8', ?, 'cosine') p ON p.id = d.id
             JOIN files f ON f.item_id = d.item_id
             WHERE s.name IN ('clip/m', 'tclip/m'))""" + tail
    sql_udf = """WITH dist AS MATERIALIZED (
             SELECT d.item_id AS item_id, f.id AS file_id, pvs_distance_cosine('clip-int8', qq.id, ?) AS d
             FROM item_data d JOIN setters s ON s.id = d.setter_id
             JOIN embedding_quants qq ON qq.id = d.id AND qq.profile_id = 5 AND qq.rev = 2
             JOIN files f ON f.item_id = d.item_id
             WHERE s.name IN ('clip/m', 'tclip/m'))""" + tail
    expected = conn.execute(sql_ref, (q.tobytes(),)).fetchall()
    li = loader.load_quant_index(conn, "int8", ["clip/m", "tclip/m"])
    assert li.rows == len(good)
    sqlite_seam.load(conn)
    sqlite_seam.bind("clip-int8", li.index)
    assert conn.execute("SELECT COUNT(*), SUM(d IS NULL) FROM pvs_dist('clip-int8', ?)", (q.tobytes(),)).fetchone() == (len(good), 1)
    got = conn.execute(sql_tvf, (q.tobytes(),)).fetchall()
    assert len(got) == 40 and got == expected
    assert conn.execute(sql_udf, (q.tobytes(),)).fetchall() == expected
    # an f32 query against the int8 index is quantized with the frozen scale on the device (compute_query_quant)
    qf = orc.synth_rows(4242, 0, 1, 128)[0]
    assert conn.execute(sql_tvf, (qf.astype("<f4").tobytes(),)).fetchall() == expected
    # k given: page 1 of the row ranking from the filter scan = ORDER BY d, id LIMIT k over the whole column
    top = conn.execute("SELECT id, d FROM pvs_dist('clip-int8', ?, 'cosine', 25)", (q.tobytes(),)).fetchall()
    full = conn.execute("SELECT id, d FROM pvs_dist('clip-int8', ?, 'cosine') WHERE d IS NOT NULL ORDER BY d, id LIMIT 25", (q.tobytes(),)).fetchall()
    assert top == full
    l2 = conn.execute("SELECT id, d FROM pvs_dist('clip-int8', ?, 'l2', 5)", (q.tobytes(),)).fetchall()
    ei, ed = orc.search(orc.I8, orc.L2, codes_now(conn, good), q[None, :], 5, ids=np.array([m[0] for m in good], np.int64))
    assert [r[0] for r in l2] == ei[0].tolist() and [np.float32(r[1]) for r in l2] == ed[0].tolist()
    # errors are SQL errors, like sqlite-vec's (db/pql.rs:18-21)
    for bad_sql, args in (("SELECT * FROM pvs_dist('clip-int8', ?)", (q.tobytes()[:-1],)), ("SELECT * FROM pvs_dist('nope', ?)", (q.tobytes(),)),
                          ("SELECT * FROM pvs_dist('clip-int8', ?, 'dot')", (q.tobytes(),)), ("SELECT * FROM pvs_dist('clip-int8', ?, 'l2', 0)", (q.tobytes(),))):
        with pytest.raises(sqlite3.OperationalError):
            conn.execute(bad_sql, args).fetchall()
    # and without SQL at all: the same first page from pvs_search_groups (items) expanded to files on the host
    gg, gv, gc = li.index.search_groups(q[None, :], 10, pvs.COSINE, pvs.AGG_MIN)
    first_items = [r[0] for r in conn.execute(
        "SELECT d.item_id, MIN(p.d) AS m FROM item_data d JOIN pvs_dist('clip-int8', ?) p ON p.id = d.id GROUP BY d.item_id ORDER BY m ASC NULLS LAST, d.item_id LIMIT 10",
        (q.tobytes(),))]
    assert gg[0, : gc[0]].tolist() == first_items
    sqlite_seam.unbind("clip-int8")
    li.index.close()


@pytest.mark.gpu
def test_native_row_streamer_loads_what_the_python_loader_loads():
    """pvs_load (the C streamer of libpvs_sqlite.so) against the Python chunk loader on the same database: same rows, ids, groups,
    fingerprint and tail; ragged and NULL payloads skipped; appends after an epoch bump; errors are SQL errors."""
    import json

    import panoptikon_amd as pvs
    from panoptikon_amd import loader, sqlite_seam

    conn, good, scale, codes = build_db(n_items=700, dim=128, ragged=True)
    names = ["clip/m", "tclip/m"]
    for kind in ("exact", "quant"):
        if kind == "exact":
            py = loader.load_exact_index(conn, names)
            nat = loader.load_exact_index(conn, names, native=True)
        else:
            py = loader.load_quant_index(conn, "int8", names)
            nat = loader.load_quant_index(conn, "int8", names, native=True)
        assert nat.rows == py.rows == len(good) and nat.dim == py.dim
        assert (nat.last_id, nat.sum_id, nat.sum_item, nat.tail) == (py.last_id, py.sum_id, py.sum_item, py.tail)
        assert np.array_equal(nat.index.read_ids(), py.index.read_ids())
        assert np.array_equal(nat.index.read_rows(0, nat.rows).view(np.uint8), py.index.read_rows(0, py.rows).view(np.uint8))
        q = orc.synth_rows(5, 0, 1, 128)
        a, b = nat.index.search_groups(q, 7, pvs.L2, pvs.AGG_MIN), py.index.search_groups(q, 7, pvs.L2, pvs.AGG_MIN)
        assert all(np.array_equal(x, y) for x, y in zip(a, b)), "group ids travelled with the rows"
        # append after an epoch bump: only the new row is streamed, the fingerprint follows
        nid = 50_000 + (0 if kind == "exact" else 1)
        conn.execute("INSERT INTO item_data (id, item_id, setter_id, data_type, idx) VALUES (?, 3, 1, 'clip', 1)", (nid,))
        conn.execute("INSERT INTO embeddings (id, embedding) VALUES (?, ?)", (nid, q[0].astype("<f4").tobytes()))
        conn.execute("INSERT INTO embedding_quants (id, profile_id, rev, quant) VALUES (?, 5, 2, ?)", (nid, orc.quantize_int8(q, scale)[0].tobytes()))
        assert loader.append_new_rows(conn, nat, names, native=True) == 1 and loader.append_new_rows(conn, py, names) == 1
        assert (nat.rows, nat.last_id, nat.sum_id, nat.sum_item, nat.tail) == (py.rows, py.last_id, py.sum_id, py.sum_item, py.tail)
        assert np.array_equal(nat.index.read_ids(), py.index.read_ids())
        conn.execute("DELETE FROM embeddings WHERE id = ?", (nid,))
        conn.execute("DELETE FROM item_data WHERE id = ?", (nid,))
        conn.execute("DELETE FROM embedding_quants WHERE id = ?", (nid,))
        nat.index.close()
        py.index.close()
    # the SQL function itself: f32 payloads into an int8 index go through the device codec; bad input is an SQL error
    ix = pvs.VectorIndex(pvs.I8, 128)
    ix.set_scale(scale)
    sqlite_seam.load(conn)
    sqlite_seam.bind("t", ix)
    stream = "SELECT d.id, d.item_id, e.embedding FROM item_data d JOIN embeddings e ON e.id = d.id WHERE d.setter_id IN (?, ?) ORDER BY d.id"
    try:
        n = conn.execute("SELECT pvs_load('t', ?, 1, 2)", (stream,)).fetchone()[0]
        assert n == len(good)
        info = json.loads(conn.execute("SELECT pvs_load_info('t')").fetchone()[0])
        assert info["rows"] == len(good) and info["skipped"] == 1 and info["last_id"] == good[-1][0]
        assert np.array_equal(ix.read_rows(0, n), codes), "device quantization of the streamed f32 rows = the stored codes"
        with pytest.raises(sqlite3.OperationalError, match="pvs_load"):
            conn.execute("SELECT pvs_load('t', ?, 1, 2)", (stream,)).fetchall()  # ids not above the loaded ones
        with pytest.raises(sqlite3.OperationalError, match="pvs_load"):
            conn.execute("SELECT pvs_load('t', 'SELECT nonsense FROM nowhere')").fetchall()
        with pytest.raises(sqlite3.OperationalError, match="pvs_load"):
            conn.execute("SELECT pvs_load('t', 'SELECT 1, 1, zeroblob(4)', 5)").fetchall()  # more parameters than placeholders
    finally:
        sqlite_seam.unbind("t")
        ix.close()


_HOST_TABLE_SCRIPT = r"""
import ctypes as C, sys
from panoptikon_amd import sqlite_seam
s3 = C.CDLL("libsqlite3.so.0")
ext = sqlite_seam.ext()
V1 = ["create_function_v2", "create_module_v2", "declare_vtab", "value_type", "value_bytes", "value_blob", "value_text", "value_int64",
      "result_double", "result_int64", "result_null", "result_error", "user_data", "get_auxdata", "set_auxdata", "mprintf", "free"]
V2 = ["prepare_v2", "step", "finalize", "column_type", "column_blob", "column_bytes", "column_int64", "bind_value", "context_db_handle",
      "errmsg", "result_text"]
V3 = ["bind_blob", "bind_int64", "reset"]
class Api(C.Structure):
    _fields_ = [("struct_size", C.c_uint32)] + [(n, C.c_void_p) for n in V1 + V2 + V3]
def table(names, size):
    a = Api()
    for n in names:
        setattr(a, n, C.cast(getattr(s3, "sqlite3_" + n), C.c_void_p).value)
    a.struct_size = size
    return a
db = C.c_void_p()
assert s3.sqlite3_open(b":memory:", C.byref(db)) == 0
ext.pvs_sqlite_register.restype = C.c_int32
ext.pvs_sqlite_register.argtypes = [C.c_void_p, C.c_void_p]
mode = sys.argv[1]
api = table(V1, Api.prepare_v2.offset) if mode == "v1" else table(V1 + V2, Api.bind_blob.offset) if mode == "v2" else table(V1 + V2 + V3, C.sizeof(Api))
assert ext.pvs_sqlite_register(db, C.byref(api)) == 0
short = table(V1, 16)
assert ext.pvs_sqlite_register(db, C.byref(short)) != 0, "a struct shorter than v1 is refused"
def has(fn):
    st = C.c_void_p()
    rc = s3.sqlite3_prepare_v2(db, ("SELECT " + fn).encode(), -1, C.byref(st), None)
    s3.sqlite3_finalize(st)
    return rc == 0
print(int(has("pvs_distance_l2('x', 1, zeroblob(4))")), int(has("pvs_load('x', 'SELECT 1')")), int(has("pvs_load_info('x')")), int(has("pvs_backfill('a', 'b', 1, 0)")))
"""


@pytest.mark.parametrize("mode,expect", [("v1", "1 0 0 0"), ("v2", "1 1 1 0"), ("v3", "1 1 1 1")])
def test_host_supplied_sqlite_entry_points(mode, expect):
    """pvs_sqlite_register with the host's own table of SQLite entry points (a Rust host links its own SQLite): the full
    struct registers everything, the v2 struct everything but pvs_backfill, the v1 struct neither that nor the row streamer; no
    dlsym, no GPU.  (Own process:
    the table is process-wide.)"""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.check_output([sys.executable, "-c", _HOST_TABLE_SCRIPT, mode], cwd=root, text=True, env=dict(os.environ, PYTHONPATH=root))
    assert out.strip().splitlines()[-1] == expect


@pytest.mark.gpu
def test_row_streamer_c_entry_point_on_a_connection_the_host_holds():
    """pvs_sqlite_load(db, sql, idx, chunk_rows, &result): the C form of pvs_load for a host that owns the sqlite3* (here: a
    connection opened through libsqlite3's C API with ctypes).  Small chunks, NULL and ragged payloads, rows without groups."""
    import ctypes as C

    import panoptikon_amd as pvs
    from panoptikon_amd import sqlite_seam

    s3 = C.CDLL("libsqlite3.so.0")
    ext = sqlite_seam.ext()
    db = C.c_void_p()
    assert s3.sqlite3_open(b":memory:", C.byref(db)) == 0
    dim, n = 16, 57
    rows = orc.synth_rows(9, 0, n, dim)
    stmts = ["CREATE TABLE v (id INTEGER PRIMARY KEY, g INTEGER, e BLOB)"]
    for i in range(n):
        stmts.append(f"INSERT INTO v VALUES ({10 + 3 * i}, {i // 2}, x'{rows[i].astype('<f4').tobytes().hex()}')")
    stmts += ["INSERT INTO v VALUES (5, 0, NULL)", "INSERT INTO v VALUES (7, 0, x'00112233')"]  # skipped: NULL, wrong length
    for sql in stmts:
        assert s3.sqlite3_exec(db, sql.encode(), None, None, None) == 0, sql

    class Res(C.Structure):
        _fields_ = [("rows", C.c_uint64), ("skipped", C.c_uint64), ("last_id", C.c_int64), ("sum_id", C.c_uint64), ("sum_group", C.c_uint64)]

    ext.pvs_sqlite_load.restype = C.c_int32
    ext.pvs_sqlite_load.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint32, C.c_void_p]
    ix = pvs.VectorIndex(pvs.F32, dim)
    try:
        r = Res()
        assert ext.pvs_sqlite_load(db, b"SELECT id, g, e FROM v ORDER BY id", ix._h, 10, C.byref(r)) == 0
        ids = 10 + 3 * np.arange(n, dtype=np.int64)
        assert (r.rows, r.skipped, r.last_id) == (n, 2, int(ids[-1]))
        assert r.sum_id == int(ids.sum()) and r.sum_group == int((np.arange(n) // 2).sum())
        assert np.array_equal(ix.read_ids(), ids) and np.array_equal(ix.read_rows(0, n).view(np.uint32), rows.view(np.uint32))
        q = rows[20:21]
        gg, gv, gn = ix.search_groups(q, 1, pvs.L2, pvs.AGG_MIN)
        assert gg[0, 0] == 10 and gv[0, 0] == 0.0, "group ids arrived"
        # a statement that fails to prepare, and rows that do not increase: status codes, nothing appended
        assert ext.pvs_sqlite_load(db, b"SELECT nope FROM nowhere", ix._h, 0, None) != 0
        assert ext.pvs_sqlite_load(db, b"SELECT id, g, e FROM v ORDER BY id", ix._h, 0, C.byref(r)) != 0 and r.rows == 0
        assert ix.stats().rows == n
    finally:
        ix.close()
        s3.sqlite3_close(db)


# The reference's two statements of a backfill chunk, restated (db/vector_quants.rs:1085-1117): what is outstanding for a
# (profile, setter) pair that is `building`, past a cursor, in id order; and the upsert that keeps a row's rowid.
_BACKFILL_SELECT = """SELECT d.id, e.embedding, c.artifact, c.artifact_rev
    FROM vector_quant_coverage c JOIN item_data d ON d.setter_id = c.setter_id JOIN embeddings e ON e.id = d.id
    WHERE c.profile_id = ? AND c.setter_id = ? AND c.state = 'building' AND c.artifact IS NOT NULL AND d.id > ?
      AND length(e.embedding) = c.dim * 4
      AND NOT EXISTS (SELECT 1 FROM embedding_quants q WHERE q.id = e.id AND q.profile_id = c.profile_id AND q.rev = c.artifact_rev)
    ORDER BY d.id LIMIT ?"""
_BACKFILL_UPSERT = """INSERT INTO embedding_quants (id, profile_id, rev, quant) VALUES (?, ?, ?, ?)
    ON CONFLICT (id, profile_id) DO UPDATE SET rev = excluded.rev, quant = excluded.quant"""


@pytest.mark.gpu
def test_backfill_quantizes_on_the_device_and_upserts_like_backfill_chunk():
    """pvs_backfill = the reference's backfill_chunk (db/vector_quants.rs:1119-1163) with quantize_int8 on the device: same rows
    written, same codes (the oracle's, bit for bit), same cursor protocol, nothing written for a pair that is not `building` or
    whose artifact is not a scale; the rows keep their rowids (upsert, not replace)."""
    from panoptikon_amd import sqlite_seam

    conn, good, scale, codes = build_db(n_items=400, dim=96, ragged=True)
    sqlite_seam.load(conn)
    mine = [m for m in good if m[2] == 1]  # setter 1's vectors
    want = {m[0]: c.tobytes() for m, c in zip(good, codes) if m[2] == 1}
    rowids = dict(conn.execute("SELECT id, rowid FROM embedding_quants WHERE profile_id = 5"))
    # a rebuild of setter 1's pair: new revision, pair back to `building`; its rows are now all outstanding
    conn.execute("UPDATE vector_quant_coverage SET state = 'building', artifact_rev = 3 WHERE profile_id = 5 AND setter_id = 1")
    conn.execute("UPDATE embedding_quants SET quant = zeroblob(96) WHERE id IN (SELECT id FROM item_data WHERE setter_id = 1)")  # stale codes
    cursor, total, chunks = 0, 0, 0
    while True:
        n = conn.execute("SELECT pvs_backfill(?, ?, 5, 0, 5, 1, ?, 64)", (_BACKFILL_SELECT, _BACKFILL_UPSERT, cursor)).fetchone()[0]
        if n == 0:
            break
        cursor = conn.execute("SELECT pvs_backfill_cursor()").fetchone()[0]
        total += n
        chunks += 1
    assert total == len(mine) and chunks == -(-len(mine) // 64) and cursor == max(m[0] for m in mine)
    got = dict(conn.execute("SELECT id, quant FROM embedding_quants WHERE profile_id = 5 AND rev = 3"))
    assert got == want, "device codes differ from quantize_int8"
    assert dict(conn.execute("SELECT id, rowid FROM embedding_quants WHERE profile_id = 5")) == rowids, "the upsert must keep rowids"
    assert conn.execute("SELECT COUNT(*) FROM embedding_quants WHERE profile_id = 5 AND rev = 2").fetchone()[0] == len(good) - len(mine)  # setter 2 untouched
    assert conn.execute("SELECT COUNT(*) FROM embedding_quants WHERE id = 7").fetchone()[0] == 0  # the ragged vector is not quantized
    # complete: nothing outstanding from the start
    assert conn.execute("SELECT pvs_backfill(?, ?, 5, 0, 5, 1, 0, 64)", (_BACKFILL_SELECT, _BACKFILL_UPSERT)).fetchone()[0] == 0
    # a pair that is not building yields nothing; an artifact that is not a scale refuses before writing anything
    conn.execute("UPDATE vector_quant_coverage SET state = 'ready' WHERE profile_id = 5 AND setter_id = 1")
    conn.execute("UPDATE vector_quant_coverage SET artifact_rev = 4 WHERE profile_id = 5 AND setter_id = 1")
    assert conn.execute("SELECT pvs_backfill(?, ?, 5, 0, 5, 1, 0, 64)", (_BACKFILL_SELECT, _BACKFILL_UPSERT)).fetchone()[0] == 0
    conn.execute("UPDATE vector_quant_coverage SET state = 'building', artifact = x'0000' WHERE profile_id = 5 AND setter_id = 1")
    with pytest.raises(sqlite3.OperationalError, match="Invalid vector quant scale artifact"):
        conn.execute("SELECT pvs_backfill(?, ?, 5, 0, 5, 1, 0, 64)", (_BACKFILL_SELECT, _BACKFILL_UPSERT)).fetchall()
    assert conn.execute("SELECT COUNT(*) FROM embedding_quants WHERE rev = 4").fetchone()[0] == 0


@pytest.mark.gpu
def test_sql_seam_reads_a_column_longer_than_one_window():
    """pvs_dist / pvs_distance_* keep the `d` column in HBM and read it through a 262,144-row window: a column of 600k rows
    (three windows) must come out exactly as pvs_score_all returns it, in order and by random id lookups."""
    import panoptikon_amd as pvs
    from panoptikon_amd import sqlite_seam

    n, dim = 600_000, 32
    rng = np.random.default_rng(5)
    rows = rng.standard_normal((n, dim)).astype(np.float32)
    ids = (np.arange(n, dtype=np.int64) * 3 + 11)  # sparse, increasing
    ix = pvs.VectorIndex(pvs.F32, dim)
    ix.add_f32(rows, row_ids=ids)
    q = rng.standard_normal(dim).astype(np.float32)
    ref = ix.score_all(q, pvs.L2)
    conn = sqlite3.connect(":memory:")
    sqlite_seam.load(conn)
    sqlite_seam.bind("big", ix)
    try:
        got = conn.execute("SELECT id, d FROM pvs_dist('big', ?, 'l2')", (q.tobytes(),)).fetchall()
        assert len(got) == n and [g[0] for g in got[:5]] == ids[:5].tolist() and got[-1][0] == int(ids[-1])
        assert np.array_equal(np.array([g[1] for g in got], np.float32).view(np.uint32), ref.view(np.uint32))
        conn.execute("CREATE TABLE probe (id INTEGER PRIMARY KEY)")
        pick = rng.choice(n, 4000, replace=False)
        conn.executemany("INSERT INTO probe (id) VALUES (?)", [(int(ids[i]),) for i in pick] + [(5,), (int(ids[-1]) + 1,)])  # two ids the index lacks
        out = dict(conn.execute("SELECT id, pvs_distance_l2('big', id, ?) FROM probe", (q.tobytes(),)))
        assert out[5] is None and out[int(ids[-1]) + 1] is None
        assert all(np.float32(out[int(ids[i])]) == ref[i] for i in pick)
    finally:
        sqlite_seam.unbind("big")
        ix.close()


@pytest.mark.gpu
def test_row_page_with_order_keys_equals_sqlites_order_by_d_last_modified_desc():
    """The page pvs_search returns once the rows carry their `last_modified` as order keys must be the page SQLite itself
    produces from the `d` column with the reference's final ordering (`ORDER BY order_rank ASC NULLS LAST, last_modified DESC`,
    pql/model.rs:547-553) — on tie-heavy int8 L2 data, where the tie-break decides who is on the page at all."""
    import panoptikon_amd as pvs
    from panoptikon_amd import sqlite_seam

    rng = np.random.default_rng(21)
    dim, distinct, copies, k = 64, 50, 60, 75
    base = orc.synth_rows(31, 0, distinct, dim)
    rows = np.tile(base, (copies, 1))[rng.permutation(distinct * copies)]
    n = len(rows)
    ids = np.arange(1, n + 1, dtype=np.int64) * 3
    mtime = rng.integers(0, 25, n).astype(np.int64) + 1_700_000_000
    scale = orc.compute_int8_scale(rows)
    ix = pvs.VectorIndex(pvs.I8, dim)
    ix.set_scale(scale)
    ix.add_f32(rows, row_ids=ids)
    ix.set_order_keys(mtime)
    conn = sqlite3.connect(":memory:")
    conn.execute("CREATE TABLE files (id INTEGER PRIMARY KEY, last_modified INTEGER NOT NULL)")
    conn.executemany("INSERT INTO files VALUES (?, ?)", list(zip(ids.tolist(), mtime.tolist())))
    sqlite_seam.load(conn)
    sqlite_seam.bind("ties", ix)
    try:
        for qi in (0, 17, 49):
            q = (base[qi] + 0.01 * orc.synth_rows(32, qi, 1, dim)[0]).astype(np.float32)
            sql = """SELECT p.id, p.d FROM pvs_dist('ties', ?, 'l2') p JOIN files f ON f.id = p.id
                     ORDER BY p.d ASC NULLS LAST, f.last_modified DESC, p.id LIMIT ?"""
            want = conn.execute(sql, (q.tobytes(), k)).fetchall()
            gi, gd, gc = ix.search(q, k, pvs.L2)
            assert gc[0] == k and gi[0, :k].tolist() == [w[0] for w in want]
            assert [np.float32(w[1]) for w in want] == gd[0, :k].tolist()
            assert len({w[1] for w in want}) < 4, "the page must consist of ties"
    finally:
        sqlite_seam.unbind("ties")
        ix.close()


@pytest.mark.gpu
def test_item_page_with_order_keys_equals_sqlites_group_by_order_by():
    """Per-item page against SQLite itself: `SELECT file_id, MIN(d) ... GROUP BY file_id ORDER BY 2 ASC NULLS LAST,
    last_modified DESC, file_id LIMIT k` over the `d` column of pvs_dist (exact.rs:67-80 + model.rs:547-553) must be the page
    pvs_search_groups returns once the rows carry their file's last_modified."""
    import panoptikon_amd as pvs
    from panoptikon_amd import sqlite_seam

    rng = np.random.default_rng(22)
    dim, distinct, files, k = 64, 30, 700, 50
    base = orc.synth_rows(41, 0, distinct, dim)
    per_file = rng.integers(1, 4, files)
    fid = np.arange(1, files + 1, dtype=np.int64) * 5
    grp = np.repeat(fid, per_file)
    rows = base[np.repeat(rng.integers(0, distinct, files), per_file)]
    n = len(rows)
    ids = np.arange(1, n + 1, dtype=np.int64)
    fm = rng.integers(0, 9, files).astype(np.int64) + 1_700_000_000
    scale = orc.compute_int8_scale(rows)
    ix = pvs.VectorIndex(pvs.I8, dim)
    ix.set_scale(scale)
    ix.add_f32(rows, row_ids=ids, group_ids=grp)
    ix.set_order_keys(np.repeat(fm, per_file))
    conn = sqlite3.connect(":memory:")
    conn.execute("CREATE TABLE files (id INTEGER PRIMARY KEY, last_modified INTEGER NOT NULL)")
    conn.execute("CREATE TABLE emb (id INTEGER PRIMARY KEY, file_id INTEGER NOT NULL)")
    conn.executemany("INSERT INTO files VALUES (?, ?)", list(zip(fid.tolist(), fm.tolist())))
    conn.executemany("INSERT INTO emb VALUES (?, ?)", list(zip(ids.tolist(), grp.tolist())))
    sqlite_seam.load(conn)
    sqlite_seam.bind("items", ix)
    try:
        for qi in (0, 11, 29):
            q = (base[qi] + 0.01 * orc.synth_rows(42, qi, 1, dim)[0]).astype(np.float32)
            for fn, agg in (("MIN", pvs.AGG_MIN), ("MAX", pvs.AGG_MAX)):
                sql = f"""SELECT e.file_id, {fn}(p.d) AS v FROM pvs_dist('items', ?, 'l2') p JOIN emb e ON e.id = p.id
                          JOIN files f ON f.id = e.file_id GROUP BY e.file_id
                          ORDER BY v ASC NULLS LAST, f.last_modified DESC, e.file_id LIMIT ?"""
                want = conn.execute(sql, (q.tobytes(), k)).fetchall()
                og, ov, oc = ix.search_groups(q, k, pvs.L2, agg)
                assert oc[0] == k and og[0, :k].tolist() == [w[0] for w in want], (qi, fn)
                assert ov[0, :k].tolist() == [w[1] for w in want]
                assert len({w[1] for w in want}) < 6, "the page must consist of ties"
    finally:
        sqlite_seam.unbind("items")
        ix.close()
