"""Index lifecycle glue: from a Panoptikon index database (SQLite) to device-resident shards.

Host-side only (stdlib ``sqlite3``; the image has no SQLite headers for a C++ twin and no Rust).
It mirrors, for a host that is not the reference's Rust process:

* readiness of a (profile, setters) pair — ``resolve_ready_pair`` / ``default_profile_name`` /
  ``active_profile_id`` (reference ``db/vector_quants.rs:1795-1929``): active profile, every *existing*
  setter ``state='ready'``, a usable scale artifact, equal scale and dim across xmodal siblings;
* the row streams the device index is loaded from: ``embeddings`` (f32 LE blobs, dim =
  len/4) and ``embedding_quants`` (int8 codes at the coverage row's ``artifact_rev``), both in
  ``item_data.id`` order — the order ``BACKFILL_CHUNK_SQL`` streams (``db/vector_quants.rs:1085-1099``) and the
  order ``pvs_index_add`` requires;
* invalidation: a cached device index is keyed by (database, kind, setters, profile id, artifact rev)
  and dropped when the caller's index epoch moves (``db/epochs.rs``: any index-DB write bumps it).

Row id = ``item_data.id``; group id = ``item_data.item_id`` (files of an item expand on the host, as the
reference's ``files`` join does).
"""
from __future__ import annotations

import sqlite3
import zlib
from dataclasses import dataclass, field
from typing import Dict, Iterable, Iterator, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib as L
from .host import artifact_scale


@dataclass(frozen=True)
class ReadyPair:
    """reference ``ReadyPair`` (db/vector_quants.rs:1784-1788)"""
    profile_id: int
    scale: float
    dim: int


def active_profile_id(conn: sqlite3.Connection, name: str) -> Optional[int]:
    row = conn.execute("SELECT id FROM vector_quant_profiles WHERE name = ? AND state = 'active'", (name,)).fetchone()
    return None if row is None else int(row[0])


def default_profile_name(conn: sqlite3.Connection) -> Optional[str]:
    row = conn.execute("SELECT name FROM vector_quant_profiles WHERE is_default = 1 AND state = 'active' LIMIT 1").fetchone()
    return None if row is None else str(row[0])


def _setter_id(conn: sqlite3.Connection, name: str) -> Optional[int]:
    row = conn.execute("SELECT id FROM setters WHERE name = ?", (name,)).fetchone()
    return None if row is None else int(row[0])


def resolve_ready_pair(conn: sqlite3.Connection, profile_name: str, setter_names: Sequence[str]) -> Optional[ReadyPair]:
    """None unless the profile is active and every existing involved setter's pair is ready with a usable
    scale and the same (scale, dim) — the ``auto`` fallback contract (db/vector_quants.rs:1795-1869).
    Setter names without a ``setters`` row are skipped."""
    profile_id = active_profile_id(conn, profile_name)
    if profile_id is None:
        return None
    result: Optional[ReadyPair] = None
    for name in setter_names:
        sid = _setter_id(conn, name)
        if sid is None:
            continue
        row = conn.execute("SELECT artifact, dim FROM vector_quant_coverage WHERE profile_id = ? AND setter_id = ? AND state = 'ready'",
                           (profile_id, sid)).fetchone()
        if row is None:
            return None
        artifact, dim = row
        scale = artifact_scale(bytes(artifact)) if artifact is not None else None
        if dim is None or scale is None:
            return None  # no dimension or no usable scale: not queryable whatever the state column says
        if result is None:
            result = ReadyPair(profile_id, float(scale), int(dim))
        elif np.float32(result.scale) != np.float32(scale) or result.dim != int(dim):
            return None  # xmodal siblings must share one artifact; a mismatch means a rebuild is pending
    return result


def _existing_setter_ids(conn: sqlite3.Connection, setter_names: Sequence[str]) -> List[int]:
    return [sid for sid in (_setter_id(conn, n) for n in setter_names) if sid is not None]


def iter_exact_rows(conn: sqlite3.Connection, setter_names: Sequence[str], chunk_rows: int = 65536, after_id: int = -1,
                    dim_bytes: Optional[int] = None) -> Iterator[Tuple[np.ndarray, np.ndarray, np.ndarray]]:
    """(row ids, item ids, f32 [n][dim]) chunks of the setters' embeddings in ``item_data.id`` order.
    dim is fixed by the first row; blobs of another length are skipped (the quant backfill applies the
    same ``length(embedding) = dim*4`` guard)."""
    sids = _existing_setter_ids(conn, setter_names)
    if not sids:
        return
    marks = ",".join("?" * len(sids))
    cur = conn.execute(f"SELECT d.id, d.item_id, e.embedding FROM item_data d JOIN embeddings e ON e.id = d.id "
                       f"WHERE d.setter_id IN ({marks}) AND d.id > ? ORDER BY d.id", [*sids, after_id])
    while True:
        rows = cur.fetchmany(chunk_rows)
        if not rows:
            return
        if dim_bytes is None:
            dim_bytes = len(rows[0][2])
        keep = [r for r in rows if r[2] is not None and len(r[2]) == dim_bytes and dim_bytes % 4 == 0 and dim_bytes > 0]
        if not keep:
            continue
        ids = np.fromiter((r[0] for r in keep), np.int64, len(keep))
        items = np.fromiter((r[1] for r in keep), np.int64, len(keep))
        mat = np.frombuffer(b"".join(r[2] for r in keep), dtype="<f4").reshape(len(keep), dim_bytes // 4)
        yield ids, items, mat


def iter_quant_rows(conn: sqlite3.Connection, profile_id: int, setter_names: Sequence[str], dim: int, chunk_rows: int = 65536,
                    after_id: int = -1) -> Iterator[Tuple[np.ndarray, np.ndarray, np.ndarray]]:
    """(row ids, item ids, int8 [n][dim]) chunks of the profile's codes at each setter's current
    ``artifact_rev``, in ``item_data.id`` order (embedding_quants schema: migrations/index/20260730150000)."""
    sids = _existing_setter_ids(conn, setter_names)
    if not sids:
        return
    marks = ",".join("?" * len(sids))
    cur = conn.execute(
        f"SELECT d.id, d.item_id, q.quant FROM vector_quant_coverage c JOIN item_data d ON d.setter_id = c.setter_id "
        f"JOIN embedding_quants q ON q.id = d.id AND q.profile_id = c.profile_id AND q.rev = c.artifact_rev "
        f"WHERE c.profile_id = ? AND c.setter_id IN ({marks}) AND d.id > ? ORDER BY d.id", [profile_id, *sids, after_id])
    while True:
        rows = cur.fetchmany(chunk_rows)
        if not rows:
            return
        keep = [r for r in rows if len(r[2]) == dim]
        if not keep:
            continue
        ids = np.fromiter((r[0] for r in keep), np.int64, len(keep))
        items = np.fromiter((r[1] for r in keep), np.int64, len(keep))
        mat = np.frombuffer(b"".join(r[2] for r in keep), dtype=np.int8).reshape(len(keep), dim)
        yield ids, items, mat


def coverage_revs(conn: sqlite3.Connection, profile_id: int, setter_names: Sequence[str]) -> Tuple[Tuple[int, int], ...]:
    """((setter id, artifact_rev), ...) — part of the cache key of a quant index."""
    out = []
    for sid in _existing_setter_ids(conn, setter_names):
        row = conn.execute("SELECT artifact_rev FROM vector_quant_coverage WHERE profile_id = ? AND setter_id = ?", (profile_id, sid)).fetchone()
        out.append((sid, -1 if row is None else int(row[0])))
    return tuple(out)


TAIL_WINDOW = 1024
_FP_MASK = (1 << 32) - 1  # sums of 32-bit residues stay below 2^63 for 2^31 rows (SQLite's SUM raises on i64 overflow)


@dataclass
class LoadedIndex:
    index: object  # VectorIndex
    kind: str      # "exact" | "quant"
    rows: int
    dim: int
    profile_id: Optional[int] = None
    scale: Optional[float] = None
    key: tuple = field(default_factory=tuple)
    last_id: int = -1  # largest item_data.id loaded (appends resume after it)
    # what validates the loaded prefix after an epoch bump (append_new_rows): integer sums over every loaded
    # (id, item_id) and the (id, item_id, crc32(payload)) of the newest TAIL_WINDOW rows
    sum_id: int = 0
    sum_item: int = 0
    tail: list = field(default_factory=list)

    def note_chunk(self, ids: np.ndarray, items: np.ndarray, mat: np.ndarray) -> None:
        self.rows += len(ids)
        self.last_id = int(ids[-1])
        self.sum_id += int(np.sum(ids & _FP_MASK, dtype=np.int64))
        self.sum_item += int(np.sum(items & _FP_MASK, dtype=np.int64))
        w = min(len(ids), TAIL_WINDOW)
        raw = np.ascontiguousarray(mat[len(ids) - w:])
        new = [(int(ids[len(ids) - w + i]), int(items[len(ids) - w + i]), zlib.crc32(raw[i].tobytes())) for i in range(w)]
        self.tail = (self.tail + new)[-TAIL_WINDOW:]


_EXACT_FROM = "FROM item_data d JOIN embeddings e ON e.id = d.id WHERE d.setter_id IN ({marks}) AND d.id > ?"
_QUANT_FROM = ("FROM vector_quant_coverage c JOIN item_data d ON d.setter_id = c.setter_id "
               "JOIN embedding_quants q ON q.id = d.id AND q.profile_id = c.profile_id AND q.rev = c.artifact_rev "
               "WHERE c.profile_id = ? AND c.setter_id IN ({marks}) AND d.id > ?")
_native_seq = 0


def _native_stream(conn: sqlite3.Connection, li: LoadedIndex, setter_names: Sequence[str], after_id: int) -> int:
    """The same row stream as iter_exact_rows / iter_quant_rows, pulled by the C streamer of libpvs_sqlite.so
    (`pvs_load`, include/pvs_sqlite.h): SQLite page cache -> staging buffer -> device, no per-row Python objects.
    Updates li's fingerprint from the streamer's sums and re-reads the newest TAIL_WINDOW rows for the tail."""
    import json

    from . import sqlite_seam as seam

    global _native_seq
    sids = _existing_setter_ids(conn, setter_names)
    if not sids:
        return 0
    marks = ",".join("?" * len(sids))
    if li.kind == "exact":
        frm, payload, args = _EXACT_FROM.format(marks=marks), "e.embedding", [*sids, after_id]
        nbytes = li.dim * 4
    else:
        frm, payload, args = _QUANT_FROM.format(marks=marks), "q.quant", [li.profile_id, *sids, after_id]
        nbytes = li.dim
    seam.load(conn)
    _native_seq += 1
    name = f"__loader_{id(li):x}_{_native_seq}"
    seam.bind(name, li.index)
    try:
        # blobs of another length are skipped by the streamer as well; the guard keeps them out of the statement entirely
        n = int(conn.execute(f"SELECT pvs_load(?, ?, {','.join('?' * (len(args) + 1))})",
                             [name, f"SELECT d.id, d.item_id, {payload} {frm} AND length({payload}) = ? ORDER BY d.id", *args, nbytes]).fetchone()[0])
        info = json.loads(conn.execute("SELECT pvs_load_info(?)", (name,)).fetchone()[0])
    finally:
        seam.unbind(name)
    if n:
        li.rows += n
        li.last_id = int(info["last_id"])
        li.sum_id += int(info["sum_id"])
        li.sum_item += int(info["sum_group"])
        tail = [(int(r[0]), int(r[1]), zlib.crc32(bytes(r[2]))) for r in conn.execute(
            f"SELECT d.id, d.item_id, {payload} {frm} AND length({payload}) = ? ORDER BY d.id DESC LIMIT {TAIL_WINDOW}", [*args, nbytes])]
        li.tail = (li.tail + tail[::-1])[-TAIL_WINDOW:]
    return n


def _first_payload_bytes(conn: sqlite3.Connection, setter_names: Sequence[str]) -> Optional[int]:
    sids = _existing_setter_ids(conn, setter_names)
    if not sids:
        return None
    marks = ",".join("?" * len(sids))
    row = conn.execute(f"SELECT length(e.embedding) {_EXACT_FROM.format(marks=marks)} ORDER BY d.id LIMIT 1", [*sids, -1]).fetchone()
    return None if row is None or row[0] is None else int(row[0])


def load_exact_index(conn: sqlite3.Connection, setter_names: Sequence[str], dtype: int = L.F32, device: int = 0,
                     chunk_rows: int = 65536, native: bool = False) -> Optional[LoadedIndex]:
    """The reference's *exact* mode on the device: the setters' f32 embeddings as an f32 (or f16) index.
    native=True streams the rows through the C streamer (pvs_load) instead of Python chunks."""
    from .index import VectorIndex

    if native:
        nbytes = _first_payload_bytes(conn, setter_names)
        if not nbytes or nbytes % 4:
            return None
        li = LoadedIndex(VectorIndex(dtype, nbytes // 4, device=device), "exact", 0, nbytes // 4)
        _native_stream(conn, li, setter_names, -1)
        return li
    li = None
    for ids, items, mat in iter_exact_rows(conn, setter_names, chunk_rows):
        if li is None:
            dim = mat.shape[1]
            li = LoadedIndex(VectorIndex(dtype, dim, device=device), "exact", 0, dim)
        li.index.add_f32(mat, row_ids=ids, group_ids=items)
        li.note_chunk(ids, items, mat)
    return li


def load_quant_index(conn: sqlite3.Connection, profile_name: str, setter_names: Sequence[str], device: int = 0,
                     chunk_rows: int = 65536, native: bool = False) -> Optional[LoadedIndex]:
    """The *quant* mode: int8 codes of a ready pair with its frozen scale.  None when the pair is not ready
    (the caller falls back to exact under ``auto``, or raises under strict selection — pql/preprocess.rs:327-383)."""
    from .index import VectorIndex

    pair = resolve_ready_pair(conn, profile_name, setter_names)
    if pair is None:
        return None
    ix = VectorIndex(L.I8, pair.dim, device=device)
    ix.set_scale(pair.scale)
    li = LoadedIndex(ix, "quant", 0, pair.dim, profile_id=pair.profile_id, scale=pair.scale)
    if native:
        _native_stream(conn, li, setter_names, -1)
        return li
    for ids, items, mat in iter_quant_rows(conn, pair.profile_id, setter_names, pair.dim, chunk_rows):
        ix.add(mat, row_ids=ids, group_ids=items)
        li.note_chunk(ids, items, mat)
    return li


def _prefix_intact(conn: sqlite3.Connection, li: LoadedIndex, setter_names: Sequence[str]) -> bool:
    """True iff the rows the loader would stream with id <= li.last_id are the rows the index holds.

    `item_data.id` is INTEGER PRIMARY KEY *without* AUTOINCREMENT (init.sql:94): SQLite hands out max(id)+1, so
    once the newest rows are deleted their ids are reused by the next extraction, and a row count alone cannot
    tell the old prefix from the new one.  Checks: (1) COUNT and integer sums of (id, item_id) over the whole
    prefix — one index scan; (2) the newest TAIL_WINDOW loaded rows are re-read and their (id, item_id,
    crc32(payload)) compared.  (2) is where reuse can hide: a reused id is larger than every id that survived
    the delete, so if the newest loaded row is still there unchanged no id at or below it was reused, and if it is
    not, the tail comparison fails."""
    sids = _existing_setter_ids(conn, setter_names)
    if not sids:
        return li.rows == 0
    marks = ",".join("?" * len(sids))
    fp = f"COUNT(*), COALESCE(SUM(d.id & {_FP_MASK}), 0), COALESCE(SUM(d.item_id & {_FP_MASK}), 0)"
    lo = li.tail[0][0] if li.tail else li.last_id + 1
    if li.kind == "exact":
        frm = (f"FROM item_data d JOIN embeddings e ON e.id = d.id WHERE d.setter_id IN ({marks}) AND d.id <= ? "
               f"AND length(e.embedding) = ?")
        args = [*sids, li.last_id, li.dim * 4]
        payload = "e.embedding"
    else:
        frm = (f"FROM vector_quant_coverage c JOIN item_data d ON d.setter_id = c.setter_id "
               f"JOIN embedding_quants q ON q.id = d.id AND q.profile_id = c.profile_id AND q.rev = c.artifact_rev "
               f"WHERE c.profile_id = ? AND c.setter_id IN ({marks}) AND d.id <= ? AND length(q.quant) = ?")
        args = [li.profile_id, *sids, li.last_id, li.dim]
        payload = "q.quant"
    cnt, sid, sitem = conn.execute(f"SELECT {fp} {frm}", args).fetchone()
    if (int(cnt), int(sid), int(sitem)) != (li.rows, li.sum_id, li.sum_item):
        return False
    tail = [(int(r[0]), int(r[1]), zlib.crc32(bytes(r[2]))) for r in
            conn.execute(f"SELECT d.id, d.item_id, {payload} {frm} AND d.id >= ? ORDER BY d.id", [*args, lo])]
    return tail == li.tail


def append_new_rows(conn: sqlite3.Connection, li: LoadedIndex, setter_names: Sequence[str], chunk_rows: int = 65536,
                    native: bool = False) -> Optional[int]:
    """After an epoch bump: if every row the index holds is still there (extractions are immutable; deletes
    cascade), only rows with a larger item_data.id are new — append them.  Returns the number appended, or
    None when the prefix changed and the caller must rebuild."""
    if not _prefix_intact(conn, li, setter_names):
        return None
    before = li.rows
    if native:
        return _native_stream(conn, li, setter_names, li.last_id)
    if li.kind == "exact":
        stream = iter_exact_rows(conn, setter_names, chunk_rows, after_id=li.last_id, dim_bytes=li.dim * 4)
        for ids, items, mat in stream:
            li.index.add_f32(mat, row_ids=ids, group_ids=items)
            li.note_chunk(ids, items, mat)
    else:
        for ids, items, mat in iter_quant_rows(conn, li.profile_id, setter_names, li.dim, chunk_rows, after_id=li.last_id):
            li.index.add(mat, row_ids=ids, group_ids=items)
            li.note_chunk(ids, items, mat)
    return li.rows - before


def reconcile_deletions(conn: sqlite3.Connection, li: LoadedIndex, setter_names: Sequence[str]) -> Optional[int]:
    """After an epoch bump whose prefix check failed: if what changed is that rows the index holds are GONE from the database
    (`embeddings ... ON DELETE CASCADE` when a file disappears: migrations/index/20250117193000_init.sql:29-33, db/files.rs:175-192),
    remove exactly those rows from the device index (pvs_index_remove_rows: compaction in HBM, milliseconds) instead of reloading
    everything from SQLite.  Returns the number of rows removed, or None when the change is not of that kind and the caller must
    rebuild.  Cost: the device side is milliseconds; the HOST side is O(rows) — every id at or below the anchor is pulled from
    SQLite (`SELECT d.id ... ORDER BY d.id`, ~1 us per row) and the index's ids and groups are read back — still far below a
    reload (which also moves and re-ingests every payload), but seconds, not milliseconds, at 10M rows.  IndexCache re-runs the
    prefix fingerprint (_prefix_intact) afterwards and rebuilds if anything else changed under the same ids.

    Safe against reused ids (`item_data.id` is not AUTOINCREMENT, see _prefix_intact): an ANCHOR is needed — the newest row of the
    loaded tail window that is still in the database with the same (id, item_id, payload crc).  Ids at or below the anchor were never
    reused (a reused id is larger than every id that survived), so a loaded id at or below it is either still there, unchanged, or
    gone; loaded rows ABOVE the anchor are dropped too (whatever sits at those ids now is new content: append_new_rows brings it in).
    No anchor in the window, or an id at or below the anchor that the index never held: rebuild."""
    sids = _existing_setter_ids(conn, setter_names)
    if not sids or not li.tail:
        return None
    marks = ",".join("?" * len(sids))
    if li.kind == "exact":
        frm = (f"FROM item_data d JOIN embeddings e ON e.id = d.id WHERE d.setter_id IN ({marks}) AND d.id <= ? "
               f"AND length(e.embedding) = ?")
        args = [*sids, li.last_id, li.dim * 4]
        payload = "e.embedding"
    else:
        frm = (f"FROM vector_quant_coverage c JOIN item_data d ON d.setter_id = c.setter_id "
               f"JOIN embedding_quants q ON q.id = d.id AND q.profile_id = c.profile_id AND q.rev = c.artifact_rev "
               f"WHERE c.profile_id = ? AND c.setter_id IN ({marks}) AND d.id <= ? AND length(q.quant) = ?")
        args = [li.profile_id, *sids, li.last_id, li.dim]
        payload = "q.quant"
    now = {int(r[0]): (int(r[1]), zlib.crc32(bytes(r[2]))) for r in
           conn.execute(f"SELECT d.id, d.item_id, {payload} {frm} AND d.id >= ? ORDER BY d.id", [*args, li.tail[0][0]])}
    anchor = None
    for rid, item, crc in reversed(li.tail):
        if now.get(rid) == (item, crc):
            anchor = rid
            break
    if anchor is None:
        return None
    current = np.fromiter((int(r[0]) for r in conn.execute(f"SELECT d.id {frm} AND d.id <= ? ORDER BY d.id", [*args, anchor])), np.int64)
    loaded, groups = li.index.read_ids(0, li.rows, groups=True)
    keep = np.isin(loaded, current, assume_unique=True) & (loaded <= anchor)
    if int(keep.sum()) != len(current):  # a row at or below the anchor that the index never held
        return None
    gone = loaded[~keep]
    removed = li.index.remove_rows(gone) if len(gone) else 0
    if removed != len(gone):
        return None
    # the fingerprint of what is left
    li.rows = int(keep.sum())
    li.last_id = int(anchor)
    li.sum_id = int(np.sum(loaded[keep] & _FP_MASK, dtype=np.int64))
    li.sum_item = int(np.sum(groups[keep] & _FP_MASK, dtype=np.int64))
    li.tail = [(int(r[0]), int(r[1]), zlib.crc32(bytes(r[2]))) for r in conn.execute(
        f"SELECT d.id, d.item_id, {payload} {frm} AND d.id <= ? ORDER BY d.id DESC LIMIT {TAIL_WINDOW}", [*args, anchor])][::-1]
    return removed


class IndexCache:
    """Device indexes keyed by what makes them stale.  ``epoch`` is the host's index epoch for the database
    (db/epochs.rs:38-46: bumped by every index-DB write); quant indexes additionally key on the coverage
    rows' ``artifact_rev`` so a rebuilt scale never serves old codes."""

    def __init__(self):
        self._items: Dict[tuple, Tuple[int, LoadedIndex]] = {}

    def get(self, conn: sqlite3.Connection, db_name: str, epoch: int, setter_names: Sequence[str],
            profile_name: Optional[str] = None, dtype: int = L.F32, device: int = 0) -> Optional[LoadedIndex]:
        names = tuple(setter_names)
        if profile_name is None:
            key = (db_name, "exact", names, int(dtype), device)
        else:
            pid = active_profile_id(conn, profile_name)
            if pid is None:
                return None
            key = (db_name, "quant", names, pid, coverage_revs(conn, pid, names), device)
        hit = self._items.get(key)
        if hit is not None and hit[0] == epoch:
            return hit[1]
        if hit is not None:  # the epoch moved under the same key (same artifact revs): try to append only
            still_ok = True
            if profile_name is not None:  # the pair must still be ready, with the scale the codes were made with
                pair = resolve_ready_pair(conn, profile_name, names)
                still_ok = pair is not None and np.float32(pair.scale) == np.float32(hit[1].scale) and pair.dim == hit[1].dim
            # Any failure of the in-place routes (a PvsError out of an add or a removal, e.g. an id the index refuses) falls through to
            # the rebuild below, which drops the index: an entry that failed once must not be retried in place on every later get
            # (before the library lowered last_id after a tail removal, re-added rowids raised here — on EVERY call)
            try:
                if still_ok and append_new_rows(conn, hit[1], names) is not None:
                    self._items[key] = (epoch, hit[1])
                    return hit[1]
                # rows the index holds were deleted: take exactly those out of HBM, then append what is new (append_new_rows checks the
                # whole-prefix fingerprint again first: surviving rows that changed under the same id mean a rebuild)
                if still_ok and reconcile_deletions(conn, hit[1], names) is not None and append_new_rows(conn, hit[1], names) is not None:
                    self._items[key] = (epoch, hit[1])
                    return hit[1]
            except L.PvsError:
                pass
        # absent, or the loaded prefix changed: drop every entry of this (database, kind, setters) and rebuild
        for k in [k for k in self._items if k[:3] == key[:3]]:
            self._items.pop(k)[1].index.close()
        loaded = (load_exact_index(conn, names, dtype, device) if profile_name is None
                  else load_quant_index(conn, profile_name, names, device))
        if loaded is None:
            return None
        loaded.key = key
        self._items[key] = (epoch, loaded)
        return loaded

    def clear(self):
        for _, li in self._items.values():
            li.index.close()
        self._items.clear()
