"""The SQLite side of the boundary from Python: loads libpvs_sqlite.so (panoptikon_amd/csrc/pvs_sqlite.cpp, declared in
include/pvs_sqlite.h) into a stdlib `sqlite3` connection and binds device indexes to the names the SQL uses.

    seam.load(conn)                      # pvs_dist(...), pvs_distance_cosine(...), pvs_distance_l2(...) now exist on conn
    seam.bind("clip", index)             # `index` is a VectorIndex
    conn.execute("SELECT p.id, p.d FROM pvs_dist('clip', ?, 'cosine') AS p ...", (query_f32.tobytes(),))

The compute is the library's: this module only loads the extension and passes handles."""
from __future__ import annotations

import ctypes as C
import os
import sqlite3

from . import _lib as L

EXT_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libpvs_sqlite.so")
_ext = None


def ext() -> C.CDLL:
    global _ext
    if _ext is None:
        if not os.path.exists(EXT_PATH):
            raise ImportError(f"{EXT_PATH} is missing: build it with `python -m panoptikon_amd.build`")
        L.lib()  # libpvs.so first (the extension links it)
        e = C.CDLL(EXT_PATH, mode=C.RTLD_GLOBAL)
        e.pvs_sqlite_bind_index.restype = C.c_int32
        e.pvs_sqlite_bind_index.argtypes = [C.c_char_p, C.c_void_p]
        e.pvs_sqlite_unbind_index.restype = C.c_int32
        e.pvs_sqlite_unbind_index.argtypes = [C.c_char_p]
        _ext = e
    return _ext


def load(conn: sqlite3.Connection) -> None:
    """Registers the SQL functions on `conn` through SQLite's loadable-extension ABI (sqlite3_extension_init)."""
    ext()
    conn.enable_load_extension(True)
    try:
        conn.load_extension(EXT_PATH)
    finally:
        conn.enable_load_extension(False)


def bind(name: str, index) -> None:
    if ext().pvs_sqlite_bind_index(name.encode(), index._h) != 0:
        raise ValueError(f"cannot bind index name {name!r}")


def unbind(name: str) -> None:
    ext().pvs_sqlite_unbind_index(name.encode())
