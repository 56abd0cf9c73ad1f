"""Control plane of the one-process-per-GPU form on ONE node: rendezvous, barrier and tiny host all-gathers over a
Unix-domain socket (rank 0 is the hub).  No torch, no MPI: the data path of a sharded search is RCCL inside libpvs
(pvs_comm.hip); this only carries the 128-byte RCCL unique id, one float for the int8 scale, barriers and the timing
reduction of bench.py.  Works under `python -m torch.distributed.run` (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_PORT /
TORCHELASTIC_RUN_ID from its environment) and under bench.py's own launcher (PVS_CTL_SOCK)."""
from __future__ import annotations

import os
import socket
import struct
import time
from typing import List, Optional

import numpy as np


def _recv_exact(s: socket.socket, n: int) -> bytes:
    buf = bytearray()
    while len(buf) < n:
        chunk = s.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("control-plane peer closed the connection")
        buf += chunk
    return bytes(buf)


def _send_msg(s: socket.socket, b: bytes) -> None:
    s.sendall(struct.pack("<Q", len(b)) + b)


def _recv_msg(s: socket.socket) -> bytes:
    (n,) = struct.unpack("<Q", _recv_exact(s, 8))
    return _recv_exact(s, n)


def default_socket_path() -> str:
    p = os.environ.get("PVS_CTL_SOCK")
    if p:
        return p
    run = os.environ.get("TORCHELASTIC_RUN_ID", "none")
    port = os.environ.get("MASTER_PORT", "0")
    return f"/tmp/pvs_ctl_{os.getuid()}_{port}_{run}.sock"


class LocalRendezvous:
    """world ranks on one node; every method is a collective all ranks call in the same order."""

    def __init__(self, rank: int, world: int, path: Optional[str] = None, timeout: float = 180.0):
        self.rank, self.world = rank, world
        self.path = path or default_socket_path()
        self.peers: List[socket.socket] = []
        self.hub: Optional[socket.socket] = None
        self._srv: Optional[socket.socket] = None
        if world <= 1:
            return
        if rank == 0:
            try:
                os.unlink(self.path)
            except FileNotFoundError:
                pass
            srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
            srv.bind(self.path)
            srv.listen(world)
            srv.settimeout(timeout)
            got = {}
            while len(got) < world - 1:
                c, _ = srv.accept()
                c.settimeout(None)
                (r,) = struct.unpack("<I", _recv_exact(c, 4))
                if r in got or not (0 < r < world):
                    raise RuntimeError(f"control plane: unexpected rank {r}")
                got[r] = c
            self.peers = [got[r] for r in range(1, world)]
            self._srv = srv
        else:
            deadline = time.time() + timeout
            while True:
                s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
                try:
                    s.connect(self.path)
                    break
                except (FileNotFoundError, ConnectionRefusedError):
                    s.close()
                    if time.time() > deadline:
                        raise TimeoutError(f"control plane: rank 0 never listened on {self.path}")
                    time.sleep(0.05)
            s.sendall(struct.pack("<I", rank))
            self.hub = s

    # ---- collectives
    def allgather_bytes(self, b: bytes) -> List[bytes]:
        if self.world <= 1:
            return [b]
        if self.rank == 0:
            parts = [b] + [_recv_msg(p) for p in self.peers]
            blob = b"".join(struct.pack("<Q", len(x)) + x for x in parts)
            for p in self.peers:
                _send_msg(p, blob)
            return parts
        _send_msg(self.hub, b)
        blob = _recv_msg(self.hub)
        parts, off = [], 0
        for _ in range(self.world):
            (n,) = struct.unpack_from("<Q", blob, off)
            parts.append(blob[off + 8: off + 8 + n])
            off += 8 + n
        return parts

    def barrier(self) -> None:
        self.allgather_bytes(b"")

    def max_float(self, v: float) -> float:
        return max(struct.unpack("<d", x)[0] for x in self.allgather_bytes(struct.pack("<d", float(v))))

    def min_float(self, v: float) -> float:
        return min(struct.unpack("<d", x)[0] for x in self.allgather_bytes(struct.pack("<d", float(v))))

    def bcast_bytes(self, b: Optional[bytes]) -> bytes:
        return self.allgather_bytes(b if self.rank == 0 and b is not None else b"")[0]

    def all_gather_np(self, a: np.ndarray) -> np.ndarray:
        a = np.ascontiguousarray(a)
        return np.stack([np.frombuffer(x, dtype=a.dtype).reshape(a.shape) for x in self.allgather_bytes(a.tobytes())])

    __call__ = all_gather_np  # usable as the `gather` argument of panoptikon_amd.sharded.merge_shard_pages

    def close(self) -> None:
        for p in self.peers:
            p.close()
        if self.hub:
            self.hub.close()
        if self._srv:
            self._srv.close()
            try:
                os.unlink(self.path)
            except OSError:
                pass
        self.peers, self.hub, self._srv = [], None, None
