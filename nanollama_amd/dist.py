"""Host-side rendezvous for one-process-per-GPU runs (bench.py under torch.distributed.run).

Only small things cross ranks on the host: the 64-byte hipIpc handles of the push all-reduce (all-gather) or the
128-byte RCCL unique id (rank 0 -> all), a barrier around the timed region, and the max over ranks of the measured time.  They travel over a plain TCP star rooted at
rank 0 (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT from the launcher's environment).  torch is NOT
imported here on purpose: the PyTorch wheel bundles its own ROCm 7.0 HIP runtime, and a process that loads
it next to /opt/rocm's runtime (which libnanollama_hip.so links) ends up with two HIP runtimes and
`hipErrorNoDevice` in the second one.  The data path (all-reduce over xGMI) is RCCL inside the library.
"""
from __future__ import annotations

import os
import socket
import struct
import time
import zlib
from typing import Callable, List, Optional

_MAGIC = b"NLRDV1"
_PORT_OFFSET = 23      # MASTER_PORT itself belongs to the launcher's own store
_PORT_TRIES = 8


def _send(sock: socket.socket, payload: bytes) -> None:
    sock.sendall(struct.pack("<I", len(payload)) + payload)


def _recv(sock: socket.socket) -> bytes:
    hdr = b""
    while len(hdr) < 4:
        chunk = sock.recv(4 - len(hdr))
        if not chunk:
            raise ConnectionError("rendezvous peer closed the connection")
        hdr += chunk
    n = struct.unpack("<I", hdr)[0]
    buf = b""
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("rendezvous peer closed the connection")
        buf += chunk
    return buf


class RendezvousDesync(RuntimeError):
    """A peer is at a different collective than this rank (one rank left the common control flow, e.g. after an
    exception): every message carries (operation, sequence number, tag) and a mismatch fails fast on both sides
    instead of pairing the wrong replies or waiting for the socket timeout."""


_OP_BCAST, _OP_GATHER, _OP_MAX = 1, 2, 3


class Rendezvous:
    def __init__(self, timeout_s: float = 300.0):
        self._seq = 0
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if os.environ.get("NL_BENCH_ONE_DEVICE"):  # developer switch: several ranks share GPU 0 (1-GPU box)
            self.local_rank = 0
        self._peers: List[socket.socket] = []   # rank 0: sockets of ranks 1..world-1 (index = rank-1)
        self._root: Optional[socket.socket] = None
        self._listener: Optional[socket.socket] = None
        if self.world > 1:
            addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
            base = int(os.environ.get("MASTER_PORT", "29500")) + _PORT_OFFSET
            if self.rank == 0:
                self._serve(addr, base, timeout_s)
            else:
                self._connect(addr, base, timeout_s)

    @classmethod
    def solo(cls) -> "Rendezvous":
        """A world of one on this rank's own GPU (a side measurement of a single rank of a larger run)."""
        r = cls.__new__(cls)
        r.rank, r.world = 0, 1
        r.local_rank = 0 if os.environ.get("NL_BENCH_ONE_DEVICE") else int(os.environ.get("LOCAL_RANK", "0"))
        r._peers, r._root, r._listener = [], None, None
        r._seq = 0
        return r

    # ---- wiring ----
    def _serve(self, addr: str, base: int, timeout_s: float) -> None:
        last = None
        for k in range(_PORT_TRIES):
            try:
                s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                s.bind(("0.0.0.0" if addr not in ("127.0.0.1", "localhost") else "127.0.0.1", base + k))
                s.listen(self.world)
                self._listener = s
                break
            except OSError as exc:
                last = exc
                s.close()
        if self._listener is None:
            raise RuntimeError(f"rendezvous: no free port in [{base}, {base + _PORT_TRIES}): {last}")
        slots: List[Optional[socket.socket]] = [None] * (self.world - 1)
        deadline = time.time() + timeout_s
        self._listener.settimeout(1.0)
        while any(x is None for x in slots):
            if time.time() > deadline:
                raise TimeoutError("rendezvous: not every rank connected")
            try:
                c, _ = self._listener.accept()
            except socket.timeout:
                continue
            try:
                c.settimeout(10.0)
                hello = _recv(c)
                if not hello.startswith(_MAGIC):
                    c.close()
                    continue
                r = struct.unpack("<I", hello[len(_MAGIC):len(_MAGIC) + 4])[0]
                if not 1 <= r < self.world or slots[r - 1] is not None:
                    c.close()
                    continue
                _send(c, _MAGIC)
                c.settimeout(timeout_s)
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                slots[r - 1] = c
            except (OSError, ConnectionError, struct.error):
                c.close()
        self._peers = [s for s in slots if s is not None]

    def _connect(self, addr: str, base: int, timeout_s: float) -> None:
        deadline = time.time() + timeout_s
        while time.time() < deadline:
            for k in range(_PORT_TRIES):
                try:
                    s = socket.create_connection((addr, base + k), timeout=2.0)
                    s.settimeout(10.0)
                    _send(s, _MAGIC + struct.pack("<I", self.rank))
                    if _recv(s) == _MAGIC:
                        s.settimeout(timeout_s)
                        s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                        self._root = s
                        return
                    s.close()
                except (OSError, ConnectionError):
                    continue
            time.sleep(0.2)
        raise TimeoutError("rendezvous: could not reach rank 0")

    # ---- collectives over the star ----
    # every message = header (operation, sequence number of the collective on the sending rank, crc32 of the caller's
    # tag) + payload; the receiver checks the header against its own position (RendezvousDesync)
    def _hdr(self, op: int, tag: str) -> bytes:
        return struct.pack("<BII", op, self._seq, zlib.crc32(tag.encode()))

    def _check(self, msg: bytes, op: int, tag: str) -> bytes:
        want = self._hdr(op, tag)
        if msg[:len(want)] != want:
            got = struct.unpack("<BII", msg[:9]) if len(msg) >= 9 else ()
            raise RendezvousDesync(f"rank {self.rank}: peer is at {got}, this rank at (op {op}, seq {self._seq}, tag {tag!r})")
        return msg[len(want):]

    def broadcast_bytes(self, make: Callable[[], bytes], tag: str = "") -> bytes:
        """Rank 0 calls make(); every rank returns the same bytes."""
        if self.world == 1:
            return make()
        try:
            if self.rank == 0:
                data = make()
                for p in self._peers:
                    _send(p, self._hdr(_OP_BCAST, tag) + data)
                return data
            return self._check(_recv(self._root), _OP_BCAST, tag)
        finally:
            self._seq += 1

    def allgather_bytes(self, mine: bytes, tag: str = "") -> List[bytes]:
        """Every rank contributes a byte string; every rank returns the rank-ordered list."""
        if self.world == 1:
            return [mine]
        try:
            if self.rank == 0:
                parts = [mine] + [self._check(_recv(p), _OP_GATHER, tag) for p in self._peers]
                blob = b"".join(struct.pack("<I", len(x)) + x for x in parts)
                for p in self._peers:
                    _send(p, self._hdr(_OP_GATHER, tag) + blob)
                return parts
            _send(self._root, self._hdr(_OP_GATHER, tag) + mine)
            blob, parts, off = self._check(_recv(self._root), _OP_GATHER, tag), [], 0
            while off < len(blob):
                n = struct.unpack_from("<I", blob, off)[0]
                parts.append(blob[off + 4:off + 4 + n])
                off += 4 + n
            return parts
        finally:
            self._seq += 1

    def barrier(self, tag: str = "barrier") -> None:
        self.max_over_ranks(0.0, tag)

    def max_over_ranks(self, value: float, tag: str = "") -> float:
        if self.world == 1:
            return value
        try:
            if self.rank == 0:
                vals = [value] + [struct.unpack("<d", self._check(_recv(p), _OP_MAX, tag))[0] for p in self._peers]
                m = max(vals)
                for p in self._peers:
                    _send(p, self._hdr(_OP_MAX, tag) + struct.pack("<d", m))
                return m
            _send(self._root, self._hdr(_OP_MAX, tag) + struct.pack("<d", value))
            return struct.unpack("<d", self._check(_recv(self._root), _OP_MAX, tag))[0]
        finally:
            self._seq += 1

    def close(self) -> None:
        for s in self._peers + [x for x in (self._root, self._listener) if x is not None]:
            try:
                s.close()
            except OSError:
                pass
        self._peers, self._root, self._listener = [], None, None
