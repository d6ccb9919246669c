"""Rendezvous, barrier and small collectives for the ranks of ONE node over TCP sockets -- harness code for `bench.py --gpus N`
and the tests, so that the N-rank path needs no PyTorch (north_star: "no PyTorch"; the library itself composes over RCCL).

Launch contract (the driver's `python -m torch.distributed.run ...` or bench.py's own launcher): RANK, WORLD_SIZE, MASTER_ADDR,
MASTER_PORT in the environment.  MASTER_PORT itself belongs to the launcher (torchrun's agent keeps its store there), so rank 0
listens on an ephemeral port and publishes it in a file named after MASTER_PORT and the ranks' common parent process; the other
ranks read it.  Star topology: every collective goes through rank 0, in rank order, so reductions are deterministic.

`SocketCollectives` is the `trc_group_set_collectives` table on top of it (host-staged), for runs with more ranks than GPUs
(RCCL refuses two ranks on one device): the same reduce / all-reduce / all-gather program as RCCL's, moved by sockets.
Nothing here computes anything of the path.
"""
import ctypes as C
import os
import pickle
import secrets
import socket
import struct
import tempfile
import time

import numpy as np

from . import abi

_MAGIC = b"TRCRDZV1"


def _send(sock, payload):
    sock.sendall(struct.pack("<Q", len(payload)))
    sock.sendall(payload)


def _recv_exact(sock, n, into=None):
    buf = into if into is not None else bytearray(n)
    view = memoryview(buf).cast("B")
    got = 0
    while got < n:
        k = sock.recv_into(view[got:], n - got)
        if k == 0:
            raise ConnectionError("peer closed the rendezvous socket")
        got += k
    return buf


def _recv(sock):
    (n,) = struct.unpack("<Q", bytes(_recv_exact(sock, 8)))
    return bytes(_recv_exact(sock, n))


class SocketGroup:
    """group = SocketGroup.from_env(); group.barrier(); group.allreduce_scalar(x, "MAX"); group.gather(obj); group.broadcast(obj)"""

    def __init__(self, rank, world, addr="127.0.0.1", key="0", timeout_s=300.0):
        self.rank, self.world = int(rank), int(world)
        self._peers = []            # rank 0: sockets of ranks 1 .. world-1, in rank order
        self._hub = None            # other ranks: the socket to rank 0
        self._file = os.path.join(tempfile.gettempdir(), f"trc_rdzv_{key}_{os.getppid()}")
        if self.world == 1:
            return
        deadline = time.monotonic() + timeout_s
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, 0))
            srv.listen(self.world)
            token = secrets.token_hex(8)
            tmp = self._file + f".{os.getpid()}"
            with open(tmp, "w") as f:
                f.write(f"{srv.getsockname()[1]} {token}")
            os.replace(tmp, self._file)                      # atomic: a reader sees the old file or the whole new one
            slots = [None] * self.world
            srv.settimeout(1.0)
            while any(s is None for s in slots[1:]):
                if time.monotonic() > deadline:
                    raise TimeoutError(f"rendezvous: {sum(s is None for s in slots[1:])} of {self.world - 1} ranks never connected")
                try:
                    c, _ = srv.accept()
                except socket.timeout:
                    continue
                c.settimeout(timeout_s)
                try:
                    hello = pickle.loads(_recv(c))
                    ok = hello.get("magic") == _MAGIC and hello.get("token") == token and hello.get("world") == self.world \
                        and 0 < hello.get("rank", 0) < self.world and slots[hello["rank"]] is None
                except Exception:
                    ok = False
                if not ok:
                    c.close()
                    continue
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                slots[hello["rank"]] = c
            srv.close()
            self._peers = slots[1:]
            for c in self._peers:
                _send(c, b"go")
        else:
            while True:
                if time.monotonic() > deadline:
                    raise TimeoutError("rendezvous: rank 0 never published its port")
                try:
                    port, token = open(self._file).read().split()
                    s = socket.create_connection((addr, int(port)), timeout=5.0)
                    s.settimeout(timeout_s)
                    s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    _send(s, pickle.dumps({"magic": _MAGIC, "token": token, "world": self.world, "rank": self.rank}))
                    if _recv(s) == b"go":
                        self._hub = s
                        break
                    s.close()
                except (OSError, ValueError, ConnectionError):
                    pass                                     # no file yet, a stale file of an earlier run, or rank 0 not listening yet
                time.sleep(0.05)

    @classmethod
    def from_env(cls, timeout_s=300.0):
        return cls(int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
                   os.environ.get("MASTER_ADDR", "127.0.0.1"), os.environ.get("MASTER_PORT", "0"), timeout_s)

    def close(self):
        for c in self._peers:
            c.close()
        if self._hub is not None:
            self._hub.close()
        if self.rank == 0 and self.world > 1:
            try:
                os.remove(self._file)
            except OSError:
                pass
        self._peers, self._hub = [], None

    # ---- objects
    def gather(self, obj):
        """every rank's object, in rank order, on every rank"""
        if self.world == 1:
            return [obj]
        if self.rank == 0:
            out = [obj] + [pickle.loads(_recv(c)) for c in self._peers]
            blob = pickle.dumps(out)
            for c in self._peers:
                _send(c, blob)
            return out
        _send(self._hub, pickle.dumps(obj))
        return pickle.loads(_recv(self._hub))

    def broadcast(self, obj, src=0):
        return self.gather(obj if self.rank == src else None)[src]

    def barrier(self):
        self.gather(None)

    def allreduce_scalar(self, x, op):
        vals = self.gather(float(x))
        return {"MAX": max, "MIN": min, "SUM": sum}[op](vals)

    # ---- arrays, in place (numpy views of host memory)
    @staticmethod
    def _combine(acc, other, op):
        if op == abi.OP_SUM:
            np.add(acc, other, out=acc)
        elif op == abi.OP_MAX:
            np.maximum(acc, other, out=acc)
        elif op == abi.OP_MIN:
            np.minimum(acc, other, out=acc)
        else:
            raise ValueError(f"unknown reduction {op}")

    def reduce(self, a, op, root=0, everywhere=False):
        """a: 1-D contiguous array, the same shape on every rank; the result lands on `root` (on every rank: all-reduce).
        Rank 0 folds the ranks in rank order -- sums of float32 are reproducible, and with zeros outside a rank's own tiles exact."""
        if self.world == 1:
            return
        if self.rank == 0:
            tmp = np.empty_like(a)
            for c in self._peers:
                _recv_exact(c, a.nbytes, tmp)
                self._combine(a, tmp, op)
            for r, c in enumerate(self._peers, start=1):
                if everywhere or r == root:
                    c.sendall(memoryview(a).cast("B"))
        else:
            self._hub.sendall(memoryview(a).cast("B"))
            if everywhere or self.rank == root:
                _recv_exact(self._hub, a.nbytes, a)

    def allgather_bytes(self, a, bytes_per_rank):
        """a: uint8 view of world * bytes_per_rank bytes; rank r's slice is valid on entry, all slices on return"""
        if self.world == 1:
            return
        mine = slice(self.rank * bytes_per_rank, (self.rank + 1) * bytes_per_rank)
        if self.rank == 0:
            for r, c in enumerate(self._peers, start=1):
                _recv_exact(c, bytes_per_rank, a[r * bytes_per_rank:(r + 1) * bytes_per_rank])
            for c in self._peers:
                c.sendall(memoryview(a).cast("B"))
        else:
            self._hub.sendall(memoryview(np.ascontiguousarray(a[mine])).cast("B"))
            _recv_exact(self._hub, a.nbytes, a)


REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)


class Collectives(C.Structure):
    """trc_collectives"""
    _fields_ = [("user", C.c_void_p), ("host_staged", C.c_int32), ("_pad", C.c_int32),
                ("reduce", REDUCE_FN), ("allreduce", ALLREDUCE_FN), ("allgather", ALLGATHER_FN)]


_NP = {abi.DT_U8: np.uint8, abi.DT_U32: np.uint32, abi.DT_F32: np.float32}


class SocketCollectives:
    """table = SocketCollectives(group); tracer.set_collectives(table, world, rank).  Keeps the ctypes callbacks alive; `calls`
    counts what the library asked for."""

    def __init__(self, group):
        self.group, self.world, self.rank = group, group.world, group.rank
        self.calls = {"reduce": 0, "allreduce": 0, "allgather": 0}
        self._cb = (REDUCE_FN(self._reduce), ALLREDUCE_FN(self._allreduce), ALLGATHER_FN(self._allgather))
        self.table = Collectives(None, 1, 0, *self._cb)

    @staticmethod
    def _view(buf, count, dtype):
        return np.ctypeslib.as_array(C.cast(buf, C.POINTER(np.ctypeslib.as_ctypes_type(_NP[dtype]))), shape=(count,))

    def _reduce(self, user, buf, count, dtype, op, root, stream):
        try:
            self.group.reduce(self._view(buf, count, dtype), op, root)
            self.calls["reduce"] += 1
            return 0
        except Exception as e:      # never let an exception cross the C boundary
            print(f"SocketCollectives.reduce: {e!r}", flush=True)
            return 1

    def _allreduce(self, user, buf, count, dtype, op, stream):
        try:
            self.group.reduce(self._view(buf, count, dtype), op, 0, everywhere=True)
            self.calls["allreduce"] += 1
            return 0
        except Exception as e:
            print(f"SocketCollectives.allreduce: {e!r}", flush=True)
            return 1

    def _allgather(self, user, buf, bytes_per_rank, stream):
        try:
            self.group.allgather_bytes(self._view(buf, bytes_per_rank * self.world, abi.DT_U8), bytes_per_rank)
            self.calls["allgather"] += 1
            return 0
        except Exception as e:
            print(f"SocketCollectives.allgather: {e!r}", flush=True)
            return 1
