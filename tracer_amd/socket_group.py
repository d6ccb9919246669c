"""Rendezvous, barrier and small collectives for the ranks of ONE node over TCP sockets -- harness code for `bench.py --gpus N`
and the tests, so that the N-rank path needs no PyTorch (north_star: "no PyTorch"; the library itself composes over RCCL).

Launch contract (the driver's `python -m torch.distributed.run ...` or bench.py's own launcher): RANK, WORLD_SIZE, MASTER_ADDR,
MASTER_PORT in the environment.  MASTER_PORT itself belongs to the launcher (torchrun's agent keeps its store there), so rank 0
listens on an ephemeral LOOPBACK port and publishes it, with a random token, in a file named after MASTER_PORT and the ranks'
common parent process; the other ranks read it.  Star topology: every collective goes through rank 0, in rank order, so
reductions are deterministic.

Trust model (one node, several users): nothing a peer sends is interpreted before the peer has proven that it knows the token.
  * the listener binds 127.0.0.1 only (the ranks of one node; a non-loopback MASTER_ADDR is not honoured);
  * the port file lives in a directory of the user's own (0700, owner checked, not a symlink), is created with
    O_EXCL | O_NOFOLLOW and mode 0600, and a reader refuses a file that is not a regular file owned by its own uid;
  * handshake: rank 0 sends a 16-byte challenge; the peer answers a FIXED-SIZE struct (magic, rank, world, its own 16-byte
    challenge, HMAC-SHA256 over all of it and rank 0's challenge under the token); rank 0 compares with hmac.compare_digest and
    answers with an HMAC over the peer's challenge, so each side knows the other holds the token;
  * objects travel as JSON (bytes as hex), never as pickles: a peer can at worst send wrong numbers.

`SocketCollectives` is the `trc_group_set_collectives` table on top of it (host-staged), for runs with more ranks than GPUs
(RCCL refuses two ranks on one device): the same reduce / all-reduce / all-gather / all-to-all / gather program as RCCL's, moved
by sockets.  Nothing here computes anything of the path.
"""
import ctypes as C
import hashlib
import hmac
import json
import os
import secrets
import socket
import stat
import struct
import tempfile
import time

import numpy as np

from . import abi

_MAGIC = b"TRCRDZV2"
_HELLO = struct.Struct("<8sII16s32s")          # magic, rank, world, peer challenge, HMAC
_MAX_OBJECT_BYTES = 1 << 26                    # a gathered object is a line of a bench record, not a frame


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


def _send(sock, payload):
    sock.sendall(struct.pack("<Q", len(payload)))
    sock.sendall(payload)


def _recv(sock):
    (n,) = struct.unpack("<Q", bytes(_recv_exact(sock, 8)))
    if n > _MAX_OBJECT_BYTES:
        raise ConnectionError(f"rendezvous: a {n}-byte object frame")
    return bytes(_recv_exact(sock, n))


def _enc(obj):
    """JSON with bytes spelled {"__bytes__": hex}; tuples become lists"""
    def default(o):
        if isinstance(o, (bytes, bytearray)):
            return {"__bytes__": bytes(o).hex()}
        if isinstance(o, (np.integer,)):
            return int(o)
        if isinstance(o, (np.floating,)):
            return float(o)
        raise TypeError(f"SocketGroup objects are JSON values or bytes, not {type(o).__name__}")
    return json.dumps(obj, default=default).encode()


def _dec(blob):
    def hook(d):
        if len(d) == 1 and "__bytes__" in d:
            return bytes.fromhex(d["__bytes__"])
        return d
    return json.loads(blob.decode(), object_hook=hook)


def _mac(token, *parts):
    return hmac.new(token, b"".join(parts), hashlib.sha256).digest()


def private_dir():
    """<tmp>/trc_rdzv_<uid>: made 0700, and refused unless it is a real directory of ours that nobody else can write"""
    path = os.path.join(tempfile.gettempdir(), f"trc_rdzv_{os.getuid()}")
    try:
        os.mkdir(path, 0o700)
    except FileExistsError:
        pass
    st = os.lstat(path)
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise PermissionError(f"rendezvous directory {path} is not a private directory of uid {os.getuid()}")
    return path


def publish(path, text):
    """a new 0600 file, never through a symlink, never over somebody else's file"""
    try:
        os.unlink(path)                      # a stale file of an earlier run of ours (the directory is ours alone)
    except FileNotFoundError:
        pass
    fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_EXCL | os.O_NOFOLLOW, 0o600)
    with os.fdopen(fd, "w") as f:
        f.write(text)


def read_published(path):
    fd = os.open(path, os.O_RDONLY | os.O_NOFOLLOW)
    with os.fdopen(fd, "r") as f:
        st = os.fstat(f.fileno())
        if not stat.S_ISREG(st.st_mode) or st.st_uid != os.getuid():
            raise PermissionError(f"{path} is not a regular file of uid {os.getuid()}")
        return f.read()


class SocketGroup:
    """group = SocketGroup.from_env(); group.barrier(); group.allreduce_scalar(x, "MAX"); group.gather(obj); group.broadcast(obj)"""

    def __init__(self, rank, world, addr="127.0.0.1", key="0", timeout_s=300.0):
        self.rank, self.world = int(rank), int(world)
        self._peers = []            # rank 0: sockets of ranks 1 .. world-1, in rank order
        self._hub = None            # other ranks: the socket to rank 0
        self._file = None
        if self.world == 1:
            return
        del addr                    # one node: loopback, whatever MASTER_ADDR says (see the trust model above)
        key = "".join(ch for ch in str(key) if ch.isalnum())[:32] or "0"
        self._file = os.path.join(private_dir(), f"{key}_{os.getppid()}")
        deadline = time.monotonic() + timeout_s
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind(("127.0.0.1", 0))
            srv.listen(self.world)
            token = secrets.token_bytes(32)
            publish(self._file, f"{srv.getsockname()[1]} {token.hex()}")
            slots = [None] * self.world
            srv.settimeout(1.0)
            while any(s is None for s in slots[1:]):
                if time.monotonic() > deadline:
                    raise TimeoutError(f"rendezvous: {sum(s is None for s in slots[1:])} of {self.world - 1} ranks never connected")
                try:
                    c, _ = srv.accept()
                except socket.timeout:
                    continue
                c.settimeout(10.0)                           # an unauthenticated peer gets seconds, not the run's timeout
                r = self._admit(c, token, slots)
                if r is None:
                    c.close()
                    continue
                c.settimeout(timeout_s)
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                slots[r] = c
            srv.close()
            self._peers = slots[1:]
            for c in self._peers:
                _send(c, b"go")
        else:
            while True:
                if time.monotonic() > deadline:
                    raise TimeoutError("rendezvous: rank 0 never published its port")
                s = None
                try:
                    port, token_hex = read_published(self._file).split()
                    token = bytes.fromhex(token_hex)
                    s = socket.create_connection(("127.0.0.1", int(port)), timeout=5.0)
                    # the handshake gets seconds: a stale port file may name a port an unrelated listener holds by now, and a
                    # wrong listener must cost one retry (the file is read again), not the run's whole timeout
                    s.settimeout(10.0)
                    s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    theirs = bytes(_recv_exact(s, 16))
                    mine = secrets.token_bytes(16)
                    head = struct.pack("<8sII16s", _MAGIC, self.rank, self.world, mine)
                    s.sendall(head + _mac(token, head, theirs))
                    proof = bytes(_recv_exact(s, 32))
                    if hmac.compare_digest(proof, _mac(token, b"rank0", mine)):
                        s.settimeout(max(10.0, deadline - time.monotonic()))      # rank 0 says "go" when EVERY rank has been admitted
                        if _recv(s) == b"go":
                            s.settimeout(timeout_s)
                            self._hub = s
                            break
                    s.close()
                except (OSError, ValueError, ConnectionError):
                    if s is not None:
                        s.close()                            # no file yet, a stale file of an earlier run, or rank 0 not listening yet
                time.sleep(0.05)

    def _admit(self, c, token, slots):
        """rank 0: challenge a new connection; its rank when it proves the token and claims a free slot of this world"""
        try:
            mine = secrets.token_bytes(16)
            c.sendall(mine)
            magic, rank, world, theirs, mac = _HELLO.unpack(bytes(_recv_exact(c, _HELLO.size)))
            head = struct.pack("<8sII16s", magic, rank, world, theirs)
            if not hmac.compare_digest(mac, _mac(token, head, mine)):
                return None
            if magic != _MAGIC or world != self.world or not 0 < rank < self.world or slots[rank] is not None:
                return None
            c.sendall(_mac(token, b"rank0", theirs))
            return rank
        except (OSError, ConnectionError, struct.error):
            return None

    @classmethod
    def from_env(cls, timeout_s=300.0):
        return cls(int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
                   os.environ.get("MASTER_ADDR", "127.0.0.1"), os.environ.get("MASTER_PORT", "0"), timeout_s)

    def close(self):
        for c in self._peers:
            c.close()
        if self._hub is not None:
            self._hub.close()
        if self.rank == 0 and self.world > 1 and self._file:
            try:
                os.remove(self._file)
            except OSError:
                pass
        self._peers, self._hub = [], None

    # ---- objects (JSON values and bytes)
    def gather(self, obj):
        """every rank's object, in rank order, on every rank"""
        if self.world == 1:
            return [obj]
        if self.rank == 0:
            out = [obj] + [_dec(_recv(c)) for c in self._peers]
            blob = _enc(out)
            for c in self._peers:
                _send(c, blob)
            return _dec(blob)                                # every rank sees the same (JSON-normalised) values
        _send(self._hub, _enc(obj))
        return _dec(_recv(self._hub))

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
        self.gather_bytes(a, bytes_per_rank, root=None)

    def gather_bytes(self, a, bytes_per_rank, root=0):
        """like allgather_bytes, the result defined on `root` only (None: on every rank)"""
        if self.world == 1:
            return
        mine = slice(self.rank * bytes_per_rank, (self.rank + 1) * bytes_per_rank)
        if self.rank == 0:
            for r, c in enumerate(self._peers, start=1):
                _recv_exact(c, bytes_per_rank, a[r * bytes_per_rank:(r + 1) * bytes_per_rank])
            for r, c in enumerate(self._peers, start=1):
                if root is None or r == root:
                    c.sendall(memoryview(a).cast("B"))
        else:
            self._hub.sendall(memoryview(np.ascontiguousarray(a[mine])).cast("B"))
            if root is None or self.rank == root:
                _recv_exact(self._hub, a.nbytes, a)

    def alltoall_bytes(self, a, bytes_per_rank):
        """a: uint8 view of world slices of bytes_per_rank; on return slice p holds what rank p had in ITS slice `rank`"""
        if self.world == 1:
            return
        n, b = self.world, bytes_per_rank
        if self.rank == 0:
            rows = [a] + [np.empty(n * b, np.uint8) for _ in self._peers]          # rows[r] = rank r's buffer
            for r, c in enumerate(self._peers, start=1):
                _recv_exact(c, n * b, rows[r])
            out = np.empty(n * b, np.uint8)
            for r in range(n):                                                     # what rank r receives: slice r of every row
                for p in range(n):
                    out[p * b:(p + 1) * b] = rows[p][r * b:(r + 1) * b]
                if r == 0:
                    keep = out.copy()
                else:
                    self._peers[r - 1].sendall(memoryview(out).cast("B"))
            a[:] = keep
        else:
            self._hub.sendall(memoryview(np.ascontiguousarray(a)).cast("B"))
            _recv_exact(self._hub, n * b, a)


REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
ALLTOALL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)


class Collectives(C.Structure):
    """trc_collectives"""
    _fields_ = [("user", C.c_void_p), ("host_staged", C.c_int32), ("_pad", C.c_int32),
                ("reduce", REDUCE_FN), ("allreduce", ALLREDUCE_FN), ("allgather", ALLGATHER_FN),
                ("alltoall", ALLTOALL_FN), ("gather", GATHER_FN)]


_NP = {abi.DT_U8: np.uint8, abi.DT_U32: np.uint32, abi.DT_F32: np.float32}


def host_view(buf, count, dtype):
    return np.ctypeslib.as_array(C.cast(buf, C.POINTER(np.ctypeslib.as_ctypes_type(_NP[dtype]))), shape=(count,))


class SocketCollectives:
    """table = SocketCollectives(group); tracer.set_collectives(table, world, rank).  Keeps the ctypes callbacks alive; `calls`
    counts what the library asked for."""

    def __init__(self, group):
        self.group, self.world, self.rank = group, group.world, group.rank
        self.calls = {"reduce": 0, "allreduce": 0, "allgather": 0, "alltoall": 0, "gather": 0}
        self._cb = (REDUCE_FN(self._reduce), ALLREDUCE_FN(self._allreduce), ALLGATHER_FN(self._allgather),
                    ALLTOALL_FN(self._alltoall), GATHER_FN(self._gather))
        self.table = Collectives(None, 1, 0, *self._cb)

    _view = staticmethod(host_view)

    def _guarded(self, name, fn):
        try:
            fn()
            self.calls[name] += 1
            return 0
        except Exception as e:      # never let an exception cross the C boundary
            print(f"SocketCollectives.{name}: {e!r}", flush=True)
            return 1

    def _reduce(self, user, buf, count, dtype, op, root, stream):
        return self._guarded("reduce", lambda: self.group.reduce(self._view(buf, count, dtype), op, root))

    def _allreduce(self, user, buf, count, dtype, op, stream):
        return self._guarded("allreduce", lambda: self.group.reduce(self._view(buf, count, dtype), op, 0, everywhere=True))

    def _allgather(self, user, buf, bytes_per_rank, stream):
        return self._guarded("allgather", lambda: self.group.allgather_bytes(self._view(buf, bytes_per_rank * self.world, abi.DT_U8), bytes_per_rank))

    def _alltoall(self, user, buf, bytes_per_rank, stream):
        return self._guarded("alltoall", lambda: self.group.alltoall_bytes(self._view(buf, bytes_per_rank * self.world, abi.DT_U8), bytes_per_rank))

    def _gather(self, user, buf, bytes_per_rank, root, stream):
        return self._guarded("gather", lambda: self.group.gather_bytes(self._view(buf, bytes_per_rank * self.world, abi.DT_U8), bytes_per_rank, root))
