"""tracer_amd/socket_group.py on CPU: the rendezvous / barrier / scalar reductions that bench.py's N-rank runs use instead of
torch.distributed, and the host-staged collectives table (trc_group_set_collectives) on top of it, called directly on host
buffers for exactly the shapes the path uses -- f32 sum to a root, u32 min / max of order-preserving keys, in-place all-gather of
byte ranges.  Three processes; started like the driver starts ranks (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT)."""
import os
import subprocess
import sys
import textwrap

import numpy as np

from conftest import ROOT

WORKER = textwrap.dedent("""
    import ctypes as C, os, sys
    import numpy as np
    sys.path.insert(0, os.environ["TRC_ROOT"])
    assert "torch" not in sys.modules
    from tracer_amd import abi
    from tracer_amd.socket_group import SocketGroup, SocketCollectives
    g = SocketGroup.from_env(timeout_s=60)
    rank, world = g.rank, g.world
    assert g.gather({"r": rank}) == [{"r": r} for r in range(world)]
    assert g.allreduce_scalar(rank + 1.5, "MAX") == world + 0.5 and g.allreduce_scalar(rank + 1, "SUM") == world * (world + 1) / 2
    assert g.broadcast("id-from-0" if rank == 0 else None) == "id-from-0"
    g.barrier()
    coll = SocketCollectives(g)
    T = coll.table
    ptr = lambda a: C.c_void_p(a.ctypes.data)
    frame = np.zeros(70001, np.float32)                      # an odd size, larger than a socket buffer
    frame[rank::world] = np.arange(70001, dtype=np.float32)[rank::world] * np.float32(0.37) + np.float32(rank)
    assert T.reduce(None, ptr(frame), frame.size, abi.DT_F32, abi.OP_SUM, 0, None) == 0
    kmin = np.array([0xFFFFFFF0 - rank, 5 + rank, 0x80000000 + rank], np.uint32)
    kmax = kmin.copy()
    assert T.allreduce(None, ptr(kmin), 3, abi.DT_U32, abi.OP_MIN, None) == 0
    assert T.allreduce(None, ptr(kmax), 3, abi.DT_U32, abi.OP_MAX, None) == 0
    per = 80 * 16
    pho = np.full(per * world, 0xEE, np.uint8)
    pho[rank * per:(rank + 1) * per] = (np.arange(per) * (rank + 3)) & 0xFF
    assert T.allgather(None, ptr(pho), per, None) == 0
    # the sample-sharded compose's two moves: all-to-all of the accumulator's slices, gather of the composed slices to a root
    sl = 4099 * 16                                           # bytes per slice: an odd pixel count
    a2a = np.empty(sl * world, np.uint8)
    for p in range(world):
        a2a[p * sl:(p + 1) * sl] = (np.arange(sl) * 7 + 31 * rank + 5 * p) & 0xFF          # what rank `rank` holds for rank p
    assert T.alltoall(None, ptr(a2a), sl, None) == 0
    gat = np.full(sl * world, 0xAB, np.uint8)
    gat[rank * sl:(rank + 1) * sl] = (np.arange(sl) * 3 + rank) & 0xFF
    assert T.gather(None, ptr(gat), sl, 1, None) == 0                                      # root 1, not the hub
    blob = bytes([0, 1, 255]) + b"binary"
    assert g.broadcast(blob) == blob                         # the RCCL id travels as bytes
    np.savez(os.path.join(os.environ["TRC_OUT"], f"r{rank}.npz"), frame=frame, kmin=kmin, kmax=kmax, pho=pho, a2a=a2a, gat=gat,
             calls=np.array([coll.calls[k] for k in ("reduce", "allreduce", "allgather", "alltoall", "gather")]))
    g.barrier(); g.close()
    assert "torch" not in sys.modules                        # the N-rank harness never pulls PyTorch in
""")


def test_socket_group_and_its_collectives_table(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    world = 3
    procs = [subprocess.Popen([sys.executable, str(script)],
                              env=dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT="29400",
                                       TRC_ROOT=ROOT, TRC_OUT=str(tmp_path))) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=120) == 0
    res = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    want = np.zeros(70001, np.float32)
    for r in range(world):
        want[r::world] = np.arange(70001, dtype=np.float32)[r::world] * np.float32(0.37) + np.float32(r)
    assert np.array_equal(res[0]["frame"].view(np.uint32), want.view(np.uint32))       # sum with zeros == gather, exact
    per = 80 * 16
    pho = np.concatenate([((np.arange(per) * (r + 3)) & 0xFF).astype(np.uint8) for r in range(world)])
    for r in range(world):
        assert list(res[r]["kmin"]) == [0xFFFFFFF0 - (world - 1), 5, 0x80000000]
        assert list(res[r]["kmax"]) == [0xFFFFFFF0, 5 + world - 1, 0x80000000 + world - 1]
        assert np.array_equal(res[r]["pho"], pho)
        assert list(res[r]["calls"]) == [1, 2, 1, 1, 1]
    sl = 4099 * 16
    for r in range(world):
        for p in range(world):                               # slice p of rank r == what rank p held for rank r
            assert np.array_equal(res[r]["a2a"][p * sl:(p + 1) * sl], ((np.arange(sl) * 7 + 31 * p + 5 * r) & 0xFF).astype(np.uint8)), (r, p)
    assert np.array_equal(res[1]["gat"], np.concatenate([((np.arange(sl) * 3 + r) & 0xFF).astype(np.uint8) for r in range(world)]))


def test_a_stale_port_file_is_survived(tmp_path):
    """a file left behind by an earlier run with the same MASTER_PORT and parent names a dead port: the ranks keep reading until
    rank 0 has published the live one"""
    from tracer_amd import socket_group as sg
    stale = os.path.join(sg.private_dir(), f"29401_{os.getpid()}")
    sg.publish(stale, "1 " + "de" * 32)
    script = tmp_path / "w.py"
    script.write_text(textwrap.dedent("""
        import os, sys, time
        sys.path.insert(0, os.environ["TRC_ROOT"])
        from tracer_amd.socket_group import SocketGroup
        if os.environ["RANK"] == "0": time.sleep(1.0)
        g = SocketGroup.from_env(timeout_s=60)
        assert g.allreduce_scalar(g.rank, "SUM") == 1.0
        g.close()
    """))
    procs = [subprocess.Popen([sys.executable, str(script)],
                              env=dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29401", TRC_ROOT=ROOT))
             for r in range(2)]
    for p in procs:
        assert p.wait(timeout=120) == 0
    assert not os.path.exists(stale)                         # rank 0 removes its file on close


def test_nothing_is_interpreted_before_the_peer_has_proven_the_token(tmp_path, monkeypatch):
    """rank 0 used to unpickle the hello of whoever connected.  Now: no pickle anywhere in the module; a peer
    that does not know the token (garbage, a well-formed hello under a wrong token) is dropped and the rendezvous still
    completes with the real rank; a non-loopback MASTER_ADDR is not honoured; the port file is 0600 in a 0700 directory of
    our own and is not read through a symlink."""
    import socket
    import stat
    import struct
    import time
    from tracer_amd import socket_group as sg

    def dropped(sock):                                       # closed, or reset because our bytes were never read
        try:
            return sock.recv(64) == b""
        except ConnectionResetError:
            return True
    src = open(sg.__file__).read()
    assert "import pickle" not in src and "pickle.loads" not in src
    d = sg.private_dir()
    assert stat.S_IMODE(os.lstat(d).st_mode) == 0o700
    key = "29477"
    path = os.path.join(d, f"{key}_{os.getpid()}")          # rank 0 below is a child of THIS process
    script = tmp_path / "r0.py"
    script.write_text(textwrap.dedent("""
        import os, sys
        sys.path.insert(0, os.environ["TRC_ROOT"])
        from tracer_amd.socket_group import SocketGroup
        g = SocketGroup(0, 2, "10.1.2.3", os.environ["KEY"], timeout_s=60)   # a non-loopback MASTER_ADDR is not honoured
        assert g.allreduce_scalar(1.0, "SUM") == 3.0
        g.close()
    """))
    p0 = subprocess.Popen([sys.executable, str(script)], env=dict(os.environ, TRC_ROOT=ROOT, KEY=key))
    try:
        t_end = time.monotonic() + 60
        while not os.path.exists(path):
            assert time.monotonic() < t_end and p0.poll() is None
            time.sleep(0.05)
        assert stat.S_IMODE(os.lstat(path).st_mode) == 0o600
        port = int(sg.read_published(path).split()[0])
        # 1. a pickle where the hello belongs: never looked at (it is not even the right size), the connection is dropped
        s = socket.create_connection(("127.0.0.1", port))
        s.settimeout(20)
        assert len(s.recv(16)) == 16
        s.sendall(b"cos" + bytes([10]) + b"system" + bytes([10]) + b"(S'true'" + bytes([10]) + b"tR." + b"." * 64)
        assert dropped(s)
        s.close()
        # 2. a well-formed hello under a wrong token: no proof comes back, no "go"
        s = socket.create_connection(("127.0.0.1", port))
        s.settimeout(20)
        ch = s.recv(16)
        head = struct.pack("<8sII16s", b"TRCRDZV2", 1, 2, b"x" * 16)
        s.sendall(head + sg._mac(b"not the token", head, ch))
        assert dropped(s)
        s.close()
        # 3. the real rank 1 still gets in (this process plays rank 0's sibling: same "parent")
        monkeypatch.setattr(os, "getppid", os.getpid)
        g1 = sg.SocketGroup(1, 2, "127.0.0.1", key, timeout_s=60)
        monkeypatch.undo()
        assert g1.allreduce_scalar(2.0, "SUM") == 3.0
        g1.close()
        assert p0.wait(timeout=60) == 0
    finally:
        if p0.poll() is None:
            p0.kill()
    # a symlink where the port file should be is not followed
    target = tmp_path / "elsewhere"
    target.write_text("1 " + "00" * 32)
    link = os.path.join(d, f"link_{os.getpid()}")
    os.symlink(target, link)
    try:
        import pytest
        with pytest.raises(OSError):
            sg.read_published(link)
    finally:
        os.unlink(link)
