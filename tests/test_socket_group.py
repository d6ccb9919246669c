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
    np.savez(os.path.join(os.environ["TRC_OUT"], f"r{rank}.npz"), frame=frame, kmin=kmin, kmax=kmax, pho=pho,
             calls=np.array([coll.calls["reduce"], coll.calls["allreduce"], coll.calls["allgather"]]))
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
        assert list(res[r]["calls"]) == [1, 2, 1]


def test_a_stale_port_file_is_survived(tmp_path):
    """a file left behind by an earlier run with the same MASTER_PORT and parent names a dead port: the ranks keep reading until
    rank 0 has published the live one"""
    import tempfile
    stale = os.path.join(tempfile.gettempdir(), f"trc_rdzv_29401_{os.getpid()}")
    open(stale, "w").write("1 deadbeef")
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
