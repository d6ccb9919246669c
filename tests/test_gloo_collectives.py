"""The harness side of trc_group_set_collectives on CPU: two gloo processes call the table's five functions directly on
host buffers (what the library does after staging a device buffer), for exactly the shapes the path uses -- f32 sum to a
root, u32 min / max of order-preserving keys, in-place all-gather of byte ranges, and the all-to-all / gather of pixel
slices the sample-sharded compose makes."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

from conftest import ROOT

WORKER = textwrap.dedent("""
    import ctypes as C, os, sys
    import numpy as np
    import torch.distributed as dist
    sys.path.insert(0, os.environ["TRC_ROOT"])
    from tracer_amd import abi
    from tracer_amd.gloo_collectives import GlooCollectives
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = GlooCollectives()
    T = g.table
    ptr = lambda a: C.c_void_p(a.ctypes.data)
    # compose: each rank holds its own stripes, zeros elsewhere
    frame = np.zeros(4096, np.float32)
    frame[rank::world] = np.arange(4096, dtype=np.float32)[rank::world] * np.float32(0.37) + np.float32(rank)
    assert T.reduce(None, ptr(frame), frame.size, abi.DT_F32, abi.OP_SUM, 0, None) == 0
    # bound keys: 3 x u32, including values above 2^31 (gloo has no unsigned reductions: the harness widens)
    kmin = np.array([0xFFFFFFF0 - rank, 5 + rank, 0x80000000 + rank], np.uint32)
    kmax = kmin.copy()
    assert T.allreduce(None, ptr(kmin), 3, abi.DT_U32, abi.OP_MIN, None) == 0
    assert T.allreduce(None, ptr(kmax), 3, abi.DT_U32, abi.OP_MAX, None) == 0
    # photon records: rank r's byte range filled, the rest garbage
    per = 80 * 16
    pho = np.full(per * world, 0xEE, np.uint8)
    pho[rank * per:(rank + 1) * per] = (np.arange(per) * (rank + 3)) & 0xFF
    assert T.allgather(None, ptr(pho), per, None) == 0
    # sample-sharded compose: all-to-all of the accumulator's slices, gather of the composed slices to a root
    sl = 4099 * 16
    a2a = np.empty(sl * world, np.uint8)
    for p in range(world):
        a2a[p * sl:(p + 1) * sl] = (np.arange(sl) * 7 + 31 * rank + 5 * p) & 0xFF
    assert T.alltoall(None, ptr(a2a), sl, None) == 0
    gat = np.full(sl * world, 0xAB, np.uint8)
    gat[rank * sl:(rank + 1) * sl] = (np.arange(sl) * 3 + rank) & 0xFF
    assert T.gather(None, ptr(gat), sl, 1, None) == 0
    np.savez(os.path.join(os.environ["TRC_OUT"], f"r{rank}.npz"), frame=frame, kmin=kmin, kmax=kmax, pho=pho, a2a=a2a, gat=gat,
             calls=np.array([g.calls[k] for k in ("reduce", "allreduce", "allgather", "alltoall", "gather")]))
    dist.barrier(); dist.destroy_process_group()
""")


def test_gloo_table_runs_the_path_s_three_collectives(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    world = 2
    procs = [subprocess.Popen([sys.executable, str(script)],
                              env=dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                                       TRC_ROOT=ROOT, TRC_OUT=str(tmp_path), OMP_NUM_THREADS="1")) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    res = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    want = np.zeros(4096, np.float32)
    for r in range(world):
        want[r::world] = np.arange(4096, dtype=np.float32)[r::world] * np.float32(0.37) + np.float32(r)
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
        for p in range(world):
            assert np.array_equal(res[r]["a2a"][p * sl:(p + 1) * sl], ((np.arange(sl) * 7 + 31 * p + 5 * r) & 0xFF).astype(np.uint8)), (r, p)
    assert np.array_equal(res[1]["gat"], np.concatenate([((np.arange(sl) * 3 + r) & 0xFF).astype(np.uint8) for r in range(world)]))
