"""N > 1 path on CPU: 2 processes over gloo.  Each rank renders ITS pixel tiles (the oracle stands in
for the kernel here -- there is no GPU in this container) into a zero frame, then one reduce(sum)
to rank 0 composes the frame exactly like trc_group_reduce_accum (ncclReduce over xGMI) does."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

from conftest import ROOT
from oracle import pyoracle as po
from tracer_amd import abi, host

WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.environ["TRC_ROOT"])
    from oracle import pyoracle as po
    from tracer_amd import abi, host
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    W, H, spp, seed = 96, 64, 3, 31
    scene = host.HostScene(abi.SCENE_CORNELL_SPHERES)
    cam = host.prepare_camera(W, H)
    acc, st = po.render(scene.view, cam, W, H, host.fill_rng(seed, W, H), spp=spp, env=(0.1, 0.2, 0.3),
                        tile_rank=rank, tile_nranks=world, n_threads=2)
    frame = torch.from_numpy(acc)
    rays = torch.tensor([float(st.rays)], dtype=torch.float64)
    dist.barrier()
    dist.reduce(frame, dst=0, op=dist.ReduceOp.SUM)          # sum-with-zeros == gather of tiles
    dist.all_reduce(rays, op=dist.ReduceOp.SUM)
    if rank == 0:
        np.save(os.environ["TRC_OUT"], frame.numpy())
        open(os.environ["TRC_OUT"] + ".rays", "w").write(str(int(rays.item())))
    dist.barrier()
    dist.destroy_process_group()
""")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_compose_the_single_rank_frame(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    out = str(tmp_path / "frame.npy")
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   TRC_ROOT=ROOT, TRC_OUT=out, OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0
    composed = np.load(out)
    W, H, spp, seed = 96, 64, 3, 31
    scene = host.HostScene(abi.SCENE_CORNELL_SPHERES)
    full, st = po.render(scene.view, host.prepare_camera(W, H), W, H, host.fill_rng(seed, W, H), spp=spp,
                         env=(0.1, 0.2, 0.3))
    assert np.array_equal(composed.view(np.uint32), full.view(np.uint32))
    assert int(open(out + ".rays").read()) == st.rays


def test_tile_owner_partition_is_a_partition():
    for W, H in [(1920, 1080), (97, 61)]:
        ty, tx = np.mgrid[0:H, 0:W] // abi.TRC_TILE
        for n in (1, 2, 4, 8):
            owner = (tx + ty) % n
            counts = np.bincount(owner.ravel(), minlength=n)
            assert counts.sum() == W * H
            if (W, H) == (1920, 1080):
                assert counts.max() / counts.min() < 1.02          # balanced pixel counts per GPU
