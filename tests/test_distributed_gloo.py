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


SAMPLES_WORKER = textwrap.dedent("""
    import ctypes as C, os, sys
    import numpy as np
    import torch.distributed as dist
    sys.path.insert(0, os.environ["TRC_ROOT"])
    from oracle import pyoracle as po
    from tracer_amd import abi, host
    from tracer_amd.gloo_collectives import GlooCollectives
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    coll = GlooCollectives()
    W, H, spp, seed = 97, 61, 8, 77                             # 5917 pixels: the last slice is one pixel short
    scene = host.HostScene(abi.SCENE_CORNELL_SPHERES)
    acc, st = po.render(scene.view, host.prepare_camera(W, H), W, H, host.fill_rng(po.shard_seed(seed, rank), W, H),
                        spp=spp // world, env=(0.1, 0.2, 0.3), n_threads=2)
    # trc_abi.hip::compose_samples, on host buffers: padded slices, all-to-all, rank-ordered fold of MY slice, gather to the root
    n_px = W * H
    sl = (n_px + world - 1) // world
    buf = np.zeros((world * sl, 4), np.float32)
    buf[:n_px] = acc.reshape(-1, 4)
    assert coll.table.alltoall(None, C.c_void_p(buf.ctypes.data), sl * 16, None) == 0
    parts = buf.reshape(world, sl, 4)
    fold = parts[0].copy()
    for p in range(1, world):
        fold = np.add(fold, parts[p], dtype=np.float32)
    out = np.zeros((world, sl, 4), np.float32)
    out[rank] = np.divide(fold, np.float32(world), dtype=np.float32)
    assert coll.table.gather(None, C.c_void_p(out.ctypes.data), sl * 16, 0, None) == 0
    if rank == 0:
        np.save(os.environ["TRC_OUT"], out.reshape(-1, 4)[:n_px].reshape(H, W, 4))
    dist.barrier()
    dist.destroy_process_group()
""")


def test_two_ranks_compose_the_sample_sharded_frame(tmp_path):
    """the N > 1 path of `--scaling samples` on CPU, world size 2 over gloo: each rank renders the WHOLE frame with spp / 2
    samples from its own seed (the oracle stands in for the kernel), the compose program of trc_group_compose_samples runs
    through the collectives table (all-to-all of pixel slices, rank-ordered fold, gather) == the oracle's definition"""
    script = tmp_path / "worker.py"
    script.write_text(SAMPLES_WORKER)
    out = str(tmp_path / "frame.npy")
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   TRC_ROOT=ROOT, TRC_OUT=out, OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0
    W, H, spp, seed = 97, 61, 8, 77
    scene = host.HostScene(abi.SCENE_CORNELL_SPHERES)
    want, _ = po.render_sample_sharded(scene.view, host.prepare_camera(W, H), W, H,
                                       [host.fill_rng(po.shard_seed(seed, g), W, H) for g in range(2)], spp, env=(0.1, 0.2, 0.3))
    assert np.array_equal(np.load(out).view(np.uint32), want.view(np.uint32))


def test_sample_sharding_definition_on_the_oracle():
    """properties of the definition itself (include/tracer_abi.h): group 0 keeps the seed, one group is the unsharded frame,
    the two statements of the seed function (harness, oracle) agree, alpha stays 1, and the composed frame estimates the same
    image as the unsharded one (both are spp-sample estimates: their means agree far better than single shards do)"""
    W, H, spp = 64, 40, 32
    scene = host.HostScene(abi.SCENE_CORNELL_SPHERES)
    cam = host.prepare_camera(W, H)
    assert po.shard_seed(5, 0) == 5 == abi.shard_seed(5, 0)
    rs = np.random.RandomState(1)
    for _ in range(64):
        seed, g = int(rs.randint(0, 2 ** 62)) * 4 + 3, int(rs.randint(0, 2 ** 31))
        assert po.shard_seed(seed, g) == abi.shard_seed(seed, g) == (seed + g * 0x9E3779B97F4A7C15) % 2 ** 64
    assert abi.shard_seed(2 ** 64 - 1, 1) == 0x9E3779B97F4A7C14
    one, _ = po.render_sample_sharded(scene.view, cam, W, H, [host.fill_rng(9, W, H)], spp)
    plain, st = po.render(scene.view, cam, W, H, host.fill_rng(9, W, H), spp=spp)
    assert np.array_equal(one.view(np.uint32), plain.view(np.uint32))
    eight, sts = po.render_sample_sharded(scene.view, cam, W, H, [host.fill_rng(po.shard_seed(9, g), W, H) for g in range(8)], spp)
    assert (eight[..., 3] == 1).all() and np.isfinite(eight).all() and not np.array_equal(eight, plain)
    assert sum(s.paths for s in sts) == st.paths == W * H * spp
    assert abs(eight[..., :3].mean() / plain[..., :3].mean() - 1) < 0.1


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
