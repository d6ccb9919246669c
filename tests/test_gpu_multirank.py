"""Two ranks, two GPUs, RCCL: the composed frame of trc_group_reduce_accum[_async], the sample-sharded compose
(trc_group_compose_samples: ncclSend / ncclRecv all-to-all + gather) and the SPPM pass with its AllReduce / AllGather must
equal the one-rank results bit for bit.  Needs a node with >= 2 GPUs (skipped on the 1-GPU boxes of this pool; the driver's
multi-GPU tier runs it)."""
import json
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import ROOT
from tracer_amd import abi, host

pytestmark = pytest.mark.gpu


def _n_gpus():
    """distinct physical GPUs a rank of this process tree can open, by PCI bus id (asked in a CHILD: collecting the tests must
    not initialise HIP here) -- not a count of environment entries"""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from tracer_amd import device\n"
            "ids = set()\n"
            "for i in range(16):\n"
            "    try:\n"
            "        t = device.Tracer(i)\n"
            "    except Exception:\n"
            "        break\n"
            "    ids.add(t.pci_bus_id()); t.close()\n"
            "print(len(ids))\n") % ROOT
    try:
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
        return int(out.stdout.strip().splitlines()[-1]) if out.returncode == 0 and out.stdout.strip() else 0
    except (OSError, ValueError, subprocess.TimeoutExpired):
        return 0


WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    sys.path.insert(0, os.environ["TRC_ROOT"])
    from tracer_amd import abi, host
    from tracer_amd.device import Tracer, group_unique_id
    from tracer_amd.socket_group import SocketGroup
    group = SocketGroup.from_env()
    rank, world = group.rank, group.world
    W, H, spp = 320, 192, 6
    scene = host.HostScene(abi.SCENE_CORNELL_SPHERES)
    t = Tracer(rank)
    t.upload_scene(scene.view); t.set_camera(host.prepare_camera(W, H)); t.set_environment((0.0, 0.0, 0.0)); t.resize(W, H)
    t.group_init(group.broadcast(group_unique_id() if rank == 0 else None), world, rank)
    out = {}
    # path-traced frame, synchronous and pipelined compose (three steps in flight)
    t.clear_accum(); t.seed(31); t.render(spp=spp, tile_rank=rank, tile_nranks=world); t.group_reduce_accum(0)
    if rank == 0: out["sync"] = t.download_accum()
    for step in range(3):
        t.clear_accum(); t.seed(40 + step); t.render(spp=spp, tile_rank=rank, tile_nranks=world)
        t.group_reduce_accum_async(0)
        if rank == 0: out[f"async{step}"] = t.download_composed()
    # sample sharding over ncclSend / ncclRecv: synchronous, pipelined, and delivered to every rank
    t.synchronize(); t.clear_accum(); t.seed(abi.shard_seed(70, rank)); t.render(spp=spp); t.group_compose_samples(0)
    if rank == 0: out["samples"] = t.download_composed()
    t.group_compose_samples_async(0, 2)
    if rank == 0: out["samples_async"] = t.download_composed()
    t.synchronize(); t.clear_accum(); t.seed(abi.shard_seed(70, rank)); t.render(spp=spp); t.group_allreduce_mean_accum()
    out[f"mean{rank}"] = t.download_accum()
    if rank == 1: np.savez(os.environ["TRC_OUT"] + ".rank1.npz", mean1=out.pop("mean1"))
    # SPPM: bounds all-reduced, photons all-gathered, frame composed
    t.synchronize(); t.clear_accum(); t.seed(8); t.sppm_init(9); t.sppm_frames(3)
    cam, pho, mark, count, cx = t.sppm_download()
    t.group_reduce_accum(0)
    if rank == 0:
        out["sppm"] = t.download_accum(); out["pho"] = pho.view(np.uint8); out["count"] = count
        out["total"] = np.float32(cx.totalPhotonSum)
        np.savez(os.environ["TRC_OUT"], **out)
    t.group_finalize(); group.barrier(); group.close(); t.close()
""")


@pytest.mark.skipif(_n_gpus() < 2, reason="needs two GPUs (RCCL refuses two ranks on one device)")
def test_two_ranks_over_rccl_equal_one_rank(gpu, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    out = str(tmp_path / "two.npz")
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541",
                   TRC_ROOT=ROOT, TRC_OUT=out, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env))
    for p in procs:
        assert p.wait(timeout=600) == 0
    two = np.load(out)
    W, H, spp = 320, 192, 6
    scene = host.HostScene(abi.SCENE_CORNELL_SPHERES)
    gpu.upload_scene(scene.view); gpu.set_camera(host.prepare_camera(W, H)); gpu.set_environment((0.0, 0.0, 0.0)); gpu.resize(W, H)
    gpu.clear_accum(); gpu.seed(31); gpu.render(spp=spp)
    assert np.array_equal(two["sync"].view(np.uint32), gpu.download_accum().view(np.uint32))
    for step in range(3):
        gpu.clear_accum(); gpu.seed(40 + step); gpu.render(spp=spp)
        assert np.array_equal(two[f"async{step}"].view(np.uint32), gpu.download_accum().view(np.uint32)), step
    shards = []
    for g in range(2):
        gpu.clear_accum(); gpu.seed(abi.shard_seed(70, g)); gpu.render(spp=spp)
        shards.append(gpu.download_accum())
    want = np.divide(np.add(shards[0], shards[1], dtype=np.float32), np.float32(2), dtype=np.float32)
    for k in ("samples", "samples_async", "mean0"):
        assert np.array_equal(two[k].view(np.uint32), want.view(np.uint32)), k
    assert np.array_equal(np.load(out + ".rank1.npz")["mean1"].view(np.uint32), want.view(np.uint32))
    gpu.clear_accum(); gpu.seed(8); gpu.sppm_init(9); gpu.sppm_frames(3)
    cam, pho, mark, count, cx = gpu.sppm_download()
    assert np.array_equal(two["sppm"].view(np.uint32), gpu.download_accum().view(np.uint32))
    assert np.array_equal(two["pho"], pho.view(np.uint8)) and np.array_equal(two["count"], count)
    assert two["total"] == np.float32(cx.totalPhotonSum)
