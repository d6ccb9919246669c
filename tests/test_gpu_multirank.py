"""Two ranks, two GPUs, RCCL: the composed frame of trc_group_reduce_accum[_async] and the SPPM pass with its
AllReduce / AllGather must equal the one-rank results bit for bit.  Needs a node with >= 2 GPUs (skipped on the
1-GPU boxes of this pool; the driver's multi-GPU tier runs it)."""
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
    sys.path.insert(0, ROOT)
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench.visible_gpus()


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
    gpu.clear_accum(); gpu.seed(8); gpu.sppm_init(9); gpu.sppm_frames(3)
    cam, pho, mark, count, cx = gpu.sppm_download()
    assert np.array_equal(two["sppm"].view(np.uint32), gpu.download_accum().view(np.uint32))
    assert np.array_equal(two["pho"], pho.view(np.uint8)) and np.array_equal(two["count"], count)
    assert two["total"] == np.float32(cx.totalPhotonSum)
