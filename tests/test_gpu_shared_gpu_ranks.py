"""Every N > 1 path executed with N > 1 on a ONE-GPU box: the ranks are processes sharing cuda:0, and the collectives of
trc_group_reduce_accum[_async], trc_group_allreduce_mean_accum and the grouped trc_sppm_frames go through the table of
trc_group_set_collectives (host-staged, gloo between the processes) instead of RCCL, which refuses a second rank on a
device.  Everything else is the code an 8-GPU run executes: tile ownership, photon index ranges, the reduce / allreduce /
allgather program, the pipelined two-accumulator compose.  Bar: N ranks == 1 rank, bit for bit -- at a small
size for 2 and 8 ranks, and at 1920x1080 for BASELINE configs 4 (teapot x 64, 2 spp + 8 spp) and 5 (4 SPPM frames).
Sample sharding (another sample set than one rank's by definition) is held to the ORACLE rendering the same definition
(oracle/pyoracle.py::render_sample_sharded): small, and at 1920x1080x64spp for BASELINE configs 2 and 4.
The reference has no counterpart (multi-device is commented out, AAPLRenderer.mm:139-146)."""
import hashlib
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from oracle import pyoracle as po
from tracer_amd import abi, host

pytestmark = pytest.mark.gpu
WORKER = os.path.join(ROOT, "tests", "_rank_worker.py")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_ranks(case, world, outdir, timeout=900):
    os.makedirs(outdir, exist_ok=True)
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), TRC_ROOT=ROOT, TRC_OUT=str(outdir), TRC_CASE=case, OMP_NUM_THREADS="4",
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, WORKER], env=env))
    try:
        for p in procs:
            assert p.wait(timeout=timeout) == 0, f"a rank of {case} x{world} failed"
    finally:
        for p in procs:                      # our own children, by PID
            if p.poll() is None:
                p.kill()
    return [np.load(os.path.join(outdir, f"rank{r}.npz")) for r in range(world)]


def owner_mask(W, H, world, rank):
    ty, tx = np.mgrid[0:H, 0:W] // abi.TRC_TILE
    return ((tx + ty) % world) == rank


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).view(np.uint8).tobytes()).hexdigest()


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.parametrize("world", [2, 8])
def test_every_group_call_with_n_ranks_on_one_gpu(gpu, tmp_path, world):
    res = run_ranks("small", world, tmp_path / f"small{world}")
    W, H, spp = 320, 192, 6
    scene = host.HostScene(abi.SCENE_CORNELL_SPHERES)
    gpu.upload_scene(scene.view); gpu.set_camera(host.prepare_camera(W, H)); gpu.set_environment((0.0, 0.0, 0.0)); gpu.resize(W, H)
    gpu.clear_accum(); gpu.seed(31); gpu.render(spp=spp)
    assert np.array_equal(bits(res[0]["sync"]), bits(gpu.download_accum()))
    for step in range(3):
        gpu.clear_accum(); gpu.seed(40 + step); gpu.render(spp=spp)
        assert np.array_equal(bits(res[0][f"async{step}"]), bits(gpu.download_accum())), step
    # sample sharding against the oracle's statement of the definition: group g's frame from seed shard_seed(seed, g),
    # rank-ordered float32 sum, one division -- bit for bit, on the root and (trc_group_allreduce_mean_accum) on every rank
    cam = host.prepare_camera(W, H)
    want, _ = po.render_sample_sharded(scene.view, cam, W, H, [host.fill_rng(po.shard_seed(100, g), W, H) for g in range(world)], 16)
    assert np.array_equal(bits(res[0]["samples"]), bits(want))
    for r in range(world):
        assert np.array_equal(bits(res[r]["mean"]), bits(want)) and bool(res[r]["untouched"]), r
        assert r == 0 or bool(res[r]["nonroot_refused"]), r
    # ... and against this GPU rendering the shards one after the other (the seeds are the documented function of the rank)
    shards = []
    for g in range(world):
        gpu.clear_accum(); gpu.seed(abi.shard_seed(100, g)); gpu.render(spp=16 // world)
        shards.append(gpu.download_accum())
    acc = shards[0]
    for a in shards[1:]:
        acc = np.add(acc, a, dtype=np.float32)
    assert np.array_equal(bits(np.divide(acc, np.float32(world), dtype=np.float32)), bits(want))
    if world >= 4:                                           # S seeds x 2 tile ranks: the zeros of the other tile rank are exact identities
        S = world // 2
        for step in range(2):
            want, _ = po.render_sample_sharded(scene.view, cam, W, H, [host.fill_rng(po.shard_seed(200 + step, g), W, H) for g in range(S)], 16)
            assert np.array_equal(bits(res[0][f"hybrid{step}"]), bits(want)), step
    # progressive + pipelined: samples 0..7 and then 8..9 of every group accumulated in place, composed after each launch
    for name, per_group in (("prog8", 8), ("prog10", 10)):
        want, _ = po.render_sample_sharded(scene.view, cam, W, H, [host.fill_rng(po.shard_seed(300, g), W, H) for g in range(world)], per_group * world)
        assert np.array_equal(bits(res[0][name]), bits(want)), name
    # SPPM
    gpu.clear_accum(); gpu.seed(8); gpu.sppm_init(9); gpu.sppm_frames(3)
    cam, pho, mark, count, cx = gpu.sppm_download()
    assert np.array_equal(bits(res[0]["sppm"]), bits(gpu.download_accum()))
    for r in range(world):
        own = owner_mask(W, H, world, r).ravel()
        assert np.array_equal(res[r]["cam_own"], cam[own].view(np.uint8)), r
        assert np.array_equal(res[r]["pho"], pho.view(np.uint8)) and np.array_equal(res[r]["count"], count)
        assert np.array_equal(bits(res[r]["mark"]), bits(mark))
        assert res[r]["total"] == np.float32(cx.totalPhotonSum)
        box = np.array([cx.photonBox.mini.x, cx.photonBox.mini.y, cx.photonBox.mini.z,
                        cx.photonBox.maxi.x, cx.photonBox.maxi.y, cx.photonBox.maxi.z], np.float32)
        assert np.array_equal(bits(res[r]["box"]), bits(box))
        # the collective program of SURVEY 8e: 1 + 3 + 1 reduces, 2 all-reduces (SPPM bound keys), 3 per-frame all-gathers of the
        # 40-byte wire records + the whole records at the download + the every-rank sample compose; one all-to-all per sample
        # compose (root, every rank, 2 pipelined hybrids), one gather per compose to a root
        hybrid = 2 if world >= 4 else 0
        assert list(res[r]["calls"]) == [5, 2, 5, 2 + hybrid + 3, 1 + hybrid + 3]                  # + the three progressive composes


@pytest.mark.parametrize("world", [2, 8])
def test_config4_as_an_n_rank_tile_split_at_1080p(gpu, tmp_path, world):
    res = run_ranks("config4", world, tmp_path / f"c4_{world}", timeout=1500)
    W, H = 1920, 1080
    scene = host.HostScene(abi.SCENE_CORNELL_MESH, host.Mesh.golden("teapot").replicate(8, 80.0))
    gpu.upload_scene(scene.view); gpu.set_camera(host.prepare_camera(W, H)); gpu.set_environment((0.0, 0.0, 0.0)); gpu.resize(W, H)
    gpu.reset_stats()
    gpu.clear_accum(); gpu.seed(0x5EED0004); gpu.render(spp=2)
    one = gpu.download_accum()
    st = gpu.stats()
    rng = gpu.download_rng()
    assert np.array_equal(bits(res[0]["frame"]), bits(one))
    assert sum(int(r["rays"]) for r in res) == st.rays and sum(int(r["paths"]) for r in res) == st.paths == W * H * 2
    for r in range(world):
        assert str(res[r]["rng_own_sha"]) == sha(rng[owner_mask(W, H, world, r)]), r
    gpu.clear_accum(); gpu.seed(0x5EED0005); gpu.render(spp=8)
    assert np.array_equal(bits(res[0]["frame8"]), bits(gpu.download_accum()))


@pytest.mark.parametrize("world", [2, 3, 8])         # 3: the 4 096 wavefronts of photons do not divide -- chunks of 1 366, 1 366, 1 364 wavefronts, padded all-gathers
def test_config5_sppm_as_an_n_rank_split_at_1080p(gpu, tmp_path, world):
    res = run_ranks("config5", world, tmp_path / f"c5_{world}", timeout=1500)
    W, H, frames = 1920, 1080, 4
    scene = host.HostScene(abi.SCENE_CORNELL_SPHERES)
    gpu.upload_scene(scene.view); gpu.set_camera(host.prepare_camera(W, H)); gpu.set_environment((0.0, 0.0, 0.0)); gpu.resize(W, H)
    gpu.clear_accum(); gpu.seed(0x5EED0050); gpu.sppm_init(0x5EED0051); gpu.sppm_frames(frames)
    cam, pho, mark, count, cx = gpu.sppm_download()
    rng = gpu.download_rng().reshape(-1, 4)
    assert np.array_equal(bits(res[0]["frame"]), bits(gpu.download_accum()))
    for r in range(world):
        own = owner_mask(W, H, world, r).ravel()
        assert str(res[r]["cam_own_sha"]) == sha(cam[own]), r
        assert str(res[r]["rng_own_sha"]) == sha(rng[own]), r
        assert str(res[r]["pho_sha"]) == sha(pho) and str(res[r]["mark_sha"]) == sha(mark) and str(res[r]["count_sha"]) == sha(count), r
        assert res[r]["total"] == np.float32(cx.totalPhotonSum) and res[r]["hash_scale"] == np.float32(cx.photonHashScale)
        assert list(res[r]["calls"]) == [1, 2, frames + 1, 0, 0]


@pytest.mark.parametrize("world", [5, 8])
def test_sppm_with_more_ranks_than_tiles_record_for_record(gpu, tmp_path, world):
    """A 40 x 24 canvas is 3 x 2 tiles: with 8 ranks four of them own none, with 5 one does and the 4 096 wavefronts of photons do not divide either
    (chunks of 820, ..., 816).  Every rank's camera records of its own pixels, all photon records (gathered at the download through the
    padded buffers), both hash grids, the photon sum, the radius and the composed frame equal the one-rank pass (round 5: a rank without a
    tile used to leave trc_sppm_frames before its collectives, and uneven photon splits were refused)."""
    res = run_ranks("sppm_small", world, tmp_path / f"ss{world}")
    W, H, frames = 40, 24, 4
    scene = host.HostScene(abi.SCENE_CORNELL_SPHERES)
    gpu.upload_scene(scene.view); gpu.set_camera(host.prepare_camera(W, H)); gpu.set_environment((0.0, 0.0, 0.0)); gpu.resize(W, H)
    gpu.clear_accum(); gpu.seed(8); gpu.sppm_init(9); gpu.sppm_frames(frames)
    cam, pho, mark, count, cx = gpu.sppm_download()
    assert np.array_equal(bits(res[0]["sppm"]), bits(gpu.download_accum()))
    empty = 0
    for r in range(world):
        own = owner_mask(W, H, world, r).ravel()
        empty += not own.any()
        assert np.array_equal(res[r]["cam_own"], cam[own].view(np.uint8)), r
        assert np.array_equal(res[r]["pho"], pho.view(np.uint8)) and np.array_equal(res[r]["count"], count), r
        assert np.array_equal(bits(res[r]["mark"]), bits(mark)), r
        assert res[r]["total"] == np.float32(cx.totalPhotonSum) and res[r]["radius"] == np.float32(cx.photonInitialRadius), r
    assert empty == (4 if world == 8 else 1)          # (tx + ty) % world over 3 x 2 tiles


@pytest.mark.parametrize("world", [2, 8])
def test_sample_shards_of_a_frame_the_ranks_do_not_divide(gpu, tmp_path, world):
    """97 x 61 = 5917 pixels over 2 / 8 ranks: the last pixel slice is short (and with 8 ranks three pixels short of the others);
    root, pipelined and every-rank compose == the oracle.  A table without alltoall / gather composes tiles and refuses sample
    shards with TRC_ERR_UNSUPPORTED instead of calling a null pointer."""
    res = run_ranks("ragged", world, tmp_path / f"ragged{world}")
    W, H = 97, 61
    scene = host.HostScene(abi.SCENE_CORNELL_SPHERES)
    cam = host.prepare_camera(W, H)
    want, _ = po.render_sample_sharded(scene.view, cam, W, H, [host.fill_rng(po.shard_seed(5, g), W, H) for g in range(world)], 16, env=(0.1, 0.2, 0.3))
    assert np.array_equal(bits(res[0]["sync"]), bits(want)) and np.array_equal(bits(res[0]["async"]), bits(want))
    for r in range(world):
        assert np.array_equal(bits(res[r]["mean"]), bits(want)), r
        assert int(res[r]["refused"]) == abi.ERR_UNSUPPORTED, r
    tiles, _ = po.render(scene.view, cam, W, H, host.fill_rng(6, W, H), spp=2, env=(0.1, 0.2, 0.3))
    assert np.array_equal(bits(res[0]["tiles"]), bits(tiles))


@pytest.mark.parametrize("world", [2, 8])
def test_config2_sample_sharded_at_1080p_is_the_oracle_s_frame(gpu, tmp_path, world):
    """bench.py --scaling samples, step for step: N ranks, each the WHOLE 1920x1080 frame with 64 / N samples from seed
    trc_shard_seed(0x5EED0000, rank), composed on rank 0 == the oracle rendering that definition, every pixel"""
    res = run_ranks("samples2", world, tmp_path / f"s2_{world}", timeout=1500)
    W, H = 1920, 1080
    scene = host.HostScene(abi.SCENE_CORNELL_SPHERES)
    want, sts = po.render_sample_sharded(scene.view, host.prepare_camera(W, H), W, H,
                                         [host.fill_rng(po.shard_seed(0x5EED0000, g), W, H) for g in range(world)], 64)
    assert np.array_equal(bits(res[0]["frame"]), bits(want))
    for r in range(world):
        assert int(res[r]["rays"]) == sts[r].rays and int(res[r]["paths"]) == W * H * (64 // world), r
        assert list(res[r]["calls"]) == [0, 0, 0, 2, 2]
    assert (want[..., 3] == 1).all()


@pytest.mark.parametrize("world,case", [(2, "samples4"), (8, "samples4"), (8, "samples3")])
def test_mesh_configs_sample_sharded_at_1080p_are_the_oracle_s_frame(gpu, tmp_path, world, case):
    """the same on BASELINE config 4's scene (teapot.obj x 64, 1 005 056 triangles, the tree read from memory; tracePath) and config 3's
    (coatball.obj, traceMIS): 64 / N samples per rank through the persistent-workgroup kernels; 1 tile in 64 of the composed frame
    re-rendered by the oracle"""
    res = run_ranks(case, world, tmp_path / f"{case}_{world}", timeout=1500)
    W, H = 1920, 1080
    if case == "samples4":
        scene, seed, integ = host.HostScene(abi.SCENE_CORNELL_MESH, host.Mesh.golden("teapot").replicate(8, 80.0)), 0x5EED0004, abi.INTEGRATOR_PATH
    else:
        scene, seed, integ = host.HostScene(abi.SCENE_CORNELL_MESH, host.Mesh.golden("coatball")), 0x5EED0003, abi.INTEGRATOR_MIS
    want, _ = po.render_sample_sharded(scene.view, host.prepare_camera(W, H), W, H,
                                       [host.fill_rng(po.shard_seed(seed, g), W, H) for g in range(world)], 64,
                                       tile_rank=0, tile_nranks=64, integrator=integ)
    mine = owner_mask(W, H, 64, 0)
    got = res[0]["frame"]
    assert mine.sum() > 5000 and np.array_equal(bits(got[mine]), bits(want[mine]))
    assert np.isfinite(got).all() and (got[..., 3] == 1).all() and sum(int(r["paths"]) for r in res) == W * H * 64


def test_device_pointer_table_and_error_propagation(gpu):
    """host_staged = 0: the table's functions get the DEVICE buffer and the stream (what a GPU-aware transport would bind).
    One rank, so every collective is the identity: the composed frame is the rendered one and the grouped SPPM pass the
    ungrouped one -- and the arguments are what the header promises.  Then a function that fails: the call returns
    TRC_ERR_RCCL with the function's code in the text, and a grouped SPPM frame that fails half-way asks for trc_sppm_init."""
    import ctypes as C
    from tracer_amd import device
    from tracer_amd.gloo_collectives import ALLGATHER_FN, ALLREDUCE_FN, REDUCE_FN, Collectives
    W, H, spp = 160, 96, 8
    scene = host.HostScene(abi.SCENE_CORNELL_SPHERES)
    seen = []

    class Table:
        def __init__(self, fail_allgather=0):
            self.cb = (REDUCE_FN(lambda user, buf, count, dtype, op, root, stream: seen.append(("reduce", bool(buf), count, dtype, op, root, bool(stream))) or 0),
                       ALLREDUCE_FN(lambda user, buf, count, dtype, op, stream: seen.append(("allreduce", bool(buf), count, dtype, op, bool(stream))) or 0),
                       ALLGATHER_FN(lambda user, buf, per_rank, stream: seen.append(("allgather", bool(buf), per_rank, bool(stream))) or fail_allgather))
            self.table = Collectives(None, 0, 0, *self.cb)               # host_staged = 0

    t = device.Tracer(0)
    try:
        t.upload_scene(scene.view); t.set_camera(host.prepare_camera(W, H)); t.set_environment((0.0, 0.0, 0.0)); t.resize(W, H)
        t.clear_accum(); t.seed(3); t.render(spp=spp)
        alone = t.download_accum()
        t.set_collectives(Table(), 1, 0)
        t.clear_accum(); t.seed(3); t.render(spp=spp); t.group_reduce_accum(0)
        assert np.array_equal(bits(t.download_accum()), bits(alone))
        t.clear_accum(); t.seed(3); t.render(spp=spp); t.group_reduce_accum_async(0)
        assert np.array_equal(bits(t.download_composed()), bits(alone))
        t.synchronize(); t.clear_accum(); t.seed(8); t.sppm_init(9); t.sppm_frames(2)
        grouped = t.download_accum()
        t.group_finalize()
        t.clear_accum(); t.seed(8); t.sppm_init(9); t.sppm_frames(2)
        assert np.array_equal(bits(t.download_accum()), bits(grouped))
        n = W * H * 4
        assert seen[:2] == [("reduce", True, n, abi.DT_F32, abi.OP_SUM, 0, True)] * 2            # device pointer + stream
        assert seen[2:4] == [("allreduce", True, 3, abi.DT_U32, abi.OP_MIN, True), ("allreduce", True, 3, abi.DT_U32, abi.OP_MAX, True)]
        assert seen[4:] == [("allgather", True, abi.PHOTON_HASHN * abi.PHOTON_HASHN * 40, True)] * 2       # 40 of a photon's 80 bytes cross the links per frame
        # a transport that fails
        t.set_collectives(Table(fail_allgather=7), 1, 0)
        t.clear_accum(); t.seed(8); t.sppm_init(9)
        with pytest.raises(device.TracerError) as e:
            t.sppm_frames(1)
        assert e.value.status == abi.ERR_RCCL and "returned 7" in str(e.value)
        with pytest.raises(device.TracerError) as e:
            t.sppm_frames(1)
        assert "trc_sppm_init" in str(e.value)
        t.group_finalize()
        t.clear_accum(); t.seed(8); t.sppm_init(9); t.sppm_frames(2)                            # a fresh pass is fine again
        assert np.array_equal(bits(t.download_accum()), bits(grouped))
        with pytest.raises(device.TracerError):                                                   # no group: the group calls say so
            t.group_reduce_accum(0)
    finally:
        t.close()
