"""examples/trc_render: a C++ host that drives the path through the C ABI only (no Python in the loop)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from tracer_amd import abi, host

EXE = os.path.join(ROOT, "examples", "trc_render")


def test_example_is_built_against_the_abi_only():
    assert os.path.exists(EXE), "run `make example`"
    src = open(os.path.join(ROOT, "examples", "trc_render.cpp")).read()
    assert '#include "tracer_abi.h"' in src and "oracle" not in src and "Python.h" not in src
    needed = subprocess.check_output(["ldd", EXE], text=True)
    assert "libtracer_amd.so" in needed and "libtrc_host.so" in needed and "liboracle" not in needed


@pytest.mark.gpu
@pytest.mark.parametrize("args", [["--scene", "spheres", "--integrator", "path"],
                                  ["--scene", "spheres", "--integrator", "path", "--device-sah"],      # the same tree, built on the GPU from the analytic leaves
                                  ["--scene", "volume", "--integrator", "volume", "--lbvh"]])
def test_example_renders_the_same_png_as_the_python_mirror(gpu, tmp_path, args):
    from PIL import Image
    W, H, spp = 160, 96, 8
    out = tmp_path / "frame.png"
    log = subprocess.check_output([EXE, *args, "--size", str(W), str(H), "--spp", str(spp), "--out", str(out)], text=True)
    assert "Mrays/s" in log
    got = np.asarray(Image.open(out).convert("RGBA"))
    volume = "volume" in args
    sc = host.HostScene(abi.SCENE_CORNELL_VOLUME if volume else abi.SCENE_CORNELL_SPHERES)
    if volume:
        gpu.upload_scene_lbvh(sc.leaves_view())
        cloud = host.make_cloud()
        gpu.upload_density(host.density_info(cloud), cloud)
    else:
        gpu.upload_scene(sc.view)
    try:
        gpu.set_camera(host.prepare_camera(W, H)); gpu.set_environment((0.0, 0.0, 0.0)); gpu.resize(W, H)
        gpu.seed(0x5EED0000); gpu.clear_accum()
        gpu.render(spp=spp, integrator=abi.INTEGRATOR_VOLUME if volume else abi.INTEGRATOR_PATH)
        want, _ = gpu.tonemap()
    finally:
        gpu.upload_density(None, None)
    assert got.shape == want.shape and (got == want).all()


@pytest.mark.gpu
def test_example_ingests_pbrt_files(gpu, tmp_path):
    """--mesh scene.pbrt (trianglemeshes through the transformation stack) and --density medium.pbrt (the block the reference
    reads through minipbrt), against the Python mirror fed with the same files"""
    from PIL import Image
    W, H, spp = 128, 80, 4
    (tmp_path / "mesh.pbrt").write_text(
        'WorldBegin\nAttributeBegin\n  Translate 278 200 278\n  Rotate 30 0 1 0\n  Scale 120 120 120\n'
        '  Shape "trianglemesh" "integer indices" [0 1 2  0 2 3  0 3 1  1 3 2]\n'
        '        "point P" [ 0 1 0   -1 -1 1   1 -1 1   0 -1 -1 ]\nAttributeEnd\nWorldEnd\n')
    grid = (np.arange(6 * 5 * 4, dtype=np.float32) % 7 / 7.0).reshape(4, 5, 6)
    (tmp_path / "medium.pbrt").write_text('MakeNamedMedium "smoke" "string type" "heterogeneous" "integer nx" 6 "integer ny" 5 '
                                          '"integer nz" 4 "float density" [ ' + " ".join(repr(float(x)) for x in grid.ravel()) + " ]\n")
    out = tmp_path / "frame.png"
    subprocess.check_output([EXE, "--scene", "volume", "--integrator", "volume", "--mesh", str(tmp_path / "mesh.pbrt"),
                             "--density", str(tmp_path / "medium.pbrt"), "--size", str(W), str(H), "--spp", str(spp),
                             "--out", str(out)], text=True)
    got = np.asarray(Image.open(out).convert("RGBA"))
    mesh = host.Mesh.load_pbrt(str(tmp_path / "mesh.pbrt"))
    assert mesh.n_triangles == 4
    sc = host.HostScene(abi.SCENE_CORNELL_VOLUME, mesh)
    dens = host.load_density_pbrt(str(tmp_path / "medium.pbrt"))
    assert np.array_equal(dens, grid)
    gpu.upload_scene(sc.view)
    gpu.upload_density(host.density_info(dens), dens)
    try:
        gpu.set_camera(host.prepare_camera(W, H)); gpu.set_environment((0.0, 0.0, 0.0)); gpu.resize(W, H)
        gpu.seed(0x5EED0000); gpu.clear_accum()
        gpu.render(spp=spp, integrator=abi.INTEGRATOR_VOLUME)
        want, _ = gpu.tonemap()
    finally:
        gpu.upload_density(None, None)
    assert got.shape == want.shape and (got == want).all()


RANKS_EXE = os.path.join(ROOT, "examples", "trc_ranks")


def test_rank_example_needs_neither_python_nor_torch():
    assert os.path.exists(RANKS_EXE), "run `make example`"
    src = open(os.path.join(ROOT, "examples", "trc_ranks.cpp")).read()
    assert '#include "tracer_abi.h"' in src and "oracle" not in src and "Python.h" not in src and "torch" not in src.lower().replace("pytorch", "")
    needed = subprocess.check_output(["ldd", RANKS_EXE], text=True)
    assert "libtracer_amd.so" in needed and "liboracle" not in needed and "libtorch" not in needed and "libpython" not in needed


@pytest.mark.gpu
@pytest.mark.parametrize("ranks", [2, 8])
def test_ranks_started_and_composed_without_pytorch(gpu, tmp_path, ranks):
    """examples/trc_ranks: fork N ranks, rendezvous on a TCP port, the collectives of trc_group_set_collectives over the same
    sockets (N ranks on one GPU), tile-split frame + grouped SPPM composed on rank 0 == the 1-rank results (the program
    checks that itself and exits non-zero otherwise); the PNG is the one the Python mirror tonemaps from its own render."""
    from PIL import Image
    W, H, spp = 480, 272, 12
    out = tmp_path / "ranks.png"
    log = subprocess.run([RANKS_EXE, "--ranks", str(ranks), "--host-collectives", "--size", str(W), str(H), "--spp", str(spp),
                          "--sppm", "2", "--out", str(out)], text=True, capture_output=True, timeout=600)
    assert log.returncode == 0, log.stdout + log.stderr
    assert f"{ranks} ranks (collectives over TCP sockets" in log.stdout and "composed frame == the 1-rank frame" in log.stdout
    assert f"the {ranks}-rank frame == the 1-rank frame" in log.stdout
    got = np.asarray(Image.open(out).convert("RGBA"))
    sc = host.HostScene(abi.SCENE_CORNELL_SPHERES)
    gpu.upload_scene(sc.view); gpu.set_camera(host.prepare_camera(W, H)); gpu.set_environment((0.0, 0.0, 0.0)); gpu.resize(W, H)
    gpu.seed(0x5EED0000); gpu.clear_accum(); gpu.render(spp=spp)
    want, _ = gpu.tonemap()
    assert got.shape == want.shape and (got == want).all()


@pytest.mark.gpu
@pytest.mark.parametrize("ranks", [2, 4])
def test_ranks_sample_sharded_without_pytorch(gpu, tmp_path, ranks):
    """examples/trc_ranks --samples: the C++ host of the split that scales -- every rank the whole frame with spp / N samples from
    trc_shard_seed(seed, rank), trc_group_compose_samples over the socket table (all-to-all + gather in C++), and rank 0's own
    check against the shards folded in rank order on the host.  The ranks are NOT told to use host collectives: they find out
    from the PCI bus ids that they share one GPU."""
    log = subprocess.run([RANKS_EXE, "--ranks", str(ranks), "--samples", "--size", "480", "272", "--spp", "16", "--out", str(tmp_path / "s.png")],
                         text=True, capture_output=True, timeout=600)
    assert log.returncode == 0, log.stdout + log.stderr
    assert "ranks share a device" in log.stderr and "collectives over TCP sockets" in log.stdout
    assert "samples split over the ranks' seeds" in log.stdout and "composed frame == the rank-ordered mean of the shards" in log.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("ranks,W,H", [(4, 9, 31), (8, 4, 5), (4, 27, 21), (3, 200, 120), (7, 20, 12)])      # the last two: rank counts that do not divide the photons
def test_more_ranks_than_tiles(gpu, ranks, W, H):
    """A frame so small that some ranks own NO tile (one, two, four tiles of 16 x 16 for 4 / 8 / 4 ranks): such a rank still bounces its
    photon range and takes part in the all-reduce of the bound keys and the all-gather of the photon records.  Up to round 5 it returned
    from trc_sppm_frames at once and the others waited for it (found by tests/campaigns/fuzz_ranks.sh).  The program compares the composed frame
    and the SPPM frame with the 1-rank results itself."""
    log = subprocess.run([RANKS_EXE, "--ranks", str(ranks), "--host-collectives", "--size", str(W), str(H), "--spp", "6", "--sppm", "3",
                          "--out", "/dev/null"], text=True, capture_output=True, timeout=600)
    assert log.returncode == 0, log.stdout + log.stderr
    assert "composed frame == the 1-rank frame" in log.stdout and f"the {ranks}-rank frame == the 1-rank frame" in log.stdout


@pytest.mark.gpu
def test_rccl_unique_id_travels_over_the_socket(gpu):
    """the default path of the example with one rank: trc_group_unique_id -> trc_group_init -> ncclReduce, no Python involved"""
    log = subprocess.run([RANKS_EXE, "--ranks", "1", "--size", "320", "200", "--spp", "8", "--out", "/dev/null"], text=True,
                         capture_output=True, timeout=600)
    assert log.returncode == 0 and "1 ranks (RCCL)" in log.stdout and "composed frame == the 1-rank frame" in log.stdout, log.stdout + log.stderr
