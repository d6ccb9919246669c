"""A pbrt-v3 scene file end to end on the GPU: trc_host_scene_load_pbrt -> trc_upload_scene -> frames of tracePath and
traceMIS bit-equal to the oracle's render of the same loaded scene, and examples/trc_render --pbrt writes the same PNG."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from oracle import pyoracle as po
from tracer_amd import abi, host
from test_pbrt_scene import CORNELL

pytestmark = pytest.mark.gpu
EXE = os.path.join(ROOT, "examples", "trc_render")


@pytest.fixture()
def cornell_pbrt(tmp_path):
    p = tmp_path / "cornell.pbrt"
    p.write_text(CORNELL)
    return str(p)


@pytest.mark.parametrize("integrator", [abi.INTEGRATOR_PATH, abi.INTEGRATOR_MIS])
def test_pbrt_scene_frames_bit_exact(gpu, cornell_pbrt, integrator):
    scene, cam, info, shapes = host.HostScene.from_pbrt(cornell_pbrt)
    W, H, spp = info.xres, info.yres, 8
    gpu.upload_scene(scene.view); gpu.set_camera(cam); gpu.set_environment((0.0, 0.0, 0.0)); gpu.resize(W, H)
    gpu.seed(77); gpu.clear_accum(); gpu.reset_stats()
    gpu.render(spp=spp, integrator=integrator)
    dev, dev_rng, st = gpu.download_accum(), gpu.download_rng(), gpu.stats()
    rng = host.fill_rng(77, W, H)
    ref, rst = po.render(scene.view, cam, W, H, rng, spp=spp, integrator=integrator)
    assert st.rays == rst.rays and ref[..., :3].max() > 0.5
    assert np.array_equal(dev.view(np.uint32), ref.view(np.uint32)) and np.array_equal(dev_rng, rng)


def test_example_renders_a_pbrt_file(gpu, cornell_pbrt, tmp_path):
    from PIL import Image
    out = tmp_path / "frame.png"
    log = subprocess.check_output([EXE, "--pbrt", cornell_pbrt, "--integrator", "mis", "--spp", "8", "--out", str(out)],
                                  text=True, stderr=subprocess.STDOUT)
    assert "160x120" in log and "10 shapes (1 not handled)" in log
    got = np.asarray(Image.open(out).convert("RGBA"))
    scene, cam, info, _ = host.HostScene.from_pbrt(cornell_pbrt)
    gpu.upload_scene(scene.view); gpu.set_camera(cam); gpu.set_environment((0.0, 0.0, 0.0)); gpu.resize(info.xres, info.yres)
    gpu.seed(0x5EED0000); gpu.clear_accum()
    gpu.render(spp=8, integrator=abi.INTEGRATOR_MIS)
    want, _ = gpu.tonemap()
    assert got.shape == want.shape and (got == want).all()
