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
