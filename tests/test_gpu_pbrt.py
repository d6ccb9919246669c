"""A pbrt-v3 scene file end to end on the GPU: trc_host_scene_load_pbrt -> trc_upload_scene -> frames of tracePath and
traceMIS bit-equal to the oracle's render of the same loaded scene, and examples/trc_render --pbrt writes the same PNG."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from oracle import pyoracle as po
from tracer_amd import abi, host
from test_pbrt_scene import CORNELL, WEDGE_PLY

pytestmark = pytest.mark.gpu
EXE = os.path.join(ROOT, "examples", "trc_render")


@pytest.fixture()
def cornell_pbrt(tmp_path):
    p = tmp_path / "cornell.pbrt"
    p.write_text(CORNELL)
    (tmp_path / "wedge.ply").write_text(WEDGE_PLY)          # the scene's Shape "plymesh"
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
    # the frame shows what round 3's ingestion added: the checkerboard cylinder + disk, the PLY wedge (triangle hits with uv)
    assert rst.n_leaf_triangle > 0 and info.n_shapes == 14


CHECKERS = '''LookAt 0 3 -12  0 1 0  0 1 0
Camera "perspective" "float fov" [ 40 ]
Film "image" "integer xresolution" [ 128 ] "integer yresolution" [ 96 ]
WorldBegin
AttributeBegin
  AreaLightSource "diffuse" "rgb L" [ 9 9 8 ]
  Shape "trianglemesh" "integer indices" [ 0 1 2 0 2 3 ] "point P" [ -3 8 -3  3 8 -3  3 8 3  -3 8 3 ]
AttributeEnd
Material "matte" "rgb Kd" [ 0.6 0.6 0.6 ]
Shape "trianglemesh" "integer indices" [ 0 1 2 0 2 3 ] "point P" [ -20 0 -20  20 0 -20  20 0 20  -20 0 20 ]
Texture "tiles" "spectrum" "checkerboard" "rgb tex1" [ 0.9 0.5 0.1 ]
AttributeBegin
  Material "matte" "texture Kd" "tiles"
  Translate -2.5 1.5 0
  Shape "sphere" "float radius" 1.5
AttributeEnd
AttributeBegin
  Material "plastic" "texture Kd" "tiles"
  Translate 2.5 0 0
  Rotate -90 1 0 0
  Shape "cylinder" "float radius" 1.2 "float zmin" 0 "float zmax" 2.5
  Translate 0 0 2.5
  Shape "disk" "float radius" 1.2
AttributeEnd
WorldEnd
'''


@pytest.mark.parametrize("integrator", [abi.INTEGRATOR_PATH, abi.INTEGRATOR_MIS])
def test_checkerboard_on_a_sphere_and_on_tessellated_quadrics(gpu, tmp_path, integrator):
    """Texture "checkerboard" -> TextureInfo{Checker} (Texture.hh:24-28) on a sphere (uv from the hit normal) and on the
    triangles of a tessellated cylinder + disk (material 19, interpolated uv): frames bit-equal to the oracle"""
    p = tmp_path / "checkers.pbrt"
    p.write_text(CHECKERS)
    scene, cam, info, shapes = host.HostScene.from_pbrt(str(p))
    v = scene.view
    assert info.n_unsupported_shapes == 0 and info.mis_ready == 1 and v.materials[19].textureInfo.type == abi.TEX_CHECKER
    assert v.materials[v.sphereList[0].material].textureInfo.type == abi.TEX_CHECKER
    W, H, spp = info.xres, info.yres, 8
    gpu.upload_scene(v); gpu.set_camera(cam); gpu.set_environment((0.05, 0.05, 0.08)); gpu.resize(W, H)
    rng = host.fill_rng(21, W, H)
    gpu.upload_rng(rng); gpu.clear_accum(); gpu.reset_stats()
    gpu.render(spp=spp, integrator=integrator)
    ref, rst = po.render(v, cam, W, H, rng, spp=spp, integrator=integrator, env=(0.05, 0.05, 0.08))
    assert gpu.stats().rays == rst.rays and rst.n_leaf_triangle > 0 and rst.n_leaf_sphere > 0
    assert np.array_equal(gpu.download_accum().view(np.uint32), ref.view(np.uint32))


def test_example_renders_a_pbrt_file(gpu, cornell_pbrt, tmp_path):
    from PIL import Image
    out = tmp_path / "frame.png"
    log = subprocess.check_output([EXE, "--pbrt", cornell_pbrt, "--integrator", "mis", "--spp", "8", "--out", str(out)],
                                  text=True, stderr=subprocess.STDOUT)
    assert "160x120" in log and "14 shapes (1 not handled), 1 materials and 1 textures not handled" in log
    got = np.asarray(Image.open(out).convert("RGBA"))
    scene, cam, info, _ = host.HostScene.from_pbrt(cornell_pbrt)
    gpu.upload_scene(scene.view); gpu.set_camera(cam); gpu.set_environment((0.0, 0.0, 0.0)); gpu.resize(info.xres, info.yres)
    gpu.seed(0x5EED0000); gpu.clear_accum()
    gpu.render(spp=8, integrator=abi.INTEGRATOR_MIS)
    want, _ = gpu.tonemap()
    assert got.shape == want.shape and (got == want).all()
