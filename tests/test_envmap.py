"""Environment map (texHDR, Render.hh:25,42-48): SampleSphericalMap + bilinear lookup on a miss."""
import numpy as np
import pytest

from oracle import pyoracle
from tracer_amd import abi, host

W, H = 64, 40


def test_constant_map_matches_constant_environment(cornell):
    cam = host.prepare_camera(W, H)
    sky = np.array([0.5, 0.7, 1.0], np.float32)
    a, _ = pyoracle.render(cornell.view, cam, W, H, host.fill_rng(2, W, H), spp=4, env=tuple(sky))
    pyoracle.set_environment_map(np.tile(sky, (8, 16, 1)).astype(np.float32))
    try:
        b, _ = pyoracle.render(cornell.view, cam, W, H, host.fill_rng(2, W, H), spp=4)
    finally:
        pyoracle.set_environment_map(None)
    assert np.abs(a - b).max() < 1e-5 and a[..., :3].max() > 0.4          # same image up to the lerp's rounding
    # a map with structure changes the picture and stays inside the map's range where only the sky is seen
    rs = np.random.RandomState(4)
    env = rs.uniform(0.2, 0.9, size=(32, 64, 3)).astype(np.float32)
    pyoracle.set_environment_map(env)
    try:
        c, _ = pyoracle.render(cornell.view, cam, W, H, host.fill_rng(2, W, H), spp=4)
        d, _ = pyoracle.render(cornell.view, cam, W, H, host.fill_rng(2, W, H), spp=4, integrator=abi.INTEGRATOR_MIS)
    finally:
        pyoracle.set_environment_map(None)
    assert np.isfinite(c).all() and np.isfinite(d).all() and not np.array_equal(c, b)
    corner = c[0, 0, :3]                                               # a pixel that looks past the box
    assert (corner >= 0.2 - 1e-6).all() and (corner <= 0.9 + 1e-6).all()


@pytest.mark.gpu
def test_device_environment_map_equals_oracle(gpu, cornell, cornell_spheres):
    rs = np.random.RandomState(9)
    env = rs.uniform(0.0, 2.0, size=(17, 33, 3)).astype(np.float32)    # odd sizes: clamp-to-edge at the poles / seam
    cam = host.prepare_camera(W, H)
    pyoracle.set_environment_map(env)
    try:
        for sc in (cornell, cornell_spheres):
            gpu.upload_scene(sc.view); gpu.set_camera(cam); gpu.set_environment((0.1, 0.1, 0.1)); gpu.resize(W, H)
            gpu.set_environment_map(env)
            for integ in (abi.INTEGRATOR_PATH, abi.INTEGRATOR_MIS, abi.INTEGRATOR_VOLUME):
                rng = host.fill_rng(6, W, H)
                gpu.upload_rng(rng); gpu.clear_accum(); gpu.render(spp=4, integrator=integ)
                ref, _ = pyoracle.render(sc.view, cam, W, H, rng, spp=4, integrator=integ, env=(0.1, 0.1, 0.1))
                assert np.array_equal(gpu.download_accum().view(np.uint32), ref.view(np.uint32)), integ
        # SPPM's camera pass looks the map up too (Photon.metal:29-33)
        gpu.seed(1); gpu.clear_accum(); gpu.sppm_init(2); gpu.sppm_frames(2)
        rng = host.fill_rng(1, W, H); acc = np.zeros((H, W, 4), np.float32)
        s = pyoracle.Sppm(W, H, 2); s.frames(cornell_spheres.view, cam, rng, acc, 2, env=(0.1, 0.1, 0.1))
        assert np.array_equal(gpu.download_accum().view(np.uint32), acc.view(np.uint32))
        # clearing the map returns to the constant environment
        gpu.set_environment_map(None)
        pyoracle.set_environment_map(None)
        rng = host.fill_rng(6, W, H)
        gpu.upload_rng(rng); gpu.clear_accum(); gpu.render(spp=2)
        ref, _ = pyoracle.render(cornell_spheres.view, cam, W, H, rng, spp=2, env=(0.1, 0.1, 0.1))
        assert np.array_equal(gpu.download_accum().view(np.uint32), ref.view(np.uint32))
    finally:
        pyoracle.set_environment_map(None)
        gpu.set_environment_map(None)
        gpu.set_environment((0.0, 0.0, 0.0))
