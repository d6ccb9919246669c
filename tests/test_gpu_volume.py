"""traceVolume on the GPU (integrator 2) against the oracle: homogeneous medium inside the glass mesh / cube
(material 19) and the GridDensity cloud container, whole frames bit for bit."""
import numpy as np
import pytest

from oracle import pyoracle
from tracer_amd import abi, host

pytestmark = pytest.mark.gpu


def both(gpu, sc, W, H, spp, seed=21, density=None):
    cam = host.prepare_camera(W, H)
    gpu.upload_scene(sc.view); gpu.set_camera(cam); gpu.resize(W, H)
    info = host.density_info(density) if density is not None else None
    gpu.upload_density(info, density)
    pyoracle.set_density(info, density)
    try:
        rng = host.fill_rng(seed, W, H)
        gpu.upload_rng(rng); gpu.clear_accum(); gpu.reset_stats()
        gpu.render(spp=spp, integrator=abi.INTEGRATOR_VOLUME, collect_stats=True)
        got, got_rng, st = gpu.download_accum(), gpu.download_rng(), gpu.stats()
        ref, ref_st = pyoracle.render(sc.view, cam, W, H, rng, spp=spp, integrator=abi.INTEGRATOR_VOLUME)
        assert (got.view(np.uint32) == ref.view(np.uint32)).all()
        assert (got_rng == rng).all()
        assert st.rays == ref_st.rays and st.n_descend == ref_st.n_descend and st.shaded == ref_st.shaded
        # production (non-instrumented) kernel, same frame
        gpu.upload_rng(host.fill_rng(seed, W, H)); gpu.clear_accum()
        gpu.render(spp=spp, integrator=abi.INTEGRATOR_VOLUME)
        assert (gpu.download_accum().view(np.uint32) == ref.view(np.uint32)).all()
        return got
    finally:
        pyoracle.set_density(None, None)
        gpu.upload_density(None, None)


def test_homogeneous_medium_in_mesh_and_cube(gpu):
    sc = host.HostScene(abi.SCENE_CORNELL_MESH, host.Mesh.ball(40, 40, 1.0))
    both(gpu, sc, 160, 96, 8)
    both(gpu, host.HostScene(abi.SCENE_CORNELL_SPHERES), 160, 96, 8)        # tree entirely in LDS


def test_grid_density_cloud(gpu):
    cloud = host.make_cloud()
    sc = host.HostScene(abi.SCENE_CORNELL_VOLUME)
    with_cloud = both(gpu, sc, 192, 128, 8, density=cloud)
    without = both(gpu, sc, 192, 128, 8, density=None)
    assert not (with_cloud.view(np.uint32) == without.view(np.uint32)).all()
    # cloud container + glass mesh with the homogeneous medium in one scene
    both(gpu, host.HostScene(abi.SCENE_CORNELL_VOLUME, host.Mesh.ball(30, 30, 1.0)), 128, 96, 4, density=cloud)


def test_volume_requires_lights_and_valid_grid(gpu):
    from tracer_amd.device import TracerError
    sc = host.HostScene(abi.SCENE_CORNELL_VOLUME)
    gpu.upload_scene(sc.view)
    info = abi.GridDensityInfo()
    with pytest.raises(TracerError):
        gpu.upload_density(info, np.zeros((1, 1, 1), np.float32))            # nx = ny = nz = 0
