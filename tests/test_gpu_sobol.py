"""TRC_FLAG_SOBOL on the GPU against the oracle: XSampler = pbrt::SobolSampler as the reference's
commented-out lines wire it (Render.metal:529-530, SobolSampler.hh:26-167) -- whole frames and RNG texels bit for bit."""
import numpy as np
import pytest

from oracle import pyoracle
from tracer_amd import abi, host
from tracer_amd.device import TracerError

pytestmark = pytest.mark.gpu


def both(gpu, sc, W, H, spp, integrator, seed=5, frame0=0, view_height=0, launches=1, max_depth=8):
    cam = host.prepare_camera(W, view_height or H)
    gpu.upload_scene(sc.view); gpu.set_camera(cam); gpu.resize(W, H)
    rng = host.fill_rng(seed, W, H)
    gpu.upload_rng(rng); gpu.clear_accum(); gpu.reset_stats()
    for i in range(launches):                    # spp split over several launches: same frames, same result
        gpu.render(spp=spp // launches, integrator=integrator, frame0=frame0 + i * (spp // launches), sobol=True,
                   view_height=view_height, max_depth=max_depth)
    got, got_rng, st = gpu.download_accum(), gpu.download_rng(), gpu.stats()
    ref, ref_st = pyoracle.render(sc.view, cam, W, H, rng, spp=spp, integrator=integrator, frame0=frame0, sobol=True,
                                  view_height=view_height, max_depth=max_depth)
    assert (got.view(np.uint32) == ref.view(np.uint32)).all()
    assert (got_rng == rng).all()
    assert st.rays == ref_st.rays and st.paths == ref_st.paths and st.shaded == ref_st.shaded
    return got


def test_path_and_mis_with_sobol_sampler(gpu):
    sc = host.HostScene(abi.SCENE_CORNELL_SPHERES)
    a = both(gpu, sc, 160, 96, 8, abi.INTEGRATOR_PATH)
    both(gpu, sc, 160, 96, 8, abi.INTEGRATOR_MIS)
    # a frame whose longest side is not a power of two and frames that do not start at 0, in two launches
    b = both(gpu, sc, 130, 75, 8, abi.INTEGRATOR_PATH, frame0=37, launches=2)
    assert np.isfinite(a).all() and np.isfinite(b).all()
    # not the random sampler's image
    gpu.resize(160, 96); gpu.set_camera(host.prepare_camera(160, 96))
    gpu.upload_rng(host.fill_rng(5, 160, 96)); gpu.clear_accum()
    gpu.render(spp=8)
    assert not (gpu.download_accum().view(np.uint32) == a.view(np.uint32)).all()


def test_sobol_on_a_mesh_scene_and_deep_paths(gpu):
    sc = host.HostScene(abi.SCENE_CORNELL_MESH, host.Mesh.ball(40, 40, 1.0))        # tree in global memory
    both(gpu, sc, 128, 96, 4, abi.INTEGRATOR_MIS)
    both(gpu, sc, 96, 64, 4, abi.INTEGRATOR_PATH, max_depth=20)                     # all 40 dimensions


def test_sobol_with_stacked_views(gpu):
    sc = host.HostScene(abi.SCENE_CORNELL_SPHERES)
    both(gpu, sc, 96, 3 * 64, 4, abi.INTEGRATOR_PATH, view_height=64)               # the sampler sees its own view


def test_sobol_degenerate_resolutions(gpu):
    """log2Resolution 0 (a 1 x 1 frame: SobolIntervalToIndex returns 0 for every frame, SobolSampler.hh:130) and 1"""
    sc = host.HostScene(abi.SCENE_CORNELL_SPHERES)
    both(gpu, sc, 1, 1, 16, abi.INTEGRATOR_PATH)
    both(gpu, sc, 2, 1, 16, abi.INTEGRATOR_MIS, frame0=3)
    both(gpu, sc, 1, 9, 8, abi.INTEGRATOR_PATH)


def test_sobol_limits(gpu):
    sc = host.HostScene(abi.SCENE_CORNELL_SPHERES)
    gpu.upload_scene(sc.view); gpu.set_camera(host.prepare_camera(64, 64)); gpu.resize(64, 64); gpu.seed(1)
    for kw in (dict(integrator=abi.INTEGRATOR_VOLUME), dict(collect_stats=True), dict(max_depth=21)):
        with pytest.raises(TracerError) as e:
            gpu.render(spp=1, sobol=True, **kw)
        assert e.value.status == abi.ERR_UNSUPPORTED
    gpu.render(spp=1, sobol=True)                                                   # still usable afterwards
    gpu.synchronize()


def test_committed_sobol_frames(gpu):
    """tests/golden/sobol_frames.npz (tests/golden/make_sobol_frames.py): accumulator, RNG texture and ray counts."""
    import os
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    import make_sobol_frames as fx
    frames = np.load(os.path.join(here, "golden", "sobol_frames.npz"))
    for name, (kind, integ, W, H, spp, seed, frame0) in fx.CASES.items():
        sc = host.HostScene(kind)
        gpu.upload_scene(sc.view); gpu.set_camera(host.prepare_camera(W, H)); gpu.set_environment((0.0, 0.0, 0.0)); gpu.resize(W, H)
        gpu.seed(seed); gpu.reset_stats()
        gpu.render(spp=spp, integrator=integ, frame0=frame0, sobol=True)
        assert np.array_equal(gpu.download_accum().view(np.uint32), frames[name + "_accum"].view(np.uint32)), name
        assert np.array_equal(gpu.download_rng(), frames[name + "_rng"]), name
        st = gpu.stats()
        assert [st.paths, st.rays, st.shaded] == list(frames[name + "_counts"]), name
