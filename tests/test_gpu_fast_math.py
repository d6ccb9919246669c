"""libtracer_amd_fast.so: the same sources under fast-math rules (approximate division / sqrt, FMA contraction,
denormals flushed), as the reference compiles its own shaders (MTL_FAST_MATH).  It is NOT the parity build: path tracing
is chaotic, so its frames are compared with the exact build's the way two exact renders with different seeds compare --
statistically -- and its first hits must be the exact build's (north_star: bit-exact hit indices, a float tolerance on
radiance)."""
import numpy as np
import pytest

from conftest import camera_rays
from tracer_amd import abi, device, host

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fast():
    t = device.Tracer(0, fast_math=True)
    yield t
    t.close()


def test_flavours(gpu, fast):
    assert device.lib().trc_build_flavor() == b"exact" and device.lib(fast_math=True).trc_build_flavor() == b"fast-math"


@pytest.mark.parametrize("scene_name", ["cornell_spheres", "coatball"])
def test_first_hits_are_the_exact_builds(gpu, fast, request, scene_name):
    scene = host.HostScene(abi.SCENE_CORNELL_MESH, host.Mesh.golden("coatball")) if scene_name == "coatball" else request.getfixturevalue(scene_name)
    W, H = 960, 540
    rays = camera_rays(host.prepare_camera(W, H), W, H)
    gpu.upload_scene(scene.view); fast.upload_scene(scene.view)
    a, b = gpu.trace_rays(rays, production=True), fast.trace_rays(rays, production=True)
    same = (a["hit"] == b["hit"]) & (a["pType"] == b["pType"]) & (a["pIndex"] == b["pIndex"])
    assert same.mean() > 0.9999, f"{(~same).sum()} of {len(rays)} primary rays hit something else"
    h = same & (a["hit"] != 0)
    rel = np.abs(a["t"][h] - b["t"][h]) / a["t"][h]
    # stated tolerance: t within 1e-3 relative (median < 1e-6), the hit point within 3e-4 of the distance travelled
    assert rel.max() < 1e-3 and np.median(rel) < 1e-6
    assert (np.abs(a["p"][h] - b["p"][h]).max(axis=1) <= 3e-4 * np.maximum(a["t"][h], 100.0)).all()


def test_frames_agree_like_two_exact_renders(gpu, fast, cornell_spheres):
    W, H, spp = 480, 270, 64
    cam = host.prepare_camera(W, H)
    lum = lambda f: f[..., 0] * 0.2126 + f[..., 1] * 0.7152 + f[..., 2] * 0.0722

    def render(t, seed):
        t.upload_scene(cornell_spheres.view); t.set_camera(cam); t.set_environment((0.0, 0.0, 0.0)); t.resize(W, H)
        t.seed(seed); t.clear_accum(); t.reset_stats(); t.render(spp=spp)
        return lum(t.download_accum().astype(np.float64)), t.stats().rays
    exact_a, rays_a = render(gpu, 11)
    exact_b, _ = render(gpu, 12)
    fast_a, rays_f = render(fast, 11)
    assert np.isfinite(fast_a).all() and abs(rays_f - rays_a) < 0.01 * rays_a
    clip = lambda x: np.minimum(x, 4.0)                     # fireflies of single pixels would dominate an RMSE
    noise = np.sqrt(np.mean((clip(exact_a) - clip(exact_b)) ** 2))
    diff = np.sqrt(np.mean((clip(fast_a) - clip(exact_a)) ** 2))
    assert diff < 1.15 * noise, (diff, noise)               # indistinguishable from a re-seeded exact render
    assert abs(clip(fast_a).mean() - clip(exact_a).mean()) < 0.01 * clip(exact_a).mean()
    # 8x8-block means are far less noisy: they must agree much more tightly than single pixels do
    blk = lambda x: clip(x)[: H // 8 * 8, : W // 8 * 8].reshape(H // 8, 8, W // 8, 8).mean(axis=(1, 3))
    assert np.sqrt(np.mean((blk(fast_a) - blk(exact_a)) ** 2)) < 1.2 * np.sqrt(np.mean((blk(exact_a) - blk(exact_b)) ** 2))
