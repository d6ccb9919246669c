"""kernelPathTracing on the CPU oracle: regression fixtures + the properties the GPU path relies on."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT
from oracle import pyoracle as po
from tracer_amd import abi, host

sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_frame_fixtures as fx  # noqa: E402

FRAMES = np.load(os.path.join(ROOT, "tests", "golden", "frames.npz"))


@pytest.mark.parametrize("name", sorted(fx.CASES))
def test_oracle_reproduces_committed_frames(name):
    acc, rng, st = fx.render_case(name)
    assert np.array_equal(rng, FRAMES[name + "_rng"])
    assert np.array_equal(acc.view(np.uint32), FRAMES[name + "_accum"].view(np.uint32))
    counts = [st.paths, st.rays, st.shaded, st.n_descend, st.n_return, st.n_leaf_sphere, st.n_leaf_square, st.n_leaf_cube]
    assert counts == list(FRAMES[name + "_counts"])


def _setup(W=64, H=40, kind=abi.SCENE_CORNELL_SPHERES):
    return host.HostScene(kind), host.prepare_camera(W, H), W, H


def test_result_is_independent_of_the_thread_count():
    scene, cam, W, H = _setup()
    a, sa = po.render(scene.view, cam, W, H, host.fill_rng(7, W, H), spp=4, n_threads=1)
    b, sb = po.render(scene.view, cam, W, H, host.fill_rng(7, W, H), spp=4, n_threads=7)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and sa.rays == sb.rays and sa.n_descend == sb.n_descend


def test_fused_spp_equals_successive_frames():
    """spp samples in one call == spp calls with frame0 = 0..spp-1: the running mean (Render.metal:540-541)
    and the RNG word swap between frames (B-1) are applied per sample either way."""
    scene, cam, W, H = _setup()
    rng1 = host.fill_rng(11, W, H)
    fused, _ = po.render(scene.view, cam, W, H, rng1, spp=5)
    rng2 = host.fill_rng(11, W, H)
    acc = np.zeros((H, W, 4), np.float32)
    for f in range(5):
        acc, _ = po.render(scene.view, cam, W, H, rng2, accum=acc, spp=1, frame0=f)
    assert np.array_equal(rng1, rng2) and np.array_equal(acc.view(np.uint32), fused.view(np.uint32))


def test_rng_words_trade_roles_every_frame():
    """B-1: the texel is read as state=(b,a), inc=(r,g) and written back as (r,g)=state, (b,a)=inc."""
    scene, cam, W, H = _setup(16, 16)
    rng = host.fill_rng(3, W, H)
    before = rng.copy()
    po.render(scene.view, cam, W, H, rng, spp=1)
    # after one frame the NEW (b,a) words are the OLD (r,g) words (the old inc, untouched by stepping)
    assert np.array_equal(rng[..., 2:4], before[..., 0:2])
    assert not np.array_equal(rng[..., 0:2], before[..., 2:4])       # state advanced


@pytest.mark.parametrize("nranks", [2, 5])
def test_tile_shards_sum_to_the_full_frame(nranks):
    scene, cam, W, H = _setup(80, 48)
    full, st_full = po.render(scene.view, cam, W, H, host.fill_rng(5, W, H), spp=3, env=(0.2, 0.2, 0.2))
    total = np.zeros_like(full)
    rays = 0
    for r in range(nranks):
        part, st = po.render(scene.view, cam, W, H, host.fill_rng(5, W, H), spp=3, env=(0.2, 0.2, 0.2),
                             tile_rank=r, tile_nranks=nranks)
        total += part
        rays += st.rays
    assert np.array_equal(total.view(np.uint32), full.view(np.uint32)) and rays == st_full.rays


def test_no_pixel_jitter_primary_ray_is_shared():
    """B-2: u = x/W, v = y/H without jitter and aperture 0: every sample of a pixel starts with the same
    primary ray, so pixels that look straight at an emitter have zero variance."""
    scene, cam, W, H = _setup(96, 54, abi.SCENE_CORNELL)
    a, _ = po.render(scene.view, cam, W, H, host.fill_rng(1, W, H), spp=1)
    b, _ = po.render(scene.view, cam, W, H, host.fill_rng(2, W, H), spp=1)
    on_light = (a[..., 0] > 1) & (b[..., 0] > 1) & (a[..., 0] == a[..., 1])   # Le*|cos| (grey), seen at a grazing angle
    assert on_light.sum() > 10 and np.array_equal(a[on_light], b[on_light])


def test_nan_and_inf_samples_are_scrubbed():
    scene, cam, W, H = _setup(48, 32)
    acc, _ = po.render(scene.view, cam, W, H, host.fill_rng(9, W, H), spp=16)
    assert np.isfinite(acc).all() and (acc[..., 3] == 1).all() and (acc[..., :3] >= 0).all()


def test_libm_variant_agrees_statistically():
    """trc_detmath.h vs glibc libm: individual samples diverge (chaotic paths), image statistics do not."""
    scene, cam, W, H = _setup(96, 54)
    a, sa = po.render(scene.view, cam, W, H, host.fill_rng(21, W, H), spp=32)
    b, sb = po.render(scene.view, cam, W, H, host.fill_rng(21, W, H), spp=32, libm=True)
    assert abs(sa.rays - sb.rays) / sa.rays < 0.01
    ma, mb = a[..., :3].mean(axis=(0, 1)), b[..., :3].mean(axis=(0, 1))
    assert np.allclose(ma, mb, rtol=0.15), (ma, mb)
    # most pixels are literally identical (same decisions, libm differs only in last bits)
    close = np.isclose(a, b, rtol=1e-4, atol=1e-6).all(axis=2).mean()
    assert close > 0.5


def test_path_and_mis_integrators_see_the_same_scene():
    scene, cam, W, H = _setup(64, 36)
    p, sp = po.render(scene.view, cam, W, H, host.fill_rng(4, W, H), spp=32, integrator=abi.INTEGRATOR_PATH)
    m, sm = po.render(scene.view, cam, W, H, host.fill_rng(4, W, H), spp=32, integrator=abi.INTEGRATOR_MIS)
    assert sm.rays > sp.rays                       # + one shadow ray per bounce
    assert m[..., :3].mean() > 0 and p[..., :3].mean() > 0
    # NEE removes the fireflies of pure BSDF sampling: far fewer black pixels
    assert (m[..., :3].sum(axis=2) == 0).mean() < (p[..., :3].sum(axis=2) == 0).mean()


def test_stacked_views_are_independent_frames(cornell_spheres):
    """trc_params.view_height: a frame that stacks k views of h rows equals k separate renders of a W x h frame, each
    with its slice of the RNG texture (the multi-GPU workload of bench.py)."""
    W, h, k = 40, 24, 3
    cam = host.prepare_camera(W, h)
    rng = host.fill_rng(1234, W, h * k)
    want_rng = rng.copy()
    stacked, st = po.render(cornell_spheres.view, cam, W, h * k, rng, spp=3, view_height=h)
    parts, rays = [], 0
    for i in range(k):
        r = np.ascontiguousarray(want_rng[i * h:(i + 1) * h])
        a, s = po.render(cornell_spheres.view, cam, W, h, r, spp=3)
        want_rng[i * h:(i + 1) * h] = r
        parts.append(a); rays += s.rays
    assert np.array_equal(stacked.view(np.uint32), np.concatenate(parts).view(np.uint32))
    assert np.array_equal(rng, want_rng) and st.rays == rays
    # the views differ from each other (different RNG texels), and view_height >= H means one view
    assert not np.array_equal(parts[0], parts[1])
    one, _ = po.render(cornell_spheres.view, cam, W, h, host.fill_rng(5, W, h), spp=2, view_height=h)
    two, _ = po.render(cornell_spheres.view, cam, W, h, host.fill_rng(5, W, h), spp=2)
    assert np.array_equal(one.view(np.uint32), two.view(np.uint32))
