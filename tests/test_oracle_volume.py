"""traceVolume (Render.metal:78-275) + Medium.hh in the oracle, and the host pieces around it."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import pyoracle
from tracer_amd import abi, host

W, H = 48, 36
REF_CLOUD = "/root/reference/RT_Metal/cloud/geometry/density_render.70.pbrt"


def render(view, integ, spp=4, seed=5, frame0=0, rng=None, accum=None):
    cam = host.prepare_camera(W, H)
    rng = host.fill_rng(seed, W, H) if rng is None else rng
    acc, st = pyoracle.render(view, cam, W, H, rng, accum=accum, spp=spp, integrator=integ, frame0=frame0)
    return acc, rng, st


def test_density_pbrt_reader(tmp_path):
    grid = (np.arange(2 * 3 * 4, dtype=np.float32) * 0.25).reshape(4, 3, 2)
    text = ('MakeNamedMedium "smoke" "string type" "heterogeneous" "integer nx" 2 "integer ny" [3] "integer nz" 4\n'
            '\t"point p0" [ 0 0 0 ] "point p1" [ 1 1 1 ]\n\t"float density" [\n' +
            " ".join(repr(float(v)) for v in grid.ravel()) + "\n]\n")
    f = tmp_path / "m.pbrt"
    f.write_text(text)
    got = host.load_density_pbrt(str(f))
    assert got.shape == (4, 3, 2) and (got == grid).all()
    info = host.density_info(got)
    assert info.sigma_t == 100.0 and info.nx == 2 and info.ny == 3 and info.nz == 4
    assert info.invMaxDensity == np.float32(1) / grid.max()
    (tmp_path / "bad.pbrt").write_text('MakeNamedMedium "x" "integer nx" 2 "float density" [ 1 2 ]')
    with pytest.raises(RuntimeError):
        host.load_density_pbrt(str(tmp_path / "bad.pbrt"))


@pytest.mark.skipif(not os.path.exists(REF_CLOUD), reason="reference cloud grid not present (GPU box)")
def test_reference_cloud_grid_loads():
    g = host.load_density_pbrt(REF_CLOUD)          # AAPLRenderer.mm:629-636 reads the same block through minipbrt
    assert g.shape == (40, 100, 100) and g.max() == 1.0 and g.min() == 0.0 and 0.2 < (g > 0).mean() < 0.4


@pytest.mark.skipif(not os.path.exists(REF_CLOUD), reason="reference cloud grid not present (GPU box)")
def test_trace_volume_through_the_reference_cloud():
    """the grid the reference itself renders (cloud/cloud.pbrt, read through its Include like AAPLRenderer.mm:629-636),
    GridDensityInfo(10, 90, 0.5, ...) as at :636, in the cloud container of the volume scene"""
    g = host.load_density_pbrt(os.path.join(os.path.dirname(os.path.dirname(REF_CLOUD)), "cloud.pbrt"))
    info = host.density_info(g)
    assert (info.sigma_a, info.sigma_s, info.g, info.invMaxDensity) == (10.0, 90.0, 0.5, 1.0)
    sc = host.HostScene(abi.SCENE_CORNELL_VOLUME)
    pyoracle.set_density(info, g)
    try:
        a, _, sa = render(sc.view, abi.INTEGRATOR_VOLUME, spp=8)
    finally:
        pyoracle.set_density(None, None)
    b, _, sb = render(sc.view, abi.INTEGRATOR_VOLUME, spp=8)
    assert np.isfinite(a).all() and (a[..., :3] >= 0).all()
    changed = (a.view(np.uint32) != b.view(np.uint32)).any(axis=2).mean()
    assert 0.05 < changed < 0.9 and sa.rays != sb.rays          # the cloud scatters light in part of the frame


def test_volume_equals_mis_without_media():
    """With every Material::medium = _NIL_ and no _NIL_-typed surface traceVolume executes exactly traceMIS."""
    sc = host.HostScene(abi.SCENE_CORNELL_SPHERES)
    mats = (abi.Material * sc.view.n_material)()
    C.memmove(mats, sc.view.materials, C.sizeof(mats))
    for m in mats:
        m.medium = abi.MEDIUM_NIL
    v = abi.Scene.from_buffer_copy(sc.view)
    v.materials = C.cast(mats, C.POINTER(abi.Material))
    a, ra, sa = render(v, abi.INTEGRATOR_MIS)
    b, rb, sb = render(v, abi.INTEGRATOR_VOLUME)
    assert (a.view(np.uint32) == b.view(np.uint32)).all() and (ra == rb).all() and sa.rays == sb.rays
    # the shipped materials differ: the small cube is glass filled with the homogeneous medium (material 19)
    c, _, _ = render(sc.view, abi.INTEGRATOR_VOLUME)
    assert not (a.view(np.uint32) == c.view(np.uint32)).all()


def test_grid_density_volume_scene():
    sc = host.HostScene(abi.SCENE_CORNELL_VOLUME)
    assert sc.n_leaves == 10 and sc.view.n_cube == 3
    cloud = host.make_cloud()
    assert cloud.shape == (40, 100, 100) and (cloud == host.make_cloud()).all() and (cloud == 0).mean() > 0.3
    info = host.density_info(cloud)
    pyoracle.set_density(info, cloud)
    try:
        a, ra, sa = render(sc.view, abi.INTEGRATOR_VOLUME, spp=8)
        assert np.isfinite(a).all() and (a[..., :3] >= 0).all() and (a[..., 3] == 1).all()
        # fused samples == one sample per call
        b = np.zeros_like(a); rb = host.fill_rng(5, W, H)
        for s in range(8):
            render(sc.view, abi.INTEGRATOR_VOLUME, spp=1, frame0=s, rng=rb, accum=b)
        assert (a.view(np.uint32) == b.view(np.uint32)).all() and (ra == rb).all()
        # the cloud matters: without the grid the container is empty space
        pyoracle.set_density(None, None)
        c, _, _ = render(sc.view, abi.INTEGRATOR_VOLUME, spp=8)
        assert not (a.view(np.uint32) == c.view(np.uint32)).all()
    finally:
        pyoracle.set_density(None, None)


def test_henyey_greenstein_against_the_published_phase_function():
    """HG (HitRecord.hh:45-77 = pbrt-v3 11.2): the phase function integrates to 1, its mean cosine is g (measured against
    -wo, pbrt's convention), the sampler's returned value is the phase function of the sampled direction, and sampled
    directions follow it."""
    L = pyoracle.lib()
    f2, f3 = C.c_float * 2, C.c_float * 3
    for g in (0.5, -0.3, 0.0, 0.85):
        mu = np.linspace(-1, 1, 20001)
        p = np.array([L.orc_phase_hg(float(m), g) for m in mu[::4]])
        assert np.trapezoid(p, mu[::4]) * 2 * np.pi == pytest.approx(1.0, rel=2e-3)
        want = (1 - g * g) / (4 * np.pi * (1 + g * g + 2 * g * mu[::4]) ** 1.5)
        assert np.allclose(p, want, rtol=1e-5)
        rs = np.random.RandomState(5)
        wo = np.array([0.3, -0.5, 0.81], np.float32); wo /= np.linalg.norm(wo)
        cosines = []
        for uu in rs.rand(4000, 2).astype(np.float32):
            wi = f3()
            pdf = L.orc_hg_sample(g, f3(*wo), f2(*uu), wi)
            w = np.array(wi)
            assert abs(np.linalg.norm(w) - 1) < 1e-5
            c = float(np.dot(wo, w))
            cosines.append(c)
            # both directions point away from the scattering point: p(wo, wi) = PhaseHG(dot(wo, wi)) and forward
            # scattering (g > 0) means wi ~ -wo, so E[dot(wo, wi)] = -g
            assert pdf == pytest.approx((1 - g * g) / (4 * np.pi * (1 + g * g + 2 * g * c) ** 1.5), rel=2e-3)
        assert np.mean(cosines) == pytest.approx(-g, abs=0.03)


def test_grid_density_is_trilinear_with_zero_outside():
    rs = np.random.RandomState(2)
    grid = rs.rand(5, 6, 7).astype(np.float32)                  # (nz, ny, nx)
    info = host.density_info(grid)
    f3 = C.c_float * 3
    L = pyoracle.lib()

    def D(x, y, z):
        return float(grid[z, y, x]) if 0 <= x < 7 and 0 <= y < 6 and 0 <= z < 5 else 0.0
    for p in rs.uniform(-0.2, 1.2, size=(300, 3)):
        s = p * np.array([7, 6, 5]) - 0.5
        i = np.floor(s).astype(int); d = s - i
        want = 0.0
        for dz in (0, 1):
            for dy in (0, 1):
                for dx in (0, 1):
                    w = (d[0] if dx else 1 - d[0]) * (d[1] if dy else 1 - d[1]) * (d[2] if dz else 1 - d[2])
                    want += w * D(i[0] + dx, i[1] + dy, i[2] + dz)
        got = L.orc_grid_density(C.byref(info), grid.ctypes.data, f3(*p.astype(np.float32)))
        assert got == pytest.approx(want, abs=2e-5)
