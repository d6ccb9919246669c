"""Analytic known-answer tests of the CPU oracle (the reference has no tests).
Each closed form is derived from the cited reference code, not from the oracle."""
import ctypes as C
import math

import numpy as np
import pytest

from conftest import make_rays
from oracle import pyoracle as po
from tracer_amd import abi, host

F32 = np.float32
FLT_MAX = float(np.finfo(np.float32).max)


def one_ray(o, d, tmax=FLT_MAX):
    return make_rays(np.array([o], F32), np.array([d], F32), tmax)


# ------------------------------------------------------------------ boxes (AABB.hh:92-112)
def box(lo, hi):
    b = abi.AABB()
    b.mini.x, b.mini.y, b.mini.z = lo
    b.maxi.x, b.maxi.y, b.maxi.z = hi
    return b


def hit_t(b, o, d, tmin=1.17549435e-38, tmax=FLT_MAX):
    r = abi.Ray()
    r.origin[:] = o
    n = math.sqrt(sum(x * x for x in d))
    r.direction[:] = [x / n for x in d]
    t = C.c_float()
    h = po.lib().orc_aabb_hit_t(C.byref(b), C.byref(r), tmin, tmax, C.byref(t))
    return bool(h), t.value


def test_aabb_hit_t_closed_forms():
    b = box((0, 0, 0), (1, 1, 1))
    assert hit_t(b, (-2, 0.5, 0.5), (1, 0, 0)) == (True, 2.0)            # entry distance
    # inside the box: tmin is clamped to range_t.x = FLT_MIN BEFORE the `tmin < 0` test (AABB.hh:104-108),
    # so Scene::hit (range_t.x = FLT_MIN, Render.hh:143) gets t = FLT_MIN, i.e. "nearest"
    assert hit_t(b, (0.5, 0.5, 0.5), (1, 0, 0)) == (True, 1.1754943508222875e-38)
    assert hit_t(b, (0.5, 0.5, 0.5), (1, 0, 0), tmin=-1.0) == (True, 0.5)  # only a negative range start returns tmax
    assert hit_t(b, (-2, 0.5, 0.5), (-1, 0, 0))[0] is False              # behind
    assert hit_t(b, (-2, 2.0, 0.5), (1, 0, 0))[0] is False               # parallel miss (inf/NaN slabs)
    assert hit_t(b, (-2, 0.5, 0.5), (1, 0, 0), tmax=1.5)[0] is False     # culled by the current closest hit
    h, t = hit_t(b, (0.5, 0.5, 0.5), (1, 0, 0), tmin=-1.0, tmax=0.25)     # inside, exit beyond range: t == range_t.y
    assert h and t == 0.25


# ------------------------------------------------------------------ primitives through Scene::hit
def two_leaf_scene(leaves, spheres=(), squares=(), cubes=(), tris=None, materials=None):
    nodes = host.build_tree(leaves)
    sv = abi.Scene()
    sv.bvhList, sv.n_bvh = C.cast(nodes, C.POINTER(abi.BVH)), len(nodes)
    keep = [nodes]
    for name, items, T in (("sphere", spheres, abi.Sphere), ("square", squares, abi.Square), ("cube", cubes, abi.Cube)):
        arr = (T * max(1, len(items)))(*items)
        keep.append(arr)
        setattr(sv, name + "List", C.cast(arr, C.POINTER(T)))
        setattr(sv, "n_" + name, len(items))
    mats = materials or [abi.Material() for _ in range(20)]
    marr = (abi.Material * len(mats))(*mats)
    keep.append(marr)
    sv.materials, sv.n_material = C.cast(marr, C.POINTER(abi.Material)), len(mats)
    if tris is not None:
        verts, idx = tris
        varr = (abi.TriangleVertex * len(verts))(*verts)
        iarr = (C.c_uint32 * len(idx))(*idx)
        keep += [varr, iarr]
        sv.triList, sv.n_vertex = C.cast(varr, C.POINTER(abi.TriangleVertex)), len(verts)
        sv.idxList, sv.n_index = C.cast(iarr, C.POINTER(C.c_uint32)), len(idx)
    sv._keep = keep
    return sv


def make_sphere(c, r, material=1):
    s = abi.Sphere()
    s.radius = r
    s.center.x, s.center.y, s.center.z = c
    s.material = material
    return s


def test_sphere_quadratic_roots_and_normal():
    # Sphere.hh:33-78: half-b quadratic, nearest root in (t_min, t_max), gn = (p - c)/r
    sp = [make_sphere((0, 0, 10), 2.0), make_sphere((100, 0, 0), 1.0)]
    leaves = [host.build_node((-2, -2, 8), (2, 2, 12), abi.PRIM_SPHERE, 0),
              host.build_node((99, -1, -1), (101, 1, 1), abi.PRIM_SPHERE, 1)]
    sv = two_leaf_scene(leaves, spheres=sp)
    h = po.trace_rays(sv, one_ray((0, 0, 0), (0, 0, 1)))[0]
    assert h["hit"] == 1 and h["pType"] == abi.PRIM_SPHERE and h["pIndex"] == 0
    assert h["t"] == 8.0 and tuple(h["gn"]) == (0.0, 0.0, -1.0) and tuple(h["sn"]) == (0.0, 0.0, -1.0)
    # from inside: the far root, geometric normal outward, shading normal flipped toward the ray
    h = po.trace_rays(sv, one_ray((0, 0, 10), (0, 0, 1)))[0]
    assert h["t"] == 2.0 and tuple(h["gn"]) == (0.0, 0.0, 1.0) and tuple(h["sn"]) == (0.0, 0.0, -1.0)
    # uv (Sphere.hh:19-24) at the -z pole of the equator: phi = atan2(-1, 0) = -pi/2 -> u = 0.75; v = 0.5
    h = po.trace_rays(sv, one_ray((0, 0, 0), (0, 0, 1)))[0]
    assert h["uv"][0] == pytest.approx(0.75, abs=1e-6) and h["uv"][1] == pytest.approx(0.5, abs=1e-6)
    # tangent ray: discriminant <= 0 rejects
    assert po.trace_rays(sv, one_ray((2.0, 0, 0), (0, 0, 1)))[0]["hit"] == 0


def make_square(ai, ri, aj, rj, ak, k, material=1):
    q = abi.Square()
    q.axis_i, q.axis_j, q.axis_k = ai, aj, ak
    q.range_i.x, q.range_i.y = ri
    q.range_j.x, q.range_j.y = rj
    q.value_k = k
    q.material = material
    return q


def square_leaf(q, index):
    lo, hi = [0, 0, 0], [0, 0, 0]
    lo[q.axis_i], hi[q.axis_i] = q.range_i.x, q.range_i.y
    lo[q.axis_j], hi[q.axis_j] = q.range_j.x, q.range_j.y
    lo[q.axis_k], hi[q.axis_k] = q.value_k - 1 / 512, q.value_k + 1 / 512
    return host.build_node(lo, hi, abi.PRIM_SQUARE, index)


def test_square_hit_uv_pdf_and_two_sidedness():
    # Square.hh:60-113; area() = 2*i*j (quirk B-7) -> PDF = 1/(2*4*2)
    qs = [make_square(0, (0, 4), 2, (0, 2), 1, 5.0, material=3), make_square(0, (10, 11), 2, (0, 1), 1, 0.0)]
    sv = two_leaf_scene([square_leaf(qs[0], 0), square_leaf(qs[1], 1)], squares=qs)
    h = po.trace_rays(sv, one_ray((1, 0, 0.5), (0, 1, 0)))[0]
    assert h["hit"] == 1 and h["t"] == 5.0 and tuple(h["p"]) == (1.0, 5.0, 0.5)
    assert tuple(h["uv"]) == (0.25, 0.25) and h["PDF"] == F32(1.0 / 16.0) and h["material"] == 3
    assert tuple(h["gn"]) == (0.0, -1.0, 0.0) and tuple(h["sn"]) == (0.0, -1.0, 0.0)   # gn = sn, facing the ray
    h = po.trace_rays(sv, one_ray((1, 9, 0.5), (0, -1, 0)))[0]
    assert h["t"] == 4.0 and tuple(h["gn"]) == (0.0, 1.0, 0.0)
    assert po.trace_rays(sv, one_ray((5, 0, 0.5), (0, 1, 0)))[0]["hit"] == 0          # outside range_i
    assert po.trace_rays(sv, one_ray((1, 0, 0.5), (1, 0, 0)))[0]["hit"] == 0          # parallel: t = inf/nan rejected


def vert(p, n=(0, 0, 1), uv=(0, 0)):
    v = abi.TriangleVertex()
    v.v[:], v.n[:], v.uv[:] = p, n, uv
    return v


def test_triangle_moller_trumbore_barycentrics():
    # Triangle.hh:31-85: p = u*b + v*c + w*a, UNNORMALISED interpolated normal, material 19, two-sided
    verts = [vert((0, 0, 5), (0, 0, -2), (0, 0)), vert((4, 0, 5), (0, 0, -2), (1, 0)), vert((0, 4, 5), (0, 0, -2), (0, 1)),
             vert((50, 0, 0)), vert((51, 0, 0)), vert((50, 1, 0))]
    idx = [0, 1, 2, 3, 4, 5]
    leaves = [host.build_node((0, 0, 5), (4, 4, 5), abi.PRIM_TRIANGLE, 0), host.build_node((50, 0, 0), (51, 1, 0), abi.PRIM_TRIANGLE, 1)]
    sv = two_leaf_scene(leaves, tris=(verts, idx))
    h = po.trace_rays(sv, one_ray((1, 1, 0), (0, 0, 1)))[0]
    assert h["hit"] == 1 and h["pType"] == abi.PRIM_TRIANGLE and h["t"] == 5.0 and h["material"] == 19
    assert tuple(h["p"]) == (1.0, 1.0, 5.0) and tuple(h["uv"]) == (0.25, 0.25)
    assert tuple(h["gn"]) == (0.0, 0.0, -2.0)                      # length 2 kept (B-5)
    assert po.trace_rays(sv, one_ray((1, 1, 9), (0, 0, -1)))[0]["t"] == 4.0     # back side also hits
    assert po.trace_rays(sv, one_ray((3, 3, 0), (0, 0, 1)))[0]["hit"] == 0      # u + v > 1


def test_cube_object_space_hit_and_world_distance():
    # Cube.hh:17-47: unit box scaled by 2 and moved to z in [4,6]; world t = distance(origin, world p)
    sc = host.HostScene(abi.SCENE_CORNELL)
    h = po.trace_rays(sc.view, one_ray((200, 100, -800), (0, 0, 1)))[0]
    assert h["hit"] == 1 and h["pType"] == abi.PRIM_CUBE and h["pIndex"] == 1 and h["material"] == 19
    assert abs(np.linalg.norm(h["gn"]) - 1) < 1e-6
    assert abs(np.linalg.norm(h["p"] - np.array([200, 100, -800], F32)) - h["t"]) < 1e-3


# ------------------------------------------------------------------ offset_ray (Math.hh:57-74)
def test_offset_ray_integer_ulp_arithmetic():
    L = po.lib()
    f3 = C.c_float * 3

    def off(p, n):
        out = f3()
        L.orc_offset_ray(f3(*p), f3(*n), out)
        return np.array(out, F32)
    # |p| >= 1/32: bits(p) +- int(256*n)
    r = off((100.0, -100.0, 0.5), (1.0, 1.0, 0.0))
    assert r[0] == F32(100.0).view(np.int32).__add__(256).view(F32) if False else True
    bits = np.array([100.0, -100.0], F32).view(np.int32)
    assert r[0].view(np.int32) == bits[0] + 256
    assert r[1].view(np.int32) == bits[1] - 256          # negative p: subtract, which moves TOWARD +y in value
    assert r[1] > F32(-100.0) and r[2] == F32(0.5)
    # |p| < 1/32: p + n/65536
    r = off((0.01, 0.0, -0.02), (0.0, 1.0, -1.0))
    assert r[1] == F32(1.0 / 65536.0) and r[2] == F32(F32(-0.02) + F32(-1.0 / 65536.0))


# ------------------------------------------------------------------ Fresnel / sampling / MIS
def test_fresnel_dielectric_normal_incidence_and_tir():
    L = po.lib()
    assert L.orc_fr_dielectric(1.0, 1.5) == pytest.approx(0.04, rel=1e-6)      # ((n-1)/(n+1))^2
    assert L.orc_fr_dielectric(-1.0, 1.5) == pytest.approx(0.04, rel=1e-6)     # from inside, eta inverted
    assert L.orc_fr_dielectric(-0.2, 1.5) == 1.0                               # total internal reflection
    assert L.orc_fr_dielectric(1.0, 1.0) == 0.0                                # the transmission lobe's own Fresnel (etaA = 1)


def test_fresnel_conductor_approximation():
    f3 = C.c_float * 3
    out = f3()
    po.lib().orc_fr_conductor(1.0, f3(0.18, 0.15, 0.81), f3(1, 1, 1), out)
    for e, got in zip((0.18, 0.15, 0.81), out):
        want = ((e * e + 1) - 2 * e + 1) / ((e * e + 1) + 2 * e + 1)              # BXDF.metal:24-34 at cos = 1
        assert got == pytest.approx(want, rel=1e-6)


def test_power_heuristic():
    L = po.lib()
    assert L.orc_power_heuristic(1, 0.3, 1, 0.3) == 0.5
    assert L.orc_power_heuristic(1, 2.0, 1, 1.0) == pytest.approx(0.8)


def test_cosine_hemisphere_is_on_the_unit_hemisphere_and_concentric():
    L = po.lib()
    f2, f3 = C.c_float * 2, C.c_float * 3
    out = f3()
    L.orc_cosine_sample_hemisphere(f2(0.5, 0.5), out)
    assert tuple(out) == (0.0, 0.0, 1.0)                                        # degenerate centre (Sampling.hh:84)
    rs = np.random.RandomState(1)
    zs = []
    for u in rs.rand(2000, 2).astype(F32):
        L.orc_cosine_sample_hemisphere(f2(*u), out)
        v = np.array(out, np.float64)
        assert abs(np.dot(v, v) - 1) < 1e-5 and v[2] >= 0
        zs.append(v[2])
    assert np.mean(zs) == pytest.approx(2 / 3, abs=0.02)                        # E[cos] under a cosine pdf


def lambert(albedo=(0.5, 0.25, 1.0), tex=abi.TEX_CONSTANT, mtype=abi.MAT_LAMBERT):
    m = abi.Material()
    m.type = mtype
    m.textureInfo.type = tex
    m.textureInfo.albedo.x, m.textureInfo.albedo.y, m.textureInfo.albedo.z = albedo
    return m


def S_F(m, wo, uu, uv=(0.3, 0.3)):
    f2, f3 = C.c_float * 2, C.c_float * 3
    wi, f, pdf = f3(), f3(), C.c_float()
    po.lib().orc_material_S_F(C.byref(m), f3(*wo), f2(*uv), f2(*uu), wi, f, C.byref(pdf))
    return np.array(wi, np.float64), np.array(f, np.float64), pdf.value


def test_lambert_furnace_f_over_pdf_is_albedo():
    # MatteBXDF.hh:6-22: F = wi.z/pi (cosine folded in), PDF = |wi.z|/pi  =>  F/PDF = 1 => throughput = albedo
    m = lambert()
    rs = np.random.RandomState(2)
    for uu in rs.rand(200, 2).astype(F32):
        wi, f, pdf = S_F(m, (0.3, 0.1, 0.9), uu)
        if pdf > 0:
            assert np.allclose(f / pdf, [0.5, 0.25, 1.0], rtol=1e-5)


def test_checker_texture_halves_albedo_on_alternate_cells():
    # Texture.hh:25-29: albedo * (0.5*step(0, sin(8 pi u) * cos(pi/2 + 4 pi v)) + 0.5)
    m = lambert(albedo=(0.8, 0.8, 0.8), tex=abi.TEX_CHECKER)
    vals = set()
    for uv in [(0.03, 0.03), (0.03, 0.3), (0.15, 0.03), (0.15, 0.3)]:
        _, f, pdf = S_F(m, (0, 0, 1), (0.3, 0.6), uv=uv)
        vals.add(round(float(f[0] / pdf), 4))
        su = math.sin(8 * math.pi * uv[0]) * math.cos(math.pi / 2 + 4 * math.pi * uv[1])
        assert f[0] / pdf == pytest.approx(0.8 if su >= 0 else 0.4, rel=1e-5)
    assert vals == {0.8, 0.4}


@pytest.mark.parametrize("mtype", [abi.MAT_METAL, abi.MAT_PLASTIC, abi.MAT_GLASS])
def test_microfacet_sample_eval_consistency(mtype):
    """S_F's (value, pdf) must equal F/PDF evaluated at the sampled direction with the same uu
    (Material.hh:77-146): checks the lobe selection plumbing of the composites."""
    m = lambert(albedo=(1, 1, 1), mtype=mtype)
    f2, f3 = C.c_float * 2, C.c_float * 3
    rs = np.random.RandomState(3)
    n_ok = 0
    for uu in rs.rand(300, 2).astype(F32):
        wo = (0.2, -0.1, 0.97)
        wi, f, pdf = S_F(m, wo, uu)
        if pdf <= 0:
            continue
        f2v, p2 = f3(), C.c_float()
        po.lib().orc_material_F(C.byref(m), f3(*wo), f3(*wi), f2(0.3, 0.3), f2(*uu), f2v, C.byref(p2))
        assert np.allclose(np.array(f2v), f, rtol=5e-3, atol=1e-7)
        # GlassMaterial::PDF scales by the lobe probability, S_F does not (MicrofacetBXDF.h:554-573)
        scale = 1.0
        if mtype == abi.MAT_GLASS:
            scale = 0.25 if uu[0] < 0.25 else 0.75
        assert p2.value == pytest.approx(pdf * scale, rel=5e-3)   # wh is re-derived from wo+wi: alpha = 0.01 lobes amplify the rounding
        n_ok += 1
    assert n_ok > 100


def test_unsupported_material_types_return_zero():
    # Material.hh:96-97,143-144: OrenNayar / Dielectric / Demofox / PBR ... return 0 and leave pdf untouched (-> 0, B-3)
    for t in (abi.MAT_ORENNAYAR, abi.MAT_DIELECTRIC, abi.MAT_DEMOFOX, abi.MAT_PBR, abi.MAT_NIL):
        _, f, pdf = S_F(lambert(mtype=t), (0, 0, 1), (0.4, 0.4))
        assert pdf == 0 and not f.any()


def test_erf_and_erfinv_roundtrip():
    L = po.lib()
    for x in (-0.9, -0.3, 0.0, 0.2, 0.7, 0.95):
        assert L.orc_erf(L.orc_erfinv(x)) == pytest.approx(x, abs=2e-3)        # A&S 7.1.26 is a 1.5e-7 fit; ErfInv ~1e-3
    assert L.orc_erf(0.5) == pytest.approx(math.erf(0.5), abs=1e-6)


def test_cast_ray_consumes_two_randoms_and_ignores_them_at_zero_aperture():
    # Camera.hh:59-69 + RandomSampler.hh:40-46: >= 2 randoms per call; origin = lookFrom when lenRadius = 0
    cam = host.prepare_camera(1920, 1080)
    f3 = C.c_float * 3
    o, d = f3(), f3()
    L = po.lib()
    s = C.c_uint64(12345)
    inc = 0xDA3E39CB94B95BDB
    s_ref = C.c_uint64(12345)
    L.orc_cast_ray(C.byref(cam), 0.5, 0.5, C.byref(s), inc, o, d)
    n = 0
    while s_ref.value != s.value and n < 64:
        L.orc_pcg32_random(C.byref(s_ref), inc)
        n += 1
    assert n >= 2 and n % 2 == 0
    assert tuple(o) == (278.0, 278.0, -800.0)
    assert np.allclose(np.array(d), [0, 0, 1], atol=1e-6)            # centre of the image looks down +z


# ------------------------------------------------------------------ published formulas, re-derived independently
# pbrt-v3 (the source the reference's MicrofacetBXDF.h follows): Beckmann / Trowbridge-Reitz D and Lambda,
# Torrance-Sparrow reflection and its pdf.  float64 numpy, written from the book's equations, not from the oracle.
def _tan2_cos2phi(w):
    c2 = w[2] * w[2]
    s2 = max(0.0, 1 - c2)
    tan2 = s2 / c2
    cphi2 = 1.0 if s2 == 0 else min(1.0, max(-1.0, w[0] / math.sqrt(s2))) ** 2
    sphi2 = 0.0 if s2 == 0 else min(1.0, max(-1.0, w[1] / math.sqrt(s2))) ** 2
    return tan2, c2, cphi2, sphi2


def _beckmann(ax, ay):
    def D(wh):
        tan2, c2, cp, sp = _tan2_cos2phi(wh)
        return math.exp(-tan2 * (cp / ax**2 + sp / ay**2)) / (math.pi * ax * ay * c2 * c2)

    def lam(w):
        tan2, c2, cp, sp = _tan2_cos2phi(w)
        a = 1 / (math.sqrt(cp * ax * ax + sp * ay * ay) * math.sqrt(tan2))
        return 0.0 if a >= 1.6 else (1 - 1.259 * a + 0.396 * a * a) / (3.535 * a + 2.181 * a * a)
    return D, lam


def _trowbridge(ax, ay):
    def D(wh):
        tan2, c2, cp, sp = _tan2_cos2phi(wh)
        e = (cp / ax**2 + sp / ay**2) * tan2
        return 1 / (math.pi * ax * ay * c2 * c2 * (1 + e) ** 2)

    def lam(w):
        tan2, c2, cp, sp = _tan2_cos2phi(w)
        return (-1 + math.sqrt(1 + (cp * ax * ax + sp * ay * ay) * tan2)) / 2
    return D, lam


def _fr_dielectric(cosi, eta):
    sin2t = (1 - cosi * cosi) / (eta * eta)
    if sin2t >= 1:
        return 1.0
    cost = math.sqrt(1 - sin2t)
    rpar = (eta * cosi - cost) / (eta * cosi + cost)
    rper = (cosi - eta * cost) / (cosi + eta * cost)
    return (rpar * rpar + rper * rper) / 2


@pytest.mark.parametrize("mtype,dist,scale,prob", [
    (abi.MAT_GLASS, _beckmann(0.01, 0.01), 0.98, 0.25),        # reflection lobe: uu.x < 0.25, kr = 0.98
    (abi.MAT_PLASTIC, _beckmann(0.01, 0.1), 0.2, 1.0),         # specular half: uu.x >= 0.5, ks = 0.2
    (abi.MAT_METAL, _trowbridge(0.01, 0.02), 1.0, 1.0)])
def test_microfacet_reflection_against_the_published_equations(mtype, dist, scale, prob):
    D, lam = dist
    m = lambert(albedo=(1, 1, 1), mtype=mtype)
    f2, f3 = C.c_float * 2, C.c_float * 3
    rs = np.random.RandomState(8)
    checked = 0
    for _ in range(400):
        # directions around a mirror configuration: the lobes are alpha = 0.01 wide
        wo = np.array([rs.uniform(-0.6, 0.6), rs.uniform(-0.6, 0.6), 0.0]); wo[2] = math.sqrt(1 - wo[0]**2 - wo[1]**2)
        wh = np.array([rs.normal(0, 0.01), rs.normal(0, 0.02), 1.0]); wh /= np.linalg.norm(wh)
        wi = -wo + 2 * wo.dot(wh) * wh
        if wi[2] <= 0.05:
            continue
        uu = (0.1, 0.5) if mtype == abi.MAT_GLASS else (0.9, 0.5)
        fv, pdf = f3(), C.c_float()
        po.lib().orc_material_F(C.byref(m), f3(*wo.astype(F32)), f3(*wi.astype(F32)), f2(0.3, 0.3), f2(*uu), fv, C.byref(pdf))
        wo32, wi32 = wo.astype(F32).astype(np.float64), wi.astype(F32).astype(np.float64)
        h = wo32 + wi32; h /= np.linalg.norm(h)
        G1 = 1 / (1 + lam(wo32))
        want_pdf = prob * D(h) * G1 * abs(wo32.dot(h)) / abs(wo32[2]) / (4 * wo32.dot(h))
        G = 1 / (1 + lam(wo32) + lam(wi32))
        want_f = scale * D(h) * G / (4 * wi32[2] * wo32[2])
        if mtype == abi.MAT_METAL:
            # FrConductor(eta = (0.18, 0.15, 0.81), k = 1), approximate form of BXDF.metal:24-34, red channel
            c, eta, k = abs(wi32.dot(h)), 0.18, 1.0
            t = (eta * eta + k * k) * c * c
            rpar = (t - 2 * eta * c + 1) / (t + 2 * eta * c + 1)
            tf = eta * eta + k * k
            rper = (tf - 2 * eta * c + c * c) / (tf + 2 * eta * c + c * c)
            want_f *= 0.5 * (rpar + rper)
        else:
            want_f *= _fr_dielectric(abs(wi32.dot(h)), 1.5)
        # the half vector of a 0.01-wide lobe is re-derived from float32 directions: allow for its conditioning
        assert pdf.value == pytest.approx(want_pdf, rel=2e-2)
        assert fv[0] == pytest.approx(want_f, rel=2e-2)
        checked += 1
    assert checked > 200


def test_glass_transmission_against_the_published_equations():
    """pbrt-v3 MicrofacetTransmission::f / Pdf (the source MicrofacetBXDF.h:64-135 follows) in float64: eta = etaB/etaA on
    the outside, wh = normalize(wo + eta wi), f = (1 - F) T |D G eta^2 |wi.wh| |wo.wh| / (cos_i cos_o (wo.wh + eta wi.wh)^2)|
    (radiance mode factor 1), pdf = D(wh) G1(wo) |wo.wh| / |cos_o| * |eta^2 wi.wh / (wo.wh + eta wi.wh)^2|.  As instantiated by
    GlassMaterial (:541): Beckmann (0.01, 0.01), T = 0.98, etaA = 1, etaB = 1.5, its own Fresnel with eta = etaA = 1 (F = 0),
    and Material::PDF scales by the lobe probability 0.75 (:554-562)."""
    D, lam = _beckmann(0.01, 0.01)
    m = lambert(albedo=(1, 1, 1), mtype=abi.MAT_GLASS)
    f2, f3 = C.c_float * 2, C.c_float * 3
    rs = np.random.RandomState(18)
    checked = 0
    for _ in range(600):
        wo = np.array([rs.uniform(-0.5, 0.5), rs.uniform(-0.5, 0.5), 0.0]); wo[2] = math.sqrt(1 - wo[0]**2 - wo[1]**2)
        wh = np.array([rs.normal(0, 0.01), rs.normal(0, 0.01), 1.0]); wh /= np.linalg.norm(wh)
        # Snell through the micro-normal, entering (eta_i / eta_t = 1 / 1.5)
        e = 1 / 1.5
        c = wo.dot(wh)
        s2t = e * e * (1 - c * c)
        wi = -e * wo + (e * c - math.sqrt(1 - s2t)) * wh
        if wi[2] >= -0.05:
            continue
        fv, pdf = f3(), C.c_float()
        po.lib().orc_material_F(C.byref(m), f3(*wo.astype(F32)), f3(*wi.astype(F32)), f2(0.3, 0.3), f2(0.6, 0.5), fv, C.byref(pdf))
        wo32, wi32 = wo.astype(F32).astype(np.float64), wi.astype(F32).astype(np.float64)
        eta = 1.5
        h = wo32 + eta * wi32; h /= np.linalg.norm(h)
        if h[2] < 0:
            h = -h
        denom = wo32.dot(h) + eta * wi32.dot(h)
        G = 1 / (1 + lam(wo32) + lam(wi32))
        want_f = 0.98 * abs(D(h) * G * eta * eta * abs(wi32.dot(h)) * abs(wo32.dot(h)) / (wi32[2] * wo32[2] * denom * denom))
        G1 = 1 / (1 + lam(wo32))
        want_pdf = 0.75 * D(h) * G1 * abs(wo32.dot(h)) / abs(wo32[2]) * abs(eta * eta * wi32.dot(h) / (denom * denom))
        assert fv[0] == fv[1] == fv[2]
        assert fv[0] == pytest.approx(want_f, rel=3e-2) and pdf.value == pytest.approx(want_pdf, rel=3e-2)
        checked += 1
    assert checked > 300


def _sampled_half_vectors(mtype, wo, n, seed, lobe_u):
    """wh = normalize(wo + wi) of n reflection samples of Material::S_F (uu.x remapped into the lobe by lobe_u)"""
    m = lambert(albedo=(1, 1, 1), mtype=mtype)
    rs = np.random.RandomState(seed)
    out = []
    for u in rs.rand(n, 2):
        wi, f, pdf = S_F(m, wo, (lobe_u(u[0]), u[1]))
        if pdf > 0:
            h = np.array(wo, np.float64) + np.array(wi, np.float64)
            out.append(h / np.linalg.norm(h))
    return np.array(out)


def test_visible_normal_sampling_at_normal_incidence_follows_the_slope_distributions():
    """at wo = +z every microfacet is visible, so the sampled slopes follow the distribution itself: Beckmann slopes / alpha
    are Gaussian, r^2 ~ Exp(1) (mean 1, median ln 2); Trowbridge-Reitz: CDF(r) = r^2 / (1 + r^2) (quartiles 1/sqrt 3, 1, sqrt 3)"""
    wo = (0.0, 0.0, 1.0)
    wh = _sampled_half_vectors(abi.MAT_PLASTIC, wo, 12000, 5, lambda u: 0.5 + 0.5 * u)          # Beckmann (0.01, 0.1), specular half
    sx, sy = -wh[:, 0] / wh[:, 2] / 0.01, -wh[:, 1] / wh[:, 2] / 0.1
    r2 = sx * sx + sy * sy
    assert len(r2) > 11000 and r2.mean() == pytest.approx(1.0, rel=0.04) and np.median(r2) == pytest.approx(math.log(2), rel=0.05)
    assert abs(sx.mean()) < 0.03 and abs(sy.mean()) < 0.03 and sx.var() == pytest.approx(0.5, rel=0.06) and sy.var() == pytest.approx(0.5, rel=0.06)
    wh = _sampled_half_vectors(abi.MAT_METAL, wo, 12000, 6, lambda u: u)                           # Trowbridge-Reitz (0.01, 0.02)
    r = np.hypot(-wh[:, 0] / wh[:, 2] / 0.01, -wh[:, 1] / wh[:, 2] / 0.02)
    q1, q2, q3 = np.percentile(r, [25, 50, 75])
    assert q1 == pytest.approx(1 / math.sqrt(3), rel=0.05) and q2 == pytest.approx(1.0, rel=0.05) and q3 == pytest.approx(math.sqrt(3), rel=0.06)


@pytest.mark.parametrize("mtype,dist,alpha,lobe_u", [
    (abi.MAT_PLASTIC, _beckmann(0.01, 0.1), (0.01, 0.1), lambda u: 0.5 + 0.5 * u),
    (abi.MAT_METAL, _trowbridge(0.01, 0.02), (0.01, 0.02), lambda u: u)])
def test_visible_normal_sampling_at_oblique_incidence(mtype, dist, alpha, lobe_u):
    """Heitz's visible-normal distribution D_wo(wh) = D(wh) G1(wo) max(0, wo.wh) / cos_o, integrated in float64 over slope
    space from the published D and Lambda: the sampled half vectors must fall into the quadrants / the 1-alpha disc with
    those probabilities (the general branch of BeckmannSample11 / TrowbridgeReitzSample11 at 60 degrees)"""
    D, lam = dist
    ax, ay = alpha
    wo = np.array([math.sin(math.radians(60)) * 0.8, math.sin(math.radians(60)) * 0.6, math.cos(math.radians(60))])
    wh = _sampled_half_vectors(mtype, tuple(wo.astype(F32)), 9000, 9, lobe_u)
    sx, sy = -wh[:, 0] / wh[:, 2], -wh[:, 1] / wh[:, 2]
    # reference probabilities on a grid in (slope / alpha) space, wide enough for Trowbridge-Reitz's heavy tails
    g = np.concatenate([-np.geomspace(400, 0.02, 500), [0.0], np.geomspace(0.02, 400, 500)])
    edges = np.concatenate([[g[0]], 0.5 * (g[1:] + g[:-1]), [g[-1]]])
    cell = np.diff(edges)
    gx, gy = np.meshgrid(g * ax, g * ay, indexing="ij")
    area = np.outer(cell * ax, cell * ay)
    nz = 1 / np.sqrt(1 + gx * gx + gy * gy)
    hx, hy, hz = -gx * nz, -gy * nz, nz
    # D(wh) of the published distributions, vectorised: tan^2 = slope^2, cos^2 phi = sx^2 / slope^2
    e = (gx / ax) ** 2 + (gy / ay) ** 2
    c4 = hz ** 4
    dens = np.exp(-e) / (math.pi * ax * ay * c4) if mtype == abi.MAT_PLASTIC else 1 / (math.pi * ax * ay * c4 * (1 + e) ** 2)
    spot = (len(g) // 2 + 37, len(g) // 2 - 11)
    assert dens[spot] == pytest.approx(D(np.array([hx[spot], hy[spot], hz[spot]])), rel=1e-9)   # same D as the scalar form above
    G1 = 1 / (1 + lam(wo))
    w = dens * G1 * np.maximum(0.0, hx * wo[0] + hy * wo[1] + hz * wo[2]) / wo[2] * nz ** 3 * area    # d(omega) = cos^3 ds_x ds_y
    assert w.sum() == pytest.approx(1.0, abs=0.01)                                 # D_wo is normalised (Heitz 2014)
    w /= w.sum()
    for name, region_g, region_s in (
            ("slope_x < 0", gx < 0, sx < 0),
            ("slope_y < 0", gy < 0, sy < 0),
            ("inside the one-alpha ellipse", (gx / ax) ** 2 + (gy / ay) ** 2 < 1, (sx / ax) ** 2 + (sy / ay) ** 2 < 1)):
        want, got = w[region_g].sum(), region_s.mean()
        assert got == pytest.approx(want, abs=0.02), (name, want, got)
