"""The constants the restatement shares with the reference, held to the reference's own TEXT.

tests/golden/reference_constants.json = numbers parsed out of /root/reference by tests/golden/make_reference_constants.py (call
arguments of MakeSquare / MakeCube / MakeSphere, material initialisers, the numeric literals of ErfInv, Erf, BeckmannSample11, ... in
order).  Here: libtrc_host.so's PODs equal the parsed arguments field for field; the literals of the oracle's functions equal the
reference's in order; the kernels' headers contain every one of them; the oracle's Erf / ErfInv / offset_ray / photon hash equal a
float64 evaluation built from the parsed lists at probe inputs; and a flipped digit fails (teeth).
"""
import ctypes as C
import importlib.util
import json
import math
import os
import re
import struct

import numpy as np
import pytest

from tracer_amd import abi, host
from oracle import pyoracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "reference_constants.json")

_spec = importlib.util.spec_from_file_location("make_reference_constants", os.path.join(ROOT, "tests", "golden", "make_reference_constants.py"))
mk = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(mk)


@pytest.fixture(scope="module")
def ref():
    with open(GOLD) as f:
        return json.load(f)


def f32(x):
    return struct.unpack("f", struct.pack("f", x))[0]


def src(rel):
    with open(os.path.join(ROOT, rel)) as f:
        return mk.strip_comments(f.read())


TRIVIAL = {0.0, 1.0, 2.0, 0.5}


def distinctive(vals):
    return [v for v in vals if abs(v) not in TRIVIAL]


# ----------------------------------------------------------------------------------------------------------------------------
@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference is present in the build container only")
def test_the_committed_file_is_what_the_parser_reads_today():
    doc = {"scene": mk.parse_scene(), "functions": mk.parse_functions()}
    with open(GOLD) as f:
        have = json.load(f)
    assert have["scene"] == json.loads(json.dumps(doc["scene"])) and have["functions"] == json.loads(json.dumps(doc["functions"]))


# ---------------------------------------------------------------------------------------------------------------------------- scene
def test_squares_equal_the_parsed_makesquare_arguments(ref, cornell_spheres):
    sc, want = cornell_spheres.view, ref["scene"]["squares"]
    assert sc.n_square == len(want) == 7
    # material variables -> table indices, from the order the reference pushes them (prepareCubeList pushed 3 before)
    names = {"light_index": 3, "red_index": 4, "green_index": 5, "white_index": 6}
    pad = f32(ref["scene"]["square_padding"])
    for i, w in enumerate(want):
        q = sc.squareList[i]
        assert (q.axis_i, q.axis_j, q.axis_k) == (w["axis_i"], w["axis_j"], w["axis_k"]), w["name"]
        assert (q.range_i.x, q.range_i.y) == tuple(map(f32, w["range_i"])) and (q.range_j.x, q.range_j.y) == tuple(map(f32, w["range_j"]))
        assert q.value_k == f32(w["value_k"]) and q.material == names[w["material_var"]], w["name"]
        lo = [q.boundingBOX.mini.x, q.boundingBOX.mini.y, q.boundingBOX.mini.z]
        hi = [q.boundingBOX.maxi.x, q.boundingBOX.maxi.y, q.boundingBOX.maxi.z]
        assert lo[w["axis_k"]] == f32(f32(w["value_k"]) - pad) and hi[w["axis_k"]] == f32(f32(w["value_k"]) + pad)
        assert (lo[w["axis_i"]], hi[w["axis_i"]]) == tuple(map(f32, w["range_i"]))


def _mat(m):
    return np.array([[getattr(m.columns[j], "xyzw"[i]) for j in range(4)] for i in range(4)], dtype=np.float64)


def test_cubes_equal_the_parsed_transforms(ref, cornell_spheres):
    sc, want = cornell_spheres.view, ref["scene"]["cubes"]
    assert sc.n_cube == len(want) == 3
    for i, w in enumerate(want):
        c = sc.cubeList[i]
        assert (c.box.mini.x, c.box.mini.y, c.box.mini.z) == tuple(w["box_min"]) and (c.box.maxi.x, c.box.maxi.y, c.box.maxi.z) == tuple(w["box_max"])
        T = np.eye(4); T[:3, 3] = w["translate"]
        S = np.diag(w["scale"] + [1.0])
        a, (x, y, z) = w["angle"], w["axis"]
        assert (x, y, z) == (0.0, 1.0, 0.0)
        R = np.array([[math.cos(a), 0, math.sin(a), 0], [0, 1, 0, 0], [-math.sin(a), 0, math.cos(a), 0], [0, 0, 0, 1]])
        M = T @ R @ S
        assert np.allclose(_mat(c.model_matrix), M, rtol=0, atol=2e-5 * np.abs(M).max()), w["name"]
        assert np.allclose(_mat(c.inverse_matrix) @ M, np.eye(4), atol=1e-5)
    # `19` is spelled in the reference (Tracer.mm:208); the other two are the indices their materials were pushed at
    assert [sc.cubeList[i].material for i in range(3)] == [0, int(want[1]["material_arg"]), 2]


def test_spheres_equal_the_parsed_makesphere_arguments(ref, cornell_spheres):
    sc, want = cornell_spheres.view, ref["scene"]["spheres"]
    infl = ref["scene"]["make_sphere_radius_inflation"]
    assert sc.n_sphere == len(want) == 12
    for i, w in enumerate(want):
        s = sc.sphereList[i]
        assert (s.center.x, s.center.y, s.center.z) == tuple(map(f32, w["center"]))
        assert s.radius == f32(w["radius"] + infl)                     # `r+0.0001`: float + double, rounded into the float member
        assert s.boundingBOX.maxi.x == f32(w["center"][0] + w["radius"])     # the box is NOT inflated (B-13)
        assert s.material == 7 + i


def test_material_table_equals_the_parsed_initialisers(ref, cornell):
    sc, want = cornell.view, ref["scene"]["materials"]
    assert ref["scene"]["material_order"] == ["prepareCubeList", "prepareCornellBox", "prepareSphereList"]
    assert sc.n_material == len(want) == 20
    types = {"Diffuse": abi.MAT_DIFFUSE, "Lambert": abi.MAT_LAMBERT, "Metal": abi.MAT_METAL, "Glass": abi.MAT_GLASS, "_NIL_": abi.MAT_NIL,
             "Dielectric": abi.MAT_DIELECTRIC, "Demofox": abi.MAT_DEMOFOX}
    tex = {"Constant": abi.TEX_CONSTANT, "Checker": abi.TEX_CHECKER, "Noise": 2}
    medium = {"_NIL_": 0, "Homogeneous": 1, "GridDensity": 2}
    for i, w in enumerate(want):
        m = sc.materials[i]
        assert m.type == types[w["type"]], i
        assert (m.textureInfo.albedo.x, m.textureInfo.albedo.y, m.textureInfo.albedo.z) == tuple(map(f32, w["albedo"])), i
        if "texture" in w:
            assert m.textureInfo.type == tex[w["texture"]], i
        if "medium" in w:
            assert m.medium == medium[w["medium"]], i
        if "eta" in w:
            assert m.eta == f32(w["eta"]), i
        assert bool(m.specular) == bool(w.get("specular", False)), i


def test_camera_defaults_equal_the_parsed_ones(ref):
    w = ref["scene"]["camera"]
    cam = host.prepare_camera(1920, 1080)
    assert (cam.lookFrom.x, cam.lookFrom.y, cam.lookFrom.z) == tuple(w["lookFrom"])
    assert (cam.lookAt.x, cam.lookAt.y, cam.lookAt.z) == tuple(w["lookAt"])
    assert (cam.viewUp.x, cam.viewUp.y, cam.viewUp.z) == tuple(w["viewUp"])
    assert cam.focus_dist == w["dist_focus"] and cam.aperture == w["aperture"] and cam.vfov == f32(w["vfov"])


# ---------------------------------------------------------------------------------------------------------------------------- literals
def oracle_functions():
    o = src("oracle/oracle.cpp")
    beck = mk.body_after(o, r"struct\s+Beckmann\s*")
    tr = mk.body_after(o, r"struct\s+TrowbridgeReitz\s*")
    pm = mk.body_after(o, r"struct\s+PlasticMaterial\s*")
    gm = mk.body_after(o, r"struct\s+GlassMaterial\s*")
    L = mk.literals
    return {
        "ErfInv": L(mk.body_after(o, r"inline\s+float\s+ErfInv\s*\(")),
        "Erf": L(mk.body_after(o, r"inline\s+float\s+Erf\s*\(")),
        "Beckmann::Lambda": L(mk.body_after(beck, r"float\s+Lambda\s*\(")),
        "BeckmannSample11": L(mk.body_after(beck, r"static\s+void\s+BeckmannSample11\s*\(")),
        "TrowbridgeReitz::Lambda": L(mk.body_after(tr, r"float\s+Lambda\s*\(")),
        "TrowbridgeReitz::D": L(mk.body_after(tr, r"float\s+D\s*\(const V3& wh\)")),
        "TrowbridgeReitzSample11": L(mk.body_after(tr, r"static\s+void\s+TrowbridgeReitzSample11\s*\(")),
        "FrConductor": L(mk.body_after(o, r"inline\s+V3\s+FrConductor\s*\(")),
        "FrDielectric": L(mk.body_after(o, r"inline\s+float\s+FrDielectric\s*\(")),
        "Photon::hash": L(mk.body_after(o, r"inline\s+float\s+ph_hash\s*\(")),
        "createMetalMaterial": L(mk.body_after(o, r"inline\s+MetalMaterial\s+createMetalMaterial\s*\(")),
        "createPlasticMaterial": L(mk.body_after(o, r"inline\s+PlasticMaterial\s+createPlasticMaterial\s*\(")),
        "createGlass": L(mk.body_after(o, r"inline\s+GlassMaterial\s+createGlass\s*\(")),
        "PlasticMaterial.ks_kd": L(pm[:pm.index("Lambertian")]),
        "GlassMaterial.kr_kt_ratio": L(gm[:gm.index("GlassMaterial(")]),
    }


ORDERED = ["ErfInv", "Erf", "Beckmann::Lambda", "BeckmannSample11", "TrowbridgeReitz::Lambda", "TrowbridgeReitz::D", "TrowbridgeReitzSample11",
           "PlasticMaterial.ks_kd", "GlassMaterial.kr_kt_ratio"]
# the oracle builds the materials in one expression each, and its hash declares the tables before the scale
UNORDERED = ["createMetalMaterial", "createPlasticMaterial", "createGlass", "Photon::hash"]


def test_the_oracles_literals_are_the_references(ref):
    ours, want = oracle_functions(), ref["functions"]
    for name in ORDERED:
        assert distinctive(ours[name]) == distinctive(want[name]), name
    for name in UNORDERED:
        assert sorted(distinctive(ours[name])) == sorted(distinctive(want[name])), name
    # Fresnel: no distinctive constant at all in either (2, 0.5, 1 only) -- the same count of each
    for name in ("FrConductor", "FrDielectric"):
        assert sorted(map(abs, ours[name])) == sorted(map(abs, want[name])), name
    # teeth: one digit of one literal of the oracle's text, and a dropped sign
    o = src("oracle/oracle.cpp")
    for a, b in (("0.246640727f", "0.246640721f"), ("-0.00125372503f", "0.00125372503f"), ("0.4265f", "0.4256f"), ("0.093073f", "0.093037f")):
        assert o.count(a) == 1
        t = o.replace(a, b)
        beck, tr = mk.body_after(t, r"struct\s+Beckmann\s*"), mk.body_after(t, r"struct\s+TrowbridgeReitz\s*")
        got = {"ErfInv": mk.literals(mk.body_after(t, r"inline\s+float\s+ErfInv\s*\(")),
               "BeckmannSample11": mk.literals(mk.body_after(beck, r"static\s+void\s+BeckmannSample11\s*\(")),
               "TrowbridgeReitzSample11": mk.literals(mk.body_after(tr, r"static\s+void\s+TrowbridgeReitzSample11\s*\("))}
        assert any(distinctive(got[k]) != distinctive(want[k]) for k in got), a


def test_the_kernels_headers_spell_every_one_of_them(ref):
    want = ref["functions"]
    bsdf = set(map(abs, mk.literals(src("tracer_amd/csrc/dev_bsdf.hpp"))))
    for name in ("ErfInv", "Erf", "Beckmann::Lambda", "BeckmannSample11", "TrowbridgeReitz::Lambda", "TrowbridgeReitz::D", "TrowbridgeReitzSample11",
                 "createMetalMaterial", "createPlasticMaterial", "createGlass", "PlasticMaterial.ks_kd", "GlassMaterial.kr_kt_ratio",
                 "Beckmann::Beckmann.alpha_floor", "TrowbridgeReitz::TrowbridgeReitz.alpha_floor"):
        missing = [v for v in distinctive(want[name]) if abs(v) not in bsdf]
        assert not missing, (name, missing)
    sppm = set(map(abs, mk.literals(src("tracer_amd/csrc/trc_sppm.hip"))))
    assert not [v for v in distinctive(want["Photon::hash"]) if abs(v) not in sppm]
    assert want["sppm_alpha"][0] in sppm and abi.PHOTON_HASHN == int(want["PHOTON_HASHN"][0]) if hasattr(abi, "PHOTON_HASHN") else True
    integ = set(map(abs, mk.literals(src("tracer_amd/csrc/dev_integrator.hpp"))))
    assert all(v in integ for v in want["rgb_to_y"]) and len(want["rgb_to_y"]) == 3
    inter = set(map(abs, mk.literals(src("tracer_amd/csrc/dev_intersect.hpp")))) | set(map(abs, mk.literals(src("tracer_amd/csrc/dev_vec.hpp"))))
    for k in ("offset_ray.origin", "offset_ray.float_scale", "offset_ray.int_scale"):
        v = want[k][0]
        assert v in inter or 1.0 / v in inter, k


# ---------------------------------------------------------------------------------------------------------------------------- functions
def erfinv64(c, x):
    """Math.hh:118-146 with the coefficients of list `c` (as parsed: clamp pair, 2.5, nine + nine Horner coefficients), in float64"""
    lo, hi, shift = c[0], c[1], c[2]
    a, b = c[3:12], c[12:21]
    x = min(max(x, lo), hi)
    w = -math.log((1 - x) * (1 + x))
    if w < 5:
        w -= shift
        p = a[0]
        for k in a[1:]:
            p = k + p * w
    else:
        w = math.sqrt(w) - 3
        p = b[0]
        for k in b[1:]:
            p = k + p * w
    return p * x


def erf64(c, x):
    a1, a2, a3, a4, a5, p = c
    s = -1 if x < 0 else 1
    x = abs(x)
    t = 1 / (1 + p * x)
    return s * (1 - (((((a5 * t + a4) * t) + a3) * t + a2) * t + a1) * t * math.exp(-x * x))


# |x| stays away from 1: there binary32's (1 - x) * (1 + x) cancels and the float64 evaluation is no longer the same function
# ... except TAIL, which reaches the second polynomial (w >= 5 <=> |x| >= 0.9966) at a tolerance that covers the cancellation
TAIL = [0.998, -0.9992, 0.99985]
PROBES = [-0.99, -0.9, -0.5, -0.1, -1e-3, 0.0, 1e-4, 0.05, 0.3, 0.7, 0.95, 0.995]


def test_erf_and_erfinv_follow_the_parsed_coefficients(ref):
    L = pyoracle.lib()
    L.orc_erf.restype = L.orc_erfinv.restype = C.c_float
    L.orc_erf.argtypes = L.orc_erfinv.argtypes = [C.c_float]
    ci, ce = ref["functions"]["ErfInv"], ref["functions"]["Erf"]
    assert len(ci) == 21 and len(ce) == 6
    def off(c, x, tol):
        want = erfinv64(c, f32(x))
        return abs(L.orc_erfinv(x) - want) > tol * max(1.0, abs(want))
    for x in PROBES:
        assert not off(ci, x, 4e-6), x
    for x in TAIL:
        assert not off(ci, x, 3e-4), x
    for x in [v * 3 for v in PROBES] + [-5.0, 4.0]:
        assert abs(L.orc_erf(x) - erf64(ce, f32(x))) <= 3e-7, x
    # teeth: one digit of one coefficient of either branch moves a probe by far more than the tolerance
    for idx, delta, probes, tol in ((10, 0.001, PROBES, 4e-6), (4, 1e-7, PROBES, 4e-6), (19, 0.01, TAIL, 3e-4)):      # (high-order coefficients move nothing measurable: the literal lists above hold those)
        bad = list(ci); bad[idx] += delta
        assert any(off(bad, x, tol) for x in probes), idx
    bad = list(ce); bad[2] += 1e-5
    assert any(abs(L.orc_erf(x) - erf64(bad, f32(x))) > 3e-7 for x in [v * 3 for v in PROBES])


def test_offset_ray_follows_the_parsed_constants(ref):
    fn = ref["functions"]
    origin, fscale, iscale = fn["offset_ray.origin"][0], fn["offset_ray.float_scale"][0], fn["offset_ray.int_scale"][0]
    L = pyoracle.lib()
    L.orc_offset_ray.argtypes = [C.POINTER(C.c_float)] * 3
    rng = np.random.default_rng(5)
    for _ in range(200):
        p = (rng.uniform(-1, 1, 3) * rng.choice([1e-3, 0.02, 0.04, 1.0, 300.0], 3)).astype(np.float32)
        n = rng.uniform(-1, 1, 3).astype(np.float32)
        n /= np.float32(np.linalg.norm(n))
        out = (C.c_float * 3)()
        L.orc_offset_ray((C.c_float * 3)(*p), (C.c_float * 3)(*n), out)
        for k in range(3):
            if abs(p[k]) < origin:
                want = np.float32(p[k] + np.float32(np.float32(fscale) * n[k]))
            else:
                of = int(np.float32(iscale) * n[k])                      # truncation, as the int3 constructor
                bits = int(np.frombuffer(np.float32(p[k]).tobytes(), dtype=np.int32)[0]) + (-of if p[k] < 0 else of)
                want = np.frombuffer(np.int32(bits).tobytes(), dtype=np.float32)[0]
            assert out[k] == want, (p, n, k)


def test_photon_hash_follows_the_parsed_constants(ref):
    c = ref["functions"]["Photon::hash"]
    scale_mul, q, r, a, m = c[0], c[1:5], c[5:9], c[9:13], c[13:17]
    assert c[17:] == [1.0, 0.5, 1.0, -1.0, 1.0, -1.0]
    n_hash = int(ref["functions"]["PHOTON_HASHN"][0])
    L = pyoracle.lib()
    L.orc_photon_hash.restype = C.c_float
    L.orc_photon_hash.argtypes = [C.POINTER(C.c_float), C.c_float]
    F = np.float32
    rng = np.random.default_rng(11)
    for _ in range(300):
        idx = rng.integers(0, 200, 3).astype(np.float32)
        hs = F(rng.choice([16.0, 64.0, 100.0, 317.0]))
        n4 = [idx[0], idx[1], idx[2], F(F(idx[0] + idx[1]) - idx[2])]
        acc = F(0)
        for k in range(4):
            n = F(F(n4[k] * F(scale_mul)) / hs)
            beta = F(np.floor(F(n / F(q[k]))))
            p = F(F(F(a[k]) * F(n - F(beta * F(q[k])))) - F(beta * F(r[k])))
            sign = F(1.0) if -p > 0 else (F(-1.0) if -p < 0 else F(0.0))
            beta = F(F(F(sign + F(1.0)) * F(0.5)) * F(m[k]))
            n = F(p + beta)
            term = F(F(n / F(m[k])) * F([1.0, -1.0, 1.0, -1.0][k]))
            acc = term if k == 0 else F(acc + term)
        fract = F(acc - F(np.floor(acc)))
        want = F(np.floor(F(fract * F(n_hash * n_hash))))
        got = L.orc_photon_hash((C.c_float * 3)(*idx), hs)
        assert got == want, (idx, hs)


def test_output_stage_texture_media_and_photon_emission_constants(ref):
    """second batch (round 6): ACESTone's five coefficients, the checker texture's frequencies, the homogeneous medium as
    Render.metal:118 constructs it, the path depth, and the photon emission (origin, flux scale, light squares) -- the reference's text
    against the oracle's and the kernels' spelling of them, and against what the library does with them where an entry point shows it"""
    fn = ref["functions"]
    o = src("oracle/oracle.cpp")
    i = o.index("const float A =")
    assert mk.literals(o[i:i + 90]) == fn["ACESTone"]                                        # orc_tonemap's copy
    abi_hip = src("tracer_amd/csrc/trc_abi.hip")
    i = abi_hip.index("const float A =")
    assert mk.literals(abi_hip[i:i + 90]) == fn["ACESTone"]                                  # k_tonemap's copy
    # the checker: sin(8 pi u) * cos(pi / 2 + 4 pi v), 0.5 * step(0, .) + 0.5
    assert fn["Texture::Checker"] == [8.0, 2.0, 4.0, 0.5, 0.0, 0.5]
    for text, a, b in ((o, "m_sin(8 * PI_F * uv.x)", "m_cos(PI_F / 2 + 4 * PI_F * uv.y)"),
                       (src("tracer_amd/csrc/dev_bsdf.hpp"), "8 * kPi * uv.x", "kPi / 2 + 4 * kPi * uv.y")):
        assert a in text and b in text
    # media, depth, photon emission
    assert fn["HomogeneousMedium.args"] == [0.02, 0.08, 0.5]
    for text in (o, src("tracer_amd/csrc/dev_integrator.hpp")):
        i = text.index("sigma_a = ")
        assert [v for v in mk.literals(text[i:i + 260]) if v in (0.02, 0.08, 0.5)][:2] == [0.02, 0.08] and "0.5f" in text[i:i + 900]
    assert fn["kernelPathTracing.depth"] == [8.0] and abi.Params().max_depth in (0, 8)
    assert fn["kernelPhotonRecording.origin"] == [450.0, 250.0, 250.0] and fn["kernelPhotonRecording.flux_scale"] == [100000.0]
    assert fn["kernelPhotonRecording.light_squares"] == [5.0, 6.0]
    for text in (o, src("tracer_amd/csrc/trc_sppm.hip")):
        assert re.search(r"\(450, 250, 250\)", text) and re.search(r"\* 100000\.0f", text)
        assert re.search(r"square_sample\(cx\.S, 5,|squareList\[5\]|sample_square\(.*5", text)
    # what the tone mapper does with the five coefficients: one grey pixel through orc_tonemap against a float64 evaluation
    acc = np.zeros((4, 4, 4), np.float32); acc[..., :3] = 0.18; acc[..., 3] = 1.0
    img, exposure = pyoracle.tonemap(acc)
    A, B, Cc, D, E = fn["ACESTone"]
    c = 0.18 * exposure
    want = (c * (A * c + B)) / (c * (Cc * c + D) + E)
    assert abs(int(img[0, 0, 0]) - int(min(max(want, 0.0), 1.0) * 255.0 + 0.5)) <= 1 and 20 < int(img[0, 0, 0]) < 250
    bad = (A + 0.3, B, Cc, D, E)                                     # teeth: a digit of the first coefficient moves the pixel
    assert abs(int(img[0, 0, 0]) - int(min((c * (bad[0] * c + B)) / (c * (Cc * c + D) + E), 1.0) * 255.0 + 0.5)) > 1
