#!/usr/bin/env python3
"""Pins the CONSTANTS the restatement shares with the reference to the reference's own text.

Runs in the build container only (the reference is not on the GPU box).  Imports and executes nothing of the reference: it
PARSES the sources where they lie under /root/reference and writes numbers -- call arguments, initialisers, the numeric literals of
named functions in the order they appear -- into tests/golden/reference_constants.json.  tests/test_reference_constants.py holds
libtrc_host.so's scene PODs, the oracle's functions and the kernels' headers to that file (a mistyped literal that kernel and oracle
share would otherwise go unseen: both restate the same reading).

    python tests/golden/make_reference_constants.py [--check]      # --check: compare with the committed file, write nothing
"""
import hashlib
import json
import math
import os
import re
import sys

REF = os.environ.get("TRC_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_constants.json")

NUM = r"(?<![\w.])(?:\d+\.\d*|\.\d+|\d+)(?:[eE][+-]?\d+)?f?(?![\w.])"


def read(rel):
    with open(os.path.join(REF, rel), encoding="utf-8", errors="replace") as f:
        return f.read()


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def body_after(text, pattern, start=0):
    """text between the first `{` after `pattern` and its matching `}` (comments already stripped)"""
    m = re.compile(pattern, re.S).search(text, start)
    if not m:
        raise SystemExit(f"pattern not found: {pattern}")
    i = text.index("{", m.end() - 1 if text[m.end() - 1] == "{" else m.end())
    depth, j = 0, i
    while True:
        c = text[j]
        if c == "{":
            depth += 1
        elif c == "}":
            depth -= 1
            if depth == 0:
                return text[i + 1:j]
        j += 1


def literals(code, floats_only=True):
    """numeric literals in order of appearance, as Python floats; a unary minus in front (after `=`, `(`, `,`, `{`, `?`, `:`, an
    operator or `return`) is taken into the value.  floats_only: literals that have a point, an exponent or an f suffix."""
    out = []
    for m in re.finditer(NUM, code):
        tok = m.group(0)
        if floats_only and not re.search(r"[.eEf]", tok):
            continue
        val = float(tok.rstrip("f"))
        before = code[:m.start()].rstrip()
        if before.endswith("-"):
            prev = before[:-1].rstrip()
            if prev == "" or prev[-1] in "=(,{?:*/+-<>&|" or prev.endswith("return"):
                val = -val
        out.append(val)
    return out


def f32(x):
    import struct
    return struct.unpack("f", struct.pack("f", x))[0]


def eval_expr(expr):
    """arithmetic of the call arguments: numbers, + - * /, parentheses, M_PI"""
    e = re.sub(r"(?<=[\d.])f\b", "", expr.strip())
    if not re.fullmatch(r"[\d\s.+\-*/()eEM_PI]+", e):
        raise SystemExit(f"unexpected expression: {expr!r}")
    return float(eval(e, {"__builtins__": {}}, {"M_PI": math.pi}))


def split_args(s):
    """top-level comma split of a call's argument text"""
    args, depth, cur = [], 0, ""
    for c in s:
        if c in "({":
            depth += 1
        elif c in ")}":
            depth -= 1
        if c == "," and depth == 0:
            args.append(cur.strip()); cur = ""
        else:
            cur += c
    if cur.strip():
        args.append(cur.strip())
    return args


def vec(s):
    """float2{a, b} / float3{a, b, c} / simd_make_float3(...) / float3(x) -> list of evaluated components"""
    m = re.search(r"[({](.*)[)}]", s, re.S)
    return [eval_expr(a) for a in split_args(m.group(1))]


def call_args(code, name, start=0):
    m = re.compile(r"\b" + name + r"\s*\(").search(code, start)
    if not m:
        return None, -1
    i, depth = m.end(), 1
    j = i
    while depth:
        if code[j] == "(":
            depth += 1
        elif code[j] == ")":
            depth -= 1
        j += 1
    return split_args(code[i:j - 1]), j


def parse_materials_in(fn_body, table):
    """appends, in emplace_back / push_back order, what each `Material x;` was given before it was pushed.  A push inside a
    `for (auto i : {..})` loop is repeated once per element of the initialiser list."""
    decl = {}
    for m in re.finditer(r"\bMaterial\s+(\w+)\s*;", fn_body):
        decl[m.group(1)] = {}
    events = []
    for m in re.finditer(r"\b(\w+)\.(type|medium|specular|eta|textureInfo\.type|textureInfo\.albedo)\s*=\s*([^;]+);", fn_body):
        if m.group(1) in decl:
            events.append((m.start(), "set", m.group(1), m.group(2), m.group(3).strip()))
    for m in re.finditer(r"materials\.(?:emplace_back|push_back)\s*\(\s*(\w+)\s*\)", fn_body):
        events.append((m.start(), "push", m.group(1), None, None))
    loops = [(m.start(), body_span(fn_body, m.end() - 1), len(split_args(m.group(1))))
             for m in re.finditer(r"for\s*\(\s*auto\s+\w+\s*:\s*\{([^}]*)\}\s*\)\s*\{", fn_body)]
    state = {k: {} for k in decl}
    for pos, kind, var, field, value in sorted(events):
        if kind == "set":
            if field == "textureInfo.albedo":
                v = vec(value) if re.search(r"[({]", value) else [eval_expr(value)]
                state[var]["albedo"] = v * 3 if len(v) == 1 else v
            elif field in ("type", "medium", "textureInfo.type"):
                state[var][{"type": "type", "medium": "medium", "textureInfo.type": "texture"}[field]] = value.split("::")[-1]
            elif field == "specular":
                state[var]["specular"] = value == "true"
            else:
                state[var]["eta"] = eval_expr(value)
        else:
            reps = 1
            for lstart, (a, b), n in loops:
                if a <= pos <= b:
                    reps = n
            for _ in range(reps):
                table.append(dict(state[var]))


def body_span(text, brace_at):
    depth, j = 0, brace_at
    while True:
        if text[j] == "{":
            depth += 1
        elif text[j] == "}":
            depth -= 1
            if depth == 0:
                return brace_at, j
        j += 1


def parse_scene():
    src = strip_comments(read("RT_Metal/Tracer/Tracer.mm"))
    out = {}
    # ---- MakeSphere: the radius is inflated, the box is not (Tracer.mm:165-172)
    ms = body_after(src, r"Sphere\s+MakeSphere\s*\(")
    out["make_sphere_radius_inflation"] = literals(ms)[0]
    sq_hdr = strip_comments(read("RT_Metal/Metal/Square.hh"))
    m = re.search(r"const\s+float\s+SquarePadding\s*=\s*([^;]+);", sq_hdr)
    out["square_padding"] = eval_expr(m.group(1))

    # ---- prepareCornellBox: MakeSquare(axis_i, range_i, axis_j, range_j, axis_k, k), materials by variable, list order
    cb = body_after(src, r"void\s+prepareCornellBox\s*\(")
    squares = {}
    for m in re.finditer(r"auto\s+(\w+)\s*=\s*MakeSquare\s*\(", cb):
        args, _ = call_args(cb, "MakeSquare", m.start())
        squares[m.group(1)] = {"axis_i": int(args[0]), "range_i": vec(args[1]), "axis_j": int(args[2]), "range_j": vec(args[3]),
                               "axis_k": int(args[4]), "value_k": eval_expr(args[5])}
    for m in re.finditer(r"(\w+)\.material\s*=\s*(\w+)\s*;", cb):
        if m.group(1) in squares:
            squares[m.group(1)]["material_var"] = m.group(2)
    out["squares"] = [dict(name=n, **squares[n]) for n in re.findall(r"list\.emplace_back\s*\(\s*(\w+)\s*\)", cb)]

    # ---- prepareCubeList: model = translate * rotate * scale, each cube takes the values assigned last before its product
    cl = body_after(src, r"void\s+prepareCubeList\s*\(")
    cubes, cur = [], {}
    ev = []
    for name in ("translation4x4", "rotation4x4", "scale4x4", "MakeCube"):
        for m in re.finditer(r"\b" + name + r"\s*\(", cl):
            args, _ = call_args(cl, name, m.start())
            ev.append((m.start(), name, args))
    for m in re.finditer(r"(\w+)\.model_matrix\s*=\s*translate\s*\*\s*rotate\s*\*\s*scale", cl):
        ev.append((m.start(), "model", m.group(1)))
    for pos, name, args in sorted(ev, key=lambda e: e[0]):
        if name == "translation4x4":
            cur["translate"] = [eval_expr(a) for a in args]
        elif name == "scale4x4":
            cur["scale"] = [eval_expr(a) for a in args]
        elif name == "rotation4x4":
            cur["angle"] = eval_expr(args[0]); cur["axis"] = vec(args[1])
        elif name == "MakeCube":
            cur["box_min"] = vec(args[0]); cur["box_max"] = vec(args[1]); cur["material_arg"] = args[2]
        else:
            cubes.append(dict(name=args, **cur))
    out["cubes"] = cubes

    # ---- prepareSphereList: MakeSphere(r, centre) -- one literal call, two loops over an initialiser list
    sl = body_after(src, r"void\s+prepareSphereList\s*\(")
    spheres = []
    loops = [(m.start(), body_span(sl, m.end() - 1), m.group(1), [int(x) for x in split_args(m.group(2))])
             for m in re.finditer(r"for\s*\(\s*auto\s+(\w+)\s*:\s*\{([^}]*)\}\s*\)\s*\{", sl)]
    for m in re.finditer(r"MakeSphere\s*\(", sl):
        args, _ = call_args(sl, "MakeSphere", m.start())
        inside = [lp for lp in loops if lp[1][0] <= m.start() <= lp[1][1]]
        if not inside:
            spheres.append({"radius": eval_expr(args[0]), "center": vec(args[1])})
        else:
            _, _, var, values = inside[0]
            comps = split_args(re.search(r"\((.*)\)", args[1], re.S).group(1))
            for i in values:
                c = [eval_expr(re.sub(r"\b" + var + r"\b", str(i), comp)) for comp in comps]
                spheres.append({"radius": eval_expr(args[0]), "center": c})
    out["spheres"] = spheres

    # ---- the material table in AAPLRenderer's order (AAPLRenderer.mm:213-246): cubes, Cornell box, spheres, testMaterial
    table = []
    for fn in ("prepareCubeList", "prepareCornellBox", "prepareSphereList"):
        parse_materials_in(body_after(src, r"void\s+" + fn + r"\s*\("), table)
    rn = strip_comments(read("RT_Metal/Tracer/AAPLRenderer.mm"))
    order = [m.group(1) for m in re.finditer(r"\b(prepareCubeList|prepareCornellBox|prepareSphereList)\s*\(\s*\w+\s*,\s*materials\s*\)", rn)]
    out["material_order"] = order
    i = rn.index("Material testMaterial;")
    j = rn.index("materials.emplace_back(testMaterial)", i)
    parse_materials_in(rn[i:j + 40], table)
    out["materials"] = table

    # ---- prepareCamera defaults (Tracer.mm:371-383)
    pc = body_after(src, r"void\s+prepareCamera\s*\(")
    cam = {}
    for key in ("lookFrom", "lookAt", "viewUp"):
        cam[key] = vec(re.search(r"auto\s+" + key + r"\s*=\s*(float3\s*\{[^}]*\})", pc).group(1))
    cam["dist_focus"] = eval_expr(re.search(r"auto\s+dist_focus\s*=\s*([^;]+);", pc).group(1))
    cam["aperture"] = eval_expr(re.search(r"auto\s+aperture\s*=\s*([^;]+);", pc).group(1))
    cam["vfov"] = eval_expr(re.search(r"auto\s+vfov\s*=\s*([^;]+);", pc).group(1))
    out["camera"] = cam
    return out


def parse_functions():
    """name -> the numeric literals of the function's body, in order"""
    math_hh = strip_comments(read("RT_Metal/Metal/Math.hh"))
    micro = strip_comments(read("RT_Metal/Metal/MicrofacetBXDF.h"))
    bxdf = strip_comments(read("RT_Metal/Metal/BXDF.metal"))
    photon_hh = strip_comments(read("RT_Metal/Metal/Photon.hh"))
    photon_metal = strip_comments(read("RT_Metal/Metal/Photon.metal"))
    common = strip_comments(read("RT_Metal/Metal/Common.hh"))
    spectrum = strip_comments(read("RT_Metal/Metal/Spectrum.hh"))
    beck = body_after(micro, r"struct\s+Beckmann\s*")
    tr = body_after(micro, r"struct\s+TrowbridgeReitz\s*")
    fn = {}
    fn["ErfInv"] = literals(body_after(math_hh, r"inline\s+float\s+ErfInv\s*\("))
    fn["Erf"] = literals(body_after(math_hh, r"inline\s+float\s+Erf\s*\("))
    fn["offset_ray.origin"] = [eval_expr(re.search(r"float\s+origin\s*\(\s*\)\s*\{\s*return\s+([^;]+);", math_hh).group(1))]
    fn["offset_ray.float_scale"] = [eval_expr(re.search(r"float\s+float_scale\s*\(\s*\)\s*\{\s*return\s+([^;]+);", math_hh).group(1))]
    fn["offset_ray.int_scale"] = [eval_expr(re.search(r"float\s+int_scale\s*\(\s*\)\s*\{\s*return\s+([^;]+);", math_hh).group(1))]
    fn["Beckmann::Lambda"] = literals(body_after(beck, r"float\s+Lambda\s*\("))
    fn["Beckmann::Beckmann.alpha_floor"] = literals(re.search(r"Beckmann\s*\(float alphax, float alphay\)\s*:([^{]*)\{", beck).group(1))
    fn["BeckmannSample11"] = literals(body_after(beck, r"void\s+BeckmannSample11\s*\("))
    fn["TrowbridgeReitz::Lambda"] = literals(body_after(tr, r"float\s+Lambda\s*\("))
    fn["TrowbridgeReitz::D"] = literals(body_after(tr, r"float\s+D\s*\(const thread float3& wh\)"))
    fn["TrowbridgeReitz::TrowbridgeReitz.alpha_floor"] = literals(body_after(tr, r"TrowbridgeReitz\s*\(float alphax, float alphay\)"))
    fn["TrowbridgeReitzSample11"] = literals(body_after(tr, r"static\s+void\s+TrowbridgeReitzSample11\s*\("))
    fn["FrConductor"] = literals(body_after(bxdf, r"float3\s+FrConductor\s*\("))
    fn["FrDielectric"] = literals(body_after(bxdf, r"float3\s+FrDielectric\s*\("))
    fn["Photon::hash"] = literals(body_after(photon_hh, r"inline\s+float\s+hash\s*\("))
    fn["createMetalMaterial"] = literals(body_after(micro, r"inline\s+MetalMaterial\s+createMetalMaterial\s*\("))
    fn["createPlasticMaterial"] = literals(body_after(micro, r"inline\s+PlasticMaterial\s+createPlasticMaterial\s*\("))
    fn["createGlass"] = literals(body_after(micro, r"inline\s+GlassMaterial\s+createGlass\s*\("))
    pm = body_after(micro, r"struct\s+PlasticMaterial\s*")
    fn["PlasticMaterial.ks_kd"] = literals(pm[:pm.index("Lambertian")])
    gm = body_after(micro, r"struct\s+GlassMaterial\s*")
    fn["GlassMaterial.kr_kt_ratio"] = literals(gm[:gm.index("GlassMaterial(")].replace("MicrofacetReflection", " ").replace("MicrofacetTransmission", " "))
    fn["rgb_to_y"] = literals(re.search(r"(0\.212671f[^;]*;)", spectrum).group(1)) if "0.212671f" in spectrum else []
    # round 6, second batch: the output stage, the checker texture, the media, the photon emission, the path depth
    render_hh = strip_comments(read("RT_Metal/Metal/Render.hh"))
    render_metal = strip_comments(read("RT_Metal/Metal/Render.metal"))
    texture_hh = strip_comments(read("RT_Metal/Metal/Texture.hh"))
    fn["ACESTone"] = literals(body_after(render_hh, r"inline\s+float3\s+ACESTone\s*\("))
    chk = re.search(r"case\s+TextureType::Checker\s*:\s*\{(.*?)\}", texture_hh, re.S).group(1)
    fn["Texture::Checker"] = literals(chk, floats_only=False)
    fn["HomogeneousMedium.args"] = [eval_expr(a) for a in call_args(render_metal, "HomogeneousMedium")[0]]
    kp = body_after(render_metal, r"kernel\s+void\s*\n?\s*kernelPathTracing\s*\(")
    fn["kernelPathTracing.depth"] = [float(re.search(r"=\s*trace(?:Path|MIS|Volume)\s*\(\s*(\d+)\s*,", kp).group(1))]
    pr = body_after(photon_metal, r"kernel\s+void\s*\n?\s*kernelPhotonRecording\s*\(")
    fn["kernelPhotonRecording.origin"] = vec(re.search(r"auto\s+_origin\s*=\s*(float3\s*\([^)]*\))", pr).group(1))
    fn["kernelPhotonRecording.flux_scale"] = [eval_expr(re.search(r"albedo\s*\*\s*([\d.]+)\s*;", pr).group(1))]
    fn["kernelPhotonRecording.light_squares"] = [float(x) for x in re.findall(r"squareList\[(\d+)\]\.sample", pr)]
    m = re.search(r"#define\s+PHOTON_HASHN\s+(\d+)", common)
    fn["PHOTON_HASHN"] = [float(m.group(1))]
    m = re.search(r"const\s+float\s+alpha\s*=\s*([\d.]+)\s*;", photon_metal)
    fn["sppm_alpha"] = [float(m.group(1))]
    return fn


def main():
    files = ["RT_Metal/Tracer/Tracer.mm", "RT_Metal/Tracer/AAPLRenderer.mm", "RT_Metal/Metal/Math.hh", "RT_Metal/Metal/MicrofacetBXDF.h",
             "RT_Metal/Metal/BXDF.metal", "RT_Metal/Metal/Photon.hh", "RT_Metal/Metal/Photon.metal", "RT_Metal/Metal/Common.hh",
             "RT_Metal/Metal/Square.hh", "RT_Metal/Metal/Spectrum.hh", "RT_Metal/Metal/Render.hh", "RT_Metal/Metal/Render.metal", "RT_Metal/Metal/Texture.hh"]
    doc = {"about": "numbers parsed out of the reference's sources by tests/golden/make_reference_constants.py (data, not source text)",
           "sources_sha256": {f: hashlib.sha256(open(os.path.join(REF, f), "rb").read()).hexdigest()[:16] for f in files},
           "scene": parse_scene(), "functions": parse_functions()}
    text = json.dumps(doc, indent=1, sort_keys=True) + "\n"
    if "--check" in sys.argv:
        ok = os.path.exists(OUT) and open(OUT).read() == text
        print("reference_constants.json is", "up to date" if ok else "STALE")
        sys.exit(0 if ok else 1)
    with open(OUT, "w") as f:
        f.write(text)
    print("wrote", OUT, len(text), "bytes")


if __name__ == "__main__":
    main()
