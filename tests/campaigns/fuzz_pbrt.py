#!/usr/bin/env python3
"""pbrt-v3 scene description campaign (CPU; needs oracle/_ref/libminipbrt_ref.so, i.e. the build container): random files -- transform
directives in random order and nesting (Translate / Scale / Rotate / LookAt / Transform / ConcatTransform / Identity, TransformBegin/End,
AttributeBegin/End, CoordinateSystem / CoordSysTransform), the seven shape classes with random parameters, materials with colours, area
lights, camera and film -- through libtrc_host.so's loader and through the reference's own minipbrt: kind, shape-to-world matrix, parameters,
material, colour, emitter, camera matrix / fov / film, shape for shape.        python3 tests/campaigns/fuzz_pbrt.py <a> <b>"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from tracer_amd import abi, host
from test_pbrt_scene import ref_describe

def num(rs, lo, hi): return f"{rs.uniform(lo, hi):.4f}"

def transform(rs, depth=0):
    k = rs.randint(0, 7)
    if k == 0: return f"Translate {num(rs,-50,50)} {num(rs,-50,50)} {num(rs,-50,50)}"
    if k == 1: return f"Scale {num(rs,0.2,4)} {num(rs,0.2,4)} {num(rs,0.2,4)}"
    if k == 2:
        ax = rs.normal(size=3); ax = ax if np.abs(ax).max() > 0.1 else np.array([0, 1, 0.0])
        return f"Rotate {num(rs,-180,180)} {ax[0]:.4f} {ax[1]:.4f} {ax[2]:.4f}"
    if k == 3:
        e = rs.uniform(-30, 30, 3); l = e + rs.normal(size=3) * 10; u = rs.normal(size=3)
        return "LookAt " + " ".join(f"{v:.4f}" for v in list(e) + list(l) + list(u))
    if k == 4 or k == 5:
        m = np.eye(4); m[:3, :3] += rs.normal(0, 0.4, (3, 3)); m[3, :3] = rs.uniform(-20, 20, 3)      # pbrt files are column-major: translation last row as written
        return ("Transform" if k == 4 else "ConcatTransform") + " [ " + " ".join(f"{v:.4f}" for v in m.reshape(-1)) + " ]"
    return "Identity" if rs.rand() < 0.3 else f"Translate {num(rs,-5,5)} 0 0"

def shape(rs):
    k = rs.randint(0, 7)
    if k == 0: return f'Shape "sphere" "float radius" {num(rs,0.5,30)}'
    if k == 1: return f'Shape "disk" "float radius" {num(rs,1,30)} "float innerradius" {num(rs,0,0.9)} "float height" {num(rs,-5,5)} "float phimax" {num(rs,30,360)}'
    if k == 2: return f'Shape "cylinder" "float radius" {num(rs,1,20)} "float zmin" {num(rs,-10,0)} "float zmax" {num(rs,0.5,30)} "float phimax" {num(rs,30,360)}'
    if k == 3: return f'Shape "cone" "float radius" {num(rs,1,20)} "float height" {num(rs,1,30)} "float phimax" {num(rs,30,360)}'
    if k == 4: return f'Shape "paraboloid" "float radius" {num(rs,1,20)} "float zmin" {num(rs,0,2)} "float zmax" {num(rs,3,30)} "float phimax" {num(rs,30,360)}'
    if k == 5: return f'Shape "hyperboloid" "point p1" [ {num(rs,1,9)} {num(rs,-3,3)} {num(rs,-5,0)} ] "point p2" [ {num(rs,1,9)} {num(rs,-3,3)} {num(rs,1,9)} ] "float phimax" {num(rs,30,360)}'
    n = rs.randint(3, 9); P = rs.uniform(-20, 20, (n, 3)); idx = rs.randint(0, n, 3 * rs.randint(1, 6))
    return 'Shape "trianglemesh" "integer indices" [ ' + " ".join(map(str, idx)) + ' ] "point P" [ ' + " ".join(f"{v:.3f}" for v in P.reshape(-1)) + " ]"

def material(rs):
    k = rs.randint(0, 6); c = " ".join(f"{v:.3f}" for v in rs.uniform(0, 1, 3))
    return [f'Material "matte" "rgb Kd" [ {c} ]', f'Material "plastic" "rgb Kd" [ {c} ]', 'Material "metal"', f'Material "mirror" "rgb Kr" [ {c} ]',
            f'Material "glass" "rgb Kt" [ {c} ]', 'Material "matte"'][k]

def body(rs, out, depth):
    for _ in range(rs.randint(1, 7)):
        k = rs.randint(0, 10)
        if k < 3: out.append("  " * depth + transform(rs))
        elif k < 5: out.append("  " * depth + material(rs))
        elif k == 5 and depth < 4:
            kind = "Attribute" if rs.rand() < 0.6 else "Transform"
            out.append("  " * depth + kind + "Begin"); body(rs, out, depth + 1); out.append("  " * depth + kind + "End")
        elif k == 6: out.append("  " * depth + (f'AreaLightSource "diffuse" "rgb L" [ {num(rs,1,20)} {num(rs,1,20)} {num(rs,1,20)} ]'))
        elif k == 7 and rs.rand() < 0.5:
            name = f"cs{rs.randint(0, 3)}"
            out.append("  " * depth + (f'CoordinateSystem "{name}"' if rs.rand() < 0.6 or name not in body.named else f'CoordSysTransform "{name}"')); body.named.add(name)
        else: out.append("  " * depth + shape(rs))

a, b = int(sys.argv[1]), int(sys.argv[2])
bad = refused = 0
with tempfile.TemporaryDirectory() as d:
    for seed in range(a, b):
        rs = np.random.RandomState(seed)
        body.named = set()
        out = [transform(rs) if rs.rand() < 0.5 else "", f"LookAt {num(rs,-100,100)} {num(rs,-100,100)} {num(rs,-300,-100)}  {num(rs,-10,10)} {num(rs,-10,10)} 0  0 1 0",
               f'Camera "perspective" "float fov" {num(rs,20,90)}', f'Film "image" "integer xresolution" {rs.randint(16, 400)} "integer yresolution" {rs.randint(16, 300)}',
               "WorldBegin"]
        body(rs, out, 1)
        out.append('  Shape "sphere" "float radius" 2'); out.append("  " + shape(rs))               # a tree needs two leaves (the loader refuses less)
        out.append("WorldEnd")
        path = os.path.join(d, f"s{seed}.pbrt"); open(path, "w").write("\n".join(out) + "\n")
        why = []
        try:
            scene, cam, info, shapes = host.HostScene.from_pbrt(path)
            rcam, rfilm, rshapes = ref_describe(path)
        except Exception as e:
            if "trc_status -7" in str(e):            # fewer than two shapes the reference's primitives can express (e.g. only non-uniformly scaled spheres): declared
                refused += 1
                continue
            why.append(f"load: {e}")
        if not why:
            if (info.xres, info.yres) != rfilm: why.append(f"film {(info.xres, info.yres)} / {rfilm}")
            if np.float32(info.fov) != rcam[16]: why.append("fov")
            c2w = np.array(info.camera_to_world[:], np.float32)
            if not np.allclose(c2w, rcam[:16], rtol=1e-4, atol=1e-3 * max(1.0, np.abs(rcam[:16]).max())): why.append("camera matrix")
            if info.n_shapes != len(rshapes) or len(shapes) != len(rshapes): why.append(f"{info.n_shapes} / {len(shapes)} / {len(rshapes)} shapes")
            else:
                for k, (mine, ref) in enumerate(zip(shapes, rshapes)):
                    if mine.kind != int(ref[0]): why.append(f"shape {k} kind {mine.kind} / {int(ref[0])}"); break
                    m = np.array(mine.shape_to_world[:], np.float32)
                    if not np.allclose(m, ref[1:17], rtol=1e-4, atol=1e-4 * max(1.0, np.abs(ref[1:17]).max())): why.append(f"shape {k} matrix"); break
                    if mine.kind in (abi.PBRT_SHAPE_SPHERE, abi.PBRT_SHAPE_DISK, abi.PBRT_SHAPE_CYLINDER) and np.float32(mine.radius) != ref[17]: why.append(f"shape {k} radius")
                    if mine.kind == abi.PBRT_SHAPE_TRIANGLEMESH and (mine.n_vertices, mine.n_indices) != (int(ref[18]), int(ref[19])): why.append(f"shape {k} mesh sizes")
                    if mine.kind in (abi.PBRT_SHAPE_DISK, abi.PBRT_SHAPE_CYLINDER) and (np.float32(mine.zmin), np.float32(mine.zmax), np.float32(mine.phimax)) != (ref[35], ref[36], ref[38]): why.append(f"shape {k} z / phi")
                    if int(ref[20]) == -1:          # no Material directive yet: minipbrt leaves the index invalid, this loader has pbrt-v3's default (matte, Kd 0.5: api.cpp)
                        if mine.material != abi.PBRT_MATTE or not np.array_equal(np.array(mine.color[:], np.float32), np.float32([0.5, 0.5, 0.5])): why.append(f"shape {k} default material")
                    elif mine.material != int(ref[20]): why.append(f"shape {k} material {mine.material} / {int(ref[20])}")
                    elif mine.material != abi.PBRT_OTHER and not np.array_equal(np.array(mine.color[:], np.float32), ref[21:24]): why.append(f"shape {k} colour")
                    if mine.emitter != int(ref[24]) or (mine.emitter and not np.array_equal(np.array(mine.L[:], np.float32), ref[25:28])): why.append(f"shape {k} emitter")
                    if why: break
        if why:
            bad += 1
            keep = os.path.join(ROOT, "gpurun_out", f"fuzz_pbrt_{seed}.pbrt"); os.makedirs(os.path.dirname(keep), exist_ok=True); open(keep, "w").write("\n".join(out) + "\n")
            print(f"MISMATCH seed {seed}: {why[:3]}  ({keep})", flush=True)
print(f"seeds {a}..{b - 1}: {b - a - bad - refused} passed, {refused} refused by the loader (fewer than two mappable shapes), {bad} FAILED")
