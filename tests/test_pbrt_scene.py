"""trc_host_scene_load_pbrt (a whole scene from a pbrt-v3 file: the reference's unchecked to-do, README.md:57) against
the REFERENCE's own parser -- RT_Metal/Tracer/minipbrt.cpp compiled where it lies into oracle/_ref/libminipbrt_ref.so
(oracle/ref_minipbrt_shim.cpp::ref_minipbrt_describe): camera matrix, fov, lens, film resolution, and for every world
shape its type, shapeToWorld, radius / mesh sizes, material type + colour and area-light radiance, field for field.
Then what the description became: spheres, squares (axis-aligned rectangle meshes, lights at squareList[5] / [6]),
triangles with material 19 -- and the loaded scene renders on the oracle (the GPU renders it in test_gpu_pbrt.py)."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import pyoracle as po
from tracer_amd import abi, host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_LIB = os.path.join(ROOT, "oracle", "_ref", "libminipbrt_ref.so")
needs_ref = pytest.mark.skipif(not os.path.exists(REF_LIB), reason="oracle/_ref/libminipbrt_ref.so not built (reference absent)")

CORNELL = '''# a Cornell box the way pbrt-v3 scene files spell one
LookAt 278 273 -800  278 273 0  0 1 0
Camera "perspective" "float fov" [ 39 ] "float lensradius" 0.5 "float focaldistance" [ 1000 ]
Film "image" "integer xresolution" [ 160 ] "integer yresolution" [ 120 ] "string filename" "cornell.exr"
Sampler "halton" "integer pixelsamples" 8
WorldBegin
MakeNamedMaterial "white" "string type" "matte" "rgb Kd" [ 0.73 0.73 0.73 ]
MakeNamedMaterial "red"   "string type" [ "matte" ] "rgb Kd" [ 0.65 0.05 0.05 ]
MakeNamedMaterial "green" "string type" "matte" "rgb Kd" [ 0.12 0.45 0.15 ]
Material "matte" "rgb Kd" [ 0.73 0.73 0.73 ]
Shape "trianglemesh" "integer indices" [ 0 1 2 0 2 3 ] "point P" [ 0 0 0  555 0 0  555 0 555  0 0 555 ]          # floor
Shape "trianglemesh" "integer indices" [ 0 1 2 0 2 3 ] "point P" [ 0 555 0  555 555 0  555 555 555  0 555 555 ]  # ceiling
Shape "trianglemesh" "integer indices" [ 0 1 2 0 2 3 ] "point P" [ 0 0 555  555 0 555  555 555 555  0 555 555 ]  # back
NamedMaterial "green"
Shape "trianglemesh" "integer indices" [ 0 1 2 0 2 3 ] "point P" [ 0 0 0  0 555 0  0 555 555  0 0 555 ]          # x = 0
NamedMaterial "red"
Shape "trianglemesh" "integer indices" [ 0 1 2 0 2 3 ] "point P" [ 555 0 0  555 555 0  555 555 555  555 0 555 ]  # x = 555
AttributeBegin
  AreaLightSource "diffuse" "rgb L" [ 17 12 4 ]
  Material "matte" "rgb Kd" [ 0.73 0.73 0.73 ]
  Shape "trianglemesh" "integer indices" [ 0 1 2 0 2 3 ] "point P" [ 213 554 227  343 554 227  343 554 332  213 554 332 ]
AttributeEnd
AttributeBegin
  Material "glass" "rgb Kt" [ 0.9 1 0.95 ]
  Translate 370 90 370
  Scale 2 2 2
  Shape "sphere" "float radius" 45
AttributeEnd
AttributeBegin
  Material "metal"
  Translate 150 60 200
  Rotate 30 0 1 0
  Shape "sphere" "float radius" [ 60 ]
AttributeEnd
AttributeBegin
  Material "plastic" "rgb Kd" [ 0.2 0.3 0.8 ]
  Translate 280 0 300
  Rotate -20 0 1 0
  Shape "trianglemesh" "integer indices" [ 0 1 2  0 2 3  0 3 1  1 3 2 ]
        "point P" [ 0 0 0   120 0 0   60 0 100   60 140 40 ]
AttributeEnd
AttributeBegin
  Material "uber"
  Shape "loopsubdiv" "integer levels" 1 "integer indices" [ 0 1 2 ] "point P" [ 0 0 0  10 0 0  0 10 0 ]
AttributeEnd
Texture "tiles" "spectrum" "checkerboard" "rgb tex1" [ 0.8 0.7 0.2 ] "rgb tex2" [ 0.1 0.1 0.1 ] "float uscale" 4 "float vscale" 4
Texture "veins" "spectrum" "marble"
AttributeBegin
  Material "matte" "texture Kd" "tiles"
  Translate 420 0 150
  Rotate -90 1 0 0
  Shape "cylinder" "float radius" 40 "float zmin" 0 "float zmax" 120 "float phimax" 270
  Translate 0 0 120
  Shape "disk" "float radius" 40 "float innerradius" 10 "float height" 0
AttributeEnd
AttributeBegin
  Material "plastic" "texture Kd" "veins"
  Translate 100 300 400
  Shape "sphere" "float radius" 30
AttributeEnd
AttributeBegin
  Material "matte" "rgb Kd" [ 0.4 0.6 0.3 ]
  Translate 300 200 450
  Scale 40 40 40
  Shape "plymesh" "string filename" "wedge.ply"
AttributeEnd
ObjectBegin "template"
  Shape "sphere" "float radius" 5
ObjectEnd
WorldEnd
'''


def ref_describe(path):
    L = C.CDLL(REF_LIB)
    L.ref_minipbrt_describe.argtypes = [C.c_char_p, C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.POINTER(C.c_float)),
                                        C.POINTER(C.c_uint)]
    L.ref_minipbrt_free.argtypes = [C.c_void_p]
    cam, film, p, n = (C.c_float * 20)(), (C.c_int * 2)(), C.POINTER(C.c_float)(), C.c_uint()
    assert L.ref_minipbrt_describe(os.fsencode(path), cam, film, C.byref(p), C.byref(n)) == 0
    try:
        shapes = np.ctypeslib.as_array(p, shape=(n.value, 48)).copy() if n.value else np.zeros((0, 48), np.float32)
    finally:
        L.ref_minipbrt_free(p)
    return np.array(cam[:], np.float32), (film[0], film[1]), shapes


# a PLY file the scene names: two triangles and a quad, normals and uv, ascii
WEDGE_PLY = """ply
format ascii 1.0
comment a wedge
element vertex 6
property float x
property float y
property float z
property float nx
property float ny
property float nz
property float u
property float v
element face 3
property list uchar int vertex_indices
end_header
0 0 0  0 -1 0  0 0
1 0 0  0 -1 0  1 0
1 0 1  0 -1 0  1 1
0 0 1  0 -1 0  0 1
0.5 1 0  0 0 -1  0.5 0
0.5 1 1  0 0 1  0.5 1
4 0 1 2 3
3 0 1 4
3 3 2 5
"""


def ref_textures(path):
    L = C.CDLL(REF_LIB)
    L.ref_minipbrt_textures.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_uint)]
    L.ref_minipbrt_free.argtypes = [C.c_void_p]
    p, n = C.POINTER(C.c_float)(), C.c_uint()
    assert L.ref_minipbrt_textures(os.fsencode(path), C.byref(p), C.byref(n)) == 0
    try:
        return np.ctypeslib.as_array(p, shape=(n.value, 8)).copy() if n.value else np.zeros((0, 8), np.float32)
    finally:
        L.ref_minipbrt_free(p)


@pytest.fixture()
def cornell_pbrt(tmp_path):
    p = tmp_path / "cornell.pbrt"
    p.write_text(CORNELL)
    (tmp_path / "wedge.ply").write_text(WEDGE_PLY)
    return str(p)


@needs_ref
def test_description_equals_minipbrt_field_for_field(cornell_pbrt):
    scene, cam, info, shapes = host.HostScene.from_pbrt(cornell_pbrt)
    rcam, rfilm, rshapes = ref_describe(cornell_pbrt)
    assert (info.xres, info.yres) == rfilm == (160, 120)
    assert info.perspective == 1 == int(rcam[19])
    assert np.float32(info.fov) == rcam[16] and np.float32(info.lensradius) == rcam[17] and np.float32(info.focaldistance) == rcam[18]
    np.testing.assert_allclose(np.array(info.camera_to_world[:], np.float32), rcam[:16], rtol=0, atol=2e-4)
    # 6 quads + 2 spheres + 1 mesh + 1 subdivision surface (not handled) + cylinder + disk + 1 sphere + 1 plymesh; no template
    assert info.n_shapes == len(rshapes) == len(shapes) == 14
    kinds = [s.kind for s in shapes]
    assert kinds.count(abi.PBRT_SHAPE_CYLINDER) == 1 and kinds.count(abi.PBRT_SHAPE_DISK) == 1 and kinds.count(abi.PBRT_SHAPE_PLYMESH) == 1
    rtex = ref_textures(cornell_pbrt)                                  # "tiles" (checkerboard), "veins" (marble), as minipbrt parsed them
    assert len(rtex) == 2 and rtex[0][0] == 1 and rtex[0][7] == 1
    n_named = n_tex = 0
    for k, (mine, ref) in enumerate(zip(shapes, rshapes)):
        assert mine.kind == int(ref[0])
        np.testing.assert_allclose(np.array(mine.shape_to_world[:], np.float32), ref[1:17], rtol=1e-6, atol=1e-4)
        if mine.kind in (abi.PBRT_SHAPE_SPHERE, abi.PBRT_SHAPE_DISK, abi.PBRT_SHAPE_CYLINDER):
            assert np.float32(mine.radius) == ref[17]
        if mine.kind in (abi.PBRT_SHAPE_TRIANGLEMESH, abi.PBRT_SHAPE_PLYMESH):     # plymesh: what minipbrt's own PLY reader loads
            assert (mine.n_vertices, mine.n_indices) == (int(ref[18]), int(ref[19])) and mine.n_indices > 0
        if mine.kind in (abi.PBRT_SHAPE_DISK, abi.PBRT_SHAPE_CYLINDER):
            assert (np.float32(mine.zmin), np.float32(mine.zmax), np.float32(mine.phimax)) == (ref[35], ref[36], ref[38])
            assert mine.kind == abi.PBRT_SHAPE_CYLINDER or np.float32(mine.innerradius) == ref[37]
        # what the colour parameter names: nothing, the checkerboard (tex1 / tex2), another texture class.  The vendored
        # minipbrt parses Texture directives but never resolves a material's reference to one (find_texture walks
        # per-attribute lists nothing appends to, minipbrt.cpp:8145-8172) -- like its named materials -- so it reports a
        # constant for every shape; the declarations themselves are compared below
        assert int(ref[28]) == 0
        if mine.texture == abi.PBRT_TEX_CHECKERBOARD:
            assert mine.kind in (abi.PBRT_SHAPE_CYLINDER, abi.PBRT_SHAPE_DISK)
            assert np.array_equal(np.array(mine.color[:], np.float32), rtex[0][1:4]) and np.array_equal(np.array(mine.tex2[:], np.float32), rtex[0][4:7])
            n_tex += 1
            continue
        if mine.texture == abi.PBRT_TEX_OTHER:
            assert rtex[1][0] == 2 and mine.material == int(ref[20]) == abi.PBRT_PLASTIC
            n_tex += 1
            continue
        if k in (3, 4):
            # NamedMaterial: the vendored minipbrt never registers named materials (find_material walks a per-attribute
            # list that nothing appends to, minipbrt.cpp:8126-8139, 6121-6144), so the directive is a no-op there and
            # these walls keep the previous material (white); this loader follows pbrt-v3 (api.cpp pbrtNamedMaterial):
            # the walls at x = 0 / 555 are green / red
            assert int(ref[20]) == abi.PBRT_MATTE and np.array_equal(ref[21:24], np.float32([0.73, 0.73, 0.73]))
            want = [0.12, 0.45, 0.15] if k == 3 else [0.65, 0.05, 0.05]
            assert mine.material == abi.PBRT_MATTE and np.array_equal(np.array(mine.color[:], np.float32), np.float32(want))
            n_named += 1
            continue
        assert mine.material == int(ref[20])
        if mine.material != abi.PBRT_OTHER and mine.texture == abi.PBRT_TEX_NONE:
            assert np.array_equal(np.array(mine.color[:], np.float32), ref[21:24])
        assert mine.emitter == int(ref[24])
        if mine.emitter:
            assert np.array_equal(np.array(mine.L[:], np.float32), ref[25:28])
    assert n_named == 2 and n_tex == 3


def test_mapping_onto_the_reference_primitives(cornell_pbrt):
    scene, cam, info, shapes = host.HostScene.from_pbrt(cornell_pbrt)
    v = scene.view
    assert info.n_unsupported_shapes == 1 and info.n_unsupported_materials == 1 and info.mis_ready == 1   # the loopsubdiv, "uber"
    assert info.n_unsupported_textures == 1                                                                # "marble" on the third sphere
    # 5 walls, then the light at squareList[5] and, being the only one, again at [6] (not in the BVH twice)
    cyl, disk, ply = (next(s for s in shapes if s.kind == k) for k in (abi.PBRT_SHAPE_CYLINDER, abi.PBRT_SHAPE_DISK, abi.PBRT_SHAPE_PLYMESH))
    assert cyl.n_indices == 6 * 48 and disk.n_indices == 6 * 64 and ply.n_indices == 12      # 270 deg = 48 of 64 steps; quad -> 2 triangles
    assert v.n_square == 7 and v.n_sphere == 3 and v.n_index == 12 + cyl.n_indices + disk.n_indices + 12
    assert v.n_vertex == 4 + cyl.n_vertices + disk.n_vertices + 6
    # the tessellations lie on the quadrics: cylinder of radius 40 around the world y axis through (420, *, 150), y in [0, 120];
    # the disk closes it at y = 120 with a hole of radius 10; normals point away from the axis / along +y
    verts = np.ctypeslib.as_array(C.cast(v.triList, C.POINTER(C.c_float)), shape=(v.n_vertex, 8))
    cv = verts[4:4 + cyl.n_vertices]
    r = np.hypot(cv[:, 0] - 420, cv[:, 2] - 150)
    assert np.allclose(r, 40, atol=1e-3) and cv[:, 1].min() > -1e-3 and cv[:, 1].max() < 120 + 1e-3
    assert np.allclose(np.hypot(cv[:, 3], cv[:, 5]), 1, atol=1e-4) and np.allclose(cv[:, 4], 0, atol=1e-5)
    assert np.allclose((cv[:, 0] - 420) * cv[:, 3] + (cv[:, 2] - 150) * cv[:, 5], 40, atol=1e-2)      # outward
    dv = verts[4 + cyl.n_vertices:4 + cyl.n_vertices + disk.n_vertices]
    rd = np.hypot(dv[:, 0] - 420, dv[:, 2] - 150)
    assert np.allclose(dv[:, 1], 120, atol=1e-3) and np.allclose(np.sort(np.unique(np.round(rd, 2))), [10, 40])
    assert np.allclose(dv[:, 3:6], [0, 1, 0], atol=1e-5)
    # total area of the disk's triangles -> pi (R^2 - r^2) from below, within the chord error of 64 steps
    tri = np.ctypeslib.as_array(C.cast(v.idxList, C.POINTER(C.c_uint32)), shape=(v.n_index,)).reshape(-1, 3)
    dt = tri[(12 + cyl.n_indices) // 3:(12 + cyl.n_indices + disk.n_indices) // 3]
    area = 0.5 * np.linalg.norm(np.cross(verts[dt[:, 1], :3] - verts[dt[:, 0], :3], verts[dt[:, 2], :3] - verts[dt[:, 0], :3]), axis=1).sum()
    assert 0.995 < area / (np.pi * (40 ** 2 - 10 ** 2)) < 1.0
    # the checkerboard became the reference's Checker texture on the cylinder / disk material (19: all triangles share it)
    mats = [v.materials[i] for i in range(v.n_material)]
    for k in (5, 6):
        m = mats[v.squareList[k].material]
        assert m.type == abi.MAT_DIFFUSE and (m.textureInfo.albedo.x, m.textureInfo.albedo.y, m.textureInfo.albedo.z) == (17, 12, 4)
    q5 = v.squareList[5]
    assert q5.axis_k == 1 and q5.value_k == 554 and (q5.range_i.x, q5.range_i.y, q5.range_j.x, q5.range_j.y) == (213, 343, 227, 332)
    assert [mats[v.squareList[i].material].type for i in range(5)] == [abi.MAT_LAMBERT] * 5
    assert (v.squareList[3].axis_k, v.squareList[3].value_k) == (0, 0.0) and (v.squareList[4].axis_k, v.squareList[4].value_k) == (0, 555.0)
    # spheres: MakeSphere (radius + 1e-4, Tracer.mm:165-172), centre from the CTM, radius times the uniform scale
    s0, s1 = v.sphereList[0], v.sphereList[1]
    assert (s0.center.x, s0.center.y, s0.center.z) == (370, 90, 370) and s0.radius == np.float32(np.float32(90) + np.float32(0.0001))
    assert mats[s0.material].type == abi.MAT_GLASS and mats[s1.material].type == abi.MAT_METAL
    assert abs(mats[s0.material].textureInfo.albedo.x - 0.9) < 1e-6
    # the tetrahedron: triangles use material 19 whatever the file says (Triangle.hh:82); 19 carries the file's material
    assert mats[19].type == abi.MAT_PLASTIC and abs(mats[19].textureInfo.albedo.z - 0.8) < 1e-6
    n_leaves = (v.n_bvh + 1) // 2
    # 6 squares in the tree (the duplicate light is not), 3 spheres, the triangles of the tetrahedron, cylinder, disk, PLY wedge
    assert n_leaves == 6 + 3 + 4 + (cyl.n_indices + disk.n_indices + 12) // 3
    # a checkerboard on a sphere keeps its own material record
    p2 = str(os.path.dirname(cornell_pbrt)) + "/tex.pbrt"
    open(p2, "w").write('WorldBegin\nTexture "c" "spectrum" "checkerboard" "rgb tex1" [ 0.9 0.2 0.1 ]\n'
                        'Material "matte" "texture Kd" "c"\nShape "sphere" "float radius" 2\n'
                        'Material "matte" "rgb Kd" [ 0.9 0.2 0.1 ]\nTranslate 5 0 0\nShape "sphere" "float radius" 2\nWorldEnd\n')
    sc2, _, _, sh2 = host.HostScene.from_pbrt(p2)
    m2 = [sc2.view.materials[i] for i in range(sc2.view.n_material)]
    a, b = m2[sc2.view.sphereList[0].material], m2[sc2.view.sphereList[1].material]
    assert a.textureInfo.type == abi.TEX_CHECKER and b.textureInfo.type == abi.TEX_CONSTANT and sh2[0].mapped_material != sh2[1].mapped_material
    assert abs(a.textureInfo.albedo.x - 0.9) < 1e-6 and sh2[0].texture == abi.PBRT_TEX_CHECKERBOARD and list(sh2[0].tex2) == [0, 0, 0]
    # camera: MakeCamera from the LookAt triple; fov over the shorter (vertical) axis; aperture = 2 * lensradius
    assert (cam.lookFrom.x, cam.lookFrom.y, cam.lookFrom.z) == (278, 273, -800) and cam.lenRadius == 0.5
    assert abs(cam.vfov - np.deg2rad(39)) < 1e-6 and abs(cam.aspect - 160 / 120) < 1e-6 and cam.focus_dist == 1000


def test_loaded_scene_renders_on_the_oracle(cornell_pbrt):
    scene, cam, info, shapes = host.HostScene.from_pbrt(cornell_pbrt)
    W, H = info.xres // 2, info.yres // 2
    for integ in (abi.INTEGRATOR_PATH, abi.INTEGRATOR_MIS):
        acc, st = po.render(scene.view, cam, W, H, host.fill_rng(5, W, H), spp=4, integrator=integ)
        assert st.rays > W * H * 4 and np.isfinite(acc).all() and acc[..., :3].max() > 0.5
        assert st.n_leaf_sphere > 0 and st.n_leaf_triangle > 0 and st.n_leaf_square > 0
    # primary rays see what the file describes: the light in the ceiling, both spheres, the tetrahedron
    from conftest import camera_rays
    hits = po.trace_rays(scene.view, camera_rays(host.make_camera((278, 273, -800), (278, 273, 0), (0, 1, 0), 0.0, W / H, np.deg2rad(39), 10.0), W, H))
    seen = set(zip(hits["pType"][hits["hit"] != 0].tolist(), hits["pIndex"][hits["hit"] != 0].tolist()))
    assert (abi.PRIM_SQUARE, 5) in seen and (abi.PRIM_SPHERE, 0) in seen and (abi.PRIM_SPHERE, 1) in seen
    assert any(t == abi.PRIM_TRIANGLE for t, _ in seen)


def test_rejects_and_reports(tmp_path):
    p = tmp_path / "bad.pbrt"
    p.write_text('WorldBegin\nShape "trianglemesh" "point P" [ 0 0 0 1 0 0 ]\nWorldEnd\n')
    with pytest.raises(RuntimeError):
        host.HostScene.from_pbrt(str(p))
    p.write_text('WorldBegin\nShape "sphere"\nWorldEnd\n')               # a single leaf cannot make a tree
    with pytest.raises(RuntimeError):
        host.HostScene.from_pbrt(str(p))
    with pytest.raises(RuntimeError):
        host.HostScene.from_pbrt(str(tmp_path / "missing.pbrt"))


QUADRICS = """LookAt 0 5 -20  0 2 0  0 1 0
Camera "perspective" "float fov" [ 45 ]
Film "image" "integer xresolution" [ 64 ] "integer yresolution" [ 48 ]
WorldBegin
Material "matte" "rgb Kd" [ 0.5 0.5 0.5 ]
AttributeBegin
  Translate -6 0 0
  Rotate -90 1 0 0
  Shape "cone" "float radius" 2 "float height" 5 "float phimax" 180
AttributeEnd
AttributeBegin
  Translate 0 0 0
  Rotate -90 1 0 0
  Shape "paraboloid" "float radius" 3 "float zmin" 1 "float zmax" 4
AttributeEnd
AttributeBegin
  Translate 6 0 0
  Rotate -90 1 0 0
  Shape "hyperboloid" "point p1" [ 1 0 0 ] "point p2" [ 2 1 3 ] "float phimax" 270
AttributeEnd
Shape "cone" "float radius" 0
WorldEnd
"""


@pytest.fixture()
def quadrics_pbrt(tmp_path):
    p = tmp_path / "quadrics.pbrt"
    p.write_text(QUADRICS)
    return str(p)


@needs_ref
def test_quadrics_equal_minipbrt_field_for_field(quadrics_pbrt):
    scene, cam, info, shapes = host.HostScene.from_pbrt(quadrics_pbrt)
    rcam, rfilm, rshapes = ref_describe(quadrics_pbrt)
    assert info.n_shapes == len(rshapes) == len(shapes) == 4
    assert [s.kind for s in shapes] == [abi.PBRT_SHAPE_CONE, abi.PBRT_SHAPE_PARABOLOID, abi.PBRT_SHAPE_HYPERBOLOID, abi.PBRT_SHAPE_CONE]
    for mine, ref in zip(shapes, rshapes):
        assert mine.kind == int(ref[0])
        np.testing.assert_allclose(np.array(mine.shape_to_world[:], np.float32), ref[1:17], rtol=1e-6, atol=1e-4)
        assert np.float32(mine.phimax) == ref[38]
        if mine.kind == abi.PBRT_SHAPE_HYPERBOLOID:
            assert np.array_equal(np.array(mine.p1[:], np.float32), ref[39:42]) and np.array_equal(np.array(mine.p2[:], np.float32), ref[42:45])
        else:
            assert (np.float32(mine.radius), np.float32(mine.zmin), np.float32(mine.zmax)) == (ref[17], ref[35], ref[36])
    assert info.n_unsupported_shapes == 1 and shapes[3].mapped_type == -1               # a cone of radius 0 is no surface


def test_quadric_tessellations_lie_on_their_surfaces(quadrics_pbrt):
    """pbrt-v3 shapes/cone.cpp, paraboloid.cpp, hyperboloid.cpp: every vertex satisfies the quadric's equation in object space,
    the normals are unit and perpendicular to the surface's tangents, the vertex counts follow the grid"""
    scene, cam, info, shapes = host.HostScene.from_pbrt(quadrics_pbrt)
    v = scene.view
    verts = np.ctypeslib.as_array(C.cast(v.triList, C.POINTER(C.c_float)), shape=(v.n_vertex, 8)).copy()
    cone, par, hyp = shapes[0], shapes[1], shapes[2]
    assert cone.n_vertices == (32 + 1) * 2 and cone.n_indices == 3 * 32                       # 180 deg = 32 of 64 steps, apex cells are one triangle
    assert par.n_vertices == 65 * 17 and par.n_indices == 6 * 64 * 16 and hyp.n_vertices == 49 * 9 and hyp.n_indices == 6 * 48 * 8
    assert v.n_vertex == cone.n_vertices + par.n_vertices + hyp.n_vertices and v.n_index == cone.n_indices + par.n_indices + hyp.n_indices
    off = 0
    for sh in (cone, par, hyp):
        M = np.array(sh.shape_to_world[:], np.float64).reshape(4, 4)
        W = verts[off:off + sh.n_vertices]; off += sh.n_vertices
        P = (np.linalg.inv(M) @ np.c_[W[:, :3], np.ones(len(W))].T).T[:, :3]               # back to object space
        N = (M[:3, :3].T @ W[:, 3:6].T).T                                                   # normals transform by the inverse transpose
        N /= np.linalg.norm(N, axis=1, keepdims=True)
        r = np.hypot(P[:, 0], P[:, 1])
        if sh is cone:
            assert np.allclose(r, 2.0 * (1.0 - P[:, 2] / 5.0), atol=1e-4) and P[:, 2].min() >= -1e-5 and P[:, 2].max() <= 5 + 1e-4
            assert np.allclose(N[:, 2], 2.0 / np.hypot(2.0, 5.0), atol=1e-4)                  # constant slope
            assert P[:, 1].min() >= -1e-4 and P[:, 0].min() < -1.9 and P[:, 0].max() > 1.9     # phimax 180: the y >= 0 half, all of it
        elif sh is par:
            assert np.allclose(P[:, 2], 4.0 * (r / 3.0) ** 2, atol=2e-4) and abs(P[:, 2].min() - 1) < 1e-5 and abs(P[:, 2].max() - 4) < 1e-5
            g = np.c_[2 * 4.0 * P[:, 0] / 9.0, 2 * 4.0 * P[:, 1] / 9.0, -np.ones(len(P))]
            assert np.allclose(N, g / np.linalg.norm(g, axis=1, keepdims=True), atol=1e-4)
        else:
            # a point of the swept segment: undo the rotation -> it lies on p1 + v (p2 - p1); x^2 + y^2 = |p(v).xy|^2 at its z
            vpar = P[:, 2] / 3.0
            want = np.hypot(1.0 + vpar * 1.0, vpar * 1.0)
            assert np.allclose(r, want, atol=2e-4) and P[:, 2].min() >= -1e-5 and P[:, 2].max() <= 3 + 1e-4
            assert np.allclose(np.linalg.norm(N, axis=1), 1.0, atol=1e-5)
    # the tessellations render: the scene builds a tree over 3 x their triangles
    assert scene.tree_depth() > 3


def test_a_paraboloid_from_its_apex_has_no_zero_area_triangles(tmp_path):
    """with zmin = 0 (pbrt's default) row 0 of the grid is the apex, ONE point: its cells are one triangle each, like
    the cone's apex cells -- not a second, zero-area triangle that becomes a leaf nothing can ever hit"""
    p = tmp_path / "apex.pbrt"
    p.write_text(QUADRICS.replace('"float radius" 3 "float zmin" 1 "float zmax" 4', '"float radius" 3 "float zmax" 4'))
    scene, cam, info, shapes = host.HostScene.from_pbrt(str(p))
    par = shapes[1]
    assert par.kind == abi.PBRT_SHAPE_PARABOLOID and par.zmin == 0.0 and par.n_vertices == 65 * 17
    assert par.n_indices == 3 * 64 * (2 * 16 - 1)                     # the apex row: one triangle per cell
    v = scene.view
    verts = np.ctypeslib.as_array(C.cast(v.triList, C.POINTER(C.c_float)), shape=(v.n_vertex, 8))[:, :3].astype(np.float64)
    idx = np.ctypeslib.as_array(v.idxList, shape=(v.n_index,)).reshape(-1, 3)
    a, b, c = verts[idx[:, 0]], verts[idx[:, 1]], verts[idx[:, 2]]
    area = 0.5 * np.linalg.norm(np.cross(b - a, c - a), axis=1)
    assert (area > 1e-6).all()


INSTANCED = '''
LookAt 278 278 -800  278 278 0  0 1 0
Camera "perspective" "float fov" 40
Film "image" "integer xresolution" 160 "integer yresolution" 120
WorldBegin
AttributeBegin
  AreaLightSource "diffuse" "rgb L" [ 15 15 15 ]
  Shape "trianglemesh" "integer indices" [ 0 1 2 0 2 3 ] "point P" [ 213 554 227  343 554 227  343 554 332  213 554 332 ]
AttributeEnd
AttributeBegin
  Translate 7 0 0
  ObjectBegin "gadget"
    Material "glass" "rgb Kt" [ 0.9 0.9 0.9 ]
    Shape "sphere" "float radius" 5
    AttributeBegin
      Material "matte" "rgb Kd" [ 0.2 0.4 0.6 ]
      Translate 0 10 0
      Scale 2 3 4
      Shape "trianglemesh" "integer indices" [ 0 1 2 ] "point P" [ 0 0 0  1 0 0  0 1 1 ] "normal N" [ 0 0 1  0 0 1  0 0 1 ]
    AttributeEnd
  ObjectEnd
AttributeEnd
Material "matte" "rgb Kd" [ 0.7 0.7 0.7 ]
AttributeBegin
  Translate 100 50 200
  ObjectInstance "gadget"
AttributeEnd
Shape "trianglemesh" "integer indices" [ 0 1 2 0 2 3 ] "point P" [ 0 0 0  555 0 0  555 0 555  0 0 555 ]
AttributeBegin
  Translate 300 80 300
  Rotate 90 0 1 0
  Scale 4 4 4
  ObjectInstance "gadget"
AttributeEnd
ObjectInstance "nothing of that name"
WorldEnd
'''


def ref_instances(path):
    L = C.CDLL(REF_LIB)
    L.ref_minipbrt_instances.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_uint)]
    L.ref_minipbrt_free.argtypes = [C.c_void_p]
    p, n = C.POINTER(C.c_float)(), C.c_uint()
    assert L.ref_minipbrt_instances(os.fsencode(path), C.byref(p), C.byref(n)) == 0
    try:
        return np.ctypeslib.as_array(p, shape=(n.value, 64)).copy() if n.value else np.zeros((0, 64), np.float32)
    finally:
        L.ref_minipbrt_free(p)


@needs_ref
def test_object_instances_are_placed_as_flat_copies(tmp_path):
    """ObjectBegin / ObjectInstance (pbrt-v3 api.cpp; minipbrt.h:1417-1436 Object / Instance): every instance becomes a copy of
    its template's shapes at instanceToWorld x shapeToWorld, after the world's own shapes, in file order -- field for field
    against what the reference's parser holds for the templates and the instances."""
    path = tmp_path / "instanced.pbrt"
    path.write_text(INSTANCED)
    scene, cam, info, shapes = host.HostScene.from_pbrt(str(path))
    _, _, rworld = ref_describe(str(path))
    rinst = ref_instances(str(path))
    assert len(rworld) == 2 and len(rinst) == 4                       # 2 world meshes; 2 instances x (sphere + mesh)
    assert info.n_instances == 3 and info.n_shapes == len(shapes) == 6 and info.n_unsupported_shapes == 1    # the instance of nothing
    for mine, ref in zip(shapes[:2], rworld):
        assert mine.kind == int(ref[0])
        np.testing.assert_allclose(np.array(mine.shape_to_world[:], np.float32), ref[1:17], rtol=1e-6, atol=1e-4)
    for mine, ref in zip(shapes[2:], rinst):
        assert mine.kind == int(ref[0])
        want = ref[48:64].reshape(4, 4).astype(np.float64) @ ref[1:17].reshape(4, 4).astype(np.float64)
        np.testing.assert_allclose(np.array(mine.shape_to_world[:], np.float64).reshape(4, 4), want, rtol=1e-5, atol=1e-3)
        assert mine.material == int(ref[20]) and np.array_equal(np.array(mine.color[:], np.float32), ref[21:24])
        if mine.kind == abi.PBRT_SHAPE_SPHERE:
            assert np.float32(mine.radius) == ref[17] == 5
        else:
            assert (mine.n_vertices, mine.n_indices) == (int(ref[18]), int(ref[19])) == (3, 3)
    v = scene.view
    # the spheres: centre = the placed origin, radius scaled by the instance (1 and 4)
    assert v.n_sphere == 2
    c0, c1 = v.sphereList[0], v.sphereList[1]
    m0 = rinst[0][48:64].reshape(4, 4).astype(np.float64) @ rinst[0][1:17].reshape(4, 4).astype(np.float64)
    m1 = rinst[2][48:64].reshape(4, 4).astype(np.float64) @ rinst[2][1:17].reshape(4, 4).astype(np.float64)
    np.testing.assert_allclose([c0.center.x, c0.center.y, c0.center.z], m0[:3, 3], atol=1e-3)
    np.testing.assert_allclose([c1.center.x, c1.center.y, c1.center.z], m1[:3, 3], atol=1e-3)
    assert abs(c0.radius - 5.0) < 1e-3 and abs(c1.radius - 20.0) < 1e-2
    # the template's triangle, placed twice: vertices = instanceToWorld x shapeToWorld x P of the source text
    P = np.array([[0, 0, 0, 1], [1, 0, 0, 1], [0, 1, 1, 1]], np.float64)
    tri = [s for s in shapes[2:] if s.kind == abi.PBRT_SHAPE_TRIANGLEMESH]
    for s, ref in zip(tri, (rinst[1], rinst[3])):
        m = ref[48:64].reshape(4, 4).astype(np.float64) @ ref[1:17].reshape(4, 4).astype(np.float64)
        first = 3 * s.mapped_index
        got = np.array([[v.triList[v.idxList[first + k]].v[a] for a in range(3)] for k in range(3)])
        np.testing.assert_allclose(got, (P @ m.T)[:, :3], rtol=1e-5, atol=1e-3)
    # and the whole thing is a scene the oracle can walk: a ray down onto the second copy's sphere hits it
    from oracle import pyoracle as po
    rays = po.make_rays(np.array([[m1[0, 3], m1[1, 3] + 100.0, m1[2, 3]]], np.float32), np.array([[0, -1, 0]], np.float32))
    h = po.trace_rays(v, rays)
    assert h["hit"][0] == 1 and h["pType"][0] == abi.PRIM_SPHERE and abs(h["t"][0] - 80.0) < 1e-2
