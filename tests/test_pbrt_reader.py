"""trc_host_load_density_pbrt against the REFERENCE's own parser: RT_Metal/Tracer/minipbrt.cpp compiled where it lies
into oracle/_ref/libminipbrt_ref.so (oracle/Makefile, oracle/ref_minipbrt_shim.cpp), doing what AAPLRenderer.mm:629-636
does.  On the reference's cloud (cloud/cloud.pbrt -> Include geometry/density_render.70.pbrt, 100 x 100 x 40) when the
reference tree is mounted, and on generated files everywhere the library exists."""
import ctypes as C
import os

import numpy as np
import pytest

from tracer_amd import host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_LIB = os.path.join(ROOT, "oracle", "_ref", "libminipbrt_ref.so")
REF_CLOUD = "/root/reference/RT_Metal/cloud"


def ref_load(path):
    L = C.CDLL(REF_LIB)
    L.ref_minipbrt_load_density.argtypes = [C.c_char_p] + [C.POINTER(C.c_int)] * 3 + [C.POINTER(C.POINTER(C.c_float))]
    L.ref_minipbrt_free.argtypes = [C.c_void_p]
    nx, ny, nz, p = C.c_int(), C.c_int(), C.c_int(), C.POINTER(C.c_float)()
    rc = L.ref_minipbrt_load_density(os.fsencode(path), C.byref(nx), C.byref(ny), C.byref(nz), C.byref(p))
    if rc != 0:
        return None
    try:
        return np.ctypeslib.as_array(p, shape=(nz.value, ny.value, nx.value)).copy()
    finally:
        L.ref_minipbrt_free(p)


needs_ref = pytest.mark.skipif(not os.path.exists(REF_LIB), reason="oracle/_ref/libminipbrt_ref.so not built (reference absent)")


@needs_ref
@pytest.mark.skipif(not os.path.isdir(REF_CLOUD), reason="reference tree not mounted")
@pytest.mark.parametrize("name", ["cloud.pbrt", "geometry/density_render.70.pbrt"])
def test_reference_cloud_reads_like_minipbrt(name):
    path = os.path.join(REF_CLOUD, name)
    ref = ref_load(path)
    mine = host.load_density_pbrt(path)
    assert ref is not None and ref.shape == (40, 100, 100) == mine.shape
    assert np.array_equal(mine.view(np.uint32), ref.view(np.uint32))          # every float, bit for bit
    assert ref.max() == 1.0 and 0.04 < ref.mean() < 0.05


MEDIUM = '''MakeNamedMedium "smoke" "string type" "heterogeneous" "integer nx" {nx} "integer ny" [ {ny} ] "integer nz" {nz}
\t"point p0" [ 0.01 0.01 0.01 ] "point p1" [ 1.99 1.99 0.79 ]   # trailing comment "integer nx" 99
\t"float density" [
{values} ]
'''


def write_scene(tmp_path, values, nx, ny, nz, include=False):
    body = MEDIUM.format(nx=nx, ny=ny, nz=nz, values=values)
    if not include:
        p = tmp_path / "medium.pbrt"
        p.write_text("# a comment first\nWorldBegin\n" + body + "WorldEnd\n")
        return str(p)
    (tmp_path / "geometry").mkdir()
    (tmp_path / "geometry" / "grid.pbrt").write_text(body)
    p = tmp_path / "top.pbrt"
    p.write_text('LookAt 0 0 5  0 0 0  0 1 0\nCamera "perspective" "float fov" [15]\nWorldBegin\n'
                 '#Include "geometry/missing.pbrt"\nTransformBegin\n\tInclude "geometry/grid.pbrt"\n'
                 '\t  "color sigma_a" [10 10 10] "color sigma_s" [90 90 90]\nTransformEnd\nWorldEnd\n')
    return str(p)


@pytest.mark.parametrize("include", [False, True])
def test_generated_files(tmp_path, include):
    nx, ny, nz = 5, 4, 3
    tokens = ["0", "1", ".5", "-0", "+2.5e+1", "1e-3", "3.", "0.1", "7.0E2", "1e-45", "0.30000001192092896", "16777217"]
    vals = [tokens[i % len(tokens)] for i in range(nx * ny * nz)]
    text = "\n".join(" ".join(vals[r * nx:(r + 1) * nx]) for r in range(ny * nz))
    path = write_scene(tmp_path, text, nx, ny, nz, include)
    mine = host.load_density_pbrt(path)
    assert mine.shape == (nz, ny, nx)
    want = np.array([np.float32(float(t)) for t in vals], dtype=np.float32).reshape(nz, ny, nx)
    assert np.array_equal(mine.view(np.uint32), want.view(np.uint32))
    if os.path.exists(REF_LIB):
        ref = ref_load(path)
        assert ref is not None and np.array_equal(mine.view(np.uint32), ref.view(np.uint32))


def test_reader_errors(tmp_path):
    p = tmp_path / "short.pbrt"
    p.write_text(MEDIUM.format(nx=2, ny=2, nz=2, values="1 2 3"))                  # 3 of 8 values
    with pytest.raises(Exception):
        host.load_density_pbrt(str(p))
    q = tmp_path / "loop.pbrt"
    q.write_text('Include "loop.pbrt"\n')                                          # include cycle
    with pytest.raises(Exception):
        host.load_density_pbrt(str(q))
    with pytest.raises(Exception):
        host.load_density_pbrt(str(tmp_path / "absent.pbrt"))


# ---------------------------------------------------------------- Shape "trianglemesh" (trc_host_mesh_load_pbrt)
SCENE = '''# generated test scene
LookAt 3 4 1.5  .5 .5 0  0 0 1
Camera "perspective" "float fov" [45]
Film "image" "integer xresolution" [64] "integer yresolution" [48] "string filename" "x.exr"
WorldBegin
LightSource "point" "rgb I" [ 1 1 1 ]
AttributeBegin
  Material "matte" "rgb Kd" [ .5 .5 .5 ]
  Translate 1 2 3
  Rotate 37 0.3 1 -0.2
  Scale 2 0.5 1.5
  Shape "trianglemesh" "integer indices" [0 1 2  0 2 3] "point P" [ 0 0 0  1 0 0  1 1 0  0 1 0 ]
        "normal N" [ 0 0 1  0 0 1  0 0 1  0.1 0 1 ] "float uv" [ 0 0 1 0 1 1 0 1 ]
  TransformBegin
    ConcatTransform [ 1 0 0 0  0 1 0 0  0 0 1 0  -4 5 6 1 ]
    Shape "trianglemesh" "integer indices" [0 1 2] "point P" [ 0 0 1  1 0 1  0 1 1 ]      # no normals: generated
  TransformEnd
  Shape "sphere" "float radius" [ 2.5 ]
AttributeEnd
ObjectBegin "template"
  Shape "trianglemesh" "integer indices" [0 1 2] "point P" [ 9 9 9  8 9 9  9 8 9 ]        # definition only: skipped
ObjectEnd
AttributeBegin
  CoordSysTransform "camera"
  Translate 0 0 10
  Shape "trianglemesh" "integer indices" [ 0 1 2 ] "point P" [ -1 -1 0  1 -1 0  0 1 0 ]
AttributeEnd
Transform [ 0 1 0 0  -1 0 0 0  0 0 1 0  0 0 0 1 ]
Include "more.pbrt"
WorldEnd
'''
MORE = 'Shape "trianglemesh" "integer indices" [ 2 1 0 ] "point P" [ 1 0 0  0 1 0  0 0 1 ] "float st" [ .1 .2 .3 .4 .5 .6 ]\n'


def ref_meshes(path):
    L = C.CDLL(REF_LIB)
    fp, up = C.POINTER(C.c_float), C.POINTER(C.c_uint)
    L.ref_minipbrt_triangle_meshes.argtypes = [C.c_char_p, C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.POINTER(fp), C.POINTER(fp),
                                               C.POINTER(fp), C.POINTER(fp), C.POINTER(up)]
    L.ref_minipbrt_free.argtypes = [C.c_void_p]
    nv, ni = C.c_uint(), C.c_uint()
    P, N, uv, M, I = fp(), fp(), fp(), fp(), up()
    assert L.ref_minipbrt_triangle_meshes(os.fsencode(path), C.byref(nv), C.byref(ni), C.byref(P), C.byref(N), C.byref(uv),
                                          C.byref(M), C.byref(I)) == 0
    arr = lambda p, shape: np.ctypeslib.as_array(p, shape=shape).copy()
    out = (arr(P, (nv.value, 3)), arr(N, (nv.value, 3)), arr(uv, (nv.value, 2)), arr(M, (nv.value, 4, 4)), arr(I, (ni.value,)))
    for p in (P, N, uv, M, I):
        L.ref_minipbrt_free(p)
    return out


def mesh_arrays(mesh):
    v = np.ctypeslib.as_array(C.cast(mesh.vertices_ptr, C.POINTER(C.c_float)), shape=(mesh.n_vertices, 8)).copy()
    i = np.ctypeslib.as_array(C.cast(mesh.indices_ptr, C.POINTER(C.c_uint32)), shape=(mesh.n_indices,)).copy()
    return v, i


def test_pbrt_triangle_meshes(tmp_path):
    (tmp_path / "scene.pbrt").write_text(SCENE)
    (tmp_path / "more.pbrt").write_text(MORE)
    mesh = host.Mesh.load_pbrt(str(tmp_path / "scene.pbrt"))
    v, idx = mesh_arrays(mesh)
    assert v.shape == (4 + 3 + 3 + 3, 8) and list(idx) == [0, 1, 2, 0, 2, 3, 4, 5, 6, 7, 8, 9, 12, 11, 10]
    # last shape: Transform replaced the CTM by a 90 degree turn about z (column-major input): (x, y, z) -> (-y, x, z)
    assert np.allclose(v[10:, :3], [[0, 1, 0], [-1, 0, 0], [0, 0, 1]], atol=1e-6)
    assert np.allclose(v[10:, 6:], [[.1, .2], [.3, .4], [.5, .6]])
    # CoordSysTransform "camera": the matrix that was current at `Camera` (the reference's parser, minipbrt.cpp:7056-7057;
    # pbrt-v3 proper would use its inverse), i.e. the LookAt matrix: p = LookAt * Translate(0, 0, 10) * P
    eye, look, up = np.array([3, 4, 1.5]), np.array([.5, .5, 0]), np.array([0, 0, 1.0])
    d = (look - eye) / np.linalg.norm(look - eye)
    r = np.cross(up, d); r /= np.linalg.norm(r)
    w2c = np.array([r, np.cross(d, r), d])
    local = np.array([[-1, -1, 10], [1, -1, 10], [0, 1, 10.0]])
    assert np.allclose(v[7:10, :3], (local - eye) @ w2c.T, atol=1e-4)
    # generated normals of the shape without N: unit length, perpendicular to its triangle
    e1, e2 = v[5, :3] - v[4, :3], v[6, :3] - v[4, :3]
    for n in v[4:7, 3:6]:
        assert abs(np.linalg.norm(n) - 1) < 1e-5 and abs(n @ e1) < 1e-5 and abs(n @ e2) < 1e-5
    # a scene can be built from it
    from tracer_amd import abi
    sc = host.HostScene(abi.SCENE_CORNELL_MESH, mesh)
    assert sc.view.n_index == 15 and sc.view.n_vertex == 13
    if not os.path.exists(REF_LIB):
        return
    P, N, uv, M, I = ref_meshes(str(tmp_path / "scene.pbrt"))                 # the reference's parser on the same file
    assert P.shape[0] == 13 and list(I) == list(idx)
    world = np.einsum("vij,vj->vi", M[:, :3, :3].astype(np.float64), P.astype(np.float64)) + M[:, :3, 3]
    assert np.allclose(v[:, :3], world, rtol=2e-5, atol=2e-5)
    assert np.array_equal(v[:10, 6:], uv[:10])         # (minipbrt does not pick up pbrt-v3's alternative spelling `st`)
    inv_t = np.linalg.inv(M.astype(np.float64)).transpose(0, 2, 1)
    want_n = np.einsum("vij,vj->vi", inv_t[:4, :3, :3], N[:4].astype(np.float64))
    assert np.allclose(v[:4, 3:6], want_n, rtol=2e-5, atol=2e-5)


def test_pbrt_lone_triangle_without_indices(tmp_path):
    """pbrt-v3 lets a three-vertex mesh omit its indices (minipbrt does not: not part of the comparison)"""
    (tmp_path / "t.pbrt").write_text('Translate 1 0 0\nShape "trianglemesh" "point P" [ 0 0 0  1 0 0  0 1 0 ]\n')
    v, idx = mesh_arrays(host.Mesh.load_pbrt(str(tmp_path / "t.pbrt")))
    assert list(idx) == [0, 1, 2] and np.array_equal(v[:, :3], [[1, 0, 0], [2, 0, 0], [1, 1, 0]])
    assert np.allclose(v[:, 3:6], [[0, 0, 1]] * 3)


def test_pbrt_mesh_errors(tmp_path):
    for name, text in (("none.pbrt", 'WorldBegin\nShape "sphere" "float radius" 1\nWorldEnd\n'),
                       ("range.pbrt", 'Shape "trianglemesh" "integer indices" [0 1 3] "point P" [0 0 0 1 0 0 0 1 0]\n'),
                       ("stack.pbrt", 'AttributeEnd\nShape "trianglemesh" "point P" [0 0 0 1 0 0 0 1 0]\n')):
        (tmp_path / name).write_text(text)
        with pytest.raises(Exception):
            host.Mesh.load_pbrt(str(tmp_path / name))
