"""Host scene construction against the reference's constants."""
import ctypes as C
import math

import numpy as np

from tracer_amd import abi, host


def test_material_table_order_and_types(cornell, cornell_spheres):
    m = [cornell.view.materials[i] for i in range(cornell.view.n_material)]
    assert len(m) == 20
    assert [x.type for x in m[:7]] == [abi.MAT_METAL, abi.MAT_GLASS, abi.MAT_NIL, abi.MAT_DIFFUSE, abi.MAT_LAMBERT,
                                       abi.MAT_LAMBERT, abi.MAT_LAMBERT]
    assert m[3].textureInfo.albedo.x == 11.0                                   # emitter Le = 11 (Tracer.mm:247-251)
    assert (round(m[4].textureInfo.albedo.x, 2), round(m[5].textureInfo.albedo.y, 2)) == (0.65, 0.65)
    assert m[6].textureInfo.type == abi.TEX_CHECKER and abs(m[6].textureInfo.albedo.x - 0.73) < 1e-6
    assert m[7].type == abi.MAT_DIELECTRIC and all(x.type == abi.MAT_DEMOFOX for x in m[8:19])   # as shipped
    assert m[19].type == abi.MAT_GLASS and m[19].medium == 1 and m[19].specular == 1 and m[19].eta == 1.5
    # config 2 remaps the unsupported sphere materials to the four supported lobes
    m2 = [cornell_spheres.view.materials[i] for i in range(20)]
    assert {x.type for x in m2[7:19]} == {abi.MAT_LAMBERT, abi.MAT_PLASTIC, abi.MAT_METAL, abi.MAT_GLASS}
    assert all(x.textureInfo.type == abi.TEX_CONSTANT for x in m2[7:19])


def test_cornell_squares(cornell):
    q = [cornell.view.squareList[i] for i in range(7)]
    assert [(s.axis_k, s.value_k) for s in q[:5]] == [(0, -245.0), (0, 800.0), (1, 555.0), (2, 555.0), (1, 0.0)]
    assert q[5].axis_k == 1 and abs(q[5].value_k - 554.9) < 1e-3 and (q[5].range_i.x, q[5].range_i.y) == (400.0, 555.0)
    assert q[6].axis_k == 0 and q[6].value_k == -300.0                          # the small side light
    assert [s.material for s in q] == [5, 4, 6, 6, 6, 3, 3]
    # AABB padded by 1/512 on the thin axis (Square.hh:7-9, Tracer.mm:137-149)
    assert q[4].boundingBOX.mini.y == -1 / 512 and q[4].boundingBOX.maxi.y == 1 / 512


def test_cubes_and_spheres(cornell):
    c = [cornell.view.cubeList[i] for i in range(3)]
    assert [x.material for x in c] == [0, 19, 2]
    col3 = c[0].model_matrix.columns[3]
    assert (col3.x, col3.y, col3.z, col3.w) == (265.0, 1.0, 295.0, 1.0)
    # normal_matrix = transpose(inverse(model)); inverse * model = I
    M = np.array([[getattr(c[0].model_matrix.columns[j], "xyzw"[i]) for j in range(4)] for i in range(4)])
    I = np.array([[getattr(c[0].inverse_matrix.columns[j], "xyzw"[i]) for j in range(4)] for i in range(4)])
    N = np.array([[getattr(c[0].normal_matrix.columns[j], "xyzw"[i]) for j in range(4)] for i in range(4)])
    assert np.allclose(I @ M, np.eye(4), atol=1e-5) and np.array_equal(N, I.T)
    s = [cornell.view.sphereList[i] for i in range(12)]
    assert abs(s[0].radius - 64.0001) < 1e-4 and (s[0].center.x, s[0].center.y, s[0].center.z) == (200.0, 250.0, 200.0)
    assert s[0].boundingBOX.maxi.x == 264.0                                     # AABB NOT inflated (B-13)
    assert [x.center.x for x in s[1:7]] == [500.0, 400.0, 300.0, 200.0, 100.0, 0.0]
    assert [x.center.x for x in s[7:12]] == [-10.0, 140.0, 290.0, 440.0, 590.0]
    assert [x.material for x in s] == list(range(7, 19))


def test_camera_defaults():
    cam = host.prepare_camera(1920, 1080)
    assert (cam.lookFrom.x, cam.lookFrom.y, cam.lookFrom.z) == (278.0, 278.0, -800.0)
    assert cam.aperture == 0.0 and cam.lenRadius == 0.0 and cam.focus_dist == 10.0
    assert abs(cam.vfov - math.radians(45)) < 1e-7 and abs(cam.aspect - 16 / 9) < 1e-6
    assert (cam.w.x, cam.w.y, cam.w.z) == (0.0, 0.0, -1.0) and (cam.u.x, cam.v.y) == (-1.0, 1.0)
    assert abs(cam.vertical.y - 2 * math.tan(math.radians(22.5)) * 10) < 1e-5


def test_tiny_obj_roundtrip(tmp_path):
    p = tmp_path / "quad.obj"
    p.write_text("# quad\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\n"
                 "f 1/1 2/2 3/3 4/4\nf -4 -3 -2\n")
    mesh = host.Mesh.load_obj(str(p))
    assert mesh.n_triangles == 3                        # the quad fans into 2, plus the negative-index triangle
    v = mesh.vertices()
    assert np.allclose(np.linalg.norm(v[:, 3:6], axis=1), 1)      # smooth normals were generated (no vn in the file)
    assert np.allclose(np.abs(v[:, 5]), 1)


def test_ball_and_replicate():
    ball = host.Mesh.ball(20, 30, 0.1)
    assert ball.n_triangles == 20 * 30 * 2
    grid = ball.replicate(3, 2.5)
    assert grid.n_triangles == 9 * ball.n_triangles and grid.n_vertices == 9 * ball.n_vertices
    assert grid.indices().max() == grid.n_vertices - 1


def test_scene_with_analytic_leaves_only():
    """trc_host_scene_create_leaves(analytic_leaves_only): the same arrays, bvhList = the analytic primitives' leaf records (the
    first leaves of the full scene), no per-triangle leaves, no tree -- the input of trc_upload_scene_device."""
    mesh = host.Mesh.ball(20, 20, 1.0)
    full = host.HostScene(abi.SCENE_CORNELL_MESH, mesh)
    lean = host.HostScene(abi.SCENE_CORNELL_MESH, mesh, analytic_leaves_only=True)
    n_tri = full.view.n_index // 3
    assert lean.view.n_bvh == full.n_leaves - n_tri == 9 and lean.view.n_index == full.view.n_index
    want = full.bvh_array()[1:1 + lean.view.n_bvh].copy()
    want[:, 0] = 0                                                   # the full scene's leaves carry their parents
    assert np.array_equal(lean.bvh_array(), want)
    a = np.frombuffer((C.c_char * (32 * full.view.n_vertex)).from_address(C.addressof(full.view.triList.contents)), dtype=np.uint8)
    b = np.frombuffer((C.c_char * (32 * lean.view.n_vertex)).from_address(C.addressof(lean.view.triList.contents)), dtype=np.uint8)
    assert np.array_equal(a, b)                                      # the placed mesh is the same
    plain = host.HostScene(abi.SCENE_CORNELL_SPHERES, analytic_leaves_only=True)
    assert plain.view.n_bvh == 21
