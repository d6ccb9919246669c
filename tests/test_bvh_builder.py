"""Host SAH BVH builder (BVH.hh:35-314): structural invariants, agreement with brute force,
determinism under the parallel build, and the reference's array layout after buildTree."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import camera_rays, random_rays
from oracle import pyoracle as po
from tracer_amd import abi, host

REF_TEAPOT = "/root/reference/RT_Metal/meshes/teapot.obj"


def check_invariants(scene):
    nodes = scene.bvh()
    n = len(nodes)
    n_leaves = (n + 1) // 2
    assert nodes[0].pType == abi.PRIM_BVH and nodes[0].parent == 0          # root at 0 (BVH.hh:263-268)
    # leaves sit at 1..n_leaves in their original order, interiors after them
    for i in range(1, n_leaves + 1):
        assert nodes[i].pType != abi.PRIM_BVH
    for i in range(n_leaves + 1, n):
        assert nodes[i].pType == abi.PRIM_BVH
    seen = np.zeros(n, dtype=int)
    stack = [0]
    while stack:
        i = stack.pop()
        seen[i] += 1
        nd = nodes[i]
        if nd.pType == abi.PRIM_BVH:
            l, r = nodes[nd.left], nodes[nd.right]
            assert l.parent == i and r.parent == i
            for ax in "xyz":                                                 # box = union of the children
                assert getattr(nd.bBOX.mini, ax) == min(getattr(l.bBOX.mini, ax), getattr(r.bBOX.mini, ax))
                assert getattr(nd.bBOX.maxi, ax) == max(getattr(l.bBOX.maxi, ax), getattr(r.bBOX.maxi, ax))
            assert nd.axis in (0, 1, 2)
            stack += [nd.left, nd.right]
    assert (seen == 1).all()                                                 # every node reachable exactly once
    assert scene.tree_depth() <= abi.TRC_MAX_BVH_DEPTH
    return n_leaves


def test_cornell_tree_shapes(cornell, cornell_spheres):
    assert check_invariants(cornell) == 9 and cornell.view.n_bvh == 17       # 2 cubes + 7 squares
    assert check_invariants(cornell_spheres) == 21 and cornell_spheres.view.n_bvh == 41
    leaves = cornell.bvh()[1:10]
    assert [l.pType for l in leaves] == [abi.PRIM_CUBE] * 2 + [abi.PRIM_SQUARE] * 7   # AAPLRenderer.mm:459-468
    assert [l.pIndex for l in leaves] == [0, 1, 0, 1, 2, 3, 4, 5, 6]


def test_mesh_tree_invariants_and_determinism(ball_mesh_scene):
    n_leaves = check_invariants(ball_mesh_scene)
    assert n_leaves == 9 + 5000
    again = host.HostScene(abi.SCENE_CORNELL_MESH, host.Mesh.ball(50, 50, 0.08))
    assert np.array_equal(again.bvh_array(), ball_mesh_scene.bvh_array())


def test_parallel_build_is_deterministic_on_a_large_mesh():
    # > 4096 primitives per subtree -> std::thread forks; post-order slots make the result schedule-independent
    mesh = host.Mesh.ball(160, 160, 0.05)            # 51 200 triangles
    a = host.HostScene(abi.SCENE_CORNELL_MESH, mesh)
    b = host.HostScene(abi.SCENE_CORNELL_MESH, mesh)
    assert np.array_equal(a.bvh_array(), b.bvh_array())
    check_invariants(a)
    assert a.tree_depth() < 40


@pytest.mark.parametrize("scene_name", ["cornell", "cornell_spheres", "ball_mesh_scene"])
def test_bvh_traversal_agrees_with_brute_force(request, scene_name):
    """Closest hit through the stackless traversal == linear scan over every leaf, exact (type, index, t)
    (ties between primitives at identical t are excluded: visiting order decides those)."""
    scene = request.getfixturevalue(scene_name)
    rays = np.concatenate([random_rays(40000, 5), random_rays(40000, 6, inside_only=True),
                           camera_rays(host.prepare_camera(192, 108), 192, 108)])
    h = po.trace_rays(scene.view, rays)
    b = po.trace_rays(scene.view, rays, brute=True)
    assert np.array_equal(h["hit"], b["hit"])
    hit = h["hit"] == 1
    same = (h["pType"] == b["pType"]) & (h["pIndex"] == b["pIndex"]) & (h["t"] == b["t"])
    ties = hit & ~same & (h["t"] == b["t"])
    assert (same | ~hit | ties).all()
    assert ties.sum() < 1e-3 * len(rays)


def test_two_leaf_ordering_and_degenerate_split():
    # two leaves: ordered by centroid on the max-extent axis (BVH.hh:60-77)
    a = host.build_node((10, 0, 0), (11, 1, 1), abi.PRIM_SPHERE, 0)
    b = host.build_node((0, 0, 0), (1, 1, 1), abi.PRIM_SPHERE, 1)
    nodes = host.build_tree([a, b])
    assert nodes[0].left == 2 and nodes[0].right == 1 and nodes[0].axis == 0
    # identical centroids: SAH cannot split -> median split of the (stable) centroid order (BVH.hh:187-195)
    same = [host.build_node((0, 0, 0), (1, 1, 1), abi.PRIM_SPHERE, i) for i in range(5)]
    nodes = host.build_tree(same)
    leaves_seen = sorted(n.pIndex for n in nodes if n.pType == abi.PRIM_SPHERE)
    assert leaves_seen == [0, 1, 2, 3, 4] and len(nodes) == 9


def test_build_node_transforms_the_eight_corners():
    # BVH.hh:273-314: AABB of the transformed corners (rotation by 90 deg about y swaps x and z extents)
    m = [[0, 0, 1, 5], [0, 1, 0, 0], [-1, 0, 0, 0], [0, 0, 0, 1]]
    n = host.build_node((0, 0, 0), (1, 2, 3), abi.PRIM_CUBE, 7, model=m)
    assert (n.bBOX.mini.x, n.bBOX.mini.y, n.bBOX.mini.z) == (5.0, 0.0, -1.0)
    assert (n.bBOX.maxi.x, n.bBOX.maxi.y, n.bBOX.maxi.z) == (8.0, 2.0, 0.0)
    assert n.pType == abi.PRIM_CUBE and n.pIndex == 7


@pytest.mark.skipif(not os.path.exists(REF_TEAPOT), reason="reference assets are not on this box")
def test_reference_teapot_obj_loads_and_traces():
    mesh = host.Mesh.load_obj(REF_TEAPOT)
    # fan triangulation: a face with k corners gives k-2 triangles (the file mixes triangles and quads)
    want = sum(len(line.split()) - 3 for line in open(REF_TEAPOT) if line.startswith("f "))
    assert mesh.n_triangles == want == 15704
    scene = host.HostScene(abi.SCENE_CORNELL_MESH, mesh)
    check_invariants(scene)
    v = scene.view
    pos = np.array([[v.triList[i].v[k] for k in range(3)] for i in range(0, v.n_vertex, 97)])
    assert 0 < pos[:, 1].min() and pos[:, 1].max() < 555      # placed on the floor inside the box
    rays = camera_rays(host.prepare_camera(256, 144), 256, 144)
    h = po.trace_rays(v, rays)
    b = po.trace_rays(v, rays, brute=True)
    assert (h["pType"] == abi.PRIM_TRIANGLE).sum() > 500
    assert np.array_equal(h["hit"], b["hit"]) and (np.abs(h["t"] - b["t"])[h["hit"] == 1] == 0).mean() > 0.999


# ---- the product's builder against the oracle's restatement of BVH::make (oracle/oracle_sah.cpp): no shared code ----

def _records(nodes, n):
    a = (C.c_uint32 * (16 * n)).from_address(C.addressof(nodes))
    return np.frombuffer(a, dtype=np.uint32).reshape(-1, 16).copy()


def _same_tree(scene):
    n = scene.n_leaves
    want = _records(po.sah_build(scene.leaves(), n), 2 * n - 1)
    got = scene.bvh_array()
    bad = np.nonzero((want != got).any(axis=1))[0]
    assert bad.size == 0, f"{bad.size} of {2 * n - 1} records differ, first at {bad[:5]}"


@pytest.mark.parametrize("scene_name", ["cornell", "cornell_spheres", "ball_mesh_scene"])
def test_host_tree_equals_the_restated_reference_build(request, scene_name):
    _same_tree(request.getfixturevalue(scene_name))


@pytest.mark.parametrize("mesh_name", ["teapot", "coatball"])
def test_host_tree_equals_the_restated_reference_build_on_the_reference_meshes(mesh_name):
    _same_tree(host.HostScene(abi.SCENE_CORNELL_MESH, host.Mesh.golden(mesh_name)))           # configs 3 and 4's assets


def test_host_tree_equals_the_restated_reference_build_on_config_4():
    # 1 005 056 triangles: the parallel build (forks above 4096 leaves) against the serial restatement
    _same_tree(host.HostScene(abi.SCENE_CORNELL_MESH, host.Mesh.golden("teapot").replicate(8, 80.0)))


def test_host_tree_equals_the_restated_reference_build_on_awkward_leaf_sets():
    rng = np.random.default_rng(11)
    cases = []
    cases.append([host.build_node((0, 0, 0), (1, 1, 1), abi.PRIM_SPHERE, i) for i in range(37)])        # one centroid
    cases.append([host.build_node((i % 2, 0, 0), (i % 2 + 1, 1, 1), abi.PRIM_SPHERE, i) for i in range(64)])   # two centroids
    cases.append([host.build_node((i, 0, 0), (i + 0.5, 1, 1), abi.PRIM_SQUARE, i) for i in range(3)])   # span 3
    pts = rng.uniform(-50, 50, (3000, 3)).astype(np.float32)
    pts[:, 1] = np.round(pts[:, 1])                                                                      # many ties on y
    pts[:, 2] = 7.0                                                                                      # flat in z
    ext = rng.uniform(0.0, 3.0, (3000, 3)).astype(np.float32)
    cases.append([host.build_node(tuple(p), tuple(p + e), abi.PRIM_TRIANGLE, i) for i, (p, e) in enumerate(zip(pts, ext))])
    clustered = np.concatenate([rng.normal(0, 0.01, (500, 3)), rng.normal(1000, 200, (20, 3))]).astype(np.float32)
    cases.append([host.build_node(tuple(p), tuple(p + np.float32(0.001)), abi.PRIM_SPHERE, i) for i, p in enumerate(clustered)])
    for leaves in cases:
        n = len(leaves)
        arr = (abi.BVH * n)(*leaves)
        want = _records(po.sah_build(arr, n), 2 * n - 1)
        got = _records(host.build_tree(leaves), 2 * n - 1)
        assert np.array_equal(want, got)


def test_host_leaf_record_equals_the_restated_buildnode():
    rng = np.random.default_rng(5)
    for i in range(200):
        lo = rng.uniform(-100, 100, 3).astype(np.float32)
        hi = lo + rng.uniform(0, 50, 3).astype(np.float32)
        m = rng.uniform(-2, 2, (4, 4)).astype(np.float32)
        m[3] = (0, 0, 0, 1)
        got = host.build_node(tuple(lo), tuple(hi), abi.PRIM_CUBE, i, model=m.tolist())
        box = abi.AABB()
        box.mini.x, box.mini.y, box.mini.z = lo
        box.maxi.x, box.maxi.y, box.maxi.z = hi
        mm = abi.float4x4()
        for c in range(4):
            mm.columns[c].x, mm.columns[c].y, mm.columns[c].z, mm.columns[c].w = m[0][c], m[1][c], m[2][c], m[3][c]
        want = po.sah_leaf(box, mm, abi.PRIM_CUBE, i)
        assert bytes(got) == bytes(want)
