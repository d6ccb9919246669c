"""LBVH build of the oracle: structure, conventions of BVH::buildTree's array layout, and
Scene::hit through the LBVH tree against brute force."""
import ctypes as C

import numpy as np
import pytest

from oracle import pyoracle
from tracer_amd import abi, host
from conftest import random_rays


def nodes_array(nodes):
    a = (C.c_uint32 * (16 * len(nodes))).from_address(C.addressof(nodes))
    return np.frombuffer(a, dtype=np.uint32).reshape(-1, 16)


def check_tree(nodes, n):
    a = nodes_array(nodes)
    boxes = a[:, 8:14].view(np.float32) if False else np.frombuffer(a.tobytes(), dtype=np.float32).reshape(-1, 16)
    parent, left, right, ptype = a[:, 0], a[:, 1], a[:, 2], a[:, 4].view(np.int32)
    assert len(a) == 2 * n - 1
    assert ptype[0] == abi.PRIM_BVH and parent[0] == 0
    assert (ptype[1:n + 1] != abi.PRIM_BVH).all() and (ptype[n + 1:] == abi.PRIM_BVH).all()
    interior = np.flatnonzero(ptype == abi.PRIM_BVH)
    kids = np.concatenate([left[interior], right[interior]])
    assert sorted(kids.tolist()) == list(range(1, 2 * n - 1))          # every non-root node is a child exactly once
    assert (parent[left[interior]] == interior).all() and (parent[right[interior]] == interior).all()
    mn, mx = boxes[:, 8:11], boxes[:, 12:15]                            # AABB = float3 mini (16 B) + float3 maxi
    for i in interior:
        l, r = left[i], right[i]
        assert (mn[i] == np.minimum(mn[l], mn[r])).all() and (mx[i] == np.maximum(mx[l], mx[r])).all()
    # depth of the deepest leaf
    depth = np.zeros(len(a), dtype=np.int64)
    order = [0]
    for i in order:
        for c in (left[i], right[i]):
            depth[c] = depth[i] + 1
            if ptype[c] == abi.PRIM_BVH:
                order.append(c)
    return int(depth[1:n + 1].max())


@pytest.mark.parametrize("kind", ["cornell", "spheres", "mesh"])
def test_lbvh_structure_and_hits(kind):
    if kind == "mesh":
        sc = host.HostScene(abi.SCENE_CORNELL_MESH, host.Mesh.ball(24, 24, 1.0))
    else:
        sc = host.HostScene(abi.SCENE_CORNELL if kind == "cornell" else abi.SCENE_CORNELL_SPHERES)
    n = sc.n_leaves
    nodes, height = pyoracle.lbvh_build(sc.leaves(), n)
    assert check_tree(nodes, n) == height
    # leaves keep their slot and payload
    src = nodes_array((abi.BVH * sc.view.n_bvh).from_address(C.addressof(sc.view.bvhList.contents)))
    dst = nodes_array(nodes)
    assert (src[1:n + 1, 4:] == dst[1:n + 1, 4:]).all()
    # Scene::hit through the LBVH tree finds the same closest hit as brute force
    view = sc.view_with_bvh(nodes)
    rays = random_rays(4000, 7)
    a = pyoracle.trace_rays(view, rays)
    b = pyoracle.trace_rays(sc.view, rays, brute=True)
    assert (a["hit"] == b["hit"]).all()
    hit = a["hit"] != 0
    assert (a["t"][hit] == b["t"][hit]).all()
    same_prim = (a["pType"][hit] == b["pType"][hit]) & (a["pIndex"][hit] == b["pIndex"][hit])
    assert same_prim.mean() > 0.999                                      # exact-t ties may resolve to a neighbour


def test_lbvh_degenerate_inputs():
    # all centroids identical (every Morton code equal: order by leaf index), and the minimum of two leaves
    for n in (2, 3, 17):
        leaves = (abi.BVH * n)()
        for k in range(n):
            leaves[k].pType = abi.PRIM_SPHERE; leaves[k].pIndex = k
            leaves[k].bBOX.mini.x = leaves[k].bBOX.mini.y = leaves[k].bBOX.mini.z = -1.0 - k
            leaves[k].bBOX.maxi.x = leaves[k].bBOX.maxi.y = leaves[k].bBOX.maxi.z = 1.0 + k
        nodes, height = pyoracle.lbvh_build(leaves, n)
        assert check_tree(nodes, n) == height
    # a flat axis (extent 0 in z) must not divide by zero
    n = 64
    leaves = (abi.BVH * n)()
    rs = np.random.RandomState(3)
    for k in range(n):
        x, y = rs.uniform(-10, 10, 2)
        leaves[k].pType = abi.PRIM_SPHERE; leaves[k].pIndex = k
        leaves[k].bBOX.mini.x, leaves[k].bBOX.mini.y, leaves[k].bBOX.mini.z = x - 1, y - 1, -1
        leaves[k].bBOX.maxi.x, leaves[k].bBOX.maxi.y, leaves[k].bBOX.maxi.z = x + 1, y + 1, 1
    nodes, height = pyoracle.lbvh_build(leaves, n)
    assert check_tree(nodes, n) == height and height < 16


def test_lbvh_random_leaf_sets():
    """property check over random leaf clouds (clustered, duplicated, huge / tiny extents): the tree is well formed,
    boxes nest, and every leaf is reachable exactly once"""
    rs = np.random.RandomState(11)
    for trial in range(25):
        n = int(rs.choice([2, 3, 5, 64, 257, 1000]))
        scale = float(rs.choice([1e-3, 1.0, 1e4]))
        centers = rs.normal(size=(n, 3)) * scale
        if trial % 3 == 0:
            centers[: n // 2] = centers[0]                      # many identical centroids
        if trial % 5 == 0:
            centers[:, 1] = 7.0                                 # flat axis
        half = np.abs(rs.normal(size=(n, 3))) * scale * 0.05
        leaves = (abi.BVH * n)()
        for k in range(n):
            leaves[k].pType = abi.PRIM_SPHERE; leaves[k].pIndex = k
            lo, hi = centers[k] - half[k], centers[k] + half[k]
            leaves[k].bBOX.mini.x, leaves[k].bBOX.mini.y, leaves[k].bBOX.mini.z = lo
            leaves[k].bBOX.maxi.x, leaves[k].bBOX.maxi.y, leaves[k].bBOX.maxi.z = hi
        nodes, height = pyoracle.lbvh_build(leaves, n)
        assert check_tree(nodes, n) == height
        assert height <= 64, (n, height)
