"""BVH::buildTree on the device (trc_upload_scene_sah, tracer_amd/csrc/trc_sah_build.hpp) against the two CPU statements of
the reference's SAH build (RT_Metal/Metal/BVH.hh:35-269): oracle/oracle_sah.cpp (serial recursion on the reference's growing
list) and the host builder of libtrc_host (parallel, slots assigned up front).  All 2n-1 records, bit for bit."""
import os
import ctypes as C

import numpy as np
import pytest

from oracle import pyoracle
from tracer_amd import abi, host
from conftest import random_rays, camera_rays

pytestmark = pytest.mark.gpu


def raw(nodes):
    return np.frombuffer(bytes(memoryview(nodes)), dtype=np.uint32).reshape(-1, 16)


def first_difference(got, want):
    bad = np.nonzero((got != want).any(axis=1))[0]
    return f"{len(bad)} of {len(want)} records differ, first at {bad[0]}: got {got[bad[0]]}, want {want[bad[0]]}" if len(bad) else ""


def build_and_compare(gpu, view, leaves, n):
    want = pyoracle.sah_build(leaves, n)
    gpu.upload_scene_sah(view)
    got = gpu.download_bvh()
    n_nodes, height, ms = gpu.lbvh_info()
    assert n_nodes == 2 * n - 1 and ms > 0
    assert not first_difference(raw(got), raw(want))
    return want, height, ms


@pytest.mark.parametrize("kind", ["cornell", "spheres", "mesh", "mesh_large"])
def test_device_sah_tree_is_the_reference_s_tree(gpu, kind):
    if kind == "mesh":
        sc = host.HostScene(abi.SCENE_CORNELL_MESH, host.Mesh.ball(60, 60, 1.0))            # 7 209 leaves: a few level rounds
    elif kind == "mesh_large":
        sc = host.HostScene(abi.SCENE_CORNELL_MESH, host.Mesh.ball(160, 160, 0.05))         # 51 209 leaves: many tasks per node
    else:
        sc = host.HostScene(abi.SCENE_CORNELL if kind == "cornell" else abi.SCENE_CORNELL_SPHERES)
    want, height, _ = build_and_compare(gpu, sc.leaves_view(), sc.leaves(), sc.n_leaves)
    assert not first_difference(raw(want), sc.bvh_array())          # ... which is also what the host builder made
    assert height == sc.tree_depth()
    # Scene::hit through the device-built tree == the oracle through the host-built tree, record for record
    rays = random_rays(20000, 12)
    got = gpu.trace_rays(rays)
    ref = pyoracle.trace_rays(sc.view, rays)
    for f in ref.dtype.names:
        assert (got[f].view(np.uint32) == ref[f].view(np.uint32)).all(), f


def leaf_records(boxes, n_sphere):
    n = len(boxes)
    leaves = (abi.BVH * n)()
    for k, (lo, hi) in enumerate(boxes):
        leaves[k].pType = abi.PRIM_SPHERE; leaves[k].pIndex = k % n_sphere
        leaves[k].bBOX.mini.x, leaves[k].bBOX.mini.y, leaves[k].bBOX.mini.z = (float(v) for v in lo)
        leaves[k].bBOX.maxi.x, leaves[k].bBOX.maxi.y, leaves[k].bBOX.maxi.z = (float(v) for v in hi)
    return leaves


def cases():
    rng = np.random.default_rng(7)
    out = {}
    for n in (2, 3, 5, 17, 64, 65, 66, 129, 300, 5000):
        out[f"identical_{n}"] = [((0, 0, 0), (1, 1, 1))] * n                                 # every split is the median split
        out[f"nested_{n}"] = [((-1.0 - k,) * 3, (1.0 + k,) * 3) for k in range(n)]           # identical centroids, growing boxes
    for n in (2, 3, 4, 9, 63, 64, 65, 257, 4097, 30000):
        c = rng.uniform(-50, 50, (n, 3)).astype(np.float32); r = rng.uniform(0.01, 3.0, (n, 3)).astype(np.float32)
        out[f"random_{n}"] = list(zip(c - r, c + r))
    out["collinear_3000"] = [((k * 0.5, 0, 0), (k * 0.5 + 1, 1, 1)) for k in range(3000)]
    out["two_clusters_9000"] = [((0, 0, 0), (1, 1, 1))] * 4500 + [((100, 0, 0), (101, 1, 1))] * 4500   # one real split, then medians
    out["geometric_400"] = [((2.0 ** (k / 8), 0, 0), (2.0 ** (k / 8) + 0.001, 1, 1)) for k in range(400)]   # one-leaf splits near the top
    grid = [((x, y, 0), (x + 0.5, y + 0.5, 0.5)) for y in range(120) for x in range(120)]      # many ties in bucket boundaries
    out["grid_14400"] = grid
    pairs = rng.integers(0, 4, (6000, 3)).astype(np.float32)                                    # many identical centroids
    out["lattice_6000"] = [(p, p + 1) for p in pairs]
    return out


CASES = cases()


@pytest.mark.parametrize("name", sorted(CASES))
def test_degenerate_and_ragged_leaf_sets(gpu, name):
    sc = host.HostScene(abi.SCENE_CORNELL_SPHERES)
    boxes = CASES[name]
    n = len(boxes)
    leaves = leaf_records(boxes, sc.view.n_sphere)
    v = abi.Scene.from_buffer_copy(sc.leaves_view()); v.bvhList = C.cast(leaves, C.POINTER(abi.BVH)); v.n_bvh = n
    want = pyoracle.sah_build(leaves, n)
    depth = tree_depth(want)
    if depth > abi.TRC_MAX_BVH_DEPTH:
        from tracer_amd.device import TracerError
        with pytest.raises(TracerError):
            gpu.upload_scene_sah(v)
        return
    gpu.upload_scene_sah(v)
    assert not first_difference(raw(gpu.download_bvh()), raw(want))
    assert gpu.lbvh_info()[1] == depth


def fuzz_boxes(seed):
    """a leaf set mixing what real scenes mix: clusters of duplicates, lattices, points, planar sheets, signed zeros, scales"""
    rng = np.random.default_rng(seed)
    n = int(rng.choice([2, 3, 7, 31, 64, 65, 100, 257, 1000, 2049, 5000, 12000]))
    scale = float(10.0 ** rng.integers(-12, 13))
    kind = seed % 6
    if kind == 0:      # clusters of exact duplicates
        centres = rng.uniform(-1, 1, (max(1, n // 50), 3))
        c = centres[rng.integers(0, len(centres), n)]
        r = np.full((n, 3), 0.01)
    elif kind == 1:    # integer lattice: many equal centroids per axis, bucket boundaries hit exactly
        c = rng.integers(-8, 9, (n, 3)).astype(np.float64); r = rng.integers(0, 3, (n, 3)) * 0.5
    elif kind == 2:    # a planar sheet (one axis degenerate) with points (zero-size boxes)
        c = rng.uniform(-1, 1, (n, 3)); c[:, rng.integers(0, 3)] = 0.25; r = np.zeros((n, 3))
    elif kind == 3:    # signed zeros and tiny values around the origin
        c = rng.choice([-0.0, 0.0, 1e-30, -1e-30, 1.0, -1.0], (n, 3)); r = rng.choice([0.0, 0.5], (n, 3))
    elif kind == 4:    # long thin boxes, wildly different sizes
        c = rng.normal(0, 1, (n, 3)); r = 10.0 ** rng.uniform(-6, 1, (n, 3))
    else:              # uniform
        c = rng.uniform(-1, 1, (n, 3)); r = rng.uniform(0, 0.2, (n, 3))
    lo = ((c - r) * scale).astype(np.float32); hi = ((c + r) * scale).astype(np.float32)
    return list(zip(lo, hi))


# TRC_FUZZ_SAH_SEEDS="a:b": a longer campaign by hand (profiles/r05/fuzz_campaign.txt)
@pytest.mark.parametrize("seed", range(*[int(x) for x in os.environ.get("TRC_FUZZ_SAH_SEEDS", "0:48").split(":")]))
def test_fuzzed_leaf_sets(gpu, seed):
    sc = host.HostScene(abi.SCENE_CORNELL_SPHERES)
    boxes = fuzz_boxes(seed)
    n = len(boxes)
    leaves = leaf_records(boxes, sc.view.n_sphere)
    v = abi.Scene.from_buffer_copy(sc.leaves_view()); v.bvhList = C.cast(leaves, C.POINTER(abi.BVH)); v.n_bvh = n
    want = pyoracle.sah_build(leaves, n)
    if tree_depth(want) > abi.TRC_MAX_BVH_DEPTH:
        pytest.skip("deeper than TRC_MAX_BVH_DEPTH")
    gpu.upload_scene_sah(v)
    assert not first_difference(raw(gpu.download_bvh()), raw(want)), (seed, n)
    # ... and the host builder agrees with both
    built = host.build_tree([leaves[k] for k in range(n)])
    assert not first_difference(raw((abi.BVH * len(built))(*built)), raw(want))


def tree_depth(nodes):
    a = raw(nodes)
    depth = np.zeros(len(a), dtype=np.int64)
    order = [0]
    for i in order:
        if a[i, 4] == np.uint32(abi.PRIM_BVH & 0xFFFFFFFF):
            for c in (a[i, 1], a[i, 2]):
                depth[c] = depth[i] + 1
                order.append(int(c))
    return int(depth.max())


def test_rejected_inputs(gpu):
    from tracer_amd.device import TracerError
    sc = host.HostScene(abi.SCENE_CORNELL_SPHERES)
    with pytest.raises(TracerError):
        gpu.upload_scene_sah(sc.view)                               # bvhList[0] is the root, an interior record
    for bad in (float("nan"), float("inf"), 3e37):
        boxes = [((k, 0, 0), (k + 1, 1, 1)) for k in range(100)]
        boxes[37] = ((0, 0, 0), (bad, 1, 1))
        leaves = leaf_records(boxes, sc.view.n_sphere)
        v = abi.Scene.from_buffer_copy(sc.leaves_view()); v.bvhList = C.cast(leaves, C.POINTER(abi.BVH)); v.n_bvh = 100
        with pytest.raises(TracerError):
            gpu.upload_scene_sah(v)
    gpu.upload_scene_sah(sc.leaves_view())                          # the context is still usable
    assert not first_difference(raw(gpu.download_bvh()), sc.bvh_array())


def test_render_through_device_sah_tree(gpu):
    sc = host.HostScene(abi.SCENE_CORNELL_MESH, host.Mesh.ball(40, 40, 1.0))
    W, H = 96, 64
    cam = host.prepare_camera(W, H)
    gpu.set_camera(cam); gpu.resize(W, H)
    rng = host.fill_rng(99, W, H)
    frames = []
    for upload in (lambda: gpu.upload_scene(sc.view), lambda: gpu.upload_scene_sah(sc.leaves_view())):
        upload()
        gpu.upload_rng(rng); gpu.clear_accum()
        gpu.render(spp=4, integrator=abi.INTEGRATOR_MIS)
        frames.append(gpu.download_accum().copy())
    assert (frames[0].view(np.uint32) == frames[1].view(np.uint32)).all()
    r = rng.copy()
    ref, _ = pyoracle.render(sc.view, cam, W, H, r, spp=4, integrator=abi.INTEGRATOR_MIS)
    assert (frames[1].view(np.uint32) == ref.view(np.uint32)).all()


def test_million_triangle_sah_build(gpu):
    mesh = host.Mesh.golden("teapot").replicate(8, 80.0)           # BASELINE config 4's scene: 1 005 056 triangles
    sc = host.HostScene(abi.SCENE_CORNELL_MESH, mesh)
    gpu.upload_scene_sah(sc.leaves_view())
    n_nodes, height, ms = gpu.lbvh_info()
    print(f"device SAH build of {sc.n_leaves} leaves: {ms:.2f} ms on the GPU, depth {height}")
    assert not first_difference(raw(gpu.download_bvh()), sc.bvh_array())     # the host builder's 2 010 111 records
    assert height == sc.tree_depth() and ms < 100.0
    W, H = 1920, 1080
    rays = camera_rays(host.prepare_camera(W, H), W, H, step=8)
    got = gpu.trace_rays(rays)
    ref = pyoracle.trace_rays(sc.view, rays)
    for f in ref.dtype.names:
        assert (got[f].view(np.uint32) == ref[f].view(np.uint32)).all(), f


def analytic_leaves_view(sc):
    """`sc`'s scene with bvhList = the leaf records of the analytic primitives only (they come first, scene.cpp)."""
    n_tri = sc.view.n_index // 3
    v = abi.Scene.from_buffer_copy(sc.leaves_view())
    v.n_bvh = sc.n_leaves - n_tri
    return v


@pytest.mark.parametrize("mesh", ["ball", "teapot", "teapots_1m"])
def test_triangle_leaves_written_on_the_device(gpu, mesh):
    """trc_upload_scene_device(TRC_TREE_TRIANGLE_LEAVES): the host hands over its handful of analytic leaves and the mesh; the
    triangles' leaf records (AAPLRenderer.mm:575-589 + BVH::buildNode) are written on the GPU.  Whole tree == the host's."""
    m = {"ball": lambda: host.Mesh.ball(60, 60, 1.0), "teapot": lambda: host.Mesh.golden("teapot"),
         "teapots_1m": lambda: host.Mesh.golden("teapot").replicate(8, 80.0)}[mesh]()
    sc = host.HostScene(abi.SCENE_CORNELL_MESH, m)
    gpu.upload_scene_device(analytic_leaves_view(sc), abi.TREE_SAH | abi.TREE_TRIANGLE_LEAVES)
    assert not first_difference(raw(gpu.download_bvh()), sc.bvh_array())
    if mesh != "teapots_1m":
        want, height = pyoracle.lbvh_build(sc.leaves(), sc.n_leaves)
        gpu.upload_scene_device(analytic_leaves_view(sc), abi.TREE_TRIANGLE_LEAVES)
        assert not first_difference(raw(gpu.download_bvh()), raw(want))
    # without the flag the same view is a tree over the analytic primitives alone; unknown flags are refused
    from tracer_amd.device import TracerError
    with pytest.raises(TracerError):
        gpu.upload_scene_device(sc.leaves_view(), 8)
    rays = random_rays(5000, 3)
    gpu.upload_scene_device(analytic_leaves_view(sc), abi.TREE_SAH | abi.TREE_TRIANGLE_LEAVES)
    got = gpu.trace_rays(rays)
    ref = pyoracle.trace_rays(sc.view, rays)
    for f in ref.dtype.names:
        assert (got[f].view(np.uint32) == ref[f].view(np.uint32)).all(), f


def test_host_without_leaf_loop_and_tree(gpu):
    """trc_host_scene_create_leaves(analytic_leaves_only): the host neither writes a leaf per triangle nor builds a tree; the
    device does both and ends with the tree of the host that did."""
    m = host.Mesh.golden("teapot").replicate(2, 80.0)
    full = host.HostScene(abi.SCENE_CORNELL_MESH, m)
    lean = host.HostScene(abi.SCENE_CORNELL_MESH, m, analytic_leaves_only=True)
    assert lean.view.n_bvh == full.n_leaves - full.view.n_index // 3 and lean.view.n_index == full.view.n_index
    gpu.upload_scene_device(lean.view, abi.TREE_SAH | abi.TREE_TRIANGLE_LEAVES)
    assert not first_difference(raw(gpu.download_bvh()), full.bvh_array())
