"""On-device LBVH build through the C ABI against oracle/oracle_lbvh.cpp: every record of the
tree bit for bit, Scene::hit through the device-built tree, rendering, edge cases and the 1 M-triangle build."""
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


def build_both(gpu, sc):
    n = sc.n_leaves
    want, height = pyoracle.lbvh_build(sc.leaves(), n)
    gpu.upload_scene_lbvh(sc.leaves_view())
    got = gpu.download_bvh()
    n_nodes, h, ms = gpu.lbvh_info()
    assert n_nodes == 2 * n - 1 and h == height and ms > 0
    assert (raw(got) == raw(want)).all()
    return want


@pytest.mark.parametrize("kind", ["cornell", "spheres", "mesh"])
def test_device_tree_equals_oracle_tree_and_hits(gpu, kind):
    if kind == "mesh":
        sc = host.HostScene(abi.SCENE_CORNELL_MESH, host.Mesh.ball(60, 60, 1.0))
    else:
        sc = host.HostScene(abi.SCENE_CORNELL if kind == "cornell" else abi.SCENE_CORNELL_SPHERES)
    want = build_both(gpu, sc)
    rays = random_rays(20000, 11)
    got = gpu.trace_rays(rays)
    ref = pyoracle.trace_rays(sc.view_with_bvh(want), rays)
    for f in ref.dtype.names:
        assert (got[f].view(np.uint32) == ref[f].view(np.uint32)).all(), f
    any_got = gpu.trace_rays(rays, any_hit=True)
    any_ref = pyoracle.trace_rays(sc.view_with_bvh(want), rays, any_hit=True)
    assert (any_got["hit"] == any_ref["hit"]).all()


def test_render_through_device_tree(gpu):
    sc = host.HostScene(abi.SCENE_CORNELL_MESH, host.Mesh.ball(40, 40, 1.0))
    want = build_both(gpu, sc)
    W, H = 96, 64
    cam = host.prepare_camera(W, H)
    gpu.set_camera(cam); gpu.resize(W, H)
    rng = host.fill_rng(99, W, H)
    for integ in (abi.INTEGRATOR_PATH, abi.INTEGRATOR_MIS):
        gpu.upload_rng(rng); gpu.clear_accum()
        gpu.render(spp=4, integrator=integ)
        got = gpu.download_accum()
        r = rng.copy()
        ref, _ = pyoracle.render(sc.view_with_bvh(want), cam, W, H, r, spp=4, integrator=integ)
        assert (got.view(np.uint32) == ref.view(np.uint32)).all()
        assert (gpu.download_rng() == r).all()


def test_degenerate_and_tiny_inputs(gpu):
    sc = host.HostScene(abi.SCENE_CORNELL_SPHERES)
    base = sc.leaves_view()
    for n in (2, 3, 17, 300):
        leaves = (abi.BVH * n)()
        for k in range(n):
            leaves[k].pType = abi.PRIM_SPHERE; leaves[k].pIndex = k % sc.view.n_sphere
            leaves[k].bBOX.mini.x = leaves[k].bBOX.mini.y = leaves[k].bBOX.mini.z = -1.0 - k
            leaves[k].bBOX.maxi.x = leaves[k].bBOX.maxi.y = leaves[k].bBOX.maxi.z = 1.0 + k
        v = abi.Scene.from_buffer_copy(base); v.bvhList = C.cast(leaves, C.POINTER(abi.BVH)); v.n_bvh = n
        want, height = pyoracle.lbvh_build(leaves, n)
        gpu.upload_scene_lbvh(v)
        assert (raw(gpu.download_bvh()) == raw(want)).all() and gpu.lbvh_info()[1] == height
    # interior records and single leaves are rejected
    from tracer_amd.device import TracerError
    with pytest.raises(TracerError):
        gpu.upload_scene_lbvh(sc.view)                 # bvhList[0] is the root, an interior record
    v = abi.Scene.from_buffer_copy(base); v.n_bvh = 1
    with pytest.raises(TracerError):
        gpu.upload_scene_lbvh(v)
    gpu.upload_scene(sc.view)                          # a host tree replaces the device tree
    with pytest.raises(TracerError):
        gpu.download_bvh()


def test_million_triangle_build(gpu):
    mesh = host.Mesh.ball(60, 60, 0.08).replicate(12, 2.4)
    sc = host.HostScene(abi.SCENE_CORNELL_MESH, mesh)
    want = build_both(gpu, sc)                         # 2.07 M records, bit for bit
    n_nodes, height, ms = gpu.lbvh_info()
    print(f"LBVH {sc.n_leaves} leaves: {ms:.2f} ms on the GPU, height {height} (host SAH tree: depth {sc.tree_depth()})")
    assert ms < 50.0
    W, H = 1920, 1080
    rays = camera_rays(host.prepare_camera(W, H), W, H, step=8)
    got = gpu.trace_rays(rays)
    ref = pyoracle.trace_rays(sc.view_with_bvh(want), rays)
    for f in ref.dtype.names:
        assert (got[f].view(np.uint32) == ref[f].view(np.uint32)).all(), f
    # same closest hits as the host SAH tree (t exact; an exact tie may name a neighbouring triangle)
    sah = pyoracle.trace_rays(sc.view, rays)
    assert (got["hit"] == sah["hit"]).all()
    hit = got["hit"] != 0
    assert (got["t"][hit] == sah["t"][hit]).all()


@pytest.mark.parametrize("seed", range(*[int(x) for x in os.environ.get("TRC_FUZZ_LBVH_SEEDS", "0:6").split(":")]))
def test_signed_zeros_in_leaf_boxes(gpu, seed):
    """-0 and +0 are equal bounds with different bits: the merged boxes keep the oracle's (std::min / std::max: the first operand
    on a tie), not whatever a hardware min would pick."""
    rng = np.random.default_rng(100 + seed)
    n = int(rng.choice([3, 64, 300, 2500]))
    c = rng.choice([-0.0, 0.0, 1e-30, -1e-30, 1.0, -1.0], (n, 3)); r = rng.choice([0.0, 0.5], (n, 3))
    lo, hi = (c - r).astype(np.float32), (c + r).astype(np.float32)
    sc = host.HostScene(abi.SCENE_CORNELL_SPHERES)
    leaves = (abi.BVH * n)()
    for k in range(n):
        leaves[k].pType = abi.PRIM_SPHERE; leaves[k].pIndex = k % sc.view.n_sphere
        leaves[k].bBOX.mini.x, leaves[k].bBOX.mini.y, leaves[k].bBOX.mini.z = (float(v) for v in lo[k])
        leaves[k].bBOX.maxi.x, leaves[k].bBOX.maxi.y, leaves[k].bBOX.maxi.z = (float(v) for v in hi[k])
    v = abi.Scene.from_buffer_copy(sc.leaves_view()); v.bvhList = C.cast(leaves, C.POINTER(abi.BVH)); v.n_bvh = n
    want, height = pyoracle.lbvh_build(leaves, n)
    gpu.upload_scene_lbvh(v)
    assert (raw(gpu.download_bvh()) == raw(want)).all() and gpu.lbvh_info()[1] == height
