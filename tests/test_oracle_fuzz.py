"""CPU half of the randomised tests: on generated scenes the oracle's BVH traversal (Scene::hit, Render.hh:135-252,
through the host SAH tree and through the LBVH tree) finds the same closest hit as testing every leaf."""
import numpy as np
import pytest

from oracle import pyoracle as po
from tracer_amd import abi
from conftest import make_rays
from test_gpu_fuzz import random_scene
import ctypes as C


@pytest.mark.parametrize("seed", list(range(1, 9)))
def test_traversal_equals_brute_force_on_generated_scenes(seed):
    rs = np.random.RandomState(7000 + seed)
    sv, keep = random_scene(rs, n_spheres=int(rs.randint(3, 20)), n_cubes=int(rs.randint(1, 6)), n_tris=int(rs.randint(5, 300)))
    n = 3000
    rays = make_rays(rs.uniform(-90, 90, (n, 3)).astype(np.float32), rs.normal(size=(n, 3)).astype(np.float32))
    brute = po.trace_rays(sv, rays, brute=True)
    n_leaves = (sv.n_bvh + 1) // 2
    leaves = C.cast(C.addressof(sv.bvhList.contents) + C.sizeof(abi.BVH), C.POINTER(abi.BVH))
    lbvh, _ = po.lbvh_build(leaves, n_leaves)
    tv = abi.Scene.from_buffer_copy(sv); tv.bvhList = C.cast(lbvh, C.POINTER(abi.BVH)); tv.n_bvh = len(lbvh)
    for view in (sv, tv):
        got = po.trace_rays(view, rays)
        assert np.array_equal(got["hit"], brute["hit"])
        hit = got["hit"] != 0
        assert np.array_equal(got["t"][hit], brute["t"][hit])
        same = (got["pType"][hit] == brute["pType"][hit]) & (got["pIndex"][hit] == brute["pIndex"][hit])
        assert same.mean() > 0.995                          # exact-t ties may name another primitive
