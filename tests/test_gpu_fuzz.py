"""Randomised differential test: generated scenes (random spheres / axis-aligned squares / rotated cubes / a small
triangle soup, random materials and textures, random camera with a real aperture) through the GPU path and the
oracle -- Scene::hit batches with counters, and small frames of all three integrators, bit for bit."""
import ctypes as C
import math
import os

import numpy as np
import pytest

from oracle import pyoracle as po
from tracer_amd import abi, host
from conftest import make_rays

pytestmark = pytest.mark.gpu
F32 = np.float32


def mat4(cols):
    m = abi.float4x4()
    for c in range(4):
        for r in range(4):
            setattr(m.columns[c], "xyzw"[r], float(cols[r][c]))
    return m


def rot_y(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, 0, s, 0], [0, 1, 0, 0], [-s, 0, c, 0], [0, 0, 0, 1]], np.float64)


def rot_x(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[1, 0, 0, 0], [0, c, -s, 0], [0, s, c, 0], [0, 0, 0, 1]], np.float64)


def random_scene(rs, n_spheres, n_cubes, n_tris):
    keep = []
    mats = []
    types = [abi.MAT_LAMBERT, abi.MAT_METAL, abi.MAT_PLASTIC, abi.MAT_GLASS, abi.MAT_LAMBERT, abi.MAT_DIFFUSE, abi.MAT_NIL]
    for i in range(24):
        m = abi.Material()
        m.type = types[rs.randint(len(types))] if i not in (19,) else abi.MAT_GLASS
        m.medium = int(rs.randint(3)) if m.type in (abi.MAT_GLASS, abi.MAT_NIL) else abi.MEDIUM_NIL
        m.textureInfo.type = abi.TEX_CHECKER if rs.rand() < 0.3 else abi.TEX_CONSTANT
        a = rs.uniform(0.2, 1.0, 3) if m.type != abi.MAT_DIFFUSE else rs.uniform(2.0, 9.0, 3)
        m.textureInfo.albedo.x, m.textureInfo.albedo.y, m.textureInfo.albedo.z = a
        mats.append(m)
    mats[5].type = abi.MAT_DIFFUSE                      # something emits
    leaves, spheres, squares, cubes = [], [], [], []
    for i in range(n_spheres):
        c, r = rs.uniform(-40, 40, 3), rs.uniform(2, 12)
        s = abi.Sphere(); s.radius = r; s.center.x, s.center.y, s.center.z = c; s.material = int(rs.randint(len(mats)))
        spheres.append(s)
        leaves.append(host.build_node(c - r, c + r, abi.PRIM_SPHERE, i))
    for i in range(8):                                  # >= 7 squares: traceMIS / traceVolume sample squareList[5], [6]
        ak = int(rs.randint(3)); ai, aj = [(1, 2), (0, 2), (0, 1)][ak]
        lo_i, lo_j = rs.uniform(-60, 20, 2); ext = rs.uniform(15, 70, 2)
        q = abi.Square(); q.axis_i, q.axis_j, q.axis_k = ai, aj, ak
        q.range_i.x, q.range_i.y = lo_i, lo_i + ext[0]; q.range_j.x, q.range_j.y = lo_j, lo_j + ext[1]
        q.value_k = rs.uniform(-60, 60); q.material = 5 if i in (5, 6) else int(rs.randint(len(mats)))
        squares.append(q)
        lo, hi = [0.0] * 3, [0.0] * 3
        lo[ai], hi[ai] = q.range_i.x, q.range_i.y; lo[aj], hi[aj] = q.range_j.x, q.range_j.y
        lo[ak], hi[ak] = q.value_k - 1 / 512, q.value_k + 1 / 512
        leaves.append(host.build_node(lo, hi, abi.PRIM_SQUARE, i))
    for i in range(n_cubes):
        T = np.eye(4); T[:3, 3] = rs.uniform(-40, 40, 3)
        S = np.diag(list(rs.uniform(5, 25, 3)) + [1.0])
        M = (T @ rot_y(rs.uniform(0, 6.28)) @ rot_x(rs.uniform(-0.5, 0.5)) @ S).astype(F32).astype(np.float64)
        Mi = np.linalg.inv(M).astype(F32).astype(np.float64)
        cb = abi.Cube()
        cb.model_matrix, cb.inverse_matrix, cb.normal_matrix = mat4(M), mat4(Mi), mat4(Mi.T)
        cb.box.mini.x = cb.box.mini.y = cb.box.mini.z = 0.0
        cb.box.maxi.x = cb.box.maxi.y = cb.box.maxi.z = 1.0
        cb.material = int(rs.randint(len(mats)))
        cubes.append(cb)
        leaves.append(host.build_node((0, 0, 0), (1, 1, 1), abi.PRIM_CUBE, i, model=M))
    verts = (abi.TriangleVertex * (3 * n_tris))()
    idx = (C.c_uint32 * (3 * n_tris))(*range(3 * n_tris))
    for t in range(n_tris):
        base = rs.uniform(-45, 45, 3)
        p = [base + rs.uniform(-9, 9, 3) for _ in range(3)]
        if t % 17 == 0: p[2] = p[1].copy()              # degenerate (zero-area) triangles now and then
        if t % 23 == 0: p[1] = p[0] + (p[2] - p[0]) * 0.5   # collinear
        n = np.cross(p[1] - p[0], p[2] - p[0]); n = n / (np.linalg.norm(n) + 1e-9)
        for k in range(3):
            v = verts[3 * t + k]
            v.v[:] = [float(x) for x in p[k]]; v.n[:] = [float(x) for x in (n + rs.normal(0, 0.1, 3))]; v.uv[:] = [float(x) for x in rs.rand(2)]
        pf = np.array([[v_ for v_ in verts[3 * t + k].v] for k in range(3)])
        leaves.append(host.build_node(pf.min(0), pf.max(0), abi.PRIM_TRIANGLE, t))
    nodes = host.build_tree(leaves)
    sv = abi.Scene()
    sv.bvhList, sv.n_bvh = C.cast(nodes, C.POINTER(abi.BVH)), len(nodes)
    for name, items, T_ in (("sphere", spheres, abi.Sphere), ("square", squares, abi.Square), ("cube", cubes, abi.Cube)):
        arr = (T_ * max(1, len(items)))(*items); keep.append(arr)
        setattr(sv, name + "List", C.cast(arr, C.POINTER(T_))); setattr(sv, "n_" + name, len(items))
    marr = (abi.Material * len(mats))(*mats); keep.append(marr)
    sv.materials, sv.n_material = C.cast(marr, C.POINTER(abi.Material)), len(mats)
    sv.triList, sv.n_vertex = C.cast(verts, C.POINTER(abi.TriangleVertex)), 3 * n_tris
    sv.idxList, sv.n_index = C.cast(idx, C.POINTER(C.c_uint32)), 3 * n_tris
    keep += [nodes, verts, idx]
    return sv, keep


# TRC_FUZZ_SEEDS="a:b" / TRC_FUZZ_SPP=n: a longer or differently shaped run by hand (spp >= 8 takes the fused kernels:
# persistent workgroups on the big trees, automatic 4x4 blocks on the small ones; the default 5 takes the strip kernels)
_SEEDS = os.environ.get("TRC_FUZZ_SEEDS", "1:31").split(":")
_SPP = int(os.environ.get("TRC_FUZZ_SPP", "5"))


@pytest.mark.parametrize("seed", list(range(int(_SEEDS[0]), int(_SEEDS[1]))))
def test_generated_scene(gpu, seed):
    rs = np.random.RandomState(1000 + seed)
    big = seed % 3 == 0                                   # every third scene: tree too large for LDS
    sv, keep = random_scene(rs, n_spheres=int(rs.randint(3, 20)), n_cubes=int(rs.randint(1, 6)),
                            n_tris=int(rs.randint(900, 1500)) if big else int(rs.randint(5, 60)))
    gpu.upload_scene(sv)
    # Scene::hit on random rays, closest and any-hit, with counters
    n = 6000
    o = rs.uniform(-90, 90, (n, 3)).astype(F32)
    d = rs.normal(size=(n, 3)).astype(F32)
    rays = make_rays(o, d)
    for any_hit in (False, True):
        got, ref = gpu.trace_rays(rays, any_hit=any_hit), po.trace_rays(sv, rays, any_hit=any_hit)
        names = ref.dtype.names if not any_hit else ("hit", "n_descend", "n_return", "n_leaf")
        for f in names:
            assert np.array_equal(got[f].view(np.uint32), ref[f].view(np.uint32)), (seed, any_hit, f)
    # frames: thin-lens camera somewhere outside, all integrators, with a density grid for the GridDensity media
    W, H = 72, 56
    look_from = rs.uniform(-150, 150, 3); look_from[2] = -170.0
    cam = host.make_camera(tuple(look_from), tuple(rs.uniform(-10, 10, 3)), (0, 1, 0), float(rs.uniform(0.0, 3.0)), W / H,
                           math.radians(55), 170.0)
    grid = rs.rand(6, 7, 8).astype(F32) * (rs.rand(6, 7, 8) > 0.4)
    info = host.density_info(np.ascontiguousarray(grid), sigma_a=0.02, sigma_s=0.05, g=0.3)
    gpu.set_camera(cam); gpu.set_environment((0.3, 0.4, 0.6)); gpu.resize(W, H)
    gpu.upload_density(info, np.ascontiguousarray(grid)); po.set_density(info, np.ascontiguousarray(grid))
    try:
        for integ in (abi.INTEGRATOR_PATH, abi.INTEGRATOR_MIS, abi.INTEGRATOR_VOLUME):
            for launch in range(2):
                rng = host.fill_rng(50 + seed, W, H)
                gpu.upload_rng(rng); gpu.clear_accum(); gpu.reset_stats(); gpu.render(spp=_SPP, integrator=integ, max_depth=6)
            got, got_rng, st = gpu.download_accum(), gpu.download_rng(), gpu.stats()
            ref, rst = po.render(sv, cam, W, H, rng, spp=_SPP, integrator=integ, max_depth=6, env=(0.3, 0.4, 0.6))
            assert st.rays == rst.rays, (seed, integ)
            assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (seed, integ)
            assert np.array_equal(got_rng, rng)
        # the same scene through a tree built on the GPU from the leaf records
        n_leaves = (sv.n_bvh + 1) // 2
        lv = abi.Scene.from_buffer_copy(sv)
        lv.bvhList = C.cast(C.addressof(sv.bvhList.contents) + C.sizeof(abi.BVH), C.POINTER(abi.BVH)); lv.n_bvh = n_leaves
        want, height = po.lbvh_build(lv.bvhList, n_leaves)
        gpu.upload_scene_lbvh(lv)
        tree = gpu.download_bvh()
        assert bytes(memoryview(tree)) == bytes(memoryview(want)) and gpu.lbvh_info()[1] == height
        tv = abi.Scene.from_buffer_copy(sv); tv.bvhList = C.cast(tree, C.POINTER(abi.BVH)); tv.n_bvh = len(tree)
        rng = host.fill_rng(90 + seed, W, H)
        gpu.upload_rng(rng); gpu.clear_accum(); gpu.render(spp=3, integrator=abi.INTEGRATOR_MIS, max_depth=6)
        ref, _ = po.render(tv, cam, W, H, rng, spp=3, integrator=abi.INTEGRATOR_MIS, max_depth=6, env=(0.3, 0.4, 0.6))
        assert np.array_equal(gpu.download_accum().view(np.uint32), ref.view(np.uint32)), (seed, "lbvh")
        # ... and through the reference's own SAH tree built on the GPU: the oracle's restatement of BVH::buildTree record for
        # record, the frame the oracle renders through it
        want = po.sah_build(lv.bvhList, n_leaves)
        if max_leaf_depth(want) <= abi.TRC_MAX_BVH_DEPTH:
            gpu.upload_scene_sah(lv)
            tree = gpu.download_bvh()
            assert bytes(memoryview(tree)) == bytes(memoryview(want)), (seed, "device sah tree")
            tv = abi.Scene.from_buffer_copy(sv); tv.bvhList = C.cast(tree, C.POINTER(abi.BVH)); tv.n_bvh = len(tree)
            rng = host.fill_rng(95 + seed, W, H)
            gpu.upload_rng(rng); gpu.clear_accum(); gpu.render(spp=3, integrator=abi.INTEGRATOR_PATH, max_depth=6)
            ref, _ = po.render(tv, cam, W, H, rng, spp=3, integrator=abi.INTEGRATOR_PATH, max_depth=6, env=(0.3, 0.4, 0.6))
            assert np.array_equal(gpu.download_accum().view(np.uint32), ref.view(np.uint32)), (seed, "device sah")
    finally:
        po.set_density(None, None); gpu.upload_density(None, None); gpu.set_environment((0.0, 0.0, 0.0))


def max_leaf_depth(nodes):
    a = np.frombuffer(bytes(memoryview(nodes)), dtype=np.uint32).reshape(-1, 16)
    depth = np.zeros(len(a), dtype=np.int64)
    order = [0]
    for i in order:
        if a[i, 4] == abi.PRIM_BVH:
            for c in (a[i, 1], a[i, 2]):
                depth[c] = depth[i] + 1
                order.append(int(c))
    return int(depth.max())


_SPPM_SEEDS = os.environ.get("TRC_FUZZ_SPPM_SEEDS", "1:5").split(":")


# 77, 534, 673: scenes of a 1 000-seed campaign in which a photon's BSDF sample is NaN -- copysign(1, NaN) and the stored NaNs took the
# platform's sign (x86 / gfx950 differ) until both sides fixed it (oracle.cpp tracePhotonRecord)
@pytest.mark.parametrize("seed", list(range(int(_SPPM_SEEDS[0]), int(_SPPM_SEEDS[1]))) + ([77, 534, 673] if _SPPM_SEEDS == ["1", "5"] else []))
def test_generated_scene_sppm(gpu, seed):
    """the SPPM pass (Photon.metal) on generated scenes: accumulator, canvas RNG, photon and camera records, hash grids"""
    rs = np.random.RandomState(3000 + seed)
    sv, keep = random_scene(rs, n_spheres=int(rs.randint(3, 12)), n_cubes=int(rs.randint(1, 4)), n_tris=int(rs.randint(5, 80)))
    W, H = 64, 48
    cam = host.make_camera((rs.uniform(-100, 100), rs.uniform(-60, 60), -160.0), (0, 0, 0), (0, 1, 0), 0.0, W / H, math.radians(50), 160.0)
    gpu.upload_scene(sv); gpu.set_camera(cam); gpu.set_environment((0.0, 0.0, 0.0)); gpu.resize(W, H)
    gpu.seed(8); gpu.clear_accum(); gpu.sppm_init(40 + seed); gpu.sppm_frames(3)
    dcam, dpho, dmark, dcount, dcx = gpu.sppm_download()
    dacc, drng = gpu.download_accum(), gpu.download_rng()
    rng = host.fill_rng(8, W, H); acc = np.zeros((H, W, 4), np.float32)
    o = po.Sppm(W, H, 40 + seed); o.frames(sv, cam, rng, acc, 3)
    ocam, opho, omark, ocount, ocx = o.download()
    assert np.array_equal(drng, rng) and np.array_equal(dacc.view(np.uint32), acc.view(np.uint32))
    assert np.array_equal(dcount, ocount) and np.array_equal(dmark, omark)
    for what, got, ref, fields in (("photon", dpho, opho, ("flux", "normal", "position", "direction", "step", "active")),
                                   ("camera", dcam, ocam, ("ratio", "position", "direction", "valid", "alternative", "flux", "radius", "photonCount"))):
        for f in fields:
            a, b = np.ascontiguousarray(got[f]), np.ascontiguousarray(ref[f])
            if bytes(memoryview(a)) != bytes(memoryview(b)):
                bad = np.flatnonzero((a.reshape(len(a), -1).view(np.uint8) != b.reshape(len(b), -1).view(np.uint8)).any(axis=1))
                raise AssertionError(f"seed {seed}: {what} record field {f} differs in {len(bad)} records, first {bad[:4]}: "
                                     f"gpu {got[bad[0]]} oracle {ref[bad[0]]}")
    assert dcx.frame_count == ocx.frame_count and dcx.totalPhotonSum == ocx.totalPhotonSum


def test_sixteen_processes_share_the_gpu_and_every_download_arrives():
    """Round 5's one unexplained mismatch -- photon records that read 0 where the oracle has the reset value -- was a TRANSFER: under 16
    processes on one GPU a D2H copy into freshly mapped pageable memory left whole ranges of the destination untouched while the device
    buffer, downloaded again, had the data (tests/campaigns/sppm_stress.py reproduced it 380 times in 9 000 scenes).  Every transfer to /
    from caller memory now goes through the context's pinned staging buffer (trc_copy_to_host / trc_copy_to_device).  Here: the same
    harness, 16 workers x 12 scenes x 3 frames, the photon records downloaded after every frame -- no mismatch."""
    import subprocess, sys
    from conftest import ROOT
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "campaigns", "sppm_stress.py"), "5000", "5192", "--procs", "16"],
                         cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "MISMATCH" not in out.stdout and "0 workers saw a mismatch" in out.stdout, out.stdout[-3000:]
