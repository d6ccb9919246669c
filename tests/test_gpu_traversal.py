"""The traversal the RENDER kernels run (production instantiations of dev_intersect.hpp::trav_iter) through the
Scene::hit test hook: trc_trace_rays(..., TRC_TRACE_PRODUCTION) against the oracle's literal Render.hh:135-252 walk.

Counters are not produced by the production walk; the whole HitRecord is compared bit for bit.  The adversarial
batches aim at the one place where a walk that reorders work across lanes could leave the reference's order: a
primitive that is tested although the reference would already have culled its box with a freshly lowered closest hit
(round 1's speculative round does exactly that: profiles/r05/exp_removed_variants.patch puts it back behind
-DTRC_SPEC_UNCHECKED; built that way it fails test_adversarial_batches, 6 of 6 seeds -- profiles/r02/traversal_teeth.txt).
  * spheres LARGER than their boxes (MakeSphere inflates the radius by 1e-4, not the AABB, Tracer.mm:165-172) with an
    occluder placed between the sphere surface and the box face,
  * rays that start inside many nested / overlapping boxes,
  * coplanar duplicates (identical squares / triangles as distinct primitives: the closest-hit tie goes to the
    reference's traversal order),
  * flat primitives lying IN a face of their box (computed t within an ulp of the box entry).
"""
import os
import ctypes as C

import numpy as np
import pytest

from conftest import camera_rays, make_rays, random_rays
from oracle import pyoracle as po
from tracer_amd import abi, host

F32 = np.float32
HIT_FIELDS = ["hit", "pType", "pIndex", "t", "p", "gn", "sn", "uv", "material", "PDF"]


def assert_records_equal(dev, ref, fields=HIT_FIELDS, what=""):
    for f in fields:
        a, b = np.ascontiguousarray(dev[f]), np.ascontiguousarray(ref[f])
        same = a.view(np.uint32) == b.view(np.uint32)
        if a.dtype == np.float32:
            same = same | (np.isnan(a) & np.isnan(b))
        assert same.all(), f"{what} field {f}: {np.count_nonzero(~same)} mismatches, first at ray {np.argwhere(~same)[0][0]}"


@pytest.mark.gpu
@pytest.mark.parametrize("scene_name", ["cornell", "cornell_spheres", "ball_mesh_scene"])
def test_production_walk_matches_the_reference_walk(gpu, request, scene_name):
    scene = request.getfixturevalue(scene_name)
    gpu.upload_scene(scene.view)
    rays = np.concatenate([random_rays(60000, 31), random_rays(60000, 32, inside_only=True),
                           camera_rays(host.prepare_camera(320, 180), 320, 180)])
    ref = po.trace_rays(scene.view, rays)
    assert_records_equal(gpu.trace_rays(rays, production=True), ref, what="closest")
    # and the instrumented walk agrees with itself through the same call (counters included)
    dev = gpu.trace_rays(rays)
    assert_records_equal(dev, ref, HIT_FIELDS + ["n_descend", "n_return", "n_leaf"], what="instrumented")
    shadow = rays.copy()
    shadow["tmax"] = np.random.RandomState(6).uniform(30, 900, len(rays)).astype(F32)
    ref = po.trace_rays(scene.view, shadow, any_hit=True)
    # any-hit, production: the order-free walk of shadow rays -- only the answer is defined (tracer_abi.h); the
    # instrumented any-hit walk is the reference's, record and counters included
    assert_records_equal(gpu.trace_rays(shadow, any_hit=True, production=True), ref, ["hit"], what="any-hit production")
    assert_records_equal(gpu.trace_rays(shadow, any_hit=True), ref, HIT_FIELDS + ["n_descend", "n_return", "n_leaf"], what="any-hit instrumented")
    assert 0.05 < (ref["hit"] != 0).mean() < 0.95
    prod = gpu.trace_rays(rays, production=True)
    assert not prod["n_descend"].any() and not prod["n_leaf"].any()


@pytest.mark.gpu
@pytest.mark.parametrize("scene_name", ["cornell", "cornell_spheres"])
def test_lanes_outside_the_guarded_ranges_share_wavefronts_with_lanes_inside(gpu, request, scene_name):
    """The short division / square-root forms of dev_vec.hpp run behind ONE wave-uniform guard per site: if any lane's operand
    is outside the proven range, the whole wavefront redoes the site with the compiler's sequence.  Batches that put such lanes
    among ordinary ones, inside the kernels: directions with exact zero components (the Cornell blocks are rotated about y, so
    a ray with d.y == 0 has an object-space direction component of exactly 0: the cube slab divides by it; the square test
    and 1 / direction likewise), denormal and huge components, origins exactly ON the planes of the walls."""
    scene = request.getfixturevalue(scene_name)
    gpu.upload_scene(scene.view)
    base = random_rays(64 * 1500, 77, inside_only=True)
    o, d = np.array(base["origin"], F32), np.array(base["direction"], F32)
    n = len(o)
    lane = np.arange(n) % 64
    d[lane == 3, 1] = 0.0                       # d.y == 0 exactly
    d[lane == 11, 0] = 0.0
    d[lane == 12, 2] = -0.0
    d[lane == 20, 1] = F32(1e-42)               # a denormal component
    d[lane == 21, 0] = F32(-3e-39)
    d[lane == 33] = np.array([0.0, 0.0, 1.0], F32)      # along an axis: two zeros
    d[lane == 34] = np.array([0.0, -1.0, 0.0], F32)
    o[lane == 40, 1] = 0.0                      # on the floor / a wall plane, exactly
    o[lane == 41, 0] = 555.0
    o[lane == 42, 2] = 555.0
    o[lane == 50] *= F32(1e12)                  # far outside: huge numerators
    rays = make_rays(o, d)
    ref = po.trace_rays(scene.view, rays)
    assert_records_equal(gpu.trace_rays(rays, production=True), ref, what="closest, production walk")
    assert_records_equal(gpu.trace_rays(rays), ref, what="closest, instrumented walk")
    assert (ref["pType"][ref["hit"] != 0] == abi.PRIM_CUBE).sum() > 2000     # the cube slab is exercised


@pytest.mark.gpu
@pytest.mark.parametrize("scene_name", ["cornell_spheres", "ball_mesh_scene"])
def test_shadow_rays_that_end_exactly_on_a_surface(gpu, request, scene_name):
    """tmax == the t of the surface the ray ends on (a light sample ON a square, a mesh vertex): Square / Triangle::hit_test
    accept t == range_t.y, Scene::hit still answers range_t.y < test_t = false (Render.hh:244,250).  The order-free walk of the
    production kernels must keep that corner -- a scene generated by the fuzzer met it (a shadow ray ending on a triangle)."""
    scene = request.getfixturevalue(scene_name)
    gpu.upload_scene(scene.view)
    rays = np.concatenate([random_rays(40000, 41, inside_only=True), camera_rays(host.prepare_camera(320, 180), 320, 180)])
    first = po.trace_rays(scene.view, rays)
    hit = first["hit"] != 0
    shadow = rays[hit].copy()
    shadow["tmax"] = first["t"][hit]                                     # exactly the distance of the closest surface
    ref = po.trace_rays(scene.view, shadow, any_hit=True)
    types = first["pType"][hit]
    on_edge = (ref["hit"] == 0) & ((types == 1) | (types == 3))          # squares / triangles at t == tmax: not occluders
    assert on_edge.sum() > 1000
    assert_records_equal(gpu.trace_rays(shadow, any_hit=True, production=True), ref, ["hit"], what="tmax == t, production")
    assert_records_equal(gpu.trace_rays(shadow, any_hit=True), ref, HIT_FIELDS, what="tmax == t, instrumented")
    for scale in (np.float32(1 + 2 ** -20), np.float32(1 - 2 ** -20)):   # and just beyond / just short of it
        s2 = shadow.copy(); s2["tmax"] = shadow["tmax"] * scale
        assert_records_equal(gpu.trace_rays(s2, any_hit=True, production=True), po.trace_rays(scene.view, s2, any_hit=True), ["hit"], what=str(scale))


def _materials(n=24):
    mats = []
    for i in range(n):
        m = abi.Material()
        m.type = [abi.MAT_LAMBERT, abi.MAT_METAL, abi.MAT_GLASS, abi.MAT_PLASTIC][i % 4]
        m.textureInfo.albedo.x = m.textureInfo.albedo.y = m.textureInfo.albedo.z = 0.5
        mats.append(m)
    return mats


def _square(ak, lo_i, hi_i, lo_j, hi_j, k, material):
    ai, aj = [(1, 2), (0, 2), (0, 1)][ak]
    q = abi.Square()
    q.axis_i, q.axis_j, q.axis_k = ai, aj, ak
    q.range_i.x, q.range_i.y, q.range_j.x, q.range_j.y = lo_i, hi_i, lo_j, hi_j
    q.value_k, q.material = k, material
    lo, hi = [0.0] * 3, [0.0] * 3
    lo[ai], hi[ai], lo[aj], hi[aj] = q.range_i.x, q.range_i.y, q.range_j.x, q.range_j.y
    lo[ak], hi[ak] = q.value_k - 1 / 512, q.value_k + 1 / 512      # MakeSquare pads the thin axis, Tracer.mm:137-149
    return q, lo, hi


def adversarial_scene(rs, n_clusters=60, n_flat_tris=120):
    """-> (scene view, keep-alive list, targets): targets = (point, axis, side, jitter, max distance) rays should graze.

    Cluster = a sphere S whose radius exceeds its box by 1e-4 (B-13), a second sphere inside it (so that the SAH
    builder pairs the two spheres and leaves the occluder as their sibling: the occluder's box is entered first and
    the spheres' boxes are tested only AFTER the occluder was accepted) and an occluder plate Q between S's surface
    and S's box face.  Reference walk: Q accepted, then S's box is culled (entry t > t_Q) -- although S is hit closer.
    """
    mats = _materials()
    spheres, squares, leaves, targets = [], [], [], []
    grid = [(x, y, z) for x in range(-2, 3) for y in range(-2, 3) for z in range(-2, 3) if (x, y, z) != (0, 0, 0)]
    rs.shuffle(grid)
    for i in range(n_clusters):
        c = (np.array(grid[i]) * 40.0 + rs.uniform(-5, 5, 3)).astype(F32).astype(np.float64)
        r = float(F32(rs.uniform(3, 9)))
        for rad in (r, r * 0.8):
            s = abi.Sphere()
            s.radius = float(F32(rad) + F32(0.0001))             # larger than its box (MakeSphere, Tracer.mm:165-172)
            s.center.x, s.center.y, s.center.z = c
            s.material = i % len(mats)
            spheres.append(s)
            leaves.append(host.build_node(c - rad, c + rad, abi.PRIM_SPHERE, len(spheres) - 1))
        ak, side = int(rs.randint(3)), float(rs.choice([-1.0, 1.0]))
        k = float(F32(c[ak] + side * (r + rs.uniform(0.00002, 0.00009))))
        ai, aj = [(1, 2), (0, 2), (0, 1)][ak]
        q, lo, hi = _square(ak, c[ai] - 1.0, c[ai] + 1.0, c[aj] - 1.0, c[aj] + 1.0, k, (i + ak) % len(mats))
        for dup in range(1 + i % 2):                             # every other plate twice: coplanar duplicates, the tie
            qq = abi.Square.from_buffer_copy(q)                  # goes to the reference's traversal order
            qq.material = (q.material + dup) % len(mats)
            squares.append(qq)
            leaves.append(host.build_node(lo, hi, abi.PRIM_SQUARE, len(squares) - 1))
        p = c.copy(); p[ak] = k
        targets.append((p, ak, side, 0.01, 10.0))                # the sphere curves away: stay within 0.01 of the pole
    while len(squares) < 7:
        q, lo, hi = _square(1, -5, 5, -5, 5, 90.0, 5)
        squares.append(q); leaves.append(host.build_node(lo, hi, abi.PRIM_SQUARE, len(squares) - 1))
    # nested cubes around the origin: rays starting inside all of them
    cubes = []
    for i in range(6):
        sc = 4.0 + 9.0 * i
        M = np.diag([sc, sc, sc, 1.0]); M[:3, 3] = -sc / 2
        Mi = np.linalg.inv(M)
        cb = abi.Cube()
        for name, mat in (("model_matrix", M), ("inverse_matrix", Mi), ("normal_matrix", Mi.T)):
            m4 = abi.float4x4()
            for col in range(4):
                for row in range(4):
                    setattr(m4.columns[col], "xyzw"[row], float(mat[row][col]))
            setattr(cb, name, m4)
        cb.box.maxi.x = cb.box.maxi.y = cb.box.maxi.z = 1.0
        cb.material = i % len(mats)
        cubes.append(cb)
        leaves.append(host.build_node((0, 0, 0), (1, 1, 1), abi.PRIM_CUBE, i, model=M))
    # flat triangles lying in a face of their own box, each twice (duplicate primitive)
    n_tris = 2 * n_flat_tris
    verts = (abi.TriangleVertex * (3 * n_tris))()
    idx = (C.c_uint32 * (3 * n_tris))(*range(3 * n_tris))
    for t in range(n_flat_tris):
        ak = t % 3
        base = rs.uniform(-70, 70, 3).astype(F32).astype(np.float64)
        p = [base.copy() for _ in range(3)]
        ai, aj = [(1, 2), (0, 2), (0, 1)][ak]
        p[1][ai] += rs.uniform(2, 8); p[2][aj] += rs.uniform(2, 8); p[2][ai] += rs.uniform(-2, 2)
        for dup in range(2):
            tt = 2 * t + dup
            for kk in range(3):
                v = verts[3 * tt + kk]
                v.v[:] = [float(x) for x in p[kk]]
                n = np.zeros(3); n[ak] = 1.0 + 0.25 * kk
                v.n[:] = [float(x) for x in n]; v.uv[:] = [0.25 * kk, 0.5 * dup]
            pf = np.array([[x for x in verts[3 * tt + kk].v] for kk in range(3)])
            leaves.append(host.build_node(pf.min(0), pf.max(0), abi.PRIM_TRIANGLE, tt))
        cen = np.mean(p, axis=0)
        targets.append((cen, ak, 1.0, 0.6, 150.0)); targets.append((cen, ak, -1.0, 0.6, 150.0))
    nodes = host.build_tree(leaves)
    keep = [nodes, verts, idx]
    sv = abi.Scene()
    sv.bvhList, sv.n_bvh = C.cast(nodes, C.POINTER(abi.BVH)), len(nodes)
    for name, items, T_ in (("sphere", spheres, abi.Sphere), ("square", squares, abi.Square), ("cube", cubes, abi.Cube)):
        arr = (T_ * len(items))(*items); keep.append(arr)
        setattr(sv, name + "List", C.cast(arr, C.POINTER(T_))); setattr(sv, "n_" + name, len(items))
    marr = (abi.Material * len(mats))(*mats); keep.append(marr)
    sv.materials, sv.n_material = C.cast(marr, C.POINTER(abi.Material)), len(mats)
    sv.triList, sv.n_vertex = C.cast(verts, C.POINTER(abi.TriangleVertex)), 3 * n_tris
    sv.idxList, sv.n_index = C.cast(idx, C.POINTER(C.c_uint32)), 3 * n_tris
    return sv, keep, targets


def adversarial_rays(rs, targets, per_target=24):
    o, d = [], []
    for p, ak, side, jit, far in targets:
        for _ in range(per_target):
            dist = rs.uniform(0.5, far)
            jitter = rs.uniform(-jit, jit, 3); jitter[ak] = 0.0
            dirv = np.zeros(3); dirv[ak] = -side                 # towards the primitive, from the `side` it faces
            if rs.rand() < 0.5:
                dirv += rs.normal(0, 0.002 if jit > 0.1 else 0.0002, 3)     # almost axis-parallel
            origin = p + jitter - dirv / np.linalg.norm(dirv) * dist
            o.append(origin); d.append(dirv)
    n_inside = 4000                                               # origins inside the nested cubes / many boxes
    o += list(rs.uniform(-2, 2, (n_inside, 3))); d += list(rs.normal(size=(n_inside, 3)))
    o += list(rs.uniform(-80, 80, (n_inside, 3))); d += list(rs.normal(size=(n_inside, 3)))
    return make_rays(np.array(o, F32), np.array(d, F32))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(*[int(x) for x in os.environ.get("TRC_FUZZ_ADV_SEEDS", "1:7").split(":")]))
def test_adversarial_batches(gpu, seed):
    rs = np.random.RandomState(7000 + seed)
    sv, keep, targets = adversarial_scene(rs)
    rays = adversarial_rays(rs, targets)
    gpu.upload_scene(sv)
    ref = po.trace_rays(sv, rays)
    # the batch does what it is meant to: spheres, squares, cubes and triangles are all hit, and some closest hits are
    # primitives the BRUTE-FORCE closest hit disagrees with (the reference culled a box the true closest hit was in)
    assert set(np.unique(ref["pType"][ref["hit"] != 0])) == {0, 1, 2, 3}
    assert_records_equal(gpu.trace_rays(rays), ref, HIT_FIELDS + ["n_descend", "n_return", "n_leaf"], what=f"seed {seed} instrumented")
    assert_records_equal(gpu.trace_rays(rays, production=True), ref, what=f"seed {seed} production")
    shadow = rays.copy()
    shadow["tmax"] = rs.uniform(1.0, 160.0, len(rays)).astype(F32)
    ref_any = po.trace_rays(sv, shadow, any_hit=True)
    assert_records_equal(gpu.trace_rays(shadow, any_hit=True, production=True), ref_any, ["hit"], what=f"seed {seed} any-hit production")
    assert_records_equal(gpu.trace_rays(shadow, any_hit=True), ref_any, HIT_FIELDS + ["n_descend", "n_return", "n_leaf"], what=f"seed {seed} any-hit instrumented")
    assert 0.05 < (ref_any["hit"] != 0).mean() < 0.95


def test_adversarial_scene_is_adversarial():
    """CPU-side sanity of the generator (no GPU): on some rays the reference walk returns an occluder although a sphere
    is hit CLOSER (brute force finds it) -- the reference culled the sphere's box with the occluder's t.  That is the
    situation a speculative walk must not 'repair'."""
    rs = np.random.RandomState(7001)
    sv, keep, targets = adversarial_scene(rs)
    rays = adversarial_rays(rs, targets)
    ref, brute = po.trace_rays(sv, rays), po.trace_rays(sv, rays, brute=True)
    culled_sphere = (brute["pType"] == 0) & (ref["pType"] == 1) & (brute["t"] < ref["t"])
    assert culled_sphere.sum() > 500, culled_sphere.sum()
