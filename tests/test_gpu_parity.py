"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on identical inputs.

Bar: BIT-EXACT for everything -- hit identity (pType, pIndex), t, the whole HitRecord, traversal
counters, RNG state and the radiance accumulator (both sides use IEEE binary32 without FMA
contraction and include/trc_detmath.h for transcendentals, so no tolerance is needed).
"""
import numpy as np
import pytest

from conftest import camera_rays, random_rays
from oracle import pyoracle as po
from tracer_amd import abi, host

pytestmark = pytest.mark.gpu

HIT_FIELDS = ["hit", "pType", "pIndex", "t", "p", "gn", "sn", "uv", "material", "PDF"]
COUNTER_FIELDS = ["n_descend", "n_return", "n_leaf"]


def assert_hits_equal(dev, ref, counters=True):
    for f in HIT_FIELDS + (COUNTER_FIELDS if counters else []):
        a, b = dev[f], ref[f]
        same = (a.view(np.uint32) == b.view(np.uint32)) if a.dtype == np.float32 else (a == b)
        if a.dtype == np.float32:   # NaN payloads aside, require identical bits; allow NaN==NaN
            same = same | (np.isnan(a) & np.isnan(b))
        assert same.all(), f"field {f}: {np.count_nonzero(~same)} mismatches, first at {np.argwhere(~same)[0]}"


@pytest.mark.parametrize("scene_name", ["cornell", "cornell_spheres", "ball_mesh_scene"])
def test_trace_rays_closest_hit_bit_exact(gpu, request, scene_name):
    scene = request.getfixturevalue(scene_name)
    gpu.upload_scene(scene.view)
    rays = np.concatenate([random_rays(60000, 11), random_rays(60000, 12, inside_only=True),
                           camera_rays(host.prepare_camera(320, 180), 320, 180)])
    dev = gpu.trace_rays(rays)
    ref = po.trace_rays(scene.view, rays)
    assert ref["hit"].mean() > 0.3
    assert_hits_equal(dev, ref)


@pytest.mark.parametrize("scene_name", ["cornell_spheres", "ball_mesh_scene"])
def test_trace_rays_any_hit_bit_exact(gpu, request, scene_name):
    """Shadow-ray mode (Render.hh:244): early exit on the first accepted primitive."""
    scene = request.getfixturevalue(scene_name)
    gpu.upload_scene(scene.view)
    rays = random_rays(80000, 21, inside_only=True)
    rays["tmax"] = np.random.RandomState(5).uniform(50, 900, len(rays)).astype(np.float32)
    dev = gpu.trace_rays(rays, any_hit=True)
    ref = po.trace_rays(scene.view, rays, any_hit=True)
    assert 0.05 < ref["hit"].mean() < 0.98
    # any-hit only promises the boolean + counters; the record of the accepted primitive is also identical
    assert_hits_equal(dev, ref)


def test_trace_rays_degenerate_directions(gpu, cornell_spheres):
    """Axis-parallel rays: 1/0 = inf and 0*inf = NaN inside the slab test (SURVEY B-10)."""
    gpu.upload_scene(cornell_spheres.view)
    o = np.array([[278, 278, -800], [278, 278, 100], [100, 300, 300], [0, 0, 0], [555, 554.9, 277.5]], np.float32)
    dirs = np.array([[0, 0, 1], [0, 1, 0], [1, 0, 0], [0, -1, 0], [-1, 0, 0], [0, 0, -1]], np.float32)
    O = np.repeat(o, len(dirs), 0)
    D = np.tile(dirs, (len(o), 1))
    from tracer_amd.dtypes import make_rays
    rays = make_rays(O, D)
    assert_hits_equal(gpu.trace_rays(rays), po.trace_rays(cornell_spheres.view, rays))


def _render_both(gpu, scene, W, H, spp, integrator, seed=0x5EED0000, env=(0.0, 0.0, 0.0), frame0=0, max_depth=8):
    cam = host.prepare_camera(W, H)
    gpu.upload_scene(scene.view)
    gpu.set_camera(cam)
    gpu.set_environment(env)
    gpu.resize(W, H)
    gpu.seed(seed)
    gpu.reset_stats()
    gpu.render(spp=spp, integrator=integrator, frame0=frame0, max_depth=max_depth, collect_stats=True)
    dev_acc, dev_rng, dev_stats = gpu.download_accum(), gpu.download_rng(), gpu.stats()
    rng = host.fill_rng(seed, W, H)
    ref_acc, ref_stats = po.render(scene.view, cam, W, H, rng, spp=spp, integrator=integrator, env=env,
                                   frame0=frame0, max_depth=max_depth)
    return dev_acc, dev_rng, dev_stats, ref_acc, rng, ref_stats


STAT_FIELDS = ["paths", "rays", "shaded", "n_descend", "n_return", "n_leaf_sphere", "n_leaf_square", "n_leaf_cube",
               "n_leaf_triangle", "n_hit_triangle", "n_hit_cube"]


def assert_frames_equal(dev_acc, dev_rng, dev_stats, ref_acc, ref_rng, ref_stats):
    assert np.array_equal(dev_rng, ref_rng), "RNG texture differs"
    bad = dev_acc.view(np.uint32) != ref_acc.view(np.uint32)
    assert not bad.any(), (f"accumulator differs in {np.count_nonzero(bad.any(axis=2))} pixels; first "
                           f"{np.argwhere(bad)[0]} dev={dev_acc[tuple(np.argwhere(bad)[0][:2])]} "
                           f"ref={ref_acc[tuple(np.argwhere(bad)[0][:2])]}")
    for f in STAT_FIELDS:
        assert getattr(dev_stats, f) == getattr(ref_stats, f), f"stat {f}: {getattr(dev_stats, f)} != {getattr(ref_stats, f)}"


@pytest.mark.parametrize("scene_name,integrator,W,H,spp", [
    ("cornell", abi.INTEGRATOR_PATH, 160, 90, 8),
    ("cornell_spheres", abi.INTEGRATOR_PATH, 160, 90, 16),
    ("cornell_spheres", abi.INTEGRATOR_MIS, 160, 90, 8),
    ("cornell_spheres", abi.INTEGRATOR_PATH, 97, 61, 4),       # ragged: partial tiles on both edges
    ("ball_mesh_scene", abi.INTEGRATOR_PATH, 128, 72, 8),
    ("ball_mesh_scene", abi.INTEGRATOR_MIS, 128, 72, 4),
])
def test_render_bit_exact(gpu, request, scene_name, integrator, W, H, spp):
    scene = request.getfixturevalue(scene_name)
    out = _render_both(gpu, scene, W, H, spp, integrator)
    assert out[3][..., :3].max() > 0, "oracle image is black: the test would be vacuous"
    assert_frames_equal(*out)


def test_render_with_sky_environment(gpu, cornell_spheres):
    """constant environment (0.5, 0.7, 1.0): the miss branch of tracePath (Render.metal:434-439)"""
    assert_frames_equal(*_render_both(gpu, cornell_spheres, 96, 54, 8, abi.INTEGRATOR_PATH, env=(0.5, 0.7, 1.0)))


@pytest.mark.parametrize("integrator", [abi.INTEGRATOR_PATH, abi.INTEGRATOR_MIS, abi.INTEGRATOR_VOLUME])
def test_shallow_and_deep_paths(gpu, cornell_spheres, integrator):
    """depth is the reference's `do { ... } while ((--depth) > 0)` (Render.metal:406,489): 0 and 1 both trace one bounce
    ray at most, large depths are cut by Russian roulette; frames that do not start at 0 weigh the running mean"""
    for depth, frame0 in ((0, 0), (1, 3), (2, 0), (3, 1000), (50, 7)):
        assert_frames_equal(*_render_both(gpu, cornell_spheres, 72, 40, 4, integrator, seed=77 + depth, frame0=frame0,
                                          max_depth=depth, env=(0.2, 0.3, 0.4)))


def test_seed_matches_host_fill(gpu):
    gpu.resize(131, 77)
    gpu.seed(1234567)
    assert np.array_equal(gpu.download_rng(), host.fill_rng(1234567, 131, 77))


def test_spp_fusion_equals_per_frame_launches(gpu, cornell_spheres):
    """One launch of spp samples == spp launches of 1 sample (the reference's 1 spp/frame progressive loop)."""
    W, H, spp = 96, 64, 6
    cam = host.prepare_camera(W, H)
    gpu.upload_scene(cornell_spheres.view)
    gpu.set_camera(cam)
    gpu.set_environment((0, 0, 0))
    gpu.resize(W, H)
    gpu.seed(77)
    gpu.render(spp=spp)
    fused_acc, fused_rng = gpu.download_accum(), gpu.download_rng()
    gpu.clear_accum()
    gpu.seed(77)
    for f in range(spp):
        gpu.render(spp=1, frame0=f)
    assert np.array_equal(gpu.download_rng(), fused_rng)
    assert np.array_equal(gpu.download_accum().view(np.uint32), fused_acc.view(np.uint32))


@pytest.mark.parametrize("nranks", [2, 3, 8])
def test_tile_shards_compose_the_single_gpu_frame(gpu, cornell_spheres, nranks):
    """Rank r renders tiles (tx+ty)%N==r into a zero frame; the SUM over ranks (what the RCCL reduce
    computes) is bit-identical to the 1-GPU frame because per-pixel RNG streams depend on (x,y,seed) only."""
    W, H, spp = 112, 80, 4
    cam = host.prepare_camera(W, H)
    gpu.upload_scene(cornell_spheres.view)
    gpu.set_camera(cam)
    gpu.set_environment((0.1, 0.1, 0.1))
    gpu.resize(W, H)
    gpu.seed(9)
    gpu.render(spp=spp)
    full = gpu.download_accum()
    total = np.zeros_like(full)
    for r in range(nranks):
        gpu.clear_accum()
        gpu.seed(9)
        gpu.render(spp=spp, tile_rank=r, tile_nranks=nranks)
        part = gpu.download_accum()
        ty, tx = np.mgrid[0:H, 0:W] // abi.TRC_TILE
        mine = ((tx + ty) % nranks) == r
        assert not part[~mine].any(), "a rank wrote outside its tiles"
        total += part
    assert np.array_equal(total.view(np.uint32), full.view(np.uint32))


def test_full_size_tile_subsample_matches_oracle(gpu, cornell_spheres):
    """BASELINE config 2 at full size (1920x1080x64spp): the oracle renders 1 tile in 64 (every tile with
    (tx+ty)%64==0, ~32k pixels, 2 M paths); those pixels must be bit-identical in the full GPU frame."""
    W, H, spp = 1920, 1080, 64
    cam = host.prepare_camera(W, H)
    gpu.upload_scene(cornell_spheres.view)
    gpu.set_camera(cam)
    gpu.set_environment((0, 0, 0))
    gpu.resize(W, H)
    gpu.seed(0x5EED0000)
    gpu.reset_stats()
    gpu.render(spp=spp)
    dev = gpu.download_accum()
    st = gpu.stats()
    assert st.paths == W * H * spp
    rng = host.fill_rng(0x5EED0000, W, H)
    ref, _ = po.render(cornell_spheres.view, cam, W, H, rng, spp=spp, tile_rank=0, tile_nranks=64)
    ty, tx = np.mgrid[0:H, 0:W] // abi.TRC_TILE
    mine = ((tx + ty) % 64) == 0
    assert mine.sum() > 20000
    assert np.array_equal(dev[mine].view(np.uint32), ref[mine].view(np.uint32))
    # size-independent properties of the full frame: finite, alpha 1, deterministic
    assert np.isfinite(dev).all() and (dev[..., 3] == 1.0).all()
    gpu.clear_accum()
    gpu.seed(0x5EED0000)
    gpu.render(spp=spp)
    assert np.array_equal(gpu.download_accum().view(np.uint32), dev.view(np.uint32))


def test_error_paths(gpu, cornell):
    from tracer_amd.device import Tracer, TracerError
    t = Tracer(0)
    try:
        with pytest.raises(TracerError) as e:
            t.render(spp=1)
        assert e.value.status == abi.ERR_NO_SCENE
        t.upload_scene(cornell.view)
        with pytest.raises(TracerError) as e:
            t.render(spp=1)
        assert e.value.status == abi.ERR_NO_FRAME
        # malformed tree: child index out of range
        import ctypes as C
        nodes = (abi.BVH * cornell.view.n_bvh)()
        C.memmove(nodes, cornell.view.bvhList, C.sizeof(nodes))
        nodes[0].left = 10_000
        bad = abi.Scene()
        C.memmove(C.byref(bad), C.byref(cornell.view), C.sizeof(bad))
        bad.bvhList = C.cast(nodes, C.POINTER(abi.BVH))
        with pytest.raises(TracerError) as e:
            t.upload_scene(bad)
        assert e.value.status == abi.ERR_BVH_INVALID
    finally:
        t.close()


def test_committed_golden_frames(gpu):
    """tests/golden/frames.npz (made by tests/golden/make_frame_fixtures.py): accumulator, RNG texture
    and work counters, bit for bit."""
    import os
    import sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_frame_fixtures as fx
    frames = np.load(os.path.join(ROOT, "tests", "golden", "frames.npz"))
    for name, (kind, integ, W, H, spp, seed, env) in fx.CASES.items():
        scene = host.HostScene(kind)
        gpu.upload_scene(scene.view)
        gpu.set_camera(host.prepare_camera(W, H))
        gpu.set_environment(env)
        gpu.resize(W, H)
        gpu.seed(seed)
        gpu.reset_stats()
        gpu.render(spp=spp, integrator=integ, collect_stats=True)
        assert np.array_equal(gpu.download_accum().view(np.uint32), frames[name + "_accum"].view(np.uint32)), name
        assert np.array_equal(gpu.download_rng(), frames[name + "_rng"]), name
        st = gpu.stats()
        counts = [st.paths, st.rays, st.shaded, st.n_descend, st.n_return, st.n_leaf_sphere, st.n_leaf_square, st.n_leaf_cube]
        assert counts == list(frames[name + "_counts"]), name


def test_rccl_group_single_rank_roundtrip(gpu, cornell_spheres):
    """trc_group_* with a 1-rank communicator: ncclGetUniqueId / ncclCommInitRank / ncclReduce(sum, root 0) /
    ncclCommDestroy resolve and run (RCCL is dlopen'ed); a 1-rank sum-reduce must leave the frame unchanged."""
    from tracer_amd.device import group_unique_id
    W, H = 64, 48
    gpu.upload_scene(cornell_spheres.view)
    gpu.set_camera(host.prepare_camera(W, H))
    gpu.resize(W, H)
    gpu.seed(3)
    gpu.render(spp=2)
    before = gpu.download_accum()
    uid = group_unique_id()
    assert len(uid) == abi.TRC_UNIQUE_ID_BYTES and any(uid)
    gpu.group_init(uid, 1, 0)
    gpu.group_reduce_accum(0)
    gpu.synchronize()
    after = gpu.download_accum()
    assert np.array_equal(before.view(np.uint32), after.view(np.uint32))
    # pipelined compose: three different frames in flight over the two accumulators; every composed frame must be
    # the frame rendered in its step, and the context keeps rendering into the other buffer meanwhile
    frames = []
    for seed in (11, 12, 13):
        gpu.clear_accum(); gpu.seed(seed); gpu.render(spp=2)
        gpu.group_reduce_accum_async(0)
        frames.append(gpu.download_composed())
    gpu.synchronize()
    for seed, got in zip((11, 12, 13), frames):
        gpu.clear_accum(); gpu.seed(seed); gpu.render(spp=2)
        assert np.array_equal(gpu.download_accum().view(np.uint32), got.view(np.uint32)), seed
    # back-to-back without reading in between: the last composed frame is the last rendered one
    for seed in (21, 22, 23, 24):
        gpu.clear_accum(); gpu.seed(seed); gpu.render(spp=1)
        gpu.group_reduce_accum_async(0)
    gpu.synchronize()
    last = gpu.download_composed()
    gpu.clear_accum(); gpu.seed(24); gpu.render(spp=1)
    assert np.array_equal(gpu.download_accum().view(np.uint32), last.view(np.uint32))
    # sample sharding over the same 1-rank communicator (ncclGroupStart / End with no peer inside, the fold kernel, the
    # all-gather): one rank's sum over one group is the frame itself, on the root, pipelined, and delivered to every rank
    gpu.clear_accum(); gpu.seed(31); gpu.render(spp=3)
    before = gpu.download_accum()
    gpu.group_compose_samples(0)
    assert np.array_equal(gpu.download_composed().view(np.uint32), before.view(np.uint32))
    gpu.group_compose_samples_async(0, 1)
    assert np.array_equal(gpu.download_composed().view(np.uint32), before.view(np.uint32))
    gpu.synchronize(); gpu.clear_accum(); gpu.seed(31); gpu.render(spp=3)
    gpu.group_allreduce_mean_accum()
    assert np.array_equal(gpu.download_accum().view(np.uint32), before.view(np.uint32))
    with pytest.raises(Exception):
        gpu.group_compose_samples(0, 3)                      # 1 rank is not 3 sample groups x tile ranks
    gpu.group_finalize()


def test_sample_sharding_is_bit_defined(gpu, cornell_spheres):
    """Sample sharding has a bit-level definition (include/tracer_abi.h): group g renders the whole frame with
    spp / S samples from trc_seed(trc_shard_seed(seed, g)); composed texel = rank-ordered binary32 sum / S.  The shards rendered
    one after the other on this GPU and folded in numpy == oracle/pyoracle.py::render_sample_sharded, bit for bit, for both
    integrators; group 0 keeps the seed, so one group IS the unsharded frame; the library's seed function is the documented one.
    (The collective itself: 2 and 8 ranks on one GPU in test_gpu_shared_gpu_ranks.py, a 1-rank group below.)"""
    import ctypes as C
    from tracer_amd import device
    W, H, spp, S = 96, 64, 16, 4
    cam = host.prepare_camera(W, H)
    gpu.upload_scene(cornell_spheres.view); gpu.set_camera(cam); gpu.set_environment((0.0, 0.0, 0.0)); gpu.resize(W, H)
    L = device.lib()
    rs = np.random.RandomState(5)
    for seed, g in [(0, 0), (0x5EED0000, 7), (2 ** 64 - 1, 1), (2 ** 63, 2 ** 32 - 1)] + [(int(rs.randint(0, 2 ** 62)) * 3, int(rs.randint(0, 2 ** 31))) for _ in range(32)]:
        assert L.trc_shard_seed(C.c_uint64(seed), C.c_uint32(g)) == abi.shard_seed(seed, g) == po.shard_seed(seed, g)
    assert abi.shard_seed(0x5EED0000, 0) == 0x5EED0000 and abi.shard_seed(1, 1) == 0x9E3779B97F4A7C16
    for integ in (abi.INTEGRATOR_PATH, abi.INTEGRATOR_MIS):
        shards = []
        for g in range(S):
            gpu.clear_accum(); gpu.seed(abi.shard_seed(321, g)); gpu.render(spp=spp // S, integrator=integ)
            shards.append(gpu.download_accum())
        acc = shards[0]
        for a in shards[1:]:
            acc = np.add(acc, a, dtype=np.float32)
        got = np.divide(acc, np.float32(S), dtype=np.float32)
        want, _ = po.render_sample_sharded(cornell_spheres.view, cam, W, H, [host.fill_rng(po.shard_seed(321, g), W, H) for g in range(S)],
                                           spp, integrator=integ)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)) and (want[..., 3] == 1).all()
    one, _ = po.render_sample_sharded(cornell_spheres.view, cam, W, H, [host.fill_rng(321, W, H)], spp)
    plain, _ = po.render(cornell_spheres.view, cam, W, H, host.fill_rng(321, W, H), spp=spp)
    assert np.array_equal(one.view(np.uint32), plain.view(np.uint32))


def test_stacked_views_sharded_over_ranks(gpu, cornell_spheres):
    """bench.py's N-GPU workload in miniature: 3 views stacked into one frame (trc_params.view_height), tiles of ranks
    0..2 rendered one after the other and sum-composed == the oracle's render of the whole stack"""
    W, h, k = 80, 56, 3
    cam = host.prepare_camera(W, h)
    gpu.upload_scene(cornell_spheres.view); gpu.set_camera(cam); gpu.set_environment((0.0, 0.0, 0.0)); gpu.resize(W, h * k)
    rng = host.fill_rng(31, W, h * k)
    composed = np.zeros((h * k, W, 4), np.float32)
    rays = 0
    for r in range(k):
        gpu.upload_rng(rng); gpu.clear_accum(); gpu.reset_stats()
        gpu.render(spp=4, tile_rank=r, tile_nranks=k, view_height=h)
        composed += gpu.download_accum()                    # non-owned tiles are zero: sum == gather
        rays += gpu.stats().rays
    ref_rng = rng.copy()
    ref, st = po.render(cornell_spheres.view, cam, W, h * k, ref_rng, spp=4, view_height=h)
    assert np.array_equal(composed.view(np.uint32), ref.view(np.uint32)) and rays == st.rays
    # every view is the single-view render of its RNG slice
    gpu.resize(W, h)
    for i in range(k):
        gpu.upload_rng(np.ascontiguousarray(rng[i * h:(i + 1) * h])); gpu.clear_accum(); gpu.render(spp=4)
        assert np.array_equal(gpu.download_accum().view(np.uint32), ref[i * h:(i + 1) * h].view(np.uint32))


def test_adaptive_launch_order_does_not_change_pixels(gpu, cornell_spheres):
    """from the second launch on the blocks are launched most-expensive-first (rays of the previous launch); the frame,
    the RNG texture and the counters must not notice"""
    W, H = 200, 136                      # not a multiple of the block size: ragged blocks take part in the sort
    cam = host.prepare_camera(W, H)
    gpu.upload_scene(cornell_spheres.view); gpu.set_camera(cam); gpu.set_environment((0.0, 0.0, 0.0)); gpu.resize(W, H)
    frames = []
    for fixed in (True, False, False, True, False):
        gpu.seed(77); gpu.clear_accum(); gpu.reset_stats()
        gpu.render(spp=6, fixed_order=fixed)
        frames.append((gpu.download_accum(), gpu.download_rng(), gpu.stats().rays))
    for acc, rng, rays in frames[1:]:
        assert np.array_equal(acc.view(np.uint32), frames[0][0].view(np.uint32))
        assert np.array_equal(rng, frames[0][1]) and rays == frames[0][2]
    ref_rng = host.fill_rng(77, W, H)
    ref, st = po.render(cornell_spheres.view, cam, W, H, ref_rng, spp=6)
    assert np.array_equal(frames[-1][0].view(np.uint32), ref.view(np.uint32)) and st.rays == frames[-1][2]


@pytest.mark.parametrize("W,H,spp,integrator", [(160, 90, 16, abi.INTEGRATOR_PATH), (97, 61, 3, abi.INTEGRATOR_MIS),
                                               (131, 77, 9, abi.INTEGRATOR_VOLUME)])
def test_small_pixel_blocks_render_the_same_frame(gpu, cornell_spheres, W, H, spp, integrator):
    """TRC_FLAG_SMALL_BLOCKS (one 4x4 pixel block on 16 lanes per wavefront, the strong-scaling launch geometry) is a
    scheduling choice: frame, RNG texture and ray count equal the 8x8-block launch, ragged frames and few-sample (strip)
    launches included, also for a rank's share of the tiles"""
    cam = host.prepare_camera(W, H)
    gpu.upload_scene(cornell_spheres.view); gpu.set_camera(cam); gpu.set_environment((0.2, 0.3, 0.4)); gpu.resize(W, H)
    out = []
    for small in (False, True):
        for rank, nranks in ((0, 1), (1, 3)):
            gpu.seed(99); gpu.clear_accum(); gpu.reset_stats()
            gpu.render(spp=spp, integrator=integrator, small_blocks=small, tile_rank=rank, tile_nranks=nranks)
            out.append((gpu.download_accum(), gpu.download_rng(), gpu.stats().rays))
    for a, b in ((out[0], out[2]), (out[1], out[3])):
        assert a[2] == b[2] and np.array_equal(a[1], b[1]) and np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))
    ref, rst = po.render(cornell_spheres.view, cam, W, H, host.fill_rng(99, W, H), spp=spp, integrator=integrator, env=(0.2, 0.3, 0.4))
    assert rst.rays == out[2][2] and np.array_equal(out[2][0].view(np.uint32), ref.view(np.uint32))


def test_the_descent_threshold_is_scheduling_only(gpu, ball_mesh_scene):
    """knob descend_min (DScene::descend_min: how many lanes of a wavefront must still be descending for the box-step loop to go on
    while others wait with a leaf; 12 by default, 6 for scenes beyond the Infinity Cache): every lane performs its own sequence of
    the reference's steps whatever the interleaving -- the frame, the RNG texture and the ray count do not move"""
    W, H, spp = 256, 160, 8
    cam = host.prepare_camera(W, H)
    gpu.upload_scene(ball_mesh_scene.view); gpu.set_camera(cam); gpu.set_environment((0.0, 0.0, 0.0)); gpu.resize(W, H)
    frames = []
    try:
        for integ in (abi.INTEGRATOR_PATH, abi.INTEGRATOR_MIS):
            for n in (0, 1, 2, 6, 24, 64):
                gpu.debug_set("descend_min", n)
                gpu.seed(99); gpu.clear_accum(); gpu.reset_stats(); gpu.render(spp=spp, integrator=integ)
                frames.append((integ, n, gpu.download_accum(), gpu.download_rng(), gpu.stats().rays))
    finally:
        gpu.debug_set("descend_min", 0)
    for integ in (abi.INTEGRATOR_PATH, abi.INTEGRATOR_MIS):
        mine = [f for f in frames if f[0] == integ]
        ref, _ = po.render(ball_mesh_scene.view, cam, W, H, host.fill_rng(99, W, H), spp=spp, integrator=integ)
        for _, n, acc, rng, rays in mine:
            assert np.array_equal(acc.view(np.uint32), ref.view(np.uint32)), (integ, n)
            assert np.array_equal(rng, mine[0][3]) and rays == mine[0][4], (integ, n)

