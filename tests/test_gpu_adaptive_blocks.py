"""Cost-adaptive block size (trc_abi.hip::k_plan_split): from the second launch of a block list on, the 8x8 blocks whose
previous launch lasted longest run as four 4x4 quarters on 16 lanes each -- and a first launch with no more blocks than
wavefront slots runs every block as quarters.  Which blocks are split is a scheduling choice made from measured durations;
pixels are independent, so every launch must produce the frame, the RNG texture and the ray count of the plain launch
(TRC_FLAG_LARGE_BLOCKS | TRC_FLAG_FIXED_ORDER), which the other parity tests compare with the oracle."""
import numpy as np
import pytest

from oracle import pyoracle as po
from tracer_amd import abi, host

pytestmark = pytest.mark.gpu


def _launch(gpu, seed, **kw):
    gpu.seed(seed); gpu.clear_accum(); gpu.reset_stats()
    gpu.render(**kw)
    frame, rng, rays = gpu.download_accum(), gpu.download_rng(), gpu.stats().rays
    _, costs, _ = gpu.block_costs()
    _launch.sixteenths = int(((costs >> 30) & 1).sum())          # blocks some of whose quarters ran as 2x2 sixteenths
    _launch.pixels = int(((costs >> 29) & 1).sum())              # ... and some of those as single pixels
    return frame, rng, rays, int((costs >> 31).sum()), len(costs)


def _same(a, b):
    return a[2] == b[2] and np.array_equal(a[1], b[1]) and np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))


def test_quarters_on_an_lds_resident_scene(gpu, cornell_spheres):
    W, H, spp = 640, 360, 9                         # 3 600 blocks <= 4 096 wavefront slots: the first launch is all quarters
    cam = host.prepare_camera(W, H)
    gpu.upload_scene(cornell_spheres.view); gpu.set_camera(cam); gpu.set_environment((0.2, 0.3, 0.4)); gpu.resize(W, H)
    gpu.debug_set("no_split", 0)                    # also resets what the context knows about the blocks' costs
    plain = _launch(gpu, 5, spp=spp, small_blocks=False, fixed_order=True)
    assert plain[3] == 0
    gpu.debug_set("no_split", 0)
    runs = [_launch(gpu, 5, spp=spp) for _ in range(4)]
    for r in runs:
        assert _same(r, plain)
    assert runs[0][3] == runs[0][4] == 3600         # launch 1: every block as quarters
    assert 0 < runs[1][3] < 3600                    # launch 2 on: only the blocks the plan picks
    ref, st = po.render(cornell_spheres.view, cam, W, H, host.fill_rng(5, W, H), spp=spp, env=(0.2, 0.3, 0.4))
    assert st.rays == plain[2] and np.array_equal(plain[0].view(np.uint32), ref.view(np.uint32))
    # a different seed every launch (a progressive renderer never replays a frame): still the plain frames
    for s in (6, 7):
        a = _launch(gpu, s, spp=spp)
        gpu.debug_set("no_split", 1)
        b = _launch(gpu, s, spp=spp, small_blocks=False, fixed_order=True)
        gpu.debug_set("no_split", 0)
        assert _same(a, b)


def test_sixteenths_where_wavefront_slots_are_idle(gpu, cornell_spheres):
    """second level of the plan: with fewer launch entries than wavefront slots, the quarters the launch ends on run as four
    2x2 blocks on 4 lanes each from the next launch on -- same frame, whatever the plan picks"""
    W, H, spp = 640, 360, 24
    cam = host.prepare_camera(W, H)
    gpu.upload_scene(cornell_spheres.view); gpu.set_camera(cam); gpu.set_environment((0.2, 0.3, 0.4)); gpu.resize(W, H)
    gpu.debug_set("no_split", 0)
    plain = _launch(gpu, 9, spp=spp, small_blocks=False, fixed_order=True)
    gpu.debug_set("no_cold_probe", 1)               # launch 1 as ONE cold launch (not head + rest: test_first_launch_head_and_rest)
    try:
        deep = []
        for k in range(7):
            r = _launch(gpu, 9, spp=spp)
            deep.append(_launch.sixteenths)
            assert _same(r, plain), k
    finally:
        gpu.debug_set("no_cold_probe", 0)
    assert deep[0] == 0 and max(deep) > 0, deep       # launch 1 is all quarters; sixteenths need a measured quarter first
    ref, st = po.render(cornell_spheres.view, cam, W, H, host.fill_rng(9, W, H), spp=spp, env=(0.2, 0.3, 0.4))
    assert st.rays == plain[2] and np.array_equal(plain[0].view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("scene_kind,integrator", [("spheres", abi.INTEGRATOR_PATH), ("mesh", abi.INTEGRATOR_PATH), ("mesh", abi.INTEGRATOR_MIS)])
def test_single_pixels_where_a_share_ends_on_one_sixteenth(gpu, cornell_spheres, scene_kind, integrator):
    """third level of the plan (round 6): a share with far fewer entries than wavefront slots ends on its slowest 2x2 sixteenths -- four
    divergent sample chains on one wavefront; those run as four single pixels (one lane each: the chain floor) from the launch after
    they were measured.  Same frame, RNG texture and ray count whatever the plan picks, on the LDS-resident kernels and on the
    persistent workgroups of a tree read from memory."""
    W, H, spp = 480, 272, 32
    scene = cornell_spheres if scene_kind == "spheres" else host.HostScene(abi.SCENE_CORNELL_MESH, host.Mesh.golden("teapot"))
    cam = host.prepare_camera(W, H)
    gpu.upload_scene(scene.view); gpu.set_camera(cam); gpu.set_environment((0.2, 0.3, 0.4)); gpu.resize(W, H)
    gpu.debug_set("no_split", 0)
    kw = dict(spp=spp, integrator=integrator, tile_rank=1, tile_nranks=2)          # 1 020 blocks for 4 096+ wavefront slots
    plain = _launch(gpu, 13, small_blocks=False, fixed_order=True, **kw)
    gpu.debug_set("no_split", 0)
    gpu.debug_set("no_plan_reuse", 1)               # re-plan before every launch
    try:
        pixels, sixteenths = [], []
        for k in range(9):
            r = _launch(gpu, 13, **kw)
            pixels.append(_launch.pixels); sixteenths.append(_launch.sixteenths)
            assert _same(r, plain), (k, pixels)
    finally:
        gpu.debug_set("no_plan_reuse", 0)
    assert max(sixteenths) > 0 and max(pixels) > 0, (sixteenths, pixels)
    first = next(k for k, v in enumerate(pixels) if v)
    assert sixteenths[first - 1] > 0, (sixteenths, pixels)          # a sixteenth is measured as one before it becomes pixels
    if scene_kind == "spheres":
        ref, st = po.render(scene.view, cam, W, H, host.fill_rng(13, W, H), spp=spp, env=(0.2, 0.3, 0.4), tile_rank=1, tile_nranks=2, integrator=integrator)
        mine = np.zeros((H, W), bool)
        ty, tx = np.meshgrid(np.arange(H) // 16, np.arange(W) // 16, indexing="ij")
        mine[(tx + ty) % 2 == 1] = True
        assert st.rays == plain[2] and np.array_equal(plain[0][mine].view(np.uint32), ref[mine].view(np.uint32))


@pytest.mark.parametrize("integrator", [abi.INTEGRATOR_PATH, abi.INTEGRATOR_MIS])
def test_quarters_on_a_tree_read_from_memory(gpu, integrator):
    """a rank's share of a mesh scene: persistent workgroups pull whole blocks and quarters from one queue"""
    W, H, spp = 960, 540, 8
    scene = host.HostScene(abi.SCENE_CORNELL_MESH, host.Mesh.golden("teapot"))
    gpu.upload_scene(scene.view); gpu.set_camera(host.prepare_camera(W, H)); gpu.set_environment((0.0, 0.0, 0.0)); gpu.resize(W, H)
    gpu.debug_set("no_split", 0)
    plain = _launch(gpu, 11, spp=spp, integrator=integrator, tile_rank=1, tile_nranks=2, small_blocks=False, fixed_order=True)
    gpu.debug_set("no_split", 0)
    runs = [_launch(gpu, 11, spp=spp, integrator=integrator, tile_rank=1, tile_nranks=2) for _ in range(4)]
    for r in runs:
        assert _same(r, plain)
    assert runs[0][3] == 0 and max(r[3] for r in runs[1:]) > 0, [r[3] for r in runs]


@pytest.mark.parametrize("integrator,mesh,spp", [(abi.INTEGRATOR_PATH, False, 20), (abi.INTEGRATOR_PATH, True, 20), (abi.INTEGRATOR_MIS, True, 20),
                                                 (abi.INTEGRATOR_PATH, False, 72), (abi.INTEGRATOR_MIS, True, 136)])
def test_first_launch_head_and_rest(gpu, cornell_spheres, integrator, mesh, spp):
    """The first launch of a block list has no durations to order or split by; trc_render runs it as a head of 8 samples (cold)
    and the rest ordered and planned by the head's durations.  A pixel's samples are one chain through its RNG texel, so the
    frame, the RNG texture, the ray count and the number of launches reported are those of one plain launch -- for a whole
    frame and for a rank's share, with a frame counter that does not start at 0; knob no_cold_probe gives the single launch.
    20 samples = 8 + 12; 72 = 8 + 16 + 48 and 136 = 8 + 16 + 32 + 80 (further passes of doubling length where four times their
    samples remain, each planned by its predecessor)."""
    W, H = (640, 360) if spp <= 72 else (320, 192)
    scene = host.HostScene(abi.SCENE_CORNELL_MESH, host.Mesh.golden("teapot")) if mesh else cornell_spheres
    for nranks, rank in ((1, 0), (4, 3)):
        gpu.upload_scene(scene.view); gpu.set_camera(host.prepare_camera(W, H)); gpu.set_environment((0.2, 0.3, 0.4)); gpu.resize(W, H)
        kw = dict(spp=spp, integrator=integrator, frame0=5, tile_rank=rank, tile_nranks=nranks)
        gpu.debug_set("no_cold_probe", 1)
        plain = _launch(gpu, 21, small_blocks=False, fixed_order=True, **kw)
        single = _launch(gpu, 21, **kw)
        gpu.debug_set("no_cold_probe", 0)               # forgets the block costs: the next launch is a first one again
        split = _launch(gpu, 21, **kw)
        assert gpu.stats().launches == 1 and gpu.stats().paths * nranks >= W * H * spp * 0.9
        again = _launch(gpu, 21, **kw)                   # a second launch: one pass, planned from the first
        assert _same(single, plain) and _same(split, plain) and _same(again, plain)


def test_the_plan_leaves_a_full_frame_alone(gpu, cornell_spheres):
    """many more blocks than wavefront slots: the work term of the plan's model wins -- nothing is split but the odd
    block whose measured duration sticks out of the ideal makespan (a stalled wavefront)"""
    W, H, spp = 1920, 1080, 8
    gpu.upload_scene(cornell_spheres.view); gpu.set_camera(host.prepare_camera(W, H)); gpu.resize(W, H)
    gpu.debug_set("no_split", 0)
    runs = [_launch(gpu, 3, spp=spp) for _ in range(3)]
    assert _same(runs[0], runs[1]) and _same(runs[0], runs[2])
    assert runs[0][3] == 0 and max(r[3] for r in runs) <= 32400 // 20, [r[3] for r in runs]      # a few per cent at most: the plan works from measured durations


def test_filtered_costs_change_the_schedule_not_the_pixels(gpu, cornell_spheres):
    """The order and the plan work on the shortest duration a block showed lately (k_filter_costs), because the SIMD serves
    its oldest wavefronts first and a block that started late measures up to twice its own work.  With the
    filter and without (knob no_cost_filter), over launches that re-plan from their predecessors: the same frame, RNG
    texture and ray count as the plain launch; the knob is reset for the tests that follow."""
    W, H, spp = 960, 544, 12                        # a share-sized list: 8 160 blocks for 5 120 wavefront slots
    gpu.upload_scene(cornell_spheres.view); gpu.set_camera(host.prepare_camera(W, H)); gpu.set_environment((0.2, 0.3, 0.4)); gpu.resize(W, H)
    gpu.debug_set("no_cost_filter", 0)
    plain = _launch(gpu, 11, spp=spp, small_blocks=False, fixed_order=True)
    try:
        for knob in (0, 1, 0):
            gpu.debug_set("no_cost_filter", knob)
            for _ in range(5):
                assert _same(_launch(gpu, 11, spp=spp), plain)
    finally:
        gpu.debug_set("no_cost_filter", 0)
    with pytest.raises(Exception):
        gpu.debug_set("no_such_knob", 1)


@pytest.mark.parametrize("policy", [0, 1, 2, 3])
def test_a_moving_camera_changes_the_schedule_not_the_pixels(gpu, cornell_spheres, policy):
    """trc_set_camera keeps the recorded block costs when the camera moved a little (knob camera_policy 0: the shipped rule; 1 always
    forgets, 2 / 3 always keep: tools/moving_camera.py) -- the next launch is then ordered and split by another view's costs.  Small
    steps, a jump, small steps again: every frame equals the plain launch of the same view, RNG texture and ray count included."""
    import math
    W, H, spp = 640, 360, 16
    gpu.upload_scene(cornell_spheres.view); gpu.set_environment((0.2, 0.3, 0.4)); gpu.resize(W, H)
    gpu.debug_set("camera_policy", policy)
    try:
        def view(deg):
            r = math.radians(deg)
            eye = (278 + 1078 * math.sin(r), 278, 278 - 1078 * math.cos(r))
            gpu.set_camera(host.make_camera(eye, (278, 278, 278), (0, 1, 0), 0.0, W / H, math.radians(45), 10.0))
        view(0.0)
        for k in range(3):
            _launch(gpu, 40 + k, spp=spp)                       # costs of the first view settle
        for k, deg in enumerate([0.5, 1.0, 1.5, 40.0, 40.5, 41.0]):      # drags, a jump, drags
            view(deg)
            got = _launch(gpu, 50 + k, spp=spp)
            view(deg)                                           # (the same camera again: nothing is forgotten, nothing changes)
            plain = _launch(gpu, 50 + k, spp=spp, small_blocks=False, fixed_order=True)
            assert _same(got, plain), (policy, deg)
    finally:
        gpu.debug_set("camera_policy", 0)
