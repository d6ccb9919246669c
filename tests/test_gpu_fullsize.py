"""BASELINE configs 2, 3, 4, 5 AS NAMED -- full frame size (1920x1080), full sample counts, the reference's own mesh
assets -- on the GPU, checked against the oracle: whole frames at low sample counts, a tile subsample at the full
count, and the whole 64-frame SPPM pass.  /root/reference does not exist on the GPU box; coatball.obj and teapot.obj
travel as vertex / index arrays (tests/golden/meshes.npz, tests/golden/make_mesh_fixtures.py)."""
import os

import numpy as np
import pytest

from oracle import pyoracle as po
from tracer_amd import abi, host

pytestmark = pytest.mark.gpu
W, H = 1920, 1080


def _tile_mask(nranks):
    ty, tx = np.mgrid[0:H, 0:W] // abi.TRC_TILE
    return ((tx + ty) % nranks) == 0


def _check_subsample(gpu, scene, integrator, spp, nranks, seed=0xC0FFEE, sobol=False, frame0=0):
    cam = host.prepare_camera(W, H)
    gpu.upload_scene(scene.view)
    gpu.set_camera(cam)
    gpu.set_environment((0.0, 0.0, 0.0))
    gpu.resize(W, H)
    gpu.seed(seed)
    gpu.reset_stats()
    gpu.render(spp=spp, integrator=integrator, sobol=sobol, frame0=frame0)
    dev = gpu.download_accum()
    st = gpu.stats()
    assert st.paths == W * H * spp and st.rays >= st.paths
    assert np.isfinite(dev).all() and (dev[..., 3] == 1.0).all() and (dev[..., :3] >= 0).all()
    ref, rst = po.render(scene.view, cam, W, H, host.fill_rng(seed, W, H), spp=spp, integrator=integrator,
                         tile_rank=0, tile_nranks=nranks, sobol=sobol, frame0=frame0)
    mine = _tile_mask(nranks)
    assert mine.sum() > 5000 and rst.rays > 0
    assert np.array_equal(dev[mine].view(np.uint32), ref[mine].view(np.uint32))
    return dev


def coatball_scene():
    """config 3: Cornell + RT_Metal/coatball/coatball.obj placed by AAPLRenderer.mm:513-525,562-572, material 19"""
    scene = host.HostScene(abi.SCENE_CORNELL_MESH, host.Mesh.golden("coatball"))
    assert scene.view.n_index // 3 == 46816
    return scene


def teapot_grid_scene():
    """config 4: RT_Metal/meshes/teapot.obj (15 704 triangles) on an 8 x 8 grid (spacing 80 object units: the pots interlock) = 1 005 056 triangles, 2.0 M BVH nodes
    (129 MB of reference nodes: far beyond the 32 MB of L2), placed by the same transform"""
    mesh = host.Mesh.golden("teapot").replicate(8, 80.0)
    assert mesh.n_triangles == 64 * 15704 >= 1_000_000
    return host.HostScene(abi.SCENE_CORNELL_MESH, mesh)


def test_scene_beyond_the_infinity_cache_4m_triangles(gpu):
    """The operating point where "HBM" means HBM: teapot.obj x 256 = 4 020 224 triangles, 8.0 M BVH records --
    ~0.7 GB of fat nodes and triangles against the 256 MiB Infinity Cache (profiles/r05/size_sweep.txt).  The host hands over
    its analytic leaves and the mesh, the tree is built on the device (the reference's SAH build, trc_upload_scene_device) and
    must be the host builder's record for record; 1920x1080 x 8 spp through it, 1 tile in 128 re-rendered by the oracle."""
    mesh = host.Mesh.golden("teapot").replicate(16, 80.0)
    assert mesh.n_triangles == 256 * 15704
    scene = host.HostScene(abi.SCENE_CORNELL_MESH, mesh)                      # host leaves + host tree: the oracle's input
    lean = host.HostScene(abi.SCENE_CORNELL_MESH, mesh, analytic_leaves_only=True)
    gpu.upload_scene_device(lean.view, abi.TREE_SAH | abi.TREE_TRIANGLE_LEAVES)
    got = np.frombuffer(bytes(memoryview(gpu.download_bvh())), dtype=np.uint32).reshape(-1, 16)
    want = scene.bvh_array()
    assert got.shape == want.shape == (2 * scene.n_leaves - 1, 16) and scene.n_leaves > 4_000_000
    assert np.array_equal(got, want)
    cam = host.prepare_camera(W, H)
    gpu.set_camera(cam); gpu.set_environment((0.0, 0.0, 0.0)); gpu.resize(W, H)
    for launch in range(2):                      # 8 spp: the persistent-workgroup kernel; the second launch in adaptive order
        gpu.seed(0xB16); gpu.clear_accum(); gpu.reset_stats(); gpu.render(spp=8)
    dev, st = gpu.download_accum(), gpu.stats()
    assert st.paths == W * H * 8 and np.isfinite(dev).all() and (dev[..., 3] == 1.0).all()
    ref, rst = po.render(scene.view, cam, W, H, host.fill_rng(0xB16, W, H), spp=8, tile_rank=0, tile_nranks=128)
    mine = _tile_mask(128)
    assert mine.sum() > 5000 and rst.rays > 0
    assert np.array_equal(dev[mine].view(np.uint32), ref[mine].view(np.uint32))


def test_config3_as_named_coatball_mis_256spp(gpu):
    """BASELINE config 3 as named: coatball.obj, traceMIS, 1920x1080 x 256 spp; 1 tile in 64 re-rendered by the oracle"""
    _check_subsample(gpu, coatball_scene(), abi.INTEGRATOR_MIS, 256, 64)


def test_headline_frame_with_the_sobol_sampler(gpu):
    """config 2's frame through TRC_FLAG_SOBOL: resolution 2048 (log2 11), pixel coordinates up to 1919 x 1079, frames 100..103;
    1 tile in 32 re-rendered by the oracle"""
    _check_subsample(gpu, host.HostScene(abi.SCENE_CORNELL_SPHERES), abi.INTEGRATOR_PATH, 4, 32, sobol=True, frame0=100)


@pytest.mark.parametrize("coalesce", [True, False])
def test_one_sample_per_launch_like_the_reference(gpu, coalesce):
    """the reference's own pattern: one sample per dispatch.  Launched one by one (knob no_coalesce) such launches run
    k_render_strip (a strip of blocks per wavefront) and reuse the launch order for a few launches; by default trc_render keeps a
    launch of few samples back and extends it with the calls that continue it, so the six calls become one launch of six
    samples at the first download.  Either way: six of them == six fused samples, and trc_stats counts six launches"""
    scene = host.HostScene(abi.SCENE_CORNELL_SPHERES)
    cam = host.prepare_camera(W, H)
    gpu.upload_scene(scene.view); gpu.set_camera(cam); gpu.set_environment((0.0, 0.0, 0.0)); gpu.resize(W, H)
    gpu.debug_set("no_coalesce", 0 if coalesce else 1)
    gpu.seed(0xABCD); gpu.clear_accum(); gpu.reset_stats()
    for f in range(6):
        gpu.render(spp=1, frame0=f)
    one_by_one, rng_a, st = gpu.download_accum(), gpu.download_rng(), gpu.stats()
    gpu.debug_set("no_coalesce", 0)
    assert st.paths == W * H * 6 and st.launches == 6
    gpu.seed(0xABCD); gpu.clear_accum()
    gpu.render(spp=6)
    assert np.array_equal(one_by_one.view(np.uint32), gpu.download_accum().view(np.uint32))
    assert np.array_equal(rng_a, gpu.download_rng())
    ref, _ = po.render(scene.view, cam, W, H, host.fill_rng(0xABCD, W, H), spp=6, tile_rank=0, tile_nranks=64)
    mine = _tile_mask(64)
    assert np.array_equal(one_by_one[mine].view(np.uint32), ref[mine].view(np.uint32))


def test_kept_launches_are_flushed_by_everything_that_could_see_them(gpu):
    """coalescing must be invisible: a run of 1-sample calls interrupted by a call that does not continue it (another frame
    counter, another depth, a tonemap, a camera, a seed, a clear) gives the frames of launching every call at once; 20 calls in a
    row are launched as 16 + 4"""
    Ws, Hs = 320, 200
    scene = host.HostScene(abi.SCENE_CORNELL_SPHERES)
    cam, cam2 = host.prepare_camera(Ws, Hs), host.make_camera((278, 278, -700), (278, 260, 0), (0, 1, 0), 0.0, Ws / Hs, 0.7, 10.0)

    def program(t):
        out = []
        t.seed(77); t.clear_accum()
        t.render(spp=1, frame0=0); t.render(spp=2, frame0=1); t.render(spp=1, frame0=3)          # continues: 4 samples
        t.render(spp=1, frame0=9)                                                                  # frame counter jumps: new launch
        t.render(spp=1, frame0=10, max_depth=3)                                                    # other depth: new launch
        out.append(t.tonemap()[0].copy())                                                          # flushes
        t.render(spp=3, frame0=11)
        t.set_camera(cam2)                                                                         # flushes BEFORE the camera changes
        t.render(spp=3, frame0=14)
        out.append(t.download_accum())
        t.set_camera(cam); t.seed(78); t.clear_accum(); t.reset_stats()
        for f in range(20):
            t.render(spp=1, frame0=f)
        out.append(t.download_accum()); out.append(t.download_rng())
        st = t.stats()
        assert st.launches == 20 and st.paths == Ws * Hs * 20
        return out

    gpu.upload_scene(scene.view); gpu.set_camera(cam); gpu.set_environment((0.1, 0.1, 0.1)); gpu.resize(Ws, Hs)
    gpu.debug_set("no_coalesce", 1)
    want = program(gpu)
    gpu.debug_set("no_coalesce", 0)
    got = program(gpu)
    for a, b in zip(want, got):
        assert np.array_equal(np.ascontiguousarray(a).view(np.uint8), np.ascontiguousarray(b).view(np.uint8))


def test_config4_as_named_million_triangles_256spp(gpu):
    """BASELINE config 4 as named on one GPU: teapot.obj replicated to >= 1 M triangles, tracePath, 1920x1080 x 256 spp;
    1 tile in 64 re-rendered by the oracle; hits on the mesh must exist among the primary rays"""
    scene = teapot_grid_scene()
    assert scene.tree_depth() <= abi.TRC_MAX_BVH_DEPTH
    _check_subsample(gpu, scene, abi.INTEGRATOR_PATH, 256, 64)
    # Scene::hit on the big tree: primary rays, bit-exact incl. traversal counters
    from conftest import camera_rays
    rays = camera_rays(host.prepare_camera(W, H), W, H, step=12)
    dev = gpu.trace_rays(rays)
    ref = po.trace_rays(scene.view, rays)
    assert (ref["pType"] == abi.PRIM_TRIANGLE).sum() > 100
    for f in ("hit", "pType", "pIndex", "n_descend", "n_return", "n_leaf"):
        assert np.array_equal(dev[f], ref[f]), f
    assert np.array_equal(dev["t"].view(np.uint32), ref["t"].view(np.uint32))


@pytest.mark.parametrize("n_frames", [2, 64])
def test_config5_sppm_full_frame(gpu, cornell_spheres, n_frames):
    """BASELINE config 5 as named on one GPU: SPPM at 1080p, 512^2 photons per frame, 64 frames (and 2), the whole pass --
    camera records, photon records, both hash grids, Complex, canvas RNG, refined frame -- against the oracle"""
    cam = host.prepare_camera(W, H)
    gpu.upload_scene(cornell_spheres.view)
    gpu.set_camera(cam)
    gpu.set_environment((0.0, 0.0, 0.0))
    gpu.resize(W, H)
    gpu.seed(8)
    gpu.sppm_init(9)
    gpu.sppm_frames(n_frames)
    dcam, dpho, dmark, dcount, dcx = gpu.sppm_download()
    dacc = gpu.download_accum()
    rng = host.fill_rng(8, W, H)
    acc = np.zeros((H, W, 4), np.float32)
    s = po.Sppm(W, H, 9)
    s.frames(cornell_spheres.view, cam, rng, acc, n_frames)
    ocam, opho, omark, ocount, ocx = s.download()
    assert dcx.frame_count == ocx.frame_count == n_frames
    assert dcx.totalPhotonSum == ocx.totalPhotonSum and dcx.photonInitialRadius == ocx.photonInitialRadius
    assert np.array_equal(dcount, ocount) and np.array_equal(dmark, omark)
    assert dpho.tobytes() == opho.tobytes()
    assert np.array_equal(dcam["radius"].view(np.uint32), ocam["radius"].view(np.uint32))
    assert np.array_equal(dcam["photonCount"], ocam["photonCount"])
    assert np.array_equal(dacc.view(np.uint32), acc.view(np.uint32))
    assert np.array_equal(gpu.download_rng(), rng)


def test_edge_sizes_and_empty_inputs(gpu, cornell):
    """1x1 frame, zero samples, an empty ray batch, and a frame larger than 4K (size-independent properties)"""
    from tracer_amd.dtypes import RAY_DTYPE
    gpu.upload_scene(cornell.view)
    assert len(gpu.trace_rays(np.zeros(0, dtype=RAY_DTYPE))) == 0
    gpu.set_camera(host.prepare_camera(1, 1))
    gpu.resize(1, 1)
    gpu.seed(1)
    gpu.render(spp=0)                                   # no-op
    assert not gpu.download_accum().any()
    gpu.render(spp=3)
    ref, _ = po.render(cornell.view, host.prepare_camera(1, 1), 1, 1, host.fill_rng(1, 1, 1), spp=3)
    assert np.array_equal(gpu.download_accum().view(np.uint32), ref.view(np.uint32))
    BW, BH = 4100, 2310                                 # ragged on both axes, 9.5 M pixels
    cam = host.prepare_camera(BW, BH)
    gpu.set_camera(cam)
    gpu.resize(BW, BH)
    gpu.seed(2)
    gpu.reset_stats()
    gpu.render(spp=1)
    big = gpu.download_accum()
    assert gpu.stats().paths == BW * BH and np.isfinite(big).all() and (big[..., 3] == 1).all()
    ref, _ = po.render(cornell.view, cam, BW, BH, host.fill_rng(2, BW, BH), spp=1, tile_rank=0, tile_nranks=97)
    ty, tx = np.mgrid[0:BH, 0:BW] // abi.TRC_TILE
    mine = ((tx + ty) % 97) == 0
    assert np.array_equal(big[mine].view(np.uint32), ref[mine].view(np.uint32))


def test_volume_full_frame_through_lbvh(gpu):
    """the two widened rows together at full size: traceVolume (cloud container + glass mesh with the homogeneous
    medium) through a tree built on the GPU; 1 tile in 128 re-rendered by the oracle through the same tree"""
    scene = host.HostScene(abi.SCENE_CORNELL_VOLUME, host.Mesh.ball(153, 153, 0.08))
    cloud = host.make_cloud()
    info = host.density_info(cloud)
    cam = host.prepare_camera(W, H)
    gpu.upload_scene_lbvh(scene.leaves_view())
    tree = gpu.download_bvh()
    gpu.upload_density(info, cloud)
    po.set_density(info, cloud)
    try:
        gpu.set_camera(cam); gpu.set_environment((0.0, 0.0, 0.0)); gpu.resize(W, H); gpu.seed(77); gpu.reset_stats()
        gpu.render(spp=4, integrator=abi.INTEGRATOR_VOLUME)
        dev = gpu.download_accum()
        assert np.isfinite(dev).all() and (dev[..., :3] >= 0).all()
        ref, rst = po.render(scene.view_with_bvh(tree), cam, W, H, host.fill_rng(77, W, H), spp=4,
                             integrator=abi.INTEGRATOR_VOLUME, tile_rank=0, tile_nranks=128)
        mine = _tile_mask(128)
        assert rst.rays > 0 and np.array_equal(dev[mine].view(np.uint32), ref[mine].view(np.uint32))
    finally:
        po.set_density(None, None)
        gpu.upload_density(None, None)


def test_config2_whole_headline_frame_bit_exact(gpu, cornell_spheres):
    """the exact workload bench.py times -- Cornell + spheres, 1920x1080, 64 spp, tracePath, seed 0x5EED0000 -- every
    pixel, the RNG texture and the ray count against the oracle (about 10 s of CPU on the GPU box's host cores)"""
    cam = host.prepare_camera(W, H)
    gpu.upload_scene(cornell_spheres.view); gpu.set_camera(cam); gpu.set_environment((0.0, 0.0, 0.0)); gpu.resize(W, H)
    for launch in range(2):                      # the second launch runs in adaptive (expensive-first) order
        gpu.seed(0x5EED0000); gpu.clear_accum(); gpu.reset_stats()
        gpu.render(spp=64)
    dev, dev_rng, st = gpu.download_accum(), gpu.download_rng(), gpu.stats()
    rng = host.fill_rng(0x5EED0000, W, H)
    ref, rst = po.render(cornell_spheres.view, cam, W, H, rng, spp=64)
    assert st.rays == rst.rays == 219978393 and st.paths == W * H * 64
    assert np.array_equal(dev.view(np.uint32), ref.view(np.uint32))
    assert np.array_equal(dev_rng, rng)
    # ... and this is the frame bench.py holds its timed steps to: the committed golden is the oracle's, here and now
    import json, zlib
    from conftest import ROOT
    with open(os.path.join(ROOT, "tests", "golden", "bench_goldens.json")) as f:
        gold = json.load(f)["config2"]
    assert gold["rays"] == rst.rays and gold["crc_accum"] == (zlib.crc32(np.ascontiguousarray(ref).view(np.uint8).tobytes()) & 0xFFFFFFFF)
    assert gold["crc_rng"] == (zlib.crc32(np.ascontiguousarray(rng).view(np.uint8).tobytes()) & 0xFFFFFFFF)


@pytest.mark.parametrize("which", ["config3_mis", "config4_path_1m", "volume_lbvh"])
def test_whole_frames_of_the_other_configurations(gpu, which):
    """every pixel of a 1920x1080 frame (few samples, so the oracle finishes in seconds) for config 3 (coatball.obj,
    traceMIS), config 4 (teapot.obj x 64, tracePath) and the participating-media scene"""
    density = None
    if which == "config3_mis":
        scene, integ, spp = coatball_scene(), abi.INTEGRATOR_MIS, 4
    elif which == "config4_path_1m":
        scene, integ, spp = teapot_grid_scene(), abi.INTEGRATOR_PATH, 2
    else:
        scene, integ, spp = host.HostScene(abi.SCENE_CORNELL_VOLUME, host.Mesh.ball(60, 60, 0.08)), abi.INTEGRATOR_VOLUME, 2
        density = host.make_cloud()
    cam = host.prepare_camera(W, H)
    view = scene.view
    if which == "volume_lbvh":
        gpu.upload_scene_lbvh(scene.leaves_view())
        tree = gpu.download_bvh()
        view = scene.view_with_bvh(tree)
        info = host.density_info(density)
        gpu.upload_density(info, density); po.set_density(info, density)
    else:
        gpu.upload_scene(scene.view)
    try:
        gpu.set_camera(cam); gpu.set_environment((0.0, 0.0, 0.0)); gpu.resize(W, H)
        for launch in range(2):
            gpu.seed(4242); gpu.clear_accum(); gpu.reset_stats(); gpu.render(spp=spp, integrator=integ)
        dev, dev_rng, st = gpu.download_accum(), gpu.download_rng(), gpu.stats()
        rng = host.fill_rng(4242, W, H)
        ref, rst = po.render(view, cam, W, H, rng, spp=spp, integrator=integ)
        assert st.rays == rst.rays and st.paths == W * H * spp
        assert np.array_equal(dev.view(np.uint32), ref.view(np.uint32)) and np.array_equal(dev_rng, rng)
    finally:
        po.set_density(None, None)
        gpu.upload_density(None, None)
