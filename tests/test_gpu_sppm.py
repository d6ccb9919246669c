"""GPU parity of the SPPM pass (Photon.metal, BASELINE config 5) against the CPU oracle: camera records,
photon records, the mark/count hash grids, Complex, the canvas RNG texture and the refined frame -- bit for bit.
The reference's point-raster "last writer wins" is deterministic here (highest photon index = last primitive
in API order), so no statistical tolerance is needed."""
import numpy as np
import pytest

from oracle import pyoracle as po
from tracer_amd import abi, host

pytestmark = pytest.mark.gpu

CAM_FIELDS = ["ratio", "position", "direction", "valid", "alternative", "flux", "radius", "photonCount"]
PHO_FIELDS = ["flux", "normal", "position", "direction", "step", "active"]


def bits_equal(a, b):
    if a.dtype == np.float32:
        return np.array_equal(a.view(np.uint32), b.view(np.uint32))
    return np.array_equal(a, b)


def run_both(gpu, scene, W, H, n_frames, canvas_seed=5, photon_seed=77):
    cam = host.prepare_camera(W, H)
    gpu.upload_scene(scene.view)
    gpu.set_camera(cam)
    gpu.set_environment((0.0, 0.0, 0.0))
    gpu.resize(W, H)
    gpu.seed(canvas_seed)
    gpu.sppm_init(photon_seed)
    gpu.sppm_frames(n_frames)
    dev = gpu.sppm_download()
    dev_acc, dev_rng = gpu.download_accum(), gpu.download_rng()
    rng = host.fill_rng(canvas_seed, W, H)
    acc = np.zeros((H, W, 4), np.float32)
    s = po.Sppm(W, H, photon_seed)
    s.frames(scene.view, cam, rng, acc, n_frames)
    return dev, dev_acc, dev_rng, s.download(), acc, rng


@pytest.mark.parametrize("n_frames", [1, 4])
def test_sppm_bit_exact(gpu, cornell_spheres, n_frames):
    (dcam, dpho, dmark, dcount, dcx), dacc, drng, (ocam, opho, omark, ocount, ocx), oacc, orng = run_both(
        gpu, cornell_spheres, 96, 54, n_frames)
    assert ocam["valid"].mean() > 0.3 and opho["active"].mean() > 0.1 and (ocount > 0).sum() > 1000
    for f in ["frame_count", "photonInitialRadius", "photonHashScale", "totalPhotonSum", "framePhotonSum"]:
        assert getattr(dcx, f) == getattr(ocx, f), f
    for ax in "xyz":
        assert getattr(dcx.photonBox.mini, ax) == getattr(ocx.photonBox.mini, ax)
        assert getattr(dcx.photonBox.maxi, ax) == getattr(ocx.photonBox.maxi, ax)
    for f in PHO_FIELDS:
        assert bits_equal(dpho[f], opho[f]), f"photon {f}"
    assert np.array_equal(dcount, ocount) and np.array_equal(dmark, omark)
    for f in CAM_FIELDS:
        assert bits_equal(dcam[f], ocam[f]), f"camera record {f}"
    assert np.array_equal(drng, orng)
    assert np.array_equal(dacc.view(np.uint32), oacc.view(np.uint32))
    assert oacc[..., :3].max() > 0


def test_a_frame_without_any_visible_point(gpu, cornell_spheres):
    """The camera looks AWAY from the box: no pixel records a visible point, every pixel contributes {FLT_MAX, -FLT_MAX} to the bound
    (Photon.metal:157-161), and kernelPhotonParams turns that into box size -inf, radius -inf, hash scale -0 (:357-372).  Found by
    tests/campaigns/fuzz_sppm.py in round 5 (cameras inside objects): the device started its min / max keys at the ends of the key range and
    decoded NaN.  Everything, the Complex block included, equals the oracle."""
    import struct
    W, H = 64, 40
    cam = host.make_camera((278, 278, -800), (278, 278, -2000), (0, 1, 0), 0.0, W / H, np.radians(40.0), 10.0)
    gpu.upload_scene(cornell_spheres.view); gpu.set_camera(cam); gpu.set_environment((0.0, 0.0, 0.0)); gpu.resize(W, H)
    gpu.seed(3); gpu.sppm_init(9); gpu.sppm_frames(3)
    dcam, dpho, dmark, dcount, dcx = gpu.sppm_download()
    dacc, drng = gpu.download_accum(), gpu.download_rng()
    rng = host.fill_rng(3, W, H); acc = np.zeros((H, W, 4), np.float32)
    s = po.Sppm(W, H, 9); s.frames(cornell_spheres.view, cam, rng, acc, 3)
    ocam, opho, omark, ocount, ocx = s.download()
    assert not ocam["valid"].any()
    bits = lambda x: struct.unpack("<I", struct.pack("<f", x))[0]
    assert ocx.photonInitialRadius == -np.inf and bits(ocx.photonHashScale) == 0x80000000
    for f in ["frame_count", "photonInitialRadius", "photonHashScale", "totalPhotonSum", "framePhotonSum"]:
        assert bits(float(getattr(dcx, f))) == bits(float(getattr(ocx, f))), f
    for ax in "xyz":
        assert getattr(dcx.photonBox.mini, ax) == getattr(ocx.photonBox.mini, ax) == np.inf
        assert getattr(dcx.photonBox.maxi, ax) == getattr(ocx.photonBox.maxi, ax) == -np.inf
    for f in PHO_FIELDS:
        assert bits_equal(dpho[f], opho[f]), f"photon {f}"
    for f in CAM_FIELDS:
        assert bits_equal(dcam[f], ocam[f]), f"camera record {f}"
    assert np.array_equal(dcount, ocount) and np.array_equal(dmark, omark) and np.array_equal(drng, rng)
    assert np.array_equal(dacc.view(np.uint32), acc.view(np.uint32))


def test_sppm_mesh_scene(gpu, ball_mesh_scene):
    (dcam, dpho, dmark, dcount, dcx), dacc, drng, (ocam, opho, omark, ocount, ocx), oacc, orng = run_both(
        gpu, ball_mesh_scene, 64, 40, 2)
    assert np.array_equal(dcount, ocount) and np.array_equal(dmark, omark)
    for f in PHO_FIELDS:
        assert bits_equal(dpho[f], opho[f]), f
    assert np.array_equal(dacc.view(np.uint32), oacc.view(np.uint32))


def test_sppm_requires_init(gpu, cornell_spheres):
    from tracer_amd.device import TracerError
    gpu.resize(32, 32)               # releases any SPPM state
    with pytest.raises(TracerError):
        gpu.sppm_frames(1)


def test_sppm_through_a_one_rank_communicator(gpu, cornell_spheres):
    """With a communicator the pass takes the multi-GPU route (tile-sharded camera/refine, photon range,
    ncclAllReduce(min/max) of the bound keys, ncclAllGather of the photon records).  One rank must reproduce the
    plain single-GPU result bit for bit."""
    from tracer_amd.device import group_unique_id
    W, H, n_frames = 80, 48, 3
    cam = host.prepare_camera(W, H)
    outs = []
    for grouped in (False, True):
        gpu.upload_scene(cornell_spheres.view)
        gpu.set_camera(cam)
        gpu.set_environment((0.0, 0.0, 0.0))
        gpu.resize(W, H)
        gpu.seed(11)
        if grouped:
            gpu.group_init(group_unique_id(), 1, 0)
        gpu.sppm_init(21)
        gpu.sppm_frames(n_frames)
        dcam, dpho, dmark, dcount, dcx = gpu.sppm_download()
        outs.append((gpu.download_accum(), dpho.copy(), dmark.copy(), dcount.copy(), dcx.totalPhotonSum))
        if grouped:
            gpu.group_finalize()
    a, b = outs
    assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))
    assert a[1].tobytes() == b[1].tobytes() and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]) and a[4] == b[4]


@pytest.mark.parametrize("scale", [0.37, 1.0, 1.5, 2.0, 3.999, 17.3, 250.0, 1.0e-3])
def test_hash_equals_the_oracle_cell_for_cell(gpu_hooks, scale):
    """The device evaluates the divisions of `hash()` (Photon.hh:71-89) with one refined reciprocal per shared divisor and
    the compiler's two fused corrections per quotient (trc_sppm.hip::DivBy).  That must be the IEEE quotient, i.e. the
    oracle's `/`: compared here directly, cell for cell, over the grid's own index range, large indices, non-integers,
    negative fourth components (x + y - z) and zeros, at hash scales from 1e-3 to 250."""
    import ctypes as C
    rs = np.random.RandomState(int(scale * 1000) % 65521)
    n = 40000
    cells = np.concatenate([
        rs.randint(0, 600, size=(n, 3)).astype(np.float32),                       # what the passes feed it
        rs.randint(0, 1 << 24, size=(n // 4, 3)).astype(np.float32),              # up to 2^24
        (rs.uniform(0, 600, size=(n // 4, 3))).astype(np.float32),                # non-integer
        np.zeros((4, 3), np.float32),
        np.array([[0, 0, 599], [599, 0, 0], [1, 1, 2], [0, 0, 1]], np.float32),   # x + y - z <= 0
    ])
    dev = gpu_hooks.sppm_hash_cells(cells, scale)
    L = po.lib()
    f3 = C.c_float * 3
    ref = np.array([L.orc_photon_hash(f3(*c), C.c_float(scale)) for c in cells], np.float32)
    bad = np.flatnonzero(dev.view(np.uint32) != ref.view(np.uint32))
    assert len(bad) == 0, f"{len(bad)} of {len(cells)} hashes differ; first: cell {cells[bad[0]]} dev {dev[bad[0]]} ref {ref[bad[0]]}"
