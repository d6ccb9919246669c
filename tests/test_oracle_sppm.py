"""SPPM pass on the CPU oracle (Photon.metal / Photon.hh): hash KATs and pipeline invariants."""
import ctypes as C

import numpy as np

from oracle import pyoracle as po
from tracer_amd import abi, host


def test_photon_hash_is_a_cell_index_and_deterministic():
    L = po.lib()
    f3 = C.c_float * 3
    seen = set()
    for ix in range(0, 40, 3):
        for iy in range(0, 40, 7):
            h = L.orc_photon_hash(f3(ix, iy, 11), 1.5)
            assert h == np.floor(h) and 0 <= h < 512 * 512          # Photon.hh:88: floor(fract(..) * N^2)
            assert h == L.orc_photon_hash(f3(ix, iy, 11), 1.5)
            seen.add(h)
    assert len(seen) > 60                                           # spreads cells over the grid


def test_pipeline_invariants():
    W, H = 64, 36
    scene = host.HostScene(abi.SCENE_CORNELL_SPHERES)
    cam = host.prepare_camera(W, H)
    rng = host.fill_rng(3, W, H)
    acc = np.zeros((H, W, 4), np.float32)
    s = po.Sppm(W, H, 9)
    s.frames(scene.view, cam, rng, acc, 1)
    c, p, mark, count, cx = s.download()
    assert cx.frame_count == 1 and cx.framePhotonSum == 0
    # kernelPhotonSumming (Photon.metal:458-496): the frame's photon sum = sum of the additive count grid
    assert cx.totalPhotonSum == count.sum()
    # only active photons are rasterised, at most one per ... each lands in exactly one cell
    assert count.sum() <= p["active"].sum()
    # marks: the winner is an ACTIVE photon and is the highest index that hashed into the cell
    cells = np.argwhere(mark[..., 0] >= 0)
    assert len(cells) == (count > 0).sum()
    for (y, x) in cells[:200]:
        winner = int(mark[y, x, 1]) * 512 + int(mark[y, x, 0])
        assert p["active"][winner] == 1 and (mark[y, x, 2], mark[y, x, 3]) == (x, y)
    # kernelPhotonParams (:357-372): radius = mean box extent * 2.5/4096, hash scale = 1/(1.5 r)
    size = np.array([cx.photonBoxSize.x, cx.photonBoxSize.y, cx.photonBoxSize.z], np.float32)
    r = np.float32(np.float32(size[0] * np.float32(1 / 3) + size[1] * np.float32(1 / 3)) + size[2] * np.float32(1 / 3)) * np.float32(2.5 / 4096)
    assert abs(cx.photonInitialRadius - r) <= 1e-6 * r
    assert abs(cx.photonHashScale - 1 / (1.5 * cx.photonInitialRadius)) < 1e-5 * cx.photonHashScale
    # progressive refinement only ever shrinks the radius (g <= 1), photon counts only grow
    s.frames(scene.view, cam, rng, acc, 2)
    c2 = s.download()[0]
    v = (c["valid"] == 1) & (c2["valid"] == 1)
    assert (c2["radius"][v] <= cx.photonInitialRadius).all() and np.isfinite(acc).all()
    # first frame: depth 3 camera chains; photons: step in 0..8
    assert p["step"].max() <= 8


def test_new_photons_start_on_the_ceiling_light():
    """kernelPhotonRecording (:316-334): `random() < 1` always picks squareList[5]; after one frame every photon
    either died (reset) or sits on a surface with step 1."""
    scene = host.HostScene(abi.SCENE_CORNELL)
    W, H = 32, 18
    s = po.Sppm(W, H, 1)
    s.frames(scene.view, host.prepare_camera(W, H), host.fill_rng(1, W, H), np.zeros((H, W, 4), np.float32), 1)
    p = s.download()[1]
    assert set(np.unique(p["step"])) <= {0, 1}
    alive = p["step"] == 1
    assert 0.3 < alive.mean() < 1.0
    assert (p["position"][alive][:, 1] < 554.95).all()        # below the light plane (y = 554.9 + offset)


def test_threaded_passes_equal_the_serial_loops(monkeypatch):
    """the oracle runs the per-pixel / per-photon passes on all host cores (so that 64 frames at 1080p can be checked on
    the GPU box); every item is independent, so any thread count gives the serial result, byte for byte"""
    scene = host.HostScene(abi.SCENE_CORNELL_SPHERES)
    W, H = 96, 54
    cam = host.prepare_camera(W, H)
    out = []
    for threads in ("1", "5"):
        monkeypatch.setenv("ORC_THREADS", threads)
        rng = host.fill_rng(3, W, H)
        acc = np.zeros((H, W, 4), np.float32)
        s = po.Sppm(W, H, 9)
        s.frames(scene.view, cam, rng, acc, 3)
        c, p, mark, count, cx = s.download()
        out.append((rng.tobytes(), acc.tobytes(), c.tobytes(), p.tobytes(), mark.tobytes(), count.tobytes(), cx.totalPhotonSum))
    assert out[0] == out[1]
