"""Two contexts in one process (include/tracer_abi.h, "Threading contract"): each owns its stream, buffers, launch plans and
its grant of > 64 KB dynamic LDS for the persistent-workgroup kernels (per context = per device: hipFuncSetAttribute
applies to the current device only; a process-wide flag made every mesh launch on a second GPU fail).  Mesh scenes --
trees read from memory, >= 8 spp, i.e. k_render_pwg with 80-160 KB of LDS -- rendered interleaved on two contexts and from
two host threads at once (the reference's completion handlers arrive off the main thread, AAPLRenderer.mm:1148-1150) must
equal what one context renders alone, bit for bit.  On a multi-GPU node the second context goes to device 1."""
import threading

import numpy as np
import pytest

from tracer_amd import abi, device, host

pytestmark = pytest.mark.gpu
W, H = 256, 160


def _second_device():
    try:
        t = device.Tracer(1)
        t.close()
        return 1
    except device.TracerError:
        return 0


def _scenes():
    a = host.HostScene(abi.SCENE_CORNELL_MESH, host.Mesh.ball(60, 60, 0.08))
    b = host.HostScene(abi.SCENE_CORNELL_MESH, host.Mesh.golden("teapot"))
    return a, b


def _setup(t, scene):
    t.upload_scene(scene.view); t.set_camera(host.prepare_camera(W, H)); t.set_environment((0.1, 0.1, 0.1)); t.resize(W, H)


def _frames(t, seeds, integrator):
    out = []
    for s in seeds:
        t.clear_accum(); t.seed(s); t.render(spp=9, integrator=integrator)
        out.append(t.download_accum())
    return out


def test_two_contexts_interleaved_and_from_two_threads(gpu):
    sa, sb = _scenes()
    seeds = [11, 12, 13]
    # what one context renders alone (the session's context)
    _setup(gpu, sa); want_a = _frames(gpu, seeds, abi.INTEGRATOR_PATH)
    _setup(gpu, sb); want_b = _frames(gpu, seeds, abi.INTEGRATOR_MIS)

    ta, tb = device.Tracer(0), device.Tracer(_second_device())
    try:
        _setup(ta, sa); _setup(tb, sb)
        # interleaved: launches of the two contexts alternate, downloads afterwards
        for i, s in enumerate(seeds):
            ta.clear_accum(); ta.seed(s); tb.clear_accum(); tb.seed(s)
            ta.render(spp=9, integrator=abi.INTEGRATOR_PATH)
            tb.render(spp=9, integrator=abi.INTEGRATOR_MIS)
            fb = tb.download_accum(); fa = ta.download_accum()
            assert np.array_equal(fa.view(np.uint32), want_a[i].view(np.uint32)), ("interleaved a", i)
            assert np.array_equal(fb.view(np.uint32), want_b[i].view(np.uint32)), ("interleaved b", i)
        # two host threads, one context each, at the same time, several rounds
        got, err = {}, []

        def work(name, t, integrator):
            try:
                got[name] = [_frames(t, seeds, integrator) for _ in range(3)]
            except Exception as e:      # surfaced below: an exception in a thread does not fail the test by itself
                err.append((name, repr(e)))

        th = [threading.Thread(target=work, args=("a", ta, abi.INTEGRATOR_PATH)),
              threading.Thread(target=work, args=("b", tb, abi.INTEGRATOR_MIS))]
        for x in th:
            x.start()
        for x in th:
            x.join()
        assert not err, err
        for rnd in range(3):
            for i in range(len(seeds)):
                assert np.array_equal(got["a"][rnd][i].view(np.uint32), want_a[i].view(np.uint32)), ("thread a", rnd, i)
                assert np.array_equal(got["b"][rnd][i].view(np.uint32), want_b[i].view(np.uint32)), ("thread b", rnd, i)
    finally:
        ta.close(); tb.close()


def test_debug_knobs_are_per_context_and_change_no_pixel(gpu):
    sa, _ = _scenes()
    _setup(gpu, sa); want = _frames(gpu, [21], abi.INTEGRATOR_PATH)[0]
    t = device.Tracer(0)
    try:
        _setup(t, sa)
        for knob, value in [("stack_lds_levels", 1), ("no_pwg", 1), ("no_lds_fit", 1), ("strip_len", 2)]:
            t.debug_set(knob, value)
            got = _frames(t, [21], abi.INTEGRATOR_PATH)[0]
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), knob
            t.debug_set(knob, 0)
        with pytest.raises(device.TracerError):
            t.debug_set("no_such_knob", 1)
        # the session context never saw those knobs
        assert np.array_equal(_frames(gpu, [21], abi.INTEGRATOR_PATH)[0].view(np.uint32), want.view(np.uint32))
    finally:
        t.close()
