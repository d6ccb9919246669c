"""pytest configuration: the `gpu` marker + shared scene/oracle helpers.

-m "not gpu": oracle vs golden vectors / analytic KATs, host logic, ABI symbols (no GPU needed).
-m gpu:       parity of the HIP path against the CPU oracle, through the C ABI, on a real MI355X.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_built():
    need = [os.path.join(ROOT, "tracer_amd", "lib", "libtrc_host.so"),
            os.path.join(ROOT, "tracer_amd", "lib", "libtracer_amd.so"),
            os.path.join(ROOT, "tracer_amd", "lib", "libtracer_amd_fast.so"),
            os.path.join(ROOT, "tracer_amd", "lib", "libtracer_amd_hooks.so"),
            os.path.join(ROOT, "oracle", "liboracle.so"),
            os.path.join(ROOT, "oracle", "liboracle_libm.so")]
    if not all(os.path.exists(p) for p in need):
        subprocess.check_call(["make", "-C", ROOT, "all"], stdout=subprocess.DEVNULL)


_ensure_built()

from tracer_amd import abi, host  # noqa: E402
from tracer_amd.dtypes import make_rays  # noqa: E402


@pytest.fixture(scope="session")
def cornell():
    return host.HostScene(abi.SCENE_CORNELL)


@pytest.fixture(scope="session")
def cornell_spheres():
    return host.HostScene(abi.SCENE_CORNELL_SPHERES)


@pytest.fixture(scope="session")
def ball_mesh_scene():
    """Cornell box + a ~5k-triangle procedural ball (stand-in for the reference's OBJ assets)."""
    mesh = host.Mesh.ball(50, 50, 0.08)
    return host.HostScene(abi.SCENE_CORNELL_MESH, mesh)


def random_rays(n, seed, inside_only=False):
    """Rays from in and around the Cornell box in random directions."""
    rs = np.random.RandomState(seed)
    zlo = 5.0 if inside_only else -700.0
    o = np.stack([rs.uniform(-240, 795, n), rs.uniform(5, 550, n), rs.uniform(zlo, 550, n)], 1).astype(np.float32)
    d = rs.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return make_rays(o, d.astype(np.float32))


def camera_rays(cam, width, height, step=1):
    """Primary rays of castRay with zero aperture (no RNG dependence)."""
    ys, xs = np.mgrid[0:height:step, 0:width:step]
    u = (xs.astype(np.float32) / np.float32(width)).ravel()
    v = (ys.astype(np.float32) / np.float32(height)).ravel()
    f = lambda a: np.array([a.x, a.y, a.z], dtype=np.float32)
    sample = f(cam.cornerLowLeft)[None] + f(cam.horizontal)[None] * u[:, None] + f(cam.vertical)[None] * v[:, None]
    o = np.repeat(f(cam.lookFrom)[None], len(u), 0)
    return make_rays(o, (sample - o).astype(np.float32))


@pytest.fixture(scope="session")
def gpu():
    """A Tracer on cuda:0; fails (not skips) when the HIP path is unusable on a GPU run."""
    from tracer_amd import device
    t = device.Tracer(0)
    yield t
    t.close()


@pytest.fixture(scope="session")
def gpu_hooks():
    """A Tracer on libtracer_amd_hooks.so: the product's sources + the entry points of include/tracer_test_hooks.h"""
    from tracer_amd import device
    t = device.Tracer(0, hooks=True)
    yield t
    t.close()


@pytest.fixture(autouse=True)
def _gpu_default_state(request):
    """The Tracer is shared by the whole session: a test that leaves a sky environment behind must not change what a
    later test (in whatever order the files are named on the command line) renders."""
    if "gpu" in request.fixturenames:
        request.getfixturevalue("gpu").set_environment((0.0, 0.0, 0.0))
    yield
