"""The BASELINE configurations as device workloads (shared by the tools; the tests build the same scenes themselves).
The reference's mesh assets travel as tests/golden/meshes.npz (tests/golden/make_mesh_fixtures.py)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tracer_amd import abi, host  # noqa: E402

W, H = 1920, 1080


def make(config):
    """-> dict(scene, integrator, spp (as BASELINE names it), kernel (rocprof name prefix), density)"""
    density = None
    if config == "2":
        scene, integ, spp = host.HostScene(abi.SCENE_CORNELL_SPHERES), abi.INTEGRATOR_PATH, 64
        what = "config 2: Cornell + 12 spheres, tracePath"
    elif config == "3":
        scene, integ, spp = host.HostScene(abi.SCENE_CORNELL_MESH, host.Mesh.golden("coatball")), abi.INTEGRATOR_MIS, 256
        what = "config 3: Cornell + coatball.obj (46 816 triangles), traceMIS"
    elif config == "4":
        scene = host.HostScene(abi.SCENE_CORNELL_MESH, host.Mesh.golden("teapot").replicate(8, 80.0))
        integ, spp = abi.INTEGRATOR_PATH, 256
        what = "config 4: Cornell + teapot.obj x 64 (1 005 056 triangles), tracePath"
    elif config == "volume":
        scene = host.HostScene(abi.SCENE_CORNELL_VOLUME, host.Mesh.golden("coatball"))
        integ, spp, density = abi.INTEGRATOR_VOLUME, 64, host.make_cloud()
        what = "traceVolume: Cornell + cloud container (100x100x40 grid) + coatball.obj as glass with the homogeneous medium"
    else:
        raise SystemExit(f"unknown config {config}")
    lds = scene.view.n_bvh <= 255          # every Cornell-only scene fits LDS; mesh scenes read nodes through L2
    # production kernels: one-wavefront workgroups on an LDS-resident tree, persistent workgroups on a tree read from memory
    kernel = ("k_render_dense" if integ == abi.INTEGRATOR_PATH else f"k_render<true, false, {integ}") if lds else f"k_render_pwg<{integ}, false>"
    return {"scene": scene, "integrator": integ, "spp": spp, "kernel": kernel, "density": density, "what": what}


def setup(trc, wl):
    trc.upload_scene(wl["scene"].view)
    if wl["density"] is not None:
        trc.upload_density(host.density_info(wl["density"]), wl["density"])
    trc.set_camera(host.prepare_camera(W, H))
    trc.set_environment((0.0, 0.0, 0.0))
    trc.resize(W, H)


def algorithmic_bytes(st, n_pixels):
    """SURVEY.md 8(d) (same formula as bench.py)"""
    leaf = (20 * st.n_leaf_sphere + 32 * st.n_leaf_square + 100 * st.n_leaf_cube + 128 * st.n_hit_cube +
            48 * st.n_leaf_triangle + 60 * st.n_hit_triangle)
    return 88 * st.n_descend + 24 * st.n_return + leaf + 64 * st.shaded + 64 * n_pixels
