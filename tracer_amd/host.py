"""Host side (CPU): scene assembly + SAH BVH builder, through libtrc_host.so.

Mirrors the reference's host half -- prepareCubeList / prepareCornellBox / prepareSphereList /
prepareCamera (RT_Metal/Tracer/Tracer.hh:43-47) and BVH::buildNode / buildTree
(RT_Metal/Metal/BVH.hh:246-314) -- behind the C ABI of include/tracer_abi.h.
"""
import ctypes as C
import os

import numpy as np

from . import abi

_LIB = None


def lib_path():
    """The in-tree build; TRC_HOST_LIB points tools/run_sanitizers.sh at the sanitized build of the same sources."""
    return os.environ.get("TRC_HOST_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libtrc_host.so")


def lib():
    global _LIB
    if _LIB is None:
        path = lib_path()
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `make host` (or __graft_entry__.build())")
        L = C.CDLL(path)
        L.trc_host_build_node.argtypes = [C.POINTER(abi.AABB), C.POINTER(abi.float4x4), C.c_int32, C.c_uint32,
                                          C.POINTER(abi.BVH)]
        L.trc_host_build_node.restype = None
        L.trc_host_build_tree.argtypes = [C.POINTER(abi.BVH), C.c_uint32, C.POINTER(C.c_uint32)]
        L.trc_host_build_tree.restype = C.c_int32
        L.trc_host_tree_depth.argtypes = [C.POINTER(abi.BVH), C.c_uint32, C.POINTER(C.c_uint32)]
        L.trc_host_tree_depth.restype = C.c_int32
        L.trc_host_make_camera.argtypes = [C.POINTER(abi.Camera), C.POINTER(C.c_float), C.POINTER(C.c_float),
                                           C.POINTER(C.c_float), C.c_float, C.c_float, C.c_float, C.c_float]
        L.trc_host_make_camera.restype = None
        L.trc_host_prepare_camera.argtypes = [C.POINTER(abi.Camera), C.c_float, C.c_float]
        L.trc_host_prepare_camera.restype = None
        L.trc_host_fill_rng.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p]
        L.trc_host_fill_rng.restype = None
        L.trc_host_scene_create.argtypes = [C.c_int32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                            C.POINTER(C.c_void_p)]
        L.trc_host_scene_create.restype = C.c_int32
        L.trc_host_scene_create_leaves.argtypes = [C.c_int32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int32, C.POINTER(C.c_void_p)]
        L.trc_host_scene_create_leaves.restype = C.c_int32
        L.trc_host_scene_load_pbrt.argtypes = [C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(abi.Camera),
                                               C.POINTER(abi.PbrtInfo), C.POINTER(abi.PbrtShape), C.c_uint32]
        L.trc_host_scene_load_pbrt.restype = C.c_int32
        L.trc_host_scene_destroy.argtypes = [C.c_void_p]
        L.trc_host_scene_destroy.restype = None
        L.trc_host_scene_view.argtypes = [C.c_void_p, C.POINTER(abi.Scene)]
        L.trc_host_scene_view.restype = None
        L.trc_host_mesh_load_obj.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        L.trc_host_mesh_load_obj.restype = C.c_int32
        L.trc_host_mesh_load_ply.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        L.trc_host_mesh_load_ply.restype = C.c_int32
        L.trc_host_load_hdr.argtypes = [C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.POINTER(C.c_float))]
        L.trc_host_load_hdr.restype = C.c_int32
        L.trc_host_mesh_make_ball.argtypes = [C.c_uint32, C.c_uint32, C.c_float, C.POINTER(C.c_void_p)]
        L.trc_host_mesh_make_ball.restype = C.c_int32
        L.trc_host_mesh_from_arrays.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]
        L.trc_host_mesh_from_arrays.restype = C.c_int32
        L.trc_host_mesh_replicate.argtypes = [C.c_void_p, C.c_uint32, C.c_float, C.POINTER(C.c_void_p)]
        L.trc_host_mesh_replicate.restype = C.c_int32
        L.trc_host_mesh_view.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint32),
                                         C.POINTER(C.c_void_p), C.POINTER(C.c_uint32)]
        L.trc_host_mesh_view.restype = None
        L.trc_host_mesh_destroy.argtypes = [C.c_void_p]
        L.trc_host_make_density_info.argtypes = [C.c_float, C.c_float, C.c_float, C.c_uint32, C.c_uint32, C.c_uint32,
                                                 C.c_void_p, C.POINTER(abi.GridDensityInfo)]
        L.trc_host_make_density_info.restype = None
        L.trc_host_make_cloud.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        L.trc_host_make_cloud.restype = None
        L.trc_host_load_density_pbrt.argtypes = [C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                                 C.POINTER(C.c_uint32), C.POINTER(C.POINTER(C.c_float))]
        L.trc_host_free.argtypes = [C.c_void_p]
        L.trc_host_write_png.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32]
        L.trc_host_write_png.restype = C.c_int32
        L.trc_host_sobol_matrices32.argtypes = [C.c_void_p]
        L.trc_host_sobol_matrices32.restype = None
        L.trc_host_sobol_interval_tables.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p]
        L.trc_host_sobol_interval_tables.restype = C.c_int32
        L.trc_host_free.restype = None
        L.trc_host_mesh_destroy.restype = None
        _LIB = L
    return _LIB


def _check(status, what):
    if status != abi.OK:
        raise RuntimeError(f"{what} failed with trc_status {status}")


class Mesh:
    """Triangle mesh in object space: 32-byte vertices + u32 triangle indices."""

    def __init__(self, handle):
        self._h = handle
        vp, ip = C.c_void_p(), C.c_void_p()
        nv, ni = C.c_uint32(), C.c_uint32()
        lib().trc_host_mesh_view(self._h, C.byref(vp), C.byref(nv), C.byref(ip), C.byref(ni))
        self.vertices_ptr, self.n_vertices = vp.value, nv.value
        self.indices_ptr, self.n_indices = ip.value, ni.value

    @classmethod
    def load_pbrt(cls, path):
        """every `Shape "trianglemesh"` of a pbrt-v3 file, transformed to world space (trc_host_mesh_load_pbrt)"""
        h = C.c_void_p()
        _check(lib().trc_host_mesh_load_pbrt(os.fsencode(path), C.byref(h)), f"trc_host_mesh_load_pbrt({path})")
        return cls(h)

    @classmethod
    def load_ply(cls, path):
        """the triangles of a PLY file (trc_host_mesh_load_ply): what a pbrt-v3 `Shape "plymesh"` refers to"""
        h = C.c_void_p()
        _check(lib().trc_host_mesh_load_ply(os.fsencode(path), C.byref(h)), f"trc_host_mesh_load_ply({path})")
        return cls(h)

    @classmethod
    def load_obj(cls, path):
        h = C.c_void_p()
        _check(lib().trc_host_mesh_load_obj(os.fsencode(path), C.byref(h)), f"trc_host_mesh_load_obj({path})")
        return cls(h)

    @classmethod
    def from_arrays(cls, vertices, indices):
        """vertices: (n, 8) float32 rows of {position, normal, uv} (trc_TriangleVertex); indices: uint32, 3 per triangle."""
        v = np.ascontiguousarray(vertices, dtype=np.float32)
        i = np.ascontiguousarray(indices, dtype=np.uint32).ravel()
        assert v.ndim == 2 and v.shape[1] == 8
        h = C.c_void_p()
        _check(lib().trc_host_mesh_from_arrays(v.ctypes.data, v.shape[0], i.ctypes.data, i.size, C.byref(h)),
               "trc_host_mesh_from_arrays")
        return cls(h)

    @classmethod
    def golden(cls, name):
        """The reference's own OBJ assets as committed vertex / index arrays (tests/golden/meshes.npz, written by
        tests/golden/make_mesh_fixtures.py from RT_Metal/coatball/coatball.obj and RT_Metal/meshes/teapot.obj)."""
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "meshes.npz")
        z = np.load(path)
        return cls.from_arrays(z[name + "_vertices"], z[name + "_indices"])

    @classmethod
    def ball(cls, n_lat, n_lon, bump=0.05):
        h = C.c_void_p()
        _check(lib().trc_host_mesh_make_ball(n_lat, n_lon, bump, C.byref(h)), "trc_host_mesh_make_ball")
        return cls(h)

    def replicate(self, k, spacing):
        h = C.c_void_p()
        _check(lib().trc_host_mesh_replicate(self._h, k, spacing, C.byref(h)), "trc_host_mesh_replicate")
        return Mesh(h)

    @property
    def n_triangles(self):
        return self.n_indices // 3

    def vertices(self):
        a = (C.c_float * (8 * self.n_vertices)).from_address(self.vertices_ptr)
        return np.frombuffer(a, dtype=np.float32).reshape(-1, 8)

    def indices(self):
        a = (C.c_uint32 * self.n_indices).from_address(self.indices_ptr)
        return np.frombuffer(a, dtype=np.uint32)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().trc_host_mesh_destroy(self._h)
            self._h = None


class HostScene:
    """Scene arrays in the reference's layouts + the built BVH (root at index 0)."""

    def __init__(self, kind=abi.SCENE_CORNELL_SPHERES, mesh=None, analytic_leaves_only=False):
        """analytic_leaves_only: no per-triangle leaves and no tree -- view.bvhList holds the analytic primitives' leaf records,
        the input of Tracer.upload_scene_device(view, TREE_SAH | TREE_TRIANGLE_LEAVES)."""
        self._h = C.c_void_p()
        self._mesh = mesh
        L = lib()
        if mesh is not None:
            st = L.trc_host_scene_create_leaves(kind, mesh.vertices_ptr, mesh.n_vertices, mesh.indices_ptr,
                                                mesh.n_indices, int(analytic_leaves_only), C.byref(self._h))
        else:
            st = L.trc_host_scene_create_leaves(kind, None, 0, None, 0, int(analytic_leaves_only), C.byref(self._h))
        _check(st, "trc_host_scene_create")
        self.view = abi.Scene()
        lib().trc_host_scene_view(self._h, C.byref(self.view))

    @classmethod
    def from_pbrt(cls, path, max_shapes=4096):
        """A whole scene from a pbrt-v3 file (trc_host_scene_load_pbrt) -> (scene, camera, info, [shape descriptions])."""
        self = cls.__new__(cls)
        self._h, self._mesh = C.c_void_p(), None
        cam, info = abi.Camera(), abi.PbrtInfo()
        shapes = (abi.PbrtShape * max_shapes)()
        _check(lib().trc_host_scene_load_pbrt(os.fsencode(path), C.byref(self._h), C.byref(cam), C.byref(info), shapes,
                                              max_shapes), f"trc_host_scene_load_pbrt({path})")
        self.view = abi.Scene()
        lib().trc_host_scene_view(self._h, C.byref(self.view))
        return self, cam, info, [shapes[i] for i in range(min(info.n_shapes, max_shapes))]

    @property
    def n_leaves(self):
        return (self.view.n_bvh + 1) // 2

    def bvh(self):
        return [self.view.bvhList[i] for i in range(self.view.n_bvh)]

    def bvh_array(self):
        """(n_bvh, 16) uint32 view of the node array (columns: parent,left,right,axis,pType,pIndex,...)"""
        a = (C.c_uint32 * (16 * self.view.n_bvh)).from_address(C.addressof(self.view.bvhList.contents))
        return np.frombuffer(a, dtype=np.uint32).reshape(-1, 16)

    def leaves(self):
        """Pointer to the n_leaves leaf records (BVH::buildTree keeps them at [1, n], BVH.hh:246-269)."""
        return C.cast(C.addressof(self.view.bvhList.contents) + C.sizeof(abi.BVH), C.POINTER(abi.BVH))

    def leaves_view(self):
        """Copy of `view` whose bvhList holds only the leaf records: the input of Tracer.upload_scene_lbvh."""
        v = abi.Scene.from_buffer_copy(self.view)
        v.bvhList = self.leaves()
        v.n_bvh = self.n_leaves
        return v

    def view_with_bvh(self, nodes):
        """Copy of `view` over another node array (ctypes array of abi.BVH); the caller keeps `nodes` alive."""
        v = abi.Scene.from_buffer_copy(self.view)
        v.bvhList = C.cast(nodes, C.POINTER(abi.BVH))
        v.n_bvh = len(nodes)
        return v

    def tree_depth(self):
        d = C.c_uint32()
        _check(lib().trc_host_tree_depth(self.view.bvhList, self.view.n_bvh, C.byref(d)), "trc_host_tree_depth")
        return d.value

    def __del__(self):
        if getattr(self, "_h", None):
            lib().trc_host_scene_destroy(self._h)
            self._h = None


def make_cloud(nx=100, ny=100, nz=40, seed=1):
    """Procedural density grid (nz, ny, nx) float32 standing in for the reference's cloud/density_render.70.pbrt."""
    out = np.empty((nz, ny, nx), dtype=np.float32)
    lib().trc_host_make_cloud(nx, ny, nz, seed, out.ctypes.data)
    return out


def load_density_pbrt(path):
    """The `float density` grid of a pbrt-v3 MakeNamedMedium block -> (nz, ny, nx) float32."""
    nx, ny, nz = C.c_uint32(), C.c_uint32(), C.c_uint32()
    p = C.POINTER(C.c_float)()
    _check(lib().trc_host_load_density_pbrt(path.encode(), C.byref(nx), C.byref(ny), C.byref(nz), C.byref(p)),
           "trc_host_load_density_pbrt")
    try:
        n = nx.value * ny.value * nz.value
        return np.ctypeslib.as_array(p, shape=(n,)).astype(np.float32).reshape(nz.value, ny.value, nx.value).copy()
    finally:
        lib().trc_host_free(p)


def load_hdr(path):
    """Radiance RGBE file -> (h, w, 3) float32, rows bottom-up: what Tracer.set_environment_map takes (trc_host_load_hdr)."""
    w, h = C.c_uint32(), C.c_uint32()
    p = C.POINTER(C.c_float)()
    _check(lib().trc_host_load_hdr(os.fsencode(path), C.byref(w), C.byref(h), C.byref(p)), f"trc_host_load_hdr({path})")
    try:
        return np.ctypeslib.as_array(p, shape=(h.value, w.value, 3)).astype(np.float32).copy()
    finally:
        lib().trc_host_free(p)


def write_png(path, rgba8):
    """(H, W, 4) uint8, rows top-down -> PNG file."""
    assert rgba8.dtype == np.uint8 and rgba8.ndim == 3 and rgba8.shape[2] == 4 and rgba8.flags.c_contiguous
    _check(lib().trc_host_write_png(os.fsencode(path), rgba8.ctypes.data, rgba8.shape[1], rgba8.shape[0]), "trc_host_write_png")


def sobol_matrices32():
    """SobolMatrices32 of the first 40 dimensions, (40, 52) uint32 (include/trc_sobol.h)."""
    out = np.empty((abi.SOBOL_DIMS, abi.SOBOL_MATRIX_SIZE), dtype=np.uint32)
    lib().trc_host_sobol_matrices32(out.ctypes.data)
    return out


def sobol_interval_tables(log2res):
    """(VdCSobolMatrices[log2res - 1], VdCSobolMatricesInv[log2res - 1]) as two (52,) uint64 arrays."""
    vdc = np.empty(abi.SOBOL_MATRIX_SIZE, dtype=np.uint64)
    inv = np.empty(abi.SOBOL_MATRIX_SIZE, dtype=np.uint64)
    _check(lib().trc_host_sobol_interval_tables(log2res, vdc.ctypes.data, inv.ctypes.data), "trc_host_sobol_interval_tables")
    return vdc, inv


def density_info(density, sigma_a=10.0, sigma_s=90.0, g=0.5):
    """GridDensityInfo(10, 90, 0.5, nx, ny, nz, density) as at AAPLRenderer.mm:636."""
    assert density.dtype == np.float32 and density.ndim == 3 and density.flags.c_contiguous
    info = abi.GridDensityInfo()
    nz, ny, nx = density.shape
    lib().trc_host_make_density_info(sigma_a, sigma_s, g, nx, ny, nz, density.ctypes.data, C.byref(info))
    return info


def prepare_camera(width, height):
    cam = abi.Camera()
    lib().trc_host_prepare_camera(C.byref(cam), float(width), float(height))
    return cam


def make_camera(look_from, look_at, view_up, aperture, aspect, vfov_radians, focus_dist):
    cam = abi.Camera()
    f3 = C.c_float * 3
    lib().trc_host_make_camera(C.byref(cam), f3(*look_from), f3(*look_at), f3(*view_up), aperture, aspect,
                               vfov_radians, focus_dist)
    return cam


def fill_rng(seed, width, height):
    out = np.empty((height, width, 4), dtype=np.uint32)
    lib().trc_host_fill_rng(seed, width, height, out.ctypes.data)
    return out


def build_tree(leaves):
    """leaves: list of abi.BVH leaf records -> ctypes array of 2n-1 nodes, root at 0."""
    n = len(leaves)
    nodes = (abi.BVH * (2 * n - 1))()
    for i, leaf in enumerate(leaves):
        nodes[i] = leaf
    count = C.c_uint32()
    _check(lib().trc_host_build_tree(nodes, n, C.byref(count)), "trc_host_build_tree")
    assert count.value == 2 * n - 1
    return nodes


def build_node(box_min, box_max, ptype, pindex, model=None):
    box = abi.AABB()
    box.mini.x, box.mini.y, box.mini.z = box_min
    box.maxi.x, box.maxi.y, box.maxi.z = box_max
    m = abi.float4x4()
    for c in range(4):
        for r in range(4):
            v = float(model[r][c]) if model is not None else (1.0 if r == c else 0.0)
            setattr(m.columns[c], "xyzw"[r], v)
    out = abi.BVH()
    lib().trc_host_build_node(C.byref(box), C.byref(m), ptype, pindex, C.byref(out))
    return out
