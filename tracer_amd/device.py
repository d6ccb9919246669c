"""Device side: the HIP path tracer behind the C ABI (libtracer_amd.so, gfx950).

`Tracer` plays the role of the reference's AAPLRenderer for the path-tracing path
(RT_Metal/Tracer/AAPLRenderer.hh:7-14: init / render); every method is one trc_* call of
include/tracer_abi.h.  There is NO CPU fallback: a missing library or a missing GPU raises.
"""
import ctypes as C
import os

import numpy as np

from . import abi

_LIB = None
_LIB_FAST = None
_LIB_HOOKS = None


class TracerError(RuntimeError):
    def __init__(self, what, status, detail=""):
        super().__init__(f"{what}: trc_status {status}" + (f" ({detail})" if detail else ""))
        self.status = status


def lib_path():
    """The in-tree build; TRC_AMD_LIB points the A/B tools (tools/ab_bench.py) at another build of the same library."""
    return os.environ.get("TRC_AMD_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libtracer_amd.so")


def fast_lib_path():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libtracer_amd_fast.so")


def hooks_lib_path():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libtracer_amd_hooks.so")


def lib(fast_math=False, hooks=False):
    """Load libtracer_amd.so (fails loudly when the HIP extension has not been built); fast_math=True loads the
    fast-math build of the same sources (libtracer_amd_fast.so) instead; hooks=True the build that also exports the test
    hooks of include/tracer_test_hooks.h (libtracer_amd_hooks.so: tests and tools only)."""
    global _LIB, _LIB_FAST, _LIB_HOOKS
    if hooks:
        if _LIB_HOOKS is None:
            _LIB_HOOKS = _load(hooks_lib_path(), hooks=True)
            assert _LIB_HOOKS.trc_has_test_hooks() == 1
        return _LIB_HOOKS
    if fast_math:
        if _LIB_FAST is None:
            _LIB_FAST = _load(fast_lib_path())
            assert _LIB_FAST.trc_build_flavor() == b"fast-math"
        return _LIB_FAST
    if _LIB is None:
        _LIB = _load(lib_path())
    return _LIB


def _load(path, hooks=False):
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: the HIP extension is not built "
                           f"(run `make hip` or __graft_entry__.build()); there is no CPU fallback")
    L = C.CDLL(path)
    vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int32
    L.trc_abi_version.restype = u32
    L.trc_build_flavor.restype = C.c_char_p
    L.trc_status_string.argtypes = [i32]
    L.trc_status_string.restype = C.c_char_p
    L.trc_last_error.argtypes = [vp]
    L.trc_last_error.restype = C.c_char_p
    L.trc_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.trc_destroy.argtypes = [vp]
    L.trc_destroy.restype = None
    L.trc_upload_scene.argtypes = [vp, C.POINTER(abi.Scene)]
    L.trc_set_environment_map.argtypes = [vp, u32, u32, vp]
    L.trc_tonemap.argtypes = [vp, vp, C.POINTER(C.c_float)]
    L.trc_upload_density.argtypes = [vp, C.POINTER(abi.GridDensityInfo), vp]
    L.trc_upload_scene_lbvh.argtypes = [vp, C.POINTER(abi.Scene)]
    L.trc_upload_scene_sah.argtypes = [vp, C.POINTER(abi.Scene)]
    L.trc_upload_scene_device.argtypes = [vp, C.POINTER(abi.Scene), u32]
    L.trc_download_bvh.argtypes = [vp, C.POINTER(abi.BVH), u32, C.POINTER(u32)]
    L.trc_lbvh_info.argtypes = [vp, C.POINTER(u32), C.POINTER(u32), C.POINTER(C.c_float)]
    L.trc_set_camera.argtypes = [vp, C.POINTER(abi.Camera)]
    L.trc_set_environment.argtypes = [vp, C.POINTER(C.c_float)]
    L.trc_resize.argtypes = [vp, u32, u32]
    L.trc_seed.argtypes = [vp, u64]
    for name in ("trc_upload_rng", "trc_download_rng", "trc_upload_accum", "trc_download_accum"):
        getattr(L, name).argtypes = [vp, vp]
    L.trc_clear_accum.argtypes = [vp]
    L.trc_render.argtypes = [vp, C.POINTER(abi.Params)]
    L.trc_synchronize.argtypes = [vp]
    L.trc_trace_rays.argtypes = [vp, vp, C.c_size_t, vp, C.c_int]
    L.trc_get_stats.argtypes = [vp, C.POINTER(abi.Stats)]
    L.trc_reset_stats.argtypes = [vp]
    L.trc_sppm_init.argtypes = [vp, u64]
    L.trc_sppm_frames.argtypes = [vp, u32]
    L.trc_sppm_download.argtypes = [vp, vp, vp, vp, vp, C.POINTER(abi.Complex)]
    L.trc_device_info.argtypes = [vp, C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_size_t)]
    L.trc_group_unique_id.argtypes = [C.POINTER(C.c_uint8)]
    L.trc_group_init.argtypes = [vp, C.POINTER(C.c_uint8), C.c_int, C.c_int]
    L.trc_group_reduce_accum.argtypes = [vp, C.c_int]
    L.trc_group_reduce_accum_async.argtypes = [vp, C.c_int]
    L.trc_group_allreduce_mean_accum.argtypes = [vp]
    L.trc_group_compose_samples.argtypes = [vp, C.c_int, u32]
    L.trc_group_compose_samples_async.argtypes = [vp, C.c_int, u32]
    L.trc_shard_seed.argtypes = [u64, u32]
    L.trc_device_pci_bus_id.argtypes = [vp, C.c_char_p, C.c_size_t]
    L.trc_download_composed.argtypes = [vp, vp]
    L.trc_group_finalize.argtypes = [vp]
    L.trc_group_set_collectives.argtypes = [vp, vp, C.c_int, C.c_int]
    L.trc_debug_set.argtypes = [vp, C.c_char_p, C.c_int]
    L.trc_debug_block_costs.argtypes = [vp, vp, vp, u32, C.POINTER(u32), C.POINTER(u32)]
    L.trc_debug_launch_shape.argtypes = [vp, C.POINTER(abi.LaunchShape)]
    if hooks:
        L.trc_debug_profile.argtypes = [vp, C.POINTER(C.c_uint64), u32]
        L.trc_sppm_hash_cells.argtypes = [vp, vp, C.c_size_t, C.c_float, vp]
        L.trc_div_by_test.argtypes = [vp, vp, vp, C.c_size_t, vp, vp]
        L.trc_unary_test.argtypes = [vp, u32, u32, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(u32)]
        for name in abi.HOOK_SYMBOLS:
            getattr(L, name).restype = i32
    L.trc_has_test_hooks.restype = C.c_int
    for name in abi.DEVICE_SYMBOLS:
        f = getattr(L, name)
        if name not in ("trc_abi_version", "trc_build_flavor", "trc_has_test_hooks", "trc_status_string", "trc_last_error", "trc_destroy", "trc_shard_seed"):
            f.restype = i32
    L.trc_shard_seed.restype = u64
    if L.trc_abi_version() != abi.TRC_ABI_VERSION:
        raise RuntimeError("libtracer_amd.so ABI version mismatch")
    return L


def group_unique_id():
    buf = (C.c_uint8 * abi.TRC_UNIQUE_ID_BYTES)()
    st = lib().trc_group_unique_id(buf)
    if st != abi.OK:
        raise TracerError("trc_group_unique_id", st)
    return bytes(buf)


class Tracer:
    """One context per GPU (single-threaded, one HIP stream)."""

    def __init__(self, device=0, fast_math=False, hooks=False):
        self._L = lib(fast_math, hooks)
        self._h = C.c_void_p()
        st = self._L.trc_create(device, C.byref(self._h))
        if st != abi.OK:
            self._h = None
            raise TracerError(f"trc_create(device={device})", st, self._L.trc_status_string(st).decode())
        self.width = self.height = 0

    def _check(self, st, what):
        if st != abi.OK:
            raise TracerError(what, st, self._L.trc_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None):
            self._L.trc_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # --- scene / camera / frame -------------------------------------------------
    def upload_scene(self, scene_view):
        self._check(self._L.trc_upload_scene(self._h, C.byref(scene_view)), "trc_upload_scene")

    def set_environment_map(self, rgb):
        """(h, w, 3) float32 equirectangular environment; None returns to the constant one."""
        if rgb is None:
            self._check(self._L.trc_set_environment_map(self._h, 0, 0, None), "trc_set_environment_map")
            return
        assert rgb.dtype == np.float32 and rgb.ndim == 3 and rgb.shape[2] == 3 and rgb.flags.c_contiguous
        self._check(self._L.trc_set_environment_map(self._h, rgb.shape[1], rgb.shape[0], rgb.ctypes.data),
                    "trc_set_environment_map")

    def tonemap(self):
        """fragmentShader's auto-exposure + ACES on the accumulator -> ((H, W, 4) uint8, rows top-down; exposure)."""
        out = np.empty((self.height, self.width, 4), dtype=np.uint8)
        e = C.c_float(0)
        self._check(self._L.trc_tonemap(self._h, out.ctypes.data, C.byref(e)), "trc_tonemap")
        return out, e.value

    def upload_density(self, info, density):
        """Density grid (nz, ny, nx) float32 of the GridDensity medium for integrator 2; None clears it."""
        if density is None:
            self._check(self._L.trc_upload_density(self._h, None, None), "trc_upload_density")
            return
        assert density.dtype == np.float32 and density.flags.c_contiguous
        self._check(self._L.trc_upload_density(self._h, C.byref(info), density.ctypes.data), "trc_upload_density")

    def upload_scene_lbvh(self, leaves_view):
        """Scene whose bvhList holds only leaf records (HostScene.leaves_view()); the tree is built on the GPU."""
        self._check(self._L.trc_upload_scene_lbvh(self._h, C.byref(leaves_view)), "trc_upload_scene_lbvh")

    def upload_scene_sah(self, leaves_view):
        """Same input as upload_scene_lbvh; the reference's binned-SAH tree (BVH::buildTree), built on the GPU."""
        self._check(self._L.trc_upload_scene_sah(self._h, C.byref(leaves_view)), "trc_upload_scene_sah")

    def upload_scene_device(self, view, flags):
        """trc_upload_scene_device: abi.TREE_SAH | abi.TREE_TRIANGLE_LEAVES (bvhList = the analytic primitives' leaves only)."""
        self._check(self._L.trc_upload_scene_device(self._h, C.byref(view), flags), "trc_upload_scene_device")

    def download_bvh(self):
        """The device-built tree in the reference's array layout: ctypes array of abi.BVH (2n-1 records)."""
        n = C.c_uint32(0)
        self._check(self._L.trc_download_bvh(self._h, None, 0, C.byref(n)), "trc_download_bvh")
        out = (abi.BVH * n.value)()
        self._check(self._L.trc_download_bvh(self._h, out, n.value, C.byref(n)), "trc_download_bvh")
        return out

    def lbvh_info(self):
        """(n_nodes, depth of the deepest leaf, GPU build time in ms) of the last upload_scene_lbvh."""
        n, h, ms = C.c_uint32(0), C.c_uint32(0), C.c_float(0)
        self._check(self._L.trc_lbvh_info(self._h, C.byref(n), C.byref(h), C.byref(ms)), "trc_lbvh_info")
        return n.value, h.value, ms.value

    def set_camera(self, camera):
        self._check(self._L.trc_set_camera(self._h, C.byref(camera)), "trc_set_camera")

    def set_environment(self, rgb):
        self._check(self._L.trc_set_environment(self._h, (C.c_float * 3)(*rgb)), "trc_set_environment")

    def resize(self, width, height):
        self._check(self._L.trc_resize(self._h, width, height), "trc_resize")
        self.width, self.height = width, height

    def seed(self, seed):
        self._check(self._L.trc_seed(self._h, seed), "trc_seed")

    def upload_rng(self, rng):
        assert rng.dtype == np.uint32 and rng.shape == (self.height, self.width, 4) and rng.flags.c_contiguous
        self._check(self._L.trc_upload_rng(self._h, rng.ctypes.data), "trc_upload_rng")

    def download_rng(self):
        out = np.empty((self.height, self.width, 4), dtype=np.uint32)
        self._check(self._L.trc_download_rng(self._h, out.ctypes.data), "trc_download_rng")
        return out

    def upload_accum(self, accum):
        assert accum.dtype == np.float32 and accum.shape == (self.height, self.width, 4) and accum.flags.c_contiguous
        self._check(self._L.trc_upload_accum(self._h, accum.ctypes.data), "trc_upload_accum")

    def download_accum(self):
        out = np.empty((self.height, self.width, 4), dtype=np.float32)
        self._check(self._L.trc_download_accum(self._h, out.ctypes.data), "trc_download_accum")
        return out

    def clear_accum(self):
        self._check(self._L.trc_clear_accum(self._h), "trc_clear_accum")

    # --- the hot path --------------------------------------------------------------
    def render(self, spp=1, max_depth=8, integrator=abi.INTEGRATOR_PATH, frame0=0, tile_rank=0, tile_nranks=1,
               collect_stats=False, view_height=0, fixed_order=False, sobol=False, small_blocks=None):
        flags = ((abi.FLAG_COLLECT_STATS if collect_stats else 0) | (abi.FLAG_FIXED_ORDER if fixed_order else 0) |
                 (abi.FLAG_SOBOL if sobol else 0) |
                 (0 if small_blocks is None else (abi.FLAG_SMALL_BLOCKS if small_blocks else abi.FLAG_LARGE_BLOCKS)))
        prm = abi.Params(spp=spp, max_depth=max_depth, integrator=integrator, frame0=frame0, tile_rank=tile_rank,
                         tile_nranks=tile_nranks, flags=flags, view_height=view_height)
        self._check(self._L.trc_render(self._h, C.byref(prm)), "trc_render")

    def synchronize(self):
        self._check(self._L.trc_synchronize(self._h), "trc_synchronize")

    def trace_rays(self, rays, any_hit=False, production=False):
        """rays: structured array with the layout of trc_ray -> structured array of trc_hit.
        production=True walks the tree as the render kernels do (no counters)."""
        from .dtypes import HIT_DTYPE, RAY_DTYPE
        assert rays.dtype == RAY_DTYPE and rays.flags.c_contiguous
        hits = np.zeros(len(rays), dtype=HIT_DTYPE)
        self._check(self._L.trc_trace_rays(self._h, rays.ctypes.data, len(rays), hits.ctypes.data,
                                           (abi.TRACE_ANY_HIT if any_hit else 0) | (abi.TRACE_PRODUCTION if production else 0)),
                    "trc_trace_rays")
        return hits

    def stats(self):
        s = abi.Stats()
        self._check(self._L.trc_get_stats(self._h, C.byref(s)), "trc_get_stats")
        return s

    PROFILE_SITES = ["loop", "box_step", "square", "sphere", "cube", "triangle", "shade", "lambert", "metal",
                     "beckmann_sample", "beckmann_eval", "path_end"]

    def debug_profile(self):
        """{site: (lanes, wavefronts, lanes/(64*wavefronts), cycles)} of the instrumented kernels."""
        n = len(self.PROFILE_SITES)
        buf = (C.c_uint64 * (3 * n))()
        self._check(self._L.trc_debug_profile(self._h, buf, n), "trc_debug_profile")
        return {s: (buf[3 * i], buf[3 * i + 1], buf[3 * i] / (64.0 * buf[3 * i + 1]) if buf[3 * i + 1] else 0.0,
                    buf[3 * i + 2]) for i, s in enumerate(self.PROFILE_SITES)}

    def reset_stats(self):
        self._check(self._L.trc_reset_stats(self._h), "trc_reset_stats")

    def device_info(self):
        name = C.create_string_buffer(256)
        cu, mem = C.c_int(), C.c_size_t()
        self._check(self._L.trc_device_info(self._h, name, 256, C.byref(cu), C.byref(mem)), "trc_device_info")
        return {"name": name.value.decode(), "cu_count": cu.value, "hbm_bytes": mem.value}

    # --- SPPM pass (Photon.metal) -----------------------------------------------------
    def sppm_init(self, photon_seed):
        self._check(self._L.trc_sppm_init(self._h, photon_seed), "trc_sppm_init")

    def sppm_frames(self, n_frames=1):
        self._check(self._L.trc_sppm_frames(self._h, n_frames), "trc_sppm_frames")

    def sppm_hash_cells(self, cells, hash_scale):
        """Photon.hh hash() of n cell indices (n x 3 float32) at one scale, evaluated on the device."""
        cells = np.ascontiguousarray(cells, dtype=np.float32).reshape(-1, 3)
        out = np.empty(len(cells), dtype=np.float32)
        self._check(self._L.trc_sppm_hash_cells(self._h, cells.ctypes.data, len(cells), float(hash_scale), out.ctypes.data),
                    "trc_sppm_hash_cells")
        return out

    def sppm_download(self):
        """(camera records, photon records, mark grid, count grid, Complex) in the reference's layouts"""
        from .dtypes import CAMREC_DTYPE, PHOTON_DTYPE
        n = abi.PHOTON_HASHN
        cam = np.zeros(self.width * self.height, dtype=CAMREC_DTYPE)
        pho = np.zeros(n * n, dtype=PHOTON_DTYPE)
        mark = np.zeros((n, n, 4), dtype=np.float32)
        count = np.zeros((n, n), dtype=np.float32)
        cx = abi.Complex()
        self._check(self._L.trc_sppm_download(self._h, cam.ctypes.data, pho.ctypes.data, mark.ctypes.data,
                                              count.ctypes.data, C.byref(cx)), "trc_sppm_download")
        return cam, pho, mark, count, cx

    # --- multi-GPU -------------------------------------------------------------------
    def group_init(self, unique_id, nranks, rank):
        buf = (C.c_uint8 * abi.TRC_UNIQUE_ID_BYTES).from_buffer_copy(unique_id)
        self._check(self._L.trc_group_init(self._h, buf, nranks, rank), "trc_group_init")

    def group_reduce_accum(self, root=0):
        self._check(self._L.trc_group_reduce_accum(self._h, root), "trc_group_reduce_accum")

    def group_allreduce_mean_accum(self):
        """Sample sharding composed into EVERY rank's accumulator: rank-ordered sum of the ranks' whole-frame accumulators
        over the number of ranks (trc_group_allreduce_mean_accum)."""
        self._check(self._L.trc_group_allreduce_mean_accum(self._h), "trc_group_allreduce_mean_accum")

    def group_compose_samples(self, root=0, sample_groups=None):
        """Sample sharding (tracer_abi.h): rank-ordered sum of the ranks' accumulators / sample_groups, to `root`
        (download_composed); sample_groups defaults to the number of ranks (no tile split inside a group)."""
        self._check(self._L.trc_group_compose_samples(self._h, root, sample_groups or 0), "trc_group_compose_samples")

    def group_compose_samples_async(self, root=0, sample_groups=None):
        self._check(self._L.trc_group_compose_samples_async(self._h, root, sample_groups or 0), "trc_group_compose_samples_async")

    def pci_bus_id(self):
        buf = C.create_string_buffer(64)
        self._check(self._L.trc_device_pci_bus_id(self._h, buf, 64), "trc_device_pci_bus_id")
        return buf.value.decode()

    def group_reduce_accum_async(self, root=0):
        """Compose on a second stream and switch to the other accumulator (see trc_group_reduce_accum_async)."""
        self._check(self._L.trc_group_reduce_accum_async(self._h, root), "trc_group_reduce_accum_async")

    def download_composed(self):
        out = np.empty((self.height, self.width, 4), dtype=np.float32)
        self._check(self._L.trc_download_composed(self._h, out.ctypes.data), "trc_download_composed")
        return out

    def set_collectives(self, table, nranks, rank):
        """Caller-supplied collectives instead of an RCCL communicator (trc_group_set_collectives); `table` is an
        object with a ctypes `table` attribute (tracer_amd.gloo_collectives.GlooCollectives) and must outlive the group."""
        self._coll = table
        self._check(self._L.trc_group_set_collectives(self._h, C.byref(table.table) if table is not None else None, nranks, rank),
                    "trc_group_set_collectives")

    def debug_set(self, knob, value):
        """A/B and test knobs of this context (trc_debug_set): scheduling only, never a pixel."""
        self._check(self._L.trc_debug_set(self._h, knob.encode(), int(value)), "trc_debug_set")

    def div_by_test(self, a, b):
        """(fast, plain): 3 quotients per operand pair through the guarded shared-divisor division and through `/`."""
        a = np.ascontiguousarray(a, dtype=np.float32); b = np.ascontiguousarray(b, dtype=np.float32)
        assert a.shape == b.shape and a.ndim == 1
        fast, plain = np.empty((len(a), 3), np.float32), np.empty((len(a), 3), np.float32)
        self._check(self._L.trc_div_by_test(self._h, a.ctypes.data, b.ctypes.data, len(a), fast.ctypes.data, plain.ctypes.data), "trc_div_by_test")
        return fast, plain

    def block_costs(self):
        """(tiles, costs, blk_shift) of the last render launch (trc_debug_block_costs)."""
        n, bs = C.c_uint32(0), C.c_uint32(0)
        self._check(self._L.trc_debug_block_costs(self._h, None, None, 0, C.byref(n), C.byref(bs)), "trc_debug_block_costs")
        tiles, costs = np.zeros(n.value, np.uint32), np.zeros(n.value, np.uint32)
        self._check(self._L.trc_debug_block_costs(self._h, tiles.ctypes.data, costs.ctypes.data, n.value, C.byref(n), C.byref(bs)),
                    "trc_debug_block_costs")
        return tiles, costs, bs.value

    def unary_test(self, op, first_bits=0, count=1 << 32):
        """(mismatches, first mismatching bit pattern) of rcp_cr / sqrt_cr / rsqrt_cr (op 0 / 1 / 2) and the constant-divisor quotients (op 3 .. 6) against the compiler's sequences"""
        n, first = C.c_uint64(0), C.c_uint32(0)
        self._check(self._L.trc_unary_test(self._h, op, first_bits, count, C.byref(n), C.byref(first)), "trc_unary_test")
        return n.value, first.value

    def launch_shape(self):
        """chain bound / work bound of the last render launch (trc_debug_launch_shape) as a dict"""
        s = abi.LaunchShape()
        self._check(self._L.trc_debug_launch_shape(self._h, C.byref(s)), "trc_debug_launch_shape")
        return {k: getattr(s, k) for k, _ in abi.LaunchShape._fields_}

    def group_finalize(self):
        self._check(self._L.trc_group_finalize(self._h), "trc_group_finalize")
