"""ctypes loader of the CPU oracle (TEST INFRASTRUCTURE -- see oracle/oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))
from tracer_amd import abi  # noqa: E402  (PODs only; the oracle never calls the product path)

_LIBS = {}


def lib(libm=False):
    key = "liboracle_libm.so" if libm else "liboracle.so"
    if key not in _LIBS:
        path = os.path.join(os.environ.get("TRC_ORACLE_DIR") or _HERE, key)      # TRC_ORACLE_DIR: the sanitized build (tools/run_sanitizers.sh)
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `make oracle`")
        L = C.CDLL(path)
        f3 = C.POINTER(C.c_float)
        L.orc_uses_libm.restype = C.c_int
        L.orc_pcg32_srandom.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_uint64, C.c_uint64]
        L.orc_pcg32_srandom.restype = None
        L.orc_pcg32_random.argtypes = [C.POINTER(C.c_uint64), C.c_uint64]
        L.orc_pcg32_random.restype = C.c_uint32
        L.orc_randomF.argtypes = [C.POINTER(C.c_uint64), C.c_uint64]
        L.orc_randomF.restype = C.c_float
        L.orc_trace_rays.argtypes = [C.POINTER(abi.Scene), C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
        L.orc_trace_rays.restype = None
        L.orc_trace_rays_brute.argtypes = [C.POINTER(abi.Scene), C.c_void_p, C.c_size_t, C.c_void_p]
        L.orc_trace_rays_brute.restype = None
        L.orc_render.argtypes = [C.POINTER(abi.Scene), C.POINTER(abi.Camera), f3, C.c_uint32, C.c_uint32,
                                 C.c_void_p, C.c_void_p, C.POINTER(abi.Params), C.POINTER(abi.Stats), C.c_int]
        L.orc_render.restype = None
        L.orc_material_S_F.argtypes = [C.POINTER(abi.Material), f3, f3, f3, f3, f3, f3]
        L.orc_material_S_F.restype = None
        L.orc_material_F.argtypes = [C.POINTER(abi.Material), f3, f3, f3, f3, f3, f3]
        L.orc_material_F.restype = None
        L.orc_material_PDF.argtypes = [C.POINTER(abi.Material), f3, f3, f3]
        L.orc_material_PDF.restype = C.c_float
        L.orc_offset_ray.argtypes = [f3, f3, f3]
        L.orc_offset_ray.restype = None
        L.orc_fr_dielectric.argtypes = [C.c_float, C.c_float]
        L.orc_fr_dielectric.restype = C.c_float
        L.orc_fr_conductor.argtypes = [C.c_float, f3, f3, f3]
        L.orc_fr_conductor.restype = None
        L.orc_power_heuristic.argtypes = [C.c_int, C.c_float, C.c_int, C.c_float]
        L.orc_power_heuristic.restype = C.c_float
        L.orc_sobol_interval_to_index.argtypes = [C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32]
        L.orc_sobol_interval_to_index.restype = C.c_uint64
        L.orc_sobol_sample_float.argtypes = [C.c_uint64, C.c_uint32]
        L.orc_sobol_sample_float.restype = C.c_float
        L.orc_sobol_sample_dimension.argtypes = [C.c_uint32] * 6
        L.orc_sobol_sample_dimension.restype = C.c_float
        L.orc_cosine_sample_hemisphere.argtypes = [f3, f3]
        L.orc_cosine_sample_hemisphere.restype = None
        L.orc_erf.argtypes = [C.c_float]
        L.orc_erf.restype = C.c_float
        L.orc_erfinv.argtypes = [C.c_float]
        L.orc_erfinv.restype = C.c_float
        L.orc_aabb_hit_t.argtypes = [C.POINTER(abi.AABB), C.POINTER(abi.Ray), C.c_float, C.c_float,
                                     C.POINTER(C.c_float)]
        L.orc_aabb_hit_t.restype = C.c_int
        L.orc_cast_ray.argtypes = [C.POINTER(abi.Camera), C.c_float, C.c_float, C.POINTER(C.c_uint64), C.c_uint64,
                                   f3, f3]
        L.orc_cast_ray.restype = None
        L.orc_sppm_create.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64]
        L.orc_sppm_create.restype = C.c_void_p
        L.orc_sppm_destroy.argtypes = [C.c_void_p]
        L.orc_sppm_destroy.restype = None
        L.orc_sppm_frames.argtypes = [C.c_void_p, C.POINTER(abi.Scene), C.POINTER(abi.Camera), f3, C.c_void_p,
                                      C.c_void_p, C.c_uint32]
        L.orc_sppm_frames.restype = None
        L.orc_sppm_download.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.POINTER(abi.Complex)]
        L.orc_sppm_download.restype = None
        L.orc_photon_hash.argtypes = [f3, C.c_float]
        L.orc_photon_hash.restype = C.c_float
        L.orc_hg_sample.argtypes = [C.c_float, f3, f3, f3]
        L.orc_hg_sample.restype = C.c_float
        L.orc_phase_hg.argtypes = [C.c_float, C.c_float]
        L.orc_phase_hg.restype = C.c_float
        L.orc_grid_density.argtypes = [C.POINTER(abi.GridDensityInfo), C.c_void_p, f3]
        L.orc_grid_density.restype = C.c_float
        L.orc_tonemap.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(C.c_float)]
        L.orc_tonemap.restype = None
        L.orc_set_environment_map.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p]
        L.orc_set_environment_map.restype = None
        L.orc_set_density.argtypes = [C.POINTER(abi.GridDensityInfo), C.c_void_p]
        L.orc_set_density.restype = None
        L.orc_lbvh_build.argtypes = [C.POINTER(abi.BVH), C.c_uint32, C.POINTER(abi.BVH), C.POINTER(C.c_uint32)]
        L.orc_lbvh_build.restype = None
        L.orc_sah_build.argtypes = [C.POINTER(abi.BVH), C.c_uint32, C.POINTER(abi.BVH)]
        L.orc_sah_build.restype = None
        L.orc_sah_leaf.argtypes = [C.POINTER(abi.AABB), C.POINTER(abi.float4x4), C.c_int32, C.c_uint32, C.POINTER(abi.BVH)]
        L.orc_sah_leaf.restype = None
        L.orc_math.argtypes = [C.c_int, C.c_float, C.c_float]
        L.orc_math.restype = C.c_float
        _LIBS[key] = L
    return _LIBS[key]


RAY_DTYPE = np.dtype([("origin", np.float32, 3), ("tmax", np.float32), ("direction", np.float32, 3),
                      ("_pad", np.uint32)])
HIT_DTYPE = np.dtype([("hit", np.int32), ("pType", np.int32), ("pIndex", np.uint32), ("t", np.float32),
                      ("p", np.float32, 3), ("gn", np.float32, 3), ("sn", np.float32, 3), ("uv", np.float32, 2),
                      ("material", np.uint32), ("PDF", np.float32),
                      ("n_descend", np.uint32), ("n_return", np.uint32), ("n_leaf", np.uint32)])
assert RAY_DTYPE.itemsize == C.sizeof(abi.Ray) and HIT_DTYPE.itemsize == C.sizeof(abi.Hit)


def make_rays(origins, directions, tmax=None):
    n = len(origins)
    rays = np.zeros(n, dtype=RAY_DTYPE)
    rays["origin"] = origins
    rays["direction"] = directions
    rays["tmax"] = np.float32(np.finfo(np.float32).max) if tmax is None else tmax
    return rays


def trace_rays(scene_view, rays, any_hit=False, brute=False, libm=False):
    hits = np.zeros(len(rays), dtype=HIT_DTYPE)
    L = lib(libm)
    if brute:
        L.orc_trace_rays_brute(C.byref(scene_view), rays.ctypes.data, len(rays), hits.ctypes.data)
    else:
        L.orc_trace_rays(C.byref(scene_view), rays.ctypes.data, len(rays), hits.ctypes.data, 1 if any_hit else 0)
    return hits


def render(scene_view, camera, width, height, rng, accum=None, spp=1, max_depth=8, integrator=0, frame0=0,
           env=(0.0, 0.0, 0.0), tile_rank=0, tile_nranks=1, n_threads=0, libm=False, view_height=0, sobol=False):
    """kernelPathTracing on the CPU.  rng: (H,W,4) uint32, updated in place.  Returns (accum, stats)."""
    assert rng.dtype == np.uint32 and rng.shape == (height, width, 4) and rng.flags.c_contiguous
    if accum is None:
        accum = np.zeros((height, width, 4), dtype=np.float32)
    assert accum.dtype == np.float32 and accum.shape == (height, width, 4) and accum.flags.c_contiguous
    prm = abi.Params(spp=spp, max_depth=max_depth, integrator=integrator, frame0=frame0,
                     tile_rank=tile_rank, tile_nranks=tile_nranks,
                     flags=abi.FLAG_COLLECT_STATS | (abi.FLAG_SOBOL if sobol else 0), view_height=view_height)
    stats = abi.Stats()
    env_c = (C.c_float * 3)(*env)
    lib(libm).orc_render(C.byref(scene_view), C.byref(camera), env_c, width, height, rng.ctypes.data,
                         accum.ctypes.data, C.byref(prm), C.byref(stats), n_threads)
    return accum, stats


def shard_seed(seed, sample_group):
    """Seed of sample group g of a sample-sharded frame (include/tracer_abi.h, "sample sharding"): the golden-ratio
    stride 0x9E3779B97F4A7C15 times g is added modulo 2^64, so group 0 renders the unsharded frame's first samples."""
    return (seed + sample_group * 0x9E3779B97F4A7C15) % (1 << 64)


def render_sample_sharded(scene_view, camera, width, height, rngs, spp, **kw):
    """The sample-sharded frame as include/tracer_abi.h defines it, on the CPU: group g renders the whole frame with
    spp / S samples (frame0 = 0) from ITS RNG texture rngs[g] (= fillRNG of shard_seed(seed, g), S = len(rngs)); the
    composed texel is (((A_0 + A_1) + A_2) + ...) / float32(S) in binary32, channel by channel.  kw: render()'s
    (integrator, max_depth, env, tile_rank / tile_nranks to restrict the pixels, n_threads ...).
    Returns (composed, [stats per group])."""
    S = len(rngs)
    assert S >= 1 and spp % S == 0, "every group renders the same number of samples"
    total, stats = None, []
    for g in range(S):
        a, st = render(scene_view, camera, width, height, rngs[g], spp=spp // S, frame0=0, **kw)
        stats.append(st)
        total = a if total is None else np.add(total, a, dtype=np.float32)
    return np.divide(total, np.float32(S), dtype=np.float32), stats


def tonemap(accum):
    """fragmentShader's exposure + ACES on an (H, W, 4) float32 accumulator -> ((H, W, 4) uint8 top-down, exposure)."""
    assert accum.dtype == np.float32 and accum.ndim == 3 and accum.shape[2] == 4 and accum.flags.c_contiguous
    H, W = accum.shape[:2]
    out = np.empty((H, W, 4), dtype=np.uint8)
    e = C.c_float(0)
    lib().orc_tonemap(accum.ctypes.data, W, H, out.ctypes.data, C.byref(e))
    return out, e.value


_ENVMAP_KEEPALIVE = []


def set_environment_map(rgb):
    """(h, w, 3) float32 equirectangular environment, or None for the constant one."""
    _ENVMAP_KEEPALIVE.clear()
    for libm in (False, True):
        if rgb is None:
            lib(libm).orc_set_environment_map(0, 0, None)
        else:
            assert rgb.dtype == np.float32 and rgb.ndim == 3 and rgb.shape[2] == 3 and rgb.flags.c_contiguous
            lib(libm).orc_set_environment_map(rgb.shape[1], rgb.shape[0], rgb.ctypes.data)
    if rgb is not None:
        _ENVMAP_KEEPALIVE.append(rgb)


_DENSITY_KEEPALIVE = []


def set_density(info, density):
    """Density grid for integrator 2 (traceVolume); `density`: float32 array (nz, ny, nx) or None to clear."""
    _DENSITY_KEEPALIVE.clear()
    for libm in (False, True):
        if density is None:
            lib(libm).orc_set_density(None, None)
        else:
            assert density.dtype == np.float32 and density.flags.c_contiguous
            lib(libm).orc_set_density(C.byref(info), density.ctypes.data)
    if density is not None:
        _DENSITY_KEEPALIVE.append(density)


def lbvh_build(leaves, n):
    """LBVH over `n` leaf records (ctypes array / pointer of abi.BVH) -> (abi.BVH * (2n-1), height)."""
    out = (abi.BVH * (2 * n - 1))()
    h = C.c_uint32(0)
    lib().orc_lbvh_build(leaves, n, out, C.byref(h))
    return out, h.value


def sah_build(leaves, n):
    """The reference's SAH build (BVH.hh:35-269) over `n` leaf records -> abi.BVH * (2n-1), root first."""
    out = (abi.BVH * (2 * n - 1))()
    lib().orc_sah_build(leaves, n, out)
    return out


def sah_leaf(box, model_matrix, ptype, pindex):
    """BVH::buildNode (BVH.hh:273-314) -> abi.BVH leaf record."""
    out = abi.BVH()
    lib().orc_sah_leaf(C.byref(box), C.byref(model_matrix), ptype, pindex, C.byref(out))
    return out


CAMREC_DTYPE = np.dtype([("ratio", np.float32, 4), ("position", np.float32, 4), ("direction", np.float32, 4),
                         ("valid", np.uint8), ("_pad0", np.uint8, 15), ("alternative", np.float32, 4),
                         ("flux", np.float32, 4), ("radius", np.float32), ("photonCount", np.uint32),
                         ("_pad1", np.uint32, 2)])
PHOTON_DTYPE = np.dtype([("flux", np.float32, 4), ("normal", np.float32, 4), ("position", np.float32, 4),
                         ("direction", np.float32, 4), ("step", np.uint8), ("active", np.uint8), ("_pad", np.uint8, 14)])
assert CAMREC_DTYPE.itemsize == 112 and PHOTON_DTYPE.itemsize == 80


class Sppm:
    """SPPM pass on the CPU oracle (Photon.metal); canvas rng / accum are updated in place."""

    def __init__(self, width, height, photon_seed):
        self.W, self.H = width, height
        self._h = lib().orc_sppm_create(width, height, photon_seed)

    def frames(self, scene_view, camera, rng, accum, n_frames=1, env=(0.0, 0.0, 0.0)):
        assert rng.dtype == np.uint32 and rng.shape == (self.H, self.W, 4) and rng.flags.c_contiguous
        assert accum.dtype == np.float32 and accum.shape == (self.H, self.W, 4) and accum.flags.c_contiguous
        lib().orc_sppm_frames(self._h, C.byref(scene_view), C.byref(camera), (C.c_float * 3)(*env), rng.ctypes.data,
                              accum.ctypes.data, n_frames)

    def download(self):
        n = abi.PHOTON_HASHN
        cam = np.zeros(self.W * self.H, dtype=CAMREC_DTYPE)
        pho = np.zeros(n * n, dtype=PHOTON_DTYPE)
        mark = np.zeros((n, n, 4), dtype=np.float32)
        count = np.zeros((n, n), dtype=np.float32)
        cx = abi.Complex()
        lib().orc_sppm_download(self._h, cam.ctypes.data, pho.ctypes.data, mark.ctypes.data, count.ctypes.data, C.byref(cx))
        return cam, pho, mark, count, cx

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_sppm_destroy(self._h)
            self._h = None
