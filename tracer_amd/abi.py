"""ctypes mirror of include/tracer_abi.h (the drop-in boundary).

Every Structure below restates one POD of the header; sizes are asserted against
SURVEY.md Appendix A at import time so a drift between the header and this file
fails loudly.  Reference definitions: RT_Metal/Metal/{AABB,BVH,Sphere,Square,Cube,
Triangle,Texture,Material,Camera}.hh (see the header for file:line).
"""
import ctypes as C

TRC_ABI_VERSION = 6
TRC_TILE = 16
TRC_MAX_BVH_DEPTH = 64
TREE_SAH, TREE_TRIANGLE_LEAVES = 1, 2
TRC_UNIQUE_ID_BYTES = 128
# enum trc_coll_dtype / trc_coll_op (ncclDataType_t / ncclRedOp_t ordinals)
DT_U8, DT_U32, DT_F32 = 1, 3, 7
OP_SUM, OP_MAX, OP_MIN = 0, 2, 3

# enum trc_PrimitiveType (BVH.hh:6-8)
PRIM_SPHERE, PRIM_SQUARE, PRIM_CUBE, PRIM_TRIANGLE, PRIM_BVH, PRIM_UNKNOW = range(6)
# enum trc_MaterialType (Material.hh:18-20)
(MAT_DIFFUSE, MAT_LAMBERT, MAT_ORENNAYAR, MAT_PLASTIC, MAT_METAL, MAT_GLASS,
 MAT_ISOTROPIC, MAT_DIELECTRIC, MAT_DEMOFOX, MAT_PBR, MAT_NIL) = range(11)
# enum trc_TextureType (Texture.hh:6)
TEX_CONSTANT, TEX_CHECKER, TEX_NOISE, TEX_IMAGE = range(4)
# enum trc_integrator
INTEGRATOR_PATH, INTEGRATOR_MIS, INTEGRATOR_VOLUME = 0, 1, 2
MEDIUM_NIL, MEDIUM_HOMOGENEOUS, MEDIUM_GRIDDENSITY = 0, 1, 2
# enum trc_host_scene_kind
SCENE_CORNELL, SCENE_CORNELL_SPHERES, SCENE_CORNELL_MESH, SCENE_CORNELL_VOLUME = 0, 1, 2, 3
FLAG_COLLECT_STATS = 1
FLAG_FIXED_ORDER = 2
FLAG_SOBOL = 4
FLAG_SMALL_BLOCKS, FLAG_LARGE_BLOCKS = 8, 16
TRACE_ANY_HIT, TRACE_PRODUCTION = 1, 2
SOBOL_DIMS, SOBOL_MATRIX_SIZE = 40, 52

SHARD_SEED_STRIDE = 0x9E3779B97F4A7C15


def shard_seed(seed, sample_group):
    """trc_shard_seed: the RNG seed of sample group g of a sample-sharded frame (group 0 keeps the seed)"""
    return (int(seed) + int(sample_group) * SHARD_SEED_STRIDE) & 0xFFFFFFFFFFFFFFFF


# status codes
OK = 0
ERR_INVALID_ARG, ERR_NO_DEVICE, ERR_HIP, ERR_NO_SCENE, ERR_NO_FRAME = -1, -2, -3, -4, -5
ERR_BVH_INVALID, ERR_UNSUPPORTED, ERR_RCCL, ERR_OOM = -6, -7, -8, -9


class float2(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float)]


class float3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float), ("_pad", C.c_float)]


class float4(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float), ("w", C.c_float)]


class float4x4(C.Structure):
    _fields_ = [("columns", float4 * 4)]


class AABB(C.Structure):
    _fields_ = [("mini", float3), ("maxi", float3)]


class BVH(C.Structure):
    _fields_ = [("parent", C.c_uint32), ("left", C.c_uint32), ("right", C.c_uint32), ("axis", C.c_uint32),
                ("pType", C.c_int32), ("pIndex", C.c_uint32), ("_pad", C.c_uint32 * 2), ("bBOX", AABB)]


class Sphere(C.Structure):
    _fields_ = [("radius", C.c_float), ("_pad0", C.c_uint32 * 3), ("center", float3),
                ("model_matrix", float4x4), ("normal_matrix", float4x4), ("inverse_matrix", float4x4),
                ("material", C.c_uint32), ("_pad1", C.c_uint32 * 3), ("boundingBOX", AABB)]


class Square(C.Structure):
    _fields_ = [("axis_i", C.c_uint8), ("axis_j", C.c_uint8), ("_pad0", C.c_uint8 * 6),
                ("range_i", float2), ("range_j", float2),
                ("axis_k", C.c_uint8), ("_pad1", C.c_uint8 * 3), ("value_k", C.c_float),
                ("model_matrix", float4x4), ("normal_matrix", float4x4), ("inverse_matrix", float4x4),
                ("material", C.c_uint32), ("_pad2", C.c_uint32 * 3), ("boundingBOX", AABB)]


class Cube(C.Structure):
    _fields_ = [("model_matrix", float4x4), ("normal_matrix", float4x4), ("inverse_matrix", float4x4),
                ("box", AABB), ("material", C.c_uint32), ("_pad", C.c_uint32 * 3)]


class TriangleVertex(C.Structure):
    _fields_ = [("v", C.c_float * 3), ("n", C.c_float * 3), ("uv", C.c_float * 2)]


class TextureInfo(C.Structure):
    _fields_ = [("type", C.c_int32), ("textureIndex", C.c_uint32), ("_pad", C.c_uint32 * 2), ("albedo", float3)]


class Material(C.Structure):
    _fields_ = [("type", C.c_int32), ("medium", C.c_int32), ("specular", C.c_uint8), ("_pad0", C.c_uint8 * 3),
                ("eta", C.c_float), ("roughness", C.c_float), ("_pad1", C.c_uint32 * 3),
                ("textureInfo", TextureInfo)]


class Camera(C.Structure):
    _fields_ = [("lookFrom", float3), ("lookAt", float3), ("viewUp", float3),
                ("vfov", C.c_float), ("aspect", C.c_float), ("aperture", C.c_float), ("lenRadius", C.c_float),
                ("focus_dist", C.c_float), ("_pad", C.c_uint32 * 3),
                ("u", float3), ("v", float3), ("w", float3),
                ("vertical", float3), ("horizontal", float3), ("cornerLowLeft", float3)]


class PhotonRecord(C.Structure):
    _fields_ = [("flux", float3), ("normal", float3), ("position", float3), ("direction", float3),
                ("step", C.c_uint8), ("active", C.c_uint8), ("_pad", C.c_uint8 * 14)]


class CameraRecord(C.Structure):
    _fields_ = [("ratio", float3), ("position", float3), ("direction", float3),
                ("valid", C.c_uint8), ("_pad0", C.c_uint8 * 15), ("alternative", float3), ("flux", float3),
                ("radius", C.c_float), ("photonCount", C.c_uint32), ("_pad1", C.c_uint32 * 2)]


class Complex(C.Structure):
    _fields_ = [("tex_size", float2), ("view_size", float2), ("running_time", C.c_float),
                ("frame_count", C.c_uint32), ("_pad0", C.c_uint32 * 2), ("photonBox", AABB),
                ("photonBoxSize", float3), ("photonInitialRadius", C.c_float), ("photonHashScale", C.c_float),
                ("totalPhotonSum", C.c_float), ("framePhotonSum", C.c_uint32)]


PHOTON_HASHN = 512


class Scene(C.Structure):
    _fields_ = [("bvhList", C.POINTER(BVH)), ("n_bvh", C.c_uint32),
                ("sphereList", C.POINTER(Sphere)), ("n_sphere", C.c_uint32),
                ("squareList", C.POINTER(Square)), ("n_square", C.c_uint32),
                ("cubeList", C.POINTER(Cube)), ("n_cube", C.c_uint32),
                ("triList", C.POINTER(TriangleVertex)), ("n_vertex", C.c_uint32),
                ("idxList", C.POINTER(C.c_uint32)), ("n_index", C.c_uint32),
                ("materials", C.POINTER(Material)), ("n_material", C.c_uint32)]


class GridDensityInfo(C.Structure):
    _fields_ = [("sigma_a", C.c_float), ("sigma_s", C.c_float), ("sigma_t", C.c_float), ("g", C.c_float),
                ("invMaxDensity", C.c_float), ("nx", C.c_uint32), ("ny", C.c_uint32), ("nz", C.c_uint32)]


# enum trc_pbrt_material
PBRT_MATTE, PBRT_PLASTIC, PBRT_METAL, PBRT_MIRROR, PBRT_GLASS, PBRT_OTHER = range(6)
# enum trc_pbrt_shape_kind / trc_pbrt_texture
PBRT_SHAPE_SPHERE, PBRT_SHAPE_TRIANGLEMESH, PBRT_SHAPE_DISK, PBRT_SHAPE_CYLINDER, PBRT_SHAPE_PLYMESH = 0, 3, 6, 7, 8
PBRT_SHAPE_CONE, PBRT_SHAPE_PARABOLOID, PBRT_SHAPE_HYPERBOLOID = 9, 10, 11
PBRT_TEX_NONE, PBRT_TEX_CHECKERBOARD, PBRT_TEX_OTHER = 0, 1, 2


class PbrtInfo(C.Structure):
    _fields_ = [("camera_to_world", C.c_float * 16), ("fov", C.c_float), ("lensradius", C.c_float),
                ("focaldistance", C.c_float), ("perspective", C.c_uint32), ("xres", C.c_uint32), ("yres", C.c_uint32),
                ("n_shapes", C.c_uint32), ("n_unsupported_shapes", C.c_uint32), ("n_unsupported_materials", C.c_uint32),
                ("n_triangle_material_conflicts", C.c_uint32), ("mis_ready", C.c_uint32), ("n_unsupported_textures", C.c_uint32),
                ("n_instances", C.c_uint32)]


class PbrtShape(C.Structure):
    _fields_ = [("kind", C.c_int32), ("shape_to_world", C.c_float * 16), ("radius", C.c_float),
                ("n_vertices", C.c_uint32), ("n_indices", C.c_uint32), ("material", C.c_int32), ("color", C.c_float * 3),
                ("emitter", C.c_int32), ("L", C.c_float * 3), ("mapped_type", C.c_int32), ("mapped_index", C.c_uint32),
                ("mapped_material", C.c_uint32), ("zmin", C.c_float), ("zmax", C.c_float), ("innerradius", C.c_float),
                ("phimax", C.c_float), ("texture", C.c_int32), ("tex2", C.c_float * 3), ("p1", C.c_float * 3), ("p2", C.c_float * 3)]


class Params(C.Structure):
    _fields_ = [("spp", C.c_uint32), ("max_depth", C.c_uint32), ("integrator", C.c_uint32),
                ("frame0", C.c_uint32), ("tile_rank", C.c_uint32), ("tile_nranks", C.c_uint32),
                ("flags", C.c_uint32), ("view_height", C.c_uint32)]


class Ray(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("tmax", C.c_float), ("direction", C.c_float * 3), ("_pad", C.c_uint32)]


class Hit(C.Structure):
    _fields_ = [("hit", C.c_int32), ("pType", C.c_int32), ("pIndex", C.c_uint32), ("t", C.c_float),
                ("p", C.c_float * 3), ("gn", C.c_float * 3), ("sn", C.c_float * 3), ("uv", C.c_float * 2),
                ("material", C.c_uint32), ("PDF", C.c_float),
                ("n_descend", C.c_uint32), ("n_return", C.c_uint32), ("n_leaf", C.c_uint32)]


class LaunchShape(C.Structure):
    """trc_launch_shape"""
    _fields_ = [("entries", C.c_uint32), ("wave_slots", C.c_uint32), ("longest_entry_ms", C.c_double),
                ("sum_entries_ms", C.c_double), ("work_over_slots_ms", C.c_double), ("clock_mhz", C.c_double)]


class Stats(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("rays", C.c_uint64), ("shaded", C.c_uint64),
                ("n_descend", C.c_uint64), ("n_return", C.c_uint64),
                ("n_leaf_sphere", C.c_uint64), ("n_leaf_square", C.c_uint64),
                ("n_leaf_cube", C.c_uint64), ("n_leaf_triangle", C.c_uint64),
                ("n_hit_triangle", C.c_uint64), ("n_hit_cube", C.c_uint64),
                ("launches", C.c_uint64), ("kernel_ms", C.c_double), ("schedule_ms", C.c_double)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


_EXPECTED_SIZES = {float2: 8, float3: 16, float4x4: 64, AABB: 32, BVH: 64, Sphere: 272, Square: 272, Cube: 240,
                   TriangleVertex: 32, TextureInfo: 32, Material: 64, Camera: 176, Ray: 32, Params: 32,
                   PhotonRecord: 80, CameraRecord: 112, Complex: 96}
for _t, _n in _EXPECTED_SIZES.items():
    assert C.sizeof(_t) == _n, f"{_t.__name__}: ctypes size {C.sizeof(_t)} != ABI size {_n}"

# every symbol include/tracer_abi.h declares, by library (checked by tests/test_abi_symbols.py)
DEVICE_SYMBOLS = [
    "trc_abi_version", "trc_build_flavor", "trc_has_test_hooks", "trc_status_string", "trc_last_error", "trc_create", "trc_destroy",
    "trc_upload_scene", "trc_upload_density", "trc_upload_scene_lbvh", "trc_upload_scene_sah", "trc_upload_scene_device", "trc_download_bvh", "trc_lbvh_info", "trc_set_camera", "trc_set_environment", "trc_set_environment_map", "trc_resize", "trc_seed",
    "trc_upload_rng", "trc_download_rng", "trc_upload_accum", "trc_download_accum", "trc_clear_accum", "trc_tonemap",
    "trc_render", "trc_synchronize", "trc_trace_rays", "trc_get_stats", "trc_reset_stats",
    "trc_sppm_init", "trc_sppm_frames", "trc_sppm_download",
    "trc_device_info", "trc_device_pci_bus_id", "trc_shard_seed", "trc_group_compose_samples", "trc_group_compose_samples_async", "trc_group_unique_id", "trc_group_init", "trc_group_reduce_accum", "trc_group_reduce_accum_async", "trc_group_allreduce_mean_accum", "trc_download_composed", "trc_group_finalize",
    "trc_group_set_collectives", "trc_debug_set", "trc_debug_block_costs", "trc_debug_launch_shape",
]
# include/tracer_test_hooks.h: exported by libtracer_amd_hooks.so only (the product's sources + -DTRC_TEST_HOOKS)
HOOK_SYMBOLS = ["trc_debug_profile", "trc_sppm_hash_cells", "trc_div_by_test", "trc_unary_test"]
HOST_SYMBOLS = [
    "trc_host_build_node", "trc_host_build_tree", "trc_host_tree_depth", "trc_host_make_camera",
    "trc_host_prepare_camera", "trc_host_fill_rng", "trc_host_scene_create", "trc_host_scene_create_leaves", "trc_host_scene_destroy",
    "trc_host_scene_view", "trc_host_scene_load_pbrt", "trc_host_mesh_load_obj", "trc_host_mesh_load_pbrt", "trc_host_mesh_load_ply", "trc_host_load_hdr", "trc_host_mesh_make_ball", "trc_host_mesh_replicate", "trc_host_mesh_from_arrays",
    "trc_host_mesh_view", "trc_host_mesh_destroy", "trc_host_make_density_info", "trc_host_make_cloud",
    "trc_host_load_density_pbrt", "trc_host_free", "trc_host_write_png", "trc_host_sobol_matrices32",
    "trc_host_sobol_interval_tables",
]
