/*
 * tracer_abi.h -- the drop-in boundary of the MI355X path-tracing hot path.
 *
 * Plain C ABI: extern "C", plain pointers and sizes, no C++/torch types.
 *
 * The reference (iaomw/Tracer, RT_Metal) has no FFI; its de-facto boundary is
 * "flat buffers of POD structs shared by host and shader + one dispatch"
 * (RT_Metal/Metal/Render.metal:495-509 kernel arguments; host side
 * RT_Metal/Tracer/AAPLRenderer.mm:702-720,1167-1196).  This header restates
 * those PODs byte-for-byte (Apple simd rules: float3 = 16 B / align 16,
 * float2 = 8 / 8, float4x4 = 64 / 16 column-major, packed_float3 = 12 / 4)
 * and declares the entry points a host would call in place of
 * `-[AAPLRenderer render:]`.  Every struct cites the reference definition
 * it mirrors; sizes/offsets are locked by static asserts below.
 *
 * Two shared libraries implement it:
 *   libtracer_amd.so  (HIP, gfx950)  -- trc_*        device path (the product)
 *   libtrc_host.so    (C++17, CPU)   -- trc_host_*   scene prep + SAH BVH builder
 */
#ifndef TRACER_ABI_H
#define TRACER_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* gcc / clang / hipcc, C and C++ alike */
#define TRC_ALIGN(n) __attribute__((aligned(n)))

/* 2: trc_trace_rays' last argument became a bit set (TRC_TRACE_*: 2 now means the production closest-hit walk, it used
 *    to mean any-hit); trc_build_flavor, trc_sppm_hash_cells, trc_host_scene_load_pbrt, trc_host_mesh_from_arrays added
 * 3: trc_group_set_collectives, trc_debug_set, trc_debug_block_costs; trc_pbrt_info / trc_pbrt_shape grew (textures,
 *    plymesh / disk / cylinder); trc_host_mesh_load_ply, trc_host_load_hdr, trc_div_by_test
 * 4: trc_debug_launch_shape, trc_unary_test; grouped trc_sppm_download of the photon records is collective; trc_stats.schedule_ms; knob no_plan_reuse; trc_debug_block_costs reports durations per SAMPLE (shader clocks / (4 spp)); knobs
 *    no_cold_probe / probe_spp (the first launch of a block list runs as an 8-sample head + the rest, trc_render)
 * 5: sample sharding with a bit-level definition: trc_shard_seed, trc_group_compose_samples[_async]; trc_collectives grew
 *    (alltoall, gather); trc_group_allreduce_mean_accum is that compose delivered to every rank (rank-ordered sum, no longer an
 *    all-reduce); trc_device_pci_bus_id; the test hooks moved to include/tracer_test_hooks.h and libtracer_amd_hooks.so
 * 6: launch lists may hold single-pixel parts (trc_debug_block_costs: bit 29; trc_launch_shape counts them as items); knob
 *    camera_policy; trc_download_composed on a non-root rank of a sample-sharded compose returns TRC_ERR_NO_FRAME (it used to
 *    hand out that rank's partial slices) */
#define TRC_ABI_VERSION 6

/* ------------------------------------------------------------------ */
/* vector / matrix PODs (Apple simd layout)                            */
/* ------------------------------------------------------------------ */

/* simd_float2: 8 bytes, align 8 (RT_Metal/Metal/Common.hh:30) */
typedef struct TRC_ALIGN(8) trc_float2 { float x, y; } trc_float2;
/* simd_float3: 16 bytes, align 16; 4th lane is padding (Common.hh:29) */
typedef struct TRC_ALIGN(16) trc_float3 { float x, y, z, _pad; } trc_float3;
/* simd_float4 (Common.hh:28) */
typedef struct TRC_ALIGN(16) trc_float4 { float x, y, z, w; } trc_float4;
/* simd_float4x4: 4 columns, column-major (Common.hh:25) */
typedef struct TRC_ALIGN(16) trc_float4x4 { trc_float4 columns[4]; } trc_float4x4;

/* ------------------------------------------------------------------ */
/* enums, with the reference's ordinals                                */
/* ------------------------------------------------------------------ */

/* RT_Metal/Metal/BVH.hh:6-8 */
enum trc_PrimitiveType {
    TRC_PRIM_SPHERE = 0, TRC_PRIM_SQUARE = 1, TRC_PRIM_CUBE = 2,
    TRC_PRIM_TRIANGLE = 3, TRC_PRIM_BVH = 4, TRC_PRIM_UNKNOW = 5
};
/* RT_Metal/Metal/Material.hh:18-20 */
enum trc_MaterialType {
    TRC_MAT_DIFFUSE = 0, TRC_MAT_LAMBERT = 1, TRC_MAT_ORENNAYAR = 2,
    TRC_MAT_PLASTIC = 3, TRC_MAT_METAL = 4, TRC_MAT_GLASS = 5,
    TRC_MAT_ISOTROPIC = 6, TRC_MAT_DIELECTRIC = 7, TRC_MAT_DEMOFOX = 8,
    TRC_MAT_PBR = 9, TRC_MAT_NIL = 10
};
/* RT_Metal/Metal/Ray.hh:6 */
enum trc_MediumType { TRC_MEDIUM_NIL = 0, TRC_MEDIUM_HOMOGENEOUS = 1, TRC_MEDIUM_GRIDDENSITY = 2 };
/* RT_Metal/Metal/Texture.hh:6 */
enum trc_TextureType { TRC_TEX_CONSTANT = 0, TRC_TEX_CHECKER = 1, TRC_TEX_NOISE = 2, TRC_TEX_IMAGE = 3 };

/* ------------------------------------------------------------------ */
/* scene PODs                                                          */
/* ------------------------------------------------------------------ */

/* RT_Metal/Metal/AABB.hh:7-9 -- 32 bytes; empty box = (+FLT_MAX, -FLT_MAX) */
typedef struct trc_AABB { trc_float3 mini, maxi; } trc_AABB;

/* RT_Metal/Metal/BVH.hh:15-22 -- 64 bytes. Array form: root at index 0,
 * leaves carry pType in {Sphere,Square,Cube,Triangle} + pIndex, interior
 * nodes pType == BVH with left/right/parent indices into the same array. */
typedef struct trc_BVH {
    uint32_t parent, left, right;
    uint32_t axis;
    int32_t  pType;              /* enum trc_PrimitiveType */
    uint32_t pIndex;
    uint32_t _pad[2];
    trc_AABB bBOX;
} trc_BVH;

/* RT_Metal/Metal/Sphere.hh:6-15 -- 272 bytes */
typedef struct trc_Sphere {
    float        radius;
    uint32_t     _pad0[3];
    trc_float3   center;
    trc_float4x4 model_matrix, normal_matrix, inverse_matrix;
    uint32_t     material;
    uint32_t     _pad1[3];
    trc_AABB     boundingBOX;
} trc_Sphere;

/* RT_Metal/Metal/Square.hh:12-27 -- 272 bytes; axis-aligned rectangle
 * spanning range_i x range_j on axes (axis_i, axis_j) at axis_k = value_k */
typedef struct trc_Square {
    uint8_t      axis_i, axis_j;
    uint8_t      _pad0[6];
    trc_float2   range_i, range_j;
    uint8_t      axis_k;
    uint8_t      _pad1[3];
    float        value_k;
    trc_float4x4 model_matrix, normal_matrix, inverse_matrix;
    uint32_t     material;
    uint32_t     _pad2[3];
    trc_AABB     boundingBOX;
} trc_Square;

/* RT_Metal/Metal/Cube.hh:6-13 -- 240 bytes; unit box under model_matrix */
typedef struct trc_Cube {
    trc_float4x4 model_matrix, normal_matrix, inverse_matrix;
    trc_AABB     box;
    uint32_t     material;
    uint32_t     _pad[3];
} trc_Cube;

/* RT_Metal/Metal/Triangle.hh:12-18 (device) == Common.hh:32-36 MeshElement
 * (host) -- 32 bytes, packed */
typedef struct trc_TriangleVertex {
    float v[3];
    float n[3];
    float uv[2];
} trc_TriangleVertex;

/* RT_Metal/Metal/Texture.hh:8-13 -- 32 bytes */
typedef struct trc_TextureInfo {
    int32_t    type;             /* enum trc_TextureType */
    uint32_t   textureIndex;
    uint32_t   _pad[2];
    trc_float3 albedo;
} trc_TextureInfo;

/* RT_Metal/Metal/Material.hh:22-40 -- 64 bytes. NOTE (reference behaviour):
 * BSDF parameters are hard-coded in MicrofacetBXDF.h create*() functions;
 * eta / roughness are carried but not read by the shading path. */
typedef struct trc_Material {
    int32_t  type;               /* enum trc_MaterialType */
    int32_t  medium;             /* enum trc_MediumType */
    uint8_t  specular;
    uint8_t  _pad0[3];
    float    eta;
    float    roughness;
    uint32_t _pad1[3];
    trc_TextureInfo textureInfo;
} trc_Material;

/* RT_Metal/Metal/Camera.hh:8-21 -- 176 bytes */
typedef struct trc_Camera {
    trc_float3 lookFrom, lookAt, viewUp;
    float vfov, aspect, aperture, lenRadius;
    float focus_dist;
    uint32_t _pad[3];
    trc_float3 u, v, w;
    trc_float3 vertical, horizontal, cornerLowLeft;
} trc_Camera;

/* RT_Metal/Metal/Camera.hh:27-55 -- 96 bytes (per-frame uniforms) */
typedef struct trc_Complex {
    trc_float2 tex_size, view_size;
    float      running_time;
    uint32_t   frame_count;
    uint32_t   _pad0[2];
    trc_AABB   photonBox;
    trc_float3 photonBoxSize;
    float      photonInitialRadius;
    float      photonHashScale;
    float      totalPhotonSum;
    uint32_t   framePhotonSum;
} trc_Complex;

/* RT_Metal/Metal/Photon.hh:12-28 -- 80 bytes (SPPM photon, persists across frames up to 8 steps) */
typedef struct trc_PhotonRecord {
    trc_float3 flux, normal, position, direction;
    uint8_t    step;
    uint8_t    active;
    uint8_t    _pad[14];
} trc_PhotonRecord;

/* RT_Metal/Metal/Photon.hh:30-53 -- 112 bytes (SPPM visible point of one pixel) */
typedef struct trc_CameraRecord {
    trc_float3 ratio, position, direction;
    uint8_t    valid;
    uint8_t    _pad0[15];
    trc_float3 alternative, flux;
    float      radius;
    uint32_t   photonCount;
    uint32_t   _pad1[2];
} trc_CameraRecord;

/* RT_Metal/Metal/Medium.hh:83-109 -- 32 bytes; describes the density grid of the GridDensity medium
 * (bound at PackageEnv ids 3/4, Render.hh:30-31) */
typedef struct trc_GridDensityInfo {
    float    sigma_a, sigma_s;
    float    sigma_t, g;
    float    invMaxDensity;
    uint32_t nx, ny, nz;
} trc_GridDensityInfo;

#define TRC_PHOTON_HASHN 512      /* PHOTON_HASHN, RT_Metal/Metal/Common.hh:4: 512x512 photons and hash cells */

/* Primitive argument table of the kernel (Render.hh:122-130) + materials of
 * PackageEnv (Render.hh:24-32), as pointers + counts.  All pointers are HOST
 * pointers; trc_upload_scene copies, the caller keeps ownership. */
typedef struct trc_scene {
    const trc_BVH*            bvhList;     uint32_t n_bvh;        /* 2*leaves-1, root at 0 */
    const trc_Sphere*         sphereList;  uint32_t n_sphere;
    const trc_Square*         squareList;  uint32_t n_square;
    const trc_Cube*           cubeList;    uint32_t n_cube;
    const trc_TriangleVertex* triList;     uint32_t n_vertex;
    const uint32_t*           idxList;     uint32_t n_index;      /* 3 per triangle */
    const trc_Material*       materials;   uint32_t n_material;
} trc_scene;

/* ------------------------------------------------------------------ */
/* status codes                                                        */
/* ------------------------------------------------------------------ */
typedef int32_t trc_status;
#define TRC_OK                  0
#define TRC_ERR_INVALID_ARG    -1
#define TRC_ERR_NO_DEVICE      -2   /* no HIP device / extension unusable: fail loudly */
#define TRC_ERR_HIP            -3
#define TRC_ERR_NO_SCENE       -4
#define TRC_ERR_NO_FRAME       -5   /* trc_resize not called */
#define TRC_ERR_BVH_INVALID    -6   /* malformed tree (bad index, depth > TRC_MAX_BVH_DEPTH, ...) */
#define TRC_ERR_UNSUPPORTED    -7
#define TRC_ERR_RCCL           -8
#define TRC_ERR_OOM            -9

#define TRC_MAX_BVH_DEPTH      64   /* reference: 32-bit stack_mark (Render.hh:140) */

/* ------------------------------------------------------------------ */
/* render parameters                                                   */
/* ------------------------------------------------------------------ */
enum trc_integrator {
    TRC_INTEGRATOR_PATH = 0,     /* tracePath, Render.metal:411-492 (the active one, :532) */
    TRC_INTEGRATOR_MIS  = 1,     /* traceMIS,  Render.metal:277-409 */
    TRC_INTEGRATOR_VOLUME = 2    /* traceVolume, Render.metal:78-275: traceMIS + participating media
                                    (Medium.hh:25-199, HitRecord.hh:37-79); SURVEY 8f-3 */
};

/* flags */
#define TRC_FLAG_FIXED_ORDER    2u  /* launch the pixel blocks in list order instead of "most expensive block of the
                                       previous launch first" (a scheduling choice only: pixels are independent) */
#define TRC_FLAG_SOBOL          4u  /* XSampler = pbrt::SobolSampler, wired as the reference's commented-out lines do
                                       (Render.metal:529-530): castRay still draws from the pixel's PCG stream; the
                                       sampler is built afterwards from a COPY of that stream, the frame number and
                                       the pixel; sample2D() (the BSDF sample of every bounce) takes successive Sobol
                                       dimensions of the pixel's sample, random() (light pick, Russian roulette)
                                       draws from the copy, and the texel keeps the stream as castRay left it.
                                       TRC_INTEGRATOR_PATH / _MIS only, 2 * max_depth <= 40 dimensions; with
                                       view_height the sampler sees the pixel and size of its own view. */
#define TRC_FLAG_SMALL_BLOCKS    8u  /* one 4x4 pixel block on 16 lanes per wavefront instead of 8x8 on 64 (a scheduling
                                       choice only: pixels are independent).  Pays when a launch has about as many
                                       blocks as the GPU has wavefront slots, i.e. a rank's share of a strong-scaled
                                       frame; chosen automatically there unless TRC_FLAG_LARGE_BLOCKS is set */
#define TRC_FLAG_LARGE_BLOCKS   16u
#define TRC_FLAG_COLLECT_STATS  1u  /* run the instrumented kernel variant: exact
                                       N_descend / N_return / leaf-test counters */

typedef struct trc_params {
    uint32_t spp;                /* samples per pixel this call; the reference does 1 per launch */
    uint32_t max_depth;          /* 8 (Render.metal:532) */
    uint32_t integrator;         /* enum trc_integrator */
    uint32_t frame0;             /* Complex.frame_count of the first sample (running-mean weight) */
    uint32_t tile_rank;          /* this call renders tiles t with trc_tile_owner(t) == tile_rank ... */
    uint32_t tile_nranks;        /* ... of tile_nranks (1 => whole frame) */
    uint32_t flags;
    uint32_t view_height;        /* 0: the frame is one view.  Otherwise the frame is a vertical stack of views of
                                    this many rows: pixel (x, y) sees the camera ray of (x, y % view_height) of a
                                    W x view_height image -- a batch of views rendered as one frame (each pixel of
                                    the stack has its own RNG texel); bench.py's multi-GPU workload */
} trc_params;

/* Pixel tiles: TRC_TILE x TRC_TILE pixels, row-major tile grid, owner of
 * tile (tx,ty) among n ranks = (tx + ty) % n. */
#define TRC_TILE 16

/* One ray / one hit of the Scene::hit test hook (Render.hh:135-252). */
typedef struct trc_ray {
    float origin[3];
    float tmax;                  /* test_t: FLT_MAX for closest hit, distance for shadow rays */
    float direction[3];          /* normalised on entry like Ray::Ray (Ray.hh:21-23) */
    uint32_t _pad;
} trc_ray;

typedef struct trc_hit {
    int32_t  hit;                /* return value of Scene::hit */
    int32_t  pType;              /* primitive type of the closest hit, -1 on miss */
    uint32_t pIndex;             /* index into its primitive list */
    float    t;
    float    p[3];
    float    gn[3];
    float    sn[3];
    float    uv[2];
    uint32_t material;
    float    PDF;
    uint32_t n_descend;          /* traversal iterations entered from the parent (Render.hh:155-187) */
    uint32_t n_return;           /* iterations entered from a child (Render.hh:189-209) */
    uint32_t n_leaf;             /* primitive hit_test calls */
} trc_hit;

/* Exact work counters of the render kernels since the last trc_reset_stats.
 * rays / paths are always counted; the n_* traversal counters only when
 * TRC_FLAG_COLLECT_STATS was set for the call. */
typedef struct trc_stats {
    uint64_t paths;              /* W*H*spp of the tiles rendered */
    uint64_t rays;               /* Scene::hit invocations (primary + bounce + shadow) */
    uint64_t shaded;             /* material lookups (S_F / F evaluations' hits) */
    uint64_t n_descend, n_return;
    uint64_t n_leaf_sphere, n_leaf_square, n_leaf_cube, n_leaf_triangle;
    uint64_t n_hit_triangle;     /* triangle tests that hit (+60 B normal/uv fetch) */
    uint64_t n_hit_cube;         /* cube tests that reach the world transform (228 B vs 100 B) */
    uint64_t launches;           /* render kernel launches */
    double   kernel_ms;          /* sum of hipEvent durations of those launches, on the ctx stream */
    double   schedule_ms;        /* ... and of the launch-list kernels in front of them (order, sort, split plan); not part of kernel_ms */
} trc_stats;

/* Threading contract.  A context owns its device buffers, one render stream (+ a communication stream and the SPPM
 * camera stream) and all per-launch state; nothing is shared between contexts except the process-wide RCCL symbol table,
 * which is resolved once under a lock.  Therefore:
 *   - calls on ONE context must not overlap: a context is used by one thread at a time (any thread -- the reference's
 *     completion handlers arrive off the main thread, AAPLRenderer.mm:1148-1150 -- every entry point selects the
 *     context's device itself);
 *   - DIFFERENT contexts, on the same GPU or on different GPUs, may be driven from different threads concurrently;
 *   - trc_* calls are asynchronous on the context's stream unless they hand host memory back (downloads, trc_get_stats,
 *     trc_trace_rays, trc_tonemap, trc_synchronize), which wait for the stream.
 * tests/test_gpu_multicontext.py drives two contexts interleaved and from two threads. */
typedef struct trc_ctx trc_ctx;

/* ------------------------------------------------------------------ */
/* device path: libtracer_amd.so                                       */
/* ------------------------------------------------------------------ */

uint32_t    trc_abi_version(void);
/* "exact" = libtracer_amd.so: IEEE division / sqrt, no FMA contraction -- the build every parity statement is about;
 * "fast-math" = libtracer_amd_fast.so: the same sources under fast-math rules, as the reference compiles its shaders
 * (MTL_FAST_MATH): approximate division / sqrt, FMA contraction, denormals flushed; results agree with the exact build
 * statistically, not bit for bit */
const char* trc_build_flavor(void);
/* 1 in libtracer_amd_hooks.so -- the same sources compiled with -DTRC_TEST_HOOKS, which additionally exports the
 * entry points of include/tracer_test_hooks.h (exhaustive arithmetic checks, the SPPM hash, the per-site cycle profile) for
 * tests/ and tools/; 0 in the product libraries, which do not carry them */
int         trc_has_test_hooks(void);
const char* trc_status_string(trc_status s);
/* last error text of this context (HIP error string etc.), never NULL */
const char* trc_last_error(const trc_ctx* ctx);

/* replaces device/queue/pipeline creation, AAPLRenderer.mm:129-181 */
trc_status trc_create(int device, trc_ctx** out);
void       trc_destroy(trc_ctx* ctx);

/* replaces buffer creation + heap copy, AAPLRenderer.mm:213-246,610-720,1233-1355 */
trc_status trc_upload_scene(trc_ctx* ctx, const trc_scene* scene);

/* On-device LBVH build + node repack.  The reference builds its BVH on the host
 * (BVH::buildTree, RT_Metal/Metal/BVH.hh:35-269) and lists "LBVHs, Morton Encoding" as a to-do
 * (RT_Metal/README.md:42); this entry point replaces BVH::buildTree for scenes where the host SAH build is the
 * bottleneck.  `scene->bvhList` holds ONLY the n_bvh LEAF records as BVH::buildNode writes them
 * (BVH.hh:273-314: bBOX, pType, pIndex); the hierarchy is built on the GPU (30-bit Morton codes of the box
 * centroids, stable radix sort, Karras' binary radix tree, bottom-up boxes) and used by every later
 * trc_render / trc_trace_rays exactly like an uploaded tree. */
trc_status trc_upload_scene_lbvh(trc_ctx* ctx, const trc_scene* scene);
/* BVH::buildTree itself on the device (RT_Metal/Metal/BVH.hh:35-269; the host builder of this package is
 * trc_host_build_tree): same input as trc_upload_scene_lbvh -- the n_bvh LEAF records -- and the tree the reference's
 * 10-bucket SAH recursion builds from them, record for record (post-order interior numbering, the partition's own leaf
 * order, median split of identical centroids, pairs ordered by centroid): rendering through it is rendering through the
 * host-built tree.  Leaf boxes must be finite with |coordinates| <= 1e37 (TRC_ERR_BVH_INVALID otherwise).
 * trc_download_bvh / trc_lbvh_info report on it like on an LBVH tree. */
trc_status trc_upload_scene_sah(trc_ctx* ctx, const trc_scene* scene);
/* Both device builds behind one call.  flags: TRC_TREE_SAH (else the LBVH); TRC_TREE_TRIANGLE_LEAVES: `scene->bvhList` holds the leaf
 * records of the analytic primitives only (n_bvh of them, possibly none) and one leaf per triangle of (idxList, triList) follows them
 * in index order, written on the device exactly as the reference's loop writes it (AAPLRenderer.mm:575-589: the box of the three
 * vertices through BVH::buildNode under the identity matrix) -- a host with a mesh of a million triangles neither loops over them
 * nor uploads 64 bytes per triangle.  The tree is the one trc_host_build_tree builds from the same leaves. */
#define TRC_TREE_SAH              1u
#define TRC_TREE_TRIANGLE_LEAVES  2u
trc_status trc_upload_scene_device(trc_ctx* ctx, const trc_scene* scene, uint32_t flags);
/* the device-built tree in the reference's array layout (BVH.hh:246-269): [root, leaf 0..n-1, interior
 * 1..n-2], 2n-1 records.  out == NULL: only *n_nodes is written. */
trc_status trc_download_bvh(trc_ctx* ctx, trc_BVH* out, uint32_t capacity, uint32_t* n_nodes);
/* node count, depth of the deepest leaf and GPU time of the last trc_upload_scene_lbvh (bounds -> emit) */
trc_status trc_lbvh_info(trc_ctx* ctx, uint32_t* n_nodes, uint32_t* height, float* device_build_ms);
/* replaces memcpy(_camera_buffer.contents, ...), AAPLRenderer.mm:1183 */
/* replaces _densityInfoBuffer / _densityDataBuffer (AAPLRenderer.mm:629-643, bound at :716-720): the
 * nx*ny*nz density grid of the GridDensity medium, x fastest (Medium.hh:121).  Consulted by
 * TRC_INTEGRATOR_VOLUME when a ray travels in a medium of type TRC_MEDIUM_GRIDDENSITY; both pointers
 * NULL clear it. */
trc_status trc_upload_density(trc_ctx* ctx, const trc_GridDensityInfo* info, const float* density);
trc_status trc_set_camera(trc_ctx* ctx, const trc_Camera* camera);
/* constant environment radiance used on a miss; stands in for
 * packageEnv.texHDR.sample (Render.metal:434-439; the HDR blob is missing) */
trc_status trc_set_environment(trc_ctx* ctx, const float rgb[3]);

/* (re)allocates accum A (RGBA32F) + RNG (RGBA32Uint) for a W x H frame and
 * zeroes them; replaces texture creation, AAPLRenderer.mm:260-288 */
/* texHDR (Render.hh:25; the reference's HDR blob is missing from its repository): equirectangular RGB float image,
 * 3*w*h floats, row 0 at v = 0.  A ray that leaves the scene returns SampleSphericalMap (Render.hh:42-48) + a bilinear,
 * clamp-to-edge lookup (Common.hh:11) instead of the constant of trc_set_environment; rgb == NULL clears the map. */
trc_status trc_set_environment_map(trc_ctx* ctx, uint32_t w, uint32_t h, const float* rgb);
trc_status trc_resize(trc_ctx* ctx, uint32_t width, uint32_t height);
/* deterministic stand-in for fillRNG (AAPLRenderer.mm:296-344, arc4random):
 * texel(x,y) = 4 successive pcg32 outputs of pcg32_srandom_r(seed, y*W+x),
 * generated on the device; identical to trc_host_fill_rng. */
trc_status trc_seed(trc_ctx* ctx, uint64_t seed);
trc_status trc_upload_rng(trc_ctx* ctx, const uint32_t* rgba /* 4*W*H */);
trc_status trc_download_rng(trc_ctx* ctx, uint32_t* rgba /* 4*W*H */);
trc_status trc_upload_accum(trc_ctx* ctx, const float* rgba /* 4*W*H */);
/* linear radiance running mean, same meaning as textureA/B (Render.metal:540-543) */
trc_status trc_download_accum(trc_ctx* ctx, float* rgba /* 4*W*H */);
trc_status trc_clear_accum(trc_ctx* ctx);

/* Output stage = what fragmentShader does to the accumulator before display (Render.metal:29-75):
 * auto-exposure from the frame's mean colour (the top mip level there; an exact mean here: per-channel sums in
 * 2^-16 fixed point, so the result does not depend on summation order), luma = dot(mean, (0.2126, 0.7152, 0.0722)),
 * expose = 1 - clamp(CETone(luma, 1), 0, 0.9999) (Render.hh:91-95), ACESTone(rgb, expose) (Render.hh:78-89),
 * no sRGB curve (commented out at :73), 8-bit = (uint8)(clamp(x, 0, 1) * 255 + 0.5), alpha 255.  Rows are
 * written top-down (the shader flips v, :51-56): out row 0 = frame row H-1.  The reference feeds its SVGF-denoised
 * texture here; this takes the raw accumulator.  rgba8: host buffer of 4*W*H bytes; exposure_out may be NULL. */
trc_status trc_tonemap(trc_ctx* ctx, uint8_t* rgba8, float* exposure_out);

/* replaces -[AAPLRenderer render:] + kernelPathTracing dispatch
 * (AAPLRenderer.mm:1134-1196, Render.metal:495-558); asynchronous on the
 * context stream; spp samples are fused into one launch with bit-identical
 * results to spp launches of 1 sample.
 * Coalesced launches: a call of fewer than 8 samples (without TRC_FLAG_COLLECT_STATS) is not launched at once but kept, and
 * extended by following calls that continue it (identical parameters, frame0 = where it ends); it is launched when 16 samples
 * have come together, when a call arrives that does not continue it, or when ANY other trc_* entry point taking this context is
 * entered (each launches what was kept before it does anything else).  Results, statistics (trc_stats.launches counts
 * calls) and error behaviour of parameter checks are those of launching every call at once; a HIP error of a kept launch
 * is returned by the call that launches it, and trc_last_error then names the frame range (frame0 .. frame0 + spp - 1) and the
 * number of trc_render calls whose samples did not run, so that a host can re-issue exactly those.  trc_destroy drops a kept
 * launch without launching it (nobody could read its frame); call trc_synchronize first to learn its status.  Knob "no_coalesce"
 * (trc_debug_set) launches every call at once.
 * First launch of a block list (new context, frame size, tile share, scene, camera, integrator), >= 16 samples: run as a
 * head of 8 samples (then passes of 16 and 32 samples where at least four times as many remain) followed by the rest, each
 * pass ordered and split by its predecessor's per-block durations -- the same pixels (a pixel's samples are one chain
 * through its RNG texel); knob "no_cold_probe" runs it as one pass, "head_stages" = n caps the passes before the rest. */
trc_status trc_render(trc_ctx* ctx, const trc_params* params);
trc_status trc_synchronize(trc_ctx* ctx);

/* test hook = Scene::hit (Render.hh:135-252) on a batch of rays (host buffers).  `any_hit` is a bit set: */
#define TRC_TRACE_ANY_HIT     1   /* stop at the first accepted hit closer than tmax (shadow rays, Render.hh:244) */
#define TRC_TRACE_PRODUCTION  2   /* walk the tree with the round the render kernels run (no instrumentation; on trees
                                     read from global memory the wavefront leaves the descent early, dev_intersect.hpp):
                                     same hit record, n_descend / n_return / n_leaf left 0.  With TRC_TRACE_ANY_HIT it
                                     is the order-free walk the render kernels give shadow rays: an any-hit query only
                                     asks WHETHER something lies in front of tmax, which does not depend on the order
                                     the tree is walked in, but which primitive is met first does -- only `hit` is
                                     defined then (pType -1, the other fields 0) */
trc_status trc_trace_rays(trc_ctx* ctx, const trc_ray* rays, size_t n, trc_hit* out, int any_hit);

trc_status trc_get_stats(trc_ctx* ctx, trc_stats* out);   /* synchronises the stream */
trc_status trc_reset_stats(trc_ctx* ctx);
/* developer diagnostic: the pixel blocks of the last trc_render (x | y << 16, in units of the block edge 1 << *blk_shift)
 * and the duration each one's wavefront measured per sample (shader clocks / (4 spp) -- the sort key of the adaptive launch order); with
 * strips (spp < 8) the costs are per strip; a block that ran in parts (four 4x4 quarters, some of them as four 2x2
 * sixteenths, some of those as four single pixels) reports its slowest part with bit 31 set, bit 30 when it had sixteenths and bit 29 when it
 * had single pixels.  Any pointer may be NULL; at most `capacity` entries are written. */
trc_status trc_debug_block_costs(trc_ctx* ctx, uint32_t* tiles, uint32_t* costs, uint32_t capacity, uint32_t* n_blocks, uint32_t* blk_shift);
/* developer diagnostic: the two lower bounds of the last trc_render's kernel time that no schedule can beat.  A pixel's samples
 * are one chain through its RNG texel (Render.metal:545-557), so a launch lasts at least as long as its slowest wavefront-sized
 * item (a whole 8x8 block, or a 4x4 / 2x2 / single-pixel part of one) -- the bound a strong-scaled share of a frame runs into -- and at least
 * the items' summed durations over the wavefront slots of the GPU.  Durations are the wavefronts' own measurements (shader
 * clocks), converted with the device's nominal shader clock. */
typedef struct trc_launch_shape {
    uint32_t entries;             /* wavefront-sized items of the launch: whole blocks + parts */
    uint32_t wave_slots;          /* wavefronts the GPU holds at once for the kernel that ran */
    double   longest_entry_ms;    /* the slowest item */
    double   sum_entries_ms;      /* all items: slot time */
    double   work_over_slots_ms;  /* sum_entries_ms / wave_slots */
    double   clock_mhz;           /* hipDeviceAttributeClockRate, used for the conversion */
} trc_launch_shape;
trc_status trc_debug_launch_shape(trc_ctx* ctx, trc_launch_shape* out);

/* device info for the bench line */
trc_status trc_device_info(trc_ctx* ctx, char* name, size_t name_len, int* cu_count, size_t* hbm_bytes);
/* which physical GPU this context sits on: "domain:bus:device.function" (hipDeviceGetPCIBusId).  Ranks exchange it to learn
 * whether they hold DIFFERENT devices (then RCCL composes) or share one (RCCL refuses a second rank on a device: the
 * collectives table composes) -- counting HIP_VISIBLE_DEVICES entries cannot tell, a launcher may give every rank its own */
trc_status trc_device_pci_bus_id(trc_ctx* ctx, char* out, size_t out_len);

/* --- SPPM pass (RT_Metal/Metal/Photon.metal, host sequencing AAPLRenderer.mm:860-1086) ------------
 * Uses the scene / camera / frame (canvas RNG + accumulator) of the context. */
/* allocates the 512x512 photon records, photon RNG texture (seeded like trc_seed), per-pixel camera
 * records and the hash grids; resets Complex (frame_count = 0, totalPhotonSum = 0) */
trc_status trc_sppm_init(trc_ctx* ctx, uint64_t photon_seed);
/* n_frames x `photon:` (AAPLRenderer.mm:1077-1086): frame 0 runs photonPrepare (camera records, their
 * bounding box, hash scale, initial radius), every frame photonWork (odd frames re-run the camera pass,
 * photon bounce, hashing, mark/count grid, photon sum, progressive refine into the accumulator).
 * In a group (trc_group_init / trc_group_set_collectives) EVERY rank calls it: rank r traces and refines the pixels of its tiles and
 * bounces its share of the 512*512 photons (whole wavefronts of 64, the last non-empty share shorter when the ranks do not divide them;
 * at most 64 ranks), the bound of the visible points
 * is all-reduced and the photon records all-gathered each frame; a rank that owns no tile of a small frame still takes part.
 * Two corners are fixed as the reference's arithmetic has them: a frame in which no pixel records a visible point ends with the
 * bound {FLT_MAX, -FLT_MAX}, i.e. radius -inf and hash scale -0 (Photon.metal:157-161,357-372); a photon whose BSDF sample is NaN
 * counts the NaN as positive where copysign(1, wi.z) picks the side of the surface, and its stored direction / flux NaNs are the
 * canonical 0x7FC00000 (the sign of a NaN is the platform's: DESIGN.md 2). */
trc_status trc_sppm_frames(trc_ctx* ctx, uint32_t n_frames);
/* any pointer may be NULL.  mark: 4 floats per cell as texturePhotonMark (x, y of the winning photon,
 * cell x, y; -1 when empty), count: 1 float per cell as texturePhotonCount.
 * With a communicator / collectives table and photon_records != NULL the call is COLLECTIVE (every rank makes it): the
 * per-frame exchange moves only the 40 bytes of a photon the hash / table / refine passes read, the whole 80-byte records
 * of the other ranks' photons are all-gathered here, on demand. */
trc_status trc_sppm_download(trc_ctx* ctx, trc_CameraRecord* camera_records /* W*H */,
                             trc_PhotonRecord* photon_records /* 512*512 */, float* mark /* 512*512*4 */,
                             float* count /* 512*512 */, trc_Complex* complex);

/* --- multi-GPU: pixel tiles sharded over ranks, one RCCL reduce -------- */
#define TRC_UNIQUE_ID_BYTES 128
/* rank 0 creates the id, every rank gets the same bytes out-of-band */
trc_status trc_group_unique_id(uint8_t id[TRC_UNIQUE_ID_BYTES]);
trc_status trc_group_init(trc_ctx* ctx, const uint8_t id[TRC_UNIQUE_ID_BYTES], int nranks, int rank);
/* ncclReduce(sum) of the full-frame accum buffer to `root` on the ctx stream:
 * every rank holds zeros outside its own tiles, so sum == gather */
trc_status trc_group_reduce_accum(trc_ctx* ctx, int root);
/* --- sample sharding: the split that scales (SURVEY 8e "alternative") -------------------------------------------
 * A pixel's samples are ONE chain through its RNG texel (Render.metal:511-519,545-557), so a rank that owns a share of
 * the TILES still runs every owned pixel's whole chain: its launch ends on its slowest 8x8 block however few blocks it
 * owns.  Splitting the SAMPLES shortens the chains instead.  Definition (bit-level; oracle/pyoracle.py::render_sample_sharded
 * restates it and the GPU tests hold the composed frame to it bit for bit):
 *   - the nranks ranks are S sample groups x T tile ranks, nranks == S * T; rank r is tile rank r % T of group g = r / T;
 *   - group g renders the WHOLE frame (its T ranks share the tiles as usual: trc_params.tile_rank = r % T, tile_nranks = T,
 *     zeros elsewhere) with spp / S samples per pixel, frame0 counting from 0 within the group, from the RNG texture
 *     trc_seed(trc_shard_seed(seed, g)) -- group 0 keeps the seed, so S == 1 is the unsharded frame;
 *   - composed pixel = (((A_0 + A_1) + A_2) + ... + A_{nranks-1}) / (float)S per channel in binary32, A_r = rank r's
 *     accumulator texel: a rank-ORDERED sum (zeros of the other tile ranks are exact identities), one IEEE division.
 * Every group must have rendered the same number of samples (the mean of running means is the running mean only then).
 * Defined for the PCG sampler: under TRC_FLAG_SOBOL a sample's Sobol' point is a function of (frame, pixel) alone, so groups that all
 * count their frames from 0 draw the SAME points and differ only in the PCG draws (light pick, Russian roulette) -- legal, bit-defined
 * the same way, and statistically worth little more than one group.
 * How it moves: each rank owns the r-th of nranks equal pixel slices; an all-to-all brings that slice of every rank's
 * accumulator (ncclSend / ncclRecv in one group call; at N = 8 and 1080p 29 MB in and out per rank, every pair on its own
 * xGMI link), a small kernel folds them in rank order, and a gather (root >= 0) brings the N slices (4 MB each) to the
 * root.  The ranks' accumulators are left untouched -- a progressive host goes on rendering and composes again. */
uint64_t   trc_shard_seed(uint64_t seed, uint32_t sample_group);   /* seed + sample_group * 0x9E3779B97F4A7C15 (mod 2^64) */
/* on the context stream; the composed frame is read with trc_download_composed (root only).  sample_groups == 0: nranks
 * (every rank its own group, no tile split) */
trc_status trc_group_compose_samples(trc_ctx* ctx, int root, uint32_t sample_groups);
/* pipelined: the exchange runs on the communication stream once the work queued so far has finished, on a SNAPSHOT of the
 * accumulator taken in render-stream order (a device copy of the frame, ~20 us at 1080p) -- the compose only reads, so unlike
 * trc_group_reduce_accum_async the context keeps its accumulator: the next trc_render overlaps with the exchange AND may go on
 * accumulating (frame0 continuing); a progressive host calls this after every trc_render and never clears.
 * trc_download_composed (root) waits for the exchange */
trc_status trc_group_compose_samples_async(trc_ctx* ctx, int root, uint32_t sample_groups);
/* the same compose with sample_groups = nranks, delivered into EVERY rank's accumulator (all-gather instead of the gather).
 * Until ABI 4 this was ncclAllReduce(sum) / nranks, whose sum order is the ring's; it is the rank-ordered sum now. */
trc_status trc_group_allreduce_mean_accum(trc_ctx* ctx);
/* same compose, pipelined: the reduce runs on a second stream as soon as the work queued so far has finished,
 * and the context switches to its OTHER accumulator (allocated on first use, zero-filled), so the next
 * trc_clear_accum / trc_render overlap with the collective.  The composed frame of the call is read with
 * trc_download_composed (root only); trc_synchronize waits for the collectives too. */
trc_status trc_group_reduce_accum_async(trc_ctx* ctx, int root);
trc_status trc_download_composed(trc_ctx* ctx, float* rgba /* 4*W*H */);
trc_status trc_group_finalize(trc_ctx* ctx);

/* The collectives behind every trc_group_* / grouped trc_sppm_frames call, as a table.  RCCL (trc_group_init) is the
 * default; a host that has its own transport -- MPI, a socket mesh, gloo in the Python harness -- or that runs several
 * ranks on ONE GPU (where RCCL refuses a second rank on a device) installs its own with trc_group_set_collectives and
 * needs no communicator.  The program of collectives is the same either way:
 *   compose          reduce(sum, f32, 4*W*H) of the accumulator to the root           (trc_group_reduce_accum[_async])
 *   sample sharding  alltoall of the accumulator's nranks pixel slices, then gather of the composed slices to the root
 *                    (or allgather to every rank)            (trc_group_compose_samples[_async], trc_group_allreduce_mean_accum)
 *   SPPM frame 0     allreduce(min, u32, 3) + allreduce(max, u32, 3) of the bound keys (Photon.metal:169-218's reduction)
 *   SPPM every frame allgather of the photon records, 512*512/N * 80 bytes per rank    (so that kernelPhotonSumming,
 *                    Photon.metal:458-496, sees every photon on every rank)
 * Every function works IN PLACE on `buf`, is called by all ranks in the same order, returns 0 on success, and must
 * be complete (result visible to work queued later on `stream`) in stream order:
 *   host_staged = 0: `buf` is DEVICE memory and `stream` the hipStream_t the surrounding work is queued on (enqueue
 *                    on it, or synchronise it);
 *   host_staged = 1: the library waits for the stream, copies the buffer into pinned HOST memory, calls the function
 *                    with that host pointer (stream = NULL; it may block), and copies the result back -- slow,
 *                    meant for tests, 1-GPU plumbing runs and hosts without a GPU-aware transport.
 * reduce: the result is defined on `root` only.  allgather: rank r's contribution sits at buf + r * bytes_per_rank on
 * entry, all of them on return.  alltoall: `buf` holds nranks slices of bytes_per_rank; on return slice p holds what rank
 * p had in ITS slice r (r = the caller's rank; slice r stays).  gather: like allgather, the result is defined on `root`
 * only.  alltoall / gather may be NULL in a table whose host never composes sample shards (TRC_ERR_UNSUPPORTED then).
 * dtype / op use ncclDataType_t / ncclRedOp_t ordinals. */
enum trc_coll_dtype { TRC_DT_U8 = 1, TRC_DT_U32 = 3, TRC_DT_F32 = 7 };
enum trc_coll_op { TRC_OP_SUM = 0, TRC_OP_MAX = 2, TRC_OP_MIN = 3 };
typedef struct trc_collectives {
    void* user;
    int32_t host_staged;
    int32_t _pad;
    int (*reduce)(void* user, void* buf, size_t count, int dtype, int op, int root, void* stream);
    int (*allreduce)(void* user, void* buf, size_t count, int dtype, int op, void* stream);
    int (*allgather)(void* user, void* buf, size_t bytes_per_rank, void* stream);
    int (*alltoall)(void* user, void* buf, size_t bytes_per_rank, void* stream);
    int (*gather)(void* user, void* buf, size_t bytes_per_rank, int root, void* stream);
} trc_collectives;
/* installs `table` (copied) and makes the context rank `rank` of `nranks`; replaces a communicator made by
 * trc_group_init.  trc_group_finalize removes it.  table == NULL: same as trc_group_finalize. */
trc_status trc_group_set_collectives(trc_ctx* ctx, const trc_collectives* table, int nranks, int rank);

/* A/B and test knobs of ONE context (the defaults come from the environment variables of the same upper-case names at
 * trc_create): "no_lds_fit", "stack_lds_levels", "strip_len", "no_pwg", "sppm_serial_camera" (tools/README.md), and
 * "no_split" (no cost-adaptive block size), "no_cost_filter" (launch order and plan from the last launch's durations instead of
 * the shortest seen lately), "force_blk_shift" (k + 1 forces 2^k x 2^k pixel blocks per wavefront, k = 0..3: measurement only), and
 * "sppm_timing" (event pairs around an SPPM frame's photon pass and its hash / table / refine passes, added to
 * trc_stats.kernel_ms; one `launch` per frame), "descend_min" (n > 0: on trees read from memory the box-step loop of a wavefront goes on
 * while at least n lanes are still descending and others wait with a leaf; 0 = the scene's own value, 12, or 6 beyond 256 MiB), "camera_policy" (what trc_set_camera does with the recorded block costs: 0 = keeps them when the camera moved a little -- the view turned by
 * at most 5 degrees and the eye moved by at most 5 % of the scene's diagonal: the next launch is one pass ordered by the last launch's raw durations -- and forgets them
 * otherwise -- the next launch runs as a head + the rest; 1 always forgets, 2 always keeps the filtered costs, 3 always keeps and takes the raw durations: tools/moving_camera.py).  They change scheduling / bookkeeping only, never a pixel.  Unknown name: TRC_ERR_INVALID_ARG. */
trc_status trc_debug_set(trc_ctx* ctx, const char* knob, int value);

/* ------------------------------------------------------------------ */
/* host path: libtrc_host.so (CPU only)                                */
/* ------------------------------------------------------------------ */

/* BVH::buildNode, RT_Metal/Metal/BVH.hh:273-314: world AABB of the 8 corners
 * of `box` under model_matrix -> one leaf record */
void trc_host_build_node(const trc_AABB* box, const trc_float4x4* model_matrix,
                         int32_t pType, uint32_t pIndex, trc_BVH* out_leaf);
/* BVH::buildTree + BVH::make, BVH.hh:35-269: 10-bucket SAH over the n_leaves
 * leaf records at nodes[0..n_leaves); writes 2*n_leaves-1 nodes (root moved to
 * index 0); `nodes` must have room for 2*n_leaves-1.  Serial post-order
 * interior numbering (one legal schedule of the reference's GCD build). */
trc_status trc_host_build_tree(trc_BVH* nodes, uint32_t n_leaves, uint32_t* out_n_nodes);
/* depth of the deepest leaf (root = 0); TRC_ERR_BVH_INVALID on malformed trees */
trc_status trc_host_tree_depth(const trc_BVH* nodes, uint32_t n_nodes, uint32_t* out_depth);

/* MakeCamera / prepareCamera defaults, Tracer.mm:87-125,371-411 */
void trc_host_make_camera(trc_Camera* out, const float lookFrom[3], const float lookAt[3],
                          const float viewUp[3], float aperture, float aspect,
                          float vfov_radians, float focus_dist);
void trc_host_prepare_camera(trc_Camera* out, float width, float height);

/* deterministic fillRNG stand-in (see trc_seed) */
void trc_host_fill_rng(uint64_t seed, uint32_t width, uint32_t height, uint32_t* rgba);

/* Scene assembly in the reference's order (AAPLRenderer.mm:213-246,459-468,
 * 513-610): prepareCubeList (materials 0-2), prepareCornellBox (3-6),
 * prepareSphereList (7-18), testMaterial (19); leaves = cubes 0..n-2, all
 * squares, [spheres], [mesh triangles]; then buildTree. */
typedef struct trc_host_scene trc_host_scene;

enum trc_host_scene_kind {
    TRC_SCENE_CORNELL          = 0,  /* as shipped: 2 cubes + 7 squares (spheres not in the BVH) */
    TRC_SCENE_CORNELL_SPHERES  = 1,  /* BASELINE config 2: + the 12 spheres, materials remapped */
    TRC_SCENE_CORNELL_MESH     = 2,  /* + a triangle mesh placed by the reference transform */
    TRC_SCENE_CORNELL_VOLUME   = 3   /* kind 0 + the third cube of prepareCubeList (Tracer.mm:221-243: material _NIL_,
                                        medium GridDensity -- the cloud container the reference keeps out of
                                        its BVH, AAPLRenderer.mm:459) as a leaf; optional mesh as in kind 2 */
};

/* mesh: optional (NULL for kinds 0/1); positions/normals/uvs as
 * trc_TriangleVertex + triangle indices in OBJECT space; placement
 * (AAPLRenderer.mm:513-525,562-572) is applied by the call. */
trc_status trc_host_scene_create(int32_t kind,
                                 const trc_TriangleVertex* mesh_vertices, uint32_t n_vertices,
                                 const uint32_t* mesh_indices, uint32_t n_indices,
                                 trc_host_scene** out);
/* analytic_leaves_only != 0: the same scene without the per-triangle leaf records and without the tree -- bvhList = the leaf
 * records of the analytic primitives, the input of trc_upload_scene_device(TRC_TREE_SAH | TRC_TREE_TRIANGLE_LEAVES), which
 * writes the triangles' leaves and builds the reference's tree on the GPU */
trc_status trc_host_scene_create_leaves(int32_t kind,
                                        const trc_TriangleVertex* mesh_vertices, uint32_t n_vertices,
                                        const uint32_t* mesh_indices, uint32_t n_indices,
                                        int32_t analytic_leaves_only, trc_host_scene** out);
void       trc_host_scene_destroy(trc_host_scene* s);
/* view of the assembled arrays (valid until destroy) */
void       trc_host_scene_view(const trc_host_scene* s, trc_scene* out);

/* GridDensityInfo::GridDensityInfo (Medium.hh:92-105): sigma_t = sigma_a + sigma_s, invMaxDensity = 1 / max */
void trc_host_make_density_info(float sigma_a, float sigma_s, float g, uint32_t nx, uint32_t ny, uint32_t nz,
                                const float* density, trc_GridDensityInfo* out);
/* procedural stand-in for cloud/geometry/density_render.70.pbrt (100x100x40, does not travel to the GPU box):
 * a few smooth blobs, values in [0, ~2], deterministic in `seed` */
void trc_host_make_cloud(uint32_t nx, uint32_t ny, uint32_t nz, uint32_t seed, float* out /* nx*ny*nz */);
/* reads the `MakeNamedMedium ... "integer nx" .. "float density" [ ... ]` block of a pbrt-v3 file (the subset
 * of minipbrt the reference uses, AAPLRenderer.mm:629-636); *out is malloc'ed, free with trc_host_free */
trc_status trc_host_load_density_pbrt(const char* path, uint32_t* nx, uint32_t* ny, uint32_t* nz, float** out);
void trc_host_free(void* p);

/* Radiance RGBE (.hdr) image -> float RGB, rows BOTTOM-UP (what trc_set_environment_map takes, and what the reference's
 * vertically flipped HDR texture holds: vulture_hide_4k.hdr, AAPLRenderer.mm:352-383, sampled through Render.hh:42-48);
 * flat and run-length-encoded scanlines; *rgb is malloc'ed (3 * w * h floats), free with trc_host_free */
trc_status trc_host_load_hdr(const char* path, uint32_t* width, uint32_t* height, float** rgb);

/* "Export as PNG file" (the reference's unchecked to-do, RT_Metal/README.md:61): 8-bit RGBA, rows top-down,
 * stored (uncompressed) deflate blocks -- no zlib dependency */
trc_status trc_host_write_png(const char* path, const uint8_t* rgba8, uint32_t width, uint32_t height);

/* Tables of pbrt::SobolSampler (SobolSampler.hh:126-160), generated by include/trc_sobol.h from the published
 * direction numbers instead of being stored: SobolMatrices32 for the first TRC_SOBOL_DIMS (40) dimensions
 * (out[dim * 52 + column]) and VdCSobolMatrices[log2res - 1] / VdCSobolMatricesInv[log2res - 1] (52 words each,
 * unused entries 0; 1 <= log2res <= 26).  libtracer_amd.so builds the same tables itself for TRC_FLAG_SOBOL. */
void       trc_host_sobol_matrices32(uint32_t* out /* [40 * 52] */);
trc_status trc_host_sobol_interval_tables(uint32_t log2res, uint64_t* vdc /* [52] */, uint64_t* inv /* [52] */);

/* minimal Wavefront OBJ reader (v / vn / vt / f; polygons fan-triangulated;
 * smooth normals generated when the file has none), standing in for ModelIO
 * (AAPLRenderer.mm:474-511).  Returns arrays owned by the mesh handle. */
typedef struct trc_host_mesh trc_host_mesh;
trc_status trc_host_mesh_load_obj(const char* path, trc_host_mesh** out);
/* every `Shape "trianglemesh"` of a pbrt-v3 file (the reference's unchecked to-do "Support pbrt-v3 file format",
 * RT_Metal/README.md:57; it vendors minipbrt for it, Tracer/minipbrt.h:1528-1546): points through the current
 * transformation matrix, normals through its inverse transpose (smooth normals when absent), uv / st, Include
 * followed; shapes inside ObjectBegin/ObjectEnd and other shape types are skipped */
trc_status trc_host_mesh_load_pbrt(const char* path, trc_host_mesh** out);
/* the triangles of a PLY file (ascii / binary_little_endian / binary_big_endian; x y z, optional nx ny nz and u v | s t;
 * triangles as they are, quads split (0 1 3) (2 3 1), larger polygons fanned) -- what `Shape "plymesh"` of a pbrt-v3 scene
 * refers to and what minipbrt's PLYMesh::triangle_mesh() returns (minipbrt.cpp:4380-4450) */
trc_status trc_host_mesh_load_ply(const char* path, trc_host_mesh** out);
/* A whole scene from a pbrt-v3 file -- the reference's unchecked to-do "Support pbrt-v3 file format"
 * (RT_Metal/README.md:57, the vendored parser RT_Metal/Tracer/minipbrt.h:1528-1546, its one call site
 * AAPLRenderer.mm:626-651): Camera "perspective" + LookAt, Film resolution, AreaLightSource "diffuse", Material
 * matte / plastic / metal / mirror / glass (+ MakeNamedMaterial / NamedMaterial), Texture "checkerboard", Shape "sphere",
 * "trianglemesh", "plymesh" (the PLY file is read: ascii / binary, triangles and quads), "disk" and "cylinder"
 * (tessellated) through the transformation and attribute stacks, Include.  Mapping onto the reference's primitives
 * (tracer_amd/host/pbrt_scene.cpp): sphere -> trc_Sphere; a trianglemesh that is an axis-aligned rectangle ->
 * trc_Square, emitters placed at squareList[5] / [6] (the two lights traceMIS samples); any other mesh -> triangles with
 * material 19.
 * `info` / `shapes` (optional, up to `capacity` entries in file order) describe what was parsed and what it became;
 * the scene handle is used like one from trc_host_scene_create. */
enum trc_pbrt_material { TRC_PBRT_MATTE = 0, TRC_PBRT_PLASTIC = 1, TRC_PBRT_METAL = 2, TRC_PBRT_MIRROR = 3,
                         TRC_PBRT_GLASS = 4, TRC_PBRT_OTHER = 5 };
/* what the file's Shape directive was (sphere / trianglemesh keep the primitive-type ordinals they map to) */
enum trc_pbrt_shape_kind { TRC_PBRT_SHAPE_SPHERE = 0, TRC_PBRT_SHAPE_TRIANGLEMESH = 3, TRC_PBRT_SHAPE_DISK = 6,
                           TRC_PBRT_SHAPE_CYLINDER = 7, TRC_PBRT_SHAPE_PLYMESH = 8, TRC_PBRT_SHAPE_CONE = 9,
                           TRC_PBRT_SHAPE_PARABOLOID = 10, TRC_PBRT_SHAPE_HYPERBOLOID = 11 };
/* what the material's colour parameter names ("texture Kd" "name"): nothing, a 2-D checkerboard (rendered through the
 * reference's TextureInfo{Checker}, Texture.hh:17-43, with albedo = tex1), any other texture class (not rendered) */
enum trc_pbrt_texture { TRC_PBRT_TEX_NONE = 0, TRC_PBRT_TEX_CHECKERBOARD = 1, TRC_PBRT_TEX_OTHER = 2 };
typedef struct trc_pbrt_info {
    float    camera_to_world[16];   /* row-major, inverse of the CTM at the Camera directive (minipbrt Camera::cameraToWorld) */
    float    fov, lensradius, focaldistance;
    uint32_t perspective;           /* Camera "perspective" */
    uint32_t xres, yres;            /* Film "image" */
    uint32_t n_shapes;              /* world shapes in file order (object templates excluded), then the shapes of every instance */
    uint32_t n_unsupported_shapes, n_unsupported_materials, n_triangle_material_conflicts;
    uint32_t mis_ready;             /* squareList[5] and [6] are emitters: TRC_INTEGRATOR_MIS / _VOLUME are usable */
    uint32_t n_unsupported_textures;/* shapes whose colour names a texture other than a 2-D checkerboard */
    uint32_t n_instances;           /* ObjectInstance directives: each placed as a flat copy of its template's shapes, after the
                                       world's own shapes, in file order (the reference's scene arrays have no instance level) */
} trc_pbrt_info;
typedef struct trc_pbrt_shape {
    int32_t  kind;                  /* enum trc_pbrt_shape_kind, or -1 (a shape class that is not handled) */
    float    shape_to_world[16];    /* row-major CTM at the Shape directive */
    float    radius;                /* spheres, disks, cylinders, cones, paraboloids */
    uint32_t n_vertices, n_indices; /* triangle meshes, PLY meshes; quadrics (disk ... hyperboloid): of their tessellation */
    int32_t  material;              /* enum trc_pbrt_material of the graphics state */
    float    color[3];              /* Kd (matte, plastic, other), Kr (mirror), Kt (glass), 1 (metal) */
    int32_t  emitter;               /* inside an AreaLightSource "diffuse" */
    float    L[3];
    int32_t  mapped_type;           /* TRC_PRIM_SPHERE / _SQUARE / _TRIANGLE it became, -1 if dropped */
    uint32_t mapped_index;          /* index in that primitive list (first triangle for a mesh) */
    uint32_t mapped_material;       /* index into the scene's material table */
    float    zmin, zmax;            /* cylinder, paraboloid; disk: both = height; cone: 0, height; hyperboloid: the z range of p1, p2 */
    float    innerradius, phimax;   /* disk; every quadric (degrees) */
    int32_t  texture;               /* enum trc_pbrt_texture of the material's colour parameter */
    float    tex2[3];               /* checkerboard: the second colour (color[] holds tex1) */
    float    p1[3], p2[3];          /* hyperboloid: the swept segment's end points */
} trc_pbrt_shape;
trc_status trc_host_scene_load_pbrt(const char* path, trc_host_scene** out_scene, trc_Camera* out_camera,
                                    trc_pbrt_info* info, trc_pbrt_shape* shapes, uint32_t capacity);

/* procedural stand-in for the missing/untravelling assets: a displaced
 * UV-sphere "ball" with n_lat x n_lon quads (2 triangles each) */
trc_status trc_host_mesh_make_ball(uint32_t n_lat, uint32_t n_lon, float bump, trc_host_mesh** out);
/* a mesh handle over caller arrays (copied): the vertex / index buffers a host already holds -- what the reference
 * gets from MDLMesh.vertexBuffers / submesh.indexBuffer (AAPLRenderer.mm:527-529) */
trc_status trc_host_mesh_from_arrays(const trc_TriangleVertex* vertices, uint32_t n_vertices,
                                     const uint32_t* indices, uint32_t n_indices, trc_host_mesh** out);
/* k x k grid replication (config 4: >= 1 M triangles) */
trc_status trc_host_mesh_replicate(const trc_host_mesh* src, uint32_t k, float spacing, trc_host_mesh** out);
void       trc_host_mesh_view(const trc_host_mesh* m, const trc_TriangleVertex** vertices,
                              uint32_t* n_vertices, const uint32_t** indices, uint32_t* n_indices);
void       trc_host_mesh_destroy(trc_host_mesh* m);

#ifdef __cplusplus
}  /* extern "C" */
#endif

/* ------------------------------------------------------------------ */
/* layout locks                                 */
/* ------------------------------------------------------------------ */
#if defined(__cplusplus)
#define TRC_SA(c, m) static_assert(c, m)
#else
#define TRC_SA(c, m) _Static_assert(c, m)
#endif
TRC_SA(sizeof(trc_float2) == 8 && sizeof(trc_float3) == 16 && sizeof(trc_float4x4) == 64, "simd sizes");
TRC_SA(sizeof(trc_AABB) == 32 && offsetof(trc_AABB, maxi) == 16, "AABB");
TRC_SA(sizeof(trc_BVH) == 64 && offsetof(trc_BVH, pType) == 16 && offsetof(trc_BVH, pIndex) == 20 &&
       offsetof(trc_BVH, bBOX) == 32, "BVH");
TRC_SA(sizeof(trc_Sphere) == 272 && offsetof(trc_Sphere, center) == 16 && offsetof(trc_Sphere, model_matrix) == 32 &&
       offsetof(trc_Sphere, normal_matrix) == 96 && offsetof(trc_Sphere, inverse_matrix) == 160 &&
       offsetof(trc_Sphere, material) == 224 && offsetof(trc_Sphere, boundingBOX) == 240, "Sphere");
TRC_SA(sizeof(trc_Square) == 272 && offsetof(trc_Square, axis_j) == 1 && offsetof(trc_Square, range_i) == 8 &&
       offsetof(trc_Square, range_j) == 16 && offsetof(trc_Square, axis_k) == 24 &&
       offsetof(trc_Square, value_k) == 28 && offsetof(trc_Square, model_matrix) == 32 &&
       offsetof(trc_Square, material) == 224 && offsetof(trc_Square, boundingBOX) == 240, "Square");
TRC_SA(sizeof(trc_Cube) == 240 && offsetof(trc_Cube, normal_matrix) == 64 && offsetof(trc_Cube, inverse_matrix) == 128 &&
       offsetof(trc_Cube, box) == 192 && offsetof(trc_Cube, material) == 224, "Cube");
TRC_SA(sizeof(trc_TriangleVertex) == 32, "TriangleVertex");
TRC_SA(sizeof(trc_TextureInfo) == 32 && offsetof(trc_TextureInfo, albedo) == 16, "TextureInfo");
TRC_SA(sizeof(trc_Material) == 64 && offsetof(trc_Material, medium) == 4 && offsetof(trc_Material, specular) == 8 &&
       offsetof(trc_Material, eta) == 12 && offsetof(trc_Material, roughness) == 16 &&
       offsetof(trc_Material, textureInfo) == 32, "Material");
TRC_SA(sizeof(trc_Camera) == 176 && offsetof(trc_Camera, vfov) == 48 && offsetof(trc_Camera, focus_dist) == 64 &&
       offsetof(trc_Camera, u) == 80 && offsetof(trc_Camera, vertical) == 128 &&
       offsetof(trc_Camera, cornerLowLeft) == 160, "Camera");
TRC_SA(sizeof(trc_Complex) == 96 && offsetof(trc_Complex, frame_count) == 20 && offsetof(trc_Complex, photonBox) == 32 &&
       offsetof(trc_Complex, photonBoxSize) == 64 && offsetof(trc_Complex, photonInitialRadius) == 80 &&
       offsetof(trc_Complex, framePhotonSum) == 92, "Complex");
TRC_SA(sizeof(trc_ray) == 32, "trc_ray");
TRC_SA(sizeof(trc_hit) == 80 && offsetof(trc_hit, p) == 16 && offsetof(trc_hit, uv) == 52 && offsetof(trc_hit, n_descend) == 68, "trc_hit");
TRC_SA(sizeof(trc_params) == 32 && sizeof(trc_stats) == 112, "trc_params / trc_stats");
TRC_SA(sizeof(trc_GridDensityInfo) == 32 && offsetof(trc_GridDensityInfo, invMaxDensity) == 16 && offsetof(trc_GridDensityInfo, nx) == 20, "GridDensityInfo");
TRC_SA(sizeof(trc_PhotonRecord) == 80 && offsetof(trc_PhotonRecord, normal) == 16 && offsetof(trc_PhotonRecord, position) == 32 &&
       offsetof(trc_PhotonRecord, direction) == 48 && offsetof(trc_PhotonRecord, step) == 64 &&
       offsetof(trc_PhotonRecord, active) == 65, "PhotonRecord");
TRC_SA(sizeof(trc_CameraRecord) == 112 && offsetof(trc_CameraRecord, position) == 16 && offsetof(trc_CameraRecord, direction) == 32 &&
       offsetof(trc_CameraRecord, valid) == 48 && offsetof(trc_CameraRecord, alternative) == 64 &&
       offsetof(trc_CameraRecord, flux) == 80 && offsetof(trc_CameraRecord, radius) == 96 &&
       offsetof(trc_CameraRecord, photonCount) == 100, "CameraRecord");

#endif /* TRACER_ABI_H */
