/*
 * oracle.h -- C API of the CPU oracle (TEST INFRASTRUCTURE, not product code).
 *
 * The oracle is a scalar, literal CPU restatement of the reference's hot path
 * (RT_Metal Metal shaders), function for function, quirks included.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it;
 * the product path (tracer_amd/) never includes, links or calls anything here.
 *
 * PARITY PIN STATUS: the reference ships no tests, golden vectors or fixtures
 * for this path and its Metal/Obj-C++/Swift sources cannot be built or run on
 * Linux.  The only executable pin is PCG32: the reference's own
 * vendored RT_Metal/Tracer/pcg_basic.c compiles here (oracle/_ref) and its
 * outputs are committed as tests/golden/pcg32_kat.json.  Everything else is
 * pinned by analytic known-answer tests and brute-force cross-checks
 * (tests/test_oracle_*.py).  Parity against the reference's actual Metal
 * output is therefore UNPINNED ("parity unpinned" beyond PCG32).
 */
#ifndef ORACLE_H
#define ORACLE_H

#include "tracer_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* 1 if built with -DORACLE_USE_LIBM (glibc sinf/cosf/... instead of trc_detmath.h) */
int orc_uses_libm(void);

/* PCG32: Random.metal:3-26 == pcg_basic.c:42-72 */
void     orc_pcg32_srandom(uint64_t* state, uint64_t* inc, uint64_t initstate, uint64_t initseq);
uint32_t orc_pcg32_random(uint64_t* state, uint64_t inc);
float    orc_randomF(uint64_t* state, uint64_t inc);

/* Scene::hit (Render.hh:135-252) on a batch */
void orc_trace_rays(const trc_scene* scene, const trc_ray* rays, size_t n, trc_hit* out, int any_hit);
/* brute force: every leaf of the BVH array tested in array order (cross-check only) */
void orc_trace_rays_brute(const trc_scene* scene, const trc_ray* rays, size_t n, trc_hit* out);

/* kernelPathTracing (Render.metal:495-558) for the tiles owned by params->tile_rank,
 * params->spp successive "frames" starting at params->frame0.  rng/accum are the
 * RGBA32Uint / RGBA32F textures, updated in place.  n_threads row-band workers
 * (0 = hardware concurrency); results do not depend on n_threads. */
void orc_render(const trc_scene* scene, const trc_Camera* camera, const float env_rgb[3],
                uint32_t width, uint32_t height, uint32_t* rng_rgba, float* accum_rgba,
                const trc_params* params, trc_stats* stats, int n_threads);

/* density grid consulted by TRC_INTEGRATOR_VOLUME (traceVolume, Render.metal:78-275; GridDensityMedium, Medium.hh:111-199);
 * `density` is borrowed and must outlive the renders; NULL clears it */
void orc_set_density(const trc_GridDensityInfo* info, const float* density);
/* equirectangular RGB float environment (texHDR, Render.hh:25,42-48) for orc_render / orc_sppm_frames: a miss returns
 * its bilinear lookup instead of the constant env_rgb; `rgb` (3*w*h floats, row 0 at v = 0) is borrowed; NULL clears */
void orc_set_environment_map(uint32_t w, uint32_t h, const float* rgb);

/* material entry points (Material.hh:77-146) in the local shading frame */
void  orc_material_S_F(const trc_Material* m, const float wo[3], const float uv[2], const float uu[2],
                       float wi_out[3], float f_out[3], float* pdf_out);
void  orc_material_F(const trc_Material* m, const float wo[3], const float wi[3], const float uv[2],
                     const float uu[2], float f_out[3], float* pdf_out);
float orc_material_PDF(const trc_Material* m, const float wo[3], const float wi[3], const float uu[2]);

/* pbrt::SobolSampler (SobolSampler.hh:126-163) over the generated tables of include/trc_sobol.h */
uint64_t orc_sobol_interval_to_index(uint32_t log2res, uint64_t sample_index, uint32_t px, uint32_t py);
float    orc_sobol_sample_float(uint64_t index, uint32_t dimension);
float    orc_sobol_sample_dimension(uint32_t frame, uint32_t x, uint32_t y, uint32_t w, uint32_t h, uint32_t dimension);

/* small pieces for known-answer tests */
void  orc_offset_ray(const float p[3], const float n[3], float out[3]);                 /* Math.hh:57-74 */
float orc_fr_dielectric(float cosi, float eta);                                          /* BXDF.metal:3-22 */
void  orc_fr_conductor(float cosi, const float eta[3], const float k[3], float out[3]);  /* BXDF.metal:24-34 */
float orc_power_heuristic(int nf, float fPdf, int ng, float gPdf);                       /* Sampling.hh:137-140 */
void  orc_cosine_sample_hemisphere(const float u[2], float out[3]);                      /* Sampling.hh:125-129 */
float orc_erf(float x);                                                                  /* Math.hh:148-167 */
float orc_erfinv(float x);                                                               /* Math.hh:118-146 */
int   orc_aabb_hit_t(const trc_AABB* box, const trc_ray* ray, float tmin, float tmax, float* t_out); /* AABB.hh:92-112 */
void  orc_cast_ray(const trc_Camera* cam, float s, float t, uint64_t* state, uint64_t inc,
                   float origin_out[3], float dir_out[3]);                               /* Camera.hh:59-69 */
/* SPPM pass (Photon.metal 3-623; host sequencing AAPLRenderer.mm:860-1086), single-threaded */
typedef struct orc_sppm orc_sppm;
orc_sppm* orc_sppm_create(uint32_t width, uint32_t height, uint64_t photon_seed);
void      orc_sppm_destroy(orc_sppm* s);
void      orc_sppm_frames(orc_sppm* s, const trc_scene* scene, const trc_Camera* camera, const float env_rgb[3],
                          uint32_t* canvas_rng, float* accum, uint32_t n_frames);
void      orc_sppm_download(const orc_sppm* s, trc_CameraRecord* cam, trc_PhotonRecord* pho, float* mark,
                            float* count, trc_Complex* complex);
float     orc_photon_hash(const float idx[3], float hash_scale);                         /* Photon.hh:71-89 */

/* LBVH (the reference lists "LBVHs, Morton Encoding" as its own to-do, RT_Metal/README.md:42, so
 * there is no reference algorithm to restate).  This is the CPU statement of the build that
 * trc_upload_scene_lbvh runs on the GPU: 30-bit Morton codes of the leaf-box centroids, keys (code << 32 | leaf
 * index) sorted ascending, T. Karras' binary radix tree (HPG 2012) over the keys, boxes fitted bottom-up.
 * leaves[0..n): leaf records as trc_host_build_node writes them; out: 2n-1 records in the layout of
 * BVH::buildTree (BVH.hh:246-269): [root, leaf 0..n-1, interior 1..n-2].  *out_height = depth of the deepest leaf. */
void orc_lbvh_build(const trc_BVH* leaves, uint32_t n, trc_BVH* out, uint32_t* out_height);

/* The reference's host SAH build restated on its own data structure (oracle_sah.cpp): BVH::buildNode
 * (BVH.hh:273-314) and BVH::buildTree + BVH::make (BVH.hh:35-269), serial schedule.  The checker of
 * trc_host_build_node / trc_host_build_tree, which share no code with it. */
void orc_sah_leaf(const trc_AABB* box, const trc_float4x4* model_matrix, int32_t pType, uint32_t pIndex, trc_BVH* out);
void orc_sah_build(const trc_BVH* leaves, uint32_t n, trc_BVH* out);

/* output stage, Render.metal:29-75 + Render.hh:78-95 (see trc_tonemap in tracer_abi.h for the exact definition) */
void orc_tonemap(const float* accum_rgba, uint32_t W, uint32_t H, uint8_t* rgba8, float* exposure_out);

/* media pieces for known-answer tests (HitRecord.hh:45-77, Medium.hh:129-145) */
float orc_hg_sample(float g, const float wo[3], const float uu[2], float wi_out[3]);
float orc_phase_hg(float cosTheta, float g);
float orc_grid_density(const trc_GridDensityInfo* info, const float* density, const float p[3]);

/* deterministic math under test (identity wrappers over trc_detmath.h / libm) */
float orc_math(int fn, float a, float b);   /* 0 sin 1 cos 2 exp 3 log 4 pow 5 asin 6 acos 7 atan2 */

#ifdef __cplusplus
}
#endif
#endif
