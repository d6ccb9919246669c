// trc_ctx.hpp -- pieces shared by the translation units of libtracer_amd.so (trc_abi.hip, trc_sppm.hip):
// kernel-side scene staging helpers, the context struct and the error helpers.
#pragma once

#include <hip/hip_runtime.h>

#include <string>
#include <utility>
#include <vector>

#include "tracer_abi.h"
#ifdef TRC_TEST_HOOKS
#include "tracer_test_hooks.h"      // libtracer_amd_hooks.so: the product's sources + the test hooks
#endif
#include "dev_integrator.hpp"

using namespace trcdev;

// ======================================================================= kernels
extern __shared__ __attribute__((aligned(16))) uint32_t trc_smem[];

// launch-list entries (KRender::order): index into the block list | part code << 25.  Code 0: the whole block; 1..4: its 4x4
// quarter code - 1 (x fastest) on 16 lanes; 5..20: the 2x2 sixteenth (code - 5) & 3 of quarter (code - 5) >> 2 on 4 lanes;
// 21..84 (round 6): ONE pixel on one lane -- pixel (code - 21) & 3 of sixteenth ((code - 21) >> 2) & 3 of quarter (code - 21) >> 4:
// the floor of a pixel's sample chain, for the shares of a frame that end on a single 4-lane wavefront.
constexpr uint32_t kLaunchCodeShift = 25u;
constexpr uint32_t kLaunchIndexMask = (1u << kLaunchCodeShift) - 1u;
// duration slots per 8x8 block of a list that may be split (KRender::cost_stride): 0..3 the quarters (0 also the whole
// block), 4..19 the sixteenths, 20..83 the pixels -- slot = code - 1
constexpr uint32_t kCostSlots = 84u;
constexpr uint32_t kPlanWords = 12u;          // trc_abi.hip k_plan_split

struct KScene {
    DScene sc;
    float root_box[6];
};

struct KRender {
    KScene ks;
    DCamera cam;
    float ambient[3];
    const float* env_rgb; uint32_t env_w, env_h;      // environment map (null: constant `ambient`)
    DFrame fr;
    uint32_t spp, max_depth, frame0;
    uint32_t view_height;               // rows per view of a stacked frame (= frame height for a single view)
    const uint32_t* tiles;              // tx | ty << 16, one per workgroup
    const float* density;               // traceVolume: GridDensity medium grid (null when absent)
    trc_GridDensityInfo dinfo;
    const uint8_t* occupancy;           // ... and its 4x4x4-brick occupancy (dev_integrator.hpp::grid_sample)
    unsigned long long* stats;          // kStatCount counters
    uint32_t* stack_ovf;                // (stack_depth - stack_lds) rows of 64 entries per wavefront (null: the stack is all LDS)
    uint32_t* queue;                    // k_render_pwg: next position of the launch order to hand out
    uint32_t blk_shift;                 // log2 of the pixel-block edge of one wavefront: 3 (8x8, 64 lanes) or 2 (4x4, 16 lanes)
    uint32_t n_tiles, strip;            // k_render_strip: blocks in `tiles`, consecutive blocks per wavefront (1: k_render)
    const uint32_t* order;              // launch list: order[slot] = index into `tiles` | part code << 27 (null: identity; codes
                                        // above: cost-adaptive block size, k_plan_split).  k_render_strip: strip indices, no codes.
    const uint32_t* n_launch;           // device word: entries of `order` in this launch (null: n_tiles); workgroups past it exit
    uint32_t cost_div;                  // a block's cost = its wavefront's duration in shader clocks / cost_div (= 4 x spp: per sample)
    uint32_t* block_cost;               // duration of each block in this launch (the next launch's sort key): slot kCostSlots * tile +
                                        // max(code - 1, 0) when the list may be split (cost_stride = kCostSlots), else slot `tile`
    uint32_t cost_stride;
    const uint32_t* sobol32;            // TRC_FLAG_SOBOL: [40][52] generator matrices (null otherwise)
    const uint64_t* sobol_vdc;          // ... [52] VdCSobolMatrices[m - 1] + [52] VdCSobolMatricesInv[m - 1]
    uint32_t sobol_m;                   // ... log2Resolution
};

struct KTrace {
    KScene ks;
    const trc_ray* rays;
    trc_hit* hits;
    uint32_t n;
};

// cooperative copy of the blob prefix (prims, materials, top fat nodes) into LDS
__device__ __forceinline__ const uint32_t* stage_scene(const DScene& sc) {
    const uint4* src = reinterpret_cast<const uint4*>(sc.blob);
    uint4* dst = reinterpret_cast<uint4*>(trc_smem);
    const uint32_t n16 = sc.lds_dwords >> 2;
    for (uint32_t i = threadIdx.x; i < n16; i += kBlock) dst[i] = src[i];
    __syncthreads();
    return trc_smem;
}

// this lane's column of the traversal stack (one row per entry); the instrumented kernels keep a second region of the
// same size for the levels
__device__ __forceinline__ uint32_t* lane_stack(const DScene& sc) { return trc_smem + sc.lds_dwords + threadIdx.x; }
__device__ __forceinline__ uint32_t* lane_lvstack(const DScene& sc) { return trc_smem + sc.lds_dwords + sc.stack_lds * kBlock + threadIdx.x; }

__device__ __forceinline__ SceneRef make_scene_ref(const DScene& sc, const uint32_t* small_base) {
    SceneRef S;
    S.small_base = small_base;
    S.blob = sc.blob;
    S.off_nodes = sc.off_nodes; S.off_spheres = sc.off_spheres; S.off_squares = sc.off_squares;
    S.off_cubes = sc.off_cubes; S.off_materials = sc.off_materials;
    S.off_tripos = sc.off_tripos; S.off_triattr = sc.off_triattr;
    S.n_lds_nodes = sc.n_lds_nodes;
    S.stack_lds = sc.stack_lds;
    S.stack_cap = sc.stack_depth;
    S.ovf = nullptr;
    S.descend_min = sc.descend_min;
    return S;
}

// Work counters live in kStatRows copies of one row; a wavefront adds to row (its index mod kStatRows) and the reader
// sums the rows.  One row for everybody meant one 64-bit atomic per wavefront and counter on ONE address: device-scope
// atomics on a line are served one after the other (~12 ns each, measured), and 32 400 wavefronts x 3 counters held a
// 1-spp frame at 1.2 ms and the SPPM camera pass at 0.4 ms -- whatever the kernels did.
constexpr uint32_t kStatRows = 1024;
constexpr uint32_t kStatRowStride = ((kStatCount + 3 * kProfCount + 15) / 16) * 16;      // 64-bit words; rows start on 128-byte lines
static_assert(kStatRowStride <= 64, "k_stats_sum sums a row with one 64-thread workgroup");
__device__ __forceinline__ unsigned long long* stat_row(unsigned long long* stats, uint32_t wave_index) {
    return stats + (size_t)(wave_index & (kStatRows - 1u)) * kStatRowStride;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}


// ======================================================================= context
struct SppmState;   // trc_sppm.hip

struct trc_ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    std::string error;

    // scene
    bool has_scene = false;
    KScene ks{};
    uint32_t* d_blob = nullptr;
    size_t blob_bytes = 0;
    bool lds_scene = false;
    bool lds_prefix_ok = false;         // the fat nodes are in top-of-tree-first order: any prefix may be staged
    uint32_t* d_stack_ovf = nullptr;    // traversal-stack overflow rows of the render launches (deep trees only)
    size_t stack_ovf_bytes = 0;
    uint32_t* d_queue = nullptr;        // block queue head of the persistent-workgroup launches
    trc_BVH* d_bvh_ref = nullptr;    // tree built by trc_upload_scene_lbvh, reference array layout (trc_download_bvh)
    uint32_t n_bvh_ref = 0, lbvh_height = 0;
    float lbvh_build_ms = 0.0f;
    float* d_density = nullptr;      // GridDensity medium (trc_upload_density)
    uint8_t* d_occupancy = nullptr;
    trc_GridDensityInfo dinfo{};

    bool has_camera = false;
    DCamera cam{};
    float ambient[3] = {0, 0, 0};
    float* d_envmap = nullptr; uint32_t env_w = 0, env_h = 0;
    uint32_t* d_sobol32 = nullptr; uint64_t* d_sobol_vdc = nullptr; uint32_t sobol_m = ~0u;   // TRC_FLAG_SOBOL tables

    // frame
    uint32_t width = 0, height = 0;
    uint32_t* d_rng = nullptr;
    float* d_accum = nullptr;

    // tiles for (nranks, rank)
    uint32_t* d_tiles = nullptr;
    // adaptive launch order: duration of every block in the previous launch -> most expensive blocks first in the next
    uint32_t* d_block_cost = nullptr;
    uint32_t* d_order_keys[2] = {nullptr, nullptr};
    uint32_t* d_order_vals[2] = {nullptr, nullptr};
    uint32_t* d_order_hist = nullptr;
    // cost-adaptive block size (trc_abi.hip::plan_split): which 8x8 blocks the last launch ran as four 4x4 quarters, the
    // launch list with the quarters spliced in, and the plan {quarters' parents K, entries}
    uint32_t* d_split = nullptr;
    uint32_t* d_whole = nullptr;        // cost of a block when it last ran whole (while it runs as quarters)
    uint32_t* d_cost_est = nullptr;     // per cost slot: the shortest duration seen lately (k_filter_costs)
    uint32_t* d_qsplit = nullptr;       // [4 n] quarter q of block i ran as four sixteenths in the last launch ...
    uint32_t* d_qwhole = nullptr;       // [4 n] ... and what it cost when it last ran as one quarter
    uint32_t* d_swhole = nullptr;       // [16 n] what a sixteenth cost when it last ran as one (while it runs as four pixels: bits 4..7 of qsplit)
    uint32_t* d_launch = nullptr;
    uint32_t* d_plan = nullptr;
    uint32_t* d_cost_scratch = nullptr;  // durations of instrumented launches (never read)
    uint32_t* d_plan_gather = nullptr;  // k_plan_gather's dense per-rank arrays (6 words per block)
    uint32_t plan_streak = 0, plan_reused = 0, plan_n = 0, plan_wave_slots = 0, plan_grid_cap = 0; bool plan_split_mode = false;   // plan reuse (trc_render)
    uint32_t launch_cap = 0;            // entries d_launch holds = the grid of a launch that may split
    bool split_live = false;            // d_split holds flags of the last launch's plan (else all zero)
    bool cost_quarters = false;         // d_block_cost / d_split describe a launch made with cost_stride 4
    bool cost_valid = false; uint32_t cost_strip = 1;
    uint32_t last_cost_div = 0, last_wave_slots = 0;     // of the last render launch (trc_debug_launch_shape)
    trc_params deferred{}; bool has_deferred = false; uint64_t deferred_calls = 0;   // a launch of few samples kept for coalescing (trc_render)
    int cost_head_age = 0;                    // 1: the costs are a cold head's (trc_render), 2: the launch after it ran on them
    bool cost_fresh_next = false;             // the next ordered launch takes the last launch's raw durations as its costs (trc_set_camera, policy 2)
    uint32_t cost_integrator = 0xFFFFFFFFu;   // integrator the recorded costs belong to
    const uint32_t* d_stale_order = nullptr;     // the launch order of the view before the camera moved: the next cold pass's prior
    const uint32_t* d_last_order = nullptr; uint32_t order_age = 0;     // most recent sorted order (short launches reuse it)
    int cu_count = 0;
    uint32_t n_tiles = 0, tiles_nranks = 0, tiles_rank = 0, tiles_view_height = 0, tiles_blk_shift = 3;

    // stats
    unsigned long long* d_stats = nullptr;       // kStatRows rows of kStatRowStride counters (stat_row)
    unsigned long long* d_stats_sum = nullptr;   // their sum, made by trc_get_stats / trc_debug_profile
    uint64_t launches = 0;
    double kernel_ms = 0.0;
    double schedule_ms = 0.0;                                   // launch-list kernels (order, sort, plan) of those launches
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending_sched;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;   // per-launch event pairs not yet read
    std::vector<hipEvent_t> event_pool;

    // RCCL
    void* comm = nullptr;
    int nranks = 1, rank = 0;
    float* d_reduce_recv = nullptr;
    // pipelined compose (trc_group_reduce_accum_async): second accumulator, communication stream, and per-buffer
    // "reduce finished" events (ev_busy belongs to d_accum, ev_busy_alt to d_accum_alt; they swap with the buffers)
    float* d_accum_alt = nullptr;
    float* d_composed = nullptr;        // buffer holding the most recently composed frame (one of the two)
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_rendered = nullptr, ev_busy = nullptr, ev_busy_alt = nullptr;
    bool busy = false, busy_alt = false;
    // sample-sharded compose (trc_group_compose_samples): the slices received from the ranks, and the composed slices
    float* d_shard_in = nullptr; float* d_shard_out = nullptr; size_t shard_px = 0; int shard_nranks = 0;
    // ... and the pipelined form's snapshot of the accumulator (trc_group_compose_samples_async), free again at ev_snapshot_free
    float* d_shard_src = nullptr; hipEvent_t ev_snapshot_free = nullptr; bool snapshot_busy = false;

    SppmState* sppm = nullptr;       // trc_sppm.hip

    // caller-supplied collectives (trc_group_set_collectives) instead of an RCCL communicator
    trc_collectives coll{};
    bool coll_active = false;
    void* h_stage = nullptr;            // pinned staging buffer of host-staged collectives
    size_t h_stage_bytes = 0;
    uint32_t* h_readback = nullptr;     // 1 KB of pinned host memory for the per-level counter read-back of trc_upload_scene_sah
    char* h_xfer = nullptr;             // 2 x kXferChunk bytes of pinned host memory: every transfer to / from caller memory goes through it (trc_copy_*)
    hipEvent_t ev_xfer[2] = {nullptr, nullptr};

    // A/B and test knobs, per context: defaults from the environment at trc_create (TRC_NO_LDS_FIT, TRC_STACK_LDS_LEVELS,
    // TRC_STRIP_LEN, TRC_NO_PWG, TRC_SPPM_SERIAL_CAMERA), changed through trc_debug_set
    struct Knobs { int no_lds_fit = 0, stack_lds_levels = 0, strip_len = 0, no_pwg = 0, sppm_serial_camera = 0, sppm_timing = 0, force_blk_shift = 0, no_split = 0, no_cost_filter = 0, no_cold_probe = 0, probe_spp = 0, no_plan_reuse = 0, no_coalesce = 0, no_dense = 0, head_stages = 0, descend_min = 0, camera_policy = 0; } knobs;
    // k_render_pwg instantiations that were granted > 64 KB of dynamic LDS on THIS context's device (bit = integrator * 2 +
    // sobol): hipFuncSetAttribute applies to the current device only, so the grant is per context, not per process
    uint32_t pwg_lds_granted = 0;

    bool grouped() const { return comm != nullptr || coll_active; }
};


inline trc_status trc_fail(trc_ctx* ctx, trc_status s, const std::string& msg) {
    if (ctx) ctx->error = msg;
    return s;
}
#define HIP_TRY(ctx, expr)                                                                 \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return trc_fail(ctx, e_ == hipErrorOutOfMemory ? TRC_ERR_OOM : TRC_ERR_HIP,     \
                            std::string(#expr) + ": " + hipGetErrorString(e_));            \
    } while (0)

// RCCL entry points, resolved at run time (dlopen) so the library loads where RCCL is absent
struct IdBlob { char internal[TRC_UNIQUE_ID_BYTES]; };   // ncclUniqueId, passed by value
struct Rccl {
    void* handle = nullptr;
    bool ready = false;              // every entry point below resolved
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, IdBlob, int) = nullptr;
    int (*Reduce)(const void*, void*, size_t, int, int, int, void*, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    // point-to-point (the sample-sharded compose): optional, a library without them still composes tiles
    int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
extern Rccl g_rccl;
bool trc_load_rccl(std::string& err);
// ncclDataType_t / ncclRedOp_t ordinals (rccl.h:448-466)
constexpr int kNcclUint8 = 1, kNcclUint32 = 3, kNcclFloat = 7, kNcclSum = 0, kNcclMax = 2, kNcclMin = 3;

// The three collectives the group calls need, in place on device buffers, through RCCL (ctx->comm) or the caller's table
// (ctx->coll; host-staged tables get the data in pinned host memory).  `what` names the call in error messages.
trc_status trc_coll_reduce(trc_ctx* ctx, void* buf, size_t count, int dtype, int op, int root, hipStream_t st, const char* what);
trc_status trc_coll_allreduce(trc_ctx* ctx, void* buf, size_t count, int dtype, int op, hipStream_t st, const char* what);
trc_status trc_coll_allgather(trc_ctx* ctx, void* buf, size_t bytes_per_rank, hipStream_t st, const char* what);

// Transfers between device memory and memory the CALLER (or a std::vector of ours) owns.  They do not hand the host pointer to the HIP
// runtime: a copy to / from pageable memory makes the runtime pin the caller's pages in place, and under many processes sharing the GPU a
// D2H copy into freshly mapped pages was seen to leave whole ranges of the destination untouched -- zeros where the device buffer,
// downloaded again, had the data (round 6: tests/campaigns/sppm_stress.py reproduced round 5's "unwritten photon records" 380 times in
// 9 000 scenes, every one of them a transfer, none a kernel).  So the bytes go through the context's own pinned
// buffer, two chunks of kXferChunk in flight, and are copied to / from the caller's memory by the CPU.  Synchronous: on return the
// transfer is complete (`st` is synchronised up to it).
constexpr size_t kXferChunk = 4u << 20;
trc_status trc_copy_to_host(trc_ctx* ctx, void* host, const void* dev, size_t bytes, hipStream_t st);
trc_status trc_copy_to_device(trc_ctx* ctx, void* dev, const void* host, size_t bytes, hipStream_t st);

// tiles owned by `rank` of `nranks` (XCD-aware order) uploaded into ctx->d_tiles; shared by render and SPPM
trc_status trc_ensure_tiles(trc_ctx* ctx, uint32_t nranks, uint32_t rank, uint32_t view_height = 0, uint32_t blk_shift = 3);
size_t trc_dyn_lds_bytes(const trc_ctx* ctx, bool stats);
// trc_lbvh.hip: stable 24-bit radix sort of (key, value) pairs
void trc_sort_pairs24(hipStream_t st, uint32_t* keys[2], uint32_t* vals[2], uint32_t* hist, uint32_t* digit_base, uint32_t n, int* result);
uint32_t trc_sort_hist_words(uint32_t n);
trc_status trc_flush(trc_ctx* ctx);       // launches what trc_render kept back (every other entry point calls it first)
hipEvent_t trc_get_event(trc_ctx* ctx);   // from the context's pool (null on failure); pairs go to ctx->pending
void trc_sppm_release(trc_ctx* ctx);   // frees ctx->sppm (no-op when absent)
void trc_sppm_order_after_camera(trc_ctx* ctx);   // context stream waits for a camera pass running ahead (no-op when none)
