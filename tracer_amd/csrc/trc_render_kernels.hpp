// trc_render_kernels.hpp -- kernelPathTracing (RT_Metal/Metal/Render.metal:495-558) as HIP kernels: DEFINITIONS.  Included by the
// two translation units that instantiate them (trc_render_config.hpp says which and why); trc_abi.hip only launches them.
#pragma once

#include <hip/hip_runtime.h>

#include "trc_render_config.hpp"

// One pixel block (8x8 pixels on 64 lanes, or 4x4 on 16) of kernelPathTracing: all `spp` samples of every pixel, RNG texel
// and accumulator read and written once.  `slot` = position of the block in the launch order; `stack` / `lvstack` / `ovf` =
// this lane's columns of the wavefront's traversal stack.  Shared by k_render (one block per one-wavefront workgroup) and
// k_render_pwg (wavefronts of a persistent workgroup pulling blocks from a queue).
//
// PARK = 8 | 10 (k_render_pwg on trees read from memory, k_render_dense): the values a lane touches only where a sample begins or ends
// -- the running mean, (u, v), the sample counter, with 10 rows the pixel's coordinates -- and the two work counters live in PARK words
// of the lane's LDS column `park` (row r at park[r * kBlock]) instead of registers: that many values fewer to carry through the walk and
// the shading code of every iteration, i.e. fewer spills to scratch -- which on a mesh scene streams through the L2 the node fetches
// live in, and on the LDS-resident headline scene are issue slots.  Same loads, same arithmetic, same stores per lane.
template <bool LDS, bool STATS, int INTEGRATOR, bool SOBOL, bool HYB, int PARK = 0, class COUNT = uint32_t>
__device__ __forceinline__ void render_block(const KRender& kp, const DScene& sc, const uint32_t* small_base, uint32_t* stack, uint32_t* lvstack,
                                             uint32_t* ovf, uint32_t* park, const uint32_t slot, const uint32_t lane,
                                             COUNT& n_rays, COUNT& n_shaded, uint32_t& n_paths, TravCounters& cnt) {
    const uint64_t t_start = clock64();          // this wavefront's own duration = the next launch's sort key
    const uint32_t entry = kp.order ? kp.order[slot] : slot;     // adaptive launch order / cost-adaptive block size (trc_render)
    const uint32_t index = entry & kLaunchIndexMask, code = entry >> kLaunchCodeShift;
    if (index >= kp.n_tiles) return;                            // padding of the launch list's part region (k_pad_launch)
    const uint32_t tile = kp.tiles[index];                      // pixel block: x | y << 16 in units of the block edge
    // 3: 8x8 pixels, all 64 lanes; 2: 4x4 pixels, lanes 0..15; 1: 2x2 pixels, lanes 0..3; 0: one pixel, lane 0 -- the whole list
    // (kp.blk_shift), or a part of an 8x8 block whose previous launch ran long (trc_ctx.hpp: launch codes)
    const uint32_t part = code >= 21u ? (code - 21u) >> 2 : (code >= 5u ? code - 5u : 0u);      // sixteenth 0..15
    const uint32_t quarter = code >= 5u ? part >> 2 : (code ? code - 1u : 0u);
    const uint32_t bs = code >= 21u ? 0u : (code >= 5u ? 1u : (code ? 2u : kp.blk_shift));
    const uint32_t pixel = code >= 21u ? (code - 21u) & 3u : 0u;
    const uint32_t qx = code ? ((quarter & 1u) << 2) + (code >= 5u ? (part & 1u) << 1 : 0u) + (pixel & 1u) : 0u;
    const uint32_t qy = code ? ((quarter >> 1) << 2) + (code >= 5u ? ((part >> 1) & 1u) << 1 : 0u) + (pixel >> 1) : 0u;
    const uint32_t px = ((tile & 0xFFFFu) << kp.blk_shift) + qx + (lane & ((1u << bs) - 1u));
    const uint32_t py = ((tile >> 16) << kp.blk_shift) + qy + (lane >> bs);
    const uint32_t W = kp.fr.width, H = kp.fr.height;
    const bool active = lane < (1u << (2u * bs)) && px < W && py < H;
    const uint32_t canon = index * kp.cost_stride + (code ? code - 1u : 0u);

    if (active) {
        PathCtx cx;
        cx.S = make_scene_ref(sc, small_base);
        cx.S.ovf = ovf;
        cx.root_min = f3(kp.ks.root_box[0], kp.ks.root_box[1], kp.ks.root_box[2]);
        cx.root_max = f3(kp.ks.root_box[3], kp.ks.root_box[4], kp.ks.root_box[5]);
        cx.sh.mats = small_base + sc.off_materials;
        cx.ambient = f3(kp.ambient[0], kp.ambient[1], kp.ambient[2]);
        cx.env.rgb = kp.env_rgb; cx.env.w = kp.env_w; cx.env.h = kp.env_h;
        cx.stack = stack;
        cx.lvstack = lvstack;
        cx.max_depth = kp.max_depth;
        cx.density = kp.density;
        cx.dinfo = kp.dinfo;
        cx.occupancy = kp.occupancy;
        if (SOBOL) {                                   // SobolSampler(rng, frame, thread_pos, vsize), SobolSampler.hh:50-61
            cx.sobol32 = kp.sobol32; cx.sobol_vdc = kp.sobol_vdc;
            cx.sobol_m = kp.sobol_m; cx.sobol_res = 1u << kp.sobol_m;
            cx.sobol_xy[0] = px; cx.sobol_xy[1] = py % kp.view_height;
        }

        // the values that are touched only where a sample begins or ends: registers, or (PARK) rows of the lane's LDS column
        F3 cached_r = f3(0);
        float u_r = 0, v_r = 0;
        uint32_t s_r = 0;
        size_t pix_r = 0;
        auto row_f = [&](uint32_t r) { return __uint_as_float(park[r * kBlock]); };
        auto put_f = [&](uint32_t r, float x) { park[r * kBlock] = __float_as_uint(x); };
        auto get_cached = [&]() { if constexpr (PARK) return f3(row_f(kParkCachedX), row_f(kParkCachedY), row_f(kParkCachedZ)); else return cached_r; };
        auto set_cached = [&](F3 c) { if constexpr (PARK) { put_f(kParkCachedX, c.x); put_f(kParkCachedY, c.y); put_f(kParkCachedZ, c.z); } else cached_r = c; };
        auto get_sample = [&]() { if constexpr (PARK) return park[kParkSample * kBlock]; else return s_r; };
        auto set_sample = [&](uint32_t s) { if constexpr (PARK) park[kParkSample * kBlock] = s; else s_r = s; };
        uint4 texel;
        {
            const size_t pix = (size_t)py * W + px;
            texel = reinterpret_cast<const uint4*>(kp.fr.rng)[pix];       // r, g, b, a
            const float4 acc = reinterpret_cast<const float4*>(kp.fr.accum)[pix];
            const float u = (float)px / (float)W;                               // no sub-pixel jitter (B-2)
            const float v = (float)(py % kp.view_height) / (float)kp.view_height;      // one view: view_height == H
            set_cached(f3(acc.x, acc.y, acc.z));
            set_sample(0u);
            if constexpr (PARK) { put_f(kParkU, u); put_f(kParkV, v); }
            if constexpr (PARK >= 10) { park[kParkPx * kBlock] = px; park[kParkPy * kBlock] = py; }
            else { u_r = u; v_r = v; pix_r = pix; }
        }

        PathState ps;
        Pcg rng;
        uint64_t state_after_cast = 0;       // SOBOL: the sampler draws from a copy, the texel keeps this (SobolSampler.hh:50)
        // castRay, then (SOBOL) the sampler of this frame: Render.metal:527-530
        auto begin_sample = [&](uint32_t s) {
            const float u = PARK ? row_f(kParkU) : u_r, v = PARK ? row_f(kParkV) : v_r;
            path_begin(ps, cast_ray(kp.cam, u, v, rng), kp.max_depth);
            if (SOBOL) {
                state_after_cast = rng.state;
                ps.sobol_index = sobol_interval_to_index(cx, (uint64_t)(kp.frame0 + s));
                ps.sobol_dim = 0;
            }
        };
        bool alive = kp.spp > 0;
        if (alive) {
            // pcg32_t rng = { rng_inc, rng_state } aggregate-initialises {state, inc}: the two 64-bit
            // words trade roles every frame (Render.metal:516-519,545-557, B-1)
            rng.state = ((uint64_t)texel.z << 32) | texel.w;
            rng.inc = ((uint64_t)texel.x << 32) | texel.y;
            begin_sample(0u);
        }
        // end of a sample: accumulate, hand the RNG words back to the texel, start the next sample (or stop)
        auto finish_sample = [&](F3 color) {
            ProfScope<STATS> scope(cnt, kProfFinish);
            const bool bad = is_inf(color.x) || is_nan(color.x) || is_inf(color.y) || is_nan(color.y) ||
                             is_inf(color.z) || is_nan(color.z);
            if (bad) color = f3(0);                                         // :537-538
            uint32_t s = get_sample();
            const uint32_t frame = kp.frame0 + s;
            set_cached(div_shared(get_cached() * (float)frame + color, (float)(frame + 1)));  // running mean, :540-541
            if (SOBOL) rng.state = state_after_cast;
            // PARK: the texel is not carried through the loop -- the last sample's write-back is the RNG's own words (below)
            if constexpr (!PARK) {
                texel.y = (uint32_t)rng.state; texel.x = (uint32_t)(rng.state >> 32);
                texel.w = (uint32_t)rng.inc;   texel.z = (uint32_t)(rng.inc >> 32);
                n_paths++;
            }
            ++s;
            set_sample(s);
            if (s == kp.spp) {
                alive = false;
            } else {
                if constexpr (PARK) { const uint64_t t = rng.state; rng.state = rng.inc; rng.inc = t; }      // the words trade roles (B-1)
                else {
                    rng.state = ((uint64_t)texel.z << 32) | texel.w;
                    rng.inc = ((uint64_t)texel.x << 32) | texel.y;
                }
                begin_sample(s);
            }
        };
        // flat loop: one Scene::hit per iteration; a finished path immediately regenerates the next sample
        while (alive) {
            ProfScope<STATS> loop_scope(cnt, kProfLoop);
            constexpr bool kVolume = INTEGRATOR == TRC_INTEGRATOR_VOLUME;
            constexpr int kDefer = STATS ? 0 : (LDS ? TRC_DEFER_LDS : TRC_DEFER_GLOBAL);      // dev_intersect.hpp: trav_test_leaf
            bool hitted = true;
            if (!(kVolume && TRC_TRACK_SLICE > 0 && ps.tracking)) {       // a lane between two slices of its delta tracker has no ray to trace
                bump(n_rays);
                hitted = scene_hit<LDS, STATS, false, false, kVolume, HYB, kDefer>(cx.S, cx.root_min, cx.root_max, ps.ray, ps.rec, FLT_MAX,
                                                                                  cx.stack, cx.lvstack, cnt);
            }
            F3 color;
            const bool finished = (INTEGRATOR == TRC_INTEGRATOR_PATH)
                                      ? path_step<STATS, SOBOL>(cx, ps, hitted, rng, cnt, n_shaded, color)
                                      : mis_step<LDS, STATS, kVolume, SOBOL, HYB>(cx, ps, hitted, rng, cnt, n_rays, n_shaded, color);
            if (finished) finish_sample(color);
        }
        if constexpr (PARK) {       // the loop is left by the last finish_sample only (trc_render launches spp >= 1): what that one would
            n_paths += kp.spp;      // have put into the texel, and one finished sample per call of it
            texel.y = (uint32_t)rng.state; texel.x = (uint32_t)(rng.state >> 32);
            texel.w = (uint32_t)rng.inc;   texel.z = (uint32_t)(rng.inc >> 32);
        }
        const F3 cached = get_cached();
        size_t pix = pix_r;
        if constexpr (PARK >= 10) pix = (size_t)park[kParkPy * kBlock] * W + park[kParkPx * kBlock];
        else if constexpr (PARK != 0) {
            // 8 rows: the pixel's coordinates are not carried at all -- the launch entry is read once more (volatile: a second load, not
            // a value kept alive through the loop) and decoded as at the top
            const uint32_t entry2 = kp.order ? *reinterpret_cast<const volatile uint32_t*>(kp.order + slot) : slot;
            const uint32_t code2 = entry2 >> kLaunchCodeShift;
            const uint32_t tile2 = *reinterpret_cast<const volatile uint32_t*>(kp.tiles + (entry2 & kLaunchIndexMask));
            const uint32_t part2 = code2 >= 21u ? (code2 - 21u) >> 2 : (code2 >= 5u ? code2 - 5u : 0u);
            const uint32_t quarter2 = code2 >= 5u ? part2 >> 2 : (code2 ? code2 - 1u : 0u);
            const uint32_t bs2 = code2 >= 21u ? 0u : (code2 >= 5u ? 1u : (code2 ? 2u : kp.blk_shift));
            const uint32_t pixel2 = code2 >= 21u ? (code2 - 21u) & 3u : 0u;
            const uint32_t qx2 = code2 ? ((quarter2 & 1u) << 2) + (code2 >= 5u ? (part2 & 1u) << 1 : 0u) + (pixel2 & 1u) : 0u;
            const uint32_t qy2 = code2 ? ((quarter2 >> 1) << 2) + (code2 >= 5u ? ((part2 >> 1) & 1u) << 1 : 0u) + (pixel2 >> 1) : 0u;
            pix = (size_t)(((tile2 >> 16) << kp.blk_shift) + qy2 + (lane >> bs2)) * W + ((tile2 & 0xFFFFu) << kp.blk_shift) + qx2 + (lane & ((1u << bs2) - 1u));
        }
        float4 out; out.x = cached.x; out.y = cached.y; out.z = cached.z; out.w = 1.0f;
        reinterpret_cast<float4*>(kp.fr.accum)[pix] = out;
        reinterpret_cast<uint4*>(kp.fr.rng)[pix] = texel;
    }

    // per SAMPLE (cost_div = 4 x spp), so that launches of different lengths speak of the same quantity
    if (lane == 0) kp.block_cost[canon] = (uint32_t)min((unsigned long long)(clock64() - t_start) / kp.cost_div, 0xFFFFFFull);
}

// the body of k_render (one one-wavefront workgroup = one entry of the launch list)
template <bool LDS, bool STATS, int INTEGRATOR, bool SOBOL, int PARK = 0>
__device__ __forceinline__ void render_workgroup(const KRender& kp) {
    if (kp.n_launch && blockIdx.x >= *kp.n_launch) return;      // the grid is sized for the most quarters a plan may splice in
    const DScene& sc = kp.ks.sc;
    const uint32_t* small_base = stage_scene(sc);
    uint32_t* stack = lane_stack(sc);
    uint32_t* lvstack = lane_lvstack(sc);

    const uint32_t lane = threadIdx.x;
    uint32_t n_rays = 0, n_shaded = 0, n_paths = 0;
    TravCounters cnt;
    counters_zero(cnt);
    constexpr bool kHybridStack = !LDS && !STATS && hybrid_stack(INTEGRATOR);     // plan_launch_lds
    uint32_t* ovf = kHybridStack ? kp.stack_ovf + (size_t)blockIdx.x * sc.stack_ovf_rows * kBlock + lane : nullptr;
    if constexpr (PARK != 0) {                  // the park rows follow the stack rows of this one-wavefront workgroup (trc_abi.hip: dense_lds_bytes)
        uint32_t* park = stack + sc.stack_lds * kBlock;
        park[kParkRays * kBlock] = 0u; park[kParkShaded * kBlock] = 0u;
        LdsCount c_rays{park + kParkRays * kBlock}, c_shaded{park + kParkShaded * kBlock};
        render_block<LDS, STATS, INTEGRATOR, SOBOL, kHybridStack, PARK>(kp, sc, small_base, stack, lvstack, ovf, park, blockIdx.x, lane, c_rays, c_shaded, n_paths, cnt);
        n_rays = park[kParkRays * kBlock]; n_shaded = park[kParkShaded * kBlock];
    } else
    render_block<LDS, STATS, INTEGRATOR, SOBOL, kHybridStack>(kp, sc, small_base, stack, lvstack, ovf, nullptr, blockIdx.x, lane, n_rays, n_shaded, n_paths, cnt);

    // exact work counters: wave reduction, one 64-bit atomic per wave and counter
    uint32_t r_paths = wave_sum(n_paths), r_rays = wave_sum(n_rays), r_shaded = wave_sum(n_shaded);
    unsigned long long* const stats = stat_row(kp.stats, blockIdx.x);
    if (lane == 0) {
        atomicAdd(&stats[kStatPaths], (unsigned long long)r_paths);
        atomicAdd(&stats[kStatRays], (unsigned long long)r_rays);
        atomicAdd(&stats[kStatShaded], (unsigned long long)r_shaded);
    }
    if (STATS) {
        uint32_t v[8] = {cnt.n_descend, cnt.n_return, cnt.leaf[0], cnt.leaf[1], cnt.leaf[2], cnt.leaf[3],
                         cnt.hit_triangle, cnt.hit_cube};
        const int slot[8] = {kStatDescend, kStatReturn, kStatLeafSphere, kStatLeafSquare, kStatLeafCube,
                             kStatLeafTriangle, kStatHitTriangle, kStatHitCube};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            uint32_t r = wave_sum(v[i]);
            if (lane == 0) atomicAdd(&stats[slot[i]], (unsigned long long)r);
        }
        for (int i = 0; i < kProfCount; ++i) {       // divergence profile: lanes and wavefronts per site
            uint32_t rl = wave_sum(cnt.prof_lane[i]), rw = wave_sum(cnt.prof_wave[i]);
            unsigned long long rc = cnt.prof_cycles[i];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) rc += __shfl_xor(rc, off);
            if (lane == 0) {
                atomicAdd(&stats[kStatCount + 3 * i], (unsigned long long)rl);
                atomicAdd(&stats[kStatCount + 3 * i + 1], (unsigned long long)rw);
                atomicAdd(&stats[kStatCount + 3 * i + 2], rc);
            }
        }
    }
}
template <bool LDS, bool STATS, int INTEGRATOR, bool SOBOL>
__global__ void __launch_bounds__(kBlock, STATS ? 1 : (INTEGRATOR == TRC_INTEGRATOR_VOLUME ? TRC_VOLUME_WAVES : (INTEGRATOR == TRC_INTEGRATOR_MIS ? (LDS ? TRC_MIS_WAVES_LDS : TRC_MIS_WAVES) : (LDS ? TRC_PATH_WAVES : TRC_PATH_WAVES_GLOBAL)))) k_render(const KRender kp) { render_workgroup<LDS, STATS, INTEGRATOR, SOBOL>(kp); }

// kernelPathTracing on a tree that is READ FROM MEMORY (mesh scenes), production launches of >= 8 spp: persistent
// workgroups.  With one wavefront per workgroup every wavefront stages its own copy of the top of the tree, and 16-24 copies
// per CU leave room for ~45 nodes (5 levels) each.  Here a workgroup is as many wavefronts as one (tracePath: half a) CU
// holds, they stage ONE prefix -- 30-60 KB, the top 9-10 levels -- and then every wavefront on its own pulls pixel
// blocks from a device-wide queue in the launch order (longest first) until it is empty, so no wavefront slot waits
// for a sibling (what cost the 4-wavefront workgroups of docs/HISTORY.md section 9 their 20 %).  Same blocks, same arithmetic per
// lane.  Measured (profiles/r02/persistent_workgroups.txt): 2.4-3 % on configs 3 / 4 and the traceVolume scene -- most
// of a mesh ray's steps are deep in the tree, below any prefix.  Workgroup shapes (trc_render_config.hpp): tracePath 4 wavefronts x 7
// per CU -- seven waves per SIMD at 72 registers, the one shape of 28 wavefronts that packs (round 5; rounds 2-4 ran 12 x 2 and 16 x 2) --,
// traceMIS 16 x 2 with 8 stack entries per lane in LDS, traceVolume 16 x 1.
template <int INTEGRATOR, bool SOBOL>
__global__ void __launch_bounds__(64 * pwg_waves(INTEGRATOR), pwg_waves(INTEGRATOR) * pwg_per_cu(INTEGRATOR) / 4) k_render_pwg(const KRender kp) {
    const DScene& sc = kp.ks.sc;
    {
        const uint4* src = reinterpret_cast<const uint4*>(sc.blob);
        uint4* dst = reinterpret_cast<uint4*>(trc_smem);
        const uint32_t n16 = sc.lds_dwords >> 2;
        for (uint32_t i = threadIdx.x; i < n16; i += blockDim.x) dst[i] = src[i];
        __syncthreads();
    }
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    constexpr bool kHybridStack = hybrid_stack(INTEGRATOR);
    constexpr uint32_t kRows = pwg_park_rows(INTEGRATOR);       // a wavefront's LDS: stack_lds stack rows, then the park rows (render_block)
    constexpr bool kPark = kRows != 0u;
    uint32_t* stack = trc_smem + sc.lds_dwords + wave * (sc.stack_lds + kRows) * kBlock + lane;
    uint32_t* park = stack + sc.stack_lds * kBlock;
    uint32_t* ovf = kHybridStack ? kp.stack_ovf + ((size_t)blockIdx.x * (blockDim.x >> 6) + wave) * sc.stack_ovf_rows * kBlock + lane : nullptr;
    uint32_t n_paths = 0;
    TravCounters cnt;
    counters_zero(cnt);
    const uint32_t n_entries = kp.n_launch ? *kp.n_launch : kp.n_tiles;
    uint32_t r_rays, r_shaded;
    if constexpr (kPark) {
        park[kParkRays * kBlock] = 0u; park[kParkShaded * kBlock] = 0u;
        LdsCount n_rays{park + kParkRays * kBlock}, n_shaded{park + kParkShaded * kBlock};
        for (;;) {
            uint32_t slot = 0;
            if (lane == 0) slot = atomicAdd(kp.queue, 1u);
            slot = __builtin_amdgcn_readfirstlane(slot);
            if (slot >= n_entries) break;
            render_block<false, false, INTEGRATOR, SOBOL, kHybridStack, (int)kRows>(kp, sc, trc_smem, stack, nullptr, ovf, park, slot, lane, n_rays, n_shaded, n_paths, cnt);
        }
        r_rays = wave_sum(park[kParkRays * kBlock]); r_shaded = wave_sum(park[kParkShaded * kBlock]);
    } else {
        uint32_t n_rays = 0, n_shaded = 0;
        for (;;) {
            uint32_t slot = 0;
            if (lane == 0) slot = atomicAdd(kp.queue, 1u);
            slot = __builtin_amdgcn_readfirstlane(slot);
            if (slot >= n_entries) break;
            render_block<false, false, INTEGRATOR, SOBOL, kHybridStack>(kp, sc, trc_smem, stack, nullptr, ovf, nullptr, slot, lane, n_rays, n_shaded, n_paths, cnt);
        }
        r_rays = wave_sum(n_rays); r_shaded = wave_sum(n_shaded);
    }
    const uint32_t r_paths = wave_sum(n_paths);
    if (lane == 0) {
        unsigned long long* const stats = stat_row(kp.stats, blockIdx.x * (blockDim.x >> 6) + wave);
        atomicAdd(&stats[kStatPaths], (unsigned long long)r_paths);
        atomicAdd(&stats[kStatRays], (unsigned long long)r_rays);
        atomicAdd(&stats[kStatShaded], (unsigned long long)r_shaded);
    }
}

// kernelPathTracing for launches of FEW samples per pixel (the reference's own pattern is one per dispatch): with one 8x8
// block per wavefront a lane that has finished its few samples waits for the longest path of the wavefront -- at 1 spp
// the wavefront runs ~9 iterations for 1.7 rays per lane.  Here a wavefront owns a strip of `kp.strip` consecutive
// blocks of the list and every lane walks its own pixel of block after block, so a lane whose pixel is done starts
// the same pixel of the next block at once (path regeneration across pixels instead of across samples).  The strip
// is the unit of the adaptive launch order.  Pixels are independent: the frame is k_render's, bit for bit.
template <bool LDS, int INTEGRATOR, bool SOBOL>
__global__ void __launch_bounds__(kBlock, INTEGRATOR == TRC_INTEGRATOR_VOLUME ? 3 : (INTEGRATOR == TRC_INTEGRATOR_PATH ? TRC_STRIP_PATH_WAVES : 4)) k_render_strip(const KRender kp) {
    const DScene& sc = kp.ks.sc;
    const uint32_t* small_base = stage_scene(sc);
    uint32_t* stack = lane_stack(sc);
    const uint64_t t_start = clock64();
    const uint32_t canon = kp.order ? kp.order[blockIdx.x] : blockIdx.x;     // strip index
    const uint32_t lane = threadIdx.x;
    const uint32_t W = kp.fr.width, H = kp.fr.height;
    // the strip's pixels form one pool: pixel p = lane (p mod block size) of block (p / block size); a lane whose pixel is
    // done takes the next unclaimed one, so no lane waits for "its" pixel of the next block while others still trace
    const uint32_t blk0 = canon * kp.strip;
    const uint32_t bshift = 2u * kp.blk_shift;                               // log2(pixels per block): 6 or 4
    const uint32_t pool_end = (min(blk0 + kp.strip, kp.n_tiles) - blk0) << bshift;
    uint32_t pool_next = 0;                                                  // wave-uniform: next unclaimed pool index

    uint32_t n_rays = 0, n_shaded = 0, n_paths = 0;
    TravCounters cnt;
    counters_zero(cnt);

    PathCtx cx;
    cx.S = make_scene_ref(sc, small_base);
    constexpr bool kHybridStack = !LDS && hybrid_stack(INTEGRATOR);
    if (kHybridStack) cx.S.ovf = kp.stack_ovf + (size_t)blockIdx.x * sc.stack_ovf_rows * kBlock + lane;
    cx.root_min = f3(kp.ks.root_box[0], kp.ks.root_box[1], kp.ks.root_box[2]);
    cx.root_max = f3(kp.ks.root_box[3], kp.ks.root_box[4], kp.ks.root_box[5]);
    cx.sh.mats = small_base + sc.off_materials;
    cx.ambient = f3(kp.ambient[0], kp.ambient[1], kp.ambient[2]);
    cx.env.rgb = kp.env_rgb; cx.env.w = kp.env_w; cx.env.h = kp.env_h;
    cx.stack = stack;
    cx.lvstack = stack;
    cx.max_depth = kp.max_depth;
    cx.density = kp.density;
    cx.dinfo = kp.dinfo;
    cx.occupancy = kp.occupancy;
    if (SOBOL) { cx.sobol32 = kp.sobol32; cx.sobol_vdc = kp.sobol_vdc; cx.sobol_m = kp.sobol_m; cx.sobol_res = 1u << kp.sobol_m; }

    PathState ps;
    Pcg rng;
    uint4 texel;
    F3 cached = f3(0);
    float u = 0, v = 0;
    uint32_t s = 0, pix = 0;
    uint64_t state_after_cast = 0;
    bool alive = false, want = true;                  // want: this lane needs a (new) pixel

    auto begin_sample = [&]() {                       // castRay, then (SOBOL) the sampler of this frame: Render.metal:527-530
        rng.state = ((uint64_t)texel.z << 32) | texel.w;      // the two words trade roles every frame (B-1)
        rng.inc = ((uint64_t)texel.x << 32) | texel.y;
        path_begin(ps, cast_ray(kp.cam, u, v, rng), kp.max_depth);
        if (SOBOL) {
            state_after_cast = rng.state;
            ps.sobol_index = sobol_interval_to_index(cx, (uint64_t)(kp.frame0 + s));
            ps.sobol_dim = 0;
        }
    };
    // hands pool indices to the lanes that want one (called where the whole wavefront is converged); a lane whose index
    // falls outside the frame (ragged edge blocks) simply asks again in the next round
    auto deal_pixels = [&]() {
        const unsigned long long m = __ballot(want);
        if (m == 0ull) return;
        const uint32_t mine = pool_next + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        pool_next += (uint32_t)__popcll(m);
        if (!want) return;
        if (mine >= pool_end) { want = false; return; }                        // the pool is empty: this lane is done
        const uint32_t tile = kp.tiles[blk0 + (mine >> bshift)];
        const uint32_t l = mine & ((1u << bshift) - 1u), bs = kp.blk_shift;
        const uint32_t px = ((tile & 0xFFFFu) << bs) + (l & ((1u << bs) - 1u));
        const uint32_t py = ((tile >> 16) << bs) + (l >> bs);
        if (px >= W || py >= H) return;                                        // not a pixel: ask again
        pix = py * W + px;
        texel = reinterpret_cast<const uint4*>(kp.fr.rng)[pix];
        const float4 acc = reinterpret_cast<const float4*>(kp.fr.accum)[pix];
        cached = f3(acc.x, acc.y, acc.z);
        u = (float)px / (float)W;                                              // no sub-pixel jitter (B-2)
        v = (float)(py % kp.view_height) / (float)kp.view_height;
        if (SOBOL) { cx.sobol_xy[0] = px; cx.sobol_xy[1] = py % kp.view_height; }
        s = 0;
        want = false;
        alive = true;
        begin_sample();
    };
    auto finish_sample = [&](F3 color) {
        const bool bad = is_inf(color.x) || is_nan(color.x) || is_inf(color.y) || is_nan(color.y) ||
                         is_inf(color.z) || is_nan(color.z);
        if (bad) color = f3(0);                                         // :537-538
        const uint32_t frame = kp.frame0 + s;
        cached = (cached * (float)frame + color) / (float)(frame + 1);  // running mean, :540-541
        if (SOBOL) rng.state = state_after_cast;
        texel.y = (uint32_t)rng.state; texel.x = (uint32_t)(rng.state >> 32);
        texel.w = (uint32_t)rng.inc;   texel.z = (uint32_t)(rng.inc >> 32);
        n_paths++;
        if (++s == kp.spp) {                                            // pixel done: write it back, take the next block's
            float4 out; out.x = cached.x; out.y = cached.y; out.z = cached.z; out.w = 1.0f;
            reinterpret_cast<float4*>(kp.fr.accum)[pix] = out;
            reinterpret_cast<uint4*>(kp.fr.rng)[pix] = texel;
            alive = false;
            want = true;
        } else {
            begin_sample();
        }
    };

    for (;;) {                                        // wave-uniform loop: every lane stays in it until nobody has or wants work
        deal_pixels();
        if (__ballot(alive || want) == 0ull) break;
        if (alive) {
            constexpr bool kVolume = INTEGRATOR == TRC_INTEGRATOR_VOLUME;
            constexpr int kDefer = LDS ? TRC_DEFER_LDS : TRC_DEFER_GLOBAL;
            bool hitted = true;
            if (!(kVolume && TRC_TRACK_SLICE > 0 && ps.tracking)) {       // render_block's loop above
                n_rays++;
                hitted = scene_hit<LDS, false, false, false, kVolume, kHybridStack, kDefer>(cx.S, cx.root_min, cx.root_max, ps.ray, ps.rec, FLT_MAX,
                                                                          cx.stack, cx.lvstack, cnt);
            }
            F3 color;
            const bool finished = (INTEGRATOR == TRC_INTEGRATOR_PATH)
                                      ? path_step<false, SOBOL>(cx, ps, hitted, rng, cnt, n_shaded, color)
                                      : mis_step<LDS, false, kVolume, SOBOL, kHybridStack>(cx, ps, hitted, rng, cnt, n_rays, n_shaded, color);
            if (finished) finish_sample(color);
        }
    }
    uint32_t r_paths = wave_sum(n_paths), r_rays = wave_sum(n_rays), r_shaded = wave_sum(n_shaded);
    if (lane == 0) {
        kp.block_cost[canon] = (uint32_t)min((unsigned long long)(clock64() - t_start) / kp.cost_div, 0xFFFFFFull);
        unsigned long long* const stats = stat_row(kp.stats, blockIdx.x);
        atomicAdd(&stats[kStatPaths], (unsigned long long)r_paths);
        atomicAdd(&stats[kStatRays], (unsigned long long)r_rays);
        atomicAdd(&stats[kStatShaded], (unsigned long long)r_shaded);
    }
}

