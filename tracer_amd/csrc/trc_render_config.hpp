// trc_render_config.hpp -- compile-time shape of the render kernels (launch bounds, persistent-workgroup geometry, stack and
// record policy per integrator), shared by the translation units that DEFINE the kernels (trc_render_lds.hip, trc_render_mem.hip)
// and the one that plans and launches them (trc_abi.hip).  Every value is an A/B macro: make variant NAME=x DEFS=-D...
#pragma once

#include "trc_ctx.hpp"

// "test now, build the record for the winner afterwards" (dev_intersect.hpp: trav_test_leaf<DEFER>), per tree residence
// two-level traversal stack (first entries in LDS, deeper ones in global rows: dev_intersect.hpp stack_put) per integrator, on
// trees read from memory: it is what lets LDS admit the occupancy the registers allow
#ifndef TRC_MIS_HYBRID
#define TRC_MIS_HYBRID 1
#endif
#ifndef TRC_VOLUME_HYBRID
#define TRC_VOLUME_HYBRID 0
#endif
constexpr bool hybrid_stack(int integrator) {
    return integrator == TRC_INTEGRATOR_PATH || (integrator == TRC_INTEGRATOR_MIS ? TRC_MIS_HYBRID != 0 : TRC_VOLUME_HYBRID != 0);
}
#ifndef TRC_DEFER_LDS
#define TRC_DEFER_LDS 0
#endif
#ifndef TRC_DEFER_GLOBAL
#define TRC_DEFER_GLOBAL 1
#endif

// kernelPathTracing, Render.metal:495-558.  One lane per pixel, one one-wavefront workgroup per 8x8 pixel block
// (DESIGN.md 4.1), all `spp` samples fused: RNG texel and accumulator are read and
// written ONCE per pixel instead of once per sample (64 B/pixel/sample in the reference).
// wavefronts per SIMD the register allocation aims at (launch bounds), each measured (profiles/r02/
// compiler_flags_and_occupancy.txt, lds_plan_and_stack.txt): tracePath on an LDS-resident tree fits 96 VGPRs with 2
// spilled dwords (5 waves: 21.5 -> 20.8 ms on config 2; 6 waves / 80 VGPRs: 21.3-21.6); on a tree read from memory the
// sixth wave hides more latency than its spills cost (1 M triangles: 32.8 -> 31.8 ms; 7 waves: 33.4) -- provided LDS
// lets it in (plan_launch_lds); traceMIS needs 128 (5 waves: 63.2 -> 63.7 ms on config 3), traceVolume 128 (5: 55 -> 84 ms)
#ifndef TRC_PATH_WAVES
#define TRC_PATH_WAVES 5
#endif
// ... and at 6 when the launch list is many times the wavefront slots (a whole 1080p frame: 19.85 -> 19.63 ms; its 25-52 spilled
// dwords lengthen a lone block's chain, so shares of a frame -- which end on their slowest block -- keep the 5-wave kernel:
// an eighth of config 2 5.6 against 5.9 ms).  k_render_dense: tracePath, LDS-resident tree, production, no Sobol'.
#ifndef TRC_PATH_WAVES_DENSE
#define TRC_PATH_WAVES_DENSE 6
#endif
#ifndef TRC_DENSE_MIN_BLOCKS_PER_SLOT
#define TRC_DENSE_MIN_BLOCKS_PER_SLOT 4
#endif
// tracePath on a tree read from memory: SEVEN waves per SIMD (72 registers: 90 spilled values and 160 B of scratch per lane where 64
// registers spill 116 / 200 B).  Round 4 had found seven slower than eight because a workgroup of 14 wavefronts does not pack -- the
// dispatcher deals a workgroup's wavefronts to the SIMDs from SIMD 0 on, 14 = 4 + 4 + 3 + 3, and the second workgroup would put an eighth
// wavefront on a SIMD whose registers hold seven, so only ONE was resident (41.6 ms).  Four-wavefront workgroups, seven per CU, pack
// (1 + 1 + 1 + 1 each).  Round 5, alternating builds (profiles/r05/ab_shapes_config*.txt, shapes_validate.txt): config 4 23.25 -> 22.87 ms
// per 32-spp launch, teapot x 1 / x 16 / x 256 16.80 / 22.38 / 20.68 -> 16.33 / 22.04 / 18.92 ms (the scene beyond the Infinity Cache
// gains most: -8.5 %), 168.5 against 169.6 ms as named; 4 x 8, 8 x 4 at eight waves and 4 x 6 at six are slower (23.66 / 23.36 / 26.73).
#ifndef TRC_PATH_WAVES_GLOBAL
#define TRC_PATH_WAVES_GLOBAL 7
#endif
#ifndef TRC_MIS_WAVES
#define TRC_MIS_WAVES 8              // traceMIS on a tree read from memory (latency-bound: occupancy pays, round 4)
#endif
#ifndef TRC_MIS_WAVES_LDS
#define TRC_MIS_WAVES_LDS 8          // ... and on an LDS-resident one (Cornell + spheres, 32 spp: 16.7 ms at 4 waves and 128 registers, 15.9 at 5, 15.6 at 8 with 64)
#endif
#ifndef TRC_VOLUME_WAVES
#define TRC_VOLUME_WAVES 4
#endif
// persistent workgroups (k_render_pwg): wavefronts per workgroup x workgroups per CU = the waves per CU above
#ifndef TRC_PWG_WAVES_PATH
#define TRC_PWG_WAVES_PATH 4
#endif
#ifndef TRC_PWG_PER_CU_PATH
#define TRC_PWG_PER_CU_PATH 7
#endif
#ifndef TRC_PWG_WAVES_MIS
#define TRC_PWG_WAVES_MIS 16
#endif
#ifndef TRC_PWG_PER_CU_MIS
#define TRC_PWG_PER_CU_MIS 2
#endif
#ifndef TRC_PWG_WAVES_VOLUME
#define TRC_PWG_WAVES_VOLUME 16
#endif
#ifndef TRC_PWG_PER_CU_VOLUME
#define TRC_PWG_PER_CU_VOLUME 1
#endif
#ifndef TRC_STRIP_PATH_WAVES
#define TRC_STRIP_PATH_WAVES 4
#endif
constexpr int pwg_waves(int integrator) { return integrator == TRC_INTEGRATOR_PATH ? TRC_PWG_WAVES_PATH : (integrator == TRC_INTEGRATOR_MIS ? TRC_PWG_WAVES_MIS : TRC_PWG_WAVES_VOLUME); }
// entries of a lane's traversal stack that live in LDS in a persistent-workgroup launch (deeper ones: global rows, dev_intersect.hpp
// stack_push); what the stacks leave of the workgroup's LDS share is node prefix.  tracePath 10 beside its park rows (round 6: 6 / 8 / 10:
// 23.2 / 22.1 / 21.9 ms per 32-spp launch of config 4; 16 without park rows: 22.6); traceMIS 8 (6 / 8 / 10 / 12 / 16: 41.7 / 40.1 / 40.2 / 40.4 / 41.0 ms per 32-spp launch of config 3, 294.4
// against 300.5 ms as named: its shadow rays walk with one entry per level and its 16 x 2 workgroups get 48 KB of prefix instead of 16)
// PARK (round 6, trc_render_kernels.hpp::render_block): words of a lane's LDS column hold the values that are touched only where a sample
// begins or ends (running mean, (u, v), sample counter) and the two work counters, so that they are not carried -- and spilled to scratch
// -- through the walk and the shading code of every iteration.  8 rows, or 10 with the pixel's coordinates (with 8 they are worked out
// again at the end from the launch entry, read once more).  Paid for with stack entries / node prefix; per kernel family, each measured
// (profiles/r06/ab_park_*.txt): 0 = off.
enum : uint32_t { kParkCachedX = 0, kParkCachedY, kParkCachedZ, kParkU, kParkV, kParkSample, kParkRays, kParkShaded, kParkPx, kParkPy };
#ifndef TRC_PARK_PATH
#define TRC_PARK_PATH 8          // 8 rows + 12 stack entries against 10 + 10: -0.6 % on config 4 and on the 4 M-triangle scene (ab_park_rows_config4.txt)
#endif
#ifndef TRC_PARK_MIS
#define TRC_PARK_MIS 0
#endif
#ifndef TRC_PARK_VOLUME
#define TRC_PARK_VOLUME 10       // (8: +3 %, ab_park_rows_volume.txt)
#endif
#ifndef TRC_PARK_DENSE
#define TRC_PARK_DENSE 0         // k_render_dense (the headline kernel: tracePath, whole tree in LDS): 8 rows are what six waves per SIMD leave, and they
#endif                           // take its scratch from 88 to 32 bytes per lane -- and its time from 16.57 to 17.19 ms (ab_park_dense_config2.txt):
                                 // the kernel is bound by issue, and an LDS access is an issued instruction + a wait where the spill was one too.  Off.
constexpr uint32_t pwg_park_rows(int integrator) { return integrator == TRC_INTEGRATOR_PATH ? TRC_PARK_PATH : (integrator == TRC_INTEGRATOR_MIS ? TRC_PARK_MIS : TRC_PARK_VOLUME); }
static_assert((TRC_PARK_PATH == 0 || TRC_PARK_PATH == 8 || TRC_PARK_PATH == 10) && (TRC_PARK_MIS == 0 || TRC_PARK_MIS == 8 || TRC_PARK_MIS == 10) &&
              (TRC_PARK_VOLUME == 0 || TRC_PARK_VOLUME == 8 || TRC_PARK_VOLUME == 10) && (TRC_PARK_DENSE == 0 || TRC_PARK_DENSE == 8 || TRC_PARK_DENSE == 10), "park rows: 0, 8 or 10");
#ifndef TRC_PWG_STACK_LDS_PATH
#define TRC_PWG_STACK_LDS_PATH (TRC_PARK_PATH == 10 ? 10 : (TRC_PARK_PATH == 8 ? 12 : 16))
#endif
#ifndef TRC_PWG_STACK_LDS_MIS
#define TRC_PWG_STACK_LDS_MIS 8
#endif
#ifndef TRC_PWG_STACK_LDS_VOLUME
#define TRC_PWG_STACK_LDS_VOLUME 16
#endif
constexpr uint32_t pwg_stack_lds_levels(int integrator) { return integrator == TRC_INTEGRATOR_MIS ? TRC_PWG_STACK_LDS_MIS : (integrator == TRC_INTEGRATOR_PATH ? TRC_PWG_STACK_LDS_PATH : TRC_PWG_STACK_LDS_VOLUME); }
constexpr int pwg_per_cu(int integrator) { return integrator == TRC_INTEGRATOR_PATH ? TRC_PWG_PER_CU_PATH : (integrator == TRC_INTEGRATOR_MIS ? TRC_PWG_PER_CU_MIS : TRC_PWG_PER_CU_VOLUME); }

// The kernels themselves (trc_render_kernels.hpp) are instantiated in five translation units -- by tree residence and integrator family,
// so that each can be compiled with the backend options that pay for it (Makefile: EXTRA_*), and in parallel:
//   trc_render_lds.hip        tracePath, whole tree staged in LDS (Cornell scenes; the bench's kernel)       -amdgpu-use-amdgpu-trackers
//   trc_render_lds_mis.hip    traceMIS / traceVolume, whole tree staged in LDS
//   trc_render_mem_path.hip   tracePath, trees read from memory (mesh scenes)                                 -disable-machine-sink
//   trc_render_mem.hip        traceMIS, trees read from memory
//   trc_render_mem_volume.hip traceVolume, trees read from memory                                             -disable-machine-sink
template <bool LDS, bool STATS, int INTEGRATOR, bool SOBOL = false>
__global__ void k_render(const KRender kp);
__global__ void k_render_dense(const KRender kp);      // trc_render_lds.hip
template <int INTEGRATOR, bool SOBOL>
__global__ void k_render_pwg(const KRender kp);
template <bool LDS, int INTEGRATOR, bool SOBOL>
__global__ void k_render_strip(const KRender kp);
