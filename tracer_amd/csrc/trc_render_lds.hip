// trc_render_lds.hip -- tracePath on scenes whose whole tree is staged in LDS (the Cornell scenes: BASELINE config 2 -- the bench's
// kernel), instantiated with the guard-free reciprocal / square root of dev_vec.hpp (TRC_FAST_UNARY, bit-identical:
// tests/test_gpu_unary.py).  traceMIS / traceVolume on such scenes: trc_render_lds_mis.hip -- two translation units since round 5
// because this one is compiled with -mllvm -amdgpu-use-amdgpu-trackers (Makefile: EXTRA_trc_render_lds; config 2 16.64 -> 16.50 ms,
// traceMIS on the same scene would lose 1.6 %: profiles/r05/ab_flags*.txt).  Definitions: trc_render_kernels.hpp; launched from trc_abi.hip.
#ifndef TRC_FAST_UNARY
#define TRC_FAST_UNARY 1
#endif
#include "trc_render_kernels.hpp"

// tracePath on an LDS-resident tree at one more wavefront per SIMD, for launch lists many times the wavefront slots (trc_render_config.hpp)
__global__ void __launch_bounds__(kBlock, TRC_PATH_WAVES_DENSE) k_render_dense(const KRender kp) {
    render_workgroup<true, false, TRC_INTEGRATOR_PATH, false, TRC_PARK_DENSE>(kp);      // + per-pixel state parked in LDS rows (render_block)
}

#define TRC_INST_RENDER(S, I, B) template __global__ void k_render<true, S, I, B>(const KRender)
#define TRC_INST_STRIP(I, B) template __global__ void k_render_strip<true, I, B>(const KRender)
// exactly the instantiations launch_render<> picks from (trc_abi.hip)
TRC_INST_RENDER(false, TRC_INTEGRATOR_PATH, false);   TRC_INST_RENDER(true, TRC_INTEGRATOR_PATH, false);   TRC_INST_RENDER(false, TRC_INTEGRATOR_PATH, true);
TRC_INST_STRIP(TRC_INTEGRATOR_PATH, false);  TRC_INST_STRIP(TRC_INTEGRATOR_PATH, true);
