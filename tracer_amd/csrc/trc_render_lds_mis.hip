// trc_render_lds_mis.hip -- traceMIS and traceVolume on scenes whose whole tree is staged in LDS (default compiler options; tracePath's
// kernels of the same scenes are in trc_render_lds.hip).  Definitions: trc_render_kernels.hpp; launched from trc_abi.hip.
#ifndef TRC_FAST_UNARY
#define TRC_FAST_UNARY 1
#endif
#include "trc_render_kernels.hpp"

#define TRC_INST_RENDER(S, I, B) template __global__ void k_render<true, S, I, B>(const KRender)
#define TRC_INST_STRIP(I, B) template __global__ void k_render_strip<true, I, B>(const KRender)
TRC_INST_RENDER(false, TRC_INTEGRATOR_MIS, false);    TRC_INST_RENDER(true, TRC_INTEGRATOR_MIS, false);    TRC_INST_RENDER(false, TRC_INTEGRATOR_MIS, true);
TRC_INST_RENDER(false, TRC_INTEGRATOR_VOLUME, false); TRC_INST_RENDER(true, TRC_INTEGRATOR_VOLUME, false);
TRC_INST_STRIP(TRC_INTEGRATOR_MIS, false);   TRC_INST_STRIP(TRC_INTEGRATOR_MIS, true);
TRC_INST_STRIP(TRC_INTEGRATOR_VOLUME, false);
