// trc_render_mem.hip -- the render kernels of trees read from memory (mesh scenes) for traceMIS and traceVolume (BASELINE config 3,
// the participating-media scene): one-wavefront workgroups, strips and the persistent workgroups.  tracePath's are in
// trc_render_mem_path.hip, a translation unit of its own since round 5 because the two families want different compiler options
// (Makefile: EXTRA_*).  Compiled WITH dev_vec.hpp's guard-free forms since their guards became one or two instructions
// (profiles/r04/guard_cost_ab.txt).  Definitions: trc_render_kernels.hpp; launched from trc_abi.hip.
#ifndef TRC_FAST_UNARY
#define TRC_FAST_UNARY 1
#endif
#include "trc_render_kernels.hpp"

#define TRC_INST_RENDER(S, I, B) template __global__ void k_render<false, S, I, B>(const KRender)
#define TRC_INST_STRIP(I, B) template __global__ void k_render_strip<false, I, B>(const KRender)
// exactly the instantiations launch_render<> picks from (trc_abi.hip)
TRC_INST_RENDER(false, TRC_INTEGRATOR_MIS, false);    TRC_INST_RENDER(true, TRC_INTEGRATOR_MIS, false);    TRC_INST_RENDER(false, TRC_INTEGRATOR_MIS, true);
TRC_INST_RENDER(false, TRC_INTEGRATOR_VOLUME, false); TRC_INST_RENDER(true, TRC_INTEGRATOR_VOLUME, false);
TRC_INST_STRIP(TRC_INTEGRATOR_MIS, false);   TRC_INST_STRIP(TRC_INTEGRATOR_MIS, true);
TRC_INST_STRIP(TRC_INTEGRATOR_VOLUME, false);
template __global__ void k_render_pwg<TRC_INTEGRATOR_MIS, false>(const KRender);
template __global__ void k_render_pwg<TRC_INTEGRATOR_MIS, true>(const KRender);
template __global__ void k_render_pwg<TRC_INTEGRATOR_VOLUME, false>(const KRender);
