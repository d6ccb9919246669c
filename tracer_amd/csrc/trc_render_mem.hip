// trc_render_mem.hip -- the render kernels of trees read from memory (mesh scenes) for traceMIS (BASELINE config 3): one-wavefront
// workgroups, strips and the persistent workgroups.  tracePath's are in trc_render_mem_path.hip and traceVolume's in
// trc_render_mem_volume.hip, translation units of their own since round 5 because the families want different compiler options
// (Makefile: EXTRA_*; this one keeps the defaults: -disable-machine-sink costs traceMIS 1.5 %).  Compiled WITH dev_vec.hpp's guard-free forms since their guards became one or two instructions
// (profiles/r04/guard_cost_ab.txt).  Definitions: trc_render_kernels.hpp; launched from trc_abi.hip.
#ifndef TRC_FAST_UNARY
#define TRC_FAST_UNARY 1
#endif
#include "trc_render_kernels.hpp"

#define TRC_INST_RENDER(S, I, B) template __global__ void k_render<false, S, I, B>(const KRender)
#define TRC_INST_STRIP(I, B) template __global__ void k_render_strip<false, I, B>(const KRender)
// exactly the instantiations launch_render<> picks from (trc_abi.hip)
TRC_INST_RENDER(false, TRC_INTEGRATOR_MIS, false);    TRC_INST_RENDER(true, TRC_INTEGRATOR_MIS, false);    TRC_INST_RENDER(false, TRC_INTEGRATOR_MIS, true);
TRC_INST_STRIP(TRC_INTEGRATOR_MIS, false);   TRC_INST_STRIP(TRC_INTEGRATOR_MIS, true);
template __global__ void k_render_pwg<TRC_INTEGRATOR_MIS, false>(const KRender);
template __global__ void k_render_pwg<TRC_INTEGRATOR_MIS, true>(const KRender);
