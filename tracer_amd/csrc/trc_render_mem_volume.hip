// trc_render_mem_volume.hip -- traceVolume on trees read from memory (the participating-media scene): one-wavefront workgroups, strips
// and the persistent workgroups.  Its own translation unit because it is compiled with -mllvm -disable-machine-sink like tracePath's
// (Makefile: EXTRA_trc_render_mem_volume; 35.97 -> 35.61 ms per 16-spp launch, profiles/r05/ab_flags_volume.txt), which traceMIS, its
// former neighbour in trc_render_mem.hip, does not want.  Definitions: trc_render_kernels.hpp; launched from trc_abi.hip.
#ifndef TRC_FAST_UNARY
#define TRC_FAST_UNARY 1
#endif
#include "trc_render_kernels.hpp"

#define TRC_INST_RENDER(S, I, B) template __global__ void k_render<false, S, I, B>(const KRender)
#define TRC_INST_STRIP(I, B) template __global__ void k_render_strip<false, I, B>(const KRender)
// exactly the instantiations launch_render<> picks from (trc_abi.hip)
TRC_INST_RENDER(false, TRC_INTEGRATOR_VOLUME, false); TRC_INST_RENDER(true, TRC_INTEGRATOR_VOLUME, false);
TRC_INST_STRIP(TRC_INTEGRATOR_VOLUME, false);
template __global__ void k_render_pwg<TRC_INTEGRATOR_VOLUME, false>(const KRender);
