// trc_render_mem_path.hip -- tracePath on trees read from memory (BASELINE config 4 and every mesh scene under the default integrator):
// one-wavefront workgroups, strips and the persistent workgroups, at 7 waves per SIMD (trc_render_config.hpp).  Its own translation
// unit because it is compiled with -mllvm -disable-machine-sink (Makefile: EXTRA_trc_render_mem_path): machine sinking moves
// computations into the branches behind the traversal loop and with them their operands' live ranges across it -- without it config 4 runs
// 22.90 -> 22.55 ms per 32-spp launch and the scene beyond the Infinity Cache 18.29 -> 17.16, while traceMIS loses 1.5 % and keeps the
// default (profiles/r05/ab_flags*.txt).  Same instructions' results either way: scheduling.  Definitions: trc_render_kernels.hpp.
#ifndef TRC_FAST_UNARY
#define TRC_FAST_UNARY 1
#endif
#include "trc_render_kernels.hpp"

#define TRC_INST_RENDER(S, I, B) template __global__ void k_render<false, S, I, B>(const KRender)
#define TRC_INST_STRIP(I, B) template __global__ void k_render_strip<false, I, B>(const KRender)
TRC_INST_RENDER(false, TRC_INTEGRATOR_PATH, false);   TRC_INST_RENDER(true, TRC_INTEGRATOR_PATH, false);   TRC_INST_RENDER(false, TRC_INTEGRATOR_PATH, true);
TRC_INST_STRIP(TRC_INTEGRATOR_PATH, false);  TRC_INST_STRIP(TRC_INTEGRATOR_PATH, true);
template __global__ void k_render_pwg<TRC_INTEGRATOR_PATH, false>(const KRender);
template __global__ void k_render_pwg<TRC_INTEGRATOR_PATH, true>(const KRender);
