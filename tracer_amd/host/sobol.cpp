// sobol.cpp -- C-ABI view of include/trc_sobol.h (tables of pbrt::SobolSampler, SobolSampler.hh:126-160) for hosts
// and tests; the device library builds the same tables itself.
#include "tracer_abi.h"
#include "trc_sobol.h"

extern "C" void trc_host_sobol_matrices32(uint32_t* out) { trc_sobol_matrices32(out); }

extern "C" trc_status trc_host_sobol_interval_tables(uint32_t log2res, uint64_t* vdc, uint64_t* inv) {
    if (!vdc || !inv) return TRC_ERR_INVALID_ARG;
    return trc_sobol_interval_tables(log2res, vdc, inv) == 0 ? TRC_OK : TRC_ERR_INVALID_ARG;
}
