// Does a VALU instruction of a wave64 cost less when only part of the EXEC mask is set?  (gfx950)
// One wave per SIMD slot, a long dependent chain of v_mul/v_add under different lane masks.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void __launch_bounds__(64) chain(float* out, uint64_t mask, int iters, int ilp) {
    const uint32_t lane = threadIdx.x;
    float a = 1.0f + lane * 1e-7f, b = 0.5f, c = 0.25f, d = 0.125f;
    if ((mask >> lane) & 1ull) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                a = a * 1.0000001f + 1e-9f;
                if (ilp > 1) { b = b * 1.0000001f + 1e-9f; c = c * 1.0000001f + 1e-9f; d = d * 1.0000001f + 1e-9f; }
            }
        }
    }
    out[blockIdx.x * 64 + lane] = a + b + c + d;
}
int main() {
    float* out; hipMalloc(&out, 4096 * 16 * 64 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct { const char* name; uint64_t mask; } cases[] = {
        {"all 64", ~0ull}, {"low 32", 0xFFFFFFFFull}, {"low 16", 0xFFFFull}, {"lane 0", 1ull},
        {"every 4th (16 lanes)", 0x1111111111111111ull}, {"16 lanes: 4 per quarter", 0x000F000F000F000Full}, {"upper 16", 0xFFFF000000000000ull}};
    for (int ilp : {1, 4})
        for (auto& cs : cases) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(chain, dim3(256 * 16), dim3(64), 0, 0, out, cs.mask, 20000, ilp);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep) printf("ilp %d  %-26s %8.3f ms\n", ilp, cs.name, ms);
            }
        }
    return 0;
}
