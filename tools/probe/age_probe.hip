// age_probe.hip -- is a SIMD's issue priority by wavefront age or by wavefront slot?  Five one-wavefront workgroups per
// SIMD; the first round (slot 0 of every SIMD) is short, a sixth round arrives in the slots it frees.  Reported: average
// duration of each round in units of a lone wavefront's duration for the same work.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void __launch_bounds__(64) k_age(float* out, unsigned long long* dur, unsigned* hw, int iters, int short_until, int prio_from) {
    float a = threadIdx.x, b = 1.0f, c = 2.0f; int s0 = blockIdx.x;
    const int n = (int)blockIdx.x < short_until ? iters / 8 : iters;
    if ((int)blockIdx.x >= prio_from) __builtin_amdgcn_s_setprio(3);
    asm volatile("v_mov_b32 v95, 0" : : : "v95");      // 96 VGPRs: five wavefronts per SIMD, like k_render
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("v_add_f32 %0, %0, %0\n v_add_f32 %1, %1, %1\n v_add_f32 %2, %2, %2\n s_add_u32 %3, %3, 1" : "+v"(a), "+v"(b), "+v"(c), "+s"(s0) : : "scc");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { dur[blockIdx.x] = t1 - t0; hw[blockIdx.x] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11)); }
    if (a + b + c + s0 == 12345.678f) out[0] = a;
}
int main() {
    float* d; unsigned long long* dd; unsigned* dh; (void)hipMalloc(&d, 64); (void)hipMalloc(&dd, 8 * 8192); (void)hipMalloc(&dh, 4 * 8192);
    std::vector<unsigned long long> h(8192); std::vector<unsigned> hw(8192);
    const int iters = 20000;
    hipLaunchKernelGGL(k_age, dim3(1024), dim3(64), 0, 0, d, dd, dh, iters, 0, 1 << 30); (void)hipDeviceSynchronize();
    hipLaunchKernelGGL(k_age, dim3(1024), dim3(64), 0, 0, d, dd, dh, iters, 0, 1 << 30); (void)hipDeviceSynchronize();
    (void)hipMemcpy(h.data(), dd, 8 * 1024, hipMemcpyDeviceToHost);
    double lone = 0; for (int i = 0; i < 1024; ++i) lone += h[i]; lone /= 1024;
    for (int prio_from : {1 << 30, 4096, 3072}) for (int short_until : {0, 1024}) {
        printf("priority 3 for workgroups >= %d; ", prio_from);
        const int n = 6 * 1024;
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k_age, dim3(n), dim3(64), 0, 0, d, dd, dh, iters, short_until, prio_from); (void)hipDeviceSynchronize(); }
        (void)hipMemcpy(h.data(), dd, 8 * n, hipMemcpyDeviceToHost); (void)hipMemcpy(hw.data(), dh, 4 * n, hipMemcpyDeviceToHost);
        printf("first round %s: duration / lone duration of the same work, and wave slot ids seen, per round:", short_until ? "SHORT (1/8)" : "as long as the others");
        for (int r = 0; r < 6; ++r) {
            double s = 0; unsigned slots = 0;
            for (int i = 1024 * r; i < 1024 * (r + 1); ++i) { s += h[i]; slots |= 1u << (hw[i] & 15u); }
            const double work = (r == 0 && short_until) ? lone / 8 : lone;
            printf("  r%d %.2f (slots %#x)", r, s / 1024 / work, slots);
        }
        printf("\n");
    }
    return 0;
}
