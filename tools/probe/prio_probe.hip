// prio_probe.hip -- does s_setprio let one wavefront of a saturated SIMD run at a lone wavefront's speed?
// 5 one-wavefront workgroups per SIMD run the same VALU (+ SALU 3:1) loop; the first 1024 workgroups raise their priority.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
template <int PRIO> __global__ void __launch_bounds__(64) k_prio(float* out, unsigned long long* dur, int iters, int n_prio) {
    float a = threadIdx.x, b = 1.0f, c = 2.0f; int s0 = blockIdx.x;
    if (PRIO && (int)blockIdx.x < n_prio) __builtin_amdgcn_s_setprio(3);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("v_add_f32 %0, %0, %0\n v_add_f32 %1, %1, %1\n v_add_f32 %2, %2, %2\n s_add_u32 %3, %3, 1" : "+v"(a), "+v"(b), "+v"(c), "+s"(s0) : : "scc");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) dur[blockIdx.x] = t1 - t0;
    if (a + b + c + s0 == 12345.678f) out[0] = a;
}
int main() {
    float* d; unsigned long long* dd; (void)hipMalloc(&d, 64); (void)hipMalloc(&dd, 8 * 8192);
    std::vector<unsigned long long> h(8192);
    for (int prio = 0; prio < 2; ++prio)
        for (int w : {1, 5}) {
            const int n = 1024 * w;
            for (int rep = 0; rep < 2; ++rep) {
                if (prio) hipLaunchKernelGGL(k_prio<1>, dim3(n), dim3(64), 0, 0, d, dd, 20000, 1024);
                else hipLaunchKernelGGL(k_prio<0>, dim3(n), dim3(64), 0, 0, d, dd, 20000, 1024);
                (void)hipDeviceSynchronize();
            }
            (void)hipMemcpy(h.data(), dd, 8 * n, hipMemcpyDeviceToHost);
            double first = 0, rest = 0;
            for (int i = 0; i < 1024; ++i) first += h[i];
            for (int i = 1024; i < n; ++i) rest += h[i];
            printf("prio %d, %d waves/SIMD: first 1024 workgroups %.0f ticks, the others %.0f ticks\n", prio, w, first / 1024, n > 1024 ? rest / (n - 1024) : 0.0);
        }
    return 0;
}
