// issue_probe.hip -- how many instructions per cycle does one gfx950 SIMD issue, and do scalar instructions share the
// slots of vector instructions?  W one-wavefront workgroups per SIMD run a loop of (a) VALU only, (b) VALU + SALU 3:1,
// (c) VALU + SALU 1:1, (d) SALU only; prints wavefront instructions per cycle per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int MODE> __global__ void __launch_bounds__(64) k_issue(float* out, int iters) {
    float a = threadIdx.x, b = 1.0f, c = 2.0f, d = 3.0f;
    typedef float v2f_t __attribute__((ext_vector_type(2)));
    v2f_t pa = {a, b}, pb = {b, c}, pc = {c, d}, pd = {d, a};
    int s0 = blockIdx.x, s1 = 1, s2 = 2, s3 = 3;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) asm volatile("v_add_f32 %0, %0, %0\n v_add_f32 %1, %1, %1\n v_add_f32 %2, %2, %2\n v_add_f32 %3, %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        } else if (MODE == 1) {
#pragma unroll
            for (int k = 0; k < 8; ++k) asm volatile("v_add_f32 %0, %0, %0\n v_add_f32 %1, %1, %1\n v_add_f32 %2, %2, %2\n s_add_u32 %3, %3, 1" : "+v"(a), "+v"(b), "+v"(c), "+s"(s0) : : "scc");
        } else if (MODE == 2) {
#pragma unroll
            for (int k = 0; k < 8; ++k) asm volatile("v_add_f32 %0, %0, %0\n s_add_u32 %2, %2, 1\n v_add_f32 %1, %1, %1\n s_add_u32 %3, %3, 1" : "+v"(a), "+v"(b), "+s"(s0), "+s"(s1) : : "scc");
        } else if (MODE == 3) {
#pragma unroll
            for (int k = 0; k < 8; ++k) asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
        } else if (MODE == 5) {   // packed FP32: two lanes' worth per instruction -- does it take two issue slots?
            typedef float v2f __attribute__((ext_vector_type(2)));
            static_assert(sizeof(v2f) == 8, "");
#pragma unroll
            for (int k = 0; k < 8; ++k) asm volatile("v_pk_mul_f32 %0, %0, %0\n v_pk_mul_f32 %1, %1, %1\n v_pk_add_f32 %2, %2, %2\n v_pk_add_f32 %3, %3, %3" : "+v"(pa), "+v"(pb), "+v"(pc), "+v"(pd));
        } else {   // dependent VALU chain: a lone wavefront's issue latency
#pragma unroll
            for (int k = 0; k < 8; ++k) asm volatile("v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0" : "+v"(a));
        }
    }
    if (a + b + c + d + s0 + s1 + s2 + s3 + pa.x + pb.y + pc.x + pd.y == 12345.678f) out[0] = a;
}
template <int MODE> void run(const char* what, float* d, double ghz) {
    const int iters = 20000;
    printf("%-28s", what);
    for (int w : {1, 2, 3, 4, 5, 6, 8}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k_issue<MODE>, dim3(1024 * w), dim3(64), 0, 0, d, 100);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_issue<MODE>, dim3(1024 * w), dim3(64), 0, 0, d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1); if (hipGetLastError() != hipSuccess) printf(" [launch error]");
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_simd = (double)w * iters * 32.0;
        printf("  W=%d: %.3f", w, instr_per_simd / (ms * 1e-3 * ghz * 1e9));
    }
    printf("   (instr / cycle / SIMD at %.2f GHz)\n", ghz);
}
int main(int argc, char** argv) {
    const double ghz = argc > 1 ? atof(argv[1]) : 2.4;
    float* d; hipMalloc(&d, 64);
    run<0>("VALU only", d, ghz);
    run<1>("VALU:SALU 3:1", d, ghz);
    run<2>("VALU:SALU 1:1", d, ghz);
    run<3>("SALU only", d, ghz);
    run<4>("dependent VALU chain", d, ghz);
    run<5>("packed FP32 (v_pk_mul/add)", d, ghz);
    return 0;
}
