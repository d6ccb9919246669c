// placement_probe.hip -- where does the dispatcher put one-wavefront workgroups at partial occupancy?
// Launches n workgroups of 64 threads that each stay resident for `us` microseconds and record HW_ID / XCC_ID;
// prints the histogram of wavefronts per SIMD and per CU.  hipcc --offload-arch=gfx950 -O2 -o placement_probe placement_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
__global__ void __launch_bounds__(64) k_probe(uint32_t* out, unsigned long long ticks, uint32_t lds_words) {
    extern __shared__ uint32_t lds[];
    if (lds_words) lds[threadIdx.x % lds_words] = threadIdx.x;
    const uint32_t hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));       // HW_REG_HW_ID
    const uint32_t xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11));     // HW_REG_XCC_ID[3:0]
    const unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}
int main(int argc, char** argv) {
    const uint32_t lds_bytes = argc > 1 ? atoi(argv[1]) : 4608;
    uint32_t* d; hipMalloc(&d, 8 * 20000);
    std::vector<uint32_t> h(2 * 20000);
    for (uint32_t n : {256u, 512u, 1016u, 2026u, 4050u, 5120u}) {
        hipLaunchKernelGGL(k_probe, dim3(n), dim3(64), lds_bytes, 0, d, 200000ull, lds_bytes / 4);   // ~2 ms at 100 MHz
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d, 8 * n, hipMemcpyDeviceToHost);
        std::map<uint32_t, int> simd, cu;
        for (uint32_t i = 0; i < n; ++i) {
            const uint32_t hw = h[2 * i], x = h[2 * i + 1] & 15u;
            const uint32_t s = (hw >> 4) & 3u, c = (hw >> 8) & 15u, sh = (hw >> 12) & 1u, se = (hw >> 13) & 7u;
            const uint32_t cuid = x << 12 | se << 8 | sh << 4 | c;
            cu[cuid]++; simd[cuid << 2 | s]++;
        }
        std::map<int, int> hs, hc;
        for (auto& kv : simd) hs[kv.second]++;
        for (auto& kv : cu) hc[kv.second]++;
        printf("n=%u lds=%u: %zu CUs used, %zu SIMDs used | waves/SIMD histogram:", n, lds_bytes, cu.size(), simd.size());
        for (auto& kv : hs) printf(" %d:%d", kv.first, kv.second);
        printf(" | waves/CU:");
        for (auto& kv : hc) printf(" %d:%d", kv.first, kv.second);
        printf("\n");
        if (n == 5120u) {
            int same = 0, same_cu = 0;
            for (uint32_t i = 0; i < 1024; ++i) for (uint32_t k = 1; k < 5; ++k) {
                auto id = [&](uint32_t j) { const uint32_t hw = h[2 * j]; return (h[2 * j + 1] & 15u) << 16 | ((hw >> 4) & 0xFFFu); };
                same += id(i) == id(i + 1024 * k); same_cu += (id(i) >> 2) == (id(i + 1024 * k) >> 2);
            }
            printf("  workgroups i and i + 1024 k: same SIMD %d of 4096, same CU %d of 4096\n", same, same_cu);
            for (uint32_t stride : {8u, 32u, 64u, 128u, 256u, 512u, 2048u}) { int sm = 0; for (uint32_t i = 0; i + stride < 5120; ++i) { auto id = [&](uint32_t j) { const uint32_t hw = h[2 * j]; return (h[2 * j + 1] & 15u) << 16 | ((hw >> 4) & 0xFFFu); }; sm += id(i) == id(i + stride); } printf("  stride %u: same SIMD %d of %u\n", stride, sm, 5120 - stride); }
        }
        if (n == 1016u) { printf("  first 24 blocks (xcc,se,sh,cu,simd):"); for (uint32_t i = 0; i < 24; ++i) { uint32_t hw = h[2*i]; printf(" %u.%u.%u.%u.%u", h[2*i+1]&15u, (hw>>13)&7u, (hw>>12)&1u, (hw>>8)&15u, (hw>>4)&3u); } printf("\n"); }
    }
    return 0;
}
