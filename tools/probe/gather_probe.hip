// What do FETCH_SIZE / TCC_EA0_RDREQ* count for THIS path's access pattern?  MI355X_MICROARCH.md calibrates FETCH_SIZE for wide
// streaming reads only (128-B requests tallied at 64 B: "double it") and says "other access widths are uncalibrated: calibrate on a
// known byte count in your own access pattern".  The render kernels gather 64-B node records (4 x global_load_dwordx4 per lane, one
// line, issued back to back, one wait) and 48-B triangle records (3 x dwordx4) at data-dependent addresses.  This probe does that on a
// table far larger than the Infinity Cache, each lane a random record, so that (nearly) every request goes to HBM and the byte count
// is known.  Modes: <64|48> = the loads of a record issued ONE AFTER THE OTHER'S RETURN (a loop with a run-time count: load, wait,
// load) -- NOT the kernels' pattern, kept because it shows what the vector L1 does not do (keep a line across a wait); chain1 / chain2 =
// dependent chains as in a traversal, the four loads together / the first alone and three after it; coop64 = four lanes per record.
//     gather_probe <record_bytes 64|48|16> <table_MB> <gathers per lane>      prints the bytes requested per launch
// Run under `rocprofv3 --pmc FETCH_SIZE`, `--pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum ...` and compare.
// mode "stream": every lane reads 16 B of a contiguous range once (the guide's calibrated case), for the same counters.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

__global__ void __launch_bounds__(256) k_gather(const float4* __restrict__ table, unsigned long long n_records, unsigned quads, unsigned iters, float* sink) {
    unsigned long long s = (blockIdx.x * 256ull + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345ull;
    float acc = 0.0f;
    for (unsigned it = 0; it < iters; ++it) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        const unsigned long long rec = (s >> 20) % n_records;
        const float4* p = table + rec * quads;
        for (unsigned q = 0; q < quads; ++q) { const float4 v = p[q]; acc += v.x + v.y + v.z + v.w; }
    }
    if (acc == 123.456f) *sink = acc;
}
// the same 64-B records, fetched by FOUR lanes each: in round r the four lanes of a quad read the four 16-B quarters of the record
// lane (quad base + r) wants -- one instruction, one 64-B request per record (the lanes' addresses fall into one line and the
// texture addresser merges them), where a lane reading its own record alone issues four requests to the same line
__global__ void __launch_bounds__(256) k_gather_coop(const float4* __restrict__ table, unsigned long long n_records, unsigned iters, float* sink) {
    unsigned long long s = (blockIdx.x * 256ull + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345ull;
    const unsigned lane = threadIdx.x & 63u, q = lane & 3u;
    float acc = 0.0f;
    for (unsigned it = 0; it < iters; ++it) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        const unsigned rec = (unsigned)((s >> 20) % n_records);
        for (unsigned r = 0; r < 4; ++r) {
            const unsigned want = __shfl(rec, (lane & ~3u) | r, 64);
            const float4 v = table[(unsigned long long)want * 4u + q];
            acc += v.x + v.y + v.z + v.w;
        }
    }
    if (acc == 123.456f) *sink = acc;
}
// the traversal's situation: the NEXT record's address depends on the record just read (a chain of dependent gathers per lane), so
// the figure is latency under load, not throughput.  phases = 1: the four 16-B loads of a record issued together (what the render
// kernels did up to round 5: four requests to L2 per record, none merged); phases = 2: the first 16 B alone, the other three once
// it has arrived (they find the line in the vector L1: one request to L2 per record, one more L1 round trip per step)
template <int PHASES>
__global__ void __launch_bounds__(256) k_chain(const float4* __restrict__ table, unsigned long long n_records, unsigned iters, float* sink) {
    unsigned long long s = (blockIdx.x * 256ull + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345ull;
    float acc = 0.0f;
    unsigned carry = 0;
    for (unsigned it = 0; it < iters; ++it) {
        s = s * 6364136223846793005ull + 1442695040888963407ull + carry;
        const unsigned long long rec = (s >> 20) % n_records;
        const float4* p = table + rec * 4u;
        float4 a = p[0], b, c, d;
        if (PHASES == 2) {
            const unsigned long long bump = (__float_as_uint(a.x) == 0x7fc12345u) ? 4u : 0u;    // never (the table is zero): orders the loads
            p += bump;
        }
        b = p[1]; c = p[2]; d = p[3];
        acc += (a.x + a.y + a.z + a.w) + (b.x + b.y + b.z + b.w) + (c.x + c.y + c.z + c.w) + (d.x + d.y + d.z + d.w);
        carry = (__float_as_uint(d.w) + __float_as_uint(a.x)) & 1u;            // 0: the next address waits for this record
    }
    if (acc == 123.456f) *sink = acc;
}
__global__ void __launch_bounds__(256) k_stream(const float4* __restrict__ table, unsigned long long n_quads, float* sink) {
    float acc = 0.0f;
    for (unsigned long long i = blockIdx.x * 256ull + threadIdx.x; i < n_quads; i += (unsigned long long)gridDim.x * 256ull) { const float4 v = table[i]; acc += v.x + v.w; }
    if (acc == 123.456f) *sink = acc;
}

int main(int argc, char** argv) {
    const bool stream = argc > 1 && !strcmp(argv[1], "stream");
    const bool coop = argc > 1 && !strcmp(argv[1], "coop64");
    const int chain = argc > 1 && !strcmp(argv[1], "chain1") ? 1 : argc > 1 && !strcmp(argv[1], "chain2") ? 2 : 0;
    const unsigned rec_bytes = stream ? 16 : (coop || chain) ? 64 : (argc > 1 ? atoi(argv[1]) : 64);
    const size_t mb = argc > 2 ? atoll(argv[2]) : 2048;
    const unsigned iters = argc > 3 ? atoi(argv[3]) : 64;
    const size_t bytes = mb << 20;
    float4* d; float* sink;
    if (hipMalloc((void**)&d, bytes) != hipSuccess || hipMalloc((void**)&sink, 4) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    (void)hipMemset(d, 0, bytes); (void)hipDeviceSynchronize();
    const unsigned quads = rec_bytes / 16, grid = 256 * 16;
    for (int rep = 0; rep < 3; ++rep) {
        if (stream) hipLaunchKernelGGL(k_stream, dim3(grid), dim3(256), 0, 0, d, bytes / 16, sink);
        else if (chain == 1) hipLaunchKernelGGL(k_chain<1>, dim3(grid), dim3(256), 0, 0, d, bytes / 64, iters, sink);
        else if (chain == 2) hipLaunchKernelGGL(k_chain<2>, dim3(grid), dim3(256), 0, 0, d, bytes / 64, iters, sink);
        else if (coop) hipLaunchKernelGGL(k_gather_coop, dim3(grid), dim3(256), 0, 0, d, bytes / 64, iters, sink);
        else hipLaunchKernelGGL(k_gather, dim3(grid), dim3(256), 0, 0, d, bytes / rec_bytes, quads, iters, sink);
    }
    (void)hipDeviceSynchronize();
    const double req = stream ? (double)bytes : (double)grid * 256 * iters * rec_bytes;
    printf("{\"mode\": \"%s\", \"record_bytes\": %u, \"table_mb\": %zu, \"bytes_requested_per_launch\": %.0f, \"launches\": 3}\n", stream ? "stream" : chain == 1 ? "dependent chain, 4 loads together" : chain == 2 ? "dependent chain, 1 + 3 loads" : coop ? "gather, four lanes per record" : "gather", rec_bytes, mb, req);
    return 0;
}
