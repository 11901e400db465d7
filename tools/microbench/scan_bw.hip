// How fast can one MI355X READ a vector without doing anything with it?  (dev tool: the zero scan of vec_degree_kernel, csrc/core.hip,
// ran at 1.3 TB/s with one element per thread and 2.2 TB/s with a strided grid and four loads in flight.)  A read-only scan of 2^24
// 16-byte elements: loads per thread and step, grid size, plain / non-temporal loads.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/microbench/scan_bw tools/microbench/scan_bw.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
struct Fe { uint64_t lo, hi; };
template <int U, bool NT>
__global__ void __launch_bounds__(256) scan_kernel(const Fe* __restrict__ v, uint64_t n, unsigned long long* out) {
    const uint64_t step = (uint64_t)gridDim.x * (256 * U);
    unsigned long long acc = 0;
    for (uint64_t base = (uint64_t)blockIdx.x * (256 * U) + threadIdx.x; base - threadIdx.x < n; base += step) {
        Fe x[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const uint64_t i = base + 256u * k;
            if (i < n) {
                if (NT) { x[k].lo = __builtin_nontemporal_load(&v[i].lo); x[k].hi = __builtin_nontemporal_load(&v[i].hi); }
                else x[k] = v[i];
            } else x[k] = Fe{0, 0};
        }
#pragma unroll
        for (int k = 0; k < U; ++k) acc |= x[k].lo | x[k].hi;
    }
    if (__ballot(acc != 0) && (threadIdx.x & 63) == 0) atomicOr(out, 1ull);
}
template <int U, bool NT>
static void run(const Fe* d, uint64_t n, unsigned long long* d_out, unsigned blocks) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((scan_kernel<U, NT>), dim3(blocks), dim3(256), 0, 0, d, n, d_out);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("  %2d loads per thread, %5u workgroups, %s: %7.1f us  %5.2f TB/s\n", U, blocks, NT ? "non-temporal" : "plain       ", best * 1e3, n * 16.0 / (best * 1e-3) / 1e12);
}
int main() {
    const uint64_t n = 1ull << 24;
    Fe* d; unsigned long long* d_out;
    if (hipMalloc(&d, n * 16) != hipSuccess || hipMalloc(&d_out, 8) != hipSuccess) { printf("no memory\n"); return 1; }
    hipMemset(d, 0, n * 16); hipMemset(d_out, 0, 8);
    printf("scan_bw: read-only scan of 2^24 16-byte elements (268 MB), best of 6\n");
    for (unsigned blocks : {1024u, 2048u, 4096u, 8192u, 16384u}) {
        run<4, false>(d, n, d_out, blocks);
        run<8, false>(d, n, d_out, blocks);
    }
    run<16, false>(d, n, d_out, 1024); run<16, false>(d, n, d_out, 2048);
    run<4, true>(d, n, d_out, 2048); run<8, true>(d, n, d_out, 2048); run<8, true>(d, n, d_out, 4096);
    run<1, false>(d, n, d_out, 65536); run<2, false>(d, n, d_out, 32768);
    return 0;
}
