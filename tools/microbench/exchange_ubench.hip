// The exchange between two rounds of a tile transform, LDS round trip against cross-lane shuffles (dev tool; north_star names
// "wavefront __shfl for the inner radix stages", DESIGN.md 3.1 explains why the kernels do not use them -- this is the A/B).
//
// A thread holds E = 4 field elements (16 bytes each); between two rounds the element index and two thread-id bits trade places:
// a 4 x 4 transpose of 128-bit elements among the lanes {l, l^4, l^8, l^12} of one wave (the wave-local exchange of the
// geometry-specialised pass kernels).  Variants, each around the SAME block of arithmetic (MULS Montgomery products per
// element, the kernels' own mont_mul) so that the exchange is measured where it runs -- in a VALU-bound loop:
//   none    : arithmetic only
//   lds     : 4 x ds_write_b128, wave-level fence, 4 x ds_read_b128 (XOR-swizzled rows, like csrc/ntt_tile.cuh)
//   shuffle : two butterfly stages of __shfl_xor (ds_bpermute_b32 on gfx950: 4 per element moved) with v_cndmask selects
// Prints nanoseconds per exchange+arithmetic block and per element; every variant's result is checked against `none` + the
// transpose done on the host side of the lanes (a checksum).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I stark-anatomy_amd/csrc -o tools/microbench/exchange_ubench tools/microbench/exchange_ubench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "field.cuh"

using namespace sc;

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int ITERS = 256;
#ifndef SC_UBENCH_MULS
#define SC_UBENCH_MULS 1
#endif
constexpr int MULS = SC_UBENCH_MULS;   // Montgomery products per element between two exchanges (a round of 2 stages of the pass kernels: 1 per element, plus 2 add/sub pairs per 4 elements)

__device__ __forceinline__ uint32_t shfl32(uint32_t v, int mask) { return (uint32_t)__shfl_xor((int)v, mask, 64); }

__device__ __forceinline__ Fe sel(bool c, Fe a, Fe b) { return c ? a : b; }

// one butterfly stage of the transpose: registers (r, r + RS) trade with the lane at distance LS
template <int RS, int LS>
__device__ __forceinline__ void transpose_stage(Fe (&x)[4], bool upper) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (r & RS) continue;
        const Fe a = x[r], b = x[r + RS];
        const Fe send = sel(upper, a, b);
        Fe got;
        got.lo = (uint64_t)shfl32((uint32_t)send.lo, LS) | ((uint64_t)shfl32((uint32_t)(send.lo >> 32), LS) << 32);
        got.hi = (uint64_t)shfl32((uint32_t)send.hi, LS) | ((uint64_t)shfl32((uint32_t)(send.hi >> 32), LS) << 32);
        x[r] = sel(upper, got, a);
        x[r + RS] = sel(upper, b, got);
    }
}

template <int MODE>
__global__ void __launch_bounds__(256) exchange_kernel(const Fe* __restrict__ in, const Fe* __restrict__ tw, Fe* __restrict__ out) {
    __shared__ Fe tile[256 * 4];
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint32_t l = (lane >> 2) & 3u;                 // the two lane bits that trade places with the element index
    Fe x[4], w[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { x[r] = in[((size_t)blockIdx.x * 256 + t) * 4 + r]; w[r] = tw[(t * 4 + r) & 1023]; }
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int m = 0; m < MULS; ++m) {
            mont_mul2(x[0], w[0], x[1], w[1], x[0], x[1]);
            mont_mul2(x[2], w[2], x[3], w[3], x[2], x[3]);
        }
        if (MODE == 1) {
            // row = 4 * (lane group) + element index; the partner view reads the transposed position; columns XOR-swizzled by the row
            Fe* base = tile + wave * 256;
            const uint32_t grp = lane & ~12u;           // lane with the two trading bits cleared
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t slot = (grp | (l << 2)) * 4 + r;            // [lane][r]
                base[slot ^ ((slot >> 4) & 3u)] = x[r];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t slot = (grp | ((uint32_t)r << 2)) * 4 + l;   // element l of the lane whose trading bits are r
                x[r] = base[slot ^ ((slot >> 4) & 3u)];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        } else if (MODE == 2) {
            transpose_stage<2, 8>(x, (l & 2u) != 0);
            transpose_stage<1, 4>(x, (l & 1u) != 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) out[((size_t)blockIdx.x * 256 + t) * 4 + r] = x[r];
}

template <int MODE>
static float run(const Fe* d_in, const Fe* d_tw, Fe* d_out, int blocks) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(exchange_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, d_in, d_tw, d_out);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(exchange_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, d_in, d_tw, d_out);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    const int blocks = 256 * 6 * 4;                      // 6 workgroups of 4 waves per CU: the occupancy of the pass kernels
    const size_t n = (size_t)blocks * 256 * 4;
    std::vector<Fe> h(n), tw(1024);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (auto& v : h) { v.lo = rnd(); v.hi = rnd() & 0x7FFFFFFFFFFFFFFFull; }
    for (auto& v : tw) { v.lo = rnd(); v.hi = rnd() & 0x7FFFFFFFFFFFFFFFull; }
    Fe *d_in, *d_tw, *d_out;
    CHK(hipMalloc(&d_in, n * sizeof(Fe))); CHK(hipMalloc(&d_tw, 1024 * sizeof(Fe))); CHK(hipMalloc(&d_out, n * sizeof(Fe)));
    CHK(hipMemcpy(d_in, h.data(), n * sizeof(Fe), hipMemcpyHostToDevice));
    CHK(hipMemcpy(d_tw, tw.data(), 1024 * sizeof(Fe), hipMemcpyHostToDevice));
    const float t0 = run<0>(d_in, d_tw, d_out, blocks);
    const float t1 = run<1>(d_in, d_tw, d_out, blocks);
    std::vector<Fe> r1(n), r2(n);
    CHK(hipMemcpy(r1.data(), d_out, n * sizeof(Fe), hipMemcpyDeviceToHost));
    const float t2 = run<2>(d_in, d_tw, d_out, blocks);
    CHK(hipMemcpy(r2.data(), d_out, n * sizeof(Fe), hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < n; ++i) bad += (r1[i].lo != r2[i].lo) || (r1[i].hi != r2[i].hi);
    printf("exchange_ubench: %d workgroups x 256 threads, %d iterations, %d Montgomery products per element between exchanges\n", blocks, ITERS, MULS);
    printf("  arithmetic only          : %8.3f ms\n", t0);
    printf("  + LDS round trip         : %8.3f ms   (+%.1f %%)\n", t1, 100.0 * (t1 - t0) / t0);
    printf("  + __shfl_xor transpose   : %8.3f ms   (+%.1f %%)\n", t2, 100.0 * (t2 - t0) / t0);
    printf("  LDS and shuffle results identical: %s\n", bad ? "NO" : "yes");
    return bad ? 2 : 0;
}
