// Does a wave64 VALU instruction cost less when only 16 (or 32) of its lanes are active?  (dev tool, gfx950.)  The narrow Merkle levels are
// bound by the dependent chain of a BLAKE2b compression in a lone wave (profiles/r06/blake2b_quad_ubench.txt: 8.7 cycles per dependent
// instruction); if the hardware skipped the 16-lane passes whose lanes are all inactive, hashes spread over more waves with 16 active lanes
// each would run a shorter chain.  One wave per workgroup, a long dependent chain of the instruction kinds of a compression, EXEC = the
// low `active` lanes; prints cycles per dependent instruction.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/exec_skip tools/microbench/exec_skip_ubench.hip && /tmp/exec_skip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int KIND>
__global__ void __launch_bounds__(64) chain_kernel(uint64_t* out, int iters, int active, unsigned long long* cycles) {
    uint64_t a = threadIdx.x * 0x9E3779B97F4A7C15ull + 1, b = a ^ 0x1234567ull, c = a + 77, d = b + 99;
    const bool on = (int)threadIdx.x < active;
    unsigned long long t0 = 0, t1 = 0;
    if (on) {
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (KIND == 0) {            // the G function's own chain: add, add, xor, rotate, add, xor, rotate
                    a = a + b; a = a + (uint64_t)k; d ^= a; d = (d >> 32) | (d << 32); c = c + d; b ^= c; b = (b >> 24) | (b << 40);
                } else if (KIND == 1) {     // 64-bit adds only
                    a = a + b; b = b + a; a = a + b; b = b + a; a = a + b; b = b + a; a = a + b;
                } else {                    // 32-bit xors only (two independent halves)
                    a ^= b; b ^= a + 0; a ^= b; b ^= a; a ^= b; b ^= a; a ^= b;
                    asm volatile("" : "+v"(a), "+v"(b));
                }
            }
        }
        t1 = __builtin_amdgcn_s_memtime();
        out[blockIdx.x * 64 + threadIdx.x] = a ^ b ^ c ^ d;
        if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    }
}

int main() {
    uint64_t* out; unsigned long long* cyc;
    hipMalloc(&out, 64 * 64 * 8); hipMalloc(&cyc, 64 * 8);
    const int iters = 2000;
    const char* names[3] = {"G chain (add add xor rot32 add xor rot24)", "64-bit adds", "xors"};
    const int per_iter[3] = {16 * 7, 16 * 7, 16 * 7};
    for (int kind = 0; kind < 3; ++kind)
        for (int blocks : {1, 4})
            for (int active : {64, 32, 16, 4}) {
                for (int rep = 0; rep < 2; ++rep) {
                    if (kind == 0) hipLaunchKernelGGL(chain_kernel<0>, dim3(blocks), dim3(64), 0, 0, out, iters, active, cyc);
                    if (kind == 1) hipLaunchKernelGGL(chain_kernel<1>, dim3(blocks), dim3(64), 0, 0, out, iters, active, cyc);
                    if (kind == 2) hipLaunchKernelGGL(chain_kernel<2>, dim3(blocks), dim3(64), 0, 0, out, iters, active, cyc);
                    hipDeviceSynchronize();
                }
                unsigned long long h = 0; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
                printf("%-44s waves %d active lanes %2d: %6.2f cycles per source-level step (%llu cycles)\n", names[kind], blocks, active, (double)h / ((double)iters * per_iter[kind]), h);
            }
    return 0;
}
