// v_cndmask variants on gfx950: VOP2 (mask in VCC) vs VOP3 (mask in an SGPR pair).  Dev tool.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
constexpr int ITERS = 2000;
#define REP8(s) s s s s s s s s
__global__ void __launch_bounds__(256) k_e32(uint32_t* out, uint32_t seed) {
    uint32_t c0 = seed + threadIdx.x, c1 = c0 + 1, c2 = c0 + 2, c3 = c0 + 3, c4 = c0 + 4, c5 = c0 + 5, c6 = c0 + 6, c7 = c0 + 7, y = seed * 3 + 1;
    asm volatile("v_cmp_gt_u32 vcc, %0, %1" :: "v"(c0), "v"(y) : "vcc");
    for (int it = 0; it < ITERS; ++it) {
        asm volatile(REP8("v_cndmask_b32_e32 %0, %0, %8, vcc\n v_cndmask_b32_e32 %1, %1, %8, vcc\n v_cndmask_b32_e32 %2, %2, %8, vcc\n v_cndmask_b32_e32 %3, %3, %8, vcc\n")
                     : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(y) : "vcc");
    }
    uint32_t s = c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7;
    if (s == 0x12345678u) out[threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_e64(uint32_t* out, uint32_t seed) {
    uint32_t c0 = seed + threadIdx.x, c1 = c0 + 1, c2 = c0 + 2, c3 = c0 + 3, c4 = c0 + 4, c5 = c0 + 5, c6 = c0 + 6, c7 = c0 + 7, y = seed * 3 + 1;
    uint64_t m;
    asm volatile("v_cmp_gt_u32_e64 %0, %1, %2" : "=s"(m) : "v"(c0), "v"(y));
    for (int it = 0; it < ITERS; ++it) {
        asm volatile(REP8("v_cndmask_b32_e64 %0, %0, %8, %9\n v_cndmask_b32_e64 %1, %1, %8, %9\n v_cndmask_b32_e64 %2, %2, %8, %9\n v_cndmask_b32_e64 %3, %3, %8, %9\n")
                     : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(y), "s"(m));
    }
    uint32_t s = c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7;
    if (s == 0x12345678u) out[threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_and(uint32_t* out, uint32_t seed) {   // class-A reference
    uint32_t c0 = seed + threadIdx.x, c1 = c0 + 1, c2 = c0 + 2, c3 = c0 + 3, c4 = c0 + 4, c5 = c0 + 5, c6 = c0 + 6, c7 = c0 + 7, y = seed * 3 + 1;
    for (int it = 0; it < ITERS; ++it) {
        asm volatile(REP8("v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n")
                     : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(y));
    }
    uint32_t s = c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7;
    if (s == 0x12345678u) out[threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_addc64(uint32_t* out, uint32_t seed) {   // VOP3 addc with SGPR-pair carry (as field_asm.cuh uses)
    uint32_t c0 = seed + threadIdx.x, c1 = c0 + 1, c2 = c0 + 2, c3 = c0 + 3, y = seed * 3 + 1;
    uint64_t m = 0, m2;
    for (int it = 0; it < ITERS; ++it) {
        asm volatile(REP8("v_addc_co_u32_e64 %0, %4, %0, %5, %6\n v_addc_co_u32_e64 %1, %4, %1, %5, %6\n v_addc_co_u32_e64 %2, %4, %2, %5, %6\n v_addc_co_u32_e64 %3, %4, %3, %5, %6\n")
                     : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "=&s"(m2) : "v"(y), "s"(m));
    }
    uint32_t s = c0 ^ c1 ^ c2 ^ c3 ^ (uint32_t)m2;
    if (s == 0x12345678u) out[threadIdx.x] = s;
}
typedef void (*kfn)(uint32_t*, uint32_t);
int main() {
    uint32_t* d; hipMalloc(&d, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct { const char* n; kfn f; } ks[] = {{"v_and_b32 (VOP2)", k_and}, {"v_cndmask_b32_e32 (vcc)", k_e32}, {"v_cndmask_b32_e64 (sgpr pair)", k_e64}, {"v_addc_co_u32_e64 (sgpr pair)", k_addc64}};
    for (int wps : {2, 4, 8}) for (auto& k : ks) {
        k.f<<<256 * wps, 256>>>(d, 1); hipDeviceSynchronize();
        float best = 1e9;
        for (int r = 0; r < 3; ++r) { hipEventRecord(e0); k.f<<<256 * wps, 256>>>(d, r + 2); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
        printf("%d waves/SIMD  %-32s %7.2f nominal cycles per wave-instruction\n", wps, k.n, best * 1e-3 * 2.4e9 / (ITERS * 32.0 * wps));
    }
    return 0;
}
