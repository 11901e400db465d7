// valu_rates.hip -- issue cost of the integer VALU instructions the BLAKE2b / Montgomery kernels are made of (dev tool, gfx950).
//   hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip && ./valu_rates
// Every kernel runs ITERS x 8 independent instructions per wave (8 accumulator chains), WAVES waves per SIMD on every SIMD;
// cycles per instruction per wave = shader-clock ticks of a wave / (ITERS * 8) * (1 / waves sharing the SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

constexpr int ITERS = 4096;

#define BODY8(INS)  INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)

template <int KIND>
__global__ void __launch_bounds__(256) rate_kernel(uint64_t* out, unsigned long long* ticks) {
    uint32_t a[8], b[8];
    uint64_t q[8], r[8];
    for (int i = 0; i < 8; ++i) {
        a[i] = threadIdx.x * 2654435761u + i;
        b[i] = threadIdx.x * 40503u + 7 * i + 1;
        q[i] = ((uint64_t)a[i] << 32) | b[i];
        r[i] = ((uint64_t)b[i] << 32) | a[i];
    }
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; ++it) {
        if constexpr (KIND == 0) {          // v_xor_b32 (VOP2)
#define INS(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            BODY8(INS)
#undef INS
        } else if constexpr (KIND == 1) {   // v_alignbit_b32
#define INS(i) asm volatile("v_alignbit_b32 %0, %0, %1, 24" : "+v"(a[i]) : "v"(b[i]));
            BODY8(INS)
#undef INS
        } else if constexpr (KIND == 2) {   // v_lshl_add_u64
#define INS(i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q[i]) : "v"(r[i]));
            BODY8(INS)
#undef INS
        } else if constexpr (KIND == 3) {   // v_add_co_u32 + v_addc_co_u32 (one 64-bit add = 2 instructions, counted as 2)
#define INS(i) asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(a[i]), "+v"(b[i]) : "v"(b[(i + 1) & 7]), "v"(a[(i + 3) & 7]) : "vcc");
            INS(0) INS(1) INS(2) INS(3)
#undef INS
        } else if constexpr (KIND == 4) {   // v_perm_b32
#define INS(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(0x02010007u));
            BODY8(INS)
#undef INS
        } else if constexpr (KIND == 5) {   // v_add3_u32
#define INS(i) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i]));
            BODY8(INS)
#undef INS
        } else if constexpr (KIND == 6) {   // v_mad_u64_u32
#define INS(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[i]) : "v"(a[i]), "v"(b[i]) : "vcc");
            BODY8(INS)
#undef INS
        } else if constexpr (KIND == 7) {   // v_xor_b32 with a DPP quad_perm operand
#define INS(i) asm volatile("v_xor_b32_dpp %0, %1, %0 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b[i]));
            BODY8(INS)
#undef INS
        } else if constexpr (KIND == 8) {   // v_add_u32 (VOP2, no carry)
#define INS(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            BODY8(INS)
#undef INS
        } else if constexpr (KIND == 9) {   // v_or3_b32 (there is no v_xor3_b32 on gfx950)
#define INS(i) asm volatile("v_or3_b32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i]));
            BODY8(INS)
#undef INS
        } else if constexpr (KIND == 10) {  // v_mov_b32 with DPP (lane rotation alone)
#define INS(i) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b[i]));
            BODY8(INS)
#undef INS
        } else if constexpr (KIND == 11) {  // v_lshlrev_b64
#define INS(i) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(q[i]));
            BODY8(INS)
#undef INS
        } else if constexpr (KIND == 12) {  // v_and_or_b32
#define INS(i) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i]));
            BODY8(INS)
#undef INS
        } else if constexpr (KIND == 14) {  // v_xor_b32 SDWA: word selects on both sources, one destination word written, the other preserved
#define INS(i) asm volatile("v_xor_b32_sdwa %0, %0, %1 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1" : "+v"(a[i]) : "v"(b[i]));
            BODY8(INS)
#undef INS
        } else if constexpr (KIND == 15) {  // v_xor_b32 SDWA writing a whole dword (dst_sel:DWORD), sources by word
#define INS(i) asm volatile("v_xor_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0" : "+v"(a[i]) : "v"(b[i]));
            BODY8(INS)
#undef INS
        } else if constexpr (KIND == 16) {  // v_cndmask_b32 e32 (mask in vcc)
#define INS(i) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]));
            BODY8(INS)
#undef INS
        } else if constexpr (KIND == 17) {  // v_cndmask_b32 e64 (mask in an SGPR pair)
#define INS(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(b[i]));
            BODY8(INS)
#undef INS
        } else if constexpr (KIND == 18) {  // v_xor_b32 e64 (VOP3 encoding of a VOP2 operation)
#define INS(i) asm volatile("v_xor_b32_e64 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            BODY8(INS)
#undef INS
        } else if constexpr (KIND == 13) {  // v_pk_add_u16 (packed math rate)
#define INS(i) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            BODY8(INS)
#undef INS
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    uint64_t acc = 0;
    for (int i = 0; i < 8; ++i) acc += a[i] + b[i] + q[i] + r[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) *ticks = t1 - t0;
}

template <int KIND>
void run(const char* name, int per_iter, int waves_per_simd, uint64_t* d_out, unsigned long long* d_ticks) {
    const int blocks = 256 * waves_per_simd;     // 256 CUs x 4 SIMDs x waves_per_simd waves, 4 waves per block
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, d_ticks);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, d_ticks);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long ticks = 0;
    hipMemcpy(&ticks, d_ticks, 8, hipMemcpyDeviceToHost);
    const double instrs = (double)ITERS * per_iter;
    printf("%-34s waves/SIMD %d  kernel %8.1f us  wave ticks %10llu  ticks/instr/wave %6.2f  (per-SIMD issue: %5.2f ticks/instr)  ns/instr/SIMD %6.3f\n", name, waves_per_simd,
           ms * 1e3, ticks, ticks / instrs, ticks / instrs / waves_per_simd, ms * 1e6 / instrs / waves_per_simd);
}

int main() {
    uint64_t* d_out;
    unsigned long long* d_ticks;
    hipMalloc(&d_out, 8ull * 256 * 16 * 256);
    hipMalloc(&d_ticks, 8);
    for (int w : {1, 4}) {
        run<0>("v_xor_b32", 8, w, d_out, d_ticks);
        run<8>("v_add_u32", 8, w, d_out, d_ticks);
        run<1>("v_alignbit_b32", 8, w, d_out, d_ticks);
        run<4>("v_perm_b32", 8, w, d_out, d_ticks);
        run<2>("v_lshl_add_u64", 8, w, d_out, d_ticks);
        run<3>("v_add_co+v_addc (x2)", 8, w, d_out, d_ticks);
        run<5>("v_add3_u32", 8, w, d_out, d_ticks);
        run<9>("v_or3_b32", 8, w, d_out, d_ticks);
        run<12>("v_and_or_b32", 8, w, d_out, d_ticks);
        run<6>("v_mad_u64_u32", 8, w, d_out, d_ticks);
        run<7>("v_xor_b32_dpp quad_perm", 8, w, d_out, d_ticks);
        run<10>("v_mov_b32_dpp quad_perm", 8, w, d_out, d_ticks);
        run<11>("v_lshlrev_b64", 8, w, d_out, d_ticks);
        run<13>("v_pk_add_u16", 8, w, d_out, d_ticks);
        run<16>("v_cndmask_b32_e32 (vcc)", 8, w, d_out, d_ticks);
        run<17>("v_cndmask_b32_e64 (sgpr mask)", 8, w, d_out, d_ticks);
        run<18>("v_xor_b32_e64", 8, w, d_out, d_ticks);
        run<14>("v_xor_b32_sdwa (word -> word, preserve)", 8, w, d_out, d_ticks);
        run<15>("v_xor_b32_sdwa (words -> dword)", 8, w, d_out, d_ticks);
    }
    return 0;
}
