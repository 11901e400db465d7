// VALU issue-rate microbenchmark for gfx950: cycles per wave-instruction for the integer / f64 ops the
// 128-bit modular multiplier can be built from.  Dev tool (not part of the product path).
//   hipcc --offload-arch=gfx950 -O3 -o valu_ubench valu_ubench.hip && ./valu_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 2000;
constexpr int UNROLL = 8;   // 8 independent chains x 4 repeats per iteration = 32 instr/iter

#define REP4(s) s s s s

// each body: 8 independent instructions (chains c0..c7), repeated 4x
#define BODY32(OP) \
  REP4(OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7))

#define KERNEL32(NAME, OPMACRO) \
__global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t seed) { \
    uint32_t c[8]; uint32_t x = seed + threadIdx.x, y = seed * 3 + 1; \
    for (int i = 0; i < 8; i++) c[i] = x + i; \
    for (int it = 0; it < ITERS; ++it) { BODY32(OPMACRO) } \
    uint32_t s = 0; for (int i = 0; i < 8; i++) s ^= c[i]; \
    if (s == 0x12345678u) out[threadIdx.x] = s; }

#define OP_ADD(i)   asm volatile("v_add_u32 %0, %0, %1" : "+v"(c[i]) : "v"(y));
#define OP_ADDCO(i) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(c[i]) : "v"(y) : "vcc");
#define OP_ADDC(i)  asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(c[i]) : "v"(y) : "vcc");
#define OP_MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(c[i]) : "v"(y));
#define OP_MULHI(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(c[i]) : "v"(y));
#define OP_MUL24(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(c[i]) : "v"(y));
#define OP_MULHI24(i) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(c[i]) : "v"(y));
#define OP_MAD24(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(c[i]) : "v"(y));
#define OP_XOR(i)   asm volatile("v_xor_b32 %0, %0, %1" : "+v"(c[i]) : "v"(y));
#define OP_ALIGN(i) asm volatile("v_alignbit_b32 %0, %0, %1, 9" : "+v"(c[i]) : "v"(y));
#define OP_CND(i)   asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(c[i]) : "v"(y) : "vcc");
#define OP_FMA32(i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(c[i]) : "v"(y));
#define OP_ADD3(i)  asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(c[i]) : "v"(y));
#define OP_LSHLADD(i) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(c[i]) : "v"(y));
#define OP_MADU16(i) asm volatile("v_mad_u32_u16 %0, %0, %1, %0" : "+v"(c[i]) : "v"(y));
#define OP_DOT2(i) asm volatile("v_dot2_u32_u16 %0, %0, %1, %0" : "+v"(c[i]) : "v"(y));
#define OP_DOT4(i) asm volatile("v_dot4_u32_u8 %0, %0, %1, %0" : "+v"(c[i]) : "v"(y));
#define OP_MOV(i)   asm volatile("v_mov_b32 %0, %1" : "+v"(c[i]) : "v"(y));

KERNEL32(k_add, OP_ADD)
KERNEL32(k_addco, OP_ADDCO)
KERNEL32(k_addc, OP_ADDC)
KERNEL32(k_mullo, OP_MULLO)
KERNEL32(k_mulhi, OP_MULHI)
KERNEL32(k_mul24, OP_MUL24)
KERNEL32(k_mulhi24, OP_MULHI24)
KERNEL32(k_mad24, OP_MAD24)
KERNEL32(k_xor, OP_XOR)
KERNEL32(k_align, OP_ALIGN)
KERNEL32(k_cnd, OP_CND)
KERNEL32(k_fma32, OP_FMA32)
KERNEL32(k_add3, OP_ADD3)
KERNEL32(k_lshladd, OP_LSHLADD)
KERNEL32(k_madu16, OP_MADU16)
KERNEL32(k_dot2, OP_DOT2)
KERNEL32(k_dot4, OP_DOT4)
KERNEL32(k_mov, OP_MOV)

// 64-bit destination ops
#define KERNEL64(NAME, OPMACRO) \
__global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t seed) { \
    uint64_t c[8]; uint32_t x = seed + threadIdx.x, y = seed * 3 + 1; uint64_t y64 = ((uint64_t)y << 32) | x; double yd = (double)y; \
    for (int i = 0; i < 8; i++) c[i] = ((uint64_t)x << 20) + i; \
    for (int it = 0; it < ITERS; ++it) { BODY32(OPMACRO) } \
    uint64_t s = 0; for (int i = 0; i < 8; i++) s ^= c[i]; \
    if (s == 0x12345678u) out[threadIdx.x] = (uint32_t)s; (void)y64; (void)yd; }

#define OP_MAD64(i)  asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c[i]) : "v"(x), "v"(y) : "vcc");
#define OP_MAD64S(i) asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0" : "+v"(c[i]) : "v"(x), "v"(y) : "s10", "s11");
#define OP_LSHLADD64(i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(c[i]) : "v"(y64));
#define OP_FMA64(i)  asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(c[i]) : "v"(yd));
#define OP_MUL64F(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(c[i]) : "v"(yd));
#define OP_ADD64F(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(c[i]) : "v"(yd));
#define OP_LSHR64(i) asm volatile("v_lshrrev_b64 %0, 9, %0" : "+v"(c[i]));
#define OP_PKFMA(i)  asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(c[i]) : "v"(y64));
#define OP_PKADD(i)  asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(c[i]) : "v"(y64));

KERNEL64(k_mad64, OP_MAD64)
KERNEL64(k_mad64s, OP_MAD64S)
KERNEL64(k_lshladd64, OP_LSHLADD64)
KERNEL64(k_fma64, OP_FMA64)
KERNEL64(k_mul64f, OP_MUL64F)
KERNEL64(k_add64f, OP_ADD64F)
KERNEL64(k_lshr64, OP_LSHR64)
KERNEL64(k_pkfma, OP_PKFMA)

typedef void (*kfn)(uint32_t*, uint32_t);
struct Ent { const char* name; kfn fn; };

int main() {
    hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount; double clk_mhz = prop.clockRate / 1000.0;
    printf("device %s CUs %d clock %.0f MHz\n", prop.name, cus, clk_mhz);
    uint32_t* d; CHK(hipMalloc(&d, 4096));
    std::vector<Ent> ents = {
        {"v_add_u32", k_add}, {"v_add_co_u32", k_addco}, {"v_addc_co_u32", k_addc}, {"v_add3_u32", k_add3}, {"v_lshl_add_u32", k_lshladd},
        {"v_xor_b32", k_xor}, {"v_mov_b32", k_mov}, {"v_alignbit_b32", k_align}, {"v_cndmask_b32", k_cnd},
        {"v_mul_lo_u32", k_mullo}, {"v_mul_hi_u32", k_mulhi}, {"v_mul_u32_u24", k_mul24}, {"v_mul_hi_u32_u24", k_mulhi24}, {"v_mad_u32_u24", k_mad24},
        {"v_mad_u32_u16", k_madu16}, {"v_dot2_u32_u16", k_dot2}, {"v_dot4_u32_u8", k_dot4},
        {"v_mad_u64_u32(vcc)", k_mad64}, {"v_mad_u64_u32(sgpr)", k_mad64s}, {"v_lshl_add_u64", k_lshladd64}, {"v_lshrrev_b64", k_lshr64},
        {"v_fma_f32", k_fma32}, {"v_pk_fma_f32", k_pkfma}, {"v_fma_f64", k_fma64}, {"v_mul_f64", k_mul64f}, {"v_add_f64", k_add64f}};
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    // waves per SIMD sweep: blocks of 256 threads = 4 waves = 1 wave/SIMD per block; k blocks/CU
    for (int wps : {1, 2, 4}) {
        printf("--- %d wave(s) per SIMD (grid = %d blocks x 256 thr)\n", wps, cus * wps);
        for (auto& e : ents) {
            e.fn<<<cus * wps, 256>>>(d, 1); CHK(hipDeviceSynchronize());
            float best = 1e30f;
            for (int r = 0; r < 3; r++) {
                CHK(hipEventRecord(e0)); e.fn<<<cus * wps, 256>>>(d, r + 2); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
                float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            double instr_per_wave = (double)ITERS * 32;
            // each SIMD runs `wps` waves; cycles per wave-instruction (throughput) = time*clk / (instr_per_wave * wps)
            double cyc = best * 1e-3 * clk_mhz * 1e6 / (instr_per_wave * wps);
            printf("%-22s %8.3f ms  %6.2f cyc/wave-instr (at nominal %.0f MHz)\n", e.name, best, cyc, clk_mhz);
        }
    }
    return 0;
}
