// fp64_modmul.hip -- integer Montgomery product (csrc/field_asm.cuh: v_mad_u64_u32 + carry chains) against the FP64-FMA
// formulation of fp64_modmul.h, in the harness of exchange_ubench.hip: the occupancy of the pass kernels (6 workgroups of 4 waves
// per CU), every thread holds E = 4 elements and runs ROUNDS of two radix-2 DIF stages on them in registers (4 butterflies per
// round: s = u + v, d = (u - v) * w), which is the arithmetic of one round of ntt_pass_kernel_fixed without its LDS exchange.
//
//   int        : fe_addsub2 + mont_mul2 (the library's hand-interleaved pairs), data canonical 2 x u64 all the way
//   fp64       : data converted to three weighted double limbs at kernel entry and back at exit (= pass load / store), lazy add /
//                sub (3 v_add_f64 each), fq::modmul under round-toward-zero; limbs are re-normalised where the growth bound of
//                the hi/lo split needs it (second-stage products and the pure-sum output of a round)
//   convert    : the fp64 kernel with zero rounds -- the price of the two conversions alone
//
// Every fp64 result is compared, element for element, with the integer kernel's (both canonical residues), and on the host the
// formulation is checked against the portable mont_mul_c on 2 * 10^5 random + edge operands (lazy and canonical inputs).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I stark-anatomy_amd/csrc -o tools/microbench/fp64_modmul tools/microbench/fp64_modmul.hip
//   ./fp64_modmul            (GPU: timing + element-wise comparison)      ./fp64_modmul --host   (CPU check only, no device needed)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cfenv>
#include <vector>
#include <type_traits>
#include "field.cuh"
#include "fp64_modmul.h"

using namespace sc;
typedef unsigned __int128 u128;

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

#ifndef FQ_ROUNDS
#define FQ_ROUNDS 64
#endif

__global__ void __launch_bounds__(256) int_kernel(const Fe* __restrict__ in, const Fe* __restrict__ tw, Fe* __restrict__ out, int rounds) {
    const uint32_t t = threadIdx.x;
    Fe x[4], w[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { x[r] = in[((size_t)blockIdx.x * 256 + t) * 4 + r]; w[r] = tw[(t * 4 + r) & 1023]; }
    for (int it = 0; it < rounds; ++it) {
        Fe s0, d0, s1, d1;
        fe_addsub2(x[0], x[2], x[1], x[3], s0, d0, s1, d1);
        mont_mul2(d0, w[0], d1, w[1], d0, d1);
        Fe a0, b0, a1, b1;
        fe_addsub2(s0, s1, d0, d1, a0, b0, a1, b1);
        mont_mul2(b0, w[2], b1, w[3], b0, b1);
        x[0] = a0; x[1] = b0; x[2] = a1; x[3] = b1;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) out[((size_t)blockIdx.x * 256 + t) * 4 + r] = x[r];
}

// carry l0 -> l1 -> l2 (two floor splits); FOLD: also take what the top limb holds above 407 * 2^119 back down with
// 407 * 2^119 = -1 (mod p) -- a pure-sum output doubles per stage and nothing else ever reduces it
template <bool FOLD>
__host__ __device__ __forceinline__ fq::F3 normalise(fq::F3 v) {
    constexpr double MA = 1.5 * fq::p2(52 + 43), MB = 1.5 * fq::p2(52 + 86), M0 = 1.5 * fq::p2(52), INV = 1.0 / (407.0 * fq::p2(119));
    if (FOLD) {
        const double q = (v.l2 * INV + M0) - M0;          // floor(l2 / (p - 1)), give or take one
        v.l2 = fq::ffma(q, fq::C407N, v.l2);
        v.l0 -= q;
    }
    const double ca = (v.l0 + MA) - MA;
    v.l0 -= ca; v.l1 += ca;
    const double cb = (v.l1 + MB) - MB;
    v.l1 -= cb; v.l2 += cb;
    return v;
}

// tw3: the twiddles as scaled Montgomery limbs (3 doubles each)
__global__ void __launch_bounds__(256) fp_kernel(const Fe* __restrict__ in, const fq::F3* __restrict__ tw3, Fe* __restrict__ out, int rounds) {
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3" ::: "memory");     // f64 rounding: toward zero
    const uint32_t t = threadIdx.x;
    fq::F3 x[4], w[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const Fe v = in[((size_t)blockIdx.x * 256 + t) * 4 + r];
        x[r] = fq::from_u128<0>(v.lo, v.hi);
        w[r] = tw3[(t * 4 + r) & 1023];
    }
    // (a pass folds a pure-sum output every few stages; here in every second round = every fourth stage.  Two rounds per loop
    // iteration so that the choice is made at compile time: a per-lane select would be a VOP2 v_cndmask, 23 cycles on gfx950)
    auto round = [&](auto fold) {
        const fq::F3 s0 = fq::add(x[0], x[2]), s1 = fq::add(x[1], x[3]);
        const fq::F3 d0 = fq::modmul<false>(fq::sub(x[0], x[2]), w[0]), d1 = fq::modmul<false>(fq::sub(x[1], x[3]), w[1]);
        x[0] = normalise<decltype(fold)::value>(fq::add(s0, s1));
        x[1] = fq::modmul<true>(fq::sub(s0, s1), w[2]);
        x[2] = normalise<false>(fq::add(d0, d1));
        x[3] = fq::modmul<true>(fq::sub(d0, d1), w[3]);
    };
    int it = 0;
    for (; it + 1 < rounds; it += 2) { round(std::false_type{}); round(std::true_type{}); }
    if (it < rounds) round(std::false_type{});
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        Fe v;
        fq::to_u128(x[r], v.lo, v.hi);
        out[((size_t)blockIdx.x * 256 + t) * 4 + r] = v;
    }
}

template <typename K, typename TW>
static float run(K kernel, const Fe* d_in, const TW* d_tw, Fe* d_out, int blocks, int rounds) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d_in, d_tw, d_out, rounds);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 7; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d_in, d_tw, d_out, rounds);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

static const u128 P = ((u128)P_HI << 64) | 1u;
static u128 to128(Fe a) { return ((u128)a.hi << 64) | a.lo; }
static Fe toFe(u128 v) { return Fe{(uint64_t)v, (uint64_t)(v >> 64)}; }

static int host_check() {
    fesetround(FE_TOWARDZERO);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    auto rndp = [&]() { u128 v; do { v = ((u128)rnd() << 64) | rnd(); } while (v >= P); return v; };
    Fe R129 = fe_add(Fe{R_LO, R_HI}, Fe{R_LO, R_HI});                            // 2^129 mod p
    std::vector<u128> edges = {0, 1, 2, P - 1, P - 2, (u128)1 << 43, ((u128)1 << 43) - 1, ((u128)1 << 86) - 1, (u128)1 << 86, (u128)1 << 119,
                               ((u128)407 << 119) - 1, (u128)0xFFFFFFFFFFFFFFFFull, (u128)1 << 127};
    size_t bad = 0, n = 0;
    auto get = [&](fq::F3 r) { uint64_t lo, hi; fq::to_u128(r, lo, hi); return ((u128)hi << 64) | lo; };
    auto check = [&](u128 a, u128 w, u128 a2) {
        const Fe wm = fe_mul(toFe(w), R129);
        const fq::F3 A = fq::from_u128<0>((uint64_t)a, (uint64_t)(a >> 64)), A2 = fq::from_u128<0>((uint64_t)a2, (uint64_t)(a2 >> 64));
        const fq::F3 B = fq::from_u128<-129>(wm.lo, wm.hi);
        const u128 want = to128(fe_mul(toFe(a), toFe(w)));
        bad += get(fq::modmul<true>(A, B)) != want; ++n;
        bad += get(fq::modmul<false>(A, B)) != want; ++n;
        // lazy inputs: un-normalised difference and sum, then a second product on the un-normalised result
        const fq::F3 d = fq::sub(A, A2), su = fq::add(A, A2);
        const Fe dm = fe_sub(toFe(a), toFe(a2)), sm = fe_add(toFe(a), toFe(a2));
        const fq::F3 r = fq::modmul<false>(d, B);
        const Fe w1 = fe_mul(dm, toFe(w));
        bad += get(r) != to128(w1); ++n;
        bad += get(fq::modmul<true>(fq::sub(r, su), B)) != to128(fe_mul(fe_sub(w1, sm), toFe(w))); ++n;
        bad += get(su) != to128(sm); ++n;
        bad += get(d) != to128(dm); ++n;
        // a sum of 32 elements, folded and normalised
        fq::F3 acc = A; Fe accm = toFe(a);
        for (int k = 0; k < 5; ++k) { acc = fq::add(acc, acc); accm = fe_add(accm, accm); }
        const fq::F3 f = normalise<true>(acc);
        bad += get(f) != to128(accm) || !(f.l2 < 408.0 * fq::p2(119) && f.l2 > -fq::p2(119)); ++n;
    };
    for (u128 a : edges) for (u128 w : edges) for (u128 a2 : {edges[3], edges[1], edges[6]}) check(a % P, w % P, a2 % P);
    for (int i = 0; i < 200000; ++i) check(rndp(), rndp(), rndp());
    printf("fp64 modmul, host check against the portable Montgomery product: %zu comparisons (%zu edge triples + 200000 random, canonical and lazy inputs), %zu mismatches\n",
           n, edges.size() * edges.size() * 3, bad);
    fesetround(FE_TONEAREST);
    return bad != 0;
}

int main(int argc, char** argv) {
    if (host_check()) return 2;
    if (argc > 1 && !strcmp(argv[1], "--host")) return 0;
    const int blocks = 256 * 6 * 4;                      // 6 workgroups of 4 waves per CU: the occupancy of the pass kernels
    const size_t n = (size_t)blocks * 256 * 4;
    std::vector<Fe> h(n), tw(1024);
    std::vector<fq::F3> tw3(1024);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    auto rndp = [&]() { u128 v; do { v = ((u128)rnd() << 64) | rnd(); } while (v >= P); return v; };
    for (auto& v : h) v = toFe(rndp());
    const Fe R129 = fe_add(Fe{R_LO, R_HI}, Fe{R_LO, R_HI});
    fesetround(FE_TOWARDZERO);
    for (int i = 0; i < 1024; ++i) {
        const Fe w = toFe(rndp());
        tw[i] = fe_mul(w, Fe{R_LO, R_HI});                // w * 2^128 mod p: the integer kernels' Montgomery form
        const Fe w129 = fe_mul(w, R129);
        tw3[i] = fq::from_u128<-129>(w129.lo, w129.hi);
    }
    fesetround(FE_TONEAREST);
    Fe *d_in, *d_tw, *d_out;
    fq::F3* d_tw3;
    CHK(hipMalloc(&d_in, n * sizeof(Fe))); CHK(hipMalloc(&d_tw, 1024 * sizeof(Fe))); CHK(hipMalloc(&d_out, n * sizeof(Fe)));
    CHK(hipMalloc(&d_tw3, 1024 * sizeof(fq::F3)));
    CHK(hipMemcpy(d_in, h.data(), n * sizeof(Fe), hipMemcpyHostToDevice));
    CHK(hipMemcpy(d_tw, tw.data(), 1024 * sizeof(Fe), hipMemcpyHostToDevice));
    CHK(hipMemcpy(d_tw3, tw3.data(), 1024 * sizeof(fq::F3), hipMemcpyHostToDevice));
    const int R = FQ_ROUNDS;
    std::vector<Fe> ri(n), rf(n);
    // correctness first, at a few depths (growth bounds are per round, so depth matters)
    size_t bad = 0;
    for (int rounds : {1, 2, 5, R}) {
        hipLaunchKernelGGL(int_kernel, dim3(blocks), dim3(256), 0, 0, d_in, d_tw, d_out, rounds);
        CHK(hipMemcpy(ri.data(), d_out, n * sizeof(Fe), hipMemcpyDeviceToHost));
        hipLaunchKernelGGL(fp_kernel, dim3(blocks), dim3(256), 0, 0, d_in, d_tw3, d_out, rounds);
        CHK(hipMemcpy(rf.data(), d_out, n * sizeof(Fe), hipMemcpyDeviceToHost));
        size_t b = 0;
        for (size_t i = 0; i < n; ++i) b += (ri[i].lo != rf[i].lo) || (ri[i].hi != rf[i].hi);
        printf("  %3d rounds: %zu of %zu elements differ between the integer and the fp64 kernel\n", rounds, b, n);
        bad += b;
    }
    const float ti0 = run(int_kernel, d_in, d_tw, d_out, blocks, 0), ti = run(int_kernel, d_in, d_tw, d_out, blocks, R);
    const float tf0 = run(fp_kernel, d_in, d_tw3, d_out, blocks, 0), tf = run(fp_kernel, d_in, d_tw3, d_out, blocks, R);
    const double bf = (double)n * R;                       // butterflies: 4 per thread-round = 1 per element-round
    printf("fp64_modmul: %d workgroups x 256 threads, %d rounds of 4 butterflies per thread (%zu elements)\n", blocks, R, n);
    printf("  integer  (mont_mul2 + fe_addsub2)  : %8.3f ms   %7.3f ps per butterfly-lane   (load+store only: %.3f ms)\n", ti, (ti - ti0) * 1e9 / bf, ti0);
    printf("  fp64     (3 weighted double limbs) : %8.3f ms   %7.3f ps per butterfly-lane   (load+convert+store only: %.3f ms)\n", tf, (tf - tf0) * 1e9 / bf, tf0);
    printf("  fp64 / integer, arithmetic only    : %.3f      conversions, per element and pass: %.3f ps = %.2f butterflies' worth\n",
           (tf - tf0) / (ti - ti0), (tf0 - ti0) * 1e9 / (double)n, (tf0 - ti0) / ((ti - ti0) / R));
    return bad ? 2 : 0;
}
