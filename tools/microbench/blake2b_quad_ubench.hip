// blake2b_quad_ubench.hip -- what bounds a narrow Merkle level: four lanes per BLAKE2b compression in a LONE wave (dev tool, round 6).
// One workgroup of 256 threads per CU-sized grid slot, i.e. one wave per SIMD, each wave running REPS compressions back to back
// (the digest of one feeds a word of the next message, like the levels of a tree), timed with s_memtime (shader cycles).
//   library  : csrc/merkle.cuh blake2b_node_4lane -- the message word index derived from the packed sigma constants at every read
//              (here the message does not move, so the compiler hoists that arithmetic out of the loop; in the tree kernels it
//              is paid per level: 117 VALU + 40 SALU of ~890 issue slots)
//   quad     : the experiment of round 6, kept in this file only -- 40 per-lane LDS addresses computed once per kernel, the words
//              of a round requested one round ahead (pinned with sched_barrier), a + b + x computed as (a + x) + b, and the quad
//              rotating a, c, d instead of b, c, d between the column and the diagonal step (b is the last value a G produces:
//              no DPP move then stands between one G and the next).  Bit-identical; see profiles/r06/blake2b_quad_ubench.txt
//              for what it did and did not buy.
// and the dependent-issue cost of the instruction kinds a compression is made of (a chain of N dependent instructions of one kind
// in a lone wave, and the same with two independent chains interleaved):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I stark-anatomy_amd/csrc -o tools/microbench/blake2b_quad_ubench tools/microbench/blake2b_quad_ubench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "merkle.cuh"

using namespace sc;

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// ---- the experiment (not in the library)
// Where lane j of the quad that hashes node n finds its message words: the LDS byte offsets, from the start of a lin array, of the
// 40 words it consumes (rounds 0..9 -- 10 and 11 repeat 0 and 1 -- each: column step x, y, diagonal step x, y).  A lane hashes the
// same node number on every level of a climb and the sigma schedule does not depend on the data, so these are computed ONCE per
// kernel: a read is then `ds_read_b64 v, a[k]` and nothing else (the library's B2_ROUND4 derives the word index from the packed sigma
// constants for every read: 117 VALU + 40 SALU of the ~890 issue slots of a compression).
struct QuadWords { uint32_t a[40]; };
__device__ __forceinline__ void quad_words(uint32_t n, uint32_t j, QuadWords& W) {
    constexpr uint32_t COL[10] = {0x76543210u, 0x6df984aeu, 0xdf250c8bu, 0xebcd1397u, 0xfa427509u, 0x38b0a6c2u, 0xa4def15cu, 0x931ce7bdu, 0x803b9ef6u, 0x5167482au};
    constexpr uint32_t DIA[10] = {0xfedcba98u, 0x357b20c1u, 0x491763eau, 0x8f04a562u, 0xd386cb1eu, 0x91ef57d4u, 0xb8293670u, 0xa2684f05u, 0x5a417d2cu, 0x0dc3e9bfu};
    const uint32_t sh = 8u * j, shd = 8u * ((j + 3u) & 3u), base = 17u * 8u * n;     // (lane j runs diagonal j - 1: see blake2b_quad)
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t bc = (COL[r] >> sh) & 0xFFu, bd = (DIA[r] >> shd) & 0xFFu;
        W.a[4 * r + 0] = base + 8u * (bc & 15u);
        W.a[4 * r + 1] = base + 8u * (bc >> 4);
        W.a[4 * r + 2] = base + 8u * (bd & 15u);
        W.a[4 * r + 3] = base + 8u * (bd >> 4);
    }
}

// Single-block BLAKE2b-512 by the four lanes of a quad, the message found through QuadWords (`lin`: the lin array).  Lane j returns
// digest words j (h_lo) and 4 + j (h_hi).  The chain is what is cut here: a + b + x is computed as (a + x) + b (a is final five
// steps before b), and between the column and the diagonal step the quad rotates a, c and d -- b, the LAST value a G produces,
// stays where it is (lane L then runs diagonal L - 1; quad_words accounts for it): 18 dependent instructions per G instead of 22.
// The four words of a round are requested one round ahead and the request is pinned where it stands (sched_barrier).
#ifndef B2_OPAQUE_ASM
#define B2_OPAQUE_ASM 0
#endif
#if B2_OPAQUE_ASM
#define B2_OPAQUE(v) asm("" : "+v"(v))      // keeps (a + x) apart from + b, at the price of register-pair copies
#else
#define B2_OPAQUE(v) ((void)0)              // hipcc then sinks (a + x) back in front of + b
#endif
template <uint32_t SWEEP = 0>
__device__ __forceinline__ void blake2b_quad(const uint64_t* lin, const QuadWords& W, uint32_t len, uint32_t j, uint64_t& h_lo, uint64_t& h_hi) {
    const uint64_t iv_a = B2_IV[j], iv_b = B2_IV[4 + j];
    const uint64_t h0 = (j == 0) ? (iv_a ^ 0x01010040ull) : iv_a;
    uint64_t a = h0, b = iv_b, c = iv_a, d = iv_b;
    if (j == 0) d ^= (uint64_t)len;   // t0 = message length
    if (j == 2) d = ~d;               // final block
    const char* base = reinterpret_cast<const char*>(lin) + SWEEP;
#define B2_W(k) (*reinterpret_cast<const uint64_t*>(base + W.a[k]))
    // one G on (a, b, c, d) with `an` = a + x already formed; leaves `an` = (a moved by PA) + xnext, c and d moved by PC / PD
#define B2_GQ(y, PA, PC, PD, xnext)                                \
    do {                                                           \
        a = an + b;                                                \
        d = rotr64(d ^ a, 32);                                     \
        c = c + d;                                                 \
        b = rotr64(b ^ c, 24);                                     \
        uint64_t ay = a + (y);                                     \
        B2_OPAQUE(ay);                                             \
        a = ay + b;                                                \
        d = rotr64(d ^ a, 16);                                     \
        an = quad_perm64<PA>(a) + (xnext);                         \
        B2_OPAQUE(an);                                             \
        c = c + d;                                                 \
        b = rotr64(b ^ c, 63);                                     \
        d = quad_perm64<PD>(d);                                    \
        c = quad_perm64<PC>(c);                                    \
    } while (0)
    uint64_t x0 = B2_W(0), x1 = B2_W(1), x2 = B2_W(2), x3 = B2_W(3);
    uint64_t an = a + x0;
    // column step in lane j = column j; then a comes from lane j - 1, c from j + 1, d from j + 2 (frame of the diagonal step: lane L
    // holds a[L-1], b[L], c[L+1], d[L+2] = diagonal L - 1); after it a from lane j + 1, c from j - 1, d from j + 2 (columns again)
#define B2_ROUNDQ(next, last)                                                        \
    do {                                                                             \
        const uint64_t y0 = B2_W(4 * (next) + 0), y1 = B2_W(4 * (next) + 1), y2 = B2_W(4 * (next) + 2), y3 = B2_W(4 * (next) + 3); \
        __builtin_amdgcn_sched_barrier(0);                                           \
        B2_GQ(x1, 0x93, 0x39, 0x4E, x2);                                             \
        B2_GQ(x3, 0x39, 0x93, 0x4E, (last) ? 0ull : y0);                             \
        x1 = y1; x2 = y2; x3 = y3;                                                   \
    } while (0)
    B2_ROUNDQ(1, false); B2_ROUNDQ(2, false); B2_ROUNDQ(3, false); B2_ROUNDQ(4, false); B2_ROUNDQ(5, false); B2_ROUNDQ(6, false);
    B2_ROUNDQ(7, false); B2_ROUNDQ(8, false); B2_ROUNDQ(9, false); B2_ROUNDQ(0, false); B2_ROUNDQ(1, false); B2_ROUNDQ(1, true);     // (the last prefetch is not used)
#undef B2_ROUNDQ
#undef B2_GQ
#undef B2_W
    h_lo = h0 ^ an ^ c;               // (after the last G `an` is a itself, back in its column)
    h_hi = iv_b ^ b ^ d;
}


constexpr int REPS = 64;

template <int VARIANT>
__global__ void __launch_bounds__(256) hash_kernel(const uint64_t* in, uint64_t* out, unsigned long long* ticks) {
    __shared__ uint64_t lin[64 * 17];
    const uint32_t t = threadIdx.x, n = t >> 2, j = t & 3u;
    for (uint32_t i = t; i < 64 * 17; i += 256) lin[i] = in[i];
    __syncthreads();
    QuadWords W;
    quad_words(n, j, W);
    uint64_t lo = 0, hi = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < REPS; ++r) {
        if (VARIANT == 0) blake2b_node_4lane(lin + 17u * n, j, lo, hi);
        else blake2b_quad<0>(lin, W, 128u, j, lo, hi);
        lin[17u * n + j] = lo;                        // the digest goes back into the message (own quad's slot only)
        lin[17u * n + 4u + j] = hi;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[(size_t)blockIdx.x * 512 + 2 * t] = lo; out[(size_t)blockIdx.x * 512 + 2 * t + 1] = hi;
    if ((t & 63u) == 0) ticks[blockIdx.x * 4 + (t >> 6)] = t1 - t0;
}

// dependent chains: KIND 0 v_xor_b32, 1 v_alignbit_b32, 2 v_lshl_add_u64, 3 v_mov_b32_dpp quad_perm, 4 v_xor_b32_dpp; CHAINS independent chains interleaved
constexpr int CHAIN = 1024;
template <int KIND, int CHAINS>
__global__ void __launch_bounds__(256) chain_kernel(uint64_t* out, unsigned long long* ticks) {
    uint32_t a[4], b[4];
    uint64_t q[4], r[4];
    for (int i = 0; i < 4; ++i) { a[i] = threadIdx.x * 2654435761u + i; b[i] = threadIdx.x * 40503u + 7 * i + 1; q[i] = ((uint64_t)a[i] << 32) | b[i]; r[i] = ((uint64_t)b[i] << 32) | a[i]; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < CHAIN / 8; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                if constexpr (KIND == 0) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[c]) : "v"(b[c]));
                else if constexpr (KIND == 1) asm volatile("v_alignbit_b32 %0, %0, %1, 24" : "+v"(a[c]) : "v"(b[c]));
                else if constexpr (KIND == 2) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q[c]) : "v"(r[c]));
                else if constexpr (KIND == 3) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf" : "+v"(a[c]));
                else asm volatile("v_xor_b32_dpp %0, %1, %0 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf" : "+v"(a[c]) : "v"(b[c]));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    uint64_t acc = 0;
    for (int i = 0; i < 4; ++i) acc ^= a[i] ^ q[i];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
    if ((threadIdx.x & 63u) == 0) ticks[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

static double mean_ticks(const std::vector<unsigned long long>& v) { double s = 0; for (auto x : v) s += (double)x; return s / v.size(); }

int main() {
    const int blocks = 256;
    uint64_t *d_in, *d_out; unsigned long long* d_ticks;
    std::vector<uint64_t> h(64 * 17);
    uint64_t s = 88172645463325252ull;
    for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = s; }
    CHK(hipMalloc(&d_in, h.size() * 8)); CHK(hipMalloc(&d_out, (size_t)blocks * 512 * 8)); CHK(hipMalloc(&d_ticks, blocks * 4 * 8));
    CHK(hipMemcpy(d_in, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    std::vector<unsigned long long> tk(blocks * 4);
    std::vector<uint64_t> o0((size_t)blocks * 512), o1((size_t)blocks * 512);
    auto run_hash = [&](int variant, std::vector<uint64_t>& o) -> double {
        for (int rep = 0; rep < 3; ++rep) {
            if (variant == 0) hipLaunchKernelGGL(hash_kernel<0>, dim3(blocks), dim3(256), 0, 0, d_in, d_out, d_ticks);
            else hipLaunchKernelGGL(hash_kernel<1>, dim3(blocks), dim3(256), 0, 0, d_in, d_out, d_ticks);
            hipDeviceSynchronize();
        }
        hipMemcpy(tk.data(), d_ticks, tk.size() * 8, hipMemcpyDeviceToHost);
        hipMemcpy(o.data(), d_out, o.size() * 8, hipMemcpyDeviceToHost);
        return mean_ticks(tk) / REPS;
    };
    const double c0 = run_hash(0, o0), c1 = run_hash(1, o1);
    size_t bad = 0;
    for (size_t i = 0; i < o0.size(); ++i) bad += o0[i] != o1[i];
    printf("four lanes per compression, one wave per SIMD, %d compressions back to back (s_memtime = shader cycles):\n", REPS);
    printf("  library (csrc/merkle.cuh blake2b_node_4lane)        : %7.1f cycles per compression\n", c0);
    printf("  quad    (addresses once, prefetch, chain cut; opaque asm %d) : %7.1f cycles per compression     results identical: %s\n", B2_OPAQUE_ASM, c1, bad ? "NO" : "yes");
    const char* names[5] = {"v_xor_b32", "v_alignbit_b32", "v_lshl_add_u64", "v_mov_b32_dpp", "v_xor_b32_dpp"};
    printf("dependent chains in a lone wave, cycles per instruction (one chain | two independent chains interleaved, per instruction):\n");
    auto run_chain = [&](auto kernel) -> double {
        for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d_out, d_ticks); hipDeviceSynchronize(); }
        hipMemcpy(tk.data(), d_ticks, tk.size() * 8, hipMemcpyDeviceToHost);
        return mean_ticks(tk) / CHAIN;
    };
    const double k0 = run_chain(chain_kernel<0, 1>), k0b = run_chain(chain_kernel<0, 2>) / 2;
    const double k1 = run_chain(chain_kernel<1, 1>), k1b = run_chain(chain_kernel<1, 2>) / 2;
    const double k2 = run_chain(chain_kernel<2, 1>), k2b = run_chain(chain_kernel<2, 2>) / 2;
    const double k3 = run_chain(chain_kernel<3, 1>), k3b = run_chain(chain_kernel<3, 2>) / 2;
    const double k4 = run_chain(chain_kernel<4, 1>), k4b = run_chain(chain_kernel<4, 2>) / 2;
    const double one[5] = {k0, k1, k2, k3, k4}, two[5] = {k0b, k1b, k2b, k3b, k4b};
    for (int i = 0; i < 5; ++i) printf("  %-16s %6.2f | %6.2f\n", names[i], one[i], two[i]);
    return bad ? 2 : 0;
}
