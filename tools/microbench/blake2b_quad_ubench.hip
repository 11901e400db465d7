// blake2b_quad_ubench.hip -- what bounds a narrow Merkle level: four lanes per BLAKE2b compression in a LONE wave (dev tool, round 6).
// One workgroup of 256 threads per CU-sized grid slot, i.e. one wave per SIMD, each wave running REPS compressions back to back
// (the digest of one feeds a word of the next message, like the levels of a tree), timed with s_memtime (shader cycles).
//   before   : the function as the library had it until the last session of round 6 (kept here, macros B2O_*): b, c, d rotate between
//              the column and the diagonal step; hipcc computes a + b + x as (b + x) + a and adds a freshly rotated d to c in two adds
//   library  : csrc/merkle.cuh blake2b_node_4lane as it is now: b stays in its lane, the early add of the a-chain an asm statement,
//              rotated words put together as vectors
// and the dependent-issue cost of the instruction kinds a compression is made of (a chain of N dependent instructions of one kind
// in a lone wave, and the same with two independent chains interleaved).  profiles/r06/blake2b_quad_ubench.txt has the numbers of every
// intermediate form (the experiment file of the first session, fixed message slots, opaque asm; this session's asm adds, vector
// rotations, chain cut) -- `git log -- tools/microbench/blake2b_quad_ubench.hip` has their sources.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I stark-anatomy_amd/csrc -o tools/microbench/blake2b_quad_ubench tools/microbench/blake2b_quad_ubench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "merkle.cuh"

using namespace sc;

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// ---- the function before this round's last session (not in the library any more)
#define B2O_G4(x, y)                      \
    do {                                 \
        a = a + b + (x);                 \
        d = rotr64(d ^ a, 32);           \
        c = c + d;                       \
        b = rotr64(b ^ c, 24);           \
        a = a + b + (y);                 \
        d = rotr64(d ^ a, 16);           \
        c = c + d;                       \
        b = rotr64(b ^ c, 63);           \
    } while (0)

// The message words of a round -- two per G, per-lane addresses from the packed sigma constants -- are REQUESTED from LDS one round
// ahead and waited for at the top of their round.  Written as loads in C++ the compiler sinks them to their first use (register
// pressure), and every round then exposes an LDS round trip on the dependent chain of the compression -- which is all a narrow
// level's time is made of; so the four ds_read_b64 are one asm statement (the hardware counts them in lgkmcnt like the compiler's
// own: its waits only become stricter), and the wait is an asm statement the words pass THROUGH, so nothing that uses them can
// move above it.  msg must be an LDS address (the low half of its flat address is the LDS offset).
#define B2O_MSG4_REQUEST(COL, DIA, X0, Y0, X1, Y1)                                    \
    do {                                                                             \
        const uint32_t bc_ = ((uint32_t)(COL) >> sh) & 0xFFu, bd_ = ((uint32_t)(DIA) >> sh) & 0xFFu; \
        const uint32_t a0_ = mbase + ((bc_ & 15u) << 3), a1_ = mbase + ((bc_ >> 4) << 3), a2_ = mbase + ((bd_ & 15u) << 3), a3_ = mbase + ((bd_ >> 4) << 3); \
        /* (`a` passes through the request and `b` through the wait: the round's arithmetic starts with a and ends with b, so  */ \
        /* the compiler can neither move a round's arithmetic above the request nor the wait above the previous round's)        */ \
        asm volatile("ds_read_b64 %0, %5\n\tds_read_b64 %1, %6\n\tds_read_b64 %2, %7\n\tds_read_b64 %3, %8"           \
                     : "=&v"(X0), "=&v"(Y0), "=&v"(X1), "=&v"(Y1), "+v"(a) : "v"(a0_), "v"(a1_), "v"(a2_), "v"(a3_) : "memory"); \
    } while (0)
#define B2O_MSG4_ARRIVED(X0, Y0, X1, Y1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(X0), "+v"(Y0), "+v"(X1), "+v"(Y1), "+v"(b))
#define B2O_ROUND4_BODY()                                                             \
    do {                                                                             \
        B2O_G4(mx0, my0);                                                             \
        b = quad_perm64<0x39>(b); c = quad_perm64<0x4E>(c); d = quad_perm64<0x93>(d); \
        B2O_G4(mx1, my1);                                                             \
        b = quad_perm64<0x93>(b); c = quad_perm64<0x4E>(c); d = quad_perm64<0x39>(d); \
    } while (0)
// one round, given the NEXT round's sigma constants
#define B2O_ROUND4_NEXT(NCOL, NDIA)                                                   \
    do {                                                                             \
        uint64_t nx0, ny0, nx1, ny1;                                                 \
        B2O_MSG4_ARRIVED(mx0, my0, mx1, my1);                                         \
        B2O_MSG4_REQUEST(NCOL, NDIA, nx0, ny0, nx1, ny1);                             \
        B2O_ROUND4_BODY();                                                            \
        mx0 = nx0; my0 = ny0; mx1 = nx1; my1 = ny1;                                  \
    } while (0)
// the twelve rounds (sigma of rounds 10 and 11 = sigma of rounds 0 and 1)
#define B2O_ROUNDS4()                                                                 \
    do {                                                                             \
        const uint32_t mbase = (uint32_t)(uintptr_t)(msg);                           \
        uint64_t mx0, my0, mx1, my1;                                                 \
        B2O_MSG4_REQUEST(0x76543210u, 0xfedcba98u, mx0, my0, mx1, my1);               \
        B2O_ROUND4_NEXT(0x6df984aeu, 0x357b20c1u); B2O_ROUND4_NEXT(0xdf250c8bu, 0x491763eau); B2O_ROUND4_NEXT(0xebcd1397u, 0x8f04a562u); \
        B2O_ROUND4_NEXT(0xfa427509u, 0xd386cb1eu); B2O_ROUND4_NEXT(0x38b0a6c2u, 0x91ef57d4u); B2O_ROUND4_NEXT(0xa4def15cu, 0xb8293670u); \
        B2O_ROUND4_NEXT(0x931ce7bdu, 0xa2684f05u); B2O_ROUND4_NEXT(0x803b9ef6u, 0x5a417d2cu); B2O_ROUND4_NEXT(0x5167482au, 0x0dc3e9bfu); \
        B2O_ROUND4_NEXT(0x76543210u, 0xfedcba98u); B2O_ROUND4_NEXT(0x6df984aeu, 0x357b20c1u);                                           \
        B2O_MSG4_ARRIVED(mx0, my0, mx1, my1);                                         \
        B2O_ROUND4_BODY();                                                            \
    } while (0)

__device__ __forceinline__ void blake2b_node_4lane_before(const uint64_t* msg, uint32_t j, uint64_t& h_lo, uint64_t& h_hi) {
    const uint32_t sh = 8u * j;
    const uint64_t iv_a = B2_IV[j], iv_b = B2_IV[4 + j];
    const uint64_t h0 = (j == 0) ? (iv_a ^ 0x01010040ull) : iv_a;
    uint64_t a = h0, b = iv_b, c = iv_a, d = iv_b;
    if (j == 0) d ^= 128ull;
    if (j == 2) d = ~d;
    B2O_ROUNDS4();
    h_lo = h0 ^ a ^ c;
    h_hi = iv_b ^ b ^ d;
}

constexpr int REPS = 64;

template <int VARIANT>
__global__ void __launch_bounds__(256) hash_kernel(const uint64_t* in, uint64_t* out, unsigned long long* ticks) {
    __shared__ uint64_t lin[64 * 17];
    const uint32_t t = threadIdx.x, n = t >> 2, j = t & 3u;
    for (uint32_t i = t; i < 64 * 17; i += 256) lin[i] = in[i];
    __syncthreads();
    uint64_t lo = 0, hi = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < REPS; ++r) {
        if (VARIANT == 0) blake2b_node_4lane_before(lin + 17u * n, j, lo, hi);
        else blake2b_node_4lane(lin + 17u * n, j, lo, hi);
        lin[17u * n + j] = lo;                        // the digest goes back into the message (own quad's slot only)
        lin[17u * n + 4u + j] = hi;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[(size_t)blockIdx.x * 512 + 2 * t] = lo; out[(size_t)blockIdx.x * 512 + 2 * t + 1] = hi;
    if ((t & 63u) == 0) ticks[blockIdx.x * 4 + (t >> 6)] = t1 - t0;
}

// a climb as the tree kernels run it: merkle_level_4lane (message addresses per level, the level's stores to LDS and to the tree)
// and a workgroup barrier per level, 128 digests -> 1, CLIMBS times; cycles per LEVEL
constexpr int CLIMBS = 16;
template <int MODE> __global__ void __launch_bounds__(256) climb_kernel(const uint64_t* in, uint64_t* tree, unsigned long long* ticks) {
    __shared__ uint64_t linA[64 * 17], linB[32 * 17];
    const uint32_t t = threadIdx.x;
    uint64_t* out = tree + (size_t)blockIdx.x * 8 * 128;
    unsigned long long total = 0;
    for (int rep = 0; rep < CLIMBS; ++rep) {
        for (uint32_t i = t; i < 64 * 17; i += 256) linA[i] = in[i] + rep;
        __syncthreads();
        uint64_t* src = linA; uint64_t* dst = linB; uint64_t* o = out;
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (uint32_t w = 128; w > 1; w >>= 1) {
            if (MODE == 1) {          // timing only: no stores of the level to the tree
                const uint32_t n = t >> 2, j = t & 3u;
                if (n < (w >> 1)) { uint64_t lo, hi; blake2b_node_4lane(src + 17u * n, j, lo, hi); dst[lin_off(n) + j] = lo; dst[lin_off(n) + 4u + j] = hi; }
            } else merkle_level_4lane(src, dst, o, w >> 1, t);
            if (MODE == 2 && (w >> 1) <= 16) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
            else __syncthreads();
            o += 8 * (w >> 1);
            uint64_t* x = src; src = dst; dst = x;
        }
        total += __builtin_amdgcn_s_memtime() - t0;
    }
    if ((t & 63u) == 0) ticks[blockIdx.x * 4 + (t >> 6)] = total;
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
            (void)hipDeviceSynchronize();
        }
        (void)hipMemcpy(tk.data(), d_ticks, tk.size() * 8, hipMemcpyDeviceToHost);
        (void)hipMemcpy(o.data(), d_out, o.size() * 8, hipMemcpyDeviceToHost);
        return mean_ticks(tk) / REPS;
    };
    const double c0 = run_hash(0, o0), c1 = run_hash(1, o1);
    size_t bad = 0;
    for (size_t i = 0; i < o0.size(); ++i) bad += o0[i] != o1[i];
    printf("four lanes per compression, one wave per SIMD, %d compressions back to back (s_memtime = shader cycles):\n", REPS);
    printf("  before  (b, c, d rotate; hipcc's association of the adds)               : %7.1f cycles per compression\n", c0);
    printf("  library (csrc/merkle.cuh blake2b_node_4lane: b stays, early asm add, vector rotations) : %7.1f cycles per compression     results identical: %s\n", c1, bad ? "NO" : "yes");
    {
        uint64_t* d_tree; CHK(hipMalloc(&d_tree, (size_t)blocks * 8 * 128 * 8));
        auto climb = [&](auto kernel) -> double {
            for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d_in, d_tree, d_ticks); (void)hipDeviceSynchronize(); }
            (void)hipMemcpy(tk.data(), d_ticks, tk.size() * 8, hipMemcpyDeviceToHost);
            double s0 = 0; for (int b = 0; b < blocks; ++b) s0 += (double)tk[b * 4];
            return s0 / blocks / CLIMBS / 7;
        };
        const double l0 = climb(climb_kernel<0>), l1 = climb(climb_kernel<1>), l2 = climb(climb_kernel<2>);
        printf("  a climb of 7 levels (128 digests -> 1) by merkle_level_4lane + a workgroup barrier per level: %7.1f cycles per level;  without the level's stores to the tree %7.1f;  wave-level sync from 16 parents down %7.1f\n", l0, l1, l2);
    }
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
