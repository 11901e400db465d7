// What would a device-side SHAKE-256 cost on the critical path of Fri.commit's tail?  (dev tool)
// Keccak-f[1600] -- the permutation of SHAKE-256 (csrc/transcript.h has the host's) -- as a LONE wave executes it, two ways:
//   one lane    the whole 25-word state in one lane's registers, every 64-bit operation on the 32-bit halves (what a kernel that
//               just calls a scalar Keccak gets)
//   25 lanes    lane x + 5 y holds word (x, y): theta's column parities and their neighbours, the rho-pi permutation and chi's two
//               row neighbours are cross-lane reads (__shfl = ds_bpermute_b32 on each half)
// against the host loop of the prover (the same permutation in portable C) -- the number next to which the measured 4.8 us of the
// persistent kernel's root -> host -> challenge round trip has to be read.  Outputs are compared with the host's.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/keccak_wave tools/microbench/keccak_wave.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

static const uint64_t RC_H[24] = {0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull, 0x000000000000808bull, 0x0000000080000001ull,
                                  0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000aull,
                                  0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull, 0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull,
                                  0x000000000000800aull, 0x800000008000000aull, 0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
static const int ROT_H[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
__device__ __constant__ uint64_t RC_D[24];
__device__ __constant__ int ROT_D[25];
__device__ __constant__ int PI_SRC[25];      // the lane whose (rotated) word lands in this lane after pi

static inline uint64_t rotl_h(uint64_t x, int n) { return n ? (x << n) | (x >> (64 - n)) : x; }
static void keccak_host(uint64_t s[25]) {
    for (int round = 0; round < 24; ++round) {
        uint64_t c[5], d[5], b[25];
        for (int x = 0; x < 5; ++x) c[x] = s[x] ^ s[x + 5] ^ s[x + 10] ^ s[x + 15] ^ s[x + 20];
        for (int x = 0; x < 5; ++x) d[x] = c[(x + 4) % 5] ^ rotl_h(c[(x + 1) % 5], 1);
        for (int i = 0; i < 25; ++i) s[i] ^= d[i % 5];
        for (int x = 0; x < 5; ++x)
            for (int y = 0; y < 5; ++y) b[y + 5 * ((2 * x + 3 * y) % 5)] = rotl_h(s[x + 5 * y], ROT_H[x + 5 * y]);
        for (int y = 0; y < 5; ++y)
            for (int x = 0; x < 5; ++x) s[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
        s[0] ^= RC_H[round];
    }
}

__device__ __forceinline__ uint64_t rotl_d(uint64_t x, int n) { return n ? (x << n) | (x >> (64 - n)) : x; }

// one lane: `perms` permutations of the state at `state`
__global__ void __launch_bounds__(64) keccak_one_lane(uint64_t* state, int perms) {
    if (threadIdx.x != 0) return;
    uint64_t s[25];
    for (int i = 0; i < 25; ++i) s[i] = state[i];
    for (int p = 0; p < perms; ++p) {
#pragma unroll 1
        for (int round = 0; round < 24; ++round) {
            uint64_t c[5], d[5], b[25];
#pragma unroll
            for (int x = 0; x < 5; ++x) c[x] = s[x] ^ s[x + 5] ^ s[x + 10] ^ s[x + 15] ^ s[x + 20];
#pragma unroll
            for (int x = 0; x < 5; ++x) d[x] = c[(x + 4) % 5] ^ rotl_d(c[(x + 1) % 5], 1);
#pragma unroll
            for (int i = 0; i < 25; ++i) s[i] ^= d[i % 5];
#pragma unroll
            for (int x = 0; x < 5; ++x)
#pragma unroll
                for (int y = 0; y < 5; ++y) {
                    constexpr int R[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
                    b[y + 5 * ((2 * x + 3 * y) % 5)] = rotl_d(s[x + 5 * y], R[x + 5 * y]);
                }
#pragma unroll
            for (int y = 0; y < 5; ++y)
#pragma unroll
                for (int x = 0; x < 5; ++x) s[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
            s[0] ^= RC_D[round];
        }
    }
    for (int i = 0; i < 25; ++i) state[i] = s[i];
}

__device__ __forceinline__ uint64_t shfl64(uint64_t v, int src) {
    const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, src), hi = (uint32_t)__shfl((int)(uint32_t)(v >> 32), src);
    return ((uint64_t)hi << 32) | lo;
}

// 25 lanes: lane l = x + 5 y holds word (x, y)
__global__ void __launch_bounds__(64) keccak_25_lanes(uint64_t* state, int perms) {
    const int l = threadIdx.x, x = l % 5, y = l / 5;
    const bool live = l < 25;
    uint64_t a = live ? state[l] : 0;
    const int rot = live ? ROT_D[l] : 0, pi_src = live ? PI_SRC[l] : l;
    const int up1 = live ? (x + 1) % 5 + 5 * y : l, up2 = live ? (x + 2) % 5 + 5 * y : l;       // chi's row neighbours
    const int colm = (x + 4) % 5, colp = (x + 1) % 5;                                              // theta's neighbour columns
    for (int p = 0; p < perms; ++p) {
#pragma unroll 1
        for (int round = 0; round < 24; ++round) {
            // theta: parity of column x - 1 and of column x + 1 (five words each), fetched directly
            uint64_t cm = 0, cp = 0;
#pragma unroll
            for (int yy = 0; yy < 5; ++yy) { cm ^= shfl64(a, colm + 5 * yy); cp ^= shfl64(a, colp + 5 * yy); }
            a ^= cm ^ rotl_d(cp, 1);
            // rho (this lane's rotation), then pi (pull from the lane whose word lands here)
            const uint64_t r = rot ? (a << rot) | (a >> (64 - rot)) : a;
            const uint64_t b = shfl64(r, pi_src);
            // chi: the two next words of the row
            const uint64_t b1 = shfl64(b, up1), b2 = shfl64(b, up2);
            a = b ^ (~b1 & b2);
            if (l == 0) a ^= RC_D[round];
        }
    }
    if (live) state[l] = a;
}

int main() {
    int pi_src[25];
    for (int xx = 0; xx < 5; ++xx)
        for (int yy = 0; yy < 5; ++yy) pi_src[yy + 5 * ((2 * xx + 3 * yy) % 5)] = xx + 5 * yy;
    CHK(hipMemcpyToSymbol(HIP_SYMBOL(RC_D), RC_H, sizeof RC_H));
    CHK(hipMemcpyToSymbol(HIP_SYMBOL(ROT_D), ROT_H, sizeof ROT_H));
    CHK(hipMemcpyToSymbol(HIP_SYMBOL(PI_SRC), pi_src, sizeof pi_src));
    uint64_t h[25], want[25], got[25];
    for (int i = 0; i < 25; ++i) h[i] = 0x9E3779B97F4A7C15ull * (uint64_t)(i + 1) ^ ((uint64_t)i << 40);
    uint64_t* d;
    CHK(hipMalloc(&d, sizeof h));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int perms = 200;
    memcpy(want, h, sizeof h);
    const auto t0 = std::chrono::steady_clock::now();
    for (int p = 0; p < perms; ++p) keccak_host(want);
    const double host_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / perms;
    printf("keccak_wave: Keccak-f[1600], %d permutations back to back, one wave on an otherwise idle MI355X\n", perms);
    printf("  host (portable C, csrc/transcript.h's loop)   %7.3f us per permutation\n", host_us);
    for (int variant = 0; variant < 2; ++variant) {
        float best = 1e30f;
        bool same = true;
        for (int rep = 0; rep < 5; ++rep) {
            CHK(hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice));
            hipEventRecord(e0, 0);
            if (variant == 0) hipLaunchKernelGGL(keccak_one_lane, dim3(1), dim3(64), 0, 0, d, perms);
            else hipLaunchKernelGGL(keccak_25_lanes, dim3(1), dim3(64), 0, 0, d, perms);
            hipEventRecord(e1, 0);
            CHK(hipEventSynchronize(e1));
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
            CHK(hipMemcpy(got, d, sizeof got, hipMemcpyDeviceToHost));
            same = same && memcmp(got, want, sizeof got) == 0;
        }
        printf("  device, %-36s %7.3f us per permutation   state %s the host's\n", variant == 0 ? "one lane (64-bit ops on 32-bit halves)" : "25 lanes (cross-lane reads: ds_bpermute)",
               1e3f * best / perms, same ? "equals" : "DIFFERS FROM");
    }
    printf("  (a round of Fri.commit has one or two permutations on its critical path: the blocks that hold the pending root)\n");
    return 0;
}
