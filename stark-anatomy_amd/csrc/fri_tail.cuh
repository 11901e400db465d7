// fri_tail.cuh -- the latency-bound tail of Fri.commit (reference code/fri.py:66-94) as ONE persistent launch.
//
// Once a codeword has at most 2^16 elements a round of the commit phase -- fold (fri.py:85), leaf hashes, the tree, the root -- is a
// chain of dependent BLAKE2b compressions a few microseconds long each, and what the classic loop adds on top is launches: two or
// three per round, the root's trip to the host and the next launch's trip back (profiles/r04/fri_prove_2p22_timeline.txt: ten rounds,
// 523 us, of which ~35 us is hash work that could not overlap).  Here all remaining rounds run inside one kernel:
//
//   per round  phase A  every active workgroup: fold of its slice of the previous codeword with the round's challenge, leaf hashes,
//                       the levels of its own subtree through LDS (four lanes per compression everywhere -- below 2^15 leaves also
//                       for the leaves: 64 leaves per workgroup, one wave per SIMD), its sub-root into an exchange buffer;
//              barrier  a counter in device memory (agent-scope stores / loads: the XCDs' L2s are not coherent with each other);
//              phase B  workgroup 0: the levels above the sub-roots, the root into a pinned host slot (the waiting host sees it a
//                       microsecond later), then it waits for the next challenge in a pinned host word and hands it to the others.
//
// The Fiat-Shamir step itself -- SHAKE-256 over pickle.dumps(transcript), ip.py:18-25 -- stays with the host thread that is polling
// the root slot anyway (csrc/transcript.h): root out and challenge back cost 4.8 us per round, 0.5-0.8 us of it hashing (the blocks in
// front of the pending root are absorbed while the device works), where ONE Keccak-f[1600] permutation takes a lone wave 7.3 us on
// 25 lanes and 8.9 us in one lane (tools/microbench/keccak_wave.hip, profiles/r05/keccak_wave_ubench.txt) -- and no launch separates
// the rounds any more.  Nothing spins for ever:
// every wait gives up after TAIL_SPIN_LIMIT polls and raises the abort flag, and the host then finishes the rounds the classic way.
#pragma once
#include "merkle.cuh"

namespace sc {

constexpr int TAIL_MAX_ROUNDS = 20;
constexpr uint32_t TAIL_MAX_LOG = 16;        // the first codeword the kernel produces has at most 2^16 elements (256 workgroups)
constexpr uint32_t TAIL_QUAD_LOG = 14;       // up to 2^14 leaves: four lanes per leaf hash, 64 leaves per workgroup
constexpr uint32_t TAIL_LAST_MAX = 4096;     // the last codeword also goes to the pinned block (fri.py:91 pushes it in the clear)
constexpr uint32_t TAIL_SPIN_LIMIT = 1u << 21;      // polls before a wait gives up (a fraction of a second)

struct TailRoundDesc {
    Fe* out;              // this round's folded codeword
    uint64_t* levels;     // its tree, (2 n - 1) digests
    Fe i2o_m;             // 1 / (2 offset) of the codeword being folded, Montgomery form
};
// pinned host memory, one block per call; every flag carries the call's sequence number
struct TailHost {
    volatile uint64_t root[TAIL_MAX_ROUNDS][16];      // device -> host: 8 words, then [8] = seq
    volatile uint64_t alpha[TAIL_MAX_ROUNDS][8];      // host -> device: lo, hi, then [2] = seq  (alpha[k]: the challenge of round k's fold)
    volatile uint64_t last_flag[8];                   // [0] = seq once `last` holds the last codeword
    volatile uint64_t abort_flag[8];                  // [0] = seq if a wait timed out
    volatile uint64_t stamps[TAIL_MAX_ROUNDS][8];     // diagnostics (TailParams::trace): workgroup 0's 100 MHz clock at the phase boundaries of each round
    Fe last[TAIL_LAST_MAX];
};
// device memory, zero when handed to a call and left zero by a call that completes
struct TailCtl {
    uint32_t arrive[TAIL_MAX_ROUNDS][16];             // one 64-byte line per round
    uint64_t alpha[TAIL_MAX_ROUNDS][8];               // lo, hi, flag (= seq)
    uint32_t abort[16];
    uint64_t exchange[256 * 8];                       // the sub-roots of one round
};
struct TailParams {
    const Fe* in0;        // the codeword the first round folds, 2 * 2^log_n0 elements
    uint32_t log_n0;      // the first codeword produced has 2^log_n0 elements
    uint32_t rounds;      // rounds to run (>= 1)
    Fe alpha0;            // the first fold's challenge, canonical
    const Fe* pw_lo;      // two-level power table of 1 / omega of in0 (Montgomery): lo[4096], hi[...], exponents < 2^log_n0
    const Fe* pw_hi;
    TailRoundDesc rd[TAIL_MAX_ROUNDS];
    TailCtl* ctl;
    TailHost* host;
    uint64_t seq;
    int trace;            // workgroup 0 stamps its phases into host->stamps
    uint32_t spin_limit;  // polls before a wait gives up (TAIL_SPIN_LIMIT; tests shorten it)
    uint64_t* alpha_bar;  // nullptr, or [TAIL_MAX_ROUNDS][8] words of FINE-GRAINED DEVICE memory the host writes the challenges to through the BAR
                          // (lo, hi, then -- behind an sfence -- [2] = seq): every workgroup polls it there, no read crosses the bus
};

#if defined(__HIPCC__)

__device__ __forceinline__ uint64_t ld_agent(const uint64_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint64_t ld_system(const uint64_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void st_agent(uint64_t* p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ Fe ld_fe_agent(const Fe* p) { return Fe{ld_agent(&p->lo), ld_agent(&p->hi)}; }
__device__ __forceinline__ void st_fe_agent(Fe* p, Fe v) { st_agent(&p->lo, v.lo); st_agent(&p->hi, v.hi); }
__device__ __forceinline__ void wait_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// single-block BLAKE2b-512 of a message of `len` <= 128 bytes held zero-padded in msg[0..16) (LDS), by the 4 lanes of a quad
__device__ __forceinline__ void blake2b_block_4lane(const uint64_t* msg, uint32_t len, uint32_t j, uint64_t& h_lo, uint64_t& h_hi) {
    const uint32_t sh = 8u * j;
    const uint64_t iv_a = B2_IV[j], iv_b = B2_IV[4 + j];
    const uint64_t h0 = (j == 0) ? (iv_a ^ 0x01010040ull) : iv_a;
    uint64_t a = h0, b = iv_b, c = iv_a, d = iv_b;
    if (j == 0) d ^= (uint64_t)len;   // t0 = message length
    if (j == 2) d = ~d;               // final block
    B2_ROUNDS4();
    h_lo = h0 ^ a ^ c;
    h_hi = iv_b ^ b ^ d;
}

// one level, four lanes per hash, any width: `parents` nodes from the 2 * parents digests in `src` (lin layout, merkle.cuh) into
// `dst` (lin layout) and to the tree (`out`: the level's place in global memory); 256 threads = 64 nodes per sweep
__device__ __forceinline__ void tail_level(const uint64_t* src, uint64_t* dst, uint64_t* __restrict__ out, uint32_t parents, uint32_t t) {
    const uint32_t j = t & 3u;
    for (uint32_t base = 0; base < parents; base += 64u) {
        const uint32_t n = base + (t >> 2);
        if (n < parents) {
            uint64_t lo, hi;
            blake2b_node_4lane(src + 17u * n, j, lo, hi);
            dst[lin_off(n) + j] = lo;
            dst[lin_off(n) + 4u + j] = hi;
            out[8u * n + j] = lo;
            out[8u * n + 4u + j] = hi;
        }
    }
}

// workgroups a round over 2^logn leaves keeps busy
__host__ __device__ __forceinline__ uint32_t tail_workgroups(uint32_t logn) {
    const uint32_t per_wg = logn <= TAIL_QUAD_LOG ? 64u : 256u, n = 1u << logn;
    return n > per_wg ? n / per_wg : 1u;
}

__global__ void __launch_bounds__(256) fri_tail_kernel(const TailParams P) {
    __shared__ uint64_t linA[128 * 17], linB[64 * 17];     // up to 256 resp. 128 digests in the lin layout (linB: also the leaf messages)
    __shared__ Fe s_alpha;
    __shared__ uint32_t s_abort;
    const uint32_t t = threadIdx.x, wg = blockIdx.x;
    TailCtl* const ctl = P.ctl;
    if (t == 0) s_abort = 0;
    Fe alpha = P.alpha0;
    const Fe* prev = P.in0;
    for (uint32_t r = 0; r < P.rounds; ++r) {
        const uint32_t logn = P.log_n0 - r, n = 1u << logn;
        const bool quad = logn <= TAIL_QUAD_LOG;
        const uint32_t per_wg = quad ? 64u : 256u;
        const uint32_t nwg = n > per_wg ? n / per_wg : 1u;           // workgroups at work in this round
        const uint32_t here = n < per_wg ? n : per_wg;               // leaves of each of them
        uint64_t* const levels = P.rd[r].levels;
        auto level_off = [n](uint32_t l) -> uint64_t { return l == 0 ? 0 : 2ull * n - ((uint64_t)n >> (l - 1)); };
        const bool last_round = r + 1 == P.rounds;
        uint32_t l = 0;                                               // level held in LDS
        uint64_t* src = linA;
        uint64_t* dst = linB;
        auto stamp = [&](int k) { if (P.trace && wg == 0 && t == 0) P.host->stamps[r][k] = __builtin_amdgcn_s_memrealtime(); };
        stamp(0);
        if (wg < nwg) {
            const Fe c_m = mont_mul(to_mont(alpha), P.rd[r].i2o_m);   // alpha / (2 offset), Montgomery form
            auto fold = [&](uint32_t i) -> Fe {
                const Fe a = ld_fe_agent(prev + i), b = ld_fe_agent(prev + i + n);
                const uint64_t e = (uint64_t)i << r;                   // (1/omega_r)^i = (1/omega_0)^(i 2^r)
                const Fe w = mont_mul(mont_mul(P.pw_lo[e & 4095u], P.pw_hi[e >> 12]), c_m);
                return fe_add(fe_half(fe_add(a, b)), mont_mul(fe_sub(a, b), w));
            };
            if (quad) {
                const uint32_t q = t >> 2, j = t & 3u, i = wg * 64u + q;
                uint32_t len = 0;
                if (q < here) {
                    const Fe e = fold(i);                               // (all four lanes: the same loads, the same arithmetic)
                    uint64_t m[16];
                    len = leaf_message(e, m);
                    if (j == 0) {
                        st_fe_agent(P.rd[r].out + i, e);
                        if (last_round && n <= TAIL_LAST_MAX) { P.host->last[i].lo = e.lo; P.host->last[i].hi = e.hi; }
#pragma unroll
                        for (int w = 0; w < 16; ++w) linB[17u * q + w] = m[w];
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (q < here) {
                    uint64_t lo, hi;
                    blake2b_block_4lane(linB + 17u * q, len, j, lo, hi);
                    linA[lin_off(q) + j] = lo;
                    linA[lin_off(q) + 4u + j] = hi;
                    levels[8ull * i + j] = lo;
                    levels[8ull * i + 4u + j] = hi;
                }
            } else {
                const uint32_t i = wg * 256u + t;
                const Fe e = fold(i);
                st_fe_agent(P.rd[r].out + i, e);
                if (last_round && n <= TAIL_LAST_MAX) { P.host->last[i].lo = e.lo; P.host->last[i].hi = e.hi; }
                uint64_t m[16], h[8];
                const uint32_t len = leaf_message(e, m);
                blake2b_single_block(m, len, h);
                ulonglong2* o = reinterpret_cast<ulonglong2*>(levels + 8ull * i);
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = make_ulonglong2(h[2 * k], h[2 * k + 1]);
#pragma unroll
                for (int w = 0; w < 8; ++w) linA[lin_off(t) + w] = h[w];
            }
            if (last_round && n <= TAIL_LAST_MAX) {
                // the last codeword is on its way to the host: flag it once every workgroup's stores are behind a system-scope fence
                // (at most TAIL_LAST_MAX / 64 workgroups; they count on arrive[TAIL_MAX_ROUNDS - 1], which no round uses)
                __threadfence_system();
                __syncthreads();
                if (t == 0) {
                    const uint32_t got = __hip_atomic_fetch_add(&ctl->arrive[TAIL_MAX_ROUNDS - 1][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (got == nwg - 1) {
                        __hip_atomic_store(&ctl->arrive[TAIL_MAX_ROUNDS - 1][0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __threadfence_system();
                        P.host->last_flag[0] = P.seq;
                    }
                }
            }
            wait_stores();                                             // this wave's codeword stores have reached memory
            __syncthreads();
            stamp(1);
            // the levels of this workgroup's subtree
            for (uint32_t width = here; width > 1; width >>= 1) {
                tail_level(src, dst, levels + 8ull * (level_off(l + 1) + (uint64_t)wg * (width >> 1)), width >> 1, t);
                __syncthreads();
                uint64_t* s = src; src = dst; dst = s;
                ++l;
            }
            stamp(2);
            if (nwg > 1) {                                             // sub-root to the exchange, then arrive
                if (t < 8) st_agent(&ctl->exchange[wg * 8u + t], src[t]);
                if (t < 64) wait_stores();
                if (t == 0) __hip_atomic_fetch_add(&ctl->arrive[r][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        // workgroups any LATER round needs (not monotone: 2^15 leaves are 128 workgroups of 256, 2^14 leaves 256 workgroups of 64)
        uint32_t nwg_next = 0;
        for (uint32_t k = r + 1; k < P.rounds; ++k) { const uint32_t w = tail_workgroups(P.log_n0 - k); nwg_next = w > nwg_next ? w : nwg_next; }
        if (wg == 0) {
            if (nwg > 1) {
                if (t == 0) {
                    uint32_t spins = 0;
                    while (__hip_atomic_load(&ctl->arrive[r][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nwg) {
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > P.spin_limit || __hip_atomic_load(&ctl->abort[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { s_abort = 1; break; }
                    }
                    __hip_atomic_store(&ctl->arrive[r][0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // left zero for the next call
                }
                __syncthreads();
                stamp(3);
                if (!s_abort) {
                    for (uint32_t q = t; q < nwg * 8u; q += 256u) linA[lin_off(q >> 3) + (q & 7u)] = ld_agent(&ctl->exchange[q]);
                    __syncthreads();
                    src = linA; dst = linB;
                    for (uint32_t width = nwg; width > 1; width >>= 1) {
                        tail_level(src, dst, levels + 8ull * level_off(l + 1), width >> 1, t);
                        __syncthreads();
                        uint64_t* s = src; src = dst; dst = s;
                        ++l;
                    }
                }
            }
            if (!s_abort) {
                // the root: 8 words, then -- ordered behind them -- the sequence number the host polls
                if (t < 8) P.host->root[r][t] = src[t];
                __threadfence_system();
                __syncthreads();
                if (t == 0) P.host->root[r][8] = P.seq;
                stamp(4);
                if (!last_round) {
                    if (t == 0 && P.alpha_bar) {
                        const uint64_t* slot = P.alpha_bar + 8u * (r + 1);
                        uint32_t spins = 0;
                        while (ld_system(slot + 2) != P.seq) {
                            __builtin_amdgcn_s_sleep(1);
                            if (++spins > P.spin_limit) { s_abort = 1; break; }
                        }
                        if (!s_abort) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, ""); s_alpha = Fe{ld_system(slot), ld_system(slot + 1)}; }
                    } else if (t == 0) {
                        uint32_t spins = 0;
                        while (P.host->alpha[r + 1][2] != P.seq) {
                            __builtin_amdgcn_s_sleep(1);
                            if (++spins > P.spin_limit) { s_abort = 1; break; }
                        }
                        if (!s_abort) {
                            const Fe a{P.host->alpha[r + 1][0], P.host->alpha[r + 1][1]};
                            s_alpha = a;
                            if (nwg_next > 1) {
                                st_agent(&ctl->alpha[r + 1][0], a.lo);
                                st_agent(&ctl->alpha[r + 1][1], a.hi);
                                wait_stores();
                                st_agent(&ctl->alpha[r + 1][2], P.seq);
                            }
                        }
                    }
                    __syncthreads();
                    stamp(5);
                }
            }
            if (s_abort) {
                if (t == 0) {
                    __hip_atomic_store(&ctl->abort[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    P.host->abort_flag[0] = P.seq;
                }
                return;
            }
            alpha = s_alpha;
        } else {
            if (wg >= nwg_next) return;                                // no later round has work for this workgroup
            if (t == 0 && P.alpha_bar) {
                const uint64_t* slot = P.alpha_bar + 8u * (r + 1);
                uint32_t spins = 0;
                while (ld_system(slot + 2) != P.seq) {
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > P.spin_limit || __hip_atomic_load(&ctl->abort[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { s_abort = 1; break; }
                }
                if (!s_abort) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, ""); s_alpha = Fe{ld_system(slot), ld_system(slot + 1)}; }
            } else if (t == 0) {
                uint32_t spins = 0;
                while (ld_agent(&ctl->alpha[r + 1][2]) != P.seq) {
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > P.spin_limit || __hip_atomic_load(&ctl->abort[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { s_abort = 1; break; }
                }
                if (!s_abort) s_alpha = Fe{ld_agent(&ctl->alpha[r + 1][0]), ld_agent(&ctl->alpha[r + 1][1])};
            }
            __syncthreads();
            if (s_abort) {
                if (t == 0) __hip_atomic_store(&ctl->abort[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
            alpha = s_alpha;
        }
        prev = P.rd[r].out;
    }
}

#endif  // __HIPCC__

}  // namespace sc
