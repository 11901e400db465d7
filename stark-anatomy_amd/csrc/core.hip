// core.hip -- host state, the device-memory pool, streams, the table caches, the transform planner and its pass launches, and the
// C-ABI entries for vectors, randomness, transforms, coset evaluation / division, pointwise work and MPolynomial evaluation.
// gfx950 only; see include/starkcore.h for the contract and the reference lines each entry replaces.
#include "core.h"

// ============================================================================ kernels

// threads per workgroup are capped per LOGE so the register allocator gets the budget the tile needs
template <int LOGE> struct PassThreads { static constexpr int value = LOGE >= 4 ? 256 : (LOGE == 3 ? 512 : 1024); };

template <int LOGE>
__global__ void __launch_bounds__(PassThreads<LOGE>::value) ntt_pass_kernel(const PassParams P, uint32_t ntiles, int xcd_remap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Fe* lds = reinterpret_cast<Fe*>(smem_raw);
    // XCD-aware tile mapping: workgroup b runs on XCD b % 8; give each XCD a contiguous range of tiles so
    // that neighbouring tiles (which share twiddle rows and adjacent memory) stay within one L2.
    // (a launch over several columns -- P.col_enable -- repeats the same mapping column after column: ntiles is per column)
    uint32_t wg = blockIdx.x, colbits = 0;
    if (P.col_enable) { colbits = (wg >> P.col_tiles_log) << P.col_tiles_log; wg -= colbits; }
    uint32_t tile = wg;
    if (xcd_remap) tile = (wg & 7u) * (ntiles >> 3) + (wg >> 3);
    tile |= colbits;
    Fe* tw = lds + (1u << (P.logR + P.logC));
    tile_twiddles_to_lds(P, P.logR, threadIdx.x, blockDim.x, tw);
    __syncthreads();
    // same schedule as make_rounds() (short round first), computed inline to keep it in SGPRs
    const int nrounds = (P.logR + LOGE - 1) / LOGE;
    int sh = P.logR;
    for (int r = 0; r < nrounds; ++r) {
        const int s = (r == 0) ? (P.logR - LOGE * (nrounds - 1)) : LOGE;
        sh -= s;
        ntt_round_dispatch<LOGE>(P, s, sh, r == 0, tile, threadIdx.x, lds, tw);
        if (r + 1 < nrounds) __syncthreads();
    }
}

// the same kernel with the tile geometry fixed at compile time (hot shapes of the default plans); see FixedRounds for what
// it does differently (one memory latency per workgroup, wave-level fences once the exchanges stay inside a wave).
// TRACE instantiations stamp s_memtime per wave at every phase boundary into P.trace (tools/pass_trace.py).
constexpr int TRACE_STAMPS = 16;
template <int LOGE, int GLR, int GLC, bool TRACE, bool ALT = false>
__global__ void __launch_bounds__(1 << (GLR + GLC - LOGE)) ntt_pass_kernel_fixed(const PassParams P, uint32_t ntiles, int xcd_remap, int wave_local) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Fe* lds = reinterpret_cast<Fe*>(smem_raw);
    unsigned long long* trow = nullptr;
    if constexpr (TRACE) {
        if (P.trace) {
            trow = P.trace + ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * TRACE_STAMPS;
            if ((threadIdx.x & 63u) == 0) { trow[0] = __builtin_amdgcn_s_memtime(); trow[15] = __builtin_amdgcn_s_memrealtime(); }
        }
    }
    uint32_t wg = blockIdx.x, colbits = 0;
    if (P.col_enable) { colbits = (wg >> P.col_tiles_log) << P.col_tiles_log; wg -= colbits; }
    uint32_t tile = wg;
    if (xcd_remap) tile = (wg & 7u) * (ntiles >> 3) + (wg >> 3);
    tile |= colbits;
    Fe* tw = lds + (1u << (GLR + GLC));
    auto stamp = [&](int i) {
        if constexpr (TRACE) {
            if (trow) {
                if (i == 2) __builtin_amdgcn_s_waitcnt(0);          // loads landed (trace only: separates latency from arithmetic)
                if ((threadIdx.x & 63u) == 0 && i < 13) trow[i] = __builtin_amdgcn_s_memtime();
            }
        }
    };
    auto sync = [] { __syncthreads(); };
    auto wsync = [] { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); };
    FixedRounds<LOGE, GLR, GLC, 0, ALT>::run(P, tile, threadIdx.x, lds, tw, sync, wsync, stamp, wave_local != 0);
    if constexpr (TRACE) {
        if (trow && (threadIdx.x & 63u) == 0) { __builtin_amdgcn_s_waitcnt(0); trow[14] = __builtin_amdgcn_s_memtime(); trow[13] = __builtin_amdgcn_s_memrealtime(); }
    }
}

// The 2^12-element shapes with EIGHT elements per thread, for the batches of columns (NttTuning::loge_cols): 512 threads, three stages per
// round, and the register budget of four waves per SIMD (128 VGPRs, one spilled), so that two workgroups share a CU and one computes
// while the other loads or drains.  No trace, no second destination (those launches take the generic kernel).
template <int GLR, int GLC>
__global__ void __launch_bounds__(1 << (GLR + GLC - 3)) __attribute__((amdgpu_waves_per_eu(4)))
ntt_pass_kernel_fixed8(const PassParams P, uint32_t ntiles, int xcd_remap, int wave_local) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Fe* lds = reinterpret_cast<Fe*>(smem_raw);
    uint32_t wg = blockIdx.x, colbits = 0;
    if (P.col_enable) { colbits = (wg >> P.col_tiles_log) << P.col_tiles_log; wg -= colbits; }
    uint32_t tile = wg;
    if (xcd_remap) tile = (wg & 7u) * (ntiles >> 3) + (wg >> 3);
    tile |= colbits;
    Fe* tw = lds + (1u << (GLR + GLC));
    auto stamp = [](int) {};
    auto sync = [] { __syncthreads(); };
    auto wsync = [] { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); };
    FixedRounds<3, GLR, GLC, 0, false>::run(P, tile, threadIdx.x, lds, tw, sync, wsync, stamp, wave_local != 0);
}

__global__ void __launch_bounds__(256) pow_table_kernel(Fe* out, uint64_t count, Fe base_m, uint64_t step, Fe scale_m) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = pow_table_entry(base_m, i, step, scale_m);
}

// direct four-step twiddle table for one column pass: out[k * B + b] = w^(b * k * scale_exp) [* n^-1 via th]
__global__ void __launch_bounds__(256) twiddle_table_kernel(Fe* out, int logB, uint64_t count, uint64_t scale_exp, const Fe* __restrict__ tl, const Fe* __restrict__ th) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    uint64_t k = i >> logB, b = i & ((1ull << logB) - 1);
    out[i] = pow2level(tl, th, b * k * scale_exp);
}

// out = a * b (canonical in, canonical out)
__global__ void __launch_bounds__(256) pointwise_mul_kernel(const Fe* __restrict__ a, const Fe* __restrict__ b, Fe* __restrict__ out, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = fe_mul(a[i], b[i]);
}

// out = a / b with Montgomery's batch-inversion trick, K elements per thread (strided for coalescing).
// flag[0] |= 1 if any divisor is zero (Field.divide asserts, code/algebra.py:91-94).
template <int K>
__global__ void __launch_bounds__(256) pointwise_div_kernel(const Fe* __restrict__ a, const Fe* __restrict__ b, Fe* __restrict__ out, uint64_t n, uint32_t* flag) {
    const uint64_t nthreads = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    Fe bm[K], pre[K];
    Fe acc = fe_mont_one();
    bool zero = false;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        uint64_t i = t + (uint64_t)k * nthreads;
        Fe v = (i < n) ? b[i] : fe_one();
        zero |= fe_is_zero(v);
        bm[k] = to_mont(v);
        pre[k] = acc;                  // product of bm[0..k)
        acc = mont_mul(acc, bm[k]);
    }
    if (zero) atomicOr(flag, 1u);
    Fe inv = mont_inv(acc);
#pragma unroll
    for (int k = K - 1; k >= 0; --k) {
        uint64_t i = t + (uint64_t)k * nthreads;
        Fe ik = mont_mul(inv, pre[k]);             // (b_k)^-1 in Montgomery form
        inv = mont_mul(inv, bm[k]);
        if (i < n) out[i] = mont_mul(a[i], ik);
    }
}

// out[i] = b(offset * root^i) for a SHORT polynomial b (nb <= SMALL_DIVISOR coefficients, canonical): what ntt(scale(b)) computes with
// three passes over `order` elements, by Horner at every point -- the two- and three-coefficient boundary zerofiers of
// fast_stark.py:93-98 are divided by on a 2^21-point coset twice per proof.  Same values (exact arithmetic), same place in the flow.
constexpr uint64_t SMALL_DIVISOR = 8;
__global__ void __launch_bounds__(256) short_poly_coset_kernel(const Fe* __restrict__ b, uint32_t nb, Fe off_m, const Fe* __restrict__ tl, const Fe* __restrict__ th, Fe* __restrict__ out, uint64_t order) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= order) return;
    const Fe x_m = mont_mul(off_m, pow2level(tl, th, i));          // (offset * root^i) in Montgomery form
    Fe acc = b[nb - 1];
    for (uint32_t k = nb - 1; k-- > 0;) acc = fe_add(mont_mul(acc, x_m), b[k]);
    out[i] = acc;
}

// out[i] = in[i] * base^i  (Polynomial.scale, code/univariate.py:153-154) via the two-level power table
__global__ void __launch_bounds__(256) scale_pow_kernel(const Fe* __restrict__ in, Fe* __restrict__ out, uint64_t n, const Fe* __restrict__ lo, const Fe* __restrict__ hi) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = mont_mul(in[i], pow2level(lo, hi, i));
}

// the same scaling on a column slab [rows][2^logcols] of a vector viewed as a rows x row_len matrix:
// out[r][c] = in[r][c] * base^(r * row_len + col_base + c)   (Polynomial.scale on the rank's columns, multi-GPU LDE / coset division)
__global__ void __launch_bounds__(256) scale_slab_kernel(const Fe* __restrict__ in, Fe* __restrict__ out, uint64_t rows, int logcols, uint64_t row_len, uint64_t col_base,
                                                         const Fe* __restrict__ lo, const Fe* __restrict__ hi) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (rows << logcols)) return;
    uint64_t r = t >> logcols, c = t & ((1ull << logcols) - 1);
    out[t] = mont_mul(in[t], pow2level(lo, hi, r * row_len + col_base + c));
}

// acc[shift + j] += weight * src[j]: one term of the nonlinear combination of code/fast_stark.py:130-145 -- `Polynomial([w]) * term`
// and `(x ^ shift) * term` are a scaling and an index shift of the coefficient vector (w_m: weight in Montgomery form)
__global__ void __launch_bounds__(256) axpy_shift_kernel(Fe* __restrict__ acc, const Fe* __restrict__ src, uint64_t n_src, uint64_t shift, Fe w_m) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_src) return;
    acc[shift + j] = fe_add(acc[shift + j], mont_mul(src[j], w_m));
}

// out[0] = max index of a non-zero element, or -1 (Polynomial.degree, code/univariate.py:7-17, on a coefficient vector in HBM)
// (one atomic per WAVE that holds a non-zero element, on the wave's highest such index: a dense vector used to issue one
// contended atomic per element -- 129 us per call at 2^21 coefficients, 9 % of the GPU time of a 2^24 proof)
// A grid of at most 2048 workgroups strides over the vector, four rows of 256 elements per step (every load a coalesced 16-byte-per-
// lane stream, four in flight per thread): the scan of the 12.6 M coefficients that must vanish after an exact division on a 2^24
// domain ran at 1.06 TB/s with one element per thread and one workgroup per 256 elements.
constexpr int DEGREE_PER_THREAD = 4;
constexpr unsigned DEGREE_MAX_BLOCKS = 2048;
constexpr int DEGREE_SLOTS = 8;                        // 64 bytes: what read_small_polled carries in one transfer
__global__ void __launch_bounds__(256) vec_degree_kernel(const Fe* __restrict__ v, uint64_t n, long long* out) {
    const uint64_t step = (uint64_t)gridDim.x * (256 * DEGREE_PER_THREAD);
    long long best = -1;                               // (wave-uniform) the highest non-zero index this wave has seen
    for (uint64_t base = (uint64_t)blockIdx.x * (256 * DEGREE_PER_THREAD) + threadIdx.x; base - threadIdx.x < n; base += step) {
        Fe x[DEGREE_PER_THREAD];
#pragma unroll
        for (int k = 0; k < DEGREE_PER_THREAD; ++k) {
            const uint64_t i = base + 256u * k;
            x[k] = i < n ? v[i] : Fe{0, 0};
        }
#pragma unroll
        for (int k = 0; k < DEGREE_PER_THREAD; ++k) {
            const unsigned long long lanes = __ballot(!fe_is_zero(x[k]));
            if (lanes) best = (long long)(base - (threadIdx.x & 63u) + 256u * k + (63 - __clzll((long long)lanes)));      // (later rows and steps are higher)
        }
    }
    // One atomic per WORKGROUP that saw a non-zero element, spread over DEGREE_SLOTS words (the host takes their maximum): every
    // wave of a strided grid finishes at the same moment, and 8192 atomics on one word cost twice the scan of a dense 2^24 vector.
    __shared__ long long wave_best[4];
    if ((threadIdx.x & 63u) == 0) wave_best[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long m = wave_best[0];
#pragma unroll
        for (int w = 1; w < 4; ++w) m = wave_best[w] > m ? wave_best[w] : m;
        if (m >= 0) atomicMax(out + (blockIdx.x & (DEGREE_SLOTS - 1)), m);
    }
}
// the maximum of the kernel's DEGREE_SLOTS words (each -1 or an index), read in one polled transfer
static int degree_read(void* fl, hipStream_t st, long long* deg) {
    long long slots[DEGREE_SLOTS];
    SCCHK(read_small_polled(fl, sizeof slots, st, slots));
    long long m = -1;
    for (int i = 0; i < DEGREE_SLOTS; ++i) m = slots[i] > m ? slots[i] : m;
    *deg = m;
    return SC_OK;
}
static inline unsigned degree_blocks(uint64_t n) {
    const uint64_t b = (n + 256 * DEGREE_PER_THREAD - 1) / (256 * DEGREE_PER_THREAD);
    return (unsigned)(b < DEGREE_MAX_BLOCKS ? (b ? b : 1) : DEGREE_MAX_BLOCKS);
}

// MPolynomial.evaluate_symbolic (code/multivariate.py:83-90) in the VALUE domain: the AIR polynomial evaluated pointwise on
// the values of the point polynomials; vals_m: [nvars][n] in Montgomery form, coef_m: [nterms] in Montgomery form,
// exps: [nterms][nvars].  out[i] = sum_t coef[t] * prod_j vals[j][i]^exps[t][j]  (canonical).  The term loop is uniform
// across the wave (scalar control flow); the value loads are coalesced and stay in L1 across the terms.
// where != nullptr: variable j is not stored but read off another one -- its value at point i is the value of variable where[2j]
// at point (i + where[2j+1]) mod n (n a power of two): q(w X) on the coset g <w> is q's own codeword turned by one place.
__global__ void __launch_bounds__(256) mpoly_eval_kernel(const Fe* __restrict__ vals_m, uint32_t nvars, uint64_t n, const uint8_t* __restrict__ exps,
                                                        const Fe* __restrict__ coef_m, uint32_t nterms, Fe* __restrict__ out, const uint64_t* __restrict__ where) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fe acc{0, 0};
    for (uint32_t t = 0; t < nterms; ++t) {
        Fe p = coef_m[t];
        const uint8_t* e = exps + (size_t)t * nvars;
        for (uint32_t j = 0; j < nvars; ++j) {
            const uint32_t ej = e[j];
            if (ej == 0) continue;
            const Fe v = where ? vals_m[where[2 * j] * n + ((i + where[2 * j + 1]) & (n - 1))] : vals_m[(uint64_t)j * n + i];
            for (uint32_t k = 0; k < ej; ++k) p = mont_mul(p, v);
        }
        acc = fe_add(acc, p);
    }
    out[i] = from_mont(acc);
}
__global__ void __launch_bounds__(256) to_mont_kernel(Fe* __restrict__ a, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = to_mont(a[i]);
}

// diagnostics: elementwise field operations exactly as the kernels use them
__global__ void __launch_bounds__(256) field_selftest_kernel(int op, const Fe* __restrict__ a, const Fe* __restrict__ b, Fe* __restrict__ out, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fe x = a[i], y = b[i], r;
    switch (op) {
        case 0: r = mont_mul(x, y); break;                 // x * y * 2^-128
        case 1: r = fe_add(x, y); break;
        case 2: r = fe_sub(x, y); break;
        case 3: r = fe_mul(x, y); break;
        case 4: r = fe_half(x); break;
        case 5: r = from_mont(mont_inv(to_mont(x))); break;
        case 7: {                                           // the interleaved pair of products the butterflies use: both halves
            Fe r0, r1;                                      // must agree with each other (operand pairs swapped between the lanes' roles)
            mont_mul2(x, y, x, y, r0, r1);
            r = fe_eq(r0, r1) ? r0 : Fe{~0ull, ~0ull};
            break;
        }
        default: r = mont_mul_c(x, y); break;               // portable reference implementation
    }
    out[i] = r;
}

// Field.sample (code/algebra.py:116-120) of `count` byte strings of `width` <= 32 bytes each: the big-endian integer mod p.
// value = hi * 2^128 + lo with hi, lo < 2^128 < 2p: one conditional subtraction each, hi * 2^128 = to_mont(hi).
__global__ void __launch_bounds__(256) sample_bytes_kernel(const uint8_t* __restrict__ bytes, uint64_t count, uint32_t width, Fe* __restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint8_t* b = bytes + i * width;
    uint64_t w[4] = {0, 0, 0, 0};                      // little-endian 64-bit words of the integer
    for (uint32_t k = 0; k < width; ++k) {
        const uint32_t pos = width - 1 - k;            // byte k has weight 256^pos
        w[pos >> 3] |= (uint64_t)b[k] << (8 * (pos & 7));
    }
    Fe lo{w[0], w[1]}, hi{w[2], w[3]};
    if (fe_ge_p(lo)) lo = fe_sub(lo, Fe{P_LO, P_HI});
    if (fe_ge_p(hi)) hi = fe_sub(hi, Fe{P_LO, P_HI});
    out[i] = fe_add(lo, to_mont(hi));
}

__global__ void __launch_bounds__(256) gather_kernel(const Fe* __restrict__ v, const uint64_t* __restrict__ idx, uint64_t k, Fe* __restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < k) out[i] = v[idx[i]];
}

// ============================================================================ host state

namespace sci {

Ctx g;
std::mutex g_mu;
hipStream_t g_comm_stream_for_free = nullptr;   // the library's communication stream once it exists (sc_fourstep_run_dev)
std::future<void> g_rand_worker;                // a draw of kernel randomness started ahead of time (sc_urandom_prefetch) ...
size_t g_rand_prefetched = 0;                   // ... and its size in bytes (0: none in flight)

// Small caching allocator for the big short-lived device objects (vectors, Merkle trees): hipMalloc/hipFree of
// hundreds of MiB cost more than the kernels that fill them.  Exact-size free lists, bounded total.
std::multimap<size_t, void*> g_pool;
size_t g_pool_bytes = 0;
// What the pool may keep: a quarter of the device's memory (72 GB of the MI355X's 288: the machine's HBM is there to be used --
// a FastStark proof at a 2^24 FRI domain cycles through ~12 GB of trees and vectors, and a buffer that does not fit the pool costs
// a hipFree now and a multi-GB hipMalloc in the next proof, tens of milliseconds each); set at init, 8 GB if the device does not say.
size_t g_pool_cap = 8ull << 30;

void reap_pending(bool block);
hipError_t pool_alloc(void** p, size_t bytes) {
    reap_pending(false);
    auto it = g_pool.find(bytes);
    if (it != g_pool.end()) {
        *p = it->second;
        g_pool.erase(it);
        g_pool_bytes -= bytes;
        return hipSuccess;
    }
    const bool slow_log = getenv("STARKCORE_LOG_SLOW_ALLOC") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t e = hipMalloc(p, bytes);
    if (slow_log) {
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (ms > 1.0) fprintf(stderr, "starkcore: hipMalloc(%zu MB) took %.1f ms (pool %zu MB in %zu buffers)\n", bytes >> 20, ms, g_pool_bytes >> 20, g_pool.size());
    }
    if (e != hipSuccess) {                             // out of memory: first what is parked behind events
        (void)hipGetLastError();
        reap_pending(true);
        auto it2 = g_pool.find(bytes);
        if (it2 != g_pool.end()) {
            *p = it2->second;
            g_pool.erase(it2);
            g_pool_bytes -= bytes;
            return hipSuccess;
        }
        e = hipMalloc(p, bytes);
    }
    if (e != hipSuccess && !g_pool.empty()) {          // still out of memory: drop the cache and retry
        (void)hipDeviceSynchronize();
        for (auto& kv : g_pool) (void)hipFree(kv.second);
        g_pool.clear();
        g_pool_bytes = 0;
        (void)hipGetLastError();
        e = hipMalloc(p, bytes);
    }
    return e;
}

void pool_free(void* p, size_t bytes) {
    if (!p) return;
    if (g_pool_bytes + bytes <= g_pool_cap) {          // (small buffers too: hipFree waits for the whole device)
        g_pool.emplace(bytes, p);
        g_pool_bytes += bytes;
    } else {
        const auto t0 = std::chrono::steady_clock::now();
        (void)hipFree(p);
        if (getenv("STARKCORE_LOG_SLOW_ALLOC")) {
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (ms > 1.0) fprintf(stderr, "starkcore: hipFree(%zu MB) took %.1f ms (pool at its cap of %zu MB)\n", bytes >> 20, ms, g_pool_cap >> 20);
        }
    }
}

void pool_clear() {
    for (auto& kv : g_pool) (void)hipFree(kv.second);
    g_pool.clear();
    g_pool_bytes = 0;
}

std::deque<PendingFree> g_pending;
std::vector<hipEvent_t> g_event_pool;

hipEvent_t event_get() {
    if (!g_event_pool.empty()) { hipEvent_t e = g_event_pool.back(); g_event_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return e;
}

// block: wait for the events instead of asking (out of memory, shutdown)
void reap_pending(bool block) {
    for (size_t i = 0; i < g_pending.size();) {
        PendingFree& f = g_pending[i];
        bool done = true;
        for (hipEvent_t e : f.evs) {
            hipError_t q = block ? hipEventSynchronize(e) : hipEventQuery(e);
            if (q == hipErrorNotReady) { (void)hipGetLastError(); done = false; break; }
            if (q != hipSuccess) (void)hipGetLastError();          // a failed event cannot hold the buffer for ever
        }
        if (!done) { ++i; continue; }
        for (hipEvent_t e : f.evs) g_event_pool.push_back(e);
        pool_free(f.p, f.bytes);
        g_pending[i] = std::move(g_pending.back());
        g_pending.pop_back();
    }
}

int fail(int code, const std::string& msg) {
    g.err = msg;
    return code;
}

int ensure_init() {
    if (g.init) {
        // HIP's current device is per host thread and starts at 0: a thread other than the one that initialised the library (a
        // prover thread of a rank that owns device k != 0) is bound to the library's device on its first call
        static thread_local int bound = -1;            // (the device, not a flag: sc_shutdown + sc_init may move the library)
        if (bound != g.device) { HIPCHK(hipSetDevice(g.device)); bound = g.device; }
        return SC_OK;
    }
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(SC_ERR_HIP, std::string("no HIP device available: ") + hipGetErrorString(e));
    int dev = 0;
    if (const char* lr = getenv("LOCAL_RANK")) dev = atoi(lr) % n;
    if (const char* sd = getenv("STARKCORE_DEVICE")) dev = atoi(sd) % n;
    HIPCHK(hipSetDevice(dev));
    HIPCHK(hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking));
    { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) g.num_cus = cus; }
    { size_t free_b = 0, total_b = 0; if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b) g_pool_cap = total_b / 4; else (void)hipGetLastError(); }
    // several processes on one device (torchrun with more ranks than GPUs: the shared-GPU functional runs) share that quarter;
    // STARKCORE_POOL_CAP_MB sets the cap outright (0: nothing is kept), sc_set_tuning("pool_cap_mb" / "pool_trim") at run time
    if (const char* lws = getenv("LOCAL_WORLD_SIZE")) { const int per_dev = (atoi(lws) + n - 1) / n; if (per_dev > 1) g_pool_cap /= (size_t)per_dev; }
    if (const char* cap = getenv("STARKCORE_POOL_CAP_MB")) g_pool_cap = (size_t)atoll(cap) << 20;
    g.device = dev;
    g.init = true;
    return SC_OK;
}

int scratch(int slot, size_t bytes, void** out) {
    DevBuf& b = g.scratch[slot];
    if (b.bytes < bytes) {
        if (b.p) { HIPCHK(hipDeviceSynchronize()); HIPCHK(hipFree(b.p)); b.p = nullptr; b.bytes = 0; }
        size_t want = bytes < (1u << 20) ? (1u << 20) : bytes;
        HIPCHK(hipMalloc(&b.p, want));
        b.bytes = want;
    }
    *out = b.p;
    return SC_OK;
}

// The intermediate vector of a multi-pass transform.  One per STREAM: calls on one stream reuse it in stream order (as the shared
// scratch slot did), calls on different streams -- two independent columns transformed side by side, tools/two_stream_ntt.py -- each
// have their own, so that sc_ntt_dev / sc_coset_evaluate_dev / sc_coset_divide_dev's transforms may be in flight on several streams at once.
int ntt_work_buffer(hipStream_t st, size_t bytes, void** out) {
    DevBuf& b = g.ntt_work[st];
    if (b.bytes < bytes) {
        if (b.p) { HIPCHK(hipDeviceSynchronize()); HIPCHK(hipFree(b.p)); b.p = nullptr; b.bytes = 0; }
        size_t want = bytes < (1u << 20) ? (1u << 20) : bytes;
        HIPCHK(hipMalloc(&b.p, want));
        b.bytes = want;
    }
    *out = b.p;
    return SC_OK;
}

// host check of ntt.py:10-11
int check_root(Fe root, uint64_t n) {
    if (fe_ge_p(root)) return fail(SC_ERR_BAD_ARG, "root is not a canonical residue");
    Fe rm = to_mont(root);
    Fe one = fe_mont_one();
    Fe half = mont_pow(rm, n / 2);
    Fe full = mont_mul(half, half);
    if (!fe_eq(full, one)) return fail(SC_ERR_ROOT_ORDER, "primitive root must be nth root of unity, where n is len(values)");
    if (fe_eq(half, one)) return fail(SC_ERR_ROOT_NOT_PRIMITIVE, "primitive root is not primitive nth root of unity, where n is len(values)");
    return SC_OK;
}

int build_pow_table(Fe** out, uint64_t count, Fe base_m, uint64_t step, Fe scale_m, hipStream_t st) {
    if (count == 0) count = 1;
    HIPCHK(hipMalloc((void**)out, count * sizeof(Fe)));
    unsigned blocks = (unsigned)((count + 255) / 256);
    hipLaunchKernelGGL(pow_table_kernel, dim3(blocks), dim3(256), 0, st, *out, count, base_m, step, scale_m);
    HIPCHK(hipGetLastError());
    return SC_OK;
}

void free_plan_tables(PlanTables& t) {
    hipFree(t.mt); hipFree(t.tl); hipFree(t.th);
    if (t.th_ninv) hipFree(t.th_ninv);
    for (int v = 0; v < 2; ++v) for (int i = 0; i < 4; ++i) if (t.twd[v][i]) hipFree(t.twd[v][i]);
    if (t.twd_b) hipFree(t.twd_b);
}

void free_plans() {
    for (auto& kv : g.plans) free_plan_tables(kv.second);
    g.plans.clear();
    for (auto& kv : g.pows) { hipFree(kv.second.lo); hipFree(kv.second.hi); }
    g.pows.clear();
    for (auto& kv : g.outers) hipFree(kv.second.d);
    g.outers.clear();
}

// Cache eviction, least recently used first and NEVER an entry looked up recently: one API call makes at most a handful of
// table lookups and keeps raw pointers to what it got (NttOpts::coset, PlanTables*), so the PIN_WINDOW most recent lookups
// are off limits -- an eviction in the middle of a call cannot free what the call still uses.  std::map nodes are stable, so
// erasing other entries leaves the kept pointers valid.  One device sync per batch, not per entry.
constexpr uint64_t PIN_WINDOW = 64;
constexpr size_t PLAN_CAP = 256, POW_CAP = 64;

template <class Map, class FreeFn>
int evict_tables(Map& m, size_t cap, FreeFn free_entry) {
    if (m.size() < cap) return SC_OK;
    std::vector<std::pair<uint64_t, typename Map::iterator>> old;
    for (auto it = m.begin(); it != m.end(); ++it)
        if (it->second.last_use + PIN_WINDOW < g.tick) old.emplace_back(it->second.last_use, it);
    if (old.empty()) return SC_OK;                       // everything is in recent use: let the cache grow
    std::sort(old.begin(), old.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
    size_t drop = cap / 4 ? cap / 4 : 1;
    if (drop > old.size()) drop = old.size();
    HIPCHK(hipDeviceSynchronize());
    for (size_t i = 0; i < drop; ++i) { free_entry(old[i].second->second); m.erase(old[i].second); }
    return SC_OK;
}

// tables for a primitive n-th root (Montgomery form entries)
int get_plan(Fe root, int logn, bool need_ninv, hipStream_t st, PlanTables** out) {
    PlanKey key{logn, root.lo, root.hi};
    auto it = g.plans.find(key);
    bool built = false;
    ++g.tick;
    if (it == g.plans.end()) {
        SCCHK(evict_tables(g.plans, PLAN_CAP, free_plan_tables));
        const uint64_t n = 1ull << logn;
        Fe rm = to_mont(root);
        PlanTables t;
        t.mt_log = logn < 12 ? logn : 12;
        SCCHK(build_pow_table(&t.mt, 1ull << (t.mt_log - 1), rm, n >> t.mt_log, fe_mont_one(), st));
        SCCHK(build_pow_table(&t.tl, n < 4096 ? n : 4096, rm, 1, fe_mont_one(), st));
        SCCHK(build_pow_table(&t.th, n > 4096 ? n >> 12 : 1, rm, 4096, fe_mont_one(), st));
        it = g.plans.emplace(key, t).first;
        built = true;
    }
    if (need_ninv && !it->second.th_ninv) {
        const uint64_t n = 1ull << logn;
        Fe ninv_m = mont_inv(to_mont(Fe{n, 0}));
        SCCHK(build_pow_table(&it->second.th_ninv, n > 4096 ? n >> 12 : 1, to_mont(root), 4096, ninv_m, st));
        built = true;
    }
    if (built) HIPCHK(hipStreamSynchronize(st));   // tables are shared across streams afterwards
    it->second.last_use = g.tick;
    *out = &it->second;
    return SC_OK;
}

// two-level power tables base^i, i < count
int get_pow(Fe base, uint64_t count, hipStream_t st, PowTables** out) {
    uint64_t hi_count = (count >> 12) + 1;
    // round up so that nearby sizes share a table
    uint64_t hc = 1; while (hc < hi_count) hc <<= 1;
    PowKey key{base.lo, base.hi, hc};
    auto it = g.pows.find(key);
    ++g.tick;
    if (it == g.pows.end()) {
        SCCHK(evict_tables(g.pows, POW_CAP, [](PowTables& t) { hipFree(t.lo); hipFree(t.hi); }));
        Fe bm = to_mont(base);
        PowTables t;
        SCCHK(build_pow_table(&t.lo, 4096, bm, 1, fe_mont_one(), st));
        SCCHK(build_pow_table(&t.hi, hc, bm, 4096, fe_mont_one(), st));
        HIPCHK(hipStreamSynchronize(st));
        it = g.pows.emplace(key, t).first;
    }
    it->second.last_use = g.tick;
    *out = &it->second;
    return SC_OK;
}

// plan a batched transform; two-pass plans get the direct inter-pass twiddle table (built once per (root, length, split))
int plan_batched_direct(NttPlanDesc& d, BatchKind kind, int loglen, int logbatch, PlanTables* pt, const Fe* in, Fe* work, Fe* out, BatchExtras ex, hipStream_t st, bool* ok) {
    NttTables tb;
    tb.mt = pt->mt; tb.mt_log = pt->mt_log; tb.tl = pt->tl; tb.th = pt->th;
    *ok = plan_batched(d, kind, loglen, logbatch, tb, in, work, out, g.tuning, ex);
    if (!*ok || d.npasses != 2 || loglen > g.tuning.direct_tw_max_log || g.tuning.direct_tw_max_log <= 0) return SC_OK;
    if (pt->twd_b && pt->twd_b_digit0 != d.digits[0]) {
        HIPCHK(hipDeviceSynchronize());
        hipFree(pt->twd_b);
        pt->twd_b = nullptr;
    }
    if (!pt->twd_b) {
        const uint64_t count = 1ull << loglen;
        HIPCHK(hipMalloc((void**)&pt->twd_b, count * sizeof(Fe)));
        hipLaunchKernelGGL(twiddle_table_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, pt->twd_b, loglen - d.digits[0], count, (uint64_t)1, pt->tl, pt->th);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(st));
        pt->twd_b_digit0 = d.digits[0];
    }
    ex.inner_twd = pt->twd_b;
    *ok = plan_batched(d, kind, loglen, logbatch, tb, in, work, out, g.tuning, ex);
    return SC_OK;
}

template <int LOGE>
void launch_pass(const NttPassDesc& pd, hipStream_t st) {
    int remap = (g.xcd_remap && pd.ntiles >= 16 && (pd.ntiles & 7u) == 0) ? 1 : 0;
    if constexpr (LOGE == 2) {
        // hot shapes of the default plans get geometry-specialised instantiations
        if (g.fixed_shapes) {
            const int lr = pd.p.logR, lc = pd.p.logC;
#define SC_LAUNCH_FIXED(LR, LC, TR, ALT) \
    hipLaunchKernelGGL((ntt_pass_kernel_fixed<2, LR, LC, TR, ALT>), dim3(pd.ntiles * pd.cols), dim3(pd.threads), pd.lds_bytes, st, pd.p, pd.ntiles, remap, g.wave_local)
            // (a launch with a second destination -- the column stage of the sharded transform -- has its own instantiation; it
            // is never traced: the generic kernel serves that combination)
#define SC_FIXED(LR, LC)                                                          \
            if (lr == LR && lc == LC && !(pd.p.trace && pd.p.blk_enable)) {          \
                if (pd.p.trace) SC_LAUNCH_FIXED(LR, LC, true, false);             \
                else if (pd.p.blk_enable) SC_LAUNCH_FIXED(LR, LC, false, true);      \
                else SC_LAUNCH_FIXED(LR, LC, false, false);                       \
                return;                                                           \
            }
            SC_FIXED(8, 3) SC_FIXED(7, 4) SC_FIXED(10, 2) SC_FIXED(6, 5) SC_FIXED(9, 3) SC_FIXED(8, 4)
#undef SC_FIXED
        }
    }
    if constexpr (LOGE == 3) {
        if (g.fixed_shapes && !pd.p.trace && !pd.p.blk_enable) {
            const int lr = pd.p.logR, lc = pd.p.logC;
#define SC_FIXED8(LR, LC) if (lr == LR && lc == LC) { hipLaunchKernelGGL((ntt_pass_kernel_fixed8<LR, LC>), dim3(pd.ntiles * pd.cols), dim3(pd.threads), pd.lds_bytes, st, pd.p, pd.ntiles, remap, g.wave_local); return; }
            SC_FIXED8(10, 2) SC_FIXED8(9, 3) SC_FIXED8(8, 4)
#undef SC_FIXED8
        }
    }
    hipLaunchKernelGGL(ntt_pass_kernel<LOGE>, dim3(pd.ntiles * pd.cols), dim3(pd.threads), pd.lds_bytes, st, pd.p, pd.ntiles, remap);
}

int run_plan(NttPlanDesc& d, hipStream_t st) {
    size_t trace_off = 0;    // diagnostics: pass i writes its stamps behind those of the passes before it
    for (int i = 0; i < d.npasses; ++i) {
        const uint32_t grid = d.pass[i].ntiles * d.pass[i].cols;
        d.pass[i].p.prio_balance = g.prio_balance >= 0 ? g.prio_balance
                                 : grid <= (uint32_t)g.num_cus ? 1
                                 : grid >= ((d.pass[i].p.logR == 10 && d.pass[i].p.logC == 2) ? 2u : 8u) * (uint32_t)g.num_cus ? 2 : 0;
        d.pass[i].p.trace = g.trace ? g.trace + trace_off : nullptr;
        trace_off += (size_t)d.pass[i].ntiles * d.pass[i].cols * (d.pass[i].threads >> 6) * TRACE_STAMPS;
        switch (d.pass[i].loge) {
            case 1: launch_pass<1>(d.pass[i], st); break;
            case 2: launch_pass<2>(d.pass[i], st); break;
            case 3: launch_pass<3>(d.pass[i], st); break;
            case 4: launch_pass<4>(d.pass[i], st); break;
            default: return fail(SC_ERR_UNSUPPORTED, "bad loge");
        }
        HIPCHK(hipGetLastError());
    }
    return SC_OK;
}

// core transform on device pointers; root already validated.  forward: out = NTT_root(in); inverse handled by caller
// passing root^-1 and inverse=true (adds the n^-1 scaling).
int ntt_device(const Fe* d_in, Fe* d_out, int logn, Fe root, bool inverse_scale, const NttOpts& o, hipStream_t st) {
    PlanTables* pt;
    SCCHK(get_plan(root, logn, inverse_scale, st, &pt));
    const uint64_t n = 1ull << logn;
    const int m = plan_num_passes(logn, g.tuning);
    NttTables tb;
    tb.mt = pt->mt; tb.mt_log = pt->mt_log; tb.tl = pt->tl; tb.th = pt->th;
    tb.th_scaled = (inverse_scale && m > 1) ? pt->th_ninv : nullptr;
    NttIo io;
    io.in = d_in; io.out = d_out; io.in_limit = o.in_limit; io.cols = o.cols; io.col_stride_in = o.col_stride_in;
    if (m > 1) { void* w; SCCHK(ntt_work_buffer(st, n * o.cols * sizeof(Fe), &w)); io.work = (Fe*)w; }
    if (o.coset) { io.ol = o.coset->lo; io.oh = o.coset->hi; }
    if (inverse_scale && m == 1) { io.scale_last = true; io.scale = mont_inv(to_mont(Fe{n, 0})); }
    NttPlanDesc d;
    if (!plan_ntt(d, logn, tb, io, g.tuning)) return fail(SC_ERR_UNSUPPORTED, "unsupported transform length");
    if (d.npasses > 1 && g.tuning.direct_tw_max_log > 0) {   // tables bigger than the cap fall back to the two-level lookup
        // direct twiddle tables (one coalesced load + one modmul per element instead of two loads + two modmuls);
        // keyed by the digit split, rebuilt if the tuning changed it
        bool same = pt->twd_passes == d.npasses;
        for (int i = 0; same && i < d.npasses; ++i) same = pt->twd_digits[i] == d.digits[i];
        if (!same) {
            HIPCHK(hipDeviceSynchronize());
            for (int v = 0; v < 2; ++v) for (int i = 0; i < 4; ++i) if (pt->twd[v][i]) { hipFree(pt->twd[v][i]); pt->twd[v][i] = nullptr; }
            pt->twd_passes = d.npasses;
            for (int i = 0; i < 4; ++i) pt->twd_digits[i] = (i < d.npasses) ? d.digits[i] : 0;
        }
        const int variant = inverse_scale ? 1 : 0;
        bool built = false;
        int logA = 0;
        for (int i = 0; i + 1 < d.npasses; ++i) {
            const int logR = d.digits[i], logB = logn - logA - logR;
            const int logcount = logR + logB;
            const bool scaled = (variant == 1 && i == 0);
            Fe*& slot = pt->twd[scaled ? 1 : 0][i];
            if (logcount <= g.tuning.direct_tw_max_log) {
                if (!slot) {
                    const uint64_t count = 1ull << logcount;
                    HIPCHK(hipMalloc((void**)&slot, count * sizeof(Fe)));
                    hipLaunchKernelGGL(twiddle_table_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, slot, logB, count, 1ull << logA,
                                       pt->tl, scaled ? pt->th_ninv : pt->th);
                    HIPCHK(hipGetLastError());
                    built = true;
                }
                tb.twd[i] = slot;
            }
            logA += logR;
        }
        if (built) HIPCHK(hipStreamSynchronize(st));
        if (!plan_ntt(d, logn, tb, io, g.tuning)) return fail(SC_ERR_UNSUPPORTED, "unsupported transform length");
    }
    return run_plan(d, st);
}

Fe root_inverse(Fe root, uint64_t n) {   // root^-1 = root^(n-1) for an n-th root of unity
    return from_mont(mont_pow(to_mont(root), n - 1));
}

int ntt_any(const Fe* d_in, Fe* d_out, uint64_t n, Fe root, bool inverse, const NttOpts& o, hipStream_t st) {
    if (n <= 1) {
        if (n == 1 && d_in != d_out) HIPCHK(hipMemcpyAsync(d_out, d_in, sizeof(Fe), hipMemcpyDeviceToDevice, st));
        return SC_OK;
    }
    if (!is_pow2(n)) return fail(SC_ERR_NOT_POW2, "cannot compute ntt of non-power-of-two sequence");
    SCCHK(check_root(root, n));
    return ntt_device(d_in, d_out, ilog2(n), inverse ? root_inverse(root, n) : root, inverse, o, st);
}

int upload(void* d, const void* h, size_t bytes, hipStream_t st) {
    if (bytes) HIPCHK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st));
    return SC_OK;
}
int download(void* h, const void* d, size_t bytes, hipStream_t st) {
    if (bytes) HIPCHK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return SC_OK;
}

// the division only enqueued: *flag_dev (a word of this stream's scratch, valid until the next division on the stream clears it)
// becomes non-zero if a divisor is zero
int pointwise_div_enqueue(const Fe* a, const Fe* b, Fe* out, uint64_t n, hipStream_t st, uint32_t** flag_dev) {
    void* fl;
    SCCHK(scratch(4, 256, &fl));
    HIPCHK(hipMemsetAsync(fl, 0, 8, st));
    static const int K = [] { const char* e = getenv("STARKCORE_DIV_K"); return e && atoi(e) == 8 ? 8 : 16; }();      // (A/B knob)
    const uint64_t threads = (n + K - 1) / K;
    const unsigned blocks = (unsigned)((threads + 255) / 256);
    if (K == 8) hipLaunchKernelGGL(pointwise_div_kernel<8>, dim3(blocks), dim3(256), 0, st, a, b, out, n, (uint32_t*)fl);
    else hipLaunchKernelGGL(pointwise_div_kernel<16>, dim3(blocks), dim3(256), 0, st, a, b, out, n, (uint32_t*)fl);
    HIPCHK(hipGetLastError());
    *flag_dev = (uint32_t*)fl;
    return SC_OK;
}
int pointwise_div_device(const Fe* a, const Fe* b, Fe* out, uint64_t n, hipStream_t st) {
    uint32_t* fl;
    SCCHK(pointwise_div_enqueue(a, b, out, n, st, &fl));
    uint64_t hflag = 0;
    SCCHK(read_small_polled(fl, 8, st, &hflag));                  // (the flag is the low 32 bits of a word of the scratch buffer)
    if ((uint32_t)hflag) return fail(SC_ERR_DIV_ZERO, "divide by zero");
    return SC_OK;
}

// A few words of device memory to the host WITHOUT a copy engine and without putting the thread to sleep: a one-wave kernel behind
// whatever produced them writes them to a pinned slot, then -- ordered behind them -- a sequence number the host polls (the scheme
// the Merkle roots travel by).  A degree, an exactness flag, a divide-by-zero flag: a proof reads two dozen of them, and a
// pageable hipMemcpyAsync + hipStreamSynchronize is 15-25 us each where this is 2-3.  Falls back to the copy when no slot is free.
__global__ void __launch_bounds__(64) words_publish_kernel(const uint64_t* __restrict__ src, uint32_t nwords, volatile uint64_t* host, uint64_t seq) {
    if (threadIdx.x < nwords) host[threadIdx.x] = src[threadIdx.x];
    __threadfence_system();
    if (threadIdx.x == 0) host[8] = seq;
}
int read_small_polled(const void* d_src, size_t bytes, hipStream_t st, void* host_out) {
    const int slot = (bytes <= 64 && ((uintptr_t)d_src & 7) == 0) ? root_slot_get() : -1;
    if (slot < 0) {
        HIPCHK(hipMemcpyAsync(host_out, d_src, bytes, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        return SC_OK;
    }
    const uint64_t seq = ++g.root_seq;
    volatile uint64_t* host = (volatile uint64_t*)(g.root_slots + ROOT_SLOT_BYTES * slot);
    hipLaunchKernelGGL(words_publish_kernel, dim3(1), dim3(64), 0, st, (const uint64_t*)d_src, (uint32_t)((bytes + 7) / 8), host, seq);
    hipError_t e = hipGetLastError();
    bool landed = false;
    if (e == hipSuccess) {
        for (long spin = 0; spin < SPIN_POLLS; ++spin) {
            if (__atomic_load_n(host + 8, __ATOMIC_ACQUIRE) == seq) { landed = true; break; }
            if ((spin & 4095) == 4095) {
                e = hipStreamQuery(st);
                if (e != hipErrorNotReady) break;              // finished (the number is there now) or failed
                (void)hipGetLastError();
                e = hipSuccess;
            }
        }
        if (!landed) {
            (void)hipGetLastError();
            e = hipStreamSynchronize(st);
            landed = e == hipSuccess && __atomic_load_n(host + 8, __ATOMIC_ACQUIRE) == seq;
            if (e == hipSuccess && !landed) e = hipErrorUnknown;
        }
    }
    if (landed) memcpy(host_out, (const void*)host, bytes);
    else (void)hipStreamSynchronize(st);                       // nothing may still write to the slot when it is reused
    g.free_root_slots.push_back(slot);
    if (!landed) return fail(SC_ERR_HIP, hipGetErrorString(e));
    return SC_OK;
}

// ---- checks that are read LATER (sc_later_t).  A division's "divide by zero" flag and the exactness of a coset division are only
// ever asserted on; a prover that waits for each of them where the reference raises leaves the GPU idle a dozen times per proof
// (tools/sync_points.py).  The flags travel like the roots of asynchronously built trees -- a one-wave kernel behind the work writes
// them to a pinned slot, then a sequence number -- and the caller collects them where it has to wait anyway.
//   words[0] != 0 : a divisor value was zero      words[1] : highest index of a non-zero coefficient above the quotient, -1 if none
__global__ void __launch_bounds__(64) divide_flags_publish_kernel(const uint32_t* __restrict__ zero_flag, const long long* __restrict__ degree_slots,
                                                                 volatile uint64_t* host, uint64_t seq) {
    if (threadIdx.x == 0) {
        long long m = -1;
        if (degree_slots) for (int i = 0; i < DEGREE_SLOTS; ++i) m = degree_slots[i] > m ? degree_slots[i] : m;
        host[0] = zero_flag ? (uint64_t)*zero_flag : 0ull;
        host[1] = (uint64_t)m;
        __threadfence_system();
        host[8] = seq;
    }
}
int divide_flags_later(const uint32_t* zero_flag, const long long* degree_slots, hipStream_t st, sc_later** out) {
    const int slot = root_slot_get();
    if (slot < 0) return fail(SC_ERR_UNSUPPORTED, "no pinned slot free for a deferred check");
    sc_later* h = new sc_later{slot, ++g.root_seq, st};
    volatile uint64_t* host = (volatile uint64_t*)(g.root_slots + ROOT_SLOT_BYTES * slot);
    hipLaunchKernelGGL(divide_flags_publish_kernel, dim3(1), dim3(64), 0, st, zero_flag, degree_slots, host, h->seq);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g.free_root_slots.push_back(slot); delete h; return fail(SC_ERR_HIP, hipGetErrorString(e)); }
    *out = h;
    return SC_OK;
}
int later_wait(sc_later* h, uint64_t words[8]) {
    volatile uint64_t* host = (volatile uint64_t*)(g.root_slots + ROOT_SLOT_BYTES * h->slot);
    bool landed = false;
    hipError_t e = hipSuccess;
    for (long spin = 0; spin < SPIN_POLLS; ++spin) {
        if (__atomic_load_n(host + 8, __ATOMIC_ACQUIRE) == h->seq) { landed = true; break; }
        if ((spin & 4095) == 4095) {
            e = hipStreamQuery(h->st);
            if (e != hipErrorNotReady) break;
            (void)hipGetLastError();
            e = hipSuccess;
        }
    }
    if (!landed) {
        (void)hipGetLastError();
        e = hipStreamSynchronize(h->st);
        landed = e == hipSuccess && __atomic_load_n(host + 8, __ATOMIC_ACQUIRE) == h->seq;
        if (e == hipSuccess && !landed) e = hipErrorUnknown;
    }
    if (landed) memcpy(words, (const void*)host, 64);
    g.free_root_slots.push_back(h->slot);
    delete h;
    if (!landed) return fail(SC_ERR_HIP, hipGetErrorString(e));
    return SC_OK;
}

int gather_device(const Fe* v, const uint64_t* d_idx, uint64_t k, Fe* d_out, hipStream_t st) {
    hipLaunchKernelGGL(gather_kernel, dim3((unsigned)((k + 255) / 256)), dim3(256), 0, st, v, d_idx, k, d_out);
    HIPCHK(hipGetLastError());
    return SC_OK;
}

int evict_outer_tables() { return evict_tables(g.outers, (size_t)16, [](OuterTable& t) { hipFree(t.d); }); }

}  // namespace sci

// ============================================================================ C ABI

extern "C" {

int sc_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int sc_init(int device) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g.init && (device < 0 || device == g.device)) return SC_OK;
    if (g.init) return fail(SC_ERR_BAD_ARG, "already initialised on another device");
    if (device >= 0) { char buf[16]; snprintf(buf, sizeof buf, "%d", device); setenv("STARKCORE_DEVICE", buf, 1); }
    return ensure_init();
}

int sc_shutdown(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_rand_worker.valid()) { g_rand_worker.get(); g_rand_prefetched = 0; }
    if (!g.init) return SC_OK;
    hipDeviceSynchronize();
    reap_pending(true);
    for (hipEvent_t e : g_event_pool) (void)hipEventDestroy(e);
    g_event_pool.clear();
    free_plans();
    pool_clear();
    for (auto& b : g.scratch) { if (b.p) hipFree(b.p); b = DevBuf{}; }
    for (auto& kv : g.ntt_work) if (kv.second.p) hipFree(kv.second.p);
    g.ntt_work.clear();
    if (g.stream) hipStreamDestroy(g.stream);
    g.stream = nullptr;
    g.seen_streams.clear();
    g.foreign_streams = false;
    g.init = false;        // (the pinned root slots stay: a tree built asynchronously may still be freed after this)
    return SC_OK;
}

const char* sc_last_error(void) { return g.err.c_str(); }

int sc_synchronize(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    HIPCHK(hipStreamSynchronize(g.stream));
    return SC_OK;
}

// The library's own stream as a raw hipStream_t: a caller that runs its other device work (torch tensors, collectives) on THIS
// stream -- torch.cuda.ExternalStream(sc_stream()) -- needs no ordering with the library at all.
int sc_stream(void** stream_out) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    if (!stream_out) return fail(SC_ERR_BAD_ARG, "null argument");
    *stream_out = (void*)g.stream;
    return SC_OK;
}

// Order the library stream and another stream with each other WITHOUT blocking the host: everything enqueued so far on either
// is finished before anything enqueued later on the other starts (two events, two stream waits).
int sc_stream_join(void* other) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    hipStream_t o = (hipStream_t)other;
    if (o == g.stream) return SC_OK;
    (void)pick_stream(other);                              // frees must respect this stream from now on
    hipEvent_t a = event_get(), b = event_get();
    if (!a || !b) { if (a) g_event_pool.push_back(a); if (b) g_event_pool.push_back(b); HIPCHK(hipStreamSynchronize(o)); HIPCHK(hipStreamSynchronize(g.stream)); return SC_OK; }
    hipError_t e = hipEventRecord(a, o);
    if (e == hipSuccess) e = hipStreamWaitEvent(g.stream, a, 0);
    if (e == hipSuccess) e = hipEventRecord(b, g.stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(o, b, 0);
    g_event_pool.push_back(a);                             // a recorded event may be re-recorded once the waits are enqueued
    g_event_pool.push_back(b);
    if (e != hipSuccess) return fail(SC_ERR_HIP, hipGetErrorString(e));
    return SC_OK;
}

int sc_set_tuning(const char* key, int value) {
    std::lock_guard<std::mutex> lk(g_mu);
    std::string k(key ? key : "");
    if (k == "max_tile_log") g.tuning.max_tile_log = value;
    else if (k == "loge") g.tuning.loge = value;
    else if (k == "max_col_log") g.tuning.max_col_log = value;
    else if (k == "min_tiles_log") g.tuning.min_tiles_log = value;
    else if (k == "single_pass_max_log") g.tuning.single_pass_max_log = value;
    else if (k == "max_digit_log") g.tuning.max_digit_log = value;
    else if (k == "direct_tw_max_log") g.tuning.direct_tw_max_log = value;
    else if (k == "xcd_remap") g.xcd_remap = value;
    else if (k == "fixed_shapes") g.fixed_shapes = value;
    else if (k == "wave_local") g.wave_local = value;
    else if (k == "prio_balance") g.prio_balance = value;
    else if (k == "loge_cols") g.tuning.loge_cols = value;
    else if (k == "tw_on_load") g.tuning.tw_on_load = value;
    else if (k == "prune") g.tuning.prune = value;
    else if (k == "merkle_big_nlev") g.merkle_big_nlev = value < 0 ? 0 : (value > 8 ? 8 : value);
    else if (k == "fri_tail") g.fri_tail = value ? 1 : 0;
    else if (k == "small_divisor_direct") g.small_divisor_direct = value ? 1 : 0;
    else if (k == "fri_tail_stall") g.fri_tail_stall = value;                               // tests only: see core.h
    else if (k == "pool_cap_mb") g_pool_cap = (size_t)(value < 0 ? 0 : value) << 20;       // what the free lists may keep from now on
    else if (k == "pool_trim") {                                                            // give everything in the free lists back to the device now
        if (g.init) { (void)hipDeviceSynchronize(); reap_pending(true); pool_clear(); }
    }
    else return fail(SC_ERR_BAD_ARG, "unknown tuning key " + k);
    return SC_OK;
}

int sc_ntt_num_passes(uint64_t n) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (n < 2 || !is_pow2(n)) return 0;
    return plan_num_passes(ilog2(n), g.tuning);
}

int sc_debug_trace(void* d_buf) {
    std::lock_guard<std::mutex> lk(g_mu);
    g.trace = (unsigned long long*)d_buf;
    return SC_OK;
}

// ---- vectors
int sc_vec_alloc(uint64_t n, sc_vec_t** out) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    sc_vec* v = new sc_vec{nullptr, n};
    hipError_t e = pool_alloc((void**)&v->d, (n ? n : 1) * sizeof(Fe));
    if (e != hipSuccess) { delete v; return fail(SC_ERR_HIP, hipGetErrorString(e)); }
    *out = v;
    return SC_OK;
}
int sc_vec_free(sc_vec_t* v) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!v) return SC_OK;
    if (v->owned) release_after_streams(v->d, (v->n ? v->n : 1) * sizeof(Fe));
    delete v;
    return SC_OK;
}
// a vector handle over device memory the CALLER owns (a torch tensor's storage): no copy in, no copy out; the memory must outlive
// the handle and every call enqueued with it
int sc_vec_wrap(void* d_elems, uint64_t n, sc_vec_t** out) {
    if (!d_elems || !out) return fail(SC_ERR_BAD_ARG, "null argument");
    sc_vec* v = new sc_vec{(Fe*)d_elems, n};
    v->owned = false;
    *out = v;
    return SC_OK;
}
uint64_t sc_vec_len(const sc_vec_t* v) { return v ? v->n : 0; }
void* sc_vec_ptr(sc_vec_t* v) { return v ? v->d : nullptr; }
int sc_vec_zero(sc_vec_t* v) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!v) return fail(SC_ERR_BAD_ARG, "null vector");
    SCCHK(ensure_init());
    if (v->n) HIPCHK(hipMemsetAsync(v->d, 0, v->n * sizeof(Fe), g.stream));
    return SC_OK;
}
// a few hundred bytes travel as a KERNEL ARGUMENT: the launch copies them, nothing is staged, nothing is waited for -- a pageable
// hipMemcpyAsync has to be followed by a wait for the stream (the host buffer is the caller's), and in the middle of a proof that
// wait is for everything the GPU still has queued (boundary zerofiers of three coefficients, 160 randomizer rows: 40 us each)
struct SmallPayload { uint64_t w[256]; };
__global__ void __launch_bounds__(256) small_upload_kernel(uint64_t* __restrict__ dst, const SmallPayload p, uint32_t nwords) {
    if (threadIdx.x < nwords) dst[threadIdx.x] = p.w[threadIdx.x];
}
// `bytes` (any number, at most 4 payloads) from host memory to 8-byte aligned device memory with room for the last word, as kernel arguments
static bool upload_small(void* d_dst, const void* host, size_t bytes, hipStream_t st) {
    if (bytes > 4 * sizeof(SmallPayload)) return false;
    const uint8_t* src = (const uint8_t*)host;
    uint64_t* dst = (uint64_t*)d_dst;
    for (size_t left = bytes; left;) {
        const size_t take = left < sizeof(SmallPayload) ? left : sizeof(SmallPayload);
        SmallPayload p;
        p.w[(take - 1) / 8] = 0;
        memcpy(p.w, src, take);
        hipLaunchKernelGGL(small_upload_kernel, dim3(1), dim3(256), 0, st, dst, p, (uint32_t)((take + 7) / 8));
        src += take; dst += (take + 7) / 8; left -= take;
    }
    return hipGetLastError() == hipSuccess;
}
int sc_vec_upload(sc_vec_t* v, uint64_t offset, const void* host, uint64_t count) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    if (!v || offset + count > v->n) return fail(SC_ERR_BAD_ARG, "upload out of range");
    if (count && upload_small(v->d + offset, host, count * sizeof(Fe), g.stream)) return SC_OK;
    SCCHK(upload(v->d + offset, host, count * sizeof(Fe), g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    return SC_OK;
}
int sc_vec_download(const sc_vec_t* v, uint64_t offset, void* host, uint64_t count) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    if (!v || offset + count > v->n) return fail(SC_ERR_BAD_ARG, "download out of range");
    return download(host, v->d + offset, count * sizeof(Fe), g.stream);
}
int sc_sample_bytes_dev(const void* bytes, uint64_t count, uint32_t width, void* d_out, void* stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    if (!count) return SC_OK;
    if (!bytes || !d_out || width == 0 || width > 32) return fail(SC_ERR_BAD_ARG, "byte strings of 1..32 bytes expected");
    hipStream_t st = pick_stream(stream);
    void* buf;
    SCCHK(scratch(6, count * width + 256, &buf));
    SCCHK(upload(buf, bytes, count * width, st));
    hipLaunchKernelGGL(sample_bytes_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, (const uint8_t*)buf, count, width, (Fe*)d_out);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));               // `bytes` is the caller's host memory
    return SC_OK;
}
// `count` draws of os.urandom(width) (code/fast_stark.py:116-117: one per coefficient of the randomizer polynomial) made by the
// library itself -- getrandom(2), which is what os.urandom calls -- and sampled into HBM (Field.sample, algebra.py:116-120).
// At a 2^24 FRI domain that is 36 MB of kernel randomness: 40-50 ms from one thread, the whole proof's budget; the kernel's
// generator is per-CPU, so the draw is split over threads writing into one pinned staging buffer (16 threads: 3 ms), which then
// goes to the device with one asynchronous copy.  Enqueued on `stream`; the staging buffer is reused by the next call, which
// first waits for this call's copy.
namespace {
uint8_t* g_rand_host = nullptr;
size_t g_rand_host_bytes = 0;
hipEvent_t g_rand_copied = nullptr;
std::atomic<int> g_rand_failed{0};

// the draw itself: `bytes` of getrandom into g_rand_host, split over host threads (at least 256 KiB each, at most 32)
void rand_fill(size_t bytes) {
    unsigned hw = std::thread::hardware_concurrency();
    size_t nthreads = bytes / (256u << 10);
    if (nthreads > 32) nthreads = 32;
    if (hw && nthreads > hw) nthreads = hw;
    static const long cap = [] { const char* e = getenv("STARKCORE_RAND_THREADS"); return e ? atol(e) : 0L; }();      // (hosts that give the process few cores)
    if (cap > 0 && nthreads > (size_t)cap) nthreads = (size_t)cap;
    if (nthreads < 1) nthreads = 1;
    auto fill = [](size_t a, size_t b) {
        while (a < b) {
            size_t want = b - a < (1u << 20) ? b - a : (1u << 20);
            ssize_t r = getrandom(g_rand_host + a, want, 0);
            if (r < 0) { if (errno == EINTR) continue; g_rand_failed = 1; return; }
            a += (size_t)r;
        }
    };
    if (nthreads == 1) { fill(0, bytes); return; }
    std::vector<std::thread> pool;
    const size_t per = (bytes + nthreads - 1) / nthreads;
    for (size_t i = 0; i < nthreads; ++i) {
        const size_t a = i * per, b = a + per < bytes ? a + per : bytes;
        if (a < b) pool.emplace_back(fill, a, b);
    }
    for (auto& t : pool) t.join();
}
// The device side of the staging: the drawn bytes' own device buffer (never the shared scratch: a prefetched copy is in flight
// while other calls run), a copy stream, and two events -- `copied`: the bytes have left the pinned buffer and are on the device;
// `consumed`: the sampling kernel that read them has finished.
uint8_t* g_rand_dev = nullptr;
size_t g_rand_dev_bytes = 0;
hipStream_t g_rand_stream = nullptr;
hipEvent_t g_rand_consumed = nullptr;

// both buffers are free (no draw running, the previous copy and the previous sampling kernel done) and hold at least `bytes`
int rand_buffer(size_t bytes) {
    if (g_rand_worker.valid()) { g_rand_worker.get(); g_rand_prefetched = 0; }
    if (g_rand_copied) HIPCHK(hipEventSynchronize(g_rand_copied));
    if (g_rand_consumed) HIPCHK(hipEventSynchronize(g_rand_consumed));
    if (g_rand_host_bytes < bytes) {
        if (g_rand_host) { (void)hipHostFree(g_rand_host); g_rand_host = nullptr; g_rand_host_bytes = 0; }
        HIPCHK(hipHostMalloc((void**)&g_rand_host, bytes, hipHostMallocDefault));
        g_rand_host_bytes = bytes;
    }
    if (g_rand_dev_bytes < bytes) {
        if (g_rand_dev) { (void)hipFree(g_rand_dev); g_rand_dev = nullptr; g_rand_dev_bytes = 0; }
        HIPCHK(hipMalloc((void**)&g_rand_dev, bytes + 256));
        g_rand_dev_bytes = bytes;
    }
    if (!g_rand_copied) HIPCHK(hipEventCreateWithFlags(&g_rand_copied, hipEventDisableTiming));
    if (!g_rand_consumed) HIPCHK(hipEventCreateWithFlags(&g_rand_consumed, hipEventDisableTiming));
    if (!g_rand_stream) HIPCHK(hipStreamCreateWithFlags(&g_rand_stream, hipStreamNonBlocking));
    return SC_OK;
}
// the prefetch worker: draw, then put the copy to the device on the copy stream (36 MB at a 2^24 FRI domain: 0.7 ms that the
// compute stream would otherwise sit through, because a stream is in-order)
void rand_fill_and_copy(size_t bytes, int device) {
    rand_fill(bytes);
    if (g_rand_failed) return;
    if (hipSetDevice(device) != hipSuccess || hipMemcpyAsync(g_rand_dev, g_rand_host, bytes, hipMemcpyHostToDevice, g_rand_stream) != hipSuccess ||
        hipEventRecord(g_rand_copied, g_rand_stream) != hipSuccess) {
        (void)hipGetLastError();
        g_rand_failed = 2;
    }
}
}
// Start the draws of a later sc_sample_urandom_dev(count, width, ...) NOW, on host threads, and return: the prover calls this at
// the top of a proof, and the 3 ms of kernel randomness for the randomizer polynomial (and their copy to the device) pass while
// the GPU interpolates the trace and commits to the boundary quotients.  (Only for the operating system's randomness, which has
// no order to keep.)
int sc_urandom_prefetch(uint64_t count, uint32_t width) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    if (!count || width == 0 || width > 32) return fail(SC_ERR_BAD_ARG, "byte strings of 1..32 bytes expected");
    const size_t bytes = (size_t)count * width;
    SCCHK(rand_buffer(bytes));
    g_rand_failed = 0;
    g_rand_prefetched = bytes;
    g_rand_worker = std::async(std::launch::async, rand_fill_and_copy, bytes, g.device);
    return SC_OK;
}
int sc_sample_urandom_dev(uint64_t count, uint32_t width, void* d_out, void* stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    if (!count) return SC_OK;
    if (!d_out || width == 0 || width > 32) return fail(SC_ERR_BAD_ARG, "byte strings of 1..32 bytes expected");
    hipStream_t st = pick_stream(stream);
    const size_t bytes = (size_t)count * width;
    if (g_rand_worker.valid() && g_rand_prefetched == bytes) {         // drawn (and copied) ahead of time: wait for the worker, order the stream behind the copy
        g_rand_worker.get();
        g_rand_prefetched = 0;
        if (g_rand_failed == 1) return fail(SC_ERR_HIP, "getrandom failed");
        if (g_rand_failed == 2) return fail(SC_ERR_HIP, "copy of the drawn bytes failed");
        HIPCHK(hipStreamWaitEvent(st, g_rand_copied, 0));
    } else {
        SCCHK(rand_buffer(bytes));
        g_rand_failed = 0;
        rand_fill(bytes);
        if (g_rand_failed) return fail(SC_ERR_HIP, "getrandom failed");
        HIPCHK(hipMemcpyAsync(g_rand_dev, g_rand_host, bytes, hipMemcpyHostToDevice, st));
        HIPCHK(hipEventRecord(g_rand_copied, st));
    }
    hipLaunchKernelGGL(sample_bytes_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, (const uint8_t*)g_rand_dev, count, width, (Fe*)d_out);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(g_rand_consumed, st));
    return SC_OK;
}

int sc_memcpy_dev(void* d_dst, const void* d_src, uint64_t count, void* stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    if (count && (!d_dst || !d_src)) return fail(SC_ERR_BAD_ARG, "null argument");
    if (count) HIPCHK(hipMemcpyAsync(d_dst, d_src, count * sizeof(Fe), hipMemcpyDeviceToDevice, pick_stream(stream)));
    return SC_OK;
}
int sc_vec_gather(const sc_vec_t* v, const uint64_t* indices, uint64_t k, void* host_out) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    if (!v) return fail(SC_ERR_BAD_ARG, "null vector");
    if (k == 0) return SC_OK;
    for (uint64_t i = 0; i < k; ++i) if (indices[i] >= v->n) return fail(SC_ERR_BAD_ARG, "gather index out of range");
    void* buf;
    const size_t idx_bytes = (k * 8 + 255) & ~255ull;
    SCCHK(scratch(4, 256 + idx_bytes + k * sizeof(Fe), &buf));
    uint64_t* d_idx = (uint64_t*)((char*)buf + 256);
    Fe* d_out = (Fe*)((char*)buf + 256 + idx_bytes);
    SCCHK(upload(d_idx, indices, k * 8, g.stream));
    hipLaunchKernelGGL(gather_kernel, dim3((unsigned)((k + 255) / 256)), dim3(256), 0, g.stream, v->d, d_idx, k, d_out);
    HIPCHK(hipGetLastError());
    return download(host_out, d_out, k * sizeof(Fe), g.stream);
}

// ---- diagnostics
int sc_field_selftest(int op, const void* a, const void* b, void* out, uint64_t n) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    if (!n) return SC_OK;
    void *da, *db, *dc;
    SCCHK(scratch(1, n * sizeof(Fe), &da));
    SCCHK(scratch(2, n * sizeof(Fe), &db));
    SCCHK(scratch(3, n * sizeof(Fe), &dc));
    SCCHK(upload(da, a, n * sizeof(Fe), g.stream));
    SCCHK(upload(db, b, n * sizeof(Fe), g.stream));
    hipLaunchKernelGGL(field_selftest_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, g.stream, op, (const Fe*)da, (const Fe*)db, (Fe*)dc, n);
    HIPCHK(hipGetLastError());
    return download(out, dc, n * sizeof(Fe), g.stream);
}

// ---- ntt
int sc_ntt_dev(const void* d_in, void* d_out, uint64_t n, const uint64_t root[2], int inverse, void* stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    return ntt_any((const Fe*)d_in, (Fe*)d_out, n, fe_from(root), inverse != 0, NttOpts{}, pick_stream(stream));
}

// `cols` independent transforms of length n, column c at element c * n, in ONE set of launches (NttIo::cols): the workgroups of one
// column start while those of another finish.  A set of launches covers at most COLS_ELEMS_PER_LAUNCH elements (64 columns of 2^20,
// 4 096 of 2^14: the grid of a set must be long whatever the length of a column), which bounds the intermediate vector at 1 GiB.
constexpr uint64_t COLS_ELEMS_PER_LAUNCH = 1ull << 26;
int sc_ntt_columns_dev(const void* d_in, void* d_out, uint64_t n, uint64_t cols, const uint64_t root[2], int inverse, void* stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    hipStream_t st = pick_stream(stream);
    if (cols == 0) return SC_OK;
    if (n > 1 && !is_pow2(n)) return fail(SC_ERR_NOT_POW2, "cannot compute ntt of non-power-of-two sequence");
    if (n <= 1) {
        if (n == 1 && d_in != d_out) HIPCHK(hipMemcpyAsync(d_out, d_in, cols * sizeof(Fe), hipMemcpyDeviceToDevice, st));
        return SC_OK;
    }
    const Fe* in = (const Fe*)d_in;
    Fe* out = (Fe*)d_out;
    uint64_t per = COLS_ELEMS_PER_LAUNCH / n;
    if (per < 1) per = 1;
    if (per > 65536) per = 65536;
    for (uint64_t done = 0; done < cols; done += per) {
        NttOpts o;
        o.cols = (uint32_t)(cols - done < per ? cols - done : per);
        SCCHK(ntt_any(in + done * n, out + done * n, n, fe_from(root), inverse != 0, o, st));
    }
    return SC_OK;
}

int sc_ntt(const void* in, void* out, uint64_t n, const uint64_t root[2], int inverse) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    if (n == 0) return SC_OK;
    if (!is_pow2(n)) return fail(SC_ERR_NOT_POW2, "cannot compute ntt of non-power-of-two sequence");
    void* a; void* b;
    SCCHK(scratch(1, n * sizeof(Fe), &a));
    SCCHK(scratch(2, n * sizeof(Fe), &b));
    SCCHK(upload(a, in, n * sizeof(Fe), g.stream));
    SCCHK(ntt_any((const Fe*)a, (Fe*)b, n, fe_from(root), inverse != 0, NttOpts{}, g.stream));
    return download(out, b, n * sizeof(Fe), g.stream);
}

// ---- coset evaluate
int sc_coset_evaluate_dev(const void* d_coeffs, uint64_t m, const uint64_t offset[2], const uint64_t generator[2], uint64_t order, void* d_out, void* stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    hipStream_t st = pick_stream(stream);
    if (m > order) return fail(SC_ERR_BAD_ARG, "more coefficients than the evaluation order");
    if (order <= 1) {
        // ntt returns its input unchanged for length <= 1 (ntt.py:5-6); coefficient 0 is scaled by offset^0 = 1
        if (order == 1) {
            if (m == 1) HIPCHK(hipMemcpyAsync(d_out, d_coeffs, sizeof(Fe), hipMemcpyDeviceToDevice, st));
            else HIPCHK(hipMemsetAsync(d_out, 0, sizeof(Fe), st));
        }
        return SC_OK;
    }
    if (!is_pow2(order)) return fail(SC_ERR_NOT_POW2, "cannot compute ntt of non-power-of-two sequence");
    Fe gen = fe_from(generator), off = fe_from(offset);
    SCCHK(check_root(gen, order));
    if (fe_ge_p(off)) return fail(SC_ERR_BAD_ARG, "offset is not a canonical residue");
    PowTables* pw;
    SCCHK(get_pow(off, m ? m : 1, st, &pw));
    NttOpts o;
    o.in_limit = m;
    o.coset = pw;
    return ntt_device((const Fe*)d_coeffs, (Fe*)d_out, ilog2(order), gen, false, o, st);
}

// the same for `cols` polynomials of m coefficients each (polynomial c at element c * m of d_coeffs, its values at c * order of d_out):
// zero padding and coset scaling apply per column, one set of launches covers COLS_ELEMS_PER_LAUNCH values
int sc_coset_evaluate_columns_dev(const void* d_coeffs, uint64_t m, uint64_t cols, const uint64_t offset[2], const uint64_t generator[2], uint64_t order, void* d_out, void* stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    hipStream_t st = pick_stream(stream);
    if (cols == 0) return SC_OK;
    if (m == 0 || m > order) return fail(SC_ERR_BAD_ARG, m ? "more coefficients than the evaluation order" : "a batch of empty polynomials");
    if (!is_pow2(order)) return fail(SC_ERR_NOT_POW2, "cannot compute ntt of non-power-of-two sequence");
    if (order < 2) return fail(SC_ERR_BAD_ARG, "evaluation order below 2");
    Fe gen = fe_from(generator), off = fe_from(offset);
    SCCHK(check_root(gen, order));
    if (fe_ge_p(off)) return fail(SC_ERR_BAD_ARG, "offset is not a canonical residue");
    PowTables* pw;
    SCCHK(get_pow(off, m, st, &pw));
    uint64_t per = COLS_ELEMS_PER_LAUNCH / order;
    if (per < 1) per = 1;
    if (per > 65536) per = 65536;
    for (uint64_t done = 0; done < cols; done += per) {
        NttOpts o;
        o.in_limit = m;
        o.coset = pw;
        o.cols = (uint32_t)(cols - done < per ? cols - done : per);
        o.col_stride_in = m;
        SCCHK(ntt_device((const Fe*)d_coeffs + done * m, (Fe*)d_out + done * order, ilog2(order), gen, false, o, st));
    }
    return SC_OK;
}

int sc_coset_evaluate(const void* coeffs, uint64_t m, const uint64_t offset[2], const uint64_t generator[2], uint64_t order, void* out) {
    void* a; void* b;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        SCCHK(ensure_init());
        if (order == 0) return SC_OK;
        if (m > order) return fail(SC_ERR_BAD_ARG, "more coefficients than the evaluation order");
        SCCHK(scratch(1, (m ? m : 1) * sizeof(Fe), &a));
        SCCHK(scratch(2, order * sizeof(Fe), &b));
        SCCHK(upload(a, coeffs, m * sizeof(Fe), g.stream));
    }
    SCCHK(sc_coset_evaluate_dev(a, m, offset, generator, order, b, nullptr));
    std::lock_guard<std::mutex> lk(g_mu);
    return download(out, b, order * sizeof(Fe), g.stream);
}

// ---- pointwise
int sc_pointwise_mul_dev(const void* d_a, const void* d_b, void* d_out, uint64_t n, void* stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    if (!n) return SC_OK;
    hipLaunchKernelGGL(pointwise_mul_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, pick_stream(stream), (const Fe*)d_a, (const Fe*)d_b, (Fe*)d_out, n);
    HIPCHK(hipGetLastError());
    return SC_OK;
}
int sc_pointwise_div_dev(const void* d_a, const void* d_b, void* d_out, uint64_t n, void* stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    if (!n) return SC_OK;
    return pointwise_div_device((const Fe*)d_a, (const Fe*)d_b, (Fe*)d_out, n, pick_stream(stream));
}
int sc_scale_dev(const void* d_in, void* d_out, uint64_t n, const uint64_t factor[2], void* stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    if (!n) return SC_OK;
    hipStream_t st = pick_stream(stream);
    PowTables* pw;
    SCCHK(get_pow(fe_from(factor), n, st, &pw));
    hipLaunchKernelGGL(scale_pow_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const Fe*)d_in, (Fe*)d_out, n, pw->lo, pw->hi);
    HIPCHK(hipGetLastError());
    return SC_OK;
}

int sc_axpy_shift_dev(void* d_acc, uint64_t n_acc, const void* d_src, uint64_t n_src, uint64_t shift, const uint64_t weight[2], void* stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    if (!n_src) return SC_OK;
    if (shift + n_src > n_acc || shift + n_src < shift) return fail(SC_ERR_BAD_ARG, "shifted term does not fit the accumulator");
    Fe w = fe_from(weight);
    if (fe_ge_p(w)) return fail(SC_ERR_BAD_ARG, "weight is not a canonical residue");
    hipLaunchKernelGGL(axpy_shift_kernel, dim3((unsigned)((n_src + 255) / 256)), dim3(256), 0, pick_stream(stream), (Fe*)d_acc, (const Fe*)d_src, n_src, shift, to_mont(w));
    HIPCHK(hipGetLastError());
    return SC_OK;
}

int sc_scale_slab_dev(const void* d_in, void* d_out, uint64_t rows, uint64_t cols, uint64_t row_len, uint64_t col_base, const uint64_t factor[2], void* stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    if (!rows || !cols) return SC_OK;
    if (!is_pow2(cols) || col_base + cols > row_len) return fail(SC_ERR_BAD_ARG, "slab columns must be a power of two inside the row");
    hipStream_t st = pick_stream(stream);
    PowTables* pw;
    SCCHK(get_pow(fe_from(factor), rows * row_len, st, &pw));
    const uint64_t cnt = rows * cols;
    hipLaunchKernelGGL(scale_slab_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, (const Fe*)d_in, (Fe*)d_out, rows, ilog2(cols), row_len, col_base, pw->lo, pw->hi);
    HIPCHK(hipGetLastError());
    return SC_OK;
}

// ---- poly mul: intt(ntt(a) * ntt(b)) truncated
int sc_poly_mul(const void* a, uint64_t na, const void* b, uint64_t nb, const uint64_t root[2], uint64_t order, void* out, uint64_t n_out) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    if (!is_pow2(order) || order < 2) return fail(SC_ERR_NOT_POW2, "cannot compute ntt of non-power-of-two sequence");
    if (na > order || nb > order || n_out > order || na == 0 || nb == 0) return fail(SC_ERR_BAD_ARG, "operand longer than the transform order");
    Fe rt = fe_from(root);
    SCCHK(check_root(rt, order));
    hipStream_t st = g.stream;
    void *da, *db, *dc;
    SCCHK(scratch(1, order * sizeof(Fe), &da));
    SCCHK(scratch(2, order * sizeof(Fe), &db));
    SCCHK(scratch(3, order * sizeof(Fe), &dc));
    const int logn = ilog2(order);
    NttOpts o;
    SCCHK(upload(dc, a, na * sizeof(Fe), st));
    o.in_limit = na;
    SCCHK(ntt_device((const Fe*)dc, (Fe*)da, logn, rt, false, o, st));
    SCCHK(upload(dc, b, nb * sizeof(Fe), st));
    o.in_limit = nb;
    SCCHK(ntt_device((const Fe*)dc, (Fe*)db, logn, rt, false, o, st));
    hipLaunchKernelGGL(pointwise_mul_kernel, dim3((unsigned)((order + 255) / 256)), dim3(256), 0, st, (const Fe*)da, (const Fe*)db, (Fe*)dc, order);
    HIPCHK(hipGetLastError());
    SCCHK(ntt_device((const Fe*)dc, (Fe*)da, logn, root_inverse(rt, order), true, NttOpts{}, st));
    return download(out, da, n_out * sizeof(Fe), st);
}

// ---- coset divide
// core of fast_coset_divide (code/ntt.py:159-176) on device operands: ALL `order` coefficients of the unscaled interpolant of
// ntt(scale(a)) / ntt(scale(b)) land in scratch slot 2 (returned in *full)
// zero_flag_dev != nullptr: the pointwise division is only enqueued and *zero_flag_dev names its "divide by zero" word (deferred check)
static int coset_divide_core(const Fe* d_a, uint64_t na, const Fe* d_b, uint64_t nb, Fe off, Fe rt, uint64_t order, Fe** full, hipStream_t st, uint32_t** zero_flag_dev = nullptr) {
    void *da, *db, *dc;
    SCCHK(scratch(1, order * sizeof(Fe), &da));
    SCCHK(scratch(2, order * sizeof(Fe), &db));
    SCCHK(scratch(3, order * sizeof(Fe), &dc));
    const int logn = ilog2(order);
    PowTables* pw;
    SCCHK(get_pow(off, order, st, &pw));
    NttOpts o;
    o.coset = pw;
    o.in_limit = na;
    SCCHK(ntt_device(d_a, (Fe*)da, logn, rt, false, o, st));
    if (nb <= SMALL_DIVISOR && g.small_divisor_direct) {
        PlanTables* pt;
        SCCHK(get_plan(rt, logn, false, st, &pt));                 // (the numerator's transform has just used these tables)
        hipLaunchKernelGGL(short_poly_coset_kernel, dim3((unsigned)((order + 255) / 256)), dim3(256), 0, st, d_b, (uint32_t)nb, to_mont(off), pt->tl, pt->th, (Fe*)db, order);
        HIPCHK(hipGetLastError());
    } else {
        o.in_limit = nb;
        SCCHK(ntt_device(d_b, (Fe*)db, logn, rt, false, o, st));
    }
    if (zero_flag_dev) SCCHK(pointwise_div_enqueue((const Fe*)da, (const Fe*)db, (Fe*)dc, order, st, zero_flag_dev));
    else SCCHK(pointwise_div_device((const Fe*)da, (const Fe*)db, (Fe*)dc, order, st));
    SCCHK(ntt_device((const Fe*)dc, (Fe*)da, logn, root_inverse(rt, order), true, NttOpts{}, st));
    // unscale by offset^-1 (ntt.py:176)
    Fe off_inv = from_mont(mont_inv(to_mont(off)));
    PowTables* pinv;
    SCCHK(get_pow(off_inv, order, st, &pinv));
    hipLaunchKernelGGL(scale_pow_kernel, dim3((unsigned)((order + 255) / 256)), dim3(256), 0, st, (const Fe*)da, (Fe*)db, order, pinv->lo, pinv->hi);
    HIPCHK(hipGetLastError());
    *full = (Fe*)db;
    return SC_OK;
}

static int coset_divide_args(uint64_t na, uint64_t nb, uint64_t n_out, const uint64_t offset[2], const uint64_t root[2], uint64_t order, Fe* rt, Fe* off) {
    if (!is_pow2(order) || order < 2) return fail(SC_ERR_NOT_POW2, "cannot compute ntt of non-power-of-two sequence");
    if (na > order || nb > order || n_out > order || na == 0 || nb == 0) return fail(SC_ERR_BAD_ARG, "operand longer than the transform order");
    *rt = fe_from(root); *off = fe_from(offset);
    SCCHK(check_root(*rt, order));
    if (fe_is_zero(*off) || fe_ge_p(*off)) return fail(SC_ERR_BAD_ARG, "bad coset offset");
    return SC_OK;
}

int sc_coset_divide(const void* a, uint64_t na, const void* b, uint64_t nb, const uint64_t offset[2], const uint64_t root[2], uint64_t order, void* out, uint64_t n_out) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    Fe rt, off;
    SCCHK(coset_divide_args(na, nb, n_out, offset, root, order, &rt, &off));
    hipStream_t st = g.stream;
    void *ua, *ub;
    SCCHK(scratch(6, (na + nb) * sizeof(Fe), &ua));
    ub = (Fe*)ua + na;
    SCCHK(upload(ua, a, na * sizeof(Fe), st));
    SCCHK(upload(ub, b, nb * sizeof(Fe), st));
    Fe* full;
    SCCHK(coset_divide_core((const Fe*)ua, na, (const Fe*)ub, nb, off, rt, order, &full, st));
    return download(out, full, n_out * sizeof(Fe), st);
}

// the same on coefficient vectors in HBM, for callers that keep their polynomials on the device.  `exact` (may be NULL): set to
// 1 iff the coefficients [n_out, order) of the interpolant vanish -- with order > deg(a) that is exactly "b divides a with
// quotient degree < n_out", the condition Polynomial.__truediv__ asserts (code/univariate.py:99-103).
int sc_coset_divide_dev(const void* d_a, uint64_t na, const void* d_b, uint64_t nb, const uint64_t offset[2], const uint64_t root[2], uint64_t order,
                        void* d_out, uint64_t n_out, int* exact, void* stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    Fe rt, off;
    SCCHK(coset_divide_args(na, nb, n_out, offset, root, order, &rt, &off));
    hipStream_t st = pick_stream(stream);
    Fe* full;
    SCCHK(coset_divide_core((const Fe*)d_a, na, (const Fe*)d_b, nb, off, rt, order, &full, st));
    if (n_out) HIPCHK(hipMemcpyAsync(d_out, full, n_out * sizeof(Fe), hipMemcpyDeviceToDevice, st));
    if (exact) {
        void* fl;
        SCCHK(scratch(7, 256, &fl));
        long long deg = -1;
        HIPCHK(hipMemsetAsync(fl, 0xFF, DEGREE_SLOTS * sizeof deg, st));      // -1 in every slot, without a pageable host-to-device copy in front of the kernel
        if (order > n_out) {
            const uint64_t cnt = order - n_out;
            hipLaunchKernelGGL(vec_degree_kernel, dim3(degree_blocks(cnt)), dim3(256), 0, st, (const Fe*)full + n_out, cnt, (long long*)fl);
            HIPCHK(hipGetLastError());
        }
        SCCHK(degree_read(fl, st, &deg));
        *exact = deg < 0 ? 1 : 0;
    }
    return SC_OK;
}

int sc_coset_divide_later_dev(const void* d_a, uint64_t na, const void* d_b, uint64_t nb, const uint64_t offset[2], const uint64_t root[2], uint64_t order,
                              void* d_out, uint64_t n_out, sc_later_t** later, void* stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    if (!later) return fail(SC_ERR_BAD_ARG, "null argument");
    Fe rt, off;
    SCCHK(coset_divide_args(na, nb, n_out, offset, root, order, &rt, &off));
    hipStream_t st = pick_stream(stream);
    Fe* full;
    uint32_t* zero_flag = nullptr;
    SCCHK(coset_divide_core((const Fe*)d_a, na, (const Fe*)d_b, nb, off, rt, order, &full, st, &zero_flag));
    if (n_out) HIPCHK(hipMemcpyAsync(d_out, full, n_out * sizeof(Fe), hipMemcpyDeviceToDevice, st));
    void* fl;
    SCCHK(scratch(7, 256, &fl));
    HIPCHK(hipMemsetAsync(fl, 0xFF, DEGREE_SLOTS * sizeof(long long), st));
    if (order > n_out) {
        const uint64_t cnt = order - n_out;
        hipLaunchKernelGGL(vec_degree_kernel, dim3(degree_blocks(cnt)), dim3(256), 0, st, (const Fe*)full + n_out, cnt, (long long*)fl);
        HIPCHK(hipGetLastError());
    }
    return divide_flags_later(zero_flag, (const long long*)fl, st, later);
}
int sc_pointwise_div_later_dev(const void* d_a, const void* d_b, void* d_out, uint64_t n, sc_later_t** later, void* stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    if (!later || !n) return fail(SC_ERR_BAD_ARG, "null argument");
    hipStream_t st = pick_stream(stream);
    uint32_t* zero_flag = nullptr;
    SCCHK(pointwise_div_enqueue((const Fe*)d_a, (const Fe*)d_b, (Fe*)d_out, n, st, &zero_flag));
    return divide_flags_later(zero_flag, nullptr, st, later);
}
int sc_later_wait(sc_later_t* later, int64_t words_out[8]) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!later || !words_out) return fail(SC_ERR_BAD_ARG, "null argument");
    SCCHK(ensure_init());
    return later_wait(later, (uint64_t*)words_out);
}

int sc_vec_degree_dev(const void* d_v, uint64_t n, int64_t* degree_out, void* stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    hipStream_t st = pick_stream(stream);
    void* fl;
    SCCHK(scratch(7, 256, &fl));
    long long deg = -1;
    HIPCHK(hipMemsetAsync(fl, 0xFF, DEGREE_SLOTS * sizeof deg, st));  // -1 in every slot, without a pageable host-to-device copy in front of the kernel
    // the leading coefficient of a polynomial is almost always in its last few entries: look at the top 2^16 first, and at
    // the rest only when those are all zero
    const uint64_t top = n < (1ull << 16) ? n : (1ull << 16);
    if (top) {
        hipLaunchKernelGGL(vec_degree_kernel, dim3(degree_blocks(top)), dim3(256), 0, st, (const Fe*)d_v + (n - top), top, (long long*)fl);
        HIPCHK(hipGetLastError());
    }
    SCCHK(degree_read(fl, st, &deg));
    if (deg >= 0) deg += (long long)(n - top);
    else if (n > top) {
        hipLaunchKernelGGL(vec_degree_kernel, dim3(degree_blocks(n - top)), dim3(256), 0, st, (const Fe*)d_v, n - top, (long long*)fl);
        HIPCHK(hipGetLastError());
        SCCHK(degree_read(fl, st, &deg));
    }
    *degree_out = (int64_t)deg;
    return SC_OK;
}

// ---- MPolynomial.evaluate_symbolic in the value domain
int sc_mpoly_eval_dev(void* d_vals, uint64_t nvars, uint64_t n, const uint8_t* exps, const void* coefs, uint64_t nterms, void* d_out, void* stream) {
    return sc_mpoly_eval_ex_dev(d_vals, nvars, n, exps, coefs, nterms, d_out, 0, stream);
}
// vals_converted != 0: d_vals has been through an earlier call already (several constraints over the same point values: the
// conversion to the library's internal form happens once)
int sc_mpoly_eval_ex_dev(void* d_vals, uint64_t nvars, uint64_t n, const uint8_t* exps, const void* coefs, uint64_t nterms, void* d_out, int vals_converted, void* stream) {
    return sc_mpoly_eval_rot_dev(d_vals, nvars, n, exps, coefs, nterms, d_out, vals_converted, nullptr, nullptr, stream);
}
// var_src / var_rot (both or neither): variable j's values are not in d_vals[j] but are those of variable var_src[j] turned by
// var_rot[j] places -- value at point i = d_vals[var_src[j]][(i + var_rot[j]) mod n], n a power of two; var_src[j] == j, var_rot[j] == 0
// for a variable stored in its own place, var_src[j] == SC_MPOLY_ABSENT for one no term uses (its place is not touched).
int sc_mpoly_eval_rot_dev(void* d_vals, uint64_t nvars, uint64_t n, const uint8_t* exps, const void* coefs, uint64_t nterms, void* d_out, int vals_converted,
                          const uint32_t* var_src, const uint64_t* var_rot, void* stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    hipStream_t st = pick_stream(stream);
    if (!d_vals || !d_out || nvars == 0 || nvars > 255 || n == 0) return fail(SC_ERR_BAD_ARG, "bad argument");
    if ((var_src == nullptr) != (var_rot == nullptr)) return fail(SC_ERR_BAD_ARG, "var_src and var_rot come together");
    std::vector<uint64_t> where;
    if (var_src) {
        if (!is_pow2(n)) return fail(SC_ERR_NOT_POW2, "turned variables need a power-of-two domain");
        const uint8_t* e = exps;
        for (uint64_t j = 0; j < nvars; ++j) {
            const uint32_t s = var_src[j];
            if (s == SC_MPOLY_ABSENT) {
                for (uint64_t t = 0; t < nterms; ++t) if (e[t * nvars + j]) return fail(SC_ERR_BAD_ARG, "a term uses a variable that is marked absent");
                where.push_back(j); where.push_back(0);
                continue;
            }
            if (s >= nvars || var_src[s] != s || var_rot[s] != 0 || (s == j && var_rot[j] != 0)) return fail(SC_ERR_BAD_ARG, "a turned variable must point at one stored in its own place");
            where.push_back(s); where.push_back(var_rot[j] & (n - 1));
        }
    }
    const Fe* c = (const Fe*)coefs;
    std::vector<Fe> cm(nterms ? nterms : 1);
    for (uint64_t t = 0; t < nterms; ++t) {
        if (fe_ge_p(c[t])) return fail(SC_ERR_BAD_ARG, "coefficient is not a canonical residue");
        cm[t] = to_mont(c[t]);
    }
    const size_t cbytes = (cm.size() * sizeof(Fe) + 255) & ~255ull;
    const size_t ebytes = (nterms * nvars + 255) & ~255ull;
    void* buf;
    SCCHK(scratch(4, cbytes + ebytes + 16 * nvars + 256, &buf));
    uint64_t* d_where = where.empty() ? nullptr : (uint64_t*)((char*)buf + cbytes + ebytes);
    // (a handful of terms: coefficients and exponents go in as kernel arguments and nothing has to be waited for afterwards)
    const bool small = upload_small(buf, cm.data(), cm.size() * sizeof(Fe), st) && (!nterms || upload_small((char*)buf + cbytes, exps, nterms * nvars, st)) &&
                       (!d_where || upload_small(d_where, where.data(), where.size() * 8, st));
    if (!small) {
        SCCHK(upload(buf, cm.data(), cm.size() * sizeof(Fe), st));
        if (nterms) SCCHK(upload((char*)buf + cbytes, exps, nterms * nvars, st));
        if (d_where) SCCHK(upload(d_where, where.data(), where.size() * 8, st));
    }
    if (!vals_converted) {
        if (!var_src) hipLaunchKernelGGL(to_mont_kernel, dim3((unsigned)((nvars * n + 255) / 256)), dim3(256), 0, st, (Fe*)d_vals, nvars * n);
        else
            for (uint64_t j = 0; j < nvars; ++j)       // only what is stored: the places of turned and of absent variables hold nothing
                if (var_src[j] == j) hipLaunchKernelGGL(to_mont_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (Fe*)d_vals + j * n, n);
    }
    hipLaunchKernelGGL(mpoly_eval_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const Fe*)d_vals, (uint32_t)nvars, n,
                       (const uint8_t*)((char*)buf + cbytes), (const Fe*)buf, (uint32_t)nterms, (Fe*)d_out, (const uint64_t*)d_where);
    HIPCHK(hipGetLastError());
    // (not waited for on the library's own stream: the scratch tables' next writer is ordered behind this kernel there; a caller's
    // stream shares the scratch buffer with other streams, so the call still ends with the kernel done)
    if (!small || st != g.stream) HIPCHK(hipStreamSynchronize(st));       // cm / exps may be host temporaries of this call
    return SC_OK;
}

}  // extern "C"
