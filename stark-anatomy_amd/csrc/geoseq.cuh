// fast_zerofier / fast_evaluate / fast_interpolate (code/ntt.py:66-130) on a GEOMETRIC PROGRESSION  x_i = c * q^i, i < n.
//
// The caller that matters hands exactly such a domain to fast_interpolate: the trace domain {omicron^i} of
// code/fast_stark.py:84-90.  Zerofier, values and the interpolant of degree < n are unique, so the device may compute them
// another way than the reference's recursion (or the general subproduct tree of polytree.cuh) and still return the same lists.
// On a progression everything is a handful of length-M convolutions (M = the power of two >= 2n - 1), Bostan & Schost 2005 /
// Bluestein 1970 (model and derivation: tests/emu/geoseq_model.py, checked against the oracle on the CPU):
//   t_j  = q^(j(j-1)/2)                  i*j = C(i+j,2) - C(i,2) - C(j,2)  =>  sum_i a_i q^(i m) = t_m^-1 * sum_i (a_i / t_i) t_(i+m)
//   A_i  = prod_{m=1..i} (q^m - 1)       Z'(q^i) = (-1)^(n-1-i) t_i A_i A_(n-1-i) q^(i(n-1-i))
//   Z(X) = prod (X - q^i):  coefficient of X^(n-k) = (-1)^k t_k A_n / (A_k A_(n-k))                    (q-binomial theorem)
//   P/Z  = sum_m s_m X^(-m-1),  s_m = sum_i (v_i / Z'(q^i)) q^(i m)   =>  rev(P) = rev(Z) * S mod y^n   (partial fractions)
// Per interpolation: 4 transforms of length M and 5 elementwise kernels; per evaluation: 2 and 3.  Per domain (once): three
// prefix-product scans, two transforms.  All tables below that end in _m hold Montgomery forms (x * 2^128): a table entry is
// always the second operand of one mont_mul with canonical data.
#pragma once
#include "field.cuh"
#include "ntt_tile.cuh"   // pow2level

namespace sc {

#define GS_INDEX() ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x)

// ---- inclusive prefix products of an array of Montgomery forms, in place: 2048 elements per workgroup (256 threads x 8) ------
constexpr int GS_E = 8;
constexpr int GS_T = 256;
constexpr uint64_t GS_BLOCK = (uint64_t)GS_E * GS_T;

// product of every workgroup's 2048 elements
__global__ void __launch_bounds__(GS_T) gs_totals_kernel(const Fe* __restrict__ a, uint64_t n, Fe* __restrict__ tot) {
    __shared__ Fe lds[GS_T];
    const uint32_t t = threadIdx.x;
    const uint64_t base = (uint64_t)blockIdx.x * GS_BLOCK + (uint64_t)t * GS_E;
    Fe prod = fe_mont_one();
#pragma unroll
    for (int k = 0; k < GS_E; ++k)
        if (base + k < n) prod = mont_mul(prod, a[base + k]);
    lds[t] = prod;
    for (uint32_t s = GS_T / 2; s > 0; s >>= 1) {
        __syncthreads();
        if (t < s) lds[t] = mont_mul(lds[t], lds[t + s]);
    }
    if (t == 0) tot[blockIdx.x] = lds[0];
}

// the scan proper; `before` (nullable): inclusive scan of the workgroup totals, so before[b - 1] is everything in front of workgroup b
__global__ void __launch_bounds__(GS_T) gs_apply_kernel(Fe* __restrict__ a, uint64_t n, const Fe* __restrict__ before) {
    __shared__ Fe lds[GS_T];
    const uint32_t t = threadIdx.x;
    const uint64_t base = (uint64_t)blockIdx.x * GS_BLOCK + (uint64_t)t * GS_E;
    Fe x[GS_E];
    Fe run = fe_mont_one();
#pragma unroll
    for (int k = 0; k < GS_E; ++k) {
        if (base + k < n) run = mont_mul(run, a[base + k]);
        x[k] = run;
    }
    lds[t] = run;
    __syncthreads();
    for (uint32_t off = 1; off < GS_T; off <<= 1) {             // Hillis-Steele over the 256 thread totals
        Fe v = lds[t];
        if (t >= off) v = mont_mul(lds[t - off], v);
        __syncthreads();
        lds[t] = v;
        __syncthreads();
    }
    Fe front = t ? lds[t - 1] : fe_mont_one();
    if (before && blockIdx.x) front = mont_mul(front, before[blockIdx.x - 1]);
#pragma unroll
    for (int k = 0; k < GS_E; ++k)
        if (base + k < n) a[base + k] = mont_mul(front, x[k]);
}

// ---- scan inputs (lo/hi: two-level power table of the base, Montgomery forms) ------------------------------------------------
// mode 0: f_j = 1, base^0, base^1, ...  (j = 0: 1; else base^(j-1))   scan -> base^(j(j-1)/2) = t_j
// mode 1: e_j = base^(j+1) - 1                                        scan -> A_(j+1)
// mode 2: e_(n-2-j) = base^(n-1-j) - 1, j < n - 1                     scan -> S_(n-2-j) = prod_{m=n-1-j..n-1} (base^m - 1)
__global__ void __launch_bounds__(256) geo_fill_kernel(Fe* __restrict__ out, uint64_t count, int mode, uint64_t n, const Fe* __restrict__ lo, const Fe* __restrict__ hi) {
    const uint64_t j = GS_INDEX();
    if (j >= count) return;
    Fe v;
    if (mode == 0) v = j ? pow2level(lo, hi, j - 1) : fe_mont_one();
    else if (mode == 1) v = fe_sub(pow2level(lo, hi, j + 1), fe_mont_one());
    else v = fe_sub(pow2level(lo, hi, n - 1 - j), fe_mont_one());
    out[j] = v;
}

// S_i = prod_{m=i+1..n-1} (q^m - 1) out of the reversed scan `rev` (n - 1 entries, rev[j] = S_(n-2-j)); S_(n-1) = 1
__device__ __forceinline__ Fe geo_S(const Fe* __restrict__ rev, uint64_t n, uint64_t i) { return i + 1 < n ? rev[n - 2 - i] : fe_mont_one(); }

// interpolation weights  wden_m[i] = 1 / (Z'(q^i) t_i) = (-1)^(n-1-i) * A_(n-1)^-2 * S_i * S_(n-1-i) * g^i,  g = q^-(n-2);
// k1_m = A_(n-1)^-2, glo/ghi = power table of g
__global__ void __launch_bounds__(256) geo_wden_kernel(Fe* __restrict__ wden_m, uint64_t n, const Fe* __restrict__ rev, Fe k1_m, const Fe* __restrict__ glo, const Fe* __restrict__ ghi) {
    const uint64_t i = GS_INDEX();
    if (i >= n) return;
    Fe v = mont_mul(mont_mul(geo_S(rev, n, i), geo_S(rev, n, n - 1 - i)), mont_mul(k1_m, pow2level(glo, ghi, i)));
    wden_m[i] = ((n - 1 - i) & 1) ? fe_neg(v) : v;
}

// reversed zerofier of {q^i}: zr[k] = coefficient of X^(n-k), k = 0..n, canonical:
//   zr[0] = 1,  zr[k] = (-1)^k t_k * (A_n A_(n-1)^-2) * S_k * S_(n-k)  (0 < k < n),  zr[n] = (-1)^n t_n;   k2_m = A_n A_(n-1)^-2
__global__ void __launch_bounds__(256) geo_zr_kernel(Fe* __restrict__ zr, uint64_t n, const Fe* __restrict__ rev, const Fe* __restrict__ t_m, Fe k2_m) {
    const uint64_t k = GS_INDEX();
    if (k > n) return;
    Fe v;
    if (k == 0) v = fe_mont_one();
    else if (k == n) v = t_m[n];
    else v = mont_mul(mont_mul(geo_S(rev, n, k), geo_S(rev, n, n - k)), mont_mul(k2_m, t_m[k]));
    v = from_mont(v);
    zr[k] = (k & 1) ? fe_neg(v) : v;
}

__global__ void __launch_bounds__(256) geo_from_mont_kernel(Fe* __restrict__ a, uint64_t n) {
    const uint64_t i = GS_INDEX();
    if (i < n) a[i] = from_mont(a[i]);
}

// ---- per operation ---------------------------------------------------------------------------------------------------------
// out[i] = in[i] * tab[i]   (canonical * Montgomery form -> canonical)
__global__ void __launch_bounds__(256) geo_mul_tab_kernel(const Fe* __restrict__ in, const Fe* __restrict__ tab_m, Fe* __restrict__ out, uint64_t n) {
    const uint64_t i = GS_INDEX();
    if (i < n) out[i] = mont_mul(in[i], tab_m[i]);
}

// cyclic correlation in the frequency domain: D[f] = c * A[-f] * B[f]   (c_m2 = c * R^2)
__global__ void __launch_bounds__(256) geo_corr_kernel(const Fe* __restrict__ A, const Fe* __restrict__ B, Fe* __restrict__ D, uint64_t M, Fe c_m2) {
    const uint64_t f = GS_INDEX();
    if (f >= M) return;
    D[f] = mont_mul(mont_mul(A[(M - f) & (M - 1)], B[f]), c_m2);
}

// evaluation input: a_j = p_j * c^j / t_j  (clo == nullptr: c = 1)
__global__ void __launch_bounds__(256) geo_eval_in_kernel(const Fe* __restrict__ p, uint64_t m, const Fe* __restrict__ clo, const Fe* __restrict__ chi, const Fe* __restrict__ tinv_m,
                                                         Fe* __restrict__ out) {
    const uint64_t j = GS_INDEX();
    if (j >= m) return;
    Fe v = mont_mul(p[j], tinv_m[j]);
    if (clo) v = mont_mul(v, pow2level(clo, chi, j));
    out[j] = v;
}

// interpolation output: P_j = Q_j * c^-j with Q reversed in `qrev` (ilo == nullptr: c = 1)
__global__ void __launch_bounds__(256) geo_rev_scale_kernel(const Fe* __restrict__ qrev, uint64_t n, const Fe* __restrict__ ilo, const Fe* __restrict__ ihi, Fe* __restrict__ out) {
    const uint64_t j = GS_INDEX();
    if (j >= n) return;
    Fe v = qrev[n - 1 - j];
    if (ilo) v = mont_mul(v, pow2level(ilo, ihi, j));
    out[j] = v;
}

// zerofier of {c q^i}: coefficient j = zr[n - j] * c^(n - j), j = 0..n
__global__ void __launch_bounds__(256) geo_zerofier_out_kernel(const Fe* __restrict__ zr, uint64_t n, const Fe* __restrict__ clo, const Fe* __restrict__ chi, Fe* __restrict__ out) {
    const uint64_t j = GS_INDEX();
    if (j > n) return;
    Fe v = zr[n - j];
    if (clo) v = mont_mul(v, pow2level(clo, chi, n - j));
    out[j] = v;
}

// Horner step over chunks of n coefficients: y_i = x_i^n = c^n * (q^n)^i as Montgomery forms (ylo/yhi: power table of q^n)
__global__ void __launch_bounds__(256) geo_chunk_power_kernel(Fe* __restrict__ y_m, uint64_t n, Fe cn_m, const Fe* __restrict__ ylo, const Fe* __restrict__ yhi) {
    const uint64_t i = GS_INDEX();
    if (i < n) y_m[i] = mont_mul(cn_m, pow2level(ylo, yhi, i));
}

// is d[i + 1] == d[i] * ratio for every i < n - 1 ?  flag[0] |= 1 where not   (ratio_m: Montgomery form)
__global__ void __launch_bounds__(256) geo_detect_kernel(const Fe* __restrict__ d, uint64_t n, Fe ratio_m, uint32_t* flag) {
    const uint64_t i = GS_INDEX();
    if (i + 1 >= n) return;
    if (!fe_eq(mont_mul(d[i], ratio_m), d[i + 1])) atomicOr(flag, 1u);
}

#undef GS_INDEX

}  // namespace sc
