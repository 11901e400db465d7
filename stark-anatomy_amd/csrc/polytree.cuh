// Subproduct tree, multipoint evaluation and interpolation on the device: code/ntt.py:66-80 (fast_zerofier), :82-100
// (fast_evaluate), :102-130 (fast_interpolate).
//
// The reference recurses node by node over Python lists (split at len//2, schoolbook remainders at every node of
// fast_evaluate, zerofiers recomputed per node).  Zerofier, values and the interpolant of degree < k are unique, so the
// device computes them LEVEL BY LEVEL over a perfect binary tree instead (model: tests/emu/polytree_model.py):
//   * the k points are padded with zeros to K = 2^L leaves (a zero leaf multiplies the zerofier by x, stripped at the end);
//   * level l holds K/2^l monic node polynomials of degree 2^l, coefficient-major [2^l][K/2^l] (top coefficient implicit):
//     one batched column transform covers a whole level, every access below is a unit-stride stream;
//   * Zf[l] = the size-2^(l+1) transforms of level l are kept: they are the operands of the products going up, of the
//     correlations going down (scaled remainder tree) and of the P_L*Z_R + P_R*Z_L combinations of interpolation.
// All kernels here are elementwise / HBM-streaming; the arithmetic is in the batched NTTs between them.
#pragma once
#include "field.cuh"

namespace sc {

#define PT_INDEX() ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x)

// level 0: node polynomial x - d_i  ->  low coefficient -d_i; padding leaves are the point 0
__global__ void __launch_bounds__(256) pt_leaves_kernel(const Fe* __restrict__ points, uint64_t k, Fe* __restrict__ zc0, uint64_t K) {
    uint64_t i = PT_INDEX();
    if (i < K) zc0[i] = (i < k) ? fe_neg(points[i]) : Fe{0, 0};
}

// [n][B] -> [2n][B]: rows < n copied, row n = `top` (1 for the implicit monic coefficient, 0 for plain zero padding), rest 0
__global__ void __launch_bounds__(256) pt_expand_kernel(const Fe* __restrict__ in, Fe* __restrict__ out, uint64_t nB, uint64_t B, uint64_t top) {
    uint64_t i = PT_INDEX();
    if (i >= 2 * nB) return;
    Fe v{0, 0};
    if (i < nB) v = in[i];
    else if (i < nB + B) v = Fe{top, 0};
    out[i] = v;
}

// frequency domain [len][B] -> [len][B/2]: out[f][j] = c * in[f][2j] * in[f][2j+1]   (c_m2 = c * R^2: undoes both Montgomery factors)
__global__ void __launch_bounds__(256) pt_mul_pairs_kernel(const Fe* __restrict__ in, Fe* __restrict__ out, uint64_t total_out, Fe c_m2) {
    uint64_t i = PT_INDEX();
    if (i >= total_out) return;
    Fe a = in[2 * i], b = in[2 * i + 1];                 // [f][2j], [f][2j+1]: B is even, so flat index 2*(f*B/2 + j)
    out[i] = mont_mul(mont_mul(a, b), c_m2);
}

// the monic coefficient x^(2n) of a product computed cyclically at size 2n wrapped onto x^0: first `count` entries -= 1
__global__ void __launch_bounds__(256) pt_sub_one_kernel(Fe* __restrict__ a, uint64_t count) {
    uint64_t i = PT_INDEX();
    if (i < count) a[i] = fe_sub(a[i], Fe{1, 0});
}

// scaled remainder tree, one level down.  C: [n][B] transforms of the nodes' series; zf: [n][2B] transforms of the children's
// polynomials; D[f][i] = c * C[f][i>>1] * zf[(-f) mod n][i^1]: the cyclic CORRELATION with the sibling, whose first n/2 outputs
// are the child's series (the index negation turns the convolution theorem into a correlation).
__global__ void __launch_bounds__(256) pt_corr_kernel(const Fe* __restrict__ C, const Fe* __restrict__ zf, Fe* __restrict__ D, uint64_t n, int logB2, Fe c_m2) {
    uint64_t t = PT_INDEX();
    if (t >= (n << logB2)) return;
    const uint64_t f = t >> logB2, i = t & ((1ull << logB2) - 1);
    const uint64_t nf = (n - f) & (n - 1);
    Fe a = C[(f << (logB2 - 1)) + (i >> 1)];
    Fe b = zf[(nf << logB2) + (i ^ 1)];
    D[t] = mont_mul(mont_mul(a, b), c_m2);
}

// interpolation, one level up: E[f][j] = c * (Ph[f][2j] * zf[f][2j+1] + Ph[f][2j+1] * zf[f][2j])
__global__ void __launch_bounds__(256) pt_comb_kernel(const Fe* __restrict__ Ph, const Fe* __restrict__ zf, Fe* __restrict__ E, uint64_t total_out, Fe c_m2) {
    uint64_t i = PT_INDEX();
    if (i >= total_out) return;
    Fe pl = Ph[2 * i], pr = Ph[2 * i + 1], zl = zf[2 * i], zr = zf[2 * i + 1];
    E[i] = mont_mul(fe_add(mont_mul(pl, zr), mont_mul(pr, zl)), c_m2);
}

// Newton step of the power-series inverse in the frequency domain: H <- c * H * (2 - T * H)   (c_m3 = c * R^3)
__global__ void __launch_bounds__(256) pt_newton_kernel(Fe* __restrict__ H, const Fe* __restrict__ T, uint64_t n, Fe c_m3) {
    uint64_t i = PT_INDEX();
    if (i >= n) return;
    Fe h = H[i];
    Fe th = mont_mul(T[i], h);                            // T*H / R
    Fe two_r = mont_mul(Fe{2, 0}, Fe{1, 0});              // 2 / R
    Fe s = fe_sub(two_r, th);                             // (2 - T*H) / R
    H[i] = mont_mul(mont_mul(h, s), c_m3);
}

// out[i] = c * a[i] * b[i]   (out may alias a)
__global__ void __launch_bounds__(256) pt_mul_scaled_kernel(const Fe* a, const Fe* __restrict__ b, Fe* out, uint64_t n, Fe c_m2) {
    uint64_t i = PT_INDEX();
    if (i < n) out[i] = mont_mul(mont_mul(a[i], b[i]), c_m2);
}

// G = rev(Z) mod y^K for the monic root polynomial Z (low coefficients top[0..K)): G[0] = 1, G[t] = top[K - t]
__global__ void __launch_bounds__(256) pt_rev_monic_kernel(const Fe* __restrict__ top, Fe* __restrict__ G, uint64_t K) {
    uint64_t t = PT_INDEX();
    if (t < K) G[t] = t ? top[K - t] : Fe{1, 0};
}

// F = rev_K(f) zero-extended to `total` entries: F[t] = f[K-1-t] for t < K (0 where K-1-t >= m)
__global__ void __launch_bounds__(256) pt_rev_poly_kernel(const Fe* __restrict__ f, uint64_t m, Fe* __restrict__ F, uint64_t K, uint64_t total) {
    uint64_t t = PT_INDEX();
    if (t >= total) return;
    Fe v{0, 0};
    if (t < K && K - 1 - t < m) v = f[K - 1 - t];
    F[t] = v;
}

// derivative of Z_real = Z / x^pad (degree k, monic), zero-padded to K entries: out[t] = (t+1) * Z_real[t+1], t < k
__global__ void __launch_bounds__(256) pt_deriv_kernel(const Fe* __restrict__ top, uint64_t K, uint64_t pad, uint64_t k, Fe* __restrict__ out) {
    uint64_t t = PT_INDEX();
    if (t >= K) return;
    Fe v{0, 0};
    if (t < k) {
        const uint64_t s = t + 1 + pad;                   // index into the full root polynomial
        Fe z = (s < K) ? top[s] : Fe{1, 0};
        v = mont_mul(z, to_mont(Fe{t + 1, 0}));
    }
    out[t] = v;
}

// zerofier coefficients (k + 1 of them): Z / x^pad with the monic top coefficient made explicit
__global__ void __launch_bounds__(256) pt_zerofier_out_kernel(const Fe* __restrict__ top, uint64_t K, uint64_t pad, uint64_t k, Fe* __restrict__ out) {
    uint64_t t = PT_INDEX();
    if (t > k) return;
    out[t] = (t + pad < K) ? top[t + pad] : Fe{1, 0};
}

// column b of a [len][B] array <-> contiguous vector (the few top levels whose columns are longer than the batched plans take)
__global__ void __launch_bounds__(256) pt_col_gather_kernel(const Fe* __restrict__ a, uint64_t len, uint64_t B, uint64_t b, Fe* __restrict__ v) {
    uint64_t i = PT_INDEX();
    if (i < len) v[i] = a[i * B + b];
}
__global__ void __launch_bounds__(256) pt_col_scatter_kernel(const Fe* __restrict__ v, uint64_t len, uint64_t B, uint64_t b, Fe* __restrict__ a) {
    uint64_t i = PT_INDEX();
    if (i < len) a[i * B + b] = v[i];
}

// polynomials with more than K coefficients are evaluated in chunks of K: f(x) = sum_j f_j(x) * (x^K)^j (Horner over the chunks)
// y[i] = points[i]^(2^logK)
__global__ void __launch_bounds__(256) pt_pow2_kernel(const Fe* __restrict__ points, uint64_t k, int logK, Fe* __restrict__ y) {
    uint64_t i = PT_INDEX();
    if (i >= k) return;
    Fe v = to_mont(points[i]);
    for (int s = 0; s < logK; ++s) v = mont_mul(v, v);
    y[i] = v;                                             // kept in Montgomery form: acc * y below is one mont_mul
}
// acc[i] = acc[i] * y[i] + e[i]
__global__ void __launch_bounds__(256) pt_horner_kernel(Fe* __restrict__ acc, const Fe* __restrict__ y_m, const Fe* __restrict__ e, uint64_t k) {
    uint64_t i = PT_INDEX();
    if (i < k) acc[i] = fe_add(mont_mul(acc[i], y_m[i]), e[i]);
}

#undef PT_INDEX

}  // namespace sc
