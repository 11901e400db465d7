// polytree_geo.hip -- fast_zerofier / fast_evaluate / fast_interpolate (code/ntt.py:66-130): the subproduct tree over arbitrary points
// (polytree.cuh) and the closed forms on geometric progressions (geoseq.cuh), resident in HBM.
#include "core.h"
#include "polytree.cuh"
#include "geoseq.cuh"

namespace sci {

// ============================================================================ subproduct tree (polytree.cuh)

// primitive 2^logn-th root the tree's internal transforms use: Field.primitive_nth_root (algebra.py:104-111), i.e. the
// order-2^119 constant squared 119 - logn times.  The results of sc_polytree_* do not depend on which roots are used.
Fe canonical_root(int logn) {
    static Fe cache[120];
    static bool have[120] = {false};
    if (!have[logn]) {
        Fe r = to_mont(Fe{0xb5038f9c18f6f7d1ull, 0x4040fbed12ee470full});
        for (int i = 119; i > logn; --i) r = mont_mul(r, r);
        cache[logn] = from_mont(r);
        have[logn] = true;
    }
    return cache[logn];
}

// c * R^j (mod p) for c = (2^logn)^-1: the constant a pointwise kernel multiplies by to apply the inverse transform's n^-1
// and cancel the R^-1 factors of its j Montgomery products
Fe ninv_scaled(int logn, int j) {
    Fe c = from_mont(mont_inv(to_mont(Fe{1ull << logn, 0})));
    for (int i = 0; i < j; ++i) c = to_mont(c);
    return c;
}

inline unsigned pt_blocks(uint64_t n) { return (unsigned)((n + 255) / 256); }

// transform along axis 0 of a [2^loglen][2^logbatch] array (natural order, out != in); the inverse uses root^-1 and does NOT
// scale by n^-1 (the pointwise kernel in front of it does)
int ntt_cols(const Fe* in, Fe* out, int loglen, int logbatch, bool inverse, hipStream_t st) {
    const uint64_t len = 1ull << loglen, B = 1ull << logbatch;
    if (loglen == 0) {
        if (in != out) HIPCHK(hipMemcpyAsync(out, in, B * sizeof(Fe), hipMemcpyDeviceToDevice, st));
        return SC_OK;
    }
    Fe rt = canonical_root(loglen);
    if (inverse) rt = root_inverse(rt, len);
    if (logbatch == 0) return ntt_device(in, out, loglen, rt, false, NttOpts(), st);
    PlanTables* pt;
    SCCHK(get_plan(rt, loglen, false, st, &pt));
    void* w;
    SCCHK(scratch(0, len * B * sizeof(Fe), &w));
    NttPlanDesc d;
    bool planned = false;
    SCCHK(plan_batched_direct(d, BATCH_COLS, loglen, logbatch, pt, in, (Fe*)w, out, BatchExtras(), st, &planned));
    if (planned) return run_plan(d, st);
    // columns longer than the batched plans take (only the top few levels of a big tree, a handful of columns each)
    void *a, *b;
    SCCHK(scratch(1, len * sizeof(Fe), &a));
    SCCHK(scratch(2, len * sizeof(Fe), &b));
    for (uint64_t c = 0; c < B; ++c) {
        hipLaunchKernelGGL(pt_col_gather_kernel, dim3(pt_blocks(len)), dim3(256), 0, st, in, len, B, c, (Fe*)a);
        SCCHK(ntt_device((const Fe*)a, (Fe*)b, loglen, rt, false, NttOpts(), st));
        hipLaunchKernelGGL(pt_col_scatter_kernel, dim3(pt_blocks(len)), dim3(256), 0, st, (const Fe*)b, len, B, c, out);
    }
    HIPCHK(hipGetLastError());
    return SC_OK;
}

}  // namespace sci

struct sc_polytree {
    uint64_t k, K;
    int L;
    Fe* zc;        // (L+1) levels of K entries: level l at zc + l*K, [2^l][K >> l], monic top coefficient implicit
    Fe* zf;        // L levels of 2K entries: level l at zf + l*2K, [2^(l+1)][K >> l] = transforms of level l at twice its size
    Fe* invg_f;    // size-2K transform of rev(Z)^-1 mod y^K (built by the first evaluation)
    size_t zc_bytes, zf_bytes;
};

namespace {

int polytree_build(const Fe* d_points, uint64_t k, sc_polytree** out, hipStream_t st) {
    if (k == 0) return fail(SC_ERR_BAD_ARG, "empty domain");
    int L = 0;
    while ((1ull << L) < k) ++L;
    if (L > 30) return fail(SC_ERR_UNSUPPORTED, "domain too large");
    const uint64_t K = 1ull << L;
    sc_polytree* t = new sc_polytree{k, K, L, nullptr, nullptr, nullptr, (size_t)(L + 1) * K * sizeof(Fe), (size_t)(L ? L : 1) * 2 * K * sizeof(Fe)};
    hipError_t e = pool_alloc((void**)&t->zc, t->zc_bytes);
    if (e == hipSuccess) e = pool_alloc((void**)&t->zf, t->zf_bytes);
    if (e != hipSuccess) {
        if (t->zc) pool_free(t->zc, t->zc_bytes);
        delete t;
        return fail(SC_ERR_HIP, hipGetErrorString(e));
    }
    auto cleanup = [&](int rc) { pool_free(t->zc, t->zc_bytes); pool_free(t->zf, t->zf_bytes); delete t; return rc; };
    PoolTmp buf, buf2;
    int rc = buf.get(2 * K * sizeof(Fe));
    if (rc == SC_OK) rc = buf2.get(K * sizeof(Fe));
    if (rc != SC_OK) return cleanup(rc);
    hipLaunchKernelGGL(pt_leaves_kernel, dim3(pt_blocks(K)), dim3(256), 0, st, d_points, k, t->zc, K);
    for (int l = 0; l < L && rc == SC_OK; ++l) {
        const uint64_t B = K >> l;
        Fe* zcl = t->zc + (uint64_t)l * K;
        Fe* zfl = t->zf + (uint64_t)l * 2 * K;
        hipLaunchKernelGGL(pt_expand_kernel, dim3(pt_blocks(2 * K)), dim3(256), 0, st, (const Fe*)zcl, buf.fe(), K, B, (uint64_t)1);
        rc = ntt_cols(buf.fe(), zfl, l + 1, L - l, false, st);
        if (rc != SC_OK) break;
        hipLaunchKernelGGL(pt_mul_pairs_kernel, dim3(pt_blocks(K)), dim3(256), 0, st, (const Fe*)zfl, buf2.fe(), K, ninv_scaled(l + 1, 2));
        rc = ntt_cols(buf2.fe(), zcl + K, l + 1, L - l - 1, true, st);
        if (rc != SC_OK) break;
        hipLaunchKernelGGL(pt_sub_one_kernel, dim3(pt_blocks(B / 2)), dim3(256), 0, st, zcl + K, B / 2);
    }
    if (rc == SC_OK && hipGetLastError() != hipSuccess) rc = fail(SC_ERR_HIP, "polytree build launch failed");
    if (rc == SC_OK && hipStreamSynchronize(st) != hipSuccess) rc = fail(SC_ERR_HIP, "polytree build failed");
    if (rc != SC_OK) return cleanup(rc);
    *out = t;
    return SC_OK;
}

// rev(Z)^-1 mod y^K by Newton iteration (h <- h (2 - G h), precision doubling), kept as its size-2K transform
int polytree_inverse_series(sc_polytree* t, hipStream_t st) {
    if (t->invg_f || t->L == 0) return SC_OK;
    const uint64_t K = t->K;
    const int L = t->L;
    PoolTmp G, h, H, T;
    SCCHK(G.get(K * sizeof(Fe)));
    SCCHK(h.get(2 * K * sizeof(Fe)));
    SCCHK(H.get(2 * K * sizeof(Fe)));
    SCCHK(T.get(2 * K * sizeof(Fe)));
    hipLaunchKernelGGL(pt_rev_monic_kernel, dim3(pt_blocks(K)), dim3(256), 0, st, (const Fe*)(t->zc + (uint64_t)L * K), G.fe(), K);
    const Fe one{1, 0};
    HIPCHK(hipMemcpyAsync(h.p, &one, sizeof(Fe), hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));                    // `one` is a stack temporary
    for (int lm = 0; lm < L; ++lm) {                      // m = 2^lm known coefficients -> 2m
        const int logn = lm + 2;
        const uint64_t n = 1ull << logn, m = 1ull << lm;
        Fe rt = canonical_root(logn);
        NttOpts o;
        o.in_limit = m;
        SCCHK(ntt_device(h.fe(), H.fe(), logn, rt, false, o, st));
        o.in_limit = 2 * m;
        SCCHK(ntt_device(G.fe(), T.fe(), logn, rt, false, o, st));
        hipLaunchKernelGGL(pt_newton_kernel, dim3(pt_blocks(n)), dim3(256), 0, st, H.fe(), (const Fe*)T.fe(), n, ninv_scaled(logn, 3));
        SCCHK(ntt_device(H.fe(), h.fe(), logn, root_inverse(rt, n), false, NttOpts(), st));
    }
    Fe* f = nullptr;
    HIPCHK(pool_alloc((void**)&f, 2 * K * sizeof(Fe)));
    NttOpts o;
    o.in_limit = K;
    int rc = ntt_device(h.fe(), f, L + 1, canonical_root(L + 1), false, o, st);
    if (rc == SC_OK && hipStreamSynchronize(st) != hipSuccess) rc = fail(SC_ERR_HIP, "inverse series failed");
    if (rc != SC_OK) { pool_free(f, 2 * K * sizeof(Fe)); return rc; }
    t->invg_f = f;
    return SC_OK;
}

// values of the polynomial d_coeffs[0..m), m <= K, at all K leaves (the k real points first); d_out holds K entries.
// bx, by: caller's temporaries of 2K entries each, tk: K entries.
int polytree_evaluate_all(sc_polytree* t, const Fe* d_coeffs, uint64_t m, Fe* d_out, Fe* bx, Fe* by, Fe* tk, hipStream_t st) {
    const uint64_t K = t->K;
    const int L = t->L;
    if (L == 0) {
        if (m) HIPCHK(hipMemcpyAsync(d_out, d_coeffs, sizeof(Fe), hipMemcpyDeviceToDevice, st));
        else HIPCHK(hipMemsetAsync(d_out, 0, sizeof(Fe), st));
        return SC_OK;
    }
    SCCHK(polytree_inverse_series(t, st));
    // root: c = first K coefficients of f/Z in 1/x = rev_K(f) * rev(Z)^-1 mod y^K
    hipLaunchKernelGGL(pt_rev_poly_kernel, dim3(pt_blocks(2 * K)), dim3(256), 0, st, d_coeffs, m, by, K, 2 * K);
    SCCHK(ntt_cols(by, bx, L + 1, 0, false, st));
    hipLaunchKernelGGL(pt_mul_scaled_kernel, dim3(pt_blocks(2 * K)), dim3(256), 0, st, (const Fe*)bx, (const Fe*)t->invg_f, bx, 2 * K, ninv_scaled(L + 1, 2));
    SCCHK(ntt_cols(bx, by, L + 1, 0, true, st));
    // down: cur = by[0..K) holds the series of the level-l nodes, [2^l][K >> l]
    for (int l = L; l >= 1; --l) {
        const uint64_t n = 1ull << l;
        const int logB = L - l;
        SCCHK(ntt_cols(by, tk, l, logB, false, st));
        hipLaunchKernelGGL(pt_corr_kernel, dim3(pt_blocks(2 * K)), dim3(256), 0, st, (const Fe*)tk, (const Fe*)(t->zf + (uint64_t)(l - 1) * 2 * K), bx, n, logB + 1,
                           ninv_scaled(l, 2));
        SCCHK(ntt_cols(bx, by, l, logB + 1, true, st));     // rows < n/2 = the first K entries = next level's series
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(d_out, by, K * sizeof(Fe), hipMemcpyDeviceToDevice, st));
    return SC_OK;
}

// d_points: the k points again (only read when m > K, for the chunk powers x^K)
int polytree_evaluate(sc_polytree* t, const Fe* d_coeffs, uint64_t m, const Fe* d_points, Fe* d_out, hipStream_t st) {
    const uint64_t K = t->K, k = t->k;
    PoolTmp bx, by, tk, all;
    SCCHK(bx.get(2 * K * sizeof(Fe)));
    SCCHK(by.get(2 * K * sizeof(Fe)));
    SCCHK(tk.get(K * sizeof(Fe)));
    SCCHK(all.get(K * sizeof(Fe)));
    if (m <= K) {
        SCCHK(polytree_evaluate_all(t, d_coeffs, m, all.fe(), bx.fe(), by.fe(), tk.fe(), st));
        HIPCHK(hipMemcpyAsync(d_out, all.p, k * sizeof(Fe), hipMemcpyDeviceToDevice, st));
    } else {
        if (!d_points) return fail(SC_ERR_BAD_ARG, "polynomial longer than the padded domain needs the points for chunked evaluation");
        PoolTmp y;
        SCCHK(y.get(k * sizeof(Fe)));
        hipLaunchKernelGGL(pt_pow2_kernel, dim3(pt_blocks(k)), dim3(256), 0, st, d_points, k, t->L, y.fe());
        const uint64_t chunks = (m + K - 1) / K;
        for (uint64_t j = chunks; j-- > 0;) {
            const uint64_t len = (j == chunks - 1) ? m - j * K : K;
            SCCHK(polytree_evaluate_all(t, d_coeffs + j * K, len, all.fe(), bx.fe(), by.fe(), tk.fe(), st));
            if (j == chunks - 1) HIPCHK(hipMemcpyAsync(d_out, all.p, k * sizeof(Fe), hipMemcpyDeviceToDevice, st));
            else hipLaunchKernelGGL(pt_horner_kernel, dim3(pt_blocks(k)), dim3(256), 0, st, d_out, (const Fe*)y.fe(), (const Fe*)all.fe(), k);
        }
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipStreamSynchronize(st));
    return SC_OK;
}

int polytree_interpolate(sc_polytree* t, const Fe* d_values, Fe* d_out, hipStream_t st) {
    const uint64_t K = t->K, k = t->k, pad = K - k;
    const int L = t->L;
    if (L == 0) {
        HIPCHK(hipMemcpyAsync(d_out, d_values, sizeof(Fe), hipMemcpyDeviceToDevice, st));
        HIPCHK(hipStreamSynchronize(st));
        return SC_OK;
    }
    PoolTmp bx, by, tk, p;
    SCCHK(bx.get(2 * K * sizeof(Fe)));
    SCCHK(by.get(2 * K * sizeof(Fe)));
    SCCHK(tk.get(K * sizeof(Fe)));
    SCCHK(p.get(K * sizeof(Fe)));
    const Fe* top = t->zc + (uint64_t)L * K;
    // weights w_i = v_i / Z_real'(d_i); padding leaves get weight 0
    hipLaunchKernelGGL(pt_deriv_kernel, dim3(pt_blocks(K)), dim3(256), 0, st, top, K, pad, k, p.fe());
    SCCHK(polytree_evaluate_all(t, p.fe(), k, tk.fe(), bx.fe(), by.fe(), p.fe(), st));     // p doubles as the K-entry temporary: its content is consumed first
    HIPCHK(hipMemsetAsync(p.p, 0, K * sizeof(Fe), st));
    SCCHK(pointwise_div_device(d_values, tk.fe(), p.fe(), k, st));
    // up: P = P_L * Z_R + P_R * Z_L
    Fe* cur = p.fe();
    Fe* nxt = tk.fe();
    for (int l = 0; l < L; ++l) {
        const uint64_t B = K >> l;
        hipLaunchKernelGGL(pt_expand_kernel, dim3(pt_blocks(2 * K)), dim3(256), 0, st, (const Fe*)cur, bx.fe(), K, B, (uint64_t)0);
        SCCHK(ntt_cols(bx.fe(), by.fe(), l + 1, L - l, false, st));
        hipLaunchKernelGGL(pt_comb_kernel, dim3(pt_blocks(K)), dim3(256), 0, st, (const Fe*)by.fe(), (const Fe*)(t->zf + (uint64_t)l * 2 * K), bx.fe(), K, ninv_scaled(l + 1, 2));
        SCCHK(ntt_cols(bx.fe(), nxt, l + 1, L - l - 1, true, st));
        Fe* s = cur; cur = nxt; nxt = s;
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(d_out, cur + pad, k * sizeof(Fe), hipMemcpyDeviceToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
    return SC_OK;
}

}  // namespace

// ============================================================================ geometric progressions (geoseq.cuh)

struct sc_geodomain {
    uint64_t n, M;
    int logM;
    Fe c, q, c_inv;        // first point, ratio, 1 / first point (canonical)
    bool unit;             // c == 1: no scaling by powers of c anywhere
    Fe* tinv_m;            // n:  q^-(j(j-1)/2)
    Fe* wden_m;            // n:  1 / (Z'(q^i) t_i)
    Fe* Bf;                // M:  transform of t_0 .. t_(M-1)
    Fe* ZRf;               // M:  transform of the reversed zerofier's first n coefficients
    Fe* zr;                // n + 1: reversed zerofier of {q^i}, canonical
};

namespace {

// inclusive prefix products of n Montgomery forms, in place (reduce per workgroup, scan the totals, apply)
int scan_products(Fe* a, uint64_t n, hipStream_t st) {
    if (n == 0) return SC_OK;
    const uint64_t nb = (n + GS_BLOCK - 1) / GS_BLOCK;
    if (nb == 1) {
        hipLaunchKernelGGL(gs_apply_kernel, dim3(1), dim3(GS_T), 0, st, a, n, (const Fe*)nullptr);
        HIPCHK(hipGetLastError());
        return SC_OK;
    }
    PoolTmpAsync tot;
    SCCHK(tot.get(nb * sizeof(Fe)));
    hipLaunchKernelGGL(gs_totals_kernel, dim3((unsigned)nb), dim3(GS_T), 0, st, (const Fe*)a, n, tot.fe());
    SCCHK(scan_products(tot.fe(), nb, st));
    hipLaunchKernelGGL(gs_apply_kernel, dim3((unsigned)nb), dim3(GS_T), 0, st, a, n, (const Fe*)tot.fe());
    HIPCHK(hipGetLastError());
    return SC_OK;
}

void geodomain_release(sc_geodomain* d) {
    if (!d) return;
    if (d->tinv_m) release_after_streams(d->tinv_m, d->n * sizeof(Fe));
    if (d->wden_m) release_after_streams(d->wden_m, d->n * sizeof(Fe));
    if (d->Bf) release_after_streams(d->Bf, d->M * sizeof(Fe));
    if (d->ZRf) release_after_streams(d->ZRf, d->M * sizeof(Fe));
    if (d->zr) release_after_streams(d->zr, (d->n + 1) * sizeof(Fe));
    delete d;
}

int geodomain_create(Fe c, Fe q, uint64_t n, sc_geodomain** out, hipStream_t st) {
    if (n < 2) return fail(SC_ERR_UNSUPPORTED, "a progression of fewer than two points");
    if (fe_ge_p(c) || fe_ge_p(q) || fe_is_zero(c) || fe_is_zero(q)) return fail(SC_ERR_UNSUPPORTED, "first point and ratio must be non-zero residues");
    const int logM = ilog2(2 * n - 1);
    if (logM > 28) return fail(SC_ERR_UNSUPPORTED, "progression too long");
    const uint64_t M = 1ull << logM;
    sc_geodomain* d = new sc_geodomain{n, M, logM, c, q, Fe{0, 0}, fe_eq(c, fe_one()), nullptr, nullptr, nullptr, nullptr, nullptr};
    auto bail = [&](int rc) { geodomain_release(d); return rc; };
    auto alloc = [&](Fe** p, uint64_t count) -> int {
        hipError_t e = pool_alloc((void**)p, count * sizeof(Fe));
        return e == hipSuccess ? SC_OK : fail(SC_ERR_HIP, hipGetErrorString(e));
    };
    int rc = alloc(&d->tinv_m, n);
    if (rc == SC_OK) rc = alloc(&d->wden_m, n);
    if (rc == SC_OK) rc = alloc(&d->Bf, M);
    if (rc == SC_OK) rc = alloc(&d->ZRf, M);
    if (rc == SC_OK) rc = alloc(&d->zr, n + 1);
    if (rc != SC_OK) return bail(rc);
    const Fe q_m = to_mont(q);
    const Fe qinv = from_mont(mont_inv(q_m));
    d->c_inv = from_mont(mont_inv(to_mont(c)));
    PoolTmpAsync tt, A, rev;
    if ((rc = tt.get(M * sizeof(Fe))) != SC_OK || (rc = A.get(n * sizeof(Fe))) != SC_OK || (rc = rev.get(n * sizeof(Fe))) != SC_OK) return bail(rc);
    PowTables *pq, *pqi, *pg;
    if ((rc = get_pow(q, M, st, &pq)) != SC_OK) return bail(rc);
    if ((rc = get_pow(qinv, n, st, &pqi)) != SC_OK) return bail(rc);
    // t_j (M of them), 1 / t_j (n), A_(j+1) (n), S_(n-2-j) (n - 1): one fill and one scan each
    hipLaunchKernelGGL(geo_fill_kernel, dim3(pt_blocks(M)), dim3(256), 0, st, tt.fe(), M, 0, n, (const Fe*)pq->lo, (const Fe*)pq->hi);
    if ((rc = scan_products(tt.fe(), M, st)) != SC_OK) return bail(rc);
    hipLaunchKernelGGL(geo_fill_kernel, dim3(pt_blocks(n)), dim3(256), 0, st, d->tinv_m, n, 0, n, (const Fe*)pqi->lo, (const Fe*)pqi->hi);
    if ((rc = scan_products(d->tinv_m, n, st)) != SC_OK) return bail(rc);
    hipLaunchKernelGGL(geo_fill_kernel, dim3(pt_blocks(n)), dim3(256), 0, st, A.fe(), n, 1, n, (const Fe*)pq->lo, (const Fe*)pq->hi);
    if ((rc = scan_products(A.fe(), n, st)) != SC_OK) return bail(rc);
    hipLaunchKernelGGL(geo_fill_kernel, dim3(pt_blocks(n - 1)), dim3(256), 0, st, rev.fe(), n - 1, 2, n, (const Fe*)pq->lo, (const Fe*)pq->hi);
    if ((rc = scan_products(rev.fe(), n - 1, st)) != SC_OK) return bail(rc);
    // scan[j] = A_(j+1).  A_(n-1) and A_n decide: a zero A_(n-1) means q^m = 1 for some m < n, i.e. the points repeat
    Fe tail[2];
    if (hipMemcpyAsync(tail, A.fe() + (n - 2), 2 * sizeof(Fe), hipMemcpyDeviceToHost, st) != hipSuccess) return bail(fail(SC_ERR_HIP, "copy of the scan's tail failed"));
    if (hipStreamSynchronize(st) != hipSuccess) return bail(fail(SC_ERR_HIP, "progression tables failed"));
    const Fe an1_m = tail[0], an_m = tail[1];
    if (fe_is_zero(an1_m)) return bail(fail(SC_ERR_UNSUPPORTED, "the points of the progression are not distinct"));
    const Fe ia_m = mont_inv(an1_m);
    const Fe k1_m = mont_mul(ia_m, ia_m);
    const Fe k2_m = mont_mul(an_m, k1_m);
    const Fe g = from_mont(mont_pow(to_mont(qinv), n - 2));
    if ((rc = get_pow(g, n, st, &pg)) != SC_OK) return bail(rc);
    hipLaunchKernelGGL(geo_wden_kernel, dim3(pt_blocks(n)), dim3(256), 0, st, d->wden_m, n, (const Fe*)rev.fe(), k1_m, (const Fe*)pg->lo, (const Fe*)pg->hi);
    hipLaunchKernelGGL(geo_zr_kernel, dim3(pt_blocks(n + 1)), dim3(256), 0, st, d->zr, n, (const Fe*)rev.fe(), (const Fe*)tt.fe(), k2_m);
    hipLaunchKernelGGL(geo_from_mont_kernel, dim3(pt_blocks(M)), dim3(256), 0, st, tt.fe(), M);
    if (hipGetLastError() != hipSuccess) return bail(fail(SC_ERR_HIP, "progression table launch failed"));
    const Fe rt = canonical_root(logM);
    if ((rc = ntt_device(tt.fe(), d->Bf, logM, rt, false, NttOpts(), st)) != SC_OK) return bail(rc);
    NttOpts o;
    o.in_limit = n;
    if ((rc = ntt_device(d->zr, d->ZRf, logM, rt, false, o, st)) != SC_OK) return bail(rc);
    *out = d;
    return SC_OK;
}

const PowTables* geo_cpow(const sc_geodomain* d, Fe base, uint64_t count, hipStream_t st, int* rc) {
    if (d->unit) { *rc = SC_OK; return nullptr; }
    PowTables* pw = nullptr;
    *rc = get_pow(base, count, st, &pw);
    return pw;
}

// values at all n points of the polynomial p[0..len), len <= n; bx, by: M entries each
int geodomain_evaluate_chunk(const sc_geodomain* d, const Fe* p, uint64_t len, Fe* dst, Fe* bx, Fe* by, hipStream_t st) {
    const uint64_t n = d->n, M = d->M;
    if (len == 0) { HIPCHK(hipMemsetAsync(dst, 0, n * sizeof(Fe), st)); return SC_OK; }
    int rc;
    const PowTables* pc = geo_cpow(d, d->c, n, st, &rc);
    SCCHK(rc);
    hipLaunchKernelGGL(geo_eval_in_kernel, dim3(pt_blocks(len)), dim3(256), 0, st, p, len, pc ? (const Fe*)pc->lo : nullptr, pc ? (const Fe*)pc->hi : nullptr, (const Fe*)d->tinv_m, bx);
    const Fe rt = canonical_root(d->logM);
    NttOpts o;
    o.in_limit = len;
    SCCHK(ntt_device(bx, by, d->logM, rt, false, o, st));
    hipLaunchKernelGGL(geo_corr_kernel, dim3(pt_blocks(M)), dim3(256), 0, st, (const Fe*)by, (const Fe*)d->Bf, bx, M, ninv_scaled(d->logM, 2));
    SCCHK(ntt_device(bx, by, d->logM, root_inverse(rt, M), false, NttOpts(), st));
    hipLaunchKernelGGL(geo_mul_tab_kernel, dim3(pt_blocks(n)), dim3(256), 0, st, (const Fe*)by, (const Fe*)d->tinv_m, dst, n);
    HIPCHK(hipGetLastError());
    return SC_OK;
}

int geodomain_evaluate(const sc_geodomain* d, const Fe* coeffs, uint64_t m, Fe* out, hipStream_t st) {
    const uint64_t n = d->n, M = d->M;
    PoolTmpAsync bx, by;
    SCCHK(bx.get(M * sizeof(Fe)));
    SCCHK(by.get(M * sizeof(Fe)));
    if (m <= n) return geodomain_evaluate_chunk(d, coeffs, m, out, bx.fe(), by.fe(), st);
    // longer polynomials: Horner over chunks of n coefficients, y_i = x_i^n = c^n (q^n)^i
    PoolTmpAsync y, vals;
    SCCHK(y.get(n * sizeof(Fe)));
    SCCHK(vals.get(n * sizeof(Fe)));
    const Fe cn_m = mont_pow(to_mont(d->c), n);
    const Fe qn = from_mont(mont_pow(to_mont(d->q), n));
    PowTables* py;
    SCCHK(get_pow(qn, n, st, &py));
    hipLaunchKernelGGL(geo_chunk_power_kernel, dim3(pt_blocks(n)), dim3(256), 0, st, y.fe(), n, cn_m, (const Fe*)py->lo, (const Fe*)py->hi);
    const uint64_t chunks = (m + n - 1) / n;
    for (uint64_t j = chunks; j-- > 0;) {
        const uint64_t len = (j == chunks - 1) ? m - j * n : n;
        if (j == chunks - 1) { SCCHK(geodomain_evaluate_chunk(d, coeffs + j * n, len, out, bx.fe(), by.fe(), st)); continue; }
        SCCHK(geodomain_evaluate_chunk(d, coeffs + j * n, len, vals.fe(), bx.fe(), by.fe(), st));
        hipLaunchKernelGGL(pt_horner_kernel, dim3(pt_blocks(n)), dim3(256), 0, st, out, (const Fe*)y.fe(), (const Fe*)vals.fe(), n);
    }
    HIPCHK(hipGetLastError());
    return SC_OK;
}

// is d_points[i + 1] == d_points[i] * ratio for all i < n - 1, with ratio = points[1] / points[0]?  (one small kernel, one sync)
int geodomain_detect(const Fe* d_points, uint64_t n, Fe* first, Fe* ratio, bool* is_geometric, hipStream_t st) {
    *is_geometric = false;
    if (n < 2) return SC_OK;
    Fe head[2];
    SCCHK(read_small_polled(d_points, 2 * sizeof(Fe), st, head));
    if (fe_is_zero(head[0]) || fe_is_zero(head[1]) || fe_ge_p(head[0]) || fe_ge_p(head[1])) return SC_OK;
    const Fe r_m = mont_mul(to_mont(head[1]), mont_inv(to_mont(head[0])));
    void* fl;
    SCCHK(scratch(7, 64, &fl));
    HIPCHK(hipMemsetAsync(fl, 0, 4, st));
    hipLaunchKernelGGL(geo_detect_kernel, dim3(pt_blocks(n)), dim3(256), 0, st, d_points, n, r_m, (uint32_t*)fl);
    HIPCHK(hipGetLastError());
    uint64_t bad = 1;
    SCCHK(read_small_polled(fl, 8, st, &bad));
    if ((uint32_t)bad) return SC_OK;
    *first = head[0];
    *ratio = from_mont(r_m);
    *is_geometric = true;
    return SC_OK;
}

// the progression tables of device points if they are a progression of at least two distinct points, else *out = nullptr
int geodomain_of_points(const Fe* d_points, uint64_t n, sc_geodomain** out, hipStream_t st) {
    *out = nullptr;
    Fe first, ratio;
    bool is = false;
    SCCHK(geodomain_detect(d_points, n, &first, &ratio, &is, st));
    if (!is) return SC_OK;
    int rc = geodomain_create(first, ratio, n, out, st);
    if (rc == SC_ERR_UNSUPPORTED) { *out = nullptr; return SC_OK; }
    return rc;
}

int geodomain_interpolate(const sc_geodomain* d, const Fe* values, Fe* out, hipStream_t st) {
    const uint64_t n = d->n, M = d->M;
    PoolTmpAsync bx, by;
    SCCHK(bx.get(M * sizeof(Fe)));
    SCCHK(by.get(M * sizeof(Fe)));
    int rc;
    const PowTables* pi = geo_cpow(d, d->c_inv, n, st, &rc);
    SCCHK(rc);
    const Fe rt = canonical_root(d->logM), rti = root_inverse(rt, M);
    const Fe c_m2 = ninv_scaled(d->logM, 2);
    NttOpts o;
    o.in_limit = n;
    // s_m = sum_i (v_i / Z'(q^i)) q^(i m): weights, correlation with t, division by t_m
    hipLaunchKernelGGL(geo_mul_tab_kernel, dim3(pt_blocks(n)), dim3(256), 0, st, values, (const Fe*)d->wden_m, bx.fe(), n);
    SCCHK(ntt_device(bx.fe(), by.fe(), d->logM, rt, false, o, st));
    hipLaunchKernelGGL(geo_corr_kernel, dim3(pt_blocks(M)), dim3(256), 0, st, (const Fe*)by.fe(), (const Fe*)d->Bf, bx.fe(), M, c_m2);
    SCCHK(ntt_device(bx.fe(), by.fe(), d->logM, rti, false, NttOpts(), st));
    hipLaunchKernelGGL(geo_mul_tab_kernel, dim3(pt_blocks(n)), dim3(256), 0, st, (const Fe*)by.fe(), (const Fe*)d->tinv_m, bx.fe(), n);
    // rev(P) = rev(Z) * S mod y^n
    SCCHK(ntt_device(bx.fe(), by.fe(), d->logM, rt, false, o, st));
    hipLaunchKernelGGL(pt_mul_scaled_kernel, dim3(pt_blocks(M)), dim3(256), 0, st, (const Fe*)by.fe(), (const Fe*)d->ZRf, bx.fe(), M, c_m2);
    SCCHK(ntt_device(bx.fe(), by.fe(), d->logM, rti, false, NttOpts(), st));
    hipLaunchKernelGGL(geo_rev_scale_kernel, dim3(pt_blocks(n)), dim3(256), 0, st, (const Fe*)by.fe(), n, pi ? (const Fe*)pi->lo : nullptr, pi ? (const Fe*)pi->hi : nullptr, out);
    HIPCHK(hipGetLastError());
    return SC_OK;
}

}  // namespace

extern "C" {

// ---- subproduct tree: fast_zerofier / fast_evaluate / fast_interpolate
int sc_polytree_build_dev(const void* d_points, uint64_t k, sc_polytree_t** tree, void* stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    if (!tree || !d_points) return fail(SC_ERR_BAD_ARG, "null argument");
    return polytree_build((const Fe*)d_points, k, tree, pick_stream(stream));
}
int sc_polytree_build(const void* points, uint64_t k, sc_polytree_t** tree) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    if (!tree || !points || !k) return fail(SC_ERR_BAD_ARG, "empty domain");
    PoolTmp dp;
    SCCHK(dp.get(k * sizeof(Fe)));
    SCCHK(upload(dp.p, points, k * sizeof(Fe), g.stream));
    return polytree_build(dp.fe(), k, tree, g.stream);       // synchronises before returning: dp may go back to the pool
}
uint64_t sc_polytree_points(const sc_polytree_t* tree) { return tree ? tree->k : 0; }
int sc_polytree_zerofier_dev(const sc_polytree_t* tree, void* d_out, void* stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    if (!tree || !d_out) return fail(SC_ERR_BAD_ARG, "null argument");
    hipStream_t st = pick_stream(stream);
    hipLaunchKernelGGL(pt_zerofier_out_kernel, dim3(pt_blocks(tree->k + 1)), dim3(256), 0, st, (const Fe*)(tree->zc + (uint64_t)tree->L * tree->K), tree->K,
                       tree->K - tree->k, tree->k, (Fe*)d_out);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    return SC_OK;
}
int sc_polytree_evaluate_dev(sc_polytree_t* tree, const void* d_coeffs, uint64_t m, const void* d_points, void* d_out, void* stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    if (!tree || !d_out || (m && !d_coeffs)) return fail(SC_ERR_BAD_ARG, "null argument");
    return polytree_evaluate(tree, (const Fe*)d_coeffs, m, (const Fe*)d_points, (Fe*)d_out, pick_stream(stream));
}
int sc_polytree_interpolate_dev(sc_polytree_t* tree, const void* d_values, void* d_out, void* stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    if (!tree || !d_out || !d_values) return fail(SC_ERR_BAD_ARG, "null argument");
    return polytree_interpolate(tree, (const Fe*)d_values, (Fe*)d_out, pick_stream(stream));
}
int sc_polytree_free(sc_polytree_t* tree) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!tree) return SC_OK;
    release_after_streams(tree->zc, tree->zc_bytes);
    release_after_streams(tree->zf, tree->zf_bytes);
    if (tree->invg_f) release_after_streams(tree->invg_f, 2 * tree->K * sizeof(Fe));
    delete tree;
    return SC_OK;
}

// host-buffer forms of ntt.py:66-80, :82-100, :102-130
// host points -> their progression tables (nullptr when they are not a progression of >= 2 distinct points: the tree serves those)
static int host_points_progression(const void* points, uint64_t k, sc_geodomain** gd) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    *gd = nullptr;
    if (!points) return fail(SC_ERR_BAD_ARG, "null argument");
    if (k < 2) return SC_OK;
    PoolTmp dp;
    SCCHK(dp.get(k * sizeof(Fe)));
    SCCHK(upload(dp.p, points, k * sizeof(Fe), g.stream));
    int rc = geodomain_of_points(dp.fe(), k, gd, g.stream);
    if (hipStreamSynchronize(g.stream) != hipSuccess && rc == SC_OK) rc = fail(SC_ERR_HIP, "progression tables failed");   // dp goes back to the pool
    return rc;
}

int sc_zerofier(const void* points, uint64_t k, void* out) {
    if (k == 0) return SC_OK;
    sc_geodomain* gd = nullptr;
    SCCHK(host_points_progression(points, k, &gd));
    if (gd) {
        std::lock_guard<std::mutex> lk(g_mu);
        PoolTmp d;
        int rc = d.get((k + 1) * sizeof(Fe));
        if (rc == SC_OK) {
            const PowTables* pc = geo_cpow(gd, gd->c, k + 1, g.stream, &rc);
            if (rc == SC_OK) {
                hipLaunchKernelGGL(geo_zerofier_out_kernel, dim3(pt_blocks(k + 1)), dim3(256), 0, g.stream, (const Fe*)gd->zr, k, pc ? (const Fe*)pc->lo : nullptr,
                                   pc ? (const Fe*)pc->hi : nullptr, d.fe());
                rc = download(out, d.p, (k + 1) * sizeof(Fe), g.stream);
            }
        }
        geodomain_release(gd);
        return rc;
    }
    sc_polytree_t* t = nullptr;
    SCCHK(sc_polytree_build(points, k, &t));
    int rc;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        PoolTmp d;
        rc = d.get((k + 1) * sizeof(Fe));
        if (rc == SC_OK) {
            hipLaunchKernelGGL(pt_zerofier_out_kernel, dim3(pt_blocks(k + 1)), dim3(256), 0, g.stream, (const Fe*)(t->zc + (uint64_t)t->L * t->K), t->K, t->K - k, k, d.fe());
            rc = download(out, d.p, (k + 1) * sizeof(Fe), g.stream);
        }
    }
    sc_polytree_free(t);
    return rc;
}
int sc_evaluate(const void* coeffs, uint64_t m, const void* points, uint64_t k, void* out) {
    if (k == 0) return SC_OK;
    sc_geodomain* gd = nullptr;
    SCCHK(host_points_progression(points, k, &gd));
    if (gd) {
        std::lock_guard<std::mutex> lk(g_mu);
        PoolTmp dc, dv;
        int rc = dc.get((m ? m : 1) * sizeof(Fe));
        if (rc == SC_OK) rc = dv.get(k * sizeof(Fe));
        if (rc == SC_OK) rc = upload(dc.p, coeffs, m * sizeof(Fe), g.stream);
        if (rc == SC_OK) rc = geodomain_evaluate(gd, dc.fe(), m, dv.fe(), g.stream);
        if (rc == SC_OK) rc = download(out, dv.p, k * sizeof(Fe), g.stream);
        else (void)hipStreamSynchronize(g.stream);
        geodomain_release(gd);
        return rc;
    }
    sc_polytree_t* t = nullptr;
    SCCHK(sc_polytree_build(points, k, &t));
    int rc;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        PoolTmp dc, dp, dv;
        rc = dc.get((m ? m : 1) * sizeof(Fe));
        if (rc == SC_OK) rc = dp.get(k * sizeof(Fe));
        if (rc == SC_OK) rc = dv.get(k * sizeof(Fe));
        if (rc == SC_OK) rc = upload(dc.p, coeffs, m * sizeof(Fe), g.stream);
        if (rc == SC_OK) rc = upload(dp.p, points, k * sizeof(Fe), g.stream);
        if (rc == SC_OK) rc = polytree_evaluate(t, dc.fe(), m, dp.fe(), dv.fe(), g.stream);
        if (rc == SC_OK) rc = download(out, dv.p, k * sizeof(Fe), g.stream);
    }
    sc_polytree_free(t);
    return rc;
}
int sc_interpolate(const void* points, const void* values, uint64_t k, void* out) {
    if (k == 0) return SC_OK;
    sc_geodomain* gd = nullptr;
    SCCHK(host_points_progression(points, k, &gd));
    if (gd) {
        std::lock_guard<std::mutex> lk(g_mu);
        PoolTmp dv, dout;
        int rc = dv.get(k * sizeof(Fe));
        if (rc == SC_OK) rc = dout.get(k * sizeof(Fe));
        if (rc == SC_OK) rc = upload(dv.p, values, k * sizeof(Fe), g.stream);
        if (rc == SC_OK) rc = geodomain_interpolate(gd, dv.fe(), dout.fe(), g.stream);
        if (rc == SC_OK) rc = download(out, dout.p, k * sizeof(Fe), g.stream);
        else (void)hipStreamSynchronize(g.stream);
        geodomain_release(gd);
        return rc;
    }
    sc_polytree_t* t = nullptr;
    SCCHK(sc_polytree_build(points, k, &t));
    int rc;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        PoolTmp dv, dout;
        rc = dv.get(k * sizeof(Fe));
        if (rc == SC_OK) rc = dout.get(k * sizeof(Fe));
        if (rc == SC_OK) rc = upload(dv.p, values, k * sizeof(Fe), g.stream);
        if (rc == SC_OK) rc = polytree_interpolate(t, dv.fe(), dout.fe(), g.stream);
        if (rc == SC_OK) rc = download(out, dout.p, k * sizeof(Fe), g.stream);
    }
    sc_polytree_free(t);
    return rc;
}

// ---- geometric progressions: fast_zerofier / fast_evaluate / fast_interpolate (ntt.py:66-130) on {first * ratio^i} ------------
int sc_geodomain_create(const uint64_t first[2], const uint64_t ratio[2], uint64_t n, sc_geodomain_t** domain, void* stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    if (!first || !ratio || !domain) return fail(SC_ERR_BAD_ARG, "null argument");
    return geodomain_create(fe_from(first), fe_from(ratio), n, domain, pick_stream(stream));
}
uint64_t sc_geodomain_points(const sc_geodomain_t* domain) { return domain ? domain->n : 0; }
int sc_geodomain_detect_dev(const void* d_points, uint64_t n, uint64_t first[2], uint64_t ratio[2], int* is_geometric, void* stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    if (!d_points || !first || !ratio || !is_geometric) return fail(SC_ERR_BAD_ARG, "null argument");
    Fe f{0, 0}, r{0, 0};
    bool is = false;
    SCCHK(geodomain_detect((const Fe*)d_points, n, &f, &r, &is, pick_stream(stream)));
    *is_geometric = is ? 1 : 0;
    if (is) { first[0] = f.lo; first[1] = f.hi; ratio[0] = r.lo; ratio[1] = r.hi; }
    return SC_OK;
}
int sc_geodomain_zerofier_dev(const sc_geodomain_t* domain, void* d_out, void* stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    if (!domain || !d_out) return fail(SC_ERR_BAD_ARG, "null argument");
    hipStream_t st = pick_stream(stream);
    int rc;
    const PowTables* pc = geo_cpow(domain, domain->c, domain->n + 1, st, &rc);
    SCCHK(rc);
    hipLaunchKernelGGL(geo_zerofier_out_kernel, dim3(pt_blocks(domain->n + 1)), dim3(256), 0, st, (const Fe*)domain->zr, domain->n, pc ? (const Fe*)pc->lo : nullptr,
                       pc ? (const Fe*)pc->hi : nullptr, (Fe*)d_out);
    HIPCHK(hipGetLastError());
    return SC_OK;
}
int sc_geodomain_evaluate_dev(const sc_geodomain_t* domain, const void* d_coeffs, uint64_t m, void* d_out, void* stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    if (!domain || !d_out || (m && !d_coeffs)) return fail(SC_ERR_BAD_ARG, "null argument");
    return geodomain_evaluate(domain, (const Fe*)d_coeffs, m, (Fe*)d_out, pick_stream(stream));
}
int sc_geodomain_interpolate_dev(const sc_geodomain_t* domain, const void* d_values, void* d_out, void* stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    if (!domain || !d_values || !d_out) return fail(SC_ERR_BAD_ARG, "null argument");
    return geodomain_interpolate(domain, (const Fe*)d_values, (Fe*)d_out, pick_stream(stream));
}
int sc_geodomain_free(sc_geodomain_t* domain) {
    std::lock_guard<std::mutex> lk(g_mu);
    geodomain_release(domain);
    return SC_OK;
}


}  // extern "C"
