/*
 * stark_oracle.c -- CPU restatement of the stark-anatomy polynomial hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (stark-anatomy_amd/, include/) may link,
 * import or call this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do,
 * and there only as the checker / reported baseline.
 *
 * Parity pin: every function here is checked in tests/test_oracle.py against golden vectors produced
 * by importing the reference (tests/golden/make_golden.py -> the JSON fixtures in tests/golden), and the BLAKE2b
 * code against CPython's hashlib (the reference's own dependency, code/merkle.py:1).
 *
 * Element layout: 16 bytes = two little-endian uint64 limbs (lo, hi), canonical residue in [0, p),
 * p = 1 + 407 * 2^119 (reference code/algebra.py:96-98).
 *
 * Arithmetic is deliberately NOT Montgomery (the device code is): products are reduced with the
 * identity 407 * 2^119 == -1 (mod p), so the oracle and the kernels cannot share a reduction bug.
 */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef uint64_t u64;

#define P_HI 0xCB80000000000000ULL
#define P_LO 1ULL
static const u128 P = ((u128)P_HI << 64) | P_LO;

/* ---------------------------------------------------------------- field (code/algebra.py:65-120) */

/* Field.add, algebra.py:78-79 */
static inline u128 f_add(u128 a, u128 b) {
    u128 s = a + b;               /* may wrap: a,b < p < 2^128, a+b < 2^129 */
    int carry = s < a;
    if (carry || s >= P) s -= P;
    return s;
}

/* Field.subtract, algebra.py:81-82 */
static inline u128 f_sub(u128 a, u128 b) { return a >= b ? a - b : a + (P - b); }

/* Field.negate, algebra.py:84-85 */
static inline u128 f_neg(u128 a) { return a ? P - a : 0; }

/* Field.multiply, algebra.py:75-76: (a*b) % p via the 256-bit product and 407*2^119 == -1. */
static u128 f_mul(u128 a, u128 b) {
    u64 a0 = (u64)a, a1 = (u64)(a >> 64), b0 = (u64)b, b1 = (u64)(b >> 64);
    u128 p00 = (u128)a0 * b0, p01 = (u128)a0 * b1, p10 = (u128)a1 * b0, p11 = (u128)a1 * b1;
    u64 t[4];
    t[0] = (u64)p00;
    u128 mid = (p00 >> 64) + (u64)p01 + (u64)p10;
    t[1] = (u64)mid;
    u128 hi = (mid >> 64) + (p01 >> 64) + (p10 >> 64) + (u64)p11;
    t[2] = (u64)hi;
    t[3] = (u64)((hi >> 64) + (p11 >> 64));
    /* T = Thi * 2^119 + Tlo */
    u128 tlo = (((u128)(t[1] & ((1ULL << 55) - 1))) << 64) | t[0];          /* low 119 bits */
    u64 h0 = (t[1] >> 55) | (t[2] << 9);
    u64 h1 = (t[2] >> 55) | (t[3] << 9);
    u64 h2 = (t[3] >> 55);                                                    /* < 2^9 */
    /* Thi = q*407 + r */
    u128 cur = h2;
    u64 q2 = (u64)(cur / 407); u64 rem = (u64)(cur % 407);
    cur = ((u128)rem << 64) | h1;
    u64 q1 = (u64)(cur / 407); rem = (u64)(cur % 407);
    cur = ((u128)rem << 64) | h0;
    u64 q0 = (u64)(cur / 407); rem = (u64)(cur % 407);
    if (q2 != 0) abort();                                                     /* q <= p-1 < 2^128 */
    u128 q = ((u128)q1 << 64) | q0;
    u128 A = ((u128)rem << 119) + tlo;                                        /* < 407*2^119 = p-1 */
    /* T == A - q (mod p), q <= p-1 */
    return A >= q ? A - q : A + (P - q);
}

/* FieldElement.__xor__, algebra.py:38-45 (square-and-multiply, MSB first) */
static u128 f_pow(u128 a, u128 e) {
    u128 acc = 1;
    for (int i = 127; i >= 0; --i) {
        acc = f_mul(acc, acc);
        if ((e >> i) & 1) acc = f_mul(acc, a);
    }
    return acc;
}

/* Field.inverse, algebra.py:87-89: xgcd inverse; inverse(0) == 0.  Fermat gives the same residues. */
static u128 f_inv(u128 a) { return f_pow(a, P - 2); }

static inline u128 ld(const u64* p) { return ((u128)p[1] << 64) | p[0]; }
static inline void st(u64* p, u128 v) { p[0] = (u64)v; p[1] = (u64)(v >> 64); }

void so_add(const u64* a, const u64* b, u64* out) { st(out, f_add(ld(a), ld(b))); }
void so_sub(const u64* a, const u64* b, u64* out) { st(out, f_sub(ld(a), ld(b))); }
void so_mul(const u64* a, const u64* b, u64* out) { st(out, f_mul(ld(a), ld(b))); }
void so_inv(const u64* a, u64* out) { st(out, f_inv(ld(a))); }
void so_pow(const u64* a, const u64* e, u64* out) { st(out, f_pow(ld(a), ld(e))); }

/* ---------------------------------------------------------------- synthetic input */
static inline u64 splitmix64(u64 x) {
    u64 z = x + 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

void so_synth(u64 seed, u64 start, u64 n, u64* out) {
    for (u64 i = 0; i < n; ++i) {
        u64 k = start + i;
        u128 x = ((u128)splitmix64(seed + 2 * k + 1) << 64) | splitmix64(seed + 2 * k);
        if (x >= P) x -= P;
        st(out + 2 * i, x);
    }
}

/* ---------------------------------------------------------------- ntt (code/ntt.py:3-18) */
/* Recursive radix-2 decimation in time exactly as ntt.py:15-18:
 *   odds = ntt(root^2, values[1::2]); evens = ntt(root^2, values[::2]);
 *   out[i] = evens[i % half] + root^i * odds[i % half]
 * root^i is carried as a running product (same residues as the reference's per-element pow). */
static void ntt_rec(u128 root, const u128* in, u64 stride, u128* out, u64 n, u128* scratch) {
    if (n == 1) { out[0] = in[0]; return; }
    u64 half = n / 2;
    u128 r2 = f_mul(root, root);
    u128* evens = scratch;
    u128* odds = scratch + half;
    ntt_rec(r2, in, stride * 2, evens, half, scratch + n);
    ntt_rec(r2, in + stride, stride * 2, odds, half, scratch + n);
    u128 w = 1;
    for (u64 i = 0; i < n; ++i) {
        out[i] = f_add(evens[i % half], f_mul(w, odds[i % half]));
        w = f_mul(w, root);
    }
}

/* returns 0 ok; -1 n not a power of two (ntt.py:4); -2 root^n != 1 (ntt.py:10); -3 root not primitive (ntt.py:11) */
int so_ntt(const u64* root_, const u64* in, u64* out, u64 n) {
    if (n & (n - 1)) return -1;
    if (n <= 1) { if (n == 1 && out != in) memcpy(out, in, 16); return 0; }
    u128 root = ld(root_);
    if (f_pow(root, n) != 1) return -2;
    if (f_pow(root, n / 2) == 1) return -3;
    u128* x = (u128*)malloc(sizeof(u128) * n);
    u128* y = (u128*)malloc(sizeof(u128) * n);
    u128* scratch = (u128*)malloc(sizeof(u128) * 2 * n);
    for (u64 i = 0; i < n; ++i) x[i] = ld(in + 2 * i);
    ntt_rec(root, x, 1, y, n, scratch);
    for (u64 i = 0; i < n; ++i) st(out + 2 * i, y[i]);
    free(x); free(y); free(scratch);
    return 0;
}

/* intt, ntt.py:20-30: ntt(root^-1, values) scaled by n^-1 */
int so_intt(const u64* root_, const u64* in, u64* out, u64 n) {
    if (n & (n - 1)) return -1;
    if (n == 1) { if (out != in) memcpy(out, in, 16); return 0; }
    if (n == 0) return 0;
    u64 rinv[2];
    st(rinv, f_inv(ld(root_)));
    int rc = so_ntt(rinv, in, out, n);
    if (rc) return rc;
    u128 ninv = f_inv((u128)n);
    for (u64 i = 0; i < n; ++i) st(out + 2 * i, f_mul(ninv, ld(out + 2 * i)));
    return 0;
}

/* Polynomial.scale, code/univariate.py:153-154: coeff[i] * factor^i */
void so_scale(const u64* in, u64 n, const u64* factor, u64* out) {
    u128 f = ld(factor), w = 1;
    for (u64 i = 0; i < n; ++i) { st(out + 2 * i, f_mul(w, ld(in + 2 * i))); w = f_mul(w, f); }
}

/* fast_coset_evaluate, ntt.py:132-135: scale by offset, zero-pad to order, ntt(generator, ...) */
int so_coset_evaluate(const u64* coeffs, u64 m, const u64* offset, const u64* generator, u64 order, u64* out) {
    if (m > order) return -4;
    u64* tmp = (u64*)calloc(order ? order : 1, 16);
    so_scale(coeffs, m, offset, tmp);
    int rc = so_ntt(generator, tmp, out, order);
    free(tmp);
    return rc;
}

/* Hadamard product, ntt.py:61 */
void so_pointwise_mul(const u64* a, const u64* b, u64 n, u64* out) {
    for (u64 i = 0; i < n; ++i) st(out + 2 * i, f_mul(ld(a + 2 * i), ld(b + 2 * i)));
}

/* pointwise field division, ntt.py:172 (Field.divide asserts a non-zero divisor, algebra.py:91-94) */
int so_pointwise_div(const u64* a, const u64* b, u64 n, u64* out) {
    for (u64 i = 0; i < n; ++i) {
        u128 d = ld(b + 2 * i);
        if (d == 0) return -5;
        st(out + 2 * i, f_mul(ld(a + 2 * i), f_inv(d)));
    }
    return 0;
}

/* split-and-fold, code/fri.py:85:
 *   c'[i] = 2^-1 * ( (1 + alpha/(offset*omega^i)) * c[i] + (1 - alpha/(offset*omega^i)) * c[N/2+i] )
 * 1/(offset*omega^i) is carried as the geometric sequence offset^-1 * (omega^-1)^i. */
int so_fold(const u64* in, u64 N, const u64* alpha_, const u64* offset_, const u64* omega_, u64* out) {
    if (N < 2 || (N & (N - 1))) return -1;
    u128 alpha = ld(alpha_), two_inv = f_inv(2);
    u128 x_inv = f_inv(ld(offset_)), w_inv = f_inv(ld(omega_));
    if (ld(offset_) == 0 || ld(omega_) == 0) return -5;
    for (u64 i = 0; i < N / 2; ++i) {
        u128 t = f_mul(alpha, x_inv);
        u128 a = ld(in + 2 * i), b = ld(in + 2 * (N / 2 + i));
        u128 v = f_add(f_mul(f_add(1, t), a), f_mul(f_sub(1, t), b));
        st(out + 2 * i, f_mul(two_inv, v));
        x_inv = f_mul(x_inv, w_inv);
    }
    return 0;
}

/* ---------------------------------------------------------------- BLAKE2b-512 (RFC 7693), unkeyed */
static const u64 B2_IV[8] = {
    0x6A09E667F3BCC908ULL, 0xBB67AE8584CAA73BULL, 0x3C6EF372FE94F82BULL, 0xA54FF53A5F1D36F1ULL,
    0x510E527FADE682D1ULL, 0x9B05688C2B3E6C1FULL, 0x1F83D9ABFB41BD6BULL, 0x5BE0CD19137E2179ULL};
static const uint8_t B2_SIGMA[12][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};

static inline u64 rotr64(u64 x, int r) { return (x >> r) | (x << (64 - r)); }

static void b2_compress(u64 h[8], const uint8_t block[128], u128 t, int last) {
    u64 m[16], v[16];
    for (int i = 0; i < 16; ++i) memcpy(&m[i], block + 8 * i, 8);   /* little-endian host */
    for (int i = 0; i < 8; ++i) { v[i] = h[i]; v[i + 8] = B2_IV[i]; }
    v[12] ^= (u64)t; v[13] ^= (u64)(t >> 64);
    if (last) v[14] = ~v[14];
#define G(a, b, c, d, x, y) do { \
    v[a] = v[a] + v[b] + (x); v[d] = rotr64(v[d] ^ v[a], 32); v[c] = v[c] + v[d]; v[b] = rotr64(v[b] ^ v[c], 24); \
    v[a] = v[a] + v[b] + (y); v[d] = rotr64(v[d] ^ v[a], 16); v[c] = v[c] + v[d]; v[b] = rotr64(v[b] ^ v[c], 63); } while (0)
    for (int r = 0; r < 12; ++r) {
        const uint8_t* s = B2_SIGMA[r];
        G(0, 4, 8, 12, m[s[0]], m[s[1]]);   G(1, 5, 9, 13, m[s[2]], m[s[3]]);
        G(2, 6, 10, 14, m[s[4]], m[s[5]]);  G(3, 7, 11, 15, m[s[6]], m[s[7]]);
        G(0, 5, 10, 15, m[s[8]], m[s[9]]);  G(1, 6, 11, 12, m[s[10]], m[s[11]]);
        G(2, 7, 8, 13, m[s[12]], m[s[13]]); G(3, 4, 9, 14, m[s[14]], m[s[15]]);
    }
#undef G
    for (int i = 0; i < 8; ++i) h[i] ^= v[i] ^ v[i + 8];
}

/* hashlib.blake2b(msg).digest(): digest_size 64, no key/salt/person (code/merkle.py:4) */
void so_blake2b(const uint8_t* msg, size_t len, uint8_t out[64]) {
    u64 h[8];
    memcpy(h, B2_IV, sizeof h);
    h[0] ^= 0x01010000ULL ^ 64;       /* depth 1, fanout 1, key 0, outlen 64 */
    uint8_t block[128];
    size_t off = 0;
    while (len - off > 128) { b2_compress(h, msg + off, (u128)(off + 128), 0); off += 128; }
    memset(block, 0, 128);
    memcpy(block, msg + off, len - off);
    b2_compress(h, block, (u128)len, 1);
    memcpy(out, h, 64);
}

/* bytes(FieldElement) = str(value).encode(): decimal ASCII, no padding (code/algebra.py:53-57) */
size_t so_leaf_bytes(const u64* elem, char* buf /* >= 40 */) {
    u128 v = ld(elem);
    char tmp[40];
    size_t k = 0;
    if (v == 0) tmp[k++] = '0';
    while (v) { tmp[k++] = (char)('0' + (int)(v % 10)); v /= 10; }
    for (size_t i = 0; i < k; ++i) buf[i] = tmp[k - 1 - i];
    return k;
}

/* Merkle tree over N = 2^k elements (code/merkle.py:6-14).  `levels` receives the whole tree, level 0 =
 * the N leaf digests H(bytes(elem)), level l at offset sum_{j<l} N/2^j digests, last = root.
 * node = H(left || right) (merkle.py:11). */
int so_merkle_tree(const u64* elems, u64 N, uint8_t* levels /* (2N-1)*64 */) {
    if (N == 0 || (N & (N - 1))) return -1;
    char buf[40];
    for (u64 i = 0; i < N; ++i) {
        size_t k = so_leaf_bytes(elems + 2 * i, buf);
        so_blake2b((const uint8_t*)buf, k, levels + 64 * i);
    }
    uint8_t* cur = levels;
    for (u64 w = N; w > 1; w /= 2) {
        uint8_t* nxt = cur + 64 * w;
        for (u64 i = 0; i < w / 2; ++i) so_blake2b(cur + 128 * i, 128, nxt + 64 * i);
        cur = nxt;
    }
    return 0;
}

/* Merkle.commit, merkle.py:13-14 */
int so_merkle_commit(const u64* elems, u64 N, uint8_t root[64]) {
    if (N == 0 || (N & (N - 1))) return -1;
    uint8_t* levels = (uint8_t*)malloc((size_t)(2 * N - 1) * 64);
    so_merkle_tree(elems, N, levels);
    memcpy(root, levels + (size_t)(2 * N - 2) * 64, 64);
    free(levels);
    return 0;
}

/* Merkle.open, merkle.py:16-27: sibling digests bottom-up (leaf sibling first, root child last);
 * log2(N) digests.  N == 1 is rejected like the reference's recursion would never terminate on it. */
int so_merkle_open(const u64* elems, u64 N, u64 index, uint8_t* path /* 64*log2 N */) {
    if (N < 2 || (N & (N - 1)) || index >= N) return -1;
    uint8_t* levels = (uint8_t*)malloc((size_t)(2 * N - 1) * 64);
    so_merkle_tree(elems, N, levels);
    uint8_t* cur = levels;
    u64 idx = index;
    size_t k = 0;
    for (u64 w = N; w > 1; w /= 2) {
        memcpy(path + 64 * k++, cur + 64 * (idx ^ 1), 64);
        cur += 64 * w;
        idx >>= 1;
    }
    free(levels);
    return 0;
}
