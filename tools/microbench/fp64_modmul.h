// fp64_modmul.h -- the FP64-FMA formulation of the modular product in F_p, p = 1 + 407 * 2^119, that VERDICT r5 asked to be
// measured against the integer Montgomery product of csrc/field_asm.cuh (dev tool; not part of the library).
//
// Representation ("R3"): x = x0 + x1 + x2, each limb a double that CARRIES ITS WEIGHT (x_i is an integer multiple of 2^(43 i)),
// signed and lazy: |x_i| < 2^(43 i + 43 + g), g <= 6 growth bits.  Add / sub are three v_add_f64 each, no carries, no
// correction.  A twiddle w is held as the three limbs of w * 2^129 mod p (Montgomery form, R = 2^129 = three limbs), each limb
// pre-scaled by 2^-129, so that column k of the product has weight 2^(43 k - 129) and the result of the reduction lands on
// weights 2^0, 2^43, 2^86 with no rescaling.
//
// Everything runs in ROUND-TOWARD-ZERO (host: fesetround; gfx950: MODE.fp_round[3:2] = 3).  With H a multiple of 2^s inside the
// binade [2^(52+s), 2^(53+s))  --  H = Mg + (small multiple of 2^s),  Mg = 1.5 * 2^(52+s)  --
//       H' = fma(a, b, H)             = H + floor(a b / 2^s) * 2^s        (the sum is positive, so RZ is a floor)
//       lo = fma(a, b, H - H')        = a b - floor(a b / 2^s) * 2^s      in [0, 2^s), exact
// i.e. three instructions per limb product, and the hi parts of one column accumulate in the H chain for free.
//
// Reduction: limb-wise Montgomery.  p = 1 + 407 * 2^119 has p mod 2^43 = 1, so the quotient digit of a column is the negated
// column itself (no multiplication), and m * p = m + m * 407 * 2^119: one more limb product (hi / lo) per column, three in all.
//
//   products 12 x 3 = 36, column sums and the two mid-column splits 24, output normalisation (two splits) 8:   68 per product
//   against 49 for the integer product; a butterfly is 68 + 6 against 49 + 21.
#pragma once
#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define FQ_HD __host__ __device__ __forceinline__
#else
#define FQ_HD inline
#endif

namespace fq {

struct F3 { double l0, l1, l2; };

FQ_HD constexpr double p2(int e) {          // 2^e, e in [-1022, 1023]
    double r = 1.0;
    if (e >= 0) for (int i = 0; i < e; ++i) r *= 2.0; else for (int i = 0; i < -e; ++i) r *= 0.5;
    return r;
}
template <int K> struct Col {                // column K of the scaled product: weight 2^E, magic constant of a split AT 2^E
    static constexpr int E = 43 * K - 129;
    static constexpr double MG = 1.5 * p2(52 + E);
};
static constexpr double C407N = -407.0 * p2(119);      // -(p - 1)
static constexpr double P2_52 = p2(52), P2_95 = p2(52 + 43), P2_138 = p2(52 + 86), TEN_P_TOP = 407.0 * p2(129);

FQ_HD double ffma(double a, double b, double c) { return __builtin_fma(a, b, c); }

// one limb product of class K-1 (hi part into column K's chain H, lo part returned: a multiple of 2^E(K-1) in [0, 2^E(K)))
FQ_HD double mac(double a, double b, double& H) {
    const double Hn = ffma(a, b, H);
    const double d = H - Hn;
    H = Hn;
    return ffma(a, b, d);
}

// out = a * w mod p (lazy limbs), w given as scaled Montgomery limbs.  NORM: carry out0 -> out1 -> out2 so the limbs are back
// within 43 bits (+ sign); without it the result carries about g + 3 growth bits.
template <bool NORM>
FQ_HD F3 modmul(const F3 a, const F3 b) {
    double H1 = Col<1>::MG, H2 = Col<2>::MG, H3 = Col<3>::MG, H4 = Col<4>::MG, H5 = Col<5>::MG;
    const double lo00 = mac(a.l0, b.l0, H1);                       // D0 = lo00
    const double lo01 = mac(a.l0, b.l1, H2);
    const double lo10 = mac(a.l1, b.l0, H2);
    // D1 (+ Mg1 riding along), split at 2^E2: u1 = Mg2 + c1
    const double D1m = (H1 + lo01) + lo10;
    const double u1 = D1m + (Col<2>::MG - Col<1>::MG);
    const double r1 = D1m - (u1 - (Col<2>::MG - Col<1>::MG));
    const double c1 = u1 - Col<2>::MG;
    const double lo02 = mac(a.l0, b.l2, H3);
    const double lo11 = mac(a.l1, b.l1, H3);
    const double lo20 = mac(a.l2, b.l0, H3);
    const double loq0 = mac(lo00, C407N, H3);                      // -D0 * 407 * 2^119
    const double lo12 = mac(a.l1, b.l2, H4);
    const double lo21 = mac(a.l2, b.l1, H4);
    const double loq1 = mac(r1, C407N, H4);
    const double D2m = ((((H2 + lo02) + lo11) + lo20) + loq0) + c1;
    const double u2 = D2m + (Col<3>::MG - Col<2>::MG);
    const double r2 = D2m - (u2 - (Col<3>::MG - Col<2>::MG));
    const double c2 = u2 - Col<3>::MG;
    const double lo22 = mac(a.l2, b.l2, H5);
    const double loq2 = mac(r2, C407N, H5);
    double o0 = ((((H3 - Col<3>::MG) + lo12) + lo21) + loq1) + c2;
    double o1 = ((H4 - Col<4>::MG) + lo22) + loq2;
    double o2 = H5 - Col<5>::MG;
    if (NORM) {
        constexpr double MA = 1.5 * p2(52 + 43), MB = 1.5 * p2(52 + 86);
        const double ca = (o0 + MA) - MA;
        o0 -= ca; o1 += ca;
        const double cb = (o1 + MB) - MB;
        o1 -= cb; o2 += cb;
    }
    return F3{o0, o1, o2};
}

FQ_HD F3 add(F3 a, F3 b) { return F3{a.l0 + b.l0, a.l1 + b.l1, a.l2 + b.l2}; }
FQ_HD F3 sub(F3 a, F3 b) { return F3{a.l0 - b.l0, a.l1 - b.l1, a.l2 - b.l2}; }

// ---- conversions (pass load / store).  u128 canonical (lo, hi) <-> limbs.
// in: a 43-bit field v goes through the mantissa of 2^(52 + 43 i): as_double(EXP | v) - 2^(52 + 43 i) is v * 2^(43 i), exact
FQ_HD double bits_to_double(uint64_t b) { union { uint64_t u; double d; } x; x.u = b; return x.d; }
FQ_HD uint64_t double_to_bits(double d) { union { uint64_t u; double d; } x; x.d = d; return x.u; }
template <int SCALE_E>
FQ_HD F3 from_u128(uint64_t lo, uint64_t hi) {
    constexpr uint64_t M43 = (1ull << 43) - 1;
    const uint64_t v0 = lo & M43, v1 = ((lo >> 43) | (hi << 21)) & M43, v2 = hi >> 22;
    constexpr uint64_t X0 = (uint64_t)(1023 + 52 + SCALE_E) << 52, X1 = (uint64_t)(1023 + 52 + 43 + SCALE_E) << 52, X2 = (uint64_t)(1023 + 52 + 86 + SCALE_E) << 52;
    return F3{bits_to_double(X0 | v0) - bits_to_double(X0), bits_to_double(X1 | v1) - bits_to_double(X1), bits_to_double(X2 | v2) - bits_to_double(X2)};
}

// out: lazy signed limbs -> canonical residue in [0, p).  The value is in (-2^137, 2^137); add 2^10 p (a multiple of p that makes
// it positive), carry the limbs into 43-bit digits (floor splits), then fold what stands above 2^119 with 407 * 2^119 = -1:
// v = q * 407 * 2^119 + r  =>  v = r - q (mod p), q < 2^19.
FQ_HD void to_u128(F3 x, uint64_t& lo, uint64_t& hi) {
    constexpr double MA = 1.5 * p2(52 + 43), MB = 1.5 * p2(52 + 86);
    // + 2^10 p = 2^10 + 407 * 2^129
    double o0 = x.l0 + 1024.0, o1 = x.l1, o2 = x.l2 + TEN_P_TOP;
    const double ca = (o0 + MA) - MA; o0 -= ca; o1 += ca;
    const double cb = (o1 + MB) - MB; o1 -= cb; o2 += cb;
    // digits: o0 in [0, 2^43), o1 / 2^43 in [0, 2^43), o2 / 2^86 in [0, 2^52)
    const uint64_t d0 = double_to_bits(o0 + P2_52) & ((1ull << 52) - 1);
    const uint64_t d1 = double_to_bits(o1 + P2_95) & ((1ull << 52) - 1);
    const uint64_t d2 = double_to_bits(o2 + P2_138) & ((1ull << 52) - 1);
    // v = d0 + d1 2^43 + d2 2^86; the part above 2^119 is d2 >> 33 (< 2^19): q = that / 407, rem goes back
    const uint64_t top = d2 >> 33, q = top / 407, rem = top - q * 407;
    const uint64_t e2 = (d2 & ((1ull << 33) - 1)) | (rem << 33);          // < 407 * 2^33
    unsigned __int128 v = (unsigned __int128)d0 + ((unsigned __int128)d1 << 43) + ((unsigned __int128)e2 << 86);   // < p + 2^86..., below 2^128
    const unsigned __int128 P = ((unsigned __int128)0xCB80000000000000ull << 64) | 1u;
    if (v >= P) v -= P;
    // v - q mod p
    if (v >= q) v -= q; else v = v + P - q;
    lo = (uint64_t)v; hi = (uint64_t)(v >> 64);
}

}  // namespace fq
