// field.cuh -- arithmetic in F_p, p = 1 + 407 * 2^119 (reference code/algebra.py:96-98), for gfx950.
//
// One element = 16 bytes = two little-endian 64-bit limbs (lo, hi).  Data lives in CANONICAL form
// everywhere (HBM, LDS, registers); only constants (twiddles, scale factors) are kept in Montgomery
// form w~ = w * 2^128 mod p, so that  mont_mul(x, w~) = x * w mod p  needs no conversion of the data.
//
// p == 1 (mod 2^64)  =>  -p^-1 mod 2^64 = 2^64 - 1, so each Montgomery step is  m = -t0  and
// m*p = m + (m * 0xcb80000000000000) << 64 : one 64x64 multiply per step instead of two.
//
// The same code compiles for the host (g++, used by the C-ABI's host-side checks and by the CPU
// emulation harness in tests/) and for the device (hipcc).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define SC_HD __host__ __device__ __forceinline__
#else
#define SC_HD inline
#endif

namespace sc {

typedef unsigned __int128 u128;

struct alignas(16) Fe {
    uint64_t lo, hi;
};

static constexpr uint64_t P_LO = 1ull;
static constexpr uint64_t P_HI = 0xCB80000000000000ull;
// R = 2^128 mod p, R2 = 2^256 mod p  (SURVEY.md 8(a) constants, re-derived in tests/test_field_consts.py)
static constexpr uint64_t R_LO = 0xFFFFFFFFFFFFFFFFull, R_HI = 0x347FFFFFFFFFFFFFull;
static constexpr uint64_t R2_LO = 0x5BD53A7F0E778236ull, R2_HI = 0xAAF4AD9A1A6AEDC2ull;

SC_HD Fe fe_zero() { return Fe{0, 0}; }
SC_HD Fe fe_one() { return Fe{1, 0}; }
SC_HD Fe fe_mont_one() { return Fe{R_LO, R_HI}; }
SC_HD bool fe_is_zero(Fe a) { return (a.lo | a.hi) == 0; }
SC_HD bool fe_eq(Fe a, Fe b) { return a.lo == b.lo && a.hi == b.hi; }
SC_HD bool fe_ge_p(Fe a) { return a.hi > P_HI || (a.hi == P_HI && a.lo >= P_LO); }

// dispatchers (defined at the end of this header): portable C on the host, hand-selected gfx950 instruction
// sequences (field_asm.cuh) on the device unless SC_ASM_MUL / SC_ASM_ADDSUB are set to 0
SC_HD Fe fe_add(Fe a, Fe b);
SC_HD Fe fe_sub(Fe a, Fe b);
SC_HD Fe mont_mul(Fe a, Fe b);

// (a + b) mod p, a, b canonical.  2p > 2^128, so the carry out of bit 127 matters.
SC_HD Fe fe_add_c(Fe a, Fe b) {
    uint64_t lo = a.lo + b.lo;
    uint64_t c0 = lo < a.lo;
    uint64_t hi = a.hi + b.hi;
    uint64_t c1 = hi < a.hi;
    uint64_t hi2 = hi + c0;
    c1 |= (hi2 < hi);
    bool ge = c1 || hi2 > P_HI || (hi2 == P_HI && lo >= P_LO);
    uint64_t sl = ge ? P_LO : 0, sh = ge ? P_HI : 0;
    uint64_t rl = lo - sl;
    uint64_t br = lo < sl;
    return Fe{rl, hi2 - sh - br};
}

// (a - b) mod p
SC_HD Fe fe_sub_c(Fe a, Fe b) {
    uint64_t lo = a.lo - b.lo;
    uint64_t b0 = a.lo < b.lo;
    uint64_t hi = a.hi - b.hi;
    uint64_t b1 = a.hi < b.hi;
    uint64_t hi2 = hi - b0;
    b1 |= (hi < b0);
    uint64_t al = b1 ? P_LO : 0, ah = b1 ? P_HI : 0;
    uint64_t rl = lo + al;
    uint64_t c = rl < lo;
    return Fe{rl, hi2 + ah + c};
}

SC_HD Fe fe_neg(Fe a) { return fe_is_zero(a) ? a : fe_sub(Fe{P_LO, P_HI}, a); }

// a/2 mod p
SC_HD Fe fe_half(Fe a) {
    if (a.lo & 1) {   // (a + p) / 2, 129-bit intermediate
        uint64_t lo = a.lo + P_LO;
        uint64_t c0 = lo < a.lo;
        uint64_t hi = a.hi + P_HI;
        uint64_t c1 = hi < a.hi;
        uint64_t hi2 = hi + c0;
        c1 |= (hi2 < hi);
        return Fe{(lo >> 1) | (hi2 << 63), (hi2 >> 1) | (c1 << 63)};
    }
    return Fe{(a.lo >> 1) | (a.hi << 63), a.hi >> 1};
}

// Montgomery product a * b * 2^-128 mod p.  Requires a * b < 2^128 * p (e.g. a < 2^128, b < p).
// Output canonical.
SC_HD Fe mont_mul_c(Fe a, Fe b) {
    u128 p00 = (u128)a.lo * b.lo, p01 = (u128)a.lo * b.hi, p10 = (u128)a.hi * b.lo, p11 = (u128)a.hi * b.hi;
    uint64_t t0 = (uint64_t)p00;
    u128 mid = (p00 >> 64) + (uint64_t)p01 + (uint64_t)p10;
    uint64_t t1 = (uint64_t)mid;
    u128 hi = (mid >> 64) + (p01 >> 64) + (p10 >> 64) + (uint64_t)p11;
    uint64_t t2 = (uint64_t)hi;
    uint64_t t3 = (uint64_t)((hi >> 64) + (p11 >> 64));
    // step 0: m0 = -t0; t0 + m0 = 2^64 * (t0 != 0)
    uint64_t m0 = 0 - t0;
    u128 x = (u128)m0 * P_HI + t1 + (uint64_t)(t0 != 0);
    uint64_t u1 = (uint64_t)x;
    u128 y = (x >> 64) + t2;
    uint64_t u2 = (uint64_t)y;
    uint64_t u3 = t3 + (uint64_t)(y >> 64);
    uint64_t c3 = u3 < t3;
    // step 1
    uint64_t m1 = 0 - u1;
    u128 x2 = (u128)m1 * P_HI + u2 + (uint64_t)(u1 != 0);
    uint64_t v2 = (uint64_t)x2;
    u128 y2 = (x2 >> 64) + u3;
    uint64_t v3 = (uint64_t)y2;
    uint64_t c4 = c3 + (uint64_t)(y2 >> 64);
    bool ge = c4 || v3 > P_HI || (v3 == P_HI && v2 >= P_LO);
    uint64_t sl = ge ? P_LO : 0, sh = ge ? P_HI : 0;
    uint64_t rl = v2 - sl;
    uint64_t br = v2 < sl;
    return Fe{rl, v3 - sh - br};
}

SC_HD Fe to_mont(Fe a) { return mont_mul(a, Fe{R2_LO, R2_HI}); }
SC_HD Fe from_mont(Fe a) { return mont_mul(a, fe_one()); }
// plain modular product of two canonical values (two Montgomery steps)
SC_HD Fe fe_mul(Fe a, Fe b) { return mont_mul(mont_mul(a, b), Fe{R2_LO, R2_HI}); }

// base^e for a Montgomery-form base; result in Montgomery form.  (square-and-multiply, LSB first)
SC_HD Fe mont_pow(Fe base_m, uint64_t e) {
    Fe acc = fe_mont_one();
    while (e) {
        if (e & 1) acc = mont_mul(acc, base_m);
        base_m = mont_mul(base_m, base_m);
        e >>= 1;
    }
    return acc;
}

// 128-bit exponent variant (used for Fermat inversion: x^(p-2))
SC_HD Fe mont_pow128(Fe base_m, uint64_t e_lo, uint64_t e_hi) {
    Fe acc = fe_mont_one();
    for (int i = 0; i < 64; ++i) {
        if ((e_lo >> i) & 1) acc = mont_mul(acc, base_m);
        base_m = mont_mul(base_m, base_m);
    }
    for (int i = 0; i < 64 && (e_hi >> i); ++i) {
        if ((e_hi >> i) & 1) acc = mont_mul(acc, base_m);
        base_m = mont_mul(base_m, base_m);
    }
    return acc;
}

// inverse of a Montgomery-form value, result in Montgomery form; inverse(0) = 0 like the reference's
// xgcd-based Field.inverse (code/algebra.py:87-89).
// x^(p - 2) by an addition chain on the shape of p - 2 = 406 * 2^119 + (2^119 - 1): y = x^(2^119 - 1) (118 squarings, 11 products:
// exponents 2^k - 1 for k = 1, 2, 3, 6, 7, 14, 28, 29, 58, 59, 118, 119), z = y x = x^(2^119), result = z^406 y (406 = 110010110b:
// 8 squarings, 4 products) -- 143 modular products where square-and-multiply over the 128 bits takes 252.
SC_HD Fe mont_sqr_n(Fe v, int n) {
    for (int i = 0; i < n; ++i) v = mont_mul(v, v);
    return v;
}
SC_HD Fe mont_inv(Fe x) {
#if !defined(__HIP_DEVICE_COMPILE__)
    // (the host inverts a handful of constants per call; its optimiser takes minutes over the chain below with mont_mul_c inlined)
    return mont_pow128(x, 0xFFFFFFFFFFFFFFFFull, P_HI - 1);
#else
    const Fe x2 = mont_mul(mont_sqr_n(x, 1), x);
    const Fe x3 = mont_mul(mont_sqr_n(x2, 1), x);
    const Fe x6 = mont_mul(mont_sqr_n(x3, 3), x3);
    const Fe x7 = mont_mul(mont_sqr_n(x6, 1), x);
    const Fe x14 = mont_mul(mont_sqr_n(x7, 7), x7);
    const Fe x28 = mont_mul(mont_sqr_n(x14, 14), x14);
    const Fe x29 = mont_mul(mont_sqr_n(x28, 1), x);
    const Fe x58 = mont_mul(mont_sqr_n(x29, 29), x29);
    const Fe x59 = mont_mul(mont_sqr_n(x58, 1), x);
    const Fe x118 = mont_mul(mont_sqr_n(x59, 59), x59);
    const Fe y = mont_mul(mont_sqr_n(x118, 1), x);           // x^(2^119 - 1)
    const Fe z = mont_mul(y, x);                              // x^(2^119)
    Fe w = mont_mul(mont_sqr_n(z, 1), z);                     // 11b
    w = mont_mul(mont_sqr_n(w, 3), z);                        // 11001b
    w = mont_mul(mont_sqr_n(w, 2), z);                        // 1100101b
    w = mont_mul(mont_sqr_n(w, 1), z);                        // 11001011b
    w = mont_sqr_n(w, 1);                                     // 110010110b = 406
    return mont_mul(w, y);
#endif
}

}  // namespace sc

#ifndef SC_ASM_MUL
#define SC_ASM_MUL 1
#endif
#ifndef SC_ASM_ADDSUB
#define SC_ASM_ADDSUB 1
#endif
#include "field_asm.cuh"

namespace sc {
SC_HD Fe mont_mul(Fe a, Fe b) {
#if defined(__HIP_DEVICE_COMPILE__) && SC_ASM_MUL
    return mont_mul_asm(a, b);
#else
    return mont_mul_c(a, b);
#endif
}
// two independent products at once (device: hand-interleaved, see field_asm.cuh; elsewhere: one after the other)
#ifndef SC_MUL2
#define SC_MUL2 1
#endif
SC_HD void mont_mul2(Fe a0, Fe b0, Fe a1, Fe b1, Fe& r0, Fe& r1) {
#if defined(__HIP_DEVICE_COMPILE__) && SC_ASM_MUL && SC_MUL2
    mont_mul2_asm(a0, b0, a1, b1, r0, r1);
#else
    r0 = mont_mul(a0, b0);
    r1 = mont_mul(a1, b1);
#endif
}
// sums and differences of two butterflies at once (device: four interleaved carry chains, see field_asm.cuh)
SC_HD void fe_addsub2(Fe u0, Fe v0, Fe u1, Fe v1, Fe& s0, Fe& d0, Fe& s1, Fe& d1) {
#if defined(__HIP_DEVICE_COMPILE__) && SC_ASM_ADDSUB && SC_MUL2
    fe_addsub2_asm(u0, v0, u1, v1, s0, d0, s1, d1);
#else
    s0 = fe_add(u0, v0); d0 = fe_sub(u0, v0);
    s1 = fe_add(u1, v1); d1 = fe_sub(u1, v1);
#endif
}
SC_HD Fe fe_add(Fe a, Fe b) {
#if defined(__HIP_DEVICE_COMPILE__) && SC_ASM_ADDSUB
    return fe_add_asm(a, b);
#else
    return fe_add_c(a, b);
#endif
}
SC_HD Fe fe_sub(Fe a, Fe b) {
#if defined(__HIP_DEVICE_COMPILE__) && SC_ASM_ADDSUB
    return fe_sub_asm(a, b);
#else
    return fe_sub_c(a, b);
#endif
}
}  // namespace sc
