// field_asm.cuh -- gfx950 device-only implementations of the hot field operations, written with
// single-instruction inline-asm wrappers so that the instruction selection is exactly the one the cycle
// model (profiles/r01/valu_ubench.txt) says is cheapest, while hipcc still does register allocation.
//
// Costs on gfx950: v_mad_u64_u32 = v_addc_co_u32 = 4 cycles per wave-instruction, plain VOP2 = 2.
// The compiler's expansion of unsigned __int128 spends ~25 % of the multiplier on v_mov (zero-extending
// high words into aligned register pairs); this version needs none in the product:
//
//   product   : 16 v_mad_u64_u32 into even (columns 2k,2k+1) and odd (2k+1,2k+2) 64-bit accumulators;
//               carry-outs are counted per accumulator and enter the NEXT same-parity accumulator as the
//               64-bit addend of its first v_mad (which cannot overflow), so no carry ever needs propagating
//   merge     : T = E + (O << 32): one 7-step v_addc chain
//   reduction : p = 1 + 407*2^119  =>  p^-1 = 1 - 407*2^119 (mod 2^128), so with x = (T_lo*407) mod 2^9
//               m' = T_lo * p^-1 = T_lo - x*2^119 (mod 2^128): T_lo with x subtracted from its top 9 bits [borrow delta]
//               (T - m'*p) / 2^128 = T_hi - (m'*PH >> 32) - delta,   PH = 407*2^23   (exact: m'*PH = x*2^23 mod 2^32)
//               -- a 4-step v_mad chain whose first low word is x*2^23 itself, then one 4-limb subtraction
//   final     : the result lies in (-p, p): add p back where the subtraction borrowed (5 instructions)
//
// Hazard: a VALU that writes an SGPR pair (carry-out) needs 2 wait states before a VALU reads it as
// carry-in / select mask; hipcc does not look inside asm, so every consumer carries its own `s_nop 1`.
#pragma once
// included from field.cuh (after the portable definitions)

#if defined(__HIP_DEVICE_COMPILE__)
namespace sc {

typedef uint64_t smask_t;   // 64-lane mask held in an SGPR pair

#ifndef SC_HAZ_NOP
// the 2 wait states between a VALU writing an SGPR pair and a VALU reading it.  Timing-only experiment (results are wrong
// without them): dropping every nop gains 6 % at 2^20, 3 % at 2^22, nothing at 2^24 -- other waves fill the slots, so
// hand-interleaving two multiplications to get rid of the nops is not worth its register cost.
#define SC_HAZ_NOP "s_nop 1\n\t"
#endif

__device__ __forceinline__ uint64_t a_mad(uint32_t a, uint32_t b, uint64_t c) {
    uint64_t d; smask_t cy;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(cy) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ uint64_t a_madc(uint32_t a, uint32_t b, uint64_t c, smask_t& cy) {
    uint64_t d;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(cy) : "v"(a), "v"(b), "v"(c));
    return d;
}
// a * k + c with an inline-constant / SGPR-free literal multiplier is not encodable for all k; pass k in a VGPR/SGPR
__device__ __forceinline__ uint32_t a_add_co(uint32_t x, uint32_t y, smask_t& co) {
    uint32_t s;
    asm("v_add_co_u32_e64 %0, %1, %2, %3" : "=v"(s), "=s"(co) : "v"(x), "v"(y));
    return s;
}
__device__ __forceinline__ uint32_t a_addc(uint32_t x, uint32_t y, smask_t ci, smask_t& co) {
    uint32_t s;
    asm(SC_HAZ_NOP "v_addc_co_u32_e64 %0, %1, %2, %3, %4" : "=v"(s), "=s"(co) : "v"(x), "v"(y), "s"(ci));
    return s;
}
__device__ __forceinline__ uint32_t a_addc_last(uint32_t x, uint32_t y, smask_t ci) {
    uint32_t s; smask_t co;
    asm(SC_HAZ_NOP "v_addc_co_u32_e64 %0, %1, %2, %3, %4" : "=v"(s), "=s"(co) : "v"(x), "v"(y), "s"(ci));
    return s;
}
// x + 0 + carry  (carry counter)
__device__ __forceinline__ uint32_t a_inc(uint32_t x, smask_t ci) {
    uint32_t s; smask_t co;
    asm(SC_HAZ_NOP "v_addc_co_u32_e64 %0, %1, 0, %2, %3" : "=v"(s), "=s"(co) : "v"(x), "s"(ci));
    return s;
}
__device__ __forceinline__ uint32_t a_sub_co(uint32_t x, uint32_t y, smask_t& bo) {
    uint32_t s;
    asm("v_sub_co_u32_e64 %0, %1, %2, %3" : "=v"(s), "=s"(bo) : "v"(x), "v"(y));
    return s;
}
__device__ __forceinline__ uint32_t a_subb(uint32_t x, uint32_t y, smask_t bi, smask_t& bo) {
    uint32_t s;
    asm(SC_HAZ_NOP "v_subb_co_u32_e64 %0, %1, %2, %3, %4" : "=v"(s), "=s"(bo) : "v"(x), "v"(y), "s"(bi));
    return s;
}
// 0 - y - borrow
__device__ __forceinline__ uint32_t a_negb(uint32_t y, smask_t bi, smask_t& bo) {
    uint32_t s;
    asm(SC_HAZ_NOP "v_subb_co_u32_e64 %0, %1, 0, %2, %3" : "=v"(s), "=s"(bo) : "v"(y), "s"(bi));
    return s;
}
// sel ? b : a   per lane
__device__ __forceinline__ uint32_t a_cnd(uint32_t a, uint32_t b, smask_t sel) {
    uint32_t r;
    asm(SC_HAZ_NOP "v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(sel));
    return r;
}

__device__ __forceinline__ uint32_t lo32(uint64_t x) { return (uint32_t)x; }
__device__ __forceinline__ uint32_t hi32(uint64_t x) { return (uint32_t)(x >> 32); }

static constexpr uint32_t PH3 = 0xCB800000u;     // top 32-bit limb of p (p = [1, 0, 0, PH3])

// R (4 limbs + carry) -> canonical: subtract p when carry | R >= p
__device__ __forceinline__ Fe a_cond_sub_p(uint32_t r0, uint32_t r1, uint32_t r2, uint32_t r3, smask_t cf) {
    smask_t b;
    uint32_t d0 = a_sub_co(r0, 1u, b);
    uint32_t d1 = a_subb(r1, 0u, b, b);
    uint32_t d2 = a_subb(r2, 0u, b, b);
    uint32_t d3 = a_subb(r3, PH3, b, b);          // b: R < p (when cf == 0)
    smask_t sel = cf | ~b;                         // take R - p
    uint32_t o0 = a_cnd(r0, d0, sel), o1 = a_cnd(r1, d1, sel), o2 = a_cnd(r2, d2, sel), o3 = a_cnd(r3, d3, sel);
    return Fe{((uint64_t)o1 << 32) | o0, ((uint64_t)o3 << 32) | o2};
}

__device__ __forceinline__ Fe mont_mul_asm(Fe a, Fe b) {
    const uint32_t a0 = lo32(a.lo), a1 = hi32(a.lo), a2 = lo32(a.hi), a3 = hi32(a.hi);
    const uint32_t b0 = lo32(b.lo), b1 = hi32(b.lo), b2 = lo32(b.hi), b3 = hi32(b.hi);
    smask_t c;
    // ---- product, even/odd accumulators
    uint64_t E0 = a_mad(a0, b0, 0);
    uint64_t O0 = a_mad(a0, b1, 0);
    O0 = a_madc(a1, b0, O0, c);
    uint32_t nO0 = a_inc(0u, c);
    uint64_t E1 = a_mad(a0, b2, 0);
    E1 = a_madc(a1, b1, E1, c);
    uint32_t nE1 = a_inc(0u, c);
    E1 = a_madc(a2, b0, E1, c);
    nE1 = a_inc(nE1, c);
    uint64_t O1 = a_mad(a0, b3, (uint64_t)nO0);
    O1 = a_madc(a1, b2, O1, c);
    uint32_t nO1 = a_inc(0u, c);
    O1 = a_madc(a2, b1, O1, c);
    nO1 = a_inc(nO1, c);
    O1 = a_madc(a3, b0, O1, c);
    nO1 = a_inc(nO1, c);
    uint64_t E2 = a_mad(a1, b3, (uint64_t)nE1);
    E2 = a_madc(a2, b2, E2, c);
    uint32_t nE2 = a_inc(0u, c);
    E2 = a_madc(a3, b1, E2, c);
    nE2 = a_inc(nE2, c);
    uint64_t O2 = a_mad(a2, b3, (uint64_t)nO1);
    O2 = a_madc(a3, b2, O2, c);
    uint32_t nO2 = a_inc(0u, c);
    uint64_t E3 = a_mad(a3, b3, (uint64_t)nE2);
    // ---- merge T = E + (O << 32)
    const uint32_t t0 = lo32(E0);
    uint32_t t1 = a_add_co(hi32(E0), lo32(O0), c);
    uint32_t t2 = a_addc(lo32(E1), hi32(O0), c, c);
    uint32_t t3 = a_addc(hi32(E1), lo32(O1), c, c);
    uint32_t t4 = a_addc(lo32(E2), hi32(O1), c, c);
    uint32_t t5 = a_addc(hi32(E2), lo32(O2), c, c);
    uint32_t t6 = a_addc(lo32(E3), hi32(O2), c, c);
    uint32_t t7 = a_addc_last(hi32(E3), nO2, c);
    // ---- reduction (subtractive form): m' = T_lo * p^-1 mod 2^128 = T_lo - x*2^119 (mod 2^128) -- only the top limb of
    // T_lo changes -- and  T * 2^-128 = T_hi - (floor(m' * PH / 2^32) + delta)  in (-p, p),  PH = 407 * 2^23,
    // delta = borrow of the top-limb subtraction; the low word of t0 * PH IS x * 2^23, so x costs nothing.
    const uint32_t PHc = PH3;                       // 407 * 2^23
    uint64_t s0 = a_mad(t0, PHc, 0);
    smask_t bw;
    uint32_t m3 = a_sub_co(t3, lo32(s0), bw);      // bw = delta
    uint64_t s1 = a_mad(t1, PHc, (uint64_t)hi32(s0));
    uint64_t s2 = a_mad(t2, PHc, (uint64_t)hi32(s1));
    uint64_t s3 = a_mad(m3, PHc, (uint64_t)hi32(s2));
    // R = T_hi - Q - delta; negative -> add p back (same tail as fe_sub)
    uint32_t r0 = a_subb(t4, lo32(s1), bw, bw);
    uint32_t r1 = a_subb(t5, lo32(s2), bw, bw);
    uint32_t r2 = a_subb(t6, lo32(s3), bw, bw);
    uint32_t r3 = a_subb(t7, hi32(s3), bw, bw);
    uint32_t ph = a_cnd(0u, PH3, bw);
    uint32_t o0 = a_addc(r0, 0u, bw, c);
    uint32_t o1 = a_addc(r1, 0u, c, c);
    uint32_t o2 = a_addc(r2, 0u, c, c);
    uint32_t o3 = a_addc_last(r3, ph, c);
    return Fe{((uint64_t)o1 << 32) | o0, ((uint64_t)o3 << 32) | o2};
}

// ---------------------------------------------------------------------------------------------------------------------
// Two independent Montgomery products, hand-interleaved (A, B, A, B, ...) so that every carry / borrow consumer sits at least two
// VALU instructions behind its producer: the 2 wait states of the SGPR hazard are filled with the other product's work instead of
// `s_nop`s (one product alone spends ~25 `s_nop 1` plus what hipcc adds between the asm statements; a pair needs 10 `s_nop 0`).
// The statements are `asm volatile`, which pins their relative order -- the compiler may put other instructions between them
// (that only adds wait states), never fewer.  Same arithmetic as mont_mul_asm, instruction for instruction.
#define SC_V_MAD(d, cy, x, y, c)   asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(cy) : "v"(x), "v"(y), "v"(c))
#define SC_V_INC(d, x, ci, co)     asm volatile("v_addc_co_u32_e64 %0, %1, 0, %2, %3" : "=v"(d), "=s"(co) : "v"(x), "s"(ci))
#define SC_V_ADDCO(d, co, x, y)    asm volatile("v_add_co_u32_e64 %0, %1, %2, %3" : "=v"(d), "=s"(co) : "v"(x), "v"(y))
#define SC_V_ADDC(d, co, x, y, ci) asm volatile("v_addc_co_u32_e64 %0, %1, %2, %3, %4" : "=v"(d), "=s"(co) : "v"(x), "v"(y), "s"(ci))
#define SC_V_SUBCO(d, bo, x, y)    asm volatile("v_sub_co_u32_e64 %0, %1, %2, %3" : "=v"(d), "=s"(bo) : "v"(x), "v"(y))
#define SC_V_SUBB(d, bo, x, y, bi) asm volatile("v_subb_co_u32_e64 %0, %1, %2, %3, %4" : "=v"(d), "=s"(bo) : "v"(x), "v"(y), "s"(bi))
#define SC_V_CND(d, x, y, m)       asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(d) : "v"(x), "v"(y), "s"(m))
#define SC_V_NOP0()                asm volatile("s_nop 0")

// Carry-outs nobody reads.  hipcc cannot look inside an asm statement, so on gfx950 it assumes the worst whenever an asm statement
// touches a register the PREVIOUS asm statement defined (the dst-forwarding hazard) and puts an `s_nop 0` between them -- with one
// shared dump register that is after every v_mad_u64_u32.  Chain A dumps into vcc, chain B into s[100:101] (named in the text and
// declared as clobbers), so that neighbouring statements of the interleave never share a register.
template <int W> __device__ __forceinline__ void v_mad_dump(uint64_t& d, uint32_t x, uint32_t y, uint64_t c) {
    if constexpr (W == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(d) : "v"(x), "v"(y), "v"(c) : "vcc");
    else asm volatile("v_mad_u64_u32 %0, s[100:101], %1, %2, %3" : "=v"(d) : "v"(x), "v"(y), "v"(c) : "s100", "s101");
}
template <int W> __device__ __forceinline__ void v_inc_dump(uint32_t& d, uint32_t x, smask_t ci) {
    if constexpr (W == 0) asm volatile("v_addc_co_u32_e64 %0, vcc, %1, 0, %2" : "=v"(d) : "v"(x), "s"(ci) : "vcc");
    else asm volatile("v_addc_co_u32_e64 %0, s[100:101], %1, 0, %2" : "=v"(d) : "v"(x), "s"(ci) : "s100", "s101");
}
template <int W> __device__ __forceinline__ void v_addc_dump(uint32_t& d, uint32_t x, uint32_t y, smask_t ci) {
    if constexpr (W == 0) asm volatile("v_addc_co_u32_e64 %0, vcc, %1, %2, %3" : "=v"(d) : "v"(x), "v"(y), "s"(ci) : "vcc");
    else asm volatile("v_addc_co_u32_e64 %0, s[100:101], %1, %2, %3" : "=v"(d) : "v"(x), "v"(y), "s"(ci) : "s100", "s101");
}
#define SC_V_MADD(d, x, y, c)      v_mad_dump<W>(d, x, y, c)
#define SC_V_INCD(d, x, ci)        v_inc_dump<W>(d, x, ci)
#define SC_V_ADDCD(d, x, y, ci)    v_addc_dump<W>(d, x, y, ci)

struct MulState {       // registers of one product in flight
    uint32_t a0, a1, a2, a3, b0, b1, b2, b3;
    uint64_t E0, O0, E1, O1, E2, O2, E3, s0, s1, s2, s3;
    uint32_t nO0, nE1, nO1, nE2, nO2, t1, t2, t3, t4, t5, t6, t7, m3, r0, r1, r2, r3, ph, o0, o1, o2, o3;
    smask_t c1, c2, c3, c4, c5, c6, c7, c8, c9, ct, bw, cx;
};

__device__ __forceinline__ void mont_mul2_asm(Fe xa, Fe wa, Fe xb, Fe wb, Fe& ra, Fe& rb) {
    MulState A, B;
    A.a0 = lo32(xa.lo); A.a1 = hi32(xa.lo); A.a2 = lo32(xa.hi); A.a3 = hi32(xa.hi);
    A.b0 = lo32(wa.lo); A.b1 = hi32(wa.lo); A.b2 = lo32(wa.hi); A.b3 = hi32(wa.hi);
    B.a0 = lo32(xb.lo); B.a1 = hi32(xb.lo); B.a2 = lo32(xb.hi); B.a3 = hi32(xb.hi);
    B.b0 = lo32(wb.lo); B.b1 = hi32(wb.lo); B.b2 = lo32(wb.hi); B.b3 = hi32(wb.hi);
    const uint64_t zero64 = 0;
    const uint32_t zero32 = 0, PHc = PH3;
#define SC_BOTH(STEP) { MulState& M = A; [[maybe_unused]] constexpr int W = 0; STEP } { MulState& M = B; [[maybe_unused]] constexpr int W = 1; STEP }
    // ---- product.  Carry counters travel in PAIRS: the odd accumulator O_k covers columns (2k+1, 2k+2), and the overflow counts
    // of O_{k-1} (weight: column 2k+1) and of E_k (weight: column 2k+2) are exactly the low and the high word of the 64-bit addend
    // of O_k's first v_mad -- {nO0, nE1} for O1, {nO1, nE2} for O2 -- so no counter is ever zero-extended into a register pair
    // of its own (4 v_mov per product before), and the even accumulators start from the shared zero.  That first v_mad cannot
    // overflow: its product has the twiddle's top limb b3 <= 0xCB800000 as a factor (a * b3 <= 0.8 * 2^64) and the addend is
    // below 2^34.  The schedule keeps every carry consumer >= 2 instructions behind its producer (4 with the interleave).
    SC_BOTH(SC_V_MADD(M.E0, M.a0, M.b0, zero64);)
    SC_BOTH(SC_V_MADD(M.O0, M.a0, M.b1, zero64);)
    SC_BOTH(SC_V_MADD(M.E1, M.a0, M.b2, zero64);)
    SC_BOTH(SC_V_MAD(M.O0, M.c1, M.a1, M.b0, M.O0);)
    SC_BOTH(SC_V_MAD(M.E1, M.c2, M.a1, M.b1, M.E1);)
    SC_BOTH(SC_V_MAD(M.E1, M.c3, M.a2, M.b0, M.E1);)
    SC_BOTH(SC_V_INCD(M.nO0, zero32, M.c1);)
    SC_BOTH(SC_V_INCD(M.nE1, zero32, M.c2);)
    SC_BOTH(SC_V_INCD(M.nE1, M.nE1, M.c3);)
    SC_BOTH(SC_V_MADD(M.O1, M.a0, M.b3, (uint64_t)M.nO0 | ((uint64_t)M.nE1 << 32));)
    SC_BOTH(SC_V_MAD(M.O1, M.c4, M.a1, M.b2, M.O1);)
    SC_BOTH(SC_V_MAD(M.O1, M.c5, M.a2, M.b1, M.O1);)
    SC_BOTH(SC_V_INCD(M.nO1, zero32, M.c4);)
    SC_BOTH(SC_V_MAD(M.O1, M.c6, M.a3, M.b0, M.O1);)
    SC_BOTH(SC_V_MADD(M.E2, M.a1, M.b3, zero64);)
    SC_BOTH(SC_V_INCD(M.nO1, M.nO1, M.c5);)
    SC_BOTH(SC_V_MAD(M.E2, M.c7, M.a2, M.b2, M.E2);)
    SC_BOTH(SC_V_INCD(M.nO1, M.nO1, M.c6);)
    SC_BOTH(SC_V_MAD(M.E2, M.c8, M.a3, M.b1, M.E2);)
    SC_BOTH(SC_V_INCD(M.nE2, zero32, M.c7);)
    SC_BOTH(SC_V_MADD(M.E3, M.a3, M.b3, zero64);)
    SC_BOTH(SC_V_INCD(M.nE2, M.nE2, M.c8);)
    SC_BOTH(SC_V_MADD(M.O2, M.a2, M.b3, (uint64_t)M.nO1 | ((uint64_t)M.nE2 << 32));)
    SC_BOTH(SC_V_MAD(M.O2, M.c9, M.a3, M.b2, M.O2);)
    // ---- merge T = E + (O << 32), interleaved with the reduction chain s0..s3 (t0 = lo32(E0)); its first two steps stand
    // between the last v_mad of the product and the counter that reads its carry
    SC_BOTH(SC_V_ADDCO(M.t1, M.ct, hi32(M.E0), lo32(M.O0));)
    SC_BOTH(SC_V_MADD(M.s0, lo32(M.E0), PHc, zero64);)
    SC_BOTH(SC_V_INCD(M.nO2, zero32, M.c9);)
    SC_BOTH(SC_V_ADDC(M.t2, M.ct, lo32(M.E1), hi32(M.O0), M.ct);)
    SC_BOTH(SC_V_MADD(M.s1, M.t1, PHc, (uint64_t)hi32(M.s0));)
    SC_BOTH(SC_V_ADDC(M.t3, M.ct, hi32(M.E1), lo32(M.O1), M.ct);)
    SC_BOTH(SC_V_MADD(M.s2, M.t2, PHc, (uint64_t)hi32(M.s1));)
    SC_BOTH(SC_V_SUBCO(M.m3, M.bw, M.t3, lo32(M.s0));)
    SC_BOTH(SC_V_ADDC(M.t4, M.ct, lo32(M.E2), hi32(M.O1), M.ct);)
    SC_BOTH(SC_V_MADD(M.s3, M.m3, PHc, (uint64_t)hi32(M.s2));)
    SC_BOTH(SC_V_ADDC(M.t5, M.ct, hi32(M.E2), lo32(M.O2), M.ct);)
    SC_BOTH(SC_V_SUBB(M.r0, M.bw, M.t4, lo32(M.s1), M.bw);)
    SC_BOTH(SC_V_ADDC(M.t6, M.ct, lo32(M.E3), hi32(M.O2), M.ct);)
    SC_BOTH(SC_V_SUBB(M.r1, M.bw, M.t5, lo32(M.s2), M.bw);)
    SC_BOTH(SC_V_ADDC(M.t7, M.cx, hi32(M.E3), M.nO2, M.ct);)
    SC_BOTH(SC_V_SUBB(M.r2, M.bw, M.t6, lo32(M.s3), M.bw);)
    SC_V_NOP0();
    SC_BOTH(SC_V_SUBB(M.r3, M.bw, M.t7, hi32(M.s3), M.bw);)
    SC_V_NOP0();
    // ---- negative -> add p back
    SC_BOTH(SC_V_CND(M.ph, zero32, PHc, M.bw);)
    SC_BOTH(SC_V_ADDC(M.o0, M.cx, M.r0, zero32, M.bw);)
    SC_V_NOP0();
    SC_BOTH(SC_V_ADDC(M.o1, M.cx, M.r1, zero32, M.cx);)
    SC_V_NOP0();
    SC_BOTH(SC_V_ADDC(M.o2, M.cx, M.r2, zero32, M.cx);)
    SC_V_NOP0();
    SC_BOTH(SC_V_ADDCD(M.o3, M.r3, M.ph, M.cx);)
#undef SC_BOTH
    ra = Fe{((uint64_t)A.o1 << 32) | A.o0, ((uint64_t)A.o3 << 32) | A.o2};
    rb = Fe{((uint64_t)B.o1 << 32) | B.o0, ((uint64_t)B.o3 << 32) | B.o2};
}

// The sums and differences of TWO butterflies (u0 +- v0, u1 +- v1): four independent carry chains issued round-robin, so every
// carry consumer is three instructions behind its producer and no `s_nop` is needed at all (fe_add_asm + fe_sub_asm spend 18 per
// butterfly).  Same arithmetic as fe_add_asm / fe_sub_asm.
struct AddSubState {
    uint32_t u0, u1, u2, u3, v0, v1, v2, v3;
    uint32_t r0, r1, r2, r3, t0, t1, t2, t3, d0, d1, d2, d3, ph, e0, e1, e2, e3, s0, s1, s2, s3;
    smask_t ca, cb, bd, ce, sel;
};

__device__ __forceinline__ void fe_addsub2_asm(Fe ua, Fe va, Fe ub, Fe vb, Fe& sa, Fe& da, Fe& sb, Fe& db) {
    AddSubState A, B;
    A.u0 = lo32(ua.lo); A.u1 = hi32(ua.lo); A.u2 = lo32(ua.hi); A.u3 = hi32(ua.hi);
    A.v0 = lo32(va.lo); A.v1 = hi32(va.lo); A.v2 = lo32(va.hi); A.v3 = hi32(va.hi);
    B.u0 = lo32(ub.lo); B.u1 = hi32(ub.lo); B.u2 = lo32(ub.hi); B.u3 = hi32(ub.hi);
    B.v0 = lo32(vb.lo); B.v1 = hi32(vb.lo); B.v2 = lo32(vb.hi); B.v3 = hi32(vb.hi);
    const uint32_t zero32 = 0, one32 = 1, PHc = PH3;
#define SC_BOTH(STEP) { AddSubState& M = A; [[maybe_unused]] constexpr int W = 0; STEP } { AddSubState& M = B; [[maybe_unused]] constexpr int W = 1; STEP }
    // sum chain (r, carry ca) and difference chain (d, borrow bd), limb by limb
    SC_BOTH(SC_V_ADDCO(M.r0, M.ca, M.u0, M.v0); SC_V_SUBCO(M.d0, M.bd, M.u0, M.v0);)
    SC_BOTH(SC_V_ADDC(M.r1, M.ca, M.u1, M.v1, M.ca); SC_V_SUBB(M.d1, M.bd, M.u1, M.v1, M.bd);)
    SC_BOTH(SC_V_ADDC(M.r2, M.ca, M.u2, M.v2, M.ca); SC_V_SUBB(M.d2, M.bd, M.u2, M.v2, M.bd);)
    SC_BOTH(SC_V_ADDC(M.r3, M.ca, M.u3, M.v3, M.ca); SC_V_SUBB(M.d3, M.bd, M.u3, M.v3, M.bd);)
    // sum - p (borrow cb)  |  difference + p where it borrowed (carry ce)
    SC_BOTH(SC_V_SUBCO(M.t0, M.cb, M.r0, one32); SC_V_CND(M.ph, zero32, PHc, M.bd);)
    SC_BOTH(SC_V_SUBB(M.t1, M.cb, M.r1, zero32, M.cb); SC_V_ADDC(M.e0, M.ce, M.d0, zero32, M.bd);)
    SC_BOTH(SC_V_SUBB(M.t2, M.cb, M.r2, zero32, M.cb); SC_V_ADDC(M.e1, M.ce, M.d1, zero32, M.ce);)
    SC_BOTH(SC_V_SUBB(M.t3, M.cb, M.r3, PHc, M.cb); SC_V_ADDC(M.e2, M.ce, M.d2, zero32, M.ce);)
    // take sum - p where the sum carried out of bit 127 or did not borrow against p (the last step of the difference chain rides
    // along so that it, too, stays three instructions behind its carry)
    A.sel = A.ca | ~A.cb;
    B.sel = B.ca | ~B.cb;
    SC_BOTH(SC_V_CND(M.s0, M.r0, M.t0, M.sel); SC_V_ADDCD(M.e3, M.d3, M.ph, M.ce);)
    SC_BOTH(SC_V_CND(M.s1, M.r1, M.t1, M.sel);)
    SC_BOTH(SC_V_CND(M.s2, M.r2, M.t2, M.sel);)
    SC_BOTH(SC_V_CND(M.s3, M.r3, M.t3, M.sel);)
#undef SC_BOTH
    sa = Fe{((uint64_t)A.s1 << 32) | A.s0, ((uint64_t)A.s3 << 32) | A.s2};
    da = Fe{((uint64_t)A.e1 << 32) | A.e0, ((uint64_t)A.e3 << 32) | A.e2};
    sb = Fe{((uint64_t)B.s1 << 32) | B.s0, ((uint64_t)B.s3 << 32) | B.s2};
    db = Fe{((uint64_t)B.e1 << 32) | B.e0, ((uint64_t)B.e3 << 32) | B.e2};
}

__device__ __forceinline__ Fe fe_add_asm(Fe a, Fe b) {
    smask_t c;
    uint32_t r0 = a_add_co(lo32(a.lo), lo32(b.lo), c);
    uint32_t r1 = a_addc(hi32(a.lo), hi32(b.lo), c, c);
    uint32_t r2 = a_addc(lo32(a.hi), lo32(b.hi), c, c);
    uint32_t r3 = a_addc(hi32(a.hi), hi32(b.hi), c, c);
    return a_cond_sub_p(r0, r1, r2, r3, c);
}

__device__ __forceinline__ Fe fe_sub_asm(Fe a, Fe b) {
    smask_t bw;
    uint32_t d0 = a_sub_co(lo32(a.lo), lo32(b.lo), bw);
    uint32_t d1 = a_subb(hi32(a.lo), hi32(b.lo), bw, bw);
    uint32_t d2 = a_subb(lo32(a.hi), lo32(b.hi), bw, bw);
    uint32_t d3 = a_subb(hi32(a.hi), hi32(b.hi), bw, bw);
    // add p = [1, 0, 0, PH3] back where the subtraction borrowed: the borrow mask is the carry-in of limb 0 and
    // selects PH3 for limb 3 (5 instructions instead of a full add chain + 4 selects)
    smask_t c;
    uint32_t ph = a_cnd(0u, PH3, bw);
    uint32_t o0 = a_addc(d0, 0u, bw, c);
    uint32_t o1 = a_addc(d1, 0u, c, c);
    uint32_t o2 = a_addc(d2, 0u, c, c);
    uint32_t o3 = a_addc_last(d3, ph, c);
    return Fe{((uint64_t)o1 << 32) | o0, ((uint64_t)o3 << 32) | o2};
}

}  // namespace sc
#endif
