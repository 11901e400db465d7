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
