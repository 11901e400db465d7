// merkle.cuh -- BLAKE2b-512 Merkle tree kernels (reference code/merkle.py:6-27).
//
//   leaf  = BLAKE2b-512( str(value).encode() )      decimal ASCII, no padding (code/algebra.py:53-57, merkle.py:14)
//   node  = BLAKE2b-512( left || right )            one 128-byte block (merkle.py:11)
//
// One thread = one compression (64-bit ARX, 12 rounds, fully unrolled: sigma is compile-time).  The whole
// tree is kept in HBM (level 0 = leaf digests, then N/2, ..., root; (2N-1) * 64 bytes) so that
// Merkle.open is a gather of log2 N digests instead of the reference's rebuild-per-open.
#pragma once
#ifndef SC_MERKLE_4LANE
#define SC_MERKLE_4LANE 1      // narrow tree levels: four lanes per BLAKE2b compression (0: one lane per hash everywhere)
#endif
#ifndef SC_FOUR_LANE_FROM
#define SC_FOUR_LANE_FROM 128    // merkle_subtree_kernel<*, FOUR_LANE>: the width (digests) from which the levels run four lanes per hash.  256 (the 128-parent
                                 // level in two sweeps of 64 instead of one lane per hash) measured no better: the eight-level climb 23.1 us against 22.5, Fri.prove +-0
#endif
#include "field.cuh"

namespace sc {

#if defined(__HIPCC__)

__device__ __constant__ const uint64_t B2_IV[8] = {
    0x6A09E667F3BCC908ull, 0xBB67AE8584CAA73Bull, 0x3C6EF372FE94F82Bull, 0xA54FF53A5F1D36F1ull,
    0x510E527FADE682D1ull, 0x9B05688C2B3E6C1Full, 0x1F83D9ABFB41BD6Bull, 0x5BE0CD19137E2179ull};

// 64-bit rotate right on the 32-bit halves: two v_alignbit_b32 (4 issue cycles each).  The compiler's expansion of the i64
// shift/or idiom is ~3.4 instructions (64-bit shifts included) per rotation -- see profiles/r01/merkle_pmc.md.
template <int R> __device__ __forceinline__ uint64_t rotr64c(uint64_t x) {
    uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    if (R == 32) return ((uint64_t)lo << 32) | hi;
    if (R < 32) {
        uint32_t nl = __builtin_amdgcn_alignbit(hi, lo, R), nh = __builtin_amdgcn_alignbit(lo, hi, R);
        return ((uint64_t)nh << 32) | nl;
    }
    uint32_t nl = __builtin_amdgcn_alignbit(lo, hi, R - 32), nh = __builtin_amdgcn_alignbit(hi, lo, R - 32);
    return ((uint64_t)nh << 32) | nl;
}
#define rotr64(x, r) rotr64c<r>(x)

#define B2_G(a, b, c, d, x, y)                   \
    do {                                         \
        v[a] = v[a] + v[b] + (x);                \
        v[d] = rotr64(v[d] ^ v[a], 32);          \
        v[c] = v[c] + v[d];                      \
        v[b] = rotr64(v[b] ^ v[c], 24);          \
        v[a] = v[a] + v[b] + (y);                \
        v[d] = rotr64(v[d] ^ v[a], 16);          \
        v[c] = v[c] + v[d];                      \
        v[b] = rotr64(v[b] ^ v[c], 63);          \
    } while (0)

#define B2_ROUND(s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15) \
    B2_G(0, 4, 8, 12, m[s0], m[s1]);   B2_G(1, 5, 9, 13, m[s2], m[s3]);                 \
    B2_G(2, 6, 10, 14, m[s4], m[s5]);  B2_G(3, 7, 11, 15, m[s6], m[s7]);                \
    B2_G(0, 5, 10, 15, m[s8], m[s9]);  B2_G(1, 6, 11, 12, m[s10], m[s11]);              \
    B2_G(2, 7, 8, 13, m[s12], m[s13]); B2_G(3, 4, 9, 14, m[s14], m[s15]);

// Single-block unkeyed BLAKE2b-512 of a message of `len` <= 128 bytes held zero-padded in m[16].
__device__ __forceinline__ void blake2b_single_block(const uint64_t m[16], uint32_t len, uint64_t h[8]) {
    uint64_t v[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) { h[i] = B2_IV[i]; }
    h[0] ^= 0x01010040ull;       // digest 64, key 0, fanout 1, depth 1
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = h[i]; v[i + 8] = B2_IV[i]; }
    v[12] ^= (uint64_t)len;
    v[14] = ~v[14];              // final block
    B2_ROUND(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
    B2_ROUND(14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3)
    B2_ROUND(11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4)
    B2_ROUND(7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8)
    B2_ROUND(9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13)
    B2_ROUND(2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9)
    B2_ROUND(12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11)
    B2_ROUND(13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10)
    B2_ROUND(6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5)
    B2_ROUND(10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0)
    B2_ROUND(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
    B2_ROUND(14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3)
#pragma unroll
    for (int i = 0; i < 8; ++i) h[i] ^= v[i] ^ v[i + 8];
}

// ---- four lanes per hash ---------------------------------------------------------------------------------------------
// Levels narrower than the machine are pure latency: a lone wave issues one VALU instruction per ~5 cycles and a DEPENDENT one
// every ~8.7 (profiles/r06/blake2b_quad_ubench.txt), whatever the level's width.  There the state is spread over a quad: lane j
// holds column j (a, b, c, d) = (v[j], v[4+j], v[8+j], v[12+j]) and runs ONE G per half-round instead of four.  What is written
// below is written for the LENGTH OF THE DEPENDENT CHAIN of a compression, which is all a narrow level's time is made of:
//   * b -- the LAST value a G produces and the first the next G consumes -- never changes lanes: between the column and the
//     diagonal step the quad rotates a, c and d instead (DPP quad_perm, no LDS), so lane L runs diagonal L - 1 (its message words
//     are picked with `shd`), and no DPP move stands between one G and the next;
//   * a + b + x is (a + x) + b: a is final five steps before b, so a + x is formed early, off the chain.  hipcc re-associates plain
//     C++ into (b + x) + a -- two adds behind b -- so the EARLY add is an asm statement (b2_add64_early; the wait state the hazard
//     recogniser puts behind an asm definition lands a dozen instructions before its use) and only the late add is C++;
//   * a rotated word is put together as a two-element vector (rotr64v): written as (hi << 32) | lo, hipcc adds the two halves of a
//     freshly rotated d to c one after the other (c + zext(lo) + (hi << 32): a seventh 64-bit add per G, on the chain).
// 584 VALU instructions per lane and compression; 2 960 cycles per compression in a lone wave (tools/microbench/blake2b_quad_ubench.hip:
// 3 290 before these three; bit-identical digests).
template <int CTRL> __device__ __forceinline__ uint64_t quad_perm64(uint64_t x) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)x, CTRL, 0xF, 0xF, true);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)(x >> 32), CTRL, 0xF, 0xF, true);
    return ((uint64_t)hi << 32) | lo;
}
typedef uint32_t b2_u32x2 __attribute__((ext_vector_type(2)));
template <int R> __device__ __forceinline__ uint64_t rotr64v(uint64_t x) {
    if (R == 32) return ((uint64_t)(uint32_t)x << 32) | (uint32_t)(x >> 32);       // (a renaming of registers as long as it is written this way)
    const b2_u32x2 v = __builtin_bit_cast(b2_u32x2, x);
    b2_u32x2 r;
    if (R < 32) { r.x = __builtin_amdgcn_alignbit(v.y, v.x, R); r.y = __builtin_amdgcn_alignbit(v.x, v.y, R); }
    else { r.x = __builtin_amdgcn_alignbit(v.x, v.y, R - 32); r.y = __builtin_amdgcn_alignbit(v.y, v.x, R - 32); }
    return __builtin_bit_cast(uint64_t, r);
}
__device__ __forceinline__ uint64_t b2_add64_early(uint64_t p, uint64_t q) {
    uint64_t r;
    asm("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(r) : "v"(p), "v"(q));
    return r;
}

// One G on (a, b, c, d) with `an` = a + x formed beforehand.  Leaves an = (a moved by PA) + xnext (just a, moved, after the LAST G of a
// compression), c and d moved by PC / PD, b where it is.
#define B2_G4(y, PA, PC, PD, xnext, LASTG)                 \
    do {                                                   \
        a = an + b;                                        \
        d = rotr64v<32>(d ^ a);                            \
        const uint64_t ay_ = b2_add64_early(a, (y));       \
        c = c + d;                                         \
        b = rotr64v<24>(b ^ c);                            \
        a = ay_ + b;                                       \
        d = rotr64v<16>(d ^ a);                            \
        an = quad_perm64<PA>(a);                           \
        if (!(LASTG)) an = b2_add64_early(an, (xnext));    \
        c = c + d;                                         \
        b = rotr64v<63>(b ^ c);                            \
        d = quad_perm64<PD>(d);                            \
        c = quad_perm64<PC>(c);                            \
    } while (0)

// The message words of a round -- two per G, per-lane addresses from the packed sigma constants: COL/DIA pack, per round, byte j =
// sigma[2j] | sigma[2j+1] << 4 resp. sigma[8+2j] | sigma[9+2j] << 4; a lane reads the column byte at `sh` and the diagonal byte at `shd`
// -- are REQUESTED from LDS one round ahead and waited for in the middle of the round before theirs.  Written as loads in C++ the compiler
// sinks them to their first use (register pressure), and every round then exposes an LDS round trip on the dependent chain; so the four
// ds_read_b64 are one asm statement (the hardware counts them in lgkmcnt like the compiler's own: its waits only become stricter), and
// the wait is an asm statement the words pass THROUGH, so nothing that uses them can move above it.  msg must be an LDS address (the
// low half of its flat address is the LDS offset).
#define B2_MSG4_REQUEST(COL, DIA, X0, Y0, X1, Y1)                                    \
    do {                                                                             \
        const uint32_t bc_ = ((uint32_t)(COL) >> sh) & 0xFFu, bd_ = ((uint32_t)(DIA) >> shd) & 0xFFu; \
        const uint32_t a0_ = mbase + ((bc_ & 15u) << 3), a1_ = mbase + ((bc_ >> 4) << 3), a2_ = mbase + ((bd_ & 15u) << 3), a3_ = mbase + ((bd_ >> 4) << 3); \
        /* (`an` passes through the request and `b` through the wait: a G starts with an and ends with b, so the compiler can   */ \
        /* neither move a round's arithmetic above the request nor the wait above the column step in front of it)                */ \
        asm volatile("ds_read_b64 %0, %5\n\tds_read_b64 %1, %6\n\tds_read_b64 %2, %7\n\tds_read_b64 %3, %8"           \
                     : "=&v"(X0), "=&v"(Y0), "=&v"(X1), "=&v"(Y1), "+v"(an) : "v"(a0_), "v"(a1_), "v"(a2_), "v"(a3_) : "memory"); \
    } while (0)
#define B2_MSG4_ARRIVED(X0, Y0, X1, Y1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(X0), "+v"(Y0), "+v"(X1), "+v"(Y1), "+v"(b))
// one round, given the NEXT round's sigma constants.  Column step in lane j = column j; then a comes from lane j - 1, c from j + 1, d from
// j + 2 (frame of the diagonal step: lane L holds a[L-1], b[L], c[L+1], d[L+2] = diagonal L - 1); after it a from lane j + 1, c from
// j - 1, d from j + 2 (columns again)
#define B2_ROUND4_NEXT(NCOL, NDIA)                                                   \
    do {                                                                             \
        uint64_t nx0, ny0, nx1, ny1;                                                 \
        B2_MSG4_REQUEST(NCOL, NDIA, nx0, ny0, nx1, ny1);                             \
        B2_G4(my0, 0x93, 0x39, 0x4E, mx1, false);                                    \
        B2_MSG4_ARRIVED(nx0, ny0, nx1, ny1);                                         \
        B2_G4(my1, 0x39, 0x93, 0x4E, nx0, false);                                    \
        mx0 = nx0; my0 = ny0; mx1 = nx1; my1 = ny1;                                  \
    } while (0)
// the twelve rounds (sigma of rounds 10 and 11 = sigma of rounds 0 and 1); expects a, b, c, d, sh (= 8 j) and msg in scope
#define B2_ROUNDS4()                                                                 \
    do {                                                                             \
        const uint32_t mbase = (uint32_t)(uintptr_t)(msg);                           \
        const uint32_t shd = (sh + 24u) & 31u;          /* 8 * ((j + 3) & 3): lane j runs diagonal j - 1 */ \
        uint64_t mx0, my0, mx1, my1;                                                 \
        uint64_t an = a;                                                             \
        B2_MSG4_REQUEST(0x76543210u, 0xfedcba98u, mx0, my0, mx1, my1);               \
        B2_MSG4_ARRIVED(mx0, my0, mx1, my1);                                         \
        an = b2_add64_early(a, mx0);                                                 \
        B2_ROUND4_NEXT(0x6df984aeu, 0x357b20c1u); B2_ROUND4_NEXT(0xdf250c8bu, 0x491763eau); B2_ROUND4_NEXT(0xebcd1397u, 0x8f04a562u); \
        B2_ROUND4_NEXT(0xfa427509u, 0xd386cb1eu); B2_ROUND4_NEXT(0x38b0a6c2u, 0x91ef57d4u); B2_ROUND4_NEXT(0xa4def15cu, 0xb8293670u); \
        B2_ROUND4_NEXT(0x931ce7bdu, 0xa2684f05u); B2_ROUND4_NEXT(0x803b9ef6u, 0x5a417d2cu); B2_ROUND4_NEXT(0x5167482au, 0x0dc3e9bfu); \
        B2_ROUND4_NEXT(0x76543210u, 0xfedcba98u); B2_ROUND4_NEXT(0x6df984aeu, 0x357b20c1u);                                           \
        B2_G4(my0, 0x93, 0x39, 0x4E, mx1, false);                                    \
        B2_G4(my1, 0x39, 0x93, 0x4E, 0ull, true);                                    \
        a = an;                                          /* (back in its column) */  \
    } while (0)

// single-block BLAKE2b-512 of the 128-byte message msg[0..16) (LDS), computed by the 4 lanes j = 0..3 of a quad (all four must
// be active).  Lane j returns digest words j (h_lo) and 4 + j (h_hi).
__device__ __forceinline__ void blake2b_node_4lane(const uint64_t* msg, uint32_t j, uint64_t& h_lo, uint64_t& h_hi) {
    const uint32_t sh = 8u * j;
    const uint64_t iv_a = B2_IV[j], iv_b = B2_IV[4 + j];
    const uint64_t h0 = (j == 0) ? (iv_a ^ 0x01010040ull) : iv_a;
    uint64_t a = h0, b = iv_b, c = iv_a, d = iv_b;
    if (j == 0) d ^= 128ull;          // t0 = message length
    if (j == 2) d = ~d;               // final block
    B2_ROUNDS4();
    h_lo = h0 ^ a ^ c;
    h_hi = iv_b ^ b ^ d;
}

// LDS layout of a level for the 4-lane path: digest n at 64-bit word (n >> 1) * 17 + (n & 1) * 8, i.e. a parent's two children
// are 16 contiguous words (its message) and consecutive messages are 136 bytes apart (spreads the quads over the banks)
__device__ __forceinline__ uint32_t lin_off(uint32_t n) { return (n >> 1) * 17u + (n & 1u) * 8u; }

// one level by the 4-lane path: `parents` nodes from the 2*parents digests in `src` (lin layout) into `dst` (lin layout) and to
// the tree (`out`: the level's place in global memory).  Threads >= 4*parents idle as whole quads; a level wider than the workgroup's
// `quads` (threads / 4) is taken in sweeps of that many nodes.
__device__ __forceinline__ void merkle_level_4lane(const uint64_t* src, uint64_t* dst, uint64_t* __restrict__ out, uint32_t parents, uint32_t t, uint32_t quads) {
    const uint32_t j = t & 3u;
    for (uint32_t n = t >> 2; n < parents; n += quads) {
        uint64_t lo, hi;
        blake2b_node_4lane(src + 17u * n, j, lo, hi);
        dst[lin_off(n) + j] = lo;
        dst[lin_off(n) + 4u + j] = hi;
        out[8u * n + j] = lo;
        out[8u * n + 4u + j] = hi;
    }
}

__device__ __forceinline__ uint32_t ndigits9(uint32_t x) {   // decimal digits of x < 10^9, x > 0
    return 1u + (x >= 10u) + (x >= 100u) + (x >= 1000u) + (x >= 10000u) + (x >= 100000u) + (x >= 1000000u) + (x >= 10000000u) + (x >= 100000000u);
}

__device__ __forceinline__ uint64_t sel6(const uint64_t W[6], uint32_t idx) {
    uint64_t r = 0;
    r = idx == 0 ? W[0] : r; r = idx == 1 ? W[1] : r; r = idx == 2 ? W[2] : r;
    r = idx == 3 ? W[3] : r; r = idx == 4 ? W[4] : r; r = idx == 5 ? W[5] : r;
    return r;
}

// Decimal ASCII of a canonical residue into the first words of a zeroed BLAKE2b block; returns the length.
__device__ __forceinline__ uint32_t leaf_message(Fe x, uint64_t m[16]) {
#ifdef SC_LEAF_ABLATION   // measurement only (wrong digests): what the tree would cost if the decimal conversion were free
#pragma unroll
    for (int i = 0; i < 16; ++i) m[i] = 0;
    m[0] = x.lo; m[1] = x.hi; m[2] = x.lo ^ 0x3030303030303030ull; m[3] = x.hi | 0x3030303030303030ull; m[4] = x.lo + x.hi;
    return 39;
#endif
    uint32_t d[4] = {(uint32_t)x.lo, (uint32_t)(x.lo >> 32), (uint32_t)x.hi, (uint32_t)(x.hi >> 32)};
    uint32_t grp[5];   // base-10^9 digits, least significant first
    // Long division by 10^9, limb by limb.  The quotient loses ~29.9 bits per group, so for ANY 128-bit input its top limbs
    // are known to be zero: after group 1 the quotient is < 2^68.2 (limb 3 gone), after group 2 < 2^38.3 (limb 2 gone), after
    // group 3 < 2^8.4 -- that is the last group itself.  13 division steps instead of 20.
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int top = (g <= 1) ? 3 : (g == 2 ? 2 : 1);
        uint64_t rem = 0;
#pragma unroll
        for (int i = 3; i >= 0; --i) {
            if (i > top) continue;
            uint64_t cur = (rem << 32) | d[i];
            uint64_t q = cur / 1000000000ull;
            rem = cur - q * 1000000000ull;
            d[i] = (uint32_t)q;
        }
        grp[g] = (uint32_t)rem;
    }
    grp[4] = d[0];
    // 45 characters, most significant first, packed little-endian into W
    uint64_t W[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int g = 4; g >= 0; --g) {
        uint32_t y = grp[g];
#pragma unroll
        for (int j = 8; j >= 0; --j) {
            uint32_t digit = 0;
            if (g < 4 || j >= 6) {              // the top group is < 2^8.4: at most three digits
                uint32_t q = y / 10u;
                digit = y - q * 10u;
                y = q;
            }
            const int pos = (4 - g) * 9 + j;
            W[pos >> 3] |= (uint64_t)(0x30u + digit) << (8 * (pos & 7));
        }
    }
    uint32_t nd = 1;
    if (grp[4]) nd = 36 + ndigits9(grp[4]);
    else if (grp[3]) nd = 27 + ndigits9(grp[3]);
    else if (grp[2]) nd = 18 + ndigits9(grp[2]);
    else if (grp[1]) nd = 9 + ndigits9(grp[1]);
    else if (grp[0]) nd = ndigits9(grp[0]);
    const uint32_t z = 45u - nd;               // leading zero characters to drop
    const uint32_t zw = z >> 3, zb = (z & 7u) * 8u;
#pragma unroll
    for (int i = 0; i < 16; ++i) m[i] = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        uint64_t lo = sel6(W, i + zw), hi = sel6(W, i + zw + 1);
        m[i] = zb ? ((lo >> zb) | (hi << (64u - zb))) : lo;
    }
    return nd;
}

// The same message with two of the conversion's three parts done differently (the throughput-bound tree kernels; ~35 % fewer
// instructions than leaf_message, whose 45-character buffer is shifted by a chain of 120 selects):
//   digits   a group y < 10^9 becomes the 32.32 fixed-point number (y * ceil(2^57 / 10^8) + 2^25) >> 25 = y / 10^8 rounded up in its
//            last place: the integer part is the first digit, and every further digit is the integer part of fraction * 10 -- ONE
//            v_mad_u64_u32 per digit, whose 64-bit addend carries the characters collected so far, shifted by a byte, into the high
//            word next to the new digit (checked against all 10^9 values of y on the host: tools/hostcheck/leaf_digits_check.c);
//   shift    the 39 characters (most significant first) go to this thread's 80 bytes of LDS followed by zeros, and the message is
//            read back from byte offset z = the number of leading '0' characters -- unaligned ds_read_b64, no select, no funnel.
// `slot`: 80 bytes of LDS owned by this thread, 16-byte aligned.
// One step of the long division by 10^9: (rem * 2^32 + d) / 10^9 -> quotient (< 2^32), rem <- remainder; rem < 10^9 in and out.
// q' = floor(cur * R / 2^64) with R = ceil(2^64 / 10^9) = 4 * 2^32 + 1266874890 is the quotient or one more (R's excess adds less
// than 0.068 to cur / 10^9 for cur < 10^9 * 2^32), so the remainder's low word, read as signed, says which: ten instructions
// where the compiler's 64-bit division by a constant takes fifteen (tools/hostcheck/leaf_div1e9_check.c: 3 * 10^8 cases and the boundaries).
__device__ __forceinline__ uint32_t div1e9_step(uint32_t& rem, uint32_t d) {
    constexpr uint32_t R0 = 1266874890u;
    uint64_t s = (uint64_t)rem * R0 + __umulhi(d, R0);
    s += (uint64_t)d << 2;
    uint32_t q = (rem << 2) + (uint32_t)(s >> 32);
    int32_t r = (int32_t)(d - q * 1000000000u);
    const int32_t over = r >> 31;                      // all ones when q is one too many
    q += (uint32_t)over;
    r += over & 1000000000;
    rem = (uint32_t)r;
    return q;
}

#ifndef SC_LEAF_LDS
#define SC_LEAF_LDS 1            // 0: the tree kernels convert their leaves with leaf_message (A/B builds)
#endif
constexpr uint32_t LEAF_SLOT_BYTES = 80;
__device__ __forceinline__ uint32_t leaf_message_lds(Fe x, uint64_t m[16], uint8_t* slot) {
    uint32_t d[4] = {(uint32_t)x.lo, (uint32_t)(x.lo >> 32), (uint32_t)x.hi, (uint32_t)(x.hi >> 32)};
    uint32_t grp[5];   // base-10^9 digits, least significant first (the long division of leaf_message: 13 steps)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int top = (g <= 1) ? 3 : (g == 2 ? 2 : 1);
        uint32_t rem = 0;
#pragma unroll
        for (int i = 3; i >= 0; --i) {
            if (i > top) continue;
            d[i] = div1e9_step(rem, d[i]);
        }
        grp[g] = rem;
    }
    grp[4] = d[0];                                     // < 341: three digits
    // 39 characters, most significant first; acc collects them big-endian, a dword is complete every fourth
    uint32_t dw[10];
    uint32_t acc = 0, f;
    int pos = 0;
#define LEAF_FIRST(digit) do { acc = (acc << 8) | (digit); if ((pos & 3) == 3) dw[pos >> 2] = acc; ++pos; } while (0)
#define LEAF_NEXT() do { const uint64_t u_ = (uint64_t)f * 10u + ((uint64_t)(acc << 8) << 32); f = (uint32_t)u_; acc = (uint32_t)(u_ >> 32); \
                         if ((pos & 3) == 3) dw[pos >> 2] = acc; ++pos; } while (0)
    {
        const uint64_t t = (uint64_t)grp[4] * 42949673u;          // ceil(2^32 / 100): integer part = the hundreds
        f = (uint32_t)t;
        LEAF_FIRST((uint32_t)(t >> 32));
        LEAF_NEXT(); LEAF_NEXT();
    }
#pragma unroll
    for (int g = 3; g >= 0; --g) {
        const uint64_t t = ((uint64_t)grp[g] * 1441151881u + (1u << 25)) >> 25;
        f = (uint32_t)t;
        LEAF_FIRST((uint32_t)(t >> 32));
#pragma unroll
        for (int k = 0; k < 8; ++k) LEAF_NEXT();
    }
    dw[9] = acc << 8;                                   // characters 36..38 and the byte behind the string
#undef LEAF_FIRST
#undef LEAF_NEXT
    // the length: the digits of the most significant non-zero group (counted once) on top of the groups below it
    uint32_t lead = grp[0], below = 0;
    if (grp[1]) { lead = grp[1]; below = 9; }
    if (grp[2]) { lead = grp[2]; below = 18; }
    if (grp[3]) { lead = grp[3]; below = 27; }
    if (grp[4]) { lead = grp[4]; below = 36; }
    const uint32_t nd = lead ? below + ndigits9(lead) : 1u;
    // to string order (first character in the lowest byte), '0' added to every character; then through LDS, shifted by z bytes
    uint64_t* w = reinterpret_cast<uint64_t*>(slot);
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const uint32_t lo = __builtin_amdgcn_perm(0u, dw[2 * i] + 0x30303030u, 0x00010203u);
        const uint32_t hi = __builtin_amdgcn_perm(0u, dw[2 * i + 1] + (i == 4 ? 0x30303000u : 0x30303030u), 0x00010203u);
        w[i] = ((uint64_t)hi << 32) | lo;
    }
#pragma unroll
    for (int i = 5; i < 10; ++i) w[i] = 0;
    const uint8_t* from = slot + (39u - nd);
#pragma unroll
    for (int i = 0; i < 5; ++i) __builtin_memcpy(&m[i], from + 8 * i, 8);
#pragma unroll
    for (int i = 5; i < 16; ++i) m[i] = 0;
    return nd;
}

// LDS staging so that every global access of the tree kernels is a fully coalesced 16-byte-per-lane stream:
// a thread's 128-byte message / 64-byte digest is strided across lanes in memory (8 resp. 4 separate partial-line
// accesses per lane otherwise).  Slots are rotated by the owning thread id to keep the per-thread LDS accesses
// at most 2-way bank-conflicted.
__device__ __forceinline__ uint32_t msg_slot(uint32_t t, uint32_t k) { return (t << 3) | ((k + t) & 7u); }   // 16-byte slots
__device__ __forceinline__ uint32_t dig_slot(uint32_t t, uint32_t k) { return (t << 2) | ((k + t) & 3u); }

// write this workgroup's digests (h of thread t) to out[base .. base + n_here) with coalesced stores
__device__ __forceinline__ void store_digests_coalesced(uint4* sm, const uint64_t h[8], bool active, uint64_t* __restrict__ out, uint64_t base, uint32_t n_here) {
    const uint32_t t = threadIdx.x;
    if (active) {
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) {
            uint4 v;
            v.x = (uint32_t)h[2 * k]; v.y = (uint32_t)(h[2 * k] >> 32); v.z = (uint32_t)h[2 * k + 1]; v.w = (uint32_t)(h[2 * k + 1] >> 32);
            sm[dig_slot(t, k)] = v;
        }
    }
    __syncthreads();
    uint4* dst = reinterpret_cast<uint4*>(out + 8 * base);
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) {
        const uint32_t q = j * 256u + t;           // 16-byte chunk index inside the workgroup's 16 KiB of digests
        if (q < n_here * 4u) dst[q] = sm[dig_slot(q >> 2, q & 3u)];
    }
}

// Priority of a workgroup's ENTRY (its global loads): in a long grid new waves arrive while others hash; at top priority until their
// loads are issued, the loads leave at once instead of queueing behind that arithmetic (the transforms: PassParams::prio_balance = 2).
#ifndef SC_MERKLE_ENTRY_PRIO
#define SC_MERKLE_ENTRY_PRIO 1
#endif
__device__ __forceinline__ void entry_prio(bool on) {
#if SC_MERKLE_ENTRY_PRIO
    if (on) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0);
#endif
}

__global__ void __launch_bounds__(256) merkle_leaf_kernel(const Fe* __restrict__ elems, uint64_t* __restrict__ digests, uint64_t N) {
    __shared__ uint4 sm[256 * 4];
    const uint64_t base = (uint64_t)blockIdx.x * 256u;
    const uint32_t n_here = (uint32_t)((N - base) < 256u ? (N - base) : 256u);
    const bool active = threadIdx.x < n_here;
    uint64_t m[16], h[8];
    if (active) {
        entry_prio(true);
        const Fe e = elems[base + threadIdx.x];
        entry_prio(false);
        uint32_t len = leaf_message(e, m);
        blake2b_single_block(m, len, h);
    }
    store_digests_coalesced(sm, h, active, digests, base, n_here);
}

__device__ __forceinline__ void merkle_node(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, uint64_t i) {
    uint64_t m[16], h[8];
    const ulonglong2* s = reinterpret_cast<const ulonglong2*>(in + 16 * i);
#pragma unroll
    for (int k = 0; k < 8; ++k) { ulonglong2 t = s[k]; m[2 * k] = t.x; m[2 * k + 1] = t.y; }
    blake2b_single_block(m, 128u, h);
    ulonglong2* o = reinterpret_cast<ulonglong2*>(out + 8 * i);
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = make_ulonglong2(h[2 * k], h[2 * k + 1]);
}

// one level: count parents from 2*count children; a workgroup handles 256 consecutive parents
__global__ void __launch_bounds__(256) merkle_level_kernel(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, uint64_t count) {
    __shared__ uint4 sm[256 * 8];                     // 32 KiB: the workgroup's 256 messages of 128 bytes
    const uint32_t t = threadIdx.x;
    const uint64_t base = (uint64_t)blockIdx.x * 256u;
    const uint32_t n_here = (uint32_t)((count - base) < 256u ? (count - base) : 256u);
    const uint4* src = reinterpret_cast<const uint4*>(in + 16 * base);
    entry_prio(true);
#pragma unroll
    for (uint32_t j = 0; j < 8; ++j) {
        const uint32_t q = j * 256u + t;              // coalesced: lane l reads chunk q of the 32 KiB block
        if (q < n_here * 8u) sm[msg_slot(q >> 3, q & 7u)] = src[q];
    }
    entry_prio(false);
    __syncthreads();
    const bool active = t < n_here;
    uint64_t m[16], h[8];
    if (active) {
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) {
            uint4 v = sm[msg_slot(t, k)];
            m[2 * k] = ((uint64_t)v.y << 32) | v.x;
            m[2 * k + 1] = ((uint64_t)v.w << 32) | v.z;
        }
        blake2b_single_block(m, 128u, h);
    }
    __syncthreads();                                   // all messages consumed before the buffer is reused for digests
    store_digests_coalesced(sm, h, active, out, base, n_here);
}

// Fused subtree kernel: a workgroup owns 256 consecutive nodes of level `lvl0` and climbs up to `nlev` further levels of
// THEIR subtree through LDS (256 -> 128 -> ... ), writing every level to its place in the tree (all levels are kept for
// openings).  LEAVES = true: level lvl0 = 0 and its digests are hashed here from the field elements; false: the 256 nodes
// are read from the tree.  One launch replaces up to nlev + 1 dependent launches (each a few microseconds of pure latency).
//   levels : base of the tree; level l starts at digest offset level_off(l) = (l == 0 ? 0 : 2N - (N >> (l-1)))
//   N      : number of leaves of the tree (power of two), width0 = N >> lvl0 nodes at the start level (multiple of 256)
//   FOUR_LANE : latency-bound launch (few workgroups): levels of <= 64 nodes run four lanes per hash; throughput-bound
//               launches keep one lane per hash (the 4-lane form costs ~1.6x the instructions per compression) and the leaner
//               instantiation (fewer registers, 16 KiB of LDS)
// FOLD (with LEAVES): the leaves are not read but COMPUTED -- leaf i is the split-and-fold of fri.py:85 of the previous round's
// codeword, out[i] = (a + b)/2 + (a - b) * c * w^-i with a = in[i], b = in[i + N] (N = this tree's leaves = half of the previous
// length) -- written to the folded codeword and hashed in the same thread: one launch less per FRI round, and the folded
// element never makes a round trip through memory between the two kernels.
struct FoldIn {
    const Fe* in;       // previous codeword, 2N elements
    Fe* out;            // folded codeword, N elements
    const Fe* lo;       // two-level power table of omega^-1
    const Fe* hi;
    Fe c_m;             // alpha / (2 * offset), Montgomery form
    // a rank's column slab [rows][2^logcols] of a codeword that is a [rows][R] matrix (multi-GPU FRI): element i of the slab is
    // codeword index (i >> logcols) * R + col_base + (i & (2^logcols - 1)); the defaults make that index i itself
    int logcols = 63;
    uint64_t R = 0, col_base = 0;
};
__device__ __forceinline__ Fe fold_element(const FoldIn& f, uint64_t i, uint64_t half) {
    const Fe a = f.in[i], b = f.in[i + half];
    const uint64_t e = (i >> f.logcols) * f.R + f.col_base + (i & ((1ull << f.logcols) - 1ull));
    const Fe t = mont_mul(mont_mul(f.lo[e & 4095u], f.hi[e >> 12]), f.c_m);          // (c * w^-e) in Montgomery form
    return fe_add(fe_half(fe_add(a, b)), mont_mul(fe_sub(a, b), t));
}

template <bool LEAVES, bool FOUR_LANE, bool FOLD = false>
__global__ void __launch_bounds__(256) merkle_subtree_kernel(const Fe* __restrict__ elems, uint64_t* __restrict__ levels, uint64_t N, int lvl0, int nlev, const FoldIn fold = FoldIn()) {
    __shared__ uint4 cur[LEAVES ? 256 * 5 : 256 * 4];  // this level's digests of the subtree (16 KiB); before that, the leaf stage's 80 bytes per thread
    constexpr bool four_lane = FOUR_LANE && (SC_MERKLE_4LANE != 0);
    constexpr uint32_t FOUR_LANE_FROM = SC_FOUR_LANE_FROM;      // the width (digests) at which a latency-bound launch hands over to the four-lane form
    __shared__ uint64_t linA[four_lane ? (SC_FOUR_LANE_FROM / 2) * 17 : 1], linB[four_lane ? (SC_FOUR_LANE_FROM / 4) * 17 : 1];   // 4-lane path: FOUR_LANE_FROM resp. half as many digests in the lin layout
    const uint32_t t = threadIdx.x;
    const uint64_t wg = blockIdx.x;
    auto level_off = [N](int l) -> uint64_t { return l == 0 ? 0 : 2 * N - (N >> (l - 1)); };
    uint64_t h[8];
    entry_prio(true);
    if (LEAVES) {
        uint64_t m[16];
        Fe e;
        if constexpr (FOLD) {
            e = fold_element(fold, wg * 256u + t, N);
            fold.out[wg * 256u + t] = e;
        } else {
            e = elems[wg * 256u + t];
        }
        entry_prio(false);
#if SC_LEAF_LDS
        uint32_t len = leaf_message_lds(e, m, reinterpret_cast<uint8_t*>(cur) + LEAF_SLOT_BYTES * t);
        blake2b_single_block(m, len, h);
        __syncthreads();                               // every thread is done with its slot of `cur` before digests are published there
#else
        uint32_t len = leaf_message(e, m);
        blake2b_single_block(m, len, h);
#endif
    } else {
        const ulonglong2* s = reinterpret_cast<const ulonglong2*>(levels + 8 * (level_off(lvl0) + wg * 256u + t));
#pragma unroll
        for (int k = 0; k < 4; ++k) { ulonglong2 v = s[k]; h[2 * k] = v.x; h[2 * k + 1] = v.y; }
        entry_prio(false);
    }
    uint32_t width = 256;
    int l = 0;
    for (;; ++l) {
        // publish this level: LDS for the next level, global for the tree (level lvl0 of a non-leaf launch is already there)
        if (t < width) {
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) {
                uint4 v;
                v.x = (uint32_t)h[2 * k]; v.y = (uint32_t)(h[2 * k] >> 32); v.z = (uint32_t)h[2 * k + 1]; v.w = (uint32_t)(h[2 * k + 1] >> 32);
                cur[dig_slot(t, k)] = v;
            }
            if constexpr (four_lane) {
                if (width == FOUR_LANE_FROM) {
#pragma unroll
                    for (uint32_t w = 0; w < 8; ++w) linA[lin_off(t) + w] = h[w];
                }
            }
        }
        __syncthreads();
        if (LEAVES || l > 0) {
            uint4* dst = reinterpret_cast<uint4*>(levels + 8 * (level_off(lvl0 + l) + wg * width));
            for (uint32_t q = t; q < width * 4u; q += 256u) dst[q] = cur[dig_slot(q >> 2, q & 3u)];     // coalesced
        }
        if (l == nlev) return;
        if constexpr (four_lane) {
            if (width == FOUR_LANE_FROM) break;        // latency-bound launch: the remaining levels go four lanes per hash
        }
        width >>= 1;
        if (t < width) {
            uint64_t m[16];
#pragma unroll
            for (uint32_t k = 0; k < 8; ++k) {
                uint4 v = cur[dig_slot(2 * t + (k >> 2), k & 3u)];
                m[2 * k] = ((uint64_t)v.y << 32) | v.x;
                m[2 * k + 1] = ((uint64_t)v.w << 32) | v.z;
            }
            blake2b_single_block(m, 128u, h);
        }
        __syncthreads();                               // everyone has read `cur` before it is overwritten
    }
    if constexpr (four_lane) {
        uint64_t* src = linA;
        uint64_t* dst = linB;
        for (++l; l <= nlev; ++l) {
            width >>= 1;
            merkle_level_4lane(src, dst, levels + 8 * (level_off(lvl0 + l) + wg * width), width, t, 64u);
            __syncthreads();
            uint64_t* s = src; src = dst; dst = s;
        }
    }
}

// finishes the tree from a level of `width` <= 2048 digests down to the root inside ONE workgroup
// (levels are written once and read only after the barrier, so L1 cannot hold a stale copy).  Levels of <= 256 parents run
// four lanes per hash out of LDS.
// host != nullptr: the root is also written to a pinned, host-coherent slot (8 words, then -- ordered behind them -- the
// sequence number the waiting host polls): the separate publish launch of an asynchronous build is saved.
__device__ __forceinline__ void publish_root(const uint64_t* root, volatile uint64_t* host, uint64_t seq) {
    __threadfence_block();
    __syncthreads();
    if (threadIdx.x < 8) host[threadIdx.x] = root[threadIdx.x];
    __threadfence_system();
    if (threadIdx.x == 0) host[8] = seq;
}

__global__ void __launch_bounds__(1024) merkle_tail_kernel(uint64_t* level, uint64_t width, volatile uint64_t* host = nullptr, uint64_t seq = 0) {
    uint64_t* cur = level;
    uint64_t w = width;
#if SC_MERKLE_4LANE
    __shared__ uint64_t linA[256 * 17], linB[128 * 17];   // 512 resp. 256 digests in the lin layout
    for (; w > 512; w >>= 1) {
#else
    for (; w > 1; w >>= 1) {
#endif
        uint64_t* nxt = cur + 8 * w;
        if (threadIdx.x < (w >> 1)) merkle_node(cur, nxt, threadIdx.x);
        __threadfence_block();
        __syncthreads();
        cur = nxt;
    }
#if SC_MERKLE_4LANE
    if (w <= 1) {
        if (host) publish_root(cur, host, seq);
        return;
    }
    for (uint32_t q = threadIdx.x; q < (uint32_t)w * 8u; q += 1024u) linA[lin_off(q >> 3) + (q & 7u)] = cur[q];
    __syncthreads();
    uint64_t* src = linA;
    uint64_t* dst = linB;
    for (; w > 1; w >>= 1) {
        uint64_t* nxt = cur + 8 * w;
        merkle_level_4lane(src, dst, nxt, (uint32_t)(w >> 1), threadIdx.x, 256u);
        __syncthreads();
        cur = nxt;
        uint64_t* s = src; src = dst; dst = s;
    }
#endif
    if (host) publish_root(cur, host, seq);
}

// authentication paths (merkle.py:16-27): for query q, digest l of the path = level_l[(index >> l) ^ 1]
__global__ void __launch_bounds__(256) merkle_open_kernel(const uint64_t* __restrict__ levels, uint64_t N, int logN,
                                                          const uint64_t* __restrict__ indices, uint64_t k, uint64_t* __restrict__ out) {
    // one thread per (query, level, 16-byte quarter)
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t total = k * (uint64_t)logN * 4;
    if (t >= total) return;
    uint32_t quarter = (uint32_t)(t & 3);
    uint64_t ql = t >> 2;
    uint32_t l = (uint32_t)(ql % (uint64_t)logN);
    uint64_t q = ql / (uint64_t)logN;
    uint64_t idx = indices[q];
    // offset of level l in digests: N + N/2 + ... = 2N - (N >> (l-1)) for l >= 1
    uint64_t off = (l == 0) ? 0 : (2 * N - (N >> (l - 1)));
    uint64_t node = (idx >> l) ^ 1ull;
    const ulonglong2* s = reinterpret_cast<const ulonglong2*>(levels + 8 * (off + node));
    ulonglong2* o = reinterpret_cast<ulonglong2*>(out + 8 * ql);
    o[quarter] = s[quarter];
}

// root of a finished tree -> a pinned, host-coherent slot: 8 words, then (ordered behind them) the sequence number the host polls
__global__ void __launch_bounds__(64) root_publish_kernel(const uint64_t* __restrict__ root, volatile uint64_t* host, uint64_t seq) {
    if (threadIdx.x < 8) host[threadIdx.x] = root[threadIdx.x];
    __threadfence_system();
    if (threadIdx.x == 0) host[8] = seq;
}

// the same for up to QUERY_MAX_TREES (tree, vector) pairs in ONE launch (Fri.prove's query phase: 15-17 trees, ~30 launches
// of a few microseconds each otherwise).  Per opened index: 4 * logN threads copy the path (as above), one more copies the
// opened element.  thread_off / idx_off / path_off are exclusive prefix sums over the pairs.
constexpr int QUERY_MAX_TREES = 32;
struct QueryTree {
    const uint64_t* levels;
    const Fe* elems;
    uint64_t N;
    uint32_t logN;
    uint32_t per_query;      // 4 * logN + 1 threads per opened index
    uint64_t thread_off;     // first thread of this pair
    uint64_t idx_off;        // first index / first output element of this pair
    uint64_t path_off;       // first output digest of this pair
};
struct QueryTrees {
    QueryTree t[QUERY_MAX_TREES];
    uint64_t total_threads;
    int count;
};
// Done (optional): the answers may go STRAIGHT to pinned host memory (elems_out / paths_out are then host pointers: the stores
// cross the bus as they are made, no staging buffer and no copy engine behind the kernel); the workgroup that finishes last -- a
// ticket counter in device memory -- writes `seq` to the host word the caller polls, behind everybody's system-scope fence.
struct QueryDone {
    unsigned* ticket = nullptr;             // device counter, zero before the launch; reset by the last workgroup
    volatile uint64_t* host_flag = nullptr;
    uint64_t seq = 0;
};
__global__ void __launch_bounds__(256) merkle_query_multi_kernel(QueryTrees Q, const uint64_t* __restrict__ indices, Fe* __restrict__ elems_out,
                                                                 uint64_t* __restrict__ paths_out, const QueryDone done = QueryDone()) {
    const uint64_t t_raw = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = t_raw < Q.total_threads;
    const uint64_t t = live ? t_raw : Q.total_threads - 1;
    int w = 0;
    for (int i = 1; i < Q.count; ++i) if (t >= Q.t[i].thread_off) w = i;
    const QueryTree& T = Q.t[w];
    const uint64_t local = t - T.thread_off;
    const uint64_t q = local / T.per_query;
    const uint32_t r = (uint32_t)(local % T.per_query);
    // the 4 logN + 1 threads of an opening share one index; it may live in host memory (uncached across the bus): the first lane of
    // every run of equal openings in the wave loads it, the others take it from that lane
    const uint64_t qn = T.idx_off + q;                        // the opening's number: non-decreasing along the lanes
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t below = __shfl_up(qn, 1);
    const bool leader = lane == 0 || below != qn;
    const uint64_t leaders = __ballot(leader);
    const uint64_t mine = leader ? indices[qn] : 0ull;
    const int src = 63 - __builtin_clzll(leaders & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull)));
    const uint64_t idx = __shfl(mine, src);
    if (live) {
        if (r == T.per_query - 1) {
            elems_out[T.idx_off + q] = T.elems[idx];
        } else {
            const uint32_t quarter = r & 3, l = r >> 2;
            const uint64_t off = (l == 0) ? 0 : (2 * T.N - (T.N >> (l - 1)));
            const uint64_t node = (idx >> l) ^ 1ull;
            const ulonglong2* s = reinterpret_cast<const ulonglong2*>(T.levels + 8 * (off + node));
            ulonglong2* o = reinterpret_cast<ulonglong2*>(paths_out + 8 * (T.path_off + q * T.logN + l));
            o[quarter] = s[quarter];
        }
    }
    if (done.host_flag) {
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned arrived = atomicAdd(done.ticket, 1u);
            if (arrived == gridDim.x - 1) {
                *done.ticket = 0;
                __threadfence_system();
                done.host_flag[0] = done.seq;
            }
        }
    }
}

#endif  // __HIPCC__

}  // namespace sc
