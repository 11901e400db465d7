// Host model of leaf_message_lds (csrc/merkle.cuh): the decimal ASCII of a 128-bit residue by the same steps -- thirteen div1e9_step
// divisions into five groups, fixed-point digits collected four to a big-endian word, byte swap, '0' added, the 39 characters
// shifted by the number of leading zeros -- printed next to the value in hexadecimal; tests/test_hostcheck.py compares every line
// with Python's str().  Values: 0, the powers of ten and their neighbours, p - 1, 2^128 - 1 (the function does not need p), random.
//   gcc -O2 -o leaf_message_model leaf_message_model.c && ./leaf_message_model
#include <stdint.h>
#include <stdio.h>
#include <string.h>
typedef unsigned __int128 u128;
static uint32_t div1e9_step(uint32_t* rem, uint32_t d) {
    const uint32_t R0 = 1266874890u;
    uint64_t s = (uint64_t)(*rem) * R0 + (uint32_t)(((uint64_t)d * R0) >> 32);
    s += (uint64_t)d << 2;
    uint32_t q = ((*rem) << 2) + (uint32_t)(s >> 32);
    int32_t r = (int32_t)(d - q * 1000000000u);
    const int32_t over = r >> 31;
    q += (uint32_t)over; r += over & 1000000000;
    *rem = (uint32_t)r; return q;
}
static uint32_t ndigits9(uint32_t x) { return 1u + (x >= 10u) + (x >= 100u) + (x >= 1000u) + (x >= 10000u) + (x >= 100000u) + (x >= 1000000u) + (x >= 10000000u) + (x >= 100000000u); }
static uint32_t bswap(uint32_t v) { return (v >> 24) | ((v >> 8) & 0xFF00u) | ((v << 8) & 0xFF0000u) | (v << 24); }
static uint32_t leaf_message_model(u128 x, char out[80]) {
    uint32_t d[4] = {(uint32_t)x, (uint32_t)(x >> 32), (uint32_t)(x >> 64), (uint32_t)(x >> 96)}, grp[5];
    for (int g = 0; g < 4; ++g) {
        const int top = (g <= 1) ? 3 : (g == 2 ? 2 : 1);
        uint32_t rem = 0;
        for (int i = top; i >= 0; --i) d[i] = div1e9_step(&rem, d[i]);
        grp[g] = rem;
    }
    grp[4] = d[0];
    uint32_t dw[10], acc = 0, f;
    int pos = 0;
#define FIRST(digit) do { acc = (acc << 8) | (digit); if ((pos & 3) == 3) dw[pos >> 2] = acc; ++pos; } while (0)
#define NEXT() do { const uint64_t u = (uint64_t)f * 10u + ((uint64_t)(acc << 8) << 32); f = (uint32_t)u; acc = (uint32_t)(u >> 32); if ((pos & 3) == 3) dw[pos >> 2] = acc; ++pos; } while (0)
    { const uint64_t t = (uint64_t)grp[4] * 42949673u; f = (uint32_t)t; FIRST((uint32_t)(t >> 32)); NEXT(); NEXT(); }
    for (int g = 3; g >= 0; --g) {
        const uint64_t t = ((uint64_t)grp[g] * 1441151881u + (1u << 25)) >> 25;
        f = (uint32_t)t; FIRST((uint32_t)(t >> 32));
        for (int k = 0; k < 8; ++k) NEXT();
    }
    dw[9] = acc << 8;
    uint32_t lead = grp[0], below = 0;
    if (grp[1]) { lead = grp[1]; below = 9; }
    if (grp[2]) { lead = grp[2]; below = 18; }
    if (grp[3]) { lead = grp[3]; below = 27; }
    if (grp[4]) { lead = grp[4]; below = 36; }
    const uint32_t nd = lead ? below + ndigits9(lead) : 1u;
    uint8_t slot[80];
    memset(slot, 0, sizeof slot);
    for (int i = 0; i < 10; ++i) {
        const uint32_t w = bswap(dw[i] + (i == 9 ? 0x30303000u : 0x30303030u));
        memcpy(slot + 4 * i, &w, 4);
    }
    memcpy(out, slot + (39u - nd), 40);                // the message's five words; everything behind the string is zero
    for (uint32_t i = nd; i < 40; ++i) if (out[i]) return 0;
    return nd;
}
static uint64_t rng = 0x9E3779B97F4A7C15ull;
static uint64_t xs(void) { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; }
static void emit(u128 x) {
    char out[80];
    const uint32_t nd = leaf_message_model(x, out);
    printf("%016llx%016llx %u %.*s\n", (unsigned long long)(x >> 64), (unsigned long long)x, nd, (int)nd, out);
}
int main(void) {
    emit(0);
    u128 pw = 1;
    for (int k = 0; k <= 38; ++k) { emit(pw - 1); emit(pw); emit(pw + 1); if (k < 38) pw *= 10; }
    emit(((u128)407 << 119));                          // p - 1
    emit(~(u128)0);
    for (int i = 0; i < 200000; ++i) {
        u128 x = ((u128)xs() << 64) | xs();
        const int cut = (int)(xs() % 129);             // every length
        emit(cut >= 128 ? x : cut == 0 ? 0 : x >> (128 - cut));
    }
    return 0;
}
