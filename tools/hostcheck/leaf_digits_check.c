// Host check of the fixed-point digit extraction of csrc/merkle.cuh (leaf_message_lds): for EVERY y < 10^9 the nine digits read off
// ((y * ceil(2^57 / 10^8) + 2^25) >> 25 as a 32.32 number, then fraction * 10 eight times) are y's; and the three digits of g < 1000
// from g * ceil(2^32 / 100).    gcc -O2 -o leaf_digits_check leaf_digits_check.c && ./leaf_digits_check   (9 s)
#include <stdint.h>
#include <stdio.h>
int main(void) {
    const uint64_t K = 1441151881ull;   // ceil(2^57 / 10^8)
    uint64_t bad = 0;
    for (uint64_t y = 0; y < 1000000000ull; ++y) {
        uint64_t t = ((y * K) >> 25) + 1;
        uint32_t d[9];
        d[0] = (uint32_t)(t >> 32);
        uint32_t f = (uint32_t)t;
        for (int i = 1; i < 9; ++i) { uint64_t u = (uint64_t)f * 10u; d[i] = (uint32_t)(u >> 32); f = (uint32_t)u; }
        uint32_t v = 0;
        for (int i = 0; i < 9; ++i) { if (d[i] > 9) { bad++; break; } v = v * 10 + d[i]; }
        if (v != y) { if (bad < 5) printf("mismatch y=%llu got %u\n", (unsigned long long)y, v); bad++; }
    }
    printf("9-digit path: %llu bad\n", (unsigned long long)bad);
    // three digits of g < 1000: t = g * ceil(2^32/100)
    const uint64_t K3 = 42949673ull;
    bad = 0;
    for (uint32_t g = 0; g < 1000; ++g) {
        uint64_t t = g * K3;
        uint32_t d0 = (uint32_t)(t >> 32), f = (uint32_t)t;
        uint64_t u = (uint64_t)f * 10u; uint32_t d1 = (uint32_t)(u >> 32); f = (uint32_t)u;
        u = (uint64_t)f * 10u; uint32_t d2 = (uint32_t)(u >> 32);
        if (d0 * 100 + d1 * 10 + d2 != g) { bad++; if (bad < 5) printf("g=%u -> %u%u%u\n", g, d0, d1, d2); }
    }
    printf("3-digit path: %llu bad\n", (unsigned long long)bad);
    return 0;
}
