// Host check of div1e9_step of csrc/merkle.cuh: (rem * 2^32 + d) / 10^9 by the 35-bit reciprocal and a sign test, against plain
// 64-bit division on 3 * 10^8 random dividends, on k * 10^9 + {-1, 0, 1}, and at the extremes.    gcc -O2 ... && ./leaf_div1e9_check   (2 s)
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
static inline uint32_t step(uint32_t* rem, uint32_t d) {
    const uint32_t r0 = 1266874890u;
    const uint32_t h = (uint32_t)(((uint64_t)d * r0) >> 32);
    uint64_t S = (uint64_t)(*rem) * r0 + h;
    S += (uint64_t)d << 2;
    uint32_t q = ((*rem) << 2) + (uint32_t)(S >> 32);
    int32_t r = (int32_t)(d - q * 1000000000u);
    const int32_t mask = r >> 31;
    q += (uint32_t)mask; r += mask & 1000000000;
    *rem = (uint32_t)r; return q;
}
static uint64_t rng = 88172645463325252ull;
static uint64_t xs(void) { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; }
int main(void) {
    uint64_t bad = 0, n = 0;
    for (uint64_t it = 0; it < 300000000ull; ++it) {
        uint32_t rem = (uint32_t)(xs() % 1000000000ull), d = (uint32_t)xs();
        if ((it & 7) == 0) {          // boundary cases: cur = k * 10^9 + {-1, 0, 1}
            uint64_t k = xs() & 0xFFFFFFFFull;
            uint64_t cur = k * 1000000000ull + (uint64_t)((int)(it >> 3) % 3 - 1);
            if ((int64_t)cur < 0) cur = 0;
            if ((cur >> 32) >= 1000000000ull) continue;
            rem = (uint32_t)(cur >> 32); d = (uint32_t)cur;
        }
        const uint64_t cur = ((uint64_t)rem << 32) | d;
        uint32_t r = rem;
        const uint32_t q = step(&r, d);
        ++n;
        if (q != cur / 1000000000ull || r != cur % 1000000000ull) { if (bad < 5) printf("bad cur=%llu q=%u r=%u\n", (unsigned long long)cur, q, r); ++bad; }
    }
    // extremes
    uint32_t rem = 999999999u; uint32_t q = step(&rem, 0xFFFFFFFFu);
    uint64_t cur = (999999999ull << 32) | 0xFFFFFFFFull;
    if (q != cur / 1000000000ull || rem != cur % 1000000000ull) { printf("bad extreme\n"); ++bad; }
    rem = 0; q = step(&rem, 0xFFFFFFFFu); if (q != 4 || rem != 294967295u) { printf("bad rem0\n"); ++bad; }
    printf("%llu cases, %llu bad\n", (unsigned long long)n, (unsigned long long)bad);
    return bad != 0;
}
