// transcript.h -- the Fiat-Shamir step of Fri.commit on the host, in C++ (no Python between a root arriving and the next launch).
//
// Reference: code/ip.py:18-25  prover_fiat_shamir() = shake_256(pickle.dumps(self.objects)).digest(32)
//            code/algebra.py:116-120  Field.sample(bytes) = big-endian integer of the bytes, mod p
//            code/fri.py:71-79  per round: push(root); alpha = field.sample(proof_stream.prover_fiat_shamir())
//
// pickle.dumps (protocol 4, CPython 3.8+) of a list whose items are all `bytes` objects shorter than 256 bytes, pairwise
// distinct objects (no memo hits), fewer than 1000 of them (one APPENDS batch) and less than 64 KiB in total (one frame) is a
// fixed byte layout; anything else stays with the Python pickler (stark-anatomy_amd/fri.py checks the conditions):
//
//   80 04                      PROTO 4
//   95 <u64 LE frame length>   FRAME            (payload = everything after these 9 bytes)
//   5d 94                      EMPTY_LIST MEMOIZE
//   28                         MARK             (only with >= 2 items)
//   43 <len> <bytes> 94        SHORT_BINBYTES MEMOIZE, per item
//   65 | 61                    APPENDS (>= 2 items) | APPEND (1 item)
//   2e                         STOP
//
// tests/test_host_cpu.py compares these bytes and the challenge with pickle / hashlib for the golden transcripts.
#pragma once
#include <stdint.h>
#include <string.h>
#include <vector>

namespace sc {

inline uint64_t keccak_rotl(uint64_t x, int n) { return n ? (x << n) | (x >> (64 - n)) : x; }

inline void keccak_f1600(uint64_t s[25]) {
    static const uint64_t RC[24] = {0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull, 0x000000000000808bull, 0x0000000080000001ull,
                                    0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000aull,
                                    0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull, 0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull,
                                    0x000000000000800aull, 0x800000008000000aull, 0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
    static const int ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
    for (int round = 0; round < 24; ++round) {
        uint64_t c[5], d[5], b[25];
        for (int x = 0; x < 5; ++x) c[x] = s[x] ^ s[x + 5] ^ s[x + 10] ^ s[x + 15] ^ s[x + 20];
        for (int x = 0; x < 5; ++x) d[x] = c[(x + 4) % 5] ^ keccak_rotl(c[(x + 1) % 5], 1);
        for (int i = 0; i < 25; ++i) s[i] ^= d[i % 5];
        for (int x = 0; x < 5; ++x)
            for (int y = 0; y < 5; ++y) b[y + 5 * ((2 * x + 3 * y) % 5)] = keccak_rotl(s[x + 5 * y], ROT[x + 5 * y]);
        for (int y = 0; y < 5; ++y)
            for (int x = 0; x < 5; ++x) s[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
        s[0] ^= RC[round];
    }
}

// SHAKE-256 (FIPS 202): rate 136 bytes, domain suffix 0x1f
inline void shake256(const uint8_t* in, size_t len, uint8_t* out, size_t outlen) {
    uint64_t s[25];
    memset(s, 0, sizeof s);
    const size_t rate = 136;
    uint8_t block[136];
    while (len >= rate) {
        for (size_t i = 0; i < rate / 8; ++i) { uint64_t w; memcpy(&w, in + 8 * i, 8); s[i] ^= w; }
        keccak_f1600(s);
        in += rate;
        len -= rate;
    }
    memset(block, 0, rate);
    memcpy(block, in, len);
    block[len] ^= 0x1f;
    block[rate - 1] ^= 0x80;
    for (size_t i = 0; i < rate / 8; ++i) { uint64_t w; memcpy(&w, block + 8 * i, 8); s[i] ^= w; }
    keccak_f1600(s);
    while (outlen) {
        const size_t take = outlen < rate ? outlen : rate;
        memcpy(out, s, take);
        out += take;
        outlen -= take;
        if (outlen) keccak_f1600(s);
    }
}


// BLAKE2b-512, unkeyed, of a message of any length (RFC 7693) on the host: Fri.sample_indices hashes `seed + bytes(counter)` with it
// (code/fri.py:36-51).  tests/test_host_cpu.py compares it with hashlib.blake2b.
inline uint64_t b2_rotr(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }
inline void blake2b_compress(uint64_t h[8], const uint8_t block[128], uint64_t t, bool last) {
    static const uint64_t IV[8] = {0x6A09E667F3BCC908ull, 0xBB67AE8584CAA73Bull, 0x3C6EF372FE94F82Bull, 0xA54FF53A5F1D36F1ull,
                                   0x510E527FADE682D1ull, 0x9B05688C2B3E6C1Full, 0x1F83D9ABFB41BD6Bull, 0x5BE0CD19137E2179ull};
    static const uint8_t SIGMA[12][16] = {{0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
                                          {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
                                          {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
                                          {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
                                          {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
                                          {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};
    uint64_t m[16], v[16];
    for (int i = 0; i < 16; ++i) memcpy(&m[i], block + 8 * i, 8);
    for (int i = 0; i < 8; ++i) { v[i] = h[i]; v[i + 8] = IV[i]; }
    v[12] ^= t;
    if (last) v[14] = ~v[14];
    auto G = [&](int a, int b, int c, int d, uint64_t x, uint64_t y) {
        v[a] = v[a] + v[b] + x; v[d] = b2_rotr(v[d] ^ v[a], 32);
        v[c] = v[c] + v[d];     v[b] = b2_rotr(v[b] ^ v[c], 24);
        v[a] = v[a] + v[b] + y; v[d] = b2_rotr(v[d] ^ v[a], 16);
        v[c] = v[c] + v[d];     v[b] = b2_rotr(v[b] ^ v[c], 63);
    };
    for (int r = 0; r < 12; ++r) {
        const uint8_t* s = SIGMA[r];
        G(0, 4, 8, 12, m[s[0]], m[s[1]]);   G(1, 5, 9, 13, m[s[2]], m[s[3]]);
        G(2, 6, 10, 14, m[s[4]], m[s[5]]);  G(3, 7, 11, 15, m[s[6]], m[s[7]]);
        G(0, 5, 10, 15, m[s[8]], m[s[9]]);  G(1, 6, 11, 12, m[s[10]], m[s[11]]);
        G(2, 7, 8, 13, m[s[12]], m[s[13]]); G(3, 4, 9, 14, m[s[14]], m[s[15]]);
    }
    for (int i = 0; i < 8; ++i) h[i] ^= v[i] ^ v[i + 8];
}
inline void blake2b_512(const uint8_t* in, size_t len, uint8_t out[64]) {
    uint64_t h[8] = {0x6A09E667F3BCC908ull ^ 0x01010040ull, 0xBB67AE8584CAA73Bull, 0x3C6EF372FE94F82Bull, 0xA54FF53A5F1D36F1ull,
                     0x510E527FADE682D1ull, 0x9B05688C2B3E6C1Full, 0x1F83D9ABFB41BD6Bull, 0x5BE0CD19137E2179ull};
    uint64_t t = 0;
    while (len > 128) {                    // the last block (also a full one) is compressed with the final flag
        t += 128;
        blake2b_compress(h, in, t, false);
        in += 128;
        len -= 128;
    }
    uint8_t block[128];
    memset(block, 0, 128);
    memcpy(block, in, len);
    blake2b_compress(h, block, t + len, true);
    memcpy(out, h, 64);
}

// Fri.sample_indices (code/fri.py:36-51) for a power-of-two `size`: `number` indices below `size`, pairwise distinct modulo
// `reduced_size`, candidate k = the big-endian integer of blake2b(seed + k zero bytes) mod size.  false: not enough room
// (fri.py:37-38 assert; the caller lets the reference's assertion speak).
inline bool fri_sample_indices(const uint8_t* seed, size_t seed_len, uint64_t size, uint64_t reduced_size, uint32_t number, uint64_t* out) {
    if (number > reduced_size || !size || (size & (size - 1)) || !reduced_size) return false;
    std::vector<uint8_t> msg(seed, seed + seed_len);
    std::vector<uint64_t> residues;
    uint32_t have = 0;
    while (have < number) {
        uint8_t d[64];
        blake2b_512(msg.data(), msg.size(), d);
        msg.push_back(0);                   // bytes(counter) is `counter` zero bytes (fri.py:44)
        uint64_t low = 0;
        for (int i = 0; i < 8; ++i) low = (low << 8) | d[56 + i];      // the integer's low 64 bits: size is a power of two <= 2^63
        const uint64_t index = low & (size - 1), res = index % reduced_size;
        bool seen = false;
        for (uint64_t r : residues) if (r == res) { seen = true; break; }
        if (seen) continue;
        residues.push_back(res);
        out[have++] = index;
    }
    return true;
}

// SHAKE-256 in two steps: whole rate blocks of a prefix absorbed ahead of time, the rest (with the padding) when it is known
inline void shake256_absorb_blocks(uint64_t s[25], const uint8_t* in, size_t nblocks) {
    for (size_t b = 0; b < nblocks; ++b, in += 136) {
        for (size_t i = 0; i < 17; ++i) { uint64_t w; memcpy(&w, in + 8 * i, 8); s[i] ^= w; }
        keccak_f1600(s);
    }
}
inline void shake256_finish(uint64_t s[25], const uint8_t* in, size_t len, uint8_t* out, size_t outlen) {
    const size_t rate = 136;
    shake256_absorb_blocks(s, in, len / rate);
    in += (len / rate) * rate;
    len %= rate;
    uint8_t block[136];
    memset(block, 0, rate);
    memcpy(block, in, len);
    block[len] ^= 0x1f;
    block[rate - 1] ^= 0x80;
    for (size_t i = 0; i < rate / 8; ++i) { uint64_t w; memcpy(&w, block + 8 * i, 8); s[i] ^= w; }
    keccak_f1600(s);
    while (outlen) {
        const size_t take = outlen < rate ? outlen : rate;
        memcpy(out, s, take);
        out += take;
        outlen -= take;
        if (outlen) keccak_f1600(s);
    }
}

constexpr size_t TRANSCRIPT_MAX_ITEMS = 999;        // one APPENDS batch
constexpr size_t TRANSCRIPT_MAX_BYTES = 60000;      // one frame (the pickler starts a new one at 64 KiB)

// pickled item: SHORT_BINBYTES len data MEMOIZE
inline void transcript_item(std::vector<uint8_t>& items, const uint8_t* data, size_t len) {
    items.push_back(0x43);
    items.push_back((uint8_t)len);
    items.insert(items.end(), data, data + len);
    items.push_back(0x94);
}

// pickle.dumps of the list whose pickled items are `items` (count of them); false when the layout above does not apply
inline bool transcript_bytes(const std::vector<uint8_t>& items, size_t count, std::vector<uint8_t>& out) {
    out.clear();
    if (count > TRANSCRIPT_MAX_ITEMS || items.size() > TRANSCRIPT_MAX_BYTES) return false;
    if (count == 0) {
        const uint8_t empty[] = {0x80, 0x04, 0x5d, 0x94, 0x2e};     // too short for a frame
        out.assign(empty, empty + sizeof empty);
        return true;
    }
    const uint64_t payload = 2 + (count >= 2 ? 1 : 0) + items.size() + 1 + 1;
    out.reserve(11 + payload);
    out.push_back(0x80); out.push_back(0x04); out.push_back(0x95);
    for (int i = 0; i < 8; ++i) out.push_back((uint8_t)(payload >> (8 * i)));
    out.push_back(0x5d); out.push_back(0x94);
    if (count >= 2) out.push_back(0x28);
    out.insert(out.end(), items.begin(), items.end());
    out.push_back(count >= 2 ? 0x65 : 0x61);
    out.push_back(0x2e);
    return true;
}


// The Fiat-Shamir step of one round of Fri.commit with its work split around the root's arrival (fri.py:71-79): the transcript is
// the pickled list of `count` items of which the LAST -- a 64-byte root -- is still being computed.  prepare() lays the bytes out
// with the root blank and absorbs every whole block in front of it (nine of ten at fifteen roots); finish() drops the root in and
// absorbs the one or two blocks that hold it.  Between a root arriving and the next launch: ~0.7 us of hashing instead of ~3.
struct PendingChallenge {
    std::vector<uint8_t> bytes;
    uint64_t state[25];
    size_t root_at = 0, absorbed = 0;
    // items: the pickled items WITHOUT the pending root; count: items including it
    bool prepare(std::vector<uint8_t>& items, size_t count) {
        static const uint8_t blank[64] = {0};
        const size_t before = items.size();
        transcript_item(items, blank, 64);
        const bool ok = transcript_bytes(items, count, bytes);
        items.resize(before);
        if (!ok) return false;
        root_at = bytes.size() - 2 - 1 - 64;            // ... 43 40 <root> 94 | 65 or 61 | 2e
        memset(state, 0, sizeof state);
        absorbed = (root_at / 136) * 136;
        shake256_absorb_blocks(state, bytes.data(), absorbed / 136);
        return true;
    }
    void finish(const uint8_t root[64], uint8_t* out, size_t outlen) {
        memcpy(bytes.data() + root_at, root, 64);
        shake256_finish(state, bytes.data() + absorbed, bytes.size() - absorbed, out, outlen);
    }
};

}  // namespace sc
