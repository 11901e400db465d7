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

}  // namespace sc
