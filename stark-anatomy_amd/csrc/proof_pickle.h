// proof_pickle.h -- ProofStream.serialize() / prover_fiat_shamir() without a Python object per digest.
//
// Reference: code/ip.py:18-25  serialize() = pickle.dumps(self.objects);  prover_fiat_shamir() = shake_256(serialize()).digest(32).
// A proof of FastStark.prove at a 2^24 FRI domain is ~3 MB: ~50 000 64-byte digests in ~2 800 authentication paths, ~3 000
// field elements.  Creating those objects and pickling them is 4-5 ms of CPython per proof, a seventh of the whole prover.
// The bytes are part of the protocol (the verifier unpickles them; the Fiat-Shamir challenges hash them), so they are produced
// HERE exactly as CPython's C pickler (Modules/_pickle.c, protocol 4, the default of 3.8+) produces them for the object graph a
// proof stream holds -- from a compact description of that graph (the "ops" below), not from objects:
//
//   list (top level and nested)      ]  MEMOIZE  [ item APPEND | ( MARK items... APPENDS, in batches of 1000 ) ]
//   bytes, fresh object              SHORT_BINBYTES / BINBYTES  MEMOIZE   (>= 64 KiB: outside any frame, as _Pickler_write_bytes does)
//   tuple of three                   items  TUPLE3  MEMOIZE
//   algebra.FieldElement, first use  global (module and name strings memoized, STACK_GLOBAL, MEMOIZE)  )  NEWOBJ  MEMOIZE
//                                    }  MEMOIZE  (  'value' int  'field' <Field object>  SETITEMS  BUILD
//     the same OBJECT again          BINGET / LONG_BINGET of its memo index -- identity is part of the description (`key`)
//   algebra.Field, first use         global  )  NEWOBJ  MEMOIZE  }  MEMOIZE  'p' int  SETITEM  BUILD
//   int                              BININT1 / BININT2 / BININT (fits 31 bits + sign)  |  LONG1 with (bit_length >> 3) + 1 bytes
//   framing                          at the start of every save() call: a frame of >= 64 KiB is closed and a new one begun
//
// tests/test_host_cpu.py compares the output with pickle.dumps byte for byte on random object graphs (shared and fresh
// elements, several fields, lists of every batch size, streams across many frames) and on the reference's golden proofs.
//
// ops (little-endian), one item after the other; the top level must be ONE list:
//   'L' u32 count            a list of the `count` items that follow
//   'B' u32 len  bytes       a bytes object
//   'D' u32 depth  64*depth  a list of `depth` bytes objects of 64 bytes each (an authentication path, merkle.py:16-27)
//   'T'                      a tuple of the 3 items that follow
//   'E' u32 field  u64 key  16 bytes   a FieldElement: `key` names the OBJECT (equal keys = the same object, pickled once and
//                            referred to afterwards), value = two u64 limbs; `field` indexes the table of field moduli
// and, inside a list only, two ops that stand for MANY of its items with their payload packed (what the provers push):
//   'R' u32 s  u32 field  u64 key_base_cur  u64 key_base_next  u32 d_cur  u32 d_next  | u32 idx_a[s] idx_b[s] idx_c[s] | values of
//       a, b, c (16 s each) | paths of a, b (64 d_cur s each), of c (64 d_next s)
//                            one round of the FRI query phase (fri.py:98-113): s tuples (cur[a], cur[b], next[c]), then per
//                            test the paths of a, b, c -- 4 s items; an element's key is key_base | index
//   'O' u32 k  u32 field  u64 key_base  u32 depth  | u32 idx[k] | values 16 k | paths 64 depth k
//                            leaf, path, leaf, path, ... of one committed codeword (fast_stark.py:154-175): 2 k items
// and the same two with their payload left WHERE THE DEVICE WROTE IT (the pinned answer buffer of sc_fri_prove_dev): the op carries
// addresses in this process, nothing is copied into the description
//   'Q' u32 s  u32 field  u32 k  | k x (u64 key_base, u32 depth) | u64 elems  u64 paths  u64 positions
//                            every round of a query phase over k codewords: codeword j's openings lie in the three arrays in
//                            the order [a (s), b (s)] if j < k - 1, [c (s)] for the last one (elements 16 bytes, paths 64 depth_j,
//                            positions u64), codeword after codeword; the c of a round whose next codeword is not the last is that
//                            codeword's a or b (the same position, opened once); 4 s (k - 1) items, as k - 1 'R' ops would produce
//   'P' u32 k  u32 field  u64 key_base  u32 depth  u64 positions  u64 values  u64 paths
//                            as 'O', positions u64
#pragma once
#include <stdint.h>
#include <string.h>
#include <unordered_map>
#include <vector>

namespace sc {

struct ProofPickler {
    std::vector<uint8_t> out;
    size_t frame_start = 0;          // offset of the open frame's 9-byte header
    uint32_t memo_next = 0;
    int64_t m_algebra = -1, m_fe_name = -1, m_fe_class = -1, m_value = -1, m_field_key = -1, m_field_name = -1, m_field_class = -1, m_p = -1;
    std::unordered_map<uint32_t, uint32_t> field_memo;     // field index -> memo index of the Field object
    std::unordered_map<uint64_t, uint32_t> elem_memo;      // (field, key) -> memo index of the FieldElement object
    const uint8_t* moduli = nullptr;                       // table of field moduli, `modulus_bytes` each, little-endian
    uint32_t nfields = 0, modulus_bytes = 0;
    bool bad = false;
    std::vector<size_t>* bytes_at = nullptr;               // optional: where each bytes object's data starts in the output (a caller that patches one in later)

    static constexpr size_t FRAME_HEADER = 9, FRAME_TARGET = 64 * 1024, FRAME_MIN = 4;

    // the output grows in big steps and is written through a raw pointer: a digest is 67 bytes in four writes, and a proof has
    // fifty thousand of them
    // (straight into the caller's buffer while it is big enough; into `out` from the moment it is not)
    uint8_t* base = nullptr;
    size_t cap = 0, used = 0;
    void use_buffer(uint8_t* p, size_t n) { base = p; cap = n; }
    void room(size_t n) {
        if (used + n <= cap) return;
        std::vector<uint8_t> bigger((used + n) * 2 + 4096);
        if (used) memcpy(bigger.data(), base, used);
        out.swap(bigger);
        base = out.data();
        cap = out.size();
    }
    void put(uint8_t b) { room(1); base[used++] = b; }
    void put(const void* p, size_t n) { room(n); memcpy(base + used, p, n); used += n; }
    void put_u32(uint32_t v) { room(4); memcpy(base + used, &v, 4); used += 4; }

    void begin() {
        used = 0;
        put(0x80); put(0x04);                              // PROTO 4
        start_frame();
    }
    void start_frame() {
        frame_start = used;
        room(FRAME_HEADER);
        used += FRAME_HEADER;
    }
    void commit_frame() {
        const size_t len = used - frame_start - FRAME_HEADER;
        if (len >= FRAME_MIN) {
            base[frame_start] = 0x95;                      // FRAME
            for (int i = 0; i < 8; ++i) base[frame_start + 1 + i] = (uint8_t)((uint64_t)len >> (8 * i));
        } else {                                           // too short to be framed
            memmove(base + frame_start, base + frame_start + FRAME_HEADER, len);
            used -= FRAME_HEADER;
        }
    }
    // every save() of the pickler starts here
    void boundary() {
        if (used - frame_start - FRAME_HEADER >= FRAME_TARGET) {
            commit_frame();
            start_frame();
        }
    }
    uint32_t memoize() { put(0x94); return memo_next++; }
    void get(uint32_t idx) {
        if (idx < 256) { put(0x68); put((uint8_t)idx); }   // BINGET
        else { put(0x6a); put_u32(idx); }                  // LONG_BINGET
    }
    // save(str): a string object that is the same object every time it appears (module / attribute names)
    void save_name(int64_t& slot, const char* text) {
        boundary();
        if (slot >= 0) { get((uint32_t)slot); return; }
        const size_t n = strlen(text);
        put(0x8c); put((uint8_t)n); put(text, n);          // SHORT_BINUNICODE
        slot = memoize();
    }
    void save_class(int64_t& slot, int64_t& name_slot, const char* name) {
        boundary();
        if (slot >= 0) { get((uint32_t)slot); return; }
        save_name(m_algebra, "algebra");
        save_name(name_slot, name);
        put(0x93);                                         // STACK_GLOBAL
        slot = memoize();
    }
    // save(int) of a non-negative integer given as little-endian bytes
    void save_uint(const uint8_t* le, size_t n) {
        boundary();
        while (n && le[n - 1] == 0) --n;
        size_t bits = n ? 8 * (n - 1) : 0;
        if (n) { uint8_t top = le[n - 1]; while (top) { ++bits; top >>= 1; } }
        if (bits <= 31) {
            uint32_t v = 0;
            for (size_t i = 0; i < n; ++i) v |= (uint32_t)le[i] << (8 * i);
            if (v < 256) { put(0x4b); put((uint8_t)v); }                       // BININT1
            else if (v < 65536) { put(0x4d); put((uint8_t)v); put((uint8_t)(v >> 8)); }   // BININT2
            else { put(0x4a); put_u32(v); }                                    // BININT
            return;
        }
        const size_t nbytes = (bits >> 3) + 1;             // always room for the sign bit
        if (nbytes < 256) { put(0x8a); put((uint8_t)nbytes); }                 // LONG1
        else { put(0x8b); put_u32((uint32_t)nbytes); }                         // LONG4
        for (size_t i = 0; i < nbytes; ++i) put(i < n ? le[i] : (uint8_t)0);
    }
    void save_bytes(const uint8_t* data, size_t len) {
        // always a FRESH object: whether two equal bytes objects are one object is the interpreter's business (b"" is a
        // singleton, one-byte objects sometimes are) -- the describer hands a stream with repeated objects to pickle itself
        boundary();
        if (len >= FRAME_TARGET) {
            // _Pickler_write_bytes of _pickle.c: a payload of a frame's size or more is not framed -- the open frame is committed
            // (dropped when it holds fewer than 4 bytes), opcode, length and data go out bare, and the MEMOIZE that follows
            // opens the next frame
            commit_frame();
            put(0x42); put_u32((uint32_t)len);                                 // BINBYTES (the description's length is a u32)
            if (bytes_at) bytes_at->push_back(used);
            put(data, len);
            start_frame();
            memoize();
            return;
        }
        if (len < 256) { put(0x43); put((uint8_t)len); }                       // SHORT_BINBYTES
        else { put(0x42); put_u32((uint32_t)len); }                            // BINBYTES
        if (bytes_at) bytes_at->push_back(used);
        put(data, len);
        memoize();
    }
    void save_field(uint32_t f) {
        boundary();
        auto it = field_memo.find(f);
        if (it != field_memo.end()) { get(it->second); return; }
        save_class(m_field_class, m_field_name, "Field");
        boundary(); put(0x29);                             // save(()) : EMPTY_TUPLE
        put(0x81);                                         // NEWOBJ
        field_memo[f] = memoize();
        boundary(); put(0x7d); memoize();                  // save(state): EMPTY_DICT MEMOIZE ... one item: SETITEM, no MARK
        save_name(m_p, "p");
        save_uint(moduli + (size_t)f * modulus_bytes, modulus_bytes);
        put(0x73);                                         // SETITEM
        put(0x62);                                         // BUILD
    }
    void save_element(uint32_t f, uint64_t key, const uint8_t value[16]) {
        boundary();
        const uint64_t id = ((uint64_t)f << 56) ^ key;
        auto it = elem_memo.find(id);
        if (it != elem_memo.end()) { get(it->second); return; }
        save_class(m_fe_class, m_fe_name, "FieldElement");
        boundary(); put(0x29);                             // EMPTY_TUPLE
        put(0x81);                                         // NEWOBJ
        elem_memo[id] = memoize();
        boundary(); put(0x7d); memoize();                  // EMPTY_DICT MEMOIZE
        put(0x28);                                         // MARK (two items)
        save_name(m_value, "value");
        save_uint(value, 16);
        save_name(m_field_key, "field");
        save_field(f);
        put(0x75);                                         // SETITEMS
        put(0x62);                                         // BUILD
    }

    // A list being written (batch_list_exact of _pickle.c): one item -> item APPEND; else MARK items... APPENDS in batches of 1000.
    // The ops of a list may produce several of its items each ('R', 'O'), so the batching is driven item by item.
    struct ListCtx { uint32_t count, done, batch; };
    void item_begin(ListCtx& L) { if (L.count != 1 && L.batch == 0) put(0x28); }      // MARK
    void item_end(ListCtx& L) {
        ++L.done;
        if (L.count == 1) { put(0x61); return; }                                       // APPEND
        if (++L.batch == 1000 || L.done == L.count) { put(0x65); L.batch = 0; }         // APPENDS
    }
    void save_path(const uint8_t* raw, uint32_t depth) {
        // fifty thousand digests per proof: when no frame can end inside this path (a save() begins a new frame only at 64 KiB) and
        // the list is one batch, its bytes are a fixed pattern -- EMPTY_LIST MEMOIZE [MARK] (SHORT_BINBYTES 64 <digest> MEMOIZE)*
        // APPEND|APPENDS -- written in one go instead of four checked writes per digest
        const size_t open_len = used - frame_start - FRAME_HEADER;
        const size_t total = 2 + (depth >= 2 ? 1 : 0) + (size_t)depth * 67 + (depth ? 1 : 0);
        if (depth <= 1000 && !bytes_at && open_len + total < FRAME_TARGET) {
            room(total);
            uint8_t* o = base + used;
            *o++ = 0x5d; *o++ = 0x94;
            if (depth >= 2) *o++ = 0x28;
            for (uint32_t i = 0; i < depth; ++i) {
                *o++ = 0x43; *o++ = 64;
                memcpy(o, raw + (size_t)i * 64, 64);
                o += 64;
                *o++ = 0x94;
            }
            if (depth) *o++ = depth == 1 ? 0x61 : 0x65;
            used += total;
            memo_next += depth + 1;
            return;
        }
        boundary(); put(0x5d); memoize();                                              // EMPTY_LIST MEMOIZE
        ListCtx L{depth, 0, 0};
        for (uint32_t i = 0; i < depth; ++i) { item_begin(L); save_bytes(raw + (size_t)i * 64, 64); item_end(L); }
    }

    // one op of the description at p, writing into list L (nullptr: an op that is not inside a list -- it must produce exactly
    // one item); returns the position behind it (nullptr: malformed)
    const uint8_t* item(const uint8_t* p, const uint8_t* end, ListCtx* L = nullptr) {
        if (p >= end) return nullptr;
        const uint8_t op = *p++;
        auto u32 = [&](uint32_t* v) { if (end - p < 4) return false; memcpy(v, p, 4); p += 4; return true; };
        auto u64 = [&](uint64_t* v) { if (end - p < 8) return false; memcpy(v, p, 8); p += 8; return true; };
        if (op == 'Q' || op == 'P') {                      // several items of the enclosing list, payload by address
            if (!L) return nullptr;
            uint32_t k, f;
            if (!u32(&k) || !u32(&f) || f >= nfields) return nullptr;
            if (op == 'P') {
                uint64_t base, p_pos, p_val, p_path; uint32_t depth;
                if (!u64(&base) || !u32(&depth) || !u64(&p_pos) || !u64(&p_val) || !u64(&p_path)) return nullptr;
                if (L->done + 2ull * k > L->count || (k && (!p_pos || !p_val || (depth && !p_path)))) return nullptr;
                const uint64_t* pos = (const uint64_t*)(uintptr_t)p_pos;
                const uint8_t *val = (const uint8_t*)(uintptr_t)p_val, *path = (const uint8_t*)(uintptr_t)p_path;
                for (uint32_t t = 0; t < k; ++t) {
                    item_begin(*L); save_element(f, base | (uint32_t)pos[t], val + 16ull * t); item_end(*L);
                    item_begin(*L); save_path(path + (size_t)t * 64 * depth, depth); item_end(*L);
                }
                return p;
            }
            // 'Q': k = s here, then the number of codewords
            const uint32_t s_ = k;
            uint32_t nk;
            if (!u32(&nk) || nk < 2 || nk > 64 || (size_t)(end - p) < (size_t)nk * 12 + 24) return nullptr;
            uint64_t base[64]; uint32_t depth[64];
            for (uint32_t j = 0; j < nk; ++j) { memcpy(&base[j], p, 8); memcpy(&depth[j], p + 8, 4); p += 12; }
            uint64_t p_el, p_path, p_pos;
            if (!u64(&p_el) || !u64(&p_path) || !u64(&p_pos) || !p_el || !p_path || !p_pos) return nullptr;
            if (L->done + 4ull * s_ * (nk - 1) > L->count) return nullptr;
            const uint8_t *el = (const uint8_t*)(uintptr_t)p_el, *paths = (const uint8_t*)(uintptr_t)p_path;
            const uint64_t* pos = (const uint64_t*)(uintptr_t)p_pos;
            size_t eo[65], po[65];
            eo[0] = po[0] = 0;
            for (uint32_t j = 0; j < nk; ++j) {
                const size_t cnt = j + 1 < nk ? 2ull * s_ : (j > 0 ? s_ : 0);
                eo[j + 1] = eo[j] + cnt;
                po[j + 1] = po[j] + cnt * 64 * depth[j];
            }
            for (uint32_t i = 0; i + 1 < nk; ++i) {
                // c of round i = position a of round i in codeword i + 1: opened there as that codeword's a (slot t) or b (slot s + t) --
                // whichever half it lies in -- unless codeword i + 1 is the last one, which holds nothing but these
                const size_t a0 = eo[i], b0 = eo[i] + s_;
                const uint64_t next_half = depth[i + 1] ? 1ull << (depth[i + 1] - 1) : 0;
                auto c_slot = [&](uint32_t t) -> size_t { return i + 2 < nk ? (pos[a0 + t] < next_half ? t : s_ + t) : t; };
                for (uint32_t t = 0; t < s_; ++t) {
                    const size_t c0t = eo[i + 1] + c_slot(t);
                    item_begin(*L);
                    boundary();                                // save(tuple)
                    save_element(f, base[i] | (uint32_t)pos[a0 + t], el + 16 * (a0 + t));
                    save_element(f, base[i] | (uint32_t)pos[b0 + t], el + 16 * (b0 + t));
                    save_element(f, base[i + 1] | (uint32_t)pos[c0t], el + 16 * c0t);
                    put(0x87); memoize();                      // TUPLE3 MEMOIZE
                    item_end(*L);
                }
                const size_t dc = depth[i], dn = depth[i + 1];
                for (uint32_t t = 0; t < s_; ++t) {
                    item_begin(*L); save_path(paths + po[i] + (size_t)t * 64 * dc, (uint32_t)dc); item_end(*L);
                    item_begin(*L); save_path(paths + po[i] + (size_t)(s_ + t) * 64 * dc, (uint32_t)dc); item_end(*L);
                    item_begin(*L); save_path(paths + po[i + 1] + c_slot(t) * 64 * dn, (uint32_t)dn); item_end(*L);
                }
            }
            return p;
        }
        if (op == 'R' || op == 'O') {                      // several items of the enclosing list
            if (!L) return nullptr;
            uint32_t k, f;
            if (!u32(&k) || !u32(&f) || f >= nfields) return nullptr;
            if (op == 'O') {
                // 'O' k field key_base depth | idx u32[k] | values 16k | paths 64*depth*k :  leaf, path, leaf, path, ...
                uint64_t base; uint32_t depth;
                if (!u64(&base) || !u32(&depth)) return nullptr;
                const size_t need = (size_t)k * (4 + 16 + (size_t)64 * depth);
                if ((size_t)(end - p) < need || L->done + 2ull * k > L->count) return nullptr;
                const uint8_t *idx = p, *val = p + 4ull * k, *path = val + 16ull * k;
                for (uint32_t t = 0; t < k; ++t) {
                    uint32_t i; memcpy(&i, idx + 4ull * t, 4);
                    item_begin(*L); save_element(f, base | i, val + 16ull * t); item_end(*L);
                    item_begin(*L); save_path(path + (size_t)t * 64 * depth, depth); item_end(*L);
                }
                return p + need;
            }
            // 'R' s field key_base_cur key_base_next d_cur d_next | idx_a idx_b idx_c u32[s] | val_a val_b val_c 16s |
            //     paths_a paths_b 64*d_cur*s | paths_c 64*d_next*s :  s triples (cur[a], cur[b], next[c]), then per test the three paths
            uint64_t bc, bn; uint32_t dc, dn;
            if (!u64(&bc) || !u64(&bn) || !u32(&dc) || !u32(&dn)) return nullptr;
            const size_t need = (size_t)k * (12 + 48 + (size_t)64 * (2 * dc + dn));
            if ((size_t)(end - p) < need || L->done + 4ull * k > L->count) return nullptr;
            const uint8_t *ia = p, *ib = ia + 4ull * k, *ic = ib + 4ull * k, *va = ic + 4ull * k, *vb = va + 16ull * k, *vc = vb + 16ull * k;
            const uint8_t *pa = vc + 16ull * k, *pb = pa + (size_t)64 * dc * k, *pc = pb + (size_t)64 * dc * k;
            for (uint32_t t = 0; t < k; ++t) {
                uint32_t a, b_, c;
                memcpy(&a, ia + 4ull * t, 4); memcpy(&b_, ib + 4ull * t, 4); memcpy(&c, ic + 4ull * t, 4);
                item_begin(*L);
                boundary();                                // save(tuple)
                save_element(f, bc | a, va + 16ull * t);
                save_element(f, bc | b_, vb + 16ull * t);
                save_element(f, bn | c, vc + 16ull * t);
                put(0x87); memoize();                      // TUPLE3 MEMOIZE
                item_end(*L);
            }
            for (uint32_t t = 0; t < k; ++t) {
                item_begin(*L); save_path(pa + (size_t)t * 64 * dc, dc); item_end(*L);
                item_begin(*L); save_path(pb + (size_t)t * 64 * dc, dc); item_end(*L);
                item_begin(*L); save_path(pc + (size_t)t * 64 * dn, dn); item_end(*L);
            }
            return p + need;
        }
        if (L) item_begin(*L);
        const uint8_t* q = single(op, p, end);
        if (q && L) item_end(*L);
        return q;
    }
    // the ops that are exactly one object
    const uint8_t* single(uint8_t op, const uint8_t* p, const uint8_t* end) {
        auto u32 = [&](uint32_t* v) { if (end - p < 4) return false; memcpy(v, p, 4); p += 4; return true; };
        switch (op) {
            case 'B': {
                uint32_t len;
                if (!u32(&len) || (size_t)(end - p) < len) return nullptr;
                save_bytes(p, len);
                return p + len;
            }
            case 'D': {
                uint32_t depth;
                if (!u32(&depth) || (size_t)(end - p) < (size_t)depth * 64) return nullptr;
                save_path(p, depth);
                return p + (size_t)depth * 64;
            }
            case 'L': {
                uint32_t count;
                if (!u32(&count)) return nullptr;
                boundary(); put(0x5d); memoize();          // EMPTY_LIST MEMOIZE
                ListCtx L{count, 0, 0};
                while (L.done < count) { p = item(p, end, &L); if (!p) return nullptr; }
                return p;
            }
            case 'T': {
                boundary();
                for (int i = 0; i < 3; ++i) { p = item(p, end); if (!p) return nullptr; }
                put(0x87); memoize();                      // TUPLE3 MEMOIZE
                return p;
            }
            case 'E': {
                uint32_t f;
                if (!u32(&f) || f >= nfields || end - p < 24) return nullptr;
                uint64_t key;
                memcpy(&key, p, 8);
                save_element(f, key, p + 8);
                return p + 24;
            }
            default: return nullptr;
        }
    }
    bool run(const uint8_t* ops, size_t len) {
        begin();
        const uint8_t* end = ops + len;
        if (!len || (ops[0] != 'L' && ops[0] != 'D')) return false;
        const uint8_t* p = item(ops, end);
        if (p != end) return false;
        put(0x2e);                                         // STOP
        commit_frame();
        return true;
    }
};

}  // namespace sc
