// merkle_fri.hip -- Merkle trees (code/merkle.py), the split-and-fold of code/fri.py:85, the commit loop of Fri.commit with the
// Fiat-Shamir step of code/ip.py:18-25 on the host side of the library, and the openings of the query phase.
#include "core.h"
#include "merkle.cuh"
#include "fri_tail.cuh"
#include "transcript.h"
#include "proof_pickle.h"
#include <immintrin.h>      // _mm_sfence: the challenge written through the BAR (write-combined) must reach the device before its flag

// split-and-fold (code/fri.py:85) rewritten as
//   out[i] = (a + b)/2 + (a - b) * c * w^-i,   a = in[i], b = in[i + N/2], c = alpha / (2 * offset)
// lo/hi are the power tables of omega^-1, c_m is c in Montgomery form.
__global__ void __launch_bounds__(256) fri_fold_kernel(const Fe* __restrict__ in, Fe* __restrict__ out, uint64_t half, const Fe* __restrict__ lo, const Fe* __restrict__ hi, Fe c_m) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= half) return;
    Fe a = in[i], b = in[i + half];
    Fe t = mont_mul(pow2level(lo, hi, i), c_m);          // (c * w^-i) in Montgomery form
    out[i] = fe_add(fe_half(fe_add(a, b)), mont_mul(fe_sub(a, b), t));
}

// the same fold on a column slab [rows][2^logcols] of the codeword viewed as a rows x R matrix (index i = row * R + col_base + col):
// partner i + N/2 is row + rows/2 of the SAME slab, so the fold is local to the rank that owns the columns.
__global__ void __launch_bounds__(256) fri_fold_slab_kernel(const Fe* __restrict__ in, Fe* __restrict__ out, uint64_t half_rows, int logcols, uint64_t R,
                                                            uint64_t col_base, const Fe* __restrict__ lo, const Fe* __restrict__ hi, Fe c_m) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (half_rows << logcols)) return;
    uint64_t row = t >> logcols, col = t & ((1ull << logcols) - 1);
    Fe a = in[t], b = in[t + (half_rows << logcols)];
    Fe w = mont_mul(pow2level(lo, hi, row * R + col_base + col), c_m);
    out[t] = fe_add(fe_half(fe_add(a, b)), mont_mul(fe_sub(a, b), w));
}

namespace sci {

// host / seq: an asynchronous build's pinned root slot; *published is set when the last launch (the tail kernel) wrote the root
// there itself, so that no separate publish launch is needed
int merkle_climb(uint64_t* levels, uint64_t N, int lvl, hipStream_t st, volatile uint64_t* host, uint64_t seq, bool* published) {
    const int logN = ilog2(N);
    uint64_t w = N >> lvl;
    auto off = [N](int l) -> uint64_t { return l == 0 ? 0 : 2 * N - (N >> (l - 1)); };
    while (w > 2048) {
        if (w > FUSE_MAX_W && g.merkle_big_nlev > 0 && (w >> g.merkle_big_nlev) >= 2048) {
            // whole waves retire as the subtree narrows (256 -> 128 -> 64 nodes: 4, 2, 1 full waves), no lane is wasted and
            // the intermediate levels are never re-read from HBM
            const int nlev = g.merkle_big_nlev;
            hipLaunchKernelGGL((merkle_subtree_kernel<false, false>), dim3((unsigned)(w / 256)), dim3(256), 0, st, (const Fe*)nullptr, levels, N, lvl, nlev, FoldIn());
            lvl += nlev;
            w >>= nlev;
        } else if (w > FUSE_MAX_W) {
            hipLaunchKernelGGL(merkle_level_kernel, dim3((unsigned)((w / 2 + 255) / 256)), dim3(256), 0, st, levels + 8 * off(lvl), levels + 8 * off(lvl + 1), w / 2);
            lvl += 1;
            w >>= 1;
        } else {
            int nlev = 8;
            if (nlev > logN - lvl) nlev = logN - lvl;
            hipLaunchKernelGGL((merkle_subtree_kernel<false, true>), dim3((unsigned)(w / 256)), dim3(256), 0, st, (const Fe*)nullptr, levels, N, lvl, nlev, FoldIn());
            lvl += nlev;
            w >>= nlev;
        }
    }
    if (w > 1) {
        hipLaunchKernelGGL(merkle_tail_kernel, dim3(1), dim3(1024), 0, st, levels + 8 * off(lvl), w, host, seq);
        if (host && published) *published = true;
    }
    HIPCHK(hipGetLastError());
    return SC_OK;
}

// finish a tree whose level 0 (the `width` digests at `levels`) is already in place
int merkle_finish(uint64_t* levels, uint64_t width, hipStream_t st) { return merkle_climb(levels, width, 0, st); }

// pinned host slots the roots of asynchronously built trees are copied to (64 bytes each)
int root_slot_get() {
    if (!g.root_slots) {
        if (hipHostMalloc((void**)&g.root_slots, ROOT_SLOT_BYTES * ROOT_SLOTS, hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); g.root_slots = nullptr; return -1; }
        memset(g.root_slots, 0, ROOT_SLOT_BYTES * ROOT_SLOTS);
        for (int i = ROOT_SLOTS - 1; i >= 0; --i) g.free_root_slots.push_back(i);
    }
    if (g.free_root_slots.empty()) return -1;
    const int s = g.free_root_slots.back();
    g.free_root_slots.pop_back();
    return s;
}

// call degrades to the synchronous form.
// BUILD_NOROOT: only enqueued as well, but nobody is expected to ask for the root (a rank's local subtree of a sharded commit: its
// sub-root level is copied out on the same stream): no pinned slot is taken and no publish kernel runs, so any number of such
// trees can be alive at once without degrading the asynchronous builds of Fri.commit to the synchronous path.
// fold != nullptr (with N >= 256): the leaves are the split-and-fold of the previous round's codeword, computed, stored to
// d_elems (= fold->out) and hashed by the leaf stage itself (merkle_subtree_kernel<true, *, true>)
int merkle_build_device(const Fe* d_elems, uint64_t N, uint8_t root_out[64], sc_merkle** tree, hipStream_t st, BuildMode mode, const FoldIn* fold) {
    if (!is_pow2(N)) return fail(SC_ERR_NOT_POW2, "length must be power of two");
    if (fold && N < 256) return fail(SC_ERR_BAD_ARG, "the fused fold needs at least 256 leaves");
    const int slot = (mode == BUILD_ASYNC && tree) ? root_slot_get() : -1;
    const uint64_t seq = slot >= 0 ? ++g.root_seq : 0;
    volatile uint64_t* host = slot >= 0 ? (volatile uint64_t*)(g.root_slots + ROOT_SLOT_BYTES * slot) : nullptr;
    bool published = false;
    uint8_t root_tmp[64];
    if (!root_out) root_out = root_tmp;
    uint64_t* levels = nullptr;
    const size_t tree_bytes = (2 * N - 1) * 64;
    HIPCHK(pool_alloc((void**)&levels, tree_bytes));
    if (N >= 256 && N <= FUSE_MAX_W) {
        int nlev = ilog2(N) < 8 ? ilog2(N) : 8;                  // leaves + up to 8 levels of every 256-leaf subtree in one launch
        if (fold) hipLaunchKernelGGL((merkle_subtree_kernel<true, true, true>), dim3((unsigned)(N / 256)), dim3(256), 0, st, d_elems, levels, N, 0, nlev, *fold);
        else hipLaunchKernelGGL((merkle_subtree_kernel<true, true>), dim3((unsigned)(N / 256)), dim3(256), 0, st, d_elems, levels, N, 0, nlev, FoldIn());
        (void)merkle_climb(levels, N, nlev, st, host, seq, &published);
    } else if (N > FUSE_MAX_W && (g.merkle_big_nlev > 0 || fold)) {
        const int nlev = g.merkle_big_nlev > 0 ? g.merkle_big_nlev : 1;
        if (fold) hipLaunchKernelGGL((merkle_subtree_kernel<true, false, true>), dim3((unsigned)(N / 256)), dim3(256), 0, st, d_elems, levels, N, 0, nlev, *fold);
        else hipLaunchKernelGGL((merkle_subtree_kernel<true, false>), dim3((unsigned)(N / 256)), dim3(256), 0, st, d_elems, levels, N, 0, nlev, FoldIn());
        (void)merkle_climb(levels, N, nlev, st, host, seq, &published);
    } else if (N > FUSE_MAX_W) {
        hipLaunchKernelGGL(merkle_leaf_kernel, dim3((unsigned)(N / 256)), dim3(256), 0, st, d_elems, levels, N);
        (void)merkle_climb(levels, N, 0, st, host, seq, &published);
    } else {
        hipLaunchKernelGGL(merkle_leaf_kernel, dim3(1), dim3(256), 0, st, d_elems, levels, N);
        (void)merkle_climb(levels, N, 0, st, host, seq, &published);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { if (slot >= 0) g.free_root_slots.push_back(slot); pool_free(levels, tree_bytes); return fail(SC_ERR_HIP, hipGetErrorString(e)); }
    if (mode == BUILD_NOROOT && tree) {
        sc_merkle* t = new sc_merkle{levels, N, ilog2(N)};
        t->st = st;
        t->lazy = true;
        *tree = t;
        return SC_OK;
    }
    if (slot >= 0) {
        // the root is WRITTEN to the host slot by the kernel that computes it (the one-workgroup tail kernel) or, where the tree
        // ends in another kernel, by a one-wave kernel behind the build -- then its sequence number: the waiting host sees it a
        // microsecond later, without a copy engine, a completion signal or a runtime call in between
        if (!published)
            hipLaunchKernelGGL(root_publish_kernel, dim3(1), dim3(64), 0, st, (const uint64_t*)(levels + 8 * (2 * N - 2)), host, seq);
        e = hipGetLastError();
        if (e != hipSuccess) { g.free_root_slots.push_back(slot); pool_free(levels, tree_bytes); return fail(SC_ERR_HIP, hipGetErrorString(e)); }
        sc_merkle* t = new sc_merkle{levels, N, ilog2(N)};
        t->slot = slot;
        t->seq = seq;
        t->st = st;
        *tree = t;
        return SC_OK;
    }
    // (the root of a synchronous build travels like the others: a pinned slot the host polls, not a copy and a sleeping wait)
    if (read_small_polled(levels + 8 * (2 * N - 2), 64, st, root_out) != SC_OK) { pool_free(levels, tree_bytes); return SC_ERR_HIP; }
    if (tree) {
        sc_merkle* t = new sc_merkle{levels, N, ilog2(N)};
        memcpy(t->root, root_out, 64);
        t->have_root = true;
        *tree = t;
    } else {
        pool_free(levels, tree_bytes);
    }
    return SC_OK;
}

// the root of a tree, waiting for an asynchronous build if that is what made it
// from_free: called by sc_merkle_free only to get the slot back -- the stream the build ran on may have been destroyed by its
// owner by then (destroying a stream lets its work finish, so the root has landed or is about to): poll, then wait for the
// device, never touch the stream handle.
int merkle_root_wait(sc_merkle* t, bool from_free) {
    if (t->have_root) return SC_OK;
    if (t->lazy) {
        if (from_free) return SC_OK;
        // nobody was expected to ask: the whole device is waited for (the build's stream may be gone), then one small copy
        HIPCHK(hipDeviceSynchronize());
        HIPCHK(hipMemcpy(t->root, t->d_levels + 8 * (2 * t->N - 2), 64, hipMemcpyDeviceToHost));
        t->have_root = true;
        return SC_OK;
    }
    if (t->slot < 0) return fail(SC_ERR_BAD_ARG, "tree has no root");
    // the prover's serial chain waits here once per round: poll the slot's sequence number (a blocking wait that has gone to
    // sleep costs tens of microseconds to wake up); every so often ask the stream, so that a failed launch cannot hang the
    // caller, and after a few milliseconds block
    volatile uint64_t* slot = (volatile uint64_t*)(g.root_slots + ROOT_SLOT_BYTES * t->slot);
    hipError_t e = hipSuccess;
    bool landed = false;
    for (long spin = 0; spin < SPIN_POLLS; ++spin) {
        if (__atomic_load_n(slot + 8, __ATOMIC_ACQUIRE) == t->seq) { landed = true; break; }
        if ((spin & 4095) == 4095) {
            if (from_free) { e = hipDeviceSynchronize(); break; }
            e = hipStreamQuery(t->st);
            if (e != hipErrorNotReady) break;              // finished (the number is there now) or failed
            (void)hipGetLastError();
            e = hipSuccess;
        }
    }
    if (!landed) {
        if (from_free) { if (e != hipSuccess) (void)hipGetLastError(); }
        else if (e == hipSuccess || e == hipErrorNotReady) { (void)hipGetLastError(); e = hipStreamSynchronize(t->st); }
        landed = (e == hipSuccess) && __atomic_load_n(slot + 8, __ATOMIC_ACQUIRE) == t->seq;
        if (e == hipSuccess && !landed) e = hipErrorUnknown;
    }
    memcpy(t->root, (const void*)slot, 64);
    if (e != hipSuccess) { if (from_free) (void)hipDeviceSynchronize(); else (void)hipStreamSynchronize(t->st); }      // nothing may still write to the slot when it is reused
    g.free_root_slots.push_back(t->slot);
    t->slot = -1;
    if (e != hipSuccess) return fail(SC_ERR_HIP, hipGetErrorString(e));
    t->have_root = true;
    return SC_OK;
}

// The commit loop of sc_fri_commit_dev spends most of its time waiting for roots.  That wait does not need the library lock:
// the tree, its slot and its sequence number belong to the calling thread until the call returns.  Poll with the lock released
// (other threads' sc_vec_free / sc_merkle_root / a second prover get through), then take it again; merkle_root_wait finds the
// root landed (or, after a failed launch, finds out why under the lock).
void root_poll_unlocked(std::unique_lock<std::mutex>& lk, const sc_merkle* t) {
    if (t->have_root || t->lazy || t->slot < 0) return;
    volatile uint64_t* slot = (volatile uint64_t*)(g.root_slots + ROOT_SLOT_BYTES * t->slot);
    const uint64_t seq = t->seq;
    lk.unlock();
    for (long spin = 0; spin < SPIN_POLLS; ++spin)
        if (__atomic_load_n(slot + 8, __ATOMIC_ACQUIRE) == seq) break;
    lk.lock();
}

// what a fold of an N-element codeword on offset * <omega> needs besides its challenge: the power tables of 1 / omega (N / 2
// exponents) and 1 / (2 offset) in Montgomery form
int fold_constants(uint64_t N, Fe offset, Fe omega, hipStream_t st, PowTables** pw_out, Fe* i2o_m_out) {
    if (N < 2 || !is_pow2(N)) return fail(SC_ERR_NOT_POW2, "codeword length must be a power of two >= 2");
    if (fe_is_zero(offset) || fe_is_zero(omega)) return fail(SC_ERR_DIV_ZERO, "divide by zero");
    // omega^-1 power tables; c = alpha / (2 * offset).  Consecutive rounds of Fri.commit square omega and offset (fri.py:86-87):
    // then 1/omega' = (1/omega)^2 and 1/(2 offset') = 2 (1/(2 offset))^2 -- three products instead of two ~250-product inversions
    // on the hand-over between rounds; the products are checked, anything else takes the inversions.
    static Fe prev_omega_m = Fe{0, 0}, prev_offset_m = Fe{0, 0}, prev_winv_m = Fe{0, 0}, prev_i2o_m = Fe{0, 0};
    static bool have_prev = false;
    const Fe omega_m = to_mont(omega), offset_m = to_mont(offset);
    const Fe two_off_m = fe_add(offset_m, offset_m);
    Fe winv_m, i2o_m;
    bool derived = false;
    if (have_prev && fe_eq(omega_m, mont_mul(prev_omega_m, prev_omega_m)) && fe_eq(offset_m, mont_mul(prev_offset_m, prev_offset_m))) {
        winv_m = mont_mul(prev_winv_m, prev_winv_m);
        Fe sq = mont_mul(prev_i2o_m, prev_i2o_m);
        i2o_m = fe_add(sq, sq);
        derived = fe_eq(mont_mul(winv_m, omega_m), fe_mont_one()) && fe_eq(mont_mul(i2o_m, two_off_m), fe_mont_one());
    }
    if (!derived) {
        winv_m = mont_inv(omega_m);
        i2o_m = mont_inv(two_off_m);
    }
    prev_omega_m = omega_m; prev_offset_m = offset_m; prev_winv_m = winv_m; prev_i2o_m = i2o_m;
    have_prev = true;
    Fe winv = from_mont(winv_m);
    SCCHK(get_pow(winv, N / 2, st, pw_out));
    *i2o_m_out = i2o_m;
    return SC_OK;
}
// everything of a fold but its launch: the power tables of omega^-1 and c = alpha / (2 * offset)
int fold_prepare(const Fe* d_in, uint64_t N, Fe alpha, Fe offset, Fe omega, Fe* d_out, hipStream_t st, FoldIn* f) {
    PowTables* pw;
    Fe i2o_m;
    SCCHK(fold_constants(N, offset, omega, st, &pw, &i2o_m));
    f->in = d_in; f->out = d_out; f->lo = pw->lo; f->hi = pw->hi;
    f->c_m = mont_mul(to_mont(alpha), i2o_m);     // alpha~ * (2 offset)^-1~ / R = c~
    return SC_OK;
}

int fold_device(const Fe* d_in, uint64_t N, Fe alpha, Fe offset, Fe omega, Fe* d_out, hipStream_t st) {
    FoldIn f;
    SCCHK(fold_prepare(d_in, N, alpha, offset, omega, d_out, st, &f));
    uint64_t half = N / 2;
    hipLaunchKernelGGL(fri_fold_kernel, dim3((unsigned)((half + 255) / 256)), dim3(256), 0, st, d_in, d_out, half, f.lo, f.hi, f.c_m);
    HIPCHK(hipGetLastError());
    return SC_OK;
}

// one round of Fri.commit (fri.py:73-88) on the device: the fold of the codeword and the tree of the folded codeword, enqueued, the
// root on its way to a pinned slot.  From 256 folded elements up the leaf stage of the tree computes the fold itself.
int fold_and_build(const Fe* d_in, uint64_t N, Fe alpha, Fe offset, Fe omega, Fe* d_out, sc_merkle** tree, hipStream_t st) {
    if (N / 2 >= 256) {
        FoldIn f;
        SCCHK(fold_prepare(d_in, N, alpha, offset, omega, d_out, st, &f));
        return merkle_build_device(d_out, N / 2, nullptr, tree, st, BUILD_ASYNC, &f);
    }
    SCCHK(fold_device(d_in, N, alpha, offset, omega, d_out, st));
    return merkle_build_device(d_out, N / 2, nullptr, tree, st, BUILD_ASYNC);
}

}  // namespace sci

extern "C" {

// ---- fold
int sc_fri_fold_dev(const void* d_in, uint64_t N, const uint64_t alpha[2], const uint64_t offset[2], const uint64_t omega[2], void* d_out, void* stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    return fold_device((const Fe*)d_in, N, fe_from(alpha), fe_from(offset), fe_from(omega), (Fe*)d_out, pick_stream(stream));
}
int sc_fri_fold(const void* in, uint64_t N, const uint64_t alpha[2], const uint64_t offset[2], const uint64_t omega[2], void* out) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    if (N < 2 || !is_pow2(N)) return fail(SC_ERR_NOT_POW2, "codeword length must be a power of two >= 2");
    void *a, *b;
    SCCHK(scratch(1, N * sizeof(Fe), &a));
    SCCHK(scratch(2, (N / 2) * sizeof(Fe), &b));
    SCCHK(upload(a, in, N * sizeof(Fe), g.stream));
    SCCHK(fold_device((const Fe*)a, N, fe_from(alpha), fe_from(offset), fe_from(omega), (Fe*)b, g.stream));
    return download(out, b, (N / 2) * sizeof(Fe), g.stream);
}

// ---- merkle
int sc_merkle_build_dev(const void* d_elems, uint64_t N, uint8_t root_out[64], sc_merkle_t** tree, void* stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    return merkle_build_device((const Fe*)d_elems, N, root_out, tree, pick_stream(stream));
}
// the build is enqueued and the call returns; sc_merkle_root waits.  (The host prepares the next round while the device hashes.)
int sc_merkle_build_async_dev(const void* d_elems, uint64_t N, sc_merkle_t** tree, void* stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    if (!tree || !d_elems) return fail(SC_ERR_BAD_ARG, "null argument");
    return merkle_build_device((const Fe*)d_elems, N, nullptr, tree, pick_stream(stream), BUILD_ASYNC);
}
// enqueue only, for a tree whose root nobody is expected to read (takes no root slot, runs no publish kernel); sc_merkle_root
// still works on it (it then waits for the device)
int sc_merkle_build_noroot_dev(const void* d_elems, uint64_t N, sc_merkle_t** tree, void* stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    if (!tree || !d_elems) return fail(SC_ERR_BAD_ARG, "null argument");
    return merkle_build_device((const Fe*)d_elems, N, nullptr, tree, pick_stream(stream), BUILD_NOROOT);
}
int sc_merkle_root(sc_merkle_t* tree, uint8_t root_out[64]) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!tree || !root_out) return fail(SC_ERR_BAD_ARG, "null argument");
    SCCHK(ensure_init());
    SCCHK(merkle_root_wait(tree));
    memcpy(root_out, tree->root, 64);
    return SC_OK;
}
// one round of Fri.commit in one call (fri.py:73-88): split-and-fold with alpha, then the Merkle tree of the folded codeword;
// nothing is waited for (sc_merkle_root fetches the root)
int sc_fri_fold_commit_dev(const void* d_in, uint64_t N, const uint64_t alpha[2], const uint64_t offset[2], const uint64_t omega[2], void* d_out,
                           sc_merkle_t** tree, void* stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    if (!tree || !d_in || !d_out) return fail(SC_ERR_BAD_ARG, "null argument");
    hipStream_t st = pick_stream(stream);
    return fold_and_build((const Fe*)d_in, N, fe_from(alpha), fe_from(offset), fe_from(omega), (Fe*)d_out, tree, st);
}
// ---- the Fiat-Shamir step on the host side of the library (csrc/transcript.h); no GPU needed
static Fe sample_field(const uint8_t* bytes, size_t len) {
    // Field.sample (algebra.py:116-120): the big-endian integer of the bytes, mod p -- 16 bytes at a time: a leading short
    // chunk, then  acc <- acc * 2^128 + chunk  with acc * 2^128 = to_mont(acc) (one Montgomery product per 16 bytes; the
    // commit loop samples 32 bytes between a root arriving and the next launch)
    Fe acc{0, 0};
    size_t i = 0;
    size_t take = len % 16 ? len % 16 : (len ? 16 : 0);
    while (i < len) {
        uint64_t w[2] = {0, 0};
        for (size_t k = 0; k < take; ++k) {
            const size_t pos = take - 1 - k;                    // byte k of the chunk has weight 256^pos
            w[pos >> 3] |= (uint64_t)bytes[i + k] << (8 * (pos & 7));
        }
        Fe c{w[0], w[1]};
        if (fe_ge_p(c)) c = fe_sub(c, Fe{P_LO, P_HI});
        acc = fe_add(i ? to_mont(acc) : acc, c);
        i += take;
        take = 16;
    }
    return acc;
}
int sc_shake256(const void* in, uint64_t len, void* out, uint64_t out_len) {
    if ((!in && len) || !out) return fail(SC_ERR_BAD_ARG, "null argument");
    shake256((const uint8_t*)in, (size_t)len, (uint8_t*)out, (size_t)out_len);
    return SC_OK;
}
int sc_field_sample(const void* bytes, uint64_t len, uint64_t out[2]) {
    if ((!bytes && len) || !out) return fail(SC_ERR_BAD_ARG, "null argument");
    const Fe v = sample_field((const uint8_t*)bytes, (size_t)len);
    out[0] = v.lo; out[1] = v.hi;
    return SC_OK;
}
// SHAKE-256(pickle.dumps(items + [root])) computed the way the commit loops compute it -- everything in front of the 64-byte `root`
// absorbed first (PendingChallenge::prepare), the root dropped in afterwards (finish): the split form pinned to hashlib by
// tests/test_host_cpu.py for every item count and alignment of the root inside a rate block
int sc_transcript_challenge(const void* data, const uint32_t* lens, uint64_t count, const uint8_t root[64], uint8_t* out, uint64_t out_len) {
    if ((!data && count) || (!lens && count) || !root || !out) return fail(SC_ERR_BAD_ARG, "null argument");
    std::vector<uint8_t> items;
    const uint8_t* p = (const uint8_t*)data;
    for (uint64_t i = 0; i < count; ++i) {
        if (lens[i] > 255) return fail(SC_ERR_UNSUPPORTED, "transcript item too long for the fixed layout");
        transcript_item(items, p, lens[i]);
        p += lens[i];
    }
    PendingChallenge pending;
    if (!pending.prepare(items, (size_t)count + 1)) return fail(SC_ERR_UNSUPPORTED, "transcript too large for the fixed layout");
    pending.finish(root, out, (size_t)out_len);
    return SC_OK;
}
int sc_blake2b(const void* in, uint64_t len, uint8_t out[64]) {
    if ((!in && len) || !out) return fail(SC_ERR_BAD_ARG, "null argument");
    blake2b_512((const uint8_t*)in, (size_t)len, out);
    return SC_OK;
}
int sc_fri_sample_indices(const void* seed, uint64_t seed_len, uint64_t size, uint64_t reduced_size, uint32_t number, uint64_t* out) {
    if ((!seed && seed_len) || (!out && number)) return fail(SC_ERR_BAD_ARG, "null argument");
    if (!fri_sample_indices((const uint8_t*)seed, (size_t)seed_len, size, reduced_size, number, out))
        return fail(SC_ERR_UNSUPPORTED, "cannot sample more indices than available in last codeword (or size not a power of two)");
    return SC_OK;
}
// pickle.dumps of a list of `count` bytes objects (lens[i] < 256 bytes each, concatenated in `data`): the transcript prefix of
// ip.py:18-19 for a proof stream that holds nothing but digests.  *out_len = bytes needed; copied when out_cap suffices.
int sc_transcript_bytes(const void* data, const uint32_t* lens, uint64_t count, void* out, uint64_t out_cap, uint64_t* out_len) {
    if ((!data && count) || (!lens && count) || !out_len) return fail(SC_ERR_BAD_ARG, "null argument");
    std::vector<uint8_t> items, bytes;
    const uint8_t* p = (const uint8_t*)data;
    for (uint64_t i = 0; i < count; ++i) {
        if (lens[i] > 255) return fail(SC_ERR_UNSUPPORTED, "transcript item too long for the fixed layout");
        transcript_item(items, p, lens[i]);
        p += lens[i];
    }
    if (!transcript_bytes(items, (size_t)count, bytes)) return fail(SC_ERR_UNSUPPORTED, "transcript too large for the fixed layout");
    *out_len = bytes.size();
    if (out && out_cap >= bytes.size()) memcpy(out, bytes.data(), bytes.size());
    return SC_OK;
}

// pickle.dumps of the object graph a proof stream holds, from its description (csrc/proof_pickle.h); host only.
// *out_len = bytes needed; they are copied into `out` when out_cap suffices.
int sc_pickle_proof(const void* ops, uint64_t ops_len, const void* moduli, uint32_t nfields, uint32_t modulus_bytes, void* out, uint64_t out_cap, uint64_t* out_len) {
    if (!ops || !out_len || (nfields && !moduli)) return fail(SC_ERR_BAD_ARG, "null argument");
    ProofPickler pk;
    pk.moduli = (const uint8_t*)moduli;
    pk.nfields = nfields;
    pk.modulus_bytes = modulus_bytes;
    if (out && out_cap) pk.use_buffer((uint8_t*)out, (size_t)out_cap);     // written in place when it fits
    if (!pk.run((const uint8_t*)ops, (size_t)ops_len)) return fail(SC_ERR_BAD_ARG, "malformed proof description");
    *out_len = pk.used;
    if (out && pk.base != (uint8_t*)out && out_cap >= pk.used) memcpy(out, pk.base, pk.used);
    return SC_OK;
}

// Fri.commit's round loop (fri.py:66-94) in ONE call: per round the Merkle tree of the codeword (asynchronous build, the root
// polled from its pinned slot), the Fiat-Shamir step on the host side of the library -- the transcript is the pickled list of
// the `prior_count` digests already in the proof stream plus this call's roots; alpha = Field.sample(SHAKE-256(transcript)) --
// and the fold of fri.py:85 with that alpha, enqueued the moment alpha exists.  Nothing crosses the language boundary between a
// root arriving and the next launch.  omega and offset are squared from round to round (fri.py:86-87).
// Out: trees_out[r] (rounds trees; [0] is over d_codeword), vecs_out[r] (rounds - 1 folded codewords, library-owned),
// roots_out (64 * rounds bytes), alphas_out (2 u64 per fold).
// ---- the rounds of the commit phase from 2^16 elements down as ONE persistent launch (csrc/fri_tail.cuh)
namespace {
std::vector<TailCtl*> g_tail_ctl_free;
std::vector<TailHost*> g_tail_host_free;
// The challenges' way back to the persistent kernel.  With a large BAR the host writes them straight into a few words of fine-grained
// DEVICE memory (one block per TailCtl) and every workgroup polls them there: the polling read of workgroup 0 across the bus, its two
// reads of the challenge and the hand-on through device memory are saved (profiles/r06/prequeue_latency.txt measured the slot).
// STARKCORE_FRI_TAIL_BAR=0 or no large BAR: the pinned word of TailHost, as before.
std::map<TailCtl*, uint64_t*> g_tail_alpha_bar;
bool g_tail_bar_gave_up = false;      // a kernel that waited for a challenge written through the BAR timed out (not a test's stall): the pinned word from then on
uint64_t* tail_alpha_bar_of(TailCtl* ctl) {
    static const int want = [] {
        const char* e = getenv("STARKCORE_FRI_TAIL_BAR");
        if (e && atoi(e) == 0) return 0;
        int large_bar = 0, dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
        return large_bar ? 1 : 0;
    }();
    if (!want || g_tail_bar_gave_up) return nullptr;
    auto it = g_tail_alpha_bar.find(ctl);
    if (it != g_tail_alpha_bar.end()) return it->second;
    uint64_t* p = nullptr;
    if (hipExtMallocWithFlags((void**)&p, 4096, hipDeviceMallocFinegrained) != hipSuccess || hipMemset(p, 0, 4096) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        (void)hipGetLastError();
        if (p) (void)hipFree(p);
        p = nullptr;
    }
    g_tail_alpha_bar[ctl] = p;
    return p;
}
uint64_t g_tail_seq = 0;
uint64_t g_tail_launches = 0, g_tail_fallbacks = 0;     // sc_fri_tail_stats
int tail_blocks_get(TailCtl** c, TailHost** h) {
    if (g_tail_ctl_free.empty()) {
        TailCtl* d = nullptr;
        HIPCHK(hipMalloc((void**)&d, sizeof(TailCtl)));
        HIPCHK(hipMemset(d, 0, sizeof(TailCtl)));
        g_tail_ctl_free.push_back(d);
    }
    if (g_tail_host_free.empty()) {
        TailHost* p = nullptr;
        HIPCHK(hipHostMalloc((void**)&p, sizeof(TailHost), hipHostMallocCoherent | hipHostMallocMapped));
        memset((void*)p, 0, sizeof(TailHost));
        g_tail_host_free.push_back(p);
    }
    *c = g_tail_ctl_free.back(); g_tail_ctl_free.pop_back();
    *h = g_tail_host_free.back(); g_tail_host_free.pop_back();
    return SC_OK;
}
constexpr int TAIL_NOT_TAKEN = 1;
constexpr int TAIL_GAVE_UP = 2;           // a launch whose wait timed out: the caller finishes this commit without trying the kernel again
}  // namespace

// Codewords `first` .. rounds - 1 of the commit phase (trees, roots, the folds between them) by fri_tail_kernel; `cur` = codeword
// first - 1 (n elements, its root already in the transcript `items`), alpha0 = the challenge of the fold that makes codeword `first`
// (into vecs_out[first - 1], allocated by the caller).  The host thread stays in the loop for the Fiat-Shamir step only: per round
// it polls the root's pinned slot, hashes the transcript and writes the next challenge to the pinned word the kernel polls.
// SC_OK: trees_out / vecs_out / roots_out / alphas_out filled to the end (and *last_there set when `last_host` got the last codeword);
// TAIL_NOT_TAKEN: the shape is not the kernel's; TAIL_GAVE_UP: a wait inside it timed out -- either way nothing the caller holds has changed.
static int fri_tail_rounds(std::unique_lock<std::mutex>& lk, const Fe* cur, uint64_t n, Fe alpha0, Fe off, Fe om, uint32_t first, uint32_t rounds,
                           std::vector<uint8_t>& items, uint64_t prior_count, sc_vec_t** vecs_out, sc_merkle_t** trees_out, uint8_t* roots_out,
                           uint64_t* alphas_out, hipStream_t st, Fe* last_host, bool* last_there, const std::function<void()>* on_last) {
    if (g.fri_tail < 0) { const char* e = getenv("STARKCORE_FRI_TAIL"); g.fri_tail = (e && atoi(e) == 0) ? 0 : 1; }      // (sc_set_tuning("fri_tail") overrides)
    const uint32_t R = rounds - first;
    const uint64_t n0 = n / 2;
    if (!g.fri_tail || R < 1 || R >= (uint32_t)TAIL_MAX_ROUNDS || n0 > (1ull << TAIL_MAX_LOG) || (n0 >> (R - 1)) < 1) return TAIL_NOT_TAKEN;
    PowTables* pw;
    Fe i2o;
    SCCHK(fold_constants(n, off, om, st, &pw, &i2o));
    TailCtl* ctl;
    TailHost* host;
    SCCHK(tail_blocks_get(&ctl, &host));
    TailParams P;
    memset((void*)&P, 0, sizeof P);
    P.in0 = cur; P.log_n0 = (uint32_t)ilog2(n0); P.rounds = R; P.alpha0 = alpha0; P.pw_lo = pw->lo; P.pw_hi = pw->hi;
    P.ctl = ctl; P.host = host; P.seq = ++g_tail_seq;
    static_assert(TAIL_MAX_ROUNDS * 8 * sizeof(uint64_t) <= 4096, "the challenge block");
    P.alpha_bar = tail_alpha_bar_of(ctl);
    static const bool tracing = getenv("STARKCORE_FRI_TIMING") != nullptr;
    P.trace = tracing ? 1 : 0;
    P.spin_limit = g.fri_tail_stall >= 0 ? (1u << 13) : TAIL_SPIN_LIMIT;
    std::vector<double> host_us;                           // tracing: when each root was seen and each challenge written (host clock)
    const auto t_launch = std::chrono::steady_clock::now();
    auto host_stamp = [&] { if (tracing) host_us.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_launch).count()); };
    std::vector<Fe*> new_vecs;             // codewords first + 1 ..: allocated here
    std::vector<uint64_t*> new_levels;
    const size_t items_before = items.size();
    auto give_back = [&](bool clean) {
        for (size_t k = 0; k < new_vecs.size(); ++k) pool_free(new_vecs[k], ((n0 >> (k + 1)) ? (n0 >> (k + 1)) : 1) * sizeof(Fe));
        for (size_t k = 0; k < new_levels.size(); ++k) pool_free(new_levels[k], (2 * (n0 >> k) - 1) * 64);
        items.resize(items_before);
        if (clean) g_tail_ctl_free.push_back(ctl);
        else if (hipMemset(ctl, 0, sizeof(TailCtl)) == hipSuccess) g_tail_ctl_free.push_back(ctl);      // counters of an aborted call
        g_tail_host_free.push_back(host);
    };
    Fe i2o_k = i2o;
    for (uint32_t k = 0; k < R; ++k) {
        const uint64_t nk = n0 >> k;
        Fe* out = nullptr;
        if (k == 0) out = vecs_out[first - 1]->d;
        else {
            if (pool_alloc((void**)&out, nk * sizeof(Fe)) != hipSuccess) { give_back(true); return fail(SC_ERR_HIP, "out of device memory"); }
            new_vecs.push_back(out);
        }
        uint64_t* levels = nullptr;
        if (pool_alloc((void**)&levels, (2 * nk - 1) * 64) != hipSuccess) { give_back(true); return fail(SC_ERR_HIP, "out of device memory"); }
        new_levels.push_back(levels);
        P.rd[k].out = out; P.rd[k].levels = levels; P.rd[k].i2o_m = i2o_k;
        const Fe sq = mont_mul(i2o_k, i2o_k);          // the next codeword's offset is this one's squared (fri.py:87): 1/(2 o^2) = 2 (1/(2 o))^2
        i2o_k = fe_add(sq, sq);
    }
    uint32_t nwg = 1;                                          // the widest round's workgroups (a workgroup leaves when no later round needs it)
    for (uint32_t k = 0; k < R; ++k) nwg = std::max(nwg, tail_workgroups(P.log_n0 - k));
    hipLaunchKernelGGL(fri_tail_kernel, dim3(nwg), dim3(256), 0, st, P);
    if (hipGetLastError() != hipSuccess) { give_back(true); return TAIL_NOT_TAKEN; }
    ++g_tail_launches;
    bool aborted = false;
    for (uint32_t k = 0; k < R && !aborted; ++k) {
        volatile uint64_t* slot = host->root[k];
        bool landed = false;
        PendingChallenge pending;                              // what the next challenge's hash can absorb before the root is there
        if (k + 1 < R && !pending.prepare(items, (size_t)(prior_count + first + k + 1))) { aborted = true; break; }
        if (k + 1 == R && last_host && last_there && (n0 >> (R - 1)) <= TAIL_LAST_MAX) {
            // the last codeword leaves the kernel BEFORE its tree is hashed: whatever the caller does with it (sc_fri_prove_dev
            // pickles it into the transcript) happens while the device builds that tree
            lk.unlock();
            bool there = false;
            for (long spin = 0; spin < SPIN_POLLS && !there; ++spin) {
                there = __atomic_load_n(&host->last_flag[0], __ATOMIC_ACQUIRE) == P.seq;
                if (!there && (spin & 1023) == 1023 && __atomic_load_n(&host->abort_flag[0], __ATOMIC_ACQUIRE) == P.seq) break;
            }
            lk.lock();
            if (there) {
                memcpy(last_host, (const void*)host->last, (n0 >> (R - 1)) * sizeof(Fe));
                *last_there = true;
                if (on_last) (*on_last)();
            }
        }
        lk.unlock();
        for (long spin = 0; spin < SPIN_POLLS; ++spin) {
            if (__atomic_load_n(slot + 8, __ATOMIC_ACQUIRE) == P.seq) { landed = true; break; }
            if ((spin & 1023) == 1023 && __atomic_load_n(&host->abort_flag[0], __ATOMIC_ACQUIRE) == P.seq) break;
        }
        lk.lock();
        if (!landed) {
            (void)hipStreamSynchronize(st);                    // the kernel gives up on its own (TAIL_SPIN_LIMIT)
            (void)hipGetLastError();
            landed = __atomic_load_n(slot + 8, __ATOMIC_ACQUIRE) == P.seq && __atomic_load_n(&host->abort_flag[0], __ATOMIC_ACQUIRE) != P.seq;
            if (!landed) { aborted = true; break; }
        }
        const uint32_t r = first + k;
        host_stamp();
        memcpy(roots_out + 64 * r, (const void*)slot, 64);
        if (k + 1 == R) break;
        uint8_t digest[32];
        pending.finish(roots_out + 64 * r, digest, 32);
        transcript_item(items, roots_out + 64 * r, 64);
        const Fe alpha = sample_field(digest, 32);
        alphas_out[2 * r] = alpha.lo; alphas_out[2 * r + 1] = alpha.hi;
        if (g.fri_tail_stall == (int)k) { aborted = true; break; }      // tests: the challenge never comes; the kernel's wait gives up
        if (P.alpha_bar) {
            volatile uint64_t* slot = P.alpha_bar + 8u * (k + 1);
            slot[0] = alpha.lo;
            slot[1] = alpha.hi;
            _mm_sfence();
            slot[2] = P.seq;
            _mm_sfence();
        } else {
            host->alpha[k + 1][0] = alpha.lo;
            host->alpha[k + 1][1] = alpha.hi;
            __atomic_store_n(&host->alpha[k + 1][2], P.seq, __ATOMIC_RELEASE);
        }
        host_stamp();
    }
    if (aborted) {
        (void)hipStreamSynchronize(st);                        // (a kernel still waiting for a challenge that will not come times out)
        (void)hipGetLastError();
        if (P.alpha_bar && g.fri_tail_stall < 0) g_tail_bar_gave_up = true;
        give_back(false);
        ++g_tail_fallbacks;
        return TAIL_GAVE_UP;
    }
    if (tracing) {
        // device stamps are a 100 MHz counter: 10 ns each
        fprintf(stderr, "fri_tail_kernel n0=2^%u, %u rounds (%u workgroups); per round, us: leaves | subtree | barrier | top levels + root out | challenge back || host: root seen, challenge written (since launch)\n", P.log_n0, R, nwg);
        for (uint32_t k = 0; k < R; ++k) {
            const volatile uint64_t* sp = host->stamps[k];
            auto d = [&](int a, int b) { return (double)(sp[b] - sp[a]) / 100.0; };
            const bool multi = tail_workgroups(P.log_n0 - k) > 1;
            fprintf(stderr, "  2^%-2u  %6.1f | %6.1f | %6.1f | %6.1f | %6.1f || %8.1f %8.1f   (round start at %.1f)\n", P.log_n0 - k, d(0, 1), d(1, 2), multi ? d(2, 3) : 0.0, multi ? d(3, 4) : d(2, 4),
                    k + 1 < R ? d(4, 5) : 0.0, host_us.size() > 2 * k ? host_us[2 * k] : 0.0, host_us.size() > 2 * k + 1 ? host_us[2 * k + 1] : 0.0, (double)(sp[0] - host->stamps[0][0]) / 100.0);
        }
    }
    for (uint32_t k = 0; k < R; ++k) {
        const uint64_t nk = n0 >> k;
        if (k > 0) vecs_out[first + k - 1] = new sc_vec{new_vecs[k - 1], nk};
        sc_merkle* t = new sc_merkle{new_levels[k], nk, ilog2(nk)};
        memcpy(t->root, roots_out + 64 * (first + k), 64);
        t->have_root = true;
        t->st = st;
        trees_out[first + k] = t;
    }
    g_tail_ctl_free.push_back(ctl);
    g_tail_host_free.push_back(host);
    return SC_OK;
}

static int fri_commit_locked(std::unique_lock<std::mutex>& lk, const void* d_codeword, uint64_t N, const uint64_t offset[2], const uint64_t omega[2], uint32_t rounds,
                             const void* prior_data, const uint32_t* prior_lens, uint64_t prior_count,
                             sc_vec_t** vecs_out, sc_merkle_t** trees_out, uint8_t* roots_out, uint64_t* alphas_out, void* stream,
                             Fe* last_host = nullptr, bool* last_there = nullptr, const std::function<void()>* on_last = nullptr) {
    SCCHK(ensure_init());
    if (!d_codeword || !trees_out || !roots_out || (rounds > 1 && (!vecs_out || !alphas_out)) || rounds < 1) return fail(SC_ERR_BAD_ARG, "null argument");
    if (N < 2 || !is_pow2(N) || rounds > 60 || (N >> (rounds - 1)) < 1) return fail(SC_ERR_NOT_POW2, "codeword length must be a power of two >= 2 that survives the folds");
    if (prior_count + rounds > TRANSCRIPT_MAX_ITEMS) return fail(SC_ERR_UNSUPPORTED, "transcript too long for the fixed layout");
    std::vector<uint8_t> items;
    {
        const uint8_t* p = (const uint8_t*)prior_data;
        for (uint64_t i = 0; i < prior_count; ++i) {
            if (prior_lens[i] > 255) return fail(SC_ERR_UNSUPPORTED, "transcript item too long for the fixed layout");
            transcript_item(items, p, prior_lens[i]);
            p += prior_lens[i];
        }
    }
    if (items.size() + 67ull * rounds > TRANSCRIPT_MAX_BYTES) return fail(SC_ERR_UNSUPPORTED, "transcript too large for the fixed layout");
    hipStream_t st = pick_stream(stream);
    Fe off = fe_from(offset), om = fe_from(omega);
    const Fe* cur = (const Fe*)d_codeword;
    uint64_t n = N;
    uint32_t made_trees = 0, made_vecs = 0;
    bool tail_gave_up = false;
    auto undo = [&](int rc) {
        (void)hipStreamSynchronize(st);
        for (uint32_t i = 0; i < made_trees; ++i) { sc_merkle* t = trees_out[i]; if (t->slot >= 0) (void)merkle_root_wait(t, true); pool_free(t->d_levels, (2 * t->N - 1) * 64); delete t; trees_out[i] = nullptr; }
        for (uint32_t i = 0; i < made_vecs; ++i) { pool_free(vecs_out[i]->d, (vecs_out[i]->n ? vecs_out[i]->n : 1) * sizeof(Fe)); delete vecs_out[i]; vecs_out[i] = nullptr; }
        return rc;
    };
    int rc = merkle_build_device(cur, n, nullptr, &trees_out[0], st, BUILD_ASYNC);
    if (rc != SC_OK) return rc;
    made_trees = 1;
    for (uint32_t r = 0; r < rounds; ++r) {
        // everything that does not need the root first: the next round's output vector
        sc_vec* nxt = nullptr;
        if (r + 1 < rounds) {
            nxt = new sc_vec{nullptr, n / 2};
            hipError_t e = pool_alloc((void**)&nxt->d, (n / 2 ? n / 2 : 1) * sizeof(Fe));
            if (e != hipSuccess) { delete nxt; return undo(fail(SC_ERR_HIP, hipGetErrorString(e))); }
            vecs_out[r] = nxt;
            ++made_vecs;
        }
        // ... and the part of the challenge's hash that does not need it either (csrc/transcript.h: PendingChallenge)
        PendingChallenge pending;
        if (r + 1 < rounds && !pending.prepare(items, (size_t)(prior_count + r + 1))) return undo(fail(SC_ERR_UNSUPPORTED, "transcript too large for the fixed layout"));
        root_poll_unlocked(lk, trees_out[r]);
        rc = merkle_root_wait(trees_out[r]);
        if (rc != SC_OK) return undo(rc);
        memcpy(roots_out + 64 * r, trees_out[r]->root, 64);
        if (r + 1 == rounds) break;
        uint8_t digest[32];
        pending.finish(trees_out[r]->root, digest, 32);
        transcript_item(items, trees_out[r]->root, 64);
        const Fe alpha = sample_field(digest, 32);
        alphas_out[2 * r] = alpha.lo; alphas_out[2 * r + 1] = alpha.hi;
        if (!tail_gave_up && n / 2 <= (1ull << TAIL_MAX_LOG)) {
            // from here on every round is latency: the rest of the commit phase is one persistent launch (csrc/fri_tail.cuh)
            rc = fri_tail_rounds(lk, cur, n, alpha, off, om, r + 1, rounds, items, prior_count, vecs_out, trees_out, roots_out, alphas_out, st, last_host, last_there, on_last);
            if (rc == SC_OK) return SC_OK;
            if (rc == TAIL_GAVE_UP) tail_gave_up = true;
            else if (rc != TAIL_NOT_TAKEN) return undo(rc);
        }
        rc = fold_and_build(cur, n, alpha, off, om, nxt->d, &trees_out[r + 1], st);
        if (rc != SC_OK) return undo(rc);
        ++made_trees;
        cur = nxt->d;
        n /= 2;
        om = fe_mul(om, om);
        off = fe_mul(off, off);
    }
    return SC_OK;
}
int sc_fri_commit_dev(const void* d_codeword, uint64_t N, const uint64_t offset[2], const uint64_t omega[2], uint32_t rounds,
                      const void* prior_data, const uint32_t* prior_lens, uint64_t prior_count,
                      sc_vec_t** vecs_out, sc_merkle_t** trees_out, uint8_t* roots_out, uint64_t* alphas_out, void* stream) {
    std::unique_lock<std::mutex> lk(g_mu);
    return fri_commit_locked(lk, d_codeword, N, offset, omega, rounds, prior_data, prior_lens, prior_count, vecs_out, trees_out, roots_out, alphas_out, stream);
}

// ---- pinned, device-visible host memory from a pool (hipHostMalloc of megabytes costs hundreds of microseconds; a proof's openings
// are 2-3 MB).  What sc_fri_prove_dev writes its answers to: the query kernel stores them across the bus itself.
namespace {
std::multimap<size_t, void*> g_host_pool;               // free buffers by (rounded) size
std::map<void*, size_t> g_host_live;                    // buffers handed out
unsigned* g_query_ticket = nullptr;                     // device counter of merkle_query_multi_kernel's completion protocol
uint64_t* g_query_flag = nullptr;                       // pinned word the last workgroup writes the call's sequence number to
uint64_t g_query_seq = 0;
size_t host_pool_size(size_t bytes) { size_t b = 4096; while (b < bytes) b <<= 1; return b; }
}  // namespace
int sc_fri_tail_stats(uint64_t out[2]) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!out) return fail(SC_ERR_BAD_ARG, "null argument");
    out[0] = g_tail_launches;
    out[1] = g_tail_fallbacks;
    return SC_OK;
}
int sc_host_alloc(uint64_t bytes, void** out) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    if (!out) return fail(SC_ERR_BAD_ARG, "null argument");
    const size_t want = host_pool_size((size_t)bytes);
    auto it = g_host_pool.find(want);
    void* p = nullptr;
    if (it != g_host_pool.end()) { p = it->second; g_host_pool.erase(it); }
    else HIPCHK(hipHostMalloc(&p, want, hipHostMallocCoherent | hipHostMallocMapped));
    g_host_live[p] = want;
    *out = p;
    return SC_OK;
}
int sc_host_free(void* p) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!p) return SC_OK;
    auto it = g_host_live.find(p);
    if (it == g_host_live.end()) return fail(SC_ERR_BAD_ARG, "not a buffer of sc_host_alloc");
    g_host_pool.emplace(it->second, p);
    g_host_live.erase(it);
    return SC_OK;
}

// openings of codeword j of a commit phase of `rounds` codewords in sc_fri_prove_dev's answers: its own round's a and b; the c positions
// of the round before are among them (c = that round's a, which is this round's a or b), except in the last codeword, which has no round of its own
static inline uint64_t fri_openings_of_codeword(uint32_t j, uint32_t rounds, uint32_t s) { return j + 1 < rounds ? 2ull * s : (j > 0 ? (uint64_t)s : 0ull); }
// Fri.prove (fri.py:115-130) in ONE call -- see include/starkcore.h.  The commit phase is sc_fri_commit_dev's; then, without
// leaving the library: the last codeword comes to the host, the transcript [prior digests..., roots..., [last codeword]] is
// pickled (csrc/proof_pickle.h) and hashed (SHAKE-256, ip.py:18-25), the top-level indices are sampled (fri.py:36-51, :122), the
// positions every round opens are derived from them (fri.py:124-128: a, b = a + half of the round, c = a of the previous round;
// each codeword's positions in the order a..., b..., c...), the caller's further (tree, vector) pairs get the sorted positions
// {i, i + shift, i + N/2, i + shift + N/2 mod N} (fast_stark.py:154-158), and ONE kernel gathers every element and path.
// `answers`: [elements 16 B each, padded to 256 B][paths][indices u64]; a buffer of sc_host_alloc is written by the kernel
// directly (no staging, no copy engine; the last workgroup flags completion in a pinned word the host polls), any other host
// pointer is served through device scratch and two copies.
int sc_fri_prove_dev(const void* d_codeword, uint64_t N, const uint64_t offset[2], const uint64_t omega[2], uint32_t rounds, uint32_t num_tests,
                     const void* prior_data, const uint32_t* prior_lens, uint64_t prior_count,
                     uint64_t extra_count, const sc_merkle_t* const* extra_trees, const void* const* extra_vecs, uint64_t extra_shift,
                     sc_vec_t** vecs_out, sc_merkle_t** trees_out, uint8_t* roots_out, uint64_t* alphas_out,
                     void* last_codeword_out, uint64_t* top_indices_out, uint64_t* extra_indices_out,
                     void* answers, uint64_t answers_bytes, void* stream) {
    std::unique_lock<std::mutex> lk(g_mu);
    if (!last_codeword_out || !top_indices_out || !answers || (extra_count && (!extra_trees || !extra_vecs || !extra_indices_out)) || !num_tests)
        return fail(SC_ERR_BAD_ARG, "null argument");
    if (rounds < 1 || rounds > 60 || N < 2 || !is_pow2(N)) return fail(SC_ERR_NOT_POW2, "codeword length must be a power of two >= 2 that survives the folds");
    // vecs_out == trees_out == NULL: nobody wants the folded codewords and their trees once the openings are on the host -- the
    // library keeps them to itself and hands the memory back before it returns (no handle crosses the boundary)
    std::vector<sc_vec_t*> own_vecs;
    std::vector<sc_merkle_t*> own_trees;
    std::vector<uint64_t> own_alphas;
    const bool keep_state = vecs_out != nullptr || trees_out != nullptr;
    if (!keep_state) {
        own_vecs.assign(rounds > 1 ? rounds - 1 : 1, nullptr);
        own_trees.assign(rounds, nullptr);
        vecs_out = own_vecs.data();
        trees_out = own_trees.data();
    }
    if (!alphas_out) { own_alphas.assign(2 * (rounds > 1 ? rounds - 1 : 1), 0); alphas_out = own_alphas.data(); }
    const uint64_t n_last = N >> (rounds - 1);
    const uint32_t s = num_tests;
    if (n_last < 1 || s > n_last || rounds + extra_count > QUERY_MAX_TREES) return fail(SC_ERR_UNSUPPORTED, "shape not served by the one-call prover");
    for (uint64_t t = 0; t < extra_count; ++t)
        if (!extra_trees[t] || !extra_vecs[t] || extra_trees[t]->N != N) return fail(SC_ERR_BAD_ARG, "further codewords must have the domain's length");
    // sizes first: nothing is enqueued for a call that cannot be answered
    uint64_t total = 0;
    size_t path_bytes = 0;
    for (uint32_t j = 0; j < rounds; ++j) {
        const uint64_t k = fri_openings_of_codeword(j, rounds, s);
        total += k;
        path_bytes += k * 64 * (size_t)ilog2(N >> j);
    }
    total += extra_count * 4ull * s;
    path_bytes += extra_count * 4ull * s * 64 * (size_t)ilog2(N);
    const size_t el_bytes = (total * sizeof(Fe) + 255) & ~(size_t)255;
    if (answers_bytes < el_bytes + path_bytes + total * 8) return fail(SC_ERR_BAD_ARG, "answer buffer too small");
    {
        const auto live = g_host_live.find(answers);           // a buffer of sc_host_alloc is written by the kernel: its real size counts, not the caller's word
        if (live != g_host_live.end() && live->second < el_bytes + path_bytes + total * 8) return fail(SC_ERR_BAD_ARG, "answer buffer too small");
    }
    // STARKCORE_FRI_TIMING=1: the call's phases on stderr (host clock, microseconds) -- tools/fri_prove_timing.py
    static const bool timing = getenv("STARKCORE_FRI_TIMING") != nullptr;
    std::chrono::steady_clock::time_point tp[8];
    auto stamp = [&](int i) { if (timing) tp[i] = std::chrono::steady_clock::now(); };
    stamp(0);
    // the transcript the indices are sampled from: [prior..., roots..., [FieldElement(v) for v in last codeword]] (fri.py:91, :122).
    // Its pickle depends on the LENGTHS of the byte strings only, so it is written as soon as the last codeword is known -- the
    // persistent tail kernel hands it over before it hashes the last tree -- with the roots that are still missing left blank.
    ProofPickler pk;
    std::vector<size_t> bytes_at;
    uint8_t modulus[17] = {0};
    { const uint64_t plo = P_LO, phi = P_HI; memcpy(modulus, &plo, 8); memcpy(modulus + 8, &phi, 8); }
    pk.moduli = modulus; pk.nfields = 1; pk.modulus_bytes = 17;
    pk.bytes_at = &bytes_at;
    bool pickled = false, pickle_ok = false;
    auto pickle_transcript = [&] {
        std::vector<uint8_t> ops;
        auto put32 = [&](uint32_t v) { for (int i = 0; i < 4; ++i) ops.push_back((uint8_t)(v >> (8 * i))); };
        ops.reserve(16 + 72 * (prior_count + rounds) + 29 * n_last);
        ops.push_back('L'); put32((uint32_t)(prior_count + rounds + 1));
        const uint8_t* p = (const uint8_t*)prior_data;
        for (uint64_t i = 0; i < prior_count; ++i) { ops.push_back('B'); put32(prior_lens[i]); ops.insert(ops.end(), p, p + prior_lens[i]); p += prior_lens[i]; }
        for (uint32_t r = 0; r < rounds; ++r) { ops.push_back('B'); put32(64); ops.insert(ops.end(), roots_out + 64 * r, roots_out + 64 * r + 64); }
        ops.push_back('L'); put32((uint32_t)n_last);
        for (uint64_t i = 0; i < n_last; ++i) {
            ops.push_back('E'); put32(0);
            for (int b = 0; b < 8; ++b) ops.push_back((uint8_t)(i >> (8 * b)));
            const uint8_t* v = (const uint8_t*)last_codeword_out + 16 * i;
            ops.insert(ops.end(), v, v + 16);
        }
        bytes_at.clear();
        pickle_ok = pk.run(ops.data(), ops.size()) && bytes_at.size() == prior_count + rounds;
        pickled = true;
    };
    const std::function<void()> on_last = pickle_transcript;
    bool last_there = false;       // the persistent tail kernel hands the last codeword over with its roots
    int rc = fri_commit_locked(lk, d_codeword, N, offset, omega, rounds, prior_data, prior_lens, prior_count, vecs_out, trees_out, roots_out, alphas_out, stream,
                               (Fe*)last_codeword_out, &last_there, &on_last);
    if (rc != SC_OK) return rc;
    stamp(1);
    hipStream_t st = pick_stream(stream);
    auto free_state = [&] {
        for (uint32_t i = 0; i < rounds; ++i) { sc_merkle* t = trees_out[i]; if (!t) continue; pool_free(t->d_levels, (2 * t->N - 1) * 64); delete t; trees_out[i] = nullptr; }
        for (uint32_t i = 0; i + 1 < rounds; ++i) { if (!vecs_out[i]) continue; pool_free(vecs_out[i]->d, (vecs_out[i]->n ? vecs_out[i]->n : 1) * sizeof(Fe)); delete vecs_out[i]; vecs_out[i] = nullptr; }
    };
    auto undo = [&](int code) {
        (void)hipStreamSynchronize(st);
        free_state();
        return code;
    };
    // the last codeword in the clear (fri.py:91): its tree is built, so the fold that made it has run
    const Fe* d_last = rounds > 1 ? vecs_out[rounds - 2]->d : (const Fe*)d_codeword;
    if (!last_there && (hipMemcpyAsync(last_codeword_out, d_last, n_last * sizeof(Fe), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess))
        return undo(fail(SC_ERR_HIP, "copy of the last codeword failed"));
    stamp(2);
    if (pickled && pickle_ok) {
        // written while the last tree was being hashed: only the roots that were not there yet go in now
        for (uint32_t r = 0; r < rounds; ++r) memcpy(pk.base + bytes_at[prior_count + r], roots_out + 64 * r, 64);
    } else {
        pickle_transcript();
        if (!pickle_ok) return undo(fail(SC_ERR_UNSUPPORTED, "transcript not described"));
    }
    uint8_t seed[32];
    shake256(pk.base, pk.used, seed, 32);
    stamp(3);
    if (!fri_sample_indices(seed, 32, N / 2, n_last, s, top_indices_out)) return undo(fail(SC_ERR_UNSUPPORTED, "cannot sample that many indices"));
    stamp(7);
    // positions: written where the kernel reads them (behind the answers)
    uint8_t* const base = (uint8_t*)answers;
    uint64_t* idx = (uint64_t*)(base + el_bytes + path_bytes);
    std::vector<uint64_t> cur(top_indices_out, top_indices_out + s), prev;
    uint64_t at = 0;
    for (uint32_t j = 0; j < rounds; ++j) {
        const uint64_t half = (N >> j) / 2;
        if (j + 1 < rounds) {
            for (uint32_t t = 0; t < s; ++t) cur[t] %= half;
            for (uint32_t t = 0; t < s; ++t) idx[at++] = cur[t];
            for (uint32_t t = 0; t < s; ++t) idx[at++] = cur[t] + half;
        } else if (j > 0) {
            // (only the LAST codeword opens the c positions of the round before on their own: everywhere else c = prev[t] is this
            // codeword's a[t] or b[t] -- prev[t] mod half resp. that + half -- and is opened once)
            for (uint32_t t = 0; t < s; ++t) idx[at++] = prev[t];
        }
        prev = cur;
    }
    if (extra_count) {
        std::vector<uint64_t> quad;
        quad.reserve(4 * s);
        for (uint32_t t = 0; t < s; ++t) quad.push_back(top_indices_out[t]);
        for (uint32_t t = 0; t < s; ++t) quad.push_back((top_indices_out[t] + extra_shift) % N);
        for (uint32_t t = 0; t < 2 * s; ++t) quad.push_back((quad[t] + N / 2) % N);
        std::sort(quad.begin(), quad.end());
        memcpy(extra_indices_out, quad.data(), 4ull * s * 8);
        for (uint64_t e = 0; e < extra_count; ++e) { memcpy(idx + at, quad.data(), 4ull * s * 8); at += 4ull * s; }
    }
    stamp(4);
    // one launch for every opening of the proof
    const bool direct = g_host_live.count(answers) != 0;
    Fe* d_el; uint64_t* d_paths; const uint64_t* d_idx;
    if (direct) {
        d_el = (Fe*)base; d_paths = (uint64_t*)(base + el_bytes); d_idx = idx;
        if (!g_query_ticket) {
            if (hipMalloc((void**)&g_query_ticket, 256) != hipSuccess || hipMemset(g_query_ticket, 0, 256) != hipSuccess ||
                hipHostMalloc((void**)&g_query_flag, 256, hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess)
                return undo(fail(SC_ERR_HIP, "completion flag of the query kernel"));
            g_query_flag[0] = 0;
        }
    } else {
        void* buf;
        rc = scratch(5, el_bytes + path_bytes + total * 8 + 256, &buf);
        if (rc != SC_OK) return undo(rc);
        d_el = (Fe*)buf; d_paths = (uint64_t*)((char*)buf + el_bytes); d_idx = (const uint64_t*)((char*)buf + el_bytes + path_bytes);
        if (hipMemcpyAsync((void*)d_idx, idx, total * 8, hipMemcpyHostToDevice, st) != hipSuccess) return undo(fail(SC_ERR_HIP, "index upload"));
    }
    QueryTrees Q;
    Q.count = 0;
    Q.total_threads = 0;
    uint64_t off = 0, poff = 0;
    for (uint64_t t = 0; t < rounds + extra_count; ++t) {
        const bool own = t < rounds;
        const sc_merkle* tree = own ? trees_out[t] : extra_trees[t - rounds];
        const uint64_t k = own ? fri_openings_of_codeword((uint32_t)t, rounds, s) : 4ull * s;
        if (!k) continue;
        QueryTree& T = Q.t[Q.count++];
        T.levels = tree->d_levels;
        T.elems = own ? (t == 0 ? (const Fe*)d_codeword : vecs_out[t - 1]->d) : (const Fe*)extra_vecs[t - rounds];
        T.N = tree->N;
        T.logN = (uint32_t)tree->logN;
        T.per_query = 4 * T.logN + 1;
        T.thread_off = Q.total_threads;
        T.idx_off = off;
        T.path_off = poff;
        Q.total_threads += k * T.per_query;
        off += k;
        poff += k * (uint64_t)tree->logN;
    }
    QueryDone done;
    const uint64_t seq = ++g_query_seq;
    if (direct) { done.ticket = g_query_ticket; done.host_flag = g_query_flag; done.seq = seq; }
    hipLaunchKernelGGL(merkle_query_multi_kernel, dim3((unsigned)((Q.total_threads + 255) / 256)), dim3(256), 0, st, Q, d_idx, d_el, d_paths, done);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return undo(fail(SC_ERR_HIP, hipGetErrorString(e)));
    stamp(5);
    if (direct) {
        volatile uint64_t* flag = g_query_flag;
        bool landed = false;
        for (long spin = 0; spin < SPIN_POLLS; ++spin) {
            if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) { landed = true; break; }
            if ((spin & 4095) == 4095) {
                e = hipStreamQuery(st);
                if (e != hipErrorNotReady) break;
                (void)hipGetLastError();
            }
        }
        if (!landed) {
            (void)hipGetLastError();
            e = hipStreamSynchronize(st);
            if (e != hipSuccess || __atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) return undo(fail(SC_ERR_HIP, "the query kernel did not complete"));
        }
    } else {
        if (hipMemcpyAsync(base, d_el, total * sizeof(Fe), hipMemcpyDeviceToHost, st) != hipSuccess ||
            (path_bytes && hipMemcpyAsync(base + el_bytes, d_paths, path_bytes, hipMemcpyDeviceToHost, st) != hipSuccess) ||
            hipStreamSynchronize(st) != hipSuccess)
            return undo(fail(SC_ERR_HIP, "copy of the openings failed"));
    }
    if (!keep_state) free_state();                            // (every kernel that read them has completed: the openings were waited for)
    if (timing) {
        stamp(6);
        auto us = [&](int a, int b) { return std::chrono::duration<double, std::micro>(tp[b] - tp[a]).count(); };
        fprintf(stderr, "sc_fri_prove_dev N=2^%d: commit %.1f | last codeword to host %.1f | pickle + shake %.1f | sample indices %.1f | positions %.1f | launch %.1f | "
                        "wait for the openings (%s) %.1f | total %.1f us\n", ilog2(N), us(0, 1), us(1, 2), us(2, 3), us(3, 7), us(7, 4), us(4, 5), direct ? "pinned" : "staged", us(5, 6), us(0, 6));
    }
    return SC_OK;
}

int sc_merkle_build(const void* elems, uint64_t N, uint8_t root_out[64], sc_merkle_t** tree) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    if (!is_pow2(N)) return fail(SC_ERR_NOT_POW2, "length must be power of two");
    void* a;
    SCCHK(scratch(1, N * sizeof(Fe), &a));
    SCCHK(upload(a, elems, N * sizeof(Fe), g.stream));
    return merkle_build_device((const Fe*)a, N, root_out, tree, g.stream);
}
int sc_merkle_commit(const void* elems, uint64_t N, uint8_t root_out[64]) { return sc_merkle_build(elems, N, root_out, nullptr); }

int sc_merkle_open_batch(const sc_merkle_t* tree, const uint64_t* indices, uint64_t k, uint8_t* paths_out) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    if (!tree) return fail(SC_ERR_BAD_ARG, "null tree");
    if (tree->N < 2) return fail(SC_ERR_BAD_ARG, "cannot open invalid index");
    for (uint64_t i = 0; i < k; ++i) if (indices[i] >= tree->N) return fail(SC_ERR_BAD_ARG, "cannot open invalid index");
    if (k == 0) return SC_OK;
    const size_t idx_bytes = (k * 8 + 255) & ~255ull;
    const size_t out_bytes = k * 64 * (size_t)tree->logN;
    void* buf;
    SCCHK(scratch(5, idx_bytes + out_bytes, &buf));
    uint64_t* d_idx = (uint64_t*)buf;
    uint64_t* d_out = (uint64_t*)((char*)buf + idx_bytes);
    SCCHK(upload(d_idx, indices, k * 8, g.stream));
    uint64_t total = k * (uint64_t)tree->logN * 4;
    hipLaunchKernelGGL(merkle_open_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, g.stream, tree->d_levels, tree->N, tree->logN, d_idx, k, d_out);
    HIPCHK(hipGetLastError());
    return download(paths_out, d_out, out_bytes, g.stream);
}
// elements + authentication paths for k indices in one round trip (the FRI query phase, code/fri.py:98-113)
int sc_merkle_query_dev(const sc_merkle_t* tree, const void* d_elems, const uint64_t* indices, uint64_t k, void* elems_out, uint8_t* paths_out) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    if (!tree || !d_elems) return fail(SC_ERR_BAD_ARG, "null tree or vector");
    for (uint64_t i = 0; i < k; ++i) if (indices[i] >= tree->N) return fail(SC_ERR_BAD_ARG, "cannot open invalid index");
    if (k == 0) return SC_OK;
    const size_t idx_bytes = (k * 8 + 255) & ~255ull;
    const size_t el_bytes = (k * sizeof(Fe) + 255) & ~255ull;
    const size_t path_bytes = k * 64 * (size_t)tree->logN;
    void* buf;
    SCCHK(scratch(5, idx_bytes + el_bytes + path_bytes + 256, &buf));
    uint64_t* d_idx = (uint64_t*)buf;
    Fe* d_el = (Fe*)((char*)buf + idx_bytes);
    uint64_t* d_paths = (uint64_t*)((char*)buf + idx_bytes + el_bytes);
    SCCHK(upload(d_idx, indices, k * 8, g.stream));
    SCCHK(gather_device((const Fe*)d_elems, d_idx, k, d_el, g.stream));
    if (tree->logN > 0) {
        uint64_t total = k * (uint64_t)tree->logN * 4;
        hipLaunchKernelGGL(merkle_open_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, g.stream, tree->d_levels, tree->N, tree->logN, d_idx, k, d_paths);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(elems_out, d_el, k * sizeof(Fe), hipMemcpyDeviceToHost, g.stream));
    if (path_bytes) HIPCHK(hipMemcpyAsync(paths_out, d_paths, path_bytes, hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    return SC_OK;
}

// the same for several (tree, vector) pairs in ONE round trip: Fri.prove's query phase opens every round's codeword
// (fri.py:124-128); counts[t] indices belong to pair t, concatenated in `indices`; outputs are concatenated in the same order
// (elements: 16 bytes each; paths: 64 * logN_t bytes per index of pair t).
int sc_merkle_query_multi_dev(uint64_t n, const sc_merkle_t* const* trees, const void* const* d_elems, const uint64_t* indices, const uint64_t* counts,
                              void* elems_out, uint8_t* paths_out) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    uint64_t total = 0;
    size_t path_bytes = 0;
    for (uint64_t t = 0; t < n; ++t) {
        if (!trees[t] || !d_elems[t]) return fail(SC_ERR_BAD_ARG, "null tree or vector");
        for (uint64_t i = 0; i < counts[t]; ++i) if (indices[total + i] >= trees[t]->N) return fail(SC_ERR_BAD_ARG, "cannot open invalid index");
        total += counts[t];
        path_bytes += counts[t] * 64 * (size_t)trees[t]->logN;
    }
    if (total == 0) return SC_OK;
    const size_t idx_bytes = (total * 8 + 255) & ~255ull;
    const size_t el_bytes = (total * sizeof(Fe) + 255) & ~255ull;
    void* buf;
    SCCHK(scratch(5, idx_bytes + el_bytes + path_bytes + 256, &buf));
    uint64_t* d_idx = (uint64_t*)buf;
    Fe* d_el = (Fe*)((char*)buf + idx_bytes);
    uint64_t* d_paths = (uint64_t*)((char*)buf + idx_bytes + el_bytes);
    SCCHK(upload(d_idx, indices, total * 8, g.stream));
    // one launch per QUERY_MAX_TREES pairs (Fri.prove: one launch)
    uint64_t off = 0, poff = 0;
    for (uint64_t t0 = 0; t0 < n; t0 += QUERY_MAX_TREES) {
        QueryTrees Q;
        Q.count = 0;
        Q.total_threads = 0;
        for (uint64_t t = t0; t < n && t < t0 + QUERY_MAX_TREES; ++t) {
            const uint64_t k = counts[t];
            if (!k) continue;
            QueryTree& T = Q.t[Q.count++];
            T.levels = trees[t]->d_levels;
            T.elems = (const Fe*)d_elems[t];
            T.N = trees[t]->N;
            T.logN = (uint32_t)trees[t]->logN;
            T.per_query = 4 * T.logN + 1;
            T.thread_off = Q.total_threads;
            T.idx_off = off;
            T.path_off = poff;
            Q.total_threads += k * T.per_query;
            off += k;
            poff += k * (uint64_t)trees[t]->logN;
        }
        if (!Q.count) continue;
        hipLaunchKernelGGL(merkle_query_multi_kernel, dim3((unsigned)((Q.total_threads + 255) / 256)), dim3(256), 0, g.stream, Q, d_idx, d_el, d_paths);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(elems_out, d_el, total * sizeof(Fe), hipMemcpyDeviceToHost, g.stream));
    if (path_bytes) HIPCHK(hipMemcpyAsync(paths_out, d_paths, path_bytes, hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    return SC_OK;
}

// ---- pieces of a Merkle tree that is sharded over ranks (stark-anatomy_amd/sharded.py: ShardedFri)
int sc_merkle_level_copy_dev(const sc_merkle_t* tree, int level, void* d_out, void* stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    if (!tree || level < 0 || level > tree->logN) return fail(SC_ERR_BAD_ARG, "no such tree level");
    const uint64_t off = (level == 0) ? 0 : (2 * tree->N - (tree->N >> (level - 1)));
    HIPCHK(hipMemcpyAsync(d_out, tree->d_levels + 8 * off, (tree->N >> level) * 64, hipMemcpyDeviceToDevice, pick_stream(stream)));
    return SC_OK;
}

int sc_merkle_from_digests_dev(const void* d_digests, uint64_t count, uint8_t root_out[64], sc_merkle_t** tree, void* stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    hipStream_t st = pick_stream(stream);
    if (!is_pow2(count)) return fail(SC_ERR_NOT_POW2, "length must be power of two");
    uint64_t* levels = nullptr;
    const size_t tree_bytes = (2 * count - 1) * 64;
    HIPCHK(pool_alloc((void**)&levels, tree_bytes));
    hipError_t e = hipMemcpyAsync(levels, d_digests, count * 64, hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) { pool_free(levels, tree_bytes); return fail(SC_ERR_HIP, hipGetErrorString(e)); }
    // root_out == NULL: asynchronous like sc_merkle_build_async_dev -- the root travels to a pinned slot behind the build and
    // sc_merkle_root polls for it (a blocking stream wait that has gone to sleep costs tens of microseconds to wake up)
    const int slot = root_out ? -1 : root_slot_get();
    if (slot >= 0) {
        const uint64_t seq = ++g.root_seq;
        volatile uint64_t* host = (volatile uint64_t*)(g.root_slots + ROOT_SLOT_BYTES * slot);
        bool published = false;
        int rc = merkle_climb(levels, count, 0, st, host, seq, &published);
        if (rc == SC_OK && !published) {
            hipLaunchKernelGGL(root_publish_kernel, dim3(1), dim3(64), 0, st, (const uint64_t*)(levels + 8 * (2 * count - 2)), host, seq);
            if (hipGetLastError() != hipSuccess) rc = fail(SC_ERR_HIP, "root publish launch failed");
        }
        if (rc != SC_OK) { (void)hipStreamSynchronize(st); g.free_root_slots.push_back(slot); pool_free(levels, tree_bytes); return rc; }
        sc_merkle* t = new sc_merkle{levels, count, ilog2(count)};
        t->slot = slot; t->seq = seq; t->st = st;
        *tree = t;
        return SC_OK;
    }
    uint8_t root_tmp[64];
    if (!root_out) root_out = root_tmp;
    int rc = merkle_finish(levels, count, st);
    if (rc == SC_OK) {
        e = hipMemcpyAsync(root_out, levels + 8 * (2 * count - 2), 64, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) rc = fail(SC_ERR_HIP, hipGetErrorString(e));
    }
    if (rc != SC_OK) { pool_free(levels, tree_bytes); return rc; }
    *tree = new sc_merkle{levels, count, ilog2(count)};
    memcpy((*tree)->root, root_out, 64);
    (*tree)->have_root = true;
    return SC_OK;
}

int sc_fri_fold_slab_dev(const void* d_in, uint64_t rows, uint64_t cols, uint64_t R, uint64_t col_base, const uint64_t alpha[2], const uint64_t offset[2],
                         const uint64_t omega[2], void* d_out, void* stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    hipStream_t st = pick_stream(stream);
    if (rows < 2 || !is_pow2(rows) || !is_pow2(cols) || !is_pow2(R) || col_base + cols > R) return fail(SC_ERR_BAD_ARG, "bad slab shape");
    Fe off = fe_from(offset), om = fe_from(omega);
    if (fe_is_zero(off) || fe_is_zero(om)) return fail(SC_ERR_DIV_ZERO, "divide by zero");
    const uint64_t N = rows * R;
    Fe winv = from_mont(mont_inv(to_mont(om)));
    PowTables* pw;
    SCCHK(get_pow(winv, N / 2, st, &pw));
    Fe c_m = mont_mul(to_mont(fe_from(alpha)), mont_inv(to_mont(fe_add(off, off))));
    const uint64_t total = (rows / 2) * cols;
    hipLaunchKernelGGL(fri_fold_slab_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const Fe*)d_in, (Fe*)d_out, rows / 2, ilog2(cols), R, col_base,
                       pw->lo, pw->hi, c_m);
    HIPCHK(hipGetLastError());
    return SC_OK;
}

int sc_fri_fold_slab_build_dev(const void* d_in, uint64_t rows, uint64_t cols, uint64_t R, uint64_t col_base, const uint64_t alpha[2], const uint64_t offset[2],
                               const uint64_t omega[2], void* d_out, sc_merkle_t** tree, void* stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    hipStream_t st = pick_stream(stream);
    if (!tree || !d_in || !d_out) return fail(SC_ERR_BAD_ARG, "null argument");
    if (rows < 2 || !is_pow2(rows) || !is_pow2(cols) || !is_pow2(R) || col_base + cols > R) return fail(SC_ERR_BAD_ARG, "bad slab shape");
    const uint64_t leaves = (rows / 2) * cols;
    FoldIn f;
    SCCHK(fold_prepare((const Fe*)d_in, rows * R, fe_from(alpha), fe_from(offset), fe_from(omega), (Fe*)d_out, st, &f));
    f.logcols = ilog2(cols);
    f.R = R;
    f.col_base = col_base;
    if (leaves >= 256) return merkle_build_device((const Fe*)d_out, leaves, nullptr, tree, st, BUILD_NOROOT, &f);
    hipLaunchKernelGGL(fri_fold_slab_kernel, dim3((unsigned)((leaves + 255) / 256)), dim3(256), 0, st, (const Fe*)d_in, (Fe*)d_out, rows / 2, f.logcols, R, col_base,
                       f.lo, f.hi, f.c_m);
    HIPCHK(hipGetLastError());
    return merkle_build_device((const Fe*)d_out, leaves, nullptr, tree, st, BUILD_NOROOT);
}

int sc_merkle_open(const sc_merkle_t* tree, uint64_t index, uint8_t* path_out) { return sc_merkle_open_batch(tree, &index, 1, path_out); }
uint64_t sc_merkle_leaves(const sc_merkle_t* tree) { return tree ? tree->N : 0; }
int sc_merkle_free(sc_merkle_t* tree) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!tree) return SC_OK;
    if (tree->slot >= 0) (void)merkle_root_wait(tree, true);  // a root still in flight: let it land, return the slot
    release_after_streams(tree->d_levels, (2 * tree->N - 1) * 64);
    delete tree;
    return SC_OK;
}

}  // extern "C"
