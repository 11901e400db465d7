// ntt_tile.cuh -- one pass of the multi-pass NTT: a tile of R rows x C columns is transformed along
// the row axis (length-R decimation-in-frequency NTT per column) with the tile held in LDS.
//
// Replaces the recursive radix-2 of reference code/ntt.py:3-18.  The whole transform of length
// n = N_1 * N_2 * ... * N_m (digits chosen by the host, ntt_plan.h, driven from core.hip) runs m such passes:
//
//   x[j],  j = j_1*(N_2..N_m) + j_2*(N_3..N_m) + ... + j_m        (natural order in)
//   X[k],  k = k_1 + N_1*k_2 + N_1*N_2*k_3 + ...                  (natural order out)
//
//   pass i < m ("column pass", in place):  memory viewed as [A][R=N_i][B], A = N_1..N_{i-1} (already
//       transformed digits), B = N_{i+1}..N_m; length-R NTT over the middle axis for every (a, b),
//       then the four-step twiddle  w_n^(A * b * k_i).  A tile = all R rows x C adjacent b.
//   pass m ("transposing pass", out of place):  contiguous rows of length R = N_m are transformed and
//       written to natural positions; a tile takes C adjacent k_1 so that the stores form C*16-byte runs.
//
// Inside a tile the R-point NTT is radix-2 DIF, log2(E) stages per round held in registers
// (E = elements per thread), rounds exchanging through LDS; DIF leaves row position rp holding output
// row bitrev(rp), which is undone for free in the store addressing.
//
// The round body is host+device so tests/ can run the identical index logic on the CPU (emulation of
// one workgroup = loop over thread ids per round).
#pragma once
#include "field.cuh"

namespace sc {

#ifndef SC_MAX_BLOCKS
#define SC_MAX_BLOCKS 16        // ranks of one node a column stage can address (destination table of PassParams)
#endif

struct PassParams {
    const Fe* in;
    Fe* out;
    int logR, logC;            // tile geometry; threads = 2^(logR+logC-LOGE)
    // tile id t -> (t_hi, t_mid, t_lo);  t_lo = t & (2^lo_log - 1), t_mid = (t >> lo_log) & (2^mid_log - 1)
    int lo_log, mid_log;
    uint64_t in_hi, in_mid, in_lo, in_rs, in_cs;      // element strides (units of Fe)
    // optional split of the input row stride (chunked input left by the all-to-all of the multi-GPU four-step):
    // row r contributes (r & (2^in_split - 1)) * in_rs + (r >> in_split) * in_rs_hi ; in_split = 0 disables the split
    int in_split;
    uint64_t in_rs_hi;
    uint64_t out_hi, out_mid, out_lo, out_rs, out_cs;
    int rfast_load;            // round 0 maps lanes along rows (transposing pass: in_rs == 1)
    // butterfly twiddles: mt[e << mt_shift] = w_R^e in Montgomery form, e < R/2
    const Fe* mt;
    int mt_shift;
    // four-step twiddle  w_n^(colidx * k * tw_scale), colidx = t_lo*C + c ; tl[e & 4095] * th[e >> 12]
    int tw_enable;
    int tw_col_shift;          // colidx >>= tw_col_shift (batched column transforms: low column bits are the batch, not the transform)
    uint64_t tw_scale;
    // generalisation used by the fused OUTER twiddle of the multi-GPU four-step: exponent =
    //   (tw_col_base + colidx) * (k * tw_row_k + t_mid * tw_row_mid) * tw_scale      (defaults 0, 1, 0 give the plain form)
    uint64_t tw_col_base, tw_row_k, tw_row_mid;
    const Fe* tl;
    const Fe* th;
    // optional direct table of the same twiddles: twd[k * twd_stride + colidx] = w_n^(colidx * k * tw_scale) [* n^-1];
    // laid out like the pass's output tile, so the load coalesces exactly like the store (nullptr = two-level lookup)
    const Fe* twd;
    uint64_t twd_stride;
    // the SAME four-step twiddles applied by the NEXT pass on load instead (twd_in[j & twd_in_mask], j = memory index of the
    // element in the work buffer): the loads go out together with the data loads at the start of the workgroup, where the
    // latency is paid anyway, instead of sitting between the last butterfly and the store of the producing pass
    const Fe* twd_in;
    uint64_t twd_in_mask;
    // final constant multiply (Montgomery form), e.g. n^-1 for a single-pass inverse transform
    int scale_enable;
    Fe scale;
    // input side (first pass only; memory index == natural index j there)
    uint64_t in_limit;         // j >= in_limit reads as zero (zero padding, ntt.py:134 / :51-56)
    int coset_enable;          // x[j] *= offset^j (Polynomial.scale, univariate.py:153-154): ol[j & 4095] * oh[j >> 12]
    const Fe* ol;
    const Fe* oh;
    // zero-padded input (LDE, fast_multiply): only the first R >> prune_log rows of every column are non-zero, so in the top
    // prune_log stages every butterfly has v = 0 and degenerates to (u, u * w): no add/sub, and nothing at all where u is a
    // zero row too (whole waves skip).  0 = off.
    int prune_log;
    // 1: the grid is a single wave of workgroups (<= one per CU, e.g. 2^20): a wave lowers its priority as it advances through a
    // round, so the waves of a SIMD reach the barrier together instead of the oldest always winning the VALU arbitration and
    // the youngest finishing alone at half the issue rate (+2-4 % at 2^20; -3..5 % when other workgroups fill the gaps anyway).
    // 2: a long grid (a batch of 2^20 columns, a 2^22 or 2^24 transform: the waves of the next workgroup arrive one by one while
    // others are computing): a wave runs at top priority until its first-round loads are issued, so they leave at once instead
    // of queueing behind that arithmetic (2^20 x 16-64 columns +3.5-6 %, 2^24 +3.5-4.5 %, 2^22 +1.6 %; -2..4 % on grids of 4 per CU)
    int prio_balance;
    // optional DESTINATION TABLE for the natural output rows (column stage of the multi-GPU four-step): the rows are cut into
    // blocks of 2^blk_log rows, block h = what rank h receives in the corner turn, and an element of natural row
    //   k * blk_row_k + t_mid * blk_row_mid
    // is stored at out_blk[row >> blk_log][j] instead of out[j] (same element index j).  Two uses: the block a rank keeps for
    // itself goes straight into its own receive buffer (the diagonal of the corner turn is neither copied nor sent; every
    // other entry = `out`), and the DIRECT-STORE corner turn: entry h points into rank h's receive buffer, mapped through
    // HIP IPC, so the stores themselves cross xGMI and no collective kernel, send buffer or extra HBM round trip exists.
    // blk_enable = 0: off (plain `out`).
    int blk_enable;
    int blk_log;
    uint32_t blk_row_k, blk_row_mid;
    Fe* out_blk[SC_MAX_BLOCKS];
    // several independent transforms in one launch (NttIo::cols, sc_ntt_columns_dev): tile index = column * 2^col_tiles_log + the
    // tile of that column's own pass; the column only shifts the element index: by column * col_stride into `out`, by column *
    // col_stride_in into `in` (they differ in the first pass of a batch of zero-padded inputs: m coefficients in, n values out).
    int col_enable;
    int col_tiles_log;
    uint64_t col_stride, col_stride_in;
    // diagnostics (tools/pass_trace.py): per-wave s_memtime stamps of the phases of a workgroup; nullptr in production
    unsigned long long* trace;
};

SC_HD uint32_t bitrev32(uint32_t x, int bits) {
#if defined(__HIP_DEVICE_COMPILE__)
    return bits ? (__brev(x) >> (32 - bits)) : 0u;
#else
    uint32_t r = 0;
    for (int i = 0; i < bits; ++i) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
#endif
}

// LDS placement of tile element (r, c): row-major with the column XOR-swizzled by the row so that both
// lane-along-column and lane-along-row access patterns spread over the 16-byte bank slots.
SC_HD uint32_t lds_index(uint32_t r, uint32_t c, int logC) {
    uint32_t cm = (1u << logC) - 1u;
    return (r << logC) | ((c ^ r) & cm);
}

// A table entry that is read exactly once per transform (the direct four-step tables): with SC_TWD_NT the load carries the
// non-temporal hint, so that the table streams past the caches instead of displacing the vector (A/B: profiles/r03).
#ifndef SC_TWD_NT
#define SC_TWD_NT 0
#endif
SC_HD Fe load_stream(const Fe* p) {
#if defined(__HIP_DEVICE_COMPILE__) && SC_TWD_NT
    typedef unsigned long long sc_u64x2 __attribute__((ext_vector_type(2)));
    const sc_u64x2 v = __builtin_nontemporal_load(reinterpret_cast<const sc_u64x2*>(p));
    return Fe{v.x, v.y};
#else
    return *p;
#endif
}

// two-level power table lookup: base^e = lo[e & 4095] * hi[e >> 12]   (Montgomery form in, Montgomery form out)
SC_HD Fe pow2level(const Fe* lo, const Fe* hi, uint64_t e) {
    Fe a = lo[e & 4095u];
    Fe b = hi[e >> 12];
    // a~ * b~ * R^-1 = (ab)~
    return mont_mul(a, b);
}

// table entry i of a power table: base^(i*step) * scale   (all Montgomery form)
SC_HD Fe pow_table_entry(Fe base_m, uint64_t i, uint64_t step, Fe scale_m) {
    return mont_mul(mont_pow(base_m, i * step), scale_m);   // (x~ * s~) R^-1 = (x s)~
}

// One round of one workgroup's tile, for thread `tid`: S radix-2 DIF stages on row bits [sh, sh+S), in three steps the
// kernel can place separately (the geometry-specialised kernel issues the first round's global loads before it stages the
// tile twiddles, so both latencies overlap): round_gather -> round_butterflies -> round_scatter.
// GLR / GLC >= 0 fix the tile geometry at compile time (the hot shapes get their own kernel instantiation: all the
// index math below then folds into immediates); -1 = read it from P.
template <int LOGE, int S, int GLR = -1, int GLC = -1>
struct Round {
    static constexpr int E = 1 << LOGE;
    static constexpr int F = 1 << S;          // elements per butterfly group
    static constexpr int G = E >> S;          // groups per thread
    uint32_t rr[G];   // per group: row bits outside the field, packed (rrem)
    uint32_t cc[G];   // per group: column
    uint32_t t_lo, t_mid, t_hi;
    uint64_t col_off, col_off_in;  // first element of this workgroup's column in `out` / `in` (0 unless the launch covers several transforms)
    int logR, logC;

    SC_HD void setup(const PassParams& P, bool first, uint32_t tile, uint32_t tid) {
        logR = (GLR >= 0) ? GLR : P.logR;
        logC = (GLC >= 0) ? GLC : P.logC;
        const uint32_t T = 1u << (logR + logC - LOGE);   // threads per workgroup
        col_off = col_off_in = 0;
        if (P.col_enable) {
            col_off = (uint64_t)(tile >> P.col_tiles_log) * P.col_stride;
            col_off_in = (uint64_t)(tile >> P.col_tiles_log) * P.col_stride_in;
            tile &= (1u << P.col_tiles_log) - 1u;
        }
        t_lo = tile & ((1u << P.lo_log) - 1u);
        t_mid = (tile >> P.lo_log) & ((1u << P.mid_log) - 1u);
        t_hi = tile >> (P.lo_log + P.mid_log);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            uint32_t rem = (uint32_t)g * T + tid;      // < 2^(logR+logC-S)
            uint32_t rrem, c;
            if (first && P.rfast_load) {
                rrem = rem & ((1u << (logR - S)) - 1u);
                c = rem >> (logR - S);
            } else {
                c = rem & ((1u << logC) - 1u);
                rrem = rem >> logC;
            }
            rr[g] = rrem;
            cc[g] = c;
        }
    }
    SC_HD uint32_t row(int i, int sh) const {
        const int g = i >> S, fi = i & (F - 1);
        const uint32_t rrem = rr[g];
        return ((rrem >> sh) << (sh + S)) | ((uint32_t)fi << sh) | (rrem & ((1u << sh) - 1u));
    }
    // memory index of element i of the first round
    SC_HD uint64_t in_index(const PassParams& P, int i, int sh) const {
        const uint32_t r = row(i, sh), c = cc[i >> S];
        uint64_t j = (uint64_t)t_hi * P.in_hi + (uint64_t)t_mid * P.in_mid + (uint64_t)t_lo * P.in_lo + (uint64_t)c * P.in_cs;
        if (P.in_split) j += (uint64_t)(r & ((1u << P.in_split) - 1u)) * P.in_rs + (uint64_t)(r >> P.in_split) * P.in_rs_hi;
        else j += (uint64_t)r * P.in_rs;
        return j;
    }
    // first round: issue the global loads (data, and the previous pass's four-step twiddles when they are applied on load)
    SC_HD void gather_global(const PassParams& P, int sh, Fe* x, Fe* tin) const {
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const uint64_t j = in_index(P, i, sh);
            Fe v = fe_zero();
            if (j < P.in_limit) v = P.in[col_off_in + j];
            x[i] = v;
            if (P.twd_in) tin[i] = P.twd_in[j & P.twd_in_mask];
        }
    }
    // first round: the multiplications that belong to the load (coset scaling, twiddle-on-load)
    SC_HD void finish_global(const PassParams& P, int sh, Fe* x, const Fe* tin) const {
        if (P.coset_enable) {
#pragma unroll
            for (int i = 0; i < E; ++i) {
                const uint64_t j = in_index(P, i, sh);
                if (j < P.in_limit) x[i] = mont_mul(x[i], pow2level(P.ol, P.oh, j));
            }
        }
        if (P.twd_in) {
#pragma unroll
            for (int i = 0; i + 1 < E; i += 2) mont_mul2(x[i], tin[i], x[i + 1], tin[i + 1], x[i], x[i + 1]);
            if (E & 1) x[E - 1] = mont_mul(x[E - 1], tin[E - 1]);
        }
    }
    SC_HD void gather_lds(int sh, Fe* x, const Fe* lds) const {
#pragma unroll
        for (int i = 0; i < E; ++i) x[i] = lds[lds_index(row(i, sh), cc[i >> S], logC)];
    }
    // S radix-2 DIF stages on the field bits, highest bit first; tw[e << tw_shift] = w_R^e (LDS copy: shift 0).
    // prune_log > 0: stages whose index from the top (0-based) is below it are the degenerate ones of a zero-padded input.
    // last_hint: 1 / 0 when the caller knows whether this is the last round (sh == 0), -1 = look at sh.
    SC_HD void butterflies(int sh, Fe* x, const Fe* tw, int tw_shift, int prune_log = 0, int last_hint = -1, int prio = 0) const {
        const bool last = last_hint < 0 ? (sh == 0) : (last_hint != 0);
        set_prio(prio, 3);
#pragma unroll
        for (int q = 0; q < S; ++q) {
            const int bit = S - 1 - q;          // field bit
            const int b = sh + bit;             // row bit
            const int tau = logR - 1 - b;       // twiddle exponent scale: w_R^(2^tau * (r mod 2^b)); also the stage index from the top
            if (tau < prune_log) {
                // v = 0 everywhere: (u, 0) -> (u, u * w).  Before this stage row r is non-zero iff (r mod 2^(b+1)) < R >> prune_log.
                const uint32_t nz = 1u << (logR - prune_log);
#pragma unroll
                for (int i0 = 0; i0 < E; ++i0) {
                    if (i0 & (1 << bit)) continue;
                    const int i1 = i0 | (1 << bit);
                    const int g = i0 >> S;
                    const uint32_t fi_low = (uint32_t)(i0 & (F - 1)) & ((1u << bit) - 1u);
                    const Fe u = x[i0];
                    Fe d = fe_zero();
                    if ((row(i0, sh) & ((2u << b) - 1u)) < nz) {
                        if (last && (bit == 0 || fi_low == 0)) {
                            d = u;
                        } else {
                            const uint32_t row_lo = rr[g] & ((1u << sh) - 1u);
                            const uint32_t e = ((fi_low << sh) | row_lo) << tau;
                            d = mont_mul(u, tw[(uint64_t)e << tw_shift]);
                        }
                    }
                    x[i1] = d;
                }
                continue;
            }
            // the E/2 butterflies of this stage: sums and differences first, then the twiddle multiplications two at a time
            // (mont_mul2: two independent products interleaved so that their carry hazards cover each other).  All register
            // indices stay compile-time constants: butterfly number bf <-> (i0, i1) is a fixed enumeration.
            Fe dd[E / 2];
            uint32_t ee[E / 2];
            bool triv[E / 2];
#pragma unroll
            for (int bf = 0; bf < E / 2; bf += 2) {
                const int i0 = ((bf >> bit) << (bit + 1)) | (bf & ((1 << bit) - 1));     // bf with a 0 inserted at position `bit`
                const int i1 = i0 | (1 << bit);
                if (bf + 1 < E / 2) {
                    const int j0 = (((bf + 1) >> bit) << (bit + 1)) | ((bf + 1) & ((1 << bit) - 1));
                    const int j1 = j0 | (1 << bit);
                    fe_addsub2(x[i0], x[i1], x[j0], x[j1], x[i0], dd[bf], x[j0], dd[bf + 1]);
                } else {
                    Fe u = x[i0], v = x[i1];
                    x[i0] = fe_add(u, v);
                    dd[bf] = fe_sub(u, v);
                }
            }
#pragma unroll
            for (int bf = 0; bf < E / 2; ++bf) {
                const int i0 = ((bf >> bit) << (bit + 1)) | (bf & ((1 << bit) - 1));
                const int g = i0 >> S;
                const uint32_t fi_low = (uint32_t)(i0 & (F - 1)) & ((1u << bit) - 1u);
                triv[bf] = last && (bit == 0 || fi_low == 0);       // twiddle is w^0 = 1 (row bit 0, or no low field bits in the last round)
                const uint32_t row_lo = rr[g] & ((1u << sh) - 1u);
                ee[bf] = ((fi_low << sh) | row_lo) << tau;            // < R/2
                if (q == 0 && bf == 0) set_prio(prio, 2);
                if (q == S - 1 && bf == 0) set_prio(prio, 1);
            }
#pragma unroll
            for (int bf = 0; bf < E / 2; bf += 2) {
                const int ia = (((bf >> bit) << (bit + 1)) | (bf & ((1 << bit) - 1))) | (1 << bit);
                if (bf + 1 < E / 2) {
                    const int ib = ((((bf + 1) >> bit) << (bit + 1)) | ((bf + 1) & ((1 << bit) - 1))) | (1 << bit);
                    if (!triv[bf] && !triv[bf + 1]) {
                        mont_mul2(dd[bf], tw[(uint64_t)ee[bf] << tw_shift], dd[bf + 1], tw[(uint64_t)ee[bf + 1] << tw_shift], x[ia], x[ib]);
                    } else {
                        x[ia] = triv[bf] ? dd[bf] : mont_mul(dd[bf], tw[(uint64_t)ee[bf] << tw_shift]);
                        x[ib] = triv[bf + 1] ? dd[bf + 1] : mont_mul(dd[bf + 1], tw[(uint64_t)ee[bf + 1] << tw_shift]);
                    }
                } else {
                    x[ia] = triv[bf] ? dd[bf] : mont_mul(dd[bf], tw[(uint64_t)ee[bf] << tw_shift]);
                }
            }
            if (q == S - 1) set_prio(prio, 0);
        }
    }
    SC_HD static void set_prio(int enable, int level) {
#if defined(__HIP_DEVICE_COMPILE__)
        if (enable) {
            switch (level) {      // s_setprio takes an immediate
                case 3: __builtin_amdgcn_s_setprio(3); break;
                case 2: __builtin_amdgcn_s_setprio(2); break;
                case 1: __builtin_amdgcn_s_setprio(1); break;
                default: __builtin_amdgcn_s_setprio(0); break;
            }
        }
#endif
    }
    SC_HD void scatter_lds(int sh, const Fe* x, Fe* lds) const {
#pragma unroll
        for (int i = 0; i < E; ++i) lds[lds_index(row(i, sh), cc[i >> S], logC)] = x[i];
    }
    // last round (sh == 0): four-step twiddle (unless the next pass applies it on load), final scale, store
    // the direct four-step twiddles of this thread's E output elements (PassParams::twd): issued early by the fixed-shape kernel,
    // at the top of the last round, so that their latency hides behind that round's butterflies
    SC_HD void load_twd(const PassParams& P, Fe* t) const {
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const uint32_t k = bitrev32(row(i, 0), logR);
            const uint64_t colidx = ((((uint64_t)t_lo << logC) | cc[i >> S]) >> P.tw_col_shift) + P.tw_col_base;
            // natural output row of element i: k for a column pass of the plain plans (tw_row_k = 1, tw_row_mid = 0), and
            // t_mid + N_1 * k for the last pass of a batched two-pass column transform with a fused outer twiddle table
            t[i] = load_stream(&P.twd[((uint64_t)k * P.tw_row_k + (uint64_t)t_mid * P.tw_row_mid) * P.twd_stride + colidx]);
        }
    }
    SC_HD void scatter_global(const PassParams& P, const Fe* x) const {
        Fe none[E];
        if (P.blk_enable) scatter_global<true>(P, x, none, false);
        else scatter_global<false>(P, x, none, false);
    }
    // ALT: the launch has a destination table (PassParams::out_blk).  A compile-time switch: with the test at run time (inside
    // the store loop, or two loops behind one branch) the plain transforms paid 1-2 % more VALU instructions for index math
    // the compiler hoisted above the branch; the geometry-specialised kernels are instantiated for both values.
    template <bool ALT>
    SC_HD void scatter_global(const PassParams& P, const Fe* x, const Fe (&twd_prefetched)[E], bool have_prefetched) const {
        Fe v[E];
#pragma unroll
        for (int i = 0; i < E; ++i) v[i] = x[i];
        if (P.tw_enable) {
            // the four-step twiddles of this thread's E elements first (table loads, or two-level lookups multiplied out in
            // pairs), then the E products in pairs
            Fe t[E];
            if (P.twd) {
                if (have_prefetched) {
#pragma unroll
                    for (int i = 0; i < E; ++i) t[i] = twd_prefetched[i];
                } else {
                    load_twd(P, t);
                }
            } else {
                Fe a[E], b[E];
#pragma unroll
                for (int i = 0; i < E; ++i) {
                    const uint32_t k = bitrev32(row(i, 0), logR);
                    const uint64_t colidx = ((((uint64_t)t_lo << logC) | cc[i >> S]) >> P.tw_col_shift) + P.tw_col_base;
                    const uint64_t e = colidx * ((uint64_t)k * P.tw_row_k + (uint64_t)t_mid * P.tw_row_mid) * P.tw_scale;
                    a[i] = P.tl[e & 4095u];
                    b[i] = P.th[e >> 12];
                }
#pragma unroll
                for (int i = 0; i + 1 < E; i += 2) mont_mul2(a[i], b[i], a[i + 1], b[i + 1], t[i], t[i + 1]);
                if (E & 1) t[E - 1] = mont_mul(a[E - 1], b[E - 1]);
            }
#pragma unroll
            for (int i = 0; i + 1 < E; i += 2) mont_mul2(v[i], t[i], v[i + 1], t[i + 1], v[i], v[i + 1]);
            if (E & 1) v[E - 1] = mont_mul(v[E - 1], t[E - 1]);
        }
        if (P.scale_enable) {
#pragma unroll
            for (int i = 0; i < E; ++i) v[i] = mont_mul(v[i], P.scale);
        }
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const uint32_t k = bitrev32(row(i, 0), logR), c = cc[i >> S];
            uint64_t j = (uint64_t)t_hi * P.out_hi + (uint64_t)t_mid * P.out_mid + (uint64_t)t_lo * P.out_lo + (uint64_t)k * P.out_rs + (uint64_t)c * P.out_cs;
            if constexpr (ALT) {
                // the block of the natural row picks the destination (same element index everywhere)
                const uint32_t nat = k * P.blk_row_k + t_mid * P.blk_row_mid;
                Fe* dst = P.out_blk[nat >> P.blk_log];
                dst[col_off + j] = v[i];
            } else {
                P.out[col_off + j] = v[i];
            }
        }
    }
};

// the three steps in order (generic kernel and the CPU emulation)
template <int LOGE, int S, int GLR = -1, int GLC = -1>
SC_HD void ntt_round(const PassParams& P, int sh, bool first, uint32_t tile, uint32_t tid, Fe* lds, const Fe* tw) {
    constexpr int E = 1 << LOGE;
    Round<LOGE, S, GLR, GLC> R;
    R.setup(P, first, tile, tid);
    Fe x[E];
    if (first) {
        Fe tin[E];
        R.gather_global(P, sh, x, tin);
        R.finish_global(P, sh, x, tin);
    } else {
        R.gather_lds(sh, x, lds);
    }
    R.butterflies(sh, x, tw, 0, P.prune_log, -1, P.prio_balance);
    if (sh == 0) R.scatter_global(P, x);
    else R.scatter_lds(sh, x, lds);
}

// dispatch on the (runtime) number of stages in this round
template <int LOGE, int GLR = -1, int GLC = -1>
SC_HD void ntt_round_dispatch(const PassParams& P, int s, int sh, bool first, uint32_t tile, uint32_t tid, Fe* lds, const Fe* tw) {
    if constexpr (LOGE >= 4) { if (s == 4) { ntt_round<LOGE, 4, GLR, GLC>(P, sh, first, tile, tid, lds, tw); return; } }
    if constexpr (LOGE >= 3) { if (s == 3) { ntt_round<LOGE, 3, GLR, GLC>(P, sh, first, tile, tid, lds, tw); return; } }
    if constexpr (LOGE >= 2) { if (s == 2) { ntt_round<LOGE, 2, GLR, GLC>(P, sh, first, tile, tid, lds, tw); return; } }
    ntt_round<LOGE, 1, GLR, GLC>(P, sh, first, tile, tid, lds, tw);
}

// The twiddles of the tile transform (w_R^i, i < R/2: 16 bytes x R/2, 8 KiB for R = 2^10) are read by every butterfly of
// every round; staged once per workgroup in LDS behind the tile they cost a ds_read instead of a global load with 64-bit
// address arithmetic (measured: 3-4 % on whole transforms at 2^22-2^24, profiles/r01/ab_lds_twiddles.txt).
// Thread `tid` of `nthreads` copies its share; the caller puts a barrier between this and the first round.
SC_HD void tile_twiddles_to_lds(const PassParams& P, int logR, uint32_t tid, uint32_t nthreads, Fe* tw) {
    const uint32_t count = logR > 0 ? (1u << (logR - 1)) : 0u;
    for (uint32_t i = tid; i < count; i += nthreads) tw[i] = P.mt[(uint64_t)i << P.mt_shift];
}

// Fully unrolled round schedule for a compile-time geometry (short round first, like make_rounds()): every `sh` is a
// constant, so row/LDS indices become base + immediate.  The CPU emulation calls ntt_round() per round with the same
// S / SH constants; the kernel uses run(), which differs only in WHEN things are issued and in how rounds are fenced:
//
//   * the first round's global loads (data, twiddle-on-load table) and the tile-twiddle staging loads are all issued
//     before the first barrier, so the workgroup pays one memory latency, not three in a row;
//   * threads that exchange elements between the round at shift SH and the next one differ only in the thread-id bits
//     [SH - LOGE + GLC, SH + GLC) (derivation in DESIGN.md 3.1): once SH + GLC <= 6 every later exchange stays inside
//     one 64-lane wave, each wave owns a closed set of LDS rows, and the workgroup barrier is replaced by a wave-level
//     fence -- the waves of a workgroup then drift apart and overlap each other's LDS traffic, arithmetic and stores.
template <int LOGE, int GLR, int GLC, int ROUND = 0, bool ALT = false>
struct FixedRounds {
    static constexpr int E = 1 << LOGE;
    static constexpr int NR = (GLR + LOGE - 1) / LOGE;
    static constexpr int S = (ROUND == 0) ? (GLR - LOGE * (NR - 1)) : LOGE;
    static constexpr int DONE = (ROUND == 0) ? 0 : (GLR - LOGE * (NR - 1)) + LOGE * (ROUND - 1);
    static constexpr int SH = GLR - DONE - S;
    static constexpr bool NEXT_WAVE_LOCAL = (ROUND >= 1) && (SH + GLC <= 6);
    static constexpr uint32_t TW_COUNT = 1u << (GLR - 1);     // tile twiddles; <= threads per workgroup for GLC >= 1

    // sync(): workgroup barrier; wsync(): wave-level fence; stamp(i): diagnostics hook (no-op in production)
    template <class Sync, class WSync, class Stamp>
    SC_HD static void run(const PassParams& P, uint32_t tile, uint32_t tid, Fe* lds, Fe* tw, Sync sync, WSync wsync, Stamp stamp, bool wave_local) {
        Round<LOGE, S, GLR, GLC> R;
        R.setup(P, ROUND == 0, tile, tid);
        Fe x[E];
        if constexpr (ROUND == 0) {
            Fe tin[E];
            Fe twv = fe_zero();
            if (P.prio_balance == 2) Round<LOGE, S, GLR, GLC>::set_prio(1, 3);
            if (tid < TW_COUNT) twv = P.mt[(uint64_t)tid << P.mt_shift];
            R.gather_global(P, SH, x, tin);
            if (P.prio_balance == 2) Round<LOGE, S, GLR, GLC>::set_prio(1, 0);
            if (tid < TW_COUNT) tw[tid] = twv;
            stamp(1);
            sync();
            stamp(2);
            R.finish_global(P, SH, x, tin);
        } else {
            R.gather_lds(SH, x, lds);
        }
        Fe tpre[E];
        bool prefetched = false;
        if constexpr (ROUND + 1 == NR && NR > 1) {
            if (P.tw_enable && P.twd) {
                R.load_twd(P, tpre);            // in flight during this round's arithmetic
                prefetched = true;
            }
        }
        // the degenerate stages of a zero-padded input lie in the first two rounds (the planner caps prune_log accordingly)
        const int prune = (ROUND <= 1) ? P.prune_log : 0;
        R.butterflies(SH, x, tw, 0, prune, ROUND + 1 == NR ? 1 : 0, P.prio_balance == 1 ? 1 : 0);
        stamp(3 + 2 * ROUND);
        if constexpr (ROUND + 1 < NR) {
            R.scatter_lds(SH, x, lds);
            if (NEXT_WAVE_LOCAL && wave_local) wsync(); else sync();
            stamp(4 + 2 * ROUND);
            FixedRounds<LOGE, GLR, GLC, ROUND + 1, ALT>::run(P, tile, tid, lds, tw, sync, wsync, stamp, wave_local);
        } else {
            R.template scatter_global<ALT>(P, x, tpre, prefetched);
            stamp(4 + 2 * ROUND);
        }
    }
};

// Round schedule shared by the kernel and the CPU emulation: the short round (if any) goes first.
struct RoundSched {
    int nrounds;
    int s[16];
    int sh[16];
};
SC_HD RoundSched make_rounds(int logR, int loge) {
    RoundSched rs;
    rs.nrounds = (logR + loge - 1) / loge;
    if (rs.nrounds == 0) rs.nrounds = 1;
    int rem = logR;
    for (int i = 0; i < rs.nrounds; ++i) {
        int s = (i == 0) ? (logR - loge * (rs.nrounds - 1)) : loge;
        if (s < 1) s = (logR == 0) ? 0 : 1;
        rem -= s;
        rs.s[i] = s;
        rs.sh[i] = rem;
    }
    return rs;
}

}  // namespace sc
