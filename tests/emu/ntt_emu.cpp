// CPU emulation of the tiled NTT kernels: runs the SAME round body (csrc/ntt_tile.cuh) and the SAME
// planner (csrc/ntt_plan.h) the HIP library uses, one workgroup at a time, one thread id at a time,
// rounds separated exactly where the kernel has its barriers.  Lets the index/twiddle logic be checked
// against the oracle in the CPU-only container.  Test infrastructure (built by tests/test_emu.py).
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../stark-anatomy_amd/csrc/ntt_plan.h"

using namespace sc;

static void fill_table(std::vector<Fe>& t, uint64_t count, Fe base_m, uint64_t step, Fe scale_m) {
    t.resize(count ? count : 1);
    for (uint64_t i = 0; i < count; ++i) t[i] = pow_table_entry(base_m, i, step, scale_m);
}

// geometry-specialised path (FixedRounds): on the CPU each round must finish for all threads before the next starts,
// so the per-round bodies are invoked directly with the same compile-time schedule the kernel unrolls
template <int LOGE, int GLR, int GLC, int ROUND = 0>
static void run_fixed_rounds(const NttPassDesc& pd, uint32_t tile, Fe* lds, const Fe* tw) {
    using FR = FixedRounds<LOGE, GLR, GLC, ROUND>;
    for (uint32_t tid = 0; tid < pd.threads; ++tid) ntt_round<LOGE, FR::S, GLR, GLC>(pd.p, FR::SH, ROUND == 0, tile, tid, lds, tw);
    if constexpr (ROUND + 1 < FR::NR) run_fixed_rounds<LOGE, GLR, GLC, ROUND + 1>(pd, tile, lds, tw);
}

template <int LOGE>
static void run_pass(const NttPassDesc& pd) {
    const PassParams& P = pd.p;
    // LDS exactly as the kernel lays it out: the tile, then the staged twiddles (pd.lds_bytes covers both)
    std::vector<Fe> lds(pd.lds_bytes / sizeof(Fe));
    if (lds.size() < ((size_t)1 << (P.logR + P.logC)) + (P.logR > 0 ? ((size_t)1 << (P.logR - 1)) : 0)) abort();
    Fe* tw = lds.data() + ((size_t)1 << (P.logR + P.logC));
    for (uint32_t tid = 0; tid < pd.threads; ++tid) tile_twiddles_to_lds(P, P.logR, tid, pd.threads, tw);
    if constexpr (LOGE == 2) {
#define EMU_FIXED(LR, LC)                                                                                   \
        if (P.logR == LR && P.logC == LC) {                                                                 \
            for (uint32_t tile = 0; tile < pd.ntiles * pd.cols; ++tile) run_fixed_rounds<2, LR, LC>(pd, tile, lds.data(), tw); \
            return;                                                                                         \
        }
        EMU_FIXED(8, 3) EMU_FIXED(7, 4) EMU_FIXED(10, 2) EMU_FIXED(6, 5) EMU_FIXED(9, 3) EMU_FIXED(8, 4)
#undef EMU_FIXED
    }
    if constexpr (LOGE == 3) {          // the batches' eight-elements-per-thread kernels (ntt_pass_kernel_fixed8)
#define EMU_FIXED8(LR, LC)                                                                                  \
        if (P.logR == LR && P.logC == LC) {                                                                 \
            for (uint32_t tile = 0; tile < pd.ntiles * pd.cols; ++tile) run_fixed_rounds<3, LR, LC>(pd, tile, lds.data(), tw); \
            return;                                                                                         \
        }
        EMU_FIXED8(10, 2) EMU_FIXED8(9, 3) EMU_FIXED8(8, 4)
#undef EMU_FIXED8
    }
    RoundSched rs = make_rounds(P.logR, LOGE);
    for (uint32_t tile = 0; tile < pd.ntiles * pd.cols; ++tile)
        for (int r = 0; r < rs.nrounds; ++r)
            for (uint32_t tid = 0; tid < pd.threads; ++tid)
                ntt_round_dispatch<LOGE>(P, rs.s[r], rs.sh[r], r == 0, tile, tid, lds.data(), tw);
}

extern "C" int emu_ntt(const uint64_t* in, uint64_t* out, int logn, const uint64_t* root, int inverse, uint64_t in_limit,
                       const uint64_t* offset, int max_tile_log, int loge, int single_pass_max_log, int min_tiles_log, int max_col_log, int max_digit_log, int direct_tw) {
    const uint64_t n = 1ull << logn;
    Fe r = Fe{root[0], root[1]};
    Fe r_m = to_mont(r);
    Fe scale_m = fe_mont_one();
    if (inverse) {
        r_m = mont_pow(r_m, n - 1);                       // root^-1 = root^(n-1)
        scale_m = mont_inv(to_mont(Fe{n, 0}));            // n^-1
    }
    NttTuning tu;
    tu.max_tile_log = max_tile_log; tu.loge = loge; tu.single_pass_max_log = single_pass_max_log;
    tu.min_tiles_log = min_tiles_log; tu.max_col_log = max_col_log; tu.max_digit_log = max_digit_log;
    tu.tw_on_load = (direct_tw != 2) ? 1 : 0;      // direct_tw: 0 = two-level lookup, 1 = direct tables applied on load by the next pass, 2 = direct tables at the store
    const int m = plan_num_passes(logn, tu);
    NttTables tb;
    std::vector<Fe> mt, tl, th, ths, ol, oh;
    tb.mt_log = logn < 12 ? logn : 12;
    fill_table(mt, 1ull << (tb.mt_log - 1), r_m, n >> tb.mt_log, fe_mont_one());
    fill_table(tl, n < 4096 ? n : 4096, r_m, 1, fe_mont_one());
    fill_table(th, n > 4096 ? n >> 12 : 1, r_m, 4096, fe_mont_one());
    fill_table(ths, n > 4096 ? n >> 12 : 1, r_m, 4096, scale_m);
    tb.mt = mt.data(); tb.tl = tl.data(); tb.th = th.data(); tb.th_scaled = (inverse && m > 1) ? ths.data() : nullptr;
    std::vector<Fe> work(n);
    NttIo io;
    io.in = (const Fe*)in; io.work = work.data(); io.out = (Fe*)out;
    io.in_limit = in_limit;
    if (offset) {
        Fe o_m = to_mont(Fe{offset[0], offset[1]});
        uint64_t cnt = in_limit < n ? in_limit : n;
        fill_table(ol, 4096, o_m, 1, fe_mont_one());
        fill_table(oh, (cnt >> 12) + 1, o_m, 4096, fe_mont_one());
        io.ol = ol.data(); io.oh = oh.data();
    }
    io.scale_last = inverse && m == 1;
    io.scale = scale_m;
    NttPlanDesc d;
    if (!plan_ntt(d, logn, tb, io, tu)) return -1;
    std::vector<Fe> twd[4];
    if (direct_tw && d.npasses > 1) {
        int logA = 0;
        for (int i = 0; i + 1 < d.npasses; ++i) {
            const int logR = d.digits[i], logB = logn - logA - logR;
            const uint64_t count = 1ull << (logR + logB);
            twd[i].resize(count);
            const Fe* thp = (i == 0 && tb.th_scaled) ? tb.th_scaled : tb.th;
            for (uint64_t q = 0; q < count; ++q) {
                uint64_t k = q >> logB, b = q & ((1ull << logB) - 1);
                twd[i][q] = pow2level(tb.tl, thp, b * k * (1ull << logA));
            }
            tb.twd[i] = twd[i].data();
            logA += logR;
        }
        if (!plan_ntt(d, logn, tb, io, tu)) return -1;
    }
    for (int i = 0; i < d.npasses; ++i) {
        switch (d.pass[i].loge) {
            case 1: run_pass<1>(d.pass[i]); break;
            case 2: run_pass<2>(d.pass[i]); break;
            case 3: run_pass<3>(d.pass[i]); break;
            case 4: run_pass<4>(d.pass[i]); break;
            default: return -2;
        }
    }
    return d.npasses;
}

// `cols` independent transforms in one set of launches (NttIo::cols, sc_ntt_columns_dev): in / out are [cols][n]
static int emu_columns_impl(const uint64_t* in, uint64_t* out, int logn, int cols, const uint64_t* root, int inverse, int direct_tw, uint64_t m, const uint64_t* offset);
extern "C" int emu_ntt_columns(const uint64_t* in, uint64_t* out, int logn, int cols, const uint64_t* root, int inverse, int direct_tw) {
    return emu_columns_impl(in, out, logn, cols, root, inverse, direct_tw, 0, nullptr);
}
// the batched LDE (sc_coset_evaluate_columns_dev): `cols` polynomials of m coefficients each (in: [cols][m]) -> [cols][n] values on offset * <root>
extern "C" int emu_coset_evaluate_columns(const uint64_t* in, uint64_t* out, int logn, int cols, const uint64_t* root, uint64_t m, const uint64_t* offset) {
    return emu_columns_impl(in, out, logn, cols, root, 0, 2, m, offset);
}
static int emu_columns_impl(const uint64_t* in, uint64_t* out, int logn, int cols, const uint64_t* root, int inverse, int direct_tw, uint64_t m_coeffs, const uint64_t* offset) {
    const uint64_t n = 1ull << logn;
    Fe r_m = to_mont(Fe{root[0], root[1]});
    Fe scale_m = fe_mont_one();
    if (inverse) {
        r_m = mont_pow(r_m, n - 1);
        scale_m = mont_inv(to_mont(Fe{n, 0}));
    }
    NttTuning tu;
    const int m = plan_num_passes(logn, tu);
    NttTables tb;
    std::vector<Fe> mt, tl, th, ths;
    tb.mt_log = logn < 12 ? logn : 12;
    fill_table(mt, 1ull << (tb.mt_log - 1), r_m, n >> tb.mt_log, fe_mont_one());
    fill_table(tl, n < 4096 ? n : 4096, r_m, 1, fe_mont_one());
    fill_table(th, n > 4096 ? n >> 12 : 1, r_m, 4096, fe_mont_one());
    fill_table(ths, n > 4096 ? n >> 12 : 1, r_m, 4096, scale_m);
    tb.mt = mt.data(); tb.tl = tl.data(); tb.th = th.data(); tb.th_scaled = (inverse && m > 1) ? ths.data() : nullptr;
    std::vector<Fe> work(n * cols);
    NttIo io;
    io.in = (const Fe*)in; io.work = work.data(); io.out = (Fe*)out;
    io.cols = (uint32_t)cols;
    std::vector<Fe> ol, oh;
    if (offset) {
        Fe o_m = to_mont(Fe{offset[0], offset[1]});
        fill_table(ol, 4096, o_m, 1, fe_mont_one());
        fill_table(oh, (m_coeffs >> 12) + 1, o_m, 4096, fe_mont_one());
        io.ol = ol.data(); io.oh = oh.data();
        io.in_limit = m_coeffs;
        io.col_stride_in = m_coeffs;
    }
    io.scale_last = inverse && m == 1;
    io.scale = scale_m;
    NttPlanDesc d;
    if (!plan_ntt(d, logn, tb, io, tu)) return -1;
    std::vector<Fe> twd[4];
    if (direct_tw && d.npasses > 1) {
        int logA = 0;
        for (int i = 0; i + 1 < d.npasses; ++i) {
            const int logR = d.digits[i], logB = logn - logA - logR;
            const uint64_t count = 1ull << (logR + logB);
            twd[i].resize(count);
            const Fe* thp = (i == 0 && tb.th_scaled) ? tb.th_scaled : tb.th;
            for (uint64_t q = 0; q < count; ++q) {
                uint64_t k = q >> logB, b = q & ((1ull << logB) - 1);
                twd[i][q] = pow2level(tb.tl, thp, b * k * (1ull << logA));
            }
            tb.twd[i] = twd[i].data();
            logA += logR;
        }
        if (!plan_ntt(d, logn, tb, io, tu)) return -1;
    }
    for (int i = 0; i < d.npasses; ++i) {
        switch (d.pass[i].loge) {
            case 1: run_pass<1>(d.pass[i]); break;
            case 2: run_pass<2>(d.pass[i]); break;
            case 3: run_pass<3>(d.pass[i]); break;
            case 4: run_pass<4>(d.pass[i]); break;
            default: return -2;
        }
    }
    return d.npasses;
}

extern "C" void emu_field(const uint64_t* a, const uint64_t* b, uint64_t* out /* 7 x 2 limbs */) {
    Fe x{a[0], a[1]}, y{b[0], b[1]};
    Fe r[7] = {fe_add(x, y), fe_sub(x, y), fe_mul(x, y), from_mont(mont_inv(to_mont(x))), fe_half(x), fe_neg(x), mont_mul(x, to_mont(y))};
    memcpy(out, r, sizeof r);
}

// batched transforms (multi-GPU building blocks): kind 0 = columns of [len][batch], kind 1 = rows of [batch][len] -> [len][batch]
// extras of the four-step plan object (sc_fourstep_*): a second destination for a range of natural output rows (kind 0), an
// explicit chunk stride (kind 1), a caller-owned work buffer and a pass range (row blocks with a deferred second pass)
struct EmuExtras {
    uint64_t* diag_out = nullptr;
    uint32_t diag_lo = 0, diag_n = 0;
    uint64_t chunk_stride = 0;
    uint64_t* work = nullptr;
    int pass_lo = 0, pass_hi = 4;
};

static int emu_batched_impl(const uint64_t* in, uint64_t* out, int kind, int loglen, int logbatch, const uint64_t* root,
                               int max_tile_log, int loge, int min_tiles_log, int max_col_log, int max_digit_log,
                               const uint64_t* outer_root, int outer_logorder, uint64_t outer_col_base, int outer_ninv, int chunks_log, int inner_direct, uint64_t out_ld,
                               const EmuExtras& xx = EmuExtras()) {
    const uint64_t len = 1ull << loglen, batch = 1ull << logbatch;
    Fe r_m = to_mont(Fe{root[0], root[1]});
    NttTuning tu;
    tu.max_tile_log = max_tile_log; tu.loge = loge; tu.min_tiles_log = min_tiles_log; tu.max_col_log = max_col_log; tu.max_digit_log = max_digit_log;
    NttTables tb;
    std::vector<Fe> mt, tl, th;
    tb.mt_log = loglen < 12 ? loglen : 12;
    fill_table(mt, 1ull << (tb.mt_log - 1), r_m, len >> tb.mt_log, fe_mont_one());
    fill_table(tl, len < 4096 ? len : 4096, r_m, 1, fe_mont_one());
    fill_table(th, len > 4096 ? len >> 12 : 1, r_m, 4096, fe_mont_one());
    tb.mt = mt.data(); tb.tl = tl.data(); tb.th = th.data();
    std::vector<Fe> work_own(xx.work ? 0 : len * batch);
    Fe* work_p = xx.work ? (Fe*)xx.work : work_own.data();
    BatchExtras ex;
    ex.chunks_log = chunks_log;
    ex.out_ld = out_ld;
    ex.chunk_stride = xx.chunk_stride;
    ex.diag_out = (Fe*)xx.diag_out;
    ex.diag_lo = xx.diag_lo;
    ex.diag_n = xx.diag_n;
    std::vector<Fe> otl, oth, otw;
    if (outer_root) {
        Fe o_m = to_mont(Fe{outer_root[0], outer_root[1]});
        const uint64_t on = 1ull << outer_logorder;
        Fe sc_m = outer_ninv ? mont_inv(to_mont(Fe{on, 0})) : fe_mont_one();
        fill_table(otl, on < 4096 ? on : 4096, o_m, 1, fe_mont_one());
        fill_table(oth, on > 4096 ? on >> 12 : 1, o_m, 4096, sc_m);
        ex.outer_tl = otl.data(); ex.outer_th = oth.data(); ex.outer_col_base = outer_col_base;
        if (inner_direct & 2) {
            // the library's direct outer table (outer_table_kernel): [r][c] = w^(r * (col_base + c)) [* order^-1]
            otw.resize(len * batch);
            for (uint64_t i = 0; i < len * batch; ++i) otw[i] = pow2level(ex.outer_tl, ex.outer_th, (i >> logbatch) * (outer_col_base + (i & (batch - 1))));
            ex.outer_twd = otw.data();
        }
    }
    NttPlanDesc d;
    if (!plan_batched(d, kind == 0 ? BATCH_COLS : BATCH_ROWS_T, loglen, logbatch, tb, (const Fe*)in, work_p, (Fe*)out, tu, ex)) return -1;
    std::vector<Fe> itw;
    if ((inner_direct & 1) && d.npasses == 2) {
        // the library's direct inter-pass table (twiddle_table_kernel): [k][b] = w^(b*k), B = len >> digits[0]
        const int logB = loglen - d.digits[0];
        itw.resize(len);
        for (uint64_t i = 0; i < len; ++i) itw[i] = pow2level(tb.tl, tb.th, (i & ((1ull << logB) - 1)) * (i >> logB));
        ex.inner_twd = itw.data();
        if (!plan_batched(d, kind == 0 ? BATCH_COLS : BATCH_ROWS_T, loglen, logbatch, tb, (const Fe*)in, work_p, (Fe*)out, tu, ex)) return -1;
    }
    for (int i = 0; i < d.npasses; ++i) {
        if (i < xx.pass_lo || i >= xx.pass_hi) continue;
        switch (d.pass[i].loge) {
            case 1: run_pass<1>(d.pass[i]); break;
            case 2: run_pass<2>(d.pass[i]); break;
            case 3: run_pass<3>(d.pass[i]); break;
            case 4: run_pass<4>(d.pass[i]); break;
            default: return -2;
        }
    }
    return d.npasses;
}

// the stages of sc_fourstep_* on one rank's buffers (csrc/fourstep.hip: fourstep_cols / fourstep_rows / fourstep_rows_finish)
extern "C" int emu_fourstep_cols(const uint64_t* src, uint64_t* send, uint64_t* recv_diag, int logR, int logcw, const uint64_t* root_cols,
                                 const uint64_t* outer_root, int outer_logorder, uint64_t col_base, int ninv, uint32_t diag_lo, uint32_t diag_n,
                                 int max_tile_log, int loge, int min_tiles_log, int max_col_log, int max_digit_log) {
    EmuExtras xx;
    xx.diag_out = recv_diag; xx.diag_lo = diag_lo; xx.diag_n = recv_diag ? diag_n : 0;
    return emu_batched_impl(src, send, 0, logR, logcw, root_cols, max_tile_log, loge, min_tiles_log, max_col_log, max_digit_log,
                            outer_root, outer_logorder, col_base, ninv, 0, 3, 0, xx);
}
// rows [row0, row0 + 2^logrk) of the rank, read in place from recv [2^chunks_log][rw][C >> chunks_log]; work: rank-sized [rw][C]
extern "C" int emu_fourstep_rows(const uint64_t* recv, uint64_t* dst, uint64_t* work, int logC, int logrk, uint64_t row0, uint64_t rw, int chunks_log,
                                 const uint64_t* root_rows, int pass_lo, int pass_hi,
                                 int max_tile_log, int loge, int min_tiles_log, int max_col_log, int max_digit_log) {
    const uint64_t C = 1ull << logC, cw = C >> chunks_log;
    EmuExtras xx;
    xx.chunk_stride = rw * cw;
    xx.work = work + 2 * row0 * C;
    xx.pass_lo = pass_lo; xx.pass_hi = pass_hi;
    return emu_batched_impl(recv + 2 * row0 * cw, dst + 2 * row0, 1, logC, logrk, root_rows, max_tile_log, loge, min_tiles_log, max_col_log, max_digit_log,
                            nullptr, 0, 0, 0, chunks_log, 1, rw, xx);
}
extern "C" int emu_fourstep_rows_finish(uint64_t* dst, uint64_t* work, int logC, int logrw, const uint64_t* root_rows,
                                        int max_tile_log, int loge, int min_tiles_log, int max_col_log, int max_digit_log) {
    EmuExtras xx;
    xx.work = work;
    xx.pass_lo = 1; xx.pass_hi = 2;
    return emu_batched_impl(work + 2, dst, 1, logC, logrw, root_rows, max_tile_log, loge, min_tiles_log, max_col_log, max_digit_log,
                            nullptr, 0, 0, 0, 0, 1, 0, xx);
}

extern "C" int emu_ntt_batched(const uint64_t* in, uint64_t* out, int kind, int loglen, int logbatch, const uint64_t* root,
                               int max_tile_log, int loge, int min_tiles_log, int max_col_log, int max_digit_log,
                               const uint64_t* outer_root, int outer_logorder, uint64_t outer_col_base, int outer_ninv, int chunks_log, int inner_direct) {
    return emu_batched_impl(in, out, kind, loglen, logbatch, root, max_tile_log, loge, min_tiles_log, max_col_log, max_digit_log,
                            outer_root, outer_logorder, outer_col_base, outer_ninv, chunks_log, inner_direct, 0);
}

// one row block of the overlapped corner turn: kind 1, chunked input, `2^logbatch` adjacent columns of a [len][out_ld] output
extern "C" int emu_ntt_rows_ld(const uint64_t* in, uint64_t* out, int loglen, int logbatch, const uint64_t* root,
                               int max_tile_log, int loge, int min_tiles_log, int max_col_log, int max_digit_log, int chunks_log, int inner_direct, uint64_t out_ld) {
    return emu_batched_impl(in, out, 1, loglen, logbatch, root, max_tile_log, loge, min_tiles_log, max_col_log, max_digit_log,
                            nullptr, 0, 0, 0, chunks_log, inner_direct, out_ld);
}
