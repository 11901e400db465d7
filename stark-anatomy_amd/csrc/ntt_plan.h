// ntt_plan.h -- host-side pass planning for the tiled NTT (digits, tile shapes, strides).
// Plain C++ shared by the C-ABI library (core.hip, fourstep.hip) and the CPU emulation used in tests.
#pragma once
#include "ntt_tile.cuh"

namespace sc {

struct NttTuning {
    // Defaults are the measured optimum on MI355X (tools/ab.py, profiles/): E = 4 elements per thread; up to 2^20 two
    // passes of <= 2^10 points with 2^12-element tiles (4+ columns), above that three passes of <= 2^8 points with
    // 2^11-element tiles.  A value of -1 means "choose by size".
    int max_tile_log = -1;   // LDS tile = 2^max_tile_log elements
    int loge = 2;            // elements per thread = 2^loge (occupancy beats register blocking: each modmul is a long dependent chain)
    int max_col_log = -1;    // columns per tile (run length = 16 B << max_col_log)
    int min_tiles_log = 8;   // shrink tiles (never below 4 columns) until there are at least this many per pass
    int single_pass_max_log = 11;
    int max_digit_log = -1;  // passes = ceil(logn / max_digit_log)
    int direct_tw_max_log = 22;  // direct four-step twiddle tables up to 2^this entries per pass (bigger ones cost more HBM than they save)
    // 1: a pass whose predecessor has a direct table applies that table on LOAD (PassParams::twd_in) instead of the predecessor
    // applying it at its store.  Round-2 history (profiles/r02): on-load won +1-2 % from 2^22 up while the at-store variant waited
    // for its table between the last butterfly and the store; since the fixed-shape kernels PREFETCH the table at the top of their
    // last round, at-store wins everywhere (2^22 +3 %, LDE 2^18 -> 2^21 71 -> 63.5 us, 2^20 +8 %).  Kept as an option.
    int tw_on_load = 0;
    int prune = 1;               // skip the degenerate top stages of a zero-padded first pass (PassParams::prune_log)
    // elements per thread (log2) for the 2^12-element tiles -- (10,2), (9,3), (8,4) -- of a BATCH of columns (NttIo::cols > 1).  With four
    // elements a thread such a tile is a 1024-thread workgroup and exactly one fits a CU; with eight it is 512 threads, three stages per
    // round instead of two (four rounds instead of five for 2^10 points), and -- built for the registers of four waves per SIMD
    // (ntt_pass_kernel_fixed8: 128 VGPRs) -- TWO fit, so one computes while the other loads or drains: 2^20 x 64 columns +7 %, 2^18 x 64
    // +18 %.  A lone transform keeps four (its single workgroup per CU loses occupancy with eight: -9 %); the 2^11-element tiles too.
    int loge_cols = 3;
};

inline NttTuning resolve_tuning(const NttTuning& in, int logn) {
    NttTuning t = in;
    const bool small = logn <= 20;
    if (t.max_digit_log < 0) t.max_digit_log = small ? 10 : 8;
    if (t.max_tile_log < 0) t.max_tile_log = small ? 12 : 11;
    if (t.max_col_log < 0) t.max_col_log = small ? 4 : 6;
    if (t.tw_on_load < 0) t.tw_on_load = 0;
    return t;
}

struct NttTables {
    const Fe* mt = nullptr;  // mt[e] = w_(2^mt_log)^e, e < 2^(mt_log-1)      (Montgomery form)
    int mt_log = 0;
    const Fe* tl = nullptr;  // tl[e] = root^e, e < min(n, 4096)
    const Fe* th = nullptr;  // th[h] = root^(4096 h), h < max(1, n/4096)
    const Fe* th_scaled = nullptr;  // th[h] * scale (n^-1 for the inverse transform); used by the FIRST pass's twiddle only
    const Fe* twd[4] = {nullptr, nullptr, nullptr, nullptr};   // optional direct twiddle table per column pass (see PassParams::twd)
};

// LDS behind the tile for the tile transform's twiddles (w_R^i, i < R/2)
inline uint32_t tile_twiddle_bytes(int logR) { return logR > 0 ? (uint32_t)sizeof(Fe) << (logR - 1) : 0u; }

struct NttPassDesc {
    PassParams p;
    int loge;
    uint32_t ntiles;         // per column
    uint32_t cols = 1;       // the grid is cols * ntiles workgroups
    uint32_t threads;
    uint32_t lds_bytes;
};

struct NttPlanDesc {
    int logn = 0;
    int npasses = 0;
    int digits[4] = {0, 0, 0, 0};
    NttPassDesc pass[4];
};

struct NttIo {
    const Fe* in = nullptr;
    Fe* work = nullptr;      // n elements of scratch (unused when npasses == 1)
    Fe* out = nullptr;
    uint64_t in_limit = ~0ull;
    const Fe* ol = nullptr;  // coset scaling tables (nullptr = none)
    const Fe* oh = nullptr;
    bool scale_last = false; // multiply outputs by `scale` in the last pass (used when there is no four-step twiddle to fold it into)
    Fe scale = Fe{0, 0};
    // `cols` independent transforms of length n in ONE set of launches (sc_ntt_columns_dev): column c of in / work / out starts
    // at element c * n, every pass is launched over cols x its tiles, and workgroup b works on column b / tiles, tile b % tiles
    // (PassParams::col_enable).  Twiddles and tables are shared by the columns; zero padding and coset scaling apply to each.
    uint32_t cols = 1;
    uint64_t col_stride_in = 0;   // elements between the columns of `in` (0 = n; a batch of zero-padded inputs: the m coefficients of a column)
};

inline int plan_num_passes(int logn, const NttTuning& tu_in) {
    const NttTuning tu = resolve_tuning(tu_in, logn);
    if (logn <= tu.single_pass_max_log) return 1;
    const int m = (logn + tu.max_digit_log - 1) / tu.max_digit_log;
    return m < 2 ? 2 : m;
}

// Fill the pass descriptors for a length-2^logn transform.  Returns false if unsupported.
inline bool plan_ntt(NttPlanDesc& d, int logn, const NttTables& tb, const NttIo& io, const NttTuning& tu_in) {
    if (logn < 1 || logn > 32) return false;
    NttTuning tu = resolve_tuning(tu_in, logn);
    d.logn = logn;
    const int m = plan_num_passes(logn, tu);
    if (m > 4) return false;
    if (io.cols < 1 || io.cols > 65536) return false;
    // a batch of columns is one long grid whatever the length of a column: short columns then take the 2^11-element tiles of the
    // long transforms (three workgroups per CU; shapes (7,4), (6,5), (8,3), which have geometry-specialised kernels) instead of the
    // tiles a lone short transform shrinks to in order to cover the machine (2^14 x 256 columns: 27 -> 49 G elements/s)
    int cols_log = 0;
    while ((2u << cols_log) <= io.cols) ++cols_log;
    if (io.cols > 1 && logn <= 16) {
        if (tu_in.max_tile_log < 0) tu.max_tile_log = 11;
        if (tu_in.max_col_log < 0) tu.max_col_log = 6;
    }
    d.npasses = m;
    {
        int base = logn / m, extra = logn % m;
        for (int i = 0; i < m; ++i) d.digits[i] = base + (i < extra ? 1 : 0);
    }
    const uint64_t n = 1ull << logn;
    int tile_cap = tu.max_tile_log;
    {
        int maxdigit = 0;
        for (int i = 0; i < m; ++i) maxdigit = d.digits[i] > maxdigit ? d.digits[i] : maxdigit;
        const int floor_log = maxdigit + 2 < tu.max_tile_log ? maxdigit + 2 : tu.max_tile_log;    // keep >= 4 columns per tile
        while (tile_cap > floor_log && logn + cols_log - tile_cap < tu.min_tiles_log && m > 1) --tile_cap;
    }

    int logA = 0;                         // log2 of the product of the digits already transformed
    for (int i = 0; i < m; ++i) {
        NttPassDesc& pd = d.pass[i];
        PassParams& p = pd.p;
        p = PassParams{};
        const int logR = d.digits[i];
        const int logB = logn - logA - logR;
        const bool lastp = (i == m - 1);
        p.logR = logR;
        p.in = (i == 0) ? io.in : io.work;
        p.out = lastp ? io.out : io.work;
        p.mt = tb.mt;
        p.mt_shift = tb.mt_log - logR;
        p.tl = tb.tl;
        p.th = (i == 0 && tb.th_scaled) ? tb.th_scaled : tb.th;
        p.in_limit = (i == 0) ? io.in_limit : ~0ull;
        p.coset_enable = (i == 0 && io.ol != nullptr) ? 1 : 0;
        p.ol = io.ol;
        p.oh = io.oh;
        p.scale_enable = (lastp && io.scale_last) ? 1 : 0;
        p.scale = io.scale;
        if (m == 1) {
            p.logC = 0;
            p.lo_log = 0; p.mid_log = 0;
            p.in_rs = 1; p.out_rs = 1;
            p.in_cs = 0; p.out_cs = 0;
            p.rfast_load = 0;
            p.tw_enable = 0;
            pd.ntiles = 1;
        } else if (!lastp) {
            // column pass on [A][R][B]
            int logC = tile_cap - logR;
            if (logC > logB) logC = logB;
            if (logC > tu.max_col_log) logC = tu.max_col_log;
            if (logC < 0) logC = 0;
            p.logC = logC;
            p.lo_log = logB - logC;       // t_lo = column block, t_mid = a
            p.mid_log = logA;
            p.in_lo = p.out_lo = 1ull << logC;
            p.in_mid = p.out_mid = 1ull << (logR + logB);
            p.in_hi = p.out_hi = 0;
            p.in_rs = p.out_rs = 1ull << logB;
            p.in_cs = p.out_cs = 1;
            p.rfast_load = 0;
            p.tw_enable = 1;
            p.tw_scale = 1ull << logA;
            p.tw_row_k = 1;
            p.twd = tb.twd[i];
            p.twd_stride = 1ull << logB;
            pd.ntiles = (uint32_t)(n >> (logR + logC));
        } else {
            // transposing pass: memory [k_1][k_2]..[k_{m-1}][j_m] -> natural k = k_1 + N_1 k_2 + ... ; C adjacent k_1 per tile
            const int logN1 = d.digits[0];
            int logC = tile_cap - logR;
            if (logC > logN1) logC = logN1;
            if (logC > tu.max_col_log) logC = tu.max_col_log;
            if (logC < 0) logC = 0;
            p.logC = logC;
            // tile id -> (t_hi = k_1 block, t_mid = k_2, t_lo = k_3); for m == 2 there is no k_2/k_3, for m == 3 no k_3
            const int logN2 = (m >= 3) ? d.digits[1] : 0;
            const int logN3 = (m >= 4) ? d.digits[2] : 0;
            p.lo_log = logN3;
            p.mid_log = logN2;
            p.in_rs = 1;
            p.in_cs = n >> logN1;                              // next k_1
            p.in_hi = (n >> logN1) << logC;
            p.in_mid = 1ull << (logR + logN3);                  // next k_2
            p.in_lo = 1ull << logR;                             // next k_3
            p.out_cs = 1;
            p.out_hi = 1ull << logC;
            p.out_mid = 1ull << logN1;
            p.out_lo = 1ull << (logN1 + logN2);
            p.out_rs = n >> logR;                               // k_m is the most significant output digit
            p.rfast_load = 1;
            p.tw_enable = 0;
            pd.ntiles = (uint32_t)(n >> (logR + logC));
        }
        int loge = tu.loge;
        const int logT = p.logR + p.logC;
        if (io.cols > 1 && loge == 2 && tu.loge_cols == 3 && logT == 12 && (p.logR == 10 || p.logR == 9 || p.logR == 8)) loge = 3;
        if (loge > logT) loge = logT;
        // threads per workgroup: <= 1024 (loge 1,2), 512 (loge 3), 256 (loge 4) -- matches the kernels' launch bounds
        while (logT - loge > (loge >= 4 ? 8 : (loge == 3 ? 9 : 10))) ++loge;
        if (loge > 4) return false;
        if (lastp && m > 1 && p.logR <= loge) loge = p.logR - 1;   // transposing pass needs >= 2 rounds (load r-fast, store c-fast)
        if (loge < 1) return false;
        pd.loge = loge;
        pd.threads = 1u << (logT - loge);
        pd.lds_bytes = ((uint32_t)sizeof(Fe) << logT) + tile_twiddle_bytes(pd.p.logR);   // the tile, then its twiddles
        pd.cols = io.cols;
        if (io.cols > 1) {
            p.col_enable = 1;
            p.col_tiles_log = 0;
            while ((1u << p.col_tiles_log) < pd.ntiles) ++p.col_tiles_log;               // tiles per column: a power of two
            p.col_stride = n;
            p.col_stride_in = (i == 0 && io.col_stride_in) ? io.col_stride_in : n;
        }
        logA += logR;
    }
    if (tu.prune && io.in_limit < n && m > 1) {
        // zero-padded input: rows j_1 >= ceil(in_limit / B) of the first pass are zero.  The fixed-shape kernels handle
        // degenerate stages in their first two rounds only, so the count is capped at what those cover.
        const int logR = d.digits[0];
        const uint64_t B = n >> logR;
        const uint64_t rows_nz = (io.in_limit + B - 1) / B;
        int k = 0;
        while (k < logR - 1 && ((uint64_t)1 << (logR - k - 1)) >= rows_nz && rows_nz > 0) ++k;
        const int loge = d.pass[0].loge;
        const int nr = (logR + loge - 1) / loge;
        const int cap = (logR - loge * (nr - 1)) + (nr > 1 ? loge : 0);
        if (k > cap) k = cap;
        d.pass[0].p.prune_log = k;
    }
    if (tu.tw_on_load) {
        // twiddle-on-load: the table of pass i-1 is indexed like the work buffer it wrote (index mod R*B), so pass i can fetch
        // it with the same addresses as its data
        int logAi = 0;
        for (int i = 0; i + 1 < m; ++i) {
            const int logR = d.digits[i], logB = logn - logAi - logR;
            if (tb.twd[i]) {
                d.pass[i].p.tw_enable = 0;
                d.pass[i].p.twd = nullptr;
                d.pass[i + 1].p.twd_in = tb.twd[i];
                d.pass[i + 1].p.twd_in_mask = (1ull << (logR + logB)) - 1;
            }
            logAi += logR;
        }
    }
    return true;
}

// ---------------------------------------------------------------------------------------------------
// Batched transforms used by the multi-GPU four-step NTT (each rank runs them on its slab).
//   BATCH_COLS  : data [len][batch] row-major, transform along axis 0 (stride = batch) for every column,
//                 output in the same layout with natural order along axis 0.
//   BATCH_ROWS_T: data [batch][len] (contiguous rows), transform every row, output TRANSPOSED [len][batch].
// At most two passes each (len <= 2^(2*max_digit_log)); tables are those of a primitive len-th root.
enum BatchKind { BATCH_COLS = 0, BATCH_ROWS_T = 1 };

// Optional extras of the batched plans (multi-GPU four-step):
//   outer_*  (BATCH_COLS): multiply output element (row r, column c) by w_n^(r * (outer_col_base + c)) [* scale folded into
//            outer_th] in the store epilogue of the last pass -- the four-step twiddle between the two stages;
//   chunks_log (BATCH_ROWS_T): the input is [2^chunks_log][batch][len / 2^chunks_log] (what all_to_all_single leaves behind)
//            instead of [batch][len]; handled in the load addresses of the first pass, no reassembly copy.
struct BatchExtras {
    const Fe* outer_tl = nullptr;
    const Fe* outer_th = nullptr;
    uint64_t outer_col_base = 0;
    // the same outer twiddles as a direct table [len][batch] of THIS rank's slab (entry (r, c) = w_n^(r * (outer_col_base + c))
    // [* scale]): one prefetched load + one modmul per element instead of two loads + two modmuls
    const Fe* outer_twd = nullptr;
    int chunks_log = 0;
    // direct table of the inter-pass twiddles of a two-pass plan: [k < 2^digits[0]][b < len >> digits[0]] = w_len^(b*k)
    // (one coalesced load + one modmul per element instead of the two-level lookup's two loads + two modmuls)
    const Fe* inner_twd = nullptr;
    // BATCH_ROWS_T: leading dimension of the transposed output (0 = batch): the rows of this call are `batch` adjacent columns
    // of a wider [len][out_ld] matrix -- a row block of the corner turn transformed as soon as it has landed (overlap)
    uint64_t out_ld = 0;
    // BATCH_ROWS_T with chunks_log > 0: distance (in elements) between consecutive chunks; 0 = batch * len / chunks, i.e. the
    // chunks of THIS call's rows lie back to back.  A row block of a bigger [chunks][all rows][len / chunks] buffer passes the
    // bigger buffer's chunk size here and points `in` at its first row.
    uint64_t chunk_stride = 0;
    // BATCH_COLS: natural output rows [diag_lo, diag_lo + diag_n) are stored into diag_out (same element index) instead of
    // `out` (PassParams::out_alt): the block of the corner turn a rank keeps for itself
    Fe* diag_out = nullptr;
    uint32_t diag_lo = 0, diag_n = 0;
    // BATCH_COLS, general form of the same: the natural output rows in blocks of `block_rows` (a power of two), block h stored
    // into block_out[h] (same element index; len / block_rows <= SC_MAX_BLOCKS entries) -- the direct-store corner turn, where
    // block_out[h] points into rank h's receive buffer.  Takes precedence over diag_out.
    Fe* const* block_out = nullptr;
    uint32_t block_rows = 0;
};

inline bool plan_batched(NttPlanDesc& d, BatchKind kind, int loglen, int logbatch, const NttTables& tb,
                         const Fe* in, Fe* work, Fe* out, const NttTuning& tu_in, const BatchExtras& ex = BatchExtras()) {
    if (loglen < 1 || loglen + logbatch > 34) return false;
    NttTuning tu = resolve_tuning(tu_in, 24);
    if (tu_in.max_digit_log < 0) tu.max_digit_log = 8;
    const int m = (loglen <= tu.max_digit_log) ? 1 : 2;
    if (loglen > 2 * tu.max_digit_log + 2) return false;
    d.logn = loglen;
    d.npasses = m;
    d.digits[0] = (m == 1) ? loglen : (loglen + 1) / 2;
    d.digits[1] = loglen - d.digits[0];
    int tile_cap = tu.max_tile_log;
    {
        const int floor_log = d.digits[0] + 2 < tu.max_tile_log ? d.digits[0] + 2 : tu.max_tile_log;
        while (tile_cap > floor_log && loglen + logbatch - tile_cap < tu.min_tiles_log) --tile_cap;
    }
    const uint64_t len = 1ull << loglen, batch = 1ull << logbatch;
    for (int i = 0; i < m; ++i) {
        NttPassDesc& pd = d.pass[i];
        PassParams& p = pd.p;
        p = PassParams{};
        const int logR = d.digits[i];
        const bool lastp = (i == m - 1);
        const int logA = (i == 0) ? 0 : d.digits[0];
        p.logR = logR;
        p.in = (i == 0) ? in : work;
        p.out = lastp ? out : work;
        p.mt = tb.mt;
        p.mt_shift = tb.mt_log - logR;
        p.tl = tb.tl;
        p.th = tb.th;
        p.in_limit = ~0ull;
        if (kind == BATCH_COLS) {
            // [A][R][Blow * batch]: column pass; the last pass writes rows in natural order k = a + A*k_m
            const int logBlow = loglen - logA - logR;       // untransformed lower digits of the transform axis
            const int logB = logBlow + logbatch;
            int logC = tile_cap - logR;
            if (logC > logB) logC = logB;
            if (logC > tu.max_col_log) logC = tu.max_col_log;
            if (logC < 0) logC = 0;
            p.logC = logC;
            p.lo_log = logB - logC;
            p.mid_log = logA;
            p.in_lo = p.out_lo = 1ull << logC;
            p.in_mid = 1ull << (logR + logB);
            p.in_rs = 1ull << logB;
            p.in_cs = p.out_cs = 1;
            if (lastp && m == 2) {
                p.out_mid = batch;                              // a = k_1 -> row k_1
                p.out_rs = batch << logA;                       // k_2 -> row N_1 * k_2
            } else {
                p.out_mid = p.in_mid;
                p.out_rs = p.in_rs;
            }
            p.rfast_load = 0;
            p.tw_enable = lastp ? 0 : 1;
            p.tw_col_shift = logbatch;
            p.tw_scale = 1ull << logA;
            p.tw_row_k = 1;
            if (!lastp && ex.inner_twd) { p.twd = ex.inner_twd; p.twd_stride = 1ull << logBlow; }
            if (lastp && ex.outer_tl) {
                // fused outer twiddle: natural output row = t_mid + N_1 * k (two passes) or k (one pass); column = local column
                p.tw_enable = 1;
                p.tw_col_shift = 0;
                p.tw_scale = 1;
                p.tw_col_base = ex.outer_col_base;
                p.tw_row_k = 1ull << logA;
                p.tw_row_mid = (m == 2) ? 1 : 0;
                p.tl = ex.outer_tl;
                p.th = ex.outer_th;
                if (ex.outer_twd) {
                    p.twd = ex.outer_twd;           // indexed by (natural row, LOCAL column)
                    p.twd_stride = batch;
                    p.tw_col_base = 0;
                }
            }
            if (lastp && ((ex.block_out && ex.block_rows) || (ex.diag_out && ex.diag_n))) {
                const uint32_t rows = ex.block_out ? ex.block_rows : ex.diag_n;
                const uint64_t nblk = len / rows;
                if ((rows & (rows - 1)) || nblk < 1 || nblk > SC_MAX_BLOCKS || nblk * rows != len) return false;
                if (!ex.block_out && (ex.diag_lo % rows)) return false;
                p.blk_enable = 1;
                p.blk_log = 0;
                while ((1u << p.blk_log) < rows) ++p.blk_log;
                p.blk_row_k = 1u << logA;                       // natural output row = t_mid + N_1 * k (two passes) or k (one pass)
                p.blk_row_mid = (m == 2) ? 1u : 0u;
                for (uint64_t h = 0; h < nblk; ++h)
                    p.out_blk[h] = ex.block_out ? ex.block_out[h] : (h == ex.diag_lo / rows ? ex.diag_out : p.out);
            }
            pd.ntiles = (uint32_t)((len * batch) >> (logR + logC));
        } else if (!lastp) {
            // rows, first digit: [batch][R][B = N_2]: t_lo = column block, t_mid = (none), t_hi = batch row
            const int logB = loglen - logR;
            int logC = tile_cap - logR;
            if (logC > logB) logC = logB;
            if (logC > tu.max_col_log) logC = tu.max_col_log;
            if (logC < 0) logC = 0;
            p.logC = logC;
            p.lo_log = logB - logC;
            p.mid_log = 0;
            p.in_lo = p.out_lo = 1ull << logC;
            p.in_hi = p.out_hi = len;
            p.in_rs = p.out_rs = 1ull << logB;
            p.in_cs = p.out_cs = 1;
            if (ex.chunks_log) {
                // element (b, j) of the row sits at chunk (j / cw), row b, offset (j % cw), cw = len / chunks
                if (ex.chunks_log > logR) return false;           // chunk boundaries must fall on whole rows of this pass
                const uint64_t cs = ex.chunk_stride ? ex.chunk_stride : (len >> ex.chunks_log) << logbatch;     // next chunk
                p.in_hi = len >> ex.chunks_log;                   // next batch row inside a chunk
                if (ex.chunks_log == logR) p.in_rs = cs;          // every row of this pass is a chunk of its own
                else { p.in_split = logR - ex.chunks_log; p.in_rs_hi = cs; }
            }
            p.rfast_load = 0;
            p.tw_enable = 1;
            p.tw_scale = 1;
            p.tw_row_k = 1;
            if (ex.inner_twd) { p.twd = ex.inner_twd; p.twd_stride = 1ull << logB; }
            pd.ntiles = (uint32_t)((len * batch) >> (logR + logC));
        } else {
            // rows, last digit: C adjacent batch rows x R contiguous; output [k][batch row], k = k_1 + N_1 * k_2
            int logC = tile_cap - logR;
            if (logC > logbatch) logC = logbatch;
            if (logC > tu.max_col_log) logC = tu.max_col_log;
            if (logC < 0) logC = 0;
            p.logC = logC;
            p.lo_log = 0;
            p.mid_log = logA;                                   // t_mid = k_1 (absent when m == 1)
            p.in_rs = 1;
            p.in_cs = len;
            p.in_hi = len << logC;
            p.in_mid = 1ull << logR;
            if (ex.chunks_log && m == 1) {
                // single (transposing) pass straight from the chunked layout
                if (ex.chunks_log > logR) return false;
                const uint64_t cs = ex.chunk_stride ? ex.chunk_stride : (len >> ex.chunks_log) << logbatch;
                p.in_cs = len >> ex.chunks_log;
                p.in_hi = (len >> ex.chunks_log) << logC;
                if (ex.chunks_log == logR) p.in_rs = cs;          // one element per chunk
                else { p.in_split = logR - ex.chunks_log; p.in_rs_hi = cs; }
            }
            const uint64_t ld = ex.out_ld ? ex.out_ld : batch;
            p.out_cs = 1;
            p.out_hi = 1ull << logC;
            p.out_mid = ld;
            p.out_rs = ld << logA;
            p.rfast_load = 1;
            p.tw_enable = 0;
            pd.ntiles = (uint32_t)((len * batch) >> (logR + logC));
        }
        int loge = tu.loge;
        const int logT = p.logR + p.logC;
        if (loge > logT) loge = logT;
        while (logT - loge > (loge >= 4 ? 8 : (loge == 3 ? 9 : 10))) ++loge;
        if (loge > 4) return false;
        if (kind == BATCH_ROWS_T && lastp && p.logR <= loge) loge = p.logR - 1;
        if (loge < 1) return false;
        pd.loge = loge;
        pd.threads = 1u << (logT - loge);
        pd.lds_bytes = ((uint32_t)sizeof(Fe) << logT) + tile_twiddle_bytes(pd.p.logR);   // the tile, then its twiddles
    }
    return true;
}

}  // namespace sc
