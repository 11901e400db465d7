// fourstep.hip -- batched transforms with the outer twiddle, the sharded four-step transform as one plan object per rank, the
// library's RCCL communicator and the direct-store corner turn over HIP IPC (code/ntt.py:3-30 on n = n1 * n2).
#include "core.h"

// direct table of one rank's outer four-step twiddles: out[r * cols + c] = w^(r * (col_base + c)) [* n^-1 via th]
__global__ void __launch_bounds__(256) outer_table_kernel(Fe* out, uint64_t count, int logcols, uint64_t col_base, const Fe* __restrict__ tl, const Fe* __restrict__ th) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    uint64_t r = i >> logcols, c = i & ((1ull << logcols) - 1);
    out[i] = pow2level(tl, th, r * (col_base + c));
}

// four-step outer twiddle on a rank's slab: data[r][c] *= w^((row_base + r) * (col_base + c)) [* scale]
__global__ void __launch_bounds__(256) twiddle_matrix_kernel(Fe* __restrict__ data, uint64_t rows, int logcols, uint64_t row_base, uint64_t col_base,
                                                             const Fe* __restrict__ tl, const Fe* __restrict__ th, int scale_enable, Fe scale_m) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (rows << logcols)) return;
    uint64_t r = i >> logcols, c = i & ((1ull << logcols) - 1);
    uint64_t e = (row_base + r) * (col_base + c);
    Fe t = pow2level(tl, th, e);
    if (scale_enable) t = mont_mul(t, scale_m);
    data[i] = mont_mul(data[i], t);
}

extern "C" {

// ---- batched transforms + outer twiddle (building blocks of the multi-GPU four-step NTT)
struct BatchCall {
    const Fe* in = nullptr;
    Fe* out = nullptr;
    uint64_t len = 0, batch = 0;
    int kind = 0;                    // 0: columns of [len][batch]; 1: rows of [batch][len] -> [len][batch]
    Fe root{0, 0};                   // primitive len-th root
    bool outer = false;              // kind 0: fused outer twiddle  outer_root^(r * (outer_col_base + c)) [* outer_order^-1]
    Fe outer_root{0, 0};
    uint64_t outer_order = 0, outer_col_base = 0;
    bool outer_ninv = false;
    uint64_t chunks = 1, out_ld = 0, chunk_stride = 0;   // kind 1 (BatchExtras)
    Fe* diag_out = nullptr;          // kind 0: second destination for natural rows [diag_lo, diag_lo + diag_n)
    uint32_t diag_lo = 0, diag_n = 0;
    Fe* const* block_out = nullptr;  // kind 0: destination table, one entry per block of block_rows natural rows (direct-store corner turn)
    uint32_t block_rows = 0;
    Fe* work = nullptr;              // work buffer of a two-pass plan (nullptr: scratch slot 0, len * batch elements)
    int pass_lo = 0, pass_hi = 4;    // run passes [pass_lo, pass_hi) of the plan only
    bool roots_checked = false;      // the caller has validated the roots already (a cached plan object)
};

// caller holds g_mu and has run ensure_init()
static int batch_call(const BatchCall& c, hipStream_t st, int* npasses_out = nullptr) {
    const uint64_t len = c.len, batch = c.batch;
    const int kind = c.kind;
    if (!is_pow2(len) || !is_pow2(batch) || len < 2) return fail(SC_ERR_NOT_POW2, "batched ntt needs power-of-two length >= 2 and batch");
    if (kind != 0 && kind != 1) return fail(SC_ERR_BAD_ARG, "kind must be 0 (columns) or 1 (rows, transposed output)");
    const uint64_t chunks = c.chunks ? c.chunks : 1;
    if (!is_pow2(chunks) || (chunks > 1 && kind != 1)) return fail(SC_ERR_BAD_ARG, "chunked input is for kind 1 and needs a power-of-two chunk count");
    if (c.outer && kind != 0) return fail(SC_ERR_BAD_ARG, "the outer twiddle belongs to the column stage (kind 0)");
    if ((c.diag_out || c.block_out) && kind != 0) return fail(SC_ERR_BAD_ARG, "the second destination belongs to the column stage (kind 0)");
    Fe rt = c.root;
    if (!c.roots_checked) SCCHK(check_root(rt, len));
    const int loglen = ilog2(len), logbatch = ilog2(batch);
    PlanTables* pt;
    SCCHK(get_plan(rt, loglen, false, st, &pt));
    NttTables tb;
    tb.mt = pt->mt; tb.mt_log = pt->mt_log; tb.tl = pt->tl; tb.th = pt->th;
    BatchExtras ex;
    ex.chunks_log = ilog2(chunks);
    ex.chunk_stride = c.chunk_stride;
    ex.diag_out = c.diag_out;
    ex.diag_lo = c.diag_lo;
    ex.diag_n = c.diag_n;
    ex.block_out = c.block_out;
    ex.block_rows = c.block_rows;
    if (c.out_ld) {
        if (kind != 1 || c.out_ld < batch) return fail(SC_ERR_BAD_ARG, "an output leading dimension belongs to kind 1 and must be >= batch");
        ex.out_ld = c.out_ld;
    }
    if (c.outer) {
        const uint64_t outer_order = c.outer_order, outer_col_base = c.outer_col_base;
        if (!is_pow2(outer_order) || outer_order < len * batch) return fail(SC_ERR_BAD_ARG, "outer twiddle order too small");
        if ((len - 1) * (outer_col_base + batch - 1) >= outer_order) return fail(SC_ERR_BAD_ARG, "outer twiddle exponent out of range");
        Fe ort = c.outer_root;
        if (!c.roots_checked) SCCHK(check_root(ort, outer_order));
        PlanTables* po;
        SCCHK(get_plan(ort, ilog2(outer_order), c.outer_ninv, st, &po));
        ex.outer_tl = po->tl;
        ex.outer_th = c.outer_ninv ? po->th_ninv : po->th;
        ex.outer_col_base = outer_col_base;
        if (g.tuning.direct_tw_max_log > 0 && len * batch <= (1ull << g.tuning.direct_tw_max_log)) {
            // a rank transforms the same slab shape over and over: keep its outer twiddles as a direct table (prefetched by the
            // kernel at the top of its last round) instead of two table loads and an extra modmul per element
            OuterKey ok{ort.lo, ort.hi, outer_order, len, batch, outer_col_base, c.outer_ninv ? 1 : 0};
            ++g.tick;
            auto it = g.outers.find(ok);
            if (it == g.outers.end()) {
                SCCHK(evict_outer_tables());
                OuterTable t;
                const uint64_t count = len * batch;
                HIPCHK(hipMalloc((void**)&t.d, count * sizeof(Fe)));
                hipLaunchKernelGGL(outer_table_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, t.d, count, logbatch, outer_col_base, ex.outer_tl, ex.outer_th);
                HIPCHK(hipGetLastError());
                HIPCHK(hipStreamSynchronize(st));
                it = g.outers.emplace(ok, t).first;
            }
            it->second.last_use = g.tick;
            ex.outer_twd = it->second.d;
        }
        // get_plan may have rehashed the map: re-fetch the inner tables
        SCCHK(get_plan(rt, loglen, false, st, &pt));
        tb.mt = pt->mt; tb.mt_log = pt->mt_log; tb.tl = pt->tl; tb.th = pt->th;
    }
    Fe* work = c.work;
    if (!work) { void* w; SCCHK(scratch(0, len * batch * sizeof(Fe), &w)); work = (Fe*)w; }
    NttPlanDesc d;
    bool planned = false;
    SCCHK(plan_batched_direct(d, kind == 0 ? BATCH_COLS : BATCH_ROWS_T, loglen, logbatch, pt, c.in, work, c.out, ex, st, &planned));
    if (!planned) return fail(SC_ERR_UNSUPPORTED, "unsupported batched transform shape");
    if (d.npasses == 2 && kind == 0 && c.in == c.out) return fail(SC_ERR_BAD_ARG, "two-pass column transform must be out of place");
    if (kind == 1 && c.in == c.out) return fail(SC_ERR_BAD_ARG, "transposing row transform must be out of place");
    if (npasses_out) *npasses_out = d.npasses;
    if (c.pass_lo > 0 || c.pass_hi < d.npasses) {
        NttPlanDesc part = d;
        part.npasses = 0;
        for (int i = c.pass_lo; i < d.npasses && i < c.pass_hi; ++i) part.pass[part.npasses++] = d.pass[i];
        return part.npasses ? run_plan(part, st) : SC_OK;
    }
    return run_plan(d, st);
}

static int batch_ex_impl(const void* d_in, void* d_out, uint64_t len, uint64_t batch, int kind, const uint64_t root[2],
                         const uint64_t outer_root[2], uint64_t outer_order, uint64_t outer_col_base, int outer_scale_ninv, uint64_t chunks, uint64_t out_ld, void* stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    BatchCall c;
    c.in = (const Fe*)d_in; c.out = (Fe*)d_out; c.len = len; c.batch = batch; c.kind = kind; c.root = fe_from(root);
    if (outer_root) { c.outer = true; c.outer_root = fe_from(outer_root); c.outer_order = outer_order; c.outer_col_base = outer_col_base; c.outer_ninv = outer_scale_ninv != 0; }
    c.chunks = chunks; c.out_ld = out_ld;
    return batch_call(c, pick_stream(stream));
}

int sc_ntt_batch_ex_dev(const void* d_in, void* d_out, uint64_t len, uint64_t batch, int kind, const uint64_t root[2],
                        const uint64_t outer_root[2], uint64_t outer_order, uint64_t outer_col_base, int outer_scale_ninv, uint64_t chunks, void* stream) {
    return batch_ex_impl(d_in, d_out, len, batch, kind, root, outer_root, outer_order, outer_col_base, outer_scale_ninv, chunks, 0, stream);
}

int sc_ntt_rows_t_ld_dev(const void* d_in, void* d_out, uint64_t len, uint64_t batch, const uint64_t root[2], uint64_t chunks, uint64_t out_ld, void* stream) {
    return batch_ex_impl(d_in, d_out, len, batch, 1, root, nullptr, 0, 0, 0, chunks, out_ld, stream);
}

int sc_ntt_batch_dev(const void* d_in, void* d_out, uint64_t len, uint64_t batch, int kind, const uint64_t root[2], void* stream) {
    return sc_ntt_batch_ex_dev(d_in, d_out, len, batch, kind, root, nullptr, 0, 0, 0, 1, stream);
}

int sc_twiddle_matrix_dev(void* d_data, uint64_t rows, uint64_t cols, uint64_t row_base, uint64_t col_base, const uint64_t root[2], uint64_t order,
                          const uint64_t scale[2], void* stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    hipStream_t st = pick_stream(stream);
    if (!is_pow2(cols) || !is_pow2(order) || order < 2) return fail(SC_ERR_NOT_POW2, "cols and order must be powers of two");
    if ((row_base + rows - 1) * (col_base + cols - 1) >= order) return fail(SC_ERR_BAD_ARG, "twiddle exponent out of range");
    Fe rt = fe_from(root);
    SCCHK(check_root(rt, order));
    PlanTables* pt;
    SCCHK(get_plan(rt, ilog2(order), false, st, &pt));
    int scale_enable = 0;
    Fe scale_m = fe_mont_one();
    if (scale && !(scale[0] == 1 && scale[1] == 0)) { scale_enable = 1; scale_m = to_mont(fe_from(scale)); }
    uint64_t total = rows * cols;
    if (!total) return SC_OK;
    hipLaunchKernelGGL(twiddle_matrix_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (Fe*)d_data, rows, ilog2(cols), row_base, col_base,
                       pt->tl, pt->th, scale_enable, scale_m);
    HIPCHK(hipGetLastError());
    return SC_OK;
}

// ---- the sharded four-step transform as ONE plan object per rank (stark-anatomy_amd/sharded.py: ShardedNtt)
//
// n = n1 * n2; rank g of G holds the column slab [R][C/G] of the row-major R x C matrix of its input (forward: R = n1, C = n2;
// inverse: R = n2, C = n1) and produces the column slab [C][R/G] of the output.  Stages, all asynchronous on the caller's stream:
//   cols : length-R column transforms + outer twiddle -> `send`, laid out [G][R/G][C/G] (block h = the rows rank h receives);
//          the block the rank keeps (h == g) can go straight into `recv` (never copied, never sent)
//   exchange : block h of `send` -> rank h's `recv` block g   (the caller's collective, or sc_fourstep_run_dev over RCCL)
//   rows : length-C row transforms reading `recv` [G][R/G][C/G] in place, written transposed into dst [C][R/G]; optionally one
//          ROW BLOCK at a time (overlap with an exchange issued in blocks), the second pass of a two-pass row transform deferred
//          to ONE full-size launch (sc_fourstep_rows_finish_dev) so that the small per-block launches are half as many
struct sc_fourstep {
    int log2n, rank, world;
    uint64_t n, n1, n2;
    struct Dir {
        uint64_t R, C;
        Fe root;          // the transform's root (forward: w, inverse: w^-1)
        Fe root_cols;     // root^C: primitive R-th root
        Fe root_rows;     // root^R: primitive C-th root
        bool ninv;
    } dir[2];
    // direct-store corner turn (sc_fourstep_set_peers): every rank's region, mapped here through HIP IPC -- [4 KiB of flags]
    // [receive buffer 0][receive buffer 1], n / world elements each; transform number `epoch` lands in buffer epoch & 1
    bool peers_set = false;
    uint8_t* region[SC_MAX_BLOCKS] = {};
    uint64_t epoch = 0;
    // pinned host word the flag barrier writes the epoch to when it gives up waiting for a peer: read for free at the top of every
    // later transform, which then FAILS (the buffers hold a transform that never completed) until the plan is set up again
    volatile uint64_t* timed_out = nullptr;
};
constexpr size_t FOURSTEP_FLAG_BYTES = 4096;

namespace {
struct RcclApi {
    void* handle = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, sc_rccl_id_t, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
static RcclApi rccl;
static void* g_comm = nullptr;
static int g_comm_rank = -1, g_comm_world = 0;
static hipStream_t g_comm_stream = nullptr;          // the exchanges of an overlapped (row-block) corner turn run here
static std::vector<hipEvent_t> g_comm_events;        // [0]: column stage done; [1 + q]: row block q has landed

static int rccl_load(const char* path) {
    if (rccl.handle) return SC_OK;
    const char* names[] = {path, getenv("STARKCORE_RCCL"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* nm : names) {
        if (!nm || !*nm) continue;
        // a copy the process has loaded already (torch's) is preferred: one RCCL per process
        h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
        if (h) break;
    }
    for (const char* nm : names) {
        if (h) break;
        if (!nm || !*nm) continue;
        h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
    }
    if (!h) return fail(SC_ERR_UNSUPPORTED, std::string("RCCL not found: ") + (dlerror() ? dlerror() : "librccl.so"));
    RcclApi a;
    a.handle = h;
    a.GetUniqueId = (int (*)(void*))dlsym(h, "ncclGetUniqueId");
    a.CommInitRank = (int (*)(void**, int, sc_rccl_id_t, int))dlsym(h, "ncclCommInitRank");
    a.CommDestroy = (int (*)(void*))dlsym(h, "ncclCommDestroy");
    a.GroupStart = (int (*)())dlsym(h, "ncclGroupStart");
    a.GroupEnd = (int (*)())dlsym(h, "ncclGroupEnd");
    a.Send = (int (*)(const void*, size_t, int, int, void*, hipStream_t))dlsym(h, "ncclSend");
    a.Recv = (int (*)(void*, size_t, int, int, void*, hipStream_t))dlsym(h, "ncclRecv");
    a.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
    if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.GroupStart || !a.GroupEnd || !a.Send || !a.Recv)
        return fail(SC_ERR_UNSUPPORTED, "the RCCL library lacks a required symbol");
    rccl = a;
    return SC_OK;
}
#define RCCLCHK(expr)                                                                                                     \
    do {                                                                                                                  \
        int _r = (expr);                                                                                                  \
        if (_r != 0) return fail(SC_ERR_HIP, std::string(#expr) + ": " + (rccl.GetErrorString ? rccl.GetErrorString(_r) : "RCCL error")); \
    } while (0)

// corner turn over the native communicator: rows [row0, row0 + nrows) of every block; block h of `send` -> rank h, block h of `recv`
// <- rank h; the rank's own block is not touched (the column stage put it into `recv` already)
static int rccl_exchange(const sc_fourstep* p, int dirn, const Fe* send, Fe* recv, uint64_t row0, uint64_t nrows, hipStream_t st) {
    const sc_fourstep::Dir& d = p->dir[dirn];
    const uint64_t G = (uint64_t)p->world, rw = d.R / G, cw = d.C / G;
    RCCLCHK(rccl.GroupStart());
    int bad = 0;                                       // a failed call still closes the group: nothing is left half-open
    for (uint64_t h = 0; h < G && !bad; ++h) {
        if ((int)h == p->rank) continue;
        const uint64_t off = (h * rw + row0) * cw;
        bad = rccl.Send(send + off, nrows * cw * sizeof(Fe), 1 /* ncclUint8 */, (int)h, g_comm, st);
        if (!bad) bad = rccl.Recv(recv + off, nrows * cw * sizeof(Fe), 1 /* ncclUint8 */, (int)h, g_comm, st);
    }
    const int ended = rccl.GroupEnd();
    RCCLCHK(bad);
    RCCLCHK(ended);
    return SC_OK;
}

// passes of a batched plan of this length (mirrors plan_batched: one pass up to the digit cap, two above)
static int batched_passes(int loglen) { return loglen <= (g.tuning.max_digit_log < 0 ? 8 : g.tuning.max_digit_log) ? 1 : 2; }

static int fourstep_cols(const sc_fourstep* p, int dirn, const Fe* src, Fe* send, Fe* recv_diag, hipStream_t st) {
    const sc_fourstep::Dir& d = p->dir[dirn];
    const uint64_t G = (uint64_t)p->world, rw = d.R / G, cw = d.C / G;
    BatchCall c;
    c.in = src; c.out = send; c.len = d.R; c.batch = cw; c.kind = 0; c.root = d.root_cols;
    c.outer = true; c.outer_root = d.root; c.outer_order = p->n; c.outer_col_base = (uint64_t)p->rank * cw; c.outer_ninv = d.ninv;
    if (recv_diag) { c.diag_out = recv_diag; c.diag_lo = (uint32_t)(p->rank * rw); c.diag_n = (uint32_t)rw; }
    c.roots_checked = true;
    return batch_call(c, st);
}

// rows [q * R/(G K), (q+1) * R/(G K)) of the rank (K = nblocks; 1 = all of them).  Two-pass row transforms with `defer`: only the
// first pass runs here, into the rank-sized work buffer; fourstep_rows_finish runs the second pass over all rows in one launch.
static int fourstep_rows(const sc_fourstep* p, int dirn, const Fe* recv, Fe* dst, uint64_t q, uint64_t K, bool defer, hipStream_t st) {
    const sc_fourstep::Dir& d = p->dir[dirn];
    const uint64_t G = (uint64_t)p->world, rw = d.R / G, cw = d.C / G;
    if (K == 0 || rw % K || !is_pow2(rw / K) || q >= K) return fail(SC_ERR_BAD_ARG, "row blocks must divide the rank's rows into powers of two");
    const uint64_t rk = rw / K;
    void* w;
    SCCHK(scratch(0, rw * d.C * sizeof(Fe), &w));
    BatchCall c;
    c.in = recv + q * rk * cw; c.out = dst + q * rk; c.len = d.C; c.batch = rk; c.kind = 1; c.root = d.root_rows;
    c.chunks = G; c.chunk_stride = rw * cw; c.out_ld = rw;
    c.work = (Fe*)w + q * rk * d.C;
    c.roots_checked = true;
    // (a single-pass plan has nothing to defer: its one pass is the transposing one)
    if (defer && batched_passes(ilog2(d.C)) == 2) { c.pass_lo = 0; c.pass_hi = 1; }
    return batch_call(c, st);
}

static int fourstep_rows_finish(const sc_fourstep* p, int dirn, Fe* dst, hipStream_t st) {
    const sc_fourstep::Dir& d = p->dir[dirn];
    const uint64_t G = (uint64_t)p->world, rw = d.R / G;
    void* w;
    SCCHK(scratch(0, rw * d.C * sizeof(Fe), &w));
    BatchCall c;
    c.in = (const Fe*)w + 1;          // (unused by the second pass; only has to differ from `out`)
    c.out = dst; c.len = d.C; c.batch = rw; c.kind = 1; c.root = d.root_rows;
    c.work = (Fe*)w;
    c.roots_checked = true;
    c.pass_lo = 1; c.pass_hi = 2;
    return batch_call(c, st);
}
}  // namespace

int sc_fourstep_create(int log2n, const uint64_t root[2], int rank, int world, sc_fourstep_t** out) {
    return sc_fourstep_create_ex(log2n, root, rank, world, 0, out);
}
int sc_fourstep_create_ex(int log2n, const uint64_t root[2], int rank, int world, int log_n1, sc_fourstep_t** out) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    if (!out || !root) return fail(SC_ERR_BAD_ARG, "null argument");
    if (log2n < 2 || log2n > 40 || world < 1 || world > SC_MAX_BLOCKS || (world & (world - 1)) || rank < 0 || rank >= world) return fail(SC_ERR_BAD_ARG, "bad four-step shape");
    const uint64_t n = 1ull << log2n;
    Fe rt = fe_from(root);
    SCCHK(check_root(rt, n));
    // small domains: square split; large ones: n1 = 2^8, so that the column stage of the forward transform is ONE pass
    // (256-point transforms) and the row stage two, and the other way round for the inverse: three passes per transform.
    // log_n1 > 0 overrides the split (e.g. the square 2^12 x 2^12 at 2^24: fewer, longer rows per rank and per message).
    const int log1 = log_n1 > 0 ? log_n1 : (log2n <= 16 ? (log2n + 1) / 2 : 8);
    if (log1 < 1 || log1 >= log2n || log1 > 18 || log2n - log1 > 18) return fail(SC_ERR_BAD_ARG, "unsupported split of the four-step transform");
    sc_fourstep* p = new sc_fourstep;
    p->log2n = log2n; p->rank = rank; p->world = world; p->n = n;
    p->n1 = 1ull << log1; p->n2 = n >> log1;
    if (p->n1 < (uint64_t)world || p->n2 < (uint64_t)world) { delete p; return fail(SC_ERR_BAD_ARG, "domain too small to shard over this many ranks"); }
    const Fe rinv = root_inverse(rt, n);
    for (int dirn = 0; dirn < 2; ++dirn) {
        sc_fourstep::Dir& d = p->dir[dirn];
        d.R = dirn == 0 ? p->n1 : p->n2;
        d.C = dirn == 0 ? p->n2 : p->n1;
        d.root = dirn == 0 ? rt : rinv;
        const Fe rm = to_mont(d.root);
        d.root_cols = from_mont(mont_pow(rm, d.C));
        d.root_rows = from_mont(mont_pow(rm, d.R));
        d.ninv = dirn == 1;
    }
    *out = p;
    return SC_OK;
}
int sc_fourstep_free(sc_fourstep_t* plan) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (plan && plan->timed_out) { (void)hipDeviceSynchronize(); (void)hipHostFree((void*)plan->timed_out); }
    delete plan;
    return SC_OK;
}
int sc_fourstep_shape(const sc_fourstep_t* plan, int inverse, uint64_t* rows, uint64_t* cols_total) {
    if (!plan) return fail(SC_ERR_BAD_ARG, "null plan");
    const sc_fourstep::Dir& d = plan->dir[inverse ? 1 : 0];
    if (rows) *rows = d.R;
    if (cols_total) *cols_total = d.C;
    return SC_OK;
}
int sc_fourstep_cols_dev(const sc_fourstep_t* plan, int inverse, const void* d_src, void* d_send, void* d_recv_diag, void* stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    if (!plan || !d_src || !d_send) return fail(SC_ERR_BAD_ARG, "null argument");
    return fourstep_cols(plan, inverse ? 1 : 0, (const Fe*)d_src, (Fe*)d_send, (Fe*)d_recv_diag, pick_stream(stream));
}
int sc_fourstep_rows_dev(const sc_fourstep_t* plan, int inverse, const void* d_recv, void* d_dst, uint64_t block, uint64_t nblocks, int defer_last_pass, void* stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    if (!plan || !d_recv || !d_dst) return fail(SC_ERR_BAD_ARG, "null argument");
    return fourstep_rows(plan, inverse ? 1 : 0, (const Fe*)d_recv, (Fe*)d_dst, block, nblocks ? nblocks : 1, defer_last_pass != 0, pick_stream(stream));
}
int sc_fourstep_rows_finish_dev(const sc_fourstep_t* plan, int inverse, void* d_dst, void* stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    if (!plan || !d_dst) return fail(SC_ERR_BAD_ARG, "null argument");
    const sc_fourstep::Dir& d = plan->dir[inverse ? 1 : 0];
    if (batched_passes(ilog2(d.C)) == 1) return SC_OK;     // single-pass rows: nothing was deferred
    return fourstep_rows_finish(plan, inverse ? 1 : 0, (Fe*)d_dst, pick_stream(stream));
}

// ---- direct-store corner turn: peers' receive buffers mapped through HIP IPC, the column stage stores across xGMI itself
struct IpcFlags {
    uint64_t* of[SC_MAX_BLOCKS];        // of[h]: rank h's flag array (entry g = the last epoch rank g has finished writing)
    volatile uint64_t* host_timed_out;  // this rank's pinned status word
    uint64_t spin_limit;                // polls of s_sleep 16 before the wait gives up (2^22: about two seconds)
};

// One workgroup, one lane per peer: tell peer t that this rank's column stage of transform `epoch` is complete (the stage is the
// PREVIOUS kernel on this stream: its stores are released at its end; the fence below orders the flag behind them once more),
// then wait until every peer has said the same to this rank.  System-scope atomics: the flags live in other GPUs' memory.
// A peer that never arrives (a crashed rank) must not hang the device for ever: after ~2 s the wait gives up and marks the
// region (flag word SC_MAX_BLOCKS), the transform's output is then garbage and the caller's checks see it.
__global__ void __launch_bounds__(64) ipc_barrier_kernel(IpcFlags flags, int rank, int world, uint64_t epoch) {
    const int t = threadIdx.x;
    __threadfence_system();
    if (t < world && t != rank) {
        __hip_atomic_store(&flags.of[t][rank], epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        uint64_t spins = 0;
        while (__hip_atomic_load(&flags.of[rank][t], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < epoch) {
            __builtin_amdgcn_s_sleep(16);
            if (++spins > flags.spin_limit) {
                __hip_atomic_store(&flags.of[rank][SC_MAX_BLOCKS], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (flags.host_timed_out && *flags.host_timed_out == 0) *flags.host_timed_out = epoch;
                break;
            }
        }
    }
    __threadfence_system();
}

static int g_ipc_fine_grained = -1;      // the kind of the last region created: 1 fine-grained, 0 coarse-grained, -1 none yet
int sc_ipc_region_create(uint64_t bytes, void** d_region, uint8_t handle_out[64]) { return sc_ipc_region_create_ex(bytes, -1, d_region, handle_out); }
// kind: 1 fine-grained (coarse-grained if the runtime cannot export one), 0 coarse-grained, -1 the default (fine-grained unless
// STARKCORE_IPC_COARSE=1).  STARKCORE_TEST_FINE_EXPORT_FAILS=1 (tests) makes the fine-grained export fail as a runtime without it would.
int sc_ipc_region_create_ex(uint64_t bytes, int kind, void** d_region, uint8_t handle_out[64]) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    if (!bytes || !d_region || !handle_out) return fail(SC_ERR_BAD_ARG, "null argument");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "HIP IPC handles are 64 bytes");
    // FINE-GRAINED device memory: other GPUs store into this region and raise flags in it WHILE kernels of this GPU poll and read
    // it.  Coarse-grained memory (plain hipMalloc) is only promised coherent between agents at kernel boundaries -- this GPU's L2
    // may keep serving a line a peer has rewritten -- which is why RCCL allocates the buffers its peers write into the same way.
    // STARKCORE_IPC_COARSE=1 selects plain hipMalloc (for an A/B on a node with several GPUs); a runtime that cannot export a
    // fine-grained allocation falls back to it as well.  sc_ipc_region_kind() tells which one the last region got.
    const char* coarse_env = getenv("STARKCORE_IPC_COARSE");
    const char* fail_env = getenv("STARKCORE_TEST_FINE_EXPORT_FAILS");
    const bool want_fine = kind < 0 ? !(coarse_env && coarse_env[0] == '1') : kind == 1;
    void* p = nullptr;
    hipIpcMemHandle_t h;
    hipError_t e = hipErrorUnknown;
    if (want_fine) {
        e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained);
        if (e == hipSuccess) e = hipMemset(p, 0, bytes);
        if (e == hipSuccess) e = (fail_env && fail_env[0] == '1') ? hipErrorInvalidValue : hipIpcGetMemHandle(&h, p);
        if (e != hipSuccess) { if (p) (void)hipFree(p); p = nullptr; (void)hipGetLastError(); }
        else g_ipc_fine_grained = 1;
    }
    if (!p) {
        HIPCHK(hipMalloc(&p, bytes));
        e = hipMemset(p, 0, bytes);
        if (e == hipSuccess) e = hipIpcGetMemHandle(&h, p);
        if (e != hipSuccess) { (void)hipFree(p); return fail(SC_ERR_HIP, std::string("hipIpcGetMemHandle: ") + hipGetErrorString(e)); }
        g_ipc_fine_grained = 0;
    }
    memcpy(handle_out, &h, 64);
    *d_region = p;
    return SC_OK;
}
int sc_ipc_region_kind(int* fine_grained) {
    if (!fine_grained) return fail(SC_ERR_BAD_ARG, "null argument");
    *fine_grained = g_ipc_fine_grained;
    return SC_OK;
}
int sc_ipc_region_open(const uint8_t handle[64], void** d_region) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    if (!handle || !d_region) return fail(SC_ERR_BAD_ARG, "null argument");
    hipIpcMemHandle_t h;
    memcpy(&h, handle, 64);
    void* p = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(SC_ERR_HIP, std::string("hipIpcOpenMemHandle: ") + hipGetErrorString(e)); }
    *d_region = p;
    return SC_OK;
}
int sc_ipc_region_close(void* d_region) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!d_region) return SC_OK;
    (void)hipDeviceSynchronize();
    HIPCHK(hipIpcCloseMemHandle(d_region));
    return SC_OK;
}
int sc_ipc_region_free(void* d_region) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!d_region) return SC_OK;
    (void)hipDeviceSynchronize();
    HIPCHK(hipFree(d_region));
    return SC_OK;
}
int sc_fourstep_region_bytes(const sc_fourstep_t* plan, uint64_t* bytes) {
    if (!plan || !bytes) return fail(SC_ERR_BAD_ARG, "null argument");
    *bytes = FOURSTEP_FLAG_BYTES + 2 * (plan->n / (uint64_t)plan->world) * sizeof(Fe);
    return SC_OK;
}
int sc_fourstep_set_peers(sc_fourstep_t* plan, void* const* regions) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!plan || !regions) return fail(SC_ERR_BAD_ARG, "null argument");
    for (int h = 0; h < plan->world; ++h) {
        if (!regions[h]) return fail(SC_ERR_BAD_ARG, "a rank's region is missing");
        plan->region[h] = (uint8_t*)regions[h];
    }
    if (!plan->timed_out) {
        uint64_t* w = nullptr;
        HIPCHK(hipHostMalloc((void**)&w, 64, hipHostMallocCoherent | hipHostMallocMapped));
        plan->timed_out = w;
    }
    *plan->timed_out = 0;
    plan->peers_set = true;
    plan->epoch = 0;
    return SC_OK;
}
// the whole transform in the direct-store form: column stage (block h stored straight into rank h's receive buffer, own block
// included), flag barrier, row stage out of this rank's receive buffer.  Collective in the sense that every rank must call it for
// the same transforms in the same order; asynchronous on `stream`.
int sc_fourstep_run_direct_dev(sc_fourstep_t* plan, int inverse, const void* d_src, void* d_dst, void* stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    if (!plan || !d_src || !d_dst) return fail(SC_ERR_BAD_ARG, "null argument");
    if (!plan->peers_set) return fail(SC_ERR_BAD_ARG, "sc_fourstep_set_peers has not been called");
    if (plan->timed_out && *plan->timed_out)
        return fail(SC_ERR_TIMEOUT, "a peer never arrived at the flag barrier of direct-store transform " + std::to_string((unsigned long long)*plan->timed_out) +
                                    ": that transform's output is invalid and the receive buffers are out of step -- set the plan's peers up again (or use the collective exchange)");
    hipStream_t st = pick_stream(stream);
    const int dirn = inverse ? 1 : 0;
    const sc_fourstep::Dir& d = plan->dir[dirn];
    const uint64_t G = (uint64_t)plan->world, g_ = (uint64_t)plan->rank, rw = d.R / G, cw = d.C / G, blk = rw * cw;
    const uint64_t per_rank = plan->n / G;
    const uint64_t parity = plan->epoch & 1;
    auto recv_of = [&](uint64_t h) { return (Fe*)(plan->region[h] + FOURSTEP_FLAG_BYTES) + parity * per_rank; };
    Fe* table[SC_MAX_BLOCKS];
    for (uint64_t h = 0; h < G; ++h) table[h] = recv_of(h) + (int64_t)(g_ - h) * (int64_t)blk;     // element j = h * blk + ... lands in block g_ of rank h
    BatchCall c;
    c.in = (const Fe*)d_src; c.out = recv_of(g_); c.len = d.R; c.batch = cw; c.kind = 0; c.root = d.root_cols;
    c.outer = true; c.outer_root = d.root; c.outer_order = plan->n; c.outer_col_base = g_ * cw; c.outer_ninv = d.ninv;
    c.block_out = table; c.block_rows = (uint32_t)rw;
    c.roots_checked = true;
    SCCHK(batch_call(c, st));
    if (G > 1) {
        IpcFlags fl;
        for (uint64_t h = 0; h < SC_MAX_BLOCKS; ++h) fl.of[h] = h < G ? (uint64_t*)plan->region[h] : nullptr;
        fl.host_timed_out = plan->timed_out;
        static const char* spin_env = getenv("STARKCORE_IPC_BARRIER_SPINS");            // (tests shorten the two seconds)
        fl.spin_limit = spin_env && atoll(spin_env) > 0 ? (uint64_t)atoll(spin_env) : (1ull << 22);
        hipLaunchKernelGGL(ipc_barrier_kernel, dim3(1), dim3(64), 0, st, fl, plan->rank, plan->world, plan->epoch + 1);
        HIPCHK(hipGetLastError());
    }
    SCCHK(fourstep_rows(plan, dirn, recv_of(g_), (Fe*)d_dst, 0, 1, false, st));
    plan->epoch += 1;
    return SC_OK;
}
// 0 while every flag barrier of this plan has completed; the epoch of the first one that gave up waiting otherwise (read after
// the stream has been synchronised)
int sc_fourstep_direct_status(const sc_fourstep_t* plan, uint64_t* timed_out_epoch) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!plan || !timed_out_epoch || !plan->peers_set) return fail(SC_ERR_BAD_ARG, "no direct-store set-up");
    *timed_out_epoch = plan->timed_out ? *plan->timed_out : 0;        // (a pinned word the barrier kernel writes: no copy, no wait)
    return SC_OK;
}

// ---- native RCCL communicator (one per process) for the corner turn of sc_fourstep_run_dev
int sc_comm_unique_id(const char* rccl_path, sc_rccl_id_t* id_out) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    if (!id_out) return fail(SC_ERR_BAD_ARG, "null argument");
    SCCHK(rccl_load(rccl_path));
    RCCLCHK(rccl.GetUniqueId(id_out));
    return SC_OK;
}
int sc_comm_init(const char* rccl_path, const sc_rccl_id_t* id, int rank, int world) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    if (!id || world < 1 || rank < 0 || rank >= world) return fail(SC_ERR_BAD_ARG, "bad communicator shape");
    if (g_comm) return (rank == g_comm_rank && world == g_comm_world) ? SC_OK : fail(SC_ERR_BAD_ARG, "a communicator of another shape exists already");
    SCCHK(rccl_load(rccl_path));
    void* comm = nullptr;
    RCCLCHK(rccl.CommInitRank(&comm, world, *id, rank));
    g_comm = comm; g_comm_rank = rank; g_comm_world = world;
    return SC_OK;
}
int sc_comm_destroy(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_comm) return SC_OK;
    (void)hipDeviceSynchronize();
    for (hipEvent_t e : g_comm_events) (void)hipEventDestroy(e);
    g_comm_events.clear();
    if (g_comm_stream) { (void)hipStreamDestroy(g_comm_stream); g_comm_stream = nullptr; g_comm_stream_for_free = nullptr; }
    int r = rccl.CommDestroy(g_comm);
    g_comm = nullptr; g_comm_rank = -1; g_comm_world = 0;
    if (r != 0) return fail(SC_ERR_HIP, "ncclCommDestroy failed");
    return SC_OK;
}

// the whole transform in one call: column stage, corner turn over the native communicator, row stage.  nblocks == 1: everything
// on `stream`, in order.  nblocks > 1: the exchange is issued as that many row blocks on the library's communication stream and
// the row transforms of block q start as soon as it has landed, while blocks q+1.. are still on the wire.
// force_diag_exchange != 0 (tests): the rank's own block travels through RCCL as well (send to self).
int sc_fourstep_run_dev(const sc_fourstep_t* plan, int inverse, const void* d_src, void* d_send, void* d_recv, void* d_dst, uint64_t nblocks, int defer_last_pass,
                        int force_diag_exchange, void* stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    SCCHK(ensure_init());
    if (!plan || !d_src || !d_send || !d_recv || !d_dst) return fail(SC_ERR_BAD_ARG, "null argument");
    const int dirn = inverse ? 1 : 0;
    const sc_fourstep::Dir& d = plan->dir[dirn];
    const uint64_t G = (uint64_t)plan->world, rw = d.R / G, cw = d.C / G;
    const uint64_t K = nblocks ? nblocks : 1;
    if (rw % K || !is_pow2(rw / K)) return fail(SC_ERR_BAD_ARG, "row blocks must divide the rank's rows into powers of two");
    const bool exchange = G > 1 || force_diag_exchange;
    if (exchange && (!g_comm || g_comm_world != plan->world || g_comm_rank != plan->rank)) return fail(SC_ERR_NOT_INIT, "sc_comm_init has not been called for this world");
    hipStream_t st = pick_stream(stream);
    const Fe* send = (const Fe*)d_send;
    Fe* recv = (Fe*)d_recv;
    SCCHK(fourstep_cols(plan, dirn, (const Fe*)d_src, (Fe*)d_send, force_diag_exchange ? nullptr : recv, st));
    auto self_block = [&](uint64_t row0, uint64_t nrows, hipStream_t s) -> int {
        if (!force_diag_exchange) return SC_OK;
        const uint64_t off = ((uint64_t)plan->rank * rw + row0) * cw;
        RCCLCHK(rccl.GroupStart());
        RCCLCHK(rccl.Send(send + off, nrows * cw * sizeof(Fe), 1, plan->rank, g_comm, s));
        RCCLCHK(rccl.Recv(recv + off, nrows * cw * sizeof(Fe), 1, plan->rank, g_comm, s));
        RCCLCHK(rccl.GroupEnd());
        return SC_OK;
    };
    if (!exchange || K == 1) {
        if (exchange) {
            if (G > 1) SCCHK(rccl_exchange(plan, dirn, send, recv, 0, rw, st));
            SCCHK(self_block(0, rw, st));
        }
        return fourstep_rows(plan, dirn, recv, (Fe*)d_dst, 0, 1, false, st);
    }
    if (!g_comm_stream) { HIPCHK(hipStreamCreateWithFlags(&g_comm_stream, hipStreamNonBlocking)); g_comm_stream_for_free = g_comm_stream; }
    while (g_comm_events.size() < 1 + K) { hipEvent_t e; HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); g_comm_events.push_back(e); }
    HIPCHK(hipEventRecord(g_comm_events[0], st));
    HIPCHK(hipStreamWaitEvent(g_comm_stream, g_comm_events[0], 0));
    const uint64_t rk = rw / K;
    for (uint64_t q = 0; q < K; ++q) {
        if (G > 1) SCCHK(rccl_exchange(plan, dirn, send, recv, q * rk, rk, g_comm_stream));
        SCCHK(self_block(q * rk, rk, g_comm_stream));
        HIPCHK(hipEventRecord(g_comm_events[1 + q], g_comm_stream));
    }
    for (uint64_t q = 0; q < K; ++q) {
        HIPCHK(hipStreamWaitEvent(st, g_comm_events[1 + q], 0));
        SCCHK(fourstep_rows(plan, dirn, recv, (Fe*)d_dst, q, K, defer_last_pass != 0, st));
    }
    if (defer_last_pass && batched_passes(ilog2(d.C)) == 2) SCCHK(fourstep_rows_finish(plan, dirn, (Fe*)d_dst, st));
    return SC_OK;
}

}  // extern "C"
