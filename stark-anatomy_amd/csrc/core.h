// core.h -- what the translation units of libstarkcore.so share: the host state (one context per process), the device-memory pool,
// error plumbing, the table caches and the entry points of the transform planner.  Nothing here is part of the C ABI
// (include/starkcore.h is); everything lives in namespace sci and is hidden from the shared object's export table.
//
//   core.hip          state, pool, streams, tables, planner + pass launches, vectors / randomness / transforms / pointwise entries
//   merkle_fri.hip    Merkle trees, the FRI fold and commit loop, the Fiat-Shamir transcript, openings
//   polytree_geo.hip  subproduct trees and geometric progressions (fast_zerofier / fast_evaluate / fast_interpolate)
//   fourstep.hip      batched transforms, the sharded four-step plan, the RCCL communicator, the direct-store corner turn
#pragma once
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cerrno>
#include <thread>
#include <sys/random.h>
#include <deque>
#include <functional>
#include <future>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#pragma GCC visibility push(default)
#include "../../include/starkcore.h"
#pragma GCC visibility pop
#include "ntt_plan.h"

namespace sc { struct FoldIn; }          // merkle.cuh (its kernels belong to merkle_fri.hip alone)
using namespace sc;

struct sc_vec {
    Fe* d;
    uint64_t n;
    bool owned = true;    // false: a view of memory somebody else owns (sc_vec_wrap: a torch tensor's storage) -- sc_vec_free leaves it alone
};
struct sc_later {           // a few words a kernel will write to a pinned slot (deferred checks: include/starkcore.h sc_later_t)
    int slot;
    uint64_t seq;
    hipStream_t st;
};
struct sc_merkle {
    uint64_t* d_levels;   // (2N-1) digests of 8 x u64
    uint64_t N;
    int logN;
    // asynchronous builds (sc_merkle_build_async_dev, sc_fri_fold_commit_dev): the root is on its way to a pinned host slot;
    // sc_merkle_root waits for `st` once and moves it to `root`
    int slot = -1;
    uint64_t seq = 0;
    hipStream_t st = nullptr;
    bool have_root = false;
    bool lazy = false;    // built "enqueue only" (BUILD_NOROOT): no slot, no publish kernel; the root is copied out if ever asked for
    uint8_t root[64] = {};
};

namespace sci {

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

struct PlanKey {
    int logn;
    uint64_t lo, hi;
    bool operator<(const PlanKey& o) const { return std::tie(logn, lo, hi) < std::tie(o.logn, o.lo, o.hi); }
};
struct PlanTables {      // power tables of one root of order n = 2^logn
    Fe* mt = nullptr;
    int mt_log = 0;
    Fe* tl = nullptr;
    Fe* th = nullptr;
    Fe* th_ninv = nullptr;   // th * n^-1 (built on first inverse use)
    // direct four-step twiddle tables per column pass for the plan's digit split, [0]: plain, [1]: first pass scaled by n^-1
    Fe* twd[2][4] = {{nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr}};
    int twd_digits[4] = {0, 0, 0, 0};
    int twd_passes = 0;
    // direct inter-pass table of the two-pass BATCHED plans of this length (first digit twd_b_digit0)
    Fe* twd_b = nullptr;
    int twd_b_digit0 = 0;
    uint64_t last_use = 0;   // lookup tick (eviction order; see evict_tables)
};
struct OuterKey {        // direct outer-twiddle table of one rank's slab (multi-GPU column stage)
    uint64_t lo, hi, order, len, batch, col_base;
    int ninv;
    bool operator<(const OuterKey& o) const { return std::tie(lo, hi, order, len, batch, col_base, ninv) < std::tie(o.lo, o.hi, o.order, o.len, o.batch, o.col_base, o.ninv); }
};
struct OuterTable {
    Fe* d = nullptr;
    uint64_t last_use = 0;
};
struct PowKey {
    uint64_t lo, hi, hi_count;
    bool operator<(const PowKey& o) const { return std::tie(lo, hi, hi_count) < std::tie(o.lo, o.hi, o.hi_count); }
};
struct PowTables {
    Fe* lo = nullptr;
    Fe* hi = nullptr;
    uint64_t last_use = 0;
};

struct Ctx {
    bool init = false;
    int device = -1;
    hipStream_t stream = nullptr;
    std::string err;
    NttTuning tuning;
    std::map<PlanKey, PlanTables> plans;
    std::map<PowKey, PowTables> pows;
    std::map<OuterKey, OuterTable> outers;
    uint64_t tick = 0;       // bumped by every table lookup
    bool foreign_streams = false;   // a caller-owned stream has been used (see pick_stream)
    int small_divisor_direct = 1;    // coset_divide_core evaluates a divisor of <= 8 coefficients point by point instead of transforming it (sc_set_tuning("small_divisor_direct", 0): A/B and tests)
    std::vector<hipStream_t> seen_streams;   // those streams, most recent last (at most SEEN_STREAMS; more: device-wide waits)
    DevBuf scratch[8];       // 0: ntt work, 1..3: poly temporaries, 4: misc small, 5: merkle staging, 6: uploaded operands, 7: degree / exactness flag
    std::map<hipStream_t, DevBuf> ntt_work;   // the work buffer of a multi-pass transform, one per stream: transforms on DIFFERENT streams may be in flight together
    int num_cus = 256;
    int xcd_remap = 1;
    int fixed_shapes = 1;    // use the geometry-specialised kernel instantiations where one matches
    int prio_balance = -1;   // -1: 1 for launches of at most one workgroup per CU, 2 for long grids (>= 8 workgroups per CU; >= 2 of 2^10 x 4 tiles), else 0; 0 / 1 / 2: force (PassParams::prio_balance)
    int wave_local = 1;      // wave-level fences instead of workgroup barriers once a tile's exchanges stay inside one wave
    unsigned long long* trace = nullptr;   // diagnostics: phase stamps of the next fixed-shape pass launches (sc_debug_trace)
    int merkle_big_nlev = 2; // levels fused per launch for Merkle levels wider than FUSE_MAX_W (0: one level kernel per level)
    int fri_tail = -1;       // the persistent tail kernel of Fri.commit (csrc/fri_tail.cuh): -1 = environment STARKCORE_FRI_TAIL (default on), 0 / 1
    int fri_tail_stall = -1; // tests: the host withholds the challenge of this round of the tail kernel (its wait then times out: the abort path)
    uint8_t* root_slots = nullptr;        // pinned host memory: roots of asynchronously built Merkle trees in flight
    uint64_t root_seq = 0;
    std::vector<int> free_root_slots;
};
constexpr int ROOT_SLOTS = 256;
constexpr size_t ROOT_SLOT_BYTES = 128;   // 64-byte root, then the 8-byte sequence number that says it has landed
constexpr long SPIN_POLLS = 40000000;     // ~ tens of milliseconds of polling before the blocking wait

// Frees never wait.  A buffer handed back while a stream may still be using it (the *_dev entries take raw device pointers on
// caller streams, so the library cannot know which) is parked with one event per stream in use -- recorded at the moment of the
// free, i.e. behind everything enqueued so far -- and returns to the pool once those events have completed; that is checked
// when the next buffer is allocated or freed (a query per event, no blocking).
struct PendingFree {
    void* p;
    size_t bytes;
    std::vector<hipEvent_t> evs;
};
#define HIPCHK(expr)                                                                                          \
    do {                                                                                                      \
        hipError_t _e = (expr);                                                                               \
        if (_e != hipSuccess) return fail(SC_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));     \
    } while (0)

#define SCCHK(expr)              \
    do {                         \
        int _rc = (expr);        \
        if (_rc != SC_OK) return _rc; \
    } while (0)

struct NttOpts {
    uint64_t in_limit = ~0ull;
    const PowTables* coset = nullptr;
    uint32_t cols = 1;       // independent transforms in one set of launches, column c at element c * n of in / out (NttIo::cols)
    uint64_t col_stride_in = 0;   // ... of `in` at c * col_stride_in instead (0 = n)
};

// Climb from level `lvl` (already in the tree, `N >> lvl` nodes) to the root.  Levels wider than FUSE_MAX_W nodes are
// throughput-bound: launches that fuse merkle_big_nlev (2) levels -- a workgroup's 256 -> 128 -> 64 nodes keep every active
// wave full, and the intermediate levels are not re-read from HBM.  Below FUSE_MAX_W the chain of dependent launches is pure
// latency: fused 8-level subtree launches (most threads idle on the upper levels, which is fine there), then the
// one-workgroup tail.  Measured (tools/merkle_timing.py, profiles/r01/merkle_timing.txt): at 2^24 leaves 2 fused levels
// 2.09 ms, 1 level per launch 2.46 ms, 4 levels 2.50 ms, 8 levels 4.5 ms.
constexpr uint64_t FUSE_MAX_W = 1ull << 17;

enum BuildMode { BUILD_SYNC = 0, BUILD_ASYNC = 1, BUILD_NOROOT = 2 };

// ---- shared host state (defined in core.hip)
extern Ctx g;
extern std::mutex g_mu;
extern hipStream_t g_comm_stream_for_free;      // the library's communication stream once it exists (sc_fourstep_run_dev)
extern std::future<void> g_rand_worker;         // a draw of kernel randomness started ahead of time (sc_urandom_prefetch) ...
extern size_t g_rand_prefetched;                // ... and its size in bytes (0: none in flight)
extern std::multimap<size_t, void*> g_pool;
extern size_t g_pool_bytes, g_pool_cap;
extern std::deque<PendingFree> g_pending;
extern std::vector<hipEvent_t> g_event_pool;

// ---- core.hip: pool, errors, streams, tables, the transform planner
void reap_pending(bool block);
hipError_t pool_alloc(void** p, size_t bytes);
void pool_free(void* p, size_t bytes);
void pool_clear();
hipEvent_t event_get();
int fail(int code, const std::string& msg);
int ensure_init();
int scratch(int slot, size_t bytes, void** out);
int ntt_work_buffer(hipStream_t st, size_t bytes, void** out);
int check_root(Fe root, uint64_t n);
int build_pow_table(Fe** out, uint64_t count, Fe base_m, uint64_t step, Fe scale_m, hipStream_t st);
void free_plans();
int evict_outer_tables();
int get_plan(Fe root, int logn, bool need_ninv, hipStream_t st, PlanTables** out);
int get_pow(Fe base, uint64_t count, hipStream_t st, PowTables** out);
int plan_batched_direct(NttPlanDesc& d, BatchKind kind, int loglen, int logbatch, PlanTables* pt, const Fe* in, Fe* work, Fe* out, BatchExtras ex, hipStream_t st, bool* ok);
int run_plan(NttPlanDesc& d, hipStream_t st);
int ntt_device(const Fe* d_in, Fe* d_out, int logn, Fe root, bool inverse_scale, const NttOpts& o, hipStream_t st);
Fe root_inverse(Fe root, uint64_t n);
int ntt_any(const Fe* d_in, Fe* d_out, uint64_t n, Fe root, bool inverse, const NttOpts& o, hipStream_t st);
int upload(void* d, const void* h, size_t bytes, hipStream_t st);
int download(void* h, const void* d, size_t bytes, hipStream_t st);
int pointwise_div_device(const Fe* a, const Fe* b, Fe* out, uint64_t n, hipStream_t st);
int pointwise_div_enqueue(const Fe* a, const Fe* b, Fe* out, uint64_t n, hipStream_t st, uint32_t** flag_dev);
int gather_device(const Fe* v, const uint64_t* d_idx, uint64_t k, Fe* d_out, hipStream_t st);
int read_small_polled(const void* d_src, size_t bytes, hipStream_t st, void* host_out);

// ---- merkle_fri.hip
int merkle_climb(uint64_t* levels, uint64_t N, int lvl, hipStream_t st, volatile uint64_t* host = nullptr, uint64_t seq = 0, bool* published = nullptr);
int merkle_finish(uint64_t* levels, uint64_t width, hipStream_t st);
int root_slot_get();
int merkle_build_device(const Fe* d_elems, uint64_t N, uint8_t root_out[64], sc_merkle** tree, hipStream_t st, BuildMode mode = BUILD_SYNC, const FoldIn* fold = nullptr);
int merkle_root_wait(sc_merkle* t, bool from_free = false);
void root_poll_unlocked(std::unique_lock<std::mutex>& lk, const sc_merkle* t);
int fold_prepare(const Fe* d_in, uint64_t N, Fe alpha, Fe offset, Fe omega, Fe* d_out, hipStream_t st, FoldIn* f);
int fold_device(const Fe* d_in, uint64_t N, Fe alpha, Fe offset, Fe omega, Fe* d_out, hipStream_t st);
int fold_and_build(const Fe* d_in, uint64_t N, Fe alpha, Fe offset, Fe omega, Fe* d_out, sc_merkle** tree, hipStream_t st);

// ---- polytree_geo.hip
Fe canonical_root(int logn);
Fe ninv_scaled(int logn, int j);
int ntt_cols(const Fe* in, Fe* out, int loglen, int logbatch, bool inverse, hipStream_t st);

// A caller stream (the *_dev entries take one; sharded.py passes torch's) may still be reading a pooled buffer when it is
// freed: once any foreign stream has been seen, frees wait for the whole device instead of the library stream only.
constexpr size_t SEEN_STREAMS = 8;
inline hipStream_t pick_stream(void* s) {
    if (s && (hipStream_t)s != g.stream) {
        g.foreign_streams = true;
        hipStream_t st = (hipStream_t)s;
        if (g.seen_streams.size() <= SEEN_STREAMS && std::find(g.seen_streams.begin(), g.seen_streams.end(), st) == g.seen_streams.end())
            g.seen_streams.push_back(st);
    }
    return s ? (hipStream_t)s : g.stream;
}
// A buffer goes back to the pool when nothing can still be using it: an event is recorded on the library stream and on every
// caller stream seen so far, and the buffer is parked until they have completed (reap_pending).  A stream that no longer takes
// an event (destroyed by its owner: its work is done) is forgotten; more streams than are tracked: wait for the whole device.
inline void release_after_streams(void* p, size_t bytes) {
    if (!p) return;
    if (g.seen_streams.size() > SEEN_STREAMS) {
        (void)hipDeviceSynchronize();
        g.seen_streams.clear();
        pool_free(p, bytes);
        return;
    }
    PendingFree f{p, bytes, {}};
    auto mark = [&](hipStream_t st) -> bool {
        if (hipStreamQuery(st) == hipSuccess) return true;          // idle: nothing of it can still touch the buffer
        (void)hipGetLastError();
        hipEvent_t e = event_get();
        if (!e) { (void)hipStreamSynchronize(st); (void)hipGetLastError(); return true; }
        if (hipEventRecord(e, st) != hipSuccess) { (void)hipGetLastError(); g_event_pool.push_back(e); return false; }
        f.evs.push_back(e);
        return true;
    };
    if (g.stream) (void)mark(g.stream);
    if (g_comm_stream_for_free) (void)mark(g_comm_stream_for_free);
    for (size_t i = 0; i < g.seen_streams.size();) {
        if (mark(g.seen_streams[i])) ++i;
        else g.seen_streams.erase(g.seen_streams.begin() + i);      // stale handle
    }
    if (f.evs.empty()) { pool_free(p, bytes); return; }
    g_pending.push_back(std::move(f));
    reap_pending(false);
}
inline Fe fe_from(const uint64_t v[2]) { return Fe{v[0], v[1]}; }
inline bool is_pow2(uint64_t n) { return n && !(n & (n - 1)); }
inline int ilog2(uint64_t n) { int l = 0; while ((1ull << l) < n) ++l; return l; }

// small RAII holder for pool temporaries
struct PoolTmp {
    void* p = nullptr;
    size_t bytes = 0;
    ~PoolTmp() { if (p) pool_free(p, bytes); }
    int get(size_t b) {
        bytes = b;
        HIPCHK(pool_alloc(&p, b));
        return SC_OK;
    }
    Fe* fe() const { return (Fe*)p; }
};

// pool temporary whose memory goes back once the streams that may still read it have passed this point (nothing here waits)
struct PoolTmpAsync {
    void* p = nullptr;
    size_t bytes = 0;
    ~PoolTmpAsync() { if (p) release_after_streams(p, bytes); }
    int get(size_t b) {
        bytes = b ? b : sizeof(Fe);
        HIPCHK(pool_alloc(&p, bytes));
        return SC_OK;
    }
    Fe* fe() const { return (Fe*)p; }
};

}  // namespace sci
using namespace sci;
