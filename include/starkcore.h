/*
 * starkcore.h -- C ABI of the MI355X-native STARK polynomial core (libstarkcore.so).
 *
 * The reference (aszepieniec/stark-anatomy, pure Python) has no FFI; its boundary for this path is
 * the set of module-level callables in code/ntt.py, the fold expression in code/fri.py:85 and
 * code/merkle.py.  Each entry point below names the reference callable it replaces (file:line);
 * INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - element = 16 bytes, two little-endian uint64 limbs (lo, hi), CANONICAL residue in [0, p),
 *     p = 1 + 407 * 2^119 (code/algebra.py:96-98); arrays contiguous, natural index order.
 *   - every function returns 0 on success or a negative SC_ERR_* code; sc_last_error() gives text.
 *     Nothing throws across the ABI; host-buffer calls block until the output buffer is valid.
 *   - host-buffer entry points (sc_ntt, ...) take caller-owned host memory.
 *   - *_dev entry points take device pointers (hipMalloc / torch tensor storage, 16-byte aligned) and
 *     a hipStream_t (NULL = the library's own stream) and are asynchronous on that stream.
 *   - sc_vec_t / sc_merkle_t are library-owned device objects behind opaque handles.
 *   - one context per PROCESS: sc_init(device) binds the process to one GPU (one process per GPU is the multi-GPU model,
 *     stark-anatomy_amd/sharded.py + torch.distributed/RCCL; there is no sc_init(ndev) -- a rank's share of the sharded
 *     transform is the sc_fourstep_* plan object below, driven by sharded.ShardedNtt).
 *   - streams: the library keeps a few scratch buffers (transform work space, temporaries) that every call reuses.  Calls
 *     that pass the SAME stream (or NULL) are ordered by that stream and need nothing else.  Calls on DIFFERENT streams must
 *     not overlap in time: order them with events, or synchronize, before switching streams (sharded.py does).  Frees of
 *     library objects never wait: the memory is parked behind an event on every stream in use (the library's and the caller
 *     streams seen so far) and goes back to the library's pool when those have completed.
 */
#ifndef STARKCORE_H
#define STARKCORE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SC_OK 0
#define SC_ERR_HIP (-1)              /* HIP runtime error (no GPU, OOM, launch failure) */
#define SC_ERR_NOT_POW2 (-2)         /* ntt.py:4 / merkle.py:7 "length must be power of two" */
#define SC_ERR_ROOT_ORDER (-3)       /* ntt.py:10 root^n != 1 */
#define SC_ERR_ROOT_NOT_PRIMITIVE (-4) /* ntt.py:11 root^(n/2) == 1 */
#define SC_ERR_DIV_ZERO (-5)         /* algebra.py:92 "divide by zero" */
#define SC_ERR_BAD_ARG (-6)
#define SC_ERR_UNSUPPORTED (-7)
#define SC_ERR_NOT_INIT (-8)
#define SC_ERR_TIMEOUT (-9)          /* a peer never arrived at a flag barrier of the direct-store corner turn: sticky until the plan's peers are set again */

typedef struct sc_vec sc_vec_t;        /* device-resident vector of field elements */
typedef struct sc_merkle sc_merkle_t;  /* device-resident BLAKE2b Merkle tree (all levels kept) */
typedef struct sc_later sc_later_t;    /* a check whose words arrive behind the work that produces them (sc_*_later_dev, sc_later_wait) */
typedef struct sc_polytree sc_polytree_t; /* device-resident subproduct tree over a list of points (all levels kept) */
typedef struct sc_geodomain sc_geodomain_t; /* tables of a domain that is a geometric progression first * ratio^i */

/* ---- lifecycle ---------------------------------------------------------------------------- */
int sc_device_count(void);
int sc_init(int device);               /* idempotent; selects the device and creates the library stream */
int sc_shutdown(void);                 /* frees plans, scratch and the stream */
const char* sc_last_error(void);
int sc_synchronize(void);              /* wait for the library stream */
/* The library stream as a raw hipStream_t, for callers that put their other device work on the same stream (no ordering needed
 * then), and a two-way, event-based ordering of the library stream with another stream that never blocks the host. */
int sc_stream(void** stream_out);
int sc_stream_join(void* other_stream);
/* tuning knobs for experiments (defaults are the measured optimum; -1 = choose by size where applicable): key in
 * {"max_tile_log","loge","max_col_log","min_tiles_log","single_pass_max_log","max_digit_log","direct_tw_max_log",
 *  "xcd_remap","fixed_shapes","merkle_big_nlev","wave_local","prio_balance","loge_cols","tw_on_load","prune","fri_tail","fri_tail_stall","small_divisor_direct"}.  Plans are re-derived on the next
 * call; results never depend on the tuning ("fri_tail_stall" = k >= 0 is a test hook: the host withholds the challenge after round k of
 * the persistent tail kernel, whose wait then gives up after 2^13 polls; -1 = off; "small_divisor_direct" = 0: sc_coset_divide* transforms a
 * divisor of <= 8 coefficients like any other instead of evaluating it point by point).  Two keys manage the device-memory pool instead (freed vectors and trees are kept
 * on exact-size free lists, by default up to a quarter of the device's memory divided by the processes sharing the device;
 * environment STARKCORE_POOL_CAP_MB): "pool_cap_mb" = what the lists may keep from now on, "pool_trim" = hand everything on them
 * back to the device now (a caller whose own allocator -- torch's -- ran out of memory). */
int sc_set_tuning(const char* key, int value);
/* kernel launches (passes over the vector) one sc_ntt_dev of length n takes at the current tuning: 1 up to 2^11, 2 up to 2^20,
 * 3 up to 2^24, 4 beyond (0: n is not a power of two >= 2).  bench.py derives the algorithmic bytes per launch from it. */
int sc_ntt_num_passes(uint64_t n);
/* diagnostics (tools/pass_trace.py): while d_buf != NULL every geometry-specialised NTT pass launch writes 16 u64 per wave
 * (s_memtime at the phase boundaries of its workgroup; slot 15 = s_memrealtime at entry) to d_buf[(block*waves + wave)*16 ..];
 * the caller sizes the buffer for the launch it traces and passes NULL afterwards. */
int sc_debug_trace(void* d_buf);

/* diagnostics: out[i] = op(a[i], b[i]) computed by the device field routines the kernels use.
 * op 0: a*b*2^-128 (Montgomery product, b < p), 1: a+b, 2: a-b, 3: a*b, 4: a/2, 5: a^-1, 6: portable Montgomery product,
 * 7: the same product through the two-at-a-time routine of the butterflies (all ones if its two halves disagree) */
int sc_field_selftest(int op, const void* a, const void* b, void* out, uint64_t n);

/* ---- device vectors ----------------------------------------------------------------------- */
int sc_vec_alloc(uint64_t n, sc_vec_t** out);
int sc_vec_free(sc_vec_t* v);
/* a handle over device memory the CALLER owns (n field elements at d_elems, e.g. a torch tensor's storage): no copy; the memory must
 * outlive the handle and every call enqueued with it; sc_vec_free releases the handle only */
int sc_vec_wrap(void* d_elems, uint64_t n, sc_vec_t** out);
uint64_t sc_vec_len(const sc_vec_t* v);
void* sc_vec_ptr(sc_vec_t* v);         /* raw device pointer */
int sc_vec_zero(sc_vec_t* v);                                   /* all elements = 0 (library stream) */
int sc_vec_upload(sc_vec_t* v, uint64_t offset, const void* host, uint64_t count);
int sc_vec_download(const sc_vec_t* v, uint64_t offset, void* host, uint64_t count);
int sc_vec_gather(const sc_vec_t* v, const uint64_t* indices, uint64_t k, void* host_out); /* host_out[i] = v[indices[i]] */
/* `count` elements device to device, asynchronous on `stream` (a library vector <-> caller-owned device memory, e.g. a torch tensor) */
int sc_memcpy_dev(void* d_dst, const void* d_src, uint64_t count, void* stream);

/* ---- ntt / intt : code/ntt.py:3-18, :20-30 -------------------------------------------------- */
/* out[i] = sum_j in[j] * root^(i*j); inverse != 0: uses root^-1 and scales by n^-1 (ntt.py:27-30).
 * n must be a power of two (n <= 1 copies).  root is validated like ntt.py:10-11.
 * _dev: enqueued on `stream` (NULL: the library's).  Transforms -- this entry and sc_coset_evaluate_dev -- may be IN FLIGHT ON SEVERAL
 * STREAMS AT ONCE (independent columns side by side: each stream has its own intermediate vector); at 2^20 two streams carry
 * 1.4 x the elements per second of one.  Every other entry keeps the one-stream-at-a-time rule of its scratch buffers. */
int sc_ntt(const void* in, void* out, uint64_t n, const uint64_t root[2], int inverse);
int sc_ntt_dev(const void* d_in, void* d_out, uint64_t n, const uint64_t root[2], int inverse, void* stream);
/* The same transform for `cols` independent vectors of length n -- the registers of a trace, the loops over columns of
 * code/fast_stark.py:84-90 and :100-104 -- column c at element c * n of d_in and of d_out (d_in == d_out is allowed): one set of
 * launches covers up to 2^26 elements (64 columns of 2^20), every pass over cols x its tiles, so the workgroups of one column start
 * while those of another finish, and the 2^12-element tiles of a batch run two workgroups per CU (a lone 2^20 transform is one
 * workgroup per CU with every CU in the same phase: 30 G elements/s; 16-64 columns: 43-44; 64 columns of 2^16: 45-49 against 3.4
 * one at a time -- DESIGN.md 3.1). */
int sc_ntt_columns_dev(const void* d_in, void* d_out, uint64_t n, uint64_t cols, const uint64_t root[2], int inverse, void* stream);

/* ---- building blocks of the multi-GPU four-step NTT (no reference counterpart: the reference is single-process;
 *      together they compute exactly ntt.py:3-18 on a domain sharded over ranks, see stark-anatomy_amd/sharded.py) */
/* kind 0: d_in [len][batch] row-major, transform along axis 0 for every column, same layout out (natural order).
 * kind 1: d_in [batch][len], transform every row, output TRANSPOSED d_out [len][batch].  root: primitive len-th root. */
int sc_ntt_batch_dev(const void* d_in, void* d_out, uint64_t len, uint64_t batch, int kind, const uint64_t root[2], void* stream);
/* the same with the two fusions the sharded transform uses: kind 0 may multiply output (row r, column c) by
 * outer_root^(r * (outer_col_base + c)) [* outer_order^-1 if outer_scale_ninv] in its store epilogue (outer_root NULL = off);
 * kind 1 may read its input as [chunks][batch][len/chunks] (the layout all_to_all_single delivers) instead of [batch][len]. */
int sc_ntt_batch_ex_dev(const void* d_in, void* d_out, uint64_t len, uint64_t batch, int kind, const uint64_t root[2],
                        const uint64_t outer_root[2], uint64_t outer_order, uint64_t outer_col_base, int outer_scale_ninv, uint64_t chunks, void* stream);
/* kind 1 for ONE ROW BLOCK of the corner turn (overlap of the all-to-all with the row stage): transforms `batch` rows given as
 * [chunks][batch][len/chunks] and writes them as `batch` adjacent columns of a wider transposed output [len][out_ld]
 * (d_out points at the block's first column). */
int sc_ntt_rows_t_ld_dev(const void* d_in, void* d_out, uint64_t len, uint64_t batch, const uint64_t root[2], uint64_t chunks, uint64_t out_ld, void* stream);
/* The sharded transform as one plan object per rank (what SURVEY.md 8(b) sketched as `sc_ntt_sharded`): computes exactly
 * ntt.py:3-18 (inverse: ntt.py:20-30) of a length-2^log2n vector partitioned over `world` ranks.  n = n1 * n2 (n1 = 2^8 above
 * 2^16, else the square split); rank g holds the column slab [R][C/G] of the row-major R x C matrix of its input (forward: R = n1,
 * C = n2; inverse: R = n2, C = n1) and receives the column slab [C][R/G] of the output (sc_fourstep_shape returns R and C).
 * `d_send` and `d_recv` are caller-owned buffers of n/G elements laid out [G][R/G][C/G].  All entries are asynchronous on `stream`.
 *   sc_fourstep_cols_dev : column transforms + outer twiddle (+ n^-1 for the inverse), d_src -> d_send; block h of d_send is what
 *       rank h must receive into block g of ITS d_recv.  With d_recv_diag != NULL the block the rank keeps (h == g) is written
 *       straight into d_recv_diag (= the rank's d_recv) and the matching block of d_send is left untouched: it is never
 *       copied or sent.
 *   sc_fourstep_rows_dev : row transforms of the rank's rows [block * R/(G nblocks), ...) (nblocks = 1: all of them) read in
 *       place from d_recv, written transposed into d_dst [C][R/G].  defer_last_pass != 0 with a two-pass row transform runs
 *       only the first pass here; sc_fourstep_rows_finish_dev then runs the second pass for ALL rows in one launch (and is a
 *       no-op when nothing was deferred).
 *   sc_fourstep_run_dev : the whole transform, with the corner turn issued over the library's own RCCL communicator
 *       (sc_comm_init; grouped ncclSend/ncclRecv to the G - 1 peers, straight from d_send into the peers' d_recv).  nblocks > 1
 *       issues the exchange as row blocks on the library's communication stream and starts the row transforms of a block as
 *       soon as it has landed.  force_diag_exchange (tests): the rank's own block goes through RCCL too. */
typedef struct sc_fourstep sc_fourstep_t;
typedef struct { char internal[128]; } sc_rccl_id_t;   /* = ncclUniqueId */
int sc_fourstep_create(int log2n, const uint64_t root[2], int rank, int world, sc_fourstep_t** plan);
/* the same with the split chosen by the caller: n1 = 2^log_n1 (0 = the default: 2^8 above 2^16, square below) */
int sc_fourstep_create_ex(int log2n, const uint64_t root[2], int rank, int world, int log_n1, sc_fourstep_t** plan);
int sc_fourstep_free(sc_fourstep_t* plan);
int sc_fourstep_shape(const sc_fourstep_t* plan, int inverse, uint64_t* rows, uint64_t* cols_total);
int sc_fourstep_cols_dev(const sc_fourstep_t* plan, int inverse, const void* d_src, void* d_send, void* d_recv_diag, void* stream);
int sc_fourstep_rows_dev(const sc_fourstep_t* plan, int inverse, const void* d_recv, void* d_dst, uint64_t block, uint64_t nblocks, int defer_last_pass, void* stream);
int sc_fourstep_rows_finish_dev(const sc_fourstep_t* plan, int inverse, void* d_dst, void* stream);
/* the library's RCCL communicator (one per process; RCCL is dlopen'ed: rccl_path NULL = the copy already in the process, e.g.
 * torch's, else librccl.so.1): rank 0 makes an id, the launcher distributes its 128 bytes (torch.distributed broadcast), every rank
 * calls sc_comm_init with it (collective, blocking).  SC_ERR_UNSUPPORTED without RCCL. */
int sc_comm_unique_id(const char* rccl_path, sc_rccl_id_t* id_out);
int sc_comm_init(const char* rccl_path, const sc_rccl_id_t* id, int rank, int world);
int sc_comm_destroy(void);
int sc_fourstep_run_dev(const sc_fourstep_t* plan, int inverse, const void* d_src, void* d_send, void* d_recv, void* d_dst, uint64_t nblocks, int defer_last_pass,
                        int force_diag_exchange, void* stream);
/* DIRECT-STORE corner turn: no collective at all.  Every rank owns a REGION of device memory -- [4 KiB of flags][receive buffer
 * 0][receive buffer 1], sc_fourstep_region_bytes() bytes -- exported with a HIP IPC handle (64 bytes, carried to the other ranks
 * by the caller's launcher) and mapped by every peer (sc_ipc_region_open).  sc_fourstep_run_direct_dev then runs the whole
 * transform: the column stage stores block h of its output STRAIGHT INTO rank h's receive buffer (the stores cross xGMI; no
 * send buffer, no copy kernel, one HBM write + read per element less than a send/recv exchange), a one-wave kernel raises this
 * rank's flag at every peer and waits for theirs (system-scope atomics), and the row stage reads the rank's own receive buffer.
 * Consecutive transforms alternate between the two receive buffers, so a peer's next column stage never overwrites what a row
 * stage is still reading.  Every rank must run the same transforms in the same order.  A barrier that waits ~2 s for a peer
 * gives up and writes the transform's number to a pinned word of the plan: that transform's output is undefined, and from then on
 * the failure is STICKY -- sc_fourstep_direct_status reports the number (no copy, no wait) and every later
 * sc_fourstep_run_direct_dev returns SC_ERR_TIMEOUT until sc_fourstep_set_peers is called again (the peers' buffers are out of
 * step; the caller re-creates the set-up or goes back to the collective exchange).
 * The region is FINE-GRAINED device memory (hipExtMallocWithFlags, as RCCL's peer-written buffers are): peers write into it
 * while this GPU's kernels poll and read it.  sc_ipc_region_create_ex chooses: kind 1 fine-grained (coarse-grained when the
 * runtime cannot export one), 0 coarse-grained = plain hipMalloc, -1 the default (fine-grained unless STARKCORE_IPC_COARSE=1);
 * sc_ipc_region_kind: 1 fine-grained, 0 coarse-grained, -1 no region created yet. */
int sc_ipc_region_create(uint64_t bytes, void** d_region, uint8_t handle_out[64]);
int sc_ipc_region_create_ex(uint64_t bytes, int kind, void** d_region, uint8_t handle_out[64]);
int sc_ipc_region_kind(int* fine_grained);
int sc_ipc_region_open(const uint8_t handle[64], void** d_region);
int sc_ipc_region_close(void* d_region);        /* a region opened from a peer's handle */
int sc_ipc_region_free(void* d_region);         /* a region this process created */
int sc_fourstep_region_bytes(const sc_fourstep_t* plan, uint64_t* bytes);
int sc_fourstep_set_peers(sc_fourstep_t* plan, void* const* regions);      /* regions[h]: rank h's region as mapped HERE (own: its own) */
int sc_fourstep_run_direct_dev(sc_fourstep_t* plan, int inverse, const void* d_src, void* d_dst, void* stream);
int sc_fourstep_direct_status(const sc_fourstep_t* plan, uint64_t* timed_out_epoch);
/* d_data[r][c] *= root^((row_base + r) * (col_base + c)) * scale, root of order `order` (scale may be NULL = 1) */
int sc_twiddle_matrix_dev(void* d_data, uint64_t rows, uint64_t cols, uint64_t row_base, uint64_t col_base, const uint64_t root[2], uint64_t order,
                          const uint64_t scale[2], void* stream);

/* ---- fast_coset_evaluate : code/ntt.py:132-135 (Polynomial.scale univariate.py:153-154 fused) -- */
/* out[i] = sum_{j<m} coeffs[j] * (offset * generator^i)^j, i < order; m <= order. */
int sc_coset_evaluate(const void* coeffs, uint64_t m, const uint64_t offset[2], const uint64_t generator[2], uint64_t order, void* out);
int sc_coset_evaluate_dev(const void* d_coeffs, uint64_t m, const uint64_t offset[2], const uint64_t generator[2], uint64_t order, void* d_out, void* stream);
/* The same for `cols` polynomials of m >= 1 coefficients each -- the loop over registers of code/fast_stark.py:100-104 -- polynomial c at
 * element c * m of d_coeffs, its values at element c * order of d_out; one set of launches (sc_ntt_columns_dev). */
int sc_coset_evaluate_columns_dev(const void* d_coeffs, uint64_t m, uint64_t cols, const uint64_t offset[2], const uint64_t generator[2], uint64_t order, void* d_out, void* stream);

/* ---- NTT core of fast_multiply : code/ntt.py:51-64 ----------------------------------------- */
/* out[0..n_out) = intt(root, ntt(root, a||0) * ntt(root, b||0))[0..n_out); na, nb, n_out <= order.
 * (the degree bookkeeping / order halving of ntt.py:38-49 stays in the host shim.) */
int sc_poly_mul(const void* a, uint64_t na, const void* b, uint64_t nb, const uint64_t root[2], uint64_t order, void* out, uint64_t n_out);

/* ---- NTT core of fast_coset_divide : code/ntt.py:159-176 ----------------------------------- */
/* out[0..n_out) = unscale( intt( ntt(scale(a)) / ntt(scale(b)) ) ); SC_ERR_DIV_ZERO if a divisor value is 0 */
int sc_coset_divide(const void* a, uint64_t na, const void* b, uint64_t nb, const uint64_t offset[2], const uint64_t root[2], uint64_t order, void* out, uint64_t n_out);
/* the same on coefficient vectors in HBM (d_out: n_out coefficients).  *exact (may be NULL; makes the call synchronous) is set to 1
 * iff the interpolant's coefficients [n_out, order) all vanish: with order > deg(a), exactly the condition "b divides a and the
 * quotient has fewer than n_out coefficients" that Polynomial.__truediv__ asserts (code/univariate.py:99-103) */
int sc_coset_divide_dev(const void* d_a, uint64_t na, const void* d_b, uint64_t nb, const uint64_t offset[2], const uint64_t root[2], uint64_t order,
                        void* d_out, uint64_t n_out, int* exact, void* stream);
/* The same division, and the pointwise division of code/ntt.py:172 on its own, with NOTHING waited for: what the reference asserts on
 * the spot -- "divide by zero" (code/algebra.py:92) and, for the coset division, a zero remainder (code/univariate.py:99-103) -- is
 * decided on the device and arrives in *later behind the work; sc_later_wait returns words_out[0] != 0: a divisor value was zero;
 * words_out[1]: the highest index, counted from n_out, of a non-zero coefficient of the interpolant above the quotient, -1 iff the
 * division is exact (always -1 for the pointwise form).  A prover collects its checks where it has to wait anyway (before the next
 * Fiat-Shamir challenge) instead of idling the GPU at every division.  sc_later_wait frees the handle (it is also how one is abandoned).
 * SC_ERR_UNSUPPORTED: no pinned slot free -- use the waiting forms. */
int sc_coset_divide_later_dev(const void* d_a, uint64_t na, const void* d_b, uint64_t nb, const uint64_t offset[2], const uint64_t root[2], uint64_t order,
                              void* d_out, uint64_t n_out, sc_later_t** later, void* stream);
int sc_pointwise_div_later_dev(const void* d_a, const void* d_b, void* d_out, uint64_t n, sc_later_t** later, void* stream);
int sc_later_wait(sc_later_t* later, int64_t words_out[8]);
/* Polynomial.degree (code/univariate.py:7-17) of a coefficient vector in HBM: index of the last non-zero entry, -1 if none (synchronous) */
int sc_vec_degree_dev(const void* d_v, uint64_t n, int64_t* degree_out, void* stream);

/* ---- pointwise helpers (ntt.py:61, :172; univariate.py:153-154) ------------------------------ */
int sc_pointwise_mul_dev(const void* d_a, const void* d_b, void* d_out, uint64_t n, void* stream);
int sc_pointwise_div_dev(const void* d_a, const void* d_b, void* d_out, uint64_t n, void* stream); /* sync; SC_ERR_DIV_ZERO */
int sc_scale_dev(const void* d_in, void* d_out, uint64_t n, const uint64_t factor[2], void* stream); /* out[i] = in[i] * factor^i */
/* acc[shift + j] += weight * src[j], j < n_src: one term `Polynomial([weight]) * (x ^ shift) * term` of the nonlinear combination
 * of code/fast_stark.py:130-145 on coefficient vectors in HBM (shift + n_src <= n_acc) */
int sc_axpy_shift_dev(void* d_acc, uint64_t n_acc, const void* d_src, uint64_t n_src, uint64_t shift, const uint64_t weight[2], void* stream);
/* the same scaling (univariate.py:153-154) on one rank's column slab [rows][cols] of a vector viewed as a rows x row_len
 * matrix (multi-GPU fast_coset_evaluate / fast_coset_divide, ntt.py:132-135, :159-176):
 * out[r][c] = in[r][c] * factor^(r * row_len + col_base + c); cols a power of two */
int sc_scale_slab_dev(const void* d_in, void* d_out, uint64_t rows, uint64_t cols, uint64_t row_len, uint64_t col_base, const uint64_t factor[2], void* stream);

/* ---- fast_zerofier / fast_evaluate / fast_interpolate : code/ntt.py:66-80, :82-100, :102-130 -- */
/* The reference recurses node by node (split at len//2, schoolbook remainders, zerofiers recomputed per node); zerofier,
 * values and interpolant are unique, so the device works level by level on a perfect tree over the k points padded with zeros
 * to a power of two (stark-anatomy_amd/csrc/polytree.cuh).  Points are arbitrary canonical residues; they need to be distinct
 * only for interpolation (a repeated point gives SC_ERR_DIV_ZERO, like the assertion of algebra.py:92 in ntt.py:124-125).
 * The transforms use the field's own 2^j-th roots (algebra.py:104-111); no root argument is needed. */
int sc_zerofier(const void* points, uint64_t k, void* out);                                   /* out: k + 1 coefficients of prod (X - d_i) */
int sc_evaluate(const void* coeffs, uint64_t m, const void* points, uint64_t k, void* out);   /* out[i] = poly(points[i]), any m */
int sc_interpolate(const void* points, const void* values, uint64_t k, void* out);            /* out: k coefficients (degree < k) */
/* the same on device pointers with the tree kept between calls (one tree serves any number of evaluations / interpolations) */
int sc_polytree_build(const void* points, uint64_t k, sc_polytree_t** tree);                  /* k >= 1 */
int sc_polytree_build_dev(const void* d_points, uint64_t k, sc_polytree_t** tree, void* stream);
uint64_t sc_polytree_points(const sc_polytree_t* tree);
int sc_polytree_zerofier_dev(const sc_polytree_t* tree, void* d_out, void* stream);           /* k + 1 coefficients */
/* d_points: the tree's points again, only read when m exceeds the padded domain size (chunked evaluation); may be NULL otherwise */
int sc_polytree_evaluate_dev(sc_polytree_t* tree, const void* d_coeffs, uint64_t m, const void* d_points, void* d_out, void* stream);
int sc_polytree_interpolate_dev(sc_polytree_t* tree, const void* d_values, void* d_out, void* stream);
int sc_polytree_free(sc_polytree_t* tree);

/* ---- the same three functions on a GEOMETRIC PROGRESSION  x_i = first * ratio^i, i < n : code/ntt.py:66-130 as called by
 * code/fast_stark.py:84-90 (the trace domain {omicron^i}) and :37 (the transition zerofier's domain) -----------------------
 * Zerofier, values and interpolant are unique, so on such a domain they are computed with O(1) transforms of length
 * M = pow2 >= 2n - 1 (Bluestein / Bostan-Schost) instead of a tree: same outputs as sc_polytree_* / the reference's recursion.
 * create: n >= 2, distinct points required (SC_ERR_UNSUPPORTED otherwise: use the tree, which reproduces the reference's
 * behaviour for repeated points).  The *_dev calls only ENQUEUE on `stream` (NULL: the library stream): results are valid in
 * stream order; nothing waits, temporaries are returned behind the stream.
 * detect: is d_points[i+1] == d_points[i] * ratio for all i (ratio = points[1] / points[0])?  one small kernel + one sync. */
int sc_geodomain_create(const uint64_t first[2], const uint64_t ratio[2], uint64_t n, sc_geodomain_t** domain, void* stream);
uint64_t sc_geodomain_points(const sc_geodomain_t* domain);
int sc_geodomain_detect_dev(const void* d_points, uint64_t n, uint64_t first[2], uint64_t ratio[2], int* is_geometric, void* stream);
int sc_geodomain_zerofier_dev(const sc_geodomain_t* domain, void* d_out, void* stream);          /* n + 1 coefficients */
int sc_geodomain_evaluate_dev(const sc_geodomain_t* domain, const void* d_coeffs, uint64_t m, void* d_out, void* stream);   /* n values, any m */
int sc_geodomain_interpolate_dev(const sc_geodomain_t* domain, const void* d_values, void* d_out, void* stream);            /* n coefficients */
int sc_geodomain_free(sc_geodomain_t* domain);

/* ---- MPolynomial.evaluate_symbolic in the value domain : code/multivariate.py:83-90 (call site fast_stark.py:109-110) ---- */
/* d_vals: [nvars][n] values of the point polynomials on an n-point domain (CONSUMED: converted in place to the library's
 * internal form); exps: host [nterms][nvars] exponents, coefs: host nterms packed residues.
 * d_out[i] = sum_t coefs[t] * prod_j vals[j][i]^exps[t][j].  With n > the degree of the result, an inverse NTT of d_out
 * gives exactly the polynomial the reference builds from schoolbook products (§8(f)-2). */
int sc_mpoly_eval_dev(void* d_vals, uint64_t nvars, uint64_t n, const uint8_t* exps, const void* coefs, uint64_t nterms, void* d_out, void* stream);
/* the same for several constraints over ONE set of point values: vals_converted != 0 says d_vals has been through an earlier call.
 * n may be any count (a rank's slab of a sharded value domain: the evaluation is pointwise). */
int sc_mpoly_eval_ex_dev(void* d_vals, uint64_t nvars, uint64_t n, const uint8_t* exps, const void* coefs, uint64_t nterms, void* d_out, int vals_converted, void* stream);
/* the same with variables that are TURNED copies of others (fast_stark.py:105-106: the point holds trace(X) and trace(omicron X); on
 * the coset g <omicron> the second one's codeword is the first one's, one place on): variable j is read as
 * d_vals[var_src[j]][(i + var_rot[j]) mod n] (n a power of two); var_src[j] == j with var_rot[j] == 0 for a variable stored in its
 * own place, var_src[j] == SC_MPOLY_ABSENT for one that no term uses (its place may hold anything and is not touched; a term that
 * does use it is an error).  A turned variable must point at a stored one.  var_src == var_rot == NULL: sc_mpoly_eval_ex_dev. */
#define SC_MPOLY_ABSENT 0xFFFFFFFFu
int sc_mpoly_eval_rot_dev(void* d_vals, uint64_t nvars, uint64_t n, const uint8_t* exps, const void* coefs, uint64_t nterms, void* d_out, int vals_converted,
                          const uint32_t* var_src, const uint64_t* var_rot, void* stream);

/* ---- FRI split-and-fold : code/fri.py:85 ---------------------------------------------------- */
/* out[i] = 2^-1 * ((1 + alpha/(offset*omega^i)) * in[i] + (1 - alpha/(offset*omega^i)) * in[N/2+i]), i < N/2 */
int sc_fri_fold(const void* in, uint64_t N, const uint64_t alpha[2], const uint64_t offset[2], const uint64_t omega[2], void* out);
int sc_fri_fold_dev(const void* d_in, uint64_t N, const uint64_t alpha[2], const uint64_t offset[2], const uint64_t omega[2], void* d_out, void* stream);

/* ---- Merkle : code/merkle.py:6-27 ------------------------------------------------------------ */
/* leaf = BLAKE2b-512(decimal ASCII of the residue) (algebra.py:53-57, merkle.py:14); node = H(left||right) */
int sc_merkle_commit(const void* elems, uint64_t N, uint8_t root_out[64]);                 /* Merkle.commit, merkle.py:13-14 */
int sc_merkle_build(const void* elems, uint64_t N, uint8_t root_out[64], sc_merkle_t** tree);
int sc_merkle_build_dev(const void* d_elems, uint64_t N, uint8_t root_out[64], sc_merkle_t** tree, void* stream); /* sync (returns root) */
/* The commit loop of Fri.commit (fri.py:66-94) without idle time on either side: the build is only ENQUEUED (the caller prepares
 * the next round meanwhile); sc_merkle_root waits for it (once) and returns the root -- it works on every tree.
 * sc_fri_fold_commit_dev = one round in one call: the fold of fri.py:85 into d_out (N/2 elements), then the asynchronous
 * build of the tree over d_out.  Fetch the root (or free the tree) before destroying a caller-owned stream the build ran on. */
int sc_merkle_build_async_dev(const void* d_elems, uint64_t N, sc_merkle_t** tree, void* stream);
/* the same, for a tree whose ROOT nobody is expected to read (a rank's local subtree of a sharded commit, whose sub-root level is
 * copied out with sc_merkle_level_copy_dev on the same stream): takes none of the 256 pinned root slots and runs no publish
 * kernel, so any number of such trees may be alive; sc_merkle_root still works (it then waits for the whole device). */
int sc_merkle_build_noroot_dev(const void* d_elems, uint64_t N, sc_merkle_t** tree, void* stream);
int sc_merkle_root(sc_merkle_t* tree, uint8_t root_out[64]);
int sc_fri_fold_commit_dev(const void* d_in, uint64_t N, const uint64_t alpha[2], const uint64_t offset[2], const uint64_t omega[2], void* d_out,
                           sc_merkle_t** tree, void* stream);
/* Fri.commit's whole round loop (fri.py:66-94) in one call, the Fiat-Shamir step included: per round the tree of the codeword
 * (built asynchronously, the root polled from its pinned slot), alpha = Field.sample(SHAKE-256(pickle.dumps(transcript)))
 * (ip.py:18-25, algebra.py:116-120) computed by the library, and the fold of fri.py:85 enqueued the moment alpha exists --
 * nothing crosses the language boundary between a root arriving and the next launch.  The transcript is the list of the
 * `prior_count` byte strings already in the proof stream (prior_lens[i] < 256 bytes each, concatenated in prior_data; the
 * caller checks that the stream holds nothing else) followed by this call's roots: that list has a fixed pickle layout
 * (csrc/transcript.h).  omega and offset are squared from round to round (fri.py:86-87).
 * Out: trees_out[rounds] ([0] over d_codeword), vecs_out[rounds - 1] folded codewords (library-owned: sc_vec_free),
 * roots_out[64 * rounds], alphas_out[2 * (rounds - 1)].  SC_ERR_UNSUPPORTED: the transcript does not have the fixed layout. */
int sc_fri_commit_dev(const void* d_codeword, uint64_t N, const uint64_t offset[2], const uint64_t omega[2], uint32_t rounds,
                      const void* prior_data, const uint32_t* prior_lens, uint64_t prior_count,
                      sc_vec_t** vecs_out, sc_merkle_t** trees_out, uint8_t* roots_out, uint64_t* alphas_out, void* stream);
/* Fri.prove (code/fri.py:115-130) in ONE call: the commit phase as above (fri.py:66-94), then -- without leaving the library --
 * the last codeword in the clear (fri.py:91), the challenge SHAKE-256(pickle.dumps([prior..., roots..., last codeword])) of
 * ip.py:18-25 (the FieldElement list pickled as CPython pickles it: csrc/proof_pickle.h), Fri.sample_indices (fri.py:36-51, :122;
 * BLAKE2b of seed + counter zero bytes), the positions each round opens (fri.py:98-113, :124-128) and ONE kernel that gathers every
 * opened element and authentication path of the proof.  `extra_count` further (tree, device vector) pairs of N leaves -- FastStark's
 * committed codewords, fast_stark.py:154-175 -- are opened in the same launch at the sorted positions
 * {i, i + extra_shift, i + N/2, i + extra_shift + N/2 (mod N)} over the top-level indices i (written to extra_indices_out, 4 *
 * num_tests of them).  Openings per pair, in this order: codeword j of the commit phase: [a (num_tests), b = a + half (num_tests)] if
 * j < rounds - 1 -- the c positions of the round before (= that round's a) are this codeword's a or b, whichever half they lie in, and
 * are opened once --, the last codeword [c (num_tests)]; every further pair: the sorted positions.
 * `answers` (answers_bytes >= the sum below) = [opened elements, 16 bytes each, padded to a multiple of 256 bytes][paths, 64 * log2
 * N_pair bytes per opening][the positions, u64 each], pairs concatenated.  A buffer of sc_host_alloc is written by the kernel itself
 * across the bus (no staging copy); any other host pointer works through two copies.  Outputs of the commit phase as for
 * sc_fri_commit_dev -- or vecs_out = trees_out = NULL (alphas_out may be NULL as well): nothing reads the folded codewords and
 * their trees once the openings are on the host, and the library then hands their memory back before it returns;
 * last_codeword_out: 16 * (N >> (rounds - 1)) bytes; top_indices_out: num_tests.
 * SC_ERR_UNSUPPORTED: a transcript or shape this entry does not serve (the caller runs the phases one by one). */
int sc_fri_prove_dev(const void* d_codeword, uint64_t N, const uint64_t offset[2], const uint64_t omega[2], uint32_t rounds, uint32_t num_tests,
                     const void* prior_data, const uint32_t* prior_lens, uint64_t prior_count,
                     uint64_t extra_count, const sc_merkle_t* const* extra_trees, const void* const* extra_vecs, uint64_t extra_shift,
                     sc_vec_t** vecs_out, sc_merkle_t** trees_out, uint8_t* roots_out, uint64_t* alphas_out,
                     void* last_codeword_out, uint64_t* top_indices_out, uint64_t* extra_indices_out,
                     void* answers, uint64_t answers_bytes, void* stream);
/* pinned, device-visible host memory from a pool kept by the library (an allocation of megabytes costs hundreds of microseconds):
 * what sc_fri_prove_dev's query kernel writes a proof's openings to */
/* diagnostics of the persistent tail kernel of the commit phase (csrc/fri_tail.cuh): out[0] = launches so far, out[1] = launches
 * whose wait for a challenge or a peer workgroup gave up, after which the rounds were finished with the per-round launches */
int sc_fri_tail_stats(uint64_t out[2]);
int sc_host_alloc(uint64_t bytes, void** out);
int sc_host_free(void* p);
/* the host-side pieces of that step on their own (no GPU needed; tests pin them to hashlib / pickle):
 * SHAKE-256 (FIPS 202); Field.sample = big-endian integer of the bytes mod p; pickle.dumps of a list of `count` byte strings
 * (*out_len = bytes needed, copied into out when out_cap suffices) */
int sc_shake256(const void* in, uint64_t len, void* out, uint64_t out_len);
/* ProofStream.serialize() (code/ip.py:18-19: pickle.dumps(self.objects), protocol 4) of a proof stream given as a DESCRIPTION of
 * its object graph instead of Python objects -- lists, bytes, authentication paths, triples, FieldElements with their object
 * identities (format: csrc/proof_pickle.h).  Byte-identical to CPython's pickler.  moduli: nfields little-endian moduli of
 * modulus_bytes each.  *out_len = bytes needed; copied into out when out_cap suffices.  Host only (no GPU needed). */
int sc_pickle_proof(const void* ops, uint64_t ops_len, const void* moduli, uint32_t nfields, uint32_t modulus_bytes, void* out, uint64_t out_cap, uint64_t* out_len);
/* Field.sample (algebra.py:116-120) of `count` host byte strings of `width` <= 32 bytes each, straight into device memory: the
 * randomizer polynomial of FastStark.prove (fast_stark.py:117: one os.urandom(17) draw per coefficient) without a Python object
 * per coefficient.  Synchronous (the bytes are the caller's host memory). */
int sc_sample_bytes_dev(const void* bytes, uint64_t count, uint32_t width, void* d_out, void* stream);
/* The same with the draws made by the library: `count` times getrandom(width bytes) -- what os.urandom(width) is -- split over
 * host threads into a pinned staging buffer, one asynchronous copy, Field.sample on the device.  Enqueued on `stream`.  For
 * callers whose os.urandom is the operating system's (a patched, seeded os.urandom must go through sc_sample_bytes_dev, whose
 * bytes and order are the caller's). */
int sc_sample_urandom_dev(uint64_t count, uint32_t width, void* d_out, void* stream);
/* start those draws NOW on host threads and return: a later sc_sample_urandom_dev of the same (count, width) uses them (the prover
 * calls this at the top of a proof; the kernel randomness is drawn while the GPU works on the trace) */
int sc_urandom_prefetch(uint64_t count, uint32_t width);
int sc_field_sample(const void* bytes, uint64_t len, uint64_t out[2]);
/* ... and of sc_fri_prove_dev's index sampling: BLAKE2b-512 (RFC 7693, unkeyed) of a message of any length; Fri.sample_indices
 * (fri.py:36-51) for a power-of-two `size`: `number` indices below size, pairwise distinct modulo reduced_size, candidate k = the
 * big-endian integer of blake2b(seed + k zero bytes) mod size (SC_ERR_UNSUPPORTED where the reference's assertion fails) */
int sc_blake2b(const void* in, uint64_t len, uint8_t out[64]);
/* SHAKE-256(pickle.dumps(items + [root])) (ip.py:18-25) in the split form the commit loops use: every whole rate block in front of
 * the pending 64-byte root absorbed before the root is known, the rest after (csrc/transcript.h: PendingChallenge) */
int sc_transcript_challenge(const void* data, const uint32_t* lens, uint64_t count, const uint8_t root[64], uint8_t* out, uint64_t out_len);
int sc_fri_sample_indices(const void* seed, uint64_t seed_len, uint64_t size, uint64_t reduced_size, uint32_t number, uint64_t* out);
int sc_transcript_bytes(const void* data, const uint32_t* lens, uint64_t count, void* out, uint64_t out_cap, uint64_t* out_len);
int sc_merkle_open(const sc_merkle_t* tree, uint64_t index, uint8_t* path_out /* 64*log2 N */); /* Merkle.open, merkle.py:16-27 */
int sc_merkle_open_batch(const sc_merkle_t* tree, const uint64_t* indices, uint64_t k, uint8_t* paths_out /* k*64*log2 N */);
/* opened elements AND their paths in one call: elems_out[i] = d_elems[indices[i]] (d_elems = the device vector the tree was built from) */
int sc_merkle_query_dev(const sc_merkle_t* tree, const void* d_elems, const uint64_t* indices, uint64_t k, void* elems_out, uint8_t* paths_out);
/* several (tree, vector) pairs in one round trip (the query phase of Fri.prove, fri.py:124-128): counts[t] of the concatenated
 * `indices` belong to pair t; outputs concatenated in the same order (16 bytes per element, 64 * log2 N_t bytes per path) */
int sc_merkle_query_multi_dev(uint64_t n, const sc_merkle_t* const* trees, const void* const* d_elems, const uint64_t* indices, const uint64_t* counts,
                              void* elems_out, uint8_t* paths_out);
/* pieces for a tree sharded over ranks: copy of one level of a built tree (level 0 = leaf digests; (N >> level) * 64 bytes),
 * a tree whose level 0 is given digests, and the fold of fri.py:85 on a rank's column slab [rows][cols] of the codeword
 * viewed as a rows x R matrix (index i = row*R + col_base + col; the partner i + N/2 is row + rows/2 of the same slab) */
int sc_merkle_level_copy_dev(const sc_merkle_t* tree, int level, void* d_out, void* stream);
int sc_merkle_from_digests_dev(const void* d_digests, uint64_t count, uint8_t root_out[64], sc_merkle_t** tree, void* stream); /* root_out NULL: only enqueued, sc_merkle_root waits */
int sc_fri_fold_slab_dev(const void* d_in, uint64_t rows, uint64_t cols, uint64_t R, uint64_t col_base, const uint64_t alpha[2], const uint64_t offset[2],
                         const uint64_t omega[2], void* d_out, void* stream);
/* that fold AND the local Merkle subtree over the folded slab in one call (a round of the sharded Fri.commit, fri.py:73-88, on one
 * rank): from 256 folded elements up the tree's leaf stage computes the fold itself; the tree is a *_noroot_dev tree (only
 * enqueued; its sub-root level is read with sc_merkle_level_copy_dev on the same stream) */
int sc_fri_fold_slab_build_dev(const void* d_in, uint64_t rows, uint64_t cols, uint64_t R, uint64_t col_base, const uint64_t alpha[2], const uint64_t offset[2],
                               const uint64_t omega[2], void* d_out, sc_merkle_t** tree, void* stream);
uint64_t sc_merkle_leaves(const sc_merkle_t* tree);
int sc_merkle_free(sc_merkle_t* tree);

#ifdef __cplusplus
}
#endif
#endif
