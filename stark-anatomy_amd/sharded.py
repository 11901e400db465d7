"""Four-step NTT sharded over the GPUs of one node (one process per GPU, RCCL all-to-all over xGMI).

Computes exactly what reference code/ntt.py:3-18 (`ntt`) / :20-30 (`intt`) compute, on a domain too large
(or too slow) for one device; the reference itself is single-process, so the partitioning is this build's.

Layout ("column slab"): a length-n vector x, n = R * C, is viewed as the row-major R x C matrix
M[r][c] = x[r*C + c]; rank g of G holds the columns c in [g*C/G, (g+1)*C/G), stored locally as a
contiguous [R][C/G] array.  forward(): R = n1, C = n2 in -> [n2][n1/G] out, i.e. the column slab of the
n2 x n1 matrix of X (X[k2*n1 + k1]); inverse() maps that layout back (n1 = 2^8 for n > 2^16, see __init__).

Per transform and rank:   (1) column NTTs of length R on the local slab          (local, HIP)
                          (2) outer twiddle  w_n^(r * c_global) [* n^-1]          (local, HIP)
                          (3) corner turn: ONE all-to-all, (G-1)/G of the slab    (RCCL over xGMI; all 7 links busy)
                          (4) row NTTs of length C, written transposed            (local, HIP)
There is no reduction anywhere, so no all-reduce / ring is used.
"""
import itertools

import torch
import torch.distributed as dist

P = 1 + 407 * (1 << 119)


def _fe(v):
    return int(v).to_bytes(16, "little")


def _exchange_single(recv, send, group=None):
    """all_to_all_single on [G][...] blocks.  Backend "nccl" (= RCCL) moves device buffers directly; under gloo with device
    tensors (functional tests: several ranks sharing one GPU) the exchange is staged through the host."""
    if send.is_cuda and dist.get_backend(group) == "gloo":
        host_recv = torch.empty(recv.shape, dtype=recv.dtype)
        dist.all_to_all_single(host_recv.view(-1), send.cpu().contiguous().view(-1), group=group)
        recv.copy_(host_recv)
        return
    dist.all_to_all_single(recv.view(-1), send.contiguous().view(-1), group=group)


def rows_to_column_slab(chunk, rows, cols, rank, world, group=None):
    """A vector that arrives in NATURAL contiguous layout -- rank g holds x[g*n/G : (g+1)*n/G], i.e. rows [g*rows/G, (g+1)*rows/G)
    of the row-major rows x cols matrix -- re-laid out as column slabs [rows][cols/G] with ONE all-to-all (the corner turn
    itself).  This is how a codeword in the contiguous-slab layout of SURVEY.md 8(e)-2 enters the zero-exchange FRI layout:
    one exchange up front instead of one neighbour exchange per fold (same total bytes: (G-1)/G of the vector)."""
    rw, cw = rows // world, cols // world
    assert tuple(chunk.shape) == (rw, cols, 2)
    if world == 1:
        return chunk
    send = chunk.view(rw, world, cw, 2).permute(1, 0, 2, 3).contiguous()       # block h = my rows x columns of rank h
    recv = torch.empty((world, rw, cw, 2), dtype=chunk.dtype, device=chunk.device)
    _exchange_single(recv, send, group)
    return recv.view(rows, cw, 2)                                              # block g = rows of rank g, my columns


def column_slab_to_rows(slab, rows, cols, rank, world, group=None):
    """inverse of rows_to_column_slab: [rows][cols/G] -> this rank's contiguous rows [rows/G][cols]"""
    rw, cw = rows // world, cols // world
    assert tuple(slab.shape) == (rows, cw, 2)
    if world == 1:
        return slab
    recv = torch.empty((world, rw, cw, 2), dtype=slab.dtype, device=slab.device)
    _exchange_single(recv, slab.contiguous().view(world, rw, cw, 2), group)
    return recv.permute(1, 0, 2, 3).contiguous().view(rw, cols, 2)


class HipEngine:
    """Local stages through the C-ABI (libstarkcore.so) on torch-owned device memory."""

    def __init__(self, stream):
        import ctypes
        import starkcore as sc
        self.sc = sc
        self.lib = sc.lib()
        self.stream = stream
        self.sptr = ctypes.c_void_p(stream.cuda_stream)
        assert stream.cuda_stream != 0, "use a non-null HIP stream"

    def cols_ntt(self, src, dst, length, batch, root):
        self.sc._check(self.lib.sc_ntt_batch_dev(src.data_ptr(), dst.data_ptr(), length, batch, 0, _fe(root), self.sptr))

    def rows_ntt_t(self, src, dst, length, batch, root):
        self.sc._check(self.lib.sc_ntt_batch_dev(src.data_ptr(), dst.data_ptr(), length, batch, 1, _fe(root), self.sptr))

    def twiddle(self, buf, rows, cols, row_base, col_base, root, order, scale):
        self.sc._check(self.lib.sc_twiddle_matrix_dev(buf.data_ptr(), rows, cols, row_base, col_base, _fe(root), order, _fe(scale), self.sptr))

    def scale_powers(self, src, dst, count, factor):
        """dst[j] = src[j] * factor^j (Polynomial.scale, code/univariate.py:153-154)"""
        self.sc._check(self.lib.sc_scale_dev(src.data_ptr(), dst.data_ptr(), count, _fe(factor), self.sptr))

    def scale_slab(self, src, dst, rows, cols, row_len, col_base, factor):
        """dst[r][c] = src[r][c] * factor^(r*row_len + col_base + c): Polynomial.scale on this rank's columns"""
        self.sc._check(self.lib.sc_scale_slab_dev(src.data_ptr(), dst.data_ptr(), rows, cols, row_len, col_base, _fe(factor), self.sptr))

    def pointwise_mul(self, a, b, out, count):
        """Hadamard product (code/ntt.py:61)"""
        self.sc._check(self.lib.sc_pointwise_mul_dev(a.data_ptr(), b.data_ptr(), out.data_ptr(), count, self.sptr))

    def pointwise_div(self, a, b, out, count):
        """pointwise quotient (code/ntt.py:172); a zero divisor raises the reference's AssertionError("divide by zero")"""
        self.sc._check(self.lib.sc_pointwise_div_dev(a.data_ptr(), b.data_ptr(), out.data_ptr(), count, self.sptr))

    def fourstep(self, log2n, root, rank, world, log_n1=0):
        """the rank's stage object for the sharded transform (sc_fourstep_t)"""
        return HipFourstep(self.sc, log2n, root, rank, world, self.sptr, log_n1)


class _DoneWork:
    def wait(self):
        return True


class _Works:
    """several asynchronous point-to-point operations as one handle"""

    def __init__(self, works):
        self.works = works

    def wait(self):
        for w in self.works:
            w.wait()
        return True


class DirectStoreTimeout(RuntimeError):
    """a flag barrier of the direct-store corner turn gave up waiting for a peer: the transform that was running has no valid output
    and the ranks' receive buffers are out of step.  ShardedNtt.fall_back_to_exchange() (collective) puts every rank on the
    collective exchange; the caller then repeats the transform."""


class HipFourstep:
    """A rank's share of the sharded transform as ONE library object (sc_fourstep_t, include/starkcore.h): roots, stage shapes,
    the outer-twiddle table and the kernel plans are fixed once; a stage is one ctypes call with pointers only."""

    def __init__(self, sc, log2n, root, rank, world, sptr, log_n1=0):
        import ctypes
        self.sc, self.lib, self.sptr, self.ct = sc, sc.lib(), sptr, ctypes
        self.rank, self.world = rank, world
        h = ctypes.c_void_p()
        sc._check(self.lib.sc_fourstep_create_ex(log2n, _fe(root), rank, world, int(log_n1), ctypes.byref(h)))
        self._h = h
        self.native = False            # sc_comm_init has been called for this world: run() may be used
        self.direct = False            # setup_direct() has mapped every rank's receive region: run_direct() may be used
        self._own_region, self._peer_regions = None, []

    def n1(self):
        rows = self.ct.c_uint64()
        self.sc._check(self.lib.sc_fourstep_shape(self._h, 0, self.ct.byref(rows), None))
        return int(rows.value)

    def setup_direct(self, device, group=None, kind=-1):
        """The direct-store corner turn (sc_fourstep_run_direct_dev): this rank's receive region is created and exported (HIP IPC),
        the 64-byte handles travel once through torch.distributed, every peer's region is mapped.  Collective.  Returns True when
        it is up on EVERY rank; on failure everything this attempt made is released again (the caller may try another kind).
        kind: 1 fine-grained device memory, 0 coarse-grained, -1 the library's default (sc_ipc_region_create_ex).
        STARKCORE_TEST_OPEN_FAILS_KIND=<0|1> (tests): mapping a peer's region of that kind fails, as a runtime that cannot import
        it would."""
        import os
        ct, sc, lib = self.ct, self.sc, self.lib
        G, g = self.world, self.rank
        size = ct.c_uint64()
        sc._check(lib.sc_fourstep_region_bytes(self._h, ct.byref(size)))
        region, handle = ct.c_void_p(), ct.create_string_buffer(64)
        ok = 1 if lib.sc_ipc_region_create_ex(size.value, int(kind), ct.byref(region), handle) == 0 else 0
        got = -1
        if ok:
            self._own_region = region
            v = ct.c_int(-1)
            lib.sc_ipc_region_kind(ct.byref(v))
            got = int(v.value)
        self.direct_kinds = [got]
        regions = [None] * G
        regions[g] = region.value
        if G > 1:
            on_dev = dist.get_backend(group) == "nccl"
            mine = torch.tensor([ok, got] + list(handle.raw), dtype=torch.int32)
            mine = mine.to(device) if on_dev else mine
            parts = [torch.empty_like(mine) for _ in range(G)]
            dist.all_gather(parts, mine, group=group)
            parts = [t.cpu() for t in parts]
            ok = int(all(int(t[0]) == 1 for t in parts))
            self.direct_kinds = [int(t[1]) for t in parts]
            failing = os.environ.get("STARKCORE_TEST_OPEN_FAILS_KIND")
            if ok:
                for h in range(G):
                    if h == g:
                        continue
                    peer = ct.c_void_p()
                    refused = failing is not None and int(parts[h][1]) == int(failing)
                    if refused or lib.sc_ipc_region_open(bytes(int(v) & 255 for v in parts[h][2:].tolist()), ct.byref(peer)) != 0:
                        ok = 0
                        break
                    self._peer_regions.append(peer)
                    regions[h] = peer.value
            flag = torch.tensor([ok], dtype=torch.int32)
            flag = flag.to(device) if on_dev else flag
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            ok = int(flag.item())
        if ok:
            sc._check(lib.sc_fourstep_set_peers(self._h, (ct.c_void_p * G)(*regions)))
            self.direct = True
        else:
            self.release_direct()
        return bool(ok)

    def region_kind(self):
        """'fine-grained' / 'coarse-grained': the kind of device memory the last receive region of this process got"""
        v = self.ct.c_int(-1)
        self.sc._check(self.lib.sc_ipc_region_kind(self.ct.byref(v)))
        return {1: "fine-grained", 0: "coarse-grained"}.get(int(v.value), "none")

    def run_direct(self, inverse, src, dst):
        rc = self.lib.sc_fourstep_run_direct_dev(self._h, inverse, src.data_ptr(), dst.data_ptr(), self.sptr)
        if rc == self.sc.SC_ERR_TIMEOUT:
            raise DirectStoreTimeout(self.lib.sc_last_error().decode())
        self.sc._check(rc)

    def direct_timed_out(self):
        """0, or the number of the first transform whose flag barrier gave up waiting for a peer (a pinned word the barrier kernel
        writes: no copy, no wait -- meaningful for transforms the stream has finished).  Sticky: once it is set, run_direct raises
        DirectStoreTimeout until the set-up is made again."""
        v = self.ct.c_uint64()
        self.sc._check(self.lib.sc_fourstep_direct_status(self._h, self.ct.byref(v)))
        return int(v.value)

    def release_direct(self):
        for peer in self._peer_regions:
            self.lib.sc_ipc_region_close(peer)
        self._peer_regions = []
        if self._own_region is not None:
            self.lib.sc_ipc_region_free(self._own_region)
            self._own_region = None
        self.direct = False

    def cols(self, inverse, src, send, recv_diag):
        self.sc._check(self.lib.sc_fourstep_cols_dev(self._h, inverse, src.data_ptr(), send.data_ptr(), None if recv_diag is None else recv_diag.data_ptr(), self.sptr))

    def rows(self, inverse, recv, dst, q, K, defer):
        self.sc._check(self.lib.sc_fourstep_rows_dev(self._h, inverse, recv.data_ptr(), dst.data_ptr(), q, K, 1 if defer else 0, self.sptr))

    def rows_finish(self, inverse, dst):
        self.sc._check(self.lib.sc_fourstep_rows_finish_dev(self._h, inverse, dst.data_ptr(), self.sptr))

    def run(self, inverse, src, send, recv, dst, K, defer, force_diag):
        self.sc._check(self.lib.sc_fourstep_run_dev(self._h, inverse, src.data_ptr(), send.data_ptr(), recv.data_ptr(), dst.data_ptr(), K, 1 if defer else 0,
                                                    1 if force_diag else 0, self.sptr))

    def __del__(self):
        try:
            if self._h is not None:
                self.release_direct()
                self.lib.sc_fourstep_free(self._h)
        except Exception:      # noqa: BLE001
            pass
        self._h = None


def init_native_comm(rank, world, device, group=None):
    """The library's own RCCL communicator (sc_comm_init), so that the corner turn is issued from C next to the kernels it
    separates (sc_fourstep_run_dev): rank 0 makes the id, torch.distributed carries its 128 bytes to the others.  Returns True
    when the communicator is up on EVERY rank (the ranks agree on the outcome), False when RCCL is not available."""
    import ctypes
    import os
    import starkcore as sc
    lib = sc.lib()
    cand = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    path = cand.encode() if os.path.exists(cand) else None
    buf = ctypes.create_string_buffer(128)
    ok = 1
    if rank == 0 and lib.sc_comm_unique_id(path, buf) != 0:
        ok = 0
    on_dev = world > 1 and dist.get_backend(group) == "nccl"
    t = torch.tensor([ok] + list(buf.raw), dtype=torch.int32)
    if world > 1:
        t = t.to(device) if on_dev else t
        dist.broadcast(t, 0, group=group)
        t = t.cpu()
    if int(t[0]) != 1:
        return False
    ident = bytes(int(v) & 255 for v in t[1:].tolist())
    rc = lib.sc_comm_init(path, ident, rank, world)
    flag = torch.tensor([1 if rc == 0 else 0], dtype=torch.int32)
    if world > 1:
        flag = flag.to(device) if on_dev else flag
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return int(flag.item()) == 1


def destroy_native_comm():
    """tear the library's RCCL communicator down (before the process group goes away); a no-op when there is none"""
    import starkcore as sc
    sc.lib().sc_comm_destroy()


class ShardedNtt:
    def __init__(self, log2n, root, rank, world, device, engine=None, group=None, always_exchange=False, overlap_chunks=1, native_exchange=False,
                 defer_last_pass=True, direct_store=False, log_n1=None):
        assert world & (world - 1) == 0, "world size must be a power of two"
        self.log2n, self.n = log2n, 1 << log2n
        self.root = int(root)
        assert pow(self.root, self.n, P) == 1 and pow(self.root, self.n // 2, P) != 1, "root must be a primitive n-th root"
        self.rank, self.world, self.device, self.group = rank, world, device, group
        # run the exchange even where there is nothing to exchange: the block a rank keeps for itself goes through the collective
        # too (a world of one rank then exercises the whole RCCL path) instead of being written in place by the column stage
        self.always_exchange = always_exchange
        # the corner turn is issued as this many row blocks; the row stage of block q runs while blocks q+1.. are still in
        # flight on the collective's own stream (1 = one blocking exchange)
        self.overlap_chunks = overlap_chunks
        # with row blocks: the second pass of a two-pass row stage runs once over all rows instead of once per block
        self.defer_last_pass = defer_last_pass
        # issue the exchange from C over the library's own RCCL communicator (init_native_comm) instead of torch.distributed
        self.native_exchange = native_exchange
        # n = n1 * n2.  Small domains: square split.  Large ones: n1 = 2^8, so that the column stage of forward() is ONE
        # pass (256-point transforms) and the row stage two, and the other way round for inverse(): 3 passes per
        # transform instead of 4 (measured per-rank compute at 2^21 local elements: 138 us -> see profiles/).
        # log_n1 overrides the split (e.g. 12 at 2^24: the square split -- 8 x fewer, 16 x longer rows per rank and message)
        self.n1 = 1 << (log_n1 if log_n1 else ((log2n + 1) // 2 if log2n <= 16 else 8))
        self.n2 = self.n // self.n1
        assert self.n2 >= world and self.n1 >= world, "domain too small to shard over this many ranks"
        self.root_inv = pow(self.root, self.n - 1, P)
        self.n_inv = pow(self.n, P - 2, P)
        if engine is None:
            cur = torch.cuda.current_stream(device)
            self.stream = cur if cur.cuda_stream != 0 else torch.cuda.Stream(device=device)
            engine = HipEngine(self.stream)
        else:
            self.stream = None
        self.engine = engine
        # stage object: one library (or test-oracle) object per rank that owns roots, shapes and plans; engines without one
        # take the primitive-by-primitive path (_transform_primitives)
        if log_n1 and hasattr(engine, "fourstep"):
            self.stages = engine.fourstep(log2n, self.root, rank, world, log_n1)
        else:
            assert not log_n1 or not hasattr(engine, "fourstep")
            self.stages = engine.fourstep(log2n, self.root, rank, world) if hasattr(engine, "fourstep") else None
        # the corner turn as the column stage's own stores into the peers' receive buffers (HIP IPC; no collective): set up
        # collectively here, used by _transform when it came up on every rank
        self.direct_store = False
        self.corner_turn_setup = []                # what was tried for the corner turn, in order, and how it went (bench: corner_turn_probes)
        if direct_store:
            assert self.stages is not None and hasattr(self.stages, "setup_direct"), "the direct-store corner turn needs the HIP stage object"
            # peers store into a region of this GPU while its kernels poll and read it: fine-grained device memory first (what RCCL
            # uses for the same purpose); a runtime that cannot export or import that kind gets coarse-grained memory; when neither
            # comes up on EVERY rank the transform keeps the collective exchange (RCCL)
            for kind, name in ((1, "fine-grained"), (0, "coarse-grained")):
                up = self.stages.setup_direct(device, group, kind)
                kinds = sorted(set({1: "fine-grained", 0: "coarse-grained"}.get(k, "none") for k in getattr(self.stages, "direct_kinds", [])))
                self.corner_turn_setup.append("direct store, %s regions requested: %s" % (name, ("up on every rank (regions: %s)" % ", ".join(kinds)) if up else "did not come up on every rank"))
                if up:
                    self.direct_store = True
                    break
            if not self.direct_store:
                self.corner_turn_setup.append("collective exchange (no direct-store set-up came up)")
        self._bufs = {}
        self._a2a_single = True
        self.bytes_exchanged = 0                   # bytes this rank has sent through the corner turn so far

    # -- helpers ---------------------------------------------------------------------------------
    def local_shape(self, forward_input=True):
        R, C = (self.n1, self.n2) if forward_input else (self.n2, self.n1)
        return (R, C // self.world, 2)

    def _buf(self, key, shape):
        """persistent work buffers, one per (purpose, shape): forward and inverse alternate between two shapes per purpose and
        neither should go back to the allocator in between"""
        k = (key, tuple(shape))
        b = self._bufs.get(k)
        if b is None:
            b = torch.empty(shape, dtype=torch.int64, device=self.device)
            self._bufs[k] = b
        return b

    def synthetic_input(self, seed=1):
        """This rank's slab of the synthetic vector synth(seed, n) in the forward-input layout."""
        import numpy as np
        import synth
        R, C = self.n1, self.n2
        w = C // self.world
        # row r of the slab = elements r*C + rank*w .. + w
        out = np.empty((R, w, 2), dtype=np.uint64)
        for r in range(R):
            out[r] = synth.synth_packed(seed, w, start=r * C + self.rank * w)
        return torch.from_numpy(out.view(np.int64)).to(self.device)

    def fall_back_to_exchange(self):
        """COLLECTIVE.  After a DirectStoreTimeout on any rank (or a non-zero stages.direct_timed_out()): every rank leaves the
        direct-store form and uses the collective exchange from now on; the caller repeats the transform that failed.  Returns
        whether any rank had seen a timeout."""
        if not getattr(self.stages, "direct", False) and not self.direct_store:
            return False
        if self.stream is not None:
            self.stream.synchronize()
        else:
            torch.cuda.synchronize()
        seen = 1 if (getattr(self.stages, "direct", False) and self.stages.direct_timed_out()) else 0
        if self.world > 1:
            on_dev = dist.get_backend(self.group) == "nccl"
            flag = torch.tensor([seen], dtype=torch.int32)
            flag = flag.to(self.device) if on_dev else flag
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group)
            seen = int(flag.item())
        self.stages.release_direct()
        self.direct_store = False
        self.corner_turn_setup.append("direct store abandoned (%s): collective exchange from here on" % ("a flag barrier timed out" if seen else "at the caller's request"))
        return bool(seen)

    # -- the transform ---------------------------------------------------------------------------
    def _transform(self, src, dst, inverse):
        """cols -> corner turn -> rows for one direction (inverse: the roles of n1 and n2 swap, root^-1, n^-1 in the twiddle)"""
        R, C = (self.n2, self.n1) if inverse else (self.n1, self.n2)
        if self.stages is None:
            return self._transform_primitives(src, dst, R, C, self.root_inv if inverse else self.root, self.n_inv if inverse else 1)
        st, G, K = self.stages, self.world, self.overlap_chunks
        rw, cw = R // G, C // G
        inv = 1 if inverse else 0
        exchange = G > 1 or self.always_exchange
        if exchange:
            self.bytes_exchanged += G * rw * cw * 16 * (G - 1) // G
        if self.direct_store:
            st.run_direct(inv, src, dst)                        # column stage stores into the peers' buffers, flag barrier, row stage
            return
        send = self._buf("send", (G * rw * cw, 2)).view(G, rw, cw, 2)
        recv = self._buf("recv", (G * rw * cw, 2)).view(G, rw, cw, 2)
        if K > 1 and (rw % K or (rw // K) & (rw // K - 1)):
            K = 1
        if not exchange:
            st.cols(inv, src, send, recv)                      # world of one rank: the whole output is the rank's own block
            st.rows(inv, recv, dst, 0, 1, False)
            return
        if self.native_exchange and getattr(st, "native", False):
            st.run(inv, src, send, recv, dst, K, self.defer_last_pass, self.always_exchange)
            return
        # the block a rank keeps for itself is written into `recv` by the column stage: neither copied nor sent
        diag_in_place = not self.always_exchange
        st.cols(inv, src, send, recv if diag_in_place else None)
        if K == 1:
            self._exchange_blocks(recv, send, diag_in_place, async_op=False)
            st.rows(inv, recv, dst, 0, 1, False)
            return
        rk = rw // K
        s5, r5 = send.view(G, K, rk, cw, 2), recv.view(G, K, rk, cw, 2)
        works = [self._exchange_blocks(r5[:, q], s5[:, q], diag_in_place, async_op=True) for q in range(K)]
        for q in range(K):
            works[q].wait()
            st.rows(inv, recv, dst, q, K, self.defer_last_pass)
        if self.defer_last_pass:
            st.rows_finish(inv, dst)

    def _exchange_blocks(self, recv, send, skip_own, async_op):
        """The corner turn of `send` [G][...] into `recv` [G][...]: block h of send -> rank h, block g of recv <- rank g.  With
        skip_own the rank's own block is left alone on both sides.  RCCL: one grouped send/recv straight from / into the
        (possibly strided) blocks.  gloo (functional runs on CPU, or with several ranks sharing one GPU): point-to-point
        messages, device tensors staged through the host.  Returns a handle with wait() when async_op."""
        G, g = self.world, self.rank
        backend = dist.get_backend(self.group)
        if backend == "nccl":
            if not skip_own and not async_op and send.is_contiguous() and recv.is_contiguous():
                dist.all_to_all_single(recv.view(-1), send.view(-1), group=self.group)
                return _DoneWork()
            empty = send.new_empty((0,))
            outs = [empty if (skip_own and h == g) else recv[h] for h in range(G)]
            ins = [empty if (skip_own and h == g) else send[h] for h in range(G)]
            work = dist.all_to_all(outs, ins, group=self.group, async_op=async_op)
            return work if async_op else _DoneWork()
        peers = [h for h in range(G) if h != g]
        if not skip_own:
            recv[g].copy_(send[g])
        staged = send.is_cuda
        if staged:
            torch.cuda.current_stream(self.device).synchronize()
        inbox = {h: torch.empty(recv[h].shape, dtype=recv.dtype) if (staged or not recv[h].is_contiguous()) else recv[h] for h in peers}
        works = [dist.irecv(inbox[h], h, group=self.group) for h in peers]
        works += [dist.isend(send[h].cpu().contiguous() if staged else send[h].contiguous(), h, group=self.group) for h in peers]

        def land():
            for w in works:
                w.wait()
            for h in peers:
                if inbox[h] is not recv[h]:
                    recv[h].copy_(inbox[h])
            return True
        if async_op and not staged:
            done = _Works(works)
            done.wait = land
            return done
        land()
        return _DoneWork()

    # -- the same transform primitive by primitive (engines without a stage object: the CPU oracle engine of the tests) ----------
    def stage_cols(self, src, R, C, root, scale):
        """(1) column transforms (out of place) + (2) outer twiddle with the GLOBAL column index -> [R][C/G]."""
        cw = C // self.world
        a = self._buf("a", (R, cw, 2))
        self.engine.cols_ntt(src, a, R, cw, pow(root, C, P))          # root^C is a primitive R-th root
        self.engine.twiddle(a, R, cw, 0, self.rank * cw, root, self.n, scale)
        return a

    def _all_to_all(self, recv, a):
        """recv[g'] <- rows [rank*rw, (rank+1)*rw) of rank g's slab.  One collective; the list form is only a fallback for
        backends without all_to_all_single."""
        self.bytes_exchanged += a.numel() * 8 * (self.world - 1) // self.world
        if self._a2a_single:
            try:
                _exchange_single(recv, a, self.group)
                return
            except (RuntimeError, NotImplementedError):
                self._a2a_single = False
        G = self.world
        if a.is_cuda and dist.get_backend(self.group) == "gloo":
            host_recv, host_a = torch.empty(recv.shape, dtype=recv.dtype), a.cpu()
            dist.all_to_all(list(host_recv.view(G, -1).unbind(0)), list(host_a.view(G, -1).unbind(0)), group=self.group)
            recv.copy_(host_recv)
            return
        dist.all_to_all(list(recv.view(G, -1).unbind(0)), list(a.view(G, -1).unbind(0)), group=self.group)

    def assemble_rows(self, recv, R, C):
        G = self.world
        rw, cw = R // G, C // G
        rows = self._buf("rows", (rw, G, cw, 2))
        rows.copy_(recv.view(G, rw, cw, 2).permute(1, 0, 2, 3))       # column block g' came from rank g'
        return rows.view(rw, C, 2)

    def stage_rows(self, rows, dst, R, C, root):
        """(4) row transforms of length C, transposed output [C][R/G]."""
        self.engine.rows_ntt_t(rows, dst, C, R // self.world, pow(root, R, P))

    def _transform_primitives(self, src, dst, R, C, root, scale):
        G = self.world
        cw, rw = C // G, R // G
        a = self.stage_cols(src, R, C, root, scale)
        if G == 1 and not self.always_exchange:
            self.stage_rows(a, dst, R, C, root)
            return
        recv = self._buf("recv", (G, rw, cw, 2))
        self._all_to_all(recv, a)
        self.stage_rows(self.assemble_rows(recv, R, C), dst, R, C, root)

    def slab_of(self, coeffs, key="slab_of"):
        """This rank's column slab [n1][n2/G] (zero-padded) of a coefficient vector `coeffs` [m][2] that is REPLICATED on every
        rank (polynomials are 1/blowup of the domain): coefficient j sits at row j // n2, column j % n2."""
        m = coeffs.shape[0]
        R, C, G = self.n1, self.n2, self.world
        assert m <= self.n and tuple(coeffs.shape) == (m, 2)
        cw = C // G
        x = self._buf(key, (R, cw, 2))
        x.zero_()
        full_rows, rest = divmod(m, C)
        lo = self.rank * cw
        if full_rows:
            x[:full_rows] = coeffs[:full_rows * C].view(full_rows, C, 2)[:, lo:lo + cw]
        if rest > lo:
            take = min(rest - lo, cw)
            x[full_rows, :take] = coeffs[full_rows * C + lo:full_rows * C + lo + take]
        return x

    def coset_scale(self, x_local, factor, out=None, rows=None):
        """x[j] * factor^j on a slab in the forward-input layout (Polynomial.scale, code/univariate.py:153-154)"""
        out = x_local if out is None else out
        R, C, cw = self.n1, self.n2, self.n2 // self.world
        rows = R if rows is None else rows
        if rows < R and out is not x_local:
            out[rows:].copy_(x_local[rows:])
        if rows:
            self._run(lambda: self.engine.scale_slab(x_local, out, rows, cw, C, self.rank * cw, int(factor)))
        return out

    def coset_evaluate(self, coeffs, offset, y_local):
        """Sharded fast_coset_evaluate (code/ntt.py:132-135): `coeffs` [m][2] is the WHOLE coefficient vector, replicated on
        every rank (it is only 1/blowup of the domain); each rank keeps the columns of its slab, scales them by offset^j,
        zero-pads to n1 rows and runs forward().  y_local [n2][n1/G] receives this rank's slab of the codeword on
        { offset * root^i }."""
        m = coeffs.shape[0]
        x = self.slab_of(coeffs.contiguous(), "lde_x")
        self.coset_scale(x, offset, rows=min(self.n1, -(-m // self.n2)) if m else 0)
        self.forward(x, y_local)

    def multiply(self, a_local, b_local, out_local):
        """Sharded fast_multiply core (code/ntt.py:58-64) on coefficient slabs [n1][n2/G] (zero-padded to the transform
        length by the caller, like ntt.py:51-56): forward both, Hadamard product on the slab -- the pointwise stage needs no
        exchange, every rank owns the same index set of both operands -- and inverse.  Two all-to-alls in, one out."""
        fa = self._buf("mul_a", self.local_shape(False))
        fb = self._buf("mul_b", self.local_shape(False))
        self.forward(a_local, fa)
        self.forward(b_local, fb)
        self._run(lambda: self.engine.pointwise_mul(fa, fb, fa, fa.numel() // 2))
        self.inverse(fa, out_local)

    def coset_divide(self, a_local, b_local, offset, out_local):
        """Sharded fast_coset_divide core (code/ntt.py:159-176) on coefficient slabs [n1][n2/G]: scale both by offset^j, forward,
        pointwise division on the slab (a zero of the divisor on the coset raises "divide by zero" like algebra.py:92),
        inverse, unscale by offset^-j.  The quotient's coefficients come back in the same slab layout."""
        sa = self.coset_scale(a_local, offset, self._buf("div_sa", self.local_shape(True)))
        sb = self.coset_scale(b_local, offset, self._buf("div_sb", self.local_shape(True)))
        fa = self._buf("div_a", self.local_shape(False))
        fb = self._buf("div_b", self.local_shape(False))
        self.forward(sa, fa)
        self.forward(sb, fb)
        self.divide_values(fa, fb, fa)
        self.inverse(fa, out_local)
        self.coset_scale(out_local, pow(int(offset), P - 2, P))

    def divide_values(self, a, b, out):
        """out = a / b pointwise on slabs of values (code/ntt.py:172); a zero of the divisor raises the reference's "divide by zero"
        on EVERY rank: it may sit in another rank's slab, so the ranks agree on the outcome (4 bytes) before anyone enters the next
        collective"""
        failed = 0
        try:
            self._run(lambda: self.engine.pointwise_div(a, b, out, a.numel() // 2))
        except AssertionError:
            failed = 1
        if self.world > 1:
            backend = dist.get_backend(self.group)
            flag = torch.tensor([failed], dtype=torch.int32, device=self.device if backend == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group)
            failed = int(flag.item())
        assert(not failed), "divide by zero"

    def forward(self, x_local, y_local):
        """x_local [n1][n2/G] -> y_local [n2][n1/G]  (column slab of X[k2*n1 + k1])."""
        self._run(lambda: self._transform(x_local, y_local, False))

    def inverse(self, y_local, x_local):
        """y_local [n2][n1/G] -> x_local [n1][n2/G]; uses root^-1 and folds n^-1 into the outer twiddle (ntt.py:27-30)."""
        self._run(lambda: self._transform(y_local, x_local, True))

    def _run(self, fn):
        # (the raw getter: a tenth of the cost of building a torch.cuda.Stream object per transform)
        if self.stream is not None and _current_raw_stream(self.device) != self.stream.cuda_stream:
            self.stream.wait_stream(torch.cuda.current_stream(self.device))
            try:
                with torch.cuda.stream(self.stream):
                    fn()
            finally:        # also when fn raises ("divide by zero"): later work on the current stream stays ordered behind the engine's
                torch.cuda.current_stream(self.device).wait_stream(self.stream)
        else:
            fn()


def gather_natural(local, n_rows, n_cols, world, group=None):
    """All ranks: assemble the full natural-order vector from column slabs [n_rows][n_cols/world] (tests only)."""
    parts = [torch.empty_like(local) for _ in range(world)]
    if world > 1:
        dist.all_gather(parts, local.contiguous(), group=group)
    else:
        parts = [local]
    return torch.cat(parts, dim=1).reshape(n_rows * n_cols, 2)


# =====================================================================================================================
# FRI over the column-slab layout
# =====================================================================================================================
def _current_raw_stream(device):
    """torch's current stream on `device` as a raw hipStream_t value (0 = the null stream); the raw getter costs a tenth of
    building a torch.cuda.Stream object per engine call"""
    getter = getattr(torch._C, "_cuda_getCurrentRawStream", None)
    if getter is not None:
        index = device.index if device.index is not None else torch.cuda.current_device()
        return int(getter(index))
    return int(torch.cuda.current_stream(device).cuda_stream)


# FRI, Merkle commitments and openings on these layouts live in sharded_fri.py (ShardedFri, ContiguousFri, ColumnReplicas,
# HipFriEngine); the names stay importable from here
def __getattr__(name):
    if name in ("HipFriEngine", "ShardedFri", "ContiguousFri", "ColumnReplicas", "_LayerEntries"):
        import sharded_fri
        return getattr(sharded_fri, name)
    raise AttributeError("module 'sharded' has no attribute %r" % name)
