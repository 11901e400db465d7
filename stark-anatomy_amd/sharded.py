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


class HipFriEngine:
    """Local FRI primitives on torch-owned slabs through the C-ABI (folds, Merkle trees, openings)."""

    def __init__(self, device):
        import ctypes
        import starkcore as sc
        self.sc, self.lib, self.device, self.ctypes = sc, sc.lib(), device, ctypes

    class _Tree:
        def __init__(self, tree, keep):
            self.tree, self.keep = tree, keep

        @property
        def root(self):
            return self.tree.root          # waits for an asynchronous build

        def open(self, indices):
            return self.tree.open_batch(list(indices))

    def _stream(self):
        """The primitives run on torch's CURRENT stream, so they are ordered with the tensor ops and collectives around them and
        need no synchronization of their own.  Only under torch's null stream (which the library cannot share) they run on the
        library stream between two explicit synchronizations."""
        raw = _current_raw_stream(self.device)
        if raw == 0:
            torch.cuda.current_stream(self.device).synchronize()
            return None
        return self.ctypes.c_void_p(raw)

    def _done(self, sptr):
        if sptr is None:
            self.sc.synchronize()

    def tree(self, elems, need_root=True):
        """Merkle tree over a contiguous tensor of field elements [..., 2].  need_root=False: the build is only enqueued on the
        current stream (a local subtree of a sharded commit: its sub-root level is copied out on the same stream, its own root
        is never looked at)."""
        elems = elems.contiguous()
        sptr = self._stream()
        if need_root or sptr is None:
            return HipFriEngine._Tree(self.sc.MerkleTree.from_device_ptr(elems.data_ptr(), elems.numel() // 2, sptr), elems)
        return HipFriEngine._Tree(self.sc.MerkleTree.from_device_ptr_noroot(elems.data_ptr(), elems.numel() // 2, sptr), elems)

    def level(self, tree, level):
        count = tree.tree.n >> level
        out = torch.empty((count, 8), dtype=torch.int64, device=self.device)
        sptr = self._stream()
        tree.tree.copy_level(level, out.data_ptr(), sptr)
        self._done(sptr)
        return out

    def tree_from_digests(self, digests):
        digests = digests.contiguous()
        return HipFriEngine._Tree(self.sc.MerkleTree.from_digests_ptr(digests.data_ptr(), digests.numel() // 8, self._stream()), digests)

    def fold_slab(self, src, rows, cols, R, col_base, alpha, offset, omega):
        dst = torch.empty((rows // 2, cols, 2), dtype=torch.int64, device=self.device)
        sptr = self._stream()
        self.sc._check(self.lib.sc_fri_fold_slab_dev(src.data_ptr(), rows, cols, R, col_base, _fe(alpha), _fe(offset), _fe(omega), dst.data_ptr(), sptr))
        self._done(sptr)
        return dst

    def fold_slab_tree(self, src, rows, cols, R, col_base, alpha, offset, omega):
        """fold_slab AND the enqueue-only local subtree over the folded slab (the next round's `tree(slab, need_root=False)`) in
        one library call: the tree's leaf stage computes the fold.  Returns (folded slab, tree); tree None under torch's null
        stream (the caller builds it the plain way)."""
        sptr = self._stream()
        if sptr is None:
            return self.fold_slab(src, rows, cols, R, col_base, alpha, offset, omega), None
        dst = torch.empty((rows // 2, cols, 2), dtype=torch.int64, device=self.device)
        tree = self.sc.MerkleTree.from_folded_slab(src.data_ptr(), rows, cols, R, col_base, _fe(alpha), _fe(offset), _fe(omega), dst.data_ptr(), sptr)
        return dst, HipFriEngine._Tree(tree, dst)

    def fold_full(self, src, N, alpha, offset, omega):
        dst = torch.empty((N // 2, 2), dtype=torch.int64, device=self.device)
        sptr = self._stream()
        self.sc._check(self.lib.sc_fri_fold_dev(src.data_ptr(), N, _fe(alpha), _fe(offset), _fe(omega), dst.data_ptr(), sptr))
        self._done(sptr)
        return dst

    class _LibraryVector:
        """a folded codeword the library handed out (sc_fri_commit_dev), behind the two things the layer records ask of a tensor"""

        def __init__(self, vec):
            self.vec = vec

        def data_ptr(self):
            return self.vec.ptr

        def contiguous(self):
            return self

    def commit_rounds(self, full, N, offset, omega, rounds, prior):
        """The remaining `rounds` rounds of the commit phase on a codeword every rank holds whole (`full`, N elements): trees,
        Fiat-Shamir steps and folds in ONE library call (sc_fri_commit_dev, what Fri.commit uses on one GPU).  prior: the byte
        strings in the proof stream so far.  [(codeword, tree, root)] per round, or None when the library does not take the
        transcript (or under torch's null stream): the caller's round loop runs then."""
        sptr = self._stream()
        if sptr is None:
            return None
        ct, sc = self.ctypes, self.sc
        full = full.contiguous()
        k = len(prior)
        vecs = (ct.c_void_p * max(1, rounds - 1))()
        trees = (ct.c_void_p * rounds)()
        roots = ct.create_string_buffer(64 * rounds)
        alphas = (ct.c_uint64 * max(2, 2 * (rounds - 1)))()
        rc = self.lib.sc_fri_commit_dev(full.data_ptr(), N, _fe(offset), _fe(omega), rounds, b"".join(prior), (ct.c_uint32 * max(1, k))(*map(len, prior)), k,
                                        vecs, trees, roots, alphas, sptr)
        if rc == sc.SC_ERR_UNSUPPORTED:
            return None
        sc._check(rc)
        out, raw = [], roots.raw
        for r in range(rounds):
            n = N >> r
            vec = full if r == 0 else HipFriEngine._LibraryVector(sc.DeviceVector.adopt(vecs[r - 1], n))
            root = raw[64 * r:64 * r + 64]
            out.append((vec, HipFriEngine._Tree(sc.MerkleTree(ct.c_void_p(trees[r]), root, n), vec), root))
        return out

    def lde(self, coeffs, offset, generator, order):
        """fast_coset_evaluate (code/ntt.py:132-135) of packed coefficients (bytes) -> device tensor [order][2]"""
        m = len(coeffs) // 16
        src = self.sc.DeviceVector.from_bytes(coeffs) if m else self.sc.DeviceVector(1)
        out = torch.empty((order, 2), dtype=torch.int64, device=self.device)
        sptr = self._stream()
        self.sc._check(self.lib.sc_coset_evaluate_dev(src.ptr, m, _fe(offset), _fe(generator), order, out.data_ptr(), sptr))
        torch.cuda.current_stream(self.device).synchronize()      # `src` is freed on return: its reader must be done
        self._done(sptr)
        return out

    def query_many(self, requests, raw_paths=False, raw_values=False):
        """[(tree, elems tensor or None, indices[, keep])] -> [(values as ints or None, authentication paths)]: every opening of
        every layer in ONE library call and one launch (sc_merkle_query_multi_dev), instead of a device round trip per tree.
        keep: only the first `keep` digests of each path are wanted (the part below a sharded commitment's sub-roots).
        raw_paths: the paths of a request come back as ONE uint8 array [openings][64 * digests] instead of lists of bytes
        objects (the sharded openings join two such parts per path before any object is made).  raw_values: the opened elements
        come back as a uint8 array [openings][16] (packed residues, as the device wrote them) instead of Python ints."""
        import numpy as np
        ct, sc = self.ctypes, self.sc
        requests = [(r[0], r[1], r[2], r[3] if len(r) > 3 else None) for r in requests]
        live = [(q, t, e, idx, keep) for q, (t, e, idx, keep) in enumerate(requests) if len(idx)]
        no_values = np.zeros((0, 16), dtype=np.uint8) if raw_values else []
        if raw_paths:
            out = [(None if e is None else no_values, np.zeros((0, 0), dtype=np.uint8)) for t, e, idx, _ in requests]
        else:
            out = [(None if e is None else no_values, [[] for _ in idx]) for t, e, idx, _ in requests]
        if not live:
            return out
        torch.cuda.current_stream(self.device).synchronize()        # the library call runs on the library's stream
        n = len(live)
        counts = [len(idx) for _, _, _, idx, _ in live]
        flat = np.fromiter(itertools.chain.from_iterable(idx for _, _, _, idx, _ in live), dtype=np.uint64, count=sum(counts))
        total = int(flat.size)
        depths = [t.tree.depth for _, t, _, _, _ in live]
        path_bytes = sum(64 * d * k for d, k in zip(depths, counts))
        elems_out = np.empty((total, 16), dtype=np.uint8)            # (every byte is written by the call: nothing to zero first)
        paths_out = np.empty(max(path_bytes, 64), dtype=np.uint8)
        # a tree over digests has no element vector: any readable pointer will do, the value is not used
        ptrs = [(e if e is not None else t.keep).data_ptr() for _, t, e, _, _ in live]
        sc._check(self.lib.sc_merkle_query_multi_dev(n, (ct.c_void_p * n)(*[t.tree._h for _, t, _, _, _ in live]), (ct.c_void_p * n)(*ptrs),
                                                     flat.ctypes.data_as(ct.POINTER(ct.c_uint64)), (ct.c_uint64 * n)(*counts),
                                                     elems_out.ctypes.data_as(ct.c_void_p), paths_out.ctypes.data_as(ct.c_void_p)))
        values = elems_out if raw_values else sc.unpack(elems_out.tobytes(), total)
        view = memoryview(paths_out)
        vo = po = 0
        for (q, t, e, idx, keep), d, k in zip(live, depths, counts):
            if raw_paths:
                whole = paths_out[po:po + 64 * k * d].reshape(k, 64 * d)
                paths = whole if keep is None or keep >= d else whole[:, :64 * keep]
            else:
                paths = sc._path_lists(view, po, d, k, keep)
            out[q] = (values[vo:vo + k] if e is not None else None, paths)
            vo += k
            po += 64 * k * d
        return out

    def read(self, elems, flat_indices):
        """values (Python ints) of elems.view(-1, 2)[flat_indices]"""
        if len(flat_indices) == 0:
            return []
        if isinstance(elems, HipFriEngine._LibraryVector):
            torch.cuda.current_stream(self.device).synchronize()        # the copy below runs on the library's stream
            values = self.sc.unpack(elems.vec.to_bytes(), elems.vec.n)
            return [values[i] for i in flat_indices]
        idx = torch.tensor(list(flat_indices), dtype=torch.int64, device=elems.device)
        got = elems.reshape(-1, 2)[idx].cpu().tolist()
        m = (1 << 64) - 1
        return [((hi & m) << 64) | (lo & m) for lo, hi in got]


class _LayerEntries:
    """the entries of one committed layer as FieldElement objects, each made once (pickle memoises by identity: the `c` of one
    round is the `a` / `b` of the next, code/fri.py:104-105) -- the object cache of a layer record behind the interface
    proof_objects' segments use"""
    _full = None

    def __init__(self, layer, field):
        # (only the cache: a reference to the layer record would close a cycle layer -> holder -> layer, and the layer's trees --
        # gigabytes at a 2^24 domain -- would then wait for the cycle collector instead of going back to the pool with the proof)
        self.cache, self.field = layer["cache"], field

    def _entries(self, indices, values):
        from algebra import FieldElement
        cache, field, new = self.cache, self.field, object.__new__
        for i, v in zip(indices, values):
            if i not in cache:
                e = new(FieldElement)
                e.value = v
                e.field = field
                cache[i] = e
        return [cache[i] for i in indices]


class ShardedFri:
    """`Fri.prove` (reference code/fri.py:115-130) on a codeword that lives in the column-slab layout.

    The codeword of length N = C*R is the row-major C x R matrix (index i = row*R + col); rank g owns the columns
    [g*R/G, (g+1)*R/G) as a contiguous [C][R/G] tensor -- exactly what ShardedNtt.forward() leaves behind.  In this layout
      * split-and-fold needs NO exchange: i and i + N/2 are rows `row` and `row + C/2` of the same columns;
      * a Merkle commit needs ONE all-gather of C digests per rank: the bottom log2(R/G) levels are whole subtrees of the
        rank's slab, the levels above are rebuilt (redundantly, identically) from the gathered sub-roots on every rank;
      * when a fold leaves a single row (length R) the codeword is all-gathered once and the remaining rounds run locally.
    Every rank drives the same Fiat-Shamir transcript (roots are replicated), so alphas and query indices agree without
    any broadcast, and every rank ends up with the identical, reference-identical proof stream.
    """

    # A 2^16-leaf tree is two latency-bound launches (0.066 ms) whatever the number of ranks; its sharded form -- local
    # subtree, level copy, all-gather, top tree -- is four steps of the same kind plus a collective.  Above 2^17 nodes the hashing is
    # throughput-bound and sharding pays.
    LOCAL_TAIL = 1 << 16

    one_rank_local = False      # (set per instance below; subclasses with their own constructor keep every round sharded)

    def __init__(self, fri, R, rank, world, device, engine=None, group=None, local_tail=None):
        """local_tail: once a round's codeword is this short it is gathered on every rank and the remaining rounds run locally
        (replicated): such rounds are bound by launch and hashing latency on any number of ranks, a collective per round only
        adds to it, and on the HIP engine the rest of the commit phase is then ONE library call (0: only when one row is left)."""
        self.fri, self.R, self.rank, self.world, self.device, self.group = fri, int(R), rank, world, device, group
        assert R % world == 0 and fri.domain_length % R == 0
        self.Rw = self.R // world
        self.engine = engine if engine is not None else HipFriEngine(device)
        self.local_tail = self.LOCAL_TAIL if local_tail is None else int(local_tail)
        # ONE rank (and no explicit local_tail, which asks for the slab rounds): the "slab" is the whole codeword in natural order --
        # one tree per commitment instead of a subtree plus a tree over its sub-roots, no sub-root level copied out, nothing
        # gathered, and the whole commit phase is one library call.  What a rank pays for the sharded layout when nobody shares it.
        self.one_rank_local = world == 1 and local_tail is None

    # -- collectives ------------------------------------------------------------------------------
    def _all_gather(self, t):
        t = t.contiguous()
        if self.world == 1:
            return t.unsqueeze(0)
        if t.is_cuda and dist.get_backend(self.group) == "gloo":        # functional tests with several ranks on one GPU: host-staged
            host = t.cpu()
            parts = [torch.empty_like(host) for _ in range(self.world)]
            dist.all_gather(parts, host, group=self.group)
            return torch.stack(parts, dim=0).to(t.device)
        if t.is_cuda:                                                   # RCCL: one output tensor, no per-rank pieces to stack
            out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
            dist.all_gather_into_tensor(out, t, group=self.group)
            return out
        parts = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(parts, t, group=self.group)
        return torch.stack(parts, dim=0)

    def _gather_answers(self, layout, mine, sizes, packed=False):
        """The owners' answers to the openings, merged with ONE fixed-shape tensor collective (no pickling, no object store).
        layout[r] = [(q, positions, ndigests)]: the runs rank r answers (one per request it owns something of), in the order it
        packs them -- every rank derives all of it from the public indices; `mine` = this rank's runs [(values, bottoms)] in that
        order, bottoms = uint8 array [openings][64 * ndigests]; sizes[q] = number of openings of request q.  Returns
        {q: (values, bottoms)}: a list and a uint8 array, both indexed by the position in the request.  No object per digest is
        made here: the caller joins these path bottoms with the path tops first.  packed: the values are uint8 arrays
        [openings][16] (packed residues) on the way in and on the way out, and no integer object is made either."""
        import numpy as np
        import starkcore as sc
        G, g = self.world, self.rank
        answers = {}

        def place(q, positions, vals, bottoms):
            if len(positions) == sizes[q]:                       # one owner for the whole request: its run IS the answer
                answers[q] = (vals, bottoms)
                return
            have = answers.get(q)
            if have is None:
                have = answers[q] = (np.zeros((sizes[q], 16), dtype=np.uint8) if packed else [None] * sizes[q],
                                     np.zeros((sizes[q], bottoms.shape[1]), dtype=np.uint8))
            if packed:
                have[0][positions] = vals
            else:
                for pos, v in zip(positions, vals):
                    have[0][pos] = v
            have[1][positions] = bottoms

        if G == 1:
            for (q, positions, nd), (vals, bottoms) in zip(layout[0], mine):
                place(q, positions, vals, bottoms)
            return answers
        words = [sum(len(positions) * (2 + 8 * nd) for _, positions, nd in layout[r]) for r in range(G)]
        width = max(max(words), 1)
        row = np.zeros(width, dtype=np.int64)
        at = 0
        for (q, positions, nd), (vals, bottoms) in zip(layout[g], mine):
            k = len(positions)
            block = np.empty((k, 2 + 8 * nd), dtype=np.int64)
            block[:, :2] = np.ascontiguousarray(vals).view(np.int64) if packed else np.frombuffer(sc.pack(vals), dtype=np.int64).reshape(k, 2)
            if nd:
                block[:, 2:] = np.ascontiguousarray(bottoms).view(np.int64)
            row[at:at + block.size] = block.reshape(-1)
            at += block.size
        assert at == words[g]
        on_device = self.device.type == "cuda" and dist.get_backend(self.group) != "gloo"
        t = torch.from_numpy(row).to(self.device) if on_device else torch.from_numpy(row)
        if on_device:
            out = torch.empty((G, width), dtype=torch.int64, device=self.device)
            dist.all_gather_into_tensor(out, t, group=self.group)
        else:
            parts = [torch.empty_like(t) for _ in range(G)]
            dist.all_gather(parts, t, group=self.group)
            out = torch.stack(parts, dim=0)
        rows = out.cpu().numpy()
        for r in range(G):
            at = 0
            for q, positions, nd in layout[r]:
                k = len(positions)
                block = rows[r, at:at + k * (2 + 8 * nd)].reshape(k, 2 + 8 * nd)
                at += block.size
                vals = np.ascontiguousarray(block[:, :2])
                vals = vals.view(np.uint8) if packed else sc.unpack(vals.tobytes(), k)
                place(q, positions, vals, np.ascontiguousarray(block[:, 2:]).view(np.uint8))
        return answers

    @staticmethod
    def _joined_paths(bottoms, tops):
        """authentication paths (lists of fresh 64-byte objects, merkle.py:16-27) from their two parts, each a uint8 array
        [openings][64 * digests]: the part below the sub-roots (from the owner's local subtree) and the part above (from the
        replicated top tree); one pass over one buffer makes all the objects"""
        import numpy as np
        import starkcore as sc
        k = bottoms.shape[0]
        if tops is not None and tops.shape[0] == k and tops.shape[1]:
            bottoms = np.concatenate((bottoms, tops), axis=1)
        depth = bottoms.shape[1] // 64
        return sc._path_lists(memoryview(np.ascontiguousarray(bottoms)).cast("B"), 0, depth, k)

    # -- layers -----------------------------------------------------------------------------------
    def _commit_sharded(self, slab, C, local=None):
        """local: the rank's subtree over `slab` if the fold that produced the slab has built it already (fold_slab_tree)"""
        eng, G, Rw = self.engine, self.world, self.Rw
        if self.one_rank_local and hasattr(eng, "query_many"):
            # one rank: its "slab" is the whole codeword in natural order and its subtree the whole tree -- ONE tree, no sub-root
            # level copied out, no gather, no second tree above it (a quarter of a world-1 proof's commitments otherwise)
            full = slab.reshape(C * self.R, 2)
            tree = local if local is not None else eng.tree(full)
            return {"kind": "local", "vec": full, "tree": tree, "root": tree.root, "length": C * self.R, "cache": {}}
        if local is None:
            local = eng.tree(slab, need_root=False)
        sub_level = Rw.bit_length() - 1
        sub = eng.level(local, sub_level)                                   # [C][8]: one sub-root per row
        top_leaves = self._all_gather(sub).permute(1, 0, 2).contiguous()    # natural order: node (row, rank)
        top = eng.tree_from_digests(top_leaves.reshape(C * G, 8))
        return {"kind": "sharded", "slab": slab, "C": C, "local": local, "top": top, "root": top.root, "length": C * self.R, "cache": {}}

    def commit(self, slab, C):
        """Merkle.commit (code/merkle.py:13-14) of a codeword held as column slabs [C][R/G]: local subtrees + one all-gather
        of C sub-roots per rank; returns the layer record `_open` answers openings from (its "root" is the commitment)."""
        return self._commit_sharded(slab, C)

    def _natural(self, slab, C):
        """the whole codeword in natural order on every rank: [C][R/G] slabs -> [C*R]"""
        return self._all_gather(slab).permute(1, 0, 2, 3).reshape(C * self.R, 2).contiguous()

    def _open_many_raw(self, requests):
        """[(values, paths)] for a list of (layer, global indices).  ONE library call for everything this rank can answer
        (values + the bottoms of the paths of the columns it owns, from its local subtrees; the tops of all paths from the
        replicated top trees) and ONE collective to merge the owners' answers."""
        eng = self.engine
        R, Rw, G, g = self.R, self.Rw, self.world, self.rank
        sub_level = Rw.bit_length() - 1
        asks, where = [], []
        for q, (layer, indices) in enumerate(requests):
            if layer["kind"] == "local":
                where.append(("local", len(asks)))
                asks.append((layer["tree"], layer["vec"], list(indices)))
                continue
            mine = [i for i in indices if (i % R) // Rw == g] if G > 1 else indices
            where.append(("sharded", len(asks)))
            asks.append((layer["local"], layer["slab"], [(i // R) * Rw + (i % R) % Rw for i in mine], sub_level))
            asks.append((layer["top"], None, [(i // R) * G + (i % R) // Rw for i in indices] if layer["C"] * G > 1 else []))
        got = eng.query_many(asks, raw_paths=True)
        layout, mine, sizes = [[] for _ in range(G)], [], [len(indices) for _, indices in requests]
        for q, ((layer, indices), w) in enumerate(zip(requests, where)):
            if w[0] == "local" or not indices:
                continue
            if G == 1:
                layout[0].append((q, range(len(indices)), sub_level))
                mine.append(got[w[1]])
                continue
            owners = [[] for _ in range(G)]
            for pos, i in enumerate(indices):
                owners[(i % R) // Rw].append(pos)
            for r in range(G):
                if owners[r]:
                    layout[r].append((q, owners[r], sub_level))
            if owners[g]:
                mine.append(got[w[1]])
        answers = self._gather_answers(layout, mine, sizes) if any(layout) else {}
        out = []
        for q, ((layer, indices), w) in enumerate(zip(requests, where)):
            if w[0] == "local":
                vals, paths = got[w[1]]
                out.append((vals, self._joined_paths(paths, None) if layer["length"] > 1 and len(indices) else [[] for _ in indices]))
                continue
            if q not in answers:
                out.append(([], []))
                continue
            vals, bottoms = answers[q]
            out.append((vals, self._joined_paths(bottoms, got[w[1] + 1][1] if layer["C"] * G > 1 else None)))     # below the sub-roots + above them
        return out

    def _open_many_arrays(self, requests):
        """[(packed residues (bytes), paths as a uint8 array [openings][64 * depth])] for a list of (layer, global indices): what
        _open_many_raw gathers, with the two parts of every path joined as arrays and NO object made (proof_objects' segments)"""
        import numpy as np
        import starkcore as sc
        eng = self.engine
        R, Rw, G, g = self.R, self.Rw, self.world, self.rank
        sub_level = Rw.bit_length() - 1
        asks, where = [], []
        for q, (layer, indices) in enumerate(requests):
            if layer["kind"] == "local":
                where.append(("local", len(asks)))
                asks.append((layer["tree"], layer["vec"], list(indices)))
                continue
            mine = [i for i in indices if (i % R) // Rw == g] if G > 1 else indices
            where.append(("sharded", len(asks)))
            asks.append((layer["local"], layer["slab"], [(i // R) * Rw + (i % R) % Rw for i in mine], sub_level))
            asks.append((layer["top"], None, [(i // R) * G + (i % R) // Rw for i in indices] if layer["C"] * G > 1 else []))
        got = eng.query_many(asks, raw_paths=True, raw_values=True)           # residues stay packed bytes from the device to the proof
        layout, mine, sizes = [[] for _ in range(G)], [], [len(indices) for _, indices in requests]
        for q, ((layer, indices), w) in enumerate(zip(requests, where)):
            if w[0] == "local" or not len(indices):
                continue
            if G == 1:
                layout[0].append((q, range(len(indices)), sub_level))
                mine.append(got[w[1]])
                continue
            owners = [[] for _ in range(G)]
            for pos, i in enumerate(indices):
                owners[(i % R) // Rw].append(pos)
            for r in range(G):
                if owners[r]:
                    layout[r].append((q, owners[r], sub_level))
            if owners[g]:
                mine.append(got[w[1]])
        answers = self._gather_answers(layout, mine, sizes, packed=True) if any(layout) else {}
        out = []
        for q, ((layer, indices), w) in enumerate(zip(requests, where)):
            k = len(indices)
            if w[0] == "local":
                vals, paths = got[w[1]]
                depth = layer["length"].bit_length() - 1
                paths = np.ascontiguousarray(paths).reshape(k, 64 * depth) if k and depth else np.zeros((k, 0), dtype=np.uint8)
            elif q not in answers:
                vals, paths = np.zeros((0, 16), dtype=np.uint8), np.zeros((0, 0), dtype=np.uint8)
            else:
                vals, bottoms = answers[q]
                tops = got[w[1] + 1][1] if layer["C"] * G > 1 else None
                paths = np.concatenate((bottoms, tops), axis=1) if tops is not None and tops.shape[0] == k and tops.shape[1] else bottoms
            out.append((np.ascontiguousarray(vals).tobytes(), np.ascontiguousarray(paths)))
        return out

    @staticmethod
    def _holder(layer, field):
        """the layer's entries as proof_objects' segments want them: a field, an identity, FieldElements made once per index"""
        holder = layer.get("holder")
        if holder is None:
            holder = layer["holder"] = _LayerEntries(layer, field)
        return holder

    def _open_many(self, requests):
        """entries as FieldElement objects (one object per index and layer, reused) + fresh path objects per request"""
        from algebra import FieldElement
        res, field, new = [], self.fri.field, object.__new__
        for (layer, indices), (values, paths) in zip(requests, self._open_many_raw(requests)):
            cache = layer["cache"]
            for i, v in zip(indices, values):
                if i not in cache:
                    # FieldElement(v, field) without the call into __init__ (algebra.py:16-18 sets exactly these two attributes),
                    # as in starkcore.DeviceCodeword._entries
                    e = new(FieldElement)
                    e.value = v
                    e.field = field
                    cache[i] = e
            res.append(([cache[i] for i in indices], paths))
        return res

    def _open(self, layer, indices):
        return self._open_many([(layer, indices)])[0]

    # -- the protocol -----------------------------------------------------------------------------
    def prove(self, slab, proof_stream, also_open=None):
        """also_open (a fri.AlsoOpen whose `requests` returns (layer records, index lists)): further committed layers opened in the
        same library call and collective as the query phase"""
        from algebra import FieldElement
        fr, eng, field = self.fri, self.engine, self.fri.field
        N, R, Rw = fr.domain_length, self.R, self.Rw
        C = N // R
        assert tuple(slab.shape) == (C, Rw, 2), "slab must be this rank's [C][R/G] columns"
        omega, offset, rounds = fr.omega, fr.offset, fr.num_rounds()
        layers, cur, full, local = [], slab, None, None
        if self.one_rank_local:
            top = self._prove_one_rank(slab, proof_stream, also_open)
            if top is not None:
                return top
        # fri.py:68 in every round: omega_r^(N_r) == 1 with omega_r = omega^(2^r), N_r = N / 2^r -- one condition, checked once
        if hasattr(fr, "_check_omega_order"):
            fr._check_omega_order(N)                         # (once per Fri instance: a power and an inversion in Python integers)
        else:
            assert(omega ^ (N - 1) == omega.inverse()), "error in commit: omega does not have the right order!"
        for r in range(rounds):
            Nr = N >> r
            if full is None and (C == 1 or Nr <= self.local_tail or (self.one_rank_local and hasattr(eng, "commit_rounds"))):      # (one rank: nothing to gather, the slab IS the codeword)
                full = self._natural(cur, C)                # one row left / a short codeword: collect it everywhere, go local
                rest = self._commit_tail(full, Nr, offset, omega, rounds - r, proof_stream)
                if rest is not None:                        # ... and the library ran every remaining round in one call
                    layers.extend(rest)
                    break
            if full is None:
                layer = self._commit_sharded(cur, C, local)
            else:
                tree = eng.tree(full)
                layer = {"kind": "local", "vec": full, "tree": tree, "root": tree.root, "length": Nr, "cache": {}}
            layers.append(layer)
            proof_stream.push(layer["root"])
            if r == rounds - 1:
                break
            alpha = field.sample(proof_stream.prover_fiat_shamir())
            if full is None:
                # the folded slab is committed to as a slab again (not gathered): fold + local subtree in one call
                if C > 2 and (Nr >> 1) > self.local_tail and hasattr(eng, "fold_slab_tree"):
                    cur, local = eng.fold_slab_tree(cur, C, Rw, R, self.rank * Rw, alpha.value, offset.value, omega.value)
                else:
                    cur, local = eng.fold_slab(cur, C, Rw, R, self.rank * Rw, alpha.value, offset.value, omega.value), None
                C //= 2
            else:
                full = eng.fold_full(full, Nr, alpha.value, offset.value, omega.value)
            omega = omega ^ 2
            offset = offset ^ 2
        # last codeword in the clear (fri.py:91): natural order, plain list; its objects are reused by the last query round
        last_layer = layers[-1]
        last_vec = last_layer["vec"] if last_layer["kind"] == "local" else self._natural(cur, C)
        last_values = eng.read(last_vec, range(last_layer["length"]))
        lazy = None
        if hasattr(eng, "query_many"):                      # (the CPU test engines answer with objects)
            import proof_objects as _po
            lazy = _po.lazy_objects(proof_stream)
        if lazy is not None:
            # described, not built (proof_objects): the transcript bytes are the same, no object per element / digest
            import starkcore as _scm
            lazy.add(_po.ElementList(self._holder(last_layer, field), _scm.pack(last_values)))
            return self._query_all_lazy(layers, len(last_values), proof_stream, lazy, also_open)
        last_list = [FieldElement(v, field) for v in last_values]
        last_layer["cache"] = dict(enumerate(last_list))
        proof_stream.push(last_list)

        return self._query_all(layers, last_list, proof_stream)

    def _prove_one_rank(self, slab, proof_stream, also_open):
        """ONE rank on the HIP engine: the slab is the codeword in natural order, so Fri.prove's one-call form (sc_fri_prove_dev:
        commit phase, index sampling and every opening -- the caller's committed layers included -- in one library call) serves
        it as it serves a single GPU's prover; the slab's memory is handed over as it is (DeviceVector.wrap).  None when a
        precondition of that form does not hold (the rounds then run as on any number of ranks)."""
        eng = self.engine
        if not isinstance(eng, HipFriEngine) or not slab.is_cuda or not slab.is_contiguous():
            return None
        import starkcore as sc
        from fri import AlsoOpen
        if _current_raw_stream(self.device) != sc.library_stream():
            return None                                    # (the one-call form runs on the library's stream)
        N = self.fri.domain_length
        inner = None
        if also_open is not None:
            more, shift = getattr(also_open, "layers", None), getattr(also_open, "shift", None)
            if more is None or shift is None or not all(layer["kind"] == "local" and isinstance(layer["tree"], HipFriEngine._Tree) and layer["length"] == N for layer in more):
                return None
            codewords = []
            for layer in more:
                cw = layer.get("codeword")
                if cw is None:
                    vec = layer["vec"]
                    cw = layer["codeword"] = sc.DeviceCodeword(sc.DeviceVector.wrap(vec.data_ptr(), N, vec), self.fri.field)
                    cw._tree = layer["tree"].tree
                codewords.append(cw)
            inner = AlsoOpen(None, codewords=codewords, shift=shift)
        codeword = sc.DeviceCodeword(sc.DeviceVector.wrap(slab.data_ptr(), N, slab), self.fri.field)
        top = self.fri._prove_in_library(codeword, proof_stream, inner)
        if top is not None and also_open is not None:
            also_open.answers = inner.answers
            also_open.position_arrays = inner.position_arrays
        return top

    def _commit_tail(self, full, Nr, offset, omega, rounds_left, proof_stream):
        """the remaining rounds of the commit phase on the gathered codeword through the engine's whole-loop call, when there is
        one and the proof stream qualifies (fri.library_transcript: what Fri.commit checks on one GPU); layer records or None"""
        eng = self.engine
        if not hasattr(eng, "commit_rounds") or Nr < 2:
            return None
        from fri import library_transcript
        prior = library_transcript(proof_stream, rounds_left)
        if prior is None:
            return None
        got = eng.commit_rounds(full, Nr, offset.value, omega.value, rounds_left, prior)
        if got is None:
            return None
        rest = []
        for k, (vec, tree, root) in enumerate(got):
            proof_stream.push(root)
            rest.append({"kind": "local", "vec": vec, "tree": tree, "root": root, "length": Nr >> k, "cache": {}})
        return rest

    def _query_requests(self, layers, last_length, proof_stream):
        """top-level indices from the transcript and what every layer has to open (fri.py:119-128)"""
        fr = self.fri
        N, s = fr.domain_length, fr.num_colinearity_tests
        top_level_indices = fr.sample_indices(proof_stream.prover_fiat_shamir(), N // 2, last_length, s)
        nq = len(layers) - 1
        per_round, indices = [], [i for i in top_level_indices]
        for i in range(nq):
            indices = [index % (layers[i]["length"] // 2) for index in indices]
            per_round.append(indices)
        requests = []
        for j, layer in enumerate(layers):
            request = []
            if j < nq:
                request += per_round[j][:s] + [index + layer["length"] // 2 for index in per_round[j][:s]]
            if j > 0:
                request += per_round[j - 1][:s]
            requests.append((layer, request))
        return top_level_indices, per_round, requests

    def _query_all_lazy(self, layers, last_length, proof_stream, lazy, also_open=None):
        """_query_all with the owners' answers pushed as they are (proof_objects.FriRound)"""
        import proof_objects as _po
        field, s = self.fri.field, self.fri.num_colinearity_tests
        top_level_indices, per_round, requests = self._query_requests(layers, last_length, proof_stream)
        if also_open is not None:
            more_layers, more_indices = also_open.requests(top_level_indices)
            requests = requests + list(zip(more_layers, more_indices))
        fetched = self._open_many_arrays(requests)
        if also_open is not None:
            also_open.answers = fetched[len(layers):]
        nq = len(layers) - 1
        for i in range(nq):
            values, paths = fetched[i]
            next_values, next_paths = fetched[i + 1]
            c_at = 2 * s if i + 1 < nq else 0
            a = per_round[i][:s]
            half = layers[i]["length"] // 2
            lazy.add(_po.FriRound(self._holder(layers[i], field), self._holder(layers[i + 1], field), a, [index + half for index in a], a,
                                  values[:16 * s], values[16 * s:32 * s], next_values[16 * c_at:16 * (c_at + s)],
                                  paths[:s], paths[s:2 * s], next_paths[c_at:c_at + s]))
        return top_level_indices

    def _query_all(self, layers, last_list, proof_stream):
        """the query phase of fri.py:124-128 over the committed layers: indices from the transcript, ONE collective for every
        opening of every round, pushes in the reference's order"""
        fr = self.fri
        N = fr.domain_length
        s = fr.num_colinearity_tests
        top_level_indices = fr.sample_indices(proof_stream.prover_fiat_shamir(), N // 2, len(last_list), s)
        nq = len(layers) - 1
        per_round, indices = [], [i for i in top_level_indices]
        for i in range(nq):
            indices = [index % (layers[i]["length"] // 2) for index in indices]
            per_round.append(indices)
        requests = []
        for j, layer in enumerate(layers):
            request = []
            if j < nq:
                request += per_round[j][:s] + [index + layer["length"] // 2 for index in per_round[j][:s]]
            if j > 0:
                request += per_round[j - 1][:s]
            requests.append((layer, request))
        fetched = self._open_many(requests)                 # one collective for the whole query phase
        # pushes in the reference's order (fri.py:104-113 per round: s triples, then per test the paths of a, b, c); a ProofStream's
        # `push` is `objects.append`, so a whole round goes in with two list extensions (as in fri.Fri._query_all)
        from ip import ProofStream
        objects = proof_stream.objects if type(proof_stream) is ProofStream else None
        for i in range(nq):
            entries, paths = fetched[i]
            next_entries, next_paths = fetched[i + 1]
            c_at = 2 * s if i + 1 < nq else 0
            triples = list(zip(entries[:s], entries[s:2 * s], next_entries[c_at:c_at + s]))
            openings = [p for trio in zip(paths[:s], paths[s:2 * s], next_paths[c_at:c_at + s]) for p in trio]
            if objects is not None:
                objects.extend(triples)
                objects.extend(openings)
            else:
                for obj in triples + openings:
                    proof_stream.push(obj)
        return top_level_indices


class ContiguousFri(ShardedFri):
    """`Fri.prove` (reference code/fri.py:115-130) on a codeword in the NATURAL contiguous layout (SURVEY.md 8(e), row "FRI
    fold"): rank g owns x[g*N/G : (g+1)*N/G] -- a single-GPU LDE cut into G pieces, or a host list scattered in order.
      * split-and-fold pairs i with i + N/2, i.e. rank g with rank g + G/2: ONE neighbour exchange per fold.  The upper rank
        ships its slab to its partner, which folds both; the folded codeword (half as long) lives contiguously on the lower
        half of the ranks, and so on until one rank holds what is left;
      * Merkle leaves are contiguous: a commit is the active ranks' local subtrees plus one all-gather of their sub-roots
        (64 bytes per rank); the levels above are rebuilt, identically, on every rank;
      * every rank -- also one that has run out of data -- follows the same transcript, so alphas and query indices agree without
        a broadcast; an opening is answered by the rank that owns the leaf, all of them merged by one collective.
    ShardedFri (column slabs: no element exchange at all) is the better layout for a codeword that comes out of ShardedNtt; this
    class serves codewords that arrive in natural order without re-slabbing them (rows_to_column_slab is the other option)."""

    def __init__(self, fri, rank, world, device, engine=None, group=None):
        self.fri, self.rank, self.world, self.device, self.group = fri, rank, world, device, group
        assert world & (world - 1) == 0 and fri.domain_length % world == 0 and fri.domain_length // world >= 1
        self.engine = engine if engine is not None else HipFriEngine(device)
        self.elements_shipped = 0

    def _ship(self, t, src, dst, count):
        """the neighbour exchange: `count` elements from rank src to rank dst (returns the received tensor on dst)"""
        staged = t is not None and t.is_cuda and dist.get_backend(self.group) == "gloo"     # functional tests: host-staged
        if self.rank == src:
            dist.send(t.cpu() if staged else t.contiguous(), dst, group=self.group)
            self.elements_shipped += count
            return None
        buf = torch.empty((count, 2), dtype=torch.int64, device="cpu" if (self.device.type == "cuda" and dist.get_backend(self.group) == "gloo") else self.device)
        dist.recv(buf, src, group=self.group)
        return buf.to(self.device)

    def _commit_contiguous(self, cur, length, active):
        eng = self.engine
        seg = length // active
        mine = self.rank < active
        local = eng.tree(cur, need_root=False) if mine else None
        sub = eng.level(local, seg.bit_length() - 1) if mine else torch.zeros((1, 8), dtype=torch.int64, device=self.device)
        top = eng.tree_from_digests(self._all_gather(sub)[:active].reshape(active, 8))
        return {"kind": "contiguous", "vec": cur, "local": local, "top": top, "root": top.root, "seg": seg, "active": active, "length": length, "cache": {}}

    def commit(self, slab, length, active=None):
        """Merkle.commit (code/merkle.py:13-14) of a codeword of `length` held contiguously by the first `active` ranks"""
        return self._commit_contiguous(slab, length, self.world if active is None else active)

    def _open_many_raw(self, requests):
        eng, g = self.engine, self.rank
        asks = []
        for layer, indices in requests:
            seg = layer["seg"]
            mine = [i % seg for i in indices if i // seg == g]
            asks.append((layer["local"], layer["vec"], mine, seg.bit_length() - 1) if mine else (None, None, []))
            asks.append((layer["top"], None, [i // seg for i in indices] if layer["active"] > 1 else []))
        got = eng.query_many(asks, raw_paths=True)
        layout, mine, sizes = [[] for _ in range(self.world)], [], [len(indices) for _, indices in requests]
        for q, (layer, indices) in enumerate(requests):
            seg = layer["seg"]
            owners = [[] for _ in range(self.world)]
            for pos, i in enumerate(indices):
                owners[i // seg].append(pos)
            for r in range(self.world):
                if owners[r]:
                    layout[r].append((q, owners[r], seg.bit_length() - 1))
            if owners[g]:
                mine.append(got[2 * q])
        answers = self._gather_answers(layout, mine, sizes)
        out = []
        for q, (layer, indices) in enumerate(requests):
            if q not in answers:
                out.append(([], []))
                continue
            vals, bottoms = answers[q]
            out.append((vals, self._joined_paths(bottoms, got[2 * q + 1][1] if layer["active"] > 1 else None)))
        return out

    def prove(self, slab, proof_stream):
        from algebra import FieldElement
        fr, eng, field, G = self.fri, self.engine, self.fri.field, self.world
        N = fr.domain_length
        assert tuple(slab.shape) == (N // G, 2), "slab must be this rank's N/G consecutive elements"
        omega, offset, rounds = fr.omega, fr.offset, fr.num_rounds()
        layers, cur, active = [], slab, G
        assert(omega ^ (N - 1) == omega.inverse()), "error in commit: omega does not have the right order!"     # every round's fri.py:68
        for r in range(rounds):
            Nr = N >> r
            layer = self._commit_contiguous(cur, Nr, active)
            layers.append(layer)
            proof_stream.push(layer["root"])
            if r == rounds - 1:
                break
            alpha = field.sample(proof_stream.prover_fiat_shamir())
            seg = Nr // active
            if active == 1:
                if self.rank == 0:
                    cur = eng.fold_full(cur, Nr, alpha.value, offset.value, omega.value)
            else:
                half = active // 2
                if self.rank < half:
                    upper = self._ship(None, self.rank + half, self.rank, seg)
                    pair = torch.cat([cur, upper], dim=0)
                    # the rank's outputs are i in [rank*seg, (rank+1)*seg): a fold of length 2*seg whose domain starts at omega^(rank*seg)
                    shifted = (offset * (omega ^ (self.rank * seg))).value
                    cur = eng.fold_full(pair, 2 * seg, alpha.value, shifted, omega.value)
                elif self.rank < active:
                    self._ship(cur, self.rank, self.rank - half, seg)
                    cur = None
                active = half
            omega = omega ^ 2
            offset = offset ^ 2
        # last codeword in the clear (fri.py:91): natural order, plain list; its objects are reused by the last query round
        last_layer = layers[-1]
        seg = last_layer["seg"]
        part = cur if self.rank < active else torch.zeros((seg, 2), dtype=torch.int64, device=self.device)
        last_vec = self._all_gather(part)[:active].reshape(last_layer["length"], 2)
        last_list = [FieldElement(v, field) for v in eng.read(last_vec, range(last_layer["length"]))]
        last_layer["cache"] = dict(enumerate(last_list))
        proof_stream.push(last_list)
        return self._query_all(layers, last_list, proof_stream)


# =====================================================================================================================
# Independent columns: one register per GPU
# =====================================================================================================================
class ColumnReplicas:
    """SURVEY.md 8(e), last row: the registers of a STARK (fast_stark.py:103-105, :113: one `fast_coset_evaluate` + one
    `Merkle.commit` per trace / quotient column) are independent units.  When the domain is too small to be worth sharding,
    column i goes to rank i % world: no element ever crosses a link, only the 64-byte roots are all-gathered (every rank
    needs all of them, in column order, for the Fiat-Shamir transcript)."""

    def __init__(self, rank, world, device, engine=None, group=None):
        self.rank, self.world, self.device, self.group = rank, world, device, group
        self.engine = engine if engine is not None else HipFriEngine(device)

    def lde_and_commit(self, columns, offset, generator, order):
        """columns: list of packed coefficient lists (bytes), identical on every rank.
        Returns (mine, roots): mine = {column index: (codeword tensor, tree)} for this rank's columns; roots = the Merkle
        roots of ALL columns in column order."""
        mine = {}
        for i, coeffs in enumerate(columns):
            if i % self.world == self.rank:
                codeword = self.engine.lde(coeffs, offset, generator, order)
                mine[i] = (codeword, self.engine.tree(codeword))
        # every rank needs all the roots, in column order: a [columns][64] byte table in which each rank fills the rows of its own
        # columns, summed over the ranks (the rows are disjoint) -- one fixed-shape tensor collective, nothing pickled
        import numpy as np
        table = np.zeros((len(columns), 64), dtype=np.int32)
        for i, (_, t) in mine.items():
            table[i] = np.frombuffer(t.root, dtype=np.uint8)
        if self.world > 1:
            on_dev = self.device.type == "cuda" and dist.get_backend(self.group) != "gloo"
            tt = torch.from_numpy(table).to(self.device) if on_dev else torch.from_numpy(table)
            dist.all_reduce(tt, op=dist.ReduceOp.SUM, group=self.group)
            table = tt.cpu().numpy()
        return mine, [bytes(table[i].astype(np.uint8)) for i in range(len(columns))]
