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
import torch
import torch.distributed as dist

P = 1 + 407 * (1 << 119)


def _fe(v):
    return int(v).to_bytes(16, "little")


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

    # fused variants (one kernel sequence each; no separate twiddle pass, no reassembly copy)
    def cols_ntt_twiddled(self, src, dst, length, batch, root, outer_root, order, col_base, scale_ninv):
        rc = self.lib.sc_ntt_batch_ex_dev(src.data_ptr(), dst.data_ptr(), length, batch, 0, _fe(root), _fe(outer_root), order, col_base,
                                          1 if scale_ninv else 0, 1, self.sptr)
        if rc == -7:            # SC_ERR_UNSUPPORTED shape: caller falls back to the unfused steps
            return False
        self.sc._check(rc)
        return True

    def rows_ntt_t_chunked(self, src, dst, length, batch, chunks, root):
        rc = self.lib.sc_ntt_batch_ex_dev(src.data_ptr(), dst.data_ptr(), length, batch, 1, _fe(root), None, 0, 0, 0, chunks, self.sptr)
        if rc == -7:
            return False
        self.sc._check(rc)
        return True


class ShardedNtt:
    def __init__(self, log2n, root, rank, world, device, engine=None, group=None, always_exchange=False):
        assert world & (world - 1) == 0, "world size must be a power of two"
        self.log2n, self.n = log2n, 1 << log2n
        self.root = int(root)
        assert pow(self.root, self.n, P) == 1 and pow(self.root, self.n // 2, P) != 1, "root must be a primitive n-th root"
        self.rank, self.world, self.device, self.group = rank, world, device, group
        self.always_exchange = always_exchange     # run the all-to-all even for a world of one rank (exercises the RCCL path)
        # n = n1 * n2.  Small domains: square split.  Large ones: n1 = 2^8, so that the column stage of forward() is ONE
        # pass (256-point transforms) and the row stage two, and the other way round for inverse(): 3 passes per
        # transform instead of 4 (measured per-rank compute at 2^21 local elements: 138 us -> see profiles/).
        self.n1 = 1 << ((log2n + 1) // 2 if log2n <= 16 else 8)
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
        self._bufs = {}
        self.launches_per_transform = None

    # -- helpers ---------------------------------------------------------------------------------
    def local_shape(self, forward_input=True):
        R, C = (self.n1, self.n2) if forward_input else (self.n2, self.n1)
        return (R, C // self.world, 2)

    def _buf(self, key, shape):
        b = self._bufs.get(key)
        if b is None or tuple(b.shape) != tuple(shape):
            b = torch.empty(shape, dtype=torch.int64, device=self.device)
            self._bufs[key] = b
        return b

    def synthetic_input(self, seed=1):
        """This rank's slab of the synthetic vector synth(seed, n) in the forward-input layout."""
        import numpy as np
        import synth
        R, C = self.n1, self.n2
        w = C // self.world
        full_rows = []
        # row r of the slab = elements r*C + rank*w .. + w
        out = np.empty((R, w, 2), dtype=np.uint64)
        for r in range(R):
            out[r] = synth.synth_packed(seed, w, start=r * C + self.rank * w)
        return torch.from_numpy(out.view(np.int64)).to(self.device)

    # -- the transform ---------------------------------------------------------------------------
    def stage_cols(self, src, R, C, root, scale):
        """(1) column transforms (out of place) + (2) outer twiddle with the GLOBAL column index -> [R][C/G]."""
        cw = C // self.world
        a = self._buf("a", (R, cw, 2))
        self.engine.cols_ntt(src, a, R, cw, pow(root, C, P))          # root^C is a primitive R-th root
        self.engine.twiddle(a, R, cw, 0, self.rank * cw, root, self.n, scale)
        return a

    def exchange(self, a, R, C):
        """(3) corner turn: rank h receives rows [h*R/G, (h+1)*R/G) of every rank's slab -> [R/G][C]."""
        G = self.world
        if G == 1 and not self.always_exchange:
            return a
        rw, cw = R // G, C // G
        recv = self._buf("recv", (G, rw, cw, 2))
        dist.all_to_all_single(recv.view(-1), a.view(-1), group=self.group)
        return self.assemble_rows(recv, R, C)

    def assemble_rows(self, recv, R, C):
        G = self.world
        rw, cw = R // G, C // G
        rows = self._buf("rows", (rw, G, cw, 2))
        rows.copy_(recv.view(G, rw, cw, 2).permute(1, 0, 2, 3))       # column block g' came from rank g'
        return rows.view(rw, C, 2)

    def stage_rows(self, rows, dst, R, C, root):
        """(4) row transforms of length C, transposed output [C][R/G]."""
        self.engine.rows_ntt_t(rows, dst, C, R // self.world, pow(root, R, P))

    def _transform(self, src, dst, R, C, root, scale):
        eng, G = self.engine, self.world
        fused = hasattr(eng, "cols_ntt_twiddled")
        cw, rw = C // G, R // G
        # (1)+(2) column transforms with the outer twiddle in their store epilogue
        a = None
        if fused:
            a = self._buf("a", (R, cw, 2))
            if not eng.cols_ntt_twiddled(src, a, R, cw, pow(root, C, P), root, self.n, self.rank * cw, scale != 1):
                a = None
        if a is None:
            a = self.stage_cols(src, R, C, root, scale)
        # (3) corner turn
        if G == 1 and not self.always_exchange:
            self.stage_rows(a, dst, R, C, root)
            return
        recv = self._buf("recv", (G, rw, cw, 2))
        dist.all_to_all_single(recv.view(-1), a.view(-1), group=self.group)
        # (4) row transforms straight from the chunked layout the all-to-all left behind
        if fused and eng.rows_ntt_t_chunked(recv, dst, C, rw, G, pow(root, R, P)):
            return
        self.stage_rows(self.assemble_rows(recv, R, C), dst, R, C, root)

    def forward(self, x_local, y_local):
        """x_local [n1][n2/G] -> y_local [n2][n1/G]  (column slab of X[k2*n1 + k1])."""
        self._run(lambda: self._transform(x_local, y_local, self.n1, self.n2, self.root, 1))

    def inverse(self, y_local, x_local):
        """y_local [n2][n1/G] -> x_local [n1][n2/G]; uses root^-1 and folds n^-1 into the outer twiddle (ntt.py:27-30)."""
        self._run(lambda: self._transform(y_local, x_local, self.n2, self.n1, self.root_inv, self.n_inv))

    def _run(self, fn):
        if self.stream is not None and torch.cuda.current_stream(self.device).cuda_stream != self.stream.cuda_stream:
            self.stream.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self.stream):
                fn()
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
