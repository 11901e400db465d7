"""Worker for tests/test_sharded_cpu.py: one rank of a world_size-N gloo job on CPU.  The local compute stages
are served by the oracle (OracleEngine) so that the data-movement logic of sharded.py (slab layout, global twiddle
indices, all-to-all corner turn, reassembly, transposed output) is what gets tested."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "stark-anatomy_amd"), REPO):
    if p not in sys.path:
        sys.path.insert(0, p)

from oracle import py_oracle as po          # noqa: E402
import synth                                 # noqa: E402
from sharded import ShardedNtt, gather_natural, P   # noqa: E402


class OracleEngine:
    def _np(self, t):
        return t.numpy().view(np.uint64)

    def cols_ntt(self, src, dst, length, batch, root):
        a = self._np(src).reshape(length, batch, 2)
        out = self._np(dst).reshape(length, batch, 2)
        for c in range(batch):
            out[:, c, :] = np.frombuffer(po.C.ntt(root, np.ascontiguousarray(a[:, c, :]).tobytes(), length), dtype=np.uint64).reshape(length, 2)

    def rows_ntt_t(self, src, dst, length, batch, root):
        a = self._np(src).reshape(batch, length, 2)
        out = self._np(dst).reshape(length, batch, 2)
        for r in range(batch):
            out[:, r, :] = np.frombuffer(po.C.ntt(root, np.ascontiguousarray(a[r]).tobytes(), length), dtype=np.uint64).reshape(length, 2)

    def twiddle(self, buf, rows, cols, row_base, col_base, root, order, scale):
        a = self._np(buf).reshape(rows, cols, 2)
        for r in range(rows):
            for c in range(cols):
                v = int(a[r, c, 0]) | (int(a[r, c, 1]) << 64)
                v = v * pow(root, (row_base + r) * (col_base + c), P) * scale % P
                a[r, c, 0], a[r, c, 1] = v & ((1 << 64) - 1), v >> 64


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ok = True
    for log2n in (6, 7, 10):
        n = 1 << log2n
        root = po.primitive_nth_root(n)
        eng = ShardedNtt(log2n, root, rank, world, torch.device("cpu"), engine=OracleEngine())
        x = eng.synthetic_input(seed=3)
        assert tuple(x.shape) == eng.local_shape(True)
        y = torch.empty(eng.local_shape(False), dtype=torch.int64)
        z = torch.empty_like(x)
        eng.forward(x, y)
        eng.inverse(y, z)
        full_in = synth.synth_packed(3, n).tobytes()
        # input slabs really are the column slabs of the natural vector
        got_in = gather_natural(x, eng.n1, eng.n2, world).numpy().tobytes()
        got = gather_natural(y, eng.n2, eng.n1, world).numpy().tobytes()
        ok &= got_in == full_in
        ok &= got == po.C.ntt(root, full_in, n)
        ok &= torch.equal(z, x)
    dist.barrier()
    dist.destroy_process_group()
    if not ok:
        sys.exit(3)
    print("rank", rank, "ok")


if __name__ == "__main__":
    main()
