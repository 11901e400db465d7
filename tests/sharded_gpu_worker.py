"""Worker for tests/test_gpu_sharded.py::test_two_ranks_share_one_gpu: one rank of a world_size-N job whose ranks all use GPU 0.
The local stages run on the HIP engine (the product path); only the exchange differs from production: backend gloo with a
host-staged all-to-all instead of RCCL (a single-GPU box cannot host an RCCL job).  Checks the sharded forward / inverse / LDE
against the oracle's transform of the full vector."""
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


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ["STARKCORE_DEVICE"] = "0"
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    import starkcore as sc
    sc.init(0)
    from sharded import ShardedNtt, gather_natural
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ok = True
    for log2n in (12, 17, 20):
        n = 1 << log2n
        root = po.primitive_nth_root(n)
        eng = ShardedNtt(log2n, root, rank, world, dev, always_exchange=True)
        x = eng.synthetic_input(seed=3)
        y = torch.empty(eng.local_shape(False), dtype=torch.int64, device=dev)
        z = torch.empty_like(x)
        eng.forward(x, y)
        eng.inverse(y, z)
        torch.cuda.synchronize()
        full_in = synth.synth_packed(3, n).tobytes()
        got = gather_natural(y.cpu(), eng.n2, eng.n1, world).numpy().tobytes()
        want = po.C.ntt(root, full_in, n)
        ok &= got == want
        ok &= torch.equal(z, x)
        m = n // 8 + 3
        coeffs = synth.synth_packed(9, m)
        lde = torch.empty(eng.local_shape(False), dtype=torch.int64, device=dev)
        eng.coset_evaluate(torch.from_numpy(coeffs.view(np.int64).copy()).to(dev), po.GENERATOR, lde)
        torch.cuda.synchronize()
        got_lde = gather_natural(lde.cpu(), eng.n2, eng.n1, world).numpy().tobytes()
        ok &= got_lde == po.C.coset_evaluate(coeffs.tobytes(), m, po.GENERATOR, root, n)
        if not ok:
            print("rank", rank, "MISMATCH at log2n", log2n, flush=True)
            break
    dist.barrier()
    dist.destroy_process_group()
    if not ok:
        sys.exit(3)
    print("rank", rank, "ok")


if __name__ == "__main__":
    main()
