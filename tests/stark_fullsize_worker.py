"""Worker for tests/test_gpu_fullsize.py::test_stark_prover_full_size_two_ranks: one rank of a world_size-N job whose ranks all use
GPU 0 (gloo, host-staged exchange -- a one-GPU box cannot host an RCCL job), proving BASELINE configs[4] at its stated size:
sharded_stark.ShardedFastStark.prove on the synthetic 2-register AIR, FRI domain 2^log_fri, from a seeded os.urandom stream.
Prints the SHA-256 of the proof it ended with; the parent compares it with the single-GPU prover's (fast_stark.FastStark)."""
import hashlib
import os
import random
import sys
import time

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "stark-anatomy_amd"), REPO):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    log_fri, seed = int(sys.argv[1]), int(sys.argv[2])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ["STARKCORE_DEVICE"] = "0"
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    import starkcore as sc
    sc.init(0)
    import workloads
    import fast_stark
    from fast_stark import DeviceTrace
    from sharded_stark import ShardedFastStark
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s = 40
    field, T, packed, air, boundary = workloads.synthetic_stark_instance(log_fri, s)
    trace = DeviceTrace.from_packed(packed, field)
    fast_stark.os.urandom = random.Random(seed).randbytes        # (rank 0's draws are the ones every rank uses: broadcast)
    stark = ShardedFastStark(field, 4, s, 2 * s, 2, T, rank, world, dev)
    assert stark.fri_domain_length == 1 << log_fri
    tz, layer, root = stark.preprocess(device_resident=True)
    t0 = time.perf_counter()
    proof = stark.prove(trace, air, boundary, tz, layer)
    torch.cuda.synchronize()
    seconds = time.perf_counter() - t0
    print("rank %d world %d fri 2^%d proof_sha256 %s proof_len %d zerofier_root %s first_proof_s %.3f" %
          (rank, world, log_fri, hashlib.sha256(proof).hexdigest(), len(proof), root.hex()[:16], seconds), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
