#!/usr/bin/env python3
"""Two INDEPENDENT forward+inverse transforms in flight on two HIP streams against the same work on one stream (dev tool, round 6).
A 2^20 pass is one workgroup per CU whose load burst, arithmetic and store drain do not overlap (74 % of the VALU floor);
every way of putting a second workgroup of the SAME transform on a CU lost.  Two different transforms -- the registers of a trace --
fill each other's gaps: a CU whose workgroup of one transform has finished takes a workgroup of the other instead of waiting for the
slowest workgroup of its own kernel.   python tools/two_stream_ntt.py [log2n=20] [pairs=400] [streams=2]"""
import ctypes, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "stark-anatomy_amd"))
import numpy as np, torch
import starkcore as sc, synth
from workloads import nth_root

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 400
S = int(sys.argv[3]) if len(sys.argv) > 3 else 2
n = 1 << log2n
sc.init(0); lib = sc.lib(); dev = torch.device("cuda", 0)
root = sc.fe_bytes(nth_root(n))
streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
bufs = []
for k in range(S):
    x = torch.from_numpy(synth.synth_packed(1 + k, n).view(np.int64)).to(dev)
    bufs.append((x, torch.empty_like(x), torch.empty_like(x)))
torch.cuda.synchronize()


def pair(k, s):
    x, y, z = bufs[k]
    p = ctypes.c_void_p(s.cuda_stream)
    sc._check(lib.sc_ntt_dev(x.data_ptr(), y.data_ptr(), n, root, 0, p))
    sc._check(lib.sc_ntt_dev(y.data_ptr(), z.data_ptr(), n, root, 1, p))


def run(two):
    for i in range(40):
        pair(i % S, streams[(i % S) if two else 0])
    torch.cuda.synchronize()
    best = None
    for _ in range(5):
        t0 = time.perf_counter()
        for i in range(pairs):
            pair(i % S, streams[(i % S) if two else 0])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    return best


one, two = run(False), run(True)
ok = all(torch.equal(z, x) for x, y, z in bufs)
print("2^%d fwd+inv pairs, %d of them: one stream %.2f us per pair (%.2f G el/s), %d streams %.2f us per pair (%.2f G el/s): %+.1f %%   round trips %s"
      % (log2n, pairs, one / pairs * 1e6, 2 * n * pairs / one / 1e9, S, two / pairs * 1e6, 2 * n * pairs / two / 1e9, 100 * (one / two - 1), "ok" if ok else "WRONG"))
