#!/usr/bin/env python3
"""sc_coset_evaluate_columns_dev (the LDE of several polynomials in one set of launches) against sc_coset_evaluate_dev per column (dev tool).
   python tools/lde_columns.py [log2m=18] [blowup_log=3] [cols=8] [reps=30]      (BASELINE configs[2] per column: 2^18 -> 2^21)"""
import ctypes, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "stark-anatomy_amd"))
import numpy as np, torch
import starkcore as sc, synth
from workloads import nth_root
logm = int(sys.argv[1]) if len(sys.argv) > 1 else 18
blow = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cols = int(sys.argv[3]) if len(sys.argv) > 3 else 8
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 30
m, n = 1 << logm, 1 << (logm + blow)
sc.init(0); lib = sc.lib(); dev = torch.device("cuda", 0)
root, off = sc.fe_bytes(nth_root(n)), sc.fe_bytes(synth.GENERATOR if hasattr(synth, "GENERATOR") else 85408008396924667383611388730472331217)
x = torch.from_numpy(synth.synth_packed(11, m * cols).view(np.int64).reshape(-1)).to(dev)
y, y1 = torch.empty(2 * n * cols, dtype=torch.int64, device=dev), torch.empty(2 * n * cols, dtype=torch.int64, device=dev)
s = torch.cuda.Stream(device=dev); sp = ctypes.c_void_p(s.cuda_stream)


def one_at_a_time():
    for c in range(cols):
        sc._check(lib.sc_coset_evaluate_dev(x.data_ptr() + 16 * m * c, m, off, root, n, y1.data_ptr() + 16 * n * c, sp))


def batch():
    sc._check(lib.sc_coset_evaluate_columns_dev(x.data_ptr(), m, cols, off, root, n, y.data_ptr(), sp))


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = None
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        best = dt if best is None or dt < best else best
    return best


one_at_a_time(); batch(); torch.cuda.synchronize()
ok = torch.equal(y, y1)
a, b = timed(one_at_a_time), timed(batch)
print("LDE 2^%d -> 2^%d x %d columns: one at a time %.1f us per column, columns call %.1f us per column (%+.0f %%)   equal: %s"
      % (logm, logm + blow, cols, a * 1e6 / cols, b * 1e6 / cols, 100 * (a / b - 1), "ok" if ok else "WRONG"))
