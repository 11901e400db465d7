#!/usr/bin/env python3
"""sc_ntt_columns_dev (a batch of independent columns in ONE set of launches) against the same columns one at a time on one stream
and alternating over two streams (dev tool, round 6).   python tools/columns_ntt.py [log2n=20] [cols=16] [reps=20]
Every figure is forward + inverse of all the columns; the round trip and column-by-column equality with sc_ntt_dev are checked."""
import ctypes, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "stark-anatomy_amd"))
import numpy as np, torch
import starkcore as sc, synth
from workloads import nth_root

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 16
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
n = 1 << log2n
sc.init(0); lib = sc.lib(); dev = torch.device("cuda", 0)
for kv in os.environ.get("TUNE", "").split(","):           # TUNE=prio_balance=1,xcd_remap=0
    if "=" in kv:
        sc.set_tuning(kv.split("=")[0], int(kv.split("=")[1]))
root = sc.fe_bytes(nth_root(n))
x = torch.from_numpy(synth.synth_packed(7, n * cols).view(np.int64)).to(dev)
y, z, y1 = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
sp = [ctypes.c_void_p(s.cuda_stream) for s in streams]
eb = 2 * n          # int64 words per column


def col(t, c):
    return t.data_ptr() + 16 * n * c


def one_at_a_time(two_streams):
    for c in range(cols):
        p = sp[c & 1] if two_streams else sp[0]
        sc._check(lib.sc_ntt_dev(col(x, c), col(y1, c), n, root, 0, p))
        sc._check(lib.sc_ntt_dev(col(y1, c), col(z, c), n, root, 1, p))


def batch(two_streams):
    if not two_streams:
        sc._check(lib.sc_ntt_columns_dev(x.data_ptr(), y.data_ptr(), n, cols, root, 0, sp[0]))
        sc._check(lib.sc_ntt_columns_dev(y.data_ptr(), z.data_ptr(), n, cols, root, 1, sp[0]))
    else:
        h = cols // 2
        for k, (c0, cn) in enumerate(((0, h), (h, cols - h))):
            sc._check(lib.sc_ntt_columns_dev(col(x, c0), col(y, c0), n, cn, root, 0, sp[k]))
            sc._check(lib.sc_ntt_columns_dev(col(y, c0), col(z, c0), n, cn, root, 1, sp[k]))


def timed(fn, *a):
    for _ in range(3):
        fn(*a)
    torch.cuda.synchronize()
    best = None
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(reps):
            fn(*a)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        best = dt if best is None or dt < best else best
    return best


# correctness first
one_at_a_time(False); torch.cuda.synchronize()
ok_one = torch.equal(z, x)
z.zero_(); torch.cuda.synchronize(); batch(False); torch.cuda.synchronize()
ok = torch.equal(z, x) and torch.equal(y, y1) and ok_one
res = {}
for name, fn, two in (("one at a time, one stream", one_at_a_time, False), ("one at a time, two streams", one_at_a_time, True),
                      ("columns call, one stream", batch, False), ("columns call, halves on two streams", batch, True)):
    if two and cols < 2:
        continue
    dt = timed(fn, two)
    res[name] = dt
    print("2^%d x %d columns fwd+inv  %-38s %9.2f us  %6.2f us per column pair  %6.2f G el/s" % (log2n, cols, name, dt * 1e6, dt * 1e6 / cols, 2 * n * cols / dt / 1e9))
print("equal to sc_ntt_dev column by column and round trip:", "ok" if ok else "WRONG")
