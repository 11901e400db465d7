#!/usr/bin/env python3
"""One-off check of the largest plans: 2^26 / 2^28-point transforms (4 and 16 GiB of traffic per pass) -- round trip of random data,
transform of a delta (all ones), of the constant vector (n * delta) and of the index-1 delta (powers of the root, spot-checked).  Dev tool."""
import json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "stark-anatomy_amd"))
import numpy as np, torch
import starkcore as sc, synth
P = synth.P; GEN = 85408008396924667383611388730472331217
def nth_root(n):
    r, order = GEN, 1 << 119
    while order != n: r, order = r * r % P, order >> 1
    return r
sc.init(0); lib = sc.lib(); dev = torch.device("cuda", 0)
for log2n in [int(a) for a in sys.argv[1:]] or (26, 28):
    n = 1 << log2n
    w = nth_root(n); root = sc.fe_bytes(w)
    g = torch.Generator(device="cpu"); g.manual_seed(log2n)
    # random canonical residues: hi limb below P_HI keeps the value < p
    lo = torch.randint(-(1 << 63), (1 << 63) - 1, (n,), dtype=torch.int64, generator=g)
    hi = torch.randint(0, 0x4B80000000000000, (n,), dtype=torch.int64, generator=g)
    x = torch.stack([lo, hi], dim=1).contiguous().to(dev)
    y, z = torch.empty_like(x), torch.empty_like(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sc._check(lib.sc_ntt_dev(x.data_ptr(), y.data_ptr(), n, root, 0, None)); sc.synchronize()
    t1 = time.perf_counter()
    sc._check(lib.sc_ntt_dev(y.data_ptr(), z.data_ptr(), n, root, 1, None)); sc.synchronize()
    t2 = time.perf_counter()
    ok_rt = bool(torch.equal(x, z))
    x.zero_(); x[0, 0] = 1
    torch.cuda.synchronize()       # torch fills x on its own stream; the library runs on its stream
    sc._check(lib.sc_ntt_dev(x.data_ptr(), y.data_ptr(), n, root, 0, None)); sc.synchronize()
    ok_delta = bool((y[:, 0] == 1).all() and (y[:, 1] == 0).all())
    x.zero_(); x[1, 0] = 1
    torch.cuda.synchronize()
    sc._check(lib.sc_ntt_dev(x.data_ptr(), y.data_ptr(), n, root, 0, None)); sc.synchronize()
    ok_pow = True
    for i in (0, 1, 2, 12345, n // 2, n - 1, (n // 3) | 1):
        v = pow(w, i, P)
        a, b = int(y[i, 0].item()) & ((1 << 64) - 1), int(y[i, 1].item()) & ((1 << 64) - 1)
        ok_pow &= (a | (b << 64)) == v
    x[:, 0] = 1; x[:, 1] = 0
    torch.cuda.synchronize()
    sc._check(lib.sc_ntt_dev(x.data_ptr(), y.data_ptr(), n, root, 0, None)); sc.synchronize()
    nz = int(torch.count_nonzero(y.abs().sum(dim=1)).item())
    ok_const = bool(int(y[0, 0].item()) == n and int(y[0, 1].item()) == 0 and nz == 1)
    if not ok_const:
        print("constant check detail: y[0] =", y[0].tolist(), "nonzero rows =", nz, flush=True)
    print(json.dumps(dict(log2n=log2n, fwd_ms=round((t1 - t0) * 1e3, 2), inv_ms=round((t2 - t1) * 1e3, 2), gelem_s=round(n / (t2 - t1) / 1e9, 2),
                          round_trip=ok_rt, delta=ok_delta, powers=ok_pow, constant=ok_const)), flush=True)
    del x, y, z
