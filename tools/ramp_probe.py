#!/usr/bin/env python3
"""Per-step GPU time of the 2^20 forward+inverse pair right after an idle period (dev tool): shows the two clock regimes of the
board -- ~79 us per pair for the first milliseconds after idle (what a 20-step driver-style run sees), ~69 us once the GPU has been
busy for tens of milliseconds (what tools/ab.py reports as steady state)."""
import ctypes, json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "stark-anatomy_amd"))
import numpy as np, torch
import starkcore as sc, synth
P = synth.P; GEN = 85408008396924667383611388730472331217
def nth_root(n):
    r, order = GEN, 1 << 119
    while order != n: r, order = r * r % P, order >> 1
    return r
sc.init(0); lib = sc.lib(); dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream); sptr = ctypes.c_void_p(stream.cuda_stream)
n = 1 << 20; root = sc.fe_bytes(nth_root(n))
x = torch.from_numpy(synth.synth_packed(1, n).view(np.int64)).to(dev); y = torch.empty_like(x); z = torch.empty_like(x)
def step():
    sc._check(lib.sc_ntt_dev(x.data_ptr(), y.data_ptr(), n, root, 0, sptr))
    sc._check(lib.sc_ntt_dev(y.data_ptr(), z.data_ptr(), n, root, 1, sptr))
for _ in range(5): step()
torch.cuda.synchronize()
time.sleep(0.5)   # idle like a fresh process between set-up and timing
evs = [torch.cuda.Event(enable_timing=True) for _ in range(61)]
t0 = time.perf_counter()
evs[0].record(stream)
host = []
for i in range(60):
    step(); evs[i + 1].record(stream); host.append(time.perf_counter() - t0)
torch.cuda.synchronize()
per = [round(evs[i].elapsed_time(evs[i + 1]) * 1e3, 1) for i in range(60)]
print("per-step us:", per)
print("host issue time of step i (us):", [round(h * 1e6) for h in host[:12]])
