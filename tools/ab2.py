#!/usr/bin/env python3
"""A/B of the round-2 pass-kernel changes: forward+inverse pairs at 2^20 / 2^22 / 2^24 under each tuning combination,
20-step (driver-style) and long runs.  Dev tool."""
import ctypes, json, os, sys
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
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream); sptr = ctypes.c_void_p(stream.cuda_stream)
cfgs = [dict(), dict(wave_local=0), dict(tw_on_load=0), dict(wave_local=0, tw_on_load=0), dict()]
extra = [json.loads(a) for a in sys.argv[1:]]
for log2n in (20, 22, 24):
    n = 1 << log2n; root = sc.fe_bytes(nth_root(n))
    x = torch.from_numpy(synth.synth_packed(1, n).view(np.int64)).to(dev); y = torch.empty_like(x); z = torch.empty_like(x)
    for cfg in cfgs + extra:
        for k, v in dict(wave_local=1, tw_on_load=1).items(): sc.set_tuning(k, v)
        for k, v in cfg.items(): sc.set_tuning(k, v)
        def f():
            sc._check(lib.sc_ntt_dev(x.data_ptr(), y.data_ptr(), n, root, 0, sptr))
            sc._check(lib.sc_ntt_dev(y.data_ptr(), z.data_ptr(), n, root, 1, sptr))
        for _ in range(5): f()
        torch.cuda.synchronize()
        assert torch.equal(x, z)
        res = {}
        for reps in (20, 400 if log2n < 24 else 60):
            best = None
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(reps): f()
                e1.record(stream); torch.cuda.synchronize()
                us = e0.elapsed_time(e1) / reps * 1e3
                best = us if best is None or us < best else best
            res["us_pair_%d" % reps] = round(best, 1)
            res["gelem_s_%d" % reps] = round(2 * n / best / 1e3, 2)
        print(json.dumps(dict(log2n=log2n, **cfg, **res)), flush=True)
for k, v in dict(wave_local=1, tw_on_load=1).items(): sc.set_tuning(k, v)
