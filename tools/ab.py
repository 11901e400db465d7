#!/usr/bin/env python3
"""A/B of pass-kernel tunings (dev tool): fwd+inv pairs at 2^20/2^22/2^24 and the LDE 2^18 -> 2^21, per tuning dict given as
JSON on the command line (defaults first).   python tools/ab.py '{"looped":0}' '{"prune":0}' ..."""
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
DEFAULTS = dict(fixed_shapes=1, wave_local=1, tw_on_load=0, prio_balance=-1, prune=1, max_tile_log=-1, max_col_log=-1, max_digit_log=-1, direct_tw_max_log=22)
cfgs = [dict()] + [json.loads(a) for a in sys.argv[1:]] + [dict()]
def timed(f, reps):
    for _ in range(3): f()
    torch.cuda.synchronize()
    best = None
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps): f()
        e1.record(stream); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        best = us if best is None or us < best else best
    return best
bufs = {}
for lg in (20, 22, 24):
    n = 1 << lg
    x = torch.from_numpy(synth.synth_packed(1, n).view(np.int64)).to(dev)
    bufs[lg] = (x, torch.empty_like(x), torch.empty_like(x), sc.fe_bytes(nth_root(n)))
m, order = 1 << 18, 1 << 21
co = torch.from_numpy(synth.synth_packed(5, m).view(np.int64)).to(dev); lde_out = torch.empty((order, 2), dtype=torch.int64, device=dev)
gen21 = sc.fe_bytes(nth_root(order))
for cfg in cfgs:
    for k, v in DEFAULTS.items(): sc.set_tuning(k, v)
    for k, v in cfg.items(): sc.set_tuning(k, v)
    res = dict(cfg=cfg)
    for lg in (20, 22, 24):
        x, y, z, root = bufs[lg]; n = 1 << lg
        def f():
            sc._check(lib.sc_ntt_dev(x.data_ptr(), y.data_ptr(), n, root, 0, sptr))
            sc._check(lib.sc_ntt_dev(y.data_ptr(), z.data_ptr(), n, root, 1, sptr))
        us = timed(f, 20 if lg == 24 else 100)
        assert torch.equal(x, z)
        res["G_el_s_2p%d" % lg] = round(2 * n / us / 1e3, 2)
    f = lambda: sc._check(lib.sc_coset_evaluate_dev(co.data_ptr(), m, sc.fe_bytes(GEN), gen21, order, lde_out.data_ptr(), sptr))
    res["lde_2p18_2p21_us"] = round(timed(f, 100), 1)
    print(json.dumps(res), flush=True)
for k, v in DEFAULTS.items(): sc.set_tuning(k, v)
