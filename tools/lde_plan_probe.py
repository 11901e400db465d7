#!/usr/bin/env python3
"""LDE 2^18 -> 2^21 and fwd+inv 2^21 / 2^22 under alternative pass plans (dev tool): python tools/lde_plan_probe.py '{"max_digit_log":11,...}' ..."""
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
DEFAULTS = dict(fixed_shapes=1, max_tile_log=-1, max_col_log=-1, max_digit_log=-1, loge=2, min_tiles_log=8)
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
m, order = 1 << 18, 1 << 21
co = torch.from_numpy(synth.synth_packed(5, m).view(np.int64)).to(dev); lde_out = torch.empty((order, 2), dtype=torch.int64, device=dev)
gen21 = sc.fe_bytes(nth_root(order))
ref = None
for cfg in [dict()] + [json.loads(a) for a in sys.argv[1:]] + [dict()]:
    for k, v in DEFAULTS.items(): sc.set_tuning(k, v)
    for k, v in cfg.items(): sc.set_tuning(k, v)
    res = dict(cfg=cfg)
    try:
        f = lambda: sc._check(lib.sc_coset_evaluate_dev(co.data_ptr(), m, sc.fe_bytes(GEN), gen21, order, lde_out.data_ptr(), sptr))
        res["lde_us"] = round(timed(f, 100), 1)
        torch.cuda.synchronize()
        if ref is None: ref = lde_out.clone()
        res["same"] = bool(torch.equal(ref, lde_out))
        for lg in (21, 22):
            n = 1 << lg
            x = torch.from_numpy(synth.synth_packed(1, n).view(np.int64)).to(dev); y = torch.empty_like(x); z = torch.empty_like(x); root = sc.fe_bytes(nth_root(n))
            def g():
                sc._check(lib.sc_ntt_dev(x.data_ptr(), y.data_ptr(), n, root, 0, sptr))
                sc._check(lib.sc_ntt_dev(y.data_ptr(), z.data_ptr(), n, root, 1, sptr))
            us = timed(g, 50)
            res["G_el_s_2p%d" % lg] = round(2 * n / us / 1e3, 2); res["rt_%d" % lg] = bool(torch.equal(x, z))
    except Exception as e:
        res["error"] = repr(e)[:200]
    print(json.dumps(res), flush=True)
for k, v in DEFAULTS.items(): sc.set_tuning(k, v)
