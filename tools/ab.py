#!/usr/bin/env python3
"""A/B: time forward NTTs with the given library build (STARKCORE_LIB) -- dev tool."""
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
tag = os.environ.get("STARKCORE_LIB", "default")
cfgs = [dict(), dict(fixed_shapes=0), dict(fixed_shapes=1), dict(fixed_shapes=0), dict(fixed_shapes=1)]
for log2n in (20, 22, 24):
    n = 1 << log2n; root = sc.fe_bytes(nth_root(n))
    x = torch.from_numpy(synth.synth_packed(1, n).view(np.int64)).to(dev); y = torch.empty_like(x)
    for cfg in cfgs:
        for k, v in dict(max_digit_log=-1, max_col_log=-1, max_tile_log=-1, loge=2, min_tiles_log=8, direct_tw_max_log=22, xcd_remap=1, fixed_shapes=1).items(): sc.set_tuning(k, v)
        for k, v in cfg.items(): sc.set_tuning(k, v)
        f = lambda: sc._check(lib.sc_ntt_dev(x.data_ptr(), y.data_ptr(), n, root, 0, sptr))
        for _ in range(3): f()
        torch.cuda.synchronize()
        reps = 20 if log2n >= 24 else 50
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps): f()
        e1.record(stream); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        print(json.dumps(dict(lib=os.path.basename(tag), log2n=log2n, us=round(us, 1), gelem_s=round(n / us / 1e3, 2), **cfg)), flush=True)
