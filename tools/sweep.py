#!/usr/bin/env python3
"""Tuning sweep on one MI355X: time per transform for tile / radix / column settings, plus Fri.prove ms.
Dev tool.  usage: python tools/sweep.py [quick]"""
import ctypes
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "stark-anatomy_amd"))
import numpy as np
import torch
import starkcore as sc
import synth

P = synth.P
GEN = 85408008396924667383611388730472331217


def nth_root(n):
    r, order = GEN, 1 << 119
    while order != n:
        r, order = r * r % P, order >> 1
    return r


def main():
    sc.init(0)
    lib = sc.lib()
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    sptr = ctypes.c_void_p(stream.cuda_stream)
    defaults = dict(max_tile_log=12, loge=3, max_col_log=6, min_tiles_log=10, single_pass_max_log=11, max_digit_log=8, xcd_remap=1)
    results = []

    def timeit(fn, reps):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            fn()
        e1.record(stream)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3     # us

    sizes = [20, 22, 24] if len(sys.argv) < 2 else [int(a) for a in sys.argv[1:]]
    for log2n in sizes:
        n = 1 << log2n
        root = sc.fe_bytes(nth_root(n))
        x = torch.from_numpy(synth.synth_packed(1, n).view(np.int64)).to(dev)
        y = torch.empty_like(x)
        cfgs = []
        for tile in (9, 10, 11, 12):
            for loge in (2, 3, 4):
                for col in (3, 4, 5, 6):
                    cfgs.append(dict(max_tile_log=tile, loge=loge, max_col_log=col, min_tiles_log=0))
        cfgs.append(dict(xcd_remap=0))
        cfgs.append(dict(max_digit_log=7))
        cfgs.append(dict(max_digit_log=10, max_tile_log=12, min_tiles_log=0))
        cfgs.append(dict(max_digit_log=11, max_tile_log=12, min_tiles_log=0, max_col_log=1))
        cfgs.append(dict(max_digit_log=12, max_tile_log=12, min_tiles_log=0, max_col_log=0))
        for cfg in cfgs:
            full = dict(defaults)
            full.update(cfg)
            for k, v in full.items():
                sc.set_tuning(k, v)
            try:
                us = timeit(lambda: sc._check(lib.sc_ntt_dev(x.data_ptr(), y.data_ptr(), n, root, 0, sptr)), 20 if log2n >= 24 else 50)
            except Exception as e:
                us = None
                print("cfg failed", cfg, e)
            rec = dict(log2n=log2n, us=us, gelem_s=(n / us / 1e3 if us else None), **cfg)
            results.append(rec)
            print(json.dumps(rec), flush=True)
    for k, v in defaults.items():
        sc.set_tuning(k, v)
    best = {}
    for r in results:
        if r["us"] and (r["log2n"] not in best or r["us"] < best[r["log2n"]]["us"]):
            best[r["log2n"]] = r
    print("BEST", json.dumps(best))

    # Fri.prove at 2^22 (BASELINE configs[3])
    from algebra import Field, FieldElement
    from univariate import Polynomial
    from ntt import fast_coset_evaluate_device
    from fri import Fri
    from ip import ProofStream
    torch.cuda.set_stream(torch.cuda.default_stream())
    field = Field.main()
    for logN in (16, 22):
        N = 1 << logN
        om = field.primitive_nth_root(N)
        coeffs = sc.DeviceVector.from_bytes(synth.synth_packed(4002, N // 4).tobytes())
        cw_vec = sc.DeviceVector(N)
        sc._check(lib.sc_coset_evaluate_dev(coeffs.ptr, N // 4, sc.fe_bytes(GEN), sc.fe_bytes(om.value), N, cw_vec.ptr, None))
        sc.synchronize()
        fr = Fri(field.generator(), om, N, 4, 40)
        for rep in range(3):
            cw = sc.DeviceCodeword(cw_vec, field)
            ps = ProofStream()
            t0 = time.perf_counter()
            fr.prove(cw, ps)
            dt = time.perf_counter() - t0
            print(json.dumps(dict(fri_prove_logN=logN, ms=dt * 1e3, rounds=fr.num_rounds(), objects=len(ps.objects))), flush=True)


if __name__ == "__main__":
    main()
