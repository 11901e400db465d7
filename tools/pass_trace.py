#!/usr/bin/env python3
"""Where does a pass workgroup spend its time?  Poor man's thread trace (rocprofv3 --att needs a decoder library this
image does not ship): the TRACE instantiation of ntt_pass_kernel_fixed stamps s_memtime per wave at every phase
boundary (sc_debug_trace).  Prints, per pass of one forward transform, the median over workgroups of each phase in
microseconds (s_memtime ticks calibrated against s_memrealtime = 100 MHz per wave) and the spread between the first and the last wave.

   python tools/pass_trace.py [log2n ...]        (dev tool; writes nothing but stdout)
"""
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
NAMES = {0: "entry", 1: "loads issued, tw staged", 2: "barrier + loads landed", 3: "round0 math"}
def names(nr):
    d = dict(NAMES)
    for r in range(nr):
        d[3 + 2 * r] = "round%d math" % r
        d[4 + 2 * r] = ("round%d exchange" % r) if r + 1 < nr else "store issued"
    d[14] = "stores drained"
    return d
logs = [int(a) for a in sys.argv[1:]] or [20, 22]
for log2n in logs:
    n = 1 << log2n; root = sc.fe_bytes(nth_root(n))
    x = torch.from_numpy(synth.synth_packed(1, n).view(np.int64)).to(dev); y = torch.empty_like(x)
    f = lambda inv=0: sc._check(lib.sc_ntt_dev(x.data_ptr(), y.data_ptr(), n, root, inv, sptr))
    for wl in (1, 0):
        sc.set_tuning("wave_local", wl)
        npass = 2 if log2n <= 20 else 3
        waves = (n // 4) // 64                       # per pass: E = 4 elements per thread
        buf = torch.zeros((npass * waves, 16), dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        # the traced transform is enqueued right behind a burst of untraced ones, with no idle gap: the first kernel after an idle
        # period runs ~2x slower (tools/ramp_probe.py) and would be what the trace shows for pass 0
        for _ in range(40 if log2n <= 22 else 8): f()
        sc._check(lib.sc_debug_trace(buf.data_ptr()))
        f()
        sc._check(lib.sc_debug_trace(None))
        torch.cuda.synchronize()
        allt = buf.cpu().numpy().astype(np.int64)
        for ps in range(npass):
            t = allt[ps * waves:(ps + 1) * waves]
            t = t[t[:, 0] != 0]
            if t.shape[0] == 0:
                print(json.dumps({"log2n": log2n, "pass": ps, "note": "no stamps (generic kernel?)"})); continue
            nr_guess = int(((t[0, 3:14] != 0).sum() + 1) // 2)
            nm = names(nr_guess)
            # s_memtime ticks per microsecond, from the 100 MHz s_memrealtime stamps at entry (slot 15) and exit (slot 13)
            tick = float(np.median((t[:, 14] - t[:, 0]) / np.maximum((t[:, 13] - t[:, 15]) / 100.0, 1e-3)))
            rel = (t[:, :15] - t[:, 0:1]) / tick            # us since the wave's own entry
            rel[t[:, :15] == 0] = np.nan
            rel[:, 13] = np.nan
            start = (t[:, 15] - t[:, 15].min()) / 100.0
            out = {"log2n": log2n, "pass": ps, "wave_local": wl, "waves": int(t.shape[0]), "memtime_ticks_per_us": round(tick, 1),
                   "entry_after_first_wave_us_p50_p90_max": [round(float(np.percentile(start, q)), 2) for q in (50, 90, 100)]}
            prev = np.zeros(t.shape[0])
            phases = {}
            for i in list(range(1, 14)) + [14]:
                col = rel[:, i]
                if np.isnan(col).all(): continue
                phases["%02d %s" % (i, nm.get(i, "?"))] = {"at_us_med": round(float(np.nanmedian(col)), 2), "at_us_max": round(float(np.nanmax(col)), 2),
                                                          "dur_us_med": round(float(np.nanmedian(col - prev)), 2)}
                prev = col
            out["phases"] = phases
            out["kernel_span_us"] = round(float((t[:, 13].max() - t[:, 15].min()) / 100.0), 2)
            print(json.dumps(out), flush=True)
sc.set_tuning("wave_local", 1)
