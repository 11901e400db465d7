#!/usr/bin/env python3
"""Phase trace of a launch over several columns (sc_ntt_columns_dev with sc_debug_trace: the TRACE build of ntt_pass_kernel_fixed
stamps s_memtime per wave at every phase boundary): per pass, the median phase durations of a workgroup in the steady state of a
long grid, next to the single transform's (tools/pass_trace.py), and how many workgroups a CU-slot ran back to back.
   python tools/columns_trace.py [log2n=20] [cols=16]"""
import ctypes, json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "stark-anatomy_amd"))
import numpy as np, torch
import starkcore as sc, synth
from workloads import nth_root
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 16
n = 1 << log2n
sc.init(0); lib = sc.lib(); dev = torch.device("cuda", 0)
sc.set_tuning("loge_cols", 2)          # the TRACE build exists for the four-elements-per-thread kernels
root = sc.fe_bytes(nth_root(n))
x = torch.from_numpy(synth.synth_packed(3, n * cols).view(np.int64).reshape(-1)).to(dev); y = torch.empty_like(x)
npass = int(lib.sc_ntt_num_passes(n))
waves = (n // 4) // 64 * cols
buf = torch.zeros((npass * waves, 16), dtype=torch.int64, device=dev)
f = lambda: sc._check(lib.sc_ntt_columns_dev(x.data_ptr(), y.data_ptr(), n, cols, root, 0, None))
for _ in range(6): f()
sc._check(lib.sc_debug_trace(buf.data_ptr())); f(); sc._check(lib.sc_debug_trace(None)); sc.synchronize()
allt = buf.cpu().numpy().astype(np.int64)
for ps in range(npass):
    t = allt[ps * waves:(ps + 1) * waves]
    t = t[t[:, 0] != 0]
    wpg = 16 if log2n <= 20 else 8
    tick = float(np.median((t[:, 14] - t[:, 0]) / np.maximum((t[:, 13] - t[:, 15]) / 100.0, 1e-3)))
    t0 = t[:, 15].min()
    # per workgroup: entry of its first wave, exit of its last (100 MHz clock -> us)
    g = t.reshape(-1, wpg, 16)
    wg_in = (g[:, :, 15].min(axis=1) - t0) / 100.0
    wg_out = (g[:, :, 13].max(axis=1) - t0) / 100.0
    dur = wg_out - wg_in
    order = np.argsort(wg_in)
    steady = order[len(order) // 4: 3 * len(order) // 4]           # the middle half of the launch by start time
    rel = (t[:, :15] - t[:, 0:1]) / tick
    rel[t[:, :15] == 0] = np.nan
    relg = rel.reshape(-1, wpg, 15)
    names = {1: "loads issued", 2: "loads landed (+barrier)"}
    nr = int(((t[0, 3:13] != 0).sum() + 1) // 2)
    for r in range(nr):
        names[3 + 2 * r] = "round%d math" % r
        names[4 + 2 * r] = "round%d exchange" % r if r + 1 < nr else "stores issued"
    names[14] = "stores drained"
    phases = {}
    prev = np.zeros(relg[steady].shape[:2])
    for i in sorted(names):
        col = relg[steady][:, :, i]
        phases["%02d %s" % (i, names[i])] = round(float(np.nanmedian(col - prev)), 2)
        prev = col
    span = float(wg_out.max())
    print(json.dumps({"log2n": log2n, "cols": cols, "pass": ps, "workgroups": int(len(dur)), "launch_span_us": round(span, 1),
                      "span_per_column_us": round(span / cols, 2), "wg_duration_us_steady_p10_p50_p90": [round(float(np.percentile(dur[steady], q)), 2) for q in (10, 50, 90)],
                      "wg_duration_us_first_256": round(float(np.median(dur[order[:256]])), 2),
                      "phase_us_median_steady": phases,
                      "concurrent_workgroups_mid_launch": int(((wg_in <= span / 2) & (wg_out >= span / 2)).sum())}))
