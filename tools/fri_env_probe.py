#!/usr/bin/env python3
"""Does the process environment change Fri.prove's time? (dev tool)  python tools/fri_env_probe.py [torch] [stream] [big]"""
import os, sys, time, gc
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "stark-anatomy_amd"))
flags = set(sys.argv[1:])
if "torch" in flags:
    import torch
    torch.zeros(4, device="cuda").sum().item()
import starkcore as sc, synth
from algebra import Field
from fri import Fri
from ip import ProofStream
GEN = 85408008396924667383611388730472331217
sc.init(0); lib = sc.lib(); field = Field.main()
if "stream" in flags:
    import ctypes
    st = torch.cuda.Stream(); torch.cuda.set_stream(st)
    v = sc.DeviceVector(1 << 12); w = sc.DeviceVector(1 << 12)
    sc._check(lib.sc_ntt_dev(v.ptr, w.ptr, 1 << 12, sc.fe_bytes(field.primitive_nth_root(1 << 12).value), 0, ctypes.c_void_p(st.cuda_stream)))
    torch.cuda.synchronize()
if "big" in flags:
    keep = [sc.DeviceVector(1 << 24) for _ in range(6)]
    del keep
N = 1 << 22
om = field.primitive_nth_root(N)
coeffs = sc.DeviceVector.from_bytes(synth.synth_packed(4002, N // 4).tobytes())
cwv = sc.DeviceVector(N)
sc._check(lib.sc_coset_evaluate_dev(coeffs.ptr, N // 4, sc.fe_bytes(GEN), sc.fe_bytes(om.value), N, cwv.ptr, None)); sc.synchronize()
fr = Fri(field.generator(), om, N, 4, 40)
runs = []
for _ in range(8):
    cw = sc.DeviceCodeword(cwv, field); ps = ProofStream()
    if "nogc" not in flags: gc.collect()
    t0 = time.perf_counter(); fr.prove(cw, ps); runs.append(round((time.perf_counter() - t0) * 1e3, 3))
print(sorted(flags), runs)
