#!/usr/bin/env python3
"""Fri.prove at 2^22, a few times (dev tool; run under rocprofv3 --kernel-trace, then tools/fri_trace_report.py)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "stark-anatomy_amd"))
import starkcore as sc, synth
from algebra import Field
from fri import Fri
from ip import ProofStream
GEN = 85408008396924667383611388730472331217
sc.init(0); lib = sc.lib(); field = Field.main()
N = 1 << 22
om = field.primitive_nth_root(N)
coeffs = sc.DeviceVector.from_bytes(synth.synth_packed(4002, N // 4).tobytes())
cwv = sc.DeviceVector(N)
sc._check(lib.sc_coset_evaluate_dev(coeffs.ptr, N // 4, sc.fe_bytes(GEN), sc.fe_bytes(om.value), N, cwv.ptr, None)); sc.synchronize()
fr = Fri(field.generator(), om, N, 4, 40)
for _ in range(6):
    t0 = time.perf_counter(); fr.prove(sc.DeviceCodeword(cwv, field), ProofStream()); print("prove_ms", round((time.perf_counter() - t0) * 1e3, 3))
    time.sleep(0.02)
