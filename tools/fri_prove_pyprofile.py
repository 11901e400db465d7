#!/usr/bin/env python3
"""Where the Python side of Fri.prove (2^22) spends its time: cProfile over many proves (dev tool)."""
import os, sys, time, cProfile, pstats
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
def run(k):
    for _ in range(k):
        ps = ProofStream(); cw = sc.DeviceCodeword(cwv, field)
        fr.prove(cw, ps)
run(5)
pr = cProfile.Profile(); pr.enable(); run(100); pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)
