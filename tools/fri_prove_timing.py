#!/usr/bin/env python3
"""Fri.prove on a 2^22 codeword (BASELINE configs[3]: expansion factor 4, 40 colinearity checks), host clock around the call, best
and median of N runs; the proof is verified once.  usage: python tools/fri_prove_timing.py [log2n] [runs]"""
import os, sys, time, statistics
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "stark-anatomy_amd"))
import starkcore as sc, synth
from algebra import Field
from fri import Fri
from ip import ProofStream
GEN = 85408008396924667383611388730472331217
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 22
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 30
sc.init(0); lib = sc.lib(); field = Field.main()
N = 1 << log2n
om = field.primitive_nth_root(N)
coeffs = sc.DeviceVector.from_bytes(synth.synth_packed(4002, N // 4).tobytes())
cwv = sc.DeviceVector(N)
sc._check(lib.sc_coset_evaluate_dev(coeffs.ptr, N // 4, sc.fe_bytes(GEN), sc.fe_bytes(om.value), N, cwv.ptr, None)); sc.synchronize()
fr = Fri(field.generator(), om, N, 4, 40)
times = []
for i in range(runs + 3):
    ps = ProofStream()
    cw = sc.DeviceCodeword(cwv, field)
    t0 = time.perf_counter()
    fr.prove(cw, ps)
    dt = time.perf_counter() - t0
    if i >= 3:
        times.append(dt)
ser = ps.serialize()
ok = fr.verify(ProofStream().deserialize(ser), [])
print("Fri.prove 2^%d: best %.3f ms  median %.3f ms  (%d runs)  proof %d bytes  verify %s" % (log2n, min(times) * 1e3, statistics.median(times) * 1e3, runs, len(ser), ok))
# where the host time of one call goes (python side): the library call, then the stream's segments
import fri as _fri
real = _fri._sc.lib().sc_fri_prove_dev
T = {"library call": 0.0}
class _Timed:
    def __call__(self, *a):
        t0 = time.perf_counter(); r = real(*a); T["library call"] += time.perf_counter() - t0; return r
lib_obj = _fri._sc.lib()
lib_obj.sc_fri_prove_dev = _Timed()
tot = 0.0
for i in range(10):
    ps = ProofStream(); cw = sc.DeviceCodeword(cwv, field)
    t0 = time.perf_counter(); fr.prove(cw, ps); tot += time.perf_counter() - t0
print("per call: total %.1f us, of which sc_fri_prove_dev %.1f us, python around it %.1f us" % (tot / 10 * 1e6, T["library call"] / 10 * 1e6, (tot - T["library call"]) / 10 * 1e6))
