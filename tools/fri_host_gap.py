#!/usr/bin/env python3
"""What the host does between the last root of Fri.commit and the query kernel of Fri.prove at 2^22 (dev tool): step timings."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "stark-anatomy_amd"))
import starkcore as sc, synth
import proof_objects as po_
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
T = {}
def clock(name, fn):
    t0 = time.perf_counter(); r = fn(); T[name] = T.get(name, 0.0) + (time.perf_counter() - t0); return r
reps = 20
for _ in range(reps):
    ps = ProofStream()
    cw = sc.DeviceCodeword(cwv, field)
    rounds = fr.num_rounds()
    codewords = clock("commit rounds in the library", lambda: fr._commit_in_library(cw, ps, rounds))
    last = codewords[-1]
    raw = clock("last codeword to the host", lambda: last.vec.to_bytes())
    lazy = po_.lazy_objects(ps)
    clock("ElementList segment", lambda: lazy.add(po_.ElementList(last, raw)))
    seed = clock("prover_fiat_shamir (describe + pickle + shake)", lambda: ps.prover_fiat_shamir())
    top = clock("sample_indices", lambda: fr.sample_indices(seed, len(codewords[0]) // 2, len(codewords[-1]), fr.num_colinearity_tests))
    clock("_query_all (requests, launch, segments)", lambda: fr._query_all(codewords, top, ps))
    clock("serialize", lambda: ps.serialize())
for k, v in T.items():
    print("%8.1f us  %s" % (v / reps * 1e6, k))
