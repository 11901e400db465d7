#!/usr/bin/env python3
"""Stress of the asynchronous commit rounds (dev tool): the golden synthetic proofs (tests/golden/fri.json) proved over and over,
every serialized proof compared with the reference's SHA-256; run several copies at once.   python tools/fri_stress.py [seconds=30]"""
import hashlib, json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "stark-anatomy_amd")); sys.path.insert(0, REPO)
import starkcore as sc, synth
from algebra import Field
from fri import Fri
from ip import ProofStream
GEN = 85408008396924667383611388730472331217
sc.init(0); lib = sc.lib(); field = Field.main()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
golden = json.load(open(os.path.join(REPO, "tests", "golden", "fri.json")))
cases = []
for rec in golden["prove_synth"]:
    N = 1 << rec["logN"]
    om = field.primitive_nth_root(N)
    coeffs = sc.DeviceVector.from_bytes(synth.synth_packed(rec["coeff_seed"], N // 4).tobytes())
    cw = sc.DeviceVector(N)
    sc._check(lib.sc_coset_evaluate_dev(coeffs.ptr, N // 4, sc.fe_bytes(GEN), sc.fe_bytes(om.value), N, cw.ptr, None))
    sc.synchronize()
    cases.append((rec, Fri(field.generator(), om, N, rec["expansion_factor"], rec["num_colinearity_tests"]), cw))
t0, n = time.time(), 0
while time.time() - t0 < budget:
    for rec, fr, cw in cases:
        ps = ProofStream()
        top = fr.prove(sc.DeviceCodeword(cw, field), ps)
        ser = ps.serialize()
        assert top == rec["top_level_indices"] and hashlib.sha256(ser).hexdigest() == rec["serialized_sha256"], ("MISMATCH", rec["logN"], n)
        n += 1
print("fri stress ok:", n, "proofs, pid", os.getpid())
