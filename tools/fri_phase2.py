#!/usr/bin/env python3
"""Finer host-side split of Fri.prove at 2^22 (dev tool): commit phase per round, index sampling, the query round trip (C call vs Python object building)."""
import os, sys, time, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "stark-anatomy_amd"))
if "torch" in sys.argv[1:]:
    import torch
    torch.zeros(4, device="cuda").sum().item()
import starkcore as sc, synth
from algebra import Field
from fri import Fri
import fri as frimod
from ip import ProofStream
GEN = 85408008396924667383611388730472331217
sc.init(0); lib = sc.lib(); field = Field.main()
N = 1 << 22
om = field.primitive_nth_root(N)
coeffs = sc.DeviceVector.from_bytes(synth.synth_packed(4002, N // 4).tobytes())
cwv = sc.DeviceVector(N)
sc._check(lib.sc_coset_evaluate_dev(coeffs.ptr, N // 4, sc.fe_bytes(GEN), sc.fe_bytes(om.value), N, cwv.ptr, None)); sc.synchronize()
fr = Fri(field.generator(), om, N, 4, 40)
for _ in range(3):
    fr.prove(sc.DeviceCodeword(cwv, field), ProofStream())
T = {}
orig_q = sc.query_codewords
def timed_q(cws, reqs):
    t0 = time.perf_counter(); r = orig_q(cws, reqs); T["query_codewords"] = time.perf_counter() - t0; return r
frimod.query_codewords = timed_q
orig_call = lib.sc_merkle_query_multi_dev
class W:
    def __call__(self, *a):
        t0 = time.perf_counter(); r = orig_call(*a); T["sc_merkle_query_multi_dev"] = time.perf_counter() - t0; return r
best = None
for _ in range(5):
    cw = sc.DeviceCodeword(cwv, field); ps = ProofStream()
    t0 = time.perf_counter()
    cws = fr.commit(cw, ps)
    t1 = time.perf_counter()
    top = fr.sample_indices(ps.prover_fiat_shamir(), len(cws[0]) // 2, len(cws[-1]), fr.num_colinearity_tests)
    t2 = time.perf_counter()
    fr._query_all(cws, top, ps)
    t3 = time.perf_counter()
    rec = dict(commit_ms=(t1 - t0) * 1e3, sample_ms=(t2 - t1) * 1e3, query_all_ms=(t3 - t2) * 1e3, query_codewords_ms=T.get("query_codewords", 0) * 1e3, total_ms=(t3 - t0) * 1e3)
    if best is None or rec["total_ms"] < best["total_ms"]: best = rec
print(json.dumps({k: round(v, 3) for k, v in best.items()}))
# inside query_codewords: time the C call alone
import ctypes
cw = sc.DeviceCodeword(cwv, field); ps = ProofStream(); cws = fr.commit(cw, ps)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); fr._query_all(cws, top, ps); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(10)
