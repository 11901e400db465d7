#!/usr/bin/env python3
"""Time of the query round trip of Fri.prove at 2^22 (dev tool): the C call alone vs the Python object building around it."""
import ctypes, json, os, sys, time
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
ps = ProofStream()
cws = fr.commit(sc.DeviceCodeword(cwv, field), ps)
top = fr.sample_indices(ps.prover_fiat_shamir(), len(cws[0]) // 2, len(cws[-1]), fr.num_colinearity_tests)
s, rounds = 40, len(cws) - 1
per_round, idx = [], list(top)
for i in range(rounds):
    idx = [j % (len(cws[i]) // 2) for j in idx]; per_round.append(idx)
requests = []
for j, cw in enumerate(cws):
    r = []
    if j < rounds: r += per_round[j][:s] + [q + len(cw) // 2 for q in per_round[j][:s]]
    if j > 0: r += per_round[j - 1][:s]
    requests.append(r)
n = len(cws); trees = [cw.tree() for cw in cws]
flat = [i for r in requests for i in r]; total = len(flat)
path_bytes = sum(64 * t.depth * len(r) for t, r in zip(trees, requests))
elems = ctypes.create_string_buffer(16 * total); paths = ctypes.create_string_buffer(path_bytes)
_vp = ctypes.c_void_p
a_tr = (_vp * n)(*[t._h for t in trees]); a_v = (_vp * n)(*[cw.vec.ptr for cw in cws])
a_i = (ctypes.c_uint64 * total)(*flat); a_c = (ctypes.c_uint64 * n)(*[len(r) for r in requests])
best = 1e9
for _ in range(20):
    t0 = time.perf_counter(); sc._check(lib.sc_merkle_query_multi_dev(n, a_tr, a_v, a_i, a_c, elems, paths)); best = min(best, time.perf_counter() - t0)
bq = 1e9
for _ in range(20):
    t0 = time.perf_counter(); sc.query_codewords(cws, requests); bq = min(bq, time.perf_counter() - t0)
ba = 1e9
for _ in range(20):
    p2 = ProofStream(); t0 = time.perf_counter(); fr._query_all(cws, top, p2); ba = min(ba, time.perf_counter() - t0)
print(json.dumps(dict(total_openings=total, path_bytes=path_bytes, c_call_us=round(best * 1e6, 1), query_codewords_us=round(bq * 1e6, 1), query_all_us=round(ba * 1e6, 1))))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(5): fr.commit(sc.DeviceCodeword(cwv, field), ProofStream())
pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(18)
bt = 1e9
for _ in range(10):
    ps = ProofStream(); t0 = time.perf_counter(); fr.prove(sc.DeviceCodeword(cwv, field), ps); bt = min(bt, time.perf_counter() - t0)
print("prove_ms", round(bt * 1e3, 3))
