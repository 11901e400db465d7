#!/usr/bin/env python3
"""Phase timing of Fri.prove on a device-resident 2^22 codeword (BASELINE configs[3]) -- dev tool."""
import ctypes, json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "stark-anatomy_amd"))
import starkcore as sc, synth
from algebra import Field
from fri import Fri
from ip import ProofStream
import fri as fri_mod

GEN = 85408008396924667383611388730472331217
sc.init(0)
lib = sc.lib()
field = Field.main()
logN = int(sys.argv[1]) if len(sys.argv) > 1 else 22
N = 1 << logN
om = field.primitive_nth_root(N)
coeffs = sc.DeviceVector.from_bytes(synth.synth_packed(4002, N // 4).tobytes())
cw_vec = sc.DeviceVector(N)
sc._check(lib.sc_coset_evaluate_dev(coeffs.ptr, N // 4, sc.fe_bytes(GEN), sc.fe_bytes(om.value), N, cw_vec.ptr, None))
sc.synchronize()
fr = Fri(field.generator(), om, N, 4, 40)

# coarse phases by monkey-patching
T = {}
def timed(name, fn):
    def w(*a, **k):
        t0 = time.perf_counter(); r = fn(*a, **k); T[name] = T.get(name, 0.0) + time.perf_counter() - t0; return r
    return w
sc.MerkleTree.from_device = classmethod(lambda cls, vec, _f=sc.MerkleTree.from_device.__func__: timed("tree_build", _f)(cls, vec))
sc.MerkleTree.open_batch = timed("open_batch", sc.MerkleTree.open_batch)
sc.DeviceCodeword.gather = timed("gather", sc.DeviceCodeword.gather)
sc.DeviceCodeword.tolist = timed("tolist", sc.DeviceCodeword.tolist)
ProofStream.prover_fiat_shamir = timed("fiat_shamir", ProofStream.prover_fiat_shamir)
Fri.commit = timed("commit_total", Fri.commit)
Fri.query = timed("query_total", Fri.query)
for rep in range(4):
    T.clear()
    cw = sc.DeviceCodeword(cw_vec, field)
    ps = ProofStream()
    t0 = time.perf_counter()
    fr.prove(cw, ps)
    dt = time.perf_counter() - t0
    t1 = time.perf_counter(); ser = ps.serialize(); tser = time.perf_counter() - t1
    print(json.dumps(dict(logN=logN, prove_ms=round(dt * 1e3, 3), serialize_ms=round(tser * 1e3, 3), **{k: round(v * 1e3, 3) for k, v in T.items()})), flush=True)
    del cw
