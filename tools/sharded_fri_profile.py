#!/usr/bin/env python3
"""cProfile of ShardedFri.prove at world 1 on the column-slab layout (dev tool): where the per-round host time goes.
   python tools/sharded_fri_profile.py [log2N=22] [log2R=8]"""
import cProfile, os, pstats, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "stark-anatomy_amd"))
import torch
import starkcore as sc, synth
from algebra import Field
from fri import Fri
from ip import ProofStream
from sharded import ShardedFri
logN = int(sys.argv[1]) if len(sys.argv) > 1 else 22
logR = int(sys.argv[2]) if len(sys.argv) > 2 else 8
sc.init(0); field = Field.main(); dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)
N, R = 1 << logN, 1 << logR
om = field.primitive_nth_root(N)
import numpy as np
slab = torch.from_numpy(synth.synth_packed(7, N).view(np.int64).reshape(N // R, R, 2).copy()).to(dev)
fr = Fri(field.generator(), om, N, 4, 40)
best = 1e9
for _ in range(4):
    ps = ProofStream(); torch.cuda.synchronize(); t0 = time.perf_counter()
    ShardedFri(fr, R, 0, 1, dev).prove(slab, ps); best = min(best, time.perf_counter() - t0)
print("sharded prove ms", round(best * 1e3, 3))
cw = sc.DeviceCodeword(sc.DeviceVector.from_bytes(slab.cpu().numpy().tobytes()), field)
b2 = 1e9
for _ in range(4):
    ps2 = ProofStream(); t0 = time.perf_counter(); fr.prove(sc.DeviceCodeword(cw.vec, field), ps2); b2 = min(b2, time.perf_counter() - t0)
print("plain prove ms", round(b2 * 1e3, 3), "same proof", ps.serialize() == ps2.serialize())
# commit phase / query phase split of the sharded path (the query phase starts when _query_all is entered)
marks = {}
inner = ShardedFri._query_all
def timed_query(self, layers, last_list, proof_stream):
    marks["q0"] = time.perf_counter(); out = inner(self, layers, last_list, proof_stream); marks["q1"] = time.perf_counter(); return out
ShardedFri._query_all = timed_query
split = []
for _ in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter(); ShardedFri(fr, R, 0, 1, dev).prove(slab, ProofStream())
    split.append((marks["q0"] - t0, marks["q1"] - marks["q0"]))
c, q = min(split, key=sum)
print("sharded phases ms: commit (incl. last codeword)", round(c * 1e3, 3), "query", round(q * 1e3, 3))
ShardedFri._query_all = inner
pr = cProfile.Profile(); pr.enable(); ShardedFri(fr, R, 0, 1, dev).prove(slab, ProofStream()); pr.disable()
rows = sorted(pstats.Stats(pr).stats.items(), key=lambda kv: -kv[1][3])[:34]      # by cumulative time, in microseconds
print("%8s %10s %10s  %s" % ("calls", "own us", "cum us", "function"))
for (fn, line, name), (cc, nc, tt, ct, _) in rows:
    print("%8d %10.0f %10.0f  %s:%d(%s)" % (nc, tt * 1e6, ct * 1e6, os.path.basename(fn), line, name))
