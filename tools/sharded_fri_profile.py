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
pr = cProfile.Profile(); pr.enable(); ShardedFri(fr, R, 0, 1, dev).prove(slab, ProofStream()); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
