#!/usr/bin/env python3
"""cProfile of sharded_stark.ShardedFastStark.prove at world 1 on the synthetic 2-register AIR (dev tool).
   python tools/sharded_stark_profile.py [log2_fri=20]"""
import cProfile, os, pstats, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "stark-anatomy_amd"))
import torch
import starkcore as sc
from algebra import Field, FieldElement
from multivariate import MPolynomial
from sharded_stark import ShardedFastStark
log_fri = int(sys.argv[1]) if len(sys.argv) > 1 else 20
k, s = log_fri - 4, 40
sc.init(0); field = Field.main(); dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)
T = (1 << k) - 4 * s
a, b, rows = 3, 5, []
for _ in range(T):
    rows.append((a, b)); a, b = b, (a * a + b) % field.p
trace = [[FieldElement(x, field), FieldElement(y, field)] for x, y in rows]
v = MPolynomial.variables(5, field)
air = [v[3] - v[2], v[4] - v[1] * v[1] - v[2]]
boundary = [(0, 0, trace[0][0]), (0, 1, trace[0][1]), (T - 1, 1, trace[T - 1][1])]
stark = ShardedFastStark(field, 4, s, 2 * s, 2, T, 0, 1, dev)
tz, layer, root = stark.preprocess()
for _ in range(2):
    t0 = time.perf_counter(); proof = stark.prove(trace, air, boundary, tz, layer); torch.cuda.synchronize(); print("prove ms", round((time.perf_counter() - t0) * 1e3, 2))
pr = cProfile.Profile(); pr.enable(); stark.prove(trace, air, boundary, tz, layer); torch.cuda.synchronize(); pr.disable()
stats = pstats.Stats(pr).stats
for title, key in (("by own time", 2), ("by cumulative time", 3)):
    print(title)
    print("%8s %10s %10s  %s" % ("calls", "own us", "cum us", "function"))
    for (fn, line, name), row in sorted(stats.items(), key=lambda kv: -kv[1][key])[:28 if key == 2 else 45]:
        print("%8d %10.0f %10.0f  %s:%d(%s)" % (row[1], row[2] * 1e6, row[3] * 1e6, os.path.basename(fn), line, name))
