#!/usr/bin/env python3
"""cProfile of fast_stark.FastStark.prove on the synthetic AIR (device-resident trace): where the host spends a proof (dev tool).
python tools/plain_stark_pyprofile.py [log2_fri=20] [proofs=30]"""
import os, sys, cProfile, pstats
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "stark-anatomy_amd")); sys.path.insert(0, REPO)
import starkcore as sc
import workloads
from fast_stark import DeviceTrace, FastStark
log_fri = int(sys.argv[1]) if len(sys.argv) > 1 else 20
proofs = int(sys.argv[2]) if len(sys.argv) > 2 else 30
sc.init(0)
field, T, packed, air, boundary = workloads.synthetic_stark_instance(log_fri, 40)
stark = FastStark(field, 4, 40, 80, 2, T)
trace = DeviceTrace.from_packed(packed, field)
tz, tzc, root = stark.preprocess(device_resident=True)
for _ in range(3):
    stark.prove(trace, air, boundary, tz, tzc)
pr = cProfile.Profile(); pr.enable()
for _ in range(proofs):
    stark.prove(trace, air, boundary, tz, tzc)
pr.disable()
stats = pstats.Stats(pr).stats
print("per proof, microseconds (cProfile inflates Python-heavy entries); %d proofs at FRI 2^%d" % (proofs, log_fri))
for title, key in (("by own time", 2), ("by cumulative time", 3)):
    print(title)
    print("%8s %10s %10s  %s" % ("calls", "own us", "cum us", "function"))
    for (fn, line, name), row in sorted(stats.items(), key=lambda kv: -kv[1][key])[:30]:
        print("%8.1f %10.1f %10.1f  %s:%d(%s)" % (row[1] / proofs, row[2] * 1e6 / proofs, row[3] * 1e6 / proofs, os.path.basename(fn), line, name))
