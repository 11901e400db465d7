#!/usr/bin/env python3
"""fast_stark.FastStark.prove in a loop on the synthetic AIR with a device-resident trace (dev tool; run under rocprofv3
--kernel-trace and feed the trace to tools/gap_report.py).   python tools/plain_stark_loop.py [log2_fri=24] [proofs=6] [eager]
("eager": FastStark.EAGER_COMMITS = True, every commitment waits for its root -- the A/B of the lazily pushed roots)"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "stark-anatomy_amd")); sys.path.insert(0, REPO)
import starkcore as sc
import workloads
from fast_stark import DeviceTrace, FastStark
log_fri = int(sys.argv[1]) if len(sys.argv) > 1 else 24
proofs = int(sys.argv[2]) if len(sys.argv) > 2 else 6
sc.init(0)
FastStark.EAGER_COMMITS = "eager" in sys.argv[3:]
field, T, packed, air, boundary = workloads.synthetic_stark_instance(log_fri, 40)
stark = FastStark(field, 4, 40, 80, 2, T)
trace = DeviceTrace.from_packed(packed, field)
tz, tzc, root = stark.preprocess(device_resident=True)
times = []
for _ in range(proofs):
    sc.synchronize(); t0 = time.perf_counter()
    proof = stark.prove(trace, air, boundary, tz, tzc)
    sc.synchronize(); times.append(round((time.perf_counter() - t0) * 1e3, 2))
print("prove ms", times, "best %.2f median %.2f" % (min(times), sorted(times)[len(times) // 2]), "eager commits" if FastStark.EAGER_COMMITS else "roots pushed lazily", "sha", __import__("hashlib").sha256(proof).hexdigest()[:12], len(proof))
