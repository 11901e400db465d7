#!/usr/bin/env python3
"""Per-phase breakdown and cProfile of sharded_stark.ShardedFastStark.prove at world 1 on the synthetic 2-register AIR with a
device-resident trace (dev tool).   python tools/sharded_stark_profile.py [log2_fri=20] [--no-cprofile]"""
import cProfile, os, pstats, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "stark-anatomy_amd"))
sys.path.insert(0, REPO)
import torch
import starkcore as sc
import bench
from fast_stark import DeviceTrace
from sharded_stark import ShardedFastStark
log_fri = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 20
s = 40
sc.init(0); dev = torch.device("cuda", 0)
t0 = time.perf_counter()
field, T, packed, air, boundary = bench.synthetic_stark_instance(log_fri, s)
print("trace generated on the host in %.2f s (T = %d rows, 2 registers)" % (time.perf_counter() - t0, T))
stark = ShardedFastStark(field, 4, s, 2 * s, 2, T, 0, 1, dev)
trace = DeviceTrace.from_packed(packed, field)
t0 = time.perf_counter(); tz, layer, root = stark.preprocess(device_resident=True); torch.cuda.synchronize(); print("preprocess ms", round((time.perf_counter() - t0) * 1e3, 2))
for _ in range(3):
    t0 = time.perf_counter(); proof = stark.prove(trace, air, boundary, tz, layer); torch.cuda.synchronize(); print("prove ms", round((time.perf_counter() - t0) * 1e3, 2))
stark.phase_log = []
t0 = time.perf_counter(); stark.prove(trace, air, boundary, tz, layer); total = time.perf_counter() - t0
print("per phase (device waited for after each phase; %.2f ms in total this way):" % (total * 1e3))
for name, sec in stark.phase_log:
    print("  %8.3f ms  %s" % (sec * 1e3, name))
stark.phase_log = None
if "--no-cprofile" not in sys.argv:
    pr = cProfile.Profile(); pr.enable(); stark.prove(trace, air, boundary, tz, layer); torch.cuda.synchronize(); pr.disable()
    stats = pstats.Stats(pr).stats
    for title, key in (("by own time", 2), ("by cumulative time", 3)):
        print(title)
        print("%8s %10s %10s  %s" % ("calls", "own us", "cum us", "function"))
        for (fn, line, name), row in sorted(stats.items(), key=lambda kv: -kv[1][key])[:28 if key == 2 else 45]:
            print("%8d %10.0f %10.0f  %s:%d(%s)" % (row[1], row[2] * 1e6, row[3] * 1e6, os.path.basename(fn), line, name))
t0 = time.perf_counter(); ok = stark.verify(proof, air, boundary, root); print("verify", ok, "in %.2f s" % (time.perf_counter() - t0))
