#!/usr/bin/env python3
"""Per-phase breakdown and cProfile of sharded_stark.ShardedFastStark.prove at world 1 on the synthetic 2-register AIR with a
device-resident trace (dev tool).   python tools/sharded_stark_profile.py [log2_fri=20] [--no-cprofile]"""
import cProfile, os, pstats, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "stark-anatomy_amd"))
sys.path.insert(0, REPO)
import torch
import starkcore as sc
import workloads
from fast_stark import DeviceTrace
from sharded_stark import ShardedFastStark
log_fri = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 20
s = 40
sc.init(0); dev = torch.device("cuda", 0)
t0 = time.perf_counter()
field, T, packed, air, boundary = workloads.synthetic_stark_instance(log_fri, s)
print("trace generated on the host in %.2f s (T = %d rows, 2 registers)" % (time.perf_counter() - t0, T))
stark = ShardedFastStark(field, 4, s, 2 * s, 2, T, 0, 1, dev)
trace = DeviceTrace.from_packed(packed, field)
t0 = time.perf_counter(); tz, layer, root = stark.preprocess(device_resident=True); torch.cuda.synchronize(); print("preprocess ms", round((time.perf_counter() - t0) * 1e3, 2))
runs = max([int(a.split("=")[1]) for a in sys.argv if a.startswith("--runs=")] + [3])
import gc
gc_pauses = []
def _gc_watch(phase, info, _t=[0.0]):
    if phase == "start": _t[0] = time.perf_counter()
    else:
        ms = (time.perf_counter() - _t[0]) * 1e3
        if ms > 2: gc_pauses.append((info.get("generation"), round(ms, 1)))
gc.callbacks.append(_gc_watch)
times = []
for k in range(runs):
    stark.phase_log = [] if "--outliers" in sys.argv else None
    t0 = time.perf_counter(); proof = stark.prove(trace, air, boundary, tz, layer); torch.cuda.synchronize(); times.append(round((time.perf_counter() - t0) * 1e3, 2))
    if times[-1] > 3 * min(times):
        st = torch.cuda.memory_stats()
        print("run", k, times[-1], "ms; torch reserved MB", torch.cuda.memory_reserved() >> 20, "device allocs", st.get("num_device_alloc"), "frees", st.get("num_device_free"),
              "retries", st.get("num_alloc_retries"), "free device MB", torch.cuda.mem_get_info()[0] >> 20, "gc pauses", gc_pauses)
    if stark.phase_log and times[-1] > 3 * min(times):
        print("run", k, times[-1], "ms; phases over 3 ms:", [(n[:40], round(s_ * 1e3, 1)) for n, s_ in stark.phase_log if s_ > 3e-3], "gc pauses", gc_pauses)
    del gc_pauses[:]
stark.phase_log = None
print("prove ms", times)
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
