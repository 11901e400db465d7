#!/usr/bin/env python3
"""BASELINE configs[4] on ONE GPU, phase by phase: fast_stark.FastStark (plain transforms) next to sharded_stark.ShardedFastStark at
world 1 (four-step slabs, slab-local FRI) -- what the sharded path costs a rank that no second rank shares (dev tool).
python tools/stark_phase_compare.py [log2_fri=24] [runs=8]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "stark-anatomy_amd")); sys.path.insert(0, REPO)
import torch
import starkcore as sc
import workloads
from fast_stark import DeviceTrace, FastStark
from sharded_stark import ShardedFastStark
log_fri = int(sys.argv[1]) if len(sys.argv) > 1 else 24
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
s = 40
sc.init(0); dev = torch.device("cuda", 0)
field, T, packed, air, boundary = workloads.synthetic_stark_instance(log_fri, s)
trace = DeviceTrace.from_packed(packed, field)
provers = [("plain (fast_stark.FastStark)", FastStark(field, 4, s, 2 * s, 2, T)), ("sharded at world 1 (sharded_stark.ShardedFastStark)", ShardedFastStark(field, 4, s, 2 * s, 2, T, 0, 1, dev))]
for name, stark in provers:
    tz, committed, root = stark.preprocess(device_resident=True)
    times = []
    for _ in range(runs + 2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        proof = stark.prove(trace, air, boundary, tz, committed)
        torch.cuda.synchronize(); times.append((time.perf_counter() - t0) * 1e3)
    times = times[2:]
    print("%s, FRI 2^%d: best %.2f ms, median %.2f ms of %d; proof %d bytes" % (name, log_fri, min(times), sorted(times)[len(times) // 2], runs, len(proof)))
    best = None
    for _ in range(3):
        stark.phase_log = []
        stark.prove(trace, air, boundary, tz, committed)
        log, stark.phase_log = stark.phase_log, None
        if best is None or sum(t for _, t in log) < sum(t for _, t in best):
            best = log
    print("  per phase (device waited for after each; %.2f ms this way):" % (1e3 * sum(t for _, t in best)))
    for phase, sec in best:
        print("  %8.3f ms  %s" % (sec * 1e3, phase))
