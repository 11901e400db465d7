#!/usr/bin/env python3
"""Which library calls of one FastStark.prove make the host wait (dev tool): every entry of libstarkcore is wrapped, and the calls
of the last of a few proofs that took longer than `min_us` are listed in order with their duration and the Python line that made
them -- the places where the host stands still for the device (a degree, an exactness flag, a root) instead of running ahead.
   python tools/sync_points.py [log2_fri=24] [min_us=25]"""
import os, sys, time, traceback
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "stark-anatomy_amd")); sys.path.insert(0, REPO)
import starkcore as sc
import workloads
from fast_stark import DeviceTrace, FastStark

log_fri = int(sys.argv[1]) if len(sys.argv) > 1 else 24
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 25.0
sc.init(0)
real = sc.lib()
log = []


class Proxy:
    def __getattr__(self, name):
        fn = getattr(real, name)
        if not name.startswith("sc_"):
            return fn

        def call(*a):
            t0 = time.perf_counter()
            r = fn(*a)
            dt = (time.perf_counter() - t0) * 1e6
            if dt >= min_us:
                frames = [f for f in traceback.extract_stack()[:-1] if "tools/sync_points" not in f.filename]
                where = " <- ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in frames[-3:][::-1])
                log.append((time.perf_counter(), name, dt, where))
            return r
        setattr(self, name, call)
        return call


field, T, packed, air, boundary = workloads.synthetic_stark_instance(log_fri, 40)
stark = FastStark(field, 4, 40, 80, 2, T)
trace = DeviceTrace.from_packed(packed, field)
tz, tzc, root = stark.preprocess(device_resident=True)
for _ in range(3):
    stark.prove(trace, air, boundary, tz, tzc)
sc._lib = Proxy()
sc.synchronize()
log.clear()
t0 = time.perf_counter()
stark.prove(trace, air, boundary, tz, tzc)
total = (time.perf_counter() - t0) * 1e3
sc._lib = real
print(f"FastStark.prove at FRI 2^{log_fri} with every library call wrapped: {total:.2f} ms; calls of {min_us:.0f} us or more ({sum(d for _, _, d, _ in log) / 1e3:.2f} ms together):")
for t, name, dt, where in log:
    print(f"  at {(t - t0) * 1e3 - dt / 1e3:7.2f} ms  {dt:8.1f} us  {name:32s} {where}")
