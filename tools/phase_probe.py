#!/usr/bin/env python3
"""where does a slow first phase come from?  times the library calls of FastStark._randomized_columns one by one (dev tool)"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "stark-anatomy_amd")); sys.path.insert(0, REPO)
import torch
import starkcore as sc
import workloads
from fast_stark import DeviceTrace
from sharded_stark import ShardedFastStark
log_fri = 20
sc.init(0); dev = torch.device("cuda", 0)
field, T, packed, air, boundary = workloads.synthetic_stark_instance(log_fri, 40)
stark = ShardedFastStark(field, 4, 40, 80, 2, T, 0, 1, dev)
trace = DeviceTrace.from_packed(packed, field)
tz, layer, root = stark.preprocess(device_resident=True)
lib = sc.lib()
def timed(label, fn):
    t0 = time.perf_counter(); r = fn(); dt = (time.perf_counter() - t0) * 1e3
    if dt > 0.5: print("   %-40s %.3f ms" % (label, dt))
    return r
for it in range(8):
    t0 = time.perf_counter(); stark.prove(trace, air, boundary, tz, layer); t1 = time.perf_counter()
    if it >= 4: torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("prove %d: %.2f ms (+%.2f sync)" % (it, (t1 - t0) * 1e3, (t2 - t1) * 1e3))
    if it >= 2:
        v = timed("DeviceVector alloc", lambda: sc.DeviceVector(T + 160))
        timed("memcpy_dev", lambda: sc._check(lib.sc_memcpy_dev(v.ptr, trace.columns[0].ptr, T, None)))
        timed("vec_upload", lambda: sc._check(lib.sc_vec_upload(v._h, T, b"\0" * 2560, 160)))
        timed("free", lambda: v.free())
