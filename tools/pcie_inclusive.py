#!/usr/bin/env python3
"""The PCIe-inclusive rate of the HOST-buffer entry (sc_ntt: host vector in, host vector out), beside the device-resident rate bench.py
reports as `value` (dev tool, round 6).  The boundary also hands over host buffers -- ntt.ntt(root, [FieldElement...]) ends in sc_ntt --
and a caller that keeps its vectors on the host pays the bus both ways: this prints what that costs, from pageable numpy memory and from
the library's pinned pool (sc_host_alloc), forward + inverse, so that nobody mistakes `value` for it.
   python tools/pcie_inclusive.py [log2n=20] [pairs=20]"""
import ctypes, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "stark-anatomy_amd"))
import numpy as np
import starkcore as sc, synth
from workloads import nth_root

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n = 1 << log2n
sc.init(0)
lib = sc.lib()
root = sc.fe_bytes(nth_root(n))
src = synth.synth_packed(1, n).view(np.uint8).reshape(-1)          # 16 bytes per element


def timed(x, y, z):
    px, py, pz = (ctypes.c_void_p(a.ctypes.data) for a in (x, y, z))
    for _ in range(2):
        sc._check(lib.sc_ntt(px, py, n, root, 0))
        sc._check(lib.sc_ntt(py, pz, n, root, 1))
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(pairs):
            sc._check(lib.sc_ntt(px, py, n, root, 0))
            sc._check(lib.sc_ntt(py, pz, n, root, 1))
        dt = (time.perf_counter() - t0) / pairs
        best = dt if best is None or dt < best else best
    return best, bool(np.array_equal(x, z))


# pageable: what a numpy array of the caller is
x = src.copy(); y = np.empty_like(x); z = np.empty_like(x)
pageable, ok1 = timed(x, y, z)
# pinned: the library's pool
hx, hy, hz = (sc.HostBuffer(16 * n) for _ in range(3))
hx.array[:] = src
pinned, ok2 = timed(hx.array, hy.array, hz.array)
for name, t, ok in (("pageable numpy memory", pageable, ok1), ("pinned (sc_host_alloc)", pinned, ok2)):
    print("sc_ntt 2^%d forward + inverse, host vector in and out, %-24s: %8.3f ms per pair = %6.2f G el/s, %5.1f GB/s over the bus (64 B per element per pair)   round trip %s"
          % (log2n, name, t * 1e3, 2 * n / t / 1e9, 64.0 * n / t / 1e9, "ok" if ok else "WRONG"))
