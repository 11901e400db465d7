#!/usr/bin/env python3
"""One build + one evaluation + one interpolation over 2^log points (dev tool; run under rocprofv3 --kernel-trace --stats)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "stark-anatomy_amd"))
import starkcore as sc, synth
sc.init(0)
logk = int(sys.argv[1]) if len(sys.argv) > 1 else 20
k = 1 << logk
pts = sc.DeviceVector.from_bytes(synth.synth_packed(11, k).tobytes())
f = sc.DeviceVector.from_bytes(synth.synth_packed(12, k).tobytes())
for _ in range(3):
    tree = sc.PolyTree(pts)
    vals = tree.evaluate(f)
    back = tree.interpolate(vals)
    sc.synchronize()
    tree.free()
print("round trip", back.to_bytes() == f.to_bytes())
