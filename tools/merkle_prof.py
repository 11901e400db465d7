#!/usr/bin/env python3
"""Build a few Merkle trees of 2^logn leaves (for rocprofv3 runs) -- dev tool.  usage: merkle_prof.py [logn] [reps]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "stark-anatomy_amd"))
import starkcore as sc, synth
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 24
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
sc.init(0)
v = sc.DeviceVector.from_bytes(synth.synth_packed(9, 1 << logn).tobytes())
for _ in range(reps):
    sc.MerkleTree.from_device(v).free()
sc.synchronize()
