#!/usr/bin/env python3
"""Can torch initialise HIP after the library has been used in this process? (dev tool)  python tools/torch_after_probe.py [vec|tree|async|prove]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "stark-anatomy_amd"))
import starkcore as sc, synth
what = sys.argv[1] if len(sys.argv) > 1 else "vec"
sc.init(0)
v = sc.DeviceVector.from_bytes(synth.synth_packed(1, 1 << 12).tobytes())
if what in ("tree", "prove"):
    t = sc.MerkleTree.from_device(v); print("root", t.root[:4].hex())
if what in ("async", "prove"):
    t2 = sc.MerkleTree.from_device_async(v); print("async root", t2.root[:4].hex())
if what == "prove":
    from algebra import Field
    from fri import Fri
    from ip import ProofStream
    f = Field.main(); N = 1 << 12
    fr = Fri(f.generator(), f.primitive_nth_root(N), N, 4, 17)
    fr.prove(sc.DeviceCodeword(v, f), ProofStream()); print("proved")
import torch
try:
    print(what, "torch ok", torch.zeros(2, device="cuda").sum().item(), torch.cuda.device_count())
except Exception as e:
    print(what, "torch FAILED", repr(e)[:200])
