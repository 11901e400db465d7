#!/usr/bin/env python3
"""Free device memory before / after many iterations of every kind of object the library hands out (vectors, trees, subproduct
trees, proofs) -- dev tool; the pool may keep up to its cap, growth beyond that would be a leak."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "stark-anatomy_amd"))
import torch
import starkcore as sc, synth
from algebra import Field
from fri import Fri
from ip import ProofStream
sc.init(0)
field = Field.main()
GEN = 85408008396924667383611388730472331217
def free_gb():
    sc.synchronize(); torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0] / 2**30
N = 1 << 18
om = field.primitive_nth_root(N)
coeffs = sc.DeviceVector.from_bytes(synth.synth_packed(1, N // 4).tobytes())
pts = synth.synth_packed(2, 5000).tobytes()
vals = sc.DeviceVector.from_bytes(synth.synth_packed(3, 5000).tobytes())
marks = []
for it in range(301):
    cwv = sc.DeviceVector(N)
    sc._check(sc.lib().sc_coset_evaluate_dev(coeffs.ptr, N // 4, sc.fe_bytes(GEN), sc.fe_bytes(om.value), N, cwv.ptr, None))
    fr = Fri(field.generator(), om, N, 4, 20)
    ps = ProofStream()
    fr.prove(sc.DeviceCodeword(cwv, field), ps)
    tree = sc.PolyTree(pts)
    back = tree.evaluate(tree.interpolate(vals))
    tree.free()
    del cwv, ps, back
    if it in (0, 10, 100, 200, 300):
        marks.append((it, round(free_gb(), 3)))
print(marks)
