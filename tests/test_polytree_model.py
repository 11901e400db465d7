"""The level-batched subproduct-tree algorithm the device runs (tests/emu/polytree_model.py mirrors csrc/polytree.cuh launch by
launch) against the oracle's restatement of the reference recursion (code/ntt.py:66-130).  CPU only."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
from oracle import py_oracle as po
import polytree_model as pm
import synth

P = po.P


@pytest.mark.parametrize("k", [1, 2, 3, 4, 5, 7, 8, 9, 16, 17, 31, 33, 50])
def test_model_matches_oracle(k):
    order = 128
    root = po.primitive_nth_root(order)
    pts = synth.synth_ints(9000 + k, k)
    if k > 3:
        pts[2] = 0
    t = pm.Tree(pts)
    assert t.zerofier() == po.fast_zerofier(pts, root, order)
    for m in sorted({0, 1, k // 2, t.K}):
        f = synth.synth_ints(9100 + k + m, m)
        assert t.evaluate(f) == [po.evaluate(f, x) for x in pts], (k, m)
    vals = synth.synth_ints(9200 + k, k)
    assert t.interpolate(vals) == po.fast_interpolate(pts, vals, root, order)


def test_model_on_a_subgroup():
    K = 32
    w = po.primitive_nth_root(K)
    t = pm.Tree([pow(w, i, P) for i in range(K)])
    assert t.zerofier() == [P - 1] + [0] * (K - 1) + [1]
    f = synth.synth_ints(9300, K)
    assert t.evaluate(f) == po.ntt(w, f)
    assert t.interpolate(po.ntt(w, f)) == f
