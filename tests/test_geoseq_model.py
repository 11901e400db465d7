"""The geometric-progression algorithm the device runs for fast_zerofier / fast_evaluate / fast_interpolate on domains
{c * q^i} (tests/emu/geoseq_model.py mirrors csrc/geoseq.cuh array by array) against the oracle's restatement of the reference
recursion (code/ntt.py:66-130).  CPU only."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
from oracle import py_oracle as po
import geoseq_model as gm
import synth

P = po.P


@pytest.mark.parametrize("n", [2, 3, 4, 5, 7, 8, 9, 16, 17, 31, 33, 50])
@pytest.mark.parametrize("kind", ["omicron", "coset", "arbitrary"])
def test_model_matches_oracle(n, kind):
    order = 128
    root = po.primitive_nth_root(order)
    if kind == "omicron":                                   # the trace domain of fast_stark.py:84-90: a prefix of a subgroup
        c, q = 1, root
    elif kind == "coset":
        c, q = po.GENERATOR, root
    else:
        c, q = synth.synth_ints(7000 + n, 2)
    dom = gm.GeometricDomain(c, q, n)
    pts = dom.points()
    assert len(set(pts)) == n
    assert dom.zerofier() == po.fast_zerofier(pts, root, order)
    for m in sorted({0, 1, n // 2, n, n + 1, 3 * n + 2}):
        f = synth.synth_ints(7100 + n + m, m)
        assert dom.evaluate(f) == [po.evaluate(f, x) for x in pts], (n, m)
    vals = synth.synth_ints(7200 + n, n)
    assert dom.interpolate(vals) == po.fast_interpolate(pts, vals, root, order)


def test_full_subgroup_and_repeated_points():
    K = 32
    w = po.primitive_nth_root(K)
    dom = gm.GeometricDomain(1, w, K)                       # ord(q) == n: A_n = 0, Z = X^n - 1
    assert dom.zerofier() == [P - 1] + [0] * (K - 1) + [1]
    f = synth.synth_ints(7300, K)
    assert dom.evaluate(f) == po.ntt(w, f)
    assert dom.interpolate(po.ntt(w, f)) == f
    with pytest.raises(ValueError):
        gm.GeometricDomain(1, w, K + 1)                     # the progression wraps: not a domain of distinct points
