"""GPU parity tests of the geometric-progression path of fast_zerofier / fast_evaluate / fast_interpolate (csrc/geoseq.cuh,
sc_geodomain_*; reference code/ntt.py:66-130 as called on the trace domain {omicron^i}, code/fast_stark.py:84-90): the CPU
oracle's restatement of the reference recursion on seeded inputs, the general subproduct tree (itself pinned to reference goldens)
at sizes the oracle cannot reach, and closed forms."""
import ctypes
import random

import pytest

from oracle import py_oracle as po
import synth

pytestmark = pytest.mark.gpu
P = po.P


@pytest.fixture(scope="module")
def sc():
    import starkcore
    assert starkcore.device_count() > 0, "no GPU visible: the HIP path is mandatory for these tests"
    starkcore.init()
    return starkcore


def progression(c, q, n):
    out, x = [], c % P
    for _ in range(n):
        out.append(x)
        x = x * q % P
    return out


def first_ratio(kind, n, order=512):
    root = po.primitive_nth_root(order)
    if kind == "omicron":                 # a prefix of a subgroup: the trace domain of fast_stark.py:84-90
        return 1, root
    if kind == "coset":
        return po.GENERATOR, root
    c, q = synth.synth_ints(8000 + n, 2)
    return c, q


def vec(sc, ints):
    return sc.DeviceVector.from_bytes(synth.pack_ints(ints)) if ints else sc.DeviceVector(1)


@pytest.mark.parametrize("kind", ["omicron", "coset", "arbitrary"])
@pytest.mark.parametrize("n", [2, 3, 4, 5, 7, 8, 9, 16, 17, 31, 32, 33, 50, 64, 65, 127, 129])
def test_vs_oracle(sc, n, kind):
    order = 512
    root = po.primitive_nth_root(order)
    c, q = first_ratio(kind, n)
    pts = progression(c, q, n)
    dom = sc.GeoDomain(c, q, n)
    assert synth.unpack_ints(dom.zerofier().to_bytes()) == po.fast_zerofier(pts, root, order)
    for m in sorted({0, 1, n // 2, n, n + 1, 3 * n + 2}):
        f = synth.synth_ints(8100 + n + m, m)
        fv = vec(sc, f)
        fv.n = m
        assert synth.unpack_ints(dom.evaluate(fv).to_bytes()) == [po.evaluate(f, x) for x in pts], (n, m)
    vals = synth.synth_ints(8200 + n, n)
    vals[1] = 0
    assert synth.unpack_ints(dom.interpolate(vec(sc, vals)).to_bytes()) == po.fast_interpolate(pts, vals, root, order)
    assert not any(dom.interpolate(vec(sc, [0] * n)).to_bytes())
    dom.free()


def test_detection_and_fallbacks(sc):
    om = po.primitive_nth_root(256)
    pts = progression(1, om, 100)
    assert isinstance(sc.domain_tables(synth.pack_ints(pts)), sc.GeoDomain)
    bent = list(pts)
    bent[57] = (bent[57] + 1) % P
    assert isinstance(sc.domain_tables(synth.pack_ints(bent)), sc.PolyTree)
    assert isinstance(sc.domain_tables(synth.pack_ints(synth.synth_ints(8300, 100))), sc.PolyTree)
    assert isinstance(sc.domain_tables(synth.pack_ints([0] + pts[1:])), sc.PolyTree)          # a zero first point is no progression
    # a progression that wraps around its subgroup repeats points: no progression tables (GeoDomain.create says so), and through
    # the tree the reference's behaviour -- interpolation is a division by zero, zerofier and evaluation do not mind
    wrapped = progression(1, po.primitive_nth_root(64), 70)
    assert sc.GeoDomain.create(1, po.primitive_nth_root(64), 70) is None
    assert isinstance(sc.domain_tables(synth.pack_ints(wrapped)), sc.PolyTree)
    out = ctypes.create_string_buffer(16 * 70)
    with pytest.raises(AssertionError, match="divide by zero"):
        sc._check(sc.lib().sc_interpolate(synth.pack_ints(wrapped), synth.pack_ints(synth.synth_ints(8301, 70)), 70, out))
    assert sc.GeoDomain.create(5, 7, 1) is None


def test_c_abi_host_functions_take_the_progression_path(sc):
    """sc_zerofier / sc_evaluate / sc_interpolate on progressions (the entry points the host shim's fast_* functions use)"""
    order = 512
    root = po.primitive_nth_root(order)
    for n, (c, q) in ((40, (1, root)), (100, (po.GENERATOR, po.primitive_nth_root(128))), (33, tuple(synth.synth_ints(8400, 2)))):
        pts = progression(c, q, n)
        out = ctypes.create_string_buffer(16 * (n + 1))
        sc._check(sc.lib().sc_zerofier(synth.pack_ints(pts), n, out))
        assert synth.unpack_ints(out.raw) == po.fast_zerofier(pts, root, order)
        f = synth.synth_ints(8401, 2 * n + 3)
        sc._check(sc.lib().sc_evaluate(synth.pack_ints(f), len(f), synth.pack_ints(pts), n, out))
        assert synth.unpack_ints(out.raw)[:n] == [po.evaluate(f, x) for x in pts]
        vals = synth.synth_ints(8402, n)
        sc._check(sc.lib().sc_interpolate(synth.pack_ints(pts), synth.pack_ints(vals), n, out))
        assert synth.unpack_ints(out.raw)[:n] == po.fast_interpolate(pts, vals, root, order)


def test_host_shim_on_the_trace_domain(sc):
    """ntt.fast_interpolate / fast_evaluate / fast_zerofier with the domain of fast_stark.py:84-90 as a list of FieldElements"""
    from algebra import Field, FieldElement
    from univariate import Polynomial
    import ntt
    field = Field.main()
    order = 256
    om = field.primitive_nth_root(order)
    n = 100
    dom = [om ^ i for i in range(n)]
    vals = [FieldElement(v, field) for v in synth.synth_ints(8500, n)]
    poly = ntt.fast_interpolate(dom, vals, om, order)
    assert [c.value for c in poly.coefficients] == po.fast_interpolate([d.value for d in dom], [v.value for v in vals], om.value, order)
    assert ntt.fast_evaluate(poly, dom, om, order) == vals
    z = ntt.fast_zerofier(dom, om, order)
    assert [c.value for c in z.coefficients] == po.fast_zerofier([d.value for d in dom], om.value, order)
    assert isinstance(ntt._device_tree(dom), sc.GeoDomain)


@pytest.mark.parametrize("logk", [10, 16, 20])
def test_full_subgroup_closed_forms(sc, logk):
    """ord(ratio) == n: A_n = 0 and the zerofier is X^n - 1, evaluation is the ntt, interpolation the intt"""
    K = 1 << logk
    w = po.primitive_nth_root(K)
    dom = sc.GeoDomain(1, w, K)
    z = dom.zerofier().to_bytes()
    assert z[:16] == (P - 1).to_bytes(16, "little") and z[-16:] == (1).to_bytes(16, "little") and not any(z[16:-16])
    f = synth.synth_packed(8600 + logk, K).tobytes()
    vals = dom.evaluate(sc.DeviceVector.from_bytes(f))
    ntt_out = ctypes.create_string_buffer(16 * K)
    sc._check(sc.lib().sc_ntt(f, ntt_out, K, sc.fe_bytes(w), 0))
    assert vals.to_bytes() == ntt_out.raw
    assert dom.interpolate(vals).to_bytes() == f
    dom.free()


@pytest.mark.parametrize("n", [1000, (1 << 16) + 5, (1 << 20) - 160, 1 << 20])
@pytest.mark.parametrize("coset", [False, True])
def test_agrees_with_the_subproduct_tree(sc, n, coset):
    """sizes the oracle cannot reach: the same zerofier, values and interpolant as the general tree over the same points (the
    tree is pinned to the reference's goldens and the oracle, tests/test_gpu_polytree.py); n = 2^20 - 160 and omicron of order
    2^22 is the trace domain of BASELINE configs[4]"""
    if coset and n > (1 << 17):
        pytest.skip("one coset case per size class is enough")
    q = po.primitive_nth_root(1 << 22)
    c = po.GENERATOR if coset else 1
    dom = sc.GeoDomain(c, q, n)
    ones = sc.DeviceVector.from_bytes(c.to_bytes(16, "little") * n)
    pts = sc.DeviceVector(n)
    sc._check(sc.lib().sc_scale_dev(ones.ptr, pts.ptr, n, sc.fe_bytes(q), None))
    tree = sc.PolyTree(pts)
    assert dom.zerofier().to_bytes() == tree.zerofier().to_bytes()
    vals = sc.DeviceVector.from_bytes(synth.synth_packed(8700, n).tobytes())
    poly = dom.interpolate(vals)
    assert poly.to_bytes() == tree.interpolate(vals).to_bytes()
    assert dom.evaluate(poly).to_bytes() == vals.to_bytes()
    f = sc.DeviceVector.from_bytes(synth.synth_packed(8701, n // 2 + 7).tobytes())
    assert dom.evaluate(f).to_bytes() == tree.evaluate(f).to_bytes()
    if n <= 1 << 17:
        g = sc.DeviceVector.from_bytes(synth.synth_packed(8702, 2 * n + n // 2 + 3).tobytes())     # more coefficients than points: chunks
        assert dom.evaluate(g).to_bytes() == tree.evaluate(g).to_bytes()
    # ... and the DEFINITIONS, by nothing but Python integers (both paths above share the transform kernels): at seeded positions of
    # the progression the interpolant takes the given values, the zerofier vanishes, the evaluation of f is Horner's; off the domain
    # the zerofier is the product of (y - x_i)  (ntt.py:66-130)
    coeffs, zer = synth.unpack_ints(poly.to_bytes()), synth.unpack_ints(dom.zerofier().to_bytes())
    assert len(coeffs) <= n and len(zer) == n + 1 and zer[n] == 1
    f_ints, f_vals = synth.unpack_ints(f.to_bytes()), dom.evaluate(f)
    rng = random.Random(8703 + n)
    for j in (0, n - 1, rng.randrange(n)):
        x = c * pow(q, j, P) % P
        assert po.evaluate(coeffs, x) == synth.synth_ints(8700, 1, j)[0], j
        assert po.evaluate(zer, x) == 0, j
        assert po.evaluate(f_ints, x) == int.from_bytes(f_vals.to_bytes(j, 1), "little"), j
    y, prod, x = synth.synth_ints(8704, 1)[0], 1, c
    for _ in range(n):
        prod, x = prod * (y - x) % P, x * q % P
    assert po.evaluate(zer, y) == prod
    dom.free()
    tree.free()
