"""GPU parity tests of the device subproduct tree (sc_zerofier / sc_evaluate / sc_interpolate, sc_polytree_*): reference
goldens (code/ntt.py:66-130 outputs), the CPU oracle on seeded inputs, and size-independent properties at full sizes."""
import ctypes
import hashlib

import pytest

from conftest import load_golden
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


def zerofier(sc, pts):
    out = ctypes.create_string_buffer(16 * (len(pts) + 1))
    sc._check(sc.lib().sc_zerofier(synth.pack_ints(pts), len(pts), out))
    return synth.unpack_ints(out.raw) if pts else []


def evaluate(sc, coeffs, pts):
    out = ctypes.create_string_buffer(16 * max(1, len(pts)))
    sc._check(sc.lib().sc_evaluate(synth.pack_ints(coeffs), len(coeffs), synth.pack_ints(pts), len(pts), out))
    return synth.unpack_ints(out.raw)[:len(pts)]


def interpolate(sc, pts, vals):
    out = ctypes.create_string_buffer(16 * max(1, len(pts)))
    sc._check(sc.lib().sc_interpolate(synth.pack_ints(pts), synth.pack_ints(vals), len(pts), out))
    return synth.unpack_ints(out.raw)[:len(pts)]


def test_reference_goldens(sc):
    g = load_golden("poly.json")
    for rec in g["zerofier"]:
        assert [str(v) for v in zerofier(sc, synth.synth_ints(rec["seed"], rec["k"]))] == rec["out"], rec["k"]
    for rec in g["evaluate"]:
        got = evaluate(sc, synth.synth_ints(rec["poly_seed"], rec["poly_len"]), synth.synth_ints(rec["dom_seed"], rec["k"]))
        assert [str(v) for v in got] == rec["out"], (rec["k"], rec["poly_len"])
    for rec in g["interpolate"]:
        if "omicron_order" in rec:
            om = po.primitive_nth_root(rec["omicron_order"])
            dom = [pow(om, i, P) for i in range(rec["k"])]
        else:
            dom = synth.synth_ints(rec["dom_seed"], rec["k"])
        got = interpolate(sc, dom, synth.synth_ints(rec["val_seed"], rec["k"]))
        assert [str(v) for v in got] == rec["out"], rec["k"]
    for rec in g.get("tree_big", []):
        dom = synth.synth_ints(rec["dom_seed"], rec["k"])
        if rec["what"] == "zerofier":
            got = zerofier(sc, dom)
        elif rec["what"] == "evaluate":
            got = evaluate(sc, synth.synth_ints(rec["poly_seed"], rec["poly_len"]), dom)
        else:
            got = interpolate(sc, dom, synth.synth_ints(rec["val_seed"], rec["k"]))
        assert len(got) == rec["out_len"] and hashlib.sha256(synth.pack_ints(got)).hexdigest() == rec["sha256"], rec


@pytest.mark.parametrize("k", [1, 2, 3, 4, 5, 7, 8, 9, 16, 17, 31, 32, 33, 50, 64, 65, 127, 129])
def test_vs_oracle(sc, k):
    order = 512
    root = po.primitive_nth_root(order)
    pts = synth.synth_ints(7000 + k, k)
    if k > 3:
        pts[2] = 0                                   # the point 0 is also what the tree pads with
    assert zerofier(sc, pts) == po.fast_zerofier(pts, root, order)
    for m in sorted({0, 1, k // 2, k, (1 << max(0, (k - 1).bit_length())), k + 3, 3 * k + 1}):
        f = synth.synth_ints(7100 + k + m, m)
        assert evaluate(sc, f, pts) == [po.evaluate(f, x) for x in pts], (k, m)
    vals = synth.synth_ints(7200 + k, k)
    if k > 1:
        vals[1] = 0
    assert interpolate(sc, pts, vals) == po.fast_interpolate(pts, vals, root, order)
    assert interpolate(sc, pts, [0] * k) == [0] * k


def test_repeated_point_is_a_division_by_zero(sc):
    pts = synth.synth_ints(7300, 40)
    pts[17] = pts[3]
    with pytest.raises(AssertionError, match="divide by zero"):
        interpolate(sc, pts, synth.synth_ints(7301, 40))
    # evaluation and the zerofier do not mind repeated points
    f = synth.synth_ints(7302, 33)
    assert evaluate(sc, f, pts) == [po.evaluate(f, x) for x in pts]
    root = po.primitive_nth_root(128)
    assert zerofier(sc, pts) == po.fast_zerofier(pts, root, 128)


@pytest.mark.parametrize("logk", [10, 16, 20])
def test_subgroup_domain_closed_forms(sc, logk):
    """On the full subgroup of order K: zerofier = X^K - 1, evaluation = ntt, interpolation = intt (all through the tree,
    including -- at 2^20 -- the levels whose columns are longer than the batched plans)."""
    K = 1 << logk
    w = po.primitive_nth_root(K)
    # powers of w: the ntt of the delta at index 1 is w^i
    delta = bytearray(16 * K)
    delta[16] = 1
    powers = po.C.ntt(w, bytes(delta), K)
    tree = sc.PolyTree(powers)
    z = tree.zerofier().to_bytes()
    assert z[:16] == (P - 1).to_bytes(16, "little") and z[-16:] == (1).to_bytes(16, "little") and not any(z[16:-16])
    f = synth.synth_packed(7400 + logk, K).tobytes()
    fv = sc.DeviceVector.from_bytes(f)
    vals = tree.evaluate(fv)
    ntt_out = ctypes.create_string_buffer(16 * K)
    sc._check(sc.lib().sc_ntt(f, ntt_out, K, sc.fe_bytes(w), 0))
    assert vals.to_bytes() == ntt_out.raw
    back = tree.interpolate(vals)
    assert back.to_bytes() == f
    tree.free()


@pytest.mark.parametrize("k", [1000, (1 << 16) + 5, (1 << 18) - 1])
def test_round_trips_on_arbitrary_points(sc, k):
    pts = synth.synth_packed(7500, k).tobytes()
    tree = sc.PolyTree(pts)
    z = tree.zerofier()
    assert z.n == k + 1 and z.to_bytes(k, 1) == (1).to_bytes(16, "little")
    assert not any(tree.evaluate(z).to_bytes())                                 # the zerofier vanishes on its points (m = k + 1 <= K or chunked)
    f = sc.DeviceVector.from_bytes(synth.synth_packed(7501, k).tobytes())
    vals = tree.evaluate(f)
    # spot-check values against Horner on the host
    fi = synth.synth_ints(7501, k)
    pi = synth.synth_ints(7500, k)
    got = synth.unpack_ints(vals.to_bytes())
    for i in (0, 1, k // 2, k - 1):
        assert got[i] == po.evaluate(fi, pi[i]), i
    assert tree.interpolate(vals).to_bytes() == f.to_bytes()
    # the definitions in Python integers, for an interpolant that is NOT a round trip of the evaluation kernels: it takes the given
    # values at its points and has fewer than k coefficients; the zerofier vanishes on the points and off them is prod (y - x_i)
    given = synth.synth_ints(7502, k)
    interpolant = synth.unpack_ints(tree.interpolate(sc.DeviceVector.from_bytes(synth.pack_ints(given))).to_bytes())
    zer = synth.unpack_ints(z.to_bytes())
    assert len(interpolant) <= k
    for i in (0, k - 1, (7 * k) // 11):
        assert po.evaluate(interpolant, pi[i]) == given[i], i
        assert po.evaluate(zer, pi[i]) == 0, i
    y, prod = synth.synth_ints(7503, 1)[0], 1
    for x in pi:
        prod = prod * (y - x) % P
    assert po.evaluate(zer, y) == prod
    if k <= 1 << 17:
        # a polynomial with more coefficients than the padded domain is evaluated in chunks
        m = 2 * k + k // 2 + 3
        big = synth.synth_ints(7502, m)
        got = synth.unpack_ints(tree.evaluate(sc.DeviceVector.from_bytes(synth.pack_ints(big))).to_bytes())
        for i in (0, 2, k - 1):
            assert got[i] == po.evaluate(big, pi[i]), i
    tree.free()
