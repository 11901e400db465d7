"""Pins the CPU oracle (oracle/stark_oracle.c + oracle/py_oracle.py) against golden vectors that
tests/golden/make_golden.py produced by importing the reference.  CPU only."""
import hashlib
import os

import pytest

from conftest import load_golden
from oracle import py_oracle as po
import synth

C = po.C
P = po.P


def packed(seed, n, start=0):
    return synth.synth_packed(seed, n, start).tobytes()


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def test_synth_three_ways():
    g = load_golden("field.json")
    ints = synth.synth_ints(1, 4)
    assert [str(v) for v in ints] == g["synth_seed1_first4"]
    assert synth.pack_ints(synth.synth_ints(7, 300, 5)) == packed(7, 300, 5) == C.synth(7, 300, 5)


def test_field_ops_c_and_py():
    g = load_golden("field.json")
    assert str(P) == g["p"] and str(po.GENERATOR) == g["generator"]
    for k, v in g["primitive_nth_root"].items():
        assert po.primitive_nth_root(1 << int(k)) == int(v)
    for a, b, r in g["mul"]:
        assert C.binop("mul", int(a), int(b)) == int(r)
    for a, b, r in g["add"]:
        assert C.binop("add", int(a), int(b)) == int(r)
    for a, b, r in g["sub"]:
        assert C.binop("sub", int(a), int(b)) == int(r)
    for a, b, r in g["div"]:
        assert C.binop("mul", int(a), C.inv(int(b))) == int(r)
    for a, e, r in g["pow"]:
        assert C.binop("pow", int(a), int(e)) == int(r) == pow(int(a), int(e), P)
    for a, r in g["inverse"].items():
        assert C.inv(int(a)) == int(r) == po.inv(int(a))
    for hx, r in g["sample"]:
        assert po.sample(bytes.fromhex(hx)) == int(r)
    # edge operands for the special-form reduction
    for a in (0, 1, P - 1, P - 2, (1 << 119), (1 << 119) - 1, (1 << 127), 407):
        for b in (0, 1, P - 1, (1 << 64) - 1, (1 << 119) + 1):
            assert C.binop("mul", a, b) == a * b % P
            assert C.binop("add", a, b) == (a + b) % P
            assert C.binop("sub", a, b) == (a - b) % P


@pytest.mark.parametrize("which", ["ntt", "intt"])
def test_ntt_golden(which):
    g = load_golden("ntt.json")
    for rec in g[which]:
        n = 1 << rec["logn"]
        data = packed(rec["seed"], n)
        root = int(rec["root"])
        out_c = getattr(C, which)(root, data, n)
        assert sha(out_c) == rec["sha256"], (which, rec["logn"])
        if "out" in rec:
            assert [str(v) for v in synth.unpack_ints(out_c)] == rec["out"]
        if rec["logn"] <= 10:
            out_py = getattr(po, which)(root, synth.synth_ints(rec["seed"], n))
            assert synth.pack_ints(out_py) == out_c
    assert [str(v) for v in po.ntt(po.primitive_nth_root(8), list(range(1, 9)))] == g["kat"]["ntt_w8_1to8"]
    assert [str(v) for v in po.intt(po.primitive_nth_root(8), list(range(1, 9)))] == g["kat"]["intt_w8_1to8"]


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(__file__), "golden", "ntt_big.json")), reason="no big golden")
def test_ntt_golden_big():
    g = load_golden("ntt_big.json")
    for rec in g["ntt"]:
        if rec["logn"] < 17:
            continue
        n = 1 << rec["logn"]
        assert sha(C.ntt(int(rec["root"]), packed(rec["seed"], n), n)) == rec["sha256"]


def test_ntt_root_checks():
    data = packed(1, 8)
    with pytest.raises(AssertionError):
        C.ntt(po.primitive_nth_root(16), data, 8)      # root^n != 1
    with pytest.raises(AssertionError):
        C.ntt(po.primitive_nth_root(4), data, 8)       # not primitive
    with pytest.raises(AssertionError):
        po.ntt(po.primitive_nth_root(4), synth.synth_ints(1, 8))
    with pytest.raises(AssertionError):
        po.ntt(po.primitive_nth_root(4), [1, 2, 3])    # not a power of two


def _poly_in(rec, key_seed, key_len, key_list):
    if key_list in rec:
        return [int(v) for v in rec[key_list]]
    return synth.synth_ints(rec[key_seed], rec[key_len])


def test_poly_golden_py():
    g = load_golden("poly.json")
    for rec in g["multiply"]:
        a = _poly_in(rec, "lhs_seed", "lhs_len", "lhs")
        b = _poly_in(rec, "rhs_seed", "rhs_len", "rhs")
        out = po.fast_multiply(a, b, int(rec["root"]), rec["order"])
        if "out" in rec:
            assert [str(v) for v in out] == rec["out"]
        else:
            assert len(out) == rec["out_len"] and sha(synth.pack_ints(out)) == rec["sha256"]
    for rec in g["coset_evaluate"]:
        c = [int(v) for v in rec["coeffs"]] if "coeffs" in rec else synth.synth_ints(rec["seed"], rec["m"])
        if rec["order"] <= 512:
            out = po.fast_coset_evaluate(c, int(rec["offset"]), int(rec["generator"]), rec["order"])
            assert sha(synth.pack_ints(out)) == rec["sha256"]
        out_c = C.coset_evaluate(synth.pack_ints(c), len(c), int(rec["offset"]), int(rec["generator"]), rec["order"])
        assert sha(out_c) == rec["sha256"]
    for rec in g["coset_divide"]:
        q = synth.synth_ints(rec["q_seed"], rec["q_len"])
        d = synth.synth_ints(rec["d_seed"], rec["d_len"])
        prod = po.schoolbook_mul(q, d)
        assert sha(synth.pack_ints(prod)) == rec["lhs_sha256"]
        out = po.fast_coset_divide(prod, d, int(rec["offset"]), int(rec["root"]), rec["order"])
        assert len(out) == rec["out_len"] and sha(synth.pack_ints(out)) == rec["sha256"]
    for rec in g["zerofier"]:
        out = po.fast_zerofier(synth.synth_ints(rec["seed"], rec["k"]), int(rec["root"]), rec["order"])
        assert [str(v) for v in out] == rec["out"]
    for rec in g["evaluate"]:
        out = po.fast_evaluate(synth.synth_ints(rec["poly_seed"], rec["poly_len"]), synth.synth_ints(rec["dom_seed"], rec["k"]), int(rec["root"]), rec["order"])
        assert [str(v) for v in out] == rec["out"]
    for rec in g["interpolate"]:
        if "omicron_order" in rec:
            om = po.primitive_nth_root(rec["omicron_order"])
            dom = [pow(om, i, P) for i in range(rec["k"])]
        else:
            dom = synth.synth_ints(rec["dom_seed"], rec["k"])
        out = po.fast_interpolate(dom, synth.synth_ints(rec["val_seed"], rec["k"]), int(rec["root"]), rec["order"])
        assert [str(v) for v in out] == rec["out"]
    for rec in g["tree_big"]:
        dom = synth.synth_ints(rec["dom_seed"], rec["k"])
        root, order = po.primitive_nth_root(1024), 1024
        if rec["what"] == "zerofier":
            out = po.fast_zerofier(dom, root, order)
        elif rec["what"] == "evaluate":
            out = po.fast_evaluate(synth.synth_ints(rec["poly_seed"], rec["poly_len"]), dom, root, order)
        else:
            out = po.fast_interpolate(dom, synth.synth_ints(rec["val_seed"], rec["k"]), root, order)
        assert len(out) == rec["out_len"] and sha(synth.pack_ints(out)) == rec["sha256"], rec["what"]
    for rec in g["scale"]:
        c = synth.synth_ints(rec["seed"], rec["m"])
        assert [str(v) for v in po.scale(c, int(rec["factor"]))] == rec["out"]
        assert synth.unpack_ints(C.scale(synth.pack_ints(c), len(c), int(rec["factor"]))) == [int(v) for v in rec["out"]]


def test_pointwise_c():
    a, b = packed(21, 100), packed(22, 100)
    ai, bi = synth.synth_ints(21, 100), synth.synth_ints(22, 100)
    assert synth.unpack_ints(C.pointwise_mul(a, b, 100)) == [x * y % P for x, y in zip(ai, bi)]
    assert synth.unpack_ints(C.pointwise_div(a, b, 100)) == [x * po.inv(y) % P for x, y in zip(ai, bi)]
    with pytest.raises(AssertionError):
        C.pointwise_div(a, bytes(16) + b[16:], 100)


def test_fold_golden():
    g = load_golden("fri.json")
    for rec in g["fold"]:
        if rec["kind"] == "test_fri_codeword":
            om = int(rec["omega"])
            cw = [po.evaluate(list(range(64)), pow(om, i, P)) for i in range(rec["n"])]
            assert sha(synth.pack_ints(cw)) == rec["in_sha256"]
        else:
            cw = synth.synth_ints(rec["seed"], rec["n"])
        args = (int(rec["alpha"]), int(rec["offset"]), int(rec["omega"]))
        out_c = C.fold(synth.pack_ints(cw), len(cw), *args)
        assert sha(out_c) == rec["sha256"]
        if len(cw) <= 256:
            assert synth.pack_ints(po.fold(cw, *args)) == out_c
        if "out" in rec:
            assert [str(v) for v in synth.unpack_ints(out_c)] == rec["out"]


def test_blake2b_vs_hashlib():
    # hashlib.blake2b is the reference's own dependency (code/merkle.py:1); RFC 7693 "abc" vector too.
    assert C.blake2b(b"abc").hex().startswith("ba80a53f981c4d0d6a2797b69f12f6e94c212f14685ac4b74b12bb6fdbffa2d1")
    for n in (0, 1, 39, 64, 127, 128, 129, 256, 300):
        msg = bytes((i * 7 + 3) & 0xFF for i in range(n))
        assert C.blake2b(msg) == hashlib.blake2b(msg).digest()


def test_merkle_golden():
    g = load_golden("merkle.json")
    for v, d in g["leaf_digest"]:
        assert C.leaf_bytes(int(v)) == str(int(v)).encode()
        assert C.blake2b(C.leaf_bytes(int(v))).hex() == d
    for v in (0, 9, 10, 10 ** 19 - 1, 10 ** 19, 10 ** 38, P - 1, (1 << 64) - 1, 1 << 64):
        assert C.leaf_bytes(v) == str(v).encode()
    for rec in g["commit"]:
        vals = [int(v) for v in rec["values"]] if "values" in rec else synth.synth_ints(rec["seed"], rec["n"])
        assert C.merkle_commit(synth.pack_ints(vals), len(vals)).hex() == rec["root"]
        if len(vals) <= 1024:
            assert po.merkle_commit(vals).hex() == rec["root"]
    for rec in g["open"]:
        vals = synth.synth_ints(rec["seed"], rec["n"])
        path = C.merkle_open(synth.pack_ints(vals), rec["n"], rec["index"])
        assert [d.hex() for d in path] == rec["path"]
        assert [d.hex() for d in po.merkle_open(rec["index"], vals)] == rec["path"]
    with pytest.raises(AssertionError):
        C.merkle_commit(packed(1, 3), 3)
