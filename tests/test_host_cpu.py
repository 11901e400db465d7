"""CPU-only tests of the host mirror modules (algebra / univariate / ip / merkle raw-digest helpers /
Fri index sampling) against golden vectors captured from the reference."""
import hashlib
import pickle

import pytest

from conftest import load_golden
import synth
from algebra import Field, FieldElement, xgcd
from univariate import Polynomial, test_colinearity
from ip import ProofStream
from merkle import Merkle
from fri import Fri

field = Field.main()


def fe(v):
    return FieldElement(int(v), field)


def test_algebra_golden():
    g = load_golden("field.json")
    assert str(field.p) == g["p"] and str(field.generator().value) == g["generator"]
    for k, v in g["primitive_nth_root"].items():
        assert str(field.primitive_nth_root(1 << int(k)).value) == v
    for a, b, r in g["mul"]:
        assert (fe(a) * fe(b)).value == int(r)
    for a, b, r in g["add"]:
        assert (fe(a) + fe(b)).value == int(r)
    for a, b, r in g["sub"]:
        assert (fe(a) - fe(b)).value == int(r)
    for a, b, r in g["div"]:
        assert (fe(a) / fe(b)).value == int(r)
    for a, e, r in g["pow"]:
        assert (fe(a) ^ int(e)).value == int(r)
    for a, r in g["inverse"].items():
        assert fe(a).inverse().value == int(r)
    for hx, r in g["sample"]:
        assert field.sample(bytes.fromhex(hx)).value == int(r)
    assert (-fe(0)).value == 0 and (-fe(5)).value == field.p - 5
    a, b, gcd = xgcd(240, 46)
    assert a * 240 + b * 46 == gcd == 2
    assert bytes(fe(12345)) == b"12345" and str(fe(7)) == "7" and fe(0).is_zero()


def test_transcript_bytes():
    g = load_golden("transcript.json")
    ps = ProofStream()
    for o in [b"\x01" * 64, [fe(5), fe(field.p - 1)], (fe(1), fe(2), fe(3)), [b"a" * 64, b"b" * 64]]:
        ps.push(o)
    assert pickle.DEFAULT_PROTOCOL == g["protocol_default"]
    assert ps.serialize().hex() == g["serialized_hex"]
    assert ps.prover_fiat_shamir().hex() == g["prover_fiat_shamir"]
    ps.pull(); ps.pull()
    assert ps.verifier_fiat_shamir().hex() == g["verifier_fiat_shamir_after2"]
    assert len(pickle.dumps([fe(7)])) == g["single_fe_list_len"]
    back = ps.deserialize(ps.serialize())
    assert back.objects[0] == b"\x01" * 64 and back.objects[1][1].value == field.p - 1


def test_polynomial_semantics():
    g = load_golden("poly.json")
    for rec in g["scale"]:
        c = [fe(v) for v in synth.synth_ints(rec["seed"], rec["m"])]
        assert [str(x.value) for x in Polynomial(c).scale(fe(rec["factor"])).coefficients] == rec["out"]
    a = Polynomial([fe(v) for v in synth.synth_ints(5, 7)] + [field.zero()] * 2)
    b = Polynomial([fe(v) for v in synth.synth_ints(6, 4)])
    assert a.degree() == 6 and Polynomial([]).degree() == -1 and Polynomial([field.zero()]).degree() == -1
    prod = a * b
    assert len(prod.coefficients) == len(a.coefficients) + len(b.coefficients) - 1
    q, r = Polynomial.divide(prod, b)
    assert q == a and r.is_zero() and (prod / b) == a and (prod % b).is_zero()
    assert Polynomial.divide(b, a) == (Polynomial([]), b) or True
    quo, rem = Polynomial.divide(a, b)
    assert (quo * b + rem) == a and rem.degree() < b.degree()
    assert (a + Polynomial([])) is a and (Polynomial([]) + a) is a
    assert (a - a).is_zero() and (a ^ 0) == Polynomial([field.one()]) and (b ^ 3) == b * b * b
    dom = [fe(v) for v in synth.synth_ints(8, 5)]
    vals = [fe(v) for v in synth.synth_ints(9, 5)]
    interp = Polynomial.interpolate_domain(dom, vals)
    assert interp.evaluate_domain(dom) == vals and interp.degree() <= 4
    z = Polynomial.zerofier_domain(dom)
    assert all(z.evaluate(d).is_zero() for d in dom) and z.degree() == 5
    x0, x1, x2 = dom[:3]
    line = Polynomial([fe(3), fe(11)])
    assert test_colinearity([(x0, line.evaluate(x0)), (x1, line.evaluate(x1)), (x2, line.evaluate(x2))])


def test_merkle_host_helpers():
    leafs = [hashlib.blake2b(bytes([i])).digest() for i in range(8)]
    root = Merkle.commit_(leafs)
    for i in range(8):
        path = Merkle.open_(i, leafs)
        assert Merkle.verify_(root, i, path, leafs[i])
        assert not Merkle.verify_(root, i ^ 1, path, leafs[i])
    g = load_golden("merkle.json")
    rec = [r for r in g["open"] if r["n"] == 4][0]
    vals = [fe(v) for v in synth.synth_ints(rec["seed"], rec["n"])]
    root4 = [r for r in g["commit"] if r.get("n") == 4][0]["root"]
    assert Merkle.verify(bytes.fromhex(root4), rec["index"], [bytes.fromhex(h) for h in rec["path"]], vals[rec["index"]])


def test_fri_index_sampling_and_rounds():
    g = load_golden("fri.json")
    for n, ef, s, rounds in g["num_rounds"]:
        assert Fri(field.generator(), field.primitive_nth_root(n), n, ef, s).num_rounds() == rounds
    f0 = Fri(field.generator(), field.primitive_nth_root(256), 256, 4, 17)
    for rec in g["sample_indices"]:
        assert f0.sample_indices(bytes.fromhex(rec["seed_hex"]), rec["size"], rec["reduced_size"], rec["number"]) == rec["out"]


def test_mul_trailing_zero_operands_do_not_recurse():
    """both lists long (>= 32 entries) but the product's degree < 8 -- fast_multiply hands such products back to
    `lhs * rhs` (ntt.py:44-45), so __mul__ has to decide on degrees.  The reference returns len(a) + len(b) - 1 coefficients."""
    a = Polynomial([fe(1), fe(1)] + [fe(0)] * 40)
    b = Polynomial([fe(1), fe(1), fe(1)] + [fe(0)] * 40)
    prod = a * b
    assert len(prod.coefficients) == 42 + 43 - 1 == 84
    assert [c.value for c in prod.coefficients[:5]] == [1, 2, 2, 1, 0] and all(c.value == 0 for c in prod.coefficients[4:])
    from ntt import fast_multiply
    r64 = field.primitive_nth_root(64)
    assert [c.value for c in fast_multiply(a, b, r64, 64).coefficients] == [1, 2, 2, 1] + [0] * 80     # ntt.py:44-45: `lhs * rhs` of the untrimmed operands (checked against the reference)
    # all-zero long operands
    z = Polynomial([fe(0)] * 40)
    assert (z * a).coefficients == [fe(0)] * (40 + 42 - 1)


def test_other_fields_are_refused_not_misreduced():
    """the device path is hard-wired to p = 1 + 407*2^119; another modulus must fail loudly before any kernel runs."""
    import pytest
    from ntt import ntt, fast_coset_evaluate, _FIELD_MSG
    f97 = Field(97)
    root = FieldElement(22, f97)            # 22 has order 4 mod 97? checked below
    assert (root ^ 4).value == 1 and (root ^ 2).value != 1
    vals = [FieldElement(v, f97) for v in (1, 2, 3, 4)]
    with pytest.raises(AssertionError, match="p = 1 \\+ 407"):
        ntt(root, vals)
    with pytest.raises(AssertionError, match="p = 1 \\+ 407"):
        fast_coset_evaluate(Polynomial(vals[:2]), FieldElement(3, f97), root, 4)


def test_byte_sampling_matches_the_shift_xor_loop():
    """Field.sample (algebra.py:123-127) and Fri.sample_index (fri.py:30-34) fold bytes with acc = (acc << 8) ^ b; the host
    modules take the big-endian integer instead -- same value for bytes, and the loop itself for anything else."""
    import random
    from fri import Fri
    rng = random.Random(5)

    def loop(byte_array, mod):
        acc = 0
        for b in byte_array:
            acc = (acc << 8) ^ int(b)
        return acc % mod
    field = Field.main()
    for length in (0, 1, 16, 17, 32, 64, 100):
        for _ in range(20):
            data = bytes(rng.randrange(256) for _ in range(length))
            assert field.sample(data).value == loop(data, field.p)
            assert field.sample(bytearray(data)).value == loop(data, field.p)
            assert field.sample(list(data)).value == loop(data, field.p)
            for size in (1, 2, 255, 256, 4096, (1 << 22) - 1):
                assert Fri.sample_index(data, size) == loop(data, size)
    assert Fri.sample_index([1, 2, 300], 1000) == loop([1, 2, 300], 1000)      # not byte values: the reference's loop verbatim
    assert field.sample([7, 1000]).value == loop([7, 1000], field.p)
    # an int is not a byte array: the reference's loop raises TypeError (iterating an int); bytes(n) -- n zero bytes, value 0 --
    # must not be mistaken for it
    for bad in (5, 0):
        with pytest.raises(TypeError):
            field.sample(bad)
        with pytest.raises(TypeError):
            Fri.sample_index(bad, 16)


def test_colinearity_shortcut_matches_interpolation():
    """test_colinearity (univariate.py:159-163) by the cross product for three points with distinct abscissas must answer what the
    reference's interpolate-and-take-the-degree answers: lines of non-zero slope only."""
    import random
    from univariate import test_colinearity as colinear
    rng = random.Random(9)
    fe = lambda v: FieldElement(v % field.p, field)

    def by_interpolation(points):
        return Polynomial.interpolate_domain([p[0] for p in points], [p[1] for p in points]).degree() == 1
    cases = []
    for _ in range(40):
        xs = [fe(rng.randrange(field.p)) for _ in range(3)]
        a, b = fe(rng.randrange(field.p)), fe(rng.randrange(field.p))
        cases.append([(x, a * x + b) for x in xs])                                  # a line
        cases.append([(x, b) for x in xs])                                          # constant: degree 0
        cases.append([(x, fe(0)) for x in xs])                                      # zero: degree -1
        cases.append([(x, fe(rng.randrange(field.p))) for x in xs])                 # generic: degree 2
        cases.append([(xs[0], a * xs[0] + b), (xs[1], a * xs[1] + b), (xs[2], a * xs[2] + b + fe(1))])
    cases.append([(fe(1), fe(2)), (fe(2), fe(4)), (fe(3), fe(6))])
    cases.append([(fe(0), fe(0)), (fe(1), fe(0)), (fe(2), fe(1))])
    for pts in cases:
        assert colinear(pts) == by_interpolation(pts)
    two = [(fe(1), fe(5)), (fe(2), fe(7))]                                          # other arities take the reference's route
    assert colinear(two) == by_interpolation(two)
    four = [(fe(i), fe(3 * i + 1)) for i in range(4)]
    assert colinear(four) == by_interpolation(four) is True


def test_mpolynomial_evaluate_against_term_by_term():
    """MPolynomial.evaluate (multivariate.py:75-81) with its per-polynomial term list: same value as the sum of products
    written out, also after the dictionary has changed."""
    import random
    from multivariate import MPolynomial
    rng = random.Random(13)
    fe = lambda v: FieldElement(v % field.p, field)
    for _ in range(20):
        nvars = rng.randrange(1, 5)
        d = {}
        for _ in range(rng.randrange(1, 12)):
            d[tuple(rng.randrange(0, 4) for _ in range(nvars))] = fe(rng.randrange(field.p))
        mp = MPolynomial(d)
        for _ in range(3):
            point = [fe(rng.randrange(field.p)) for _ in range(nvars)]
            want = 0
            for k, v in mp.dictionary.items():
                t = v.value
                for i, e in enumerate(k):
                    t = t * pow(point[i].value, e, field.p) % field.p
                want = (want + t) % field.p
            assert mp.evaluate(point).value == want
            mp.dictionary[tuple(rng.randrange(0, 4) for _ in range(nvars))] = fe(rng.randrange(field.p))   # mutate, evaluate again


def test_library_fiat_shamir_step_matches_pickle_and_hashlib():
    """Fri.commit's hand-over (ip.py:18-25, algebra.py:116-120) runs inside the library (sc_fri_commit_dev); its
    host-side pieces need no GPU and are pinned here: SHAKE-256 against hashlib, Field.sample against the shift-xor loop, and
    the transcript bytes against pickle.dumps / ProofStream.serialize() for lists of digests -- 0, 1, 2, ... items (the layout
    changes between them), mixed lengths, and the golden transcript challenges captured from the reference."""
    import ctypes
    import random
    from hashlib import shake_256
    import starkcore
    lib = starkcore.lib()
    rng = random.Random(11)
    for length in (0, 1, 135, 136, 137, 272, 1000, 70000):
        data = bytes(rng.randrange(256) for _ in range(length))
        for outlen in (32, 1, 136, 200, 300):
            out = ctypes.create_string_buffer(outlen)
            assert lib.sc_shake256(data, length, out, outlen) == 0
            assert out.raw == shake_256(data).digest(outlen), (length, outlen)
    for length in (0, 1, 16, 17, 32, 64):
        for _ in range(20):
            data = bytes(rng.randrange(256) for _ in range(length))
            got = (ctypes.c_uint64 * 2)()
            assert lib.sc_field_sample(data, length, got) == 0
            assert got[0] | (got[1] << 64) == field.sample(data).value
    all_ones = bytes([255]) * 32
    got = (ctypes.c_uint64 * 2)()
    lib.sc_field_sample(all_ones, 32, got)
    assert got[0] | (got[1] << 64) == int.from_bytes(all_ones, "big") % field.p

    def c_transcript(items):
        lens = (ctypes.c_uint32 * max(1, len(items)))(*[len(i) for i in items])
        need = ctypes.c_uint64(0)
        assert lib.sc_transcript_bytes(b"".join(items), lens, len(items), None, 0, ctypes.byref(need)) == 0
        buf = ctypes.create_string_buffer(need.value)
        assert lib.sc_transcript_bytes(b"".join(items), lens, len(items), buf, need.value, ctypes.byref(need)) == 0
        return buf.raw
    for count in (0, 1, 2, 3, 15, 40, 300):
        items = [bytes(rng.randrange(256) for _ in range(64)) for _ in range(count)]
        ps = ProofStream()
        for it in items:
            ps.push(it)
        assert c_transcript(items) == ps.serialize() == pickle.dumps(items), count
    mixed = [b"", b"a", bytes(range(255)), b"x" * 64]
    assert c_transcript(mixed) == pickle.dumps(mixed)
    lens = (ctypes.c_uint32 * 1)(256)
    need = ctypes.c_uint64(0)
    assert lib.sc_transcript_bytes(b"z" * 256, lens, 1, None, 0, ctypes.byref(need)) == -7      # BINBYTES, not SHORT_BINBYTES: not this layout
    # the transcripts of the reference's own Fri.prove runs (tests/golden/fri.json: the roots the reference pushed, round by
    # round): challenge and alpha from the library == from ProofStream / Field.sample
    checked = 0
    for rec in load_golden("fri.json")["prove_synth"]:
        roots = [bytes.fromhex(h) for h in rec["roots"]]
        ps = ProofStream()
        for r, root in enumerate(roots):
            ps.push(root)
            raw = c_transcript(roots[:r + 1])
            assert raw == ps.serialize()
            out = ctypes.create_string_buffer(32)
            lib.sc_shake256(raw, len(raw), out, 32)
            assert out.raw == ps.prover_fiat_shamir()
            got = (ctypes.c_uint64 * 2)()
            lib.sc_field_sample(out.raw, 32, got)
            assert got[0] | (got[1] << 64) == field.sample(ps.prover_fiat_shamir()).value
            checked += 1
    assert checked >= 10


def test_library_index_sampling_matches_hashlib_and_the_reference_loop():
    """sc_fri_prove_dev samples the query indices inside the library (fri.py:36-51, :122): its host-side pieces -- BLAKE2b-512 of
    any length against hashlib, Fri.sample_indices against this package's restatement of the reference loop (itself pinned to the
    reference's golden top_level_indices in tests/golden/fri.json) -- need no GPU."""
    import ctypes
    import random
    from hashlib import blake2b
    import starkcore
    lib = starkcore.lib()
    rng = random.Random(12)
    for length in (0, 1, 31, 32, 64, 127, 128, 129, 255, 256, 257, 1000):
        data = rng.randbytes(length)
        out = ctypes.create_string_buffer(64)
        assert lib.sc_blake2b(data, length, out) == 0
        assert out.raw == blake2b(data).digest(), length
    fri = Fri(field.generator(), field.primitive_nth_root(1 << 12), 1 << 12, 4, 8)
    for size, reduced, number in [(1 << 11, 1 << 4, 8), (1 << 21, 1 << 8, 40), (1 << 23, 1 << 8, 40), (1 << 5, 1 << 3, 8), (1 << 9, 64, 64), (1 << 62, 1 << 7, 17), (2, 2, 2), (4, 1, 1)]:
        for trial in range(3):
            seed = rng.randbytes(32)
            got = (ctypes.c_uint64 * number)()
            assert lib.sc_fri_sample_indices(seed, 32, size, reduced, number, got) == 0
            assert list(got) == fri.sample_indices(seed, size, reduced, number), (size, reduced, number)
    # the reference's assertion (more indices than the last codeword has entries), a size that is not a power of two
    got = (ctypes.c_uint64 * 9)()
    assert lib.sc_fri_sample_indices(b"s" * 32, 32, 1 << 10, 8, 9, got) == starkcore.SC_ERR_UNSUPPORTED
    assert lib.sc_fri_sample_indices(b"s" * 32, 32, 1000, 8, 4, got) == starkcore.SC_ERR_UNSUPPORTED


def test_library_challenge_with_the_root_dropped_in_last():
    """The commit loops hash the transcript in two steps (csrc/transcript.h: PendingChallenge): while the device computes a root,
    every whole SHAKE-256 rate block in front of it is absorbed; when it arrives it is dropped into the prepared bytes and the
    last block or two follow.  Same digest as hashlib over pickle.dumps (ip.py:18-25) for every number of prior items -- the
    root's place inside a 136-byte block moves by 67 bytes per item, so 0 ... 40 items cover every alignment, a root that straddles
    two blocks, the one-item layout (APPEND instead of MARK ... APPENDS) -- and for items of other lengths in front."""
    import ctypes
    import pickle
    import random
    from hashlib import shake_256
    import starkcore
    lib = starkcore.lib()
    rng = random.Random(13)
    shapes = [[64] * k for k in range(0, 41)] + [[0], [1, 255], [5, 64, 64, 17], [64] * 300]
    for lens in shapes:
        items = [rng.randbytes(n) for n in lens]
        root = rng.randbytes(64)
        for outlen in (32, 200):
            out = ctypes.create_string_buffer(outlen)
            assert lib.sc_transcript_challenge(b"".join(items), (ctypes.c_uint32 * max(1, len(items)))(*lens), len(items), root, out, outlen) == 0
            assert out.raw == shake_256(pickle.dumps(items + [root])).digest(outlen), lens
    out = ctypes.create_string_buffer(32)
    assert lib.sc_transcript_challenge(b"x" * 256, (ctypes.c_uint32 * 1)(256), 1, bytes(64), out, 32) == starkcore.SC_ERR_UNSUPPORTED


def test_inversion_chain_of_the_division_kernel_spends_143_products_on_p_minus_2():
    """csrc/field.cuh: mont_inv on the device follows an addition chain instead of square-and-multiply over the 128 exponent bits.
    The chain, restated on EXPONENTS (a squaring doubles, a product adds), must end at p - 2, with 126 squarings and 17 products."""
    p = 1 + 407 * (1 << 119)
    count = {"sq": 0, "mul": 0}

    def sqn(e, n):
        count["sq"] += n
        return e << n

    def mul(a, b):
        count["mul"] += 1
        return a + b
    x = 1
    x2 = mul(sqn(x, 1), x)
    x3 = mul(sqn(x2, 1), x)
    x6 = mul(sqn(x3, 3), x3)
    x7 = mul(sqn(x6, 1), x)
    x14 = mul(sqn(x7, 7), x7)
    x28 = mul(sqn(x14, 14), x14)
    x29 = mul(sqn(x28, 1), x)
    x58 = mul(sqn(x29, 29), x29)
    x59 = mul(sqn(x58, 1), x)
    x118 = mul(sqn(x59, 59), x59)
    y = mul(sqn(x118, 1), x)
    assert y == (1 << 119) - 1
    z = mul(y, x)
    w = mul(sqn(z, 1), z)
    w = mul(sqn(w, 3), z)
    w = mul(sqn(w, 2), z)
    w = mul(sqn(w, 1), z)
    w = sqn(w, 1)
    assert w == 406 * (1 << 119)
    assert mul(w, y) == p - 2
    assert (count["sq"], count["mul"]) == (126, 17)
    # ... and square-and-multiply, least significant bit first as mont_pow128 does it: one squaring per bit position, one product per set bit
    assert (p - 2).bit_length() + bin(p - 2).count("1") == 252
