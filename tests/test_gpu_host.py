"""The reference's own hot-path tests (code/test_ntt.py, code/test_fri.py) replayed through the host
mirror modules, i.e. through the same call signatures a user of the reference has -- plus golden proof
hashes captured from the reference, which pin Fri.prove end to end (roots, alphas, indices, paths,
pickle bytes)."""
import hashlib
import random

import pytest

from conftest import load_golden
import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    import starkcore
    assert starkcore.device_count() > 0, "no GPU visible"
    starkcore.init()


from algebra import Field, FieldElement          # noqa: E402
from univariate import Polynomial               # noqa: E402
from ntt import *                                # noqa: E402,F401,F403
from fri import Fri                              # noqa: E402
from ip import ProofStream                       # noqa: E402
from merkle import Merkle                        # noqa: E402

field = Field.main()
rng = random.Random(20240607)


def rand_fe():
    return field.sample(bytes(rng.getrandbits(8) for _ in range(17)))   # test_ntt.py:12 with a seeded stream


def sha_packed(lst):
    return hashlib.sha256(synth.pack_ints([x.value for x in lst])).hexdigest()


def test_ntt():                                   # code/test_ntt.py:6-19
    n = 1 << 8
    primitive_root = field.primitive_nth_root(n)
    coefficients = [rand_fe() for _ in range(n)]
    before = [c.value for c in coefficients]
    values = ntt(primitive_root, coefficients)
    values_again = Polynomial(coefficients).evaluate_domain([primitive_root ^ i for i in range(n)])
    assert values == values_again, "ntt does not compute correct batch-evaluation"
    assert [c.value for c in coefficients] == before and values is not coefficients
    one = [rand_fe()]
    assert ntt(primitive_root, one) is one and intt(primitive_root, one) is one    # ntt.py:5-6, :23-24
    with pytest.raises(AssertionError):
        ntt(primitive_root, coefficients[:6])
    with pytest.raises(AssertionError):
        ntt(field.primitive_nth_root(2 * n), coefficients)
    with pytest.raises(AssertionError):
        ntt(field.primitive_nth_root(n // 2), coefficients)


def test_intt():                                  # code/test_ntt.py:21-32
    n = 1 << 7
    primitive_root = field.primitive_nth_root(n)
    values = [field.sample(bytes([rng.getrandbits(8)])) for _ in range(n)]
    coeffs = ntt(primitive_root, values)
    assert intt(primitive_root, coeffs) == values, "inverse ntt is different from forward ntt"


def test_multiply_and_divide():                   # code/test_ntt.py:34-70
    n = 1 << 6
    primitive_root = field.primitive_nth_root(n)
    for trial in range(20):
        lhs = Polynomial([rand_fe() for _ in range(rng.randrange(n // 2) + 1)])
        rhs = Polynomial([rand_fe() for _ in range(rng.randrange(n // 2) + 1)])
        fast_product = fast_multiply(lhs, rhs, primitive_root, n)
        slow = lhs * rhs
        assert fast_product == slow, "fast product does not equal slow product"
        assert len(fast_product.coefficients) == lhs.degree() + rhs.degree() + 1
        quotient = fast_coset_divide(fast_product, lhs, field.generator(), primitive_root, n)
        assert quotient == rhs, "fast divide does not equal original factor"
    assert fast_multiply(Polynomial([]), lhs, primitive_root, n).coefficients == []


def test_poly_golden_through_host_api():
    g = load_golden("poly.json")

    def fes(seed, n):
        return [FieldElement(v, field) for v in synth.synth_ints(seed, n)]

    for rec in g["multiply"]:
        a = [FieldElement(int(v), field) for v in rec["lhs"]] if "lhs" in rec else fes(rec["lhs_seed"], rec["lhs_len"])
        b = [FieldElement(int(v), field) for v in rec["rhs"]] if "rhs" in rec else fes(rec["rhs_seed"], rec["rhs_len"])
        out = fast_multiply(Polynomial(a), Polynomial(b), FieldElement(int(rec["root"]), field), rec["order"])
        if "out" in rec:
            assert [str(c.value) for c in out.coefficients] == rec["out"]
        else:
            assert len(out.coefficients) == rec["out_len"] and sha_packed(out.coefficients) == rec["sha256"]
    for rec in g["coset_divide"]:
        q, d = fes(rec["q_seed"], rec["q_len"]), fes(rec["d_seed"], rec["d_len"])
        prod = Polynomial(q) * Polynomial(d)
        out = fast_coset_divide(prod, Polynomial(d), FieldElement(int(rec["offset"]), field), FieldElement(int(rec["root"]), field), rec["order"])
        assert len(out.coefficients) == rec["out_len"] and sha_packed(out.coefficients) == rec["sha256"]
    for rec in g["zerofier"]:
        out = fast_zerofier(fes(rec["seed"], rec["k"]), FieldElement(int(rec["root"]), field), rec["order"])
        assert [str(c.value) for c in out.coefficients] == rec["out"]
    for rec in g["evaluate"]:
        out = fast_evaluate(Polynomial(fes(rec["poly_seed"], rec["poly_len"])), fes(rec["dom_seed"], rec["k"]), FieldElement(int(rec["root"]), field), rec["order"])
        assert [str(c.value) for c in out] == rec["out"]
    for rec in g["interpolate"]:
        root = FieldElement(int(rec["root"]), field)
        dom = [field.primitive_nth_root(rec["omicron_order"]) ^ i for i in range(rec["k"])] if "omicron_order" in rec else fes(rec["dom_seed"], rec["k"])
        out = fast_interpolate(dom, fes(rec["val_seed"], rec["k"]), root, rec["order"])
        assert [str(c.value) for c in out.coefficients] == rec["out"]
    for rec in g["coset_evaluate"]:
        c = [FieldElement(int(v), field) for v in rec["coeffs"]] if "coeffs" in rec else fes(rec["seed"], rec["m"])
        out = fast_coset_evaluate(Polynomial(c), FieldElement(int(rec["offset"]), field), FieldElement(int(rec["generator"]), field), rec["order"])
        assert len(out) == rec["order"] and sha_packed(out) == rec["sha256"]


def test_interpolate():                           # code/test_ntt.py:72-96: 10 trials of random size below 512
    n = 1 << 9
    primitive_root = field.primitive_nth_root(n)
    for N in [1, 2, 15, 16, 37, 511] + [1 + rng.randrange(n - 1) for _ in range(10)]:
        values = [rand_fe() for _ in range(N)]
        domain = [rand_fe() for _ in range(N)]
        poly = fast_interpolate(domain, values, primitive_root, n)
        assert len(poly.coefficients) == N
        assert fast_evaluate(poly, domain, primitive_root, n)[0:N] == values


def test_tree_paths_agree_and_big_goldens():
    """the device subproduct tree (>= DEVICE_TREE_MIN_POINTS points) and the reference-shaped host recursion return the same
    lists; larger reference goldens through the host API; the device-resident DeviceDomain API"""
    import ntt as ntt_mod
    from starkcore import DeviceCodeword
    n = 128
    root = field.primitive_nth_root(n)
    for k in (16, 23, 40):
        dom = [rand_fe() for _ in range(k)]
        vals = [rand_fe() for _ in range(k)]
        pol = Polynomial([rand_fe() for _ in range(k + 5)])
        dev = (fast_zerofier(dom, root, n), fast_evaluate(pol, dom, root, n), fast_interpolate(dom, vals, root, n))
        keep = ntt_mod.DEVICE_TREE_MIN_POINTS
        ntt_mod.DEVICE_TREE_MIN_POINTS = 1 << 30
        try:
            host = (fast_zerofier(dom, root, n), fast_evaluate(pol, dom, root, n), fast_interpolate(dom, vals, root, n))
        finally:
            ntt_mod.DEVICE_TREE_MIN_POINTS = keep
        assert dev[0].coefficients == host[0].coefficients and dev[1] == host[1] and dev[2].coefficients == host[2].coefficients
    g = load_golden("poly.json")
    root4 = field.primitive_nth_root(1024)

    def fes(seed, cnt):
        return [FieldElement(v, field) for v in synth.synth_ints(seed, cnt)]

    for rec in g["tree_big"]:
        dom = fes(rec["dom_seed"], rec["k"])
        if rec["what"] == "zerofier":
            out = fast_zerofier(dom, root4, 1024).coefficients
        elif rec["what"] == "evaluate":
            out = fast_evaluate(Polynomial(fes(rec["poly_seed"], rec["poly_len"])), dom, root4, 1024)
        else:
            out = fast_interpolate(dom, fes(rec["val_seed"], rec["k"]), root4, 1024).coefficients
        assert len(out) == rec["out_len"] and sha_packed(out) == rec["sha256"], rec["what"]
    # device-resident API: nothing is marshalled per element
    k = 5000
    dd = DeviceDomain(fes(31, k))
    vals = DeviceCodeword.from_list(fes(32, k), field)
    coeffs = fast_interpolate_device(dd, vals)
    assert len(coeffs) == k and fast_evaluate_device(coeffs, dd).vec.to_bytes() == vals.vec.to_bytes()
    z = fast_zerofier_device(dd)
    assert len(z) == k + 1 and z[k] == field.one() and not any(fast_evaluate_device(z, dd).vec.to_bytes())


def test_coset_evaluate():                        # code/test_ntt.py:98-116
    n = 1 << 9
    primitive_root = field.primitive_nth_root(n)
    two = FieldElement(2, field)
    domain = [two * (primitive_root ^ i) for i in range(n)]
    for degree in (-1, 0, 200, n - 1):
        poly = Polynomial([rand_fe() for _ in range(degree + 1)])
        values_fast = fast_coset_evaluate(poly, two, primitive_root, n)
        assert len(values_fast) == n
        assert all(vf == vt for (vf, vt) in zip(values_fast, [poly.evaluate(d) for d in domain]))


def _test_fri_instance():
    degree, expansion_factor, num_colinearity_tests = 63, 4, 17
    n = (degree + 1) * expansion_factor
    omega = field.primitive_nth_root(n)
    fri = Fri(field.generator(), omega, n, expansion_factor, num_colinearity_tests)
    polynomial = Polynomial([FieldElement(i, field) for i in range(degree + 1)])
    codeword = polynomial.evaluate_domain([omega ^ i for i in range(n)])
    return fri, polynomial, codeword, omega, degree


def test_fri():                                   # code/test_fri.py:4-59 + golden proof bytes
    g = load_golden("fri.json")
    fri, polynomial, codeword, omega, degree = _test_fri_instance()
    proof_stream = ProofStream()
    top = fri.prove(codeword, proof_stream)
    ser = proof_stream.serialize()
    assert top == g["test_fri"]["top_level_indices"]
    assert len(proof_stream.objects) == g["test_fri"]["num_objects"] and len(ser) == g["test_fri"]["serialized_len"]
    assert [o.hex() for o in proof_stream.objects[:fri.num_rounds()]] == g["test_fri"]["roots"]
    assert hashlib.sha256(ser).hexdigest() == g["test_fri"]["serialized_sha256"]     # byte-identical proof
    points = []
    assert fri.verify(proof_stream, points) == True
    for (x, y) in points:
        assert polynomial.evaluate(omega ^ x) == y, "polynomial evaluates to wrong value"
    # disturb then test for failure
    proof_stream = ProofStream()
    for i in range(0, degree // 3):
        codeword[i] = field.zero()
    top2 = fri.prove(codeword, proof_stream)
    assert top2 == g["test_fri_corrupt"]["top_level_indices"]
    assert hashlib.sha256(proof_stream.serialize()).hexdigest() == g["test_fri_corrupt"]["serialized_sha256"]
    assert False == fri.verify(proof_stream, []), "proof should fail, but is accepted ..."


def test_fri_prove_synthetic_goldens():
    g = load_golden("fri.json")
    for rec in g["prove_synth"]:
        N = 1 << rec["logN"]
        om = field.primitive_nth_root(N)
        coeffs = [FieldElement(v, field) for v in synth.synth_ints(rec["coeff_seed"], N // 4)]
        cw = fast_coset_evaluate_device(Polynomial(coeffs), field.generator(), om, N)      # stays in HBM
        assert hashlib.sha256(cw.vec.to_bytes()).hexdigest() == rec["codeword_sha256"]
        fr = Fri(field.generator(), om, N, rec["expansion_factor"], rec["num_colinearity_tests"])
        assert fr.num_rounds() == rec["num_rounds"]
        ps = ProofStream()
        top = fr.prove(cw, ps)
        assert top == rec["top_level_indices"]
        assert [o.hex() for o in ps.objects[:rec["num_rounds"]]] == rec["roots"]
        ser = ps.serialize()
        assert len(ps.objects) == rec["num_objects"] and len(ser) == rec["serialized_len"]
        assert hashlib.sha256(ser).hexdigest() == rec["serialized_sha256"], rec["logN"]
        assert fr.verify(ps, []) == True
        # same proof from a plain list input
        if rec["logN"] <= 10:
            ps2 = ProofStream()
            assert fr.prove(cw.tolist(), ps2) == top and ps2.serialize() == ser


def test_fri_prove_in_one_call_commit_in_library_and_in_python_agree(monkeypatch):
    """Fri.prove has three forms: sc_fri_prove_dev (commit phase, index sampling and every opening in ONE library call, taken when
    the proof stream holds only digests), Fri.commit through sc_fri_commit_dev (trees, Fiat-Shamir step and folds in one call) with
    the query phase driven from Python, and the per-round loop with the Fiat-Shamir step in Python.  All must produce the
    reference's proof: the same golden hash, byte-identical streams also after digests pushed beforehand (FastStark pushes its
    commitments before FRI), and a stream that holds something else takes the Python loop and still verifies."""
    import starkcore as sc
    calls, one_calls = [], []
    real, real_one = Fri._commit_in_library, Fri._prove_in_library
    monkeypatch.setattr(Fri, "_commit_in_library", lambda self, *a: (calls.append(1), real(self, *a))[1])

    def counted_one_call(self, *a):
        out = real_one(self, *a)
        if out is not None:
            one_calls.append(1)
        return out
    monkeypatch.setattr(Fri, "_prove_in_library", counted_one_call)
    rec = [r for r in load_golden("fri.json")["prove_synth"] if r["logN"] == 12][0]
    N = 1 << rec["logN"]
    om = field.primitive_nth_root(N)
    poly = Polynomial([FieldElement(v, field) for v in synth.synth_ints(rec["coeff_seed"], N // 4)])
    fr = Fri(field.generator(), om, N, rec["expansion_factor"], rec["num_colinearity_tests"])

    def prove(prior, form="one call"):
        cw = fast_coset_evaluate_device(poly, field.generator(), om, N)
        ps = ProofStream()
        for o in prior:
            ps.push(o)
        with monkeypatch.context() as m:
            if form != "one call":
                m.setattr(Fri, "_prove_in_library", lambda self, *a: None)
            if form == "python":
                m.setattr(Fri, "_commit_in_library", lambda self, codeword, proof_stream, rounds: self._commit_rounds(codeword, proof_stream, rounds))
            top = fr.prove(cw, ps)
        return top, ps
    top, ps = prove([])
    assert one_calls and not calls and top == rec["top_level_indices"] and hashlib.sha256(ps.serialize()).hexdigest() == rec["serialized_sha256"]
    top1, ps1 = prove([], form="commit in library")
    assert calls and top1 == top and ps1.serialize() == ps.serialize()
    top2, ps2 = prove([], form="python")
    assert top2 == top and ps2.serialize() == ps.serialize()
    # what a verifier reads back from the one-call proof is the reference's object graph
    back = ProofStream().deserialize(ps.serialize())
    assert fr.verify(back, []) is True
    assert [type(o) for o in ps.objects] == [type(o) for o in ps2.objects] and list(ps.objects) == list(ps2.objects)
    prior = [bytes([i]) * 64 for i in range(3)] + [b"short", b""]
    n_calls, n_one = len(calls), len(one_calls)
    top3, ps3 = prove(prior)
    top3b, ps3b = prove(prior, form="commit in library")
    top4, ps4 = prove(prior, form="python")
    assert len(calls) == n_calls + 1 and len(one_calls) == n_one + 1 and top3 == top3b == top4 and ps3.serialize() == ps3b.serialize() == ps4.serialize() and top3 != top
    # the same object twice is a pickle memo hit, a list is not a digest: neither has the fixed layout -> the Python loop
    same = b"r" * 64
    for odd in ([same, same], [[1, 2, 3]], [b"x" * 300]):
        n_calls, n_one = len(calls), len(one_calls)
        top5, ps5 = prove(odd)
        assert len(calls) == n_calls and len(one_calls) == n_one
        for _ in odd:
            ps5.pull()
        assert fr.verify(ps5, []) is True


def test_fri_commit_persistent_tail_kernel_equals_the_per_round_launches():
    """csrc/fri_tail.cuh: from 2^16 elements down the commit phase is ONE persistent launch (fold, leaf hashes, tree and root of every
    remaining round inside it, the host only answering each root with the next challenge).  Same roots, same folded codewords, same
    trees (every authentication path the query phase opens) and the same proof bytes as the per-round launches (sc_set_tuning
    fri_tail 0), at every size class: one workgroup only (<= 64 leaves), four lanes per leaf (<= 2^14), one lane per leaf (2^15,
    2^16: 128 then 256 workgroups -- fewer workgroups in the round BEFORE the widest one), and a domain whose first rounds are the
    classic ones (2^18).  The last codeword comes back through the kernel's pinned block; the oracle folds one round independently."""
    import starkcore as sc
    from oracle import py_oracle as po
    for logN, s in [(5, 2), (7, 4), (9, 8), (12, 17), (15, 40), (16, 40), (17, 40), (18, 40)]:
        N = 1 << logN
        om = field.primitive_nth_root(N)
        coeffs = synth.synth_packed(6100 + logN, N // 4).tobytes()
        fr = Fri(field.generator(), om, N, 4, s)
        proofs = []
        for tail in (1, 0):
            sc.set_tuning("fri_tail", tail)
            try:
                vec = sc.DeviceVector(N)
                src = sc.DeviceVector.from_bytes(coeffs)
                sc._check(sc.lib().sc_coset_evaluate_dev(src.ptr, N // 4, sc.fe_bytes(field.generator().value), sc.fe_bytes(om.value), N, vec.ptr, None))
                ps = ProofStream()
                top = fr.prove(sc.DeviceCodeword(vec, field), ps)
                proofs.append((top, ps.serialize()))
            finally:
                sc.set_tuning("fri_tail", 1)
        assert proofs[0][0] == proofs[1][0], logN
        assert proofs[0][1] == proofs[1][1], logN
        back = ProofStream().deserialize(proofs[0][1])
        assert fr.verify(back, []) is True, logN
        # the first fold against the oracle, through the first two roots of the proof
        objects = ProofStream().deserialize(proofs[0][1]).objects
        cw0 = po.C.coset_evaluate(coeffs, N // 4, po.GENERATOR, om.value, N)
        assert objects[0] == po.C.merkle_commit(cw0, N)
        if fr.num_rounds() > 1:
            ps = ProofStream()
            ps.push(objects[0])
            alpha = field.sample(ps.prover_fiat_shamir())
            cw1 = po.C.fold(cw0, N, alpha.value, po.GENERATOR, om.value)
            assert objects[1] == po.C.merkle_commit(cw1, N // 2), logN


def test_fri_tail_kernel_whose_challenge_never_comes_gives_up_and_the_rounds_finish_the_classic_way():
    """Nothing in csrc/fri_tail.cuh may spin for ever: a wait that is not answered gives up, raises the abort flag, every workgroup
    leaves, and the host finishes the commit phase with the per-round launches -- the same proof.  sc_set_tuning("fri_tail_stall", k)
    makes the host withhold the challenge after the tail kernel's k-th root (and shortens the kernel's patience to milliseconds)."""
    import starkcore as sc
    rec = [r for r in load_golden("fri.json")["prove_synth"] if r["logN"] == 12][0]
    N = 1 << rec["logN"]
    om = field.primitive_nth_root(N)
    poly = Polynomial([FieldElement(v, field) for v in synth.synth_ints(rec["coeff_seed"], N // 4)])
    fr = Fri(field.generator(), om, N, rec["expansion_factor"], rec["num_colinearity_tests"])

    def prove():
        ps = ProofStream()
        top = fr.prove(fast_coset_evaluate_device(poly, field.generator(), om, N), ps)
        return top, hashlib.sha256(ps.serialize()).hexdigest()
    import ctypes

    def stats():
        out = (ctypes.c_uint64 * 2)()
        sc._check(sc.lib().sc_fri_tail_stats(out))
        return out[0], out[1]
    want = (rec["top_level_indices"], rec["serialized_sha256"])
    launches, fallbacks = stats()
    assert prove() == want and stats() == (launches + 1, fallbacks)
    try:
        for k in (0, 2):
            sc.set_tuning("fri_tail_stall", k)
            before = stats()
            assert prove() == want, k
            assert stats() == (before[0] + 1, before[1] + 1), k     # the kernel was launched, gave up, and the classic rounds finished the proof
    finally:
        sc.set_tuning("fri_tail_stall", -1)
    before = stats()
    assert prove() == want and stats() == (before[0] + 1, before[1])   # ... and the next proof takes the persistent kernel again


def test_fri_commit_with_a_proof_stream_subclass_and_a_long_transcript(monkeypatch):
    """(1) a ProofStream SUBCLASS that derives its challenges differently (the reference's SignatureProofStream prefixes
    the document, code/rpsss.py) must never take the library's commit loop, which hashes pickle(objects) itself -- prover and
    verifier then disagree on every alpha and the proof is rejected without an error; (2) a transcript of many short digests that
    passes a byte-count test but not the library's (3 bytes of pickle opcodes per item) must fall back to the Python loop instead
    of raising."""
    from hashlib import shake_256
    calls = []
    real, real_one = Fri._commit_in_library, Fri._prove_in_library
    monkeypatch.setattr(Fri, "_commit_in_library", lambda self, *a: (calls.append(1), real(self, *a))[1])

    def counted_one_call(self, *a):                                   # (the one-call form hashes the transcript in the library as well)
        out = real_one(self, *a)
        if out is not None:
            calls.append(1)
        return out
    monkeypatch.setattr(Fri, "_prove_in_library", counted_one_call)

    class PrefixedProofStream(ProofStream):
        def __init__(self, document):
            ProofStream.__init__(self)
            self.prefix = shake_256(document).digest(32)

        def prover_fiat_shamir(self, num_bytes=32):
            return shake_256(self.prefix + self.serialize()).digest(num_bytes)

        def verifier_fiat_shamir(self, num_bytes=32):
            import pickle
            return shake_256(self.prefix + pickle.dumps(self.objects[:self.read_index])).digest(num_bytes)

    N = 1 << 10
    om = field.primitive_nth_root(N)
    poly = Polynomial([FieldElement(v, field) for v in synth.synth_ints(77, N // 4)])
    fr = Fri(field.generator(), om, N, 4, 10)
    cw = fast_coset_evaluate_device(poly, field.generator(), om, N)
    ps = PrefixedProofStream(b"a document")
    fr.prove(cw, ps)
    assert not calls                                                  # the subclass went through proof_stream.prover_fiat_shamir()
    assert fr.verify(ps, []) is True
    wrong = PrefixedProofStream(b"another document")
    wrong.objects = ps.objects
    assert fr.verify(wrong, []) is False                              # the prefix really is part of every challenge
    # a plain stream still takes the library loop
    plain = ProofStream()
    fr.prove(fast_coset_evaluate_device(poly, field.generator(), om, N), plain)
    assert calls and fr.verify(plain, []) is True
    # 990 distinct one-byte "digests": few bytes, many items
    n_calls = len(calls)
    prior = [bytes([i % 256, i // 256]) for i in range(990)]
    long_stream = ProofStream()
    for o in prior:
        long_stream.push(o)
    fr.prove(fast_coset_evaluate_device(poly, field.generator(), om, N), long_stream)      # no exception: Python loop (or the library, if it takes it)
    for _ in prior:
        long_stream.pull()
    assert fr.verify(long_stream, []) is True
    del n_calls


def test_a_kept_proof_stream_does_not_keep_the_codewords(monkeypatch):
    """The proof stream of Fri.prove describes the device's answers (proof_objects); somebody who keeps the stream must not keep the
    codewords and their Merkle trees alive with it (the object path never did): the segments refer to the codewords weakly, and the
    stream still serializes to the same bytes, materialises and verifies after the codewords are gone."""
    import gc
    import weakref
    import starkcore as sc
    rec = [r for r in load_golden("fri.json")["prove_synth"] if r["logN"] == 12][0]
    N = 1 << rec["logN"]
    om = field.primitive_nth_root(N)
    poly = Polynomial([FieldElement(v, field) for v in synth.synth_ints(rec["coeff_seed"], N // 4)])
    fr = Fri(field.generator(), om, N, rec["expansion_factor"], rec["num_colinearity_tests"])
    cw = fast_coset_evaluate_device(poly, field.generator(), om, N)
    seen = [weakref.ref(cw)]
    real = sc.DeviceCodeword.__init__                      # every folded codeword the prover makes (whichever form of Fri.prove runs)
    monkeypatch.setattr(sc.DeviceCodeword, "__init__", lambda self, *a, **k: (real(self, *a, **k), seen.append(weakref.ref(self)))[0])
    ps = ProofStream()
    top = fr.prove(cw, ps)
    # (the one-call prover makes no Python object for a folded codeword at all: the library frees them before it returns)
    assert top == rec["top_level_indices"] and len(seen) in (1, fr.num_rounds())
    before = ps.serialize()
    assert hashlib.sha256(before).hexdigest() == rec["serialized_sha256"]
    del cw
    gc.collect()
    assert all(ref() is None for ref in seen), "the proof stream keeps codewords (and their trees) alive"
    assert ps.serialize() == before                        # described from packed answers, not from the codewords
    assert fr.verify(ps, []) is True                       # materialised without them: objects from the stream's own caches
    import pickle
    assert pickle.dumps(list(ps.objects)) == before


def test_a_detached_proof_stream_holds_plain_objects_and_no_pinned_memory():
    """The described segments of a one-call Fri.prove are views of pinned host memory (starkcore.HostBuffer).  objects.detach()
    swaps them for the reference's objects: the buffer's memory goes back to the pool (seen through the weak reference numpy's base
    object allows), and the stream serializes to the same bytes and verifies."""
    import gc
    import weakref
    import numpy as np
    import proof_objects
    rec = [r for r in load_golden("fri.json")["prove_synth"] if r["logN"] == 12][0]
    N = 1 << rec["logN"]
    om = field.primitive_nth_root(N)
    poly = Polynomial([FieldElement(v, field) for v in synth.synth_ints(rec["coeff_seed"], N // 4)])
    fr = Fri(field.generator(), om, N, rec["expansion_factor"], rec["num_colinearity_tests"])
    ps = ProofStream()
    fr.prove(fast_coset_evaluate_device(poly, field.generator(), om, N), ps)
    before = ps.serialize()
    assert hashlib.sha256(before).hexdigest() == rec["serialized_sha256"]
    lazy = ps.objects
    assert isinstance(lazy, proof_objects.LazyProofObjects)
    pinned = [weakref.ref(seg.elems.base) for seg in lazy._segments if isinstance(getattr(seg, "elems", None), np.ndarray) and seg.elems.base is not None]
    assert lazy.detach() is lazy
    assert all(isinstance(seg, proof_objects._Real) for seg in lazy._segments)
    gc.collect()
    assert all(ref() is None for ref in pinned), "a detached stream still refers to the pinned answers"
    assert ps.serialize() == before and fr.verify(ps, []) is True


def test_merkle_through_host_api():
    g = load_golden("merkle.json")
    for rec in g["commit"]:
        vals = [int(v) for v in rec["values"]] if "values" in rec else synth.synth_ints(rec["seed"], rec["n"])
        els = [FieldElement(v, field) for v in vals]
        assert Merkle.commit(els).hex() == rec["root"]
    for rec in g["open"]:
        els = [FieldElement(v, field) for v in synth.synth_ints(rec["seed"], rec["n"])]
        path = Merkle.open(rec["index"], els)
        assert [d.hex() for d in path] == rec["path"]
        assert Merkle.verify(Merkle.commit(els), rec["index"], path, els[rec["index"]])
        assert not Merkle.verify(Merkle.commit(els), rec["index"], path, els[rec["index"]] + field.one())
    with pytest.raises(AssertionError):
        Merkle.commit([field.one()] * 3)


def test_evaluate_symbolic_value_domain_matches_schoolbook():
    """MPolynomial.evaluate_symbolic through the value domain on the GPU (NTTs + mpoly_eval_kernel + inverse NTT) gives the
    polynomial the reference's sums of schoolbook products give (multivariate.py:83-90)."""
    from multivariate import MPolynomial

    def trim(p):
        return [c.value for c in p.coefficients[:p.degree() + 1]]

    for nvars, nterms, maxe, plen in [(5, 40, 3, 40), (3, 7, 5, 33), (2, 3, 9, 20), (6, 60, 2, 70)]:
        d = {}
        for _ in range(nterms):
            k = tuple(rng.randrange(maxe + 1) for _ in range(nvars))
            d[k] = rand_fe()
        d[(0,) * nvars] = rand_fe()                       # constant term
        d[tuple([1] + [0] * (nvars - 1))] = field.zero()  # a zero coefficient
        mp = MPolynomial(d)
        point = [Polynomial([rand_fe() for _ in range(1 + rng.randrange(plen))]) for _ in range(nvars)]
        point[-1] = Polynomial([field.zero(), field.one()])             # X itself, as fast_stark.py:108 passes it
        if nvars == 6:
            point[2] = Polynomial([])                                   # a zero polynomial kills the terms that use it
        keep = MPolynomial.VALUE_DOMAIN_MIN_DEGREE
        try:
            MPolynomial.VALUE_DOMAIN_MIN_DEGREE = 0
            dev = mp.evaluate_symbolic(point)
            MPolynomial.VALUE_DOMAIN_MIN_DEGREE = 1 << 40
            host = mp.evaluate_symbolic(point)
        finally:
            MPolynomial.VALUE_DOMAIN_MIN_DEGREE = keep
        assert trim(dev) == trim(host), (nvars, nterms)
