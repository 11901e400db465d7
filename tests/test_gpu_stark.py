"""FastStark (the caller of the hot path) end to end on the GPU: the reference's test (code/test_fast_stark.py:9-65)
with a seeded os.urandom, plus golden proof hashes captured from the reference with the same seeds -- the whole
prover (interpolation, LDEs, coset divisions, Merkle commits, FRI, openings, pickle bytes) must be byte-identical."""
import hashlib
import random

import pytest

from conftest import load_golden
from workload_rescue_prime import RescuePrime

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    import starkcore
    assert starkcore.device_count() > 0, "no GPU visible"
    starkcore.init()


import fast_stark                                  # noqa: E402
from fast_stark import FastStark                   # noqa: E402
from algebra import Field, FieldElement            # noqa: E402
from ip import ProofStream                         # noqa: E402


def _seed_urandom(seed):
    rng = random.Random(seed)
    fast_stark.os.urandom = lambda k: bytes(rng.getrandbits(8) for _ in range(k))
    return rng


@pytest.mark.parametrize("device_min", [32, 10 ** 9])
def test_fast_stark_seeded_golden_proofs(device_min, monkeypatch):
    """device_min = 32: every polynomial of the prover lives in HBM (DevicePolynomial pipeline: interpolation, exact boundary
    quotients, value-domain AIR substitution, transition quotients, LDEs, device combination); 10^9: the reference's host-list
    data flow with the GPU behind each fast_* call.  Both must reproduce the reference's proofs byte for byte."""
    monkeypatch.setattr(FastStark, "DEVICE_MIN", device_min)
    g = load_golden("fast_stark.json")
    field = Field.main()
    rp = RescuePrime()
    for rec in g["runs"]:
        _seed_urandom(rec["urandom_seed"])
        input_element = FieldElement(int(rec["input"]), field)
        output_element = rp.hash(input_element)
        assert str(output_element.value) == rec["output"]
        stark = FastStark(field, rec["expansion_factor"], rec["num_colinearity_checks"], rec["security_level"], rp.m, rp.N + 1)
        assert (stark.omicron_domain_length, stark.fri_domain_length) == (rec["omicron_domain_length"], rec["fri_domain_length"])
        transition_zerofier, transition_zerofier_codeword, transition_zerofier_root = stark.preprocess()
        assert transition_zerofier_root.hex() == rec["zerofier_root"]
        trace = rp.trace(input_element)
        air = rp.transition_constraints(stark.omicron)
        boundary = rp.boundary_constraints(output_element)
        proof = stark.prove(trace, air, boundary, transition_zerofier, transition_zerofier_codeword)
        ps = ProofStream().deserialize(proof)
        assert [o.hex() for o in ps.objects[:rp.m + 1]] == rec["first_roots"]
        assert len(ps.objects) == rec["num_objects"]
        assert len(proof) == rec["proof_len"]
        assert hashlib.sha256(proof).hexdigest() == rec["proof_sha256"]            # byte-identical to the reference
        assert stark.verify(proof, air, boundary, transition_zerofier_root) == rec["verifies"] == True
        assert stark.verify(proof, air, rp.boundary_constraints(output_element + field.one()), transition_zerofier_root) == rec["false_claim_verifies"] == False


@pytest.mark.parametrize("device_min", [32, 10 ** 9])
def test_synthetic_air_reference_golden_proofs(device_min, monkeypatch):
    """The workload bench.py TIMES for BASELINE configs[4] (workloads.synthetic_stark_instance: 2-register AIR (a, b) -> (b, a*a + b),
    T = 2^(log_fri - 4) - 4 s rows, expansion factor 4, s colinearity checks) as the REFERENCE's FastStark.prove proved it with the
    same seeded os.urandom (tests/golden/fast_stark_synth.json, make_golden.py --stark-synth; code/fast_stark.py:76-178): FRI domains
    2^10 ... 2^17 (2.2 hours of reference time for the last), i.e. the multi-pass LDE plans, the progression interpolation, the value-domain transition quotients and the
    library commit loop at sizes where they are the code that runs -- byte for byte, from host rows and from device-resident columns."""
    import workloads
    import synth
    monkeypatch.setattr(FastStark, "DEVICE_MIN", device_min)
    genuine = fast_stark.os.urandom
    try:
        for rec in load_golden("fast_stark_synth.json")["runs"]:
            log_fri, s = rec["log_fri"], rec["num_colinearity_checks"]
            if device_min > 32 and log_fri > 14:
                continue                                   # (the host-list data flow at 2^16: minutes of Python lists, nothing new)
            field, T, packed, air, boundary = workloads.synthetic_stark_instance(log_fri, s)
            assert T == rec["original_trace_length"]
            stark = FastStark(field, rec["expansion_factor"], s, rec["security_level"], 2, T)
            assert (stark.omicron_domain_length, stark.fri_domain_length) == (rec["omicron_domain_length"], rec["fri_domain_length"])
            for resident in ([False, True] if device_min == 32 else [False]):
                _seed_urandom(rec["urandom_seed"])
                tz, tz_codeword, tz_root = stark.preprocess(device_resident=True) if resident else stark.preprocess()
                assert tz_root.hex() == rec["zerofier_root"]
                if resident:
                    trace = fast_stark.DeviceTrace.from_packed(packed, field)
                else:
                    trace = [[FieldElement(a, field), FieldElement(b, field)] for a, b in zip(*synth.synthetic_air_columns(T))]
                proof = stark.prove(trace, air, boundary, tz, tz_codeword)
                ps = ProofStream().deserialize(proof)
                assert [o.hex() for o in ps.objects[:3]] == rec["first_roots"], (log_fri, resident)
                assert (len(ps.objects), len(proof)) == (rec["num_objects"], rec["proof_len"])
                assert hashlib.sha256(proof).hexdigest() == rec["proof_sha256"], (log_fri, resident)      # byte-identical to the reference
            if log_fri <= 12:
                assert stark.verify(proof, air, boundary, tz_root) is True
                wrong = boundary[:2] + [(T - 1, 1, boundary[2][2] + FieldElement(1, field))]
                assert stark.verify(proof, air, wrong, tz_root) is False
    finally:
        fast_stark.os.urandom = genuine


def test_trace_at_omicron_x_is_read_off_the_trace_codeword_not_scaled_and_transformed(monkeypatch):
    """fast_stark.py:105-106 puts trace(omicron X) in the point.  On the coset g <omicron> -- the one the value-domain transition
    quotients use unless the degree bound lets it shrink -- that polynomial's codeword is trace(X)'s, one place on: the prover
    neither scales the coefficients nor transforms them (sc_mpoly_eval_rot_dev reads the variable off its source), and the proof
    is, byte for byte, the one the scale-and-transform way gives.  (A randomized trace of 296 rows: 2 * 295 > 512, so the coset
    keeps the omicron domain's 1024 points; bench.py's sizes, whose randomized trace is a power of two, halve it -- there the
    odd points of the omicron coset are needed, and the variable is transformed as before.)"""
    import synth
    import ntt as ntt_mod
    import starkcore as sc
    from multivariate import MPolynomial
    field, s, T = Field.main(), 40, 136
    col_a, col_b = synth.synthetic_air_columns(T)
    v = MPolynomial.variables(5, field)
    air = [v[3] - v[2], v[4] - v[1] * v[1] - v[2]]
    boundary = [(0, 0, FieldElement(col_a[0], field)), (0, 1, FieldElement(col_b[0], field)), (T - 1, 1, FieldElement(col_b[T - 1], field))]
    packed = [synth.pack_ints(col_a), synth.pack_ints(col_b)]
    stark = FastStark(field, 4, s, 2 * s, 2, T)
    assert stark.omicron_domain_length == 1024
    lib = sc.lib()
    real_scale, real_eval = ntt_mod.DevicePolynomial.scale, lib.sc_coset_evaluate_dev
    scaled, evaluated, counts = [], [], []

    class Spy:                                                                # notes the order of every coset transform
        def __getattr__(self, name):
            fn = getattr(lib, name)
            if name != "sc_coset_evaluate_dev":
                return fn
            return lambda *a: (evaluated.append(int(a[4])), real_eval(*a))[1]
    monkeypatch.setattr(ntt_mod.DevicePolynomial, "scale", lambda self, factor: (scaled.append(factor.value), real_scale(self, factor))[1])
    genuine = fast_stark.os.urandom
    proofs = []
    try:
        tz, tz_codeword, tz_root = stark.preprocess(device_resident=True)
        for turned in (True, False):
            if not turned:
                monkeypatch.setattr(ntt_mod.DevicePolynomial, "scaled_later", lambda self, factor: ntt_mod.DevicePolynomial.scale(self, factor))
            _seed_urandom(4242)
            del scaled[:], evaluated[:]
            stark._zerofier_values.clear()                                    # (kept between proofs: both runs evaluate them once)
            monkeypatch.setattr(sc, "_lib", Spy())
            try:
                proofs.append(stark.prove(fast_stark.DeviceTrace.from_packed(packed, field), air, boundary, tz, tz_codeword))
            finally:
                monkeypatch.setattr(sc, "_lib", lib)
            counts.append((scaled.count(stark.omicron.value), evaluated.count(stark.omicron_domain_length)))
    finally:
        fast_stark.os.urandom = genuine
    assert proofs[0] == proofs[1]
    assert stark.verify(proofs[0], air, boundary, tz_root) is True
    # b' = b(omicron X) belongs to the quadratic constraint, whose coset is the whole omicron domain: read off b's codeword -- one
    # scaling and one transform of that order fewer.  a' belongs to the linear constraint, whose coset is half as large: the odd
    # points of the omicron coset -- a itself, transformed with the offset g * omicron (no scaled coefficient vector either).
    assert counts[1] == (2, counts[0][1] + 1) and counts[0][0] == 0, counts


@pytest.mark.parametrize("device_min", [32, 10 ** 9])
def test_fast_stark(device_min, monkeypatch):      # code/test_fast_stark.py:9-65 with its 20 chained trials (:18), seeded
    monkeypatch.setattr(FastStark, "DEVICE_MIN", device_min)
    field = Field.main()
    rng = _seed_urandom(2024)
    expansion_factor, num_colinearity_checks, security_level = 4, 2, 2
    rp = RescuePrime()
    output_element = field.sample(bytes(b'0xdeadbeef'))
    for trial in range(20):
        input_element = output_element
        output_element = rp.hash(input_element)
        num_cycles, state_width = rp.N + 1, rp.m
        stark = FastStark(field, expansion_factor, num_colinearity_checks, security_level, state_width, num_cycles)
        transition_zerofier, transition_zerofier_codeword, transition_zerofier_root = stark.preprocess()
        trace = rp.trace(input_element)
        air = rp.transition_constraints(stark.omicron)
        boundary = rp.boundary_constraints(output_element)
        proof = stark.prove(trace, air, boundary, transition_zerofier, transition_zerofier_codeword)
        assert stark.verify(proof, air, boundary, transition_zerofier_root) == True, "valid stark proof fails to verify"
        # false claim
        boundary_ = rp.boundary_constraints(output_element + field.one())
        assert stark.verify(proof, air, boundary_, transition_zerofier_root) == False, "invalid stark proof verifies"
        # false witness: perturb one trace cell that is NOT boundary-constrained (the reference's own test can hit
        # (0, reg 1) / (27, reg 0) and then fails in the prover's exact division -- SURVEY.md section 4)
        while True:
            cycle = rng.randrange(len(trace))
            register = rng.randrange(state_width)
            if (cycle, register) not in [(c, r) for c, r, v in boundary]:
                break
        trace[cycle][register] = trace[cycle][register] + field.sample(bytes(rng.getrandbits(8) for _ in range(17)))
        proof = stark.prove(trace, air, boundary, transition_zerofier, transition_zerofier_codeword)
        assert stark.verify(proof, air, boundary, transition_zerofier_root) == False, "STARK produced from false witness verifies :("
    # a perturbed boundary cell makes the boundary quotient inexact: the reference raises from Polynomial.__truediv__
    trace = rp.trace(input_element)
    trace[0][1] = trace[0][1] + field.one()
    with pytest.raises(AssertionError):
        stark.prove(trace, air, boundary, transition_zerofier, transition_zerofier_codeword)


def test_verifier_rejects_tampered_proofs():
    """Every rejection branch of FastStark.verify / Fri.verify that a single changed proof object reaches: an opened leaf, a digest of
    an authentication path, a colinear triple of the low-degree test, the last codeword."""
    import pickle
    field = Field.main()
    _seed_urandom(77)
    rp = RescuePrime()
    input_element = field.sample(b"tamper")
    output_element = rp.hash(input_element)
    stark = FastStark(field, 4, 2, 2, rp.m, rp.N + 1)
    tz, tz_codeword, tz_root = stark.preprocess()
    air, boundary = rp.transition_constraints(stark.omicron), rp.boundary_constraints(output_element)
    proof = stark.prove(rp.trace(input_element), air, boundary, tz, tz_codeword)
    assert stark.verify(proof, air, boundary, tz_root) == True
    objects = pickle.loads(proof)

    def verdict(changed):
        try:
            return stark.verify(pickle.dumps(changed), air, boundary, tz_root)
        except AssertionError:            # a changed transcript may also trip an assertion on the way (e.g. "divide by zero")
            return False

    def flipped(digest):
        return bytes([digest[0] ^ 1]) + digest[1:]

    # the last two objects: the leaf and the path of the last opening of the transition zerofier codeword
    leaf_at, path_at = len(objects) - 2, len(objects) - 1
    assert isinstance(objects[leaf_at], FieldElement) and isinstance(objects[path_at], list)
    changed = list(objects); changed[leaf_at] = objects[leaf_at] + field.one()
    assert verdict(changed) == False, "changed leaf accepted"
    changed = list(objects); changed[path_at] = [flipped(objects[path_at][0])] + objects[path_at][1:]
    assert verdict(changed) == False, "changed authentication path accepted"
    # the low-degree test's part of the stream starts after the registers' roots and the randomizer root
    first_fri = rp.m + 1
    rounds = stark.fri.num_rounds()
    last_codeword_at = first_fri + rounds
    assert isinstance(objects[last_codeword_at], list) and isinstance(objects[last_codeword_at][0], FieldElement)
    changed = list(objects); changed[last_codeword_at] = [objects[last_codeword_at][0] + field.one()] + objects[last_codeword_at][1:]
    assert verdict(changed) == False, "changed last codeword accepted"
    triple_at = last_codeword_at + 1
    assert isinstance(objects[triple_at], tuple) and len(objects[triple_at]) == 3
    a, b, c = objects[triple_at]
    changed = list(objects); changed[triple_at] = (a, b, c + field.one())
    assert verdict(changed) == False, "changed colinearity triple accepted"
    fri_path_at = triple_at + stark.fri.num_colinearity_tests
    assert isinstance(objects[fri_path_at], list) and isinstance(objects[fri_path_at][0], bytes)
    changed = list(objects); changed[fri_path_at] = [flipped(objects[fri_path_at][0])] + objects[fri_path_at][1:]
    assert verdict(changed) == False, "changed FRI authentication path accepted"
    changed = list(objects); changed[first_fri] = flipped(objects[first_fri])
    assert verdict(changed) == False, "changed FRI root accepted"


def test_device_resident_trace_reproduces_the_reference_proofs():
    """fast_stark.DeviceTrace (the trace as columns in HBM) and preprocess(device_resident=True) (the transition zerofier from
    its closed form on the progression {omicron^i}) against the REFERENCE's golden proofs: byte-identical."""
    from fast_stark import DeviceTrace
    g = load_golden("fast_stark.json")
    field = Field.main()
    rp = RescuePrime()
    for rec in g["runs"]:
        input_element = FieldElement(int(rec["input"]), field)
        output_element = rp.hash(input_element)
        stark = FastStark(field, rec["expansion_factor"], rec["num_colinearity_checks"], rec["security_level"], rp.m, rp.N + 1)
        for resident in (False, True):
            _seed_urandom(rec["urandom_seed"])
            tz, tz_codeword, tz_root = stark.preprocess(device_resident=resident)
            assert tz_root.hex() == rec["zerofier_root"]
            rows = rp.trace(input_element)
            trace = DeviceTrace.from_rows(rows, field)
            assert len(trace) == len(rows) and trace.entry(3, 1) == rows[3][1]
            air, boundary = rp.transition_constraints(stark.omicron), rp.boundary_constraints(output_element)
            proof = stark.prove(trace, air, boundary, tz, tz_codeword)
            assert hashlib.sha256(proof).hexdigest() == rec["proof_sha256"], resident
            assert stark.verify(proof, air, boundary, tz_root) == True


def test_library_draws_of_the_randomizer_polynomial(monkeypatch):
    """sc_sample_urandom_dev: the library's own getrandom draws (what os.urandom is), threaded, sampled on the device.  The bytes
    are the operating system's, so only properties can be checked: canonical residues, no repeats, a fresh draw every call; and a
    proof made with them verifies.  Field.sample itself is pinned through sc_sample_bytes_dev (seeded goldens above)."""
    import os as real_os
    import importlib
    import starkcore as sc
    from starkcore import DeviceVector
    count = (1 << 16) + 3                         # > 1 MiB of draws: several threads
    a, b = DeviceVector(count), DeviceVector(count)
    sc._check(sc.lib().sc_sample_urandom_dev(count, 17, a.ptr, None))
    sc._check(sc.lib().sc_sample_urandom_dev(count, 17, b.ptr, None))
    va, vb = sc.unpack(a.to_bytes()), sc.unpack(b.to_bytes())
    p = Field.main().p
    assert all(0 <= v < p for v in va) and len(set(va)) == count and len(set(va) & set(vb)) == 0
    assert max(va) > p // 2 and min(va) < p // 2 and sum(v >> 120 != 0 for v in va) > count // 4
    # a whole proof with the operating system's randomness: os.urandom as the interpreter provides it
    genuine = importlib.import_module("posix").urandom
    monkeypatch.setattr(real_os, "urandom", genuine)
    assert fast_stark.os_urandom_is_genuine()
    field = Field.main()
    rp = RescuePrime()
    input_element = field.sample(b"library draws")
    output_element = rp.hash(input_element)
    stark = FastStark(field, 4, 2, 2, rp.m, rp.N + 1)
    tz, tz_codeword, tz_root = stark.preprocess(device_resident=True)
    air, boundary = rp.transition_constraints(stark.omicron), rp.boundary_constraints(output_element)
    proofs = [stark.prove(fast_stark.DeviceTrace.from_rows(rp.trace(input_element), field), air, boundary, tz, tz_codeword) for _ in range(2)]
    assert proofs[0] != proofs[1]                 # zero knowledge needs fresh randomizers every time
    assert all(stark.verify(proof, air, boundary, tz_root) == True for proof in proofs)
