#!/usr/bin/env python3
"""Generate golden vectors by importing the reference implementation.

Runs ONLY in the build container (needs /root/reference/code on disk); the GPU box never
runs this.  Output: small JSON fixtures next to this file.  Only DATA is written (inputs are
regenerated from seeds, outputs stored in full for small sizes and as SHA-256 of the packed
16-byte-LE output for larger ones).  No reference source is copied.

usage:  python tests/golden/make_golden.py [--big] [--fri-big] [--host-mirror] [--stark-synth [log_fri:s:seed ...]]     (--big adds 2^18 / 2^20 NTT digests, --fri-big 2^14 and 2^16 Fri.prove runs; minutes)
"""
import hashlib
import json
import os
import random
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/code"
sys.path.insert(0, REF)
sys.path.insert(1, os.path.join(REPO, "stark-anatomy_amd"))
sys.setrecursionlimit(10000)

import algebra as ref_algebra            # noqa: E402  (reference)
import univariate as ref_univariate      # noqa: E402
import ntt as ref_ntt                    # noqa: E402
import merkle as ref_merkle              # noqa: E402
import ip as ref_ip                      # noqa: E402
import fri as ref_fri                    # noqa: E402

assert ref_ntt.__file__.startswith(REF), ref_ntt.__file__

import importlib.util                    # noqa: E402
_spec = importlib.util.spec_from_file_location("sa_synth", os.path.join(REPO, "stark-anatomy_amd", "synth.py"))
synth = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(synth)

Field, FieldElement, Polynomial = ref_algebra.Field, ref_algebra.FieldElement, ref_univariate.Polynomial
field = Field.main()


def fe(v):
    return FieldElement(v, field)


def fes(seed, n, start=0):
    return [fe(v) for v in synth.synth_ints(seed, n, start)]


def vals(lst):
    return [str(x.value) for x in lst]


def sha_packed(lst):
    return hashlib.sha256(synth.pack_ints([x.value for x in lst])).hexdigest()


def dump(name, obj):
    path = os.path.join(HERE, name)
    with open(path, "w") as f:
        json.dump(obj, f, indent=0, separators=(",", ":"))
        f.write("\n")
    print("wrote", name, os.path.getsize(path), "bytes")


def gen_field():
    out = {"p": str(field.p), "generator": str(field.generator().value)}
    out["primitive_nth_root"] = {str(k): str(field.primitive_nth_root(1 << k).value) for k in range(1, 33)}
    out["inverse"] = {str(v): str(fe(v).inverse().value) for v in [0, 1, 2, 3, 8, 407, field.p - 1, 1 << 64, (1 << 127) + 12345]}
    xs = synth.synth_ints(11, 8)
    ys = synth.synth_ints(12, 8)
    out["mul"] = [[str(a), str(b), str((fe(a) * fe(b)).value)] for a, b in zip(xs, ys)]
    out["add"] = [[str(a), str(b), str((fe(a) + fe(b)).value)] for a, b in zip(xs, ys)]
    out["sub"] = [[str(a), str(b), str((fe(a) - fe(b)).value)] for a, b in zip(xs, ys)]
    out["div"] = [[str(a), str(b), str((fe(a) / fe(b)).value)] for a, b in zip(xs, ys)]
    out["pow"] = [[str(a), str(e), str((fe(a) ^ e).value)] for a, e in zip(xs, [0, 1, 2, 3, 65537, (1 << 64) + 3, field.p - 2, field.p - 1])]
    out["sample"] = [[b.hex(), str(field.sample(b).value)] for b in [b"", b"\x01", b"0xdeadbeef", bytes(range(17)), b"\xff" * 32, b"\x01" * 32]]
    out["synth_seed1_first4"] = [str(v) for v in synth.synth_ints(1, 4)]
    dump("field.json", out)


def gen_ntt(big):
    out = {"ntt": [], "intt": [], "kat": {}}
    w8 = field.primitive_nth_root(8)
    out["kat"]["ntt_w8_1to8"] = vals(ref_ntt.ntt(w8, [fe(i) for i in range(1, 9)]))
    out["kat"]["intt_w8_1to8"] = vals(ref_ntt.intt(w8, [fe(i) for i in range(1, 9)]))
    sizes = list(range(0, 17))
    if big:
        sizes += [18, 20]
    for logn in sizes:
        n = 1 << logn
        for seed in ([1, 2] if logn <= 10 else [1]):
            t0 = time.time()
            x = fes(seed, n)
            root = field.primitive_nth_root(n) if n > 1 else field.one()
            y = ref_ntt.ntt(root, x)
            rec = {"logn": logn, "seed": seed, "root": str(root.value), "sha256": sha_packed(y)}
            if logn <= 6:
                rec["out"] = vals(y)
            else:
                rec["out_head"] = vals(y[:4])
                rec["out_tail"] = vals(y[-2:])
            out["ntt"].append(rec)
            if logn <= 14:
                z = ref_ntt.intt(root, x)
                rec2 = {"logn": logn, "seed": seed, "root": str(root.value), "sha256": sha_packed(z)}
                if logn <= 6:
                    rec2["out"] = vals(z)
                else:
                    rec2["out_head"] = vals(z[:4])
                out["intt"].append(rec2)
            print("ntt logn", logn, "seed", seed, "%.1fs" % (time.time() - t0), flush=True)
    # a non-canonical root choice: inverse root and a root of a larger order squared down
    n = 64
    x = fes(5, n)
    r = field.primitive_nth_root(n).inverse()
    out["ntt"].append({"logn": 6, "seed": 5, "root": str(r.value), "sha256": sha_packed(ref_ntt.ntt(r, x)), "out": vals(ref_ntt.ntt(r, x))})
    r = field.primitive_nth_root(n) ^ 3   # another primitive 64th root
    out["ntt"].append({"logn": 6, "seed": 5, "root": str(r.value), "sha256": sha_packed(ref_ntt.ntt(r, x)), "out": vals(ref_ntt.ntt(r, x))})
    dump("ntt_big.json" if big else "ntt.json", out)


def gen_poly():
    out = {"multiply": [], "coset_evaluate": [], "coset_divide": [], "zerofier": [], "evaluate": [], "interpolate": [], "scale": []}
    n = 64
    root = field.primitive_nth_root(n)
    rng = random.Random(1234)
    # (lhs_len, rhs_len) incl. schoolbook fallback (deg<8), zero polys, trailing zeros, order shrinking
    cases = [(1, 1), (3, 4), (4, 5), (5, 5), (9, 1), (17, 13), (32, 32), (31, 2), (0, 5), (7, 0), (20, 20), (12, 30)]
    for i, (la, lb) in enumerate(cases):
        a = fes(100 + i, la)
        b = fes(200 + i, lb)
        prod = ref_ntt.fast_multiply(Polynomial(a), Polynomial(b), root, n)
        out["multiply"].append({"lhs_seed": 100 + i, "lhs_len": la, "rhs_seed": 200 + i, "rhs_len": lb, "order": n,
                                "root": str(root.value), "out": vals(prod.coefficients)})
    # trailing zero coefficients (degree() < len-1)
    a = fes(300, 10) + [field.zero()] * 3
    b = fes(301, 6) + [field.zero()]
    prod = ref_ntt.fast_multiply(Polynomial(a), Polynomial(b), root, n)
    out["multiply"].append({"lhs": vals(a), "rhs": vals(b), "order": n, "root": str(root.value), "out": vals(prod.coefficients)})
    # larger: order 1024, degrees 300/200
    n2 = 1024
    root2 = field.primitive_nth_root(n2)
    a = fes(310, 301)
    b = fes(311, 201)
    prod = ref_ntt.fast_multiply(Polynomial(a), Polynomial(b), root2, n2)
    out["multiply"].append({"lhs_seed": 310, "lhs_len": 301, "rhs_seed": 311, "rhs_len": 201, "order": n2,
                            "root": str(root2.value), "sha256": sha_packed(prod.coefficients), "out_len": len(prod.coefficients),
                            "out_head": vals(prod.coefficients[:3])})

    # coset evaluate: offsets 2 and generator, lengths incl. empty-ish, full, with trailing zeros
    for i, (m, order, offv) in enumerate([(3, 8, field.generator().value), (1, 8, 2), (8, 8, 2), (100, 512, 2), (512, 512, field.generator().value),
                                          (37, 256, 12345678901234567890123), (1 << 10, 1 << 13, field.generator().value)]):
        coeffs = fes(400 + i, m)
        gen = field.primitive_nth_root(order)
        v = ref_ntt.fast_coset_evaluate(Polynomial(coeffs), fe(offv), gen, order)
        rec = {"seed": 400 + i, "m": m, "order": order, "offset": str(offv), "generator": str(gen.value), "sha256": sha_packed(v)}
        if order <= 8:
            rec["out"] = vals(v)
        else:
            rec["out_head"] = vals(v[:3])
        out["coset_evaluate"].append(rec)
    v = ref_ntt.fast_coset_evaluate(Polynomial([fe(1), fe(2), fe(3)]), field.generator(), field.primitive_nth_root(8), 8)
    out["coset_evaluate"].append({"coeffs": ["1", "2", "3"], "order": 8, "offset": str(field.generator().value),
                                  "generator": str(field.primitive_nth_root(8).value), "out": vals(v), "sha256": sha_packed(v)})

    # coset divide: quotient * divisor products (clean division), incl. fallback
    for i, (lq, ld, order) in enumerate([(3, 2, 64), (10, 9, 64), (30, 20, 64), (20, 30, 64), (1, 33, 64), (200, 300, 1024)]):
        q = fes(500 + i, lq)
        d = fes(600 + i, ld)
        rt = field.primitive_nth_root(order)
        prod = Polynomial(q) * Polynomial(d)
        quo = ref_ntt.fast_coset_divide(prod, Polynomial(d), field.generator(), rt, order)
        out["coset_divide"].append({"q_seed": 500 + i, "q_len": lq, "d_seed": 600 + i, "d_len": ld, "order": order, "root": str(rt.value),
                                    "offset": str(field.generator().value), "lhs_sha256": sha_packed(prod.coefficients),
                                    "out_len": len(quo.coefficients), "sha256": sha_packed(quo.coefficients), "out_head": vals(quo.coefficients[:3])})

    # zerofier / evaluate / interpolate on arbitrary domains
    n3 = 512
    root3 = field.primitive_nth_root(n3)
    for i, k in enumerate([0, 1, 2, 3, 5, 8, 27, 100]):
        dom = fes(700 + i, k)
        z = ref_ntt.fast_zerofier(dom, root3, n3)
        out["zerofier"].append({"seed": 700 + i, "k": k, "order": n3, "root": str(root3.value), "out": vals(z.coefficients)})
    for i, (deg1, k) in enumerate([(5, 0), (5, 1), (5, 7), (40, 33), (100, 64)]):
        poly = fes(800 + i, deg1)
        dom = fes(900 + i, k)
        ev = ref_ntt.fast_evaluate(Polynomial(poly), dom, root3, n3)
        out["evaluate"].append({"poly_seed": 800 + i, "poly_len": deg1, "dom_seed": 900 + i, "k": k, "order": n3, "root": str(root3.value), "out": vals(ev)})
    for i, k in enumerate([0, 1, 2, 5, 36, 64]):
        dom = fes(1000 + i, k)
        vv = fes(1100 + i, k)
        poly = ref_ntt.fast_interpolate(dom, vv, root3, n3)
        out["interpolate"].append({"dom_seed": 1000 + i, "val_seed": 1100 + i, "k": k, "order": n3, "root": str(root3.value), "out": vals(poly.coefficients)})
    # omicron-prefix domain as fast_stark.py:86-90 uses it
    om = field.primitive_nth_root(128)
    dom = [om ^ i for i in range(36)]
    vv = fes(1200, 36)
    poly = ref_ntt.fast_interpolate(dom, vv, om, 128)
    out["interpolate"].append({"omicron_order": 128, "k": 36, "val_seed": 1200, "order": 128, "root": str(om.value), "out": vals(poly.coefficients)})
    for i, (m, f) in enumerate([(0, 2), (1, 2), (9, field.generator().value), (9, 0)]):
        c = fes(1300 + i, m)
        out["scale"].append({"seed": 1300 + i, "m": m, "factor": str(f), "out": vals(Polynomial(c).scale(fe(f)).coefficients)})
    # larger trees (digest only): the device computes these level by level instead of by the reference's recursion
    out["tree_big"] = []
    n4 = 1024
    root4 = field.primitive_nth_root(n4)
    t0 = time.time()
    dom = fes(1400, 300)
    z = ref_ntt.fast_zerofier(dom, root4, n4)
    out["tree_big"].append({"what": "zerofier", "dom_seed": 1400, "k": 300, "out_len": len(z.coefficients), "sha256": sha_packed(z.coefficients)})
    dom = fes(1401, 257)
    pol = Polynomial(fes(1402, 300))
    ev = ref_ntt.fast_evaluate(pol, dom, root4, n4)
    out["tree_big"].append({"what": "evaluate", "dom_seed": 1401, "k": 257, "poly_seed": 1402, "poly_len": 300, "out_len": len(ev), "sha256": sha_packed(ev)})
    dom = fes(1403, 200)
    vv = fes(1404, 200)
    ip = ref_ntt.fast_interpolate(dom, vv, root4, n4)
    out["tree_big"].append({"what": "interpolate", "dom_seed": 1403, "k": 200, "val_seed": 1404, "out_len": len(ip.coefficients), "sha256": sha_packed(ip.coefficients)})
    print("tree_big %.1fs" % (time.time() - t0), flush=True)
    # schoolbook divide (univariate.py:80-97) incl. operands carrying trailing zeros: the LIST LENGTHS are part of the contract
    out["divmod"] = []
    for a, b in [([1], [1, 0]), ([1, 2, 3], [1, 1, 0, 0]), ([0, 0, 5, 0], [2, 0]), ([1, 2, 3, 4, 0, 0], [3, 1, 0]), ([5, 4, 3, 2, 1], [7, 1]),
                 ([1, 2], [1, 2, 3]), ([], [1]), ([0, 0], [4]), ([field.p - 1, 1 << 64, 12345], [field.p - 2, 9])]:
        q, r = Polynomial.divide(Polynomial([fe(v) for v in a]), Polynomial([fe(v) for v in b]))
        out["divmod"].append({"num": [str(v) for v in a], "den": [str(v) for v in b], "quo": vals(q.coefficients), "rem": vals(r.coefficients)})
    dump("poly.json", out)


def gen_merkle():
    out = {"commit": [], "open": []}
    for vlist in ([0], [0, 1], [0, 1, 2, 3], [field.p - 1, 10 ** 18, 10 ** 19 - 1, 10 ** 19, 10 ** 37, 10 ** 38 - 1, 10 ** 38, 1 << 64]):
        els = [fe(v) for v in vlist]
        out["commit"].append({"values": [str(v) for v in vlist], "root": ref_merkle.Merkle.commit(els).hex()})
    for logn in range(0, 13):
        n = 1 << logn
        els = fes(2000 + logn, n)
        out["commit"].append({"seed": 2000 + logn, "n": n, "root": ref_merkle.Merkle.commit(els).hex()})
    for logn, idxs in [(1, [0, 1]), (2, [0, 2, 3]), (5, [0, 13, 31]), (10, [0, 1, 511, 512, 777, 1023])]:
        n = 1 << logn
        els = fes(2000 + logn, n)
        for idx in idxs:
            path = ref_merkle.Merkle.open(idx, els)
            out["open"].append({"seed": 2000 + logn, "n": n, "index": idx, "path": [d.hex() for d in path]})
    out["leaf_digest"] = [[str(v), ref_merkle.Merkle.H(bytes(fe(v))).digest().hex()] for v in [0, 7, 10 ** 19, field.p - 1]]
    dump("merkle.json", out)


def gen_fri():
    out = {}
    # fri.py:36-51
    f0 = ref_fri.Fri(field.generator(), field.primitive_nth_root(256), 256, 4, 17)
    out["sample_indices"] = [{"seed_hex": (b"\x00" * 32).hex(), "size": 128, "reduced_size": 32, "number": 17,
                              "out": f0.sample_indices(b"\x00" * 32, 128, 32, 17)},
                             {"seed_hex": bytes(range(32)).hex(), "size": 1 << 21, "reduced_size": 256, "number": 40,
                              "out": f0.sample_indices(bytes(range(32)), 1 << 21, 256, 40)}]
    out["num_rounds"] = [[n, ef, s, ref_fri.Fri(field.generator(), field.primitive_nth_root(n), n, ef, s).num_rounds()]
                         for (n, ef, s) in [(256, 4, 17), (512, 4, 2), (4096, 4, 64), (1 << 12, 4, 40), (1 << 22, 4, 40), (1 << 24, 4, 40), (1 << 26, 4, 64)]]

    # one fold (fri.py:85) of the test_fri codeword with a fixed alpha
    degree = 63
    n = 256
    omega = field.primitive_nth_root(n)
    poly = Polynomial([fe(i) for i in range(degree + 1)])
    codeword = poly.evaluate_domain([omega ^ i for i in range(n)])
    alpha = field.sample(b"\x01" * 32)
    offset = field.generator()
    one = field.one()
    two = fe(2)
    N = n
    folded = [two.inverse() * ((one + alpha / (offset * (omega ^ i))) * codeword[i] + (one - alpha / (offset * (omega ^ i))) * codeword[N // 2 + i]) for i in range(N // 2)]
    out["fold"] = [{"kind": "test_fri_codeword", "n": n, "alpha": str(alpha.value), "offset": str(offset.value), "omega": str(omega.value),
                    "in_sha256": sha_packed(codeword), "in_head": vals(codeword[:3]), "out": vals(folded), "sha256": sha_packed(folded)}]
    for i, logn in enumerate([1, 2, 3, 6, 11]):
        N = 1 << logn
        cw = fes(3000 + i, N)
        om = field.primitive_nth_root(N)
        al = fes(3100 + i, 1)[0]
        off = field.generator() if i % 2 == 0 else fe(3)
        fo = [two.inverse() * ((one + al / (off * (om ^ j))) * cw[j] + (one - al / (off * (om ^ j))) * cw[N // 2 + j]) for j in range(N // 2)]
        rec = {"kind": "synth", "seed": 3000 + i, "n": N, "alpha": str(al.value), "offset": str(off.value), "omega": str(om.value), "sha256": sha_packed(fo)}
        if N <= 64:
            rec["out"] = vals(fo)
        out["fold"].append(rec)

    # deterministic test_fri instance (test_fri.py:4-59)
    fr = ref_fri.Fri(field.generator(), omega, n, 4, 17)
    ps = ref_ip.ProofStream()
    top = fr.prove(list(codeword), ps)
    ser = ps.serialize()
    out["test_fri"] = {"n": n, "expansion_factor": 4, "num_colinearity_tests": 17, "top_level_indices": top,
                       "num_objects": len(ps.objects), "serialized_len": len(ser), "serialized_sha256": hashlib.sha256(ser).hexdigest(),
                       "roots": [o.hex() for o in ps.objects[:fr.num_rounds()]],
                       "last_codeword_sha256": sha_packed(ps.objects[fr.num_rounds()])}
    # corrupted codeword variant (test_fri.py:53-54)
    cw2 = list(codeword)
    for i in range(0, degree // 3):
        cw2[i] = field.zero()
    ps2 = ref_ip.ProofStream()
    top2 = fr.prove(cw2, ps2)
    ser2 = ps2.serialize()
    out["test_fri_corrupt"] = {"top_level_indices": top2, "serialized_sha256": hashlib.sha256(ser2).hexdigest(), "serialized_len": len(ser2)}

    # synthetic LDE codewords: Fri(generator, omega, N, 4, s)
    # (--fri-big adds 2^14 and 2^16 codewords with the BASELINE number of colinearity checks: a minute of reference time, most of it
    # Merkle.open rebuilding the tree for each opening)
    for (logN, s, seed) in [(6, 4, 4000), (10, 10, 4001), (12, 40, 4002)] + ([(14, 40, 4003), (16, 40, 4004)] if "--fri-big" in sys.argv else []):
        N = 1 << logN
        om = field.primitive_nth_root(N)
        coeffs = fes(seed, N // 4)
        cw = ref_ntt.fast_coset_evaluate(Polynomial(coeffs), field.generator(), om, N)
        fr = ref_fri.Fri(field.generator(), om, N, 4, s)
        ps = ref_ip.ProofStream()
        t0 = time.time()
        top = fr.prove(cw, ps)
        ser = ps.serialize()
        nr = fr.num_rounds()
        out.setdefault("prove_synth", []).append({
            "logN": logN, "num_colinearity_tests": s, "expansion_factor": 4, "coeff_seed": seed, "num_rounds": nr,
            "codeword_sha256": sha_packed(cw), "top_level_indices": top, "num_objects": len(ps.objects),
            "serialized_len": len(ser), "serialized_sha256": hashlib.sha256(ser).hexdigest(),
            "roots": [o.hex() for o in ps.objects[:nr]], "last_codeword_sha256": sha_packed(ps.objects[nr])})
        print("fri prove logN", logN, "%.1fs" % (time.time() - t0), flush=True)
    dump("fri.json", out)


def gen_pickle():
    """Byte-level pins for the Fiat-Shamir transcript (ip.py:18-25)."""
    import pickle
    out = {}
    objs = [b"\x01" * 64, [fe(5), fe(field.p - 1)], (fe(1), fe(2), fe(3)), [b"a" * 64, b"b" * 64]]
    ps = ref_ip.ProofStream()
    for o in objs:
        ps.push(o)
    ser = ps.serialize()
    out["protocol_default"] = pickle.DEFAULT_PROTOCOL
    out["python"] = sys.version.split()[0]
    out["serialized_hex"] = ser.hex()
    out["prover_fiat_shamir"] = ps.prover_fiat_shamir().hex()
    ps.pull()
    ps.pull()
    out["verifier_fiat_shamir_after2"] = ps.verifier_fiat_shamir().hex()
    out["single_fe_list_len"] = len(pickle.dumps([fe(7)]))
    dump("transcript.json", out)


def gen_stark():
    """Rescue-Prime parameters (data) + a seeded FastStark run (fast_stark.py:76-178 with os.urandom patched)."""
    import fast_stark as ref_fast_stark
    import rescue_prime as ref_rp
    rp = ref_rp.RescuePrime()
    params = {"p": str(rp.p), "m": rp.m, "N": rp.N, "alpha": rp.alpha, "alphainv": str(rp.alphainv),
              "MDS": [[str(x.value) for x in row] for row in rp.MDS], "MDSinv": [[str(x.value) for x in row] for row in rp.MDSinv],
              "round_constants": [str(c.value) for c in rp.round_constants]}
    params["kat_hash"] = [[str(v), str(rp.hash(fe(v)).value)] for v in [1, 57322816861100832358702415967512842988, int(field.sample(b"0xdeadbeef").value)]]
    dump("rescue_prime_params.json", params)

    out = {"runs": []}
    for seed, s_checks in [(7, 2), (11, 3)]:
        rng = random.Random(seed)
        ref_fast_stark.os.urandom = lambda k, rng=rng: bytes(rng.getrandbits(8) for _ in range(k))
        ef, sec = 4, 2
        input_element = field.sample(b"0xdeadbeef")
        output_element = rp.hash(input_element)
        stark = ref_fast_stark.FastStark(field, ef, s_checks, sec, rp.m, rp.N + 1)
        tz, tzc, tzr = stark.preprocess()
        trace = rp.trace(input_element)
        air = rp.transition_constraints(stark.omicron)
        boundary = rp.boundary_constraints(output_element)
        t0 = time.time()
        proof = stark.prove(trace, air, boundary, tz, tzc)
        ok = stark.verify(proof, air, boundary, tzr)
        bad = stark.verify(proof, air, rp.boundary_constraints(output_element + field.one()), tzr)
        ps = ref_ip.ProofStream().deserialize(proof)
        out["runs"].append({"urandom_seed": seed, "expansion_factor": ef, "num_colinearity_checks": s_checks, "security_level": sec,
                            "input": str(input_element.value), "output": str(output_element.value),
                            "omicron_domain_length": stark.omicron_domain_length, "fri_domain_length": stark.fri_domain_length,
                            "zerofier_coeffs_sha256": sha_packed(tz.coefficients), "zerofier_root": tzr.hex(),
                            "proof_len": len(proof), "proof_sha256": hashlib.sha256(proof).hexdigest(), "num_objects": len(ps.objects),
                            "first_roots": [o.hex() for o in ps.objects[:rp.m + 1]], "verifies": ok, "false_claim_verifies": bad})
        print("fast stark seed", seed, "%.1fs" % (time.time() - t0), ok, bad, flush=True)
    dump("fast_stark.json", out)


def gen_host_mirror():
    """tests/golden/host_mirror_cases.py run on the reference's own algebra / univariate / multivariate: digests per seeded case"""
    import algebra as ref_algebra
    import univariate as ref_univariate
    import multivariate as ref_multivariate
    sys.path.insert(0, HERE)
    import host_mirror_cases
    for m in (ref_algebra, ref_univariate, ref_multivariate):
        assert m.__file__.startswith("/root/reference/"), m.__file__
    dump("host_mirror.json", host_mirror_cases.run_cases(ref_algebra, ref_univariate, ref_multivariate))


def gen_stark_synth(cases):
    """The workload bench.py times for BASELINE configs[4] -- workloads.synthetic_stark_instance: the 2-register AIR (a, b) -> (b, a*a + b),
    T = 2^(log_fri - 4) - 4 s rows, expansion factor 4, s colinearity checks, security level 2 s -- proven by the REFERENCE's
    FastStark (fast_stark.py:76-178) with a seeded os.urandom.  cases: [(log_fri, s, seed)].  Records are merged into
    fast_stark_synth.json by (log_fri, s, seed), so the long sizes can be added one at a time (2^14: minutes; 2^16: about an hour)."""
    import fast_stark as ref_fast_stark
    import multivariate as ref_multivariate
    path = os.path.join(HERE, "fast_stark_synth.json")
    for log_fri, s_checks, seed in cases:
        rng = random.Random(seed)
        ref_fast_stark.os.urandom = lambda k, rng=rng: bytes(rng.getrandbits(8) for _ in range(k))
        T = (1 << (log_fri - 4)) - 4 * s_checks
        col_a, col_b = synth.synthetic_air_columns(T)
        trace = [[fe(a), fe(b)] for a, b in zip(col_a, col_b)]
        v = ref_multivariate.MPolynomial.variables(5, field)                  # X, a, b, a', b'
        air = [v[3] - v[2], v[4] - v[1] * v[1] - v[2]]
        boundary = [(0, 0, fe(col_a[0])), (0, 1, fe(col_b[0])), (T - 1, 1, fe(col_b[T - 1]))]
        stark = ref_fast_stark.FastStark(field, 4, s_checks, 2 * s_checks, 2, T)
        assert stark.fri_domain_length == 1 << log_fri, (stark.fri_domain_length, log_fri)
        t0 = time.time()
        tz, tzc, tzr = stark.preprocess()
        t1 = time.time()
        proof = stark.prove(trace, air, boundary, tz, tzc)
        t2 = time.time()
        ok = stark.verify(proof, air, boundary, tzr)
        bad = stark.verify(proof, air, [(0, 0, fe(col_a[0])), (0, 1, fe(col_b[0])), (T - 1, 1, fe(col_b[T - 1] + 1))], tzr)
        t3 = time.time()
        ps = ref_ip.ProofStream().deserialize(proof)
        rec = {"log_fri": log_fri, "num_colinearity_checks": s_checks, "urandom_seed": seed, "expansion_factor": 4, "security_level": 2 * s_checks,
               "original_trace_length": T, "omicron_domain_length": stark.omicron_domain_length, "fri_domain_length": stark.fri_domain_length,
               "zerofier_coeffs_sha256": sha_packed(tz.coefficients), "zerofier_root": tzr.hex(),
               "proof_len": len(proof), "proof_sha256": hashlib.sha256(proof).hexdigest(), "num_objects": len(ps.objects),
               "first_roots": [o.hex() for o in ps.objects[:3]], "verifies": ok, "false_claim_verifies": bad,
               "reference_seconds": {"preprocess": round(t1 - t0, 1), "prove": round(t2 - t1, 1), "verify_twice": round(t3 - t2, 1)}}
        out = {"runs": []}
        if os.path.exists(path):                  # read at merge time: a long size may have been running beside a short one
            with open(path) as f:
                out = json.load(f)
        out["runs"] = [r for r in out["runs"] if (r["log_fri"], r["num_colinearity_checks"], r["urandom_seed"]) != (log_fri, s_checks, seed)] + [rec]
        out["runs"].sort(key=lambda r: (r["log_fri"], r["num_colinearity_checks"], r["urandom_seed"]))
        print("synthetic AIR, fri 2^%d s=%d seed %d: preprocess %.1fs prove %.1fs verify x2 %.1fs" % (log_fri, s_checks, seed, t1 - t0, t2 - t1, t3 - t2), ok, bad, flush=True)
        dump("fast_stark_synth.json", out)


if __name__ == "__main__":
    big = "--big" in sys.argv
    if "--fri-big" in sys.argv:
        gen_fri()                  # fri.json again, with the 2^14 proof (the other records are reproduced identically)
    elif "--stark-synth" in sys.argv:
        # --stark-synth [log_fri:s:seed ...]; default: the sizes that take minutes in all
        given = [tuple(int(x) for x in a.split(":")) for a in sys.argv[1:] if ":" in a]
        gen_stark_synth(given or [(10, 8, 21), (12, 40, 22), (14, 40, 23)])
    elif "--host-mirror" in sys.argv:
        gen_host_mirror()
    elif "--stark" in sys.argv:
        gen_stark()
    elif "--poly" in sys.argv:
        gen_poly()
    elif big:
        gen_ntt(True)
    else:
        gen_stark()
        gen_field()
        gen_ntt(False)
        gen_poly()
        gen_merkle()
        gen_fri()
        gen_pickle()
