#!/usr/bin/env python3
"""FastStark.prove on a LONG synthetic trace (the caller rows 8(f)-2 / 8(f)-3 at scale).  The reference's only
workload is the 28-row Rescue-Prime trace; here a 2-register quadratic recurrence  (a, b) -> (b, a*a + b)  is run for 2^k - 4s
cycles so that the whole prover works on ~2^k-row columns: interpolation through 2^k points, two exact boundary quotients, the
AIR substitution, two transition quotients, 4 LDEs to 4 * 2^(k+1), FRI.  Prints where the time goes, with every polynomial in
HBM (FastStark.DEVICE_MIN = 32, the default) and, for small k, with the reference's host-list data flow for comparison.

   python tools/stark_big_trace.py [log2_rows=14] [--host-too]            (dev tool)"""
import json, os, random, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "stark-anatomy_amd"))
sys.setrecursionlimit(100000)
import starkcore as sc
import fast_stark
from fast_stark import FastStark
from algebra import Field, FieldElement
from multivariate import MPolynomial
sc.init(0)
k = int(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else 14
field = Field.main()
s = 40
T = (1 << k) - 4 * s
rng = random.Random(5)


def run(device_min):
    rng.seed(5)
    fast_stark.os.urandom = lambda n: bytes(rng.getrandbits(8) for _ in range(n))
    FastStark.DEVICE_MIN = device_min
    stark = FastStark(field, 4, s, 2 * s, 2, T)
    p = field.p
    a, b = 3, 5
    rows = []
    for _ in range(T):
        rows.append((a, b))
        a, b = b, (a * a + b) % p
    trace = [[FieldElement(x, field), FieldElement(y, field)] for x, y in rows]
    v = MPolynomial.variables(5, field)                  # X, a, b, a', b'
    air = [v[3] - v[2], v[4] - v[1] * v[1] - v[2]]
    boundary = [(0, 0, trace[0][0]), (0, 1, trace[0][1]), (T - 1, 1, trace[T - 1][1])]
    t0 = time.perf_counter(); tz, tzc, tzr = stark.preprocess(); t_pre = time.perf_counter() - t0
    t0 = time.perf_counter(); proof = stark.prove(trace, air, boundary, tz, tzc); t_prove = time.perf_counter() - t0
    import cProfile, pstats, io
    prof = cProfile.Profile(); prof.enable()
    t0 = time.perf_counter(); proof = stark.prove(trace, air, boundary, tz, tzc); t_prove2 = time.perf_counter() - t0
    prof.disable()
    t0 = time.perf_counter(); ok = stark.verify(proof, air, boundary, tzr); t_ver = time.perf_counter() - t0
    out = io.StringIO(); pstats.Stats(prof, stream=out).sort_stats("tottime").print_stats(8)
    top = [l.strip() for l in out.getvalue().splitlines() if l.strip() and l.strip()[0].isdigit()][:8]
    import hashlib
    return dict(device_min=device_min, rows=1 << k, omicron_domain=stark.omicron_domain_length, fri_domain=stark.fri_domain_length, preprocess_s=round(t_pre, 3),
                prove_first_s=round(t_prove, 3), prove_s=round(t_prove2, 3), verify_s=round(t_ver, 3), verifies=ok, proof_bytes=len(proof),
                proof_sha256_16=hashlib.sha256(proof).hexdigest()[:16], top_tottime=top)


res = [run(32)]
if "--host-too" in sys.argv:
    res.append(run(10 ** 9))
    res.append({"proofs_identical": res[0]["proof_sha256_16"] == res[1]["proof_sha256_16"]})
for r in res:
    print(json.dumps(r), flush=True)
