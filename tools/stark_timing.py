#!/usr/bin/env python3
"""FastStark.prove / verify at the reference's FastRPSSS parameters (code/fast_rpsss.py:27-36: expansion factor 4, 64 colinearity checks,
security level 128, transition_constraints_degree 3 => FRI domain 4096) -- the one end-to-end timing the reference publishes
(docs/faster.md:469: 72 s; 38.6 s in the survey container).  Dev tool; prints a cProfile summary with --profile."""
import cProfile, json, os, pstats, random, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "stark-anatomy_amd")); sys.path.insert(0, os.path.join(REPO, "tests"))
sys.setrecursionlimit(10000)
import starkcore as sc
import fast_stark
from fast_stark import FastStark
from algebra import Field
from workload_rescue_prime import RescuePrime
sc.init(0)
rng = random.Random(5)
fast_stark.os.urandom = lambda k: bytes(rng.getrandbits(8) for _ in range(k))
field = Field.main()
rp = RescuePrime()
stark = FastStark(field, 4, 64, 128, rp.m, rp.N + 1, transition_constraints_degree=3)
t0 = time.perf_counter(); tz, tzc, tzr = stark.preprocess(); t_pre = time.perf_counter() - t0
inp = field.sample(b"0xdeadbeef"); out = rp.hash(inp)
trace = rp.trace(inp); air = rp.transition_constraints(stark.omicron); boundary = rp.boundary_constraints(out)
stark.prove(trace, air, boundary, tz, tzc)          # warm-up (plans, tables)
prof = cProfile.Profile() if "--profile" in sys.argv else None
t0 = time.perf_counter()
if prof: prof.enable()
proof = stark.prove(trace, air, boundary, tz, tzc)
if prof: prof.disable()
t_prove = time.perf_counter() - t0
vprof = cProfile.Profile() if "--profile-verify" in sys.argv else None
t0 = time.perf_counter()
if vprof: vprof.enable()
ok = stark.verify(proof, air, boundary, tzr)
if vprof: vprof.disable()
t_ver = time.perf_counter() - t0
print(json.dumps(dict(fri_domain=stark.fri_domain_length, omicron_domain=stark.omicron_domain_length, preprocess_s=round(t_pre, 3), prove_s=round(t_prove, 3),
                      verify_s=round(t_ver, 3), verifies=ok, proof_bytes=len(proof))))
if vprof:
    pstats.Stats(vprof).sort_stats("cumulative").print_stats(22)
if prof:
    pstats.Stats(prof).sort_stats("cumulative").print_stats(18)
