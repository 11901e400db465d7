#!/usr/bin/env python3
"""Several prover THREADS in one process (dev tool; what tools/sanitize.sh runs under ThreadSanitizer): each thread proves the
golden synthetic FRI instances over and over and compares every serialized proof with the reference's SHA-256.  ctypes drops the
GIL inside library calls, so the threads really meet inside the library: its mutex, the pinned root-slot ring, the polled root
slots (sc_fri_commit_dev releases the lock while it polls), the pooled allocator with event-parked frees.
   python tools/thread_stress.py [seconds=20] [threads=3]"""
import hashlib, json, os, sys, threading, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "stark-anatomy_amd")); sys.path.insert(0, REPO)
import starkcore as sc, synth
from algebra import Field
from fri import Fri
from ip import ProofStream
GEN = 85408008396924667383611388730472331217
sc.init(0); lib = sc.lib(); field = Field.main()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
nthreads = int(sys.argv[2]) if len(sys.argv) > 2 else 3
golden = json.load(open(os.path.join(REPO, "tests", "golden", "fri.json")))
cases = []
for rec in golden["prove_synth"]:
    N = 1 << rec["logN"]
    om = field.primitive_nth_root(N)
    coeffs = sc.DeviceVector.from_bytes(synth.synth_packed(rec["coeff_seed"], N // 4).tobytes())
    cw = sc.DeviceVector(N)
    sc._check(lib.sc_coset_evaluate_dev(coeffs.ptr, N // 4, sc.fe_bytes(GEN), sc.fe_bytes(om.value), N, cw.ptr, None))
    sc.synchronize()
    cases.append((rec, Fri(field.generator(), om, N, rec["expansion_factor"], rec["num_colinearity_tests"]), cw))
counts, errors = [0] * nthreads, []


def worker(k):
    t0 = time.time()
    try:
        while time.time() - t0 < budget and not errors:
            for rec, fr, cw in cases:
                ps = ProofStream()
                top = fr.prove(sc.DeviceCodeword(cw, field), ps)
                ser = ps.serialize()
                if top != rec["top_level_indices"] or hashlib.sha256(ser).hexdigest() != rec["serialized_sha256"]:
                    errors.append(("MISMATCH", k, rec["logN"], counts[k]))
                    return
                counts[k] += 1
    except Exception as e:       # noqa: BLE001
        errors.append((k, repr(e)))


threads = [threading.Thread(target=worker, args=(k,)) for k in range(nthreads)]
for t in threads:
    t.start()
for t in threads:
    t.join()
if errors:
    sys.exit("thread stress FAILED: %r" % (errors,))
print("thread stress ok: %d threads, proofs per thread %s" % (nthreads, counts))
