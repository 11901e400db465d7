#!/usr/bin/env python3
"""Fri.prove under the microscope (dev tool; one script for what used to be eleven).  BASELINE configs[3]: 2^22 codeword, expansion
factor 4, 40 colinearity checks, codeword resident in HBM.

  python tools/fri_profile.py timing [log2n=22] [runs=30]   host clock around the call: best / median, the proof verified once, and how much
                                                            of a call is the library (set STARKCORE_FRI_TIMING=1 for the library's own phases
                                                            and the persistent tail kernel's per-round stamps on stderr)
  python tools/fri_profile.py trace-run                     six proofs 20 ms apart: run under `rocprofv3 --kernel-trace`, then
  python tools/fri_profile.py trace-report <kernel_trace.csv>   the kernel timeline of the LAST proof in that trace (start, duration, gap, grid)
  python tools/fri_profile.py pyprofile                     cProfile of the Python side over 100 proofs
  python tools/fri_profile.py stress [seconds=30]           the golden synthetic proofs (tests/golden/fri.json) proved over and over, every
                                                            serialized proof compared with the reference's SHA-256; run several copies at once
"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "stark-anatomy_amd"))
sys.path.insert(0, REPO)
GEN = 85408008396924667383611388730472331217


def setup(log2n):
    """(sc, field, Fri instance, device vector of the LDE codeword)"""
    import starkcore as sc
    import synth
    from algebra import Field
    from fri import Fri
    sc.init(0)
    lib, field = sc.lib(), Field.main()
    N = 1 << log2n
    om = field.primitive_nth_root(N)
    coeffs = sc.DeviceVector.from_bytes(synth.synth_packed(4002, N // 4).tobytes())
    cwv = sc.DeviceVector(N)
    sc._check(lib.sc_coset_evaluate_dev(coeffs.ptr, N // 4, sc.fe_bytes(GEN), sc.fe_bytes(om.value), N, cwv.ptr, None))
    sc.synchronize()
    return sc, field, Fri(field.generator(), om, N, 4, 40), cwv


def timing(argv):
    import statistics
    from ip import ProofStream
    log2n = int(argv[0]) if argv else 22
    runs = int(argv[1]) if len(argv) > 1 else 30
    sc, field, fr, cwv = setup(log2n)
    times = []
    for i in range(runs + 3):
        ps, cw = ProofStream(), sc.DeviceCodeword(cwv, field)
        t0 = time.perf_counter()
        fr.prove(cw, ps)
        if i >= 3:
            times.append(time.perf_counter() - t0)
    ser = ps.serialize()
    ok = fr.verify(ProofStream().deserialize(ser), [])
    print("Fri.prove 2^%d: best %.3f ms  median %.3f ms  (%d runs)  proof %d bytes  verify %s" % (log2n, min(times) * 1e3, statistics.median(times) * 1e3, runs, len(ser), ok))
    # where the host time of one call goes: the library call, the Python around it
    import fri as _fri
    lib = _fri._sc.lib()
    real, spent = lib.sc_fri_prove_dev, [0.0]

    def timed(*a):
        t0 = time.perf_counter()
        r = real(*a)
        spent[0] += time.perf_counter() - t0
        return r

    lib.sc_fri_prove_dev = timed
    tot = 0.0
    for _ in range(10):
        ps, cw = ProofStream(), sc.DeviceCodeword(cwv, field)
        t0 = time.perf_counter()
        fr.prove(cw, ps)
        tot += time.perf_counter() - t0
    print("per call: total %.1f us, of which sc_fri_prove_dev %.1f us, python around it %.1f us" % (tot / 10 * 1e6, spent[0] / 10 * 1e6, (tot - spent[0]) / 10 * 1e6))


def trace_run(argv):
    from ip import ProofStream
    sc, field, fr, cwv = setup(22)
    for _ in range(6):
        t0 = time.perf_counter()
        fr.prove(sc.DeviceCodeword(cwv, field), ProofStream())
        print("prove_ms", round((time.perf_counter() - t0) * 1e3, 3))
        time.sleep(0.02)


def trace_report(argv):
    import csv
    rows = list(csv.DictReader(open(argv[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    groups, cur, last_end = [], [], None                 # proofs are separated by >= 10 ms of idle
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if last_end is not None and s - last_end > 10_000_000:
            groups.append(cur)
            cur = []
        cur.append(r)
        last_end = e
    groups.append(cur)
    g = groups[-1]
    t0 = int(g[0]["Start_Timestamp"])
    busy, prev_end = 0, t0
    print("%9s %8s %8s  %-44s %s" % ("t_us", "dur_us", "gap_us", "kernel", "grid"))
    for r in g:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("sc::", "")
        print("%9.1f %8.1f %8.1f  %-44s %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, name[:44], r.get("Grid_Size_X", r.get("Grid_Size", ""))))
        busy += e - s
        prev_end = e
    print("span_us %.1f busy_us %.1f kernels %d" % ((prev_end - t0) / 1e3, busy / 1e3, len(g)))


def pyprofile(argv):
    import cProfile
    import pstats
    from ip import ProofStream
    sc, field, fr, cwv = setup(22)

    def run(k):
        for _ in range(k):
            fr.prove(sc.DeviceCodeword(cwv, field), ProofStream())

    run(5)
    pr = cProfile.Profile()
    pr.enable()
    run(100)
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(22)


def stress(argv):
    import hashlib
    import json
    import starkcore as sc
    import synth
    from algebra import Field
    from fri import Fri
    from ip import ProofStream
    sc.init(0)
    lib, field = sc.lib(), Field.main()
    budget = float(argv[0]) if argv else 30.0
    cases = []
    for rec in json.load(open(os.path.join(REPO, "tests", "golden", "fri.json")))["prove_synth"]:
        N = 1 << rec["logN"]
        om = field.primitive_nth_root(N)
        coeffs = sc.DeviceVector.from_bytes(synth.synth_packed(rec["coeff_seed"], N // 4).tobytes())
        cw = sc.DeviceVector(N)
        sc._check(lib.sc_coset_evaluate_dev(coeffs.ptr, N // 4, sc.fe_bytes(GEN), sc.fe_bytes(om.value), N, cw.ptr, None))
        sc.synchronize()
        cases.append((rec, Fri(field.generator(), om, N, rec["expansion_factor"], rec["num_colinearity_tests"]), cw))
    t0, n = time.time(), 0
    while time.time() - t0 < budget:
        for rec, fr, cw in cases:
            ps = ProofStream()
            top = fr.prove(sc.DeviceCodeword(cw, field), ps)
            assert top == rec["top_level_indices"] and hashlib.sha256(ps.serialize()).hexdigest() == rec["serialized_sha256"], ("MISMATCH", rec["logN"], n)
            n += 1
    print("fri stress ok:", n, "proofs, pid", os.getpid())


if __name__ == "__main__":
    commands = {"timing": timing, "trace-run": trace_run, "trace-report": trace_report, "pyprofile": pyprofile, "stress": stress}
    if len(sys.argv) < 2 or sys.argv[1] not in commands:
        sys.exit(__doc__)
    commands[sys.argv[1]](sys.argv[2:])
