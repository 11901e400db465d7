"""The workloads bench.py times and the tests pin to the reference's own proofs (BASELINE.json configs[4]; SURVEY.md 8(d) "C5").

* `synthetic_stark_instance` -- the synthetic 2-register AIR (a, b) -> (b, a*a + b) at a chosen FRI domain: what
  `FastStark.prove` / `ShardedFastStark.prove` are timed on, and what tests/golden/fast_stark_synth.json holds the REFERENCE's
  proofs of (tests/golden/make_golden.py --stark-synth builds the same columns from synth.synthetic_air_columns).
* `stark_census` / `sharded_census` -- the polynomial-core call census of FastStark.prove (reference code/fast_stark.py:101-151)
  replayed at a chosen size on one GPU / on the sharded layout.
* `plain_stark_prove_measure` / `stark_prove_measure` -- one prover run loop each (single GPU / one process per GPU), returning
  the record a bench line carries.

Definitions only: nothing here parses arguments or prints; bench.py does the timing of record, tests import from here.
"""
import itertools
import os
import sys
import time

GEN = 85408008396924667383611388730472331217          # Field.generator() (reference code/algebra.py:100-102)


def nth_root(n):
    """primitive n-th root of unity: Field.primitive_nth_root (code/algebra.py:104-114) on ints"""
    import synth
    r, order = GEN, 1 << 119
    while order != n:
        r, order = r * r % synth.P, order >> 1
    return r


def synthetic_stark_instance(log_fri, s=40):
    """The synthetic configs[4] workload: the 2-register AIR (a, b) -> (b, a*a + b) over a trace of T = 2^(log_fri - 4) - 4 s rows,
    so that the randomized trace has 2^(log_fri - 4) rows and the FRI domain 2^log_fri points (expansion factor 4).  Returns
    (field, T, packed columns (bytes per register), air, boundary): the columns are handed to the prover as a device-resident
    fast_stark.DeviceTrace -- a trace of 2^20 rows as the reference's list of lists is two million Python objects."""
    from algebra import Field, FieldElement
    from multivariate import MPolynomial
    k = log_fri - 4
    field = Field.main()
    p = field.p
    T = (1 << k) - 4 * s
    from synth import synthetic_air_columns
    col_a, col_b = synthetic_air_columns(T)
    pack = lambda col: b"".join(map(int.to_bytes, col, itertools.repeat(16), itertools.repeat("little")))
    v = MPolynomial.variables(5, field)                  # X, a, b, a', b'
    air = [v[3] - v[2], v[4] - v[1] * v[1] - v[2]]
    boundary = [(0, 0, FieldElement(col_a[0], field)), (0, 1, FieldElement(col_b[0], field)), (T - 1, 1, FieldElement(col_b[T - 1], field))]
    return field, T, [pack(col_a), pack(col_b)], air, boundary


def sharded_census(log_fri, rank, world, dev, stream, group=None, checks=40):
    """BASELINE configs[4]: the polynomial-core call census of FastStark.prove (reference code/fast_stark.py:101-151; SURVEY.md
    8(d)) replayed on the SHARDED layout at fri_domain_length 2^log_fri, omicron_domain_length 2^(log_fri-2), 2 registers:
    4 LDEs to 2^log_fri (ShardedNtt.coset_evaluate: one all-to-all each) + 3 sharded Merkle commits, 2 sharded coset divisions
    at 2^(log_fri-2) (3 all-to-alls each), ShardedFri.prove on the last codeword (no element exchange), 4 * checks openings on
    each of the three committed codewords.  Returns per-stage seconds of this rank and the bytes it sent."""
    import numpy as np
    import torch
    import synth
    from algebra import Field
    from fri import Fri
    from ip import ProofStream
    from sharded import ShardedNtt, ShardedFri
    field = Field.main()
    Nf, No = 1 << log_fri, 1 << (log_fri - 2)
    omega, omicron = field.primitive_nth_root(Nf), field.primitive_nth_root(No)
    ntt_f = ShardedNtt(log_fri, omega.value, rank, world, dev, group=group)
    ntt_o = ShardedNtt(log_fri - 2, omicron.value, rank, world, dev, group=group)
    polys = [torch.from_numpy(synth.synth_packed(60 + i, No // 2).view(np.int64)).to(dev) for i in range(4)]
    fr = Fri(field.generator(), omega, Nf, 4, checks)
    sfri = ShardedFri(fr, ntt_f.n1, rank, world, dev, group=group)
    C = ntt_f.n2

    def sync():
        torch.cuda.synchronize()

    sync()
    times = {}
    t0 = time.perf_counter()
    ps = ProofStream()
    slabs, layers = [], []
    for i, pv in enumerate(polys):                       # 2 boundary quotients, randomizer, combination
        slab = torch.empty(ntt_f.local_shape(False), dtype=torch.int64, device=dev)
        ntt_f.coset_evaluate(pv, GEN, slab)
        slabs.append(slab)
        if i < 3:
            sync()
            layers.append(sfri.commit(slab, C))
            ps.push(layers[-1]["root"])
    sync()
    times["lde_and_commit"] = time.perf_counter() - t0
    t1 = time.perf_counter()
    den = ntt_o.slab_of(polys[3][:No // 4], "census_den").clone()
    q = torch.empty(ntt_o.local_shape(True), dtype=torch.int64, device=dev)
    for i in range(2):                                   # 2 transition quotients (fast_stark.py:113)
        num = ntt_o.slab_of(polys[i], "census_num")
        ntt_o.coset_divide(num, den, GEN, q)
    sync()
    times["coset_divide"] = time.perf_counter() - t1
    t2 = time.perf_counter()
    indices = sfri.prove(slabs[3], ps)
    sync()
    times["fri_prove"] = time.perf_counter() - t2
    t3 = time.perf_counter()
    dup = [i for i in indices] + [(i + 4) % Nf for i in indices]
    quad = sorted(dup + [(i + Nf // 2) % Nf for i in dup])
    for layer in layers:
        entries, paths = sfri._open(layer, quad)
        for e, pth in zip(entries, paths):
            ps.push(e)
            ps.push(pth)
    times["openings"] = time.perf_counter() - t3
    times["total"] = time.perf_counter() - t0
    info = {"fri_rounds": fr.num_rounds(), "proof_objects": len(ps.objects), "proof_sha256_16": __import__("hashlib").sha256(ps.serialize()).hexdigest()[:16],
            "all_to_all_bytes_sent_per_rank": ntt_f.bytes_exchanged + ntt_o.bytes_exchanged, "roots": [l["root"].hex()[:16] for l in layers]}
    return times, info


def stark_census(sc, lib, field, log_fri):
    import ctypes
    import synth
    from fri import Fri
    from ip import ProofStream
    GEN = 85408008396924667383611388730472331217
    Nf, No = 1 << log_fri, 1 << (log_fri - 2)
    omega, omicron = field.primitive_nth_root(Nf), field.primitive_nth_root(No)
    polys = [sc.DeviceVector.from_bytes(synth.synth_packed(60 + i, No // 2).tobytes()) for i in range(4)]
    sc.synchronize()
    t0 = time.perf_counter()
    ps = ProofStream()
    codewords = []
    for i, pv in enumerate(polys):                       # 2 boundary quotients, randomizer, combination
        cw = sc.DeviceVector(Nf)
        sc._check(lib.sc_coset_evaluate_dev(pv.ptr, No // 2, sc.fe_bytes(GEN), sc.fe_bytes(omega.value), Nf, cw.ptr, None))
        codewords.append(sc.DeviceCodeword(cw, field))
        if i < 3:
            ps.push(codewords[i].tree().root)
    t_lde_commit = time.perf_counter() - t0
    # 2 transition quotients: coset NTTs of numerator and zerofier, pointwise division, inverse NTT (device-resident core)
    t1 = time.perf_counter()
    a, b, q = sc.DeviceVector(No), sc.DeviceVector(No), sc.DeviceVector(No)
    for i in range(2):
        sc._check(lib.sc_coset_evaluate_dev(polys[i].ptr, No // 2, sc.fe_bytes(GEN), sc.fe_bytes(omicron.value), No, a.ptr, None))
        sc._check(lib.sc_coset_evaluate_dev(polys[3].ptr, No // 4, sc.fe_bytes(GEN), sc.fe_bytes(omicron.value), No, b.ptr, None))
        sc._check(lib.sc_pointwise_div_dev(a.ptr, b.ptr, q.ptr, No, None))
        sc._check(lib.sc_ntt_dev(q.ptr, a.ptr, No, sc.fe_bytes(omicron.value), 1, None))
    sc.synchronize()
    t_div = time.perf_counter() - t1
    t2 = time.perf_counter()
    fr = Fri(field.generator(), omega, Nf, 4, 40)
    indices = fr.prove(codewords[3], ps)
    t_fri = time.perf_counter() - t2
    t3 = time.perf_counter()
    dup = [i for i in indices] + [(i + 4) % Nf for i in indices]
    quad = sorted(dup + [(i + Nf // 2) % Nf for i in dup])
    for cw in codewords[:3]:
        entries, paths = cw.query(quad)
        for e, pth in zip(entries, paths):
            ps.push(e)
            ps.push(pth)
    t_open = time.perf_counter() - t3
    total = time.perf_counter() - t0
    import hashlib
    return {"ms": total * 1e3, "lde_and_commit_ms": t_lde_commit * 1e3, "coset_divide_ms": t_div * 1e3, "fri_prove_ms": t_fri * 1e3,
            "openings_ms": t_open * 1e3, "fri_rounds": fr.num_rounds(), "proof_objects": len(ps.objects),
            "proof_sha256_16": hashlib.sha256(ps.serialize()).hexdigest()[:16], "roots": [o.hex()[:16] for o in ps.objects[:3]]}


def census_record(times, info, log_fri, world, dist, backend, dev):
    """max over ranks of every stage time (ms) + what rank 0 saw"""
    import torch
    keys = sorted(times)
    t = torch.tensor([times[k] for k in keys], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    rec = {"log2_fri_domain": log_fri, "world_size": world}
    for k, v in zip(keys, t.tolist()):
        rec[k + "_ms"] = v * 1e3
    rec.update(info)
    return rec


def stark_prove_measure(log_fri, steps, warmup, rank, world, dev, dist, backend, phases=False):
    """sharded_stark.ShardedFastStark.prove (reference code/fast_stark.py:76-178) on the synthetic 2-register AIR (a, b) -> (b, a*a + b)
    with a 2^(log_fri - 4)-row randomized trace that is RESIDENT IN HBM as columns when the timed region starts: (seconds summed
    over `steps` proofs, max over ranks per proof; record for rank 0).  Every rank must end with the same proof; rank 0 verifies
    it with FastStark.verify outside the timed region.  phases: one more (untimed) proof with the per-phase breakdown."""
    import hashlib
    import torch
    from fast_stark import DeviceTrace
    from sharded_stark import ShardedFastStark
    s = 40
    field, T, packed, air, boundary = synthetic_stark_instance(log_fri, s)
    stark = ShardedFastStark(field, 4, s, 2 * s, 2, T, rank, world, dev)
    assert stark.fri_domain_length == 1 << log_fri
    trace = DeviceTrace.from_packed(packed, field)
    t0 = time.perf_counter()
    tz, layer, root = stark.preprocess(device_resident=True)
    torch.cuda.synchronize()
    preprocess_s = time.perf_counter() - t0
    for _ in range(warmup):
        stark.prove(trace, air, boundary, tz, layer)
    dist.barrier()
    torch.cuda.synchronize()
    totals, proof = [], None
    for _ in range(steps):
        t0 = time.perf_counter()
        proof = stark.prove(trace, air, boundary, tz, layer)
        torch.cuda.synchronize()
        totals.append(time.perf_counter() - t0)
    dist.barrier()
    t = torch.tensor(totals, dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)              # per step: the slowest rank
    elapsed = float(t.sum().item())
    digest = hashlib.sha256(proof).digest()
    mine = torch.tensor(list(digest[:8]), dtype=torch.int32, device=dev if backend == "nccl" else "cpu")
    lo, hi = mine.clone(), mine.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    same_everywhere = bool(torch.equal(lo, hi))
    phase_ms = None
    if phases:
        stark.phase_log = []
        stark.prove(trace, air, boundary, tz, layer)
        phase_ms = [[name, round(1e3 * sec, 3)] for name, sec in stark.phase_log]
        stark.phase_log = None
    rec = None
    if rank == 0:
        t0 = time.perf_counter()
        verifies = bool(stark.verify(proof, air, boundary, root))
        verify_s = time.perf_counter() - t0
        rec = {"workload": "faststark_prove_synthetic_air_trace_2^%d_fri_2^%d_%dgpu" % (log_fri - 4, log_fri, world), "log2n": log_fri, "world_size": world,
               "registers": 2, "colinearity_checks": s, "expansion_factor": 4, "ms_per_proof": 1e3 * elapsed / steps,
               "trace": "device-resident columns (fast_stark.DeviceTrace), generated on the host outside the timed region",
               "parallelism": "sharded LDEs (1 corner turn each), commitments, quotients, FRI and openings; trace interpolation one register per rank (broadcast); combination replicated",
               "proof_bytes": len(proof), "proof_sha256_16": digest.hex()[:16], "same_proof_on_every_rank": same_everywhere,
               "verify_accepts": verifies, "verify_s": verify_s, "preprocess_s": preprocess_s, "runs_ms": [round(x * 1e3, 3) for x in t.tolist()]}
        if phase_ms is not None:
            rec["phases_ms_synchronised_after_each"] = phase_ms
    return elapsed, same_everywhere, rec


def plain_stark_prove_measure(log_fri, steps):
    """fast_stark.FastStark.prove on one GPU (no process group): ms per proof from a device-resident trace to the serialized proof"""
    from fast_stark import DeviceTrace, FastStark
    s = 40
    field, T, packed, air, boundary = synthetic_stark_instance(log_fri, s)
    stark = FastStark(field, 4, s, 2 * s, 2, T)
    trace = DeviceTrace.from_packed(packed, field)
    sc = sys.modules["starkcore"]
    t0 = time.perf_counter()
    tz, tz_codeword, root = stark.preprocess(device_resident=True)
    sc.synchronize()
    preprocess_s = time.perf_counter() - t0
    for _ in range(4):                                 # untimed: pool, tables, plans and the board's clock settle over the first few proofs
        stark.prove(trace, air, boundary, tz, tz_codeword)
    runs, proof = [], None
    for _ in range(steps):
        sc.synchronize()
        t0 = time.perf_counter()
        proof = stark.prove(trace, air, boundary, tz, tz_codeword)
        sc.synchronize()
        runs.append(time.perf_counter() - t0)
    t0 = time.perf_counter()
    verifies = bool(stark.verify(proof, air, boundary, root))
    return {"workload": "faststark_prove_synthetic_air_trace_2^%d_fri_2^%d_1gpu" % (log_fri - 4, log_fri), "ms_per_proof": 1e3 * min(runs), "median_ms": 1e3 * sorted(runs)[len(runs) // 2],
            "runs_ms": [round(1e3 * r, 3) for r in runs], "registers": 2, "colinearity_checks": s, "expansion_factor": 4,
            "trace": "device-resident columns (fast_stark.DeviceTrace)", "proof_bytes": len(proof), "verify_accepts": verifies,
            "verify_s": time.perf_counter() - t0, "preprocess_s": preprocess_s,
            "randomness": "the operating system's (getrandom, drawn by the library)" if sys.modules["fast_stark"].os_urandom_is_genuine() else "patched os.urandom"}
