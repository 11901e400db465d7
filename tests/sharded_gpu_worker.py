"""Worker for tests/test_gpu_sharded.py::test_two_ranks_share_one_gpu: one rank of a world_size-N job whose ranks all use GPU 0.
The local stages run on the HIP engine (the product path); only the exchange differs from production: backend gloo with a
host-staged all-to-all instead of RCCL (a single-GPU box cannot host an RCCL job).  Checks the sharded forward / inverse / LDE
against the oracle's transform of the full vector."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "stark-anatomy_amd"), REPO):
    if p not in sys.path:
        sys.path.insert(0, p)

from oracle import py_oracle as po          # noqa: E402
import synth                                 # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ["STARKCORE_DEVICE"] = "0"
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    import starkcore as sc
    sc.init(0)
    from sharded import ShardedNtt, gather_natural
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ok = True
    for log2n in (12, 17, 20):
        n = 1 << log2n
        root = po.primitive_nth_root(n)
        full_in = synth.synth_packed(3, n).tobytes()
        want = po.C.ntt(root, full_in, n)
        # the forms of the corner turn: the rank's own block written in place (default) or through the exchange; one blocking
        # exchange or row blocks, with the second pass of the row stage deferred or not -- always the same transform
        forms = [dict(), dict(overlap_chunks=4), dict(overlap_chunks=2, defer_last_pass=False), dict(always_exchange=True), dict(always_exchange=True, overlap_chunks=2),
                 # the direct-store corner turn: the column stage stores block h into rank h's receive buffer (mapped through HIP
                 # IPC -- here between processes on ONE device), a flag barrier instead of a collective
                 dict(direct_store=True)]
        if log2n == 20:
            forms += [dict(log_n1=10), dict(direct_store=True, log_n1=10)]          # another split of the same transform
        for kw in forms:
            eng = ShardedNtt(log2n, root, rank, world, dev, **kw)
            if kw.get("direct_store") and not eng.direct_store:
                print("rank", rank, "the direct-store corner turn did not come up", flush=True)
                ok = False
                break
            x = eng.synthetic_input(seed=3)
            y = torch.empty(eng.local_shape(False), dtype=torch.int64, device=dev)
            z = torch.empty_like(x)
            eng.forward(x, y)
            eng.inverse(y, z)
            torch.cuda.synchronize()
            got = gather_natural(y.cpu(), eng.n2, eng.n1, world).numpy().tobytes()
            ok &= got == want
            ok &= torch.equal(z, x)
            if kw.get("direct_store"):
                # several transforms back to back without waiting in between: the two receive buffers alternate, a rank that runs
                # ahead must not overwrite what a slower rank's row stage still reads
                for _ in range(5):
                    eng.forward(x, y)
                    eng.inverse(y, z)
                eng.forward(z, y)
                torch.cuda.synchronize()
                ok &= gather_natural(y.cpu(), eng.n2, eng.n1, world).numpy().tobytes() == want and torch.equal(z, x)
                ok &= eng.stages.direct_timed_out() == 0
                dist.barrier()
                eng.stages.release_direct()
            if not ok:
                print("rank", rank, "MISMATCH at log2n", log2n, kw, flush=True)
                break
        eng = ShardedNtt(log2n, root, rank, world, dev)
        m = n // 8 + 3
        coeffs = synth.synth_packed(9, m)
        lde = torch.empty(eng.local_shape(False), dtype=torch.int64, device=dev)
        eng.coset_evaluate(torch.from_numpy(coeffs.view(np.int64).copy()).to(dev), po.GENERATOR, lde)
        torch.cuda.synchronize()
        got_lde = gather_natural(lde.cpu(), eng.n2, eng.n1, world).numpy().tobytes()
        ok &= got_lde == po.C.coset_evaluate(coeffs.tobytes(), m, po.GENERATOR, root, n)
        if not ok:
            print("rank", rank, "MISMATCH at log2n", log2n, flush=True)
            break
        if log2n <= 17:
            ok &= poly_check(eng, rank, world, n, dev)
            if not ok:
                print("rank", rank, "POLY MISMATCH at log2n", log2n, flush=True)
                break
    ok &= fri_check(rank, world, dev)
    ok &= stark_check(rank, world, dev)
    dist.barrier()
    dist.destroy_process_group()
    if not ok:
        sys.exit(3)
    print("rank", rank, "ok")


def poly_check(eng, rank, world, n, dev):
    """SURVEY 8(e)-5 with the HIP engine on every rank: fast_multiply / fast_coset_divide cores on slabs (sc_scale_slab_dev,
    sc_pointwise_mul_dev / _div_dev between the sharded transforms) against the oracle's products."""
    from sharded import gather_natural
    as_t = lambda vals: torch.from_numpy(np.frombuffer(synth.pack_ints(vals), dtype=np.int64).reshape(len(vals), 2).copy()).to(dev)
    la, lb = n // 2 - 3, n // 2 + 1
    a, b = synth.synth_ints(21, la), synth.synth_ints(22, lb)
    root = po.primitive_nth_root(n)                           # product through the C oracle (ntt, Hadamard, intt: code/ntt.py:58-64)
    pad = lambda v: synth.pack_ints(v) + bytes(16 * (n - len(v)))
    prod = po.C.intt(root, po.C.pointwise_mul(po.C.ntt(root, pad(a), n), po.C.ntt(root, pad(b), n), n), n)
    want = synth.unpack_ints(prod)[:la + lb - 1]
    out = torch.empty(eng.local_shape(True), dtype=torch.int64, device=dev)
    eng.multiply(eng.slab_of(as_t(a), "ta").clone(), eng.slab_of(as_t(b), "tb").clone(), out)
    torch.cuda.synchronize()
    got = synth.unpack_ints(gather_natural(out.cpu(), eng.n1, eng.n2, world).numpy().tobytes())
    ok = got[:len(want)] == want and not any(got[len(want):])
    q = torch.empty(eng.local_shape(True), dtype=torch.int64, device=dev)
    eng.coset_divide(eng.slab_of(as_t(want), "ta").clone(), eng.slab_of(as_t(b), "tb").clone(), po.GENERATOR, q)
    torch.cuda.synchronize()
    gq = synth.unpack_ints(gather_natural(q.cpu(), eng.n1, eng.n2, world).numpy().tobytes())
    ok = ok and gq[:la] == a and not any(gq[la:])
    try:                                                     # a divisor with a zero on the coset: every rank raises together
        eng.coset_divide(eng.slab_of(as_t(want), "ta").clone(), eng.slab_of(as_t([(-po.GENERATOR) % po.P, 1]), "tb").clone(), po.GENERATOR, q)
        ok = False
    except AssertionError as e:
        ok = ok and "divide by zero" in str(e)
    return bool(ok)


def stark_check(rank, world, dev):
    """BASELINE configs[4] as a real prover: sharded_stark.ShardedFastStark (sharded LDEs, commitments, quotients, FRI and openings;
    rank 0's random bytes broadcast) on a synthetic 2-register AIR must produce, on every rank, the byte string that
    fast_stark.FastStark.prove produces on one GPU from the same random bytes (reference code/fast_stark.py:76-178)."""
    import random
    import fast_stark
    from fast_stark import FastStark
    from sharded_stark import ShardedFastStark
    from algebra import Field, FieldElement
    from multivariate import MPolynomial
    field = Field.main()
    ok = True
    for k, s in ((10, 8), (12, 40)):
        T = (1 << k) - 4 * s
        a, b, rows = 3, 5, []
        for _ in range(T):
            rows.append((a, b))
            a, b = b, (a * a + b) % field.p
        trace = [[FieldElement(x, field), FieldElement(y, field)] for x, y in rows]
        v = MPolynomial.variables(5, field)                  # X, a, b, a', b'
        air = [v[3] - v[2], v[4] - v[1] * v[1] - v[2]]
        boundary = [(0, 0, trace[0][0]), (0, 1, trace[0][1]), (T - 1, 1, trace[T - 1][1])]

        def seeded():
            rng = random.Random(1000 + k)
            fast_stark.os.urandom = lambda n: bytes(rng.getrandbits(8) for _ in range(n))
        seeded()
        one = FastStark(field, 4, s, 2 * s, 2, T)
        tz, tzc, tzr = one.preprocess()
        want = one.prove(trace, air, boundary, tz, tzc)
        seeded()
        many = ShardedFastStark(field, 4, s, 2 * s, 2, T, rank, world, dev)
        tz2, layer, root = many.preprocess()
        got = many.prove(trace, air, boundary, tz2, layer)
        good = root == tzr and got == want and one.verify(got, air, boundary, tzr) is True
        # the same proof from a device-resident trace (columns in HBM) and a zerofier made on the device from its closed form
        seeded()
        columns = fast_stark.DeviceTrace.from_rows(trace, field)
        tz3, layer3, root3 = many.preprocess(device_resident=True)
        good = good and root3 == tzr and many.prove(columns, air, boundary, tz3, layer3) == want
        # a FALSE WITNESS (a trace cell that no boundary condition pins, perturbed): the transition quotients are not exact any more,
        # the value-domain route must notice and hand the constraint to the reference's way -- same bytes as the one-GPU prover,
        # and a proof the verifier rejects
        bent = [list(row) for row in trace]
        bent[T // 2][0] = bent[T // 2][0] + FieldElement(12345, field)
        seeded()
        want_bent = one.prove(bent, air, boundary, tz, tzc)
        seeded()
        got_bent = many.prove(bent, air, boundary, tz2, layer)
        good = good and got_bent == want_bent and one.verify(got_bent, air, boundary, tzr) is False
        if not good:
            print("rank", rank, "STARK MISMATCH k", k, root == tzr, len(got), len(want), flush=True)
        ok &= good
    ok &= stark_reference_goldens(rank, world, dev)
    return ok


def stark_reference_goldens(rank, world, dev, max_log_fri=18, max_log_fri_host_rows=14):
    """The workload bench.py times for BASELINE configs[4] (workloads.synthetic_stark_instance), as the REFERENCE's FastStark proved it
    (tests/golden/fast_stark_synth.json, written by make_golden.py --stark-synth with the same seeded os.urandom): every rank must
    end with those bytes, from the host-list trace (up to FRI 2^14: above that the lists only cost time) and from device-resident
    columns (every size the fixture holds: the reference needs 20 minutes for the 2^16 proof and hours for 2^18)."""
    import hashlib
    import json
    import random
    import workloads
    import fast_stark
    from algebra import FieldElement
    from ip import ProofStream
    from sharded_stark import ShardedFastStark
    golden = json.load(open(os.path.join(REPO, "tests", "golden", "fast_stark_synth.json")))
    genuine, ok = fast_stark.os.urandom, True
    try:
        for rec in golden["runs"]:
            log_fri, s = rec["log_fri"], rec["num_colinearity_checks"]
            if log_fri > max_log_fri:
                continue
            field, T, packed, air, boundary = workloads.synthetic_stark_instance(log_fri, s)
            stark = ShardedFastStark(field, 4, s, rec["security_level"], 2, T, rank, world, dev)
            for resident in ((False, True) if log_fri <= max_log_fri_host_rows else (True,)):
                rng = random.Random(rec["urandom_seed"])
                fast_stark.os.urandom = lambda k, rng=rng: bytes(rng.getrandbits(8) for _ in range(k))
                tz, layer, root = stark.preprocess(device_resident=resident)
                if resident:
                    trace = fast_stark.DeviceTrace.from_packed(packed, field)
                else:
                    trace = [[FieldElement(a, field), FieldElement(b, field)] for a, b in zip(*synth.synthetic_air_columns(T))]
                proof = stark.prove(trace, air, boundary, tz, layer)
                objects = ProofStream().deserialize(proof).objects
                good = (root.hex() == rec["zerofier_root"] and [o.hex() for o in objects[:3]] == rec["first_roots"] and len(objects) == rec["num_objects"]
                        and len(proof) == rec["proof_len"] and hashlib.sha256(proof).hexdigest() == rec["proof_sha256"])
                if not good:
                    print("rank", rank, "REFERENCE GOLDEN MISMATCH fri 2^%d resident %s" % (log_fri, resident), root.hex() == rec["zerofier_root"], len(proof), rec["proof_len"], flush=True)
                ok &= good
    finally:
        fast_stark.os.urandom = genuine
    return ok


def fri_check(rank, world, dev):
    """ShardedFri with the HIP engine on every rank (slab-local folds, sharded Merkle commits, collective openings): the proof
    must be the reference's, byte for byte (golden SHA-256 of the serialized proof stream)."""
    import hashlib
    import json
    from sharded import ShardedFri
    from algebra import Field
    from fri import Fri
    from ip import ProofStream
    golden = json.load(open(os.path.join(REPO, "tests", "golden", "fri.json")))
    field = Field.main()
    ok = True
    for rec in golden["prove_synth"]:
        logN = rec["logN"]
        N = 1 << logN
        om = field.primitive_nth_root(N)
        coeffs = synth.synth_packed(rec["coeff_seed"], N // 4).tobytes()
        cw = np.frombuffer(po.C.coset_evaluate(coeffs, N // 4, po.GENERATOR, om.value, N), dtype=np.int64).reshape(N, 2)
        for logR in {6: (2, 3), 10: (3, 5, 8), 12: (4, 6)}.get(logN, ()):
            R = 1 << logR
            if R < world:
                continue
            C, Rw = N // R, R // world
            slab = torch.from_numpy(cw.reshape(C, R, 2)[:, rank * Rw:(rank + 1) * Rw, :].copy()).to(dev)
            fr = Fri(field.generator(), om, N, rec["expansion_factor"], rec["num_colinearity_tests"])
            # local_tail: never gather early / gather half way through the rounds / the default (these sizes: before round 0)
            for tail in (0, N >> 2, None):
                ps = ProofStream()
                top = ShardedFri(fr, R, rank, world, dev, local_tail=tail).prove(slab, ps)
                ser = ps.serialize()
                good = (top == rec["top_level_indices"] and len(ser) == rec["serialized_len"] and hashlib.sha256(ser).hexdigest() == rec["serialized_sha256"])
                if not good:
                    print("rank", rank, "FRI MISMATCH logN", logN, "R", R, "tail", tail, flush=True)
                ok &= good
        # the NATURAL contiguous layout (SURVEY 8(e) "FRI fold": one neighbour exchange per fold), HIP engine on every rank
        from sharded import ContiguousFri
        seg = N // world
        chunk = torch.from_numpy(cw[rank * seg:(rank + 1) * seg].copy()).to(dev)
        fr = Fri(field.generator(), om, N, rec["expansion_factor"], rec["num_colinearity_tests"])
        ps = ProofStream()
        top = ContiguousFri(fr, rank, world, dev).prove(chunk, ps)
        ser = ps.serialize()
        good = (top == rec["top_level_indices"] and len(ser) == rec["serialized_len"] and hashlib.sha256(ser).hexdigest() == rec["serialized_sha256"])
        if not good:
            print("rank", rank, "CONTIGUOUS FRI MISMATCH logN", logN, flush=True)
        ok &= good
    # independent columns, one register per rank (HIP LDE + Merkle commit), roots gathered in column order
    from sharded import ColumnReplicas
    order = 1 << 12
    gen = po.primitive_nth_root(order)
    cols = [synth.synth_packed(50 + i, 1000 + i).tobytes() for i in range(5)]
    mine, roots = ColumnReplicas(rank, world, dev).lde_and_commit(cols, po.GENERATOR, gen, order)
    want = [po.C.merkle_commit(po.C.coset_evaluate(c, len(c) // 16, po.GENERATOR, gen, order), order) for c in cols]
    ok &= roots == want and sorted(mine) == [i for i in range(5) if i % world == rank]
    return ok


def direct_faults_main():
    """tests/test_gpu_sharded.py::test_direct_store_faults_end_in_a_correct_transform: what the environment makes fail, the set-up
    must survive -- a fine-grained export that fails, peers that cannot import one kind of region, no kind at all, a peer that is
    late for a flag barrier -- every time ending in a transform equal to the oracle's through the next form in line."""
    import time
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    case = sys.argv[2]
    os.environ["STARKCORE_DEVICE"] = "0"
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    import starkcore as sc
    sc.init(0)
    from sharded import ShardedNtt, DirectStoreTimeout, gather_natural
    dist.init_process_group("gloo", rank=rank, world_size=world)
    log2n = 14
    n = 1 << log2n
    root = po.primitive_nth_root(n)
    want = po.C.ntt(root, synth.synth_packed(3, n).tobytes(), n)
    eng = ShardedNtt(log2n, root, rank, world, dev, direct_store=True)
    x = eng.synthetic_input(seed=3)
    y = torch.empty(eng.local_shape(False), dtype=torch.int64, device=dev)
    z = torch.empty_like(x)

    def correct():
        eng.forward(x, y)
        eng.inverse(y, z)
        torch.cuda.synchronize()
        return gather_natural(y.cpu(), eng.n2, eng.n1, world).numpy().tobytes() == want and torch.equal(z, x)
    setup = "; ".join(eng.corner_turn_setup)
    ok = True
    if case == "fine_export_fails":              # STARKCORE_TEST_FINE_EXPORT_FAILS=1: the region silently becomes coarse-grained
        ok &= eng.direct_store and eng.stages.region_kind() == "coarse-grained" and correct() and eng.stages.direct_timed_out() == 0
    elif case == "fine_import_fails":            # STARKCORE_TEST_OPEN_FAILS_KIND=1: first attempt down on every rank, the coarse-grained one up
        ok &= eng.direct_store and "fine-grained regions requested: did not come up" in setup and "coarse-grained regions requested: up" in setup
        ok &= eng.stages.region_kind() == "coarse-grained" and correct() and eng.stages.direct_timed_out() == 0
    elif case == "nothing_imports":              # neither kind can be mapped: the collective exchange carries the corner turn
        ok &= (not eng.direct_store) and "collective exchange" in setup and correct()
    elif case == "late_peer":                    # STARKCORE_IPC_BARRIER_SPINS small: a rank that is late by a second misses the barrier
        ok &= eng.direct_store and correct()
        dist.barrier()
        if rank == 1:
            time.sleep(1.5)
        try:
            eng.forward(x, y)
            torch.cuda.synchronize()
        except DirectStoreTimeout:
            pass
        if rank != 1:
            ok &= eng.stages.direct_timed_out() != 0
            try:                                 # sticky: the plan refuses every later transform
                eng.forward(x, y)
                ok = False
            except DirectStoreTimeout as e:
                ok &= "never arrived" in str(e)
        ok &= eng.fall_back_to_exchange() is True           # collective: every rank leaves the direct-store form
        ok &= (not eng.direct_store) and correct()
    else:
        ok = False
    if not ok:
        print("rank", rank, "case", case, "FAILED; set-up:", setup, flush=True)
    dist.barrier()
    dist.destroy_process_group()
    if not ok:
        sys.exit(3)
    print("rank", rank, "ok")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "direct_faults":
        direct_faults_main()
    else:
        main()
