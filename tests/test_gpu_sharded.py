"""Multi-GPU four-step NTT, GPU side: the batched HIP stages (column transforms, outer twiddle, transposed row
transforms) driven by sharded.py for a SIMULATED world of ranks on one device -- the all-to-all is replaced by the
equivalent tensor shuffle -- and compared with the single-GPU transform and the oracle."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import py_oracle as po
import synth

import os as _os
import sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sc():
    import starkcore
    assert starkcore.device_count() > 0
    starkcore.init()
    return starkcore


def _simulate(sc, log2n, world, seed, fused=True, blocks=1, defer=True, diag_in_place=True):
    """fused: the ranks' stage objects (sc_fourstep_*: column stage with the outer twiddle and, with diag_in_place, the rank's own
    block written straight into its receive buffer; row stage reading the [G][R/G][C/G] layout in place, optionally in `blocks`
    row blocks with the second pass deferred); otherwise the primitive-by-primitive path (separate twiddle and reassembly)."""
    from sharded import ShardedNtt
    dev = torch.device("cuda", 0)
    n = 1 << log2n
    root = po.primitive_nth_root(n)
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        engs = [ShardedNtt(log2n, root, r, world, dev) for r in range(world)]
        xs = [e.synthetic_input(seed) for e in engs]
        n1, n2 = engs[0].n1, engs[0].n2

        def run(srcs, R, C, inverse):
            rt, scale = (engs[0].root_inv, engs[0].n_inv) if inverse else (root, 1)
            rw, cw = R // world, C // world
            outs = []
            if not fused:
                a = [e.stage_cols(s_, R, C, rt, scale).clone() for e, s_ in zip(engs, srcs)]
                for h, e in enumerate(engs):
                    # what all_to_all_single delivers to rank h: from every rank g its rows [h*rw, (h+1)*rw)
                    recv = torch.stack([a[g][h * rw:(h + 1) * rw] for g in range(world)], dim=0).contiguous()
                    dst = torch.empty((C, rw, 2), dtype=torch.int64, device=dev)
                    e.stage_rows(e.assemble_rows(recv, R, C) if world > 1 else a[0], dst, R, C, rt)
                    outs.append(dst)
                return outs
            inv = 1 if inverse else 0
            # poisoned buffers: whatever is not written by the stage that owns it shows up in the result
            sends = [torch.full((world, rw, cw, 2), -1, dtype=torch.int64, device=dev) for _ in engs]
            recvs = [torch.full((world, rw, cw, 2), -1, dtype=torch.int64, device=dev) for _ in engs]
            for e, s_, snd, rcv in zip(engs, srcs, sends, recvs):
                e.stages.cols(inv, s_, snd, rcv if diag_in_place else None)
            for h in range(world):
                for g_ in range(world):
                    if g_ != h or not diag_in_place:
                        recvs[h][g_].copy_(sends[g_][h])          # the corner turn: block h of rank g -> block g of rank h
            K = blocks if (rw % blocks == 0 and rw // blocks >= 1) else 1
            for e, rcv in zip(engs, recvs):
                dst = torch.empty((C, rw, 2), dtype=torch.int64, device=dev)
                for q in range(K):
                    e.stages.rows(inv, rcv, dst, q, K, defer and K > 1)
                if defer and K > 1:
                    e.stages.rows_finish(inv, dst)
                outs.append(dst)
            return outs

        ys = run(xs, n1, n2, False)
        zs = run(ys, n2, n1, True)
    torch.cuda.synchronize()
    full_in = synth.synth_packed(seed, n).tobytes()
    got = torch.cat(ys, dim=1).reshape(n, 2).cpu().numpy().tobytes()
    back = torch.cat(zs, dim=1).reshape(n, 2).cpu().numpy().tobytes()
    return full_in, got, back, root


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("log2n,world", [(8, 1), (10, 2), (13, 4), (16, 8), (18, 8), (21, 2)])
def test_sharded_simulated_world(sc, log2n, world, fused):
    full_in, got, back, root = _simulate(sc, log2n, world, seed=11, fused=fused)
    n = 1 << log2n
    assert got == po.C.ntt(root, full_in, n)
    assert back == full_in


@pytest.mark.parametrize("blocks,defer,diag", [(2, True, True), (4, True, True), (4, False, True), (1, False, False), (2, True, False)])
@pytest.mark.parametrize("log2n,world", [(10, 2), (16, 4), (18, 8), (21, 8), (21, 1)])
def test_sharded_row_blocks_and_diagonal(sc, log2n, world, blocks, defer, diag):
    """the row stage in row blocks (what an overlapped corner turn needs), with and without the deferred second pass, and the
    rank's own block written in place or travelling through the exchange: always the same transform"""
    full_in, got, back, root = _simulate(sc, log2n, world, seed=13, blocks=blocks, defer=defer, diag_in_place=diag)
    n = 1 << log2n
    assert got == po.C.ntt(root, full_in, n)
    assert back == full_in


def test_sharded_2p24_matches_single_gpu(sc):
    """BASELINE target size: 2^24 over 8 (simulated) ranks == the single-GPU transform, and round trip."""
    full_in, got, back, root = _simulate(sc, 24, 8, seed=12)
    n = 1 << 24
    x = sc.DeviceVector.from_bytes(full_in)
    y = sc.DeviceVector(n)
    sc._check(sc.lib().sc_ntt_dev(x.ptr, y.ptr, n, sc.fe_bytes(root), 0, None))
    sc.synchronize()
    assert y.to_bytes() == got
    assert back == full_in


def test_batched_entry_points_vs_oracle(sc):
    dev = torch.device("cuda", 0)
    lib = sc.lib()
    for kind, loglen, logbatch in [(0, 5, 3), (0, 11, 4), (0, 12, 6), (1, 5, 3), (1, 11, 4), (1, 12, 6), (1, 8, 0), (0, 9, 0)]:
        ln, bt = 1 << loglen, 1 << logbatch
        host = synth.synth_packed(40 + loglen + kind, ln * bt)
        src = torch.from_numpy(host.view(np.int64)).to(dev)
        dst = torch.empty_like(src)
        root = po.primitive_nth_root(ln)
        sc._check(lib.sc_ntt_batch_dev(src.data_ptr(), dst.data_ptr(), ln, bt, kind, sc.fe_bytes(root), None))
        sc.synchronize()
        got = dst.cpu().numpy().view(np.uint64)
        if kind == 0:
            m = host.reshape(ln, bt, 2)
            exp = np.stack([np.frombuffer(po.C.ntt(root, np.ascontiguousarray(m[:, c]).tobytes(), ln), dtype=np.uint64).reshape(ln, 2) for c in range(bt)], axis=1)
        else:
            m = host.reshape(bt, ln, 2)
            exp = np.stack([np.frombuffer(po.C.ntt(root, m[r].tobytes(), ln), dtype=np.uint64).reshape(ln, 2) for r in range(bt)], axis=1)
        assert got.reshape(-1).tobytes() == exp.tobytes(), (kind, loglen, logbatch)
    # outer twiddle
    rows, cols, order = 64, 32, 1 << 12
    host = synth.synth_packed(77, rows * cols)
    buf = torch.from_numpy(host.view(np.int64)).to(dev)
    root = po.primitive_nth_root(order)
    scale = 12345678901234567890
    sc._check(lib.sc_twiddle_matrix_dev(buf.data_ptr(), rows, cols, 0, 32, sc.fe_bytes(root), order, sc.fe_bytes(scale), None))
    sc.synchronize()
    got = synth.unpack_ints(buf.cpu().numpy().tobytes())
    ints = synth.unpack_ints(host.tobytes())
    exp = [ints[r * cols + c] * pow(root, r * (32 + c), po.P) * scale % po.P for r in range(rows) for c in range(cols)]
    assert got == exp


def test_bench_sharded_path_under_torchrun_one_rank():
    """bench.py's N > 1 code path (RCCL process group, all_to_all_single on the bench stream, max-over-ranks timing,
    JSON line) launched exactly like the driver launches it, with a world of one rank on this box's single GPU."""
    import json
    import os
    import socket
    import subprocess
    import sys
    from conftest import REPO
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(REPO, "bench.py"), "--gpus", "1", "--force-sharded", "--log2n", "20", "--steps", "5", "--warmup", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, MASTER_ADDR="127.0.0.1"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["config"]["roundtrip_bit_exact"] is True and out["n_gpus"] == 1 and out["value"] > 0
    assert "fourstep" in out["config"]["workload"]
    assert "nothing to exchange" in out["config"]["corner_turn"]            # a world of one rank moves no bytes
    # the same with the rank's own block pushed through the collectives: torch.distributed over RCCL and the library's own RCCL
    # communicator (sc_comm_init / sc_fourstep_run_dev), one exchange and row blocks, are all probed and must all be correct
    r = subprocess.run(cmd + ["--force-diag-exchange"], capture_output=True, text=True, timeout=600, env=dict(os.environ, MASTER_ADDR="127.0.0.1"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["config"]["roundtrip_bit_exact"] is True
    probes = out["config"]["corner_turn"]
    assert "torch.distributed, one blocking exchange" in probes and "torch.distributed, 4 asynchronous row blocks" in probes
    assert "library RCCL communicator, one exchange" in probes and "library RCCL communicator, 4 row blocks" in probes, probes + r.stderr[-2000:]


@pytest.mark.parametrize("world", [2, 4])
def test_ranks_share_one_gpu(world):
    """Several PROCESSES, each driving the HIP engine on its slab (all on GPU 0), exchanging through gloo with a host-staged
    all-to-all: everything of the N > 1 path except RCCL itself, against the oracle's transform of the full vector."""
    import os
    import socket
    import subprocess
    import sys
    from conftest import REPO
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(REPO, "tests", "sharded_gpu_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.count("ok") == world


@pytest.mark.parametrize("case,env", [("fine_export_fails", {"STARKCORE_TEST_FINE_EXPORT_FAILS": "1"}),
                                      ("fine_import_fails", {"STARKCORE_TEST_OPEN_FAILS_KIND": "1"}),
                                      ("nothing_imports", {"STARKCORE_TEST_OPEN_FAILS_KIND": "0", "STARKCORE_TEST_FINE_EXPORT_FAILS": "1"}),
                                      ("late_peer", {"STARKCORE_IPC_BARRIER_SPINS": "20000"})])
def test_direct_store_faults_end_in_a_correct_transform(case, env):
    """the direct-store corner turn must be impossible to ignore when it fails and easy to survive.  Two
    processes on GPU 0 (real HIP IPC handles); the environment makes one thing fail per case: the fine-grained export (the region
    becomes coarse-grained), the import of fine-grained regions on the peers (second attempt: coarse-grained), every import (the
    collective exchange), a peer that is 1.5 s late for a flag barrier that waits ~50 ms (the barrier's rank sees the time-out in its
    pinned status word, every later direct-store transform of the plan raises DirectStoreTimeout -- sticky --, a collective
    fall_back_to_exchange puts both ranks on the exchange).  Every case ends in a transform equal to the oracle's."""
    import os
    import socket
    import subprocess
    import sys
    from conftest import REPO
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    full = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1", **env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(REPO, "tests", "sharded_gpu_worker.py"), "direct_faults", case]
    r = subprocess.run(cmd, env=full, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert r.stdout.count("ok") == 2


def test_sharded_fast_stark_one_rank(sc):
    """sharded_stark.ShardedFastStark at world 1 (no process group): the same proof bytes as fast_stark.FastStark.prove from the
    same random bytes (worlds of 2 and 4 ranks: test_ranks_share_one_gpu)."""
    import sharded_gpu_worker
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        assert sharded_gpu_worker.stark_check(0, 1, dev)


def test_sharded_fast_stark_reproduces_the_reference_proofs_one_rank(sc, monkeypatch):
    """The REFERENCE's golden Rescue-Prime proofs (tests/golden/fast_stark.json) through sharded_stark.ShardedFastStark with the HIP
    engine at world 1 and every division forced through the sharded route: the AIR there uses the variable X (round constants)
    and has one constraint per register, so the value-domain transition quotients (one set of point values, several constraints,
    X evaluated on the coset) are pinned to the reference byte for byte."""
    import hashlib
    import random
    import fast_stark
    import sharded_stark
    from algebra import Field, FieldElement
    from conftest import load_golden
    from workload_rescue_prime import RescuePrime
    monkeypatch.setattr(sharded_stark.ShardedFastStark, "MIN_SHARDED_LOG2", 4)
    taken = []
    real = sharded_stark.ShardedFastStark._coset_divide
    monkeypatch.setattr(sharded_stark.ShardedFastStark, "_coset_divide", lambda self, *a, **k: (taken.append(1), real(self, *a, **k))[1])
    field = Field.main()
    rp = RescuePrime()
    dev = torch.device("cuda", 0)
    genuine = fast_stark.os.urandom
    try:
        for rec in load_golden("fast_stark.json")["runs"]:
            rng = random.Random(rec["urandom_seed"])
            fast_stark.os.urandom = lambda k, rng=rng: bytes(rng.getrandbits(8) for _ in range(k))
            input_element = FieldElement(int(rec["input"]), field)
            output_element = rp.hash(input_element)
            stark = sharded_stark.ShardedFastStark(field, rec["expansion_factor"], rec["num_colinearity_checks"], rec["security_level"], rp.m, rp.N + 1, 0, 1, dev)
            tz, layer, root = stark.preprocess()
            assert root.hex() == rec["zerofier_root"]
            del taken[:]
            proof = stark.prove(rp.trace(input_element), rp.transition_constraints(stark.omicron), rp.boundary_constraints(output_element), tz, layer)
            assert hashlib.sha256(proof).hexdigest() == rec["proof_sha256"]
            assert len(taken) == rp.m                      # only the boundary quotients went through _coset_divide: the transition quotients took the value-domain route
    finally:
        fast_stark.os.urandom = genuine


def test_a_proof_leaves_no_device_memory_to_the_cycle_collector(sc):
    """Everything a proof allocates on the device (trees of gigabytes at a 2^24 domain) must go back to the pool when the proof's
    objects die -- by reference counting, not whenever the cycle collector gets round to it: a tree that waits in a cycle makes the
    next proof miss the pool, and a 2 GB hipMalloc costs 60 ms (seen as 300-800 ms outliers, profiles/r04)."""
    import gc
    import workloads
    import starkcore
    from fast_stark import DeviceTrace, FastStark
    from sharded_stark import ShardedFastStark
    field, T, packed, air, boundary = workloads.synthetic_stark_instance(14, 40)
    trace = DeviceTrace.from_packed(packed, field)
    dev = torch.device("cuda", 0)
    provers = [ShardedFastStark(field, 4, 40, 80, 2, T, 0, 1, dev), FastStark(field, 4, 40, 80, 2, T)]
    setups = [p.preprocess(device_resident=True) for p in provers]
    for p, (tz, committed, root) in zip(provers, setups):
        p.prove(trace, air, boundary, tz, committed)
    gc.collect()
    gc.disable()
    try:
        gc.set_debug(gc.DEBUG_SAVEALL)
        for p, (tz, committed, root) in zip(provers, setups):
            p.prove(trace, air, boundary, tz, committed)
        gc.collect()
        held = [o for o in gc.garbage if isinstance(o, (starkcore.MerkleTree, starkcore.DeviceVector, starkcore.DeviceCodeword, torch.Tensor))]
        assert not held, [type(o).__name__ for o in held][:10]
    finally:
        gc.set_debug(0)
        gc.garbage.clear()
        gc.enable()


def test_bench_stark_prove_workload(sc):
    """`bench.py --workload stark_prove`: BASELINE configs[4] as a prover, on one rank and on two ranks sharing the GPU -- the same
    proof (randomness is rank 0's; the proofs differ between RUNS, so only the verdicts are compared), accepted by the verifier"""
    for n in (1, 2):
        out = _run_bench(["--gpus", str(n), "--workload", "stark_prove", "--log2n", "14", "--steps", "1", "--warmup", "1"])
        assert out["metric"] == "stark_prove_ms" and out["n_gpus"] == n and out["value"] > 0
        c = out["config"]
        assert c["verify_accepts"] is True and c["same_proof_on_every_rank"] is True and c["proof_bytes"] > 10000


def test_sharded_fri_hip_engine_matches_reference_proofs(sc):
    """ShardedFri with the HIP engine (slab fold kernel, tree levels, tree from digests, openings) on one rank: the proof
    must be byte-identical to the reference's golden Fri.prove for every slab shape (the multi-rank orchestration is
    covered under gloo in tests/test_sharded_cpu.py)."""
    import hashlib
    from conftest import load_golden
    from sharded import ShardedFri
    from algebra import Field
    from fri import Fri
    from ip import ProofStream
    dev = torch.device("cuda", 0)
    field = Field.main()
    g = load_golden("fri.json")
    for rec in g["prove_synth"]:
        N = 1 << rec["logN"]
        om = field.primitive_nth_root(N)
        coeffs = sc.DeviceVector.from_bytes(synth.synth_packed(rec["coeff_seed"], N // 4).tobytes())
        cwv = sc.DeviceVector(N)
        sc._check(sc.lib().sc_coset_evaluate_dev(coeffs.ptr, N // 4, sc.fe_bytes(po.GENERATOR), sc.fe_bytes(om.value), N, cwv.ptr, None))
        sc.synchronize()
        cw = torch.from_numpy(np.frombuffer(cwv.to_bytes(), dtype=np.int64).reshape(N, 2).copy()).to(dev)
        for logR in sorted({1, 3, rec["logN"] // 2, rec["logN"] - 1, rec["logN"]}):
            R = 1 << logR
            fr = Fri(field.generator(), om, N, rec["expansion_factor"], rec["num_colinearity_tests"])
            # under torch's null stream the engine runs on the library's stream and folds with a kernel of its own; on a side
            # stream (how the sharded prover runs) a round's fold happens in the leaf stage of the next local subtree
            # local_tail: the codeword is never gathered early / gathered half way through the rounds / the default (2^16: from
            # the first round on for these records) -- on a side stream the rounds after the
            # gather are ONE library call (sc_fri_commit_dev), under the null stream the Python loop
            for side_stream, tail in ((False, 0), (True, 0), (False, N >> 2), (True, N >> 2), (False, None), (True, None)):
                ps = ProofStream()
                if side_stream:
                    torch.cuda.synchronize()
                    with torch.cuda.stream(torch.cuda.Stream(device=dev)):
                        top = ShardedFri(fr, R, 0, 1, dev, local_tail=tail).prove(cw.reshape(N // R, R, 2), ps)
                    torch.cuda.synchronize()
                else:
                    top = ShardedFri(fr, R, 0, 1, dev, local_tail=tail).prove(cw.reshape(N // R, R, 2), ps)
                ser = ps.serialize()
                assert top == rec["top_level_indices"], (rec["logN"], R, side_stream, tail)
                assert hashlib.sha256(ser).hexdigest() == rec["serialized_sha256"], (rec["logN"], R, side_stream, tail)


def test_sharded_lde_then_sharded_fri_one_rank(sc):
    """The whole sharded polynomial core on one rank with the HIP engines: coefficients -> ShardedNtt.coset_evaluate
    (slab codeword) -> ShardedFri.prove, against the reference's golden proof for that LDE (tests/golden/fri.json)."""
    import hashlib
    from conftest import load_golden
    from sharded import ShardedNtt, ShardedFri
    from algebra import Field
    from fri import Fri
    from ip import ProofStream
    dev = torch.device("cuda", 0)
    field = Field.main()
    rec = [r for r in load_golden("fri.json")["prove_synth"] if r["logN"] == 12][0]
    N = 1 << rec["logN"]
    om = field.primitive_nth_root(N)
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        eng = ShardedNtt(rec["logN"], om.value, 0, 1, dev)
        coeffs = torch.from_numpy(synth.synth_packed(rec["coeff_seed"], N // 4).view(np.int64).copy()).to(dev)
        slab = torch.empty(eng.local_shape(False), dtype=torch.int64, device=dev)
        eng.coset_evaluate(coeffs, po.GENERATOR, slab)
    torch.cuda.synchronize()
    assert hashlib.sha256(slab.cpu().numpy().tobytes()).hexdigest() == rec["codeword_sha256"]       # world 1: slab == natural order
    fr = Fri(field.generator(), om, N, rec["expansion_factor"], rec["num_colinearity_tests"])
    ps = ProofStream()
    top = ShardedFri(fr, eng.n1, 0, 1, dev).prove(slab, ps)
    assert top == rec["top_level_indices"]
    assert hashlib.sha256(ps.serialize()).hexdigest() == rec["serialized_sha256"]


def _run_bench(args, timeout=900, env_extra=None):
    """bench.py from a BARE shell (no RANK / WORLD_SIZE in the environment), exactly as the driver's `python3 bench.py --gpus N`"""
    import json
    import os
    import subprocess
    import sys
    from conftest import REPO
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def test_bench_headline_line_is_the_contract_and_matches_the_reference(sc):
    """the N = 1 headline exactly as the driver runs it (BASELINE configs[1]): the JSON contract's keys, `roofline` and -- with
    cpu_baseline switched off here for time -- a forward transform whose SHA-256 equals the one the reference's code/ntt.py produced
    for the same input (tests/golden/ntt_big.json)"""
    out = _run_bench(["--gpus", "1", "--steps", "5", "--warmup", "2", "--no-extras", "--no-cpu-baseline"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in out, key
    assert out["metric"] == "ntt_field_elements_per_sec" and out["n_gpus"] == 1 and out["steps"] == 5 and out["dtype"] == "u128"
    assert out["config"]["workload"] == "ntt_fwd_inv_2^20_1gpu" and out["config"]["roundtrip_bit_exact"] is True
    assert out["config"]["forward_sha256_equals_reference_output"] is True
    r = out["roofline"]
    cols = out["config"]["columns_per_step"]
    assert cols == 64 and out["config"]["elements_per_step"] == 2 * cols << 20           # one step = one batch of columns (sc_ntt_columns_dev)
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["alg_bytes_per_launch"] == 32 * cols * (1 << 20) / out["config"]["passes_per_transform"]
    assert r["traffic"] is None or r["traffic"] > r["alg_bytes_per_launch"]
    assert abs(out["value"] - 2 * cols * (1 << 20) * 5 / (out["ms_per_step"] * 5e-3)) < 1e-3 * out["value"]
    assert "one_column_at_a_time" not in out                                                # a side leg, off with --no-extras
    # ... and --columns 1 makes that the step again
    out1 = _run_bench(["--gpus", "1", "--steps", "5", "--warmup", "2", "--no-extras", "--no-cpu-baseline", "--columns", "1"])
    assert out1["config"]["columns_per_step"] == 1 and out1["config"]["forward_sha256_equals_reference_output"] is True
    assert "one_column_at_a_time" not in out1 and out1["roofline"]["traffic"] > out1["roofline"]["alg_bytes_per_launch"]
    assert abs(out1["value"] - 2 * (1 << 20) * 5 / (out1["ms_per_step"] * 5e-3)) < 1e-3 * out1["value"]


def test_bench_two_ranks_print_the_north_star_record(sc):
    """`python bench.py --gpus 2` times the north_star's transform -- forward + inverse at 2^24, strong scaling --
    and carries it as extras.ntt_2p24_strong with the roofline fraction, the bytes exchanged, the form of the corner turn and a
    round trip checked on every element (on this box: two ranks sharing the GPU over gloo, labelled functional)."""
    out = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--no-extras", "--no-cpu-baseline"])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["config"]["log2n"] == 24
    assert "ntt_fwd_inv_2^24_fourstep_2gpu" == out["config"]["workload"] and "north_star" in out["config"]["series"]
    rec = out["extras"]["ntt_2p24_strong"]
    assert rec["log2n"] == 24 and rec["n_gpus"] == 2 and rec["roundtrip_bit_exact"] is True and "all 2^24" in rec["roundtrip_check"]
    assert 0 < rec["frac"] < 1 and rec["elements_per_s"] > 0 and rec["bytes_sent_per_rank_per_pair"] == 2 * (1 << 23) * 16 // 2
    assert abs(rec["elements_per_s"] - out["value"]) < 1e-6 * out["value"]
    assert rec["corner_turn"] == out["config"]["corner_turn"]
    # the first multi-GPU run explains itself -- per-stage times (max over ranks), bytes per peer, the probes of every
    # form of the corner turn (the direct-store form among them), the split, RCCL version and peer-access matrix, and the OTHER
    # member of strong / weak in the same run
    for direction in ("forward", "inverse"):
        st = out["roofline"]["stages_us"][direction]
        assert st["cols_us"] > 0 and st["rows_us"] > 0 and st["whole_us"] > 0 and "exchange_and_waiting_us" in st
    assert out["roofline"]["stages_us"]["bytes_to_each_peer_per_transform"] == (1 << 24) // 4 * 16
    probes = out["config"]["corner_turn_probes"]
    assert probes["chosen"] in out["config"]["corner_turn"] and any("direct store" in p["form"] for p in probes["probes"])
    direct = [p for p in probes["probes"] if p["form"].startswith("direct store: ")][0]
    assert direct["available"] is True and direct["correct"] is True and direct["ms_per_pair"] > 0
    assert direct["receive_region_memory"] == "fine-grained"      # peers store into it while this GPU's kernels poll and read it
    # ... and only after a child of every rank had exported, mapped and stored across processes without taking its process down
    pre = [p for p in probes["probes"] if p["form"].startswith("direct-store pre-flight")][0]
    assert pre["passed"] is True and pre["this_rank_status"] == 0
    node = out["config"]["node"]
    assert "rccl_version" in node and node["visible_gpus"] >= 1 and len(node["can_access_peer"]) == node["visible_gpus"]
    # what a reader of the driver's SCALE record needs from the line alone: the count, the rate, which collective library carried the
    # corner turn (here: gloo, labelled; on a node with a GPU per rank the label names RCCL -- tests/test_sharded_cpu.py), the time
    # every leg of the command took and which legs the time budget dropped (none at the default budget)
    assert out["value"] > 0 and "gloo" in out["config"]["collective_backend"]
    legs = out["config"]["legs"]
    assert legs["dropped_for_the_budget"] == [] and legs["budget_s"] == 1200.0
    assert {"set_up_probes_and_headline", "stage_breakdown", "other_member_of_strong_weak", "whole_command_so_far"} <= set(legs["seconds"])
    assert "n1 = 2^" in out["config"]["split"]
    other = out["extras"]["ntt_other_scaling"]
    assert other["log2n"] == 22 and other["roundtrip_bit_exact"] is True and "weak" in other["scaling"] and other["stages_us"]["forward"]["cols_us"] > 0
    # the weak series is still there on request
    weak = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--no-extras", "--no-cpu-baseline", "--scaling", "weak"])
    assert weak["scaling"] == "weak" and weak["config"]["log2n"] == 22 and "ntt_2p24_strong" not in weak.get("extras", {})


def test_bench_time_budget_drops_the_side_legs_in_the_stated_order(sc):
    """`--budget-s`: the legs behind the headline start only while the command has used less than a fraction of its budget (other
    member of strong / weak 0.35, census 0.5, prover 0.7), every rank taking the same branch; with a budget the set-up alone
    exhausts all three are dropped and named, and the headline, its stage breakdown and the north_star record are still there."""
    out = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--budget-s", "1"])
    legs = out["config"]["legs"]
    assert legs["dropped_for_the_budget"] == ["other_member_of_strong_weak", "stark_census_sharded", "stark_prove_sharded"]
    assert "ntt_other_scaling" not in out["extras"] and "stark_census_sharded" not in out["extras"] and "stark_prove_sharded" not in out["extras"]
    assert out["extras"]["ntt_2p24_strong"]["roundtrip_bit_exact"] is True and out["roofline"]["stages_us"]["forward"]["whole_us"] > 0
    assert legs["seconds"]["whole_command_so_far"] >= legs["seconds"]["set_up_probes_and_headline"] > 1


def test_bench_bare_launch_two_ranks_and_census_parity(sc):
    """`python bench.py --gpus 2` must launch itself (one rank per GPU; on this 1-GPU box the two ranks share
    the device and exchange through gloo, labelled as such), carry the config-5 census on the sharded layout, and the census
    must produce the SAME commitments and the SAME proof whatever the world size -- and the oracle's commitment."""
    out = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--log2n", "16", "--no-cpu-baseline"])
    assert out["n_gpus"] == 2 and out["config"]["world_size"] == 2 and out["config"]["roundtrip_bit_exact"] is True
    assert "fourstep_2gpu" in out["config"]["workload"] and out["value"] > 0
    assert "gloo" in out["config"]["collective_backend"] and "NOT a scaling measurement" in out["config"]["collective_backend"]
    assert out["config"]["all_to_all_bytes_sent_per_rank_per_step"] == 2 * (1 << 15) * 16 // 2
    c2 = out["extras"]["stark_census_sharded"]
    assert "error" not in c2, c2
    pr = out["extras"]["stark_prove_sharded"]                 # configs[4] as a prover, beside the census
    assert "error" not in pr, pr
    assert pr["verify_accepts"] is True and pr["same_proof_on_every_rank"] is True and pr["world_size"] == 2 and pr["ms_per_proof"] > 0
    for k in ("lde_and_commit_ms", "coset_divide_ms", "fri_prove_ms", "openings_ms", "total_ms"):
        assert c2[k] > 0
    assert c2["world_size"] == 2 and c2["all_to_all_bytes_sent_per_rank"] > 0
    # the same census as the timed workload on 1 and 4 ranks
    c1 = _run_bench(["--gpus", "1", "--workload", "stark_census", "--log2n", "16", "--steps", "1", "--warmup", "1"])
    c4 = _run_bench(["--gpus", "4", "--workload", "stark_census", "--log2n", "16", "--steps", "1", "--warmup", "1"])
    assert c1["metric"] == "stark_census_ms" and c1["higher_is_better"] is False and c4["n_gpus"] == 4
    s1, s4 = c1["stages_best_run"], c4["stages_best_run"]
    assert s1["proof_sha256_16"] == s4["proof_sha256_16"] == c2["proof_sha256_16"]
    assert s1["roots"] == s4["roots"] == c2["roots"] and s1["fri_rounds"] == s4["fri_rounds"] == 9
    assert s1["all_to_all_bytes_sent_per_rank"] == 0 and s4["all_to_all_bytes_sent_per_rank"] > 0
    # commitment of the first LDE against the oracle (code/fast_stark.py:104-105 on code/ntt.py:132-135)
    Nf, No = 1 << 16, 1 << 14
    coeffs = synth.synth_packed(60, No // 2).tobytes()
    lde = po.C.coset_evaluate(coeffs, No // 2, po.GENERATOR, po.primitive_nth_root(Nf), Nf)
    assert po.C.merkle_commit(lde, Nf).hex()[:16] == s1["roots"][0]


def test_bench_keeps_to_the_collectives_when_a_pre_flight_child_dies(sc):
    """What may go wrong with the direct-store corner turn between two physical GPUs is not only an error code: a store into memory
    the runtime mapped badly is a GPU memory fault, and that ends the process.  bench.py lets a child of every rank try the
    ingredients first (stark-anatomy_amd/direct_preflight.py); here rank 1's child dies by SIGABRT, as a fault would end it --
    every rank leaves the direct-store forms alone and the job still prints its line, measured through the collectives."""
    out = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--no-extras", "--no-cpu-baseline", "--scaling", "weak"],
                     env_extra={"STARKCORE_TEST_PREFLIGHT_DIES": "1", "STARKCORE_PREFLIGHT_WAIT_S": "6"})
    probes = out["config"]["corner_turn_probes"]
    pre = [p for p in probes["probes"] if p["form"].startswith("direct-store pre-flight")][0]
    assert pre["passed"] is False
    assert not any(p["form"].startswith("direct store") for p in probes["probes"]) and "direct store" not in probes["chosen"]
    assert out["config"]["roundtrip_bit_exact"] is True and out["value"] > 0


def test_bench_measures_again_when_the_direct_store_fails_in_the_timed_run(sc):
    """The direct-store corner turn has never run between two physical GPUs; if it passes its probe and the timed run's round trip
    (or a flag barrier) then fails, the bench must not end without a number: it discards the form on every rank together and
    measures again through the collectives.  The failure is injected after the timed windows (BENCH_INJECT_DIRECT_STORE_FAULT)."""
    out = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--no-extras", "--no-cpu-baseline", "--scaling", "weak"],
                     env_extra={"BENCH_INJECT_DIRECT_STORE_FAULT": "1"})
    probes = out["config"]["corner_turn_probes"]
    assert "direct store" in probes["discarded_after_timed_run"] and "direct store" not in probes["chosen"]
    assert any("direct store" in p["form"] for p in probes["probes"])            # its probe times stay in the record
    assert out["config"]["roundtrip_bit_exact"] is True and out["value"] > 0 and probes["chosen"] in out["config"]["corner_turn"]
