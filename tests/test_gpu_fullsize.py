"""Full-size parity for BASELINE configs[3] and configs[4]: the workloads bench.py times, checked against
the oracle at their real sizes.

configs[3]  Fri.prove on a 2^22 codeword (ef 4, 40 checks; reference code/fri.py:115-130, code/test_fri.py:36-47):
            the LDE, EVERY round's Merkle root, every folded codeword (through its root) and the last codeword against the C
            oracle; the Fiat-Shamir indices recomputed; the whole proof through Fri.verify (1 680 openings re-hashed with hashlib).
configs[4]  the FastStark call census at a 2^24 FRI domain (code/fast_stark.py:101-151): the four-step (sharded-layout) path
            and the plain single-GPU path must produce the same commitments and the same proof bytes; one commitment
            against the C oracle's LDE + Merkle tree of 2^24 leaves."""
import hashlib
import os
import sys

import pytest

from conftest import REPO
from oracle import py_oracle as po
import synth

pytestmark = pytest.mark.gpu
C = po.C


@pytest.fixture(scope="module")
def sc():
    import starkcore
    assert starkcore.device_count() > 0, "no GPU visible"
    starkcore.init()
    return starkcore


def test_fri_prove_2p22_against_oracle(sc):
    from algebra import Field
    from fri import Fri
    from ip import ProofStream
    field = Field.main()
    N, m = 1 << 22, 1 << 20
    om = field.primitive_nth_root(N)
    coeffs = synth.synth_packed(4002, m).tobytes()
    cwv = sc.DeviceVector(N)
    src = sc.DeviceVector.from_bytes(coeffs)
    sc._check(sc.lib().sc_coset_evaluate_dev(src.ptr, m, sc.fe_bytes(po.GENERATOR), sc.fe_bytes(om.value), N, cwv.ptr, None))
    sc.synchronize()
    codeword = C.coset_evaluate(coeffs, m, po.GENERATOR, om.value, N)          # oracle LDE (code/ntt.py:132-135)
    assert cwv.to_bytes() == codeword
    fr = Fri(field.generator(), om, N, 4, 40)
    assert fr.num_rounds() == 15
    ps = ProofStream()
    top = fr.prove(sc.DeviceCodeword(cwv, field), ps)
    # the commit phase replayed by the oracle: root, alpha from the transcript so far, fold (code/fri.py:56-96)
    replay = ProofStream()
    omega, offset, cur, n = om.value, po.GENERATOR, codeword, N
    for r in range(15):
        root = C.merkle_commit(cur, n)
        assert ps.objects[r] == root, "round %d root" % r
        replay.push(root)
        if r == 14:
            break
        alpha = field.sample(replay.prover_fiat_shamir()).value
        cur = C.fold(cur, n, alpha, offset, omega)
        omega, offset, n = omega * omega % po.P, offset * offset % po.P, n // 2
    assert n == 256 and [e.value for e in ps.objects[15]] == synth.unpack_ints(cur)        # last codeword in the clear (fri.py:91)
    replay.push(ps.objects[15])
    assert top == fr.sample_indices(replay.prover_fiat_shamir(), N // 2, 256, 40)
    assert len(ps.objects) == 15 + 1 + 14 * 40 * 4                                          # 14 query rounds x (40 triples + 120 paths)
    assert fr.verify(ps, []) is True                                                        # colinearity + 1 680 Merkle paths (hashlib)
    # the proof's bytes come from the library's pickler working on a DESCRIPTION of the proof (csrc/proof_pickle.h); CPython's own
    # pickle over the materialised objects -- 2 MB, ~25 000 objects, dozens of 64 KiB frames -- must give the same bytes
    import pickle
    assert pickle.dumps(list(ps.objects)) == ps.serialize()


def test_stark_census_2p24_two_paths_and_oracle(sc):
    import torch
    sys.path.insert(0, REPO)
    import workloads
    from algebra import Field
    field = Field.main()
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        times, info = workloads.sharded_census(24, 0, 1, dev, stream)
    plain = workloads.stark_census(sc, sc.lib(), field, 24)
    assert info["fri_rounds"] == plain["fri_rounds"] == 17
    assert info["roots"] == plain["roots"]
    assert info["proof_objects"] == plain["proof_objects"] == 3 + 17 + 1 + 16 * 40 * 4 + 3 * 160 * 2
    assert info["proof_sha256_16"] == plain["proof_sha256_16"]          # four-step slab path == plain path, byte for byte
    # one commitment against the oracle at full size
    Nf, No = 1 << 24, 1 << 22
    coeffs = synth.synth_packed(60, No // 2).tobytes()
    lde = C.coset_evaluate(coeffs, No // 2, po.GENERATOR, po.primitive_nth_root(Nf), Nf)
    assert C.merkle_commit(lde, Nf).hex()[:16] == info["roots"][0]


@pytest.mark.parametrize("log_fri", [22, 24])
def test_stark_prover_full_size_two_provers_one_proof(sc, log_fri):
    """BASELINE configs[4] as a real prover at its stated size (reference code/fast_stark.py:76-178): the synthetic 2-register AIR
    with a 2^(log_fri - 4)-row randomized trace resident in HBM, FRI domain 2^log_fri.  The single-GPU prover
    (fast_stark.FastStark, plain transforms) and the sharded prover at world 1 (sharded_stark.ShardedFastStark: four-step slabs,
    slab-local FRI) must produce the SAME proof bytes from the same seeded os.urandom stream, and FastStark.verify -- host
    arithmetic only: hashlib, Python ints -- accepts it and rejects it for a false boundary claim."""
    import random
    import torch
    import workloads
    import fast_stark
    from algebra import FieldElement
    from fast_stark import DeviceTrace, FastStark
    from sharded_stark import ShardedFastStark
    s = 40
    field, T, packed, air, boundary = workloads.synthetic_stark_instance(log_fri, s)
    trace = DeviceTrace.from_packed(packed, field)
    genuine = fast_stark.os.urandom

    def seeded():
        rng = random.Random(4242 + log_fri)
        fast_stark.os.urandom = rng.randbytes
    try:
        seeded()
        one = FastStark(field, 4, s, 2 * s, 2, T)
        tz, tz_codeword, root = one.preprocess(device_resident=True)
        want = one.prove(trace, air, boundary, tz, tz_codeword)
        seeded()
        many = ShardedFastStark(field, 4, s, 2 * s, 2, T, 0, 1, torch.device("cuda", 0))
        tz2, layer, root2 = many.preprocess(device_resident=True)
        got = many.prove(trace, air, boundary, tz2, layer)
    finally:
        fast_stark.os.urandom = genuine
    assert root2 == root
    assert hashlib.sha256(got).hexdigest() == hashlib.sha256(want).hexdigest() and len(got) > 2_000_000
    assert one.verify(want, air, boundary, root) is True
    # ... and CPython's pickle writes the same bytes for the object graph those bytes load as (sharing included)
    import pickle
    assert pickle.dumps(pickle.loads(want)) == want
    wrong = [(0, 0, boundary[0][2] + FieldElement(1, field))] + boundary[1:]
    assert one.verify(want, air, wrong, root) is False


@pytest.mark.parametrize("world", [2])
def test_stark_prover_full_size_two_ranks(sc, world):
    """BASELINE configs[4] at its stated size WITH world > 1 (reference code/fast_stark.py:76-178): two processes, each driving the HIP
    engine on its slabs of the 2^24 FRI domain (both on GPU 0, exchanging through gloo -- everything of the N > 1 prover except RCCL
    itself), must each end with the byte string the single-GPU prover produces from the same seeded os.urandom stream."""
    import os
    import random
    import socket
    import subprocess
    import sys
    import workloads
    import fast_stark
    from conftest import REPO
    from fast_stark import DeviceTrace, FastStark
    log_fri, s, seed = 24, 40, 5151
    field, T, packed, air, boundary = workloads.synthetic_stark_instance(log_fri, s)
    genuine = fast_stark.os.urandom
    try:
        fast_stark.os.urandom = random.Random(seed).randbytes
        one = FastStark(field, 4, s, 2 * s, 2, T)
        tz, tz_codeword, root = one.preprocess(device_resident=True)
        want = one.prove(DeviceTrace.from_packed(packed, field), air, boundary, tz, tz_codeword)
    finally:
        fast_stark.os.urandom = genuine
    want_sha = hashlib.sha256(want).hexdigest()
    del tz, tz_codeword, one
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(REPO, "tests", "stark_fullsize_worker.py"), str(log_fri), str(seed)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if "proof_sha256" in l]
    assert len(lines) == world, r.stdout[-2000:]
    for line in lines:
        words = line.split()
        assert words[words.index("proof_sha256") + 1] == want_sha, line
        assert int(words[words.index("proof_len") + 1]) == len(want) > 2_000_000
        assert words[words.index("zerofier_root") + 1] == root.hex()[:16]
    out_dir = os.path.join(REPO, "gpurun_out")
    if os.path.isdir(out_dir):                       # kept as a record of the run (profiles/r05/)
        with open(os.path.join(out_dir, "stark_prove_fri2p24_world%d_shared_gpu.txt" % world), "w") as f:
            f.write("single-GPU prover (fast_stark.FastStark): proof_sha256 %s proof_len %d\n" % (want_sha, len(want)) + "\n".join(lines) + "\n")
