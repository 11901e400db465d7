"""N > 1 path on CPU: world_size-2 (and 4) gloo jobs running the four-step sharded NTT orchestration."""
import os
import socket
import subprocess
import sys

import pytest

from conftest import REPO


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_fourstep_gloo(world):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", env["MASTER_PORT"], os.path.join(REPO, "tests", "sharded_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.count("ok") == world


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_sharded_fri_gloo_matches_reference_proofs(world):
    """ShardedFri (slab-local folds, sharded Merkle commits, collective openings) and ContiguousFri (natural layout, one
    neighbour exchange per fold) must reproduce the reference's Fri.prove byte for byte (golden SHA-256 of the serialized
    proof stream) for several slab shapes; world 8 = the node the sharding is designed for."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", env["MASTER_PORT"], os.path.join(REPO, "tests", "sharded_worker.py"), "fri"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-4000:]
    assert r.stdout.count("ok") == world


@pytest.mark.parametrize("world", [1, 2, 4])
def test_sharded_stark_prover_gloo_matches_reference_proofs(world):
    """sharded_stark.ShardedFastStark (BASELINE configs[4] as a prover: sharded LDEs, commitments, quotients, FRI, openings; rank 0's
    random bytes broadcast) on the reference's Rescue-Prime workload: every rank ends with the REFERENCE's proof (golden SHA-256 from
    code/fast_stark.py run with the same seeded random bytes).  Local computations come from the oracle; what runs for real is the
    orchestration: what is sharded, what is gathered, which collective carries what."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", env["MASTER_PORT"], os.path.join(REPO, "tests", "sharded_worker.py"), "stark"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-4000:]
    assert r.stdout.count("ok") == world


@pytest.mark.parametrize("case", ["no_gpu", "all_children_pass", "one_child_dies"])
def test_direct_store_preflight_agreement_gloo(case, tmp_path):
    """sharded_setup.direct_store_preflight (the children that try the direct-store corner turn's ingredients before any rank does): the
    ranks must end with ONE answer -- here, without a GPU, 'no' because no child can initialise the library; with a stand-in child
    that only walks through the file rendezvous, 'yes' when every child returns 0 and 'no' when one of them dies by SIGABRT."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), OMP_NUM_THREADS="1", STARKCORE_PREFLIGHT_WAIT_S="5",
               PREFLIGHT_TMP=str(tmp_path))
    if case != "no_gpu":
        env["PREFLIGHT_FAKE"] = "1"
    if case == "all_children_pass":
        env["PREFLIGHT_EXPECT"] = "pass"
    if case == "one_child_dies":
        env["PREFLIGHT_FAKE_FAILS"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", env["MASTER_PORT"], os.path.join(REPO, "tests", "preflight_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-4000:]
    assert r.stdout.count("ok rank") == 2


def test_the_line_of_a_run_with_a_gpu_per_rank_names_rccl():
    """config.collective_backend of bench.py's N > 1 line: on a node where every rank owns a GPU the process group is "nccl", which
    on ROCm IS RCCL, and the label says so; ranks sharing GPUs (every functional run on a 1-GPU box) are labelled as not a measurement"""
    from sharded_setup import collective_label
    assert collective_label("nccl", 8, 8, False) == "nccl (RCCL), 8 ranks on 8 GPUs"
    shared = collective_label("gloo", 8, 1, True)
    assert "gloo" in shared and "8 ranks sharing 1 GPU" in shared and "NOT a scaling measurement" in shared
