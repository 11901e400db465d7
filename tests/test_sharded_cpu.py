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
