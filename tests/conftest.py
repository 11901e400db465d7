import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "stark-anatomy_amd")
GOLDEN = os.path.join(REPO, "tests", "golden")
# Host modules are flat (algebra, univariate, ntt, ...) exactly like the reference's code/ directory,
# so the package directory itself goes on sys.path (pickle parity needs the module name `algebra`).
for p in (PKG, REPO):
    if p not in sys.path:
        sys.path.insert(0, p)
sys.setrecursionlimit(10000)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")
    # the oracle is test infrastructure; build it on demand (gcc, <1 s)
    so = os.path.join(REPO, "oracle", "libstark_oracle.so")
    src = os.path.join(REPO, "oracle", "stark_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(REPO, "oracle")], stdout=subprocess.DEVNULL)


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden():
    return load_golden


def has_gpu():
    try:
        import starkcore
        return starkcore.device_count() > 0
    except Exception:
        return False
