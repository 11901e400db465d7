"""The two arithmetic shortcuts of the tree kernels' leaf stage (csrc/merkle.cuh: div1e9_step, the fixed-point digits of
leaf_message_lds), restated in C and checked against plain division on the host: all 10^9 values of a nine-digit group, 3 * 10^8
random and boundary dividends of the long division.  (The device code itself is pinned by the Merkle goldens in the -m gpu tests;
these programs are what the constants were chosen with.)"""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "..", "tools", "hostcheck")


@pytest.mark.parametrize("name", ["leaf_digits_check", "leaf_div1e9_check"])
def test_leaf_stage_shortcuts_on_the_host(name, tmp_path):
    exe = str(tmp_path / name)
    subprocess.check_call(["gcc", "-O2", "-o", exe, os.path.join(SRC, name + ".c")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert lines and all(l.endswith(" 0 bad") or ", 0 bad" in l for l in lines), out.stdout


def test_leaf_message_model_against_python_str(tmp_path):
    """tools/hostcheck/leaf_message_model.c walks through leaf_message_lds's steps on the host (the divisions, the fixed-point digits,
    the big-endian collection, the byte swap, the shift by the leading zeros): 0, every power of ten and its neighbours, p - 1,
    2^128 - 1 and 200 000 random values of every length must come out as Python's str() writes them (merkle.py:13-14 hashes
    `bytes(da)`, and FieldElement.__bytes__, algebra.py:56-57, is the decimal string of the value)."""
    exe = str(tmp_path / "leaf_message_model")
    subprocess.check_call(["gcc", "-O2", "-o", exe, os.path.join(SRC, "leaf_message_model.c")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0
    lines = out.stdout.splitlines()
    assert len(lines) > 200000
    for line in lines:
        value, length, text = line.split()
        assert text == str(int(value, 16)) and int(length) == len(text), line
