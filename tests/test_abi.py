"""CPU-only checks of the drop-in boundary: libstarkcore.so loads and exports exactly what
include/starkcore.h declares; the ctypes binding covers all of it; no compute is attempted."""
import ctypes
import os
import re
import subprocess

import pytest

from conftest import REPO, PKG

HEADER = os.path.join(REPO, "include", "starkcore.h")
LIB = os.path.join(PKG, "libstarkcore.so")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sc_[a-z_0-9]+)\s*\(", src)))


@pytest.fixture(scope="module")
def built_lib():
    if not os.path.exists(LIB):
        subprocess.check_call(["make", "-C", os.path.join(PKG, "csrc")])
    return ctypes.CDLL(LIB)


def test_header_symbols_exported(built_lib):
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(built_lib, n), n


def test_binding_matches_header():
    import starkcore
    assert sorted(starkcore.SIGNATURES) == declared_symbols()
    starkcore.lib()      # sets restype/argtypes for every symbol


def test_no_gpu_fails_loudly():
    import starkcore
    if starkcore.device_count() > 0:
        pytest.skip("a GPU is present")
    import ntt
    from algebra import Field
    f = Field.main()
    with pytest.raises(RuntimeError):
        ntt.ntt(f.primitive_nth_root(4), [f.one()] * 4)     # no CPU fallback: the HIP path is mandatory


def test_product_does_not_import_oracle():
    # the oracle is test infrastructure; nothing under the package may reference it
    for root, _, files in os.walk(PKG):
        for fn in files:
            if fn.endswith((".py", ".hip", ".cuh", ".h")):
                txt = open(os.path.join(root, fn), errors="replace").read()
                assert "py_oracle" not in txt and "stark_oracle" not in txt and "libstark_oracle" not in txt, fn


def test_header_names_every_tuning_key_the_library_takes():
    """sc_set_tuning's keys live in a chain of string comparisons (csrc/core.hip) and in a comment of include/starkcore.h: the
    comment must name every key the code takes (a knob nobody can find is a knob nobody uses -- or resets)"""
    import re
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    src = open(os.path.join(root, "stark-anatomy_amd", "csrc", "core.hip")).read()
    body = src[src.index("int sc_set_tuning("):]
    body = body[:body.index("\nint ", 10)]
    keys = set(re.findall(r'strcmp\(key, "([a-z_0-9]+)"\)', body)) or set(re.findall(r'"([a-z_0-9]+)"', body))
    assert len(keys) >= 15, keys
    header = open(os.path.join(root, "include", "starkcore.h")).read()
    comment = header[:header.index("int sc_set_tuning(")]
    comment = comment[comment.rindex("/*"):]
    missing = sorted(k for k in keys if '"%s"' % k not in comment)
    assert not missing, missing


def test_integration_notes_list_every_environment_variable():
    """every STARKCORE_* variable the sources read is in INTEGRATION.md's table"""
    import glob
    import re
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    found = set()
    for path in glob.glob(os.path.join(root, "stark-anatomy_amd", "csrc", "*.*")) + glob.glob(os.path.join(root, "stark-anatomy_amd", "*.py")) + [os.path.join(root, "bench.py")]:
        if path.endswith((".hip", ".h", ".cuh", ".py")):
            found |= set(re.findall(r"STARKCORE_[A-Z_0-9]+", open(path).read()))
    notes = open(os.path.join(root, "INTEGRATION.md")).read()
    missing = sorted(v for v in found if v not in notes)
    assert found and not missing, missing
