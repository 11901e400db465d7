"""Deterministic synthetic field elements (counter-based, no state).

SURVEY.md §8(d): ``lo = splitmix64(seed + 2i)``, ``hi = splitmix64(seed + 2i + 1)``,
``x_i = (hi * 2**64 + lo) mod p``.  Mirrors the distribution the reference tests draw with
``field.sample(os.urandom(17))`` (code/test_ntt.py:12) but reproducibly, and identically in
Python, numpy and C (the test oracle has its own copy).

Packed layout everywhere: one element = 16 bytes = two little-endian uint64 limbs (lo, hi),
canonical residue in [0, p).
"""
import numpy as np

P = 1 + 407 * (1 << 119)
P_LO = 1
P_HI = 0xCB80000000000000
MASK64 = (1 << 64) - 1


def splitmix64(x):
    """Scalar splitmix64 finaliser of the counter x (Python ints)."""
    z = (x + 0x9E3779B97F4A7C15) & MASK64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
    return z ^ (z >> 31)


def _splitmix64_np(x):
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def synth_packed(seed, n, start=0):
    """(n, 2) uint64 array of limbs (lo, hi) for elements start .. start+n-1 of stream `seed`."""
    i = np.arange(start, start + n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        lo = _splitmix64_np(np.uint64(seed) + np.uint64(2) * i)
        hi = _splitmix64_np(np.uint64(seed) + np.uint64(2) * i + np.uint64(1))
    # x < 2**128 < 2p, so one conditional subtraction of p canonicalises.
    ge = (hi > np.uint64(P_HI)) | ((hi == np.uint64(P_HI)) & (lo >= np.uint64(P_LO)))
    with np.errstate(over="ignore"):
        borrow = (lo < np.uint64(P_LO)) & ge
        lo = np.where(ge, lo - np.uint64(P_LO), lo)
        hi = np.where(ge, hi - np.uint64(P_HI) - borrow.astype(np.uint64), hi)
    out = np.empty((n, 2), dtype=np.uint64)
    out[:, 0] = lo
    out[:, 1] = hi
    return out


def synth_ints(seed, n, start=0):
    """Same stream as Python ints (small n)."""
    out = []
    for i in range(start, start + n):
        lo = splitmix64((seed + 2 * i) & MASK64)
        hi = splitmix64((seed + 2 * i + 1) & MASK64)
        out.append(((hi << 64) | lo) % P)
    return out


def pack_ints(values):
    """list[int] -> bytes (16 B little-endian each)."""
    return b"".join(int(v).to_bytes(16, "little") for v in values)


def unpack_ints(buf):
    """bytes / buffer -> list[int]."""
    mv = bytes(buf)
    return [int.from_bytes(mv[i:i + 16], "little") for i in range(0, len(mv), 16)]


def synthetic_air_columns(T, a=3, b=5):
    """The two registers of the synthetic configs[4] workload as Python ints: T rows of (a, b) -> (b, a*a + b) mod p from (3, 5).
    workloads.synthetic_stark_instance, the golden generator (tests/golden/make_golden.py --stark-synth, which hands the same rows to the
    reference's FastStark) and the tests all take their trace from here."""
    col_a, col_b = [], []
    for _ in range(T):
        col_a.append(a)
        col_b.append(b)
        a, b = b, (a * a + b) % P
    return col_a, col_b
