"""GPU parity tests proper: the HIP path, called through the C-ABI (libstarkcore.so), against the CPU
oracle on the same seeded inputs, against golden vectors from the reference, and -- at full sizes --
through size-independent properties.  Integer work => bit-exact everywhere."""
import ctypes
import hashlib

import numpy as np

import pytest

from conftest import load_golden
from oracle import py_oracle as po
import synth

pytestmark = pytest.mark.gpu
C = po.C
P = po.P


@pytest.fixture(scope="module")
def sc():
    import starkcore
    assert starkcore.device_count() > 0, "no GPU visible: the HIP path is mandatory for these tests"
    starkcore.init()
    yield starkcore
    for k, v in (("max_tile_log", -1), ("loge", 2), ("max_col_log", -1), ("min_tiles_log", 8), ("single_pass_max_log", 11), ("max_digit_log", -1), ("xcd_remap", 1), ("direct_tw_max_log", 22), ("fixed_shapes", 1), ("wave_local", 1), ("tw_on_load", 0)):
        starkcore.set_tuning(k, v)


def packed(seed, n, start=0):
    return synth.synth_packed(seed, n, start).tobytes()


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def gpu_ntt(sc, data, n, root, inverse=0):
    out = ctypes.create_string_buffer(16 * n if n else 16)
    sc._check(sc.lib().sc_ntt(bytes(data), out, n, sc.fe_bytes(root), inverse))
    return out.raw[:16 * n]


def test_ntt_golden_reference_vectors(sc):
    g = load_golden("ntt.json")
    for which, inv in (("ntt", 0), ("intt", 1)):
        for rec in g[which]:
            n = 1 << rec["logn"]
            out = gpu_ntt(sc, packed(rec["seed"], n), n, int(rec["root"]), inv)
            assert sha(out) == rec["sha256"], (which, rec["logn"])
            if "out" in rec:
                assert [str(v) for v in synth.unpack_ints(out)] == rec["out"]


@pytest.mark.parametrize("logn", [1, 2, 5, 11, 12, 13, 15, 16, 17, 18, 20])
def test_ntt_vs_oracle(sc, logn):
    n = 1 << logn
    data = packed(500 + logn, n)
    root = po.primitive_nth_root(n)
    assert gpu_ntt(sc, data, n, root) == C.ntt(root, data, n)
    assert gpu_ntt(sc, data, n, root, 1) == C.intt(root, data, n)
    other = pow(root, 3, P)          # a different primitive n-th root
    if logn <= 16:
        assert gpu_ntt(sc, data, n, other) == C.ntt(other, data, n)


@pytest.mark.parametrize("logn,cols", [(1, 3), (5, 9), (11, 4), (12, 5), (14, 70), (16, 3), (18, 2), (20, 3), (21, 2), (22, 1)])
def test_ntt_columns_vs_oracle(sc, logn, cols):
    """sc_ntt_columns_dev: `cols` independent transforms (code/ntt.py:3-30 once per column) in one set of launches -- one, two and three
    passes, column counts that are not powers of two, short columns on the long transforms' tiles -- equal the oracle column by column,
    forward and inverse, out of place and in place."""
    import torch
    dev = torch.device("cuda", 0)
    lib = sc.lib()
    n = 1 << logn
    root = po.primitive_nth_root(n)
    rt = sc.fe_bytes(root)
    data = packed(1200 + logn, n * cols)
    x = torch.from_numpy(np.frombuffer(data, dtype=np.int64).copy()).to(dev)
    y, z = torch.empty_like(x), torch.empty_like(x)
    sc._check(lib.sc_ntt_columns_dev(x.data_ptr(), y.data_ptr(), n, cols, rt, 0, None))
    sc._check(lib.sc_ntt_columns_dev(y.data_ptr(), z.data_ptr(), n, cols, rt, 1, None))
    sc.synchronize()
    got = y.cpu().numpy().tobytes()
    for c in range(cols):
        assert got[16 * n * c:16 * n * (c + 1)] == C.ntt(root, data[16 * n * c:16 * n * (c + 1)], n), (logn, cols, c)
    assert z.cpu().numpy().tobytes() == data
    # in place, and the inverse against the oracle's own
    w = x.clone()
    torch.cuda.synchronize()                # (torch's copy and the library's stream are ordered by nothing else)
    sc._check(lib.sc_ntt_columns_dev(w.data_ptr(), w.data_ptr(), n, cols, rt, 1, None))
    sc.synchronize()
    got = w.cpu().numpy().tobytes()
    for c in range(0, cols, max(1, cols // 3)):
        assert got[16 * n * c:16 * n * (c + 1)] == C.intt(root, data[16 * n * c:16 * n * (c + 1)], n), (logn, cols, c)
    # the reference's assertions (ntt.py:10-11) and zero columns
    assert lib.sc_ntt_columns_dev(x.data_ptr(), y.data_ptr(), n, 0, rt, 0, None) == 0
    if logn >= 2:
        assert lib.sc_ntt_columns_dev(x.data_ptr(), y.data_ptr(), n, cols, sc.fe_bytes(pow(root, 2, P)), 0, None) != 0
        assert b"primitive" in lib.sc_last_error()


def test_ntt_columns_full_size_round_trip(sc):
    """BASELINE configs[1] as a batch: 66 columns of 2^20 (what bench.py times per step, and two more: a set of launches covers 2^26
    elements, so the last two columns are a second set): columns of both sets equal sc_ntt_dev's transform of that column alone, and
    the whole batch round-trips bit for bit."""
    import torch
    dev = torch.device("cuda", 0)
    lib = sc.lib()
    n, cols = 1 << 20, 66
    rt = sc.fe_bytes(po.primitive_nth_root(n))
    x = torch.from_numpy(synth.synth_packed(77, n * cols).view(np.int64).reshape(-1)).to(dev)
    y, z, one = torch.empty_like(x), torch.empty_like(x), torch.empty(2 * n, dtype=torch.int64, device=dev)
    sc._check(lib.sc_ntt_columns_dev(x.data_ptr(), y.data_ptr(), n, cols, rt, 0, None))
    sc._check(lib.sc_ntt_columns_dev(y.data_ptr(), z.data_ptr(), n, cols, rt, 1, None))
    for c in (0, 63, 64, cols - 1):
        sc._check(lib.sc_ntt_dev(x.data_ptr() + 16 * n * c, one.data_ptr(), n, rt, 0, None))
        sc.synchronize()
        assert torch.equal(one, y[2 * n * c:2 * n * (c + 1)]), c
    assert torch.equal(z, x)


@pytest.mark.parametrize("logn,cols,m", [(4, 3, 16), (10, 5, 100), (12, 9, 1024), (14, 3, 2048), (16, 6, 1 << 13), (18, 3, 1 << 15), (21, 3, 1 << 18)])
def test_coset_evaluate_columns_vs_oracle(sc, logn, cols, m):
    """sc_coset_evaluate_columns_dev: the LDE of `cols` polynomials in one set of launches (fast_coset_evaluate, code/ntt.py:132-135, once per
    column: the loop of code/fast_stark.py:100-104) -- zero padding and coset scaling per column, the pruned first pass, one to three passes,
    the configs[2] shape 2^18 -> 2^21 -- equals the oracle column by column, and sc_coset_evaluate_dev of each column alone."""
    import torch
    dev = torch.device("cuda", 0)
    lib = sc.lib()
    n = 1 << logn
    root = po.primitive_nth_root(n)
    rt, off = sc.fe_bytes(root), sc.fe_bytes(po.GENERATOR)
    data = packed(1300 + logn, m * cols)
    x = torch.from_numpy(np.frombuffer(data, dtype=np.int64).copy()).to(dev)
    y = torch.empty(2 * n * cols, dtype=torch.int64, device=dev)
    one = torch.empty(2 * n, dtype=torch.int64, device=dev)
    sc._check(lib.sc_coset_evaluate_columns_dev(x.data_ptr(), m, cols, off, rt, n, y.data_ptr(), None))
    sc.synchronize()
    got = y.cpu().numpy().tobytes()
    for c in range(cols):
        if logn <= 18 or c == cols - 1:
            assert got[16 * n * c:16 * n * (c + 1)] == C.coset_evaluate(data[16 * m * c:16 * m * (c + 1)], m, po.GENERATOR, root, n), (logn, cols, c)
        sc._check(lib.sc_coset_evaluate_dev(x.data_ptr() + 16 * m * c, m, off, rt, n, one.data_ptr(), None))
        sc.synchronize()
        assert torch.equal(one, y[2 * n * c:2 * n * (c + 1)]), (logn, cols, c)
    assert lib.sc_coset_evaluate_columns_dev(x.data_ptr(), n + 1, 1, off, rt, n, y.data_ptr(), None) != 0      # more coefficients than points


def test_transforms_in_flight_on_two_streams(sc):
    """Two independent columns transformed side by side (one per HIP stream, many times over, never waiting in between): every
    multi-pass transform keeps its intermediate vector in a buffer of ITS stream (csrc/core.hip ntt_work_buffer), so the results are
    the oracle's whatever the interleaving -- with one shared work buffer the two transforms overwrite each other's passes
    (tools/two_stream_ntt.py found it).  Sizes with two and three passes, forward, inverse and an LDE."""
    import torch
    dev = torch.device("cuda", 0)
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    lib = sc.lib()
    for logn in (14, 18):
        n = 1 << logn
        root = po.primitive_nth_root(n)
        rt = sc.fe_bytes(root)
        cols = []
        for k in range(2):
            data = packed(900 + 10 * logn + k, n)
            x = torch.from_numpy(np.frombuffer(data, dtype=np.int64).copy()).to(dev)
            cols.append((data, x, torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)))
        torch.cuda.synchronize()
        for rep in range(24):
            for k, (data, x, y, z, w) in enumerate(cols):
                p = ctypes.c_void_p(streams[k].cuda_stream)
                sc._check(lib.sc_ntt_dev(x.data_ptr(), y.data_ptr(), n, rt, 0, p))
                sc._check(lib.sc_ntt_dev(y.data_ptr(), z.data_ptr(), n, rt, 1, p))
                sc._check(lib.sc_coset_evaluate_dev(x.data_ptr(), n // 4, sc.fe_bytes(po.GENERATOR), rt, n, w.data_ptr(), p))
        torch.cuda.synchronize()
        for data, x, y, z, w in cols:
            assert y.cpu().numpy().tobytes() == C.ntt(root, data, n)
            assert z.cpu().numpy().tobytes() == data
            assert w.cpu().numpy().tobytes() == C.coset_evaluate(data[:16 * (n // 4)], n // 4, po.GENERATOR, root, n)


def test_ntt_big_golden(sc):
    try:
        g = load_golden("ntt_big.json")
    except FileNotFoundError:
        pytest.skip("no big golden")
    for rec in g["ntt"]:
        n = 1 << rec["logn"]
        assert sha(gpu_ntt(sc, packed(rec["seed"], n), n, int(rec["root"]))) == rec["sha256"], rec["logn"]


TUNINGS = [dict(max_tile_log=12, loge=3, max_digit_log=8), dict(max_tile_log=11, loge=3, max_digit_log=8, max_col_log=6), dict(max_tile_log=10, loge=2), dict(max_tile_log=12, loge=3, max_col_log=4),
           dict(max_tile_log=12, loge=3, min_tiles_log=0), dict(max_tile_log=9, loge=2, max_digit_log=5), dict(max_tile_log=12, loge=3, xcd_remap=0),
           dict(max_tile_log=12, loge=3, single_pass_max_log=12), dict(direct_tw_max_log=0), dict(direct_tw_max_log=16, max_tile_log=11), dict(fixed_shapes=0),
           # round 2: workgroup barriers everywhere / four-step tables applied on load by the next pass / at the store
           dict(wave_local=0), dict(tw_on_load=1), dict(tw_on_load=0), dict(tw_on_load=1, wave_local=0, max_tile_log=11, max_digit_log=8)]


@pytest.mark.parametrize("tune", TUNINGS)
def test_ntt_tunings_agree_with_oracle(sc, tune):
    defaults = dict(max_tile_log=-1, loge=2, max_col_log=-1, min_tiles_log=8, single_pass_max_log=11, max_digit_log=-1, xcd_remap=1, direct_tw_max_log=22, fixed_shapes=1,
                    wave_local=1, tw_on_load=0)
    defaults.update(tune)
    for k, v in defaults.items():
        sc.set_tuning(k, v)
    try:
        for logn in (12, 14, 17, 19):
            n = 1 << logn
            data = packed(600 + logn, n)
            root = po.primitive_nth_root(n)
            assert gpu_ntt(sc, data, n, root) == C.ntt(root, data, n), (tune, logn)
            assert gpu_ntt(sc, data, n, root, 1) == C.intt(root, data, n), (tune, logn)
    finally:
        for k, v in dict(max_tile_log=-1, loge=2, max_col_log=-1, min_tiles_log=8, single_pass_max_log=11, max_digit_log=-1, xcd_remap=1, direct_tw_max_log=22, fixed_shapes=1,
                         wave_local=1, tw_on_load=0).items():
            sc.set_tuning(k, v)


def test_table_cache_eviction_keeps_tables_in_use(sc):
    """the plan / power-table caches evict (256 plans, 64 power tables).  Registering more distinct roots and
    offsets than that must never free tables the running call still points to: every result stays bit-exact."""
    lib = sc.lib()
    n, m = 1 << 10, 200
    w = po.primitive_nth_root(n)
    data, coeffs = packed(4100, n), packed(4101, m)
    expect_ntt = {}
    for i in range(300):                                   # 300 distinct primitive 1024-th roots -> 300 plan entries
        root = pow(w, 2 * i + 1, P)
        got = gpu_ntt(sc, data, n, root)
        if i % 37 == 0:
            assert got == C.ntt(root, data, n), i
    for i in range(80):                                    # 80 distinct coset offsets -> 80 power-table entries, each with a fresh plan too
        off, gen = 3 + i, pow(w, 2 * (300 + i) + 1, P)
        out = ctypes.create_string_buffer(16 * n)
        sc._check(lib.sc_coset_evaluate(coeffs, m, sc.fe_bytes(off), sc.fe_bytes(gen), n, out))
        if i % 9 == 0 or i >= 76:
            assert out.raw == C.coset_evaluate(coeffs, m, off, gen, n), i
    assert gpu_ntt(sc, data, n, w) == C.ntt(w, data, n)     # the very first (long evicted) root is rebuilt on demand


@pytest.mark.parametrize("logn", [22, 24])
def test_ntt_full_size_properties(sc, logn):
    """BASELINE sizes: round trip, agreement of two different pass decompositions, DC / Nyquist sums, and the forward and
    inverse transforms element for element against the C oracle (north_star: "2^24 ... bit-exact vs code/ntt.py")."""
    n = 1 << logn
    root = po.primitive_nth_root(n)
    x = sc.DeviceVector.from_bytes(packed(700 + logn, n))
    y = sc.DeviceVector(n)
    z = sc.DeviceVector(n)
    lib = sc.lib()
    sc._check(lib.sc_ntt_dev(x.ptr, y.ptr, n, sc.fe_bytes(root), 0, None))
    sc._check(lib.sc_ntt_dev(y.ptr, z.ptr, n, sc.fe_bytes(root), 1, None))
    sc.synchronize()
    xin = x.to_bytes()
    assert z.to_bytes() == xin                      # intt(ntt(x)) == x, all n elements
    yb = y.to_bytes()
    # X[0] = sum x, X[n/2] = sum (-1)^j x_j  (exact, via numpy-free Python ints on a strided sample is not enough:
    # use the full sums)
    import numpy as np
    a = np.frombuffer(xin, dtype=np.uint64).reshape(n, 2)
    tot_even = (int(a[0::2, 0].astype(object).sum()) + (int(a[0::2, 1].astype(object).sum()) << 64)) % P
    tot_odd = (int(a[1::2, 0].astype(object).sum()) + (int(a[1::2, 1].astype(object).sum()) << 64)) % P
    X0 = int.from_bytes(yb[:16], "little")
    Xh = int.from_bytes(yb[16 * (n // 2):16 * (n // 2) + 16], "little")
    assert X0 == (tot_even + tot_odd) % P and Xh == (tot_even - tot_odd) % P
    # a different decomposition (other tile / radix schedule) must give the identical transform
    sc.set_tuning("loge", 4)
    sc.set_tuning("max_col_log", 4)
    try:
        sc._check(lib.sc_ntt_dev(x.ptr, z.ptr, n, sc.fe_bytes(root), 0, None))
        sc.synchronize()
        assert z.to_bytes() == yb
    finally:
        sc.set_tuning("loge", 2)
        sc.set_tuning("max_col_log", -1)
    # directly against the C oracle (reference code/ntt.py:3-30), forward AND inverse, at both BASELINE sizes (2^24: ~15 s of
    # oracle time per transform).  The inverse is taken of an independent vector, not of the forward output, so it is pinned by
    # the oracle and not only by the round trip.
    assert yb == C.ntt(root, xin, n)
    del yb
    w = sc.DeviceVector.from_bytes(packed(900 + logn, n))
    sc._check(lib.sc_ntt_dev(w.ptr, z.ptr, n, sc.fe_bytes(root), 1, None))
    sc.synchronize()
    assert z.to_bytes() == C.intt(root, w.to_bytes(), n)


def test_ntt_four_pass_plan_2p25(sc):
    """Largest plan shape (four passes, digits 7+6+6+6): round trip, and agreement with a different digit split."""
    n = 1 << 25
    root = po.primitive_nth_root(n)
    x = sc.DeviceVector.from_bytes(packed(725, n))
    y = sc.DeviceVector(n)
    z = sc.DeviceVector(n)
    lib = sc.lib()
    sc._check(lib.sc_ntt_dev(x.ptr, y.ptr, n, sc.fe_bytes(root), 0, None))
    sc._check(lib.sc_ntt_dev(y.ptr, z.ptr, n, sc.fe_bytes(root), 1, None))
    sc.synchronize()
    xin = x.to_bytes()
    assert z.to_bytes() == xin
    yb = y.to_bytes()
    sc.set_tuning("max_digit_log", 9)          # three passes (9+8+8) instead of four
    try:
        sc._check(lib.sc_ntt_dev(x.ptr, z.ptr, n, sc.fe_bytes(root), 0, None))
        sc.synchronize()
        assert z.to_bytes() == yb
    finally:
        sc.set_tuning("max_digit_log", -1)
    # spot check against the definition: X[1] = sum_j x_j root^j  (exact, on the host)
    import numpy as np
    a = np.frombuffer(xin, dtype=np.uint64).reshape(n, 2)
    m = 1 << 12                                  # X[k] for k = n/m * t is a length-m DFT of the m-decimated sums
    lo = a[:, 0].reshape(-1, m).astype(object).sum(axis=0)
    hi = a[:, 1].reshape(-1, m).astype(object).sum(axis=0)
    folded = [(int(l) + (int(h) << 64)) % P for l, h in zip(lo, hi)]       # s[r] = sum_q x[q*m + r]
    rm = pow(root, n // m, P)
    for t in (1, 5, m - 1):
        expect = sum(v * pow(rm, (t * r) % m, P) for r, v in enumerate(folded)) % P
        k = (n // m) * t
        assert int.from_bytes(yb[16 * k:16 * k + 16], "little") == expect


def test_ntt_errors(sc):
    data = packed(1, 8)
    with pytest.raises(AssertionError):
        gpu_ntt(sc, data, 8, po.primitive_nth_root(16))
    with pytest.raises(AssertionError):
        gpu_ntt(sc, data, 8, po.primitive_nth_root(4))
    with pytest.raises(AssertionError):
        gpu_ntt(sc, packed(1, 6), 6, po.primitive_nth_root(8))
    assert gpu_ntt(sc, data[:16], 1, 1) == data[:16]


def test_coset_evaluate(sc):
    g = load_golden("poly.json")
    lib = sc.lib()
    for rec in g["coset_evaluate"]:
        c = [int(v) for v in rec["coeffs"]] if "coeffs" in rec else synth.synth_ints(rec["seed"], rec["m"])
        out = ctypes.create_string_buffer(16 * rec["order"])
        sc._check(lib.sc_coset_evaluate(synth.pack_ints(c), len(c), sc.fe_bytes(int(rec["offset"])), sc.fe_bytes(int(rec["generator"])), rec["order"], out))
        assert sha(out.raw) == rec["sha256"], rec
    for (m, logn) in [(1, 4), (5, 13), (1 << 12, 15), (1 << 15, 18), (3000, 17), (1 << 18, 21)]:
        n = 1 << logn
        coeffs = packed(800 + logn, m)
        gen = po.primitive_nth_root(n)
        out = ctypes.create_string_buffer(16 * n)
        sc._check(lib.sc_coset_evaluate(coeffs, m, sc.fe_bytes(po.GENERATOR), sc.fe_bytes(gen), n, out))
        assert out.raw == C.coset_evaluate(coeffs, m, po.GENERATOR, gen, n), (m, logn)


def test_poly_mul_and_divide(sc):
    lib = sc.lib()
    for (la, lb, order) in [(5, 4, 16), (33, 31, 64), (301, 201, 1024), (5000, 3000, 1 << 13), (40000, 25000, 1 << 17)]:
        a, b = synth.synth_ints(900 + la, la), synth.synth_ints(901 + lb, lb)
        root = po.primitive_nth_root(order)
        n_out = la + lb - 1
        out = ctypes.create_string_buffer(16 * n_out)
        sc._check(lib.sc_poly_mul(synth.pack_ints(a), la, synth.pack_ints(b), lb, sc.fe_bytes(root), order, out, n_out))
        if la * lb <= 301 * 201:
            expect = po.schoolbook_mul(a, b)
        else:
            ca = C.ntt(root, synth.pack_ints(a) + bytes(16 * (order - la)), order)
            cb = C.ntt(root, synth.pack_ints(b) + bytes(16 * (order - lb)), order)
            expect = synth.unpack_ints(C.intt(root, C.pointwise_mul(ca, cb, order), order))[:n_out]
        prod = synth.unpack_ints(out.raw)
        assert prod == expect, (la, lb)
        # exact division of the product by `a` gives back `b` (test_ntt.py:53-70)
        quo = ctypes.create_string_buffer(16 * lb)
        sc._check(lib.sc_coset_divide(out.raw, n_out, synth.pack_ints(a), la, sc.fe_bytes(po.GENERATOR), sc.fe_bytes(root), order, quo, lb))
        assert synth.unpack_ints(quo.raw) == b, (la, lb)
    # a divisor that vanishes on the coset -> "divide by zero" (algebra.py:92): (X - g) is zero at the first coset point g*w^0
    order = 16
    root = po.primitive_nth_root(order)
    divisor = [(-po.GENERATOR) % P, 1]
    lhs = po.schoolbook_mul(divisor, synth.synth_ints(5, 9))
    quo = ctypes.create_string_buffer(16 * 9)
    with pytest.raises(AssertionError):
        sc._check(lib.sc_coset_divide(synth.pack_ints(lhs), len(lhs), synth.pack_ints(divisor), 2,
                                      sc.fe_bytes(po.GENERATOR), sc.fe_bytes(root), order, quo, 9))
    # and the library still works afterwards
    sc._check(lib.sc_coset_divide(synth.pack_ints(lhs), len(lhs), synth.pack_ints(divisor), 2,
                                  sc.fe_bytes(3), sc.fe_bytes(root), order, quo, 9))
    assert synth.unpack_ints(quo.raw) == synth.synth_ints(5, 9)


def test_coset_divide_by_a_short_divisor_direct_and_transformed(sc):
    """A divisor of at most 8 coefficients (the boundary zerofiers of fast_stark.py:93-98 have two or three) is evaluated on the coset
    point by point (short_poly_coset_kernel) instead of being scaled and transformed: the same values, hence the same quotient as with
    sc_set_tuning("small_divisor_direct", 0) and as the oracle's restatement of ntt.py:137-176 -- for exact and inexact divisions, one-
    to eight- and nine-coefficient divisors (the last takes the transform either way), orders with one, two and three passes."""
    lib = sc.lib()
    try:
        for k, (lq, ld, order) in enumerate([(9, 1, 16), (30, 2, 64), (700, 3, 1024), (3000, 8, 1 << 12), (3000, 9, 1 << 12), (50000, 3, 1 << 16), ((1 << 20) + 5, 2, 1 << 21)]):
            q, d = synth.synth_ints(3000 + k, lq), synth.synth_ints(3100 + k, ld)
            d[-1] = d[-1] or 1
            root = po.primitive_nth_root(order)
            if lq * ld <= 3000 * 9:
                lhs = po.schoolbook_mul(q, d)
            else:                                    # (exact product through the oracle's transforms)
                big = order * 2
                rb = po.primitive_nth_root(big)
                ca = C.ntt(rb, synth.pack_ints(q) + bytes(16 * (big - lq)), big)
                cb = C.ntt(rb, synth.pack_ints(d) + bytes(16 * (big - ld)), big)
                lhs = synth.unpack_ints(C.intt(rb, C.pointwise_mul(ca, cb, big), big))[:lq + ld - 1]
            for exact in (True, False):
                num = list(lhs) if exact else [(v + 1) % P for v in lhs]       # an inexact division: what comes out is still the reference's
                n_out = lq
                got = {}
                for direct in (1, 0):
                    sc.set_tuning("small_divisor_direct", direct)
                    out = ctypes.create_string_buffer(16 * n_out)
                    sc._check(lib.sc_coset_divide(synth.pack_ints(num), len(num), synth.pack_ints(d), ld, sc.fe_bytes(po.GENERATOR), sc.fe_bytes(root), order, out, n_out))
                    got[direct] = out.raw
                assert got[1] == got[0], (lq, ld, order, exact)
                if exact:
                    assert synth.unpack_ints(got[1]) == q, (lq, ld, order)
                elif order <= 1 << 12:
                    # ntt.py:159-176 with the oracle's primitives: transform both on the coset, divide value by value, come back, unscale
                    ca = C.coset_evaluate(synth.pack_ints(num), len(num), po.GENERATOR, root, order)
                    cb = C.coset_evaluate(synth.pack_ints(d), ld, po.GENERATOR, root, order)
                    back = C.scale(C.intt(root, C.pointwise_div(ca, cb, order), order), order, pow(po.GENERATOR, -1, P))
                    assert got[1] == back[:16 * n_out], (lq, ld, order)
    finally:
        sc.set_tuning("small_divisor_direct", 1)


def test_golden_multiply_divide_via_cabi_sizes(sc):
    g = load_golden("poly.json")
    rec = [r for r in g["multiply"] if r.get("order") == 1024][0]
    a, b = synth.synth_ints(rec["lhs_seed"], rec["lhs_len"]), synth.synth_ints(rec["rhs_seed"], rec["rhs_len"])
    out = ctypes.create_string_buffer(16 * rec["out_len"])
    root, order = int(rec["root"]), rec["order"]
    deg = rec["out_len"] - 1
    while deg < order // 2:
        root, order = root * root % P, order // 2
    sc._check(sc.lib().sc_poly_mul(synth.pack_ints(a), len(a), synth.pack_ints(b), len(b), sc.fe_bytes(root), order, out, rec["out_len"]))
    assert sha(out.raw) == rec["sha256"]


def test_fold(sc):
    g = load_golden("fri.json")
    lib = sc.lib()
    for rec in g["fold"]:
        if rec["kind"] == "test_fri_codeword":
            om = int(rec["omega"])
            cw = [po.evaluate(list(range(64)), pow(om, i, P)) for i in range(rec["n"])]
        else:
            cw = synth.synth_ints(rec["seed"], rec["n"])
        out = ctypes.create_string_buffer(8 * len(cw))
        sc._check(lib.sc_fri_fold(synth.pack_ints(cw), len(cw), sc.fe_bytes(int(rec["alpha"])), sc.fe_bytes(int(rec["offset"])), sc.fe_bytes(int(rec["omega"])), out))
        assert sha(out.raw) == rec["sha256"], rec["n"]
    for logn in (4, 13, 16, 20):
        N = 1 << logn
        data = packed(1000 + logn, N)
        om = po.primitive_nth_root(N)
        alpha = synth.synth_ints(77, 1)[0]
        out = ctypes.create_string_buffer(8 * N)
        sc._check(lib.sc_fri_fold(data, N, sc.fe_bytes(alpha), sc.fe_bytes(po.GENERATOR), sc.fe_bytes(om), out))
        assert out.raw == C.fold(data, N, alpha, po.GENERATOR, om), logn


def test_merkle(sc):
    g = load_golden("merkle.json")
    lib = sc.lib()
    for rec in g["commit"]:
        vals = [int(v) for v in rec["values"]] if "values" in rec else synth.synth_ints(rec["seed"], rec["n"])
        root = ctypes.create_string_buffer(64)
        sc._check(lib.sc_merkle_commit(synth.pack_ints(vals), len(vals), root))
        assert root.raw.hex() == rec["root"], len(vals)
    for rec in g["open"]:
        tree = sc.MerkleTree.from_bytes(packed(rec["seed"], rec["n"]))
        assert [d.hex() for d in tree.open(rec["index"])] == rec["path"]
    # leaf encoding edge cases: every decimal length 1..39
    vals = [0] + [10 ** k for k in range(39)] + [10 ** k - 1 for k in range(1, 39)] + [P - 1, (1 << 64) - 1, 1 << 64, (1 << 127)]
    vals = vals[:64] + synth.synth_ints(3, 128 - len(vals[:64]))
    root = ctypes.create_string_buffer(64)
    sc._check(lib.sc_merkle_commit(synth.pack_ints(vals), len(vals), root))
    assert root.raw == po.merkle_commit(vals)
    for logn in (12, 16, 18, 19, 20):             # 18..20: the wide-level launches that fuse two levels
        N = 1 << logn
        data = packed(1100 + logn, N)
        tree = sc.MerkleTree.from_bytes(data)
        levels = C.merkle_tree(data, N)
        assert tree.root == levels[-64:], logn
        idxs = [0, 1, N // 2 - 1, N // 2, N - 1, 12345 % N]
        paths = tree.open_batch(idxs)
        assert paths == [C.merkle_open(data, N, i) for i in idxs[:2]] + paths[2:]
        for i, path in zip(idxs, paths):
            # path checks out against the root with hashlib (the reference's Merkle.verify_)
            node = hashlib.blake2b(str(int.from_bytes(data[16 * i:16 * i + 16], "little")).encode()).digest()
            j = i
            for sib in path:
                node = hashlib.blake2b(node + sib).digest() if j % 2 == 0 else hashlib.blake2b(sib + node).digest()
                j >>= 1
            assert node == tree.root
    with pytest.raises(AssertionError):
        sc._check(lib.sc_merkle_commit(packed(1, 6), 6, root))


@pytest.mark.parametrize("logn", list(range(0, 15)))
def test_merkle_every_small_size(sc, logn):
    """every tree shape of the latency-bound kernels (one workgroup tail, 4-lane narrow levels, fused subtrees) against the oracle"""
    N = 1 << logn
    data = packed(1200 + logn, N)
    tree = sc.MerkleTree.from_bytes(data)
    levels = C.merkle_tree(data, N)
    assert tree.root == levels[-64:]
    if N > 1:                                       # the reference cannot open a one-leaf tree either (merkle.py:17 on the empty half)
        for i in sorted({0, N - 1, N // 3}):
            assert tree.open(i) == C.merkle_open(data, N, i)


def test_merkle_launch_shapes_do_not_change_the_tree(sc):
    """the number of levels fused per launch on wide levels is a tuning knob: every setting must give the oracle's tree"""
    N = 1 << 19
    data = packed(1250, N)
    levels = C.merkle_tree(data, N)
    try:
        for nlev in (0, 1, 2, 3, 5, 8):
            sc.set_tuning("merkle_big_nlev", nlev)
            tree = sc.MerkleTree.from_bytes(data)
            assert tree.root == levels[-64:], nlev
            assert tree.open(N - 3) == C.merkle_open(data, N, N - 3), nlev
            tree.free()
    finally:
        sc.set_tuning("merkle_big_nlev", 2)


def test_device_vector_api(sc):
    n = 1 << 14
    data = packed(1200, n)
    v = sc.DeviceVector.from_bytes(data)
    assert v.to_bytes() == data and v.to_bytes(5, 3) == data[80:128]
    idx = [0, 5, n - 1, 77, 5]
    assert v.gather(idx) == [int.from_bytes(data[16 * i:16 * i + 16], "little") for i in idx]
    out = sc.DeviceVector(n)
    sc._check(sc.lib().sc_pointwise_mul_dev(v.ptr, v.ptr, out.ptr, n, None))
    sc.synchronize()
    assert out.to_bytes() == C.pointwise_mul(data, data, n)
    sc._check(sc.lib().sc_pointwise_div_dev(out.ptr, v.ptr, out.ptr, n, None))
    assert out.to_bytes() == data
    sc._check(sc.lib().sc_scale_dev(v.ptr, out.ptr, n, sc.fe_bytes(po.GENERATOR), None))
    sc.synchronize()
    assert out.to_bytes() == C.scale(data, n, po.GENERATOR)


def test_device_field_ops(sc):
    """The device field routines (hand-selected gfx950 sequences in csrc/field_asm.cuh) vs exact integer arithmetic,
    on edge operands and 2^16 random pairs; op 6 is the portable implementation of the same Montgomery product."""
    lib = sc.lib()
    R = 1 << 128
    Rinv = pow(R, -1, P)
    edge = [0, 1, 2, P - 1, P - 2, (1 << 119), (1 << 119) - 1, (1 << 127), 407, (1 << 64) - 1, 1 << 64, (1 << 96) + 1, 0xCB800000 << 96, 511, 512]
    a = [x for x in edge for _ in edge] + synth.synth_ints(31, 1 << 16)
    b = [y for _ in edge for y in edge] + synth.synth_ints(32, 1 << 16)
    # Montgomery product accepts any a < 2^128 (lazy inputs) with b < p
    a_lazy = [R - 1, R - 2, P, P + 1, (1 << 127) + (1 << 126)] + [(v + P) % R if v + P < R else v for v in a[5:]]
    n = len(a)
    out = ctypes.create_string_buffer(16 * n)

    def run(op, xs, ys):
        sc._check(lib.sc_field_selftest(op, synth.pack_ints(xs), synth.pack_ints(ys), out, n))
        return synth.unpack_ints(out.raw)

    assert run(0, a, b) == [x * y * Rinv % P for x, y in zip(a, b)]
    assert run(0, a_lazy, b) == [x * y * Rinv % P for x, y in zip(a_lazy, b)]
    assert run(6, a_lazy, b) == [x * y * Rinv % P for x, y in zip(a_lazy, b)]
    # the two-at-a-time product of the butterflies (paired carry counters: its first odd-column v_mad relies on the second
    # operand's top limb being <= p's): any 128-bit first operand, second operand up to p - 1
    assert run(7, a, b) == [x * y * Rinv % P for x, y in zip(a, b)]
    assert run(7, a_lazy, b) == [x * y * Rinv % P for x, y in zip(a_lazy, b)]
    worst_a = ([(1 << 128) - 1, (1 << 128) - 1, 0xFFFFFFFF, (1 << 128) - (1 << 96) + 0xFFFFFFFF] * (n // 4 + 1))[:n]
    worst_b = ([P - 1, P - (1 << 32), P - 1, P - 2] * (n // 4 + 1))[:n]
    assert run(7, worst_a, worst_b) == [x * y * Rinv % P for x, y in zip(worst_a, worst_b)]
    assert run(1, a, b) == [(x + y) % P for x, y in zip(a, b)]
    assert run(2, a, b) == [(x - y) % P for x, y in zip(a, b)]
    assert run(3, a, b) == [x * y % P for x, y in zip(a, b)]
    assert run(4, a, b) == [x * pow(2, -1, P) % P for x in a]
    assert run(5, a[:4096] + [1] * (n - 4096), b)[:4096] == [po.inv(x) for x in a[:4096]]


def test_sharded_merkle_and_fold_primitives(sc):
    """sc_merkle_query_dev, sc_merkle_level_copy_dev, sc_merkle_from_digests_dev, sc_fri_fold_slab_dev against the oracle."""
    import numpy as np
    lib = sc.lib()
    N = 1 << 12
    data = packed(1500, N)
    levels = C.merkle_tree(data, N)
    vec = sc.DeviceVector.from_bytes(data)
    cw = sc.DeviceCodeword(vec, None)
    tree = cw.tree()
    # query: elements + paths in one call
    idxs = [0, 1, 77, N // 2, N - 1, 77]
    k = len(idxs)
    arr = (ctypes.c_uint64 * k)(*idxs)
    el = ctypes.create_string_buffer(16 * k)
    pa = ctypes.create_string_buffer(64 * 12 * k)
    sc._check(lib.sc_merkle_query_dev(tree._h, vec.ptr, arr, k, el, pa))
    assert el.raw == b"".join(data[16 * i:16 * i + 16] for i in idxs)
    assert [pa.raw[64 * 12 * q:64 * 12 * (q + 1)] for q in range(k)] == [b"".join(C.merkle_open(data, N, i)) for i in idxs]
    # level copy: level l of the tree == the oracle's level l
    for level in (0, 3, 11, 12):
        cnt = N >> level
        out = sc.DeviceVector(cnt * 4)                       # 64 bytes per digest = 4 elements of 16 bytes
        tree.copy_level(level, out.ptr)
        sc.synchronize()
        off = 0 if level == 0 else 2 * N - (N >> (level - 1))
        assert out.to_bytes() == levels[64 * off:64 * (off + cnt)], level
    # tree from digests: feeding level 4 of the tree reproduces the same root and upper paths
    cnt = N >> 4
    lvl = sc.DeviceVector(cnt * 4)
    tree.copy_level(4, lvl.ptr)
    sc.synchronize()
    top = sc.MerkleTree.from_digests_ptr(lvl.ptr, cnt)
    assert top.root == tree.root == levels[-64:]
    assert top.open(5) == C.merkle_open(data, N, 5 << 4)[4:]
    # slab fold == natural fold restricted to the slab's columns
    R, cols, col_base = 64, 16, 32
    rows = N // R
    alpha, omega = synth.synth_ints(88, 1)[0], po.primitive_nth_root(N)
    full = np.frombuffer(data, dtype=np.uint64).reshape(rows, R, 2)
    slab = np.ascontiguousarray(full[:, col_base:col_base + cols, :])
    src = sc.DeviceVector.from_bytes(slab.tobytes())
    dst = sc.DeviceVector(rows // 2 * cols)
    sc._check(lib.sc_fri_fold_slab_dev(src.ptr, rows, cols, R, col_base, sc.fe_bytes(alpha), sc.fe_bytes(po.GENERATOR), sc.fe_bytes(omega), dst.ptr, None))
    sc.synchronize()
    want = np.frombuffer(C.fold(data, N, alpha, po.GENERATOR, omega), dtype=np.uint64).reshape(rows // 2, R, 2)[:, col_base:col_base + cols, :]
    assert dst.to_bytes() == np.ascontiguousarray(want).tobytes()
    # the same fold with the local subtree over the folded slab built by the same call (sc_fri_fold_slab_build_dev): the fused
    # leaf stage (512 leaves here) and, with a narrower slab, the separate fold kernel in front of a small tree (128 leaves)
    for cols2 in (cols, 4):
        slab2 = np.ascontiguousarray(full[:, col_base:col_base + cols2, :])
        want2 = np.ascontiguousarray(np.frombuffer(C.fold(data, N, alpha, po.GENERATOR, omega), dtype=np.uint64).reshape(rows // 2, R, 2)[:, col_base:col_base + cols2, :]).tobytes()
        src2 = sc.DeviceVector.from_bytes(slab2.tobytes())
        dst2 = sc.DeviceVector(rows // 2 * cols2)
        built = sc.MerkleTree.from_folded_slab(src2.ptr, rows, cols2, R, col_base, sc.fe_bytes(alpha), sc.fe_bytes(po.GENERATOR), sc.fe_bytes(omega), dst2.ptr, None)
        sc.synchronize()
        assert dst2.to_bytes() == want2, cols2
        leaves = rows // 2 * cols2
        assert built.n == leaves
        assert built.root == C.merkle_tree(want2, leaves)[-64:], cols2
        assert built.open(3) == C.merkle_open(want2, leaves, 3), cols2


def test_query_multi_and_mpoly_eval(sc):
    """sc_merkle_query_multi_dev (all rounds of a FRI query phase in one round trip) and sc_mpoly_eval_dev (pointwise AIR
    evaluation, multivariate.py:75-81 at every point of a domain) against the oracle / plain modular arithmetic."""
    lib = sc.lib()
    sizes = [1 << 10, 1 << 7, 1 << 3, 2]
    datas = [packed(1300 + i, N) for i, N in enumerate(sizes)]
    vecs = [sc.DeviceVector.from_bytes(d) for d in datas]
    trees = [sc.MerkleTree.from_device(v) for v in vecs]
    reqs = [[0, 5, 1023, 77], [127], [], [1, 0]]
    n = len(sizes)
    flat = [i for r in reqs for i in r]
    el = ctypes.create_string_buffer(16 * len(flat))
    pbytes = sum(64 * (N.bit_length() - 1) * len(r) for N, r in zip(sizes, reqs))
    pa = ctypes.create_string_buffer(pbytes)
    sc._check(lib.sc_merkle_query_multi_dev(n, (ctypes.c_void_p * n)(*[t._h for t in trees]), (ctypes.c_void_p * n)(*[v.ptr for v in vecs]),
                                            (ctypes.c_uint64 * len(flat))(*flat), (ctypes.c_uint64 * n)(*[len(r) for r in reqs]), el, pa))
    eo = po_ = 0
    for data, N, r in zip(datas, sizes, reqs):
        d = N.bit_length() - 1
        for i in r:
            assert el.raw[eo:eo + 16] == data[16 * i:16 * i + 16]
            assert pa.raw[po_:po_ + 64 * d] == b"".join(C.merkle_open(data, N, i))
            eo += 16
            po_ += 64 * d
    assert eo == 16 * len(flat) and po_ == pbytes
    with pytest.raises(sc.StarkCoreError):
        sc._check(lib.sc_merkle_query_multi_dev(1, (ctypes.c_void_p * 1)(trees[3]._h), (ctypes.c_void_p * 1)(vecs[3].ptr), (ctypes.c_uint64 * 1)(2),
                                                (ctypes.c_uint64 * 1)(1), el, pa))
    # pointwise multivariate evaluation
    import random
    rng = random.Random(11)
    nvars, npts, nterms = 4, 300, 25
    vals = [synth.synth_ints(1400 + j, npts) for j in range(nvars)]
    terms = [(tuple(rng.randrange(4) for _ in range(nvars)), rng.randrange(P)) for _ in range(nterms)] + [((0,) * nvars, 5), ((9, 0, 0, 1), P - 1)]
    dv = sc.DeviceVector.from_bytes(b"".join(synth.pack_ints(v) for v in vals))
    out = sc.DeviceVector(npts)
    exps = bytes(e for k, _ in terms for e in k)
    sc._check(lib.sc_mpoly_eval_dev(dv.ptr, nvars, npts, exps, synth.pack_ints([c for _, c in terms]), len(terms), out.ptr, None))
    want = []
    for i in range(npts):
        acc = 0
        for k, c in terms:
            t = c
            for j, e in enumerate(k):
                t = t * pow(vals[j][i], e, P) % P
            acc = (acc + t) % P
        want.append(acc)
    assert synth.unpack_ints(out.to_bytes()) == want
    # the same with turned variables (sc_mpoly_eval_rot_dev): variable 2 = variable 0 three places on, variable 3 = variable 1 one
    # place on (mod n), variable 4 in no term and marked absent -- its place is never written and never read
    n2, nv2 = 256, 5
    stored = [synth.synth_ints(1500 + j, n2) for j in range(2)]
    terms2 = [((1, 0, 2, 0, 0), 7), ((0, 3, 0, 1, 0), P - 2), ((2, 1, 1, 1, 0), rng.randrange(P)), ((0, 0, 0, 0, 0), 11)]
    dv2 = sc.DeviceVector(nv2 * n2)
    sc._check(lib.sc_vec_upload(dv2._h, 0, b"".join(synth.pack_ints(v) for v in stored), 2 * n2))
    out2 = sc.DeviceVector(n2)
    exps2 = bytes(e for k, _ in terms2 for e in k)
    src = (ctypes.c_uint32 * nv2)(0, 1, 0, 1, 0xFFFFFFFF)
    rot = (ctypes.c_uint64 * nv2)(0, 0, 3, 1, 0)
    for converted in (0, 1):                       # (the second call finds the stored variables converted already)
        sc._check(lib.sc_mpoly_eval_rot_dev(dv2.ptr, nv2, n2, exps2, synth.pack_ints([c for _, c in terms2]), len(terms2), out2.ptr, converted, src, rot, None))
        value = lambda j, i: stored[0][(i + 3) % n2] if j == 2 else stored[1][(i + 1) % n2] if j == 3 else stored[j][i]
        want2 = []
        for i in range(n2):
            acc = 0
            for k, c in terms2:
                t = c
                for j, e in enumerate(k):
                    if e:
                        t = t * pow(value(j, i), e, P) % P
                acc = (acc + t) % P
            want2.append(acc)
        assert synth.unpack_ints(out2.to_bytes()) == want2, converted
    bad_terms = bytes([0, 0, 0, 0, 1])             # a term that uses the absent variable; a turned variable pointing at a turned one
    with pytest.raises(sc.StarkCoreError):
        sc._check(lib.sc_mpoly_eval_rot_dev(dv2.ptr, nv2, n2, bad_terms, synth.pack_ints([1]), 1, out2.ptr, 1, src, rot, None))
    with pytest.raises(sc.StarkCoreError):
        sc._check(lib.sc_mpoly_eval_rot_dev(dv2.ptr, nv2, n2, exps2, synth.pack_ints([c for _, c in terms2]), len(terms2), out2.ptr, 1,
                                            (ctypes.c_uint32 * nv2)(0, 1, 3, 1, 0xFFFFFFFF), rot, None))


def test_round2_device_entry_points(sc):
    """Direct parity of the entries added in round 2: slab scaling, axpy with shift, degree, exact coset division on device
    operands, and one row block of the overlapped corner turn -- each against the oracle / plain integer arithmetic."""
    import numpy as np
    lib = sc.lib()
    # sc_scale_slab_dev: out[r][c] = in[r][c] * f^(r*row_len + col_base + c)
    rows, cols, row_len, col_base, f = 12, 8, 32, 16, po.GENERATOR
    vals = synth.synth_ints(5100, rows * cols)
    v = sc.DeviceVector.from_ints(vals)
    out = sc.DeviceVector(rows * cols)
    sc._check(lib.sc_scale_slab_dev(v.ptr, out.ptr, rows, cols, row_len, col_base, sc.fe_bytes(f), None))
    want = [vals[r * cols + c] * pow(f, r * row_len + col_base + c, P) % P for r in range(rows) for c in range(cols)]
    assert synth.unpack_ints(out.to_bytes()) == want
    # sc_vec_zero / sc_axpy_shift_dev / sc_vec_degree_dev
    acc = sc.DeviceVector.zeros(300)
    assert acc.to_bytes() == bytes(16 * 300)
    src = synth.synth_ints(5101, 100)
    w = synth.synth_ints(5102, 2)
    sv = sc.DeviceVector.from_ints(src)
    acc.axpy_shift(sv, 0, w[0])
    acc.axpy_shift(sv, 57, w[1])
    model = [0] * 300
    for j, x in enumerate(src):
        model[j] = (model[j] + w[0] * x) % P
        model[57 + j] = (model[57 + j] + w[1] * x) % P
    assert synth.unpack_ints(acc.to_bytes()) == model
    deg = ctypes.c_int64(0)
    sc._check(lib.sc_vec_degree_dev(acc.ptr, 300, ctypes.byref(deg), None))
    assert deg.value == po.degree(model) == 156
    sc._check(lib.sc_vec_degree_dev(sc.DeviceVector.zeros(64).ptr, 64, ctypes.byref(deg), None))
    assert deg.value == -1
    with pytest.raises(sc.StarkCoreError):
        acc.axpy_shift(sv, 201, 1)                                        # does not fit
    # sc_coset_divide_dev: exact quotient, exactness flag, inexact numerator
    a, b = synth.synth_ints(5103, 300), synth.synth_ints(5104, 41)
    prod = po.schoolbook_mul(a, b)
    order = 512
    root = po.primitive_nth_root(order)
    dp, db = sc.DeviceVector.from_ints(prod), sc.DeviceVector.from_ints(b)
    q = sc.DeviceVector(len(a))
    flag = ctypes.c_int(-1)
    sc._check(lib.sc_coset_divide_dev(dp.ptr, len(prod), db.ptr, len(b), sc.fe_bytes(po.GENERATOR), sc.fe_bytes(root), order, q.ptr, len(a), ctypes.byref(flag), None))
    assert flag.value == 1 and synth.unpack_ints(q.to_bytes()) == a
    assert synth.unpack_ints(q.to_bytes()) == po.fast_coset_divide(prod, b, po.GENERATOR, root, order)
    bad = list(prod); bad[3] = (bad[3] + 1) % P                             # remainder != 0
    sc._check(lib.sc_coset_divide_dev(sc.DeviceVector.from_ints(bad).ptr, len(bad), db.ptr, len(b), sc.fe_bytes(po.GENERATOR), sc.fe_bytes(root), order, q.ptr, len(a),
                                      ctypes.byref(flag), None))
    assert flag.value == 0
    # sc_ntt_rows_t_ld_dev: two row blocks of 8 rows each, chunked input, written into one [len][16] output
    ln, bt, ch = 1 << 9, 16, 4
    host = synth.synth_packed(5105, ln * bt)
    rt = po.primitive_nth_root(ln)
    m = host.reshape(bt, ln, 2)
    exp = np.stack([np.frombuffer(C.ntt(rt, m[r].tobytes(), ln), dtype=np.uint64).reshape(ln, 2) for r in range(bt)], axis=1)
    wide = sc.DeviceVector(ln * bt)
    for blk in range(2):
        part = np.ascontiguousarray(m[blk * 8:(blk + 1) * 8].reshape(8, ch, ln // ch, 2).transpose(1, 0, 2, 3))
        pv = sc.DeviceVector.from_bytes(part.tobytes())
        sc._check(lib.sc_ntt_rows_t_ld_dev(pv.ptr, wide.ptr + 16 * blk * 8, ln, 8, sc.fe_bytes(rt), ch, bt, None))
        sc.synchronize()
    assert wide.to_bytes() == exp.tobytes()


def test_async_commit_round_and_wide_query(sc):
    """sc_merkle_build_async_dev / sc_merkle_root / sc_fri_fold_commit_dev against the synchronous entries and the oracle, and
    sc_merkle_query_multi_dev over more (tree, vector) pairs than one launch takes (chunks of 32), a one-leaf tree among them."""
    lib = sc.lib()
    N = 1 << 11
    data = packed(1500, N)
    v = sc.DeviceVector.from_bytes(data)
    t_async = sc.MerkleTree.from_device_async(v)
    t_sync = sc.MerkleTree.from_device(v)
    assert t_async.root == t_sync.root == C.merkle_commit(data, N)
    assert t_async.root is t_async.root                       # fetched once
    assert t_async.open_batch([0, 5, N - 1]) == t_sync.open_batch([0, 5, N - 1])
    # many builds in flight (more than there are pinned root slots): every root still arrives
    vs = [sc.DeviceVector.from_bytes(packed(1600 + i, 256)) for i in range(8)]
    many = [sc.MerkleTree.from_device_async(vs[i % 8]) for i in range(300)]
    for i, t in enumerate(many):
        assert t.root == C.merkle_commit(packed(1600 + i % 8, 256), 256)
    del many
    unfetched = sc.MerkleTree.from_device_async(vs[0])         # freed with its root still in flight
    del unfetched
    # enqueue-only builds whose root nobody reads (local subtrees of a sharded commit): they take no root slot, so
    # more of them than there are slots can be alive while asynchronous builds still publish; their levels and (on demand)
    # their root are right
    quiet = [sc.MerkleTree.from_device_ptr_noroot(vs[i % 8].ptr, 256) for i in range(300)]
    again = sc.MerkleTree.from_device_async(vs[3])
    assert again.root == C.merkle_commit(packed(1603, 256), 256)
    assert quiet[299].open_batch([17]) == [C.merkle_open(packed(1600 + 299 % 8, 256), 256, 17)]
    assert quiet[5].root == C.merkle_commit(packed(1605, 256), 256)
    del quiet, again
    # one commit round in one call
    from algebra import Field
    field = Field.main()
    cw = sc.DeviceCodeword(v, field)
    alpha, offset, omega = field.sample(b"a"), field.generator(), field.primitive_nth_root(N)
    folded = cw.fold_commit(alpha, offset, omega, sc.DeviceVector(N // 2))
    want = C.fold(data, N, alpha.value, offset.value, omega.value)
    assert folded.tree().root == C.merkle_commit(want, N // 2)
    assert folded.vec.to_bytes() == want
    h = ctypes.c_void_p()
    assert lib.sc_fri_fold_commit_dev(v.ptr, 12, sc.fe_bytes(1), sc.fe_bytes(1), sc.fe_bytes(1), v.ptr, ctypes.byref(h), None) == sc.SC_ERR_NOT_POW2
    assert lib.sc_merkle_root(None, ctypes.create_string_buffer(64)) == -6          # SC_ERR_BAD_ARG
    # 40 pairs in one query call
    sizes = [1 << (1 + i % 9) for i in range(39)] + [1]
    datas = [packed(1700 + i, n) for i, n in enumerate(sizes)]
    vecs = [sc.DeviceVector.from_bytes(d) for d in datas]
    trees = [sc.MerkleTree.from_device_async(x) for x in vecs]
    reqs = [[(7 * i + j) % n for j in range(i % 4)] for i, n in enumerate(sizes)]
    reqs[-1] = [0, 0]
    n = len(sizes)
    flat = [i for r in reqs for i in r]
    el = ctypes.create_string_buffer(16 * len(flat))
    pbytes = sum(64 * (m.bit_length() - 1) * len(r) for m, r in zip(sizes, reqs))
    pa = ctypes.create_string_buffer(pbytes)
    sc._check(lib.sc_merkle_query_multi_dev(n, (ctypes.c_void_p * n)(*[t._h for t in trees]), (ctypes.c_void_p * n)(*[x.ptr for x in vecs]),
                                            (ctypes.c_uint64 * len(flat))(*flat), (ctypes.c_uint64 * n)(*[len(r) for r in reqs]), el, pa))
    eo = po_ = 0
    for d_, m, r in zip(datas, sizes, reqs):
        d = m.bit_length() - 1
        for i in r:
            assert el.raw[eo:eo + 16] == d_[16 * i:16 * i + 16]
            if d:
                assert pa.raw[po_:po_ + 64 * d] == b"".join(C.merkle_open(d_, m, i))
            eo += 16
            po_ += 64 * d
    assert eo == 16 * len(flat) and po_ == pbytes
    fetched = sc.query_codewords([sc.DeviceCodeword(x, field) for x in vecs[:3]], [[0, 1], [], [3]])
    assert [len(e) for e, _ in fetched] == [2, 0, 1] and fetched[2][1] == [C.merkle_open(datas[2], sizes[2], 3)]
    with pytest.raises(AssertionError):
        sc.query_codewords([sc.DeviceCodeword(vecs[0], field)], [[-1]])


def test_sample_bytes_on_device(sc):
    """sc_sample_bytes_dev == Field.sample (code/algebra.py:116-120) per byte string, for every width up to 32 bytes, edge values
    included (all ones, p, p - 1, multiples of 2^128)"""
    import random
    from algebra import Field
    field = Field.main()
    rng = random.Random(77)
    lib = sc.lib()
    for width in (1, 8, 16, 17, 24, 32):
        rows = [bytes(rng.randrange(256) for _ in range(width)) for _ in range(300)]
        rows += [bytes([255]) * width, bytes(width), (1).to_bytes(width, "big")]
        if width >= 16:
            rows += [P.to_bytes(width, "big"), (P - 1).to_bytes(width, "big")]
        if width >= 17:
            rows += [(1 << 128).to_bytes(width, "big"), ((1 << 128) + P).to_bytes(width, "big")]
        out = sc.DeviceVector(len(rows))
        sc._check(lib.sc_sample_bytes_dev(b"".join(rows), len(rows), width, out.ptr, None))
        assert synth.unpack_ints(out.to_bytes()) == [field.sample(r).value for r in rows], width
    assert lib.sc_sample_bytes_dev(b"x" * 33, 1, 33, sc.DeviceVector(1).ptr, None) == -6


def test_vec_degree_looks_at_the_top_first_and_then_at_the_rest(sc):
    """sc_vec_degree_dev (Polynomial.degree, code/univariate.py:7-17): the top 2^16 entries first, the rest only if those are zero"""
    lib = sc.lib()
    n = (1 << 17) + 77
    deg = ctypes.c_int64(0)
    for want in (-1, 0, 5, 63, 64, n - (1 << 16) - 1, n - (1 << 16), n - (1 << 16) + 1, n - 2, n - 1):
        v = sc.DeviceVector.zeros(n)
        if want >= 0:
            sc._check(lib.sc_vec_upload(v._h, want, (12345).to_bytes(16, "little"), 1))
            if want >= 3:
                sc._check(lib.sc_vec_upload(v._h, want // 2, (1).to_bytes(16, "little"), 1))     # a lower non-zero entry must not win
        sc._check(lib.sc_vec_degree_dev(v.ptr, n, ctypes.byref(deg), None))
        assert deg.value == want


def test_stream_handle_and_event_join(sc):
    """sc_stream / sc_stream_join: work enqueued on the library's stream, then consumed on ANOTHER stream with nothing but the
    event-based join in between (the host waits for the second stream only); and the library stream wrapped as a torch stream"""
    import torch
    lib = sc.lib()
    n = 1 << 20
    root = po.primitive_nth_root(n)
    data = synth.synth_packed(321, n).tobytes()
    want = po.C.ntt(root, data, n)
    src, dst = sc.DeviceVector.from_bytes(data), sc.DeviceVector(n)
    assert sc.library_stream() != 0
    other = torch.cuda.Stream()
    out = torch.empty((n, 2), dtype=torch.int64, device="cuda")
    for _ in range(3):
        sc._check(lib.sc_ntt_dev(src.ptr, dst.ptr, n, sc.fe_bytes(root), 0, None))        # asynchronous, library stream
        sc.stream_join(other.cuda_stream)
        sc._check(lib.sc_memcpy_dev(out.data_ptr(), dst.ptr, n, ctypes.c_void_p(other.cuda_stream)))   # on the other stream
        other.synchronize()
        assert out.cpu().numpy().tobytes() == want
        sc._check(lib.sc_vec_zero(dst._h))
    sc.stream_join(sc.library_stream())                                                   # joining a stream with itself: nothing to do
    ext = torch.cuda.ExternalStream(sc.library_stream())
    with torch.cuda.stream(ext):
        sc._check(lib.sc_ntt_dev(src.ptr, dst.ptr, n, sc.fe_bytes(root), 0, None))
        t = torch.empty((n, 2), dtype=torch.int64, device="cuda")
        sc._check(lib.sc_memcpy_dev(t.data_ptr(), dst.ptr, n, None))
        doubled = t.clone()                                                                # a torch op on the library's stream: ordered by construction
    ext.synchronize()
    assert doubled.cpu().numpy().tobytes() == want


def test_urandom_prefetch_then_sample(sc):
    """sc_urandom_prefetch + sc_sample_urandom_dev: the prefetched draws are used when the sizes match, a mismatch draws afresh;
    canonical residues, no repeats either way"""
    lib = sc.lib()
    p = po.P
    count = (1 << 17) + 11
    for ahead in (count, count - 5):
        sc._check(lib.sc_urandom_prefetch(ahead, 17))
        v = sc.DeviceVector(count)
        sc._check(lib.sc_sample_urandom_dev(count, 17, v.ptr, None))
        vals = sc.unpack(v.to_bytes())
        assert all(0 <= x < p for x in vals) and len(set(vals)) == count
    sc._check(lib.sc_urandom_prefetch(1000, 17))                                          # left unused: the next call (or shutdown) collects it


def test_pool_cap_and_trim(sc):
    """the free lists of the device-memory pool must be steerable -- a cap the caller sets, and a trim that hands the
    idle buffers back to the device (torch's allocator in the same process, or another rank on the same GPU, may need them)."""
    lib = sc.lib()
    # the HIP runtime this process is bound to, without torch (the ASan runs cannot initialise torch.cuda): by symbol if it was
    # loaded globally, else by the path it is mapped from
    hip = ctypes.CDLL(None)
    if not hasattr(hip, "hipMemGetInfo"):
        with open("/proc/self/maps") as maps:
            path = next(line.split()[-1] for line in maps if "libamdhip64" in line)
        hip = ctypes.CDLL(path)

    def free_bytes():
        free, total = ctypes.c_size_t(), ctypes.c_size_t()
        assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
        return free.value
    n = 1 << 22                                   # 64 MB vectors
    sc.synchronize()
    sc.set_tuning("pool_trim", 1)
    free0 = free_bytes()
    vs = [sc.DeviceVector(n) for _ in range(4)]
    for v in vs:
        v.free() if hasattr(v, "free") else None
    del vs
    sc.synchronize()
    held = free0 - free_bytes()
    assert held >= 3 * 16 * n, held               # the freed vectors sit in the pool
    sc.set_tuning("pool_trim", 1)
    assert free0 - free_bytes() < 16 * n
    sc.set_tuning("pool_cap_mb", 0)               # nothing is kept from now on
    try:
        v = sc.DeviceVector(n)
        del v
        sc.synchronize()
        w = sc.DeviceVector(16)                   # (an allocation reaps what was parked behind events)
        del w
        sc.set_tuning("pool_trim", 1)
        assert free0 - free_bytes() < 16 * n
    finally:
        sc.set_tuning("pool_cap_mb", 72 * 1024)


def test_fri_prove_in_one_call_refuses_what_it_cannot_answer(sc):
    """sc_fri_prove_dev checks its arguments before anything is enqueued: no rounds, a length that is not a power of two, more
    colinearity tests than the last codeword has points (fri.py:36-51 could never sample them), an answer buffer that is too small
    -- and a PINNED answer buffer that is smaller than the caller says, which the query kernel would write past."""
    lib = sc.lib()
    N, rounds, s = 1 << 8, 3, 4
    g, om = po.GENERATOR, po.primitive_nth_root(N)
    vec = sc.DeviceVector.from_bytes(packed(9300, N))
    need = 1 << 20

    def rc(n=N, r=rounds, tests=s, answers=None, nbytes=need):
        answers = answers if answers is not None else ctypes.create_string_buffer(need)
        roots_out, last_raw = ctypes.create_string_buffer(64 * max(r, 1)), ctypes.create_string_buffer(16 * N)
        top_out = (ctypes.c_uint64 * max(tests, 1))()
        return lib.sc_fri_prove_dev(vec.ptr, n, sc.fe_bytes(g), sc.fe_bytes(om), r, tests, b"", (ctypes.c_uint32 * 1)(), 0, 0, None, None, 0,
                                    None, None, roots_out, None, last_raw, top_out, None, answers, nbytes, None)
    assert rc() == 0                                             # the shape itself is fine ...
    assert rc(r=0) < 0 and rc(n=N - 1) < 0                       # ... these are not
    assert rc(tests=(N >> (rounds - 1)) + 1) == sc.SC_ERR_UNSUPPORTED
    assert rc(nbytes=64) < 0
    small = sc.HostBuffer(4096)                                  # pinned, 4 KiB: the openings of this proof need more
    assert rc(answers=ctypes.c_void_p(small.ptr.value), nbytes=need) < 0
    assert b"too small" in lib.sc_last_error()
    assert rc() == 0                                             # and the library is none the worse for it


@pytest.mark.parametrize("logN,s,prior_count", [(6, 2, 0), (10, 8, 3), (13, 40, 1), (18, 40, 2)])
def test_fri_prove_in_one_call_through_the_cabi(sc, logN, s, prior_count):
    """sc_fri_prove_dev (reference code/fri.py:115-130) called as a C caller would: the commit phase's roots against the oracle's
    Merkle trees of the oracle's folds (alphas recomputed with hashlib from pickle.dumps of the roots so far, ip.py:18-25), the
    top-level indices against blake2b (fri.py:36-51) over SHAKE-256 of the transcript with the last codeword pickled by CPython,
    every opened element and authentication path against the oracle's codewords and trees, the positions against fri.py:98-113 --
    with the answers in pinned memory of sc_host_alloc (written by the kernel itself) and in plain memory (two copies), with the
    commit phase's device state handed out (trees and vectors then serve the same openings again) and kept by the library."""
    import hashlib
    import pickle
    from hashlib import blake2b, shake_256
    from algebra import Field, FieldElement
    lib, field = sc.lib(), Field.main()
    N = 1 << logN
    om, g = field.primitive_nth_root(N), field.generator()
    coeffs = packed(9100 + logN, N // 4)
    cw0 = C.coset_evaluate(coeffs, N // 4, po.GENERATOR, om.value, N)
    vec = sc.DeviceVector.from_bytes(cw0)
    rounds, length = 0, N
    while length > 4 and length > 4 * s:
        rounds, length = rounds + 1, length // 2
    n_last = N >> (rounds - 1)
    prior = [bytes([i + 1]) * 64 for i in range(prior_count)]
    # what the reference computes, with the oracle's arithmetic
    codewords, roots, transcript, omega, offset = [cw0], [], list(prior), om.value, g.value
    for r in range(rounds):
        roots.append(C.merkle_commit(codewords[-1], N >> r))
        transcript.append(roots[-1])
        if r + 1 < rounds:
            alpha = field.sample(shake_256(pickle.dumps(transcript)).digest(32))
            codewords.append(C.fold(codewords[-1], N >> r, alpha.value, offset, omega))
            omega, offset = omega * omega % po.P, offset * offset % po.P
    last = [FieldElement(v, field) for v in synth.unpack_ints(codewords[-1])]
    seed = shake_256(pickle.dumps(transcript + [last])).digest(32)
    top, residues, counter = [], set(), 0
    while len(top) < s:
        i = int.from_bytes(blake2b(seed + bytes(counter)).digest(), "big") % (N // 2)
        counter += 1
        if i % n_last not in residues:
            residues.add(i % n_last)
            top.append(i)
    positions, idx, prev = [], list(top), None
    for j in range(rounds):
        half, here = (N >> j) // 2, []
        if j + 1 < rounds:
            idx = [i % half for i in idx]
            here += idx + [i + half for i in idx]
            assert j == 0 or all(c in (a, a + half) for c, a in zip(prev, idx))      # (the c of the round before is opened as this a or b)
        elif j > 0:
            here += prev                                                          # the last codeword: nothing but the c of the round before
        prev = idx
        positions.append(here)
    extra_shift = 4
    quad = sorted(top + [(i + extra_shift) % N for i in top] + [(i + N // 2) % N for i in top] + [(i + extra_shift + N // 2) % N for i in top])
    extra_data = packed(9200 + logN, N)
    extra_vec = sc.DeviceVector.from_bytes(extra_data)
    extra_tree = sc.MerkleTree.from_device(extra_vec)
    counts = [len(p) for p in positions] + [4 * s]
    depths = [(N >> j).bit_length() - 1 for j in range(rounds)] + [logN]
    total = sum(counts)
    el_bytes = (16 * total + 255) & ~255
    path_bytes = sum(64 * c * d for c, d in zip(counts, depths))
    nbytes = el_bytes + path_bytes + 8 * total

    def call(answers_ptr, keep):
        vecs = (ctypes.c_void_p * max(1, rounds - 1))()
        trees = (ctypes.c_void_p * rounds)()
        roots_out = ctypes.create_string_buffer(64 * rounds)
        alphas = (ctypes.c_uint64 * max(2, 2 * (rounds - 1)))()
        last_raw = ctypes.create_string_buffer(16 * n_last)
        top_out = (ctypes.c_uint64 * s)()
        quad_out = (ctypes.c_uint64 * (4 * s))()
        sc._check(lib.sc_fri_prove_dev(vec.ptr, N, sc.fe_bytes(g.value), sc.fe_bytes(om.value), rounds, s, b"".join(prior), (ctypes.c_uint32 * max(1, prior_count))(*map(len, prior)), prior_count,
                                       1, (ctypes.c_void_p * 1)(extra_tree._h), (ctypes.c_void_p * 1)(extra_vec.ptr), extra_shift,
                                       vecs if keep else None, trees if keep else None, roots_out, alphas if keep else None, last_raw, top_out, quad_out, answers_ptr, nbytes, None))
        assert [roots_out.raw[64 * r:64 * r + 64] for r in range(rounds)] == roots
        assert last_raw.raw == codewords[-1] and list(top_out) == top and list(quad_out) == quad
        return vecs, trees

    def check(raw):
        vo, po_ = 0, el_bytes
        where = np.frombuffer(raw[el_bytes + path_bytes:el_bytes + path_bytes + 8 * total], dtype=np.uint64).tolist()
        for j, (c, d) in enumerate(zip(counts, depths)):
            data, n, want_pos = (codewords[j], N >> j, positions[j]) if j < rounds else (extra_data, N, quad)
            assert where[vo:vo + c] == want_pos, j
            for t, i in enumerate(want_pos):
                assert raw[16 * (vo + t):16 * (vo + t + 1)] == data[16 * i:16 * i + 16], (j, t)
            for t in ([0, c // 2, c - 1] if c else []):                 # three paths per codeword against the oracle's tree
                got = raw[po_ + 64 * d * t:po_ + 64 * d * (t + 1)]
                assert got == b"".join(C.merkle_open(data, n, want_pos[t])), (j, t)
            vo += c
            po_ += 64 * c * d
    pinned = sc.HostBuffer(nbytes)
    call(pinned.ptr, keep=False)
    first = pinned.array.tobytes()
    check(first)
    plain = ctypes.create_string_buffer(nbytes)
    vecs, trees = call(ctypes.cast(plain, ctypes.c_void_p), keep=True)
    # the staged form: the same answers (the bytes between the elements and the 256-byte boundary the paths start at are nobody's)
    assert plain.raw[:16 * total] == first[:16 * total] and plain.raw[el_bytes:el_bytes + path_bytes] == first[el_bytes:el_bytes + path_bytes]
    check(plain.raw)
    # the device state that was handed out: codeword r's vector and tree
    for r in range(rounds):
        tree = sc.MerkleTree(ctypes.c_void_p(trees[r]), roots[r], N >> r)
        if r > 0:
            folded = sc.DeviceVector.adopt(vecs[r - 1], N >> r)
            assert folded.to_bytes() == codewords[r]
        assert tree.open_batch([0])[0] == C.merkle_open(codewords[r], N >> r, 0) if (N >> r) > 1 else True
    # a buffer that is too small, and one more colinearity test than the last codeword has entries
    assert lib.sc_fri_prove_dev(vec.ptr, N, sc.fe_bytes(g.value), sc.fe_bytes(om.value), rounds, s, b"", (ctypes.c_uint32 * 1)(), 0, 0, None, None, 0,
                                None, None, ctypes.create_string_buffer(64 * rounds), None, ctypes.create_string_buffer(16 * n_last), (ctypes.c_uint64 * s)(), None,
                                pinned.ptr, 16, None) == -6
    assert lib.sc_host_free(ctypes.cast(plain, ctypes.c_void_p)) == -6                          # not a buffer of sc_host_alloc


def test_vec_wrap_is_a_view_of_the_callers_memory(sc):
    """sc_vec_wrap: a vector handle over device memory the caller owns (a torch tensor's storage) -- the library reads and writes
    it in place, sc_vec_free leaves the memory alone; and a library vector seen by torch through the CUDA array interface."""
    import torch
    n = 1 << 12
    data = packed(9300, n)
    t = torch.from_numpy(np.frombuffer(data, dtype=np.int64).reshape(n, 2).copy()).cuda()
    v = sc.DeviceVector.wrap(t.data_ptr(), n, t)
    assert v.to_bytes() == data
    root = sc.MerkleTree.from_device(v).root
    assert root == C.merkle_commit(data, n)
    sc._check(sc.lib().sc_vec_zero(v._h))
    sc.synchronize()
    assert not t.any()                                        # written through the handle
    v.free()
    t.fill_(7)                                                # the memory is still the tensor's
    torch.cuda.synchronize()
    assert int(t[5, 1]) == 7
    w = sc.DeviceVector.from_bytes(data)
    view = torch.as_tensor(w, device="cuda")
    assert tuple(view.shape) == (n, 2) and view.data_ptr() == int(w.ptr)
    assert view.cpu().numpy().tobytes() == data
