"""CPU emulation of the HIP NTT tile kernels (tests/emu/ntt_emu.cpp runs the same round body and planner
as the device code) against the oracle.  Catches index / twiddle / planner bugs without a GPU."""
import ctypes
import os
import subprocess

import pytest

from conftest import REPO
from oracle import py_oracle as po
P = po.P
import synth

EMU_DIR = os.path.join(REPO, "tests", "emu")


@pytest.fixture(scope="module")
def emu():
    so = os.environ.get("NTT_EMU_LIB") or os.path.join(EMU_DIR, "libntt_emu.so")       # (tools/sanitize.sh: an ASan + UBSan build)
    srcs = [os.path.join(EMU_DIR, "ntt_emu.cpp")] + [os.path.join(REPO, "stark-anatomy_amd", "csrc", f) for f in ("field.cuh", "ntt_tile.cuh", "ntt_plan.h")]
    if not os.environ.get("NTT_EMU_LIB") and (not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so, srcs[0]])
    lib = ctypes.CDLL(so)
    lib.emu_ntt.restype = ctypes.c_int
    lib.emu_ntt.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64,
                            ctypes.c_void_p] + [ctypes.c_int] * 7
    lib.emu_field.restype = None
    lib.emu_field.argtypes = [ctypes.c_void_p] * 3
    return lib


def run_emu(lib, data, logn, root, inverse=0, in_limit=None, offset=None, tile=12, loge=3, single=11, min_tiles=10, max_col=6, digit=8, direct=1):
    n = 1 << logn
    out = ctypes.create_string_buffer(16 * n)
    rc = lib.emu_ntt(data, out, logn, int(root).to_bytes(16, "little"), inverse, (1 << 64) - 1 if in_limit is None else in_limit,
                     None if offset is None else int(offset).to_bytes(16, "little"), tile, loge, single, min_tiles, max_col, digit, direct)
    assert rc > 0, rc
    return out.raw, rc


def test_field_host_device_code(emu):
    P = po.P
    cases = [(0, 0), (1, 1), (P - 1, P - 1), (P - 1, 1), (1 << 127, (1 << 127) + 5), (12345, P - 2), ((1 << 119) + 1, 407)]
    cases += list(zip(synth.synth_ints(91, 200), synth.synth_ints(92, 200)))
    for a, b in cases:
        out = ctypes.create_string_buffer(16 * 7)
        emu.emu_field(a.to_bytes(16, "little"), b.to_bytes(16, "little"), out)
        r = synth.unpack_ints(out.raw)
        assert r[0] == (a + b) % P and r[1] == (a - b) % P and r[2] == a * b % P
        assert r[3] == po.inv(a) and r[4] == a * po.inv(2) % P and r[5] == (-a) % P and r[6] == a * b % P


@pytest.mark.parametrize("logn", list(range(1, 13)))
def test_emu_single_pass(emu, logn):
    n = 1 << logn
    data = synth.synth_packed(40 + logn, n).tobytes()
    root = po.primitive_nth_root(n)
    for loge in (1, 2, 3, 4):
        out, npass = run_emu(emu, data, logn, root, loge=loge, single=12)
        assert npass == 1
        assert out == po.C.ntt(root, data, n), (logn, loge)
    out, _ = run_emu(emu, data, logn, root, inverse=1, single=12)
    assert out == po.C.intt(root, data, n)


# (logn, tile cap, loge, single_pass_max, min_tiles_log, max_col_log) -> forces 2-, 3- and 4-pass plans at small n
MULTI = [(6, 4, 1, 2, 0, 2, 8), (8, 6, 2, 3, 0, 3, 8), (10, 7, 2, 4, 0, 3, 8), (12, 8, 3, 5, 0, 4, 8), (13, 12, 3, 11, 10, 6, 8), (14, 12, 4, 11, 4, 6, 8),
         (16, 12, 3, 11, 10, 6, 8), (17, 12, 3, 11, 10, 6, 8), (12, 6, 2, 3, 0, 2, 4), (15, 7, 2, 3, 0, 3, 4), (9, 4, 1, 2, 0, 1, 3), (18, 10, 3, 11, 10, 6, 8),
         (12, 5, 1, 2, 0, 2, 3), (11, 6, 2, 3, 0, 2, 3), (16, 7, 2, 3, 0, 3, 4),
         # shapes with geometry-specialised kernel instantiations (logR, logC) = (8,3), (7,4), (6,5), (10,2)
         (16, 11, 2, 11, 0, 6, 8), (14, 11, 2, 11, 0, 6, 8), (12, 11, 2, 11, 0, 6, 8), (20, 12, 2, 11, 8, 4, 10),
         (18, 12, 2, 11, 8, 4, 10), (16, 12, 2, 11, 8, 4, 10)]


@pytest.mark.parametrize("cfg", MULTI)
def test_emu_multi_pass(emu, cfg):
    logn, tile, loge, single, min_tiles, max_col, digit = cfg
    n = 1 << logn
    data = synth.synth_packed(70 + logn, n).tobytes()
    root = po.primitive_nth_root(n)
    kw = dict(tile=tile, loge=loge, single=single, min_tiles=min_tiles, max_col=max_col, digit=digit)
    out, npass = run_emu(emu, data, logn, root, **kw)
    assert npass >= 2
    assert out == po.C.ntt(root, data, n), cfg
    out, _ = run_emu(emu, data, logn, root, inverse=1, **kw)
    assert out == po.C.intt(root, data, n), cfg
    # two-level twiddle lookup instead of the direct tables (0); direct tables applied at the store of the producing pass
    # instead of on load by the next one (2)
    for direct in (0, 2):
        out, _ = run_emu(emu, data, logn, root, direct=direct, **kw)
        assert out == po.C.ntt(root, data, n), (cfg, direct)
        out, _ = run_emu(emu, data, logn, root, inverse=1, direct=direct, **kw)
        assert out == po.C.intt(root, data, n), (cfg, direct)


def test_emu_pass_counts(emu):
    seen = set()
    for cfg in MULTI:
        logn, tile, loge, single, min_tiles, max_col, digit = cfg
        n = 1 << logn
        data = synth.synth_packed(1, n).tobytes()
        _, npass = run_emu(emu, data, logn, po.primitive_nth_root(n), tile=tile, loge=loge, single=single, min_tiles=min_tiles, max_col=max_col, digit=digit)
        seen.add(npass)
    assert {2, 3, 4} <= seen, seen


@pytest.mark.parametrize("cfg", [(3, 3, 12, 3, 11, 10, 6), (8, 100, 12, 3, 11, 10, 6), (10, 1000, 7, 2, 4, 0, 3), (13, 1 << 10, 12, 3, 11, 10, 6), (12, 37, 6, 2, 3, 0, 2)])
def test_emu_coset_evaluate(emu, cfg):
    logn, m, tile, loge, single, min_tiles, max_col = cfg
    n = 1 << logn
    coeffs = synth.synth_packed(300 + logn, m).tobytes()
    gen = po.primitive_nth_root(n)
    for offset in (po.GENERATOR, 2):
        out, _ = run_emu(emu, coeffs + bytes(16), logn, gen, in_limit=m, offset=offset, tile=tile, loge=loge, single=single, min_tiles=min_tiles, max_col=max_col)
        assert out == po.C.coset_evaluate(coeffs, m, offset, gen, n), cfg


# zero-padded inputs through the geometry-specialised shapes: the degenerate top stages of the first pass are pruned
# (logn, m, tile, loge, single, min_tiles, max_col, digit): blowup 8 / 4 / 2 / 16 and ragged lengths on (8,3), (7,4), (10,2), (9,3)
PRUNE = [(16, 1 << 13, 11, 2, 11, 0, 6, 8), (16, (1 << 13) - 5, 11, 2, 11, 0, 6, 8), (14, 1 << 12, 11, 2, 11, 0, 6, 8), (14, 1 << 13, 11, 2, 11, 0, 6, 8),
         (16, 1 << 12, 11, 2, 11, 0, 6, 8), (16, 3, 11, 2, 11, 0, 6, 8), (18, 1 << 15, 12, 2, 11, 8, 4, 10), (20, 1 << 17, 12, 2, 11, 8, 4, 10), (18, 40000, 12, 2, 11, 8, 4, 10)]


@pytest.mark.parametrize("cfg", PRUNE)
def test_emu_pruned_first_pass(emu, cfg):
    logn, m, tile, loge, single, min_tiles, max_col, digit = cfg
    n = 1 << logn
    coeffs = synth.synth_packed(350 + logn, m).tobytes()
    gen = po.primitive_nth_root(n)
    kw = dict(tile=tile, loge=loge, single=single, min_tiles=min_tiles, max_col=max_col, digit=digit)
    want = po.C.coset_evaluate(coeffs, m, po.GENERATOR, gen, n)
    for direct in (1, 2):
        out, npass = run_emu(emu, coeffs + bytes(16), logn, gen, in_limit=m, offset=po.GENERATOR, direct=direct, **kw)
        assert npass >= 2 and out == want, (cfg, direct)
    # plain zero padding without the coset scaling (fast_multiply's operands, code/ntt.py:51-56)
    out, _ = run_emu(emu, coeffs + bytes(16), logn, gen, in_limit=m, **kw)
    assert out == po.C.ntt(gen, coeffs + bytes(16 * (n - m)), n), cfg


def _batched_expect(data, kind, loglen, logbatch, root):
    """oracle: transform every column (kind 0) or every row with transposed output (kind 1)."""
    import numpy as np
    ln, bt = 1 << loglen, 1 << logbatch
    a = np.frombuffer(data, dtype=np.uint64)
    if kind == 0:
        m = a.reshape(ln, bt, 2)
        cols = [po.C.ntt(root, m[:, c, :].tobytes(), ln) for c in range(bt)]
        out = np.stack([np.frombuffer(c, dtype=np.uint64).reshape(ln, 2) for c in cols], axis=1)      # [len][batch][2]
    else:
        m = a.reshape(bt, ln, 2)
        rows = [po.C.ntt(root, m[r].tobytes(), ln) for r in range(bt)]
        out = np.stack([np.frombuffer(r, dtype=np.uint64).reshape(ln, 2) for r in rows], axis=1)      # [len][batch][2]
    return out.tobytes()


@pytest.mark.parametrize("cfg", [(0, 3, 2, 6, 2, 0, 3, 8), (0, 6, 4, 8, 2, 0, 4, 3), (0, 7, 3, 9, 2, 0, 3, 4), (0, 12, 2, 11, 2, 4, 6, 8), (0, 5, 0, 6, 2, 0, 3, 8),
                                 (1, 3, 2, 6, 2, 0, 3, 8), (1, 6, 4, 8, 2, 0, 4, 3), (1, 7, 3, 9, 3, 0, 3, 4), (1, 12, 2, 11, 2, 4, 6, 8), (1, 9, 1, 11, 2, 0, 6, 8),
                                 (0, 11, 4, 11, 2, 10, 6, 8), (1, 11, 4, 11, 2, 10, 6, 8),
                                 # short columns, many of them: the level-batched transforms of the subproduct tree (polytree.cuh)
                                 (0, 1, 9, 11, 2, 8, 6, 8), (0, 2, 8, 11, 2, 8, 6, 8), (0, 1, 1, 11, 2, 8, 6, 8), (0, 2, 1, 11, 2, 8, 6, 8), (0, 3, 1, 11, 2, 8, 6, 8),
                                 (0, 4, 7, 11, 2, 8, 6, 8), (0, 9, 3, 11, 2, 8, 6, 8), (0, 10, 1, 11, 2, 8, 6, 8)])
def test_emu_batched(emu, cfg):
    kind, loglen, logbatch, tile, loge, min_tiles, max_col, digit = cfg
    emu.emu_ntt_batched.restype = ctypes.c_int
    emu.emu_ntt_batched.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p] + [ctypes.c_int] * 5 + \
        [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    total = 1 << (loglen + logbatch)
    data = synth.synth_packed(900 + loglen + 7 * kind, total).tobytes()
    root = po.primitive_nth_root(1 << loglen)
    out = ctypes.create_string_buffer(16 * total)
    expect = _batched_expect(data, kind, loglen, logbatch, root)
    for inner_direct in (0, 1):                     # inter-pass twiddles by two-level lookup / from the direct table
        rc = emu.emu_ntt_batched(data, out, kind, loglen, logbatch, root.to_bytes(16, "little"), tile, loge, min_tiles, max_col, digit, None, 0, 0, 0, 0, inner_direct)
        assert rc > 0, rc
        assert out.raw == expect, (cfg, inner_direct)
    import numpy as np
    ln, bt = 1 << loglen, 1 << logbatch
    if kind == 0:
        # fused outer twiddle: out[r][c] *= w^(r * (col_base + c)) [* order^-1]
        ologn = loglen + logbatch + 2
        w = po.primitive_nth_root(1 << ologn)
        for col_base, ninv in ((0, 0), (bt * 3, 1)):
            ints = synth.unpack_ints(expect)
            sc_ = pow(1 << ologn, P - 2, P) if ninv else 1
            want = [ints[r * bt + c] * pow(w, r * (col_base + c), P) * sc_ % P for r in range(ln) for c in range(bt)]
            for tables in (col_base & 1 or ninv, 2, 3):          # bit 0: direct inter-pass table, bit 1: direct outer-twiddle table
                rc = emu.emu_ntt_batched(data, out, kind, loglen, logbatch, root.to_bytes(16, "little"), tile, loge, min_tiles, max_col, digit,
                                         w.to_bytes(16, "little"), ologn, col_base, ninv, 0, tables)
                assert rc > 0
                assert synth.unpack_ints(out.raw) == want, (cfg, col_base, tables)
    else:
        # chunked input [chunks][batch][len/chunks]
        for chunks_log in (1, 2):
            first_digit = loglen if loglen <= digit else (loglen + 1) // 2
            if chunks_log >= first_digit or chunks_log > loglen - 1:
                continue
            ch = 1 << chunks_log
            a = np.frombuffer(data, dtype=np.uint64).reshape(bt, ch, ln // ch, 2)
            chunked = np.ascontiguousarray(a.transpose(1, 0, 2, 3)).tobytes()
            rc = emu.emu_ntt_batched(chunked, out, kind, loglen, logbatch, root.to_bytes(16, "little"), tile, loge, min_tiles, max_col, digit, None, 0, 0, 0, chunks_log, chunks_log & 1)
            assert rc > 0, (cfg, chunks_log)
            assert out.raw == expect, (cfg, chunks_log)
            # the same rows as ROW BLOCKS of an overlapped corner turn: each block of bt/2 rows is transformed on its own and
            # written as bt/2 adjacent columns of the full [len][bt] output (leading dimension bt)
            if bt >= 2:
                emu.emu_ntt_rows_ld.restype = ctypes.c_int
                emu.emu_ntt_rows_ld.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p] + [ctypes.c_int] * 7 + [ctypes.c_uint64]
                wide = ctypes.create_string_buffer(16 * total)
                half = bt // 2
                for blk in range(2):
                    part = np.ascontiguousarray(a[blk * half:(blk + 1) * half].transpose(1, 0, 2, 3)).tobytes()
                    dst = ctypes.c_void_p(ctypes.addressof(wide) + 16 * blk * half)
                    rc = emu.emu_ntt_rows_ld(part, dst, loglen, logbatch - 1, root.to_bytes(16, "little"), tile, loge, min_tiles, max_col, digit, chunks_log, 1, bt)
                    assert rc > 0, (cfg, chunks_log, blk)
                assert wide.raw == expect, (cfg, chunks_log, "row blocks")


@pytest.mark.parametrize("log2n,world,blocks,defer,diag", [(8, 2, 1, False, True), (10, 4, 2, True, True), (12, 4, 4, True, True), (12, 2, 2, False, True),
                                                         (12, 4, 2, True, False), (12, 1, 4, True, True), (10, 8, 1, False, True)])
def test_emu_fourstep_stages(emu, log2n, world, blocks, defer, diag):
    """The stages of the sharded transform's plan object (sc_fourstep_cols_dev / _rows_dev / _rows_finish_dev) with the kernel
    body and planner the device uses: column stage with the outer twiddle, the rank's own block written straight into its
    receive buffer (PassParams::out_alt); row stage reading [G][R/G][C/G] in place with an explicit chunk stride, in row blocks,
    second pass deferred to one full launch.  A simulated world of ranks against the oracle's transform of the whole vector."""
    import numpy as np
    n = 1 << log2n
    root = po.primitive_nth_root(n)
    log1 = (log2n + 1) // 2
    n1, n2 = 1 << log1, n >> log1
    tune = (8, 2, 2, 3, 4)       # tile, loge, min_tiles, max_col, digit: two-pass stages from 2^5 up, several tiles per pass
    i32, u64, vp = ctypes.c_int, ctypes.c_uint64, ctypes.c_void_p
    emu.emu_fourstep_cols.restype = i32
    emu.emu_fourstep_cols.argtypes = [vp, vp, vp, i32, i32, vp, vp, i32, u64, i32, ctypes.c_uint32, ctypes.c_uint32] + [i32] * 5
    emu.emu_fourstep_rows.restype = i32
    emu.emu_fourstep_rows.argtypes = [vp, vp, vp, i32, i32, u64, u64, i32, vp, i32, i32] + [i32] * 5
    emu.emu_fourstep_rows_finish.restype = i32
    emu.emu_fourstep_rows_finish.argtypes = [vp, vp, i32, i32, vp] + [i32] * 5
    full = synth.synth_packed(77 + log2n, n)
    want = np.frombuffer(po.C.ntt(root, full.tobytes(), n), dtype=np.uint64).reshape(n2, n1, 2)
    G = world
    lg = G.bit_length() - 1

    def direction(slabs, R, C, rt, ninv):
        rw, cw = R // G, C // G
        logR, logC = R.bit_length() - 1, C.bit_length() - 1
        sends, recvs = [], []
        for g_ in range(G):
            send = np.full((G, rw, cw, 2), 0xEE, dtype=np.uint64)
            recv = np.full((G, rw, cw, 2), 0xDD, dtype=np.uint64)
            src = np.ascontiguousarray(slabs[g_])
            rc = emu.emu_fourstep_cols(src.ctypes.data, send.ctypes.data, recv.ctypes.data if diag else None, logR, cw.bit_length() - 1,
                                       pow(rt, C, P).to_bytes(16, "little"), rt.to_bytes(16, "little"), log2n, g_ * cw, ninv, g_ * rw, rw, *tune)
            assert rc > 0, rc
            sends.append(send)
            recvs.append(recv)
        for h in range(G):
            for g_ in range(G):
                if g_ != h or not diag:
                    recvs[h][g_] = sends[g_][h]
        outs = []
        K = blocks if rw % blocks == 0 else 1
        rk = rw // K
        for h in range(G):
            dst = np.full((C, rw, 2), 0xCC, dtype=np.uint64)
            work = np.zeros((rw, C, 2), dtype=np.uint64)
            two_pass = logC > tune[4]
            for q in range(K):
                lo, hi = (0, 1) if (defer and K > 1 and two_pass) else (0, 4)
                rc = emu.emu_fourstep_rows(recvs[h].ctypes.data, dst.ctypes.data, work.ctypes.data, logC, rk.bit_length() - 1, q * rk, rw, lg,
                                           pow(rt, R, P).to_bytes(16, "little"), lo, hi, *tune)
                assert rc > 0, rc
            if defer and K > 1 and two_pass:
                rc = emu.emu_fourstep_rows_finish(dst.ctypes.data, work.ctypes.data, logC, rw.bit_length() - 1, pow(rt, R, P).to_bytes(16, "little"), *tune)
                assert rc > 0, rc
            outs.append(dst)
        return outs

    m = full.reshape(n1, n2, 2)
    cw = n2 // G
    xs = [m[:, g_ * cw:(g_ + 1) * cw] for g_ in range(G)]
    ys = direction(xs, n1, n2, root, 0)
    rw = n1 // G
    for g_ in range(G):
        assert ys[g_].tobytes() == np.ascontiguousarray(want[:, g_ * rw:(g_ + 1) * rw]).tobytes(), ("forward", g_)
    zs = direction(ys, n2, n1, pow(root, n - 1, P), 1)
    for g_ in range(G):
        assert zs[g_].tobytes() == np.ascontiguousarray(xs[g_]).tobytes(), ("inverse", g_)


# (logn, cols): one pass (<= 2^11), two passes (default plans up to 2^20), three passes (above)
COLUMNS = [(6, 5), (11, 2), (12, 7), (12, 64), (13, 32), (14, 3), (14, 16), (15, 8), (16, 2), (16, 4), (17, 3), (19, 2), (21, 2)]


@pytest.mark.parametrize("cfg", COLUMNS)
def test_emu_column_batches(emu, cfg):
    """NttIo::cols (sc_ntt_columns_dev): several transforms in one set of launches equal the oracle column by column,
    forward and inverse, with the direct four-step tables (2: at the store) and with the two-level lookup (0)."""
    logn, cols = cfg
    n = 1 << logn
    emu.emu_ntt_columns.restype = ctypes.c_int
    emu.emu_ntt_columns.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    data = synth.synth_packed(300 + logn, n * cols).tobytes()
    root = po.primitive_nth_root(n)
    for inverse in (0, 1):
        want = b"".join((po.C.intt if inverse else po.C.ntt)(root, data[16 * n * c:16 * n * (c + 1)], n) for c in range(cols))
        for direct in ((2, 0) if logn <= 16 else (2,)):
            out = ctypes.create_string_buffer(16 * n * cols)
            rc = emu.emu_ntt_columns(data, out, logn, cols, int(root).to_bytes(16, "little"), inverse, direct)
            assert rc == (1 if logn <= 11 else 2 if logn <= 20 else 3), rc
            assert out.raw == want, (cfg, inverse, direct)


@pytest.mark.parametrize("cfg", [(8, 3, 100), (12, 5, 1024), (14, 3, 2048), (15, 4, 4096), (18, 3, 1 << 15), (21, 2, 1 << 18)])
def test_emu_lde_column_batches(emu, cfg):
    """sc_coset_evaluate_columns_dev (NttIo::col_stride_in): `cols` polynomials of m coefficients each, zero-padded, scaled by offset^j and
    transformed in one set of launches -- one, two and three passes, the pruned first pass, the eight-element kernels -- equal the oracle's
    fast_coset_evaluate column by column."""
    logn, cols, m = cfg
    n = 1 << logn
    emu.emu_coset_evaluate_columns.restype = ctypes.c_int
    emu.emu_coset_evaluate_columns.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
    data = synth.synth_packed(400 + logn, m * cols).tobytes()
    root = po.primitive_nth_root(n)
    out = ctypes.create_string_buffer(16 * n * cols)
    rc = emu.emu_coset_evaluate_columns(data, out, logn, cols, int(root).to_bytes(16, "little"), m, int(po.GENERATOR).to_bytes(16, "little"))
    assert rc > 0, rc
    want = b"".join(po.C.coset_evaluate(data[16 * m * c:16 * m * (c + 1)], m, po.GENERATOR, root, n) for c in range(cols))
    assert out.raw == want, cfg
