"""Seeded differential fuzzing of the C-ABI against the oracle: random shapes (ragged lengths, tiny and mid sizes, random
primitive roots, edge residues 0 / 1 / p-1 mixed in), every entry point of the hot path.  Bit-exact."""
import ctypes
import random

import pytest

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
    return starkcore


def rand_vals(rng, n):
    edge = (0, 1, 2, P - 1, P - 2, 1 << 64, (1 << 64) - 1, 1 << 127)
    return [rng.choice(edge) if rng.random() < 0.15 else rng.randrange(P) for _ in range(n)]


def rand_root(rng, n):
    """a random primitive n-th root: the canonical one raised to a random odd power"""
    return pow(po.primitive_nth_root(n), rng.randrange(n) | 1, P) if n > 1 else 1


def test_fuzz_ntt_and_lde(sc):
    rng = random.Random(101)
    lib = sc.lib()
    for _ in range(60):
        logn = rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14])
        n = 1 << logn
        root = rand_root(rng, n)
        data = synth.pack_ints(rand_vals(rng, n))
        out = ctypes.create_string_buffer(16 * n)
        inv = rng.randrange(2)
        sc._check(lib.sc_ntt(data, out, n, sc.fe_bytes(root), inv))
        assert out.raw == (C.intt(root, data, n) if inv else C.ntt(root, data, n)), (logn, inv)
        m = rng.randrange(0, n + 1)
        offset = rng.choice([1, 2, po.GENERATOR, rng.randrange(1, P)])
        coeffs = synth.pack_ints(rand_vals(rng, m))
        sc._check(lib.sc_coset_evaluate(coeffs, m, sc.fe_bytes(offset), sc.fe_bytes(root), n, out))
        assert out.raw == C.coset_evaluate(coeffs, m, offset, root, n), (logn, m)


def test_fuzz_multiply_divide(sc):
    rng = random.Random(102)
    lib = sc.lib()
    for _ in range(40):
        la, lb = rng.randrange(1, 300), rng.randrange(1, 300)
        a, b = rand_vals(rng, la), rand_vals(rng, lb)
        if a[-1] == 0:
            a[-1] = 7                                  # exact division needs the true degree
        if b[-1] == 0:
            b[-1] = 9
        order = 1 << max(1, (la + lb - 1).bit_length())
        root = rand_root(rng, order)
        n_out = la + lb - 1
        out = ctypes.create_string_buffer(16 * n_out)
        sc._check(lib.sc_poly_mul(synth.pack_ints(a), la, synth.pack_ints(b), lb, sc.fe_bytes(root), order, out, n_out))
        assert synth.unpack_ints(out.raw) == po.schoolbook_mul(a, b), (la, lb)
        quo = ctypes.create_string_buffer(16 * lb)
        offset = rng.choice([po.GENERATOR, 3])
        try:
            sc._check(lib.sc_coset_divide(out.raw, n_out, synth.pack_ints(a), la, sc.fe_bytes(offset), sc.fe_bytes(root), order, quo, lb))
        except AssertionError:
            continue                                   # the divisor happened to vanish on the coset (algebra.py:92)
        assert synth.unpack_ints(quo.raw) == b, (la, lb)


def test_fuzz_fold_and_merkle(sc):
    rng = random.Random(103)
    lib = sc.lib()
    for _ in range(30):
        logn = rng.randrange(1, 13)
        N = 1 << logn
        data = synth.pack_ints(rand_vals(rng, N))
        om = rand_root(rng, N)
        alpha, offset = rng.randrange(P), rng.randrange(1, P)
        out = ctypes.create_string_buffer(8 * N)
        sc._check(lib.sc_fri_fold(data, N, sc.fe_bytes(alpha), sc.fe_bytes(offset), sc.fe_bytes(om), out))
        assert out.raw == C.fold(data, N, alpha, offset, om), logn
        tree = sc.MerkleTree.from_bytes(data)
        assert tree.root == C.merkle_commit(data, N)
        idx = [rng.randrange(N) for _ in range(5)]
        assert tree.open_batch(idx) == [C.merkle_open(data, N, i) for i in idx]
        tree.free()


def test_fuzz_subproduct_tree(sc):
    rng = random.Random(104)
    lib = sc.lib()
    for _ in range(25):
        k = rng.randrange(1, 120)
        pts = rng.sample(range(1, 1 << 20), k)         # distinct
        if rng.random() < 0.3:
            pts[rng.randrange(k)] = 0
        pts = [p_ if rng.random() < 0.5 else (p_ * 0x9E3779B97F4A7C15) % P for p_ in pts]
        if len(set(pts)) != k:
            continue
        vals = rand_vals(rng, k)
        m = rng.randrange(0, 3 * k + 2)
        f = rand_vals(rng, m)
        order = 256
        root = po.primitive_nth_root(order)
        out = ctypes.create_string_buffer(16 * (k + 1))
        sc._check(lib.sc_zerofier(synth.pack_ints(pts), k, out))
        assert synth.unpack_ints(out.raw) == po.fast_zerofier(pts, root, order), k
        sc._check(lib.sc_evaluate(synth.pack_ints(f), m, synth.pack_ints(pts), k, out))
        assert synth.unpack_ints(out.raw)[:k] == [po.evaluate(f, x) for x in pts], (k, m)
        sc._check(lib.sc_interpolate(synth.pack_ints(pts), synth.pack_ints(vals), k, out))
        assert synth.unpack_ints(out.raw)[:k] == po.fast_interpolate(pts, vals, root, order), k


def test_fuzz_commit_rounds_and_wide_queries(sc):
    """Round 2 entries: one commit round in one call (sc_fri_fold_commit_dev: fold, then the tree of the folded codeword, root
    fetched later) and all openings of several trees in one launch (sc_merkle_query_multi_dev), random shapes vs the oracle."""
    from algebra import Field, FieldElement
    field = Field.main()
    rng = random.Random(303)
    lib = sc.lib()
    for _ in range(40):
        logn = rng.randrange(1, 13)
        N = 1 << logn
        vals = rand_vals(rng, N)
        data = synth.pack_ints(vals)
        v = sc.DeviceVector.from_bytes(data)
        alpha, offset = rng.randrange(P), rng.randrange(1, P)
        omega = rand_root(rng, N)
        cw = sc.DeviceCodeword(v, field)
        folded = cw.fold_commit(FieldElement(alpha, field), FieldElement(offset, field), FieldElement(omega, field), sc.DeviceVector(N // 2))
        want = C.fold(data, N, alpha, offset, omega)
        assert folded.vec.to_bytes() == want
        assert folded.tree().root == C.merkle_commit(want, N // 2)
        assert cw.start_tree().root == C.merkle_commit(data, N)
    for _ in range(6):
        ntrees = rng.randrange(1, 45)
        sizes = [1 << rng.randrange(0, 11) for _ in range(ntrees)]
        datas = [synth.pack_ints(rand_vals(rng, m)) for m in sizes]
        vecs = [sc.DeviceVector.from_bytes(d) for d in datas]
        trees = [sc.MerkleTree.from_device_async(x) if rng.random() < 0.5 else sc.MerkleTree.from_device(x) for x in vecs]
        reqs = [[rng.randrange(m) for _ in range(rng.randrange(0, 6))] for m in sizes]
        flat = [i for r in reqs for i in r]
        if not flat:
            continue
        el = ctypes.create_string_buffer(16 * len(flat))
        pbytes = sum(64 * (m.bit_length() - 1) * len(r) for m, r in zip(sizes, reqs))
        pa = ctypes.create_string_buffer(max(pbytes, 64))
        sc._check(lib.sc_merkle_query_multi_dev(ntrees, (ctypes.c_void_p * ntrees)(*[t._h for t in trees]), (ctypes.c_void_p * ntrees)(*[x.ptr for x in vecs]),
                                                (ctypes.c_uint64 * len(flat))(*flat), (ctypes.c_uint64 * ntrees)(*[len(r) for r in reqs]), el, pa))
        eo = po_ = 0
        for d_, m, r in zip(datas, sizes, reqs):
            d = m.bit_length() - 1
            for i in r:
                assert el.raw[eo:eo + 16] == d_[16 * i:16 * i + 16]
                if d:
                    assert pa.raw[po_:po_ + 64 * d] == b"".join(C.merkle_open(d_, m, i))
                eo += 16
                po_ += 64 * d


def test_fuzz_progressions(sc):
    """fast_zerofier / fast_evaluate / fast_interpolate on geometric progressions (csrc/geoseq.cuh) with random first points, ratios
    of every kind (roots of unity of order >= n, arbitrary residues, 2, p - 1 squared away), edge residues among the values, and
    polynomials shorter and longer than the domain -- through the host-buffer C-ABI entries (which must DETECT the progression) and
    against the oracle's restatement of the reference recursion"""
    rng = random.Random(104)
    lib = sc.lib()
    order = 1 << 10
    root = po.primitive_nth_root(order)
    for trial in range(40):
        n = rng.choice([2, 3, 5, 8, 13, 32, 33, 64, 100, 129, 200])
        kind = rng.randrange(4)
        if kind == 0:
            q = pow(root, rng.randrange(order) | 1, P)                     # a primitive 2^10-th root: a prefix of a subgroup
        elif kind == 1:
            q = pow(po.primitive_nth_root(1 << rng.choice([8, 9, 12, 20])), rng.randrange(1 << 8) | 1, P)
            if pow(q, 1, P) == 1 or any(pow(q, k, P) == 1 for k in range(1, n)):
                continue
        elif kind == 2:
            q = rng.randrange(2, P)
        else:
            q = 2
        c = rng.choice([1, P - 1, po.GENERATOR, rng.randrange(1, P)])
        pts, x = [], c
        for _ in range(n):
            pts.append(x)
            x = x * q % P
        if len(set(pts)) != n:
            continue
        out = ctypes.create_string_buffer(16 * (n + 1))
        sc._check(lib.sc_zerofier(synth.pack_ints(pts), n, out))
        assert synth.unpack_ints(out.raw) == po.fast_zerofier(pts, root, order), (trial, n, kind)
        m = rng.choice([0, 1, n // 2, n, n + 1, 2 * n + 3])
        f = rand_vals(rng, m)
        sc._check(lib.sc_evaluate(synth.pack_ints(f), m, synth.pack_ints(pts), n, out))
        assert synth.unpack_ints(out.raw)[:n] == [po.evaluate(f, x) for x in pts], (trial, n, m, kind)
        vals = rand_vals(rng, n)
        sc._check(lib.sc_interpolate(synth.pack_ints(pts), synth.pack_ints(vals), n, out))
        assert synth.unpack_ints(out.raw)[:n] == po.fast_interpolate(pts, vals, root, order), (trial, n, kind)
        dom = sc.GeoDomain.create(c, q, n)
        assert dom is not None and synth.unpack_ints(dom.interpolate(sc.DeviceVector.from_bytes(synth.pack_ints(vals))).to_bytes()) == po.fast_interpolate(pts, vals, root, order)
        dom.free()


def test_fuzz_column_batches(sc):
    """sc_ntt_columns_dev against sc_ntt_dev column by column (the single transform is pinned to the oracle and the reference's goldens
    elsewhere): random lengths 2^1 ... 2^21, column counts that are not powers of two, random primitive roots, forward and inverse,
    in place and out of place -- the one-pass plans, the short columns on the long transforms' tiles, the eight-elements-per-thread
    kernels of the 2^12-element tiles (2^17 ... 2^20), the three-pass plans, and a batch that needs two sets of launches."""
    import numpy as np
    import torch
    rng = random.Random(103)
    lib = sc.lib()
    dev = torch.device("cuda", 0)
    cases = [(rng.randrange(1, 22), None) for _ in range(36)] + [(17, 5), (18, 3), (19, 2), (20, 2), (12, 70), (16, 33), (21, 33)]
    for logn, cols in cases:
        n = 1 << logn
        if cols is None:
            cols = rng.randrange(1, max(2, min(40, (1 << 23) >> logn)) + 1)
        root = sc.fe_bytes(rand_root(rng, n))
        inv = rng.randrange(2)
        host = synth.synth_packed(rng.randrange(1 << 30), n * cols)
        for i in range(0, n * cols, max(1, (n * cols) // 64)):                       # edge residues sprinkled in
            v = rng.choice((0, 1, P - 1, (1 << 64) - 1, 1 << 64))
            host[i, 0], host[i, 1] = v & ((1 << 64) - 1), v >> 64
        x = torch.from_numpy(host.view(np.int64).reshape(-1)).to(dev)
        want = torch.empty_like(x)
        for c in range(cols):
            sc._check(lib.sc_ntt_dev(x.data_ptr() + 16 * n * c, want.data_ptr() + 16 * n * c, n, root, inv, None))
        if rng.random() < 0.5:
            got = x.clone()
            torch.cuda.synchronize()        # (the copy runs on torch's stream, the library call on the library's: nothing else orders them)
            sc._check(lib.sc_ntt_columns_dev(got.data_ptr(), got.data_ptr(), n, cols, root, inv, None))
        else:
            got = torch.empty_like(x)
            sc._check(lib.sc_ntt_columns_dev(x.data_ptr(), got.data_ptr(), n, cols, root, inv, None))
        sc.synchronize()
        assert torch.equal(got, want), (logn, cols, inv)
