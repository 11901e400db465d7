"""Worker for tests/test_sharded_cpu.py: one rank of a world_size-N gloo job on CPU.  The local compute stages
are served by the oracle (OracleEngine) so that the data-movement logic of sharded.py (slab layout, global twiddle
indices, all-to-all corner turn, reassembly, transposed output) is what gets tested."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "stark-anatomy_amd"), REPO):
    if p not in sys.path:
        sys.path.insert(0, p)

from oracle import py_oracle as po          # noqa: E402
import synth                                 # noqa: E402
from sharded import ShardedNtt, gather_natural, rows_to_column_slab, column_slab_to_rows, P   # noqa: E402


class OracleEngine:
    def _np(self, t):
        return t.numpy().view(np.uint64)

    def cols_ntt(self, src, dst, length, batch, root):
        a = self._np(src).reshape(length, batch, 2)
        out = self._np(dst).reshape(length, batch, 2)
        for c in range(batch):
            out[:, c, :] = np.frombuffer(po.C.ntt(root, np.ascontiguousarray(a[:, c, :]).tobytes(), length), dtype=np.uint64).reshape(length, 2)

    def rows_ntt_t(self, src, dst, length, batch, root):
        a = self._np(src).reshape(batch, length, 2)
        out = self._np(dst).reshape(length, batch, 2)
        for r in range(batch):
            out[:, r, :] = np.frombuffer(po.C.ntt(root, np.ascontiguousarray(a[r]).tobytes(), length), dtype=np.uint64).reshape(length, 2)

    def scale_powers(self, src, dst, count, factor):
        raw = po.C.scale(src.contiguous().numpy().tobytes(), count, factor)
        dst.copy_(torch.from_numpy(np.frombuffer(raw, dtype=np.int64).reshape(count, 2).copy()))

    def scale_slab(self, src, dst, rows, cols, row_len, col_base, factor):
        a = self._np(src).reshape(-1, cols, 2)
        out = self._np(dst).reshape(-1, cols, 2)
        for r in range(rows):
            for c in range(cols):
                v = (int(a[r, c, 0]) | (int(a[r, c, 1]) << 64)) * pow(factor, r * row_len + col_base + c, P) % P
                out[r, c, 0], out[r, c, 1] = v & ((1 << 64) - 1), v >> 64

    def pointwise_mul(self, a, b, out, count):
        raw = po.C.pointwise_mul(a.contiguous().numpy().tobytes(), b.contiguous().numpy().tobytes(), count)
        out.view(-1).copy_(torch.from_numpy(np.frombuffer(raw, dtype=np.int64).copy()))

    def pointwise_div(self, a, b, out, count):
        raw = po.C.pointwise_div(a.contiguous().numpy().tobytes(), b.contiguous().numpy().tobytes(), count)
        out.view(-1).copy_(torch.from_numpy(np.frombuffer(raw, dtype=np.int64).copy()))

    def twiddle(self, buf, rows, cols, row_base, col_base, root, order, scale):
        a = self._np(buf).reshape(rows, cols, 2)
        for r in range(rows):
            for c in range(cols):
                v = int(a[r, c, 0]) | (int(a[r, c, 1]) << 64)
                v = v * pow(root, (row_base + r) * (col_base + c), P) * scale % P
                a[r, c, 0], a[r, c, 1] = v & ((1 << 64) - 1), v >> 64


class OracleStages:
    """The stage object of the HIP engine (sharded.HipFourstep over sc_fourstep_t) served by the oracle: same layouts, same
    contract -- column stage into `send` [G][R/G][C/G] with the rank's own block optionally written straight into `recv`, row
    stage reading `recv` in place, in row blocks, with an optionally deferred second pass (modelled by doing the whole
    transform of a block's rows either right away or in rows_finish) -- so that the orchestration of ShardedNtt._transform
    (blocks never written are poisoned) runs under gloo on the CPU."""

    def __init__(self, log2n, root, rank, world):
        self.n, self.root, self.rank, self.world = 1 << log2n, root, rank, world
        self.n1 = 1 << ((log2n + 1) // 2 if log2n <= 16 else 8)
        self.n2 = self.n // self.n1
        self.pending = {}

    def _dir(self, inverse):
        R, C = (self.n2, self.n1) if inverse else (self.n1, self.n2)
        root = pow(self.root, self.n - 1, P) if inverse else self.root
        return R, C, root, (pow(self.n, P - 2, P) if inverse else 1)

    def cols(self, inverse, src, send, recv_diag):
        R, C, root, scale = self._dir(inverse)
        G, g = self.world, self.rank
        rw, cw = R // G, C // G
        eng = OracleEngine()
        a = torch.empty((R, cw, 2), dtype=torch.int64)
        eng.cols_ntt(src.contiguous(), a, R, cw, pow(root, C, P))
        eng.twiddle(a, R, cw, 0, g * cw, root, self.n, scale)
        blocks = a.view(G, rw, cw, 2)
        for h in range(G):
            if h == g and recv_diag is not None:
                recv_diag.view(G, rw, cw, 2)[g].copy_(blocks[g])
                send.view(G, rw, cw, 2)[g].fill_(-1)               # never sent: poison it
            else:
                send.view(G, rw, cw, 2)[h].copy_(blocks[h])

    def _rows_now(self, R, C, root, recv, dst, row0, nrows):
        G = self.world
        rw, cw = R // G, C // G
        a = recv.numpy().view(np.uint64).reshape(G, rw, cw, 2)
        out = dst.numpy().view(np.uint64).reshape(C, rw, 2)
        rt = pow(root, R, P)
        for r in range(row0, row0 + nrows):
            row = np.ascontiguousarray(a[:, r]).reshape(C, 2)          # [G][cw] -> the row's C elements in column order
            out[:, r, :] = np.frombuffer(po.C.ntt(rt, row.tobytes(), C), dtype=np.uint64).reshape(C, 2)

    def rows(self, inverse, recv, dst, q, K, defer):
        R, C, root, _ = self._dir(inverse)
        rk = R // self.world // K
        if defer:
            self.pending.setdefault(inverse, []).append((recv.clone(), q * rk, rk))   # a snapshot: later blocks must not be needed
            return
        self._rows_now(R, C, root, recv, dst, q * rk, rk)

    def rows_finish(self, inverse, dst):
        R, C, root, _ = self._dir(inverse)
        for recv, row0, nrows in self.pending.pop(inverse, []):
            self._rows_now(R, C, root, recv, dst, row0, nrows)


class StagesOracleEngine(OracleEngine):
    def fourstep(self, log2n, root, rank, world):
        return OracleStages(log2n, root, rank, world)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ok = True
    # (size, engine, row blocks of the corner turn, deferred second pass, own block through the exchange)
    for log2n, engine, chunks, defer, always in ((6, OracleEngine(), 1, True, False), (7, StagesOracleEngine(), 2, True, False), (10, StagesOracleEngine(), 4, True, False),
                                                 (10, StagesOracleEngine(), 4, False, True), (10, StagesOracleEngine(), 1, True, False), (10, StagesOracleEngine(), 1, True, True),
                                                 (10, OracleEngine(), 1, True, True)):
        n = 1 << log2n
        root = po.primitive_nth_root(n)
        eng = ShardedNtt(log2n, root, rank, world, torch.device("cpu"), engine=engine, overlap_chunks=chunks, defer_last_pass=defer, always_exchange=always)
        x = eng.synthetic_input(seed=3)
        assert tuple(x.shape) == eng.local_shape(True)
        y = torch.empty(eng.local_shape(False), dtype=torch.int64)
        z = torch.empty_like(x)
        eng.forward(x, y)
        eng.inverse(y, z)
        full_in = synth.synth_packed(3, n).tobytes()
        # input slabs really are the column slabs of the natural vector
        got_in = gather_natural(x, eng.n1, eng.n2, world).numpy().tobytes()
        got = gather_natural(y, eng.n2, eng.n1, world).numpy().tobytes()
        ok &= got_in == full_in
        ok &= got == po.C.ntt(root, full_in, n)
        ok &= torch.equal(z, x)
        # sharded LDE: replicated coefficients (ragged length), coset offset = generator
        for m in (n // 4, n // 8 + 3, 1):
            coeffs = synth.synth_packed(9, m)
            lde = torch.empty(eng.local_shape(False), dtype=torch.int64)
            eng.coset_evaluate(torch.from_numpy(coeffs.view(np.int64).copy()), po.GENERATOR, lde)
            got_lde = gather_natural(lde, eng.n2, eng.n1, world).numpy().tobytes()
            ok &= got_lde == po.C.coset_evaluate(coeffs.tobytes(), m, po.GENERATOR, root, n)
        ok &= poly_checks(eng, rank, world, n, root)
    dist.barrier()
    dist.destroy_process_group()
    if not ok:
        sys.exit(3)
    print("rank", rank, "ok")


def poly_checks(eng, rank, world, n, root):
    """SURVEY 8(e)-5 on slabs: fast_multiply / fast_coset_divide cores (code/ntt.py:58-64, :159-176), and 8(e)-2's contiguous
    layout entering / leaving the column-slab layout with one exchange."""
    ok = True
    ints = lambda seed, k: synth.synth_ints(seed, k)
    as_t = lambda vals: torch.from_numpy(np.frombuffer(synth.pack_ints(vals), dtype=np.int64).reshape(len(vals), 2).copy())
    # product of two polynomials whose degrees add up to < n
    la, lb = n // 2 - 3, n // 2 + 1
    a, b = ints(21, la), ints(22, lb)
    out = torch.empty(eng.local_shape(True), dtype=torch.int64)
    eng.multiply(eng.slab_of(as_t(a), "ta").clone(), eng.slab_of(as_t(b), "tb").clone(), out)
    got = synth.unpack_ints(gather_natural(out, eng.n1, eng.n2, world).numpy().tobytes())
    want = po.schoolbook_mul(a, b)
    ok &= got[:len(want)] == want and not any(got[len(want):])
    # exact quotient (a * b) / b on the coset of offset = generator, like fast_stark.py:113
    prod = want
    q = torch.empty(eng.local_shape(True), dtype=torch.int64)
    eng.coset_divide(eng.slab_of(as_t(prod), "ta").clone(), eng.slab_of(as_t(b), "tb").clone(), po.GENERATOR, q)
    gq = synth.unpack_ints(gather_natural(q, eng.n1, eng.n2, world).numpy().tobytes())
    ok &= gq[:la] == a and not any(gq[la:])
    # divisor vanishing on the coset: the reference's field division asserts (algebra.py:92)
    zero_at = po.GENERATOR                                   # (X - offset*root^0) has a zero on the coset
    try:
        eng.coset_divide(eng.slab_of(as_t(prod), "ta").clone(), eng.slab_of(as_t([(-zero_at) % P, 1]), "tb").clone(), po.GENERATOR, q)
        ok = False
    except AssertionError as e:
        ok &= "divide by zero" in str(e)
    # natural contiguous chunks <-> column slabs
    full = np.frombuffer(synth.synth_packed(31, n).tobytes(), dtype=np.int64).reshape(eng.n1, eng.n2, 2)
    rw = eng.n1 // world
    chunk = torch.from_numpy(full[rank * rw:(rank + 1) * rw].copy())
    slab = rows_to_column_slab(chunk, eng.n1, eng.n2, rank, world)
    cw = eng.n2 // world
    ok &= torch.equal(slab, torch.from_numpy(full[:, rank * cw:(rank + 1) * cw].copy()))
    ok &= torch.equal(column_slab_to_rows(slab, eng.n1, eng.n2, rank, world), chunk)
    return bool(ok)


if __name__ == "__main__" and not (len(sys.argv) > 1 and sys.argv[1] in ("fri", "stark")):
    main()


# ---------------------------------------------------------------------------------------------------------------------
# ShardedFri under gloo: local primitives served by the oracle, proof compared with the reference's golden hashes
# ---------------------------------------------------------------------------------------------------------------------
class OracleFriEngine:
    class _Tree:
        def __init__(self, levels, n):
            self.levels, self.n, self.depth = levels, n, n.bit_length() - 1
            self.root = levels[-64:]

        def _node(self, level, idx):
            off = 0 if level == 0 else 2 * self.n - (self.n >> (level - 1))
            return self.levels[64 * (off + idx):64 * (off + idx + 1)]

        def open(self, indices):
            return [[self._node(l, (i >> l) ^ 1) for l in range(self.depth)] for i in indices]

    def tree(self, elems, need_root=True):
        data = elems.contiguous().numpy().tobytes()
        n = len(data) // 16
        return OracleFriEngine._Tree(po.C.merkle_tree(data, n), n)

    def level(self, tree, level):
        cnt = tree.n >> level
        raw = b"".join(tree._node(level, i) for i in range(cnt))
        return torch.from_numpy(np.frombuffer(raw, dtype=np.int64).reshape(cnt, 8).copy())

    def tree_from_digests(self, digests):
        import hashlib
        raw = digests.contiguous().numpy().tobytes()
        n = len(raw) // 64
        level = [raw[64 * i:64 * (i + 1)] for i in range(n)]
        allb = [raw]
        while len(level) > 1:
            level = [hashlib.blake2b(level[i] + level[i + 1]).digest() for i in range(0, len(level), 2)]
            allb.append(b"".join(level))
        return OracleFriEngine._Tree(b"".join(allb), n)

    def lde(self, coeffs, offset, generator, order):
        raw = po.C.coset_evaluate(coeffs, len(coeffs) // 16, offset, generator, order)
        return torch.from_numpy(np.frombuffer(raw, dtype=np.int64).reshape(order, 2).copy())

    def query_many(self, requests, raw_paths=False, raw_values=False):
        """[(tree, elems or None, indices[, keep])] -> [(values or None, paths cut to their first `keep` digests)] (the HIP engine
        answers all of them in one library call); raw_values: packed residues, a uint8 array [openings][16]"""
        out = []
        for req in requests:
            t, e, idx = req[:3]
            keep = req[3] if len(req) > 3 else None
            paths = [list(p) if keep is None else list(p[:keep]) for p in t.open(idx)] if t is not None else []
            if raw_paths:
                width = 64 * len(paths[0]) if paths else 0
                paths = np.frombuffer(b"".join(d for p in paths for d in p), dtype=np.uint8).reshape(len(paths), width)
            values = self.read(e, idx) if e is not None else None
            if raw_values and values is not None:
                values = np.frombuffer(b"".join(v.to_bytes(16, "little") for v in values), dtype=np.uint8).reshape(len(values), 16)
            out.append((values, paths))
        return out

    def read(self, elems, flat_indices):
        a = elems.contiguous().numpy().view(np.uint64).reshape(-1, 2)
        return [int(a[i, 0]) | (int(a[i, 1]) << 64) for i in flat_indices]

    def fold_slab(self, src, rows, cols, R, col_base, alpha, offset, omega):
        a = src.contiguous().numpy().view(np.uint64).reshape(rows, cols, 2)
        out = np.empty((rows // 2, cols, 2), dtype=np.uint64)
        inv2 = po.inv(2)
        for row in range(rows // 2):
            for col in range(cols):
                i = row * R + col_base + col
                t = alpha * po.inv(offset * pow(omega, i, P) % P) % P
                x = int(a[row, col, 0]) | (int(a[row, col, 1]) << 64)
                y = int(a[row + rows // 2, col, 0]) | (int(a[row + rows // 2, col, 1]) << 64)
                v = inv2 * ((1 + t) * x + (1 - t) * y) % P
                out[row, col, 0], out[row, col, 1] = v & ((1 << 64) - 1), v >> 64
        return torch.from_numpy(out.view(np.int64))

    def fold_full(self, src, N, alpha, offset, omega):
        raw = po.C.fold(src.contiguous().numpy().tobytes(), N, alpha, offset, omega)
        return torch.from_numpy(np.frombuffer(raw, dtype=np.int64).reshape(N // 2, 2).copy())


def fri_main():
    import hashlib
    import json
    from sharded import ShardedFri
    from algebra import Field
    from fri import Fri
    from ip import ProofStream
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    golden = json.load(open(os.path.join(REPO, "tests", "golden", "fri.json")))
    field = Field.main()
    ok = True
    for rec in golden["prove_synth"]:
        logN = rec["logN"]
        if logN > 10:
            continue
        N = 1 << logN
        om = field.primitive_nth_root(N)
        coeffs = synth.synth_packed(rec["coeff_seed"], N // 4).tobytes()
        cw = np.frombuffer(po.C.coset_evaluate(coeffs, N // 4, po.GENERATOR, om.value, N), dtype=np.int64).reshape(N, 2)
        for logR in ({6: (2, 3), 10: (3, 5, 8)}[logN]):
            R = 1 << logR
            if R < world:
                continue
            C, Rw = N // R, R // world
            slab = torch.from_numpy(cw.reshape(C, R, 2)[:, rank * Rw:(rank + 1) * Rw, :].copy())
            fr = Fri(field.generator(), om, N, rec["expansion_factor"], rec["num_colinearity_tests"])
            # local_tail: never gather early / gather half way through the rounds / the default (these sizes: before round 0)
            for tail in (0, N >> 2, None):
                ps = ProofStream()
                top = ShardedFri(fr, R, rank, world, torch.device("cpu"), engine=OracleFriEngine(), local_tail=tail).prove(slab, ps)
                ser = ps.serialize()
                good = (top == rec["top_level_indices"] and [o.hex() for o in ps.objects[:rec["num_rounds"]]] == rec["roots"]
                        and len(ser) == rec["serialized_len"] and hashlib.sha256(ser).hexdigest() == rec["serialized_sha256"])
                if not good:
                    print("rank", rank, "MISMATCH logN", logN, "R", R, "tail", tail, top == rec["top_level_indices"], len(ser), rec["serialized_len"], flush=True)
                ok &= good
        # SURVEY 8(e), row "FRI fold": the same proof from the NATURAL contiguous layout (one neighbour exchange per fold)
        from sharded import ContiguousFri
        if N // world >= 1:
            seg = N // world
            chunk = torch.from_numpy(cw[rank * seg:(rank + 1) * seg].copy())
            fr = Fri(field.generator(), om, N, rec["expansion_factor"], rec["num_colinearity_tests"])
            ps = ProofStream()
            cf = ContiguousFri(fr, rank, world, torch.device("cpu"), engine=OracleFriEngine())
            top = cf.prove(chunk, ps)
            ser = ps.serialize()
            good = (top == rec["top_level_indices"] and [o.hex() for o in ps.objects[:rec["num_rounds"]]] == rec["roots"]
                    and len(ser) == rec["serialized_len"] and hashlib.sha256(ser).hexdigest() == rec["serialized_sha256"])
            # what the upper halves shipped: N/2 + N/4 + ... while more than one rank holds data
            shipped = sum((N >> r) // 2 for r in range(min(rec["num_rounds"] - 1, world.bit_length() - 1)))
            total = torch.tensor([cf.elements_shipped], dtype=torch.int64)
            dist.all_reduce(total)
            good = good and int(total.item()) == shipped
            if not good:
                print("rank", rank, "CONTIGUOUS MISMATCH logN", logN, top == rec["top_level_indices"], len(ser), rec["serialized_len"], int(total.item()), shipped, flush=True)
            ok &= good
    # independent columns: one register per rank, roots gathered in column order
    from sharded import ColumnReplicas
    order = 256
    gen = po.primitive_nth_root(order)
    cols = [synth.synth_packed(50 + i, 40 + i).tobytes() for i in range(5)]
    mine, roots = ColumnReplicas(rank, world, torch.device("cpu"), engine=OracleFriEngine()).lde_and_commit(cols, po.GENERATOR, gen, order)
    want = [po.C.merkle_commit(po.C.coset_evaluate(c, len(c) // 16, po.GENERATOR, gen, order), order) for c in cols]
    ok &= roots == want and sorted(mine) == [i for i in range(5) if i % world == rank]
    dist.barrier()
    dist.destroy_process_group()
    if not ok:
        sys.exit(3)
    print("rank", rank, "ok")


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "fri":
    fri_main()


# =====================================================================================================================
# ShardedFastStark under gloo: the orchestration of the sharded prover with every local computation served by the oracle
# =====================================================================================================================
def stark_main():
    """sharded_stark.ShardedFastStark on the reference's Rescue-Prime workload (code/test_fast_stark.py) with the reference's seeded
    random bytes: every rank must end with the proof the REFERENCE produced (tests/golden/fast_stark.json: SHA-256 of the proof,
    the first roots, the zerofier commitment).  The replicated steps (trace-domain polynomials) are host Polynomial arithmetic
    and the oracle's fast_* restatements; the sharded ones run sharded.py's real data movement over the oracle engines."""
    import hashlib
    import json
    import random
    import fast_stark
    import sharded_stark
    from algebra import Field, FieldElement
    from univariate import Polynomial
    from ip import ProofStream
    from workload_rescue_prime import RescuePrime
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    field = Field.main()
    from multivariate import MPolynomial
    Polynomial.FAST_MUL_MIN_LEN = 10 ** 9            # host products stay schoolbook: no GPU in this process
    MPolynomial.VALUE_DOMAIN_MIN_DEGREE = 10 ** 9    # and the AIR substitution stays the reference's sums of products
    sharded_stark.ShardedFastStark.MIN_SHARDED_LOG2 = 4   # the quotients of this small workload (order 64 / 128) take the sharded route too

    def ints(polynomial):
        return [c.value for c in polynomial.coefficients]

    def poly(values):
        return Polynomial([FieldElement(int(v), field) for v in values])

    class OracleReplicatedSteps:
        def __init__(self, stark):
            self.stark = stark

        def ntt_engine(self):
            return StagesOracleEngine()

        def fri_engine(self):
            return OracleFriEngine()

        def join(self):
            pass

        def lift(self, polynomial):
            return polynomial

        def zero(self):
            return Polynomial([])

        def subtract(self, lhs, rhs):
            return lhs - rhs

        def zerofier(self, domain, root, order):
            return poly(po.fast_zerofier([d.value for d in domain], root.value, order))

        def trace_polynomials(self, trace, rows, registers, raw, only=None):
            s = self.stark
            width = s.num_registers
            draws = [field.sample(raw[17 * i:17 * i + 17]) for i in range(len(raw) // 17)]
            trace = trace + [draws[r * width:(r + 1) * width] for r in range(s.num_randomizers)]
            assert len(trace) == rows
            domain = [pow(s.omicron.value, i, P) for i in range(rows)]
            return [poly(po.fast_interpolate(domain, [row[r].value for row in trace], s.omicron.value, s.omicron_domain_length)) if only is None or r in only else None
                    for r in registers]

        def coset_divide(self, lhs, rhs, exact):
            if exact:
                return lhs / rhs                       # univariate.py's division: asserts a zero remainder
            s = self.stark
            return poly(po.fast_coset_divide(ints(lhs), ints(rhs), s.generator.value, s.omicron.value, s.omicron_domain_length))

        def sampled_polynomial(self, raw):
            return Polynomial([field.sample(raw[17 * i:17 * i + 17]) for i in range(len(raw) // 17)])

        def combination(self, shifted, weights, max_degree):
            x = Polynomial([field.zero(), field.one()])
            terms = []
            for polynomial, shift in shifted:
                terms += [polynomial] if shift is None else [polynomial, (x ^ shift) * polynomial]
            total = Polynomial([])
            for weight, term in zip(weights, terms):
                total = total + Polynomial([weight]) * term
            return total

        def coefficients(self, polynomial, length=None):
            values = ints(polynomial)
            length = len(values) if length is None else length
            values = (values + [0] * length)[:length]
            raw = b"".join(v.to_bytes(16, "little") for v in values)
            return torch.from_numpy(np.frombuffer(raw, dtype=np.int64).reshape(length, 2).copy()) if length else torch.empty((0, 2), dtype=torch.int64)

        def polynomial(self, tensor, length):
            a = tensor.contiguous().numpy().view(np.uint64).reshape(-1, 2)
            return poly(int(a[i, 0]) | (int(a[i, 1]) << 64) for i in range(length))

    golden = json.load(open(os.path.join(REPO, "tests", "golden", "fast_stark.json")))
    rp = RescuePrime()
    ok = True
    for rec in golden["runs"]:
        rng = random.Random(rec["urandom_seed"])
        fast_stark.os.urandom = lambda k, rng=rng: bytes(rng.getrandbits(8) for _ in range(k))
        input_element = FieldElement(int(rec["input"]), field)
        output_element = rp.hash(input_element)
        stark = sharded_stark.ShardedFastStark(field, rec["expansion_factor"], rec["num_colinearity_checks"], rec["security_level"], rp.m, rp.N + 1,
                                               rank, world, torch.device("cpu"), replicated_steps=OracleReplicatedSteps)
        transition_zerofier, layer, root = stark.preprocess()
        proof = stark.prove(rp.trace(input_element), rp.transition_constraints(stark.omicron), rp.boundary_constraints(output_element), transition_zerofier, layer)
        objects = ProofStream().deserialize(proof).objects
        good = (root.hex() == rec["zerofier_root"] and [o.hex() for o in objects[:rp.m + 1]] == rec["first_roots"] and len(objects) == rec["num_objects"]
                and len(proof) == rec["proof_len"] and hashlib.sha256(proof).hexdigest() == rec["proof_sha256"])
        # the quotients really took the sharded route: transforms of other orders than the FRI domain's were planned
        good = good and len(stark._ntts) >= 2
        if not good:
            print("rank", rank, "STARK MISMATCH seed", rec["urandom_seed"], root.hex() == rec["zerofier_root"], len(proof), rec["proof_len"], flush=True)
        ok &= good
    dist.barrier()
    dist.destroy_process_group()
    if not ok:
        sys.exit(3)
    print("rank", rank, "ok")


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "stark":
    stark_main()
