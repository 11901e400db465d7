"""FastStark.prove with its FRI-domain work sharded over the GPUs of one node (BASELINE configs[4]).

Interface and transcript of reference code/fast_stark.py:76-178 (`FastStark.prove`), one process per GPU.  What is sharded is
what is big -- everything that lives on the FRI domain (expansion_factor x the trace domain):

  * every low-degree extension (fast_stark.py:103-104, :119, :148) is `ShardedNtt.coset_evaluate`: the coefficient vector (1/blowup
    of the domain) is replicated, each rank transforms its column slab, ONE corner turn per LDE;
  * every Merkle commitment (fast_stark.py:105, :120) is `ShardedFri.commit`: local subtrees + one all-gather of sub-roots;
  * the transition and boundary quotients (fast_stark.py:93-98, :113) are `ShardedNtt.coset_divide` on the trace domain's
    coset (pointwise division on slabs, no exchange beyond the transforms' corner turns); the quotient's coefficients are
    all-gathered once, because the next steps need them replicated;
  * the low-degree test is `ShardedFri.prove` (slab-local folds, no element exchange), and the openings of the committed
    codewords (fast_stark.py:154-175) are answered by the owning ranks and merged with one collective per codeword.

The trace interpolation (fast_stark.py:84-87) is spread by COLUMN: register s is interpolated by rank s mod G (closed forms on the
progression {omicron^i}: csrc/geoseq.cuh) and broadcast -- the registers are independent, and the next steps need every
polynomial's coefficients on every rank.  What is small stays replicated on every rank and runs exactly as in
fast_stark.FastStark (same code): the degree bookkeeping and the weighted sum of the nonlinear combination -- polynomials of the
trace domain's size.  Byte parity ties the rest to the host: the os.urandom draws
(rank 0 draws, in the reference's order, and broadcasts the bytes), Fiat-Shamir, the proof stream.  Every rank ends with the
same proof, byte-identical to `FastStark.prove` on one GPU with the same random bytes (tests/test_gpu_sharded.py).
"""
import torch
import torch.distributed as dist

import fast_stark as _fs
from fast_stark import *                      # noqa: F401,F403  (FastStark and the reference's star-imported names)
from ntt import DevicePolynomial, DeviceDomain, _shrink_order, coset_divide_device, fast_interpolate_device, fast_zerofier_device
from sharded import ShardedNtt, _current_raw_stream
from sharded_fri import ShardedFri
import starkcore as _sc


class HipReplicatedSteps:
    """The steps of the prover that are NOT sharded -- polynomials of the trace domain's size, identical on every rank -- on the
    rank's GPU through the library (DevicePolynomial: the same code fast_stark.FastStark runs).  ShardedFastStark reaches them
    only through this object, so the multi-rank orchestration around them (what is sharded, what is gathered, which collective
    carries what) can be run under gloo on CPU with another implementation of these few operations (tests/sharded_worker.py)."""

    def __init__(self, stark):
        self.stark, self.field, self.device = stark, stark.field, stark.device

    # engines of the sharded parts (None: sharded.ShardedNtt / ShardedFri build their HIP engines)
    def ntt_engine(self):
        return None

    def fri_engine(self):
        return None

    def join(self):
        """The replicated steps run on the library's stream, the sharded ones on torch's current stream.  ShardedFastStark makes
        them the SAME stream (torch.cuda.ExternalStream over sc_stream()), and then there is nothing to do here; under any other
        current stream the two are ordered with each other on the device (sc_stream_join: two events).  The host never waits."""
        raw = _current_raw_stream(self.device)
        if raw != _sc.library_stream():
            if raw == 0:                                  # torch's null stream cannot be waited on from another stream's side
                torch.cuda.current_stream(self.device).synchronize()
                _sc.synchronize()
            else:
                _sc.stream_join(raw)

    # -- polynomials in, polynomials out
    def lift(self, polynomial):
        return self.stark._lift(polynomial)                # the same host Polynomial is packed and uploaded once, not per proof

    def zero(self):
        return DevicePolynomial(_sc.DeviceVector(1), self.field, 0)

    def subtract(self, lhs, host_polynomial):
        return lhs.minus(host_polynomial)

    def zerofier(self, domain, root, order):
        return fast_zerofier(domain, root, order)

    def zerofier_device(self, count):
        """fast_zerofier of {omicron^i, i < count} (fast_stark.py:37) as a DevicePolynomial: closed form on a progression"""
        domain = DeviceDomain.geometric(self.field.one(), self.stark.omicron, count)
        return DevicePolynomial.from_codeword(fast_zerofier_device(domain))

    def trace_polynomials(self, trace, rows, registers, raw, only=None):
        """fast_stark.py:79-87: the trace with its randomizer rows (`raw`: the os.urandom draws, in the reference's order)
        interpolated column by column through {omicron^i, i < rows} -- a geometric progression: a handful of convolutions per
        column over tables built once (csrc/geoseq.cuh).  trace: the reference's list of rows, or a fast_stark.DeviceTrace.
        only: the registers THIS rank interpolates (None: all of them); the others' places hold None."""
        stark, field = self.stark, self.field
        if isinstance(trace, _fs.DeviceTrace):
            columns = stark._randomized_columns(trace, raw)
        else:
            width = stark.num_registers
            draws = [field.sample(raw[17 * i:17 * i + 17]) for i in range(len(raw) // 17)]
            full = trace + [draws[r * width:(r + 1) * width] for r in range(stark.num_randomizers)]
            columns = [DeviceCodeword.from_list([row[s] for row in full], field) for s in registers]
        domain = stark._trace_domain(rows)
        return [DevicePolynomial.from_codeword(fast_interpolate_device(domain, column)) if only is None or s in only else None for s, column in zip(registers, columns)]

    def coset_divide(self, lhs, rhs, exact):
        s = self.stark
        return coset_divide_device(lhs, rhs, s.generator, s.omicron, s.omicron_domain_length, exact=exact)

    def sampled_polynomial(self, raw):
        return _fs.sampled_polynomial(raw, self.field)

    def random_polynomial(self, count):
        return _fs.random_polynomial(count, self.field)

    def combination(self, shifted, weights, max_degree):
        return self.stark._combination_on_device(shifted, weights, max_degree)

    def air_values(self, vals, nvars, count, terms, out, converted):
        """out[i] = sum_t coef_t * prod_j vals[j][i]^e_tj on a rank's slab (multivariate.py:83-90 pointwise; sc_mpoly_eval_ex_dev);
        vals [nvars][count][2] is converted to the library's internal form by the first call (converted=False)"""
        exps = bytes(e for k, _ in terms for e in k)
        coefs = b"".join(v.to_bytes(16, "little") for _, v in terms)
        self.join()
        _sc._check(_sc.lib().sc_mpoly_eval_ex_dev(vals.data_ptr(), nvars, count, exps, coefs, len(terms), out.data_ptr(), 1 if converted else 0, None))
        self.join()

    # -- between a polynomial and the torch tensor the collectives and the sharded transforms work on
    def coefficients(self, poly, length=None):
        """coefficients as a torch tensor [length][2]: a VIEW of the library's vector (CUDA array interface; the tensor keeps the
        vector alive) -- nothing on the sharded side writes into a polynomial's coefficients (slab_of copies what it keeps)"""
        length = len(poly) if length is None else length
        if not length:
            return torch.empty((0, 2), dtype=torch.int64, device=self.device)
        self.join()
        return torch.as_tensor(poly.vec, device=self.device)[:length]

    def polynomial(self, tensor, length):
        """the first `length` rows of a [..][2] tensor as a polynomial: a vector handle over the tensor's own memory (sc_vec_wrap;
        the polynomial keeps the tensor alive), no copy"""
        if not length:
            return DevicePolynomial(_sc.DeviceVector(1), self.field, 0)
        tensor = tensor.contiguous()
        self.join()
        return DevicePolynomial(_sc.DeviceVector.wrap(tensor.data_ptr(), length, tensor), self.field, length)


class ShardedFastStark(FastStark):
    # divisions whose transform is shorter than this stay on one GPU (replicated): nothing to win from an exchange of a few KiB
    MIN_SHARDED_LOG2 = 10

    def __init__(self, field, expansion_factor, num_colinearity_checks, security_level, num_registers, num_cycles, rank, world, device, group=None,
                 transition_constraints_degree=2, replicated_steps=None):
        super().__init__(field, expansion_factor, num_colinearity_checks, security_level, num_registers, num_cycles, transition_constraints_degree)
        assert field.p == Field.P_MAIN, "the sharded prover works in the main field"
        self.rank, self.world, self.device, self.group = rank, world, device, group
        # ONE stream for everything: torch's tensor ops and collectives are put on the library's own stream, so the replicated
        # steps (library stream) and the sharded ones (torch's current stream) are ordered by construction and nothing waits
        self._stream = None
        if replicated_steps is None and getattr(device, "type", None) == "cuda":
            self._stream = torch.cuda.ExternalStream(_sc.library_stream(), device=device)
        with self._on_stream():
            self.steps = HipReplicatedSteps(self) if replicated_steps is None else replicated_steps(self)
            self._ntts = {}
            self._zerofier_values = {}        # transform order -> (the zerofier, its values on this rank's slab of that coset)
            self.ntt_fri = self._ntt(self.fri_domain_length, self.omega.value)
            self.sfri = ShardedFri(self.fri, self.ntt_fri.n1, rank, world, device, engine=self.steps.fri_engine(), group=group)

    # a list here turns on the per-phase breakdown: after each phase of prove() the device is waited for and
    # (phase, seconds since the previous mark) is appended -- measurement only (tools/sharded_stark_profile.py, bench.py)
    phase_log = None

    def _mark(self, phase):
        if self.phase_log is None:
            return
        import time
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        now = time.perf_counter()
        if phase is not None:
            self.phase_log.append((phase, now - self._phase_t0))
        self._phase_t0 = now

    def _on_stream(self):
        import contextlib
        return torch.cuda.stream(self._stream) if self._stream is not None else contextlib.nullcontext()

    # -- plumbing ---------------------------------------------------------------------------------
    def _ntt(self, order, root):
        key = (order, int(root))
        if key not in self._ntts:
            self._ntts[key] = ShardedNtt(order.bit_length() - 1, int(root), self.rank, self.world, self.device, engine=self.steps.ntt_engine(), group=self.group)
        return self._ntts[key]

    def _tensor(self, poly, length=None):
        return self.steps.coefficients(poly, length)

    def _polynomial(self, tensor, length):
        return self.steps.polynomial(tensor, length)

    def _gather_natural(self, slab, rows, cols):
        """column slabs [rows][cols/G] of every rank -> the whole natural-order vector [rows*cols][2], on every rank"""
        parts = self.sfri._all_gather(slab)                           # [G][rows][cols/G][2]
        return parts.permute(1, 0, 2, 3).reshape(rows * cols, 2).contiguous()

    def _broadcast(self, t, src):
        if t.is_cuda and dist.get_backend(self.group) == "gloo":        # functional runs with several ranks on one GPU: host-staged
            host = t.cpu()
            dist.broadcast(host, src=src, group=self.group)
            t.copy_(host)
        else:
            dist.broadcast(t, src=src, group=self.group)
        return t

    def _trace_polynomials_by_column(self, trace, rows, registers, raw):
        steps, G, g = self.steps, self.world, self.rank
        if G == 1:
            return steps.trace_polynomials(trace, rows, registers, raw)
        registers = list(registers)
        mine = [s for s in registers if s % G == g]
        polys = steps.trace_polynomials(trace, rows, registers, raw, only=mine)
        for k, s in enumerate(registers):
            if s % G == g:
                t = steps.coefficients(polys[k], rows).contiguous()
                if t.shape[0] < rows:                      # (an interpolant whose top coefficients vanish is still `rows` long on the wire)
                    t = torch.cat([t, torch.zeros((rows - t.shape[0], 2), dtype=t.dtype, device=t.device)])
            else:
                t = torch.empty((rows, 2), dtype=torch.int64, device=self.device)
            self._broadcast(t, s % G)
            if s % G != g:
                polys[k] = steps.polynomial(t, rows)
        return polys

    def _shared_random_bytes(self, count):
        """`count` draws of os.urandom(17) in the reference's order (fast_stark.py:80, :117), made by rank 0 and broadcast: every
        rank must randomise the same trace"""
        raw = _fs.draw_random_bytes(count) if self.rank == 0 else bytes(17 * count)
        if self.world > 1:
            on_dev = dist.get_backend(self.group) == "nccl"
            t = torch.frombuffer(bytearray(raw), dtype=torch.uint8)
            t = t.to(self.device) if on_dev else t
            dist.broadcast(t, 0, group=self.group)
            raw = bytes(t.cpu().numpy())
        return raw

    def _shared_random_polynomial(self, count):
        """Polynomial([field.sample(os.urandom(17)) ...]) of `count` coefficients (fast_stark.py:116-117), the same on every rank.
        A patched (seeded) os.urandom: rank 0's draws in the reference's order, broadcast as bytes.  The operating system's:
        the library draws and samples them on the device (fast_stark.random_polynomial), each rank its share."""
        steps = self.steps
        if not (_fs.os_urandom_is_genuine() and hasattr(steps, "random_polynomial")):
            return steps.sampled_polynomial(self._shared_random_bytes(count))
        if self.world == 1:
            return steps.random_polynomial(count)
        # the operating system's randomness has no order to keep: every rank draws 1/G of the coefficients (the draws are the
        # serial part -- 3 ms for 2^21 coefficients on one rank) and one all-gather makes the polynomial the same everywhere
        G = self.world
        share = -(-count // G)
        mine = steps.coefficients(steps.random_polynomial(share), share)
        on_dev = dist.get_backend(self.group) == "nccl"
        if on_dev:
            whole = torch.empty((G * share, 2), dtype=torch.int64, device=self.device)
            dist.all_gather_into_tensor(whole, mine.contiguous(), group=self.group)
        else:
            parts = [torch.empty((share, 2), dtype=torch.int64) for _ in range(G)]
            dist.all_gather(parts, mine.cpu().contiguous(), group=self.group)
            whole = torch.cat(parts, dim=0).to(self.device)
        return steps.polynomial(whole, count)

    # -- sharded building blocks ---------------------------------------------------------------------
    def _lde_commit(self, poly):
        """fast_coset_evaluate onto the FRI coset (fast_stark.py:58-59) + Merkle.commit, both sharded: the layer record ShardedFri
        answers openings from; layer["root"] is the commitment"""
        slab = torch.empty(self.ntt_fri.local_shape(False), dtype=torch.int64, device=self.device)
        self.ntt_fri.coset_evaluate(self._tensor(poly), self.generator.value, slab)
        return self.sfri.commit(slab, self.ntt_fri.n2)

    def _coset_divide(self, lhs, rhs, exact=False):
        """fast_coset_divide (code/ntt.py:137-176) with the transforms and the pointwise division sharded; the quotient's
        coefficients replicated again.  Transform order chosen like ntt.py:155-157, so the result is the reference's also when
        the division is not exact.  exact=True: Polynomial.__truediv__'s zero-remainder assertion (univariate.py:99-103)."""
        assert(not rhs.is_zero()), "cannot divide by zero polynomial"
        if lhs.is_zero():
            return self.steps.zero()
        dl, dr = lhs.degree(), rhs.degree()
        assert(dr <= dl), "cannot divide by polynomial of larger degree"
        root, order = _shrink_order(self.omicron, self.omicron_domain_length, max(dl, dr))
        log2 = order.bit_length() - 1
        if log2 < ShardedFastStark.MIN_SHARDED_LOG2 or (1 << (log2 // 2)) < self.world:
            return self.steps.coset_divide(lhs, rhs, exact)
        ntt = self._ntt(order, root.value)
        num = ntt.slab_of(self._tensor(lhs, dl + 1), "div_num").clone()
        den = ntt.slab_of(self._tensor(rhs, dr + 1), "div_den").clone()
        q = torch.empty(ntt.local_shape(True), dtype=torch.int64, device=self.device)
        ntt.coset_divide(num, den, self.generator.value, q)
        full = self._gather_natural(q, ntt.n1, ntt.n2)
        n_out = dl - dr + 1
        if exact:
            tail = self._polynomial(full[n_out:], order - n_out)
            assert(tail.degree() == -1), "cannot perform polynomial division because remainder is not zero"
        return self._polynomial(full, n_out)

    def _transition_quotients(self, constraints, point, tz_dev):
        """fast_stark.py:107-113: `a.evaluate_symbolic(point)` divided by the transition zerofier, for every constraint a.

        The reference builds the transition polynomial from schoolbook products and divides it on the coset g * <root'>, root' of the
        order ntt.py:155-157 shrinks to.  Here the whole computation stays in the VALUE domain of that coset, SHARDED: the point
        polynomials (X, trace(X), trace(omicron X)) are evaluated on it with the sharded transform -- one per variable, shared by
        all constraints of the same order --, the AIR is evaluated pointwise on the rank's slab (mpoly_eval_kernel: values are
        values, whatever the layout), divided pointwise by the zerofier's values (kept per order: they depend on the prover's
        parameters only), and ONE sharded inverse per constraint returns the quotient's coefficients.  4 + 2 transforms instead of
        the 9 + 4 of "substitute (replicated), then divide (sharded)", and nothing of it replicated.
        Same result as the reference: the order is chosen from the degree BOUND of the transition polynomial; an exact division
        gives the same quotient on any coset large enough, and its length is its degree + 1.  Where that cannot be guaranteed --
        the interpolant of the pointwise quotient is longer than an exact quotient could be (a false witness: the reference then
        returns a truncation that depends on the transition polynomial's true degree), a shape the kernel does not take, an order
        too small to shard -- the constraint goes the replicated way."""
        steps = self.steps
        replicated = lambda a: self._coset_divide(a.evaluate_symbolic(point), tz_dev)
        if not hasattr(steps, "air_values"):
            return [replicated(a) for a in constraints]
        degrees = [q.degree() for q in point]
        dr = tz_dev.degree()
        out, groups = [None] * len(constraints), {}
        for i, a in enumerate(constraints):
            plan = a.value_domain_terms(degrees)
            if plan is NotImplemented or plan[0] < max(dr, 8) or dr < 0:
                continue
            bound, terms = plan
            root, order = _shrink_order(self.omicron, self.omicron_domain_length, max(bound, dr))
            log2 = order.bit_length() - 1
            if log2 < ShardedFastStark.MIN_SHARDED_LOG2 or (1 << (log2 // 2)) < self.world or len(tz_dev) > order or any(len(q) > order for q in point):
                continue
            groups.setdefault(order, (root, []))[1].append((i, bound, terms))
        for order, (root, members) in groups.items():
            ntt = self._ntt(order, root.value)
            shape = ntt.local_shape(False)
            count = shape[0] * shape[1]
            nvars = len(point)
            used = [any(k[j] for _, _, terms in members for k, _ in terms) for j in range(nvars)]
            vals = torch.empty((nvars,) + tuple(shape), dtype=torch.int64, device=self.device)
            for j, q in enumerate(point):
                if used[j]:
                    ntt.coset_evaluate(self._tensor(q, degrees[j] + 1), self.generator.value, vals[j])
            kept = self._zerofier_values.get(order)
            if kept is not None and kept[0] is tz_dev:
                zvals = kept[1]
            else:
                zvals = torch.empty(shape, dtype=torch.int64, device=self.device)
                ntt.coset_evaluate(self._tensor(tz_dev, dr + 1), self.generator.value, zvals)
                if len(self._zerofier_values) >= 4:
                    self._zerofier_values.clear()
                self._zerofier_values[order] = (tz_dev, zvals)
            converted = False
            for i, bound, terms in members:
                tvals = torch.empty(shape, dtype=torch.int64, device=self.device)
                steps.air_values(vals, nvars, count, terms, tvals, converted)
                converted = True
                ntt.divide_values(tvals, zvals, tvals)                   # "divide by zero" on every rank together
                q = torch.empty(ntt.local_shape(True), dtype=torch.int64, device=self.device)
                ntt.inverse(tvals, q)
                ntt.coset_scale(q, pow(self.generator.value, self.field.p - 2, self.field.p))
                full = self._gather_natural(q, ntt.n1, ntt.n2)
                whole = self._polynomial(full, order)
                degree = whole.degree()
                if degree > bound - dr:
                    continue                                             # not the quotient of an exact division: the reference's way
                out[i] = DevicePolynomial(whole.vec, self.field, degree + 1) if degree >= 0 else steps.zero()
        return [q if q is not None else replicated(a) for q, a in zip(out, constraints)]

    # -- preprocessing (fast_stark.py:36-40) -----------------------------------------------------------
    def preprocess(self, device_resident=False):
        """-> (transition_zerofier, its committed codeword as a sharded layer record, the root).  device_resident=True: the
        zerofier of {omicron^i, i < T - 1} is made on the device from its closed form (no host list of the domain)"""
        with self._on_stream():
            if device_resident and self.original_trace_length - 1 >= 2 and hasattr(self.steps, "zerofier_device"):
                transition_zerofier = self.steps.zerofier_device(self.original_trace_length - 1)
            else:
                transition_zerofier = self.steps.zerofier(self.omicron_domain[:(self.original_trace_length - 1)], self.omicron, len(self.omicron_domain))
            layer = self._lde_commit(self.steps.lift(transition_zerofier))
            return transition_zerofier, layer, layer["root"]

    # -- prover (fast_stark.py:76-178) -------------------------------------------------------------------
    def prove(self, trace, transition_constraints, boundary, transition_zerofier, transition_zerofier_layer, proof_stream=None):
        """trace: the reference's list of rows, or a fast_stark.DeviceTrace (columns resident in HBM)"""
        with self._on_stream():
            return self._prove(trace, transition_constraints, boundary, transition_zerofier, transition_zerofier_layer, proof_stream)

    def _prove(self, trace, transition_constraints, boundary, transition_zerofier, transition_zerofier_layer, proof_stream):
        if proof_stream == None:
            proof_stream = ProofStream()
        field, registers = self.field, range(self.num_registers)
        self._mark(None)

        # randomizer rows appended to the trace (draw order: row by row, register by register)
        raw = self._shared_random_bytes(self.num_randomizers * self.num_registers)
        trace_rows = len(trace) + self.num_randomizers
        if hasattr(self.steps, "random_polynomial"):
            # this rank's share of the randomizer polynomial's draws starts now and passes while the GPU works on the trace
            _fs.prefetch_random_polynomial(-(-(self.max_degree(transition_constraints) + 1) // self.world))

        interpolants = self.boundary_interpolants(boundary)
        zerofiers = self.boundary_zerofiers(boundary)
        # the trace polynomials through {omicron^i} (fast_stark.py:84-87): the registers are independent columns --
        # register s is interpolated by rank s mod G and broadcast; every rank needs every polynomial's coefficients next (the
        # boundary quotients and the AIR's point cut their own slabs out of them).  [Sharding ONE column's interpolation instead --
        # its four transforms of twice the trace length through ShardedNtt -- is four corner turns per column for half a
        # millisecond of arithmetic: DESIGN.md section 4.]
        steps = self.steps
        trace_polynomials = self._trace_polynomials_by_column(trace, trace_rows, registers, raw)
        self._mark("trace interpolation (one register per rank, broadcast)")
        # sharded: boundary quotients, their LDEs and commitments (fast_stark.py:89-105)
        zerofiers_dev = [steps.lift(z) for z in zerofiers]
        boundary_quotients = [self._coset_divide(steps.subtract(trace_polynomials[s], interpolants[s]), zerofiers_dev[s], exact=True) for s in registers]
        self._mark("boundary quotients (sharded division)")
        boundary_layers = []
        for s in registers:
            boundary_layers.append(self._lde_commit(boundary_quotients[s]))
            proof_stream.push(boundary_layers[s]["root"])
        self._mark("boundary quotient LDEs + commitments (sharded)")

        # replicated: the AIR substituted in (X, trace(X), trace(omicron X)) in the value domain; sharded: the quotients
        x = Polynomial([field.zero(), field.one()])
        point = [steps.lift(x)] + trace_polynomials + [tp.scale(self.omicron) for tp in trace_polynomials]
        tz_dev = steps.lift(transition_zerofier)
        transition_quotients = self._transition_quotients(transition_constraints, point, tz_dev)
        self._mark("AIR substitution + transition quotients (value domain, sharded)")

        # randomizer polynomial (rank 0's draws), its sharded LDE and commitment
        max_degree = self.max_degree(transition_constraints)
        randomizer_polynomial = self._shared_random_polynomial(max_degree + 1)
        self._mark("randomizer polynomial: os.urandom / getrandom draws and Field.sample")
        randomizer_layer = self._lde_commit(randomizer_polynomial)
        proof_stream.push(randomizer_layer["root"])
        self._mark("randomizer polynomial: LDE, commitment (sharded)")

        weights = self.sample_weights(1 + 2 * len(transition_quotients) + 2 * len(boundary_quotients), proof_stream.prover_fiat_shamir())
        tq_bounds = self.transition_quotient_degree_bounds(transition_constraints)
        assert([tq.degree() for tq in transition_quotients] == tq_bounds), "transition quotient degrees do not match with expectation"

        # nonlinear combination (replicated axpys over coefficient vectors), its sharded LDE, the sharded low-degree test
        bq_bounds = self.boundary_quotient_degree_bounds(trace_rows, boundary)
        shifted = [(randomizer_polynomial, None)]
        for i, tq in enumerate(transition_quotients):
            shifted.append((tq, max_degree - tq_bounds[i]))
        for i in registers:
            shifted.append((boundary_quotients[i], max_degree - bq_bounds[i]))
        combination = steps.combination(shifted, weights, max_degree)
        self._mark("weights, degree checks, nonlinear combination (replicated)")
        slab = torch.empty(self.ntt_fri.local_shape(False), dtype=torch.int64, device=self.device)
        self.ntt_fri.coset_evaluate(self._tensor(combination), self.generator.value, slab)
        self._mark("LDE of the combination (sharded)")
        # the openings of the committed codewords depend on the same sampled indices as the query phase: fetched with it, in ONE
        # library call and ONE collective (fri.AlsoOpen), when the stream takes the device's answers as they are
        N = self.fri.domain_length
        layers = boundary_layers + [randomizer_layer, transition_zerofier_layer]

        def opened_positions(indices):
            # the queried positions and their expansion_factor / half-domain companions (fast_stark.py:154-158)
            duplicated_indices = [i for i in indices] + [(i + self.expansion_factor) % N for i in indices]
            quadrupled_indices = [i for i in duplicated_indices] + [(i + (N // 2)) % N for i in duplicated_indices]
            quadrupled_indices.sort()
            return quadrupled_indices
        import proof_objects as _po
        together = None
        if hasattr(self.sfri.engine, "query_many") and type(proof_stream) is ProofStream:
            together = AlsoOpen(lambda indices: (layers, [opened_positions(indices)] * len(layers)))
            together.layers, together.shift = list(layers), self.expansion_factor      # the same request as data (ShardedFri on one rank hands it to the library)
        indices = self.sfri.prove(slab, proof_stream, together) if together is not None else self.sfri.prove(slab, proof_stream)
        self._mark("FRI: commit + query phases (sharded), openings fetched with them")

        quadrupled_indices = opened_positions(indices)
        lazy = _po.lazy_objects(proof_stream) if hasattr(self.sfri.engine, "query_many") else None
        if lazy is not None:
            # the owners' answers as they are (proof_objects.Openings): same transcript bytes, no object per digest
            answers = together.answers if together is not None and together.answers is not None else self.sfri._open_many_arrays([(layer, quadrupled_indices) for layer in layers])
            arrays = getattr(together, "position_arrays", None) if together is not None and together.answers is answers else None
            for layer, (values, paths), where in zip(layers, answers, arrays or [None] * len(layers)):
                lazy.add(_po.Openings(self.sfri._holder(layer, field), quadrupled_indices, values, paths, where))
            layers = []
        for entries, paths in self.sfri._open_many([(layer, quadrupled_indices) for layer in layers]) if layers else []:
            if type(proof_stream) is ProofStream:            # push == objects.append
                proof_stream.objects.extend(x for pair in zip(entries, paths) for x in pair)
                continue
            for entry, path in zip(entries, paths):
                proof_stream.push(entry)
                proof_stream.push(path)
        self._mark("openings of the committed codewords (sharded)")
        proof = proof_stream.serialize()
        self._mark("proof serialization (host pickle)")
        return proof
