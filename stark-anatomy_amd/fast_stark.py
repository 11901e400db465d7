"""NTT-based STARK prover / verifier -- the main CALLER of the GPU hot path.

Host mirror of the interface of reference code/fast_stark.py:8-286:
`FastStark(field, expansion_factor, num_colinearity_checks, security_level, num_registers, num_cycles,
transition_constraints_degree=2)` with `preprocess / prove / verify` and the degree-bound helpers.  Every
`fast_zerofier`, `fast_interpolate`, `fast_coset_evaluate`, `fast_coset_divide`, `Merkle.commit/open` and
`Fri.prove` inside goes through ntt.py / merkle.py / fri.py to the MI355X.  The order of `os.urandom` draws and of
`proof_stream.push` calls is the reference's, so with a patched `fast_stark.os.urandom` the proof bytes are identical.
"""
from functools import reduce
import ctypes
import os

from fri import *
from univariate import *
from multivariate import *
from ntt import *
from ntt import _View, _shrink_order
import starkcore as _sc
import proof_objects as _po


def draw_random_bytes(count, width=17):
    """`count` draws of os.urandom(width) as one byte string, in draw order.  Long runs are drawn in blocks: os.urandom(n) is
    the next n bytes of the stream, so the bytes and their order are those of the individual draws."""
    block = 4096
    return b"".join(os.urandom(width * min(block, count - i)) for i in range(0, count, block))


class DeviceTrace:
    """The execution trace as device-resident COLUMNS: one DeviceVector of `rows` field elements per register -- what the
    `trace` argument of FastStark.prove (a list of rows of FieldElements, fast_stark.py:76) is to a caller whose trace never
    was a Python list.  A 2^20-row trace of two registers is two million Python objects and seconds of marshalling as lists
    (SURVEY.md App. C); as columns it is 32 MiB in HBM."""

    def __init__(self, columns, field):
        assert len(columns) >= 1 and all(c.n == columns[0].n for c in columns), "columns of one length, one per register"
        self.columns, self.field = list(columns), field

    @classmethod
    def from_rows(cls, rows, field):
        """rows: the reference's list of rows (lists of FieldElement or ints)"""
        width = len(rows[0])
        cols = [DeviceVector.from_bytes(b"".join(int(getattr(row[s], "value", row[s])).to_bytes(16, "little") for row in rows)) for s in range(width)]
        return cls(cols, field)

    @classmethod
    def from_packed(cls, packed_columns, field):
        """packed_columns: per register, the column as packed bytes (16 little-endian bytes per element)"""
        return cls([DeviceVector.from_bytes(c) for c in packed_columns], field)

    def __len__(self):
        return self.columns[0].n

    def entry(self, cycle, register):
        """one cell as a FieldElement (boundary conditions are read from the trace: 16 bytes from HBM)"""
        return FieldElement(int.from_bytes(self.columns[register].to_bytes(cycle, 1), "little"), self.field)


def sampled_polynomial(raw, field, width=17):
    """Polynomial([field.sample(raw[17 i : 17 i + 17]) ...]) as a DevicePolynomial: Field.sample on the device (sc_sample_bytes_dev)"""
    count = len(raw) // width
    vec = DeviceVector(max(count, 1))
    _sc._check(_sc.lib().sc_sample_bytes_dev(raw, count, width, vec.ptr, None))
    return DevicePolynomial(vec, field, count)


def os_urandom_is_genuine():
    """is os.urandom the interpreter's own (the operating system's generator), not a seeded stand-in a test has put in its place?
    Only then may the draws be made anywhere but through os.urandom itself, in any order."""
    return type(os.urandom).__name__ == "builtin_function_or_method"


def random_polynomial(count, field, width=17):
    """Polynomial([field.sample(os.urandom(17)) for i in range(count)]) (fast_stark.py:116-117) as a DevicePolynomial.  With the
    operating system's os.urandom the library makes the draws itself -- getrandom(2), several host threads, straight into a
    pinned buffer (sc_sample_urandom_dev); a patched os.urandom is called draw by draw in the reference's order."""
    if not os_urandom_is_genuine():
        return sampled_polynomial(draw_random_bytes(count, width), field, width)
    vec = DeviceVector(max(count, 1))
    _sc._check(_sc.lib().sc_sample_urandom_dev(count, width, vec.ptr, None))
    return DevicePolynomial(vec, field, count)


def prefetch_random_polynomial(count, width=17):
    """start the draws of a later random_polynomial(count, ...) on the library's host threads and return (a no-op under a patched
    os.urandom, whose draws must be made one by one in the reference's order)"""
    if count > 0 and os_urandom_is_genuine():
        _sc._check(_sc.lib().sc_urandom_prefetch(count, width))


def device_powers(base, count):
    """base^i, i < count, as a DeviceVector (Polynomial.scale of the all-ones vector: no host loop)"""
    ones = DeviceVector.from_bytes((1).to_bytes(16, "little") * count)
    out = DeviceVector(count)
    _sc._check(_sc.lib().sc_scale_dev(ones.ptr, out.ptr, count, _sc.fe_bytes(base.value), None))
    _sc.synchronize()
    return out


class FastStark:
    def __init__(self, field, expansion_factor, num_colinearity_checks, security_level, num_registers, num_cycles, transition_constraints_degree=2):
        assert(field.p.bit_length() >= security_level), "p must have at least as many bits as security level"
        assert(expansion_factor & (expansion_factor - 1) == 0), "expansion factor must be a power of 2"
        assert(expansion_factor >= 4), "expansion factor must be 4 or greater"
        assert(num_colinearity_checks * 2 >= security_level), "number of colinearity checks must be at least half of security level"

        # parameters (names are the reference's, fast_stark.py:14-35: callers read them)
        self.field, self.security_level = field, security_level
        self.expansion_factor, self.num_colinearity_checks = expansion_factor, num_colinearity_checks
        self.num_registers, self.original_trace_length = num_registers, num_cycles
        self.num_randomizers = 4 * num_colinearity_checks
        self.randomized_trace_length = num_cycles + self.num_randomizers

        # domains: the trace lives on <omicron>, the smallest power-of-two subgroup strictly larger than the degree of the AIR
        # substituted into the trace polynomials; FRI runs on the coset generator * <omega>, expansion_factor times larger
        self.omicron_domain_length = 1 << max(1, (self.randomized_trace_length * transition_constraints_degree).bit_length())
        self.fri_domain_length = self.omicron_domain_length * expansion_factor
        self.generator = field.generator()
        self.omega = field.primitive_nth_root(self.fri_domain_length)
        self.omicron = field.primitive_nth_root(self.omicron_domain_length)
        self._omicron_domain = None
        self._trace_domains = {}          # rows -> DeviceDomain of {omicron^i, i < rows} (progression tables, built once)
        self._lifted = {}                 # id(host Polynomial) -> (the Polynomial, its DevicePolynomial): lifted once, not per proof
        self._zerofier_values = {}        # transform order -> (the transition zerofier, its values on that order's coset)

        self.fri = Fri(self.generator, self.omega, self.fri_domain_length, expansion_factor, num_colinearity_checks)

    @property
    def omicron_domain(self):
        """omicron^i for i < omicron_domain_length (fast_stark.py:33), by running product instead of one exponentiation per
        entry; built when first read (a 2^22-entry list of objects is seconds of host time the device-resident prover never needs)"""
        if self._omicron_domain is None:
            domain, power = [], self.field.one()
            for _ in range(self.omicron_domain_length):
                domain.append(power)
                power = power * self.omicron
            self._omicron_domain = domain
        return self._omicron_domain

    def _trace_domain(self, rows):
        """{omicron^i, i < rows} as a DeviceDomain: a geometric progression, so interpolation through it is a handful of
        convolutions (csrc/geoseq.cuh) instead of a subproduct tree"""
        domain = self._trace_domains.get(rows)
        if domain is None:
            if len(self._trace_domains) >= 4:
                self._trace_domains.clear()
            domain = self._trace_domains[rows] = DeviceDomain.geometric(self.field.one(), self.omicron, rows)
        return domain

    def _lift(self, polynomial):
        """a host Polynomial (or an already device-resident one) as a DevicePolynomial; the same object is lifted once"""
        if isinstance(polynomial, DevicePolynomial):
            return polynomial
        hit = self._lifted.get(id(polynomial))
        if hit is not None and hit[0] is polynomial:
            return hit[1]
        if len(self._lifted) >= 16:
            self._lifted.clear()
        dev = DevicePolynomial.from_polynomial(polynomial, self.field)
        self._lifted[id(polynomial)] = (polynomial, dev)
        return dev

    # -- preprocessing (fast_stark.py:36-40) ------------------------------------------------------
    def preprocess(self, device_resident=False):
        """device_resident=True: the transition zerofier comes back as a DevicePolynomial (prove() takes either form) and no
        host list of the omicron domain is ever built -- the zerofier of {omicron^i, i < T - 1} has a closed form on the device"""
        if device_resident:
            assert(self.field.p == Field.P_MAIN), "the device-resident prover works in the main field"
            count = self.original_trace_length - 1
            if count >= 2:
                zerofier_domain = DeviceDomain.geometric(self.field.one(), self.omicron, count)
                transition_zerofier = DevicePolynomial.from_codeword(fast_zerofier_device(zerofier_domain))
            else:
                transition_zerofier = DevicePolynomial.from_polynomial(fast_zerofier(self.omicron_domain[:count], self.omicron, self.omicron_domain_length), self.field)
            transition_zerofier_codeword = transition_zerofier.coset_evaluate(self.generator, self.omega, self.fri_domain_length)
            return transition_zerofier, transition_zerofier_codeword, Merkle.commit(transition_zerofier_codeword)
        transition_zerofier = fast_zerofier(self.omicron_domain[:(self.original_trace_length - 1)], self.omicron, len(self.omicron_domain))
        if self.randomized_trace_length >= FastStark.DEVICE_MIN and self.field.p == Field.P_MAIN:
            # long traces: the codeword is committed here and opened in prove() where it lies, in HBM (a DeviceCodeword is list-like)
            transition_zerofier_codeword = fast_coset_evaluate_device(transition_zerofier, self.generator, self.omega, self.fri_domain_length)
        else:
            transition_zerofier_codeword = self._lde(transition_zerofier)
        transition_zerofier_root = Merkle.commit(transition_zerofier_codeword)
        return transition_zerofier, transition_zerofier_codeword, transition_zerofier_root

    # Traces of at least this many rows (randomizers included) are proved with every polynomial resident in HBM (see prove());
    # shorter ones follow the reference's host-list data flow with the GPU behind each fast_* call.  Same polynomials, same
    # objects pushed in the same order -- the proofs are byte-identical either way (tests/test_gpu_stark.py runs both settings
    # against the reference's golden proofs).
    DEVICE_MIN = 32
    # True: every commitment waits for its root before the prover goes on (the reference's order of events; A/B of proof_objects.RootLater)
    EAGER_COMMITS = False

    def _lde(self, polynomial):
        """Low-degree extension onto the FRI coset  generator * omega^i  (the LDE kernel)."""
        return fast_coset_evaluate(polynomial, self.generator, self.omega, self.fri_domain_length)

    # -- degree bookkeeping (fast_stark.py:42-56) ---------------------------------------------------
    def transition_degree_bounds(self, transition_constraints):
        """degree bound of every AIR polynomial once X (degree 1) and the 2 * num_registers trace polynomials (current and next
        row, degree randomized_trace_length - 1 each) are substituted for its variables: the worst monomial decides"""
        trace_degree = self.randomized_trace_length - 1
        variables = 1 + 2 * self.num_registers
        bounds = []
        for constraint in transition_constraints:
            monomial_degrees = [sum(exponents[:1]) + trace_degree * sum(exponents[1:variables]) for exponents in constraint.dictionary]
            bounds.append(max(monomial_degrees))
        return bounds

    def transition_quotient_degree_bounds(self, transition_constraints):
        # the transition zerofier vanishes on the first original_trace_length - 1 points of <omicron>
        zerofier_degree = self.original_trace_length - 1
        return [bound - zerofier_degree for bound in self.transition_degree_bounds(transition_constraints)]

    def max_degree(self, transition_constraints):
        # one less than the next power of two above the largest quotient bound (fast_stark.py:50-52)
        largest = max(self.transition_quotient_degree_bounds(transition_constraints))
        return (1 << max(1, largest.bit_length())) - 1

    def _boundary_points(self, boundary, register):
        """(domain points, values) of the boundary conditions (cycle, register, value) that concern `register`"""
        mine = [(cycle, value) for cycle, reg, value in boundary if reg == register]
        return [self.omicron ^ cycle for cycle, _ in mine], [value for _, value in mine]

    def boundary_zerofiers(self, boundary):
        return [Polynomial.zerofier_domain(self._boundary_points(boundary, s)[0]) for s in range(self.num_registers)]

    def boundary_interpolants(self, boundary):
        return [Polynomial.interpolate_domain(*self._boundary_points(boundary, s)) for s in range(self.num_registers)]

    def boundary_quotient_degree_bounds(self, randomized_trace_length, boundary):
        return [(randomized_trace_length - 1) - zerofier.degree() for zerofier in self.boundary_zerofiers(boundary)]

    def sample_weights(self, number, randomness):
        # bytes(i) is i zero bytes (fast_stark.py:74)
        return [self.field.sample(blake2b(randomness + bytes(i)).digest()) for i in range(0, number)]

    # a list here turns on the per-phase breakdown: after each phase of prove() the device is waited for and (phase, seconds since
    # the previous mark) is appended -- measurement only (tools/stark_phase_compare.py); the phases are sharded_stark's
    phase_log = None

    def _mark(self, phase):
        if self.phase_log is None:
            return
        import time
        _sc.synchronize()
        now = time.perf_counter()
        if phase is not None:
            self.phase_log.append((phase, now - self._phase_t0))
        self._phase_t0 = now

    # -- prover (fast_stark.py:76-178) -------------------------------------------------------------
    def prove(self, trace, transition_constraints, boundary, transition_zerofier, transition_zerofier_codeword, proof_stream=None):
        if proof_stream == None:
            proof_stream = ProofStream()
        field, registers = self.field, range(self.num_registers)
        self._mark(None)

        # randomizer rows appended to the trace (draw order: row by row, register by register); one concatenation instead of one
        # per row -- the caller's list is not touched either way
        if isinstance(trace, DeviceTrace):
            assert(field.p == Field.P_MAIN), "a device-resident trace lives in the main field"
            # the randomizer polynomial's draws (fast_stark.py:116-117: one os.urandom(17) per coefficient, 36 MB at a 2^24 FRI
            # domain) start now and pass while the GPU works on the trace
            prefetch_random_polynomial(self.max_degree(transition_constraints) + 1)
            columns = self._randomized_columns(trace, draw_random_bytes(self.num_randomizers * self.num_registers))
            trace_rows, on_device = len(trace) + self.num_randomizers, True
        else:
            trace = trace + [[field.sample(os.urandom(17)) for s in registers] for _ in range(self.num_randomizers)]
            trace_rows = len(trace)
            on_device = trace_rows >= FastStark.DEVICE_MIN and field.p == Field.P_MAIN
            if on_device:
                columns = [DeviceCodeword.from_list([row[s] for row in trace], field) for s in registers]
        interpolants = self.boundary_interpolants(boundary)
        zerofiers = self.boundary_zerofiers(boundary)
        # Checks the reference makes on the spot and nothing but an assertion reads -- a zero remainder of the boundary divisions
        # (univariate.py:99-103), "divide by zero" in the pointwise divisions (algebra.py:92) -- are decided on the device and COLLECTED
        # here; they are run where the prover has to wait for the device anyway, before the Fiat-Shamir challenge below, and raise
        # there what the reference raises at the division (the proof stream then holds the commitments pushed so far).
        pending = [] if on_device and not FastStark.EAGER_COMMITS else None
        if on_device:
            # Polynomials live in HBM from here on (DevicePolynomial): interpolation, boundary quotients (exact coset division,
            # exactness decided on the device), the AIR substitution in the value domain, the transition quotients, the LDEs and
            # the combination.  The host keeps what byte parity ties to it: os.urandom draws, Fiat-Shamir, the proof stream.
            trace_domain = self._trace_domain(trace_rows)
            trace_polynomials = [DevicePolynomial.from_codeword(fast_interpolate_device(trace_domain, column)) for column in columns]
            self._mark("trace interpolation")
            zerofiers_dev = [DevicePolynomial.from_polynomial(z, field) for z in zerofiers]
            boundary_quotients = [coset_divide_device(trace_polynomials[s].minus(interpolants[s]), zerofiers_dev[s], self.generator, self.omicron,
                                                      self.omicron_domain_length, exact=True, later=pending) for s in registers]
            lde = lambda poly: poly.coset_evaluate(self.generator, self.omega, self.fri_domain_length)
        else:
            # trace polynomials through {omicron^i}
            trace_domain = [self.omicron ^ i for i in range(trace_rows)]
            trace_polynomials = [fast_interpolate(trace_domain, [row[s] for row in trace], self.omicron, self.omicron_domain_length) for s in registers]
            # boundary quotients (exact schoolbook division by the small boundary zerofiers)
            boundary_quotients = [(trace_polynomials[s] - interpolants[s]) / zerofiers[s] for s in registers]
            lde = self._lde

        self._mark("boundary quotients (division)")
        # commit to their low-degree extensions
        boundary_quotient_codewords = []
        # a commitment whose codeword lives on the device is pushed as "the root of this tree, once it is built" (proof_objects.RootLater):
        # the stream's first reader -- the Fiat-Shamir challenge below -- waits for it, the GPU queue never does
        later = _po.lazy_objects(proof_stream) if on_device and not FastStark.EAGER_COMMITS else None

        def commit(codeword):
            if later is not None and isinstance(codeword, DeviceCodeword):
                later.add(_po.RootLater(codeword.start_tree()))
            else:
                proof_stream.push(Merkle.commit(codeword))
        for s in registers:
            boundary_quotient_codewords.append(lde(boundary_quotients[s]))
            commit(boundary_quotient_codewords[s])
        self._mark("boundary quotient LDEs + commitments")

        # transition polynomials: AIR evaluated symbolically in (X, trace(X), trace(omicron X)), then quotients
        x = Polynomial([field.zero(), field.one()])
        point = [DevicePolynomial.from_polynomial(x, field) if on_device else x] + trace_polynomials + \
                [tp.scaled_later(self.omicron) if on_device else tp.scale(self.omicron) for tp in trace_polynomials]
        if on_device:
            transition_quotients = self._transition_quotients_on_device(transition_constraints, point, self._lift(transition_zerofier), pending)
        else:
            transition_polynomials = [a.evaluate_symbolic(point) for a in transition_constraints]
            transition_quotients = [fast_coset_divide(tp, transition_zerofier, self.generator, self.omicron, self.omicron_domain_length) for tp in transition_polynomials]

        self._mark("AIR substitution + transition quotients (value domain)")
        # randomizer polynomial
        max_degree = self.max_degree(transition_constraints)
        if on_device:
            # the same draws (max_degree + 1 times os.urandom(17), fast_stark.py:117), sampled into HBM without a Python object each
            randomizer_polynomial = random_polynomial(max_degree + 1, field)
        else:
            randomizer_polynomial = Polynomial([field.sample(os.urandom(17)) for i in range(max_degree + 1)])
        self._mark("randomizer polynomial: os.urandom / getrandom draws and Field.sample")
        randomizer_codeword = lde(randomizer_polynomial)
        commit(randomizer_codeword)
        self._mark("randomizer polynomial: LDE, commitment")

        for verdict in pending or ():                    # the collected checks: the device has long decided them
            verdict()
        # Fiat-Shamir weights: 1 randomizer + 2 per transition quotient + 2 per boundary quotient
        weights = self.sample_weights(1 + 2 * len(transition_quotients) + 2 * len(boundary_quotients), proof_stream.prover_fiat_shamir())
        tq_bounds = self.transition_quotient_degree_bounds(transition_constraints)
        assert([tq.degree() for tq in transition_quotients] == tq_bounds), "transition quotient degrees do not match with expectation"

        # nonlinear combination: each quotient and its degree-shifted copy
        bq_bounds = self.boundary_quotient_degree_bounds(trace_rows, boundary)
        shifted = [(randomizer_polynomial, None)]
        for i, tq in enumerate(transition_quotients):
            shifted.append((tq, max_degree - tq_bounds[i]))
        for i in registers:
            shifted.append((boundary_quotients[i], max_degree - bq_bounds[i]))
        if on_device:
            combined_codeword = self._combine_on_device(shifted, weights, max_degree)
            self._mark("weights, degree checks, nonlinear combination + its LDE")
        else:
            terms = []
            for poly, shift in shifted:
                terms += [poly] if shift is None else [poly, (x ^ shift) * poly]
            combination = reduce(lambda a, b: a + b, [Polynomial([weights[i]]) * terms[i] for i in range(len(terms))], Polynomial([]))
            combined_codeword = self._lde(combination)

        # low-degree test of the combination; the openings of the committed codewords depend on the same sampled indices, so they
        # are fetched in the query phase's own device round trip (AlsoOpen) when every codeword lives on the device
        N = self.fri.domain_length
        committed = boundary_quotient_codewords + [randomizer_codeword, transition_zerofier_codeword]

        def opened_positions(indices):
            # the queried positions and their expansion_factor / half-domain companions (fast_stark.py:154-158)
            duplicated_indices = [i for i in indices] + [(i + self.expansion_factor) % N for i in indices]
            quadrupled_indices = [i for i in duplicated_indices] + [(i + (N // 2)) % N for i in duplicated_indices]
            quadrupled_indices.sort()
            return quadrupled_indices
        together = None
        if all(_po.eligible(codeword) for codeword in committed) and type(proof_stream) is ProofStream:
            together = AlsoOpen(lambda indices: (committed, [opened_positions(indices)] * len(committed)), codewords=committed, shift=self.expansion_factor)
        indices = self.fri.prove(combined_codeword, proof_stream, together) if together is not None else self.fri.prove(combined_codeword, proof_stream)
        self._mark("FRI: commit + query phases, openings fetched with them")

        quadrupled_indices = opened_positions(indices)
        lazy = _po.lazy_objects(proof_stream) if all(_po.eligible(codeword) for codeword in committed) else None
        if lazy is not None:
            # the device's answers as they are (proof_objects.Openings): same transcript bytes, no object per digest
            answers = together.answers if together is not None and together.answers is not None else _sc.query_codewords_raw(committed, [quadrupled_indices] * len(committed))
            arrays = together.position_arrays if together is not None and together.answers is answers and together.position_arrays else [None] * len(committed)
            for codeword, (values, paths), where in zip(committed, answers, arrays):
                lazy.add(_po.Openings(codeword, quadrupled_indices, values, paths, where))
        elif all(isinstance(codeword, DeviceCodeword) for codeword in committed):
            # every codeword's openings in ONE device round trip; pushed leaf, path, leaf, path, ... codeword by codeword
            for entries, paths in query_codewords(committed, [quadrupled_indices] * len(committed)):
                self._push_openings(entries, paths, proof_stream)
        else:
            for codeword in committed:
                self._open_all(codeword, quadrupled_indices, proof_stream)

        self._mark("openings of the committed codewords")
        proof = proof_stream.serialize()
        self._mark("proof serialization (host pickle)")
        return proof

    def _transition_quotients_on_device(self, constraints, point, tz_dev, pending=None):
        """fast_stark.py:107-113 -- `a.evaluate_symbolic(point)` divided by the transition zerofier -- without ever building the
        transition polynomial: on the coset g * <root'> (root' of the order code/ntt.py:155-157 shrinks to, taken from the degree
        BOUND) the point polynomials are evaluated once for all constraints of that order, the AIR is evaluated value by value
        (mpoly_eval_kernel), divided pointwise by the zerofier's values, and one inverse transform per constraint returns the
        quotient's coefficients: 4 + 2 transforms for the two-register AIR instead of the 9 + 4 of "substitute, then divide".
        The same polynomial as the reference's: an exact division gives the same quotient on any coset that is large enough, and
        its list length is its degree + 1.  Exactness is DECIDED, not assumed: the interpolant Q of the pointwise quotient satisfies
        Q * Z = transition polynomial as polynomials whenever deg Q <= bound - deg Z (both sides then have degree below the order
        and agree on the coset); a longer interpolant means the division is not exact (a false witness), and the reference's
        result then depends on the transition polynomial's true degree -- that constraint goes the reference's way
        (evaluate_symbolic, then coset_divide_device), as does any shape the kernel does not take (sharded_stark.py does the same
        on slabs)."""
        field = self.field
        reference_way = lambda a: coset_divide_device(a.evaluate_symbolic(point), tz_dev, self.generator, self.omicron, self.omicron_domain_length)
        degrees = [q.degree() for q in point]
        dr = tz_dev.degree()
        out, groups = [None] * len(constraints), {}
        for i, a in enumerate(constraints):
            plan = a.value_domain_terms(degrees)
            if plan is NotImplemented or dr < 0 or plan[0] < max(dr, MPolynomial.VALUE_DOMAIN_MIN_DEGREE):
                continue
            bound, terms = plan
            root, order = _shrink_order(self.omicron, self.omicron_domain_length, max(bound, dr))
            if len(tz_dev) > order or any(len(q) > order for q in point):
                continue
            groups.setdefault(order, (root, []))[1].append((i, bound, terms))
        lib, gen = _sc.lib(), _sc.fe_bytes(self.generator.value)
        for order, (root, members) in groups.items():
            nvars, rt = len(point), _sc.fe_bytes(root.value)
            used = [any(k[j] for _, _, terms in members for k, _ in terms) for j in range(nvars)]
            # q(root X) on the coset g <root> is q's codeword there, one place on: such a variable (the trace polynomials at
            # omicron X, fast_stark.py:105-106, whenever the coset's root is omicron itself) is read off its source's values
            # instead of being scaled and transformed
            turned = {}
            for j, q in enumerate(point):
                source = getattr(q, "scaled_from", None)
                if used[j] and source is not None and source[1].value == root.value:
                    k = next((k for k, other in enumerate(point) if other is source[0]), None)
                    if k is not None and getattr(point[k], "scaled_from", None) is None:
                        turned[j] = k
            stored = [(used[j] and j not in turned) or j in turned.values() for j in range(nvars)]
            vals = DeviceVector(nvars * order)
            for j, q in enumerate(point):
                if stored[j]:
                    source = getattr(q, "scaled_from", None)
                    if source is not None:
                        # q(f X) on g <root> is q on (g f) <root>: the transform's own offset does the scaling, the scaled
                        # coefficient vector is never made
                        shifted = _sc.fe_bytes((self.generator * source[1]).value)
                        _sc._check(lib.sc_coset_evaluate_dev(source[0].vec.ptr, degrees[j] + 1, shifted, rt, order, vals.ptr + 16 * j * order, None))
                    else:
                        _sc._check(lib.sc_coset_evaluate_dev(q.vec.ptr, degrees[j] + 1, gen, rt, order, vals.ptr + 16 * j * order, None))
            var_src = (ctypes.c_uint32 * nvars)(*[turned.get(j, j if stored[j] else 0xFFFFFFFF) for j in range(nvars)])
            var_rot = (ctypes.c_uint64 * nvars)(*[1 if j in turned else 0 for j in range(nvars)])
            kept = self._zerofier_values.get(order)
            if kept is not None and kept[0] is tz_dev:
                zvals = kept[1]
            else:
                zvals = DeviceVector(order)
                _sc._check(lib.sc_coset_evaluate_dev(tz_dev.vec.ptr, dr + 1, gen, rt, order, zvals.ptr, None))
                if len(self._zerofier_values) >= 4:
                    self._zerofier_values.clear()
                self._zerofier_values[order] = (tz_dev, zvals)
            converted = 0
            for i, bound, terms in members:
                exps = bytes(e for k, _ in terms for e in k)
                coefs = b"".join(v.to_bytes(16, "little") for _, v in terms)
                tvals, whole = DeviceVector(order), DeviceVector(order)
                _sc._check(lib.sc_mpoly_eval_rot_dev(vals.ptr, nvars, order, exps, coefs, len(terms), tvals.ptr, converted, var_src, var_rot, None))
                converted = 1
                self._pointwise_divide(tvals, zvals, order, pending)                                    # "divide by zero" like algebra.py:92
                _sc._check(lib.sc_ntt_dev(tvals.ptr, whole.ptr, order, rt, 1, None))
                _sc._check(lib.sc_scale_dev(whole.ptr, whole.ptr, order, _sc.fe_bytes(self.generator.inverse().value), None))
                quotient = DevicePolynomial(whole, field, order)
                degree = quotient.degree()
                if degree > bound - dr:
                    continue                                             # not exact: the reference's way decides what comes out
                out[i] = DevicePolynomial(whole, field, degree + 1) if degree >= 0 else DevicePolynomial(DeviceVector(1), field, 0)
                out[i]._degree = degree                                  # just read: the degree check of fast_stark.py:124 need not ask the device again
        return [q if q is not None else reference_way(a) for q, a in zip(out, constraints)]

    @staticmethod
    def _pointwise_divide(numerator, denominator, count, pending):
        """numerator[i] /= denominator[i] on the device; with a list of pending checks the "divide by zero" verdict is collected, not waited for"""
        lib = _sc.lib()
        if pending is not None:
            handle = ctypes.c_void_p()
            rc = lib.sc_pointwise_div_later_dev(numerator.ptr, denominator.ptr, numerator.ptr, count, ctypes.byref(handle), None)
            if rc != _sc.SC_ERR_UNSUPPORTED:
                _sc._check(rc)
                check = _sc.Later(handle)

                def verdict():
                    assert(not check.wait()[0]), "divide by zero"
                pending.append(verdict)
                return
        _sc._check(lib.sc_pointwise_div_dev(numerator.ptr, denominator.ptr, numerator.ptr, count, None))

    def _randomized_columns(self, trace, raw):
        """the columns of a DeviceTrace with the randomizer rows appended (fast_stark.py:79-81): `raw` holds the draws of
        os.urandom(17) in the reference's order -- row by row, register by register; 4 * num_colinearity_checks rows, sampled on
        the host and written behind each column's copy"""
        assert(len(trace.columns) == self.num_registers), "one column per register"
        width, rows, extra = self.num_registers, len(trace), self.num_randomizers
        # Field.sample (algebra.py:116-120: big-endian accumulate, then % p) of every 17-byte draw, without an object per draw
        p, big = self.field.p, int.from_bytes
        columns = []
        for s in range(width):
            tail = b"".join((big(raw[17 * (r * width + s):17 * (r * width + s) + 17], "big") % p).to_bytes(16, "little") for r in range(extra))
            column = DeviceVector(rows + extra)
            _sc._check(_sc.lib().sc_memcpy_dev(column.ptr, trace.columns[s].ptr, rows, None))
            if extra:
                _sc._check(_sc.lib().sc_vec_upload(column._h, rows, tail, extra))
            columns.append(DeviceCodeword(column, self.field))
        return columns

    def _combine_on_device(self, shifted, weights, max_degree):
        """sum_i weights[i] * terms[i] (fast_stark.py:130-145) as axpys over coefficient vectors in HBM, then the LDE straight from
        the accumulator: `Polynomial([w]) * t` scales t, `(x ^ k) * t` shifts it by k places.  The combination never visits the host."""
        return self._combination_on_device(shifted, weights, max_degree).coset_evaluate(self.generator, self.omega, self.fri_domain_length)

    def _combination_on_device(self, shifted, weights, max_degree):
        width = max(max_degree + 1, max(len(p) + (k or 0) for p, k in shifted))
        acc = DeviceVector.zeros(width)
        w = iter(weights)
        for poly, shift in shifted:
            for k in ([0] if shift is None else [0, shift]):
                weight = next(w)
                if len(poly):
                    acc.axpy_shift(_View(poly.vec, len(poly)), k, weight.value)
        return DevicePolynomial(acc, self.field, width)

    def _open_all(self, codeword, indices, proof_stream):
        """leaf, path, leaf, path, ... for one codeword -- one resident tree, one batched gather of all paths."""
        if isinstance(codeword, DeviceCodeword):
            entries, paths = codeword.query(indices)          # entries and paths in one device round trip
        else:
            entries, paths = [codeword[i] for i in indices], Merkle._tree(codeword).open_batch(indices)
        self._push_openings(entries, paths, proof_stream)

    @staticmethod
    def _push_openings(entries, paths, proof_stream):
        if type(proof_stream) is ProofStream:            # push == objects.append: one list extension for the whole codeword
            proof_stream.objects.extend(x for pair in zip(entries, paths) for x in pair)
            return
        for entry, path in zip(entries, paths):
            proof_stream.push(entry)
            proof_stream.push(path)

    # -- verifier (fast_stark.py:180-286) -----------------------------------------------------------
    def verify(self, proof, transition_constraints, boundary, transition_zerofier_root, proof_stream=None):
        stream = (ProofStream() if proof_stream == None else proof_stream).deserialize(proof)
        registers = range(self.num_registers)
        trace_rows = 1 + max(cycle for cycle, _, _ in boundary) + self.num_randomizers

        # the commitments, and the combination weights they determine
        quotient_roots = [stream.pull() for _ in registers]
        randomizer_root = stream.pull()
        weights = self.sample_weights(1 + 2 * len(transition_constraints) + 2 * self.num_registers, stream.verifier_fiat_shamir())

        # low-degree test of the combination; it reports the combination's values at the points it opened
        opened = []
        accepted = self.fri.verify(stream, opened)
        opened.sort(key=lambda index_value: index_value[0])
        if not accepted:
            return False

        # every committed codeword opened at those points and at their successors on the trace domain
        N, step = self.fri.domain_length, self.expansion_factor
        positions = sorted([i for i, _ in opened] + [(i + step) % N for i, _ in opened])
        quotient_leaves = []
        for root in quotient_roots:
            quotient_leaves.append(self._pull_openings(stream, root, positions))
            if quotient_leaves[-1] is None:
                return False
        randomizer = self._pull_openings(stream, randomizer_root, positions)
        if randomizer is None:
            return False
        zerofier_values = self._pull_openings(stream, transition_zerofier_root, positions)
        if zerofier_values is None:
            return False

        # the combination recomputed from the openings must agree with FRI's view of it at every queried point
        zerofiers, interpolants = self.boundary_zerofiers(boundary), self.boundary_interpolants(boundary)
        max_degree = self.max_degree(transition_constraints)
        transition_shifts = [max_degree - bound for bound in self.transition_quotient_degree_bounds(transition_constraints)]
        boundary_shifts = [max_degree - bound for bound in self.boundary_quotient_degree_bounds(trace_rows, boundary)]
        constraint_at = [constraint.evaluator() for constraint in transition_constraints]      # term lists extracted once, not per point

        def trace_row(index, x):
            # undo the boundary quotient: trace = quotient * zerofier + interpolant
            return [quotient_leaves[s][index] * zerofiers[s].evaluate(x) + interpolants[s].evaluate(x) for s in registers]

        for index, claimed in opened:
            successor = (index + step) % N
            x = self.generator * (self.omega ^ index)
            point = [x] + trace_row(index, x) + trace_row(successor, self.generator * (self.omega ^ successor))
            weight = iter(weights)
            total = randomizer[index] * next(weight)
            for evaluate, shift in zip(constraint_at, transition_shifts):
                quotient = evaluate(point) / zerofier_values[index]
                total = total + quotient * next(weight) + quotient * (x ^ shift) * next(weight)
            for s, shift in zip(registers, boundary_shifts):
                quotient = quotient_leaves[s][index]
                total = total + quotient * next(weight) + quotient * (x ^ shift) * next(weight)
            if not (total == claimed):
                return False
        return True

    @staticmethod
    def _pull_openings(stream, root, positions):
        """{position: leaf} from the (leaf, authentication path) pairs the prover pushed for one commitment, in the order of
        `positions`; None as soon as a path does not lead to `root`"""
        leaves = {}
        for position in positions:
            leaf, path = stream.pull(), stream.pull()
            if not Merkle.verify(root, position, path, leaf):
                return None
            leaves[position] = leaf
        return leaves
