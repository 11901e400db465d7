"""NTT-based STARK prover / verifier -- the main CALLER of the GPU hot path.

Host mirror of the interface of reference code/fast_stark.py:8-286:
`FastStark(field, expansion_factor, num_colinearity_checks, security_level, num_registers, num_cycles,
transition_constraints_degree=2)` with `preprocess / prove / verify` and the degree-bound helpers.  Every
`fast_zerofier`, `fast_interpolate`, `fast_coset_evaluate`, `fast_coset_divide`, `Merkle.commit/open` and
`Fri.prove` inside goes through ntt.py / merkle.py / fri.py to the MI355X.  The order of `os.urandom` draws and of
`proof_stream.push` calls is the reference's, so with a patched `fast_stark.os.urandom` the proof bytes are identical.
"""
from functools import reduce
import os

from fri import *
from univariate import *
from multivariate import *
from ntt import *
from ntt import _View


def draw_random_bytes(count, width=17):
    """`count` draws of os.urandom(width) as one byte string, in draw order.  Long runs are drawn in blocks: os.urandom(n) is
    the next n bytes of the stream, so the bytes and their order are those of the individual draws."""
    block = 4096
    return b"".join(os.urandom(width * min(block, count - i)) for i in range(0, count, block))


def sampled_polynomial(raw, field, width=17):
    """Polynomial([field.sample(raw[17 i : 17 i + 17]) ...]) as a DevicePolynomial: Field.sample on the device (sc_sample_bytes_dev)"""
    import starkcore as _sc
    count = len(raw) // width
    vec = DeviceVector(max(count, 1))
    _sc._check(_sc.lib().sc_sample_bytes_dev(raw, count, width, vec.ptr, None))
    return DevicePolynomial(vec, field, count)


def device_powers(base, count):
    """base^i, i < count, as a DeviceVector (Polynomial.scale of the all-ones vector: no host loop)"""
    import starkcore as _sc
    ones = DeviceVector.from_bytes((1).to_bytes(16, "little") * count)
    out = DeviceVector(count)
    _sc._check(_sc.lib().sc_scale_dev(ones.ptr, out.ptr, count, _sc.fe_bytes(base.value), None))
    _sc.synchronize()
    return out


class FastStark:
    def __init__(self, field, expansion_factor, num_colinearity_checks, security_level, num_registers, num_cycles, transition_constraints_degree=2):
        assert(len(bin(field.p)) - 2 >= security_level), "p must have at least as many bits as security level"
        assert(expansion_factor & (expansion_factor - 1) == 0), "expansion factor must be a power of 2"
        assert(expansion_factor >= 4), "expansion factor must be 4 or greater"
        assert(num_colinearity_checks * 2 >= security_level), "number of colinearity checks must be at least half of security level"

        self.field = field
        self.expansion_factor = expansion_factor
        self.num_colinearity_checks = num_colinearity_checks
        self.security_level = security_level
        self.num_randomizers = 4 * num_colinearity_checks
        self.num_registers = num_registers
        self.original_trace_length = num_cycles
        self.randomized_trace_length = self.original_trace_length + self.num_randomizers
        # smallest power of two strictly above randomized_trace_length * constraint degree (fast_stark.py:27)
        self.omicron_domain_length = 1 << len(bin(self.randomized_trace_length * transition_constraints_degree)[2:])
        self.fri_domain_length = self.omicron_domain_length * expansion_factor

        self.generator = self.field.generator()
        self.omega = self.field.primitive_nth_root(self.fri_domain_length)
        self.omicron = self.field.primitive_nth_root(self.omicron_domain_length)
        # omicron^i for i < omicron_domain_length (fast_stark.py:33), by running product instead of one exponentiation per entry
        self.omicron_domain, acc = [], self.field.one()
        for _ in range(self.omicron_domain_length):
            self.omicron_domain.append(acc)
            acc = acc * self.omicron

        self.fri = Fri(self.generator, self.omega, self.fri_domain_length, self.expansion_factor, self.num_colinearity_checks)

    # -- preprocessing (fast_stark.py:36-40) ------------------------------------------------------
    def preprocess(self):
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

    def _lde(self, polynomial):
        """Low-degree extension onto the FRI coset  generator * omega^i  (the LDE kernel)."""
        return fast_coset_evaluate(polynomial, self.generator, self.omega, self.fri_domain_length)

    # -- degree bookkeeping (fast_stark.py:42-56) ---------------------------------------------------
    def transition_degree_bounds(self, transition_constraints):
        point_degrees = [1] + [self.original_trace_length + self.num_randomizers - 1] * 2 * self.num_registers
        return [max(sum(r * l for r, l in zip(point_degrees, k)) for k, v in a.dictionary.items()) for a in transition_constraints]

    def transition_quotient_degree_bounds(self, transition_constraints):
        return [d - (self.original_trace_length - 1) for d in self.transition_degree_bounds(transition_constraints)]

    def max_degree(self, transition_constraints):
        md = max(self.transition_quotient_degree_bounds(transition_constraints))
        return (1 << (len(bin(md)[2:]))) - 1

    def boundary_zerofiers(self, boundary):
        return [Polynomial.zerofier_domain([self.omicron ^ c for c, r, v in boundary if r == s]) for s in range(self.num_registers)]

    def boundary_interpolants(self, boundary):
        interpolants = []
        for s in range(self.num_registers):
            points = [(c, v) for c, r, v in boundary if r == s]
            interpolants.append(Polynomial.interpolate_domain([self.omicron ^ c for c, v in points], [v for c, v in points]))
        return interpolants

    def boundary_quotient_degree_bounds(self, randomized_trace_length, boundary):
        randomized_trace_degree = randomized_trace_length - 1
        return [randomized_trace_degree - bz.degree() for bz in self.boundary_zerofiers(boundary)]

    def sample_weights(self, number, randomness):
        # bytes(i) is i zero bytes (fast_stark.py:74)
        return [self.field.sample(blake2b(randomness + bytes(i)).digest()) for i in range(0, number)]

    # -- prover (fast_stark.py:76-178) -------------------------------------------------------------
    def prove(self, trace, transition_constraints, boundary, transition_zerofier, transition_zerofier_codeword, proof_stream=None):
        if proof_stream == None:
            proof_stream = ProofStream()
        field, registers = self.field, range(self.num_registers)

        # randomizer rows appended to the trace (draw order: row by row, register by register); one concatenation instead of one
        # per row -- the caller's list is not touched either way
        trace = trace + [[field.sample(os.urandom(17)) for s in registers] for _ in range(self.num_randomizers)]

        on_device = len(trace) >= FastStark.DEVICE_MIN and field.p == Field.P_MAIN
        interpolants = self.boundary_interpolants(boundary)
        zerofiers = self.boundary_zerofiers(boundary)
        if on_device:
            # Polynomials live in HBM from here on (DevicePolynomial): interpolation, boundary quotients (exact coset division,
            # exactness decided on the device), the AIR substitution in the value domain, the transition quotients, the LDEs and
            # the combination.  The host keeps what byte parity ties to it: os.urandom draws, Fiat-Shamir, the proof stream.
            trace_domain = DeviceDomain(device_powers(self.omicron, len(trace)), field)
            trace_polynomials = [DevicePolynomial.from_codeword(fast_interpolate_device(trace_domain, DeviceCodeword.from_list([row[s] for row in trace], field)))
                                 for s in registers]
            zerofiers_dev = [DevicePolynomial.from_polynomial(z, field) for z in zerofiers]
            boundary_quotients = [coset_divide_device(trace_polynomials[s].minus(interpolants[s]), zerofiers_dev[s], self.generator, self.omicron,
                                                      self.omicron_domain_length, exact=True) for s in registers]
            lde = lambda poly: poly.coset_evaluate(self.generator, self.omega, self.fri_domain_length)
        else:
            # trace polynomials through {omicron^i}
            trace_domain = [self.omicron ^ i for i in range(len(trace))]
            trace_polynomials = [fast_interpolate(trace_domain, [row[s] for row in trace], self.omicron, self.omicron_domain_length) for s in registers]
            # boundary quotients (exact schoolbook division by the small boundary zerofiers)
            boundary_quotients = [(trace_polynomials[s] - interpolants[s]) / zerofiers[s] for s in registers]
            lde = self._lde

        # commit to their low-degree extensions
        boundary_quotient_codewords = []
        for s in registers:
            boundary_quotient_codewords.append(lde(boundary_quotients[s]))
            proof_stream.push(Merkle.commit(boundary_quotient_codewords[s]))

        # transition polynomials: AIR evaluated symbolically in (X, trace(X), trace(omicron X)), then quotients
        x = Polynomial([field.zero(), field.one()])
        point = [DevicePolynomial.from_polynomial(x, field) if on_device else x] + trace_polynomials + [tp.scale(self.omicron) for tp in trace_polynomials]
        transition_polynomials = [a.evaluate_symbolic(point) for a in transition_constraints]
        if on_device:
            tz_dev = DevicePolynomial.from_polynomial(transition_zerofier, field)
            transition_quotients = [coset_divide_device(tp, tz_dev, self.generator, self.omicron, self.omicron_domain_length) for tp in transition_polynomials]
        else:
            transition_quotients = [fast_coset_divide(tp, transition_zerofier, self.generator, self.omicron, self.omicron_domain_length) for tp in transition_polynomials]

        # randomizer polynomial
        max_degree = self.max_degree(transition_constraints)
        if on_device:
            # the same draws (max_degree + 1 times os.urandom(17), fast_stark.py:117), sampled into HBM without a Python object each
            randomizer_polynomial = sampled_polynomial(draw_random_bytes(max_degree + 1), field)
        else:
            randomizer_polynomial = Polynomial([field.sample(os.urandom(17)) for i in range(max_degree + 1)])
        randomizer_codeword = lde(randomizer_polynomial)
        proof_stream.push(Merkle.commit(randomizer_codeword))

        # Fiat-Shamir weights: 1 randomizer + 2 per transition quotient + 2 per boundary quotient
        weights = self.sample_weights(1 + 2 * len(transition_quotients) + 2 * len(boundary_quotients), proof_stream.prover_fiat_shamir())
        tq_bounds = self.transition_quotient_degree_bounds(transition_constraints)
        assert([tq.degree() for tq in transition_quotients] == tq_bounds), "transition quotient degrees do not match with expectation"

        # nonlinear combination: each quotient and its degree-shifted copy
        bq_bounds = self.boundary_quotient_degree_bounds(len(trace), boundary)
        shifted = [(randomizer_polynomial, None)]
        for i, tq in enumerate(transition_quotients):
            shifted.append((tq, max_degree - tq_bounds[i]))
        for i in registers:
            shifted.append((boundary_quotients[i], max_degree - bq_bounds[i]))
        if on_device:
            combined_codeword = self._combine_on_device(shifted, weights, max_degree)
        else:
            terms = []
            for poly, shift in shifted:
                terms += [poly] if shift is None else [poly, (x ^ shift) * poly]
            combination = reduce(lambda a, b: a + b, [Polynomial([weights[i]]) * terms[i] for i in range(len(terms))], Polynomial([]))
            combined_codeword = self._lde(combination)

        # low-degree test of the combination
        indices = self.fri.prove(combined_codeword, proof_stream)

        # open the queried positions (and their expansion_factor / half-domain companions)
        N = self.fri.domain_length
        duplicated_indices = [i for i in indices] + [(i + self.expansion_factor) % N for i in indices]
        quadrupled_indices = [i for i in duplicated_indices] + [(i + (N // 2)) % N for i in duplicated_indices]
        quadrupled_indices.sort()
        for codeword in boundary_quotient_codewords + [randomizer_codeword, transition_zerofier_codeword]:
            self._open_all(codeword, quadrupled_indices, proof_stream)

        return proof_stream.serialize()

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
        if type(proof_stream) is ProofStream:            # push == objects.append: one list extension for the whole codeword
            proof_stream.objects.extend(x for pair in zip(entries, paths) for x in pair)
            return
        for entry, path in zip(entries, paths):
            proof_stream.push(entry)
            proof_stream.push(path)

    # -- verifier (fast_stark.py:180-286) -----------------------------------------------------------
    def verify(self, proof, transition_constraints, boundary, transition_zerofier_root, proof_stream=None):
        original_trace_length = 1 + max(c for c, r, v in boundary)
        randomized_trace_length = original_trace_length + self.num_randomizers

        if proof_stream == None:
            proof_stream = ProofStream()
        proof_stream = proof_stream.deserialize(proof)

        boundary_quotient_roots = [proof_stream.pull() for s in range(self.num_registers)]
        randomizer_root = proof_stream.pull()
        interpolants = self.boundary_interpolants(boundary)
        zerofiers = self.boundary_zerofiers(boundary)
        weights = self.sample_weights(1 + 2 * len(transition_constraints) + 2 * len(interpolants), proof_stream.verifier_fiat_shamir())

        polynomial_values = []
        verifier_accepts = self.fri.verify(proof_stream, polynomial_values)
        polynomial_values.sort(key=lambda iv: iv[0])
        if not verifier_accepts:
            return False
        indices = [i for i, v in polynomial_values]
        values = [v for i, v in polynomial_values]

        N = self.fri.domain_length
        duplicated_indices = [i for i in indices] + [(i + self.expansion_factor) % N for i in indices]
        duplicated_indices.sort()

        def read_leafs(root):
            table = dict()
            for i in duplicated_indices:
                table[i] = proof_stream.pull()
                path = proof_stream.pull()
                if not Merkle.verify(root, i, path, table[i]):
                    return None
            return table

        leafs = []
        for root in boundary_quotient_roots:
            table = read_leafs(root)
            if table is None:
                return False
            leafs.append(table)
        randomizer = read_leafs(randomizer_root)
        if randomizer is None:
            return False
        transition_zerofier = read_leafs(transition_zerofier_root)
        if transition_zerofier is None:
            return False

        max_degree = self.max_degree(transition_constraints)
        tq_bounds = self.transition_quotient_degree_bounds(transition_constraints)
        bq_bounds = self.boundary_quotient_degree_bounds(randomized_trace_length, boundary)
        constraint_at = [tc.evaluator() for tc in transition_constraints]      # term lists extracted once, not per queried point
        for position, current_index in enumerate(indices):
            next_index = (current_index + self.expansion_factor) % N
            x_current = self.generator * (self.omega ^ current_index)
            x_next = self.generator * (self.omega ^ next_index)
            # undo the boundary quotient to recover the trace values at both points
            current_trace = [leafs[s][current_index] * zerofiers[s].evaluate(x_current) + interpolants[s].evaluate(x_current) for s in range(self.num_registers)]
            next_trace = [leafs[s][next_index] * zerofiers[s].evaluate(x_next) + interpolants[s].evaluate(x_next) for s in range(self.num_registers)]
            point = [x_current] + current_trace + next_trace
            constraint_values = [at(point) for at in constraint_at]

            terms = [randomizer[current_index]]
            for s, tcv in enumerate(constraint_values):
                quotient = tcv / transition_zerofier[current_index]
                terms += [quotient, quotient * (x_current ^ (max_degree - tq_bounds[s]))]
            for s in range(self.num_registers):
                bqv = leafs[s][current_index]
                terms += [bqv, bqv * (x_current ^ (max_degree - bq_bounds[s]))]
            combination = reduce(lambda a, b: a + b, [terms[j] * weights[j] for j in range(len(terms))], self.field.zero())
            if not (combination == values[position]):
                return False
        return True
