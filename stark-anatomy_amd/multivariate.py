"""Sparse multivariate polynomials (AIR bookkeeping for the STARK caller).

Host mirror of the interface of reference code/multivariate.py:3-123: `MPolynomial(dictionary)` maps exponent
tuples to FieldElement coefficients; `zero / constant / variables / lift`, `+ - * ^ neg`, `is_zero`,
`evaluate(point)` and `evaluate_symbolic(point)` (point = list of Polynomial).  `evaluate_symbolic` of large points runs
in the value domain on the GPU; everything else is host bookkeeping.
"""
from univariate import *


def _padded(key, width):
    return tuple(key) + (0,) * (width - len(key))


class MPolynomial:
    def __init__(self, dictionary):
        self.dictionary = dictionary

    def zero():
        return MPolynomial(dict())

    def constant(element):
        return MPolynomial({(0,): element})

    def variables(num_variables, field):
        one = field.one()
        return [MPolynomial({tuple(1 if j == i else 0 for j in range(num_variables)): one}) for i in range(num_variables)]

    def _width(self, other):
        return max([len(k) for k in self.dictionary] + [len(k) for k in other.dictionary])

    def __add__(self, other):
        width = self._width(other)
        acc = dict()
        for k, v in self.dictionary.items():
            acc[_padded(k, width)] = v
        for k, v in other.dictionary.items():
            key = _padded(k, width)
            acc[key] = acc[key] + v if key in acc else v
        return MPolynomial(acc)

    def __neg__(self):
        return MPolynomial({k: -v for k, v in self.dictionary.items()})

    def __sub__(self, other):
        return self + (-other)

    def __mul__(self, other):
        width = self._width(other)
        acc = dict()
        for k0, v0 in self.dictionary.items():
            p0 = _padded(k0, width)
            for k1, v1 in other.dictionary.items():
                p1 = _padded(k1, width)
                key = tuple(a + b for a, b in zip(p0, p1))
                term = v0 * v1
                acc[key] = acc[key] + term if key in acc else term
        return MPolynomial(acc)

    def __xor__(self, exponent):
        if self.is_zero():
            return MPolynomial(dict())
        field = next(iter(self.dictionary.values())).field
        width = len(next(iter(self.dictionary.keys())))
        acc = MPolynomial({(0,) * width: field.one()})
        for bit in bin(exponent)[2:]:
            acc = acc * acc
            if bit == '1':
                acc = acc * self
        return acc

    def is_zero(self):
        return all(v.is_zero() for v in self.dictionary.values())

    def evaluate(self, point):
        # multivariate.py:75-81 on residues: each power point[i]^e is computed once per call instead of once per term,
        # and factors with exponent 0 (the field's one) are not multiplied out -- same value
        return self.evaluator()(point)

    def evaluator(self):
        """point -> self.evaluate(point) for the polynomial AS IT IS NOW: the term list and the powers each call needs are
        extracted once.  The verifier evaluates every constraint at every queried point (fast_stark.py:201-205)."""
        terms = [(v.value, tuple((i, e) for i, e in enumerate(k) if e)) for k, v in self.dictionary.items()]
        needed = sorted({ie for _, f in terms for ie in f})

        def run(point):
            field = point[0].field
            p = field.p
            vals = [q.value for q in point]
            powers = {(i, e): pow(vals[i], e, p) for i, e in needed}
            acc = 0
            for coefficient, factors in terms:
                term = coefficient
                for ie in factors:
                    term = term * powers[ie] % p
                acc += term
            return FieldElement(acc % p, field)
        return run

    # degree bound of the result from which evaluate_symbolic goes through the value domain on the GPU
    VALUE_DOMAIN_MIN_DEGREE = 48

    def evaluate_symbolic(self, point):
        device = self._evaluate_symbolic_value_domain(point)
        if device is not None and device is not NotImplemented:
            return device
        to_device = None
        if any(hasattr(q, "vec") for q in point):
            # device operands the value-domain route declined (an exponent above 255, another field, more exponents than
            # variables -- NOT "every term vanishes", which it answers itself with a zero polynomial): the reference's sums of
            # products on the host, then back to HBM, because the caller works there
            from ntt import DevicePolynomial
            to_device = next(q.field for q in point if hasattr(q, "vec"))
            point = [q.to_polynomial() if hasattr(q, "vec") else q for q in point]
        acc = self._evaluate_symbolic_host(point)
        if to_device is not None:
            return DevicePolynomial.from_polynomial(acc, to_device)
        return acc

    def _evaluate_symbolic_host(self, point):
        # Same sums of products as multivariate.py:83-90.  The reference recomputes point[i] ^ e for every term; the powers
        # are the same polynomials each time, so they are computed once per call, and a factor that is the constant 1
        # (e = 0) is not multiplied out: `term * Polynomial([1])` has the same coefficient list as `term`.
        powers = {}
        acc = Polynomial([])
        for k, v in self.dictionary.items():
            term = Polynomial([v])
            for i, e in enumerate(k):
                power = powers.get((i, e))
                if power is None:
                    power = powers[(i, e)] = point[i] ^ e
                if len(power.coefficients) == 1 and power.coefficients[0].value == 1 and term.coefficients != []:
                    continue
                term = term * power
            acc = acc + term
        return acc

    def value_domain_terms(self, degrees):
        """What evaluating this polynomial POINTWISE on the values of point polynomials of the given degrees needs: (degree bound of
        the result, [(exponents padded to len(degrees), coefficient residue)] of the terms that do not vanish); bound -1: every
        term vanishes; NotImplemented for a shape mpoly_eval_kernel does not take."""
        nvars = len(degrees)
        if nvars > 255:
            return NotImplemented
        bound, terms = -1, []
        for k, v in self.dictionary.items():
            if len(k) > nvars or max(k, default=0) > 255:
                return NotImplemented
            d, dead = 0, False
            for i, e in enumerate(k):
                if degrees[i] < 0:
                    dead = True               # Polynomial.__xor__ returns the zero polynomial for a zero base whatever the
                    break                     # exponent, 0 included (univariate.py:139-140): the term vanishes
                d += e * degrees[i]
            if dead or v.value == 0:
                continue
            terms.append((tuple(k) + (0,) * (nvars - len(k)), v.value))
            bound = max(bound, d)
        return bound, terms

    def _evaluate_symbolic_value_domain(self, point):
        """The same polynomial computed pointwise: the point polynomials are evaluated on a power-of-two domain larger than
        the degree of the result (one zero-padded NTT each), the AIR is evaluated value by value (`mpoly_eval_kernel`), and
        one inverse NTT returns the coefficients.  The result is the unique polynomial the reference builds from schoolbook
        products; only its list length (trailing zeros) may differ, which no caller observes (fast_stark.py:110 divides it
        by the transition zerofier, which trims by degree).  Returns None when the point is too small to be worth it (host
        operands only), NotImplemented for a shape mpoly_eval_kernel does not take, and -- for device operands -- the zero
        polynomial when every term vanishes."""
        if not self.dictionary or not point:
            return None if not any(hasattr(q, "vec") for q in point) else NotImplemented
        on_device = [hasattr(q, "vec") for q in point]        # DevicePolynomial operands: coefficients already in HBM
        field = None
        for q, dev in zip(point, on_device):
            if dev:
                field = q.field
                break
            if q.coefficients:
                field = q.coefficients[0].field
                break
        if field is None or field.p != Field.P_MAIN:
            return NotImplemented if any(on_device) else None
        nvars = len(point)
        degs = [q.degree() for q in point]
        plan = self.value_domain_terms(degs)
        if plan is NotImplemented:
            return NotImplemented
        bound, terms = plan
        if bound < 0:
            # no live term: a zero factor kills its term whatever the exponent (univariate.py:139-140), zero coefficients add
            # nothing -- the sum is the zero polynomial
            if any(on_device):
                from ntt import DevicePolynomial
                import starkcore as _sc
                return DevicePolynomial(_sc.DeviceVector(1), field, 0)
            return None
        if bound < MPolynomial.VALUE_DOMAIN_MIN_DEGREE and not any(on_device):
            return None
        import starkcore as _sc
        n = 1 << max(1, bound.bit_length())            # > bound
        root = field.primitive_nth_root(n)
        one = _sc.fe_bytes(1)
        vals = _sc.DeviceVector(nvars * n)
        lib = _sc.lib()
        keep = []
        # only variables that occur in a live term are transformed: an unused one may be longer than the domain (n is sized
        # from the variables that do appear), and mpoly_eval_kernel never reads the values of a variable whose exponent is 0
        used = [any(k[j] for k, _ in terms) for j in range(nvars)]
        for j, q in enumerate(point):
            if not used[j]:
                continue
            m = degs[j] + 1
            if on_device[j]:
                src = q.vec
            else:
                src = _sc.DeviceVector.from_bytes(b"".join(c.value.to_bytes(16, "little") for c in q.coefficients[:m])) if m else _sc.DeviceVector(1)
            keep.append(src)
            _sc._check(lib.sc_coset_evaluate_dev(src.ptr, m, one, _sc.fe_bytes(root.value), n, vals.ptr + 16 * j * n, None))
        exps = bytes(e for k, _ in terms for e in k)
        coefs = b"".join(v.to_bytes(16, "little") for _, v in terms)
        out = _sc.DeviceVector(n)
        _sc._check(lib.sc_mpoly_eval_dev(vals.ptr, nvars, n, exps, coefs, len(terms), out.ptr, None))
        coeffs = _sc.DeviceVector(n)
        _sc._check(lib.sc_ntt_dev(out.ptr, coeffs.ptr, n, _sc.fe_bytes(root.value), 1, None))
        if any(on_device):
            from ntt import DevicePolynomial
            return DevicePolynomial(coeffs, field, bound + 1)     # stays in HBM for callers that work there
        raw = coeffs.to_bytes(0, bound + 1)
        frm = int.from_bytes
        return Polynomial([FieldElement(frm(raw[16 * i:16 * i + 16], "little"), field) for i in range(bound + 1)])

    def lift(polynomial, variable_index):
        if polynomial.is_zero():
            return MPolynomial({})
        field = polynomial.coefficients[0].field
        x = MPolynomial.variables(variable_index + 1, field)[-1]
        acc = MPolynomial({})
        for i, c in enumerate(polynomial.coefficients):
            acc = acc + MPolynomial.constant(c) * (x ^ i)
        return acc
