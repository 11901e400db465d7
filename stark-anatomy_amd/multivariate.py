"""Sparse multivariate polynomials (AIR bookkeeping for the STARK caller).

Host mirror of the interface of reference code/multivariate.py:3-123: `MPolynomial(dictionary)` maps exponent
tuples to FieldElement coefficients; `zero / constant / variables / lift`, `+ - * ^ neg`, `is_zero`,
`evaluate(point)` and `evaluate_symbolic(point)` (point = list of Polynomial).  Not on the GPU hot path.
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
        acc = point[0].field.zero()
        for k, v in self.dictionary.items():
            term = v
            for i, e in enumerate(k):
                term = term * (point[i] ^ e)
            acc = acc + term
        return acc

    def evaluate_symbolic(self, point):
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

    def lift(polynomial, variable_index):
        if polynomial.is_zero():
            return MPolynomial({})
        field = polynomial.coefficients[0].field
        x = MPolynomial.variables(variable_index + 1, field)[-1]
        acc = MPolynomial({})
        for i, c in enumerate(polynomial.coefficients):
            acc = acc + MPolynomial.constant(c) * (x ^ i)
        return acc
