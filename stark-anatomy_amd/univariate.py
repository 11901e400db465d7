"""Dense univariate polynomials over the field: host boundary type `Polynomial`.

Mirrors the interface and list-length conventions of reference code/univariate.py:3-160 (coefficient
lists are little-endian, may carry trailing zeros; `degree()` ignores them, equality ignores them).
Quadratic paths here (schoolbook multiply / long division / Lagrange) are the reference's own
small-degree fallbacks and test oracles; the fast paths live in ntt.py and run on the GPU.
Arithmetic is done on plain ints and wrapped back into FieldElement at the boundary.
"""
from algebra import *


def _wrap(ints, field):
    return [FieldElement(v, field) for v in ints]


class Polynomial:
    def __init__(self, coefficients):
        self.coefficients = [c for c in coefficients]

    # -- structure ---------------------------------------------------------------------------
    def degree(self):
        """Index of the last non-zero coefficient; -1 for the zero polynomial (univariate.py:7-17)."""
        for i in range(len(self.coefficients) - 1, -1, -1):
            if self.coefficients[i].value != 0:
                return i
        return -1

    def is_zero(self):
        return all(c.value == 0 for c in self.coefficients)

    def leading_coefficient(self):
        return self.coefficients[self.degree()]

    def __eq__(self, other):
        d = self.degree()
        if d != other.degree():
            return False
        if d == -1:
            return True
        return all(self.coefficients[i] == other.coefficients[i] for i in range(len(self.coefficients)))

    def __neq__(self, other):
        return not self.__eq__(other)

    __hash__ = None

    def __str__(self):
        return "[" + ",".join(str(c) for c in self.coefficients) + "]"

    # -- ring operations ---------------------------------------------------------------------
    def __neg__(self):
        return Polynomial([-c for c in self.coefficients])

    def __add__(self, other):
        # a zero operand hands back the other operand itself (univariate.py:22-26)
        if self.degree() == -1:
            return other
        if other.degree() == -1:
            return self
        field = self.coefficients[0].field
        p = field.p
        a, b = self.coefficients, other.coefficients
        out = [0] * max(len(a), len(b))
        for i, c in enumerate(a):
            out[i] = c.value
        for i, c in enumerate(b):
            out[i] = (out[i] + c.value) % p
        return Polynomial(_wrap(out, field))

    def __sub__(self, other):
        return self.__add__(-other)

    # operands at least this long go through the GPU transform instead of the schoolbook loop (same coefficients,
    # same list length; the threshold stays above fast_multiply's own "degree < 8 -> lhs * rhs" fallback)
    FAST_MUL_MIN_LEN = 32

    def __mul__(self, other):
        if self.coefficients == [] or other.coefficients == []:
            return Polynomial([])
        field = self.coefficients[0].field
        if min(len(self.coefficients), len(other.coefficients)) >= Polynomial.FAST_MUL_MIN_LEN and field.p == Field.P_MAIN:
            # Decide on DEGREES, not list lengths: lists may carry trailing zeros, and fast_multiply hands products of
            # degree < 8 straight back to `lhs * rhs` (ntt.py:44-45) -- taking the fast path for those would recurse forever.
            dl, dr = self.degree(), other.degree()
            if dl + dr >= 8:
                # Polynomial.__mul__ dominates MPolynomial.evaluate_symbolic (fast_stark.py:109-110) at scale; the product is
                # the same polynomial, so only the time changes
                from ntt import fast_multiply
                full_len = len(self.coefficients) + len(other.coefficients) - 1
                order = 1 << max(1, (dl + dr).bit_length())
                product = fast_multiply(self, other, field.primitive_nth_root(order), order).coefficients
                return Polynomial(product + [field.zero()] * (full_len - len(product)))
        return self._schoolbook_mul(other)

    def _schoolbook_mul(self, other):
        """univariate.py:48-57: len(a) + len(b) - 1 coefficients, trailing zeros included"""
        field = self.coefficients[0].field
        p = field.p
        b = [c.value for c in other.coefficients]
        out = [0] * (len(self.coefficients) + len(b) - 1)
        for i, c in enumerate(self.coefficients):
            x = c.value
            if x == 0:
                continue
            for j, y in enumerate(b):
                out[i + j] = (out[i + j] + x * y) % p
        return Polynomial(_wrap(out, field))

    def divide(numerator, denominator):
        """Schoolbook long division -> (quotient, remainder); None for a zero denominator (univariate.py:80-97)."""
        dd = denominator.degree()
        if dd == -1:
            return None
        dn = numerator.degree()
        if dn < dd:
            return (Polynomial([]), numerator)
        field = denominator.coefficients[0].field
        p = field.p
        den = [c.value for c in denominator.coefficients]
        # the reference's running remainder grows to the subtractee's length when the denominator list
        # carries trailing zeros; keep the same list length
        rem = [c.value for c in numerator.coefficients]
        rem += [0] * max(0, (dn - dd) + len(den) - len(rem))
        quo = [0] * (dn - dd + 1)
        lead_inv = pow(den[dd], -1, p)
        top = dn
        while top >= dd:
            c = rem[top] * lead_inv % p
            shift = top - dd
            quo[shift] = c
            for j in range(dd + 1):
                rem[shift + j] = (rem[shift + j] - c * den[j]) % p
            top -= 1
            while top >= 0 and rem[top] == 0:
                top -= 1
        return Polynomial(_wrap(quo, field)), Polynomial(_wrap(rem, field))

    def __truediv__(self, other):
        quo, rem = Polynomial.divide(self, other)
        assert(rem.is_zero()), "cannot perform polynomial division because remainder is not zero"
        return quo

    def __mod__(self, other):
        quo, rem = Polynomial.divide(self, other)
        return rem

    def __xor__(self, exponent):
        if self.is_zero():
            return Polynomial([])
        one = Polynomial([self.coefficients[0].field.one()])
        if exponent == 0:
            return one
        acc = one
        for i in reversed(range(exponent.bit_length())):
            acc = acc * acc
            if (exponent >> i) & 1:
                acc = acc * self
        return acc

    # -- evaluation / interpolation ----------------------------------------------------------
    def evaluate(self, point):
        field = point.field
        p = field.p
        x = point.value
        acc, xi = 0, 1
        for c in self.coefficients:
            acc = (acc + c.value * xi) % p
            xi = xi * x % p
        return FieldElement(acc, field)

    def evaluate_domain(self, domain):
        return [self.evaluate(d) for d in domain]

    def interpolate_domain(domain, values):
        """Lagrange interpolation (univariate.py:107-121)."""
        assert(len(domain) == len(values)), "number of elements in domain does not match number of values -- cannot interpolate"
        assert(len(domain) > 0), "cannot interpolate between zero points"
        field = domain[0].field
        x = Polynomial([field.zero(), field.one()])
        acc = Polynomial([])
        for i in range(len(domain)):
            prod = Polynomial([values[i]])
            for j in range(len(domain)):
                if j == i:
                    continue
                prod = prod * (x - Polynomial([domain[j]])) * Polynomial([(domain[i] - domain[j]).inverse()])
            acc = acc + prod
        return acc

    def zerofier_domain(domain):
        field = domain[0].field
        x = Polynomial([field.zero(), field.one()])
        acc = Polynomial([field.one()])
        for d in domain:
            acc = acc * (x - Polynomial([d]))
        return acc

    def scale(self, factor):
        """coefficient i times factor^i (univariate.py:153-154)."""
        if not self.coefficients:
            return Polynomial([])
        field = self.coefficients[0].field
        p = field.p
        out, f, acc = [], factor.value, 1
        for c in self.coefficients:
            out.append(acc * c.value % p)
            acc = acc * f % p
        return Polynomial(_wrap(out, field))


def test_colinearity(points):
    if len(points) == 3:
        (x0, y0), (x1, y1), (x2, y2) = points
        if x0 != x1 and x0 != x2 and x1 != x2:
            # univariate.py:159-163 interpolates and asks for degree == 1: three points with distinct abscissas lie on a line
            # of non-zero slope (a constant interpolant has degree 0, a parabola 2).  The verifier runs this 3 s (rounds - 1)
            # times (fri.py:207); the cross product is ~40x cheaper than the Lagrange interpolation.
            return (y1 - y0) * (x2 - x0) == (y2 - y0) * (x1 - x0) and y1 != y0
    domain = [p[0] for p in points]
    values = [p[1] for p in points]
    polynomial = Polynomial.interpolate_domain(domain, values)
    return polynomial.degree() == 1


test_colinearity.__test__ = False   # a helper (code/fri.py:207 calls it by this name), not a pytest case
