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


def _last_nonzero(ints):
    k = len(ints) - 1
    while k >= 0 and ints[k] == 0:
        k -= 1
    return k


class Polynomial:
    def __init__(self, coefficients):
        self.coefficients = list(coefficients)

    def _values(self):
        return [c.value for c in self.coefficients]

    # -- structure ---------------------------------------------------------------------------
    def degree(self):
        """Index of the last non-zero coefficient; -1 for the zero polynomial (univariate.py:7-17)."""
        return _last_nonzero(self._values())

    def is_zero(self):
        return _last_nonzero(self._values()) < 0

    def leading_coefficient(self):
        # (the zero polynomial indexes with -1 like the reference: last list entry, IndexError on an empty list)
        return self.coefficients[self.degree()]

    def __eq__(self, rhs):
        """univariate.py:66-71: equal degrees, then the reference walks the WHOLE of the left list against the right one, so a
        left operand with more trailing zeros than the right list is long raises IndexError once the real coefficients agree."""
        mine, theirs = self._values(), rhs._values()
        top = _last_nonzero(mine)
        if top != _last_nonzero(theirs):
            return False
        if mine[:top + 1] != theirs[:top + 1]:
            return False
        if top >= 0 and len(mine) > len(theirs):
            raise IndexError("list index out of range")
        return True

    def __neq__(self, rhs):
        return not self == rhs

    __hash__ = None

    def __str__(self):
        return "[%s]" % ",".join(str(c) for c in self.coefficients)

    # -- ring operations ---------------------------------------------------------------------
    def __neg__(self):
        if not self.coefficients:
            return Polynomial([])
        field = self.coefficients[0].field
        return Polynomial(_wrap([(field.p - v) % field.p for v in self._values()], field))

    def __add__(self, rhs):
        # a zero operand hands back the other operand itself (univariate.py:22-26)
        if self.is_zero():
            return rhs
        if rhs.is_zero():
            return self
        field = self.coefficients[0].field
        p = field.p
        a, b = self._values(), rhs._values()
        if len(a) < len(b):
            a, b = b, a
        return Polynomial(_wrap([(v + b[i]) % p if i < len(b) else v for i, v in enumerate(a)], field))

    def __sub__(self, rhs):
        return self + (-rhs)

    # operands at least this long go through the GPU transform instead of the schoolbook loop (same coefficients,
    # same list length; the threshold stays above fast_multiply's own "degree < 8 -> lhs * rhs" fallback)
    FAST_MUL_MIN_LEN = 32

    def __mul__(self, rhs):
        if not self.coefficients or not rhs.coefficients:
            return Polynomial([])
        field = self.coefficients[0].field
        if min(len(self.coefficients), len(rhs.coefficients)) >= Polynomial.FAST_MUL_MIN_LEN and field.p == Field.P_MAIN:
            # Decide on DEGREES, not list lengths: lists may carry trailing zeros, and fast_multiply hands products of
            # degree < 8 straight back to `lhs * rhs` (ntt.py:44-45) -- taking the fast path for those would recurse forever.
            dl, dr = self.degree(), rhs.degree()
            if dl + dr >= 8:
                # Polynomial.__mul__ dominates MPolynomial.evaluate_symbolic (fast_stark.py:109-110) at scale; the product is
                # the same polynomial, so only the time changes
                from ntt import fast_multiply
                full_len = len(self.coefficients) + len(rhs.coefficients) - 1
                order = 1 << max(1, (dl + dr).bit_length())
                product = fast_multiply(self, rhs, field.primitive_nth_root(order), order).coefficients
                return Polynomial(product + [field.zero()] * (full_len - len(product)))
        return self._schoolbook_mul(rhs)

    def _schoolbook_mul(self, rhs):
        """univariate.py:48-57: len(a) + len(b) - 1 coefficients, trailing zeros included"""
        field = self.coefficients[0].field
        p = field.p
        b = rhs._values()
        out = [0] * (len(self.coefficients) + len(b) - 1)
        for i, x in enumerate(self._values()):
            if x:
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
        den = denominator._values()
        # the reference's running remainder grows to the subtractee's length when the denominator list
        # carries trailing zeros; keep the same list length
        rem = numerator._values()
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

    def __truediv__(self, rhs):
        quotient, remainder = self.divide(rhs)            # (a zero divisor: divide gives None and the unpacking raises TypeError)
        assert remainder.is_zero(), "cannot perform polynomial division because remainder is not zero"
        return quotient

    def __mod__(self, rhs):
        _, remainder = self.divide(rhs)
        return remainder

    def __xor__(self, exponent):
        """self^exponent; the list is exponent * (len - 1) + 1 long whatever the order of the products, so the binary method
        runs from the low bit here (univariate.py:140-150 runs from the high bit)."""
        if self.is_zero():
            return Polynomial([])
        power = Polynomial([self.coefficients[0].field.one()])
        square, e = self, exponent
        while e > 0:
            if e & 1:
                power = power * square
            e >>= 1
            if e:
                square = square * square
        return power

    # -- evaluation / interpolation ----------------------------------------------------------
    def evaluate(self, point):
        field = point.field
        p = field.p
        x = point.value
        acc = 0
        for v in reversed(self._values()):                # Horner
            acc = (acc * x + v) % p
        return FieldElement(acc, field)

    def evaluate_domain(self, domain):
        return [self.evaluate(d) for d in domain]

    @staticmethod
    def _zerofier_ints(xs, p):
        """coefficients of prod (X - x) over xs, low order first"""
        out = [1]
        for x in xs:
            out = [((out[k - 1] if k else 0) - (x * out[k] if k < len(out) else 0)) % p for k in range(len(out) + 1)]
        return out

    def interpolate_domain(domain, values):
        """Lagrange interpolation with the results (values AND list length len(domain)) of univariate.py:107-121, in
        O(n^2) instead of its O(n^3): the master polynomial Z = prod (X - x_j) once, then for every point the synthetic
        division Z / (X - x_i) weighted by y_i / prod_{j != i} (x_i - x_j).  The reference inverts each difference through
        Field.inverse, which maps 0 to 0 (algebra.py:87-89): a repeated abscissa silently drops its terms, and so it does here."""
        n = len(domain)
        assert n == len(values), "number of elements in domain does not match number of values -- cannot interpolate"
        assert n > 0, "cannot interpolate between zero points"
        field = domain[0].field
        p = field.p
        xs = [d.value for d in domain]
        master = Polynomial._zerofier_ints(xs, p)
        out = [0] * n
        for i, xi in enumerate(xs):
            denom = 1
            for j, xj in enumerate(xs):
                if j != i:
                    denom = denom * (xi - xj) % p
            weight = values[i].value * pow(denom, -1, p) % p if denom else 0
            if weight == 0:
                continue
            carry = 0
            for k in range(n, 0, -1):                      # quotient of master by (X - xi), high order first
                carry = (master[k] + xi * carry) % p
                out[k - 1] = (out[k - 1] + weight * carry) % p
        return Polynomial(_wrap(out, field))

    def zerofier_domain(domain):
        """prod (X - d) over the domain, len(domain) + 1 coefficients (univariate.py:123-128)"""
        field = domain[0].field
        return Polynomial(_wrap(Polynomial._zerofier_ints([d.value for d in domain], field.p), field))

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
    polynomial = Polynomial.interpolate_domain([x for x, _ in points], [y for _, y in points])
    return polynomial.degree() == 1


test_colinearity.__test__ = False   # a helper (code/fri.py:207 calls it by this name), not a pytest case
