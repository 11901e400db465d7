"""Field arithmetic boundary types: `Field`, `FieldElement`, `xgcd`.

Host-side counterpart of the reference's code/algebra.py interface (same names, argument meaning, error
messages and -- because the Fiat-Shamir transcript pickles these objects, code/ip.py:18-25 -- the same
module name, class names and instance attributes: FieldElement{value, field}, Field{p}).
Values are canonical residues after every operation, exactly like algebra.py:75-94.
The heavy lifting (NTT, fold, Merkle) happens on the GPU through starkcore; these classes only carry
single values across the API, so the arithmetic here is plain Python integers: an element's operators work
on the residues directly, and `Field`'s methods (the reference's algebra.py:75-94 spelling of the same
operations) are thin views of them.
"""

P_MAIN = 1 + 407 * (1 << 119)
# "generator" of the reference (algebra.py:100-102): an element of multiplicative order exactly 2^119
G_MAIN = 85408008396924667383611388730472331217
LOG_ORDER_MAIN = 119


def xgcd(x, y):
    """Bezout coefficients: (a, b, g) with a*x + b*y == g = gcd   (interface of code/algebra.py:1-12; same outputs:
    the remainder sequence of Euclid's algorithm is unique, and so are the cofactors that go with it)."""
    rows = [(x, 1, 0), (y, 0, 1)]                  # (remainder, cofactor of x, cofactor of y)
    while rows[1][0] != 0:
        (r_old, a_old, b_old), (r_new, a_new, b_new) = rows
        quotient = r_old // r_new
        rows = [rows[1], (r_old - quotient * r_new, a_old - quotient * a_new, b_old - quotient * b_new)]
    g, a, b = rows[0]
    return a, b, g


def _inverse_residue(value, p):
    """value^-1 mod p, with the reference's convention inverse(0) == 0 (xgcd(0, p) gives the cofactor 0, algebra.py:87-89)"""
    value %= p
    if not value:
        return 0
    try:
        return pow(value, -1, p)
    except ValueError:
        # not invertible (composite modulus): the reference returns the Bezout cofactor of its xgcd without complaint (algebra.py:87-89)
        return xgcd(value, p)[0] % p


class FieldElement:
    def __init__(self, value, field):
        self.value = value
        self.field = field

    def _with(self, residue):
        return FieldElement(residue, self.field)

    # -- ring operations: canonical residue after each one
    def __add__(self, right):
        return self._with((self.value + right.value) % self.field.p)

    def __sub__(self, right):
        return self._with((self.value - right.value) % self.field.p)

    def __neg__(self):
        return self._with(-self.value % self.field.p)

    def __mul__(self, right):
        return self._with(self.value * right.value % self.field.p)

    def __truediv__(self, right):
        assert(not right.is_zero()), "divide by zero"
        p = self.field.p
        return self._with(self.value * _inverse_residue(right.value, p) % p)

    def inverse(self):
        return self._with(_inverse_residue(self.value, self.field.p))

    def __xor__(self, exponent):
        # the reference writes exponentiation as `^` (square-and-multiply, code/algebra.py:38-45); same residues
        return self._with(pow(self.value, exponent, self.field.p))

    # -- comparisons and conversions
    def __eq__(self, other):
        return self.value == other.value

    def __neq__(self, other):
        return self.value != other.value

    def is_zero(self):
        return self.value == 0

    def __str__(self):
        return str(self.value)

    def __bytes__(self):
        # decimal ASCII -- this is what Merkle leaves hash (code/algebra.py:56-57, code/merkle.py:14)
        return b"%d" % self.value


class Field:
    P_MAIN = P_MAIN
    G_MAIN = G_MAIN

    def __init__(self, p):
        self.p = p

    def main():
        return Field(P_MAIN)

    # -- constants
    def zero(self):
        return FieldElement(0, self)

    def one(self):
        return FieldElement(1, self)

    def generator(self):
        assert(self.p == P_MAIN), "Do not know generator for other fields beyond 1+407*2^119"
        return FieldElement(G_MAIN, self)

    def primitive_nth_root(self, n):
        if self.p != P_MAIN:
            assert(False), "Unknown field, can't return root of unity."
        assert(n <= 1 << LOG_ORDER_MAIN and (n & (n - 1)) == 0), "Field does not have nth root of unity where n > 2^119 or not power of two."
        # G_MAIN has order 2^119: squaring it 119 - log2(n) times leaves order n (algebra.py:107-111)
        squarings = LOG_ORDER_MAIN - (n.bit_length() - 1)
        return FieldElement(pow(G_MAIN, 1 << squarings, self.p), self)

    # -- the operations by name (what the operators of FieldElement do, results tagged with THIS field)
    def add(self, left, right):
        return FieldElement((left.value + right.value) % self.p, self)

    def subtract(self, left, right):
        return FieldElement((left.value - right.value) % self.p, self)

    def negate(self, operand):
        return FieldElement(-operand.value % self.p, self)

    def multiply(self, left, right):
        return FieldElement(left.value * right.value % self.p, self)

    def inverse(self, operand):
        return FieldElement(_inverse_residue(operand.value, self.p), self)

    def divide(self, left, right):
        assert(not right.is_zero()), "divide by zero"
        return FieldElement(left.value * _inverse_residue(right.value, self.p) % self.p, self)

    def sample(self, byte_array):
        # algebra.py:116-120 folds the bytes in with acc = (acc << 8) ^ b: for byte values that is the big-endian integer
        # (only for byte strings: bytes(n) of an int n would be n zero bytes, where the reference's loop raises TypeError)
        if isinstance(byte_array, (bytes, bytearray, memoryview)):
            acc = int.from_bytes(byte_array, "big")
        else:
            acc = 0
            for b in byte_array:
                acc = (acc << 8) ^ int(b)
        return FieldElement(acc % self.p, self)
