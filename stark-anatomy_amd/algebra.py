"""Field arithmetic boundary types: `Field`, `FieldElement`, `xgcd`.

Host-side mirror of the reference's code/algebra.py interface (same names, argument meaning, error
messages and -- because the Fiat-Shamir transcript pickles these objects, code/ip.py:18-25 -- the same
module name, class names and instance attributes: FieldElement{value, field}, Field{p}).
Values are canonical residues after every operation, exactly like algebra.py:75-94.
The heavy lifting (NTT, fold, Merkle) happens on the GPU through starkcore; these classes only carry
single values across the API.
"""


def xgcd(x, y):
    """Extended Euclid: returns (a, b, g) with a*x + b*y == g   (code/algebra.py:1-12)."""
    r0, r1 = x, y
    s0, s1 = 1, 0
    t0, t1 = 0, 1
    while r1 != 0:
        q = r0 // r1
        r0, r1 = r1, r0 - q * r1
        s0, s1 = s1, s0 - q * s1
        t0, t1 = t1, t0 - q * t1
    return s0, t0, r0


class FieldElement:
    def __init__(self, value, field):
        self.value = value
        self.field = field

    def __add__(self, right):
        return self.field.add(self, right)

    def __sub__(self, right):
        return self.field.subtract(self, right)

    def __mul__(self, right):
        return self.field.multiply(self, right)

    def __truediv__(self, right):
        return self.field.divide(self, right)

    def __neg__(self):
        return self.field.negate(self)

    def inverse(self):
        return self.field.inverse(self)

    def __xor__(self, exponent):
        # modular exponentiation (code/algebra.py:38-45); same residues as square-and-multiply
        return FieldElement(pow(self.value, exponent, self.field.p), self.field)

    def __eq__(self, other):
        return self.value == other.value

    def __neq__(self, other):
        return self.value != other.value

    def __str__(self):
        return str(self.value)

    def __bytes__(self):
        # decimal ASCII -- this is what Merkle leaves hash (code/algebra.py:56-57, code/merkle.py:14)
        return str(self.value).encode()

    def is_zero(self):
        return self.value == 0


class Field:
    P_MAIN = 1 + 407 * (1 << 119)
    G_MAIN = 85408008396924667383611388730472331217

    def __init__(self, p):
        self.p = p

    def zero(self):
        return FieldElement(0, self)

    def one(self):
        return FieldElement(1, self)

    def multiply(self, left, right):
        return FieldElement(left.value * right.value % self.p, self)

    def add(self, left, right):
        return FieldElement((left.value + right.value) % self.p, self)

    def subtract(self, left, right):
        return FieldElement((left.value - right.value) % self.p, self)

    def negate(self, operand):
        return FieldElement(-operand.value % self.p, self)

    def inverse(self, operand):
        # inverse(0) == 0 like the reference's xgcd(0, p) (code/algebra.py:87-89)
        a, _, _ = xgcd(operand.value, self.p)
        return FieldElement(a % self.p, self)

    def divide(self, left, right):
        assert(not right.is_zero()), "divide by zero"
        a, _, _ = xgcd(right.value, self.p)
        return FieldElement(left.value * a % self.p, self)

    def main():
        return Field(Field.P_MAIN)

    def generator(self):
        assert(self.p == Field.P_MAIN), "Do not know generator for other fields beyond 1+407*2^119"
        return FieldElement(Field.G_MAIN, self)

    def primitive_nth_root(self, n):
        if self.p == Field.P_MAIN:
            assert(n <= 1 << 119 and (n & (n - 1)) == 0), "Field does not have nth root of unity where n > 2^119 or not power of two."
            # G_MAIN has order 2^119: square it down to order n
            value, order = Field.G_MAIN, 1 << 119
            while order != n:
                value = value * value % self.p
                order >>= 1
            return FieldElement(value, self)
        else:
            assert(False), "Unknown field, can't return root of unity."

    def sample(self, byte_array):
        # algebra.py:123-127 folds the bytes in with acc = (acc << 8) ^ b: for byte values that is the big-endian integer
        # (only for byte strings: bytes(n) of an int n would be n zero bytes, where the reference's loop raises TypeError)
        if isinstance(byte_array, (bytes, bytearray, memoryview)):
            acc = int.from_bytes(byte_array, "big")
        else:
            acc = 0
            for b in byte_array:
                acc = (acc << 8) ^ int(b)
        return FieldElement(acc % self.p, self)
