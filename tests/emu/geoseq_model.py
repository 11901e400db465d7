"""Model of csrc/geoseq.cuh: fast_zerofier / fast_evaluate / fast_interpolate (code/ntt.py:66-130) when the domain is a
geometric progression  x_i = c * q^i, i < n  (the trace domain of code/fast_stark.py:84-90 is {omicron^i}).

Plain Python ints, array by array the way the device computes them (every list below is one device array, every loop one
kernel or one transform), so that the index and sign conventions are checked on the CPU against the oracle's restatement of the
reference recursion before they run on the GPU.  Test infrastructure only.

Facts used (Bostan-Schost, "Polynomial evaluation and interpolation on special sets of points", 2005; Bluestein 1970):
  t_j   = q^(j(j-1)/2)                     i*j = C(i+j, 2) - C(i, 2) - C(j, 2)
  A_i   = prod_{m=1..i} (q^m - 1)          prod_{j != i} (q^i - q^j) = (-1)^(n-1-i) * t_i * A_i * A_(n-1-i) * q^(i(n-1-i))
  Z(X)  = prod_{i<n} (X - q^i):  coefficient of X^(n-k) is (-1)^k * t_k * A_n / (A_k * A_(n-k))   (q-binomial theorem)
  P / Z = sum_m s_m X^(-m-1) with s_m = sum_i w_i q^(i m),  w_i = v_i / Z'(q^i)   (partial fractions, expanded at infinity)
"""
P = 1 + 407 * (1 << 119)


def inv(a):
    return pow(a, -1, P)


def ntt(root, v):
    n = len(v)
    if n == 1:
        return v[:]
    even, odd = ntt(root * root % P, v[0::2]), ntt(root * root % P, v[1::2])
    out, w = [0] * n, 1
    for i in range(n // 2):
        out[i] = (even[i] + w * odd[i]) % P
        out[i + n // 2] = (even[i] - w * odd[i]) % P
        w = w * root % P
    return out


_G119 = 85408008396924667383611388730472331217


def root_of_order(M):
    r = _G119
    for _ in range(119 - (M.bit_length() - 1)):
        r = r * r % P
    return r


def prefix_products(values):
    out, acc = [], 1
    for v in values:
        acc = acc * v % P
        out.append(acc)
    return out


class GeometricDomain:
    """tables of one domain {c * q^i, i < n}; n >= 2 and the points pairwise distinct (ord(q) >= n), else ValueError"""

    def __init__(self, c, q, n):
        assert n >= 2 and c % P and q % P
        self.c, self.q, self.n = c % P, q % P, n
        M = 1
        while M < 2 * n - 1:
            M *= 2
        self.M, self.w = M, root_of_order(M)
        qinv = inv(q)
        # scans: t_j = q^C(j,2) for j < M (input 1, q^0, q^1, ...), the same for 1/q
        self.t = prefix_products([1] + [pow(q, j - 1, P) for j in range(1, M)])
        self.tinv = prefix_products([1] + [pow(qinv, j - 1, P) for j in range(1, n)])
        # e_j = q^(j+1) - 1, j < n; A_(j+1) = inclusive scan; S_i = prod_{j=i..n-2} e_j  (so that 1/A_i = S_i / A_(n-1))
        e = [(pow(q, j + 1, P) - 1) % P for j in range(n)]
        A = [1] + prefix_products(e)                                   # A_0 .. A_n
        if A[n - 1] == 0:
            raise ValueError("the points of the progression are not distinct")
        rev = prefix_products([e[n - 2 - j] for j in range(n - 1)])    # rev[j] = S_(n-2-j)
        S = [rev[n - 2 - i] for i in range(n - 1)] + [1]               # S_0 .. S_(n-1)
        ia = inv(A[n - 1])
        g = inv(pow(q, n - 2, P)) if n >= 2 else 1
        # interpolation weights: 1 / (Z'(q^i) * t_i) = (-1)^(n-1-i) / (A_i * A_(n-1-i) * q^(i(n-2)))
        self.wden = [(-1 if (n - 1 - i) & 1 else 1) * ia * ia % P * S[i] % P * S[n - 1 - i] % P * pow(g, i, P) % P for i in range(n)]
        # reversed zerofier zr_k = coefficient of X^(n-k), k = 0..n
        zr = [1] + [(-1 if k & 1 else 1) * self.t[k] % P * A[n] % P * ia % P * ia % P * S[k] % P * S[n - k] % P for k in range(1, n)]
        zr.append((-1 if n & 1 else 1) * self.t[n] % P)
        self.zr = zr
        self.Bf = ntt(self.w, self.t)                                   # transform of t_0 .. t_(M-1)
        self.ZRf = ntt(self.w, zr[:n] + [0] * (M - n))

    def points(self):
        return [self.c * pow(self.q, i, P) % P for i in range(self.n)]

    def zerofier(self):
        """coefficients of prod (X - c q^i), low to high: z_j * c^(n-j)"""
        n = self.n
        return [self.zr[n - j] * pow(self.c, n - j, P) % P for j in range(n + 1)]

    def _correlate(self, a):
        """corr[m] = sum_i a_i t_(i+m), m < n, through the transforms: D[f] = A[-f] * B[f]"""
        M, n = self.M, self.n
        Af = ntt(self.w, a + [0] * (M - len(a)))
        D = [Af[(M - f) % M] * self.Bf[f] % P for f in range(M)]
        full = ntt(inv(self.w), D)
        minv = inv(M)
        return [x * minv % P for x in full[:n]]

    def _evaluate_chunk(self, coeffs):
        n = len(coeffs)
        assert n <= self.n
        a = [coeffs[j] * pow(self.c, j, P) % P * self.tinv[j] % P for j in range(n)]
        corr = self._correlate(a)
        return [corr[i] * self.tinv[i] % P for i in range(self.n)]

    def evaluate(self, coeffs):
        n = self.n
        if len(coeffs) <= n:
            return self._evaluate_chunk(list(coeffs))
        # longer polynomials in chunks of n coefficients: Horner over y_i = x_i^n = c^n * (q^n)^i
        y = [pow(self.c, n, P) * pow(pow(self.q, n, P), i, P) % P for i in range(n)]
        chunks = [coeffs[j:j + n] for j in range(0, len(coeffs), n)]
        acc = self._evaluate_chunk(chunks[-1])
        for chunk in reversed(chunks[:-1]):
            vals = self._evaluate_chunk(chunk)
            acc = [(acc[i] * y[i] + vals[i]) % P for i in range(n)]
        return acc

    def interpolate(self, values):
        n, M = self.n, self.M
        assert len(values) == n
        a = [values[i] * self.wden[i] % P for i in range(n)]
        corr = self._correlate(a)
        s = [corr[m] * self.tinv[m] % P for m in range(n)]
        Sf = ntt(self.w, s + [0] * (M - n))
        prod = ntt(inv(self.w), [Sf[f] * self.ZRf[f] % P for f in range(M)])
        minv = inv(M)
        qrev = [x * minv % P for x in prod[:n]]
        cinv = inv(self.c)
        return [qrev[n - 1 - j] * pow(cinv, j, P) % P for j in range(n)]
