"""Executable model of the device algorithm behind sc_polytree_* (csrc/polytree.cuh): the subproduct tree, multipoint
evaluation and interpolation of code/ntt.py:66-130 as LEVEL-BATCHED transforms over a perfect binary tree.

The reference recurses node by node (split at len//2, schoolbook remainders); the results -- zerofier, values, the
interpolant of degree < k -- are unique, so the device is free to compute them another way:

  * the k points are padded with zeros to K = 2^L leaves (a zero leaf multiplies the zerofier by x: strip `pad` low coefficients);
  * level l holds K/2^l monic node polynomials of degree 2^l, stored coefficient-major [2^l][K/2^l] (top coefficient implicit)
    so that ONE batched column transform (sc_ntt_batch kind 0) handles a whole level, and their size-2^(l+1) transforms
    Zf[l] are kept: they serve the products going up, the correlations going down and the combinations of interpolation;
  * evaluation is the scaled remainder tree: c = first K coefficients of f/Z in 1/x (one power-series inverse of rev(Z), Newton),
    then child_L = first half of corr(c, Z_R), child_R = first half of corr(c, Z_L) level by level; leaves are the values;
  * interpolation: w_i = v_i / Z_real'(d_i) (Z_real' evaluated with the same tree), then P = P_L*Z_R + P_R*Z_L upwards.

Every array operation below corresponds to one kernel launch / batched transform of the device code.  Test-only.
"""
from oracle import py_oracle as po

P = po.P


def root_of(n):
    return po.primitive_nth_root(n)


def ntt_cols(a, length, batch, inverse=False):
    """transform along axis 0 of the [length][batch] array `a` (flat list), natural order in and out"""
    if length == 1:
        return list(a)
    r = root_of(length)
    out = [0] * (length * batch)
    for b in range(batch):
        col = a[b::batch]
        res = po.intt(r, col) if inverse else po.ntt(r, col)
        out[b::batch] = res
    return out


class Tree:
    def __init__(self, points):
        k = len(points)
        L = max(1, (k - 1).bit_length()) if k > 1 else 0
        K = 1 << L
        self.k, self.K, self.L, self.pad = k, K, L, K - k
        self.Zc = [[(-d) % P for d in points] + [0] * (K - k)]         # level 0: [1][K]
        self.Zf = []
        for l in range(L):
            n, B = 1 << l, K >> l                                     # children: degree n, B of them
            buf = self.Zc[l] + [1] * B + [0] * ((n - 1) * B)          # expand to [2n][B] with the monic coefficient at row n
            zf = ntt_cols(buf, 2 * n, B)
            self.Zf.append(zf)
            prod = [zf[f * B + 2 * j] * zf[f * B + 2 * j + 1] % P for f in range(2 * n) for j in range(B // 2)]
            co = ntt_cols(prod, 2 * n, B // 2, inverse=True)
            for j in range(B // 2):                                   # the monic top coefficient x^(2n) wrapped onto x^0
                co[j] = (co[j] - 1) % P
            self.Zc.append(co)
        self.invG = None

    def zerofier(self):
        full = self.Zc[self.L] + [1]
        return full[self.pad:] if self.k else []

    def _inverse_series(self):
        if self.invG is not None:
            return self.invG
        K = self.K
        top = self.Zc[self.L]
        G = [1] + [top[K - t] for t in range(1, K)]                   # rev(Z) mod y^K
        h = [1]
        m = 1
        while m < K:
            n = 4 * m
            hh = ntt_cols(h + [0] * (n - m), n, 1)
            tt = ntt_cols(G[:2 * m] + [0] * (n - 2 * m), n, 1)
            nw = [hh[f] * ((2 - tt[f] * hh[f]) % P) % P for f in range(n)]
            h = ntt_cols(nw, n, 1, inverse=True)[:2 * m]
            m *= 2
        self.invG = h[:K]
        return self.invG

    def evaluate(self, coeffs):
        K, L = self.K, self.L
        assert len(coeffs) <= K
        if L == 0:
            return [coeffs[0] % P if coeffs else 0][:self.k]
        f = list(coeffs) + [0] * (K - len(coeffs))
        F = f[::-1]
        a = ntt_cols(F + [0] * K, 2 * K, 1)
        b = ntt_cols(self._inverse_series() + [0] * K, 2 * K, 1)
        c = ntt_cols([x * y % P for x, y in zip(a, b)], 2 * K, 1, inverse=True)[:K]       # [K][1]
        for l in range(L, 0, -1):
            n, B = 1 << l, K >> l                                     # nodes: size n, B of them; children 2B
            C = ntt_cols(c, n, B)
            zf = self.Zf[l - 1]                                       # [n][2B]
            D = [C[f * B + (i >> 1)] * zf[((n - f) % n) * 2 * B + (i ^ 1)] % P for f in range(n) for i in range(2 * B)]
            c = ntt_cols(D, n, 2 * B, inverse=True)[:K]               # first n/2 rows of [n][2B]
        return c[:self.k]

    def interpolate(self, values):
        k, K, L, pad = self.k, self.K, self.L, self.pad
        assert len(values) == k
        if k == 0:
            return []
        if L == 0:
            return [values[0] % P]
        zr = self.zerofier()                                          # k + 1 coefficients
        der = [(t + 1) * zr[t + 1] % P for t in range(k)]
        e = self.evaluate(der)
        assert all(x != 0 for x in e), "divide by zero"
        p = [v * po.inv(x) % P for v, x in zip(values, e)] + [0] * pad      # leaves [1][K]
        for l in range(L):
            n, B = 1 << l, K >> l
            ph = ntt_cols(p + [0] * (n * B), 2 * n, B)
            zf = self.Zf[l]
            E = [(ph[f * B + 2 * j] * zf[f * B + 2 * j + 1] + ph[f * B + 2 * j + 1] * zf[f * B + 2 * j]) % P
                 for f in range(2 * n) for j in range(B // 2)]
            p = ntt_cols(E, 2 * n, B // 2, inverse=True)
        return p[pad:]
