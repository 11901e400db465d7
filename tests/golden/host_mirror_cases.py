"""Seeded differential cases for the host mirror (algebra / univariate / multivariate): the SAME function is run by
tests/golden/make_golden.py on the REFERENCE's modules (-> tests/golden/host_mirror.json, digests only) and by
tests/test_host_properties.py on this repository's modules.  What a case pins: coefficient VALUES and LIST LENGTHS (trailing zeros
travel through + - * divide % / ^ scale), the ORDER of an MPolynomial's dictionary, which exception a degenerate operand raises
(its type; messages differ between Python versions), `Field.inverse(0) = 0` reaching `interpolate_domain` through repeated
abscissas, `==` on operands whose lists differ in trailing zeros, `str()`, and the pickle bytes of FieldElement lists.
Inputs come from `random.Random(seed)`; the output of a case is the SHA-256 of the repr of its results."""
import hashlib
import pickle
import random


def _dump(x):
    if x is None or isinstance(x, (bool, int, str)):
        return x
    if isinstance(x, bytes):
        return x.hex()
    if isinstance(x, (tuple, list)):
        return [_dump(t) for t in x]
    if hasattr(x, "coefficients"):
        return ["P"] + [c.value for c in x.coefficients]
    if hasattr(x, "dictionary"):
        return ["M"] + [[list(k), v.value] for k, v in x.dictionary.items()]          # in dictionary ORDER
    if hasattr(x, "value"):
        return ["F", x.value]
    return repr(x)


def _try(f):
    try:
        return _dump(f())
    except Exception as e:          # noqa: BLE001  (the type of the exception is part of the behaviour)
        return "EXC:" + type(e).__name__


def _digest(results):
    return hashlib.sha256(repr(results).encode()).hexdigest()


def run_cases(algebra, univariate, multivariate, count=400):
    """-> {"univariate": [...], "interpolate": [...], "multivariate": [...], "field": [...]} of hex digests"""
    Field, FieldElement = algebra.Field, algebra.FieldElement
    Polynomial, test_colinearity = univariate.Polynomial, univariate.test_colinearity
    MPolynomial = multivariate.MPolynomial
    field = Field.main()
    p = field.p

    def fe(v):
        return FieldElement(v % p, field)

    out = {"univariate": [], "interpolate": [], "multivariate": [], "field": []}
    rng = random.Random(20260601)

    def rpoly(maxlen=12):
        n = rng.randrange(0, maxlen)
        c = [fe(rng.choice([0, 0, 1, p - 1, rng.getrandbits(127)])) for _ in range(n)]
        if rng.random() < 0.4:
            c += [fe(0)] * rng.randrange(1, 4)
        return Polynomial(c)

    for _ in range(count):
        a, b = rpoly(), rpoly()
        e = rng.randrange(0, 6)
        pt = fe(rng.getrandbits(120))
        longer = Polynomial(a.coefficients + [fe(0)])
        out["univariate"].append(_digest([
            _try(lambda: a + b), _try(lambda: a - b), _try(lambda: a * b), _try(lambda: Polynomial.divide(a, b)), _try(lambda: a % b), _try(lambda: a / b),
            _try(lambda: a ^ e), _try(lambda: a == b), _try(lambda: a == longer), _try(lambda: longer == a), _try(lambda: a.__neq__(b)), _try(lambda: -a),
            _try(lambda: a.degree()), _try(lambda: a.is_zero()), _try(lambda: a.leading_coefficient()), _try(lambda: a.evaluate(pt)),
            _try(lambda: a.evaluate_domain([pt, fe(0), fe(1)])), _try(lambda: str(a)), _try(lambda: a.scale(pt)),
            _try(lambda: (a * b) / b), _try(lambda: (a * b) % b)]))
    for _ in range(count // 2):
        n = rng.randrange(0, 9)
        dom = [fe(rng.choice([0, 1, 2, 3, rng.getrandbits(100)])) for _ in range(n)]          # repeated abscissas are likely
        if rng.random() < 0.5:
            dom = [fe(rng.getrandbits(100) + i) for i in range(n)]
        vals = [fe(rng.choice([0, rng.getrandbits(100)])) for _ in range(n if rng.random() < 0.9 else n + 1)]
        pts = list(zip(dom, vals))
        out["interpolate"].append(_digest([
            _try(lambda: Polynomial.interpolate_domain(dom, vals)), _try(lambda: Polynomial.zerofier_domain(dom)),
            _try(lambda: test_colinearity(pts[:3])), _try(lambda: test_colinearity(pts)),
            _try(lambda: Polynomial.interpolate_domain(dom, vals).evaluate_domain(dom) if len(set(d.value for d in dom)) == len(dom) == len(vals) else None)]))

    def rmpoly(nvars):
        d = {}
        for _ in range(rng.randrange(0, 5)):
            k = tuple(rng.randrange(0, 3) for _ in range(rng.choice([nvars, nvars, nvars - 1]) if nvars > 1 else nvars))
            d[k] = fe(rng.choice([0, 1, rng.getrandbits(90)]))
        return MPolynomial(d)

    for _ in range(count // 2):
        nv = rng.randrange(1, 4)
        a, b = rmpoly(nv), rmpoly(nv)
        e = rng.randrange(0, 4)
        point = [fe(rng.getrandbits(80)) for _ in range(nv)]
        ppoint = [rpoly(4) for _ in range(nv)]
        u = rpoly(5)
        out["multivariate"].append(_digest([
            _try(lambda: a + b), _try(lambda: a - b), _try(lambda: a * b), _try(lambda: -a), _try(lambda: a ^ e), _try(lambda: a.is_zero()),
            _try(lambda: a.evaluate(point)), _try(lambda: a.evaluate_symbolic(ppoint)), _try(lambda: MPolynomial.lift(u, rng.randrange(0, 3))),
            _try(lambda: MPolynomial.constant(point[0])), _try(lambda: MPolynomial.variables(nv, field)), _try(lambda: MPolynomial.zero())]))
    for _ in range(count // 2):
        x, y = rng.choice([0, 1, p - 1, rng.getrandbits(127)]), rng.choice([0, 1, p - 1, rng.getrandbits(127)])
        a, b = fe(x), fe(y)
        e = rng.choice([0, 1, 2, rng.getrandbits(130)])
        raw = bytes(rng.getrandbits(8) for _ in range(rng.randrange(1, 40)))
        out["field"].append(_digest([
            _try(lambda: a + b), _try(lambda: a - b), _try(lambda: a * b), _try(lambda: a / b), _try(lambda: -a), _try(lambda: a.inverse()), _try(lambda: a ^ e),
            _try(lambda: a == b), _try(lambda: a != b), _try(lambda: a.is_zero()), _try(lambda: str(a)), _try(lambda: bytes(a)), _try(lambda: field.sample(raw)),
            _try(lambda: pickle.dumps([a, b, a])), _try(lambda: pickle.dumps([fe(7)]))]))
    return out
