"""NTT-based polynomial arithmetic -- host shim over the HIP kernels (libstarkcore.so).

Same callables, argument meaning, assertion messages and list-length conventions as the reference's
code/ntt.py (ntt :3, intt :20, fast_multiply :32, fast_zerofier :66, fast_evaluate :82,
fast_interpolate :102, fast_coset_evaluate :132, fast_coset_divide :137); the transforms, the coset
scaling, the Hadamard product / pointwise division and the truncations run on the MI355X.

`list[FieldElement]` in -> new `list[FieldElement]` out (arguments are never mutated).  The *_device
variants take/return `DeviceCodeword` (data stays in HBM; no per-element marshalling) and are what
fri.py / bench.py use for large domains.
"""
import ctypes

from univariate import *
import starkcore as _sc
from starkcore import DeviceCodeword, DeviceVector

_ROOT_ORDER_MSG = "supplied root does not have supplied order"
_ROOT_PRIM_MSG = "supplied root is not primitive root of supplied order"


_FIELD_MSG = "the MI355X polynomial core implements the field p = 1 + 407 * 2^119 only"


def _require_main_field(field):
    """The kernels are hard-wired to p = 1 + 407*2^119 (the only field whose primitive_nth_root the reference supports,
    algebra.py:104-114).  There is no CPU fallback, so any other modulus is refused loudly instead of being mis-reduced."""
    assert(field.p == Field.P_MAIN), _FIELD_MSG


def _pack(elements):
    return _sc.pack(list(map(_sc._value_of, elements)))


def _unpack(raw, count, field):
    return [FieldElement(v, field) for v in _sc.unpack(raw, count)]


_verified_roots = set()       # (p, root, order) triples that already passed the two assertions below


def _check_root(primitive_root, root_order):
    key = (primitive_root.field.p, primitive_root.value, root_order)
    if key in _verified_roots:
        return
    assert(primitive_root ^ root_order == primitive_root.field.one()), _ROOT_ORDER_MSG
    assert(primitive_root ^ (root_order // 2) != primitive_root.field.one()), _ROOT_PRIM_MSG
    if len(_verified_roots) < 4096:
        _verified_roots.add(key)


def _transform(primitive_root, values, inverse):
    n = len(values)
    field = values[0].field
    _require_main_field(field)
    if isinstance(values, DeviceCodeword):
        out = DeviceVector(n)
        _sc._check(_sc.lib().sc_ntt_dev(values.vec.ptr, out.ptr, n, _sc.fe_bytes(primitive_root.value), inverse, None))
        return DeviceCodeword(out, field)
    out = ctypes.create_string_buffer(16 * n)
    _sc._check(_sc.lib().sc_ntt(_pack(values), out, n, _sc.fe_bytes(primitive_root.value), inverse))
    return _unpack(out.raw, n, field)


def ntt(primitive_root, values):
    assert(len(values) & (len(values) - 1) == 0), "cannot compute ntt of non-power-of-two sequence"
    if len(values) <= 1:
        return values
    field = values[0].field
    assert(primitive_root ^ len(values) == field.one()), "primitive root must be nth root of unity, where n is len(values)"
    assert(primitive_root ^ (len(values) // 2) != field.one()), "primitive root is not primitive nth root of unity, where n is len(values)"
    return _transform(primitive_root, values, 0)


def intt(primitive_root, values):
    assert(len(values) & (len(values) - 1) == 0), "cannot compute intt of non-power-of-two sequence"
    if len(values) == 1:
        return values
    field = values[0].field
    # the reference runs ntt(root^-1, .) and so inherits its root checks (ntt.py:27-29, :10-11)
    assert(primitive_root ^ len(values) == field.one()), "primitive root must be nth root of unity, where n is len(values)"
    assert(primitive_root ^ (len(values) // 2) != field.one()), "primitive root is not primitive nth root of unity, where n is len(values)"
    return _transform(primitive_root, values, 1)


def _shrink_order(root, order, degree):
    # smallest power-of-two transform that still holds `degree+1` coefficients (ntt.py:47-49, :155-157)
    while degree < order // 2:
        root = root ^ 2
        order = order // 2
    return root, order


def fast_multiply(lhs, rhs, primitive_root, root_order):
    _check_root(primitive_root, root_order)
    if lhs.is_zero() or rhs.is_zero():
        return Polynomial([])
    field = lhs.coefficients[0].field
    dl, dr = lhs.degree(), rhs.degree()
    degree = dl + dr
    if degree < 8:
        return lhs * rhs
    _require_main_field(field)
    root, order = _shrink_order(primitive_root, root_order, degree)
    out = ctypes.create_string_buffer(16 * (degree + 1))
    _sc._check(_sc.lib().sc_poly_mul(_pack(lhs.coefficients[:dl + 1]), dl + 1, _pack(rhs.coefficients[:dr + 1]), dr + 1,
                                     _sc.fe_bytes(root.value), order, out, degree + 1))
    return Polynomial(_unpack(out.raw, degree + 1, field))


# Domains of at least this many points go to the device -- the level-batched subproduct tree (csrc/polytree.cuh), or, when the
# points are a geometric progression, a handful of convolutions (csrc/geoseq.cuh); smaller ones follow the reference's
# recursion below.  Zerofier, values and interpolant are unique, so all three give the same lists.
DEVICE_TREE_MIN_POINTS = 16

_tree_memo = {}           # points (as packed bytes) -> PolyTree; fast_interpolate / fast_evaluate revisit the same domains


def _device_tree(domain):
    _require_main_field(domain[0].field)
    key = _pack(domain)
    tree = _tree_memo.get(key)
    if tree is None:
        # a tree keeps ~1 KiB of HBM per point (all levels and their transforms): bound what the memo retains
        if len(_tree_memo) >= 8 or sum(t.k for t in _tree_memo.values()) + len(domain) > (1 << 22):
            for t in _tree_memo.values():
                t.free()
            _tree_memo.clear()
        # a geometric progression (the trace domain {omicron^i}, fast_stark.py:84-90) gets the progression tables, anything
        # else the subproduct tree: same zerofier, values and interpolant either way
        tree = _tree_memo[key] = _sc.domain_tables(key)
    return tree


# The reference recomputes the zerofier of every sub-domain at every node of fast_evaluate / fast_interpolate
# (ntt.py:92-93, :115-116); the polynomials are the same each time, so they are remembered per call tree.
_zerofier_memo = {}


def fast_zerofier(domain, primitive_root, root_order):
    _check_root(primitive_root, root_order)
    if len(domain) == 0:
        return Polynomial([])
    if len(domain) == 1:
        return Polynomial([-domain[0], primitive_root.field.one()])
    if len(domain) >= DEVICE_TREE_MIN_POINTS:
        return Polynomial(_unpack(_device_tree(domain).zerofier().to_bytes(), len(domain) + 1, primitive_root.field))
    key = (primitive_root.value, root_order, tuple(d.value for d in domain))
    hit = _zerofier_memo.get(key)
    if hit is not None:
        return hit
    half = len(domain) // 2
    result = fast_multiply(fast_zerofier(domain[:half], primitive_root, root_order),
                           fast_zerofier(domain[half:], primitive_root, root_order), primitive_root, root_order)
    if len(_zerofier_memo) >= 8192:
        _zerofier_memo.clear()
    _zerofier_memo[key] = result
    return result


def fast_evaluate(polynomial, domain, primitive_root, root_order):
    _check_root(primitive_root, root_order)
    if len(domain) == 0:
        return []
    if len(domain) == 1:
        return [polynomial.evaluate(domain[0])]
    if len(domain) >= DEVICE_TREE_MIN_POINTS:
        coeffs = DeviceVector.from_bytes(_pack(polynomial.coefficients))
        return _unpack(_device_tree(domain).evaluate(coeffs).to_bytes(), len(domain), primitive_root.field)
    half = len(domain) // 2
    lower, upper = domain[:half], domain[half:]
    lower_rem = polynomial % fast_zerofier(lower, primitive_root, root_order)
    upper_rem = polynomial % fast_zerofier(upper, primitive_root, root_order)
    return fast_evaluate(lower_rem, lower, primitive_root, root_order) + fast_evaluate(upper_rem, upper, primitive_root, root_order)


def fast_interpolate(domain, values, primitive_root, root_order):
    _check_root(primitive_root, root_order)
    assert(len(domain) == len(values)), "cannot interpolate over domain of different length than values list"
    if len(domain) == 0:
        return Polynomial([])
    if len(domain) == 1:
        return Polynomial([values[0]])
    if len(domain) >= DEVICE_TREE_MIN_POINTS:
        # a repeated point raises AssertionError("divide by zero") like the field division at ntt.py:124-125
        out = _device_tree(domain).interpolate(DeviceVector.from_bytes(_pack(values)))
        return Polynomial(_unpack(out.to_bytes(), len(domain), primitive_root.field))
    half = len(domain) // 2
    lower, upper = domain[:half], domain[half:]
    lower_zerofier = fast_zerofier(lower, primitive_root, root_order)
    upper_zerofier = fast_zerofier(upper, primitive_root, root_order)
    # each half is interpolated against values divided by the OTHER half's zerofier there (ntt.py:121-125)
    lower_offset = fast_evaluate(upper_zerofier, lower, primitive_root, root_order)
    upper_offset = fast_evaluate(lower_zerofier, upper, primitive_root, root_order)
    if not all(not v.is_zero() for v in lower_offset):
        print("left_offset:", " ".join(str(v) for v in lower_offset))
    lower_targets = [n / d for (n, d) in zip(values[:half], lower_offset)]
    upper_targets = [n / d for (n, d) in zip(values[half:], upper_offset)]
    lower_interpolant = fast_interpolate(lower, lower_targets, primitive_root, root_order)
    upper_interpolant = fast_interpolate(upper, upper_targets, primitive_root, root_order)
    return lower_interpolant * upper_zerofier + upper_interpolant * lower_zerofier


class DeviceDomain:
    """A list of evaluation points resident in HBM together with its subproduct tree (built once): the device-resident form
    of the `domain` argument of fast_zerofier / fast_evaluate / fast_interpolate for domains too large to marshal per call."""

    def __init__(self, points, field=None):
        """points: DeviceCodeword, DeviceVector (give `field`) or list[FieldElement]"""
        if isinstance(points, DeviceCodeword):
            vec, field = points.vec, points.field
        elif isinstance(points, DeviceVector):
            vec = points
        else:
            vec, field = DeviceVector.from_bytes(_pack(points)), points[0].field
        assert(vec.n > 0), "empty domain"
        _require_main_field(field)
        self.field = field
        self.tree = _sc.domain_tables(vec)

    @classmethod
    def geometric(cls, first, ratio, count):
        """the domain first * ratio^i, i < count (FieldElements), without materialising or inspecting the points: the trace
        domain of fast_stark.py:84-90 is {omicron^i}.  Falls back to the general tree where the progression tables do not apply."""
        field = ratio.field
        _require_main_field(field)
        tables = _sc.GeoDomain.create(first.value, ratio.value, count) if count >= 2 else None
        if tables is None:
            ones = DeviceVector.from_bytes((first.value).to_bytes(16, "little") * count)
            points = DeviceVector(count)
            _sc._check(_sc.lib().sc_scale_dev(ones.ptr, points.ptr, count, _sc.fe_bytes(ratio.value), None))
            return cls(points, field)
        domain = cls.__new__(cls)
        domain.field, domain.tree = field, tables
        return domain

    def __len__(self):
        return self.tree.k


class DevicePolynomial:
    """A coefficient list resident in HBM: what `Polynomial` is to the host path.  `len(p)` is the reference's LIST length
    (trailing zeros count, like `polynomial.coefficients`); `degree()` is Polynomial.degree (code/univariate.py:7-17), computed
    on the device once and cached.  Used by the callers that keep their polynomials on the device (fast_stark.py)."""

    def __init__(self, vec, field, length=None):
        self.vec, self.field = vec, field
        self.n = vec.n if length is None else int(length)
        self._degree = None

    @classmethod
    def from_polynomial(cls, polynomial, field=None):
        coeffs = polynomial.coefficients
        field = field if field is not None else coeffs[0].field
        _require_main_field(field)
        out = cls(DeviceVector.from_bytes(_pack(coeffs)) if coeffs else DeviceVector(1), field, len(coeffs))
        out._degree = polynomial.degree()                  # known on the host: no round trip to the device for it later
        return out

    @classmethod
    def from_codeword(cls, codeword):
        return cls(codeword.vec, codeword.field)

    def __len__(self):
        return self.n

    def to_polynomial(self):
        return Polynomial(_unpack(self.vec.to_bytes(0, self.n), self.n, self.field))

    def degree(self):
        if self._degree is None:
            deg = ctypes.c_int64(-1)
            _sc._check(_sc.lib().sc_vec_degree_dev(self.vec.ptr, self.n, ctypes.byref(deg), None))
            self._degree = int(deg.value)
        return self._degree

    def is_zero(self):
        return self.degree() == -1

    def copy(self, length=None):
        length = self.n if length is None else length
        out = DeviceVector.zeros(max(length, 1))
        if min(length, self.n):
            out.axpy_shift(_View(self.vec, min(length, self.n)), 0, 1)
        return DevicePolynomial(out, self.field, length)

    def minus(self, polynomial):
        """self - polynomial for a (short) host Polynomial: list length max(len, len), like Polynomial.__sub__"""
        coeffs = polynomial.coefficients
        out = self.copy(max(self.n, len(coeffs)))
        if coeffs:
            out.vec.axpy_shift(DeviceVector.from_bytes(_pack(coeffs)), 0, self.field.p - 1)
        # a subtrahend of lower degree leaves the degree alone: asked of `self` (once, and remembered there -- the prover needs the
        # trace polynomials' degrees again for the transition quotients) instead of of the difference
        if polynomial.degree() < self.degree():
            out._degree = self._degree
        return out

    def scale(self, factor):
        """Polynomial.scale (code/univariate.py:153-154): coefficient i times factor^i"""
        out = DeviceVector(max(self.n, 1))
        if self.n:
            _sc._check(_sc.lib().sc_scale_dev(self.vec.ptr, out.ptr, self.n, _sc.fe_bytes(factor.value), None))
        scaled = DevicePolynomial(out, self.field, self.n)
        if factor.value % self.field.p != 0:
            scaled._degree = self._degree                  # coefficient i times factor^i: the same coefficients vanish (None stays None)
        return scaled

    def scaled_later(self, factor):
        """`scale(factor)` whose coefficients are only computed if somebody asks for them: a caller that evaluates the result on the
        coset g <factor> never does -- q(factor X) there is q's own codeword, one place on (fast_stark.py:105-113)"""
        return ScaledLater(self, factor)

    def coset_evaluate(self, offset, generator, order):
        """fast_coset_evaluate (code/ntt.py:132-135) -> DeviceCodeword"""
        out = DeviceVector(order)
        # (enqueued, not waited for: whatever reads the codeword next runs on the same stream, and vectors freed meanwhile are parked
        # behind an event -- sc_vec_free never hands memory back while a stream may still use it)
        _sc._check(_sc.lib().sc_coset_evaluate_dev(self.vec.ptr, self.n, _sc.fe_bytes(offset.value), _sc.fe_bytes(generator.value), order, out.ptr, None))
        return DeviceCodeword(out, self.field)


class ScaledLater(DevicePolynomial):
    """source.scale(factor), made when `vec` is first read; `scaled_from` = (source, factor) for whoever can do without"""

    def __init__(self, source, factor):
        self.field, self.n = source.field, source.n
        self.scaled_from = (source, factor)
        self._made = None
        self._degree = source._degree if factor.value % source.field.p != 0 else None

    @property
    def vec(self):
        if self._made is None:
            source, factor = self.scaled_from
            self._made = DevicePolynomial.scale(source, factor)
            if self._degree is None:
                self._degree = self._made._degree
        return self._made.vec

    @vec.setter
    def vec(self, value):
        # (DevicePolynomial.__init__ is deliberately not run: there is no vector until somebody reads it)
        raise AttributeError("ScaledLater.vec is derived from scaled_from; wrap a new vector in a DevicePolynomial instead")

    def degree(self):
        if self._degree is None and self.scaled_from[1].value % self.field.p != 0:
            self._degree = self.scaled_from[0].degree()      # the same coefficients vanish
        return DevicePolynomial.degree(self)


class _View:
    """the first n elements of a DeviceVector (for axpy_shift sources)"""

    def __init__(self, vec, n):
        self.ptr, self.n = vec.ptr, n


def coset_divide_device(lhs, rhs, offset, primitive_root, root_order, exact=False, later=None):
    """fast_coset_divide (code/ntt.py:137-176) on DevicePolynomials: lhs.degree() - rhs.degree() + 1 coefficients, in HBM.
    exact=True additionally asserts what Polynomial.__truediv__ asserts (a zero remainder), decided on the device.
    later (a list, with exact=True): the division is only enqueued and a check is appended to the list -- a callable that waits for
    the device's verdict and raises what this call would have raised ("divide by zero", a non-zero remainder); the caller runs its
    checks where it has to wait anyway.  The quotient's degree is what an exact division leaves, which the check confirms."""
    _check_root(primitive_root, root_order)
    assert(not rhs.is_zero()), "cannot divide by zero polynomial"
    field = lhs.field
    if lhs.is_zero():
        return DevicePolynomial(DeviceVector(1), field, 0)
    dl, dr = lhs.degree(), rhs.degree()
    assert(dr <= dl), "cannot divide by polynomial of larger degree"
    root, order = _shrink_order(primitive_root, root_order, max(dl, dr))
    n_out = dl - dr + 1
    out = DeviceVector(n_out)
    if exact and later is not None:
        handle = ctypes.c_void_p()
        rc = _sc.lib().sc_coset_divide_later_dev(lhs.vec.ptr, dl + 1, rhs.vec.ptr, dr + 1, _sc.fe_bytes(offset.value), _sc.fe_bytes(root.value), order,
                                                 out.ptr, n_out, ctypes.byref(handle), None)
        if rc != _sc.SC_ERR_UNSUPPORTED:
            _sc._check(rc)
            check = _sc.Later(handle)

            def verdict():
                zero_divisor, remainder = check.wait()
                assert(not zero_divisor), "divide by zero"
                assert(not remainder), "cannot perform polynomial division because remainder is not zero"
            later.append(verdict)
            quotient = DevicePolynomial(out, field, n_out)
            quotient._degree = dl - dr
            return quotient
    flag = ctypes.c_int(0)
    _sc._check(_sc.lib().sc_coset_divide_dev(lhs.vec.ptr, dl + 1, rhs.vec.ptr, dr + 1, _sc.fe_bytes(offset.value), _sc.fe_bytes(root.value), order,
                                             out.ptr, n_out, ctypes.byref(flag) if exact else None, None))
    quotient = DevicePolynomial(out, field, n_out)
    if exact:
        assert(flag.value == 1), "cannot perform polynomial division because remainder is not zero"
        quotient._degree = dl - dr                         # an exact quotient's leading coefficient is lhs's over rhs's: not zero
    return quotient


def fast_zerofier_device(domain):
    """fast_zerofier (ntt.py:66-80) of a DeviceDomain -> DeviceCodeword of len(domain) + 1 coefficients"""
    return DeviceCodeword(domain.tree.zerofier(), domain.field)


def fast_evaluate_device(coefficients, domain):
    """fast_evaluate (ntt.py:82-100): coefficients (DeviceCodeword) at a DeviceDomain -> DeviceCodeword of values"""
    return DeviceCodeword(domain.tree.evaluate(coefficients.vec), domain.field)


def fast_interpolate_device(domain, values):
    """fast_interpolate (ntt.py:102-130): values (DeviceCodeword) on a DeviceDomain -> DeviceCodeword of len(domain) coefficients"""
    assert(len(domain) == len(values)), "cannot interpolate over domain of different length than values list"
    return DeviceCodeword(domain.tree.interpolate(values.vec), domain.field)


def fast_coset_evaluate_device(polynomial, offset, generator, order):
    """fast_coset_evaluate with the result left in HBM as a DeviceCodeword."""
    coeffs = polynomial.coefficients
    m = len(coeffs)
    _require_main_field(offset.field)
    out = DeviceVector(order)
    src = DeviceVector.from_bytes(_pack(coeffs)) if m else DeviceVector(1)
    _sc._check(_sc.lib().sc_coset_evaluate_dev(src.ptr, m, _sc.fe_bytes(offset.value), _sc.fe_bytes(generator.value), order, out.ptr, None))
    return DeviceCodeword(out, offset.field)          # (not waited for: `src` is parked behind an event when it is freed)


def fast_coset_evaluate(polynomial, offset, generator, order):
    coeffs = polynomial.coefficients
    m = len(coeffs)
    if order <= 1 or m > order:
        # degenerate shapes: follow the reference's own formula on the host
        padded = polynomial.scale(offset).coefficients + [offset.field.zero()] * (order - m)
        return ntt(generator, padded)
    assert(order & (order - 1) == 0), "cannot compute ntt of non-power-of-two sequence"
    field = offset.field
    assert(generator ^ order == field.one()), "primitive root must be nth root of unity, where n is len(values)"
    assert(generator ^ (order // 2) != field.one()), "primitive root is not primitive nth root of unity, where n is len(values)"
    _require_main_field(field)
    out = ctypes.create_string_buffer(16 * order)
    _sc._check(_sc.lib().sc_coset_evaluate(_pack(coeffs), m, _sc.fe_bytes(offset.value), _sc.fe_bytes(generator.value), order, out))
    return _unpack(out.raw, order, field)


def fast_coset_divide(lhs, rhs, offset, primitive_root, root_order):  # clean division only!
    _check_root(primitive_root, root_order)
    assert(not rhs.is_zero()), "cannot divide by zero polynomial"
    if lhs.is_zero():
        return Polynomial([])
    dl, dr = lhs.degree(), rhs.degree()
    assert(dr <= dl), "cannot divide by polynomial of larger degree"
    field = lhs.coefficients[0].field
    degree = max(dl, dr)
    if degree < 8:
        return lhs / rhs
    _require_main_field(field)
    root, order = _shrink_order(primitive_root, root_order, degree)
    n_out = dl - dr + 1
    out = ctypes.create_string_buffer(16 * n_out)
    _sc._check(_sc.lib().sc_coset_divide(_pack(lhs.coefficients[:dl + 1]), dl + 1, _pack(rhs.coefficients[:dr + 1]), dr + 1,
                                         _sc.fe_bytes(offset.value), _sc.fe_bytes(root.value), order, out, n_out))
    return Polynomial(_unpack(out.raw, n_out, field))
