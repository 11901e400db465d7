"""What a proof stream holds, without a Python object per digest.

Reference: code/ip.py:4-30 -- `ProofStream.objects` is a plain list; `serialize()` is `pickle.dumps(objects)` and every Fiat-Shamir
challenge hashes those bytes.  A proof of FastStark.prove at a 2^24 FRI domain is ~50 000 digest objects in ~2 800 authentication
paths plus ~3 000 field elements: creating and pickling them costs 4-5 ms of CPython per proof, a seventh of the prover, although
nothing on the prover's side ever looks at them.  The prover therefore pushes what the device answered -- index lists, packed
residues, packed paths -- as SEGMENTS of a `LazyProofObjects`, which

  * pickles itself from a description of the object graph (csrc/proof_pickle.h, `sc_pickle_proof`): byte-identical to
    `pickle.dumps` of the materialised list, including which FieldElement OBJECTS are shared (pickle memoises by identity; the
    reference pushes the same object when a codeword entry appears twice: last codeword <-> query triple, the `c` of one FRI round
    <-> the `a`/`b` of the next, code/fri.py:91, :104-105);
  * turns into the reference's objects the moment anything indexes, iterates or compares it (a verifier running on the prover's
    stream, a test looking at `objects[0]`), through the same identity-preserving caches the object path uses.

Anything it cannot describe (an object of another type pushed by the caller) makes it fall back to materialise + `pickle.dumps`.
"""
import pickle
import struct as _struct

import numpy as np

import starkcore as _sc

_E = np.dtype([("op", "u1"), ("f", "<u4"), ("key", "<u8"), ("v", "u1", (16,))])          # 'E' field key value: 29 bytes, packed
assert _E.itemsize == 29


class _Unsupported(Exception):
    pass


def _path_dtype(depth):
    return np.dtype([("op", "u1"), ("depth", "<u4"), ("raw", "u1", (64 * depth,))])


class _Context:
    """per serialization: the table of Field objects (by identity) the elements refer to"""

    def __init__(self):
        self.fields = []
        self.seen = set()                                  # ids of the real objects described so far
        self.uids = {}                                     # process-wide codeword id -> small id of this serialization

    def uid(self, holder):
        """The codeword's id inside THIS serialization (1, 2, ...).  The pickler's memo key of an element is
        (field << 56) ^ (id << 32 | index): 24 bits for the id.  The process-wide counter behind `_codeword_uid` grows by one per
        codeword and round of every proof a long-lived prover makes and would spill into the field bits after 2^24 of them;
        a proof describes a few dozen codewords."""
        local = self.uids.setdefault(_codeword_uid(holder), len(self.uids) + 1)
        assert local < (1 << 24)
        return local

    def field_index(self, field):
        for i, f in enumerate(self.fields):
            if f is field:
                return i
        self.fields.append(field)
        return len(self.fields) - 1


def _codeword_uid(cw):
    uid = getattr(cw, "_uid", None)
    if uid is None:
        _codeword_uid.counter += 1
        uid = cw._uid = _codeword_uid.counter
    return uid


_codeword_uid.counter = 0


class _Entries:
    """What a segment keeps of the codeword it describes: its field, its identity, and a WEAK reference -- a proof stream that
    somebody holds on to must not keep gigabytes of device memory (the codeword, its Merkle tree) alive, which the object path
    never did.  Objects are made through the codeword's own cache while it lives (so `cw[i]` and the stream's entry are one
    object), through a cache of this holder's afterwards.  One holder per codeword (`cw._proof_entries`)."""
    _full = None

    def __init__(self, codeword):
        import weakref
        self.field, self._uid = codeword.field, _codeword_uid(codeword)
        self._alive = weakref.ref(codeword)
        # the codeword's OWN cache of entry objects (index -> FieldElement), shared: whatever the codeword hands out -- to the
        # object path of a mixed stream, to a caller indexing it -- is the object this holder hands out, also after the codeword
        # is gone (DeviceCodeword.tolist, which retires that cache, copies its objects here first)
        self._own = codeword._elems if codeword._elems is not None else dict(enumerate(codeword._full))

    def _entries(self, indices, values):
        cw = self._alive()
        if cw is not None:
            return cw._entries(indices, values)
        from algebra import FieldElement
        own, field, new = self._own, self.field, object.__new__
        for i, v in zip(indices, values):
            if i not in own:
                e = new(FieldElement)
                e.value = v
                e.field = field
                own[i] = e
        return [own[i] for i in indices]


class DetachedEntries(_Entries):
    """the holder of a codeword that never had a Python object: the folded codewords of a Fri.prove that ran as one library call
    (sc_fri_prove_dev) are freed before the call returns; what the stream describes of them are residues the device handed over"""

    def __init__(self, field):
        _codeword_uid.counter += 1
        self.field, self._uid, self._own = field, _codeword_uid.counter, {}

    @staticmethod
    def _alive():
        return None


def entries_of(codeword):
    """the holder a segment keeps instead of the codeword itself (holders that are not device codewords -- the sharded prover's
    layer caches -- are kept as they are: they hold no device memory)"""
    if not isinstance(codeword, _sc.DeviceCodeword):
        return codeword
    holder = getattr(codeword, "_proof_entries", None)
    if holder is None:
        holder = codeword._proof_entries = _Entries(codeword)
    return holder


def _element_ops(ctx, cw, indices, values):
    """the 'E' records of entries `indices` of device codeword `cw` (values: packed residues, 16 bytes each)"""
    k = len(indices)
    rec = np.empty(k, dtype=_E)
    rec["op"] = ord("E")
    rec["f"] = ctx.field_index(cw.field)
    rec["key"] = (np.uint64(ctx.uid(cw)) << np.uint64(32)) | np.asarray(indices, dtype=np.uint64)
    rec["v"] = np.frombuffer(values, dtype=np.uint8).reshape(k, 16)
    return rec


def _path_ops(paths, depth):
    """the 'D' records of `paths` (uint8 array [k][64 * depth])"""
    rec = np.empty(paths.shape[0], dtype=_path_dtype(depth))
    rec["op"] = ord("D")
    rec["depth"] = depth
    if depth:
        rec["raw"] = paths
    return rec


def eligible(codeword):
    """a codeword whose entries have no Python objects yet other than the ones its own cache hands out"""
    return isinstance(codeword, _sc.DeviceCodeword) and codeword._full is None


class ElementList:
    """ONE object: the list of all entries of a device codeword (the last FRI codeword, pushed in the clear: fri.py:91)"""
    count = 1

    def __init__(self, codeword, values):
        self.cw, self.values = entries_of(codeword), values

    def ops(self, ctx):
        n = len(self.values) // 16
        return b"L" + int(n).to_bytes(4, "little") + _element_ops(ctx, self.cw, range(n), self.values).tobytes()

    def materialize(self):
        n = len(self.values) // 16
        return [self.cw._entries(range(n), _sc.unpack(self.values, n))]


class FriRound:
    """one round of the query phase (fri.py:98-113): s triples (current[a], current[b], next[c]), then per test the
    authentication paths of a, b, c"""

    def __init__(self, current, following, idx_a, idx_b, idx_c, val_a, val_b, val_c, paths_a, paths_b, paths_c):
        self.cur, self.nxt = entries_of(current), entries_of(following)
        self.idx = (idx_a, idx_b, idx_c)
        self.val = (val_a, val_b, val_c)
        self.paths = (paths_a, paths_b, paths_c)
        self.s = len(idx_a)
        self.count = 4 * self.s

    def ops(self, ctx):
        """the 'R' op of csrc/proof_pickle.h: header, the three index lists, the packed residues and paths as they came back"""
        s = self.s
        if s == 0:
            return b""
        d_cur, d_nxt = self.paths[0].shape[1] // 64, self.paths[2].shape[1] // 64
        f = ctx.field_index(self.cur.field)
        if ctx.field_index(self.nxt.field) != f:
            raise _Unsupported("two fields in one round")
        head = _struct.pack("<cIIQQII", b"R", s, f, ctx.uid(self.cur) << 32, ctx.uid(self.nxt) << 32, d_cur, d_nxt)
        idx = np.asarray(self.idx, dtype=np.uint32).tobytes()
        return b"".join((head, idx, bytes(self.val[0]), bytes(self.val[1]), bytes(self.val[2]),
                         self.paths[0].tobytes(), self.paths[1].tobytes(), self.paths[2].tobytes()))

    def materialize(self):
        s = self.s
        ent = [holder._entries(np.asarray(idx).tolist(), _sc.unpack(bytes(val), s)) for holder, idx, val in zip((self.cur, self.cur, self.nxt), self.idx, self.val)]
        lists = [_sc._path_lists(memoryview(np.ascontiguousarray(p)).cast("B"), 0, p.shape[1] // 64, s) for p in self.paths]
        return list(zip(*ent)) + [path for trio in zip(*lists) for path in trio]


class FriQueryPhase:
    """every round of the query phase at once (fri.py:124-128), as sc_fri_prove_dev answered it: views of one pinned buffer, cut
    into rounds only when somebody serializes or reads the stream.
    holders[j]: codeword j's entry holder; counts / depths: openings and path depth per codeword, in the buffer's order ([a, b] of
    its own round; the last codeword: [c] of the round before -- everywhere else that c is the codeword's own a or b, opened once);
    elems / paths: uint8 views of the opened residues (16 bytes each) and of the authentication paths of those codewords, codeword
    after codeword; positions: the opened indices (uint64 view), likewise."""

    def __init__(self, holders, s, counts, depths, elems, paths, positions):
        self.holders, self.s, self.counts, self.depths = holders, s, counts, depths
        self.elems, self.paths, self.positions = elems, paths, positions
        self.count = 4 * s * (len(holders) - 1)
        self._rounds = None

    def rounds(self):
        if self._rounds is None:
            s = self.s
            values, paths, where, vo, po = [], [], [], 0, 0
            for c, d in zip(self.counts, self.depths):
                values.append(self.elems[16 * vo:16 * (vo + c)])
                paths.append(self.paths[po:po + 64 * c * d].reshape(c, 64 * d))
                where.append(self.positions[vo:vo + c])
                vo += c
                po += 64 * c * d
            k = len(self.holders)
            self._rounds = []
            for i in range(k - 1):
                if i + 2 < k:
                    # c = this round's a, in the next codeword: that codeword's own a (slot t) or b (slot s + t), whichever half it lies in
                    slot = np.arange(s) + np.where(where[i][:s] < np.uint64(1 << (self.depths[i + 1] - 1)), 0, s)
                else:
                    slot = np.arange(s)
                c_values = np.ascontiguousarray(values[i + 1].reshape(-1, 16)[slot]).reshape(-1)
                self._rounds.append(FriRound(self.holders[i], self.holders[i + 1], where[i][:s], where[i][s:2 * s], where[i + 1][slot],
                                             values[i][:16 * s], values[i][16 * s:32 * s], c_values,
                                             paths[i][:s], paths[i][s:2 * s], paths[i + 1][slot]))
        return self._rounds

    payload_bytes = property(lambda self: int(self.elems.nbytes + self.paths.nbytes))

    def ops(self, ctx):
        """the 'Q' op of csrc/proof_pickle.h: the three arrays stay where the device wrote them, the description carries their
        addresses (this segment keeps them alive)"""
        k = len(self.holders)
        if k < 2 or self.s == 0:
            return b""
        f = ctx.field_index(self.holders[0].field)
        if any(ctx.field_index(h.field) != f for h in self.holders):
            raise _Unsupported("two fields in one query phase")
        if not (self.elems.flags["C_CONTIGUOUS"] and self.paths.flags["C_CONTIGUOUS"] and self.positions.flags["C_CONTIGUOUS"]) or self.positions.dtype != np.uint64:
            return b"".join(r.ops(ctx) for r in self.rounds())
        head = _struct.pack("<cIII", b"Q", self.s, f, k)
        per = b"".join(_struct.pack("<QI", ctx.uid(h) << 32, d) for h, d in zip(self.holders, self.depths))
        return head + per + _struct.pack("<QQQ", self.elems.ctypes.data, self.paths.ctypes.data, self.positions.ctypes.data)

    def materialize(self):
        return [obj for r in self.rounds() for obj in r.materialize()]


class Openings:
    """leaf, path, leaf, path, ... of one committed codeword (fast_stark.py:154-175)"""

    def __init__(self, codeword, indices, values, paths, positions=None):
        """positions (optional): the indices once more as a contiguous uint64 array -- with values and paths as contiguous arrays
        too (the pinned answers of sc_fri_prove_dev) the description carries addresses instead of copies"""
        self.cw, self.indices, self.values, self.paths, self.positions = entries_of(codeword), indices, values, paths, positions
        self.count = 2 * len(indices)

    @property
    def payload_bytes(self):
        return 16 * len(self.indices) + int(getattr(self.paths, "nbytes", 0))

    def ops(self, ctx):
        """the 'O' op of csrc/proof_pickle.h ('P' -- payload by address -- when it lies in contiguous arrays)"""
        k = len(self.indices)
        if k == 0:
            return b""
        depth = self.paths.shape[1] // 64
        pos, val = self.positions, self.values
        if (isinstance(pos, np.ndarray) and pos.dtype == np.uint64 and pos.flags["C_CONTIGUOUS"] and len(pos) == k and isinstance(val, np.ndarray)
                and val.dtype == np.uint8 and val.flags["C_CONTIGUOUS"] and val.nbytes == 16 * k and self.paths.flags["C_CONTIGUOUS"]):
            return _struct.pack("<cIIQIQQQ", b"P", k, ctx.field_index(self.cw.field), ctx.uid(self.cw) << 32, depth,
                                pos.ctypes.data, val.ctypes.data, self.paths.ctypes.data if depth else 0)
        head = _struct.pack("<cIIQI", b"O", k, ctx.field_index(self.cw.field), ctx.uid(self.cw) << 32, depth)
        return b"".join((head, np.asarray(self.indices, dtype=np.uint32).tobytes(), bytes(self.values), self.paths.tobytes()))

    def materialize(self):
        k = len(self.indices)
        entries = self.cw._entries(list(self.indices), _sc.unpack(bytes(self.values), k))
        paths = _sc._path_lists(memoryview(np.ascontiguousarray(self.paths)).cast("B"), 0, self.paths.shape[1] // 64, k)
        return [x for pair in zip(entries, paths) for x in pair]


class _Real:
    """objects the caller pushed as objects"""

    def __init__(self, objects):
        self.objects = objects

    @property
    def count(self):
        return len(self.objects)

    def ops(self, ctx):
        out, seen = [], ctx.seen
        for o in self.objects:
            if type(o) is bytes:
                out.append(b"B" + len(o).to_bytes(4, "little") + o)
                fresh = [o]
            elif type(o) is list and all(type(x) is bytes and len(x) == 64 for x in o):
                out.append(b"D" + len(o).to_bytes(4, "little") + b"".join(o))
                fresh = o + [o]
            else:
                # a FieldElement object (or anything holding one) may be THE SAME object as one a lazy segment describes by
                # (codeword, index): only the pickler walking real objects gets that right
                raise _Unsupported(type(o).__name__)
            for x in fresh:                                # the same object twice is a memo hit in pickle: not described here
                if id(x) in seen:
                    raise _Unsupported("the same object pushed twice")
                seen.add(id(x))
        return b"".join(out)

    def materialize(self):
        return self.objects


class RootLater:
    """ONE object: the root of a Merkle tree whose build is only enqueued (starkcore.MerkleTree.from_device_async).  The prover pushes
    the commitment and goes on enqueueing -- the next LDE stands in the queue right behind the tree -- and the root is waited for
    when somebody needs the stream's bytes or objects: the next Fiat-Shamir challenge (fast_stark.py:125), serialization.  With
    `Merkle.commit` returning the root at once the GPU stood idle after every commitment for as long as the host took to come back
    with the next launch (tools/sync_points.py, tools/gap_report.py)."""
    count = 1

    def __init__(self, tree):
        self.tree = tree

    def ops(self, ctx):
        root = self.tree.root                              # (waits for the build; the same bytes object from then on)
        if id(root) in ctx.seen:
            raise _Unsupported("the same object pushed twice")
        ctx.seen.add(id(root))
        return b"B" + len(root).to_bytes(4, "little") + root

    def materialize(self):
        return [self.tree.root]


class LazyProofObjects:
    """`ProofStream.objects` once a prover has pushed device answers: a sequence that reads like the reference's list"""

    def __init__(self, objects=()):
        self._segments = [_Real(list(objects))]
        self._cache = {}                                   # segment index -> its materialised objects
        self._all = None

    # -- the prover's side
    def append(self, obj):
        self._all = None
        if not isinstance(self._segments[-1], _Real):
            self._segments.append(_Real([]))
        self._segments[-1].objects.append(obj)

    def extend(self, objs):
        for o in objs:
            self.append(o)

    def add(self, segment):
        self._all = None
        self._segments.append(segment)

    def pickled(self):
        """pickle.dumps(list(self)), without making the list"""
        try:
            ctx = _Context()
            body = [seg.ops(ctx) for seg in self._segments]
            ops = b"L" + len(self).to_bytes(4, "little") + b"".join(body)
            moduli = b"".join(f.p.to_bytes(32, "little") for f in ctx.fields)
        except (_Unsupported, OverflowError):
            return pickle.dumps(self.materialized())
        # (segments whose payload stays where the device wrote it describe it by address: the output is that much longer than the ops)
        by_address = sum(getattr(seg, "payload_bytes", 0) for seg in self._segments)
        return _sc.pickle_proof(ops, moduli, len(ctx.fields), 32, expect=by_address)

    # -- the reader's side: the reference's objects, made once
    def _segment_objects(self, k):
        got = self._cache.get(k)
        if got is None:
            got = self._cache[k] = self._segments[k].materialize()
        return got

    def materialized(self):
        if self._all is None:
            out = []
            for k, seg in enumerate(self._segments):
                out.extend(seg.objects if isinstance(seg, _Real) else self._segment_objects(k))
            self._all = out
        return self._all

    def detach(self):
        """Make the reference's objects once and keep ONLY them.  The described segments are views into the pinned host buffer the
        query kernel wrote (starkcore.HostBuffer: megabytes per proof, page-locked, back in the library's pool when the last view
        dies); serializing and dropping the stream -- what a prover does -- releases it by itself.  A process that RETAINS proof
        streams calls this (or reads the objects and drops the stream) so that it holds ordinary Python objects, as the
        reference's streams do, instead of page-locked memory."""
        objects = list(self.materialized())
        self._segments, self._cache, self._all = [_Real(objects)], {}, None
        return self

    def __len__(self):
        return sum(seg.count for seg in self._segments)

    def __getitem__(self, i):
        return self.materialized()[i]

    def __iter__(self):
        return iter(self.materialized())

    def __eq__(self, other):
        return self.materialized() == (other.materialized() if isinstance(other, LazyProofObjects) else other)

    __hash__ = None

    def __reduce__(self):
        raise TypeError("a proof stream's objects pickle through ProofStream.serialize()")


def lazy_objects(proof_stream):
    """`proof_stream.objects` as a LazyProofObjects (swapped in on first use), or None when the stream is not a plain
    ip.ProofStream holding a plain list (a subclass may serialize differently; the reference's SignatureProofStream does)"""
    from ip import ProofStream
    if type(proof_stream) is not ProofStream or pickle.DEFAULT_PROTOCOL != 4:
        return None                                        # (the library writes protocol 4, the default of CPython 3.8 - 3.13)
    objects = proof_stream.objects
    if isinstance(objects, LazyProofObjects):
        return objects
    if type(objects) is not list:
        return None
    proof_stream.objects = LazyProofObjects(objects)
    return proof_stream.objects
