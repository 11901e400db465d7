"""ProofStream.serialize() without Python objects (reference code/ip.py:18-25: pickle.dumps(self.objects)): the library's pickler
(csrc/proof_pickle.h, sc_pickle_proof) and proof_objects.LazyProofObjects against CPython's own pickle.dumps, byte for byte.
Host only: runs without a GPU."""
import pickle
import random
import struct

import numpy as np
import pytest

import starkcore as sc
import proof_objects as po_
from algebra import Field, FieldElement
from ip import ProofStream

MAIN = Field.main()
SMALL = Field(97)


def describe(obj, keys, fields, rng):
    """the ops of csrc/proof_pickle.h for a graph of lists / bytes / 3-tuples / FieldElements; object identity -> key"""
    if type(obj) is bytes:
        return b"B" + struct.pack("<I", len(obj)) + obj
    if type(obj) is list:
        if obj and all(type(o) is bytes and len(o) == 64 for o in obj) and rng.random() < 0.7:
            return b"D" + struct.pack("<I", len(obj)) + b"".join(obj)
        return b"L" + struct.pack("<I", len(obj)) + b"".join(describe(o, keys, fields, rng) for o in obj)
    if type(obj) is tuple:
        return b"T" + b"".join(describe(o, keys, fields, rng) for o in obj)
    assert type(obj) is FieldElement
    f = [id(x) for x in fields].index(id(obj.field))
    return b"E" + struct.pack("<IQ", f, keys.setdefault(id(obj), len(keys))) + obj.value.to_bytes(16, "little")


def library_pickle(obj, fields, rng):
    return sc.pickle_proof(describe(obj, {}, fields, rng), b"".join(f.p.to_bytes(17, "little") for f in fields), len(fields), 17)


def random_graph(rng, n):
    fields = [MAIN] if rng.random() < 0.7 else [MAIN, SMALL]
    pools = {id(f): [] for f in fields}

    def element(field):
        pool = pools[id(field)]
        if pool and rng.random() < 0.4:
            return rng.choice(pool)                      # the SAME object again: a memo hit in pickle
        v = rng.choice([0, 1, 255, 256, 65535, 65536, 2 ** 31 - 1, 2 ** 31, 2 ** 32, 2 ** 64 - 1, 2 ** 64, 2 ** 120, MAIN.p - 1, rng.randrange(MAIN.p)]) % field.p
        pool.append(FieldElement(v, field))
        return pool[-1]
    items = []
    for _ in range(n):
        k, field = rng.random(), rng.choice(fields)
        if k < 0.3:
            # (fresh objects: int.to_bytes never returns the interpreter's shared one-byte objects; b"" IS shared, so at most one)
            size = rng.choice([0, 1, 2, 64, 64, 64, 255, 256, 300])
            if size == 0 and any(type(o) is bytes and not o for o in items):
                size = 1
            items.append(rng.randbytes(size))
        elif k < 0.6:
            items.append([rng.randbytes(64) for _ in range(rng.choice([0, 1, 2, 12, 24]))])
        elif k < 0.75:
            items.append((element(field), element(field), element(field)))
        elif k < 0.9:
            items.append(element(field))
        else:
            items.append([element(field) for _ in range(rng.choice([0, 1, 2, 256, 1001]))])
    return items, fields


@pytest.mark.parametrize("n", [0, 1, 2, 3, 10, 50, 999, 1000, 1001, 2000, 2001, 3500])
def test_library_pickler_matches_cpython(n):
    """lists of every batch size (APPEND / MARK ... APPENDS in thousands), fresh and shared FieldElements of two fields, every
    integer opcode (BININT1/2, BININT, LONG1 of 5..17 bytes), short and long bytes, streams across many 64 KiB frames"""
    rng = random.Random(100 + n)
    for _ in range(4 if n > 100 else 40):
        items, fields = random_graph(rng, n)
        want = pickle.dumps(items)
        got = library_pickle(items, fields, rng)
        assert got == want, (n, len(got), len(want))
        assert pickle.loads(got)[:3] == items[:3]


def test_library_pickler_rejects_malformed_descriptions():
    for ops in (b"", b"X", b"L\x02\x00\x00\x00B\x01\x00\x00\x00a", b"L\x01\x00\x00\x00E\x05\x00\x00\x00" + bytes(24), b"B\x01\x00\x00\x00a", b"L\x00\x00\x00\x00junk"):
        with pytest.raises(sc.StarkCoreError):
            sc.pickle_proof(ops, MAIN.p.to_bytes(17, "little"), 1, 17)


class FakeCodeword:
    """the part of starkcore.DeviceCodeword the segments use: a field, and entries created once per index"""
    _full = None

    def __init__(self, n, field, rng):
        self.field, self.values = field, [rng.randrange(field.p) for _ in range(n)]
        self._elems = {}

    def __len__(self):
        return len(self.values)

    def raw(self, indices):
        return b"".join(self.values[i].to_bytes(16, "little") for i in indices)

    def _entries(self, indices, values):
        for i, v in zip(indices, values):
            assert v == self.values[i]
            if i not in self._elems:
                self._elems[i] = FieldElement(v, self.field)
        return [self._elems[i] for i in indices]


@pytest.mark.parametrize("seed", range(6))
def test_lazy_proof_objects_pickle_like_the_objects_they_stand_for(seed):
    """a stream shaped like FastStark.prove's: roots, the last FRI codeword, query rounds whose `c` entries are the next round's
    `a` / `b` entries and finally the last codeword's own objects, openings of committed codewords with repeated indices"""
    rng = random.Random(seed)
    field = MAIN
    s, rounds = rng.choice([2, 17, 40]), rng.choice([1, 3, 6])
    sizes = [(64 if s > 20 else 16) << (rounds - r) for r in range(rounds + 1)]
    cws = [FakeCodeword(n, field, rng) for n in sizes]
    stream = ProofStream()
    for _ in range(rng.choice([0, 3])):
        stream.push(rng.randbytes(64))
    for _ in range(rounds + 1):
        stream.push(rng.randbytes(64))
    lazy = po_.lazy_objects(stream)
    assert lazy is stream.objects and po_.lazy_objects(stream) is lazy
    last = cws[-1]
    lazy.add(po_.ElementList(last, last.raw(range(len(last)))))
    challenge_before_queries = stream.prover_fiat_shamir()
    top = rng.sample(range(sizes[0] // 2), s)
    idx = top
    for r in range(rounds):
        cur, nxt = cws[r], cws[r + 1]
        half = len(cur) // 2
        a = [i % half for i in idx]
        b = [i + half for i in a]
        depth_c, depth_n = len(cur).bit_length() - 1, len(nxt).bit_length() - 1
        paths = [np.frombuffer(rng.randbytes(64 * d * s), dtype=np.uint8).reshape(s, 64 * d) for d in (depth_c, depth_c, depth_n)]
        lazy.add(po_.FriRound(cur, nxt, a, b, a, cur.raw(a), cur.raw(b), nxt.raw(a), *paths))
        idx = a
    committed = [FakeCodeword(sizes[0], field, rng) for _ in range(3)] + [cws[0]]          # the last one also went through FRI
    opened = sorted(top + [(i + 4) % sizes[0] for i in top] + [top[0]])                     # a repeated index
    for cw in committed:
        d = len(cw).bit_length() - 1
        lazy.add(po_.Openings(cw, opened, cw.raw(opened), np.frombuffer(rng.randbytes(64 * d * len(opened)), dtype=np.uint8).reshape(len(opened), 64 * d)))
    got = stream.serialize()
    objects = list(stream.objects)
    assert len(objects) == len(stream.objects)
    assert got == pickle.dumps(objects)
    assert stream.prover_fiat_shamir() != challenge_before_queries
    # what a verifier reads back is the same graph
    back = ProofStream().deserialize(got)
    assert [type(o) for o in back.objects] == [type(o) for o in objects]
    assert back.objects[len(objects) - 2].value == objects[-2].value and back.objects[-1] == objects[-1]
    # an object the description does not cover: the stream falls back to the interpreter's pickler, same bytes
    stream.push((objects[-2], 5))
    assert stream.serialize() == pickle.dumps(list(stream.objects))
    # a subclass keeps its plain list
    class Other(ProofStream):
        pass
    assert po_.lazy_objects(Other()) is None


def test_proof_stream_without_lazy_objects_is_the_reference_stream():
    stream = ProofStream()
    stream.push(b"root")
    stream.push([FieldElement(5, MAIN)])
    assert type(stream.objects) is list and stream.serialize() == pickle.dumps(stream.objects)


@pytest.mark.parametrize("size", [65531, 65535, 65536, 65537, 70000, 262143, 262144, 300000])
@pytest.mark.parametrize("around", [0, 1, 3])
def test_bytes_objects_of_a_frame_or_more_go_out_unframed(size, around):
    """_Pickler_write_bytes of _pickle.c: a bytes object of 64 KiB or more commits the open frame and is written outside any frame;
    the library's pickler and a LazyProofObjects stream must write what CPython writes (a caller-pushed blob next to roots)."""
    rng = random.Random(size + around)
    objects = [rng.randbytes(64) for _ in range(around)] + [rng.randbytes(size)] + [rng.randbytes(64) for _ in range(around)]
    expected = pickle.dumps(objects)
    assert library_pickle(objects, [MAIN], rng) == expected
    stream = ProofStream()
    for o in objects:
        stream.push(o)
    lazy = po_.lazy_objects(stream)
    assert lazy is not None
    assert stream.serialize() == expected
    import hashlib
    assert stream.prover_fiat_shamir() == hashlib.shake_256(expected).digest(32)
    # two blobs in a row, and one as the only item of a nested list
    objects = [rng.randbytes(size), rng.randbytes(size), [rng.randbytes(size)], rng.randbytes(7)]
    assert library_pickle(objects, [MAIN], random.Random(1)) == pickle.dumps(objects)


@pytest.mark.parametrize("seed", range(4))
def test_query_phase_segment_from_the_one_call_provers_buffer(seed):
    """proof_objects.FriQueryPhase + DetachedEntries: the whole query phase described from ONE buffer in sc_fri_prove_dev's layout
    (opened elements padded to 256 bytes, paths, positions; per codeword [a, b] of its own round -- the c of the round before is one of
    them and is opened once -- and [c] for the last codeword) must pickle like the per-round segments and like the objects -- including the sharing between a round's c entries and the next
    round's a / b entries and the last codeword (pickle memoises by identity, fri.py:91, :104-105)."""
    rng = random.Random(100 + seed)
    field = MAIN
    s, rounds = rng.choice([2, 5, 40]), rng.choice([2, 3, 7])            # `rounds` codewords, rounds - 1 folds
    sizes = [(128 if s > 20 else 16) << (rounds - 1 - r) for r in range(rounds)]
    cws = [FakeCodeword(n, field, rng) for n in sizes]
    top = rng.sample(range(sizes[0] // 2), s)
    counts = [2 * s if j + 1 < rounds else (s if j > 0 else 0) for j in range(rounds)]
    depths = [n.bit_length() - 1 for n in sizes]
    positions, idx, prev = [], list(top), None
    for j in range(rounds):
        half = sizes[j] // 2
        here = []
        if j + 1 < rounds:
            idx = [i % half for i in idx]
            here += idx + [i + half for i in idx]
        elif j > 0:
            here += prev
        prev = idx
        positions.append(here)
    total = sum(counts)
    el_bytes = (16 * total + 255) & ~255
    elems = b"".join(cw.raw(p) for cw, p in zip(cws, positions))
    path_arrays = [rng.randbytes(64 * d * c) for c, d in zip(counts, depths)]
    extra_bytes = 64 * rng.choice([0, 3, 40])               # the paths of a caller's further codewords lie between the two (fast_stark.py:154-175)
    buf = np.frombuffer(elems + bytes(el_bytes - len(elems)) + b"".join(path_arrays) + rng.randbytes(extra_bytes)
                        + np.asarray([i for p in positions for i in p], dtype=np.uint64).tobytes(), dtype=np.uint8)

    def stream(kind):
        ps = ProofStream()
        for _ in range(rounds):
            ps.push(rng.randbytes(64))
        lazy = po_.lazy_objects(ps)
        if kind == "one buffer":
            holders = [cws[0]] + [po_.DetachedEntries(field) for _ in range(rounds - 1)]
        else:
            holders = cws
        lazy.add(po_.ElementList(holders[-1], cws[-1].raw(range(sizes[-1]))))
        if kind == "one buffer":
            own_paths = sum(64 * c * d for c, d in zip(counts, depths))
            lazy.add(po_.FriQueryPhase(holders, s, counts, depths, buf[:16 * total], buf[el_bytes:el_bytes + own_paths],
                                       buf[el_bytes + own_paths + extra_bytes:el_bytes + own_paths + extra_bytes + 8 * total].view(np.uint64)))
        else:
            paths = [np.frombuffer(raw, dtype=np.uint8).reshape(c, 64 * d) for raw, c, d in zip(path_arrays, counts, depths)]
            for i in range(rounds - 1):
                a, half = positions[i][:s], sizes[i] // 2
                # where the next codeword's buffer holds the path of c = a: among its own a / b, or (last codeword) on its own
                slot = [t + (0 if a[t] < sizes[i + 1] // 2 else s) for t in range(s)] if i + 2 < rounds else list(range(s))
                assert [positions[i + 1][q] for q in slot] == a
                lazy.add(po_.FriRound(cws[i], cws[i + 1], a, [x + half for x in a], a, cws[i].raw(a), cws[i].raw([x + half for x in a]), cws[i + 1].raw(a),
                                      paths[i][:s], paths[i][s:2 * s], paths[i + 1][slot]))
        return ps
    rng_state = rng.getstate()
    one = stream("one buffer")
    rng.setstate(rng_state)
    per_round = stream("per round")
    got = one.serialize()
    assert got == per_round.serialize()
    assert got == pickle.dumps(list(per_round.objects))
    assert pickle.dumps(list(one.objects)) == got           # materialised from detached holders: the same sharing
    assert len(one.objects) == rounds + 1 + 4 * s * (rounds - 1)


@pytest.mark.parametrize("seed", range(3))
def test_openings_described_by_address(seed):
    """proof_objects.Openings with its payload left in contiguous arrays (values uint8 [16 k], paths uint8 [k][64 depth], positions
    uint64 [k] -- the pinned answer buffer of sc_fri_prove_dev): the 'P' op of csrc/proof_pickle.h must write what the copying 'O' op
    and CPython write, repeated positions (one object, a memo hit) included."""
    rng = random.Random(300 + seed)
    cw = FakeCodeword(1 << 9, MAIN, rng)
    k, depth = rng.choice([1, 7, 160]), 9
    opened = sorted(rng.choice(range(len(cw))) for _ in range(k))
    values = np.frombuffer(cw.raw(opened), dtype=np.uint8).copy()
    paths = np.frombuffer(rng.randbytes(64 * depth * k), dtype=np.uint8).reshape(k, 64 * depth).copy()

    def stream(by_address):
        ps = ProofStream()
        ps.push(b"r" * 64)
        lazy = po_.lazy_objects(ps)
        lazy.add(po_.Openings(cw, opened, values if by_address else values.tobytes(), paths, np.asarray(opened, dtype=np.uint64) if by_address else None))
        return ps
    a, b = stream(True), stream(False)
    assert b"P" in a.objects._segments[-1].ops(po_._Context())[:1] and b"O" in b.objects._segments[-1].ops(po_._Context())[:1]
    assert a.serialize() == b.serialize() == pickle.dumps(list(b.objects))
