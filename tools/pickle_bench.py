#!/usr/bin/env python3
"""ProofStream.serialize() of a proof shaped like FastStark's at a 2^24 FRI domain, on the HOST only (no GPU): 3 roots + 17 FRI
roots, a last codeword, the query phase of 16 folds (40 colinearity checks) described from one buffer (proof_objects.FriQueryPhase)
and the openings of four committed codewords (proof_objects.Openings) -- the 3.2 MB the prover's last phase pickles in C
(csrc/proof_pickle.h).  Prints the time per serialize() and, once, that the bytes equal pickle.dumps of the materialised objects.
   python tools/pickle_bench.py [runs=30]"""
import os, pickle, random, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "stark-anatomy_amd"))
import proof_objects as po_
from algebra import Field
from ip import ProofStream

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = random.Random(7)
field = Field.main()
p = field.p
s, log_fri, rounds = 40, 24, 17
sizes = [1 << (log_fri - r) for r in range(rounds)]


class Holder:                                   # what the segments need of a codeword: a field, an identity, entries created once per index
    def __init__(self, f):
        self.field, self._elems = f, {}

    def _entries(self, indices, vals):
        from algebra import FieldElement
        return [self._elems.setdefault(i, FieldElement(v, self.field)) for i, v in zip(indices, vals)]


def residues(count):
    raw = np.frombuffer(rng.randbytes(16 * count), dtype=np.uint64).reshape(count, 2).copy()
    raw[:, 1] &= (1 << 62) - 1
    return raw.tobytes()


top = rng.sample(range(sizes[0] // 2), s)
counts = [2 * s if j + 1 < rounds else (s if j > 0 else 0) for j in range(rounds)]
depths = [n.bit_length() - 1 for n in sizes]
positions, idx, prev = [], list(top), None
for j in range(rounds):
    half, here = sizes[j] // 2, []
    if j + 1 < rounds:
        idx = [i % half for i in idx]
        here += idx + [i + half for i in idx]
    elif j > 0:
        here += prev
    prev = idx
    positions.append(here)
total = sum(counts)
# elements: a round's c entries ARE the next round's values at those positions -- keep them consistent (the pickler shares them)
values = [dict() for _ in range(rounds)]
elems = bytearray()
for j in range(rounds):
    for i in positions[j]:
        v = values[j].setdefault(i, residues(1))
        elems += v
el_bytes = (16 * total + 255) & ~255
own_paths = sum(64 * c * d for c, d in zip(counts, depths))
quad = sorted(rng.sample(range(sizes[0]), 4 * s))                    # (distinct positions: a repeated one would need the same value twice)
buf = np.frombuffer(bytes(elems) + bytes(el_bytes - len(elems)) + rng.randbytes(own_paths) + np.asarray([i for q in positions for i in q], dtype=np.uint64).tobytes(), dtype=np.uint8).copy()
last_values = b"".join(values[-1].get(i, residues(1)) for i in range(sizes[-1]))
committed = [(Holder(field), residues(4 * s), np.frombuffer(rng.randbytes(64 * log_fri * 4 * s), dtype=np.uint8).reshape(4 * s, 64 * log_fri).copy()) for _ in range(4)]


roots = [rng.randbytes(64) for _ in range(3 + rounds)]


def stream():
    ps = ProofStream()
    for root in roots:
        ps.push(root)
    lazy = po_.lazy_objects(ps)
    holders = [Holder(field)] + [po_.DetachedEntries(field) for _ in range(rounds - 1)]
    lazy.add(po_.ElementList(holders[-1], last_values))
    lazy.add(po_.FriQueryPhase(holders, s, counts, depths, buf[:16 * total], buf[el_bytes:el_bytes + own_paths],
                               buf[el_bytes + own_paths:el_bytes + own_paths + 8 * total].view(np.uint64)))
    for holder, vals, paths in committed:
        lazy.add(po_.Openings(holder, quad, vals, paths, np.asarray(quad, dtype=np.uint64)))
    return ps


ps = stream()
out = ps.serialize()
times = []
for _ in range(runs):
    ps = stream()
    t0 = time.perf_counter()
    out = ps.serialize()
    times.append((time.perf_counter() - t0) * 1e3)
print("serialize(): %d bytes, best %.3f ms, median %.3f ms of %d" % (len(out), min(times), sorted(times)[len(times) // 2], runs))
if "--check" in sys.argv:
    ref = pickle.dumps(list(stream().objects))
    if ref != out:
        k = next((i for i, (x, y) in enumerate(zip(ref, out)) if x != y), min(len(ref), len(out)))
        print("DIFFERS from pickle.dumps at byte %d of %d / %d: %r | %r" % (k, len(ref), len(out), ref[max(0, k - 24):k + 24], out[max(0, k - 24):k + 24]))
        sys.exit(1)
    print("equals pickle.dumps of the materialised objects")
