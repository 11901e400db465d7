"""Fiat-Shamir proof stream (host side; mirrors the interface of reference code/ip.py:4-30).

Must stay byte-identical to the reference: challenges are SHAKE-256 over pickle.dumps(objects), and
every alpha / query index of FRI derives from it (code/fri.py:79, :122).  Tiny and sequential, so it
stays on the host by design.
"""
import pickle
from hashlib import shake_256


class ProofStream:
    def __init__(self):
        self.objects = []
        self.read_index = 0

    def push(self, obj):
        self.objects.append(obj)

    def pull(self):
        assert(self.read_index < len(self.objects)), "ProofStream: cannot pull object; queue empty."
        obj = self.objects[self.read_index]
        self.read_index += 1
        return obj

    def serialize(self):
        return pickle.dumps(self.objects)

    def prover_fiat_shamir(self, num_bytes=32):
        return shake_256(self.serialize()).digest(num_bytes)

    def verifier_fiat_shamir(self, num_bytes=32):
        return shake_256(pickle.dumps(self.objects[:self.read_index])).digest(num_bytes)

    def deserialize(self, bb):
        ps = ProofStream()
        ps.objects = pickle.loads(bb)
        return ps
