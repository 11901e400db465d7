"""Fiat-Shamir proof stream (host side; mirrors the interface of reference code/ip.py:4-30).

Must stay byte-identical to the reference: challenges are SHAKE-256 over pickle.dumps(objects), and
every alpha / query index of FRI derives from it (code/fri.py:79, :122).  Tiny and sequential, so it
stays on the host by design.
"""
import pickle
from hashlib import shake_256


def _pickled(objects):
    """pickle.dumps(objects) -- of the plain list the reference keeps, or of the prover's lazily described objects
    (proof_objects.LazyProofObjects: the same bytes without a Python object per digest)"""
    pickled = getattr(objects, "pickled", None)
    return pickled() if pickled is not None else pickle.dumps(objects)


def _challenge(objects, num_bytes):
    """SHAKE-256 of the pickled transcript prefix (default pickle protocol: the bytes are part of the protocol)."""
    return shake_256(_pickled(objects)).digest(num_bytes)


class ProofStream:
    """An append-only list of proof objects plus the verifier's read cursor."""

    def __init__(self):
        self.objects = []
        self.read_index = 0

    # -- prover side
    def push(self, obj):
        self.objects.append(obj)

    def serialize(self):
        return _pickled(self.objects)

    def prover_fiat_shamir(self, num_bytes=32):
        # the prover hashes everything sent so far
        return _challenge(self.objects, num_bytes)

    # -- verifier side
    def deserialize(self, bb):
        stream = ProofStream()
        stream.objects = pickle.loads(bb)
        return stream

    def pull(self):
        assert(self.read_index < len(self.objects)), "ProofStream: cannot pull object; queue empty."
        self.read_index += 1
        return self.objects[self.read_index - 1]

    def verifier_fiat_shamir(self, num_bytes=32):
        # the verifier hashes only what it has read so far
        return _challenge(self.objects[:self.read_index], num_bytes)
