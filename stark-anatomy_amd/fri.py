"""FRI low-degree test -- host orchestration over device-resident codewords.

Interface of reference code/fri.py:11-231 (`Fri(offset, omega, initial_domain_length, expansion_factor,
num_colinearity_tests)` with num_rounds / sample_index / sample_indices / eval_domain / commit / query /
prove / verify).  What moved to the GPU: the split-and-fold of fri.py:85, the Merkle commitment of every
round (fri.py:71) and the authentication paths of the query phase (fri.py:108-111), served from trees
that stay in HBM.  The Fiat-Shamir transcript (ip.py) stays on the host and is byte-identical, so alphas
and query indices match the reference.
"""
import pickle as _pickle
from hashlib import blake2b

from algebra import *
from merkle import *
from ip import *
from ntt import *
from univariate import *
import starkcore as _sc
import proof_objects as _po
from starkcore import DeviceCodeword, DeviceVector, query_codewords


class AlsoOpen:
    """Openings a caller wants together with the query phase: `requests(top_level_indices)` -> (codewords, index lists); `answers`
    = [(packed residues, paths array)] per codeword once the query phase has fetched them, or None if it took another route.
    codewords / shift (optional): the same request as data -- device codewords of the domain's length, all opened at the sorted
    positions {i, i + shift, i + N/2, i + shift + N/2 (mod N)} over the top-level indices i (fast_stark.py:154-158) -- which lets
    the library derive the positions itself and serve Fri.prove in one call (sc_fri_prove_dev)"""

    def __init__(self, requests, codewords=None, shift=None):
        self.requests, self.answers = requests, None
        self.codewords, self.shift = codewords, shift
        self.position_arrays = None                        # per codeword: its opened positions as a uint64 array next to the answers


def library_transcript(proof_stream, rounds):
    """The byte strings in `proof_stream` if the library may run `rounds` rounds of the commit phase on it itself
    (sc_fri_commit_dev computes SHAKE-256(pickle(objects)) on its own), else None.  Only a plain ProofStream: a subclass may derive
    its challenges differently (the reference's SignatureProofStream prefixes the document).  Only distinct byte strings so far
    (roots): that list has a fixed pickle layout.  The size test counts what the C side counts (3 bytes of pickle opcodes per prior
    item, 67 per root of this commit); the library still answers "unsupported" for anything else it does not take, and then the
    caller's Python loop runs."""
    if type(proof_stream) is not ProofStream or _pickle.DEFAULT_PROTOCOL != 4:
        return None                                      # (csrc/transcript.h writes protocol 4, the default of CPython 3.8 - 3.13)
    prior = list(proof_stream.objects)
    if (len(prior) + rounds < 999 and all(type(o) is bytes and len(o) < 256 for o in prior) and len(set(map(id, prior))) == len(prior)
            and sum(len(o) + 3 for o in prior) + 67 * rounds < 60000):
        return prior
    return None


class Fri:
    def __init__(self, offset, omega, initial_domain_length, expansion_factor, num_colinearity_tests):
        self.offset, self.omega, self.field = offset, omega, omega.field
        self.domain_length = initial_domain_length
        self.expansion_factor, self.num_colinearity_tests = expansion_factor, num_colinearity_tests
        assert(self.num_rounds() >= 1), "cannot do FRI with less than one round"

    def num_rounds(self):
        """commitments made by `commit`: the codeword is halved while it is longer than the expansion factor and than four times
        the number of colinearity tests (fri.py:22-28)"""
        rounds, length = 0, self.domain_length
        while length > self.expansion_factor and length > 4 * self.num_colinearity_tests:
            rounds, length = rounds + 1, length / 2
        return rounds

    def sample_index(byte_array, size):
        # fri.py:30-34 folds the bytes in with acc = (acc << 8) ^ b: the big-endian integer of the array
        # (only for byte strings: bytes(n) of an int n would be n zero bytes, where the reference's loop raises TypeError)
        if isinstance(byte_array, (bytes, bytearray, memoryview)):
            return int.from_bytes(byte_array, "big") % size
        acc = 0
        for b in byte_array:
            acc = (acc << 8) ^ int(b)
        return acc % size

    def sample_indices(self, seed, size, reduced_size, number):
        """`number` indices below `size`, pairwise distinct modulo `reduced_size` (they must not collide in the last codeword),
        drawn from blake2b(seed + counter zero bytes) for counter = 0, 1, ... (fri.py:36-51)"""
        assert(number <= reduced_size), f"cannot sample more indices than available in last codeword; requested: {number}, available: {reduced_size}"
        assert(number <= 2 * reduced_size), "not enough entropy in indices wrt last codeword"
        chosen, residues, counter = [], set(), 0
        while len(chosen) < number:
            # bytes(counter) is `counter` zero bytes (fri.py:44), not an encoding of the counter
            candidate = Fri.sample_index(blake2b(seed + bytes(counter)).digest(), size)
            counter += 1
            if candidate % reduced_size in residues:
                continue
            residues.add(candidate % reduced_size)
            chosen.append(candidate)
        return chosen

    def eval_domain(self):
        # offset * omega^i, i < domain_length, by running product
        points, x = [], self.offset
        for _ in range(self.domain_length):
            points.append(x)
            x = x * self.omega
        return points

    def _check_omega_order(self, length):
        """fri.py:68 asserts omega_r^(N_r - 1) == omega_r^-1, i.e. omega_r^(N_r) == 1, in every round; omega_r = omega^(2^r) and
        N_r = N / 2^r, so every round's condition is omega^N == 1: checked once per (omega, length) -- a power and an inversion in
        Python integers are 0.1 ms, a twentieth of a 2^22 proof"""
        key = (self.omega.value, length)
        if getattr(self, "_order_checked", None) != key:
            assert(self.omega ^ (length - 1) == self.omega.inverse()), "error in commit: omega does not have the right order!"
            self._order_checked = key

    def _on_device(self, codeword):
        if isinstance(codeword, DeviceCodeword):
            return codeword
        return DeviceCodeword.from_list(codeword, self.field)

    def commit(self, codeword, proof_stream, round_index=0):
        """fri.py:66-94.  The chain root -> alpha -> fold -> next root is strictly serial.  When the proof stream holds nothing but
        digests so far (always the case for Fri.prove on its own and inside FastStark.prove) the whole round loop -- trees, the
        Fiat-Shamir step of ip.py:18-25, folds -- is ONE library call (sc_fri_commit_dev): nothing crosses the language boundary
        between a root arriving and the next launch.  Otherwise (_commit_rounds) a round's fold and tree are enqueued in one call
        and only the Fiat-Shamir step sits between a root arriving and the next launch."""
        codeword = self._on_device(codeword)
        rounds = self.num_rounds()
        self._check_omega_order(len(codeword))             # (before anything is enqueued)
        codewords = None
        if len(codeword) >= 2 and codeword._tree is None and library_transcript(proof_stream, rounds) is not None:
            codewords = self._commit_in_library(codeword, proof_stream, rounds)
        if codewords is None:
            codewords = self._commit_rounds(codeword, proof_stream, rounds)
        # the last codeword goes out in the clear, as a plain list (it is pickled into the transcript) -- described, not built,
        # when the stream can hold it that way (proof_objects: the transcript bytes are the same)
        lazy = _po.lazy_objects(proof_stream) if _po.eligible(codewords[-1]) else None
        if lazy is not None:
            lazy.add(_po.ElementList(codewords[-1], codewords[-1].vec.to_bytes()))
        else:
            proof_stream.push(codewords[-1].tolist())
        return codewords

    def _commit_in_library(self, codeword, proof_stream, rounds):
        import ctypes
        prior = proof_stream.objects
        k = len(prior)
        vecs = (ctypes.c_void_p * max(1, rounds - 1))()
        trees = (ctypes.c_void_p * rounds)()
        roots = ctypes.create_string_buffer(64 * rounds)
        alphas = (ctypes.c_uint64 * max(2, 2 * (rounds - 1)))()
        rc = _sc.lib().sc_fri_commit_dev(codeword.vec.ptr, len(codeword), _sc.fe_bytes(self.offset.value), _sc.fe_bytes(self.omega.value), rounds,
                                         b"".join(prior), (ctypes.c_uint32 * max(1, k))(*map(len, prior)), k, vecs, trees, roots, alphas, None)
        if rc == _sc.SC_ERR_UNSUPPORTED:
            return None                                  # a transcript shape the library does not write: the caller runs _commit_rounds
        _sc._check(rc)
        codewords, cur, raw = [], codeword, roots.raw
        for r in range(rounds):
            n = len(codeword) >> r
            if r > 0:
                cur = DeviceCodeword(DeviceVector.adopt(vecs[r - 1], n), self.field)
            root = raw[64 * r:64 * r + 64]
            cur._tree = _sc.MerkleTree(ctypes.c_void_p(trees[r]), root, n)
            proof_stream.push(root)
            codewords.append(cur)
        return codewords

    def _commit_rounds(self, codeword, proof_stream, rounds):
        """the same loop with the Fiat-Shamir step in Python (any proof stream): a round's fold and tree are ENQUEUED in one library
        call (sc_fri_fold_commit_dev) and everything the host can do without the root (the next output vector) happens while the
        device hashes"""
        omega, offset = self.omega, self.offset
        codewords = []
        for r in range(rounds):
            N = len(codeword)
            if r == 0:
                codeword.start_tree()
            folded = DeviceVector(N // 2) if r < rounds - 1 else None
            # Merkle root of this round's codeword (waits for the device); the tree stays in HBM for the query phase
            proof_stream.push(codeword.tree().root)
            if r == rounds - 1:
                break
            alpha = self.field.sample(proof_stream.prover_fiat_shamir())
            codewords.append(codeword)
            codeword = codeword.fold_commit(alpha, offset, omega, folded)
            omega = omega ^ 2
            offset = offset ^ 2
        codewords.append(codeword)
        return codewords

    def query(self, current_codeword, next_codeword, c_indices, proof_stream):
        """one round of the query phase (fri.py:98-113): the s colinear triples (current[c], current[c + half], next[c]), then their
        authentication paths; returns the opened positions of the current codeword"""
        current, following = self._on_device(current_codeword), self._on_device(next_codeword)
        s, half = self.num_colinearity_tests, len(current) // 2
        lower = list(c_indices)
        upper = [c + half for c in lower]
        # one round trip per codeword: opened entries and their authentication paths together
        entries, paths = current.query(lower[:s] + upper[:s])
        next_entries, next_paths = following.query(lower[:s])
        for t in range(s):
            proof_stream.push((entries[t], entries[s + t], next_entries[t]))
        for t in range(s):
            for path in (paths[t], paths[s + t], next_paths[t]):
                proof_stream.push(path)
        return lower + upper

    def prove(self, codeword, proof_stream, also_open=None):
        """also_open (optional, an AlsoOpen): further device codewords to open at positions that depend on the sampled indices --
        FastStark's committed codewords (fast_stark.py:154-175) -- fetched in the SAME device round trip as the query phase"""
        assert(self.domain_length == len(codeword)), "initial codeword length does not match length of initial codeword"
        top_level_indices = self._prove_in_library(codeword, proof_stream, also_open)
        if top_level_indices is not None:
            return top_level_indices
        codewords = self.commit(codeword, proof_stream)
        top_level_indices = self.sample_indices(proof_stream.prover_fiat_shamir(), len(codewords[0]) // 2, len(codewords[-1]), self.num_colinearity_tests)
        self._query_all(codewords, top_level_indices, proof_stream, also_open)
        return top_level_indices

    def _prove_in_library(self, codeword, proof_stream, also_open):
        """fri.py:115-130 as ONE library call (sc_fri_prove_dev): commit phase, the challenge over the transcript with the last
        codeword, the sampled indices, and one kernel that writes every opening of the proof into pinned host memory.  What comes
        back is pushed as the reference pushes it -- roots, the last codeword, per round s triples and 3 s paths -- in described
        form (proof_objects); None when the stream or the codewords are not of the kind the library serves (the phases then run
        one by one, with the same bytes)."""
        import ctypes
        import numpy as np
        rounds, s = self.num_rounds(), self.num_colinearity_tests
        N = self.domain_length
        if not (isinstance(codeword, DeviceCodeword) and _po.eligible(codeword) and codeword._tree is None and N >= 2 and N & (N - 1) == 0):
            return None
        if rounds < 2:                  # (fri.py:122 reads codewords[1]: the reference itself has no proof with fewer than two codewords)
            return None
        if self.field.p != Field.P_MAIN or s > (N >> (rounds - 1)) or s < 1 or library_transcript(proof_stream, rounds) is None:
            return None
        extra = []
        if also_open is not None:
            extra = also_open.codewords
            if extra is None or also_open.shift is None or not all(isinstance(cw, DeviceCodeword) and _po.eligible(cw) and len(cw) == N for cw in extra):
                return None
        if rounds + len(extra) > 32:
            return None
        self._check_omega_order(N)
        prior = list(proof_stream.objects)
        k, ne = len(prior), len(extra)
        roots = ctypes.create_string_buffer(64 * rounds)
        n_last = N >> (rounds - 1)
        last_raw = ctypes.create_string_buffer(16 * n_last)
        top = (ctypes.c_uint64 * s)()
        quad = (ctypes.c_uint64 * (4 * s))()
        # openings per pair (see include/starkcore.h): codeword j: the 2 s of its own round (j < rounds - 1) -- the c positions of the round
        # before are among them -- and the last codeword the s of the round before
        counts = [2 * s if j + 1 < rounds else (s if j > 0 else 0) for j in range(rounds)] + [4 * s] * ne
        depths = [(N >> j).bit_length() - 1 for j in range(rounds)] + [N.bit_length() - 1] * ne
        total = sum(counts)
        el_bytes = (16 * total + 255) & ~255
        path_bytes = sum(64 * c * d for c, d in zip(counts, depths))
        # Lifetime: `answers` is pinned host memory from the library's pool; the segments added to the stream below (and
        # also_open.answers) are VIEWS of it, so it stays page-locked for as long as the stream's described objects live and goes
        # back to the pool when the stream is dropped after serialize().  A caller that keeps streams calls
        # proof_stream.objects.detach() (proof_objects.LazyProofObjects.detach) to hold plain objects instead.
        answers = _sc.HostBuffer(el_bytes + path_bytes + 8 * total)
        extra_trees = [cw.tree() for cw in extra]
        # no handle of the commit phase comes back (vecs_out = trees_out = NULL): nothing reads the folded codewords or their trees
        # once the openings are on the host, and the library hands their memory back before it returns
        rc = _sc.lib().sc_fri_prove_dev(codeword.vec.ptr, N, _sc.fe_bytes(self.offset.value), _sc.fe_bytes(self.omega.value), rounds, s,
                                        b"".join(prior), (ctypes.c_uint32 * max(1, k))(*map(len, prior)), k,
                                        ne, (ctypes.c_void_p * max(1, ne))(*[t._h for t in extra_trees]),
                                        (ctypes.c_void_p * max(1, ne))(*[cw.vec.ptr for cw in extra]), int(also_open.shift) if ne else 0,
                                        None, None, roots, None, last_raw, top, quad, answers.ptr, answers.nbytes, None)
        if rc == _sc.SC_ERR_UNSUPPORTED:
            return None
        _sc._check(rc)
        # the commit phase's objects (fri.py:71, :91): the roots, then the last codeword in the clear
        raw = roots.raw
        for r in range(rounds):
            proof_stream.push(raw[64 * r:64 * r + 64])
        holders = [_po.entries_of(codeword)] + [_po.DetachedEntries(self.field) for _ in range(rounds - 1)]
        lazy = _po.lazy_objects(proof_stream)
        lazy.add(_po.ElementList(holders[-1], last_raw.raw))
        # the query phase's objects (fri.py:104-113): described by the answers as they lie in the pinned buffer
        data = answers.array
        own = sum(counts[:rounds])                               # openings of the commit phase's codewords; the caller's come behind them
        own_paths = sum(64 * c * d for c, d in zip(counts[:rounds], depths[:rounds]))
        if rounds > 1:
            positions = data[el_bytes + path_bytes:el_bytes + path_bytes + 8 * own].view(np.uint64)
            lazy.add(_po.FriQueryPhase(holders, s, counts[:rounds], depths[:rounds], data[:16 * own], data[el_bytes:el_bytes + own_paths], positions))
        if also_open is not None:
            vo, po, fetched = own, el_bytes + own_paths, []
            where = data[el_bytes + path_bytes:el_bytes + path_bytes + 8 * total].view(np.uint64)
            for c, d in zip(counts[rounds:], depths[rounds:]):
                fetched.append((data[16 * vo:16 * (vo + c)], data[po:po + 64 * c * d].reshape(c, 64 * d)))
                vo += c
                po += 64 * c * d
            also_open.answers = fetched
            also_open.position_arrays = [where[own + 4 * s * e:own + 4 * s * (e + 1)] for e in range(ne)]      # the same, as the library wrote them
        return list(top)

    def _query_all(self, codewords, top_level_indices, proof_stream, also_open=None):
        """The query phase of fri.py:124-128 with ONE device round trip for all rounds: everything a codeword has to open
        (its a/b entries for its own round, the c entries of the previous round) is fetched together, then pushed in the
        reference's order (per round: s triples, then 3*s paths as a, b, c)."""
        s = self.num_colinearity_tests
        rounds = len(codewords) - 1
        halves = [len(cw) // 2 for cw in codewords]           # (len() of a device codeword is a call into the binding: once each)
        per_round, indices = [], [index for index in top_level_indices]
        for i in range(rounds):
            half = halves[i]
            indices = [index % half for index in indices]
            per_round.append(indices)
        requests = []
        for j in range(len(codewords)):
            request = []
            if j < rounds:
                half = halves[j]
                request += per_round[j][:s] + [index + half for index in per_round[j][:s]]
            if j > 0:
                request += per_round[j - 1][:s]
            requests.append(request)
        lazy = _po.lazy_objects(proof_stream) if all(_po.eligible(cw) for cw in codewords) else None
        if lazy is not None:
            # the device's answers go into the stream as they are: residues and paths stay packed, the transcript is pickled from
            # their description (proof_objects.FriRound); the reference's objects exist only if somebody reads them
            more_codewords, more_requests = also_open.requests(top_level_indices) if also_open is not None else ([], [])
            fetched = _sc.query_codewords_raw(list(codewords) + list(more_codewords), requests + list(more_requests))
            if also_open is not None:
                also_open.answers = fetched[len(codewords):]
            for i in range(rounds):
                values, paths = fetched[i]
                next_values, next_paths = fetched[i + 1]
                c_at = 2 * s if i + 1 < rounds else 0
                a = per_round[i][:s]
                half = halves[i]
                lazy.add(_po.FriRound(codewords[i], codewords[i + 1], a, [index + half for index in a], a,
                                      values[:16 * s], values[16 * s:32 * s], next_values[16 * c_at:16 * (c_at + s)],
                                      paths[:s], paths[s:2 * s], next_paths[c_at:c_at + s]))
            return
        if all(isinstance(cw, DeviceCodeword) for cw in codewords):
            fetched = query_codewords(codewords, requests)          # every round's openings in one device round trip
        else:
            fetched = [cw.query(request) for cw, request in zip(codewords, requests)]
        # pushes in the reference's order (fri.py:104-113 per round: s triples, then per test the paths of a, b, c); a ProofStream's
        # `push` is `objects.append`, so a whole round goes in with two list extensions instead of 4 s method calls
        objects = proof_stream.objects if type(proof_stream) is ProofStream else None
        for i in range(rounds):
            entries, paths = fetched[i]
            next_entries, next_paths = fetched[i + 1]
            c_at = 2 * s if i + 1 < rounds else 0
            triples = list(zip(entries[:s], entries[s:2 * s], next_entries[c_at:c_at + s]))
            openings = [p for trio in zip(paths[:s], paths[s:2 * s], next_paths[c_at:c_at + s]) for p in trio]
            if objects is not None:
                objects.extend(triples)
                objects.extend(openings)
            else:
                for obj in triples + openings:
                    proof_stream.push(obj)

    def _last_codeword_degree(self, last_codeword, last_omega, last_offset):
        """Degree of the interpolant of the last codeword on its coset: intt + unscale (the route the
        reference's fri.py:165-166 / docs describe) -- same unique polynomial as Lagrange at fri.py:164."""
        coefficients = intt(last_omega, last_codeword)
        poly = Polynomial(coefficients).scale(last_offset.inverse())
        assert(fast_coset_evaluate(poly, last_offset, last_omega, len(last_codeword)) == last_codeword), "re-evaluated codeword does not match original!"
        return poly.degree()

    @staticmethod
    def _reject(*lines):
        for line in lines:
            print(line)
        return False

    def verify(self, proof_stream, polynomial_values):
        rounds, s = self.num_rounds(), self.num_colinearity_tests
        # the commit phase replayed: per round a root and the challenge it determines, then the last codeword in the clear
        roots, alphas = [], []
        for _ in range(rounds):
            roots.append(proof_stream.pull())
            alphas.append(self.field.sample(proof_stream.verifier_fiat_shamir()))
        last_codeword = proof_stream.pull()
        if Merkle.commit(last_codeword) != roots[-1]:
            return Fri._reject("last codeword is not well formed")
        # the last codeword lives on the coset after rounds - 1 squarings; it must be of low degree there
        squarings = 1 << (rounds - 1)
        last_omega, last_offset = self.omega ^ squarings, self.offset ^ squarings
        assert(last_omega.inverse() == last_omega ^ (len(last_codeword) - 1)), "omega does not have right order"
        allowed = len(last_codeword) // self.expansion_factor - 1
        observed = self._last_codeword_degree(last_codeword, last_omega, last_offset)
        if observed > allowed:
            return Fri._reject("last codeword does not correspond to polynomial of low enough degree", "observed degree: %s" % observed,
                               "but should be: %s" % allowed)
        # the query phase: consistency of consecutive codewords at the sampled positions
        top_level_indices = self.sample_indices(proof_stream.verifier_fiat_shamir(), self.domain_length >> 1, self.domain_length >> (rounds - 1), s)
        omega, offset = self.omega, self.offset
        for r in range(rounds - 1):
            half = self.domain_length >> (r + 1)
            lower = [index % half for index in top_level_indices]
            upper = [index + half for index in lower]
            triples = [proof_stream.pull() for _ in range(s)]
            for a, b, (ya, yb, yc) in zip(lower, upper, triples):
                if r == 0:
                    polynomial_values += [(a, ya), (b, yb)]
                # (x_a, y_a), (x_b, y_b) and (alpha, y_c) lie on one line: that is the fold of fri.py:85 read backwards
                if not test_colinearity([(offset * (omega ^ a), ya), (offset * (omega ^ b), yb), (alphas[r], yc)]):
                    return Fri._reject("colinearity check failure")
            for a, b, (ya, yb, yc) in zip(lower, upper, triples):
                for root, position, leaf, label in ((roots[r], a, ya, "aa"), (roots[r], b, yb, "bb"), (roots[r + 1], a, yc, "cc")):
                    if not Merkle.verify(root, position, proof_stream.pull(), leaf):
                        return Fri._reject("merkle authentication path verification fails for " + label)
            omega, offset = omega ^ 2, offset ^ 2
        return True
