"""FRI on a codeword sharded over the GPUs of one node: Merkle commitments, folds and openings on column slabs (ShardedFri: the
layout ShardedNtt's transforms leave a codeword in -- no element ever crosses a rank boundary) and on contiguous slabs
(ContiguousFri: one neighbour exchange per fold), independent columns spread over the ranks (ColumnReplicas), and the local
primitives they run on (HipFriEngine: the C-ABI on torch-owned memory).

Reference: code/fri.py:56-130 (commit, query, prove), code/merkle.py:13-27; the reference is single-process, so the partitioning is
this build's; every rank ends with the reference's transcript, byte for byte (tests/test_sharded_cpu.py under gloo with the oracle
as the engine, tests/test_gpu_sharded.py with the HIP engine on ranks sharing the GPU).
"""
import itertools

import torch
import torch.distributed as dist

from sharded import P, _fe, _current_raw_stream, gather_natural      # noqa: F401


class HipFriEngine:
    """Local FRI primitives on torch-owned slabs through the C-ABI (folds, Merkle trees, openings)."""

    def __init__(self, device):
        import ctypes
        import starkcore as sc
        self.sc, self.lib, self.device, self.ctypes = sc, sc.lib(), device, ctypes

    class _Tree:
        def __init__(self, tree, keep):
            self.tree, self.keep = tree, keep

        @property
        def root(self):
            return self.tree.root          # waits for an asynchronous build

        def open(self, indices):
            return self.tree.open_batch(list(indices))

    def _stream(self):
        """The primitives run on torch's CURRENT stream, so they are ordered with the tensor ops and collectives around them and
        need no synchronization of their own.  Only under torch's null stream (which the library cannot share) they run on the
        library stream between two explicit synchronizations."""
        raw = _current_raw_stream(self.device)
        if raw == 0:
            torch.cuda.current_stream(self.device).synchronize()
            return None
        return self.ctypes.c_void_p(raw)

    def _done(self, sptr):
        if sptr is None:
            self.sc.synchronize()

    def tree(self, elems, need_root=True):
        """Merkle tree over a contiguous tensor of field elements [..., 2].  need_root=False: the build is only enqueued on the
        current stream (a local subtree of a sharded commit: its sub-root level is copied out on the same stream, its own root
        is never looked at)."""
        elems = elems.contiguous()
        sptr = self._stream()
        if need_root or sptr is None:
            return HipFriEngine._Tree(self.sc.MerkleTree.from_device_ptr(elems.data_ptr(), elems.numel() // 2, sptr), elems)
        return HipFriEngine._Tree(self.sc.MerkleTree.from_device_ptr_noroot(elems.data_ptr(), elems.numel() // 2, sptr), elems)

    def level(self, tree, level):
        count = tree.tree.n >> level
        out = torch.empty((count, 8), dtype=torch.int64, device=self.device)
        sptr = self._stream()
        tree.tree.copy_level(level, out.data_ptr(), sptr)
        self._done(sptr)
        return out

    def tree_from_digests(self, digests):
        digests = digests.contiguous()
        return HipFriEngine._Tree(self.sc.MerkleTree.from_digests_ptr(digests.data_ptr(), digests.numel() // 8, self._stream()), digests)

    def fold_slab(self, src, rows, cols, R, col_base, alpha, offset, omega):
        dst = torch.empty((rows // 2, cols, 2), dtype=torch.int64, device=self.device)
        sptr = self._stream()
        self.sc._check(self.lib.sc_fri_fold_slab_dev(src.data_ptr(), rows, cols, R, col_base, _fe(alpha), _fe(offset), _fe(omega), dst.data_ptr(), sptr))
        self._done(sptr)
        return dst

    def fold_slab_tree(self, src, rows, cols, R, col_base, alpha, offset, omega):
        """fold_slab AND the enqueue-only local subtree over the folded slab (the next round's `tree(slab, need_root=False)`) in
        one library call: the tree's leaf stage computes the fold.  Returns (folded slab, tree); tree None under torch's null
        stream (the caller builds it the plain way)."""
        sptr = self._stream()
        if sptr is None:
            return self.fold_slab(src, rows, cols, R, col_base, alpha, offset, omega), None
        dst = torch.empty((rows // 2, cols, 2), dtype=torch.int64, device=self.device)
        tree = self.sc.MerkleTree.from_folded_slab(src.data_ptr(), rows, cols, R, col_base, _fe(alpha), _fe(offset), _fe(omega), dst.data_ptr(), sptr)
        return dst, HipFriEngine._Tree(tree, dst)

    def fold_full(self, src, N, alpha, offset, omega):
        dst = torch.empty((N // 2, 2), dtype=torch.int64, device=self.device)
        sptr = self._stream()
        self.sc._check(self.lib.sc_fri_fold_dev(src.data_ptr(), N, _fe(alpha), _fe(offset), _fe(omega), dst.data_ptr(), sptr))
        self._done(sptr)
        return dst

    class _LibraryVector:
        """a folded codeword the library handed out (sc_fri_commit_dev), behind the two things the layer records ask of a tensor"""

        def __init__(self, vec):
            self.vec = vec

        def data_ptr(self):
            return self.vec.ptr

        def contiguous(self):
            return self

    def commit_rounds(self, full, N, offset, omega, rounds, prior):
        """The remaining `rounds` rounds of the commit phase on a codeword every rank holds whole (`full`, N elements): trees,
        Fiat-Shamir steps and folds in ONE library call (sc_fri_commit_dev, what Fri.commit uses on one GPU).  prior: the byte
        strings in the proof stream so far.  [(codeword, tree, root)] per round, or None when the library does not take the
        transcript (or under torch's null stream): the caller's round loop runs then."""
        sptr = self._stream()
        if sptr is None:
            return None
        ct, sc = self.ctypes, self.sc
        full = full.contiguous()
        k = len(prior)
        vecs = (ct.c_void_p * max(1, rounds - 1))()
        trees = (ct.c_void_p * rounds)()
        roots = ct.create_string_buffer(64 * rounds)
        alphas = (ct.c_uint64 * max(2, 2 * (rounds - 1)))()
        rc = self.lib.sc_fri_commit_dev(full.data_ptr(), N, _fe(offset), _fe(omega), rounds, b"".join(prior), (ct.c_uint32 * max(1, k))(*map(len, prior)), k,
                                        vecs, trees, roots, alphas, sptr)
        if rc == sc.SC_ERR_UNSUPPORTED:
            return None
        sc._check(rc)
        out, raw = [], roots.raw
        for r in range(rounds):
            n = N >> r
            vec = full if r == 0 else HipFriEngine._LibraryVector(sc.DeviceVector.adopt(vecs[r - 1], n))
            root = raw[64 * r:64 * r + 64]
            out.append((vec, HipFriEngine._Tree(sc.MerkleTree(ct.c_void_p(trees[r]), root, n), vec), root))
        return out

    def lde(self, coeffs, offset, generator, order):
        """fast_coset_evaluate (code/ntt.py:132-135) of packed coefficients (bytes) -> device tensor [order][2]"""
        m = len(coeffs) // 16
        src = self.sc.DeviceVector.from_bytes(coeffs) if m else self.sc.DeviceVector(1)
        out = torch.empty((order, 2), dtype=torch.int64, device=self.device)
        sptr = self._stream()
        self.sc._check(self.lib.sc_coset_evaluate_dev(src.ptr, m, _fe(offset), _fe(generator), order, out.data_ptr(), sptr))
        torch.cuda.current_stream(self.device).synchronize()      # `src` is freed on return: its reader must be done
        self._done(sptr)
        return out

    def query_many(self, requests, raw_paths=False, raw_values=False):
        """[(tree, elems tensor or None, indices[, keep])] -> [(values as ints or None, authentication paths)]: every opening of
        every layer in ONE library call and one launch (sc_merkle_query_multi_dev), instead of a device round trip per tree.
        keep: only the first `keep` digests of each path are wanted (the part below a sharded commitment's sub-roots).
        raw_paths: the paths of a request come back as ONE uint8 array [openings][64 * digests] instead of lists of bytes
        objects (the sharded openings join two such parts per path before any object is made).  raw_values: the opened elements
        come back as a uint8 array [openings][16] (packed residues, as the device wrote them) instead of Python ints."""
        import numpy as np
        ct, sc = self.ctypes, self.sc
        requests = [(r[0], r[1], r[2], r[3] if len(r) > 3 else None) for r in requests]
        live = [(q, t, e, idx, keep) for q, (t, e, idx, keep) in enumerate(requests) if len(idx)]
        no_values = np.zeros((0, 16), dtype=np.uint8) if raw_values else []
        if raw_paths:
            out = [(None if e is None else no_values, np.zeros((0, 0), dtype=np.uint8)) for t, e, idx, _ in requests]
        else:
            out = [(None if e is None else no_values, [[] for _ in idx]) for t, e, idx, _ in requests]
        if not live:
            return out
        torch.cuda.current_stream(self.device).synchronize()        # the library call runs on the library's stream
        n = len(live)
        counts = [len(idx) for _, _, _, idx, _ in live]
        flat = np.fromiter(itertools.chain.from_iterable(idx for _, _, _, idx, _ in live), dtype=np.uint64, count=sum(counts))
        total = int(flat.size)
        depths = [t.tree.depth for _, t, _, _, _ in live]
        path_bytes = sum(64 * d * k for d, k in zip(depths, counts))
        elems_out = np.empty((total, 16), dtype=np.uint8)            # (every byte is written by the call: nothing to zero first)
        paths_out = np.empty(max(path_bytes, 64), dtype=np.uint8)
        # a tree over digests has no element vector: any readable pointer will do, the value is not used
        ptrs = [(e if e is not None else t.keep).data_ptr() for _, t, e, _, _ in live]
        sc._check(self.lib.sc_merkle_query_multi_dev(n, (ct.c_void_p * n)(*[t.tree._h for _, t, _, _, _ in live]), (ct.c_void_p * n)(*ptrs),
                                                     flat.ctypes.data_as(ct.POINTER(ct.c_uint64)), (ct.c_uint64 * n)(*counts),
                                                     elems_out.ctypes.data_as(ct.c_void_p), paths_out.ctypes.data_as(ct.c_void_p)))
        values = elems_out if raw_values else sc.unpack(elems_out.tobytes(), total)
        view = memoryview(paths_out)
        vo = po = 0
        for (q, t, e, idx, keep), d, k in zip(live, depths, counts):
            if raw_paths:
                whole = paths_out[po:po + 64 * k * d].reshape(k, 64 * d)
                paths = whole if keep is None or keep >= d else whole[:, :64 * keep]
            else:
                paths = sc._path_lists(view, po, d, k, keep)
            out[q] = (values[vo:vo + k] if e is not None else None, paths)
            vo += k
            po += 64 * k * d
        return out

    def read(self, elems, flat_indices):
        """values (Python ints) of elems.view(-1, 2)[flat_indices]"""
        if len(flat_indices) == 0:
            return []
        if isinstance(elems, HipFriEngine._LibraryVector):
            torch.cuda.current_stream(self.device).synchronize()        # the copy below runs on the library's stream
            values = self.sc.unpack(elems.vec.to_bytes(), elems.vec.n)
            return [values[i] for i in flat_indices]
        idx = torch.tensor(list(flat_indices), dtype=torch.int64, device=elems.device)
        got = elems.reshape(-1, 2)[idx].cpu().tolist()
        m = (1 << 64) - 1
        return [((hi & m) << 64) | (lo & m) for lo, hi in got]


class _LayerEntries:
    """the entries of one committed layer as FieldElement objects, each made once (pickle memoises by identity: the `c` of one
    round is the `a` / `b` of the next, code/fri.py:104-105) -- the object cache of a layer record behind the interface
    proof_objects' segments use"""
    _full = None

    def __init__(self, layer, field):
        # (only the cache: a reference to the layer record would close a cycle layer -> holder -> layer, and the layer's trees --
        # gigabytes at a 2^24 domain -- would then wait for the cycle collector instead of going back to the pool with the proof)
        self.cache, self.field = layer["cache"], field

    def _entries(self, indices, values):
        from algebra import FieldElement
        cache, field, new = self.cache, self.field, object.__new__
        for i, v in zip(indices, values):
            if i not in cache:
                e = new(FieldElement)
                e.value = v
                e.field = field
                cache[i] = e
        return [cache[i] for i in indices]


class ShardedFri:
    """`Fri.prove` (reference code/fri.py:115-130) on a codeword that lives in the column-slab layout.

    The codeword of length N = C*R is the row-major C x R matrix (index i = row*R + col); rank g owns the columns
    [g*R/G, (g+1)*R/G) as a contiguous [C][R/G] tensor -- exactly what ShardedNtt.forward() leaves behind.  In this layout
      * split-and-fold needs NO exchange: i and i + N/2 are rows `row` and `row + C/2` of the same columns;
      * a Merkle commit needs ONE all-gather of C digests per rank: the bottom log2(R/G) levels are whole subtrees of the
        rank's slab, the levels above are rebuilt (redundantly, identically) from the gathered sub-roots on every rank;
      * when a fold leaves a single row (length R) the codeword is all-gathered once and the remaining rounds run locally.
    Every rank drives the same Fiat-Shamir transcript (roots are replicated), so alphas and query indices agree without
    any broadcast, and every rank ends up with the identical, reference-identical proof stream.
    """

    # A 2^16-leaf tree is two latency-bound launches (0.066 ms) whatever the number of ranks; its sharded form -- local
    # subtree, level copy, all-gather, top tree -- is four steps of the same kind plus a collective.  Above 2^17 nodes the hashing is
    # throughput-bound and sharding pays.
    LOCAL_TAIL = 1 << 16

    one_rank_local = False      # (set per instance below; subclasses with their own constructor keep every round sharded)

    def __init__(self, fri, R, rank, world, device, engine=None, group=None, local_tail=None):
        """local_tail: once a round's codeword is this short it is gathered on every rank and the remaining rounds run locally
        (replicated): such rounds are bound by launch and hashing latency on any number of ranks, a collective per round only
        adds to it, and on the HIP engine the rest of the commit phase is then ONE library call (0: only when one row is left)."""
        self.fri, self.R, self.rank, self.world, self.device, self.group = fri, int(R), rank, world, device, group
        assert R % world == 0 and fri.domain_length % R == 0
        self.Rw = self.R // world
        self.engine = engine if engine is not None else HipFriEngine(device)
        self.local_tail = self.LOCAL_TAIL if local_tail is None else int(local_tail)
        # ONE rank (and no explicit local_tail, which asks for the slab rounds): the "slab" is the whole codeword in natural order --
        # one tree per commitment instead of a subtree plus a tree over its sub-roots, no sub-root level copied out, nothing
        # gathered, and the whole commit phase is one library call.  What a rank pays for the sharded layout when nobody shares it.
        self.one_rank_local = world == 1 and local_tail is None

    # -- collectives ------------------------------------------------------------------------------
    def _all_gather(self, t):
        t = t.contiguous()
        if self.world == 1:
            return t.unsqueeze(0)
        if t.is_cuda and dist.get_backend(self.group) == "gloo":        # functional tests with several ranks on one GPU: host-staged
            host = t.cpu()
            parts = [torch.empty_like(host) for _ in range(self.world)]
            dist.all_gather(parts, host, group=self.group)
            return torch.stack(parts, dim=0).to(t.device)
        if t.is_cuda:                                                   # RCCL: one output tensor, no per-rank pieces to stack
            out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
            dist.all_gather_into_tensor(out, t, group=self.group)
            return out
        parts = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(parts, t, group=self.group)
        return torch.stack(parts, dim=0)

    def _gather_answers(self, layout, mine, sizes, packed=False):
        """The owners' answers to the openings, merged with ONE fixed-shape tensor collective (no pickling, no object store).
        layout[r] = [(q, positions, ndigests)]: the runs rank r answers (one per request it owns something of), in the order it
        packs them -- every rank derives all of it from the public indices; `mine` = this rank's runs [(values, bottoms)] in that
        order, bottoms = uint8 array [openings][64 * ndigests]; sizes[q] = number of openings of request q.  Returns
        {q: (values, bottoms)}: a list and a uint8 array, both indexed by the position in the request.  No object per digest is
        made here: the caller joins these path bottoms with the path tops first.  packed: the values are uint8 arrays
        [openings][16] (packed residues) on the way in and on the way out, and no integer object is made either."""
        import numpy as np
        import starkcore as sc
        G, g = self.world, self.rank
        answers = {}

        def place(q, positions, vals, bottoms):
            if len(positions) == sizes[q]:                       # one owner for the whole request: its run IS the answer
                answers[q] = (vals, bottoms)
                return
            have = answers.get(q)
            if have is None:
                have = answers[q] = (np.zeros((sizes[q], 16), dtype=np.uint8) if packed else [None] * sizes[q],
                                     np.zeros((sizes[q], bottoms.shape[1]), dtype=np.uint8))
            if packed:
                have[0][positions] = vals
            else:
                for pos, v in zip(positions, vals):
                    have[0][pos] = v
            have[1][positions] = bottoms

        if G == 1:
            for (q, positions, nd), (vals, bottoms) in zip(layout[0], mine):
                place(q, positions, vals, bottoms)
            return answers
        words = [sum(len(positions) * (2 + 8 * nd) for _, positions, nd in layout[r]) for r in range(G)]
        width = max(max(words), 1)
        row = np.zeros(width, dtype=np.int64)
        at = 0
        for (q, positions, nd), (vals, bottoms) in zip(layout[g], mine):
            k = len(positions)
            block = np.empty((k, 2 + 8 * nd), dtype=np.int64)
            block[:, :2] = np.ascontiguousarray(vals).view(np.int64) if packed else np.frombuffer(sc.pack(vals), dtype=np.int64).reshape(k, 2)
            if nd:
                block[:, 2:] = np.ascontiguousarray(bottoms).view(np.int64)
            row[at:at + block.size] = block.reshape(-1)
            at += block.size
        assert at == words[g]
        on_device = self.device.type == "cuda" and dist.get_backend(self.group) != "gloo"
        t = torch.from_numpy(row).to(self.device) if on_device else torch.from_numpy(row)
        if on_device:
            out = torch.empty((G, width), dtype=torch.int64, device=self.device)
            dist.all_gather_into_tensor(out, t, group=self.group)
        else:
            parts = [torch.empty_like(t) for _ in range(G)]
            dist.all_gather(parts, t, group=self.group)
            out = torch.stack(parts, dim=0)
        rows = out.cpu().numpy()
        for r in range(G):
            at = 0
            for q, positions, nd in layout[r]:
                k = len(positions)
                block = rows[r, at:at + k * (2 + 8 * nd)].reshape(k, 2 + 8 * nd)
                at += block.size
                vals = np.ascontiguousarray(block[:, :2])
                vals = vals.view(np.uint8) if packed else sc.unpack(vals.tobytes(), k)
                place(q, positions, vals, np.ascontiguousarray(block[:, 2:]).view(np.uint8))
        return answers

    @staticmethod
    def _joined_paths(bottoms, tops):
        """authentication paths (lists of fresh 64-byte objects, merkle.py:16-27) from their two parts, each a uint8 array
        [openings][64 * digests]: the part below the sub-roots (from the owner's local subtree) and the part above (from the
        replicated top tree); one pass over one buffer makes all the objects"""
        import numpy as np
        import starkcore as sc
        k = bottoms.shape[0]
        if tops is not None and tops.shape[0] == k and tops.shape[1]:
            bottoms = np.concatenate((bottoms, tops), axis=1)
        depth = bottoms.shape[1] // 64
        return sc._path_lists(memoryview(np.ascontiguousarray(bottoms)).cast("B"), 0, depth, k)

    # -- layers -----------------------------------------------------------------------------------
    def _commit_sharded(self, slab, C, local=None):
        """local: the rank's subtree over `slab` if the fold that produced the slab has built it already (fold_slab_tree)"""
        eng, G, Rw = self.engine, self.world, self.Rw
        if self.one_rank_local and hasattr(eng, "query_many"):
            # one rank: its "slab" is the whole codeword in natural order and its subtree the whole tree -- ONE tree, no sub-root
            # level copied out, no gather, no second tree above it (a quarter of a world-1 proof's commitments otherwise)
            full = slab.reshape(C * self.R, 2)
            tree = local if local is not None else eng.tree(full)
            return {"kind": "local", "vec": full, "tree": tree, "root": tree.root, "length": C * self.R, "cache": {}}
        if local is None:
            local = eng.tree(slab, need_root=False)
        sub_level = Rw.bit_length() - 1
        sub = eng.level(local, sub_level)                                   # [C][8]: one sub-root per row
        top_leaves = self._all_gather(sub).permute(1, 0, 2).contiguous()    # natural order: node (row, rank)
        top = eng.tree_from_digests(top_leaves.reshape(C * G, 8))
        return {"kind": "sharded", "slab": slab, "C": C, "local": local, "top": top, "root": top.root, "length": C * self.R, "cache": {}}

    def commit(self, slab, C):
        """Merkle.commit (code/merkle.py:13-14) of a codeword held as column slabs [C][R/G]: local subtrees + one all-gather
        of C sub-roots per rank; returns the layer record `_open` answers openings from (its "root" is the commitment)."""
        return self._commit_sharded(slab, C)

    def _natural(self, slab, C):
        """the whole codeword in natural order on every rank: [C][R/G] slabs -> [C*R]"""
        return self._all_gather(slab).permute(1, 0, 2, 3).reshape(C * self.R, 2).contiguous()

    def _open_many_raw(self, requests):
        """[(values, paths)] for a list of (layer, global indices).  ONE library call for everything this rank can answer
        (values + the bottoms of the paths of the columns it owns, from its local subtrees; the tops of all paths from the
        replicated top trees) and ONE collective to merge the owners' answers."""
        eng = self.engine
        R, Rw, G, g = self.R, self.Rw, self.world, self.rank
        sub_level = Rw.bit_length() - 1
        asks, where = [], []
        for q, (layer, indices) in enumerate(requests):
            if layer["kind"] == "local":
                where.append(("local", len(asks)))
                asks.append((layer["tree"], layer["vec"], list(indices)))
                continue
            mine = [i for i in indices if (i % R) // Rw == g] if G > 1 else indices
            where.append(("sharded", len(asks)))
            asks.append((layer["local"], layer["slab"], [(i // R) * Rw + (i % R) % Rw for i in mine], sub_level))
            asks.append((layer["top"], None, [(i // R) * G + (i % R) // Rw for i in indices] if layer["C"] * G > 1 else []))
        got = eng.query_many(asks, raw_paths=True)
        layout, mine, sizes = [[] for _ in range(G)], [], [len(indices) for _, indices in requests]
        for q, ((layer, indices), w) in enumerate(zip(requests, where)):
            if w[0] == "local" or not indices:
                continue
            if G == 1:
                layout[0].append((q, range(len(indices)), sub_level))
                mine.append(got[w[1]])
                continue
            owners = [[] for _ in range(G)]
            for pos, i in enumerate(indices):
                owners[(i % R) // Rw].append(pos)
            for r in range(G):
                if owners[r]:
                    layout[r].append((q, owners[r], sub_level))
            if owners[g]:
                mine.append(got[w[1]])
        answers = self._gather_answers(layout, mine, sizes) if any(layout) else {}
        out = []
        for q, ((layer, indices), w) in enumerate(zip(requests, where)):
            if w[0] == "local":
                vals, paths = got[w[1]]
                out.append((vals, self._joined_paths(paths, None) if layer["length"] > 1 and len(indices) else [[] for _ in indices]))
                continue
            if q not in answers:
                out.append(([], []))
                continue
            vals, bottoms = answers[q]
            out.append((vals, self._joined_paths(bottoms, got[w[1] + 1][1] if layer["C"] * G > 1 else None)))     # below the sub-roots + above them
        return out

    def _open_many_arrays(self, requests):
        """[(packed residues (bytes), paths as a uint8 array [openings][64 * depth])] for a list of (layer, global indices): what
        _open_many_raw gathers, with the two parts of every path joined as arrays and NO object made (proof_objects' segments)"""
        import numpy as np
        import starkcore as sc
        eng = self.engine
        R, Rw, G, g = self.R, self.Rw, self.world, self.rank
        sub_level = Rw.bit_length() - 1
        asks, where = [], []
        for q, (layer, indices) in enumerate(requests):
            if layer["kind"] == "local":
                where.append(("local", len(asks)))
                asks.append((layer["tree"], layer["vec"], list(indices)))
                continue
            mine = [i for i in indices if (i % R) // Rw == g] if G > 1 else indices
            where.append(("sharded", len(asks)))
            asks.append((layer["local"], layer["slab"], [(i // R) * Rw + (i % R) % Rw for i in mine], sub_level))
            asks.append((layer["top"], None, [(i // R) * G + (i % R) // Rw for i in indices] if layer["C"] * G > 1 else []))
        got = eng.query_many(asks, raw_paths=True, raw_values=True)           # residues stay packed bytes from the device to the proof
        layout, mine, sizes = [[] for _ in range(G)], [], [len(indices) for _, indices in requests]
        for q, ((layer, indices), w) in enumerate(zip(requests, where)):
            if w[0] == "local" or not len(indices):
                continue
            if G == 1:
                layout[0].append((q, range(len(indices)), sub_level))
                mine.append(got[w[1]])
                continue
            owners = [[] for _ in range(G)]
            for pos, i in enumerate(indices):
                owners[(i % R) // Rw].append(pos)
            for r in range(G):
                if owners[r]:
                    layout[r].append((q, owners[r], sub_level))
            if owners[g]:
                mine.append(got[w[1]])
        answers = self._gather_answers(layout, mine, sizes, packed=True) if any(layout) else {}
        out = []
        for q, ((layer, indices), w) in enumerate(zip(requests, where)):
            k = len(indices)
            if w[0] == "local":
                vals, paths = got[w[1]]
                depth = layer["length"].bit_length() - 1
                paths = np.ascontiguousarray(paths).reshape(k, 64 * depth) if k and depth else np.zeros((k, 0), dtype=np.uint8)
            elif q not in answers:
                vals, paths = np.zeros((0, 16), dtype=np.uint8), np.zeros((0, 0), dtype=np.uint8)
            else:
                vals, bottoms = answers[q]
                tops = got[w[1] + 1][1] if layer["C"] * G > 1 else None
                paths = np.concatenate((bottoms, tops), axis=1) if tops is not None and tops.shape[0] == k and tops.shape[1] else bottoms
            out.append((np.ascontiguousarray(vals).tobytes(), np.ascontiguousarray(paths)))
        return out

    @staticmethod
    def _holder(layer, field):
        """the layer's entries as proof_objects' segments want them: a field, an identity, FieldElements made once per index"""
        holder = layer.get("holder")
        if holder is None:
            holder = layer["holder"] = _LayerEntries(layer, field)
        return holder

    def _open_many(self, requests):
        """entries as FieldElement objects (one object per index and layer, reused) + fresh path objects per request"""
        from algebra import FieldElement
        res, field, new = [], self.fri.field, object.__new__
        for (layer, indices), (values, paths) in zip(requests, self._open_many_raw(requests)):
            cache = layer["cache"]
            for i, v in zip(indices, values):
                if i not in cache:
                    # FieldElement(v, field) without the call into __init__ (algebra.py:16-18 sets exactly these two attributes),
                    # as in starkcore.DeviceCodeword._entries
                    e = new(FieldElement)
                    e.value = v
                    e.field = field
                    cache[i] = e
            res.append(([cache[i] for i in indices], paths))
        return res

    def _open(self, layer, indices):
        return self._open_many([(layer, indices)])[0]

    # -- the protocol -----------------------------------------------------------------------------
    def prove(self, slab, proof_stream, also_open=None):
        """also_open (a fri.AlsoOpen whose `requests` returns (layer records, index lists)): further committed layers opened in the
        same library call and collective as the query phase"""
        from algebra import FieldElement
        fr, eng, field = self.fri, self.engine, self.fri.field
        N, R, Rw = fr.domain_length, self.R, self.Rw
        C = N // R
        assert tuple(slab.shape) == (C, Rw, 2), "slab must be this rank's [C][R/G] columns"
        omega, offset, rounds = fr.omega, fr.offset, fr.num_rounds()
        layers, cur, full, local = [], slab, None, None
        if self.one_rank_local:
            top = self._prove_one_rank(slab, proof_stream, also_open)
            if top is not None:
                return top
        # fri.py:68 in every round: omega_r^(N_r) == 1 with omega_r = omega^(2^r), N_r = N / 2^r -- one condition, checked once
        if hasattr(fr, "_check_omega_order"):
            fr._check_omega_order(N)                         # (once per Fri instance: a power and an inversion in Python integers)
        else:
            assert(omega ^ (N - 1) == omega.inverse()), "error in commit: omega does not have the right order!"
        for r in range(rounds):
            Nr = N >> r
            if full is None and (C == 1 or Nr <= self.local_tail or (self.one_rank_local and hasattr(eng, "commit_rounds"))):      # (one rank: nothing to gather, the slab IS the codeword)
                full = self._natural(cur, C)                # one row left / a short codeword: collect it everywhere, go local
                rest = self._commit_tail(full, Nr, offset, omega, rounds - r, proof_stream)
                if rest is not None:                        # ... and the library ran every remaining round in one call
                    layers.extend(rest)
                    break
            if full is None:
                layer = self._commit_sharded(cur, C, local)
            else:
                tree = eng.tree(full)
                layer = {"kind": "local", "vec": full, "tree": tree, "root": tree.root, "length": Nr, "cache": {}}
            layers.append(layer)
            proof_stream.push(layer["root"])
            if r == rounds - 1:
                break
            alpha = field.sample(proof_stream.prover_fiat_shamir())
            if full is None:
                # the folded slab is committed to as a slab again (not gathered): fold + local subtree in one call
                if C > 2 and (Nr >> 1) > self.local_tail and hasattr(eng, "fold_slab_tree"):
                    cur, local = eng.fold_slab_tree(cur, C, Rw, R, self.rank * Rw, alpha.value, offset.value, omega.value)
                else:
                    cur, local = eng.fold_slab(cur, C, Rw, R, self.rank * Rw, alpha.value, offset.value, omega.value), None
                C //= 2
            else:
                full = eng.fold_full(full, Nr, alpha.value, offset.value, omega.value)
            omega = omega ^ 2
            offset = offset ^ 2
        # last codeword in the clear (fri.py:91): natural order, plain list; its objects are reused by the last query round
        last_layer = layers[-1]
        last_vec = last_layer["vec"] if last_layer["kind"] == "local" else self._natural(cur, C)
        last_values = eng.read(last_vec, range(last_layer["length"]))
        lazy = None
        if hasattr(eng, "query_many"):                      # (the CPU test engines answer with objects)
            import proof_objects as _po
            lazy = _po.lazy_objects(proof_stream)
        if lazy is not None:
            # described, not built (proof_objects): the transcript bytes are the same, no object per element / digest
            import starkcore as _scm
            lazy.add(_po.ElementList(self._holder(last_layer, field), _scm.pack(last_values)))
            return self._query_all_lazy(layers, len(last_values), proof_stream, lazy, also_open)
        last_list = [FieldElement(v, field) for v in last_values]
        last_layer["cache"] = dict(enumerate(last_list))
        proof_stream.push(last_list)

        return self._query_all(layers, last_list, proof_stream)

    def _prove_one_rank(self, slab, proof_stream, also_open):
        """ONE rank on the HIP engine: the slab is the codeword in natural order, so Fri.prove's one-call form (sc_fri_prove_dev:
        commit phase, index sampling and every opening -- the caller's committed layers included -- in one library call) serves
        it as it serves a single GPU's prover; the slab's memory is handed over as it is (DeviceVector.wrap).  None when a
        precondition of that form does not hold (the rounds then run as on any number of ranks)."""
        eng = self.engine
        if not isinstance(eng, HipFriEngine) or not slab.is_cuda or not slab.is_contiguous():
            return None
        import starkcore as sc
        from fri import AlsoOpen
        if _current_raw_stream(self.device) != sc.library_stream():
            return None                                    # (the one-call form runs on the library's stream)
        N = self.fri.domain_length
        inner = None
        if also_open is not None:
            more, shift = getattr(also_open, "layers", None), getattr(also_open, "shift", None)
            if more is None or shift is None or not all(layer["kind"] == "local" and isinstance(layer["tree"], HipFriEngine._Tree) and layer["length"] == N for layer in more):
                return None
            codewords = []
            for layer in more:
                cw = layer.get("codeword")
                if cw is None:
                    vec = layer["vec"]
                    cw = layer["codeword"] = sc.DeviceCodeword(sc.DeviceVector.wrap(vec.data_ptr(), N, vec), self.fri.field)
                    cw._tree = layer["tree"].tree
                codewords.append(cw)
            inner = AlsoOpen(None, codewords=codewords, shift=shift)
        codeword = sc.DeviceCodeword(sc.DeviceVector.wrap(slab.data_ptr(), N, slab), self.fri.field)
        top = self.fri._prove_in_library(codeword, proof_stream, inner)
        if top is not None and also_open is not None:
            also_open.answers = inner.answers
            also_open.position_arrays = inner.position_arrays
        return top

    def _commit_tail(self, full, Nr, offset, omega, rounds_left, proof_stream):
        """the remaining rounds of the commit phase on the gathered codeword through the engine's whole-loop call, when there is
        one and the proof stream qualifies (fri.library_transcript: what Fri.commit checks on one GPU); layer records or None"""
        eng = self.engine
        if not hasattr(eng, "commit_rounds") or Nr < 2:
            return None
        from fri import library_transcript
        prior = library_transcript(proof_stream, rounds_left)
        if prior is None:
            return None
        got = eng.commit_rounds(full, Nr, offset.value, omega.value, rounds_left, prior)
        if got is None:
            return None
        rest = []
        for k, (vec, tree, root) in enumerate(got):
            proof_stream.push(root)
            rest.append({"kind": "local", "vec": vec, "tree": tree, "root": root, "length": Nr >> k, "cache": {}})
        return rest

    def _query_requests(self, layers, last_length, proof_stream):
        """top-level indices from the transcript and what every layer has to open (fri.py:119-128)"""
        fr = self.fri
        N, s = fr.domain_length, fr.num_colinearity_tests
        top_level_indices = fr.sample_indices(proof_stream.prover_fiat_shamir(), N // 2, last_length, s)
        nq = len(layers) - 1
        per_round, indices = [], [i for i in top_level_indices]
        for i in range(nq):
            indices = [index % (layers[i]["length"] // 2) for index in indices]
            per_round.append(indices)
        requests = []
        for j, layer in enumerate(layers):
            request = []
            if j < nq:
                request += per_round[j][:s] + [index + layer["length"] // 2 for index in per_round[j][:s]]
            if j > 0:
                request += per_round[j - 1][:s]
            requests.append((layer, request))
        return top_level_indices, per_round, requests

    def _query_all_lazy(self, layers, last_length, proof_stream, lazy, also_open=None):
        """_query_all with the owners' answers pushed as they are (proof_objects.FriRound)"""
        import proof_objects as _po
        field, s = self.fri.field, self.fri.num_colinearity_tests
        top_level_indices, per_round, requests = self._query_requests(layers, last_length, proof_stream)
        if also_open is not None:
            more_layers, more_indices = also_open.requests(top_level_indices)
            requests = requests + list(zip(more_layers, more_indices))
        fetched = self._open_many_arrays(requests)
        if also_open is not None:
            also_open.answers = fetched[len(layers):]
        nq = len(layers) - 1
        for i in range(nq):
            values, paths = fetched[i]
            next_values, next_paths = fetched[i + 1]
            c_at = 2 * s if i + 1 < nq else 0
            a = per_round[i][:s]
            half = layers[i]["length"] // 2
            lazy.add(_po.FriRound(self._holder(layers[i], field), self._holder(layers[i + 1], field), a, [index + half for index in a], a,
                                  values[:16 * s], values[16 * s:32 * s], next_values[16 * c_at:16 * (c_at + s)],
                                  paths[:s], paths[s:2 * s], next_paths[c_at:c_at + s]))
        return top_level_indices

    def _query_all(self, layers, last_list, proof_stream):
        """the query phase of fri.py:124-128 over the committed layers: indices from the transcript, ONE collective for every
        opening of every round, pushes in the reference's order"""
        fr = self.fri
        N = fr.domain_length
        s = fr.num_colinearity_tests
        top_level_indices = fr.sample_indices(proof_stream.prover_fiat_shamir(), N // 2, len(last_list), s)
        nq = len(layers) - 1
        per_round, indices = [], [i for i in top_level_indices]
        for i in range(nq):
            indices = [index % (layers[i]["length"] // 2) for index in indices]
            per_round.append(indices)
        requests = []
        for j, layer in enumerate(layers):
            request = []
            if j < nq:
                request += per_round[j][:s] + [index + layer["length"] // 2 for index in per_round[j][:s]]
            if j > 0:
                request += per_round[j - 1][:s]
            requests.append((layer, request))
        fetched = self._open_many(requests)                 # one collective for the whole query phase
        # pushes in the reference's order (fri.py:104-113 per round: s triples, then per test the paths of a, b, c); a ProofStream's
        # `push` is `objects.append`, so a whole round goes in with two list extensions (as in fri.Fri._query_all)
        from ip import ProofStream
        objects = proof_stream.objects if type(proof_stream) is ProofStream else None
        for i in range(nq):
            entries, paths = fetched[i]
            next_entries, next_paths = fetched[i + 1]
            c_at = 2 * s if i + 1 < nq else 0
            triples = list(zip(entries[:s], entries[s:2 * s], next_entries[c_at:c_at + s]))
            openings = [p for trio in zip(paths[:s], paths[s:2 * s], next_paths[c_at:c_at + s]) for p in trio]
            if objects is not None:
                objects.extend(triples)
                objects.extend(openings)
            else:
                for obj in triples + openings:
                    proof_stream.push(obj)
        return top_level_indices


class ContiguousFri(ShardedFri):
    """`Fri.prove` (reference code/fri.py:115-130) on a codeword in the NATURAL contiguous layout (SURVEY.md 8(e), row "FRI
    fold"): rank g owns x[g*N/G : (g+1)*N/G] -- a single-GPU LDE cut into G pieces, or a host list scattered in order.
      * split-and-fold pairs i with i + N/2, i.e. rank g with rank g + G/2: ONE neighbour exchange per fold.  The upper rank
        ships its slab to its partner, which folds both; the folded codeword (half as long) lives contiguously on the lower
        half of the ranks, and so on until one rank holds what is left;
      * Merkle leaves are contiguous: a commit is the active ranks' local subtrees plus one all-gather of their sub-roots
        (64 bytes per rank); the levels above are rebuilt, identically, on every rank;
      * every rank -- also one that has run out of data -- follows the same transcript, so alphas and query indices agree without
        a broadcast; an opening is answered by the rank that owns the leaf, all of them merged by one collective.
    ShardedFri (column slabs: no element exchange at all) is the better layout for a codeword that comes out of ShardedNtt; this
    class serves codewords that arrive in natural order without re-slabbing them (rows_to_column_slab is the other option)."""

    def __init__(self, fri, rank, world, device, engine=None, group=None):
        self.fri, self.rank, self.world, self.device, self.group = fri, rank, world, device, group
        assert world & (world - 1) == 0 and fri.domain_length % world == 0 and fri.domain_length // world >= 1
        self.engine = engine if engine is not None else HipFriEngine(device)
        self.elements_shipped = 0

    def _ship(self, t, src, dst, count):
        """the neighbour exchange: `count` elements from rank src to rank dst (returns the received tensor on dst)"""
        staged = t is not None and t.is_cuda and dist.get_backend(self.group) == "gloo"     # functional tests: host-staged
        if self.rank == src:
            dist.send(t.cpu() if staged else t.contiguous(), dst, group=self.group)
            self.elements_shipped += count
            return None
        buf = torch.empty((count, 2), dtype=torch.int64, device="cpu" if (self.device.type == "cuda" and dist.get_backend(self.group) == "gloo") else self.device)
        dist.recv(buf, src, group=self.group)
        return buf.to(self.device)

    def _commit_contiguous(self, cur, length, active):
        eng = self.engine
        seg = length // active
        mine = self.rank < active
        local = eng.tree(cur, need_root=False) if mine else None
        sub = eng.level(local, seg.bit_length() - 1) if mine else torch.zeros((1, 8), dtype=torch.int64, device=self.device)
        top = eng.tree_from_digests(self._all_gather(sub)[:active].reshape(active, 8))
        return {"kind": "contiguous", "vec": cur, "local": local, "top": top, "root": top.root, "seg": seg, "active": active, "length": length, "cache": {}}

    def commit(self, slab, length, active=None):
        """Merkle.commit (code/merkle.py:13-14) of a codeword of `length` held contiguously by the first `active` ranks"""
        return self._commit_contiguous(slab, length, self.world if active is None else active)

    def _open_many_raw(self, requests):
        eng, g = self.engine, self.rank
        asks = []
        for layer, indices in requests:
            seg = layer["seg"]
            mine = [i % seg for i in indices if i // seg == g]
            asks.append((layer["local"], layer["vec"], mine, seg.bit_length() - 1) if mine else (None, None, []))
            asks.append((layer["top"], None, [i // seg for i in indices] if layer["active"] > 1 else []))
        got = eng.query_many(asks, raw_paths=True)
        layout, mine, sizes = [[] for _ in range(self.world)], [], [len(indices) for _, indices in requests]
        for q, (layer, indices) in enumerate(requests):
            seg = layer["seg"]
            owners = [[] for _ in range(self.world)]
            for pos, i in enumerate(indices):
                owners[i // seg].append(pos)
            for r in range(self.world):
                if owners[r]:
                    layout[r].append((q, owners[r], seg.bit_length() - 1))
            if owners[g]:
                mine.append(got[2 * q])
        answers = self._gather_answers(layout, mine, sizes)
        out = []
        for q, (layer, indices) in enumerate(requests):
            if q not in answers:
                out.append(([], []))
                continue
            vals, bottoms = answers[q]
            out.append((vals, self._joined_paths(bottoms, got[2 * q + 1][1] if layer["active"] > 1 else None)))
        return out

    def prove(self, slab, proof_stream):
        from algebra import FieldElement
        fr, eng, field, G = self.fri, self.engine, self.fri.field, self.world
        N = fr.domain_length
        assert tuple(slab.shape) == (N // G, 2), "slab must be this rank's N/G consecutive elements"
        omega, offset, rounds = fr.omega, fr.offset, fr.num_rounds()
        layers, cur, active = [], slab, G
        assert(omega ^ (N - 1) == omega.inverse()), "error in commit: omega does not have the right order!"     # every round's fri.py:68
        for r in range(rounds):
            Nr = N >> r
            layer = self._commit_contiguous(cur, Nr, active)
            layers.append(layer)
            proof_stream.push(layer["root"])
            if r == rounds - 1:
                break
            alpha = field.sample(proof_stream.prover_fiat_shamir())
            seg = Nr // active
            if active == 1:
                if self.rank == 0:
                    cur = eng.fold_full(cur, Nr, alpha.value, offset.value, omega.value)
            else:
                half = active // 2
                if self.rank < half:
                    upper = self._ship(None, self.rank + half, self.rank, seg)
                    pair = torch.cat([cur, upper], dim=0)
                    # the rank's outputs are i in [rank*seg, (rank+1)*seg): a fold of length 2*seg whose domain starts at omega^(rank*seg)
                    shifted = (offset * (omega ^ (self.rank * seg))).value
                    cur = eng.fold_full(pair, 2 * seg, alpha.value, shifted, omega.value)
                elif self.rank < active:
                    self._ship(cur, self.rank, self.rank - half, seg)
                    cur = None
                active = half
            omega = omega ^ 2
            offset = offset ^ 2
        # last codeword in the clear (fri.py:91): natural order, plain list; its objects are reused by the last query round
        last_layer = layers[-1]
        seg = last_layer["seg"]
        part = cur if self.rank < active else torch.zeros((seg, 2), dtype=torch.int64, device=self.device)
        last_vec = self._all_gather(part)[:active].reshape(last_layer["length"], 2)
        last_list = [FieldElement(v, field) for v in eng.read(last_vec, range(last_layer["length"]))]
        last_layer["cache"] = dict(enumerate(last_list))
        proof_stream.push(last_list)
        return self._query_all(layers, last_list, proof_stream)


# =====================================================================================================================
# Independent columns: one register per GPU
# =====================================================================================================================
class ColumnReplicas:
    """SURVEY.md 8(e), last row: the registers of a STARK (fast_stark.py:103-105, :113: one `fast_coset_evaluate` + one
    `Merkle.commit` per trace / quotient column) are independent units.  When the domain is too small to be worth sharding,
    column i goes to rank i % world: no element ever crosses a link, only the 64-byte roots are all-gathered (every rank
    needs all of them, in column order, for the Fiat-Shamir transcript)."""

    def __init__(self, rank, world, device, engine=None, group=None):
        self.rank, self.world, self.device, self.group = rank, world, device, group
        self.engine = engine if engine is not None else HipFriEngine(device)

    def lde_and_commit(self, columns, offset, generator, order):
        """columns: list of packed coefficient lists (bytes), identical on every rank.
        Returns (mine, roots): mine = {column index: (codeword tensor, tree)} for this rank's columns; roots = the Merkle
        roots of ALL columns in column order."""
        mine = {}
        for i, coeffs in enumerate(columns):
            if i % self.world == self.rank:
                codeword = self.engine.lde(coeffs, offset, generator, order)
                mine[i] = (codeword, self.engine.tree(codeword))
        # every rank needs all the roots, in column order: a [columns][64] byte table in which each rank fills the rows of its own
        # columns, summed over the ranks (the rows are disjoint) -- one fixed-shape tensor collective, nothing pickled
        import numpy as np
        table = np.zeros((len(columns), 64), dtype=np.int32)
        for i, (_, t) in mine.items():
            table[i] = np.frombuffer(t.root, dtype=np.uint8)
        if self.world > 1:
            on_dev = self.device.type == "cuda" and dist.get_backend(self.group) != "gloo"
            tt = torch.from_numpy(table).to(self.device) if on_dev else torch.from_numpy(table)
            dist.all_reduce(tt, op=dist.ReduceOp.SUM, group=self.group)
            table = tt.cpu().numpy()
        return mine, [bytes(table[i].astype(np.uint8)) for i in range(len(columns))]
