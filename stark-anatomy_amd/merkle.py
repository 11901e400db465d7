"""BLAKE2b-512 Merkle commitments -- host shim (interface of reference code/merkle.py:3-43).

`commit` / `open` on field-element arrays run on the GPU (leaf = H(decimal ASCII of the residue),
node = H(left || right)); the tree stays resident in HBM, so an opening is a gather of log2 N digests
instead of the reference's full rebuild per call.  The raw-digest variants (`commit_`, `open_`,
`verify_`) and `verify` are tiny and stay on the host with hashlib.
"""
from hashlib import blake2b

from starkcore import DeviceCodeword, MerkleTree


class Merkle:
    H = blake2b

    def commit_(leafs):
        assert(len(leafs) & (len(leafs) - 1) == 0), "length must be power of two"
        level = list(leafs)
        while len(level) > 1:
            level = [Merkle.H(level[i] + level[i + 1]).digest() for i in range(0, len(level), 2)]
        return level[0]

    def _tree(data_array):
        if isinstance(data_array, DeviceCodeword):
            return data_array.tree()
        # leaves travel as 16-byte residues; a field wider than that has no device leaf format (and no CPU fallback)
        assert(len(data_array) == 0 or data_array[0].field.p <= (1 << 128)), "the MI355X Merkle kernels take residues below 2^128 only"
        return MerkleTree.from_bytes(b"".join(da.value.to_bytes(16, "little") for da in data_array))

    def commit(data_array):
        assert(len(data_array) & (len(data_array) - 1) == 0), "length must be power of two"
        return Merkle._tree(data_array).root

    def open_(index, leafs):
        assert(len(leafs) & (len(leafs) - 1) == 0), "length must be power of two"
        assert(0 <= index and index < len(leafs)), "cannot open invalid index"
        level, path = list(leafs), []
        while len(level) > 1:
            path.append(level[index ^ 1])
            level = [Merkle.H(level[i] + level[i + 1]).digest() for i in range(0, len(level), 2)]
            index >>= 1
        return path

    def open(index, data_array):
        assert(len(data_array) & (len(data_array) - 1) == 0), "length must be power of two"
        assert(0 <= index and index < len(data_array)), "cannot open invalid index"
        return Merkle._tree(data_array).open(index)

    def verify_(root, index, path, leaf):
        assert(0 <= index and index < (1 << len(path))), "cannot verify invalid index"
        node = leaf
        for sibling in path:
            node = Merkle.H(node + sibling).digest() if index % 2 == 0 else Merkle.H(sibling + node).digest()
            index >>= 1
        return root == node

    def verify(root, index, path, data_element):
        return Merkle.verify_(root, index, path, Merkle.H(bytes(data_element)).digest())
