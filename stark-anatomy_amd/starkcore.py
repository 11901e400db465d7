"""ctypes binding of libstarkcore.so (include/starkcore.h) plus the device-resident codeword type.

This is the ONLY path from the host modules (ntt.py, fri.py, merkle.py) to the GPU.  There is no CPU
fallback: if the shared library is missing or no MI355X is visible, calls raise RuntimeError.
"""
import array
import ctypes
import itertools
import os
from collections.abc import Sequence

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("STARKCORE_LIB") or os.path.join(_HERE, "libstarkcore.so")   # override only for A/B experiments

_u64 = ctypes.c_uint64
_vp = ctypes.c_void_p
_int = ctypes.c_int

SC_ERR_NOT_POW2 = -2
SC_ERR_ROOT_ORDER = -3
SC_ERR_ROOT_NOT_PRIMITIVE = -4
SC_ERR_DIV_ZERO = -5
SC_ERR_UNSUPPORTED = -7
SC_ERR_TIMEOUT = -9

# every symbol include/starkcore.h declares: (restype, argtypes)
SIGNATURES = {
    "sc_device_count": (_int, []),
    "sc_init": (_int, [_int]),
    "sc_shutdown": (_int, []),
    "sc_last_error": (ctypes.c_char_p, []),
    "sc_synchronize": (_int, []),
    "sc_stream": (_int, [ctypes.POINTER(_vp)]),
    "sc_stream_join": (_int, [_vp]),
    "sc_set_tuning": (_int, [ctypes.c_char_p, _int]),
    "sc_ntt_num_passes": (_int, [_u64]),
    "sc_debug_trace": (_int, [_vp]),
    "sc_field_selftest": (_int, [_int, _vp, _vp, _vp, _u64]),
    "sc_vec_alloc": (_int, [_u64, ctypes.POINTER(_vp)]),
    "sc_vec_free": (_int, [_vp]),
    "sc_vec_wrap": (_int, [_vp, _u64, ctypes.POINTER(_vp)]),
    "sc_vec_len": (_u64, [_vp]),
    "sc_vec_ptr": (_vp, [_vp]),
    "sc_vec_zero": (_int, [_vp]),
    "sc_vec_upload": (_int, [_vp, _u64, _vp, _u64]),
    "sc_vec_download": (_int, [_vp, _u64, _vp, _u64]),
    "sc_vec_gather": (_int, [_vp, _vp, _u64, _vp]),
    "sc_memcpy_dev": (_int, [_vp, _vp, _u64, _vp]),
    "sc_sample_bytes_dev": (_int, [_vp, _u64, ctypes.c_uint32, _vp, _vp]),
    "sc_sample_urandom_dev": (_int, [_u64, ctypes.c_uint32, _vp, _vp]),
    "sc_urandom_prefetch": (_int, [_u64, ctypes.c_uint32]),
    "sc_ntt": (_int, [_vp, _vp, _u64, _vp, _int]),
    "sc_ntt_dev": (_int, [_vp, _vp, _u64, _vp, _int, _vp]),
    "sc_ntt_columns_dev": (_int, [_vp, _vp, _u64, _u64, _vp, _int, _vp]),
    "sc_ntt_batch_dev": (_int, [_vp, _vp, _u64, _u64, _int, _vp, _vp]),
    "sc_ntt_batch_ex_dev": (_int, [_vp, _vp, _u64, _u64, _int, _vp, _vp, _u64, _u64, _int, _u64, _vp]),
    "sc_ntt_rows_t_ld_dev": (_int, [_vp, _vp, _u64, _u64, _vp, _u64, _u64, _vp]),
    "sc_fourstep_create": (_int, [_int, _vp, _int, _int, ctypes.POINTER(_vp)]),
    "sc_fourstep_create_ex": (_int, [_int, _vp, _int, _int, _int, ctypes.POINTER(_vp)]),
    "sc_ipc_region_create": (_int, [_u64, ctypes.POINTER(_vp), _vp]),
    "sc_ipc_region_create_ex": (_int, [_u64, _int, ctypes.POINTER(_vp), _vp]),
    "sc_ipc_region_open": (_int, [_vp, ctypes.POINTER(_vp)]),
    "sc_ipc_region_kind": (_int, [ctypes.POINTER(_int)]),
    "sc_ipc_region_close": (_int, [_vp]),
    "sc_ipc_region_free": (_int, [_vp]),
    "sc_fourstep_region_bytes": (_int, [_vp, ctypes.POINTER(_u64)]),
    "sc_fourstep_set_peers": (_int, [_vp, _vp]),
    "sc_fourstep_run_direct_dev": (_int, [_vp, _int, _vp, _vp, _vp]),
    "sc_fourstep_direct_status": (_int, [_vp, ctypes.POINTER(_u64)]),
    "sc_fourstep_free": (_int, [_vp]),
    "sc_fourstep_shape": (_int, [_vp, _int, ctypes.POINTER(_u64), ctypes.POINTER(_u64)]),
    "sc_fourstep_cols_dev": (_int, [_vp, _int, _vp, _vp, _vp, _vp]),
    "sc_fourstep_rows_dev": (_int, [_vp, _int, _vp, _vp, _u64, _u64, _int, _vp]),
    "sc_fourstep_rows_finish_dev": (_int, [_vp, _int, _vp, _vp]),
    "sc_comm_unique_id": (_int, [ctypes.c_char_p, _vp]),
    "sc_comm_init": (_int, [ctypes.c_char_p, _vp, _int, _int]),
    "sc_comm_destroy": (_int, []),
    "sc_fourstep_run_dev": (_int, [_vp, _int, _vp, _vp, _vp, _vp, _u64, _int, _int, _vp]),
    "sc_twiddle_matrix_dev": (_int, [_vp, _u64, _u64, _u64, _u64, _vp, _u64, _vp, _vp]),
    "sc_coset_evaluate": (_int, [_vp, _u64, _vp, _vp, _u64, _vp]),
    "sc_coset_evaluate_dev": (_int, [_vp, _u64, _vp, _vp, _u64, _vp, _vp]),
    "sc_coset_evaluate_columns_dev": (_int, [_vp, _u64, _u64, _vp, _vp, _u64, _vp, _vp]),
    "sc_poly_mul": (_int, [_vp, _u64, _vp, _u64, _vp, _u64, _vp, _u64]),
    "sc_coset_divide": (_int, [_vp, _u64, _vp, _u64, _vp, _vp, _u64, _vp, _u64]),
    "sc_coset_divide_dev": (_int, [_vp, _u64, _vp, _u64, _vp, _vp, _u64, _vp, _u64, ctypes.POINTER(_int), _vp]),
    "sc_vec_degree_dev": (_int, [_vp, _u64, ctypes.POINTER(ctypes.c_int64), _vp]),
    "sc_pointwise_mul_dev": (_int, [_vp, _vp, _vp, _u64, _vp]),
    "sc_pointwise_div_dev": (_int, [_vp, _vp, _vp, _u64, _vp]),
    "sc_coset_divide_later_dev": (_int, [_vp, _u64, _vp, _u64, _vp, _vp, _u64, _vp, _u64, ctypes.POINTER(_vp), _vp]),
    "sc_pointwise_div_later_dev": (_int, [_vp, _vp, _vp, _u64, ctypes.POINTER(_vp), _vp]),
    "sc_later_wait": (_int, [_vp, ctypes.POINTER(ctypes.c_int64)]),
    "sc_scale_dev": (_int, [_vp, _vp, _u64, _vp, _vp]),
    "sc_axpy_shift_dev": (_int, [_vp, _u64, _vp, _u64, _u64, _vp, _vp]),
    "sc_scale_slab_dev": (_int, [_vp, _vp, _u64, _u64, _u64, _u64, _vp, _vp]),
    "sc_fri_fold": (_int, [_vp, _u64, _vp, _vp, _vp, _vp]),
    "sc_fri_fold_dev": (_int, [_vp, _u64, _vp, _vp, _vp, _vp, _vp]),
    "sc_merkle_commit": (_int, [_vp, _u64, _vp]),
    "sc_merkle_build": (_int, [_vp, _u64, _vp, ctypes.POINTER(_vp)]),
    "sc_merkle_build_dev": (_int, [_vp, _u64, _vp, ctypes.POINTER(_vp), _vp]),
    "sc_merkle_build_async_dev": (_int, [_vp, _u64, ctypes.POINTER(_vp), _vp]),
    "sc_merkle_build_noroot_dev": (_int, [_vp, _u64, ctypes.POINTER(_vp), _vp]),
    "sc_merkle_root": (_int, [_vp, _vp]),
    "sc_fri_fold_commit_dev": (_int, [_vp, _u64, _vp, _vp, _vp, _vp, ctypes.POINTER(_vp), _vp]),
    "sc_fri_commit_dev": (_int, [_vp, _u64, _vp, _vp, ctypes.c_uint32, _vp, _vp, _u64, _vp, _vp, _vp, _vp, _vp]),
    "sc_fri_prove_dev": (_int, [_vp, _u64, _vp, _vp, ctypes.c_uint32, ctypes.c_uint32, _vp, _vp, _u64, _u64, _vp, _vp, _u64, _vp, _vp, _vp, _vp,
                                _vp, _vp, _vp, _vp, _u64, _vp]),
    "sc_fri_tail_stats": (_int, [_vp]),
    "sc_host_alloc": (_int, [_u64, _vp]),
    "sc_host_free": (_int, [_vp]),
    "sc_shake256": (_int, [_vp, _u64, _vp, _u64]),
    "sc_pickle_proof": (_int, [_vp, _u64, _vp, ctypes.c_uint32, ctypes.c_uint32, _vp, _u64, ctypes.POINTER(_u64)]),
    "sc_field_sample": (_int, [_vp, _u64, _vp]),
    "sc_blake2b": (_int, [_vp, _u64, _vp]),
    "sc_transcript_challenge": (_int, [_vp, _vp, _u64, _vp, _vp, _u64]),
    "sc_fri_sample_indices": (_int, [_vp, _u64, _u64, _u64, ctypes.c_uint32, _vp]),
    "sc_transcript_bytes": (_int, [_vp, _vp, _u64, _vp, _u64, ctypes.POINTER(_u64)]),
    "sc_merkle_open": (_int, [_vp, _u64, _vp]),
    "sc_merkle_open_batch": (_int, [_vp, _vp, _u64, _vp]),
    "sc_merkle_query_dev": (_int, [_vp, _vp, _vp, _u64, _vp, _vp]),
    "sc_merkle_query_multi_dev": (_int, [_u64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sc_merkle_level_copy_dev": (_int, [_vp, _int, _vp, _vp]),
    "sc_merkle_from_digests_dev": (_int, [_vp, _u64, _vp, ctypes.POINTER(_vp), _vp]),
    "sc_fri_fold_slab_dev": (_int, [_vp, _u64, _u64, _u64, _u64, _vp, _vp, _vp, _vp, _vp]),
    "sc_fri_fold_slab_build_dev": (_int, [_vp, _u64, _u64, _u64, _u64, _vp, _vp, _vp, _vp, ctypes.POINTER(_vp), _vp]),
    "sc_merkle_leaves": (_u64, [_vp]),
    "sc_merkle_free": (_int, [_vp]),
    "sc_mpoly_eval_dev": (_int, [_vp, _u64, _u64, _vp, _vp, _u64, _vp, _vp]),
    "sc_mpoly_eval_ex_dev": (_int, [_vp, _u64, _u64, _vp, _vp, _u64, _vp, _int, _vp]),
    "sc_mpoly_eval_rot_dev": (_int, [_vp, _u64, _u64, _vp, _vp, _u64, _vp, _int, _vp, _vp, _vp]),
    "sc_zerofier": (_int, [_vp, _u64, _vp]),
    "sc_evaluate": (_int, [_vp, _u64, _vp, _u64, _vp]),
    "sc_interpolate": (_int, [_vp, _vp, _u64, _vp]),
    "sc_polytree_build": (_int, [_vp, _u64, ctypes.POINTER(_vp)]),
    "sc_polytree_build_dev": (_int, [_vp, _u64, ctypes.POINTER(_vp), _vp]),
    "sc_polytree_points": (_u64, [_vp]),
    "sc_polytree_zerofier_dev": (_int, [_vp, _vp, _vp]),
    "sc_polytree_evaluate_dev": (_int, [_vp, _vp, _u64, _vp, _vp, _vp]),
    "sc_polytree_interpolate_dev": (_int, [_vp, _vp, _vp, _vp]),
    "sc_polytree_free": (_int, [_vp]),
    "sc_geodomain_create": (_int, [_vp, _vp, _u64, ctypes.POINTER(_vp), _vp]),
    "sc_geodomain_points": (_u64, [_vp]),
    "sc_geodomain_detect_dev": (_int, [_vp, _u64, _vp, _vp, ctypes.POINTER(_int), _vp]),
    "sc_geodomain_zerofier_dev": (_int, [_vp, _vp, _vp]),
    "sc_geodomain_evaluate_dev": (_int, [_vp, _vp, _u64, _vp, _vp]),
    "sc_geodomain_interpolate_dev": (_int, [_vp, _vp, _vp, _vp]),
    "sc_geodomain_free": (_int, [_vp]),
}

_lib = None


def _one_hip_runtime():
    """torch wheels carry their own copy of the HIP runtime (same soname as /opt/rocm's).  Two copies in one process do not share
    the device: if libstarkcore.so pulled in the system copy first, a later `import torch` would find "No HIP GPUs".  When torch is
    installed but not imported yet, its copy of libamdhip64 is therefore loaded (by path, RTLD_GLOBAL) before the library, so
    that libstarkcore's DT_NEEDED binds to it by soname and a later `import torch` (sharded.py and bench.py use torch for memory,
    streams and collectives in the same process) finds the same runtime -- without importing torch here: a pure C-ABI user pays
    milliseconds, not the seconds and side effects of the import.  Only if that preload fails torch itself is imported; a failure
    of both is reported, not hidden.  STARKCORE_NO_TORCH=1 skips all of it (a process that never touches torch)."""
    import importlib.util
    import sys
    import warnings
    if "torch" in sys.modules or os.environ.get("STARKCORE_NO_TORCH") == "1":
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    runtime = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    try:
        if os.path.exists(runtime):
            ctypes.CDLL(runtime, mode=ctypes.RTLD_GLOBAL)
            return
    except OSError as e:
        warnings.warn("starkcore: could not preload torch's HIP runtime (%s); importing torch instead" % e)
    try:
        import torch  # noqa: F401
    except Exception as e:      # noqa: BLE001  a broken torch install must not take the library down with it -- but say so
        warnings.warn("starkcore: torch is installed but neither its HIP runtime nor torch itself could be loaded (%r); a later "
                      "`import torch` in this process may not see the GPU (set STARKCORE_NO_TORCH=1 to silence)" % (e,))


def lib():
    """The loaded library (loads on first use; raises if the HIP extension was not built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError("libstarkcore.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "or `make -C stark-anatomy_amd/csrc` (the HIP extension is mandatory, there is no CPU fallback)")
        _one_hip_runtime()
        l = ctypes.CDLL(_LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


class StarkCoreError(RuntimeError):
    pass


def _check(rc):
    """Map a negative return code to the exception the reference's `assert` would have raised."""
    if rc == 0:
        return
    msg = lib().sc_last_error().decode(errors="replace")
    if rc in (SC_ERR_NOT_POW2, SC_ERR_ROOT_ORDER, SC_ERR_ROOT_NOT_PRIMITIVE, SC_ERR_DIV_ZERO):
        raise AssertionError(msg)
    raise StarkCoreError("starkcore error %d: %s" % (rc, msg))


def device_count():
    return lib().sc_device_count()


def init(device=-1):
    _check(lib().sc_init(device))


def set_tuning(key, value):
    _check(lib().sc_set_tuning(key.encode(), int(value)))


def synchronize():
    _check(lib().sc_synchronize())


def library_stream():
    """the library's own HIP stream as an integer handle (for torch.cuda.ExternalStream: work put on it needs no ordering with
    the library's)"""
    h = _vp()
    _check(lib().sc_stream(ctypes.byref(h)))
    return int(h.value or 0)


def stream_join(other):
    """order the library stream and the raw stream handle `other` with each other on the device; the host does not wait"""
    _check(lib().sc_stream_join(_vp(int(other))))


def pickle_proof(ops, moduli, nfields, modulus_bytes, expect=0):
    """pickle.dumps of the object graph described by `ops` (csrc/proof_pickle.h; host only).  expect: bytes of payload the
    description refers to by address (they appear in the output, not in `ops`)"""
    import numpy as np
    need = ctypes.c_uint64()
    size = len(ops) + int(expect)
    cap = size + size // 8 + 4096
    # one scratch buffer per THREAD (the library call releases the GIL), grown when a proof needs more: fresh megabytes from the
    # allocator are page faults on first touch (a 3 MB proof: more time than pickling it), a zeroed ctypes buffer a memset on top
    out = getattr(_pickle_scratch, "buffer", None)
    if out is None or out.size < cap:
        out = _pickle_scratch.buffer = np.empty(cap + cap // 4, dtype=np.uint8)
    _check(lib().sc_pickle_proof(ops, len(ops), moduli, nfields, modulus_bytes, _vp(out.ctypes.data), out.size, ctypes.byref(need)))
    if need.value > out.size:
        out = _pickle_scratch.buffer = np.empty(need.value + need.value // 4, dtype=np.uint8)
        _check(lib().sc_pickle_proof(ops, len(ops), moduli, nfields, modulus_bytes, _vp(out.ctypes.data), out.size, ctypes.byref(need)))
    return out[:need.value].tobytes()


import threading as _threading
_pickle_scratch = _threading.local()


def fe_bytes(v):
    """int residue -> 16 little-endian bytes (the ABI's element layout)."""
    return int(v).to_bytes(16, "little")


def pack(values):
    """iterable of ints -> packed bytes (one C-level pass: int.to_bytes mapped over the values; anything that is not an int --
    numpy scalars, for instance -- goes through int() first)."""
    values = values if isinstance(values, (list, tuple)) else list(values)
    try:
        return b"".join(map(int.to_bytes, values, itertools.repeat(16), itertools.repeat("little")))
    except TypeError:
        return b"".join(int(v).to_bytes(16, "little") for v in values)


def unpack(buf, count=None):
    """packed elements -> Python ints (one C-level pass over the limbs, then lo | hi << 64)"""
    if count is None:
        count = len(buf) // 16
    limbs = _limb_struct(count).unpack_from(buf)
    return [lo | (hi << 64) for lo, hi in zip(limbs[0::2], limbs[1::2])]


_limb_structs = {}


def _limb_struct(count):
    import struct
    st = _limb_structs.get(count)
    if st is None:
        if len(_limb_structs) > 64:
            _limb_structs.clear()
        st = _limb_structs[count] = struct.Struct("<%dQ" % (2 * count))
    return st


# ------------------------------------------------------------------------------------------------
class DeviceVector:
    """Owner of an sc_vec_t (device-resident field elements)."""

    def __init__(self, n):
        self.n = int(n)
        h = _vp()
        _check(lib().sc_vec_alloc(self.n, ctypes.byref(h)))
        self._h = h

    @classmethod
    def wrap(cls, ptr, n, keep):
        """a vector over device memory somebody else owns (sc_vec_wrap): `keep` -- the owner, e.g. a torch tensor -- stays alive
        as long as this object does.  No copy."""
        v = cls.__new__(cls)
        v.n = int(n)
        h = _vp()
        _check(lib().sc_vec_wrap(_vp(int(ptr)), v.n, ctypes.byref(h)))
        v._h, v._keep = h, keep
        return v

    @property
    def __cuda_array_interface__(self):
        """the elements as an [n][2] array of int64 limbs for whoever speaks the CUDA array interface (torch.as_tensor(vec,
        device=...) is a view: no copy; the consumer keeps this object alive)"""
        return {"shape": (self.n, 2), "typestr": "<i8", "data": (int(self.ptr), False), "version": 3, "strides": None}

    @classmethod
    def adopt(cls, handle, n):
        """owner of an sc_vec_t the library handed out (sc_fri_commit_dev's folded codewords)"""
        v = cls.__new__(cls)
        v.n = int(n)
        v._h = _vp(handle) if not isinstance(handle, _vp) else handle
        return v

    @classmethod
    def from_bytes(cls, data):
        v = cls(len(data) // 16)
        if v.n:
            _check(lib().sc_vec_upload(v._h, 0, bytes(data), v.n))
        return v

    @classmethod
    def from_ints(cls, values):
        return cls.from_bytes(pack(values))

    @classmethod
    def zeros(cls, n):
        v = cls(n)
        _check(lib().sc_vec_zero(v._h))
        return v

    def axpy_shift(self, src, shift, weight):
        """self[shift + j] += weight * src[j]  (one term of the nonlinear combination, code/fast_stark.py:130-145)"""
        _check(lib().sc_axpy_shift_dev(self.ptr, self.n, src.ptr, src.n, int(shift), fe_bytes(weight), None))

    @property
    def ptr(self):
        return lib().sc_vec_ptr(self._h)

    def to_bytes(self, offset=0, count=None):
        count = self.n - offset if count is None else count
        out = ctypes.create_string_buffer(16 * count if count else 16)
        if count:
            _check(lib().sc_vec_download(self._h, offset, out, count))
        return out.raw[:16 * count]

    def gather(self, indices):
        k = len(indices)
        if k == 0:
            return []
        idx = (ctypes.c_uint64 * k)(*[int(i) for i in indices])
        out = ctypes.create_string_buffer(16 * k)
        _check(lib().sc_vec_gather(self._h, idx, k, out))
        return unpack(out.raw, k)

    def free(self):
        if self._h is not None and _lib is not None:
            _lib.sc_vec_free(self._h)
        self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class _HostMemory:
    """one sc_host_alloc'ed block; goes back to the library's pool when the last reference dies"""

    def __init__(self, nbytes):
        p = _vp()
        _check(lib().sc_host_alloc(max(int(nbytes), 1), ctypes.byref(p)))
        self.ptr = p

    def __del__(self):
        p, self.ptr = getattr(self, "ptr", None), None
        if p is not None and _lib is not None:
            _lib.sc_host_free(p)


class Later:
    """A check that is read later (include/starkcore.h sc_later_t): the flags of a division enqueued with sc_*_later_dev.  `wait()`
    -> (a divisor value was zero, the division left a remainder), once; a handle nobody waited for is collected when it is dropped."""

    def __init__(self, handle):
        self._h = handle

    def wait(self):
        words = (ctypes.c_int64 * 8)()
        h, self._h = self._h, None
        _check(lib().sc_later_wait(h, words))
        return words[0] != 0, words[1] >= 0

    def __del__(self):
        if getattr(self, "_h", None) is not None:
            try:
                words = (ctypes.c_int64 * 8)()
                lib().sc_later_wait(self._h, words)
            except Exception:      # noqa: BLE001  (interpreter shutdown)
                pass
            self._h = None


class HostBuffer:
    """Pinned, device-visible host memory from the library's pool (sc_host_alloc): what sc_fri_prove_dev's query kernel writes a
    proof's openings to.  `array` is a uint8 numpy view; views of it keep the memory alive, and it goes back to the pool when the
    last of them dies -- by reference counting (array -> ctypes block -> _HostMemory, no cycle): a block that waited for the cycle
    collector would make the next proof allocate a fresh one (hipHostMalloc of megabytes: 0.1 ms)."""

    def __init__(self, nbytes):
        import numpy as np
        self.nbytes = int(nbytes)
        memory = _HostMemory(self.nbytes)
        self.ptr = memory.ptr
        raw = (ctypes.c_uint8 * max(self.nbytes, 1)).from_address(memory.ptr.value)
        raw._memory = memory                                # numpy keeps `raw` as the array's base; `raw` keeps the block
        self.array = np.frombuffer(raw, dtype=np.uint8, count=self.nbytes)


def query_codewords(codewords, requests):
    """DeviceCodeword.query for several codewords in ONE device round trip: requests[t] = indices into codewords[t];
    returns [(entries, paths)] in the same order (entries identity-preserving, paths fresh objects)."""
    n = len(codewords)
    trees = [cw.tree() for cw in codewords]
    try:
        flat = array.array("Q", itertools.chain.from_iterable(requests))  # (the upper bound is checked by the library)
    except OverflowError:
        raise AssertionError("cannot open invalid index")
    total = len(flat)
    if total == 0:
        return [([], []) for _ in codewords]
    depths = [t.depth for t in trees]
    path_bytes = sum(64 * d * len(req) for d, req in zip(depths, requests))
    elems = ctypes.create_string_buffer(16 * total)
    paths = ctypes.create_string_buffer(path_bytes if path_bytes else 64)
    _check(lib().sc_merkle_query_multi_dev(n, (_vp * n)(*[t._h for t in trees]), (_vp * n)(*[cw.vec.ptr for cw in codewords]),
                                           (ctypes.c_uint64 * total).from_buffer(flat), (ctypes.c_uint64 * n)(*[len(req) for req in requests]), elems, paths))
    values = unpack(elems.raw, total)
    view = memoryview(paths)
    out, vo, po = [], 0, 0
    for cw, req, d in zip(codewords, requests, depths):
        k = len(req)
        out.append((cw._entries(req, values[vo:vo + k]), _path_lists(view, po, d, k)))
        vo += k
        po += 64 * k * d
    return out


def query_codewords_raw(codewords, requests):
    """query_codewords without objects: [(packed residues (bytes, 16 per opening), paths as a uint8 array [openings][64 * depth])]
    -- what proof_objects' segments keep until somebody asks for the objects"""
    import numpy as np
    n = len(codewords)
    trees = [cw.tree() for cw in codewords]
    try:
        flat = array.array("Q", itertools.chain.from_iterable(requests))
    except OverflowError:
        raise AssertionError("cannot open invalid index")
    total = len(flat)
    if total == 0:
        return [(b"", np.zeros((0, 64 * t.depth), dtype=np.uint8)) for t in trees]
    depths = [t.depth for t in trees]
    path_bytes = sum(64 * d * len(req) for d, req in zip(depths, requests))
    elems = np.empty(16 * total, dtype=np.uint8)                  # (uninitialised: the library fills every byte it is asked for;
    raw_p = np.empty(path_bytes if path_bytes else 64, dtype=np.uint8)   # a zeroed ctypes buffer of megabytes is a memset for nothing)
    _check(lib().sc_merkle_query_multi_dev(n, (_vp * n)(*[t._h for t in trees]), (_vp * n)(*[cw.vec.ptr for cw in codewords]),
                                           (ctypes.c_uint64 * total).from_buffer(flat), (ctypes.c_uint64 * n)(*[len(req) for req in requests]),
                                           _vp(elems.ctypes.data), _vp(raw_p.ctypes.data)))
    raw_e = elems.tobytes()
    out, vo, po = [], 0, 0
    for req, d in zip(requests, depths):
        k = len(req)
        out.append((raw_e[16 * vo:16 * (vo + k)], raw_p[po:po + 64 * k * d].reshape(k, 64 * d)))
        vo += k
        po += 64 * k * d
    return out


class PolyTree:
    """Owner of an sc_polytree_t: the subproduct tree over a list of points, resident in HBM (code/ntt.py:66-130).

    One tree serves the zerofier, any number of multipoint evaluations and interpolations over the same points."""

    def __init__(self, points):
        """points: DeviceVector or packed bytes (16 bytes per point), at least one point"""
        self.points = points if isinstance(points, DeviceVector) else DeviceVector.from_bytes(points)
        self.k = self.points.n
        h = _vp()
        _check(lib().sc_polytree_build_dev(self.points.ptr, self.k, ctypes.byref(h), None))
        self._h = h

    def zerofier(self):
        out = DeviceVector(self.k + 1)
        _check(lib().sc_polytree_zerofier_dev(self._h, out.ptr, None))
        return out

    def evaluate(self, coeffs):
        """coeffs: DeviceVector of polynomial coefficients (any length) -> DeviceVector of the k values"""
        out = DeviceVector(self.k)
        _check(lib().sc_polytree_evaluate_dev(self._h, coeffs.ptr, coeffs.n, self.points.ptr, out.ptr, None))
        return out

    def interpolate(self, values):
        """values: DeviceVector of k values -> DeviceVector of the k coefficients of the interpolant (degree < k)"""
        assert values.n == self.k
        out = DeviceVector(self.k)
        _check(lib().sc_polytree_interpolate_dev(self._h, values.ptr, out.ptr, None))
        return out

    def free(self):
        if self._h is not None and _lib is not None:
            _lib.sc_polytree_free(self._h)
        self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class GeoDomain:
    """Owner of an sc_geodomain_t: the tables of a domain that is a geometric progression first * ratio^i, i < n (the trace
    domain {omicron^i} of code/fast_stark.py:84-90).  Same three operations as PolyTree, same results (zerofier, values and
    interpolant are unique), a handful of transforms instead of a tree.  The operations are enqueued on the library stream."""

    def __init__(self, first, ratio, n):
        """first, ratio: int residues; n >= 2 distinct points -- StarkCoreError(unsupported) otherwise (see `create`)"""
        self.first, self.ratio, self.k = int(first), int(ratio), int(n)
        h = _vp()
        _check(lib().sc_geodomain_create(fe_bytes(first), fe_bytes(ratio), self.k, ctypes.byref(h), None))
        self._h = h

    @classmethod
    def create(cls, first, ratio, n):
        """the domain, or None where the progression path does not apply (fewer than two points, repeated points)"""
        dom = cls.__new__(cls)
        dom.first, dom.ratio, dom.k, dom._h = int(first), int(ratio), int(n), None
        h = _vp()
        rc = lib().sc_geodomain_create(fe_bytes(first), fe_bytes(ratio), dom.k, ctypes.byref(h), None)
        if rc == SC_ERR_UNSUPPORTED:
            return None
        _check(rc)
        dom._h = h
        return dom

    @classmethod
    def detect(cls, points):
        """points: DeviceVector.  The GeoDomain of these points if they form a progression of distinct points, else None."""
        if points.n < 2:
            return None
        first, ratio, flag = (ctypes.c_uint64 * 2)(), (ctypes.c_uint64 * 2)(), _int(0)
        _check(lib().sc_geodomain_detect_dev(points.ptr, points.n, first, ratio, ctypes.byref(flag), None))
        if not flag.value:
            return None
        return cls.create(first[0] | (first[1] << 64), ratio[0] | (ratio[1] << 64), points.n)

    def zerofier(self):
        out = DeviceVector(self.k + 1)
        _check(lib().sc_geodomain_zerofier_dev(self._h, out.ptr, None))
        return out

    def evaluate(self, coeffs):
        out = DeviceVector(self.k)
        _check(lib().sc_geodomain_evaluate_dev(self._h, coeffs.ptr, coeffs.n, out.ptr, None))
        return out

    def interpolate(self, values):
        assert values.n == self.k
        out = DeviceVector(self.k)
        _check(lib().sc_geodomain_interpolate_dev(self._h, values.ptr, out.ptr, None))
        return out

    def free(self):
        if self._h is not None and _lib is not None:
            _lib.sc_geodomain_free(self._h)
        self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def domain_tables(points):
    """the device structure that serves fast_zerofier / fast_evaluate / fast_interpolate over `points` (DeviceVector or packed
    bytes): the progression tables where the points are a geometric progression of distinct points, the subproduct tree otherwise"""
    points = points if isinstance(points, DeviceVector) else DeviceVector.from_bytes(points)
    return GeoDomain.detect(points) or PolyTree(points)


_struct_cache = {}


def _digest_struct(count, skipped=0):
    import struct
    st = _struct_cache.get((count, skipped))
    if st is None:
        st = _struct_cache[(count, skipped)] = struct.Struct("64s" * count + ("%dx" % (64 * skipped) if skipped else ""))
    return st


def _path_lists(view, offset, depth, k, keep=None):
    """k authentication paths of `depth` 64-byte digests each, packed back to back in `view` from byte `offset`, as the
    reference's lists of fresh bytes objects (merkle.py:16-27).  One C-level pass per tree: the ~25 000 digest objects of a
    2^22 Fri.prove are the largest host cost of its query phase.  keep < depth: only the first `keep` digests of every path
    become objects (the part of a path below a sharded commitment's sub-roots)."""
    keep = depth if keep is None else min(keep, depth)
    if keep == 0:
        return [[] for _ in range(k)]
    return list(map(list, _digest_struct(keep, depth - keep).iter_unpack(view[offset:offset + 64 * k * depth])))


class MerkleTree:
    """Owner of an sc_merkle_t: all levels resident in HBM, so `open` is a gather (code/merkle.py:16-27)."""

    def __init__(self, handle, root, n):
        self._h = handle
        self._root = root        # None: built asynchronously, fetched (and waited for) on first use
        self.n = n
        self.depth = n.bit_length() - 1

    @property
    def root(self):
        if self._root is None:
            out = ctypes.create_string_buffer(64)
            _check(lib().sc_merkle_root(self._h, out))
            self._root = out.raw
        return self._root

    @classmethod
    def from_device(cls, vec):
        root = ctypes.create_string_buffer(64)
        h = _vp()
        _check(lib().sc_merkle_build_dev(vec.ptr, vec.n, root, ctypes.byref(h), None))
        return cls(h, root.raw, vec.n)

    @classmethod
    def from_device_async(cls, vec):
        """the build is enqueued, not waited for: `.root` waits (the caller prepares its next step meanwhile)"""
        h = _vp()
        _check(lib().sc_merkle_build_async_dev(vec.ptr, vec.n, ctypes.byref(h), None))
        return cls(h, None, vec.n)

    @classmethod
    def from_bytes(cls, data):
        n = len(data) // 16
        root = ctypes.create_string_buffer(64)
        h = _vp()
        _check(lib().sc_merkle_build(bytes(data), n, root, ctypes.byref(h)))
        return cls(h, root.raw, n)

    @classmethod
    def from_device_ptr(cls, ptr, n, stream=None):
        """tree over n field elements at a raw device pointer (e.g. a torch tensor's storage); built on `stream` (a
        ctypes.c_void_p hipStream_t; None = the library's), which is synchronized before the root is returned"""
        root = ctypes.create_string_buffer(64)
        h = _vp()
        _check(lib().sc_merkle_build_dev(ptr, n, root, ctypes.byref(h), stream))
        return cls(h, root.raw, n)

    @classmethod
    def from_device_ptr_async(cls, ptr, n, stream=None):
        """as from_device_ptr, but only enqueued on `stream`: `.root` waits (a caller that needs a level of the tree, not its
        root, never waits at all -- copy_level on the same stream is ordered behind the build)"""
        h = _vp()
        _check(lib().sc_merkle_build_async_dev(ptr, n, ctypes.byref(h), stream))
        return cls(h, None, n)

    @classmethod
    def from_device_ptr_noroot(cls, ptr, n, stream=None):
        """enqueue-only build of a tree whose root is not expected to be read (a rank's local subtree of a sharded commit): no
        pinned root slot, no publish kernel.  `.root` still works, at the price of a device-wide wait."""
        h = _vp()
        _check(lib().sc_merkle_build_noroot_dev(ptr, n, ctypes.byref(h), stream))
        return cls(h, None, n)

    @classmethod
    def from_folded_slab(cls, src_ptr, rows, cols, R, col_base, alpha, offset, omega, dst_ptr, stream=None):
        """the fold of fri.py:85 on a rank's column slab [rows][cols] (sc_fri_fold_slab_dev) AND the enqueue-only local subtree
        over the folded slab, in one library call: the tree's leaf stage computes the fold (alpha, offset, omega: packed bytes)"""
        h = _vp()
        _check(lib().sc_fri_fold_slab_build_dev(src_ptr, rows, cols, R, col_base, alpha, offset, omega, dst_ptr, ctypes.byref(h), stream))
        return cls(h, None, (rows // 2) * cols)

    @classmethod
    def from_digests_ptr(cls, ptr, count, stream=None):
        """tree whose level 0 is `count` given 64-byte digests at a raw device pointer; only enqueued: `.root` waits (polling the
        pinned slot the root is published to)"""
        h = _vp()
        _check(lib().sc_merkle_from_digests_dev(ptr, count, None, ctypes.byref(h), stream))
        return cls(h, None, count)

    def copy_level(self, level, dst_ptr, stream=None):
        """copy the (n >> level) digests of one level into caller-owned device memory"""
        _check(lib().sc_merkle_level_copy_dev(self._h, level, dst_ptr, stream))

    def open_batch(self, indices):
        k = len(indices)
        for i in indices:
            assert 0 <= i < self.n, "cannot open invalid index"
        if k == 0:
            return []
        idx = (ctypes.c_uint64 * k)(*[int(i) for i in indices])
        out = ctypes.create_string_buffer(64 * self.depth * k)
        _check(lib().sc_merkle_open_batch(self._h, idx, k, out))
        # one C-level pass creates all the 64-byte digest objects (they end up, one by one, in the transcript)
        return _path_lists(memoryview(out), 0, self.depth, k)

    def open(self, index):
        return self.open_batch([index])[0]

    def free(self):
        if self._h is not None and _lib is not None:
            _lib.sc_merkle_free(self._h)
        self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


import operator as _operator

_value_of = _operator.attrgetter("value")
_FieldElement = None


def _field_element():
    """algebra.FieldElement, imported on first use (algebra does not depend on this module, but importing it at load time would
    make every C-ABI-only user pay for it)"""
    global _FieldElement
    if _FieldElement is None:
        from algebra import FieldElement
        _FieldElement = FieldElement
    return _FieldElement


class DeviceCodeword(Sequence):
    """A list-like view of a device-resident codeword.

    Behaves like the `list[FieldElement]` the reference passes around (len, indexing, iteration,
    equality with lists) but keeps the data in HBM; `Fri` and `Merkle` recognise it and stay on the
    device.  Elements are materialised as FieldElement only when asked for.
    """

    def __init__(self, vec, field, elements=None):
        self.vec = vec
        self.field = field
        self._tree = None
        # index -> FieldElement.  Identity matters: the Fiat-Shamir transcript is pickle.dumps(objects)
        # (code/ip.py:18-25) and pickle memoises by object identity; the reference pushes the SAME
        # FieldElement object when a codeword entry appears twice (last codeword + query triple, or the `c`
        # of one round and the `a`/`b` of the next: code/fri.py:91,104-105), so entries are created once.
        self._elems = {}
        self._full = None
        if elements is not None:
            self._full = elements
            self._elems = None

    # -- construction helpers
    @classmethod
    def from_list(cls, values, field):
        values = list(values)
        return cls(DeviceVector.from_ints(list(map(_value_of, values))), field, elements=values)

    def __len__(self):
        return self.vec.n

    def _fe(self, v):
        return _field_element()(v, self.field)

    def tolist(self):
        if self._full is None:
            known = self._elems
            ints = unpack(self.vec.to_bytes(), self.vec.n)
            self._full = [known[i] if i in known else self._fe(v) for i, v in enumerate(ints)]
            holder = getattr(self, "_proof_entries", None)
            if holder is not None:                          # a proof stream describes entries of this codeword: it shares the
                known.update(enumerate(self._full))         # cache that is retired here and must know every object
            self._elems = None
        return self._full

    def __getitem__(self, i):
        if isinstance(i, slice):
            return self.tolist()[i]
        n = self.vec.n
        if i < 0:
            i += n
        if not 0 <= i < n:
            raise IndexError("codeword index out of range")
        return self.gather([i])[0]

    def gather(self, indices):
        """Entries at `indices` (one D2H gather for the ones not materialised yet)."""
        if self._full is not None:
            return [self._full[i] for i in indices]
        known = self._elems
        missing = [i for i in dict.fromkeys(indices) if i not in known]
        if missing:
            for i, v in zip(missing, self.vec.gather(missing)):
                known[i] = self._fe(v)
        return [known[i] for i in indices]

    def __iter__(self):
        return iter(self.tolist())

    def __eq__(self, other):
        if isinstance(other, DeviceCodeword):
            return self.vec.to_bytes() == other.vec.to_bytes()
        return self.tolist() == list(other)

    __hash__ = None

    def tree(self):
        if self._tree is None:
            self._tree = MerkleTree.from_device(self.vec)
        return self._tree

    def start_tree(self):
        """enqueue the build of the tree and return at once (`tree().root` waits for it)"""
        if self._tree is None:
            self._tree = MerkleTree.from_device_async(self.vec)
        return self._tree

    def fold_commit(self, alpha, offset, omega, out_vec):
        """One round of Fri.commit in one library call (fri.py:85 then the tree of the folded codeword, both only enqueued):
        the folded DeviceCodeword, whose tree().root waits for the device."""
        h = _vp()
        _check(lib().sc_fri_fold_commit_dev(self.vec.ptr, self.vec.n, fe_bytes(alpha.value), fe_bytes(offset.value), fe_bytes(omega.value),
                                            out_vec.ptr, ctypes.byref(h), None))
        folded = DeviceCodeword(out_vec, self.field)
        folded._tree = MerkleTree(h, None, out_vec.n)
        return folded

    def query(self, indices):
        """One device round trip: the entries at `indices` (identity-preserving, like gather) and one freshly created
        authentication path per requested index (paths are new objects every time, as in the reference, which
        recomputes them per Merkle.open call)."""
        tree = self.tree()
        k = len(indices)
        if k == 0:
            return [], []
        idx = (ctypes.c_uint64 * k)(*[int(i) for i in indices])
        d = tree.depth
        elems = ctypes.create_string_buffer(16 * k)
        paths = ctypes.create_string_buffer(64 * d * k if d else 64)
        _check(lib().sc_merkle_query_dev(tree._h, self.vec.ptr, idx, k, elems, paths))
        return self._entries(indices, unpack(elems.raw, k)), _path_lists(memoryview(paths), 0, d, k)

    def _entries(self, indices, values):
        """FieldElement objects for freshly fetched residues, created once per index (see __init__)"""
        if self._full is not None:
            return [self._full[i] for i in indices]
        known, fe, field, new = self._elems, _field_element(), self.field, object.__new__
        for i, v in zip(indices, values):
            if i not in known:
                # FieldElement(v, field) without the call into __init__ (algebra.py:16-18 sets exactly these two attributes):
                # half the cost per object, and a query phase creates over a thousand of them
                e = new(fe)
                e.value = v
                e.field = field
                known[i] = e
        return [known[i] for i in indices]
