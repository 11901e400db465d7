"""A/B of the batch-inversion division kernel (csrc/core.hip: pointwise_div_kernel<K>) and of the exactness scan (vec_degree_kernel):
sc_pointwise_div_dev and sc_vec_degree_dev on vectors of the size a 2^24 proof divides (2^22), wall clock per call with the stream
drained -- run once per STARKCORE_DIV_K in {8, 16}; the kernel's own duration is in the rocprofv3 kernel stats of the same command."""
import os
import sys
import time
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "stark-anatomy_amd"))
import starkcore as sc                                  # noqa: E402

P = 1 + 407 * (1 << 119)


def main():
    logn = int(sys.argv[1]) if len(sys.argv) > 1 else 22
    n = 1 << logn
    rng = np.random.default_rng(5)
    raw = rng.integers(0, 1 << 63, size=(2, n, 2), dtype=np.uint64)
    raw[:, :, 1] &= (1 << 62) - 1                       # below p
    a, b = sc.DeviceVector.from_bytes(raw[0].tobytes()), sc.DeviceVector.from_bytes(raw[1].tobytes())
    out = sc.DeviceVector(n)
    lib = sc.lib()
    for name, call in (("sc_pointwise_div_dev", lambda: sc._check(lib.sc_pointwise_div_dev(a.ptr, b.ptr, out.ptr, n, None))),
                       ("sc_vec_degree_dev of a vector whose top 3/4 is zero", None)):
        if call is None:
            z = np.zeros((n, 2), dtype=np.uint64)
            z[: n // 4] = raw[0][: n // 4]
            v = sc.DeviceVector.from_bytes(z.tobytes())
            import ctypes
            deg = ctypes.c_int64()
            call = lambda: sc._check(lib.sc_vec_degree_dev(v.ptr, n, ctypes.byref(deg), None))
        for _ in range(3):
            call()
        ts = []
        for _ in range(20):
            t0 = time.perf_counter()
            call()
            ts.append(time.perf_counter() - t0)
        print(f"{name} n=2^{logn} K={os.environ.get('STARKCORE_DIV_K', 'default')}: best {min(ts) * 1e6:.1f} us, median {sorted(ts)[10] * 1e6:.1f} us")
    # the quotient against Python ints at a few positions
    got = np.frombuffer(out.to_bytes(), dtype=np.uint64).reshape(n, 2)
    for i in (0, 1, n // 3, n - 1):
        x = int(raw[0][i][0]) | int(raw[0][i][1]) << 64
        y = int(raw[1][i][0]) | int(raw[1][i][1]) << 64
        q = int(got[i][0]) | int(got[i][1]) << 64
        assert q == x * pow(y, P - 2, P) % P, i
    print("quotients at 4 positions equal a * b^(p-2) mod p in Python ints")


if __name__ == "__main__":
    main()
