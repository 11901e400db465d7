#!/usr/bin/env python3
"""Per-rank stage timing of the sharded four-step NTT for a given (log2n, world), on one GPU (local shapes only) -- dev tool.
Stages of the plan object (sc_fourstep_*): column stage (outer twiddle fused, own block in place), row stage in one piece, and
the row stage in K row blocks with the second pass deferred to one launch (what an overlapped corner turn runs) or not."""
import ctypes, json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "stark-anatomy_amd"))
import torch
import starkcore as sc
from sharded import ShardedNtt, P
GEN = 85408008396924667383611388730472331217
def nth_root(n):
    r, order = GEN, 1 << 119
    while order != n: r, order = r * r % P, order >> 1
    return r
sc.init(0)
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)
sp = ctypes.c_void_p(stream.cuda_stream)
def timeit(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    best = None
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps): fn()
        e1.record(stream); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / reps * 1e3
        best = t if best is None or t < best else best
    return best
for log2n, world in [(22, 2), (23, 4), (24, 8), (21, 1), (24, 1)]:
    n = 1 << log2n
    eng = ShardedNtt(log2n, nth_root(n), 0, world, dev)
    st = eng.stages
    res = dict(log2n=log2n, world=world, local_elems=n // world)
    for inv, tag in ((0, "forward"), (1, "inverse")):
        R, C = (eng.n2, eng.n1) if inv else (eng.n1, eng.n2)
        rw, cw = R // world, C // world
        x = torch.randint(0, 1 << 62, (R, cw, 2), dtype=torch.int64, device=dev)   # timing only
        y = torch.empty((C, rw, 2), dtype=torch.int64, device=dev)
        send = torch.empty((world, rw, cw, 2), dtype=torch.int64, device=dev)
        recv = torch.randint(0, 1 << 62, (world, rw, cw, 2), dtype=torch.int64, device=dev)
        r = {}
        r["cols_us"] = round(timeit(lambda: st.cols(inv, x, send, recv)), 1)
        r["rows_us"] = round(timeit(lambda: st.rows(inv, recv, y, 0, 1, False)), 1)
        for K in (2, 4):
            if rw % K:
                continue
            def blocks(defer, K=K):
                for q in range(K):
                    st.rows(inv, recv, y, q, K, defer)
                if defer:
                    st.rows_finish(inv, y)
            r["rows_%d_blocks_deferred_us" % K] = round(timeit(lambda: blocks(True)), 1)
            r["rows_%d_blocks_us" % K] = round(timeit(lambda: blocks(False)), 1)
        res[tag] = r
    lg = log2n - (world.bit_length() - 1)
    xs = sc.DeviceVector(1 << lg); ys = sc.DeviceVector(1 << lg)
    rt = sc.fe_bytes(nth_root(1 << lg))
    res["single_gpu_ntt_same_size_us"] = round(timeit(lambda: sc._check(sc.lib().sc_ntt_dev(xs.ptr, ys.ptr, 1 << lg, rt, 0, sp))), 1)
    print(json.dumps(res), flush=True)
