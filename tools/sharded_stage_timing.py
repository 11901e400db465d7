#!/usr/bin/env python3
"""Per-rank stage timing of the sharded four-step NTT for a given (log2n, world), on one GPU (local shapes only) -- dev tool."""
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
def timeit(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps): fn()
    e1.record(stream); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for log2n, world in [(22, 2), (23, 4), (24, 8), (24, 1), (21, 1)]:
    n = 1 << log2n
    eng = ShardedNtt(log2n, nth_root(n), 0, world, dev)
    R, C = eng.n1, eng.n2
    x = torch.randint(0, 1 << 62, eng.local_shape(True), dtype=torch.int64, device=dev)   # timing only
    y = torch.empty(eng.local_shape(False), dtype=torch.int64, device=dev)
    rw, cw = R // world, C // world
    a = eng._buf("a", (R, cw, 2))
    recv = torch.empty((world, rw, cw, 2), dtype=torch.int64, device=dev)
    res = dict(log2n=log2n, world=world, local_elems=n // world)
    res["cols_ntt_us"] = round(timeit(lambda: eng.engine.cols_ntt(x, a, R, cw, pow(eng.root, C, P))), 1)
    res["twiddle_us"] = round(timeit(lambda: eng.engine.twiddle(a, R, cw, 0, 0, eng.root, n, 1)), 1)
    res["assemble_us"] = round(timeit(lambda: eng.assemble_rows(recv, R, C)), 1) if world > 1 else 0.0
    rows = eng.assemble_rows(recv, R, C) if world > 1 else a
    res["rows_ntt_us"] = round(timeit(lambda: eng.engine.rows_ntt_t(rows, y, C, rw, pow(eng.root, R, P))), 1)
    res["sum_us"] = round(res["cols_ntt_us"] + res["twiddle_us"] + res["assemble_us"] + res["rows_ntt_us"], 1)
    res["fused_cols_us"] = round(timeit(lambda: eng.engine.cols_ntt_twiddled(x, a, R, cw, pow(eng.root, C, P), eng.root, n, 0, False)), 1)
    res["fused_rows_us"] = round(timeit(lambda: eng.engine.rows_ntt_t_chunked(recv, y, C, rw, world, pow(eng.root, R, P))), 1) if world > 1 else res["rows_ntt_us"]
    res["fused_sum_us"] = round(res["fused_cols_us"] + res["fused_rows_us"], 1)
    lg = log2n - (world.bit_length() - 1)
    xs = sc.DeviceVector(1 << lg); ys = sc.DeviceVector(1 << lg)
    rt = sc.fe_bytes(nth_root(1 << lg))
    res["single_gpu_ntt_same_size_us"] = round(timeit(lambda: sc._check(sc.lib().sc_ntt_dev(xs.ptr, ys.ptr, 1 << lg, rt, 0, sp))), 1)
    print(json.dumps(res), flush=True)

# ---- overlap of the corner turn with the row stage (ShardedNtt._transform_overlapped), SIMULATED on one GPU: there is no xGMI on a
# 1-GPU box, so the exchange of row block q is stood in for by device-to-device copies of `slow` x its bytes on a side stream
# (slow = 1: as fast as HBM allows; slow = 8: a link ~8x slower than an HBM copy, the order of 7 xGMI links vs HBM); what is
# measured is the pipeline: K async "exchanges" back to back on the side stream, the row transforms of block q on the compute
# stream as soon as block q has landed (event wait), against the blocking form (whole exchange, then the whole row stage).
side = torch.cuda.Stream(device=dev)
for log2n, world, K in [(24, 8, 4), (24, 8, 8), (23, 4, 4)]:
    n = 1 << log2n
    eng = ShardedNtt(log2n, nth_root(n), 0, world, dev)
    for (R, C, tag) in ((eng.n1, eng.n2, "forward"), (eng.n2, eng.n1, "inverse")):
        rw, cw = R // world, C // world
        if rw % K:
            continue
        rk = rw // K
        root_rows = pow(eng.root, R, P)
        a = torch.randint(0, 1 << 62, (R, cw, 2), dtype=torch.int64, device=dev)
        recv = torch.empty((K, world, rk, cw, 2), dtype=torch.int64, device=dev)
        recv1 = recv.view(world * K, rk, cw, 2)
        y = torch.empty((C, rw, 2), dtype=torch.int64, device=dev)
        for slow in (1, 8):
            def exch_block(q):
                for _ in range(slow):
                    recv[q].copy_(a.view(world, K, rk, cw, 2)[:, q], non_blocking=True)
            def blocking():
                side.wait_stream(stream)
                with torch.cuda.stream(side):
                    for q in range(K): exch_block(q)
                stream.wait_stream(side)
                for q in range(K):
                    assert eng.engine.rows_ntt_t_block(recv[q], y, q * rk, C, rk, world, root_rows, rw)
            def overlapped():
                side.wait_stream(stream)
                evs = []
                with torch.cuda.stream(side):
                    for q in range(K):
                        exch_block(q)
                        ev = torch.cuda.Event(); ev.record(side); evs.append(ev)
                for q in range(K):
                    stream.wait_event(evs[q])
                    assert eng.engine.rows_ntt_t_block(recv[q], y, q * rk, C, rk, world, root_rows, rw)
            def only_exchange():
                side.wait_stream(stream)
                with torch.cuda.stream(side):
                    for q in range(K): exch_block(q)
                stream.wait_stream(side)
            def only_rows():
                for q in range(K):
                    assert eng.engine.rows_ntt_t_block(recv[q], y, q * rk, C, rk, world, root_rows, rw)
            t_x, t_r, t_b, t_o = timeit(only_exchange, 10), timeit(only_rows, 10), timeit(blocking, 10), timeit(overlapped, 10)
            hidden = (t_b - t_o) / t_x if t_x > 0 else 0.0
            print(json.dumps(dict(overlap_sim=tag, log2n=log2n, world=world, blocks=K, exchange_slowdown=slow, exchange_us=round(t_x, 1), rows_us=round(t_r, 1),
                                  blocking_us=round(t_b, 1), overlapped_us=round(t_o, 1), exchange_hidden_frac=round(hidden, 2))), flush=True)
