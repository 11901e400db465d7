#!/usr/bin/env python3
"""Where a sharded transform's time goes at world 1 over RCCL (dev tool).

  python tools/sharded_timeline.py run [log2n=21] [steps=200]     # one rank, backend nccl: host enqueue time vs device time per step
  python tools/sharded_timeline.py report <kernel_trace.csv>      # timeline of the last few steps of a rocprofv3 --kernel-trace of `run`

`run` prints one JSON line: ms per forward+inverse step with the host clock (barrier on both sides), the host time spent
ENQUEUEING the same steps (no synchronisation inside the window: if it equals the total, the path is host-bound), and the same two
numbers for the plain single-GPU transform of the same size."""
import csv
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "stark-anatomy_amd"))


def run(log2n, steps):
    import ctypes
    import torch
    import torch.distributed as dist
    import starkcore as sc
    from sharded import ShardedNtt, P
    GEN = 85408008396924667383611388730472331217
    r, order = GEN, 1 << 119
    n = 1 << log2n
    while order != n:
        r, order = r * r % P, order >> 1
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    sc.init(0)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    out = {"log2n": log2n, "steps": steps}
    from sharded import init_native_comm
    native = init_native_comm(0, 1, dev)
    out["native_rccl_comm"] = bool(native)
    forms = [("in_place_nothing_to_exchange", {}),
             ("torch_own_block_through_rccl", dict(always_exchange=True)),
             ("torch_own_block_through_rccl_4_blocks", dict(always_exchange=True, overlap_chunks=4))]
    if native:
        forms += [("native_own_block_through_rccl", dict(always_exchange=True, native_exchange=True)),
                  ("native_own_block_through_rccl_2_blocks", dict(always_exchange=True, native_exchange=True, overlap_chunks=2)),
                  ("native_own_block_through_rccl_4_blocks", dict(always_exchange=True, native_exchange=True, overlap_chunks=4)),
                  ("native_own_block_through_rccl_4_blocks_not_deferred", dict(always_exchange=True, native_exchange=True, overlap_chunks=4, defer_last_pass=False))]
    forms.append(("direct_store_no_collective", dict(direct_store=True)))
    only = os.environ.get("TIMELINE_FORMS")
    for name, kw in forms:
        if only and name not in only.split(","):
            continue
        eng = ShardedNtt(log2n, r, 0, 1, dev, **kw)
        if kw.get("native_exchange"):
            eng.stages.native = True
        x = eng.synthetic_input(seed=1)
        y = torch.empty(eng.local_shape(False), dtype=torch.int64, device=dev)
        z = torch.empty_like(x)

        def step():
            eng.forward(x, y)
            eng.inverse(y, z)

        for _ in range(20):
            step()
        torch.cuda.synchronize()
        assert torch.equal(z, x), name
        best = None
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            rec = {"ms_per_step": 1e3 * (t2 - t0) / steps, "host_enqueue_ms_per_step": 1e3 * (t1 - t0) / steps}
            if best is None or rec["ms_per_step"] < best["ms_per_step"]:
                best = rec
        out[name] = best
    # the plain transform of the same size through the same stream
    lib = sc.lib()
    a, b, c = sc.DeviceVector(n), sc.DeviceVector(n), sc.DeviceVector(n)
    rt = sc.fe_bytes(r)
    sp = ctypes.c_void_p(stream.cuda_stream)

    def plain():
        sc._check(lib.sc_ntt_dev(a.ptr, b.ptr, n, rt, 0, sp))
        sc._check(lib.sc_ntt_dev(b.ptr, c.ptr, n, rt, 1, sp))

    for _ in range(20):
        plain()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        plain()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    out["plain_single_gpu"] = {"ms_per_step": 1e3 * (t2 - t0) / steps, "host_enqueue_ms_per_step": 1e3 * (t1 - t0) / steps}
    print(json.dumps(out), flush=True)
    from sharded import destroy_native_comm
    destroy_native_comm()
    dist.destroy_process_group()


def report(path, nsteps=3):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))

    def short(r):
        return r["Kernel_Name"].split("(")[0].replace("void ", "").replace("sc::", "")[:60]

    # per-kernel averages over the whole trace
    agg = {}
    for r in rows:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        a = agg.setdefault(short(r), [0, 0])
        a[0] += 1
        a[1] += d
    print("%-60s %8s %10s" % ("kernel", "calls", "avg_us"))
    for k, (cnt, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-60s %8d %10.2f" % (k, cnt, tot / cnt / 1e3))
    # the busiest stretch: the last 40 kernels before the last gap of > 5 ms (i.e. the end of a timed window)
    cut = len(rows)
    for i in range(len(rows) - 1, 0, -1):
        if int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"]) > 5_000_000:
            cut = i
        if cut - i > 2000:
            break
    tail = rows[max(0, cut - 40):cut]
    if not tail:
        return
    t0 = int(tail[0]["Start_Timestamp"])
    prev_end = t0
    print("\n%9s %8s %8s  %s" % ("t_us", "dur_us", "gap_us", "kernel"))
    busy = 0
    for r in tail:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print("%9.1f %8.1f %8.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, short(r)))
        busy += e - s
        prev_end = max(prev_end, e)
    print("span_us %.1f busy_us %.1f kernels %d" % ((prev_end - t0) / 1e3, busy / 1e3, len(tail)))


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "report":
        report(sys.argv[2])
    else:
        args = [a for a in sys.argv[1:] if a != "run"]
        log2n = int(args[0]) if args else 21
        steps = int(args[1]) if len(args) > 1 else 200
        run(log2n, steps)
