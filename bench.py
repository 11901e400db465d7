#!/usr/bin/env python3
"""bench.py -- NTT field-elements/s on MI355X (BASELINE.json metric), one JSON line on stdout.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

N = 1: workload = BASELINE configs[1]: forward + inverse 2^20-point NTT, data resident in HBM, over one BATCH of independent
       columns per step (--columns, default 64: the registers of a trace, code/fast_stark.py:84-104; 1 GiB per vector, well past
       the 256 MiB Infinity Cache) through sc_ntt_columns_dev -- one set of launches per direction;
       value = 2 * n * columns * K / elapsed  (field elements / s).  `one_column_at_a_time` carries the same transforms issued one
       column after the other with sc_ntt_dev (--columns 1 makes that the step: the headline of rounds 1-5),
       `extras.ntt_2p24_strong` the N = 1 member of the north_star series (forward + inverse at 2^24, with `frac`).
N > 1: the north_star line: forward + inverse 2^24-point NTT as a four-step transform sharded over the ranks (STRONG scaling:
       the same 2^24 for every N; `--scaling weak` times 2^21 elements per GPU instead, `--log2n` any size), one corner turn per
       transform (stark-anatomy_amd/sharded.py).  A bare `python bench.py --gpus N` re-launches itself under
       torch.distributed.run; the line also carries the whole BASELINE configs[4] call census on the sharded layout
       (extras.stark_census_sharded); `--workload stark_census` makes that the timed step.  With fewer GPUs than ranks the
       ranks share devices and exchange through gloo (labelled functional run).
       The forms of the corner turn (torch.distributed or the library's own RCCL communicator; one blocking exchange or row
       blocks overlapped with the row stage) are each checked against the first and timed for a few steps; the fastest correct
       one is measured (`config.corner_turn` names it and lists the probe times).

Timing: W untimed steps, then exactly K steps between barrier + torch.cuda.synchronize(), max over ranks -> `value`,
`ms_per_step`, `roofline` (launch duration by HIP events on the launch stream).  The same window is then repeated after
CLOCK_RAMP_MS of untimed steps and reported beside it as `clock_ramp.steady_state` (information: a short window straight after
start-up runs at the board's idle clock).  `cpu_baseline` and `extras` (Fri.prove, LDE, census, Merkle, ...) follow, rank 0, N = 1.
"""
import argparse
import itertools
import json
import math
import os
import sys
import time

T0 = time.perf_counter()             # the process's start, for --budget-s
REPO = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(REPO, "stark-anatomy_amd")
for p in (PKG, REPO):
    if p not in sys.path:
        sys.path.insert(0, p)

# the workloads (what a step IS) and the set-up of the sharded transform live in the package; this script times them
from workloads import (nth_root, sharded_census, stark_census, census_record, stark_prove_measure,          # noqa: E402
                       plain_stark_prove_measure)
from sharded_setup import (HBM_PEAK_GBS, BYTES_PER_ELEMENT_PER_TRANSFORM, strong_record, sharded_setup,    # noqa: E402
                           stage_breakdown, node_facts, collective_label)
sys.path.insert(0, os.path.join(REPO, "tools"))
from pmc_records import measured_valu, measured_traffic, pmc_figures_are_current, kernel_source_digest       # noqa: E402,F401  (what profiles/ holds)

COLS_ELEMS_PER_LAUNCH = 1 << 26       # csrc/core.hip sc_ntt_columns_dev: elements one set of launches covers (64 columns of 2^20)


def column_launch_sets(n, cols):
    per = max(1, min(65536, COLS_ELEMS_PER_LAUNCH // n))
    return (cols + per - 1) // per


CLOCK_RAMP_MS = 150.0      # untimed load between the contract's window (`value`) and its repetition (`clock_ramp.steady_state`)


def cpu_baseline(sample_log2n):
    """Pure-Python port of the reference's recursive ntt/intt (oracle/py_oracle.py), 1 core, bounded sample; next to it the
    C restatement (oracle/stark_oracle.c) on 1 core and on all cores (independent transforms, one per thread)."""
    from oracle import py_oracle as po
    import synth
    n = 1 << sample_log2n
    xs = synth.synth_ints(1, n)
    root = po.primitive_nth_root(n)
    sys.setrecursionlimit(10000)
    t0 = time.perf_counter()
    ys = po.ntt(root, xs)
    zs = po.intt(root, ys)
    dt = time.perf_counter() - t0
    assert zs == xs
    out = {"value": 2 * n / dt, "unit": "field-elements/s", "cores": 1, "kind": "port",
           "sample": "pure-Python port of code/ntt.py ntt+intt at n=2^%d, %.1f s, host has %d cores; the port works on ints with the built-in "
                     "pow(root, i, p) where code/ntt.py runs FieldElement.__xor__ on objects (algebra.py:38-45), so it is about 6x FASTER "
                     "than the reference itself (5.5-8.7 k elements/s, SURVEY.md App. C): every GPU/CPU ratio quoted from it is conservative"
                     % (sample_log2n, dt, os.cpu_count())}
    try:
        m = 1 << 18
        data = synth.synth_packed(1, m).tobytes()
        r2 = po.primitive_nth_root(m)

        def pair(_):
            y = po.C.ntt(r2, data, m)
            return po.C.intt(r2, y, m) == data

        t0 = time.perf_counter()
        assert pair(0)
        dtc = time.perf_counter() - t0
        out["c_port"] = {"value": 2 * m / dtc, "unit": "field-elements/s", "cores": 1, "sample": "oracle/stark_oracle.c ntt+intt at n=2^18, %.2f s" % dtc}
        # all cores: one independent 2^18 transform pair per thread (ctypes releases the GIL inside the C call)
        from concurrent.futures import ThreadPoolExecutor
        cores = os.cpu_count() or 1
        with ThreadPoolExecutor(max_workers=cores) as ex:
            t0 = time.perf_counter()
            assert all(ex.map(pair, range(cores)))
            dta = time.perf_counter() - t0
        out["c_port_all_cores"] = {"value": 2 * m * cores / dta, "unit": "field-elements/s", "cores": cores,
                                   "sample": "oracle/stark_oracle.c: %d independent ntt+intt pairs at n=2^18, one per thread, %.2f s" % (cores, dta)}
    except Exception as e:      # the C oracle is optional for the baseline
        out.setdefault("c_port", {"error": str(e)})
    return out


def emit(out):
    """the ONE JSON line, as the last line of stdout: RCCL writes a version banner through C stdio, which sits in libc's buffer
    until the process exits -- flush it first so that it cannot land behind the line"""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:      # noqa: BLE001
        pass
    sys.stdout.flush()
    print(json.dumps(out), flush=True)


def _free_port():
    import socket
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        return s_.getsockname()[1]


def self_launch(args):
    """`python bench.py --gpus N` from a bare shell: re-exec under torch.distributed.run, one rank per GPU."""
    import subprocess
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default 2000 (10 for a functional run whose ranks share GPUs)")
    ap.add_argument("--warmup", type=int, default=None, help="default 200 (2 for a functional run whose ranks share GPUs)")
    ap.add_argument("--workload", choices=("ntt", "stark_census", "stark_prove"), default="ntt",
                    help="ntt (headline, BASELINE configs[1]; N > 1: the sharded four-step transform), stark_census (the polynomial-core call census of "
                         "BASELINE configs[4] on the sharded layout) or stark_prove (sharded_stark.ShardedFastStark.prove on a synthetic AIR: configs[4] as a prover)")
    ap.add_argument("--log2n", type=int, default=None, help="override the transform size (ntt) / the FRI domain (stark_census)")
    ap.add_argument("--columns", type=int, default=None,
                    help="N = 1: independent columns per step (default 64 at 2^20, fewer above so that a vector stays at 1 GiB; 1 = one "
                         "transform at a time through sc_ntt_dev, the step of rounds 1-5)")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE", help="sc_set_tuning(KEY, VALUE) before anything runs (A/B of a library knob in bench conditions; the line carries it in config.tune)")
    ap.add_argument("--cpu-sample-log2n", type=int, default=18)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the Fri.prove / LDE / census side measurements")
    ap.add_argument("--force-sharded", action="store_true", help="run the multi-GPU four-step code path (process group, corner turn) even with one rank")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="N > 1: strong = the north_star's 2^24 transform for every N (default); weak = 2^21 elements per GPU")
    ap.add_argument("--force-diag-exchange", action="store_true",
                    help="send the block a rank keeps for itself through the collective as well (a one-rank world then exercises the whole RCCL path)")
    ap.add_argument("--no-native-exchange", action="store_true", help="do not probe the library's own RCCL communicator, only torch.distributed")
    ap.add_argument("--no-direct-store", action="store_true", help="do not probe the direct-store corner turn (HIP IPC, no collective)")
    ap.add_argument("--budget-s", type=float, default=1200.0,
                    help="N > 1: seconds this command may take in all (the driver allows 1800).  The legs behind the headline are started only "
                         "while the time used is below a fraction of it -- the other member of strong/weak 0.35, the call census 0.5, the prover "
                         "0.7 -- so they are dropped in that order, and the line says which were (config.legs); 0 = no limit")
    ap.add_argument("--allow-replicas", action="store_true",
                    help="N > 1: if the sharded path cannot be set up, time N independent single-GPU transforms instead of exiting non-zero")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))            # bare `python bench.py --gpus N`: become N ranks under torch.distributed.run

    # the host driver of these nodes only supports dmabuf IPC: without this RCCL and hipIpcGetMemHandle fail between processes.
    # The image exports it; a launcher that cleaned the environment must not change what the ranks can do (read at HSA start-up)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import numpy as np
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node == --gpus (or plain `python bench.py --gpus N`)"
    ngpu = torch.cuda.device_count()
    # One rank per GPU over RCCL is the production shape.  With fewer GPUs than ranks (functional runs on a 1-GPU box) the ranks
    # share devices and the collectives go through gloo, staged over the host: correct, labelled, and not a scaling measurement.
    shared_gpus = world > ngpu
    batch_default = world == 1 and not args.force_sharded and args.workload == "ntt" and args.columns != 1
    if args.steps is None:
        args.steps = 10 if shared_gpus else (200 if batch_default else 2000)
    if args.warmup is None:
        args.warmup = 2 if shared_gpus else (20 if batch_default else 200)
    dev_index = local_rank % ngpu
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    os.environ["STARKCORE_DEVICE"] = str(dev_index)
    import starkcore as sc
    import synth
    sc.init(dev_index)
    lib = sc.lib()
    for kv in args.tune:
        sc.set_tuning(kv.split("=")[0], int(kv.split("=")[1]))

    sharded = world > 1 or args.force_sharded or args.workload in ("stark_census", "stark_prove")

    # a dedicated (non-null) HIP stream: the library launches on it and the timing events are recorded on it
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    sptr = ctypes_void(stream.cuda_stream)
    assert stream.cuda_stream != 0

    backend = None
    dist = None
    replicas_reason = None
    if sharded:
        # N > 1: the path sharded over the ranks.  Safety net: if the process group cannot even be set up and warmed up on this
        # node (RCCL init, all-to-all), every rank falls back to independent single-GPU transforms of the same per-GPU size and the
        # JSON line says so ("replicas"); nothing is silently substituted.
        try:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            backend = "gloo" if shared_gpus else "nccl"
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            else:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            if args.workload == "ntt":
                if args.log2n:
                    log2n = args.log2n
                elif args.scaling == "strong" and world > 1:
                    log2n = 24                                                   # north_star: the same 2^24 for every N
                else:
                    log2n = 20 + (world.bit_length() - 1) + 1                   # 2^21 per GPU: 8 GPUs -> 2^24
                n = 1 << log2n
                step, eng, (x, y, z), corner_turn, corner_probes = sharded_setup(args, log2n, rank, world, dev, dist, backend)
                launches_per_step = 2      # N > 1: roofline is reported per whole transform (local passes + corner turn)
                workload = "ntt_fwd_inv_2^%d_fourstep_%dgpu" % (log2n, world)
                total_n = n
                parallelism = "four-step, column-sharded, 1 corner turn per transform"
        except Exception as e:       # noqa: BLE001
            replicas_reason = repr(e)[:300]
            if not args.allow_replicas:
                # a scaling run that silently measured N independent transforms would report a meaningless number with rc 0
                sys.stderr.write("bench.py: the sharded path could not be set up (%s); --allow-replicas would time independent replicas instead\n" % replicas_reason)
                sys.exit(4)
            sharded = False
            sys.stderr.write("bench.py: sharded path failed (%s); falling back to independent replicas (--allow-replicas)\n" % replicas_reason)

    if sharded and args.workload == "stark_census":
        return run_census_workload(args, rank, world, dev, stream, dist, backend, shared_gpus)
    if sharded and args.workload == "stark_prove":
        return run_stark_prove_workload(args, rank, world, dev, stream, dist, backend, shared_gpus)

    if not sharded:
        log2n = args.log2n or (20 if world == 1 else 21)
        n = 1 << log2n
        root = sc.fe_bytes(nth_root(n))
        cols = args.columns if args.columns else (max(1, (1 << 26) >> log2n) if world == 1 else 1)
        # column 0 is the vector of rounds 1-5 (synth seed 1: the reference's own transform of it is pinned in tests/golden/ntt_big.json)
        x = torch.empty(2 * n * cols, dtype=torch.int64, device=dev)
        for c in range(cols):        # (a column at a time: the host never holds more than one)
            x[2 * n * c:2 * n * (c + 1)] = torch.from_numpy(synth.synth_packed(1, n, start=c * n).view(np.int64).reshape(-1)).to(dev)
        y = torch.empty_like(x)
        z = torch.empty_like(x)

        if cols > 1:
            def step():
                sc._check(lib.sc_ntt_columns_dev(x.data_ptr(), y.data_ptr(), n, cols, root, 0, sptr))
                sc._check(lib.sc_ntt_columns_dev(y.data_ptr(), z.data_ptr(), n, cols, root, 1, sptr))
            launches_per_step = 2 * int(lib.sc_ntt_num_passes(n)) * column_launch_sets(n, cols)
        else:
            def step():
                sc._check(lib.sc_ntt_dev(x.data_ptr(), y.data_ptr(), n, root, 0, sptr))
                sc._check(lib.sc_ntt_dev(y.data_ptr(), z.data_ptr(), n, root, 1, sptr))
            launches_per_step = 2 * int(lib.sc_ntt_num_passes(n))

        def one_column_pair():
            sc._check(lib.sc_ntt_dev(x.data_ptr(), y.data_ptr(), n, root, 0, sptr))
            sc._check(lib.sc_ntt_dev(y.data_ptr(), z.data_ptr(), n, root, 1, sptr))

        if world == 1:
            workload = "ntt_fwd_inv_2^%d_1gpu" % log2n
            parallelism = "single"
        else:
            workload = "ntt_fwd_inv_2^%d_x%d_independent_replicas" % (log2n, world)
            parallelism = "replicas (sharded path failed: %s); value = n_gpus x rank-0 rate" % replicas_reason
        total_n = n * cols * world

    def barrier():
        if sharded:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_window():
        """the contract's measurement: W untimed steps, then EXACTLY K steps between barrier + synchronize on both sides;
        (seconds on the host clock, max over ranks; milliseconds between HIP events on the launch stream)"""
        for _ in range(args.warmup):
            step()
        barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(stream)
        for _ in range(args.steps):
            step()
        e1.record(stream)
        barrier()
        dt = time.perf_counter() - t0
        if sharded:
            t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, e0.elapsed_time(e1)

    # The board idles at a low clock and needs tens of milliseconds of load to reach its sustained one (profiles/r02/ramp_probe.txt:
    # 79-82 us per 2^20 pair for the first ~5 ms after idle, 69 us from then on), so a short window straight after start-up
    # measures the clock ramp as much as the transform.  `value` is the contract's window, run first, straight after start-up;
    # then CLOCK_RAMP_MS of the same untimed steps and the same window once more, reported beside it as
    # "clock_ramp.steady_state" (information only).  With the default 200 + 2000 steps the two agree to ~1 %.
    elapsed, ev_ms = timed_window()
    ramp_steps = max(1, min(20000, int(math.ceil(CLOCK_RAMP_MS * 1e-3 / (elapsed / args.steps)))))
    for _ in range(ramp_steps):
        step()
    steady_elapsed, steady_ev_ms = timed_window()

    # correctness guard inside the bench: the round trip must reproduce the input bit for bit
    ok = bool(torch.equal(z, x))
    one_column = None
    if not sharded and cols > 1 and not args.no_extras:
        # (a side leg like `extras`: --no-extras, which the profiling commands use, keeps every launch of the run a launch of the batch)
        # the same kernels with the columns issued one after the other (sc_ntt_dev on column 0: the step of rounds 1-5), at the clock
        # the windows above have brought the board to
        pairs = max(200, min(4000, int(0.1 / (elapsed / args.steps / cols))))
        for _ in range(50):
            one_column_pair()
        barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(stream)
        for _ in range(pairs):
            one_column_pair()
        e1.record(stream)
        barrier()
        dt1 = time.perf_counter() - t0
        lps1 = 2 * int(lib.sc_ntt_num_passes(n))
        one_column = {"value": 2.0 * n * pairs / dt1, "unit": "field-elements/s", "us_per_pair": 1e6 * dt1 / pairs, "pairs_timed": pairs,
                      "avg_launch_us": e0.elapsed_time(e1) * 1e3 / (pairs * lps1),
                      "roofline_frac": BYTES_PER_ELEMENT_PER_TRANSFORM * n / (lps1 // 2) / (e0.elapsed_time(e1) * 1e-3 / (pairs * lps1)) / 1e9 / HBM_PEAK_GBS,
                      "roundtrip_bit_exact": bool(torch.equal(z[:2 * n], x[:2 * n])),
                      "note": "sc_ntt_dev, one 2^%d forward + inverse after the other on one stream (what `value` was in rounds 1-5; --columns 1 times it as the step)" % log2n}
        step()           # leave y and z as the batch left them (the checks below read them)
        barrier()
    if sharded and args.workload == "ntt":
        direct = bool(corner_probes["chosen_kwargs"].get("direct_store"))
        if direct:
            ok = ok and eng.stages.direct_timed_out() == 0
            if os.environ.get("BENCH_INJECT_DIRECT_STORE_FAULT") == "1":          # tests: the path below
                ok = False
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok = int(flag.item()) == 1
        if not ok and direct:
            # The direct-store corner turn passed its probe and then failed in the timed run (peers' stores into this GPU's memory
            # not seen in time, a flag barrier that gave up): it has never run between two physical GPUs before the first SCALE run.
            # The collective forms have; measure again with them instead of reporting nothing.
            if rank == 0:
                sys.stderr.write("bench.py: the direct-store corner turn FAILED in the timed run; measuring again without it\n")
            discarded = corner_probes
            dist.barrier()
            eng.stages.release_direct()
            del step, eng, x, y, z
            args.no_direct_store = True
            step, eng, (x, y, z), corner_turn, corner_probes = sharded_setup(args, log2n, rank, world, dev, dist, backend)
            corner_probes["discarded_after_timed_run"] = discarded["chosen"]
            corner_probes["probes"] = [q for q in discarded["probes"] if "direct store" in q["form"]] + corner_probes["probes"]
            elapsed, ev_ms = timed_window()
            for _ in range(ramp_steps):
                step()
            steady_elapsed, steady_ev_ms = timed_window()
            flag = torch.tensor([1 if torch.equal(z, x) else 0], dtype=torch.int32, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = int(flag.item()) == 1
    # ... and the forward transform of the timed workload is the REFERENCE's: tests/golden/ntt_big.json holds the SHA-256 of
    # code/ntt.py's own output for this very input (synth seed 1, n = 2^20, Field.primitive_nth_root), generated by
    # tests/golden/make_golden.py from the imported reference (BASELINE configs[1]: "bit-exact vs code/ntt.py")
    reference_sha = None
    if not sharded and world == 1:
        try:
            import hashlib
            gold = json.load(open(os.path.join(REPO, "tests", "golden", "ntt_big.json")))
            want = [r["sha256"] for r in gold["ntt"] if r["logn"] == log2n and r["seed"] == 1 and int(r["root"]) == nth_root(n)]
            if want:
                reference_sha = hashlib.sha256(y[:2 * n].cpu().numpy().tobytes()).hexdigest() == want[0]       # column 0
                ok = ok and reference_sha
        except Exception:       # noqa: BLE001  (no fixture: the round trip stays the guard)
            reference_sha = None

    legs = {"seconds": {"set_up_probes_and_headline": round(time.perf_counter() - T0, 2)}, "dropped_for_the_budget": [], "budget_s": args.budget_s}

    def leg(name, fraction):
        """may the optional leg `name` start?  Decided on the slowest rank's clock so that every rank takes the same branch (the
        legs are collective); leg_done(name) books its duration"""
        used = time.perf_counter() - T0
        if sharded and world > 1:
            t = torch.tensor([used], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            used = float(t.item())
        ok = args.budget_s <= 0 or used < fraction * args.budget_s
        if not ok:
            legs["dropped_for_the_budget"].append(name)
        legs["_t"] = time.perf_counter()
        return ok

    def leg_done(name):
        legs["seconds"][name] = round(time.perf_counter() - legs.pop("_t"), 2)

    stages = weak = None
    if sharded and args.workload == "ntt":
        # where the time of the sharded transform goes (per stage, max over ranks), and the OTHER member of the pair strong / weak:
        # the first multi-GPU run should need no second run to be read
        leg("stage_breakdown", float("inf"))          # (always runs; booked like the others)
        try:
            stages = stage_breakdown(eng, (x, y, z), rank, world, dev, dist, backend, reps=3 if shared_gpus else 10)
        except Exception as e:       # noqa: BLE001
            stages = {"error": repr(e)[:300]}
        leg_done("stage_breakdown")
        try:
            scaling_is_strong = args.scaling == "strong" and world > 1
            other_log2n = (20 + (world.bit_length() - 1) + 1) if scaling_is_strong else 24
            if world > 1 and not args.log2n and other_log2n != log2n and leg("other_member_of_strong_weak", 0.35):
                other_steps = 5 if shared_gpus else 50
                step2, eng2, xyz2, _, _ = sharded_setup(args, other_log2n, rank, world, dev, dist, backend, only=(corner_probes["chosen"], corner_probes["chosen_kwargs"]))
                for _ in range(2 if shared_gpus else 5):
                    step2()
                barrier()
                t0 = time.perf_counter()
                for _ in range(other_steps):
                    step2()
                barrier()
                t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                sec = float(t.item()) / other_steps
                weak = strong_record(other_log2n, world, sec, bool(torch.equal(xyz2[2], xyz2[0])), corner_probes["chosen"],
                                     2 * ((1 << other_log2n) // world) * 16 * (world - 1) // world,
                                     {"scaling": "weak (2^21 elements per GPU)" if scaling_is_strong else "strong (2^24 in total)", "steps": other_steps,
                                      "stages_us": stage_breakdown(eng2, xyz2, rank, world, dev, dist, backend, reps=3 if shared_gpus else 10)})
                if getattr(eng2.stages, "direct", False):
                    dist.barrier()
                    eng2.stages.release_direct()
                del step2, eng2, xyz2
                leg_done("other_member_of_strong_weak")
        except Exception as e:       # noqa: BLE001
            weak = {"error": repr(e)[:300]}

    census = prover = None
    if sharded and world > 1 and not args.no_extras:
        # the whole config-5 pipeline on the same ranks, once warm and once timed (all ranks take part; rank 0 reports)
        if leg("stark_census_sharded", 0.5):
            try:
                lf = 16 if shared_gpus else (args.log2n or (20 + (world.bit_length() - 1) + 1))
                sharded_census(lf, rank, world, dev, stream)
                times, info = sharded_census(lf, rank, world, dev, stream)
                census = census_record(times, info, lf, world, dist, backend, dev)
            except Exception as e:       # noqa: BLE001  side measurements never invalidate the headline
                census = {"error": repr(e)[:300]}
            leg_done("stark_census_sharded")
        if leg("stark_prove_sharded", 0.7):
            try:
                # ... and configs[4] as a prover at its stated size: ShardedFastStark.prove on the synthetic AIR, FRI domain 2^24 sharded
                # over the ranks (2^14 when the ranks share GPUs: a functional run); the reference proves this workload byte for byte
                # at 2^10 ... 2^16 (tests/golden/fast_stark_synth.json)
                _, _, prover = stark_prove_measure(14 if shared_gpus else 24, 2, 1, rank, world, dev, dist, backend)
            except Exception as e:       # noqa: BLE001
                prover = {"error": repr(e)[:300]}
            leg_done("stark_prove_sharded")
    legs.pop("_t", None)
    legs["seconds"]["whole_command_so_far"] = round(time.perf_counter() - T0, 2)

    # what N means for the work: the N > 1 default is the north_star's strong-scaling series (2^24 for every N; its N = 1 member is
    # extras.ntt_2p24_strong of the N = 1 run, whose headline stays BASELINE configs[1] = 2^20); --scaling weak / replicas: work per GPU fixed
    if not sharded or world == 1:
        scaling_label = "weak"
    else:
        scaling_label = args.scaling
    if rank == 0:
        value = 2.0 * total_n * args.steps / elapsed
        ms_per_step = 1e3 * elapsed / args.steps
        # dominant kernel: ntt_pass_kernel (every launch in the timed region is one pass of it).
        # algorithmic bytes per launch = 32 B/element/transform * n elements / passes-per-transform
        passes = launches_per_step // 2 if sharded else int(lib.sc_ntt_num_passes(n))
        avg_launch_s = (ev_ms * 1e-3) / (args.steps * launches_per_step)
        if sharded:
            passes = None
            alg_bytes_per_launch = BYTES_PER_ELEMENT_PER_TRANSFORM * (total_n / world)     # per rank, per transform
        else:
            # (a launch of the batch covers every column of its set: per launch = 32 B x n x columns / passes, spread over the sets)
            alg_bytes_per_launch = BYTES_PER_ELEMENT_PER_TRANSFORM * (total_n / world) / (launches_per_step // 2)
        achieved = alg_bytes_per_launch / avg_launch_s / 1e9
        pmc_key = None if sharded else ("%d" % log2n if cols == 1 else "%dx%d" % (log2n, cols))     # profiles/rNN/traffic.json, pmc_summary.json
        out = {
            "metric": "ntt_field_elements_per_sec", "value": value, "unit": "field-elements/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": scaling_label, "vs_baseline": None, "dtype": "u128", "data": "synthetic",
            "config": {"workload": workload, "log2n": log2n, "columns_per_step": (cols if not sharded else 1), "elements_per_step": 2 * total_n, "parallelism": parallelism,
                       "passes_per_transform": passes, "roundtrip_bit_exact": ok, "forward_sha256_equals_reference_output": reference_sha,
                       **({"tune": args.tune} if args.tune else {})},
            "clock_ramp": {"untimed_steps_between_windows": ramp_steps, "target_ms": CLOCK_RAMP_MS,
                           "steady_state": {"value": 2.0 * total_n * args.steps / steady_elapsed, "ms_per_step": 1e3 * steady_elapsed / args.steps,
                                            "avg_launch_us": steady_ev_ms * 1e3 / (args.steps * launches_per_step),
                                            "roofline_frac": alg_bytes_per_launch / ((steady_ev_ms * 1e-3) / (args.steps * launches_per_step)) / 1e9 / HBM_PEAK_GBS},
                           "note": "information only: the same W + K window repeated after the board has clocked up (`value` is the first window, straight after start-up)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": measured_traffic(pmc_key) if not sharded else None, "kernel": "ntt_pass_kernel" if not sharded else "whole sharded transform (per rank)", "avg_launch_us": avg_launch_s * 1e6,
                         "alg_bytes_per_launch": alg_bytes_per_launch,
                         "valu_insts_per_launch": measured_valu(pmc_key) if not sharded else None,
                         "pmc_collected_with_these_kernel_sources": pmc_figures_are_current(),
                         "note": "traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024 and valu_insts (SQ_INSTS_VALU, wave-level) per launch from profiles/ (PMC passes); "
                                 "the kernel is VALU-bound: valu_insts / 1024 SIMDs x ~4.2 cycles at the 2.1 GHz the board holds under this load is its issue time -- "
                                 "12.3 us per 2^20 column of a launch over 64 columns (eight elements per thread, two workgroups per CU: 12.0 measured); "
                                 "12.8 us for the lone column's kernel, which takes 17.4: one "
                                 "workgroup per CU with every CU in the same phase), see DESIGN.md 3.1"},
        }
        if sharded:
            # the N > 1 line prices a rank's WHOLE transform (column stage, corner turn, row stage), not one launch of one kernel: no
            # per-launch PMC figures belong here (they are the single-GPU kernel's, in the N = 1 line and profiles/); what bounds this
            # number is in stages_us -- the two compute stages run the same ntt_pass_kernel at its VALU issue rate, the rest is the
            # exchange and waiting for the slowest peer
            del out["roofline"]["traffic"], out["roofline"]["valu_insts_per_launch"]
            out["roofline"]["traffic"] = None
            out["roofline"]["note"] = ("per rank and transform: 32 B/element algorithmic over the whole sharded transform (cols + corner turn + rows); the stages' own times are "
                                       "in stages_us (compute stages = ntt_pass_kernel launches, priced per launch in the N = 1 line; the exchange moves "
                                       "all_to_all_bytes_sent_per_rank_per_step / 2 bytes per transform over xGMI)")
            out["config"]["collective_backend"] = collective_label(backend, world, ngpu, shared_gpus)
            out["config"]["legs"] = legs
            out["config"]["world_size"] = world
            out["config"]["corner_turn"] = corner_turn
            out["config"]["corner_turn_probes"] = corner_probes
            out["config"]["corner_turn_setup"] = list(getattr(eng, "corner_turn_setup", []))
            out["config"]["split"] = "n1 = 2^%d x n2 = 2^%d" % (eng.n1.bit_length() - 1, eng.n2.bit_length() - 1)
            out["config"]["node"] = node_facts(dev)
            out["roofline"]["stages_us"] = stages
            if weak is not None:
                out.setdefault("extras", {})["ntt_other_scaling"] = weak
            out["config"]["all_to_all_bytes_sent_per_rank_per_step"] = 2 * (total_n // world) * 16 * (world - 1) // world
            out["config"]["series"] = ("north_star strong-scaling series: forward + inverse 2^%d for every N; the N = 1 member is extras.ntt_2p24_strong of the "
                                       "N = 1 run (whose headline is BASELINE configs[1], 2^20)" % log2n) if scaling_label == "strong" else \
                ("2^%d elements per GPU (weak); the N = 1 headline is BASELINE configs[1], 2^20 on the single-GPU plan" % (log2n - (world.bit_length() - 1)))
            if log2n == 24:
                out.setdefault("extras", {})["ntt_2p24_strong"] = strong_record(24, world, elapsed / args.steps, ok, corner_turn, out["config"]["all_to_all_bytes_sent_per_rank_per_step"])
        if census is not None:
            out.setdefault("extras", {})["stark_census_sharded"] = census
        if prover is not None:
            out.setdefault("extras", {})["stark_prove_sharded"] = prover
        if not args.no_extras and not sharded and world == 1:
            try:
                out["extras"] = extras(sc, lib, stream)
            except Exception as e:       # side measurements never invalidate the headline
                out["extras"] = {"error": repr(e)}
        two = out.get("extras", {}).get("ntt_fwd_inv_2p20_two_columns_two_streams") if isinstance(out.get("extras"), dict) else None
        if two and "elements_per_s" in two:
            # beside the headline (one transform at a time, BASELINE configs[1]): what the same kernels carry with two independent
            # columns in flight on two HIP streams
            out["two_columns_in_flight"] = {"value": two["elements_per_s"], "unit": "field-elements/s", "frac_of_hbm_roofline": two["frac"],
                                            "gain_over_one_stream": two["gain"], "roundtrip_bit_exact": two["roundtrip_bit_exact"],
                                            "note": "NOT the headline: two independent 2^20 forward+inverse transforms on two HIP streams; details in extras"}
        if one_column is not None:
            out["one_column_at_a_time"] = one_column
        if not args.no_cpu_baseline and not sharded and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample_log2n)
        emit(out)
    if sharded:
        try:
            from sharded import destroy_native_comm
            destroy_native_comm()
        except Exception:      # noqa: BLE001
            pass
        dist.destroy_process_group()
    elif world > 1:
        try:
            import torch.distributed as dist
            if dist.is_initialized():
                dist.destroy_process_group()
        except Exception:      # noqa: BLE001
            pass
    if not ok:
        sys.exit("round trip mismatch")


def run_census_workload(args, rank, world, dev, stream, dist, backend, shared_gpus):
    """--workload stark_census: one step = the whole sharded census; value = ms per census (max over ranks)."""
    import torch
    ngpu = torch.cuda.device_count()
    log_fri = args.log2n or (16 if shared_gpus else 20 + (world.bit_length() - 1) + 1)
    steps, warmup = min(args.steps, 20), max(1, min(args.warmup, 2))
    for _ in range(warmup):
        sharded_census(log_fri, rank, world, dev, stream)
    dist.barrier()
    torch.cuda.synchronize()
    best, totals = None, []
    for _ in range(steps):
        # one step = the census from the first LDE to the last opening; synthesising and uploading the inputs (host numpy) is
        # set-up and is not part of the step
        times, info = sharded_census(log_fri, rank, world, dev, stream)
        totals.append(times["total"])
        if best is None or times["total"] < best[0]["total"]:
            best = (times, info)
    dist.barrier()
    torch.cuda.synchronize()
    t = torch.tensor(totals, dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)              # per step: the slowest rank
    elapsed = float(t.sum().item())
    rec = census_record(best[0], best[1], log_fri, world, dist, backend, dev)
    if rank == 0:
        out = {"metric": "stark_census_ms", "value": 1e3 * elapsed / steps, "unit": "ms", "n_gpus": world, "steps": steps, "warmup": warmup,
               "ms_per_step": 1e3 * elapsed / steps, "higher_is_better": False, "scaling": "weak", "vs_baseline": None, "dtype": "u128", "data": "synthetic",
               "config": {"workload": "faststark_call_census_fri_2^%d_sharded_%dgpu" % (log_fri, world), "log2n": log_fri, "world_size": world,
                          "collective_backend": collective_label(backend, world, ngpu, shared_gpus),
                          "parallelism": "four-step LDE (1 all-to-all each), slab-local folds, sharded Merkle (1 all-gather per commit)"},
               "stages_best_run": rec}
        emit(out)
    dist.destroy_process_group()


def run_stark_prove_workload(args, rank, world, dev, stream, dist, backend, shared_gpus):
    """--workload stark_prove: one step = one ShardedFastStark.prove from the trace to the serialized proof (FRI domain 2^log2n,
    default 2^24 = BASELINE configs[4]; 2^16 when the ranks share GPUs); value = ms per proof (max over ranks)."""
    import torch
    ngpu = torch.cuda.device_count()
    log_fri = args.log2n or (16 if shared_gpus else 24)
    steps, warmup = max(1, min(args.steps, 5)), max(1, min(args.warmup, 1))
    elapsed, same_everywhere, rec = stark_prove_measure(log_fri, steps, warmup, rank, world, dev, dist, backend, phases=True)
    if rank == 0:
        rec["collective_backend"] = collective_label(backend, world, ngpu, shared_gpus)
        out = {"metric": "stark_prove_ms", "value": 1e3 * elapsed / steps, "unit": "ms", "n_gpus": world, "steps": steps, "warmup": warmup,
               "ms_per_step": 1e3 * elapsed / steps, "higher_is_better": False, "scaling": "strong", "vs_baseline": None, "dtype": "u128", "data": "synthetic",
               "config": rec}
        emit(out)
    dist.destroy_process_group()
    if not same_everywhere:
        sys.exit("ranks disagree on the proof")


def extras(sc, lib, stream=None):
    """The other BASELINE.json configs, timed on the side (device-resident inputs).  Kernel-only quantities (LDE, NTT pairs) are
    timed with HIP events on the bench stream around a burst of back-to-back calls, best of 3 -- the same way the headline's
    launch duration is measured; host-driven ones (Fri.prove, census, trees) on the host clock.
    configs[2] LDE of 2^18 coefficients at blowup 8, configs[3] Fri.prove on a 2^22 codeword (ef 4, 40 checks)."""
    import ctypes
    import numpy as np
    import torch
    sptr = ctypes.c_void_p(stream.cuda_stream) if stream is not None else None

    def device_time(fn, reps):
        """seconds per call of fn (which launches on `stream`), HIP events, best of 3"""
        if stream is None:
            best = None
            for _ in range(3):
                sc.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    fn()
                sc.synchronize()
                dt = (time.perf_counter() - t0) / reps
                best = dt if best is None or dt < best else best
            return best
        fn()
        torch.cuda.synchronize()
        best = None
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(reps):
                fn()
            e1.record(stream)
            torch.cuda.synchronize()
            dt = e0.elapsed_time(e1) * 1e-3 / reps
            best = dt if best is None or dt < best else best
        return best

    import synth
    from algebra import Field
    from fri import Fri
    from ip import ProofStream
    GEN = 85408008396924667383611388730472331217
    field = Field.main()
    res = {}
    # configs[2]: fast_coset_evaluate, 2^18 coefficients -> 2^21 values
    m, order = 1 << 18, 1 << 21
    om = field.primitive_nth_root(order)
    coeffs = sc.DeviceVector.from_bytes(synth.synth_packed(5, m).tobytes())
    outv = sc.DeviceVector(order)
    best = device_time(lambda: sc._check(lib.sc_coset_evaluate_dev(coeffs.ptr, m, sc.fe_bytes(GEN), sc.fe_bytes(om.value), order, outv.ptr, sptr)), 50)
    res["lde_2p18_to_2p21"] = {"ms": best * 1e3, "alg_GBps": 16 * (m + order) / best / 1e9, "timing": "HIP events, 50 calls back to back, best of 3"}
    # ... and eight such columns in one set of launches (sc_coset_evaluate_columns_dev: the registers of a trace, fast_stark.py:100-104)
    try:
        k = 8
        coeffs8 = sc.DeviceVector.from_bytes(synth.synth_packed(5, m * k).tobytes())
        out8 = sc.DeviceVector(order * k)
        best8 = device_time(lambda: sc._check(lib.sc_coset_evaluate_columns_dev(coeffs8.ptr, m, k, sc.fe_bytes(GEN), sc.fe_bytes(om.value), order, out8.ptr, sptr)), 20)
        same = out8.to_bytes(0, order) == outv.to_bytes()
        res["lde_2p18_to_2p21_x8_columns"] = {"ms_per_column": best8 * 1e3 / k, "alg_GBps": 16 * (m + order) * k / best8 / 1e9, "column_0_equals_the_single_call": same}
        del coeffs8, out8
    except Exception as e:       # noqa: BLE001
        res["lde_2p18_to_2p21_x8_columns"] = {"error": repr(e)}
    # configs[3]: Fri.prove, N = 2^22
    N = 1 << 22
    om = field.primitive_nth_root(N)
    coeffs = sc.DeviceVector.from_bytes(synth.synth_packed(4002, N // 4).tobytes())
    cwv = sc.DeviceVector(N)
    sc._check(lib.sc_coset_evaluate_dev(coeffs.ptr, N // 4, sc.fe_bytes(GEN), sc.fe_bytes(om.value), N, cwv.ptr, None))
    sc.synchronize()
    fr = Fri(field.generator(), om, N, 4, 40)
    for _ in range(8):                   # untimed: tables, pool, and the board's clock (the first proofs after an idle second run ~10 % slow)
        fr.prove(sc.DeviceCodeword(cwv, field), ProofStream())
    best, runs = None, []
    for _ in range(16):                  # every run is listed: an occasional one pays a full
        cw = sc.DeviceCodeword(cwv, field)   # collection of this process's heap (torch, numpy) by CPython's collector
        ps = ProofStream()
        t0 = time.perf_counter()
        fr.prove(cw, ps)
        dt = time.perf_counter() - t0
        runs.append(round(dt * 1e3, 3))
        best = dt if best is None or dt < best else best
    t0 = time.perf_counter()
    serialized = ps.serialize()          # (not part of Fri.prove in the reference either: fri.py:115-130 returns the indices; ip.py:18 serializes)
    serialize_ms = (time.perf_counter() - t0) * 1e3
    # like for like with the reference's Fri.prove, which builds every tuple and path INSIDE the call (fri.py:98-113): here the call
    # describes the proof objects and serialize() materialises them, so prove + serialize is the figure that covers the same work
    both = []
    for _ in range(5):
        cw2, ps2 = sc.DeviceCodeword(cwv, field), ProofStream()
        t0 = time.perf_counter()
        fr.prove(cw2, ps2)
        ps2.serialize()
        both.append(round((time.perf_counter() - t0) * 1e3, 3))
    t0 = time.perf_counter()
    verified = fr.verify(ps, [])        # outside the timed loop: the proof that was timed is a proof the verifier accepts
    res["fri_prove_2p22_ef4_s40"] = {"ms": best * 1e3, "median_ms": sorted(runs)[len(runs) // 2], "rounds": fr.num_rounds(), "proof_objects": len(ps.objects),
                                     "verify_accepts": bool(verified), "verify_s": time.perf_counter() - t0, "runs_ms": runs,
                                     "proof_bytes": len(serialized), "serialize_ms_outside_the_timed_call": serialize_ms,
                                     "prove_plus_serialize_ms": {"best": min(both), "median": sorted(both)[len(both) // 2], "runs": both},
                                     "how": "one library call (sc_fri_prove_dev): commit phase with the rounds below 2^17 in one persistent launch (fri_tail_kernel), "
                                            "transcript challenge, index sampling, one query kernel writing the openings to pinned host memory"}
    del cw, cw2, cwv, coeffs
    # configs[4] on ONE GPU: the polynomial-core call census of FastStark.prove replayed at
    # fri_domain_length 2^24, omicron_domain_length 2^22, 2 registers: 4 LDEs to 2^24, 2 coset divisions at 2^22,
    # 3 Merkle commits of 2^24 leaves, Fri.prove on the combined codeword (17 rounds), 4 x 160 openings.
    try:
        runs = [stark_census(sc, lib, field, 24) for _ in range(2)]      # the first run also pays for mapping ~10 GB of fresh HBM
        res["stark_census_2p24_1gpu"] = min(runs, key=lambda r: r["ms"])
    except Exception as e:
        res["stark_census_2p24_1gpu"] = {"error": repr(e)}
    # the metric's other sizes (BASELINE.json: NTT elements/s at 2^22 and 2^24, forward + inverse), library stream, best of 3
    for lg in (22, 24):
        try:
            nn = 1 << lg
            rt = sc.fe_bytes(field.primitive_nth_root(nn).value)
            a = sc.DeviceVector.from_bytes(synth.synth_packed(1, nn).tobytes())
            b, c = sc.DeviceVector(nn), sc.DeviceVector(nn)
            def pair():
                sc._check(lib.sc_ntt_dev(a.ptr, b.ptr, nn, rt, 0, sptr))
                sc._check(lib.sc_ntt_dev(b.ptr, c.ptr, nn, rt, 1, sptr))

            best = device_time(pair, 40 if lg == 22 else 10)
            if stream is not None:
                torch.cuda.synchronize()
            same = c.to_bytes() == a.to_bytes()                      # the whole vector, not a sample
            res["ntt_fwd_inv_2p%d" % lg] = {"ms_per_pair": best * 1e3, "elements_per_s": 2 * nn / best, "roundtrip_bit_exact": same,
                                            "roundtrip_check": "all 2^%d elements" % lg,
                                            "alg_GBps": 2 * BYTES_PER_ELEMENT_PER_TRANSFORM * nn / best / 1e9,
                                            "frac": 2 * BYTES_PER_ELEMENT_PER_TRANSFORM * nn / best / 1e9 / HBM_PEAK_GBS,
                                            "passes_per_transform": int(lib.sc_ntt_num_passes(nn)),
                                            "timing": "HIP events, pairs back to back, best of 3"}
            if lg == 24:
                # the N = 1 member of the north_star series (forward + inverse 2^24 at 1/2/4/8 GPUs; the N > 1 members are the
                # headline of `bench.py --gpus N`)
                res["ntt_2p24_strong"] = strong_record(24, 1, best, same, "single GPU: three-pass plan, no exchange", 0,
                                                       {"timing": "HIP events, pairs back to back, best of 3"})
            del a, b, c
        except Exception as e:
            res["ntt_fwd_inv_2p%d" % lg] = {"error": repr(e)}
    # Two INDEPENDENT columns side by side (the registers of a trace: fast_stark.py:103-105 transforms them one after the other), one HIP
    # stream each: a CU whose workgroup of one transform is done takes one of the other instead of waiting for the slowest workgroup
    # of its own kernel.  The headline stays one transform at a time (BASELINE configs[1]); this is the throughput a prover with two
    # columns in flight gets from the same kernels.  Host clock over 300 pairs, best of 3; every round trip compared.
    for lg in (20, 22):
        try:
            nn = 1 << lg
            rt = sc.fe_bytes(field.primitive_nth_root(nn).value)
            three = [torch.cuda.Stream() for _ in range(3)]
            cols = []
            for k in range(2):
                xk = torch.from_numpy(synth.synth_packed(1 + k, nn).view(np.int64)).cuda()
                cols.append((xk, torch.empty_like(xk), torch.empty_like(xk)))
            torch.cuda.synchronize()                       # (the uploads ran on the bench stream: done before another stream reads them)

            def pairs(count, streams_used):
                for i in range(count):
                    xk, yk, zk = cols[i & 1]
                    p = ctypes.c_void_p(streams_used[i % len(streams_used)].cuda_stream)
                    sc._check(lib.sc_ntt_dev(xk.data_ptr(), yk.data_ptr(), nn, rt, 0, p))
                    sc._check(lib.sc_ntt_dev(yk.data_ptr(), zk.data_ptr(), nn, rt, 1, p))

            def per_pair(streams_used, count, reps):
                pairs(40, streams_used)
                torch.cuda.synchronize()
                best = None
                for _ in range(reps):
                    t0 = time.perf_counter()
                    pairs(count, streams_used)
                    torch.cuda.synchronize()
                    dt = (time.perf_counter() - t0) / count
                    best = dt if best is None or dt < best else best
                return best

            # HIP maps its streams onto a handful of hardware queues, and two streams that share one run in turn (measured: of six
            # streams, three pairs do -- tools/stream_pair_probe.py): of three streams take the pair that overlaps
            probe = {(i, j): per_pair([three[i], three[j]], 60, 1) for i, j in ((0, 1), (0, 2), (1, 2))}
            chosen = min(probe, key=probe.get)
            rates = {1: per_pair([three[0]], 300, 3), 2: per_pair([three[chosen[0]], three[chosen[1]]], 300, 3)}
            same = all(bool(torch.equal(zk, xk)) for xk, yk, zk in cols)
            res["ntt_fwd_inv_2p%d_two_columns_two_streams" % lg] = {
                "ms_per_pair": rates[2] * 1e3, "elements_per_s": 2 * nn / rates[2], "roundtrip_bit_exact": same,
                "alg_GBps": 2 * BYTES_PER_ELEMENT_PER_TRANSFORM * nn / rates[2] / 1e9,
                "frac": 2 * BYTES_PER_ELEMENT_PER_TRANSFORM * nn / rates[2] / 1e9 / HBM_PEAK_GBS,
                "one_stream_same_loop": {"ms_per_pair": rates[1] * 1e3, "elements_per_s": 2 * nn / rates[1]},
                "gain": rates[1] / rates[2],
                "stream_pair_probe_us_per_pair": {"%d+%d" % k: round(v * 1e6, 2) for k, v in probe.items()},
                "what": "two independent columns, each forward + inverse on its own HIP stream (two transforms in flight); host clock over 300 pairs, best of 3"}
            del cols
        except Exception as e:
            res["ntt_fwd_inv_2p%d_two_columns_two_streams" % lg] = {"error": repr(e)}
    # Merkle.commit on 2^24 leaves (2^25 BLAKE2b compressions) and the subproduct tree of ntt.py:66-130 over 2^20 arbitrary points
    try:
        v = sc.DeviceVector.from_bytes(synth.synth_packed(9, 1 << 24).tobytes())
        best = None
        for _ in range(4):
            t0 = time.perf_counter()
            t = sc.MerkleTree.from_device(v)
            dt = time.perf_counter() - t0
            t.free()
            best = dt if best is None or dt < best else best
        res["merkle_commit_2p24"] = {"ms": best * 1e3, "gcompress_s": (2 ** 25 - 1) / best / 1e9}
        del v
        k = 1 << 20
        pts = sc.DeviceVector.from_bytes(synth.synth_packed(11, k).tobytes())
        f = sc.DeviceVector.from_bytes(synth.synth_packed(12, k).tobytes())

        def timed(fn, reps):
            b, r = None, None
            for _ in range(reps):
                sc.synchronize()
                t0 = time.perf_counter()
                r = fn()
                sc.synchronize()
                d = time.perf_counter() - t0
                b = d if b is None or d < b else b
            return b, r

        tb, tree = timed(lambda: sc.PolyTree(pts), 3)
        timed(lambda: tree.evaluate(f), 1)                       # builds the tree's power-series inverse (once per tree)
        te, vals = timed(lambda: tree.evaluate(f), 3)
        ti, back = timed(lambda: tree.interpolate(vals), 3)
        res["polytree_2p20_points"] = {"build_ms": tb * 1e3, "evaluate_ms": te * 1e3, "interpolate_ms": ti * 1e3,
                                       "round_trip_ok": back.to_bytes() == f.to_bytes()}
        tree.free()
        # the same three functions on a geometric progression (the trace domain {omicron^i} of fast_stark.py:84-90: 2^20 - 160
        # points, omicron of order 2^22): convolutions instead of a tree (csrc/geoseq.cuh)
        kg = (1 << 20) - 160
        omicron = field.primitive_nth_root(1 << 22).value
        tg, dom = timed(lambda: sc.GeoDomain(1, omicron, kg), 2)
        vals = sc.DeviceVector.from_bytes(synth.synth_packed(13, kg).tobytes())
        ti, poly = timed(lambda: dom.interpolate(vals), 5)
        te, back = timed(lambda: dom.evaluate(poly), 5)
        res["progression_2p20_points"] = {"tables_ms": tg * 1e3, "evaluate_ms": te * 1e3, "interpolate_ms": ti * 1e3, "points": kg,
                                          "round_trip_ok": back.to_bytes() == vals.to_bytes()}
        dom.free()
    except Exception as e:
        res["merkle_polytree"] = {"error": repr(e)}
    # configs[4] as a real prover on ONE GPU: fast_stark.FastStark.prove (reference code/fast_stark.py:76-178) on the synthetic
    # 2-register AIR, 2^20-row randomized trace resident in HBM, FRI domain 2^24; verified outside the timed region
    try:
        res["stark_prove_2p24_1gpu"] = plain_stark_prove_measure(24, 8)
    except Exception as e:
        res["stark_prove_2p24_1gpu"] = {"error": repr(e)}
    return res


def ctypes_void(v):
    import ctypes
    return ctypes.c_void_p(v)


if __name__ == "__main__":
    main()
