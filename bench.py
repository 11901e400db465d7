#!/usr/bin/env python3
"""bench.py -- NTT field-elements/s on MI355X (BASELINE.json metric), one JSON line on stdout.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

N = 1: workload = BASELINE configs[1]: forward + inverse 2^20-point NTT, data resident in HBM.
       One step = one forward + one inverse transform;  value = 2 * n * K / elapsed  (field elements / s).
       `extras.ntt_2p24_strong` carries the N = 1 member of the north_star series (forward + inverse at 2^24, with `frac`).
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

REPO = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(REPO, "stark-anatomy_amd")
for p in (PKG, REPO):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BYTES_PER_ELEMENT_PER_TRANSFORM = 32   # read once + write once, 16-byte elements
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
           "sample": "pure-Python port of code/ntt.py ntt+intt at n=2^%d, %.1f s, host has %d cores" % (sample_log2n, dt, os.cpu_count())}
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


def nth_root(n):
    import synth
    r, order = GEN, 1 << 119
    while order != n:
        r, order = r * r % synth.P, order >> 1
    return r


GEN = 85408008396924667383611388730472331217


def sharded_census(log_fri, rank, world, dev, stream, group=None, checks=40):
    """BASELINE configs[4]: the polynomial-core call census of FastStark.prove (reference code/fast_stark.py:101-151; SURVEY.md
    8(d)) replayed on the SHARDED layout at fri_domain_length 2^log_fri, omicron_domain_length 2^(log_fri-2), 2 registers:
    4 LDEs to 2^log_fri (ShardedNtt.coset_evaluate: one all-to-all each) + 3 sharded Merkle commits, 2 sharded coset divisions
    at 2^(log_fri-2) (3 all-to-alls each), ShardedFri.prove on the last codeword (no element exchange), 4 * checks openings on
    each of the three committed codewords.  Returns per-stage seconds of this rank and the bytes it sent."""
    import numpy as np
    import torch
    import synth
    from algebra import Field
    from fri import Fri
    from ip import ProofStream
    from sharded import ShardedNtt, ShardedFri
    field = Field.main()
    Nf, No = 1 << log_fri, 1 << (log_fri - 2)
    omega, omicron = field.primitive_nth_root(Nf), field.primitive_nth_root(No)
    ntt_f = ShardedNtt(log_fri, omega.value, rank, world, dev, group=group)
    ntt_o = ShardedNtt(log_fri - 2, omicron.value, rank, world, dev, group=group)
    polys = [torch.from_numpy(synth.synth_packed(60 + i, No // 2).view(np.int64)).to(dev) for i in range(4)]
    fr = Fri(field.generator(), omega, Nf, 4, checks)
    sfri = ShardedFri(fr, ntt_f.n1, rank, world, dev, group=group)
    C = ntt_f.n2

    def sync():
        torch.cuda.synchronize()

    sync()
    times = {}
    t0 = time.perf_counter()
    ps = ProofStream()
    slabs, layers = [], []
    for i, pv in enumerate(polys):                       # 2 boundary quotients, randomizer, combination
        slab = torch.empty(ntt_f.local_shape(False), dtype=torch.int64, device=dev)
        ntt_f.coset_evaluate(pv, GEN, slab)
        slabs.append(slab)
        if i < 3:
            sync()
            layers.append(sfri.commit(slab, C))
            ps.push(layers[-1]["root"])
    sync()
    times["lde_and_commit"] = time.perf_counter() - t0
    t1 = time.perf_counter()
    den = ntt_o.slab_of(polys[3][:No // 4], "census_den").clone()
    q = torch.empty(ntt_o.local_shape(True), dtype=torch.int64, device=dev)
    for i in range(2):                                   # 2 transition quotients (fast_stark.py:113)
        num = ntt_o.slab_of(polys[i], "census_num")
        ntt_o.coset_divide(num, den, GEN, q)
    sync()
    times["coset_divide"] = time.perf_counter() - t1
    t2 = time.perf_counter()
    indices = sfri.prove(slabs[3], ps)
    sync()
    times["fri_prove"] = time.perf_counter() - t2
    t3 = time.perf_counter()
    dup = [i for i in indices] + [(i + 4) % Nf for i in indices]
    quad = sorted(dup + [(i + Nf // 2) % Nf for i in dup])
    for layer in layers:
        entries, paths = sfri._open(layer, quad)
        for e, pth in zip(entries, paths):
            ps.push(e)
            ps.push(pth)
    times["openings"] = time.perf_counter() - t3
    times["total"] = time.perf_counter() - t0
    info = {"fri_rounds": fr.num_rounds(), "proof_objects": len(ps.objects), "proof_sha256_16": __import__("hashlib").sha256(ps.serialize()).hexdigest()[:16],
            "all_to_all_bytes_sent_per_rank": ntt_f.bytes_exchanged + ntt_o.bytes_exchanged, "roots": [l["root"].hex()[:16] for l in layers]}
    return times, info


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default 2000 (10 for a functional run whose ranks share GPUs)")
    ap.add_argument("--warmup", type=int, default=None, help="default 200 (2 for a functional run whose ranks share GPUs)")
    ap.add_argument("--workload", choices=("ntt", "stark_census", "stark_prove"), default="ntt",
                    help="ntt (headline, BASELINE configs[1]; N > 1: the sharded four-step transform), stark_census (the polynomial-core call census of "
                         "BASELINE configs[4] on the sharded layout) or stark_prove (sharded_stark.ShardedFastStark.prove on a synthetic AIR: configs[4] as a prover)")
    ap.add_argument("--log2n", type=int, default=None, help="override the transform size (ntt) / the FRI domain (stark_census)")
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
    if args.steps is None:
        args.steps = 10 if shared_gpus else 2000
    if args.warmup is None:
        args.warmup = 2 if shared_gpus else 200
    dev_index = local_rank % ngpu
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    os.environ["STARKCORE_DEVICE"] = str(dev_index)
    import starkcore as sc
    import synth
    sc.init(dev_index)
    lib = sc.lib()

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
        host = synth.synth_packed(1, n)
        x = torch.from_numpy(host.view(np.int64)).to(dev)
        y = torch.empty_like(x)
        z = torch.empty_like(x)

        def step():
            sc._check(lib.sc_ntt_dev(x.data_ptr(), y.data_ptr(), n, root, 0, sptr))
            sc._check(lib.sc_ntt_dev(y.data_ptr(), z.data_ptr(), n, root, 1, sptr))

        launches_per_step = 2 * int(lib.sc_ntt_num_passes(n))
        if world == 1:
            workload = "ntt_fwd_inv_2^%d_1gpu" % log2n
            parallelism = "single"
        else:
            workload = "ntt_fwd_inv_2^%d_x%d_independent_replicas" % (log2n, world)
            parallelism = "replicas (sharded path failed: %s); value = n_gpus x rank-0 rate" % replicas_reason
        total_n = n * world

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
                reference_sha = hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest() == want[0]
                ok = ok and reference_sha
        except Exception:       # noqa: BLE001  (no fixture: the round trip stays the guard)
            reference_sha = None

    stages = weak = None
    if sharded and args.workload == "ntt":
        # where the time of the sharded transform goes (per stage, max over ranks), and the OTHER member of the pair strong / weak:
        # the first multi-GPU run should need no second run to be read
        try:
            stages = stage_breakdown(eng, (x, y, z), rank, world, dev, dist, backend, reps=3 if shared_gpus else 10)
        except Exception as e:       # noqa: BLE001
            stages = {"error": repr(e)[:300]}
        try:
            scaling_is_strong = args.scaling == "strong" and world > 1
            other_log2n = (20 + (world.bit_length() - 1) + 1) if scaling_is_strong else 24
            if world > 1 and not args.log2n and other_log2n != log2n:
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
        except Exception as e:       # noqa: BLE001
            weak = {"error": repr(e)[:300]}

    census = prover = None
    if sharded and world > 1 and not args.no_extras:
        # the whole config-5 pipeline on the same ranks, once warm and once timed (all ranks take part; rank 0 reports)
        try:
            lf = 16 if shared_gpus else (args.log2n or (20 + (world.bit_length() - 1) + 1))
            sharded_census(lf, rank, world, dev, stream)
            times, info = sharded_census(lf, rank, world, dev, stream)
            census = census_record(times, info, lf, world, dist, backend, dev)
        except Exception as e:       # noqa: BLE001  side measurements never invalidate the headline
            census = {"error": repr(e)[:300]}
        try:
            # ... and configs[4] as a prover at its stated size: ShardedFastStark.prove on the synthetic AIR, FRI domain 2^24 sharded
            # over the ranks (2^14 when the ranks share GPUs: a functional run); the reference proves this workload byte for byte
            # at 2^10 ... 2^16 (tests/golden/fast_stark_synth.json)
            _, _, prover = stark_prove_measure(14 if shared_gpus else 24, 2, 1, rank, world, dev, dist, backend)
        except Exception as e:       # noqa: BLE001
            prover = {"error": repr(e)[:300]}

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
        passes = launches_per_step // 2
        avg_launch_s = (ev_ms * 1e-3) / (args.steps * launches_per_step)
        if sharded:
            passes = None
            alg_bytes_per_launch = BYTES_PER_ELEMENT_PER_TRANSFORM * (total_n / world)     # per rank, per transform
        else:
            alg_bytes_per_launch = BYTES_PER_ELEMENT_PER_TRANSFORM * (total_n / world) / passes
        achieved = alg_bytes_per_launch / avg_launch_s / 1e9
        out = {
            "metric": "ntt_field_elements_per_sec", "value": value, "unit": "field-elements/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": scaling_label, "vs_baseline": None, "dtype": "u128", "data": "synthetic",
            "config": {"workload": workload, "log2n": log2n, "elements_per_step": 2 * total_n, "parallelism": parallelism,
                       "passes_per_transform": passes, "roundtrip_bit_exact": ok, "forward_sha256_equals_reference_output": reference_sha},
            "clock_ramp": {"untimed_steps_between_windows": ramp_steps, "target_ms": CLOCK_RAMP_MS,
                           "steady_state": {"value": 2.0 * total_n * args.steps / steady_elapsed, "ms_per_step": 1e3 * steady_elapsed / args.steps,
                                            "avg_launch_us": steady_ev_ms * 1e3 / (args.steps * launches_per_step),
                                            "roofline_frac": alg_bytes_per_launch / ((steady_ev_ms * 1e-3) / (args.steps * launches_per_step)) / 1e9 / HBM_PEAK_GBS},
                           "note": "information only: the same W + K window repeated after the board has clocked up (`value` is the first window, straight after start-up)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": measured_traffic(log2n) if not sharded else None, "kernel": "ntt_pass_kernel" if not sharded else "whole sharded transform (per rank)", "avg_launch_us": avg_launch_s * 1e6,
                         "alg_bytes_per_launch": alg_bytes_per_launch,
                         "valu_insts_per_launch": measured_valu(log2n) if not sharded else None,
                         "pmc_collected_with_these_kernel_sources": pmc_figures_are_current(),
                         "note": "traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024 and valu_insts (SQ_INSTS_VALU, wave-level) per launch from profiles/ (PMC passes); "
                                 "the kernel is VALU-bound: valu_insts / 1024 SIMDs x ~4.2 cycles is its issue floor (13 of 17.5 us at 2^20), see DESIGN.md 3.1"},
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


def strong_record(log2n, world, seconds_per_pair, roundtrip_ok, corner_turn, bytes_sent_per_rank_per_pair, extra=None):
    """one member of the north_star series: forward + inverse 2^log2n at `world` GPUs, absolute and as a fraction of the HBM
    roofline (32 B per element per transform, over the N x 8 TB/s of the GPUs taking part)"""
    n = 1 << log2n
    rec = {"log2n": log2n, "n_gpus": world, "ms_per_pair": seconds_per_pair * 1e3, "elements_per_s": 2 * n / seconds_per_pair,
           "alg_GBps": 2 * BYTES_PER_ELEMENT_PER_TRANSFORM * n / seconds_per_pair / 1e9,
           "frac": 2 * BYTES_PER_ELEMENT_PER_TRANSFORM * n / seconds_per_pair / 1e9 / (HBM_PEAK_GBS * world),
           "roundtrip_bit_exact": bool(roundtrip_ok), "roundtrip_check": "all 2^%d elements" % log2n,
           "corner_turn": corner_turn, "bytes_sent_per_rank_per_pair": bytes_sent_per_rank_per_pair}
    if extra:
        rec.update(extra)
    return rec


_DIRECT_PREFLIGHT = None          # the job's one pre-flight of the direct-store corner turn: {"passed": bool, ...}


def direct_store_preflight(rank, world, dev, dist, backend):
    """The ingredients of the direct-store corner turn -- a HIP IPC region of one process opened in another, kernels of one GPU storing
    into another's memory -- tried by a CHILD of every rank first (stark-anatomy_amd/direct_preflight.py): between two physical GPUs they have
    never run, and what goes wrong there may be a GPU memory fault that ends the process instead of an error the library could
    return.  The ranks use the direct-store forms only if every child came back with status 0.  Once per job."""
    global _DIRECT_PREFLIGHT
    if _DIRECT_PREFLIGHT is not None:
        return _DIRECT_PREFLIGHT
    import shutil
    import subprocess
    import tempfile
    import torch
    on_dev = backend == "nccl"
    # the rendezvous directory: rank 0 makes it, its name travels as numbers (a plain tensor broadcast, like every other exchange here)
    made = tempfile.mkdtemp(prefix="starkcore_preflight_") if rank == 0 else ""
    name = torch.zeros(256, dtype=torch.int32)
    if rank == 0:
        raw = made.encode()
        assert len(raw) < 255, made
        name[0] = len(raw)
        name[1:1 + len(raw)] = torch.tensor(list(raw), dtype=torch.int32)
    name = name.to(dev) if on_dev else name
    dist.broadcast(name, 0)
    name = name.cpu().tolist()
    where = bytes(name[1:1 + name[0]]).decode()
    local = dev.index if on_dev and dev.index is not None else int(os.environ.get("LOCAL_RANK", "0"))
    if not on_dev:
        local = local % max(1, torch.cuda.device_count())
    t0 = time.perf_counter()
    try:
        child = subprocess.run([sys.executable, os.path.join(REPO, "stark-anatomy_amd", "direct_preflight.py"), str(rank), str(world), str(local), where],
                               capture_output=True, text=True, timeout=120, env=dict(os.environ, STARKCORE_NO_TORCH="1"))
        status, said = child.returncode, child.stderr.strip().splitlines()[-1:] if child.stderr.strip() else []
    except subprocess.TimeoutExpired:
        status, said = -1, ["no answer within 120 s"]
    t = torch.tensor([status == 0], dtype=torch.int32, device=dev if on_dev else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    passed = int(t.item()) == 1
    if status != 0:
        sys.stderr.write("bench.py: direct-store pre-flight, rank %d: status %d %s\n" % (rank, status, " ".join(said)))
    dist.barrier()
    if rank == 0:
        shutil.rmtree(where, ignore_errors=True)
    _DIRECT_PREFLIGHT = {"form": "direct-store pre-flight (a child of every rank exports, maps and stores across processes)", "passed": passed,
                         "this_rank_status": status, "seconds": round(time.perf_counter() - t0, 2)}
    return _DIRECT_PREFLIGHT


def sharded_setup(args, log2n, rank, world, dev, dist, backend, probe_steps=4, only=None):
    """The sharded transform of length 2^log2n ready to be timed: every form of the corner turn this job can run is built, its
    forward transform compared with the first form's element for element, its round trip checked, and timed for a few steps;
    the fastest correct one is returned as (step, engine, (x, y, z), description, probes).  The choice is the same on every
    rank: a form is dropped on ALL ranks as soon as any rank fails to build it or gets a wrong result (the ranks agree on a flag
    before the next collective), and the probe times are all-reduced.  only: a (label, kwargs) pair to build without probing."""
    import torch
    from sharded import ShardedNtt, init_native_comm
    n = 1 << log2n
    root = nth_root(n)
    on_dev = backend == "nccl"

    def agreed(flag):
        """True iff `flag` is true on every rank"""
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=dev if on_dev else "cpu")
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return int(t.item()) == 1

    forms = []                                    # (label, ShardedNtt kwargs)
    own = dict(always_exchange=True) if args.force_diag_exchange else {}
    if only is not None:
        forms = [only]
    elif world == 1 and not args.force_diag_exchange:
        forms.append(("one rank: nothing to exchange, the column stage writes the rank's own block in place", {}))
    else:
        if world > 1 and not args.force_diag_exchange:
            # the most conservative form first (it is the reference the others are compared with): one plain all_to_all_single
            # that carries the rank's own block as well
            forms.append(("torch.distributed all_to_all_single, own block included", dict(always_exchange=True)))
        forms.append(("torch.distributed, one blocking exchange", dict(own)))
        forms.append(("torch.distributed, 4 asynchronous row blocks overlapped with the row stage", dict(own, overlap_chunks=4)))
        native = False
        if on_dev and not args.no_native_exchange:
            try:
                native = init_native_comm(rank, world, dev)
            except Exception as e1:       # noqa: BLE001
                sys.stderr.write("bench.py: the library's RCCL communicator is unavailable (%r)\n" % (e1,))
            native = agreed(native)
        if native:
            forms.append(("library RCCL communicator, one exchange on the compute stream", dict(own, native_exchange=True)))
            forms.append(("library RCCL communicator, 2 row blocks on the communication stream overlapped with the row stage", dict(own, native_exchange=True, overlap_chunks=2)))
            forms.append(("library RCCL communicator, 4 row blocks on the communication stream overlapped with the row stage", dict(own, native_exchange=True, overlap_chunks=4)))
        preflight = None
        if not args.no_direct_store and world > 1:
            preflight = direct_store_preflight(rank, world, dev, dist, backend)
        if not args.no_direct_store and (preflight is None or preflight["passed"]):
            # no collective at all: the column stage stores block h straight into rank h's receive buffer (HIP IPC over xGMI),
            # a flag barrier, the row stage -- with the default split and, above 2^16, with the square one (fewer, longer rows)
            forms.append(("direct store: column stage writes into the peers' receive buffers (HIP IPC), flag barrier, no collective", dict(direct_store=True)))
            if log2n >= 20:
                forms.append(("direct store, square split n1 = 2^%d" % (log2n // 2), dict(direct_store=True, log_n1=log2n // 2)))
        if log2n >= 20 and native:
            forms.append(("library RCCL communicator, one exchange, square split n1 = 2^%d" % (log2n // 2), dict(own, native_exchange=True, log_n1=log2n // 2)))
    candidates, y_ref, probes = [], None, []
    if only is None and world > 1 and _DIRECT_PREFLIGHT is not None:
        probes.append(dict(_DIRECT_PREFLIGHT))
    for label, kw in forms:
        eng = x = y = z = None
        built = True
        try:
            eng = ShardedNtt(log2n, root, rank, world, dev, **kw)
            if kw.get("native_exchange"):
                eng.stages.native = True
            if kw.get("direct_store") and not eng.direct_store:
                raise RuntimeError("the peers' regions could not be mapped (%s)" % "; ".join(eng.corner_turn_setup))
            x = eng.synthetic_input(seed=1)
            y = torch.empty(eng.local_shape(False), dtype=torch.int64, device=dev)
            z = torch.empty_like(x)
        except Exception as e1:       # noqa: BLE001
            built = False
            sys.stderr.write("bench.py: corner turn form '%s' unavailable on rank %d (%r)\n" % (label, rank, e1))
        if not agreed(built):                      # nobody enters this form's collectives unless everybody can
            probes.append({"form": label, "available": False})
            continue

        def step(eng=eng, x=x, y=y, z=z):
            eng.forward(x, y)
            eng.inverse(y, z)

        step()
        dist.barrier()
        torch.cuda.synchronize()
        same = torch.equal(z, x)
        if kw.get("log_n1") and y_ref is not None:
            pass                                   # another split leaves another slab layout: the round trip is its check
        elif y_ref is None and not kw.get("log_n1"):
            y_ref = y.clone()
        elif y_ref is not None:
            same = same and torch.equal(y_ref, y)
        if kw.get("direct_store"):
            same = same and eng.stages.direct_timed_out() == 0
        if not agreed(same):
            sys.stderr.write("bench.py: corner turn form '%s' gave a WRONG result (rank %d: %s)\n" % (label, rank, "ok here" if same else "mismatch"))
            probes.append({"form": label, "available": True, "correct": False})
            if kw.get("direct_store"):
                eng.stages.release_direct()
            continue
        for _ in range(2):
            step()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(probe_steps):
            step()
        dist.barrier()
        torch.cuda.synchronize()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev if on_dev else "cpu")
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        sec = float(t.item()) / probe_steps
        # once more after the timed steps: a form whose hand-over only fails now and then must not be chosen either
        same = torch.equal(z, x) and (not kw.get("direct_store") or eng.stages.direct_timed_out() == 0)
        if not agreed(same):
            sys.stderr.write("bench.py: corner turn form '%s' gave a WRONG result after %d steps (rank %d: %s)\n" % (label, probe_steps + 3, rank, "ok here" if same else "mismatch"))
            probes.append({"form": label, "available": True, "correct": False, "failed_after_steps": probe_steps + 3})
            if kw.get("direct_store"):
                eng.stages.release_direct()
            continue
        candidates.append((sec, label, step, eng, (x, y, z), kw))
        probes.append({"form": label, "available": True, "correct": True, "ms_per_pair": sec * 1e3})
        if kw.get("direct_store"):
            probes[-1]["setup"] = list(eng.corner_turn_setup)       # which kind of region came up (fine-grained first, then coarse-grained)
        if kw.get("direct_store"):
            probes[-1]["receive_region_memory"] = eng.stages.region_kind()
    if not candidates:
        raise RuntimeError("no working corner turn")
    best = min(candidates, key=lambda c: c[0])         # the same choice on every rank (times are all-reduced)
    desc = best[1] + "; probe ms/step: " + ", ".join("[%s] %.3f" % (c[1], c[0] * 1e3) for c in candidates)
    for c in candidates:                               # the losers give their buffers (and mapped regions) back
        if c is not best and getattr(c[3].stages, "direct", False):
            dist.barrier()
            c[3].stages.release_direct()
    return best[2], best[3], best[4], desc, {"chosen": best[1], "chosen_kwargs": {k: v for k, v in best[5].items()}, "probes": probes}


def stage_breakdown(eng, xyz, rank, world, dev, dist, backend, reps=10):
    """Where a sharded transform's time goes, per direction: the column stage and the row stage of this rank timed ALONE with HIP
    events (local kernels, no exchange: the stage object's cols / rows on scratch buffers), the whole transform the same way,
    `exchange_and_waiting_us` = whole - cols - rows (the corner turn plus whatever the stages wait for: a derived figure -- in
    the direct-store form the column stage's own stores ARE the exchange, so it also holds the slower remote stores).  Max over
    ranks.  Bytes: what one rank sends to ONE peer per transform."""
    import torch
    x, y, z = xyz
    st = eng.stages
    G = world
    on_dev = backend == "nccl"
    stream = torch.cuda.current_stream(dev)
    out = {}
    if st is None:
        return out
    scratch_send = torch.empty((eng.n // G, 2), dtype=torch.int64, device=dev)
    scratch_recv = torch.empty((eng.n // G, 2), dtype=torch.int64, device=dev)

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if world > 1:
            dist.barrier()
        e0.record(stream)
        for _ in range(reps):
            fn()
        e1.record(stream)
        torch.cuda.synchronize()
        t = torch.tensor([e0.elapsed_time(e1) * 1e3 / reps], dtype=torch.float64, device=dev if on_dev else "cpu")
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    for name, inv, src, dst in (("forward", 0, x, y), ("inverse", 1, y, z)):
        cols = timed(lambda: eng._run(lambda: st.cols(inv, src, scratch_send, scratch_recv)))
        rows = timed(lambda: eng._run(lambda: st.rows(inv, scratch_recv, dst, 0, 1, False)))
        whole = timed((lambda: eng.forward(x, y)) if inv == 0 else (lambda: eng.inverse(y, z)))
        out[name] = {"cols_us": cols, "rows_us": rows, "whole_us": whole, "exchange_and_waiting_us": whole - cols - rows}
    # the timed stages have overwritten y / z with transforms of scratch data: restore the pair the caller checks
    eng.forward(x, y)
    eng.inverse(y, z)
    torch.cuda.synchronize()
    out["bytes_to_each_peer_per_transform"] = (eng.n // G // G) * 16 if G > 1 else 0
    out["messages_per_rank_per_transform"] = G - 1
    return out


def node_facts(dev):
    """what the first multi-GPU run should say about the node without a second run: RCCL / HIP versions, the peer-access matrix"""
    import torch
    facts = {}
    try:
        facts["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception as e:       # noqa: BLE001
        facts["rccl_version"] = repr(e)[:80]
    facts["hip_version"] = getattr(torch.version, "hip", None)
    n = torch.cuda.device_count()
    facts["visible_gpus"] = n
    try:
        facts["can_access_peer"] = [[bool(i == j or torch.cuda.can_device_access_peer(i, j)) for j in range(n)] for i in range(n)]
    except Exception as e:       # noqa: BLE001
        facts["can_access_peer"] = repr(e)[:80]
    try:
        facts["device_name"] = torch.cuda.get_device_name(dev)
    except Exception:            # noqa: BLE001
        pass
    return facts


def collective_label(backend, world, ngpu, shared_gpus):
    if backend == "nccl":
        return "nccl (RCCL), %d ranks on %d GPUs" % (world, ngpu)
    return "gloo, host-staged: %d ranks sharing %d GPU(s) -- functional run, NOT a scaling measurement" % (world, ngpu)


def census_record(times, info, log_fri, world, dist, backend, dev):
    """max over ranks of every stage time (ms) + what rank 0 saw"""
    import torch
    keys = sorted(times)
    t = torch.tensor([times[k] for k in keys], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    rec = {"log2_fri_domain": log_fri, "world_size": world}
    for k, v in zip(keys, t.tolist()):
        rec[k + "_ms"] = v * 1e3
    rec.update(info)
    return rec


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


def synthetic_stark_instance(log_fri, s=40):
    """The synthetic configs[4] workload: the 2-register AIR (a, b) -> (b, a*a + b) over a trace of T = 2^(log_fri - 4) - 4 s rows,
    so that the randomized trace has 2^(log_fri - 4) rows and the FRI domain 2^log_fri points (expansion factor 4).  Returns
    (field, T, packed columns (bytes per register), air, boundary): the columns are handed to the prover as a device-resident
    fast_stark.DeviceTrace -- a trace of 2^20 rows as the reference's list of lists is two million Python objects."""
    from algebra import Field, FieldElement
    from multivariate import MPolynomial
    k = log_fri - 4
    field = Field.main()
    p = field.p
    T = (1 << k) - 4 * s
    from synth import synthetic_air_columns
    col_a, col_b = synthetic_air_columns(T)
    pack = lambda col: b"".join(map(int.to_bytes, col, itertools.repeat(16), itertools.repeat("little")))
    v = MPolynomial.variables(5, field)                  # X, a, b, a', b'
    air = [v[3] - v[2], v[4] - v[1] * v[1] - v[2]]
    boundary = [(0, 0, FieldElement(col_a[0], field)), (0, 1, FieldElement(col_b[0], field)), (T - 1, 1, FieldElement(col_b[T - 1], field))]
    return field, T, [pack(col_a), pack(col_b)], air, boundary


def stark_prove_measure(log_fri, steps, warmup, rank, world, dev, dist, backend, phases=False):
    """sharded_stark.ShardedFastStark.prove (reference code/fast_stark.py:76-178) on the synthetic 2-register AIR (a, b) -> (b, a*a + b)
    with a 2^(log_fri - 4)-row randomized trace that is RESIDENT IN HBM as columns when the timed region starts: (seconds summed
    over `steps` proofs, max over ranks per proof; record for rank 0).  Every rank must end with the same proof; rank 0 verifies
    it with FastStark.verify outside the timed region.  phases: one more (untimed) proof with the per-phase breakdown."""
    import hashlib
    import torch
    from fast_stark import DeviceTrace
    from sharded_stark import ShardedFastStark
    s = 40
    field, T, packed, air, boundary = synthetic_stark_instance(log_fri, s)
    stark = ShardedFastStark(field, 4, s, 2 * s, 2, T, rank, world, dev)
    assert stark.fri_domain_length == 1 << log_fri
    trace = DeviceTrace.from_packed(packed, field)
    t0 = time.perf_counter()
    tz, layer, root = stark.preprocess(device_resident=True)
    torch.cuda.synchronize()
    preprocess_s = time.perf_counter() - t0
    for _ in range(warmup):
        stark.prove(trace, air, boundary, tz, layer)
    dist.barrier()
    torch.cuda.synchronize()
    totals, proof = [], None
    for _ in range(steps):
        t0 = time.perf_counter()
        proof = stark.prove(trace, air, boundary, tz, layer)
        torch.cuda.synchronize()
        totals.append(time.perf_counter() - t0)
    dist.barrier()
    t = torch.tensor(totals, dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)              # per step: the slowest rank
    elapsed = float(t.sum().item())
    digest = hashlib.sha256(proof).digest()
    mine = torch.tensor(list(digest[:8]), dtype=torch.int32, device=dev if backend == "nccl" else "cpu")
    lo, hi = mine.clone(), mine.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    same_everywhere = bool(torch.equal(lo, hi))
    phase_ms = None
    if phases:
        stark.phase_log = []
        stark.prove(trace, air, boundary, tz, layer)
        phase_ms = [[name, round(1e3 * sec, 3)] for name, sec in stark.phase_log]
        stark.phase_log = None
    rec = None
    if rank == 0:
        t0 = time.perf_counter()
        verifies = bool(stark.verify(proof, air, boundary, root))
        verify_s = time.perf_counter() - t0
        rec = {"workload": "faststark_prove_synthetic_air_trace_2^%d_fri_2^%d_%dgpu" % (log_fri - 4, log_fri, world), "log2n": log_fri, "world_size": world,
               "registers": 2, "colinearity_checks": s, "expansion_factor": 4, "ms_per_proof": 1e3 * elapsed / steps,
               "trace": "device-resident columns (fast_stark.DeviceTrace), generated on the host outside the timed region",
               "parallelism": "sharded LDEs (1 corner turn each), commitments, quotients, FRI and openings; trace interpolation one register per rank (broadcast); combination replicated",
               "proof_bytes": len(proof), "proof_sha256_16": digest.hex()[:16], "same_proof_on_every_rank": same_everywhere,
               "verify_accepts": verifies, "verify_s": verify_s, "preprocess_s": preprocess_s, "runs_ms": [round(x * 1e3, 3) for x in t.tolist()]}
        if phase_ms is not None:
            rec["phases_ms_synchronised_after_each"] = phase_ms
    return elapsed, same_everywhere, rec


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
    # configs[3]: Fri.prove, N = 2^22
    N = 1 << 22
    om = field.primitive_nth_root(N)
    coeffs = sc.DeviceVector.from_bytes(synth.synth_packed(4002, N // 4).tobytes())
    cwv = sc.DeviceVector(N)
    sc._check(lib.sc_coset_evaluate_dev(coeffs.ptr, N // 4, sc.fe_bytes(GEN), sc.fe_bytes(om.value), N, cwv.ptr, None))
    sc.synchronize()
    fr = Fri(field.generator(), om, N, 4, 40)
    best, runs = None, []
    for _ in range(16):                  # every run is listed: the first pays allocations and tables, an occasional one a full
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
    t0 = time.perf_counter()
    verified = fr.verify(ps, [])        # outside the timed loop: the proof that was timed is a proof the verifier accepts
    res["fri_prove_2p22_ef4_s40"] = {"ms": best * 1e3, "median_ms": sorted(runs)[len(runs) // 2], "rounds": fr.num_rounds(), "proof_objects": len(ps.objects),
                                     "verify_accepts": bool(verified), "verify_s": time.perf_counter() - t0, "runs_ms": runs,
                                     "proof_bytes": len(serialized), "serialize_ms_outside_the_timed_call": serialize_ms,
                                     "how": "one library call (sc_fri_prove_dev): commit phase with the rounds below 2^17 in one persistent launch (fri_tail_kernel), "
                                            "transcript challenge, index sampling, one query kernel writing the openings to pinned host memory"}
    del cw, cwv, coeffs
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
        res["stark_prove_2p24_1gpu"] = plain_stark_prove_measure(24, 3)
    except Exception as e:
        res["stark_prove_2p24_1gpu"] = {"error": repr(e)}
    return res


def plain_stark_prove_measure(log_fri, steps):
    """fast_stark.FastStark.prove on one GPU (no process group): ms per proof from a device-resident trace to the serialized proof"""
    from fast_stark import DeviceTrace, FastStark
    s = 40
    field, T, packed, air, boundary = synthetic_stark_instance(log_fri, s)
    stark = FastStark(field, 4, s, 2 * s, 2, T)
    trace = DeviceTrace.from_packed(packed, field)
    sc = sys.modules["starkcore"]
    t0 = time.perf_counter()
    tz, tz_codeword, root = stark.preprocess(device_resident=True)
    sc.synchronize()
    preprocess_s = time.perf_counter() - t0
    stark.prove(trace, air, boundary, tz, tz_codeword)
    runs, proof = [], None
    for _ in range(steps):
        sc.synchronize()
        t0 = time.perf_counter()
        proof = stark.prove(trace, air, boundary, tz, tz_codeword)
        sc.synchronize()
        runs.append(time.perf_counter() - t0)
    t0 = time.perf_counter()
    verifies = bool(stark.verify(proof, air, boundary, root))
    return {"workload": "faststark_prove_synthetic_air_trace_2^%d_fri_2^%d_1gpu" % (log_fri - 4, log_fri), "ms_per_proof": 1e3 * min(runs),
            "runs_ms": [round(1e3 * r, 3) for r in runs], "registers": 2, "colinearity_checks": s, "expansion_factor": 4,
            "trace": "device-resident columns (fast_stark.DeviceTrace)", "proof_bytes": len(proof), "verify_accepts": verifies,
            "verify_s": time.perf_counter() - t0, "preprocess_s": preprocess_s,
            "randomness": "the operating system's (getrandom, drawn by the library)" if sys.modules["fast_stark"].os_urandom_is_genuine() else "patched os.urandom"}


def stark_census(sc, lib, field, log_fri):
    import ctypes
    import synth
    from fri import Fri
    from ip import ProofStream
    GEN = 85408008396924667383611388730472331217
    Nf, No = 1 << log_fri, 1 << (log_fri - 2)
    omega, omicron = field.primitive_nth_root(Nf), field.primitive_nth_root(No)
    polys = [sc.DeviceVector.from_bytes(synth.synth_packed(60 + i, No // 2).tobytes()) for i in range(4)]
    sc.synchronize()
    t0 = time.perf_counter()
    ps = ProofStream()
    codewords = []
    for i, pv in enumerate(polys):                       # 2 boundary quotients, randomizer, combination
        cw = sc.DeviceVector(Nf)
        sc._check(lib.sc_coset_evaluate_dev(pv.ptr, No // 2, sc.fe_bytes(GEN), sc.fe_bytes(omega.value), Nf, cw.ptr, None))
        codewords.append(sc.DeviceCodeword(cw, field))
        if i < 3:
            ps.push(codewords[i].tree().root)
    t_lde_commit = time.perf_counter() - t0
    # 2 transition quotients: coset NTTs of numerator and zerofier, pointwise division, inverse NTT (device-resident core)
    t1 = time.perf_counter()
    a, b, q = sc.DeviceVector(No), sc.DeviceVector(No), sc.DeviceVector(No)
    for i in range(2):
        sc._check(lib.sc_coset_evaluate_dev(polys[i].ptr, No // 2, sc.fe_bytes(GEN), sc.fe_bytes(omicron.value), No, a.ptr, None))
        sc._check(lib.sc_coset_evaluate_dev(polys[3].ptr, No // 4, sc.fe_bytes(GEN), sc.fe_bytes(omicron.value), No, b.ptr, None))
        sc._check(lib.sc_pointwise_div_dev(a.ptr, b.ptr, q.ptr, No, None))
        sc._check(lib.sc_ntt_dev(q.ptr, a.ptr, No, sc.fe_bytes(omicron.value), 1, None))
    sc.synchronize()
    t_div = time.perf_counter() - t1
    t2 = time.perf_counter()
    fr = Fri(field.generator(), omega, Nf, 4, 40)
    indices = fr.prove(codewords[3], ps)
    t_fri = time.perf_counter() - t2
    t3 = time.perf_counter()
    dup = [i for i in indices] + [(i + 4) % Nf for i in indices]
    quad = sorted(dup + [(i + Nf // 2) % Nf for i in dup])
    for cw in codewords[:3]:
        entries, paths = cw.query(quad)
        for e, pth in zip(entries, paths):
            ps.push(e)
            ps.push(pth)
    t_open = time.perf_counter() - t3
    total = time.perf_counter() - t0
    import hashlib
    return {"ms": total * 1e3, "lde_and_commit_ms": t_lde_commit * 1e3, "coset_divide_ms": t_div * 1e3, "fri_prove_ms": t_fri * 1e3,
            "openings_ms": t_open * 1e3, "fri_rounds": fr.num_rounds(), "proof_objects": len(ps.objects),
            "proof_sha256_16": hashlib.sha256(ps.serialize()).hexdigest()[:16], "roots": [o.hex()[:16] for o in ps.objects[:3]]}


def measured_valu(log2n):
    """wave-level VALU instructions per ntt_pass_kernel launch from the committed PMC runs (profiles/*/pmc_summary.json), or None"""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(REPO, "profiles", "r*", "pmc_summary.json"))):
        try:
            d = json.load(open(f))
            for run, kernels in d.items():
                if not run.endswith("_%d" % log2n):
                    continue
                for name, ctrs in kernels.items():
                    if "ntt_pass" in name and "SQ_INSTS_VALU" in ctrs:
                        best = ctrs["SQ_INSTS_VALU"]["avg_per_dispatch"]
        except Exception:
            pass
    return best


KERNEL_SOURCES = ("ntt_tile.cuh", "ntt_plan.h", "field.cuh", "field_asm.cuh")


def kernel_source_digest():
    """SHA-256 over the sources that define ntt_pass_kernel: tools/gpu_record.sh stores it next to the PMC figures it collects
    (profiles/rNN/traffic.json), and a bench line says so when the figures it quotes predate a change to those files"""
    import hashlib
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(REPO, "stark-anatomy_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def pmc_figures_are_current():
    """True / False / None (no record): were the latest committed PMC figures collected with today's kernel sources?"""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r*", "traffic.json")))
    if not files:
        return None
    try:
        recorded = json.load(open(files[-1])).get("kernel_source_sha256_16")
    except Exception:
        return None
    return None if recorded is None else recorded == kernel_source_digest()


def measured_traffic(log2n):
    """HBM bytes per ntt_pass_kernel launch from the committed rocprofv3 PMC runs (profiles/*/traffic.json), or None."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(REPO, "profiles", "r*", "traffic.json"))):
        try:
            rec = json.load(open(f)).get(str(log2n))
            if rec:
                best = rec["hbm_bytes_per_launch"]
        except Exception:
            pass
    return best


def ctypes_void(v):
    import ctypes
    return ctypes.c_void_p(v)


if __name__ == "__main__":
    main()
