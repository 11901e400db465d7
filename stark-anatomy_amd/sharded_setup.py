"""Setting up the sharded four-step transform for a measurement (bench.py --gpus N): every form of the corner turn this job can run
is built, checked against the first and timed for a few steps (`sharded_setup`); the direct-store forms only after a child of
every rank has tried their ingredients (`direct_store_preflight`); where a sharded transform's time goes (`stage_breakdown`);
what the node looks like (`node_facts`).  SURVEY.md 8(e); no reference counterpart (the reference is single-process Python).
"""
import os
import sys
import time

from workloads import nth_root

PKG = os.path.dirname(os.path.abspath(__file__))
HBM_PEAK_GBS = 8000.0                    # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BYTES_PER_ELEMENT_PER_TRANSFORM = 32     # read once + write once, 16-byte elements
_DIRECT_PREFLIGHT = None                 # the job's one pre-flight of the direct-store corner turn: {"passed": bool, ...}


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
        child = subprocess.run([sys.executable, os.path.join(PKG, "direct_preflight.py"), str(rank), str(world), str(local), where],
                               capture_output=True, text=True, timeout=120, env=dict(os.environ, STARKCORE_NO_TORCH="1"))
        status, said = child.returncode, child.stderr.strip().splitlines()[-1:] if child.stderr.strip() else []
    except subprocess.TimeoutExpired:
        status, said = -1, ["no answer within 120 s"]
    t = torch.tensor([status == 0], dtype=torch.int32, device=dev if on_dev else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    passed = int(t.item()) == 1
    if status != 0:
        sys.stderr.write("direct-store pre-flight, rank %d: status %d %s\n" % (rank, status, " ".join(said)))
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
