#!/usr/bin/env python3
"""Stress test of the hand-scheduled field arithmetic (dev tool): thousands of transforms of random data, the default plan against
two differently shaped plans (generic kernel with 8 elements per thread; smaller tiles) bit for bit, and against the C oracle for the
sizes it finishes quickly.  Run several copies at once to perturb the wave scheduling (the hazard spacing of mont_mul2 / fe_addsub2 must
not depend on it).   python tools/stress_parity.py [seconds=40]"""
import ctypes, os, random, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "stark-anatomy_amd")); sys.path.insert(0, REPO)
import numpy as np, torch
import starkcore as sc, synth
from oracle import py_oracle as po
P = synth.P
sc.init(0); lib = sc.lib(); dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream); sptr = ctypes.c_void_p(stream.cuda_stream)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 40.0
rng = random.Random(os.getpid())
DEF = dict(fixed_shapes=1, loge=2, max_tile_log=-1, max_digit_log=-1, wave_local=1, prio_balance=-1)
ALTS = [dict(fixed_shapes=0, loge=3), dict(max_tile_log=10), dict(wave_local=0, prio_balance=1), dict(fixed_shapes=0, loge=2), dict(prio_balance=2), dict(prio_balance=0)]
def tune(d):
    for k, v in DEF.items(): sc.set_tuning(k, v)
    for k, v in d.items(): sc.set_tuning(k, v)
t_end, count, oracle_checks = time.time() + budget, 0, 0
while time.time() < t_end:
    lg = rng.randint(12, 21)
    n = 1 << lg
    seed = rng.randrange(1 << 30)
    host = synth.synth_packed(seed, n)
    if rng.random() < 0.2:                                      # edge-heavy data: 0, 1, p-1, 2^64-1 ... sprinkled in
        ints = synth.unpack_ints(host[:64].tobytes())
        edge = [0, 1, P - 1, P - 2, (1 << 64) - 1, 1 << 64, (1 << 127), P >> 1]
        a = host.reshape(-1, 2).copy()
        for i in range(0, n, max(1, n // 512)):
            v = edge[rng.randrange(len(edge))]
            a[i, 0], a[i, 1] = v & ((1 << 64) - 1), v >> 64
        host = a.reshape(-1)
    x = torch.from_numpy(host.view(np.int64)).to(dev)
    root = sc.fe_bytes(po.primitive_nth_root(n)); inv = rng.randrange(2)
    outs = []
    for cfg in [dict()] + rng.sample(ALTS, 2):
        tune(cfg)
        y = torch.empty_like(x)
        sc._check(lib.sc_ntt_dev(x.data_ptr(), y.data_ptr(), n, root, inv, sptr))
        outs.append(y)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), ("plans disagree", lg, seed, inv)
    # the same vector as column 1 of a batch of three (sc_ntt_columns_dev: one set of launches), in place, under one of the plans
    if lg <= 20 and rng.random() < 0.5:
        tune(rng.choice([dict()] + ALTS))
        flat = x.reshape(-1)
        three = torch.cat([torch.roll(flat, 2), flat, torch.roll(flat, 4)]).contiguous()
        sc._check(lib.sc_ntt_columns_dev(three.data_ptr(), three.data_ptr(), n, 3, root, inv, sptr))
        torch.cuda.synchronize()
        assert torch.equal(three[flat.numel():2 * flat.numel()], outs[0].reshape(-1)), ("columns call disagrees", lg, seed, inv)
        count += 3
    if lg <= 15:
        raw = host.tobytes()
        want = po.C.intt(po.primitive_nth_root(n), raw, n) if inv else po.C.ntt(po.primitive_nth_root(n), raw, n)
        assert outs[0].cpu().numpy().tobytes() == want, ("oracle", lg, seed, inv)
        oracle_checks += 1
    count += 3
tune({})
print("stress ok: %d transforms, %d oracle checks, pid %d" % (count, oracle_checks, os.getpid()))
