#!/usr/bin/env python3
"""Merkle trees and subproduct trees on one GPU (dev tool; one script for merkle_prof / merkle_timing / polytree_prof / polytree_timing).

  python tools/tree_profile.py merkle-timing [big_nlev ...]     build time vs leaf count (2^12 ... 2^24, best of 6), optionally per value of
                                                               the "merkle_big_nlev" tuning
  python tools/tree_profile.py merkle-run [logn=24] [reps=4]    just build trees (run under rocprofv3 --kernel-trace --stats)
  python tools/tree_profile.py polytree-timing [logk ...]       subproduct tree: build / first evaluation / evaluation / interpolation, best of 3
  python tools/tree_profile.py polytree-run [logk=20]           one build + evaluation + interpolation, three times (for rocprofv3)
"""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "stark-anatomy_amd"))
import starkcore as sc          # noqa: E402
import synth                    # noqa: E402


def merkle_timing(argv):
    nlevs = [int(a) for a in argv] or [None]
    for logn in (12, 16, 20, 22, 24):
        n = 1 << logn
        v = sc.DeviceVector.from_bytes(synth.synth_packed(9, n).tobytes())
        for nl in nlevs:
            if nl is not None:
                sc.set_tuning("merkle_big_nlev", nl)
            best, root = None, None
            for _ in range(6):
                t0 = time.perf_counter()
                t = sc.MerkleTree.from_device(v)
                dt = time.perf_counter() - t0
                root = t.root.hex()[:16]
                t.free()
                best = dt if best is None or dt < best else best
            print(json.dumps(dict(logn=logn, big_nlev=nl, ms=round(best * 1e3, 3), gcompress_s=round((2 * n - 1) / best / 1e9, 2), root=root)), flush=True)


def merkle_run(argv):
    logn = int(argv[0]) if argv else 24
    reps = int(argv[1]) if len(argv) > 1 else 4
    v = sc.DeviceVector.from_bytes(synth.synth_packed(9, 1 << logn).tobytes())
    for _ in range(reps):
        sc.MerkleTree.from_device(v).free()
    sc.synchronize()


def _best(fn, reps=3):
    b = r = None
    for _ in range(reps):
        sc.synchronize()
        t0 = time.perf_counter()
        r = fn()
        sc.synchronize()
        dt = time.perf_counter() - t0
        b = dt if b is None or dt < b else b
    return b, r


def polytree_timing(argv):
    for logk in [int(a) for a in argv] or (10, 14, 16, 18, 20, 22):
        k = 1 << logk
        pts = sc.DeviceVector.from_bytes(synth.synth_packed(11, k).tobytes())
        f = sc.DeviceVector.from_bytes(synth.synth_packed(12, k).tobytes())
        t_build, tree = _best(lambda: sc.PolyTree(pts))
        t_first, vals = _best(lambda: tree.evaluate(f), reps=1)          # includes the power-series inverse of the root (once per tree)
        t_eval, vals = _best(lambda: tree.evaluate(f))
        t_interp, back = _best(lambda: tree.interpolate(vals))
        print(json.dumps(dict(points=k, build_ms=round(t_build * 1e3, 3), first_evaluate_ms=round(t_first * 1e3, 3), evaluate_ms=round(t_eval * 1e3, 3),
                              interpolate_ms=round(t_interp * 1e3, 3), round_trip_ok=back.to_bytes() == f.to_bytes())), flush=True)
        tree.free()


def polytree_run(argv):
    k = 1 << (int(argv[0]) if argv else 20)
    pts = sc.DeviceVector.from_bytes(synth.synth_packed(11, k).tobytes())
    f = sc.DeviceVector.from_bytes(synth.synth_packed(12, k).tobytes())
    for _ in range(3):
        tree = sc.PolyTree(pts)
        back = tree.interpolate(tree.evaluate(f))
        sc.synchronize()
        tree.free()
    print("round trip", back.to_bytes() == f.to_bytes())


if __name__ == "__main__":
    commands = {"merkle-timing": merkle_timing, "merkle-run": merkle_run, "polytree-timing": polytree_timing, "polytree-run": polytree_run}
    if len(sys.argv) < 2 or sys.argv[1] not in commands:
        sys.exit(__doc__)
    sc.init(0)
    commands[sys.argv[1]](sys.argv[2:])
