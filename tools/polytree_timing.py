#!/usr/bin/env python3
"""Device subproduct tree: build / zerofier / multipoint evaluation / interpolation time vs number of points (best of 3) -- dev tool."""
import json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "stark-anatomy_amd"))
import starkcore as sc, synth
sc.init(0)


def best(fn, reps=3):
    b = None
    for _ in range(reps):
        sc.synchronize()
        t0 = time.perf_counter()
        r = fn()
        sc.synchronize()
        dt = time.perf_counter() - t0
        b = dt if b is None or dt < b else b
    return b, r


for logk in [int(a) for a in sys.argv[1:]] or (10, 14, 16, 18, 20, 22):
    k = 1 << logk
    pts = sc.DeviceVector.from_bytes(synth.synth_packed(11, k).tobytes())
    f = sc.DeviceVector.from_bytes(synth.synth_packed(12, k).tobytes())
    t_build, tree = best(lambda: sc.PolyTree(pts))
    t_first, vals = best(lambda: tree.evaluate(f), reps=1)          # includes the power-series inverse of the root (once per tree)
    t_eval, vals = best(lambda: tree.evaluate(f))
    t_interp, back = best(lambda: tree.interpolate(vals))
    ok = back.to_bytes() == f.to_bytes()
    print(json.dumps(dict(points=k, build_ms=round(t_build * 1e3, 3), first_evaluate_ms=round(t_first * 1e3, 3), evaluate_ms=round(t_eval * 1e3, 3),
                          interpolate_ms=round(t_interp * 1e3, 3), round_trip_ok=ok)), flush=True)
    tree.free()
