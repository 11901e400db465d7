#!/usr/bin/env python3
"""Idle gaps of the GPU in a rocprofv3 --kernel-trace (dev tool): the last `span_ms` of the trace, every gap longer than
`min_us` between the end of one kernel and the start of the next, with the kernels on either side, and the busy fraction.
   python tools/gap_report.py <kernel_trace.csv> [span_ms=30 | 0 = the last-but-one burst of >= 500 kernels] [min_us=40]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
span = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 40.0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
if span > 0:
    end = max(int(r["End_Timestamp"]) for r in rows)
    rows = [r for r in rows if int(r["Start_Timestamp"]) >= end - span * 1e6]
else:
    # span 0: a window of `WINDOW_MS` (env, default 24.5) starting `OFFSET_MS` (env, default 45) after the first Merkle kernel of
    # the trace -- one proof out of the middle of a loop of proofs
    import os
    first = min(int(r["Start_Timestamp"]) for r in rows if "merkle_subtree" in r["Kernel_Name"])
    lo = first + float(os.environ.get("OFFSET_MS", "45")) * 1e6
    hi = lo + float(os.environ.get("WINDOW_MS", "24.5")) * 1e6
    rows = [r for r in rows if lo <= int(r["Start_Timestamp"]) < hi]
short = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "").replace("sc::", "")[:46]
busy, cursor, gaps = 0, None, []
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if cursor is not None and s > cursor:
        gaps.append(((s - cursor) / 1e3, (cursor - t0) / 1e3, prev, short(r)))
    if cursor is None or e > cursor:
        busy += e - max(s, cursor or s)
        cursor, prev = e, short(r)
total = (cursor - t0) / 1e3
print("window %.1f us, busy %.1f us (%.1f %%), %d kernels, idle %.1f us in %d gaps" % (total, busy / 1e3, 100 * busy / 1e3 / total, len(rows), total - busy / 1e3, len(gaps)))
big = [g for g in gaps if g[0] >= min_us]
print("gaps >= %.0f us: %d, together %.1f us; smaller ones together %.1f us" % (min_us, len(big), sum(g[0] for g in big), sum(g[0] for g in gaps if g[0] < min_us)))
print("%10s %9s  %-46s -> %s" % ("at_us", "gap_us", "after", "before"))
for g in sorted(big, key=lambda g: -g[0])[:45]:
    print("%10.1f %9.1f  %-46s -> %s" % (g[1], g[0], g[2], g[3]))
