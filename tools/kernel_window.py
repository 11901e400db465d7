#!/usr/bin/env python3
"""Every kernel of one window of a rocprofv3 --kernel-trace, in time order, with the idle time before it (dev tool).
   python tools/kernel_window.py <kernel_trace.csv>      env OFFSET_MS (after the first Merkle kernel, default 40), WINDOW_MS (default 8)"""
import csv, os, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = min(int(r["Start_Timestamp"]) for r in rows if "merkle_subtree" in r["Kernel_Name"])
lo = first + float(os.environ.get("OFFSET_MS", "40")) * 1e6
hi = lo + float(os.environ.get("WINDOW_MS", "8")) * 1e6
rows = [r for r in rows if lo <= int(r["Start_Timestamp"]) < hi]
short = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "").replace("sc::", "")[:60]
cursor = None
print("%10s %8s %8s  %s" % ("at_us", "idle_us", "run_us", "kernel"))
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    idle = 0.0 if cursor is None else max(0.0, (s - cursor) / 1e3)
    print("%10.1f %8.1f %8.1f  %s" % ((s - lo) / 1e3, idle, (e - s) / 1e3, short(r)))
    cursor = e if cursor is None else max(cursor, e)
