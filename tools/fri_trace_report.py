#!/usr/bin/env python3
"""Timeline of the LAST Fri.prove in a rocprofv3 kernel trace (dev tool): python tools/fri_trace_report.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# proves are separated by >= 10 ms of idle
groups, cur, last_end = [], [], None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if last_end is not None and s - last_end > 10_000_000:
        groups.append(cur); cur = []
    cur.append(r); last_end = e
groups.append(cur)
g = groups[-1]
t0 = int(g[0]["Start_Timestamp"]); busy = 0; prev_end = t0
print("%9s %8s %8s  %-44s %s" % ("t_us", "dur_us", "gap_us", "kernel", "grid"))
for r in g:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("sc::", "")
    print("%9.1f %8.1f %8.1f  %-44s %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, name[:44], r.get("Grid_Size_X", r.get("Grid_Size", ""))))
    busy += e - s; prev_end = e
print("span_us %.1f busy_us %.1f kernels %d" % ((prev_end - t0) / 1e3, busy / 1e3, len(g)))
