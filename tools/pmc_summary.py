#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc CSV output: average counter value per dispatch, per kernel.  usage: pmc_summary.py <dir>..."""
import csv, glob, json, os, sys, collections
out = {}
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            k = (row["Kernel_Name"].split("(")[0][:60], row["Counter_Name"])
            acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
        for (kn, cn), (tot, cnt) in sorted(acc.items()):
            out.setdefault(os.path.basename(d.rstrip("/")), {}).setdefault(kn, {})[cn] = {"avg_per_dispatch": tot / cnt, "dispatches": cnt}
print(json.dumps(out, indent=1))
