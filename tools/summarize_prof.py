#!/usr/bin/env python
"""Summarise a tools/profile.sh output directory into one JSON/markdown (kernel time from --kernel-trace --stats,
PMC counters per dispatch of clouds_kernel averaged over dispatches)."""
import csv
import glob
import json
import os
import sys

d = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else "clouds_kernel"
out = {"dir": d, "kernel": kern}
for f in glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    out["kernel_stats"] = [{k: r[k] for k in r} for r in rows]
cnt = {}
for f in sorted(glob.glob(os.path.join(d, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        if kern not in r.get("Kernel_Name", ""):
            continue
        c = cnt.setdefault(r["Counter_Name"], [0.0, 0])
        c[0] += float(r["Counter_Value"]); c[1] += 1
out["counters_per_dispatch"] = {k: v[0] / v[1] for k, v in sorted(cnt.items())}
out["dispatches_per_counter"] = {k: v[1] for k, v in sorted(cnt.items())}
print(json.dumps(out, indent=1))
