#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc csv output: mean counter value per kernel. usage: pmc_summary.py DIR [filter]"""
import collections
import csv
import glob
import sys

# --min-frac F: of every kernel's launches only those whose counter value is at least F x the kernel's largest (loops enqueue launches past
# convergence that do no work, and a solve's first pass moves less than the others: the mean over ALL launches says nothing)
args = sys.argv[1:]
minfrac = 0.0
if "--min-frac" in args:
    i = args.index("--min-frac")
    minfrac = float(args[i + 1])
    del args[i : i + 2]
d = args[0]
flt = args[1] if len(args) > 1 else ""
for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if flt not in k:
            continue
        agg[k.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        for c, vals in v.items():
            if minfrac > 0.0 and vals:
                top = max(vals)
                vals = [x for x in vals if x >= minfrac * top]
            print("%-62s %-30s n=%3d mean=%.6g" % (k[:62], c, len(vals), sum(vals) / max(len(vals), 1)))
