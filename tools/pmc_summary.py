#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc csv output: mean counter value per kernel. usage: pmc_summary.py DIR [filter]"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if flt not in k:
            continue
        agg[k.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        for c, vals in v.items():
            print("%-50s %-30s n=%3d mean=%.6g" % (k[:50], c, len(vals), sum(vals) / len(vals)))
