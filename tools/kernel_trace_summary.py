#!/usr/bin/env python3
"""Per (kernel, grid size) summary of a rocprofv3 --kernel-trace csv: a bench run launches the same kernel on several problem sizes, so the
per-name averages of --stats mix them; this groups the launches by grid size too and — with --min-frac F — keeps, per group, the launches that last
at least F x the group's longest (loops enqueue launches past convergence that stop at their prologue).
usage: kernel_trace_summary.py DIR [name filter] [--min-frac F]"""
import collections
import csv
import glob
import sys

args = sys.argv[1:]
minfrac = 0.0
if "--min-frac" in args:
    i = args.index("--min-frac")
    minfrac = float(args[i + 1])
    del args[i : i + 2]
d = args[0]
flt = args[1] if len(args) > 1 else ""
groups = collections.defaultdict(list)
for f in sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if flt in name:
            groups[(name, int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), int(r["LDS_Block_Size"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("%-58s %7s %8s %6s %6s %10s %10s %10s" % ("kernel", "blocks", "LDS", "calls", "kept", "avg us", "min us", "max us"))
for (name, blocks, lds), v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
    keep = [x for x in v if x >= minfrac * max(v)] if minfrac > 0 else v
    if sum(v) < 2000.0:
        continue
    print("%-58s %7d %8d %6d %6d %10.1f %10.1f %10.1f" % (name[:58], blocks, lds, len(v), len(keep), sum(keep) / len(keep), min(keep), max(keep)))
