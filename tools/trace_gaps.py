#!/usr/bin/env python3
"""Where does a step's time go besides its two big kernels?  Reads a rocprofv3 --kernel-trace csv of a bench run and prints,
for the launches of the stepping loop in start order, kernel durations and the idle gaps between consecutive kernels.
usage: rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --steps 40 --warmup 5 --no-other-configs --no-multi-iteration --no-cpu-baseline
       python tools/trace_gaps.py DIR"""
import csv
import glob
import sys
from collections import defaultdict

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the stepping loop: from the first K2S (or fused step) launch to the last
MARK = next((m for m in ("fused_chunk_kernel", "fused_step_kernel") if any(m in r["Kernel_Name"] for r in rows)), "pcg_update_spec_kernel")
idx = [i for i, r in enumerate(rows) if MARK in r["Kernel_Name"]]
lo, hi = idx[5], idx[-1]
dur, gap, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
steps = sum(1 for i in range(lo, hi) if MARK in rows[i]["Kernel_Name"])
for i in range(lo, hi):
    k = rows[i]["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:48]
    d = int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])
    g = int(rows[i + 1]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"])
    dur[k] += d
    gap[k] += g
    cnt[k] += 1
total = int(rows[hi]["Start_Timestamp"]) - int(rows[lo]["Start_Timestamp"])
print("%d steps, %.4f ms per step (start of first K2S / fused launch to start of last)" % (steps, total / steps / 1e6))
for k in sorted(dur, key=lambda k: -dur[k]):
    print("  %-50s %5d launches  %8.4f ms/step in kernel   %7.4f ms/step idle after it (avg gap %.1f us)" % (k, cnt[k], dur[k] / steps / 1e6, gap[k] / steps / 1e6, gap[k] / cnt[k] / 1e3))
