#!/usr/bin/env python3
"""Quick look at the fused step on a tile-form box: iteration counts, launches, heads against the unfused chain.
usage: python tools/fused_check.py [n1 n2 n3] [dt] [nsteps] [rtol]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_fused as T  # noqa: E402

fv = load_package()
ns = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else T.BOX
dt = float(sys.argv[4]) if len(sys.argv) > 4 else T.DT
nsteps = int(sys.argv[5]) if len(sys.argv) > 5 else 21
rtol = float(sys.argv[6]) if len(sys.argv) > 6 else 1e-11
case = T._problem(fv, ns)
out = {}
for fused in (1, 0):
    out[fused] = T._run(fv, case, bool(fused), [(dt, nsteps, rtol)])
    print("fused" if fused else "plain", "iterations", out[fused][1], "fused_form", out[fused][2], "spmv form", out[fused][3], flush=True)
print("relative difference of the heads: %.3e" % T.relerr(out[1][0], out[0][0]))
