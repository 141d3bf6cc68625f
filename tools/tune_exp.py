#!/usr/bin/env python3
"""Timing-only experiments (forms 3,4 give wrong results by design)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
import bench
fv = load_package()
ns = [464] * 3
mins, maxs = bench.spacing_box(ns)
dn, src = bench.box_setup(ns)
p = fv.Problem.regulargrid(mins, maxs, ns, dn)
p.assemble(np.array([1e-5]), src, np.full(len(dn), 1e3))
p.transient_begin(0.1, None, np.full(p.N, 1e3))
lib = fv.load()
bytes_ = 12 * p.nnz + 28 * p.n
for name, form in [("wstream", 2), ("coalesced-gather (timing only)", 3), ("no gather (timing only)", 4), ("wstream", 2)]:
    lib.fv_tune(0, form)
    ms = [p.bench_spmv(1 / 60.0, 10) for _ in range(3)]
    print("%-32s %.3f ms  %.0f GB/s" % (name, min(ms), bytes_ / min(ms) / 1e6))
