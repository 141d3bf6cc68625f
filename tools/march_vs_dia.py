#!/usr/bin/env python3
"""Plane-marching vs plain sliced-DIA SpMV over grid shapes (short and long pencils), interleaved in one process.
usage: python tools/march_vs_dia.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402
import bench  # noqa: E402

fv = load_package()
lib = fv.load()
shapes = [[18, 464, 464], [34, 464, 464], [60, 464, 464], [118, 464, 464], [234, 464, 464], [216, 216, 216], [128, 128, 128], [1000, 100, 100], [100, 1000, 100]]
if len(sys.argv) > 1:
    shapes = [[int(v) for v in a.split("x")] for a in sys.argv[1:]]
variants = [("march auto", 2, 0), ("march m=1", 2, 1), ("march m=2", 2, 2), ("march m=4", 2, 4), ("plain DIA", 0, 0), ("library's choice", 1, 0)]
for ns in shapes:
    mins, maxs = bench.spacing_box(ns)
    dn, src = bench.box_setup(ns)
    p = fv.Problem.regulargrid(mins, maxs, ns, dn)
    p.assemble(np.array([1e-5]), src, np.full(len(dn), 1e3))
    p.transient_begin(0.1, None, np.full(p.N, 1e3))
    b = 12 * p.nnz + 20 * p.n
    res = {v[0]: [] for v in variants}
    for r in range(4):
        for name, march, m in variants:
            lib.fv_tune(9, march)
            lib.fv_tune(10, m)
            res[name].append(p.bench_spmv(1 / 60.0, 20))
    lib.fv_tune(9, 1)
    lib.fv_tune(10, 0)
    print("%-16s rows %9d  " % ("x".join(map(str, ns)), p.n) + "  ".join("%s %.1f us (%.2f TB/s)" % (k, np.median(v) * 1e3, b / np.median(v) / 1e9) for k, v in res.items()), flush=True)
    p.close()
