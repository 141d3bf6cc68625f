#!/usr/bin/env python3
"""A/B the SpMV variants in ONE process (interleaved rounds) on the bench workload.
usage: python tools/tune_spmv.py [ns] [rounds]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402
import bench  # noqa: E402

fv = load_package()
ns_ = int(sys.argv[1]) if len(sys.argv) > 1 else 464
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ns = [ns_] * 3
mins, maxs = bench.spacing_box(ns)
dn, src = bench.box_setup(ns)
p = fv.Problem.regulargrid(mins, maxs, ns, dn)
p.assemble(np.array([1e-5]), src, np.full(len(dn), 1e3))
p.transient_begin(0.1, None, np.full(p.N, 1e3))
lib = fv.load()
bytes_ = 12 * p.nnz + 20 * p.n  # CSR accounting with the shift folded (variants that read D move 8 n more)
# name, form, order, fold, nt, dia, march (0 = off, m = segments per XCD)
variants = [("DIA plane-marching auto", 2, 1, 1, 1, 1, -1), ("DIA plane-marching m=2", 2, 1, 1, 1, 1, 2), ("DIA plane-marching m=3", 2, 1, 1, 1, 1, 3),
            ("DIA plane-marching m=7", 2, 1, 1, 1, 1, 7), ("sliced-DIA plane-blocked", 2, 1, 1, 1, 1, 0), ("sliced-DIA natural order", 2, 0, 1, 1, 1, 0),
            ("wstream+order+fold+nt", 2, 1, 1, 1, 0, 0), ("lanes-per-row(8)", 1, 0, 0, 0, 0, 0)]


def select(form, order, fold, nt, dia, march):
    lib.fv_tune(0, form)
    lib.fv_tune(2, order)
    lib.fv_tune(3, fold)
    lib.fv_tune(4, nt)
    lib.fv_tune(6, dia)
    lib.fv_tune(9, 2 if march else 0)
    if march:
        lib.fv_tune(10, max(march, 0))  # -1: let the library choose the segment count


res = {v[0]: [] for v in variants}
rng = np.random.default_rng(0)
x = rng.standard_normal(p.n)
ref = None
for name, *knobs in variants:  # correctness of every variant against the last (lanes-per-row)
    select(*knobs)
    y = p.spmv(x, sigma=1 / 60.0)
    if ref is None:
        ref = y
    else:
        err = np.abs(y - ref).max() / np.abs(ref).max()
        print("%-28s max rel diff to the first variant %.2e" % (name, err), flush=True)
        assert err < 1e-13, (name, err)
for r in range(rounds):
    for name, *knobs in variants:
        select(*knobs)
        res[name].append(p.bench_spmv(1 / 60.0, 10))
for name, v in res.items():
    v = np.array(v)
    print("%-24s median %.3f ms  min %.3f ms  -> %.0f GB/s (median), %.1f%% of 8 TB/s" % (name, np.median(v), v.min(), bytes_ / np.median(v) / 1e6, bytes_ / np.median(v) / 1e6 / 80))
