#!/usr/bin/env python3
"""In-process A/B of the symmetric plane-marching SpMV (K1) against the seven-diagonal marching kernel and of its knobs:
back-to-back launches of the PCG's own kernel (SpMV + p.q) on the bench operator, interleaved rounds.
usage: python tools/sym_ab.py [ns=464] ["k=v,k=v" variants ...]   (each variant: fv_tune settings on top of the defaults)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402
import bench  # noqa: E402

fv = load_package()
lib = fv.load()
ns_ = sys.argv[1] if len(sys.argv) > 1 else "464"
variants = sys.argv[2:] or ["27=0", "27=1", "27=1,28=0", "27=1,28=1", "27=1,28=3", "27=1,28=7", "27=1,10=1", "27=1,10=2", "27=1,10=3", "27=1,10=4", "27=1,10=6",
                            "27=1,29=1", "27=1,29=2", "27=1,29=3", "27=1,29=4", "27=1,29=7"]
DEFAULTS = {27: 4, 37: 1, 9: 1}
ns = [int(v) for v in ns_.split("x")] if "x" in ns_ else [int(ns_)] * 3
mins, maxs = bench.spacing_box(ns)
dn, src = bench.box_setup(ns)
p = fv.Problem.regulargrid(mins, maxs, ns, dn)
p.assemble(np.array([1e-5]), src, np.full(len(dn), 1e3))
p.transient_begin(0.1, None, np.full(p.N, 1e3))
res = {v: [] for v in variants}
forms = {}
for r in range(4):
    for v in variants:
        for k, d in DEFAULTS.items():
            lib.fv_tune(k, d)
        for kv in v.split(","):
            k, val = kv.split("=")
            assert lib.fv_tune(int(k), int(val)) == 0, kv
        res[v].append(p.bench_spmv(1 / 60.0, 20))
        forms[v] = p.spmv_form()
for k, d in DEFAULTS.items():
    lib.fv_tune(k, d)
print("%s, rows %d; median ms per launch (min), bytes of the form, TB/s" % (ns_, p.n))
for v in variants:
    t = np.array(res[v])
    fid, name, nbytes = forms[v]
    print("  %-22s form %d  %.4f ms (%.4f)  %.2f GB  %.2f TB/s" % (v, fid, np.median(t), t.min(), nbytes / 1e9, nbytes / np.median(t) / 1e9), flush=True)
