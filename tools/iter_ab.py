#!/usr/bin/env python3
"""In-process A/B of fv_tune settings on the many-iteration regime of the bench operator (dt = 1 h, ~11 PCG iterations per step):
ms per iteration, interleaved rounds; heads of the variants against the first one.
usage: python tools/iter_ab.py [--ns 464] 46=0 46=1 ..."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402
import bench  # noqa: E402

args = sys.argv[1:]
ns_ = 464
if "--ns" in args:
    ns_ = int(args[args.index("--ns") + 1])
    del args[args.index("--ns") : args.index("--ns") + 2]
variants = args or ["46=0", "46=1"]
fv = load_package()
lib = fv.load()
ns = [ns_] * 3
mins, maxs = bench.spacing_box(ns)
dn, src = bench.box_setup(ns)
p = fv.Problem.regulargrid(mins, maxs, ns, dn)
p.assemble(np.array([1e-5]), src, np.full(len(dn), 1e3))
res = {v: [] for v in variants}
heads = {}
for r in range(3):
    for v in variants:
        for kv in v.split(","):
            k, val = kv.split("=")
            assert lib.fv_tune(int(k), int(val)) == 0
        st = p.transient_begin(0.1, None, np.full(p.N, 1e3))
        p.run_fixed(st, 3600.0, 2, 1e-10)
        p.ctx.synchronize()
        t0 = time.perf_counter()
        it, info, ms = p.run_fixed(st, 3600.0, 6, 1e-10)
        p.ctx.synchronize()
        sec = time.perf_counter() - t0
        assert info.converged
        res[v].append(sec / it.sum() * 1e3)
        heads[v] = (st.free_values(), it.copy())
ref = heads[variants[0]]
for v in variants:
    print("%-12s median %.4f ms per iteration (min %.4f), iterations %s, heads vs %s: %.2e" % (v, float(np.median(res[v])), min(res[v]), heads[v][1].tolist(), variants[0],
          np.abs(heads[v][0] - ref[0]).max() / np.abs(ref[0]).max()), flush=True)
