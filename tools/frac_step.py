#!/usr/bin/env python3
"""The fractures-like 5M-cell configuration (BASELINE configs[3]) on its own: fv_problem_create, transient steps, SpMV.
usage: python tools/frac_step.py [steps]   (FV_TUNE=key=value,... for A/B)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

fv = load_package()
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
w = fv.workloads.fractures_like(20, 500, seed=0)
t0 = time.perf_counter()
p = fv.Problem.create((w["node1"], w["node2"]), w["aol"], w["N"], w["dnodes"])
t_create = time.perf_counter() - t0
p.assemble(w["K"], np.zeros(w["N"]), w["dheads"])
st = p.transient_begin(1e-9, w["volumes"], np.full(w["N"], 1.5e6))
p.run_fixed(st, 1.0, 8, 1e-10, maxiter=5000)
res = []
for r in range(3):
    p.ctx.synchronize()
    t0 = time.perf_counter()
    it, info, _ = p.run_fixed(st, 1.0, steps, 1e-10, maxiter=5000)
    p.ctx.synchronize()
    res.append((time.perf_counter() - t0) / steps * 1e3)
ms = p.bench_spmv(1.0, 20)
fid, fname, fbytes = p.spmv_form()
print("create %.3f s (%r); %.4f ms per step (min %.4f) = %.3e DoF-updates/s, %.2f its/step; SpMV %s %.4f ms = %.2f TB/s on %.0f MB; K2S %d B/row" %
      (t_create, p.reorder_info(), float(np.median(res)), min(res), w["N"] / (float(np.median(res)) * 1e-3), float(it.mean()), fname, ms, fbytes / ms / 1e9, fbytes / 1e6, p.update_form()))
