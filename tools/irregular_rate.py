"""The fractures-like 5M-cell mesh (BASELINE configs[3]): ms per SpMV and per implicit step, A/B of fv_tune keys in one process.
    python tools/irregular_rate.py [54=1 54=0 ...]   (each argument: comma-separated key=value pairs of one variant)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

fv = load_package()
lib = fv.load()
variants = sys.argv[1:] or ["54=1", "54=0"]
w = fv.workloads.fractures_like(20, 500, seed=0)
N = w["N"]
keys = sorted({int(kv.split("=")[0]) for v in variants for kv in v.split(",")})
for rep in range(2):
    for v in variants:
        for kv in v.split(","):
            assert lib.fv_tune(int(kv.split("=")[0]), int(kv.split("=")[1])) == 0
        p = fv.Problem.create((w["node1"], w["node2"]), w["aol"], N, w["dnodes"])
        p.assemble(w["K"], np.zeros(N), w["dheads"])
        st = p.transient_begin(1e-9, w["volumes"], np.full(N, 1.5e6))
        p.run_fixed(st, 1.0, 20, rtol=1e-10, maxiter=2000)
        secs = []
        for r in range(3):
            p.ctx.synchronize()
            t0 = time.perf_counter()
            it, info, _ = p.run_fixed(st, 1.0, 100, rtol=1e-10, maxiter=2000)
            p.ctx.synchronize()
            secs.append(time.perf_counter() - t0)
        ms = p.bench_spmv(1.0, 20)
        form = p.spmv_form()
        print("%-16s step %.4f ms (%.3e DoF-updates/s, %.2f it/step, fused launches %d at %d B/row)  SpMV %.4f ms = %.0f GB/s on %d B/row (%s)" %
              (v, np.median(secs) * 10, N * 100 / np.median(secs), it.mean(), p.fused_form()[0], p.fused_form()[1], ms, form[2] / ms / 1e6, form[2] // p.n, form[1][:30]), flush=True)
        p.close()
