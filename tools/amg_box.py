"""Steady box_model solve (BASELINE configs[1] geometry, smooth log-K field): Jacobi-PCG vs AMG-PCG on the GPU.
    python tools/amg_box.py 256 3.0 [jacobi_maxiter]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402
from tests import workloads  # noqa: E402

fv = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
sigma = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
jmax = int(sys.argv[3]) if len(sys.argv) > 3 else 40000
ns = [n, n, n]
mins, maxs = [-50.0, -50.0, 0.0], [50.0, 50.0, 10.0]
dn, dh = workloads.box_model_dirichlet(ns)
p = fv.Problem.regulargrid(mins, maxs, ns, dn)
logk = np.log(1e-5) + sigma * workloads.smooth_gaussian_field(ns, seed=0)
n1 = np.empty(p.F, np.int64)
n2 = np.empty(p.F, np.int64)
p.check(fv.load().fv_problem_get_grid(p.handle, n1.ctypes.data, n2.ctypes.data, None, None))
Kf = fv.nodehycos2neighborhycos((n1, n2), logk, True)
del n1, n2
p.assemble(Kf, np.zeros(p.N), dh, None, True)
print("cells %d unknowns %d nnz %d" % (p.N, p.n, p.nnz), flush=True)
p.set_preconditioner("amg")
t0 = time.perf_counter()
rows, nnz = p.amg_info()
t_setup = time.perf_counter() - t0
print("AMG setup %.3f s; rows %s; nnz %s; operator complexity %.2f" % (t_setup, rows.tolist(), nnz.tolist(), nnz.sum() / nnz[0]), flush=True)
t0 = time.perf_counter()
head, res, ch = p.solve_steady(None, 1e-8, 400, want_resnorm=True)
t_amg = time.perf_counter() - t0
r = p.spmv(res) - p.b()
print("AMG-PCG: iters %d converged %s device ms %.1f wall %.3f s; true relres %.2e" % (ch.iters, ch.isconverged, ch.solve_ms, t_amg, np.linalg.norm(r) / np.linalg.norm(p.b())), flush=True)
if jmax > 0:
    p.set_preconditioner("jacobi")
    t0 = time.perf_counter()
    head_j, res_j, ch_j = p.solve_steady(None, 1e-8, jmax, want_resnorm=False)
    print("Jacobi-PCG: iters %d converged %s device ms %.1f wall %.3f s" % (ch_j.iters, ch_j.isconverged, ch_j.solve_ms, time.perf_counter() - t0), flush=True)
    print("heads: max |amg - jacobi| = %.3e" % np.abs(head - head_j).max())
