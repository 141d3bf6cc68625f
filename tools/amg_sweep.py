"""AMG parameter sweep on the steady box_model solve (BASELINE configs[1] geometry, smooth log-K field) in one process:
    python tools/amg_sweep.py [n=256] [sigma=3.0]     -> per (theta, omega, passes, rounds): levels, set-up s, iterations, solve s"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402
from tests import workloads  # noqa: E402

fv = load_package()
lib = fv.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
sigma = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
ns = [n, n, n]
mins, maxs = [-50.0, -50.0, 0.0], [50.0, 50.0, 10.0]
dn, dh = workloads.box_model_dirichlet(ns)
p = fv.Problem.regulargrid(mins, maxs, ns, dn)
logk = np.log(1e-5) + sigma * workloads.smooth_gaussian_field(ns, seed=0)
n1 = np.empty(p.F, np.int64)
n2 = np.empty(p.F, np.int64)
p.check(lib.fv_problem_get_grid(p.handle, n1.ctypes.data, n2.ctypes.data, None, None))
Kf = fv.nodehycos2neighborhycos((n1, n2), logk, True)
del n1, n2
configs = [(0.25, 2 / 3, 3, 6), (0.25, 2 / 3, 2, 6), (0.25, 0.8, 3, 6), (0.25, 0.9, 3, 6), (0.25, 0.8, 2, 6), (0.1, 2 / 3, 3, 6), (0.5, 2 / 3, 3, 6), (0.25, 2 / 3, 4, 6)]
if os.environ.get("FV_AMG_SWEEP") == "2":
    configs = [(th, om, 3, 6) for th in (0.0, 0.05, 0.1, 0.15) for om in (2 / 3, 0.85, 1.0)] + [(0.1, 0.85, 3, 10), (0.05, 0.85, 2, 6)]
if os.environ.get("FV_AMG_SWEEP") == "3":  # the K-cycle (fv_tune 52: levels 1 .. k by two flexible-CG steps)
    configs = [(0.1, 0.85, 3, 10, k) for k in (0, 1, 2, 3)] + [(0.1, 0.85, 2, 10, k) for k in (0, 1, 2, 3, 5)] + [(0.05, 0.85, 2, 10, 3), (0.1, 0.7, 3, 10, 2), (0.1, 1.0, 3, 10, 2)]
if os.environ.get("FV_AMG_SWEEP") == "4":  # handshake rounds / threshold under the K-cycle
    configs = [(0.1, 0.85, 2, r, 2) for r in (3, 4, 6, 8, 10)] + [(th, 0.85, 2, 6, 2) for th in (0.05, 0.2, 0.3)] + [(0.1, om, 2, 6, 2) for om in (0.7, 1.0)]
if os.environ.get("FV_AMG_SWEEP") == "5":  # round 4 (rows without a candidate leave the matching): rounds / threshold / damping / K levels again
    configs = [(0.1, 0.85, 2, r, 2) for r in (4, 6, 10)] + [(th, 0.85, 2, 10, 2) for th in (0.05, 0.15, 0.25)] + [(0.1, om, 2, 10, 2) for om in (0.75, 0.95)] + \
              [(0.1, 0.85, 2, 10, k) for k in (1, 3)]
for cfg in configs:
    theta, omega, passes, rounds = cfg[:4]
    kc = cfg[4] if len(cfg) > 4 else 0
    os.environ["FV_AMG_KCYCLE"] = str(kc)  # (round 5: the K-cycle's depth is an environment variable, read at every use)
    lib.fv_amg_configure(theta, omega, passes, rounds)
    p.assemble(Kf, np.zeros(p.N), dh, None, True)  # (a new assembly epoch: the hierarchy is rebuilt with the new parameters)
    p.set_preconditioner("amg")
    p.ctx.synchronize()
    t0 = time.perf_counter()
    rows, nnz = p.amg_info()
    p.ctx.synchronize()
    t_setup = time.perf_counter() - t0
    t0 = time.perf_counter()
    head, res, ch = p.solve_steady(None, 1e-8, 600, want_head=False, want_resnorm=False)
    p.ctx.synchronize()
    t_solve = time.perf_counter() - t0
    print("theta %.2f omega %.2f passes %d rounds %d K-levels %d: rows %s complexity %.2f set-up %.3f s, %d iterations (%s) %.3f s -> %.2f ms per iteration" %
          (theta, omega, passes, rounds, kc, rows.tolist(), nnz.sum() / nnz[0], t_setup, ch.iters, "converged" if ch.isconverged else "NOT converged", t_solve,
           t_solve / max(ch.iters, 1) * 1e3), flush=True)
lib.fv_amg_configure(0.10, 0.85, 2, 10)
