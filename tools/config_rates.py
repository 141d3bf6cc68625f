"""Throughput of every BASELINE.json configuration that fits one GPU (SURVEY.md §8d inputs), one line each.
    python tools/config_rates.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402
import bench  # noqa: E402
from tests import workloads  # noqa: E402

fv = load_package()


def line(name, cells, steps, sec, extra):
    print("%-44s %11d cells x %4d steps in %8.3f s -> %.3e DoF-updates/s   %s" % (name, cells, steps, sec, cells * steps / sec, extra), flush=True)


# configs[1]: box_model 256^3, one steady solve (Jacobi and AMG)
ns = [256, 256, 256]
mins, maxs = [-50.0, -50.0, 0.0], [50.0, 50.0, 10.0]
dn, dh = workloads.box_model_dirichlet(ns)
p = fv.Problem.regulargrid(mins, maxs, ns, dn)
for sigma in (0.0, 3.0):
    if sigma == 0.0:
        p.assemble(np.array([1e-5]), np.zeros(p.N), dh)
    else:
        logk = np.log(1e-5) + sigma * workloads.smooth_gaussian_field(ns, seed=0)
        n1 = np.empty(p.F, np.int64)
        n2 = np.empty(p.F, np.int64)
        p.check(fv.load().fv_problem_get_grid(p.handle, n1.ctypes.data, n2.ctypes.data, None, None))
        Kf = fv.nodehycos2neighborhycos((n1, n2), logk, True)
        del n1, n2
        p.assemble(Kf, np.zeros(p.N), dh, None, True)
    for pre in ("jacobi", "amg"):
        p.set_preconditioner(pre)
        t0 = time.perf_counter()
        if pre == "amg":
            p.amg_info()
        t_setup = time.perf_counter() - t0
        t0 = time.perf_counter()
        head, res, ch = p.solve_steady(None, 1e-8, 60000, want_head=False, want_resnorm=False)
        sec = time.perf_counter() - t0
        line("box_model 256^3 steady sigma=%g %s-PCG" % (sigma, pre), p.N, 1, sec + t_setup, "iters %d converged %s (set-up %.3f s)" % (ch.iters, ch.isconverged, t_setup))
p.close()

# configs[2]: watertable-like 216^3 transient, 100 fixed steps (dt = 60 s as in §8d, and dt = 1 h)
ns = [216, 216, 216]
mins, maxs = [0.0, 0.0, 0.0], [1000.0, 1000.0, 100.0]
dn, src = bench.box_setup(ns)
p = fv.Problem.regulargrid(mins, maxs, ns, dn)
p.assemble(np.array([1e-5]), src, np.full(len(dn), 1e3))
for dt in (60.0, 3600.0):
    st = p.transient_begin(0.1, None, np.full(p.N, 1e3))
    p.run_fixed(st, dt, 2, 1e-10)
    p.ctx.synchronize()
    t0 = time.perf_counter()
    iters, info, ms = p.run_fixed(st, dt, 100, 1e-10)
    p.ctx.synchronize()
    line("watertable-like 216^3 transient dt=%gs" % dt, p.N, 100, time.perf_counter() - t0, "%.1f PCG iters/step, converged %s" % (iters.mean(), info.converged))
p.close()

# configs[3]: fractures-like 5M cells, irregular CSR, 100 fixed steps
w = workloads.fractures_like(20, 500, seed=0)
p = fv.Problem.create((w["node1"], w["node2"]), w["aol"], w["N"], w["dnodes"])
p.assemble(w["K"], np.zeros(w["N"]), w["dheads"])
for dt in (1e-3, 1.0):
    st = p.transient_begin(1e-9, w["volumes"], np.full(w["N"], 1.5e6))
    p.run_fixed(st, dt, 2, 1e-10)
    p.ctx.synchronize()
    t0 = time.perf_counter()
    iters, info, ms = p.run_fixed(st, dt, 100, 1e-10, maxiter=5000)
    p.ctx.synchronize()
    line("fractures-like 5M transient dt=%gs" % dt, p.N, 100, time.perf_counter() - t0, "%.1f PCG iters/step, converged %s" % (iters.mean(), info.converged))
for pre in ("jacobi", "amg"):
    p.set_preconditioner(pre)
    t0 = time.perf_counter()
    head, res, ch = p.solve_steady(None, 1e-10, 60000, want_head=False, want_resnorm=False)
    line("fractures-like 5M steady %s-PCG" % pre, p.N, 1, time.perf_counter() - t0, "iters %d converged %s" % (ch.iters, ch.isconverged))
p.close()

# configs[4] on one GPU: the bench workload
ns = [464, 464, 464]
mins, maxs = bench.spacing_box(ns)
dn, src = bench.box_setup(ns)
p = fv.Problem.regulargrid(mins, maxs, ns, dn)
p.assemble(np.array([1e-5]), src, np.full(len(dn), 1e3))
for dt in (60.0, 3600.0):
    st = p.transient_begin(0.1, None, np.full(p.N, 1e3))
    p.run_fixed(st, dt, 2, 1e-10)
    p.ctx.synchronize()
    t0 = time.perf_counter()
    iters, info, ms = p.run_fixed(st, dt, 20, 1e-10)
    p.ctx.synchronize()
    line("synthetic 464^3 transient dt=%gs" % dt, p.N, 20, time.perf_counter() - t0, "%.1f PCG iters/step, converged %s" % (iters.mean(), info.converged))
