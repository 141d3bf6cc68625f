#!/usr/bin/env python3
"""Where does the time go on the small Theis problem (15 650 unknowns, ~3 300 solves)?"""
import cProfile, os, pstats, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
from tests import refcases
fv = load_package()
grid = lambda a, b, n: (lambda r: (r[0], r[1][:, 0], r[1][:, 1], r[2], r[3]))(fv.regulargrid(a, b, n))
c = refcases.theis(grid)
nb = np.stack([c["node1"], c["node2"]], 1)
solver = fv.DevicePCG(rtol=1e-12, maxiter=2000)
def run():
    return fv.backwardeulerintegrate(c["u0"], c["tspan"], c["Ss"], c["volumes"], nb, c["aol"], c["K"], c["sources"], c["dnodes"], c["dheads"], atol=c["atol"], dt0=c["dt0"], linearsolver=solver)
t0 = time.perf_counter(); us, ts = run(); t1 = time.perf_counter()
print("theis: %.2f s, %d steps, %d solves, %d PCG iterations (%.1f per solve), %.0f us per solve" % (t1 - t0, len(ts) - 1, solver.solves, solver.total_iters, solver.total_iters / solver.solves, (t1 - t0) / solver.solves * 1e6))
pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(8)
