"""Large-dt implicit steps at 464^3: Jacobi-PCG vs AMG-PCG (the hierarchy carries the storage term)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
import bench
fv = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 464
ns = [n] * 3
mins, maxs = bench.spacing_box(ns)
dn, src = bench.box_setup(ns)
p = fv.Problem.regulargrid(mins, maxs, ns, dn)
p.assemble(np.array([1e-5]), src, np.full(len(dn), 1e3))
for dt in (3600.0, 86400.0, 864000.0):
    for pre in ("jacobi", "amg"):
        p.set_preconditioner(pre)
        st = p.transient_begin(0.1, None, np.full(p.N, 1e3))
        p.run_fixed(st, dt, 2, 1e-10, maxiter=5000)
        p.ctx.synchronize()
        t0 = time.perf_counter()
        it, info, ms = p.run_fixed(st, dt, 10, 1e-10, maxiter=5000)
        p.ctx.synchronize()
        sec = time.perf_counter() - t0
        print("%d^3 dt=%gs %-6s: %.1f its/step, %.2f ms/step, converged %s" % (n, dt, pre, it.mean(), sec / 10 * 1e3, info.converged), flush=True)
