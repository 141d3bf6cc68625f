"""SpMV rate on the fractures-like 5M-cell mesh (nodes randomly permuted inside every fracture), as given and after
the locality re-ordering of finitevolume.jl_amd/meshio.py (reverse Cuthill-McKee on the host)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
from tests import workloads
fv = load_package()
nfrac, m = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (20, 500)
w = workloads.fractures_like(nfrac, m, seed=0)
t0 = time.perf_counter()
order, rank = fv.meshio.locality_order(w["node1"], w["node2"], w["N"])
t_rcm = time.perf_counter() - t0
w2 = fv.meshio.reorder_mesh(dict(node1=w["node1"], node2=w["node2"], aol=w["aol"], K=w["K"], volumes=w["volumes"], dnodes=w["dnodes"], dheads=w["dheads"]), rank)
for name, ww in (("as given", w), ("re-ordered (RCM, %.1f s on the host)" % t_rcm, w2)):
    p = fv.Problem.create((ww["node1"], ww["node2"]), ww["aol"], w["N"], ww["dnodes"])
    p.assemble(ww["K"], np.zeros(w["N"]), ww["dheads"])
    st = p.transient_begin(1e-9, ww["volumes"], np.full(w["N"], 1.5e6))
    ms = min(p.bench_spmv(1.0, 20) for _ in range(3))
    b = 12 * p.nnz + 20 * p.n
    p.run_fixed(st, 1.0, 3, 1e-10)
    p.ctx.synchronize()
    t0 = time.perf_counter()
    it, info, _ = p.run_fixed(st, 1.0, 100, 1e-10)
    p.ctx.synchronize()
    sec = time.perf_counter() - t0
    head, res, ch = p.solve_steady(None, 1e-10, 60000, want_head=False, want_resnorm=False)
    print("fractures-like %s: n %d nnz %d | SpMV %.4f ms -> %.0f GB/s (CSR accounting) | transient dt=1s: %.3f ms/step (%.1f it) | steady Jacobi-PCG %d it %.2f s" % (name, p.n, p.nnz, ms, b / ms / 1e6, sec * 10, it.mean(), ch.iters, ch.solve_ms / 1e3), flush=True)
    p.close()
