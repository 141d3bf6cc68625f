"""A/B of the burst length (fv_tune key 13) on the fixed-dt bench workload at several sizes, one process."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
import bench
fv = load_package()
lib = fv.load()
for n in (128, 216, 464):
    ns = [n] * 3
    mins, maxs = bench.spacing_box(ns)
    dn, src = bench.box_setup(ns)
    p = fv.Problem.regulargrid(mins, maxs, ns, dn)
    p.assemble(np.array([1e-5]), src, np.full(len(dn), 1e3))
    st = p.transient_begin(0.1, None, np.full(p.N, 1e3))
    p.run_fixed(st, 60.0, 4, 1e-10)
    res = {}
    for r in range(3):
        for chain in (0, 4, 8, 16):
            lib.fv_tune(13, chain)
            p.ctx.synchronize()
            t0 = time.perf_counter()
            it, info, ms = p.run_fixed(st, 60.0, 100, 1e-10)
            p.ctx.synchronize()
            res.setdefault(chain, []).append((time.perf_counter() - t0) / 100 * 1e3)
            assert info.converged and (it == 1).all()
    print("%d^3: ms/step by burst length: %s" % (n, ", ".join("%d: %.4f" % (c, min(v)) for c, v in sorted(res.items()))), flush=True)
    p.close()
lib.fv_tune(13, 8)
