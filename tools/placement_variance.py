"""Does the physical placement of the big arrays explain the run-to-run spread of the kernel times?  The bench problem is
built, stepped and destroyed several times in ONE process; K1 / K2S averages per incarnation (HIP events)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402
import bench  # noqa: E402

fv = load_package()
ns = [int(sys.argv[1]) if len(sys.argv) > 1 else 464] * 3
mins, maxs = bench.spacing_box(ns)
dn, src = bench.box_setup(ns)
keep = []
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 6):
    p = fv.Problem.regulargrid(mins, maxs, ns, dn, lean=(len(sys.argv) > 3 and sys.argv[3] == "lean") or None)
    p.assemble(np.array([1e-5]), src, np.full(len(dn), 1e3))
    st = p.transient_begin(0.1, None, np.full(p.N, 1e3))
    p.run_fixed(st, 60.0, 4, 1e-10)
    p.profile(True)
    it, info, ms = p.run_fixed(st, 60.0, 60, 1e-10)
    prof = p.profile_get()
    p.profile(False)
    print("incarnation %d: %.3f ms/step, K1 %.3f ms, K2S %.3f ms" % (rep, ms / 60, prof["spmv_dot"][0] / prof["spmv_dot"][1], prof["update"][0] / prof["update"][1]), flush=True)
    p.close()
    if rep % 2 == 0:  # shift the allocator's state between incarnations
        keep.append(fv.Problem.regulargrid(mins, maxs, [64, 64, 64 + 16 * rep], bench.box_setup([64, 64, 64 + 16 * rep])[0]))
