"""Differential check across the optimisation switches of the fixed-dt run on random boxes and schedules: streaming hints
(fv_tune 26) must not change a bit; the sparse-b forms (12), speculation (8) and the residual carry-over (7) may differ in
rounding only.  Measured: everything bit-identical except carry-over off, 1e-14 relative, same iteration counts."""
import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from __graft_entry__ import load_package
import bench
fv = load_package(); lib = fv.load()
bad = 0
for seed in range(10):
    rng = np.random.default_rng(seed)
    ns = [int(rng.integers(12, 34)), int(rng.integers(12, 30)), int(rng.integers(8, 26))]
    mins, maxs = bench.spacing_box(ns); dn, src = bench.box_setup(ns)
    tight = seed % 2
    dts, rtols = ([2.0**-10, 2.0**-8, 2.0**-6, 1.0, 3600.0], [1e-8, 1e-10, 1e-12]) if tight else ([2.0**-8, 1.0, 20.0, 60.0, 600.0, 3600.0], [1e-3, 1e-5, 3e-5, 1e-8, 1e-12])
    schedule = [(float(rng.choice(dts)), int(rng.integers(1, 46)), float(rng.choice(rtols))) for _ in range(int(rng.integers(3, 9)))]
    out = {}
    for name, key, val in (("base", 26, 3), ("no hints", 26, 0), ("p streamed", 26, 7), ("b dense", 12, 0), ("b gather in vector blocks", 12, 2), ("no speculation", 8, 0), ("no carry", 7, 0)):
        lib.fv_tune(26, 3); lib.fv_tune(12, 1); lib.fv_tune(8, 1); lib.fv_tune(7, 128)
        lib.fv_tune(key, val)
        p = fv.Problem.regulargrid(mins, maxs, ns, dn)
        K = 1e-5 * np.exp(np.random.default_rng(100 + seed).standard_normal(p.F))
        p.assemble(K, src, np.full(len(dn), 1e3))
        st = p.transient_begin(0.1, None, np.full(p.N, 1e3) + np.random.default_rng(200 + seed).standard_normal(p.N))
        its = np.concatenate([p.run_fixed(st, dt, k, rtol, maxiter=5000)[0] for dt, k, rtol in schedule])
        out[name] = (st.free_values(), its); p.close()
    ref = out["base"]
    for name, (state, its) in out.items():
        same = np.array_equal(state, ref[0]) and np.array_equal(its, ref[1])
        rel = np.abs(state - ref[0]).max() / np.abs(ref[0]).max()
        dit = int(np.abs(its.astype(int) - ref[1].astype(int)).sum())
        flag = ""
        if name in ("no hints", "p streamed") and not same: flag = "  <-- expected bitwise"; bad += 1
        print("seed %d %-28s bitwise %s  rel diff %.2e  iteration-count diff %d%s" % (seed, name, same, rel, dit, flag), flush=True)
lib.fv_tune(26, 3); lib.fv_tune(12, 1); lib.fv_tune(8, 1); lib.fv_tune(7, 128)
print("unexpected:", bad)
