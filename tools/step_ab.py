"""In-process A/B of one fv_tune key on the bench workload (fixed-dt stepping, one-iteration regime): interleaved rounds,
ms per step.  usage: python tools/step_ab.py KEY V1 V2 [V3 ...] [--ns 464] [--steps 100]
or, for combinations of keys:  python tools/step_ab.py 36=1 36=0 36=1,26=0 ... (every key named anywhere is set in every variant: give all of them)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402
import bench  # noqa: E402

args = sys.argv[1:]
ns_, steps = 464, 100
if "--ns" in args:
    ns_ = int(args[args.index("--ns") + 1])
    del args[args.index("--ns") : args.index("--ns") + 2]
dt = 60.0
if "--dt" in args:
    dt = float(args[args.index("--dt") + 1])
    del args[args.index("--dt") : args.index("--dt") + 2]
if "--steps" in args:
    steps = int(args[args.index("--steps") + 1])
    del args[args.index("--steps") : args.index("--steps") + 2]
if "=" in args[0]:
    key, values = -1, args
else:
    key, values = int(args[0]), [int(v) for v in args[1:]]


def apply(v):
    if key >= 0:
        assert lib.fv_tune(key, v) == 0
    else:
        for kv in v.split(","):
            k, val = kv.split("=")
            assert lib.fv_tune(int(k), int(val)) == 0


fv = load_package()
lib = fv.load()
ns = [ns_] * 3
mins, maxs = bench.spacing_box(ns)
dn, src = bench.box_setup(ns)
p = fv.Problem.regulargrid(mins, maxs, ns, dn)
p.assemble(np.array([1e-5]), src, np.full(len(dn), 1e3))
st = p.transient_begin(0.1, None, np.full(p.N, 1e3))
res = {v: [] for v in values}
for r in range(5):
    for v in values:
        apply(v)
        p.run_fixed(st, dt, 8, 1e-10)
        p.ctx.synchronize()
        t0 = time.perf_counter()
        it, info, ms = p.run_fixed(st, dt, steps, 1e-10)
        p.ctx.synchronize()
        res[v].append((time.perf_counter() - t0) / steps * 1e3)
        assert info.converged and (dt != 60.0 or (it == 1).all())
print("%d^3, dt %g s, %.1f PCG iterations per step, fv_tune key %d: " % (ns_, dt, float(it.mean()), key) + "; ".join("%s -> median %.4f ms/step (min %.4f)" % (v, float(np.median(t)), min(t)) for v, t in res.items()), flush=True)
