"""In-process A/B of fv_tune settings on the bench's heterogeneous-conductivity problem (bench.hetero_face_K / the smooth Gaussian
field), fixed-dt stepping: interleaved rounds, ms per step, PCG iterations per step, the fused launch's form.
usage: python tools/hetero_ab.py 60=1,62=5 60=1,62=4 60=0 [--ns 464] [--dt 7.5] [--steps 40]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402
import bench  # noqa: E402

args = sys.argv[1:]


def opt(name, default, cast):
    if name in args:
        i = args.index(name)
        v = cast(args[i + 1])
        del args[i : i + 2]
        return v
    return default


ns_ = opt("--ns", 464, int)
dt = opt("--dt", 7.5, float)
steps = opt("--steps", 40, int)
rounds = opt("--rounds", 5, int)
values = args
fv = load_package()
lib = fv.load()
ns = [ns_] * 3
mins, maxs = bench.spacing_box(ns)
dn, src = bench.box_setup(ns)
p = fv.Problem.regulargrid(mins, maxs, ns, dn)
K = bench.hetero_face_K(ns, p.F, p.N)
p.assemble(K, src, np.full(len(dn), 1e3))
del K
st = p.transient_begin(0.1, None, np.full(p.N, 1e3))


def apply(v):
    for kv in v.split(","):
        k, val = kv.split("=")
        assert lib.fv_tune(int(k), int(val)) == 0


res = {v: [] for v in values}
form = {}
for r in range(rounds):
    for v in values:
        apply(v)
        st = p.transient_begin(0.1, None, np.full(p.N, 1e3))  # (from the start state every time: a long run reaches steps that are converged at their set-up)
        p.run_fixed(st, dt, 8, 1e-10, 2000)
        p.ctx.synchronize()
        t0 = time.perf_counter()
        it, info, ms = p.run_fixed(st, dt, steps, 1e-10, 2000)
        p.ctx.synchronize()
        res[v].append((time.perf_counter() - t0) / steps * 1e3)
        assert info.converged
        form[v] = (float(it.mean()), p.fused_form(), p.fused_traversal(), p.loop_form())
for v, t in res.items():
    print("%d^3 heterogeneous, dt %g s: %s -> median %.4f ms/step (min %.4f); iterations/step %.2f, fused form %s, traversal %d, loop form %s"
          % (ns_, dt, v, float(np.median(t)), min(t), form[v][0], form[v][1], form[v][2], form[v][3]), flush=True)
