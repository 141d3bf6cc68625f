"""A forward + adjoint pair of the transient-adjoint workflow at 10^7 cells with every state in HBM: 216^3 box of the
watertable-like configuration, `nsteps` fixed steps forward recorded into an fv_trajectory, observation series at `nobs` rows, then
fv_adjoint_run over the same steps; ms per forward step, ms per adjoint step, and the same sweep through the host-closure path (dgdu
evaluated on the host, a dense forcing uploaded per solve) over a few steps for comparison.
usage: python tools/adjoint_rate.py [--ns 216] [--steps 60] [--dt 3600] [--nobs 64]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--ns", type=int, default=216)
ap.add_argument("--steps", type=int, default=60)
ap.add_argument("--dt", type=float, default=3600.0)
ap.add_argument("--nobs", type=int, default=64)
ap.add_argument("--rtol", type=float, default=1e-10)
ap.add_argument("--host-steps", type=int, default=6)
args = ap.parse_args()
fv = load_package()
ctx = fv.default_context()
ns = [args.ns] * 3
dn, src = bench.box_setup(ns)
p = fv.Problem.regulargrid([0.0, 0.0, 0.0], [1000.0, 1000.0, 100.0], ns, dn, ctx)
p.assemble(np.array([1e-5]), src, np.full(len(dn), 1e3))
st = p.transient_begin(0.1, None, np.full(p.N, 1e3))
T = args.dt * args.steps
free0, _ = ctx.mem_info()
# warm-up on a scratch state (the solver's storage forms are built at the first products)
ws = p.new_state()
ws.set_nodes(np.full(p.N, 1e3))
p.run_fixed(ws, args.dt, 3, args.rtol, 5000)
del ws
# forward, recorded
tr = p.new_trajectory()
tr.push(st, 0.0)
p.record(tr, 0.0)
ctx.synchronize()
t0 = time.perf_counter()
its_f, info, _ = p.run_fixed(st, args.dt, args.steps, args.rtol, 5000)
ctx.synchronize()
t_fwd = time.perf_counter() - t0
p.record(None)
# forward again without recording (bursts allowed): what recording costs
st2 = p.new_state()
st2.set_nodes(np.full(p.N, 1e3))
ctx.synchronize()
t0 = time.perf_counter()
p.run_fixed(st2, args.dt, args.steps, args.rtol, 5000)
ctx.synchronize()
t_fwd_plain = time.perf_counter() - t0
# observations: the recorded drawdown scaled by 1.1 at rows spread along the well column's plane
rng = np.random.default_rng(0)
obs = np.sort(rng.choice(p.n, args.nobs, replace=False)) + 1
knots = np.linspace(0.0, T, 8)
uobs = np.stack([1e3 - 1.1 * (1e3 - tr.at(t)[obs - 1]) for t in knots])
ob = fv.core.Observation(p, obs, knots, uobs, np.full((len(knots), len(obs)), 0.03)) if hasattr(fv, "core") else None
if ob is None:
    from fvamd.core import Observation

    ob = Observation(p, obs, knots, uobs, np.full((len(knots), len(obs)), 0.03))
G = ob.integral(tr, 0.0, T)
ctx.synchronize()
t0 = time.perf_counter()
lam, nout, nsol, ainfo = p.adjoint_run(tr, ob, 0.0, T, dt0=args.dt, adaptive=False, rtol=args.rtol, maxiter=5000)
ctx.synchronize()
t_adj = time.perf_counter() - t0
used = (free0 - ctx.mem_info()[0]) / 1e9
# the host-closure path over a few steps: interpolate u on the host, build dgdu, upload it, one adjoint step
from fvamd import _lib  # noqa: E402

hs = p.new_state()
hs.set_free(np.zeros(p.n))
ctx.synchronize()
t0 = time.perf_counter()
for k in range(args.host_steps):
    t = k * args.dt
    u = tr.at(T - t)  # (already a device interpolation + download: the host mirror would hold the states in host memory)
    f = np.zeros(p.n)
    f[obs - 1] = 2 * 0.03**2 * (u[obs - 1] - np.array([np.interp(T - t, knots, uobs[:, j]) for j in range(len(obs))]))
    p.step(hs, hs, args.dt, f, _lib.FV_STEP_ADJOINT, args.rtol, 5000)
ctx.synchronize()
t_host = (time.perf_counter() - t0) / args.host_steps
print(json.dumps({"cells": p.N, "unknowns": p.n, "steps": args.steps, "dt": args.dt, "trajectory_knots": len(tr), "hbm_in_use_gb": used,
                  "forward": {"ms_per_step_recording": t_fwd / args.steps * 1e3, "ms_per_step_plain": t_fwd_plain / args.steps * 1e3, "pcg_iters_per_step": float(its_f.mean())},
                  "adjoint_device": {"ms_per_step": t_adj / max(nout, 1) * 1e3, "outer_steps": nout, "solves": nsol, "last_iters": ainfo.iters, "lambda_knots": len(lam)},
                  "adjoint_host_closure": {"ms_per_step": t_host * 1e3, "steps": args.host_steps}, "G": G, "nobs": int(args.nobs)}))
