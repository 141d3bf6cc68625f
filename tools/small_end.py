#!/usr/bin/env python3
"""The small end: /root/reference/test/theis.jl:21-54 — 101 x 101 x 2 cells, 15 650 unknowns, the default adaptive
stepper, ~1 090 outer steps = ~3 280 solves — through the three ways this build can run it, and the test/theisadjoint.jl workflow
(25 x 25 x 2) through the host-closure path and with every state in HBM.  Prints one JSON line: seconds, solves, microseconds per solve.
Under `rocprofv3 --kernel-trace --stats` the kernel count / solves gives the launches per solve (tools/run_small_end.sh)."""
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402
from tests import refcases  # noqa: E402

fv = load_package()
ctx = fv.default_context()
grid = lambda a, b, n: (lambda r: (r[0], r[1][:, 0], r[1][:, 1], r[2], r[3]))(fv.regulargrid(a, b, n))  # noqa: E731
c = refcases.theis(grid)
nb = np.stack([c["node1"], c["node2"]], 1)
args = (c["u0"], c["tspan"], c["Ss"], c["volumes"], nb, c["aol"], c["K"], c["sources"], c["dnodes"], c["dheads"])
out = {"theis": {"unknowns": int(len(c["u0"]) - len(c["dnodes"]))}}
only = sys.argv[1] if len(sys.argv) > 1 else "all"


def timed(f):
    ctx.synchronize()
    t0 = time.perf_counter()
    r = f()
    ctx.synchronize()
    return r, time.perf_counter() - t0


if only in ("all", "theis"):
    fv.backwardeulerintegrate(*args, atol=c["atol"], dt0=c["dt0"], keep="last")  # warm-up: library load, first kernels
    solver = fv.DevicePCG(rtol=math.sqrt(np.finfo(float).eps), maxiter=1000)
    (us, ts), sec = timed(lambda: fv.backwardeulerintegrate(*args, atol=c["atol"], dt0=c["dt0"], linearsolver=solver))
    out["theis"]["host_loop_every_state_downloaded"] = {"seconds": sec, "outer_steps": len(ts) - 1, "solves": solver.solves, "pcg_iters_per_solve": solver.total_iters / solver.solves,
                                                        "us_per_solve": sec / solver.solves * 1e6}
    nsolves = solver.solves
    (r, sec) = timed(lambda: fv.backwardeulerintegrate(*args, atol=c["atol"], dt0=c["dt0"], keep="last"))
    out["theis"]["device_stepper_keep_last"] = {"seconds": sec, "outer_steps": len(r[1]) - 1, "us_per_solve": sec / nsolves * 1e6}
    (r, sec) = timed(lambda: fv.backwardeulerintegrate(*args, atol=c["atol"], dt0=c["dt0"], keep="device"))
    out["theis"]["device_stepper_states_in_hbm"] = {"seconds": sec, "outer_steps": len(r[1]) - 1, "us_per_solve": sec / nsolves * 1e6}
if only in ("all", "adjoint"):
    atol, side, thick = 1e-4, 50.0, 10.0
    coords, neighbors, aol, volumes = fv.regulargrid([-side, -side, 0.0], [side, side, thick], [25, 25, 2])
    F, N = len(aol), coords.shape[1]
    center = np.nonzero((coords[0] == 0) & (coords[1] == 0))[0]
    sources = np.zeros(N)
    sources[center] = -1e-3
    sources[center[0]] = sources[center[-1]] = -0.5e-3
    dnodes = np.nonzero(np.hypot(coords[0], coords[1]) - side >= 0)[0] + 1
    dheads = np.zeros(len(dnodes))
    u0 = np.zeros(N)
    tspan = (0.0, 60 * 60 * 24 * 1e1)
    kw = dict(atol=atol, dt0=60.0)
    mesh = (0.1, volumes, neighbors, aol)
    K0 = np.full(F, math.log(1e-5))
    rest = (K0, sources, dnodes, dheads, None, True)
    p0 = np.r_[K0, sources, dheads]
    us, ts = fv.backwardeulerintegrate(u0, tspan, *mesh, K0 + 1, *rest[1:], **kw)
    uobs = fv.getcontinuoussolution(us, ts)
    freenodes, n2f = fv.getfreenodes(N, dnodes)
    g, dgdu, dfdp, dgdp, du0dp, G = fv.getadjointfunctions(lambda i, t: 0.03, [int(n2f[i]) for i in center], uobs, u0, tspan, *mesh, *rest, **kw)

    def host():
        us_i, ts_i = fv.backwardeulerintegrate(u0, tspan, *mesh, *rest, **kw)
        uc = fv.getcontinuoussolution(us_i, ts_i)
        lam, tl = fv.adjointintegrate(lambda t: dgdu(uc, t), tspan, *mesh, *rest, **kw)
        idl = fv.integratedfdplambda(fv.getcontinuoussolution(us_i, ts_i, 2), p0, lam, tl, tspan, *mesh, *rest, complete=True)
        return fv.gradientintegrate(lam[0], du0dp, lambda t: dgdp(uc, t, p0), idl, tspan), G(uc), len(ts_i)

    def device():
        dus, dts = fv.backwardeulerintegrate(u0, tspan, *mesh, *rest, keep="device", **kw)
        duc = fv.getcontinuoussolution(dus, dts)
        lam, tl = fv.adjointintegrate(dgdu.bind(duc), tspan, *mesh, *rest, **kw)
        idl = fv.integratedfdplambda(duc, p0, lam, tl, tspan, *mesh, *rest, complete=True)
        return fv.gradientintegrate(lam[0], du0dp, lambda t: dgdp(duc, t, p0), idl, tspan), G(duc), len(dts)

    device()
    (gh, Gh, nh), sh = timed(host)
    (gd, Gd, nd), sd = timed(device)
    out["theisadjoint_25x25x2"] = {"objective_and_gradient_host_closure_s": sh, "objective_and_gradient_states_in_hbm_s": sd, "outer_steps": nh - 1,
                                   "gradient_rel_diff": float(np.abs(gd - gh).max() / np.abs(gh).max()), "G_rel_diff": abs(Gd - Gh) / abs(Gh)}
print(json.dumps(out))
