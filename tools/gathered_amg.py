"""Steady solve of a high-contrast box with the AMG preconditioner: one GPU, then row blocks on loopback ranks (threads of this process
on one GPU) with block-Jacobi AMG and with the gathered coarse levels — iteration counts and wall time.
    python tools/gathered_amg.py [nx ny nz] [sigma]"""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402
from tests import workloads  # noqa: E402

fv = load_package()
from fvamd import dist  # noqa: E402

ns = [int(v) for v in sys.argv[1:4]] if len(sys.argv) > 3 else [128, 128, 64]
sigma = float(sys.argv[4]) if len(sys.argv) > 4 else 3.0
mins, maxs = [-50.0, -50.0, 0.0], [50.0, 50.0, 10.0]
dn, dh = workloads.box_model_dirichlet(ns)
logk = np.log(1e-5) + sigma * workloads.smooth_gaussian_field(ns, seed=0)


def build(ctx=None):
    p = fv.Problem.regulargrid(mins, maxs, ns, dn, ctx) if ctx is not None else fv.Problem.regulargrid(mins, maxs, ns, dn)
    n1, n2 = np.empty(p.F, np.int64), np.empty(p.F, np.int64)
    p.check(fv.load().fv_problem_get_grid(p.handle, n1.ctypes.data, n2.ctypes.data, None, None))
    Kf = fv.nodehycos2neighborhycos((n1, n2), logk, True)
    return p.assemble(Kf, np.zeros(p.N), dh, None, True)


p = build()
p.set_preconditioner("amg")
t0 = time.perf_counter()
head, res, ch = p.solve_steady(None, 1e-8, 400, want_head=False, want_resnorm=False)
print("one GPU: %d iterations, %.3f s (with set-up)" % (ch.iters, time.perf_counter() - t0), flush=True)
p.close()
gid = 4000
for kind in ("amg_gathered", "amg"):
    for nranks in [int(v) for v in os.environ.get("FV_RANKS", "2,4,8").split(",")]:
        gid += 1
        out, errors = [None] * nranks, []

        def worker(rank):
            try:
                ctx = fv.Context(0)
                dist.comm_init_local(ctx, nranks, rank, gid)
                pg = build(ctx)
                pg.transient_begin(0.1, None, np.zeros(pg.N))
                blk = dist.RowBlock(pg, nranks, rank).set_preconditioner(kind)
                pg.close()
                t = time.perf_counter()
                x, info = blk.solve_steady(None, 1e-8, 2000)
                out[rank] = (info.iters, info.converged, time.perf_counter() - t)
                blk.close()
                fv.load().fv_comm_destroy(ctx.handle)
            except BaseException as e:  # noqa: BLE001
                errors.append((rank, repr(e)))

        th = [threading.Thread(target=worker, args=(r,), daemon=True) for r in range(nranks)]
        for t in th:
            t.start()
        for t in th:
            t.join(timeout=900)
        if errors:
            print(kind, nranks, "FAILED", errors, flush=True)
            continue
        print("%-13s %d loopback ranks: %d iterations (converged %s), %.3f s on rank 0 (threads share one GPU; reductions through the host)" %
              (kind, nranks, out[0][0], out[0][1], out[0][2]), flush=True)
