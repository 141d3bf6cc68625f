"""Per-step time of ONE rank's share of the strong-scaling bench (464^3 over 8 GPUs = 58 x-planes = 1.25e7 rows):
the single-GPU loop and the row-block driver (1-rank RCCL communicator: the all-reduces and the stream hand-offs are
issued, nothing travels).  Shows what launch latency and host polling cost at that block size; the ideal is an eighth
of the 464^3 step.  Usage: python tools/block_rate.py [planes] [steps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

fv = load_package()
from fvamd import dist as fvdist  # noqa: E402

for kv in os.environ.get("FV_TUNE", "").split(","):  # e.g. FV_TUNE=21=1: the one-rank run issues its all-reduces through RCCL
    if "=" in kv:
        fv.load().fv_tune(int(kv.split("=")[0]), int(kv.split("=")[1]))
planes = int(sys.argv[1]) if len(sys.argv) > 1 else 58
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
ns = [planes + 2, 464, 464]  # + the two Dirichlet planes
mins, maxs = bench.spacing_box(ns)
dn, src = bench.box_setup(ns)
ctx = fv.default_context()


def problem():
    p = fv.Problem.regulargrid(mins, maxs, ns, dn, ctx)
    p.assemble(np.array([1e-5]), src, np.full(len(dn), 1e3))
    p.transient_begin(0.1, None, np.full(p.N, 1e3))
    return p


p = problem()
st = fv.DeviceVector(p, 0, owned=False)
p.run_fixed(st, 60.0, 8, 1e-10)
for rep in range(3):
    ctx.synchronize()
    t = time.perf_counter()
    it, info, ms = p.run_fixed(st, 60.0, steps, 1e-10)
    ctx.synchronize()
    t = time.perf_counter() - t
    print("single-GPU loop  rows %d  %.3f ms/step (device %.3f)  %.2f it/step" % (p.n, t / steps * 1e3, ms / steps, it.mean()), flush=True)
p.close()

fvdist.comm_init(ctx, 1, 0, fvdist.comm_unique_id())
p = problem()
blk = fvdist.RowBlock(p, 1, 0)
p.close()
blk.run_fixed(60.0, 8, 1e-10)
for rep in range(3):
    ctx.synchronize()
    t = time.perf_counter()
    it, info, ms = blk.run_fixed(60.0, steps, 1e-10)
    ctx.synchronize()
    t = time.perf_counter() - t
    print("row-block driver rows %d  %.3f ms/step (device %.3f)  %.2f it/step" % (blk.nloc, t / steps * 1e3, ms / steps, it.mean()), flush=True)
if os.environ.get("FV_BLOCK_RATE_AB"):  # in-process A/B of a tune key, e.g. FV_BLOCK_RATE_AB=22 (process-to-process noise is ~10 us)
    spec = os.environ["FV_BLOCK_RATE_AB"]  # "22" (values 1, 0) or "50=1,0"
    key = int(spec.split("=")[0])
    vals = [int(v) for v in spec.split("=")[1].split(",")] if "=" in spec else [1, 0]
    for rep in range(4):
        for val in vals:
            fv.load().fv_tune(key, val)
            blk.run_fixed(60.0, 16, 1e-10)
            ctx.synchronize()
            t = time.perf_counter()
            it, info, ms = blk.run_fixed(60.0, steps, 1e-10)
            ctx.synchronize()
            print("row-block driver, fv_tune(%d, %d): %.4f ms/step" % (key, val, (time.perf_counter() - t) / steps * 1e3), flush=True)
blk.close()
