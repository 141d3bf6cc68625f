#!/usr/bin/env python3
"""bench.py — DoF-updates/s of the implicit transient hot path on MI355X.

A "step" is one backward-Euler time step of the whole grid: RHS build + the
Jacobi-PCG solve of (D/dt + A) u+ = D u/dt + b, everything resident in HBM.
Workload (BASELINE.json configs[4] / the 10^8-cell target of
`north_star`): structured box of ns^3 cells (default 464^3 = 9.99e7), 1000 x 1000
x 100 m, K = 1e-5 m/s, Ss = 0.1, Dirichlet head 1e3 on the four lateral faces,
u0 = 1e3, one pumping well (-1e-3 m^3/s) down the centre column, fixed dt = 60 s,
PCG rtol 1e-10.  Grid, connectivity and CSR are generated/assembled on the device.

    python bench.py --gpus 1 --steps 20 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line (rank 0).  value = cells x steps / seconds, whole job.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6290 GB/s is the measured copy ceiling


def box_setup(ns, i1_lo=None, i1_hi=None):
    """Lateral-Dirichlet box of the transient configs.  Returns mins, maxs, dirichlet nodes (1-based), sources."""
    n1, n2, n3 = ns
    N = n1 * n2 * n3
    i1 = np.arange(n1)
    i2 = np.arange(n2)
    lat = np.zeros((n1, n2), bool)
    lat[0, :] = lat[-1, :] = True
    lat[:, 0] = lat[:, -1] = True
    cols = np.nonzero(lat.ravel())[0]  # (i1,i2) columns on the lateral boundary
    dn = (cols[:, None] * n3 + np.arange(n3)[None, :]).ravel() + 1
    dn.sort()
    sources = np.zeros(N)
    c1, c2 = n1 // 2, n2 // 2
    well = (c1 * n2 + c2) * n3 + np.arange(n3)
    Q = 1e-3
    sources[well] = -2 * Q / (2 * n3 - 2)  # theis.jl:40-42 weighting along the column
    sources[well[0]] = sources[well[-1]] = -Q / (2 * n3 - 2)
    return dn.astype(np.int64), sources


def spacing_box(ns, ref_ns=464):
    """Box extents that keep the cell size of the 464^3 / 1000x1000x100 m configuration."""
    hx = 1000.0 / (ref_ns - 1)
    hz = 100.0 / (ref_ns - 1)
    return [0.0, 0.0, 0.0], [hx * (ns[0] - 1), hx * (ns[1] - 1), hz * (ns[2] - 1)]


def cpu_baseline(dt, rtol):
    """The oracle (CPU restatement of the reference: assembleA/b, scalebyvolume!,
    fixedbackwardeulerstep!, IterativeSolvers-style CG) timed on ONE host core on a
    bounded sample of the same workload: same cell size / K / Ss / dt / BCs, 256^3
    cells, 30 steps (10-20 s of CPU).  Only the stepping loop is timed, as for the GPU."""
    from oracle import fv_oracle as o

    ns = [256, 256, 256]
    steps = 30
    mins, maxs = spacing_box(ns)
    _, n1, n2, aol, vol = o.regulargrid(mins, maxs, ns, want_coords=False)
    dn, src = box_setup(ns)
    dh = np.full(len(dn), 1e3)
    K = np.full(len(aol), 1e-5)
    Ss = 0.1
    freenodes, n2f = o.getfreenodes(len(vol), dn)
    f2n = o.freenodei2nodei(n2f)
    A = o.assembleA(n1, n2, aol, K, src, dn, dh)
    b = o.assembleb(n1, n2, aol, K, src, dn, dh)
    o.scalebyvolume_A(A, Ss * vol, f2n)
    b = o.scalebyvolume_b(b, Ss * vol, f2n)
    u0 = np.full(A.n, 1e3)
    iters = []

    def solver(Am, rhs, x0):
        x, ch = o.cg(Am, rhs, x0=x0, tol=rtol, maxiter=1000)
        iters.append(ch.iters)
        return x

    t0 = time.perf_counter()
    us, ts = o.backwardeulerintegrate_generic(u0, A, b, dt, 0.0, dt * steps, stepper=o.fixedbackwardeulerstep, linearsolver=solver)
    sec = time.perf_counter() - t0
    N = len(vol)
    return {
        "value": N * steps / sec,
        "unit": "DoF-updates/s",
        "cores": 1,
        "kind": "port",
        "sample": "oracle (C restatement of the reference path, unpreconditioned CG rtol %.0e) on %dx%dx%d cells x %d steps, same cell size/K/Ss/dt/BCs; %.1f s; %.1f CG iters/step" % (rtol, ns[0], ns[1], ns[2], steps, sec, float(np.mean(iters))),
    }


def other_baseline_configs(fv, ctx):
    """The other single-GPU configurations of BASELINE.json (SURVEY 8d inputs), one short measurement each, reported inside
    `config` — the headline `value` stays the 10^8-cell run.  The same code as tools/config_rates.py."""
    from tests import workloads

    rows = []

    def row(name, cells, steps, sec, **extra):
        rows.append(dict(config=name, cells=int(cells), steps=int(steps), seconds=sec, dof_updates_per_s=cells * steps / sec, **extra))

    # configs[1]: box_model 256^3, one steady solve, sigma = 3 log-conductivity field; PCG with the aggregation-AMG V-cycle
    ns = [256, 256, 256]
    dn, dh = workloads.box_model_dirichlet(ns)
    p = fv.Problem.regulargrid([-50.0, -50.0, 0.0], [50.0, 50.0, 10.0], ns, dn, ctx)
    logk = np.log(1e-5) + 3.0 * workloads.smooth_gaussian_field(ns, seed=0)
    n1, n2 = np.empty(p.F, np.int64), np.empty(p.F, np.int64)
    p.check(fv.load().fv_problem_get_grid(p.handle, n1.ctypes.data, n2.ctypes.data, None, None))
    Kf = fv.nodehycos2neighborhycos((n1, n2), logk, True)
    del n1, n2
    p.assemble(Kf, np.zeros(p.N), dh, None, True)
    p.set_preconditioner("amg")
    ctx.synchronize()
    t0 = time.perf_counter()
    levels = p.amg_info()
    head, res, ch = p.solve_steady(None, 1e-8, 400, want_head=False, want_resnorm=False)
    ctx.synchronize()
    row("box_model 256^3 steady, sigma=3 field, AMG-PCG rtol 1e-8 (hierarchy set-up included)", p.N, 1, time.perf_counter() - t0,
        pcg_iters=int(ch.iters), converged=bool(ch.isconverged), amg_levels=int(len(levels[0])))
    p.close()
    # configs[2]: watertable-like 10 M cells, 100 implicit steps
    ns = [216, 216, 216]
    dn, src = box_setup(ns)
    p = fv.Problem.regulargrid([0.0, 0.0, 0.0], [1000.0, 1000.0, 100.0], ns, dn, ctx)
    p.assemble(np.array([1e-5]), src, np.full(len(dn), 1e3))
    st = p.transient_begin(0.1, None, np.full(p.N, 1e3))
    p.run_fixed(st, 60.0, 3, 1e-10)
    ctx.synchronize()
    t0 = time.perf_counter()
    iters, info, _ = p.run_fixed(st, 60.0, 100, 1e-10)
    ctx.synchronize()
    row("watertable-like 216^3 transient, 100 steps, dt=60s, Jacobi-PCG rtol 1e-10", p.N, 100, time.perf_counter() - t0,
        pcg_iters_per_step=float(iters.mean()), converged=bool(info.converged))
    p.close()
    # configs[3]: fractures-like 5 M cells, irregular CSR (cells numbered at random inside each fracture), 100 implicit steps
    w = workloads.fractures_like(20, 500, seed=0)
    p = fv.Problem.create((w["node1"], w["node2"]), w["aol"], w["N"], w["dnodes"], ctx)
    p.assemble(w["K"], np.zeros(w["N"]), w["dheads"])
    st = p.transient_begin(1e-9, w["volumes"], np.full(w["N"], 1.5e6))
    p.run_fixed(st, 1.0, 3, 1e-10, maxiter=5000)
    ctx.synchronize()
    t0 = time.perf_counter()
    iters, info, _ = p.run_fixed(st, 1.0, 100, 1e-10, maxiter=5000)
    ctx.synchronize()
    row("fractures-like 5M cells (irregular CSR, as numbered), transient, 100 steps, dt=1s, Jacobi-PCG rtol 1e-10", p.N, 100, time.perf_counter() - t0,
        pcg_iters_per_step=float(iters.mean()), converged=bool(info.converged))
    p.close()
    # the same mesh after the locality re-numbering of meshio.locality_order (host pre-processing, reverse Cuthill-McKee)
    t0 = time.perf_counter()
    order, rank = fv.meshio.locality_order(w["node1"], w["node2"], w["N"])
    t_rcm = time.perf_counter() - t0
    w2 = fv.meshio.reorder_mesh(dict(node1=w["node1"], node2=w["node2"], aol=w["aol"], K=w["K"], volumes=w["volumes"], dnodes=w["dnodes"], dheads=w["dheads"]), rank)
    p = fv.Problem.create((w2["node1"], w2["node2"]), w2["aol"], w["N"], w2["dnodes"], ctx)
    p.assemble(w2["K"], np.zeros(w["N"]), w2["dheads"])
    st = p.transient_begin(1e-9, w2["volumes"], np.full(w["N"], 1.5e6))
    p.run_fixed(st, 1.0, 3, 1e-10, maxiter=5000)
    ctx.synchronize()
    t0 = time.perf_counter()
    iters, info, _ = p.run_fixed(st, 1.0, 100, 1e-10, maxiter=5000)
    ctx.synchronize()
    row("fractures-like 5M cells re-numbered for locality (meshio.locality_order), transient, 100 steps, dt=1s", p.N, 100, time.perf_counter() - t0,
        pcg_iters_per_step=float(iters.mean()), converged=bool(info.converged), renumbering_host_s=t_rcm)
    p.close()
    return rows


def main():
    # multi-process GPU work on this pool needs dmabuf IPC (RCCL's peer mappings fail with the legacy mode); the launcher
    # normally exports it already — set before anything touches HIP
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)  # 0.65 s of stepping: one hiccup of the box no longer decides the number
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--ns", type=int, default=464, help="cells per dimension of the box (464 -> 9.99e7 cells)")
    ap.add_argument("--dt", type=float, default=60.0)
    ap.add_argument("--rtol", type=float, default=1e-10)
    ap.add_argument("--maxiter", type=int, default=2000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short runs of BASELINE.json's other single-GPU configurations")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-kernel HIP-event timing inside the timed region")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print("bench.py: --gpus %d needs the torch.distributed.run launcher (one process per GPU)" % args.gpus, file=sys.stderr)
            sys.exit(2)
    from __graft_entry__ import load_package

    fv = load_package()
    for kv in os.environ.get("FV_TUNE", "").split(","):  # A/B knobs of libfvhip (fv_tune), e.g. FV_TUNE=5=0,6=0
        if "=" in kv:
            k, v = kv.split("=")
            fv.load().fv_tune(int(k), int(v))
    if world > 1 or os.environ.get("FV_BENCH_FORCE_DIST") == "1":  # the env var rehearses the multi-GPU driver with one rank
        from bench_dist import run_distributed

        return run_distributed(fv, args, world, rank)

    ctx = fv.default_context()
    name, cus, mem = ctx.device_info()
    free0, _ = ctx.mem_info()
    ns = [args.ns] * 3
    mins, maxs = spacing_box(ns)
    dn, src = box_setup(ns)
    t_setup = time.perf_counter()
    p = fv.Problem.regulargrid(mins, maxs, ns, dn, ctx)
    t_symbolic = time.perf_counter() - t_setup
    t1 = time.perf_counter()
    p.assemble(np.array([1e-5]), src, np.full(len(dn), 1e3))
    t_assemble = time.perf_counter() - t1
    state = p.transient_begin(0.1, None, np.full(p.N, 1e3))
    t_setup = time.perf_counter() - t_setup

    # warmup steps (untimed)
    if args.warmup > 0:
        p.run_fixed(state, args.dt, args.warmup, args.rtol, args.maxiter)
    if not args.no_profile:
        p.profile(True)
    ctx.synchronize()
    t0 = time.perf_counter()
    iters, info, dev_ms = p.run_fixed(state, args.dt, args.steps, args.rtol, args.maxiter)
    ctx.synchronize()
    sec = time.perf_counter() - t0
    prof = p.profile_get() if not args.no_profile else None
    p.profile(False)

    value = p.N * args.steps / sec
    # algorithmic bytes of the dominant kernel, the PCG SpMV q = (A + D/dt) p with the p.q epilogue, in
    # SURVEY.md §8d's CSR accounting: vals 8 + colind 4 per entry; rowptr 4 + x 8 + y 8 per row.  The fixed-dt
    # run folds D/dt into the stored diagonal, so the "+8 n if the shift vector is read separately" does not apply.
    spmv_bytes = 12 * p.nnz + 20 * p.n
    roof = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
            "kernel": ("K1 q=(A+D/dt)p with p.q: spmv_dia_march_kernel<true,true,true> (plane-marching sliced-DIA, 16-byte window accesses; + spmv_wstream_kernel<512,true,true> on non-grid-like slices, none on this grid)"
                       if 8 * p.n > 160 * 2**20 else
                       "K1 q=(A+D/dt)p with p.q: spmv_dia_kernel<true,true,false> (slice-by-slice sliced-DIA: x fits the last-level cache at this size)"),
            "algorithmic_bytes_per_launch": spmv_bytes}
    kern = {}
    if prof and prof["spmv_dot"][1] > 0:
        ms, cnt = prof["spmv_dot"]
        ach = spmv_bytes / (ms / cnt * 1e-3) / 1e9
        roof.update(achieved=ach, frac=ach / HBM_PEAK_GBS, avg_launch_ms=ms / cnt, launches=cnt)
        tune = dict(kv.split("=") for kv in os.environ.get("FV_TUNE", "").split(",") if "=" in kv)
        fused = tune.get("7", "32") != "0" and tune.get("8", "1") != "0" and float(np.mean(iters)) == 1.0
        # K2 in the one-iteration regime also prepares the next step (pcg_update_spec_kernel): 7 streams in, 3 out
        fused_bytes = 72 if tune.get("12", "1") != "0" else 80  # the sparse b's share of |rhs|^2 comes from a gather (fv_tune key 12)
        for k, bytes_ in (("update", (fused_bytes if fused else 56) * p.n), ("pupdate", 32 * p.n)):
            kms, kc = prof[k]
            if kc:
                kern[k] = {"avg_ms": kms / kc, "launches": kc}
                if kms / kc > 0.05 * (32 * p.n / 5e12 * 1e3):  # the last p-update of a solve is skipped on convergence: no bandwidth figure for no-ops
                    kern[k]["GB/s"] = bytes_ / (kms / kc * 1e-3) / 1e9
    tfile = os.path.join(ROOT, "profiles", "spmv_traffic.json")
    if os.path.exists(tfile):
        try:
            t = json.load(open(tfile))
            roof["traffic"] = t.get(str(args.ns))
        except Exception:
            pass

    out = {
        "metric": "DoF-updates/s (cells\u00d7steps) implicit transient; SpMV HBM GB/s vs peak",
        "value": value,
        "unit": "DoF-updates/s",
        "n_gpus": 1,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": sec / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": "synthetic %d^3 box (%.3g cells), transient, fixed dt=%gs, Jacobi-PCG rtol %.0e, lateral Dirichlet + centre well (SURVEY 8d / BASELINE configs[4] on one GPU)" % (args.ns, p.N, args.dt, args.rtol),
            "cells": p.N, "unknowns": p.n, "nnz": p.nnz, "faces": p.F,
            "pcg_iters_per_step": float(np.mean(iters)), "pcg_iters_total": int(np.sum(iters)),
            "last_relres": info.relres, "converged": bool(info.converged),
            "device": name, "compute_units": cus,
            "setup_s": {"grid+symbolic": t_symbolic, "assemble": t_assemble, "total": t_setup},
            "device_ms_total": dev_ms,
            "hbm_in_use_gb": (free0 - ctx.mem_info()[0]) / 1e9,
            "other_kernels": kern,
        },
        "roofline": roof,
    }
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.dt, args.rtol)
    if not args.no_other_configs:
        p.close()  # 45 GB back before the next problems
        try:
            out["config"]["other_baseline_configs"] = other_baseline_configs(fv, ctx)
        except Exception as e:  # never lose the headline line to a side measurement
            out["config"]["other_baseline_configs"] = "failed: %r" % (e,)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
