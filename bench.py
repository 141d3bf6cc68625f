#!/usr/bin/env python3
"""bench.py — DoF-updates/s of the implicit transient hot path on MI355X.

A "step" is one backward-Euler time step of the whole grid: RHS build + the
Jacobi-PCG solve of (D/dt + A) u+ = D u/dt + b, everything resident in HBM.
Workload (BASELINE.json configs[4] / the 10^8-cell target of
`north_star`): structured box of ns^3 cells (default 464^3 = 9.99e7), 1000 x 1000
x 100 m, K = 1e-5 m/s, Ss = 0.1, Dirichlet head 1e3 on the four lateral faces,
u0 = 1e3, one pumping well (-1e-3 m^3/s) down the centre column, fixed dt = 60 s,
PCG rtol 1e-10.  The grid is generated and assembled on the device; by default the problem is lean (FV_OPT_LEAN_SETUP: no faces / CSR in HBM,
the same doubles and kernels), the line's `csr_route_block` times the route through faces and CSR beside it (--lean off swaps the two).

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
KERNEL_NAMES = {  # fv_spmv_form id -> kernel(s) that ran
    5: "spmv_sell_kernel<true> (SELL-64: per 64-row group lane-major blocks of 64 values + 64 sixteen-bit column offsets, the diagonal first; one wave per group)",
    0: "spmv_wstream_kernel<512,true,true> (wave-private CSR stream)",
    1: "spmv_dia_kernel<true,true,false> (sliced-DIA, slice by slice: x fits the last-level cache at this size)",
    2: "spmv_dia_march_kernel<true,true,true> (plane-marching sliced-DIA, 16-byte window accesses)",
    4: "spmv_symdia_tile_kernel<true,4> (symmetric form, tiled: a block owns 1024 rows of a plane and marches through the planes; 3 upper diagonals streamed once, lower arms and the in-plane x arms through LDS, +-plane arms in registers, the diagonal re-derived from the six arms where the row sum is zero; first/last plane by spmv_dia_kernel)",
    3: "spmv_symdia_march_kernel<true,4,true,true> (symmetric plane-marching: 3 upper diagonals streamed, lower arms from the upper arrays, the diagonal re-derived from the six arms where the row sum is zero and streamed elsewhere; first/last plane by spmv_dia_kernel)",
}


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
    cells, 30 steps (10-20 s of CPU).  Only the stepping loop is timed, as for the GPU.
    Beside it the path north_star names — `linearsolver = (A, b, x0) -> A \\ b` (test/ode.jl:36), i.e. a sparse direct
    solve per step — on 32^3 cells x 3 steps with SciPy's SuperLU standing in for Julia's UMFPACK/CHOLMOD (`direct`)."""
    from oracle import fv_oracle as o

    def problem(ns):
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
        return A, b, len(vol)

    ns, steps = [256, 256, 256], 30
    A, b, N = problem(ns)
    u0 = np.full(A.n, 1e3)
    iters = []

    def solver(Am, rhs, x0):
        x, ch = o.cg(Am, rhs, x0=x0, tol=rtol, maxiter=1000)
        iters.append(ch.iters)
        return x

    t0 = time.perf_counter()
    us, ts = o.backwardeulerintegrate_generic(u0, A, b, dt, 0.0, dt * steps, stepper=o.fixedbackwardeulerstep, linearsolver=solver)
    sec = time.perf_counter() - t0
    out = {
        "value": N * steps / sec,
        "unit": "DoF-updates/s",
        "cores": 1,
        "cores_available": os.cpu_count(),
        "kind": "port",
        "sample": "oracle (C restatement of the reference path, unpreconditioned CG rtol %.0e) on %dx%dx%d cells x %d steps, same cell size/K/Ss/dt/BCs; %.1f s; %.1f CG iters/step" % (rtol, ns[0], ns[1], ns[2], steps, sec, float(np.mean(iters))),
    }
    del us
    try:  # the same steps on the host's cores (OpenMP; oracle/fv_oracle_mt.c): SURVEY 8(d) asks for both figures, core count printed
        import ctypes as C

        mt = C.CDLL(os.path.join(ROOT, "oracle", "_build", "libfvoracle_mt.so"))
        avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        nthreads = max(1, min(avail, int(os.environ.get("FV_BENCH_CPU_THREADS", "16"))))  # (a one-GPU box of this pool has a 16-core share)
        _, n1, n2, aol, vol = o.regulargrid(*spacing_box(ns), ns, want_coords=False)
        dn, src = box_setup(ns)
        A0 = o.assembleA(n1, n2, aol, np.full(len(aol), 1e-5), src, dn, np.full(len(dn), 1e3))
        _, n2f = o.getfreenodes(len(vol), dn)
        dvol = np.ascontiguousarray((0.1 * vol)[np.asarray(o.freenodei2nodei(n2f)) - 1])
        del n1, n2, aol
        vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        u = np.full(A0.n, 1e3)
        itot = C.c_int64()
        t0 = time.perf_counter()
        rc = mt.fvo_mt_fixed_steps(C.c_int64(A0.n), vp(A0.colptr), vp(A0.rowval), vp(A0.nzval), vp(dvol), vp(b), vp(u), C.c_double(dt), C.c_int64(steps),
                                   C.c_double(rtol), C.c_int64(1000), C.byref(itot), C.c_int(nthreads))
        secm = time.perf_counter() - t0
        if rc != 0:
            raise RuntimeError("fvo_mt_fixed_steps returned %d" % rc)
        out["all_cores"] = {"value": N * steps / secm, "unit": "DoF-updates/s", "cores": nthreads, "cores_available": os.cpu_count(), "kind": "port-threaded",
                            "sample": "the same %dx%dx%d cells x %d steps, the same CG with OpenMP over the rows (gather product on the symmetric matrix, dot products summed "
                                      "per thread: oracle/fv_oracle_mt.c); %.1f s; %.1f CG iters/step; the reference itself runs on one thread" % (ns[0], ns[1], ns[2], steps, secm, itot.value / steps)}
        del A0, u, dvol
    except Exception as e:
        out["all_cores"] = "failed: %r" % (e,)
    del A, b
    try:  # the backslash path: one sparse LU per step (the reference factorises the shifted matrix anew in every step)
        nsd, stepsd = [32, 32, 32], 3
        A, b, Nd = problem(nsd)
        t0 = time.perf_counter()
        o.backwardeulerintegrate_generic(np.full(A.n, 1e3), A, b, dt, 0.0, dt * stepsd, stepper=o.fixedbackwardeulerstep, linearsolver=o.directlinearsolver)
        secd = time.perf_counter() - t0
        out["direct"] = {"value": Nd * stepsd / secd, "unit": "DoF-updates/s", "cores": 1, "kind": "port-direct",
                         "sample": "linearsolver = A \\ b (test/ode.jl:36) as scipy.sparse.linalg.splu per step on %dx%dx%d cells x %d steps; %.1f s" % (nsd[0], nsd[1], nsd[2], stepsd, secd)}
    except Exception as e:
        out["direct"] = "failed: %r" % (e,)
    return out


def other_baseline_configs(fv, ctx):
    """The other single-GPU configurations of BASELINE.json (SURVEY 8d inputs), one short measurement each, reported inside
    `config` — the headline `value` stays the 10^8-cell run.  The same code as tools/config_rates.py."""
    workloads = fv.workloads

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
    sec_amg = time.perf_counter() - t0
    ms0 = p.bench_spmv(0.0, 20)
    _, form_name0, form_bytes0 = p.spmv_form()
    row("box_model 256^3 steady, sigma=3 field, AMG-PCG rtol 1e-8 (hierarchy set-up included)", p.N, 1, sec_amg,
        pcg_iters=int(ch.iters), converged=bool(ch.isconverged), amg_levels=int(len(levels[0])),
        roofline={"bound": "hbm", "kernel": "the level-0 product of the cycle and of the PCG (%s)" % form_name0, "achieved": form_bytes0 / (ms0 * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                  "unit": "GB/s", "frac": form_bytes0 / (ms0 * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "avg_launch_ms": ms0, "launches": 20,
                  "bytes_per_row": form_bytes0 / max(p.n, 1),
                  "note": "a V-cycle iteration is ~100 launches over six levels (DESIGN 4a: 1.0 of its 2.55 ms are level-0 kernels at this rate, the rest coarse-level products "
                          "within a fifth of their byte bounds); no single byte model covers the solve, so the figure is the dominant kernel's"})
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
    f0, m0 = p.fused_form()[0], p.bytes_moved()
    iters, info, _ = p.run_fixed(st, 60.0, 100, 1e-10)
    ctx.synchronize()
    sec = time.perf_counter() - t0
    launches, brow, bl = p.fused_form()
    row("watertable-like 216^3 transient, 100 steps, dt=60s, Jacobi-PCG rtol 1e-10", p.N, 100, sec,
        pcg_iters_per_step=float(iters.mean()), converged=bool(info.converged), roofline=step_regime_roofline(p, iters, sec / 100, launches - f0, brow, bl, p.bytes_moved() - m0))
    p.close()
    # ... "plus the same field at sigma = 1" (SURVEY 8d): node log K = log(1e-5) + g, the Gaussian field of the 256^3 case at this size
    p = fv.Problem.regulargrid([0.0, 0.0, 0.0], [1000.0, 1000.0, 100.0], ns, dn, ctx)
    Kf, _ = gaussian_face_logK(fv, p, ns, 1.0, seed=0)
    p.assemble(Kf, src, np.full(len(dn), 1e3), None, True)
    del Kf
    st = p.transient_begin(0.1, None, np.full(p.N, 1e3))
    p.run_fixed(st, 60.0, 3, 1e-10)
    ctx.synchronize()
    t0 = time.perf_counter()
    f0, m0 = p.fused_form()[0], p.bytes_moved()
    iters, info, _ = p.run_fixed(st, 60.0, 100, 1e-10)
    ctx.synchronize()
    sec = time.perf_counter() - t0
    launches, brow, bl = p.fused_form()
    row("watertable-like 216^3 transient, sigma=1 Gaussian log-K field, 100 steps, dt=60s, Jacobi-PCG rtol 1e-10", p.N, 100, sec,
        pcg_iters_per_step=float(iters.mean()), converged=bool(info.converged), roofline=step_regime_roofline(p, iters, sec / 100, launches - f0, brow, bl, p.bytes_moved() - m0))
    p.close()
    # configs[3]: fractures-like 5 M cells, irregular CSR (cells numbered at random inside each fracture), 100 implicit steps.
    # The mesh is handed over as it is numbered; fv_problem_create re-numbers the free cells for locality by itself
    # (reverse Cuthill-McKee inside the library, invisible at the ABI).  Second row: the same with that switched off.
    w = workloads.fractures_like(20, 500, seed=0)
    for label, mode in (("as numbered; fv_problem_create re-numbers the free cells for locality inside the library", 1),
                        ("as numbered, the library's re-numbering switched off (fv_ctx_set_option(FV_OPT_REORDER, 0))", 0)):
        ctx.set_option(1, mode)  # FV_OPT_REORDER
        try:
            t0 = time.perf_counter()
            p = fv.Problem.create((w["node1"], w["node2"]), w["aol"], w["N"], w["dnodes"], ctx)
            t_create = time.perf_counter() - t0
        finally:
            ctx.set_option(1, 1)
        info = p.reorder_info()
        p.assemble(w["K"], np.zeros(w["N"]), w["dheads"])
        st = p.transient_begin(1e-9, w["volumes"], np.full(w["N"], 1.5e6))
        p.run_fixed(st, 1.0, 3, 1e-10, maxiter=5000)
        ctx.synchronize()
        t0 = time.perf_counter()
        f0, m0 = p.fused_form()[0], p.bytes_moved()
        iters, sinfo, _ = p.run_fixed(st, 1.0, 100, 1e-10, maxiter=5000)
        ctx.synchronize()
        sec = time.perf_counter() - t0
        launches, brow, bl = p.fused_form()
        regime = step_regime_roofline(p, iters, sec / 100, launches - f0, brow, bl, p.bytes_moved() - m0)
        ms = p.bench_spmv(1.0, 20)
        form_id, form_name, form_bytes = p.spmv_form()
        row("fractures-like 5M cells (irregular CSR, %s), transient, 100 steps, dt=1s, Jacobi-PCG rtol 1e-10" % label, p.N, 100, sec,
            pcg_iters_per_step=float(iters.mean()), converged=bool(sinfo.converged), problem_create_s=t_create,
            renumbered=info["reordered"], renumbering_s=info["seconds"], mean_face_distance=[info["mean_before"], info["mean_after"]],
            spmv={"form": form_name, "ms": ms, "GB/s": form_bytes / (ms * 1e-3) / 1e9, "frac_of_peak": form_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}, roofline=regime)
        p.close()
    return rows


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start N fresh child processes of this script, one per GPU, with the
    torch.distributed.run environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT), relay rank 0's single JSON
    line and return non-zero if any rank fails.  The parent never loads libfvhip or touches HIP (a process that has
    initialised the GPU must not be replaced or forked on this pool), and a failed rank takes the others down with it
    instead of leaving them waiting in a collective."""
    import signal
    import socket
    import subprocess

    with socket.socket() as s:  # a free rendezvous port
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, start_new_session=True))
    rc = 0
    out0 = b""
    try:
        pending = set(range(n))
        import selectors
        sel = selectors.DefaultSelector()
        sel.register(procs[0].stdout, selectors.EVENT_READ)
        eof0 = False
        while pending:
            if not eof0:
                for key, _ in sel.select(timeout=0.2):
                    chunk = os.read(key.fileobj.fileno(), 65536)
                    if chunk:
                        out0 += chunk
                    else:
                        eof0 = True
                        sel.unregister(key.fileobj)
            else:
                time.sleep(0.2)
            for r in sorted(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    print("bench.py: rank %d exited with code %d; stopping the other ranks" % (r, code), file=sys.stderr)
                    for q in pending:
                        try:
                            os.killpg(procs[q].pid, signal.SIGTERM)
                        except ProcessLookupError:
                            pass
        if not eof0:
            out0 += procs[0].stdout.read()
    finally:
        for pr in procs:
            if pr.poll() is None:
                try:
                    os.killpg(pr.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
    lines = [l for l in out0.decode(errors="replace").splitlines() if l.strip().startswith("{")]
    if rc == 0 and len(lines) != 1:
        print("bench.py: rank 0 printed %d JSON lines instead of one" % len(lines), file=sys.stderr)
        rc = 3
    for l in lines:  # a failed run prints no headline number on stdout (a driver that parses the last JSON line must not find one)
        print(l, file=sys.stdout if rc == 0 else sys.stderr)
    sys.stdout.flush()
    return rc


def main():
    # multi-process GPU work on this pool needs dmabuf IPC (RCCL's peer mappings fail with the legacy mode); the launcher
    # normally exports it already — set before anything touches HIP
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)  # 0.65 s of stepping: one hiccup of the box no longer decides the number
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--repeats", type=int, default=3, help="timed regions of --steps steps each; value = the median region (all are listed)")
    ap.add_argument("--ns", type=int, default=464, help="cells per dimension of the box (464 -> 9.99e7 cells)")
    ap.add_argument("--dt", type=float, default=60.0)
    ap.add_argument("--rtol", type=float, default=1e-10)
    ap.add_argument("--maxiter", type=int, default=2000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short runs of BASELINE.json's other single-GPU configurations")
    ap.add_argument("--no-multi-iteration", action="store_true", help="skip the second measured block (dt = 1 h, ~10 PCG iterations per step)")
    ap.add_argument("--no-hetero", action="store_true", help="skip the same workload with a heterogeneous conductivity (the matrix then streams as doubles: 73 B per row)")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-kernel HIP-event timing inside the timed region")
    ap.add_argument("--no-lean-block", action="store_true", help="skip the same steps on a lean problem (no faces / CSR in HBM)")
    ap.add_argument("--lean", choices=("auto", "on", "off"), default="on",
                    help="FV_OPT_LEAN_SETUP for the bench problem: no face arrays / incident lists / CSR in HBM — the same doubles, the same kernels (tests/test_gpu_lean.py); "
                         "off: the route through faces and CSR (the side block of the line measures the other route); auto: lean only where the CSR would not fit int32 offsets")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: this process (which has not touched HIP) starts the N ranks itself
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    if world != args.gpus:
        print("bench.py: --gpus %d but WORLD_SIZE=%d: launch one process per GPU (or let bench.py start them: unset WORLD_SIZE)" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    from __graft_entry__ import load_package

    fv = load_package()
    # (A/B knobs of libfvhip: FV_TUNE=key=value,... in the environment, read by the library itself — csrc/fv_tune.h)
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
    p = fv.Problem.regulargrid(mins, maxs, ns, dn, ctx, lean={"auto": None, "on": True, "off": False}[args.lean])
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
        p.profile(2)  # HIP event pairs around every K1 / fused-step launch of the timed regions (and nothing else: an event is a barrier)
    fused0 = p.fused_form()[0]
    regions, iters_all = [], []
    for rep in range(max(args.repeats, 1)):  # each region: exactly --steps steps between two synchronisations; consecutive fixed-dt
        ctx.synchronize()                     # runs go on where the previous one stopped (fv_problem::resume)
        t0 = time.perf_counter()
        iters, info, dev_ms = p.run_fixed(state, args.dt, args.steps, args.rtol, args.maxiter)
        ctx.synchronize()
        regions.append(time.perf_counter() - t0)
        iters_all.append(iters.copy())
    sec = float(np.median(regions))
    iters = np.concatenate(iters_all)
    fused_launches = p.fused_form()[0] - fused0
    prof = p.profile_get() if not args.no_profile else None
    p.profile(False)
    if prof is not None and not fused_launches:  # the other kernels of the step: 16 more steps after the timed region with all event pairs on
        p.profile(1)
        p.run_fixed(state, args.dt, 16, args.rtol, args.maxiter)
        extra = p.profile_get()
        p.profile(False)
        prof["update"], prof["pupdate"] = extra["update"], extra["pupdate"]

    value = p.N * args.steps / sec
    roof, kern = roofline_block(p, prof, float(np.mean(iters)), args.ns, fused_launches, len(iters))

    out = {
        "metric": "DoF-updates/s (cells\u00d7steps) implicit transient; SpMV HBM GB/s vs peak",
        "value": value,
        "unit": "DoF-updates/s",
        "n_gpus": 1,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": sec / args.steps * 1e3,
        "repeats": {"regions": len(regions), "ms_per_step_each": [r / args.steps * 1e3 for r in regions], "value_is": "the median region",
                    "spread": (max(regions) - min(regions)) / sec},
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": "synthetic %d^3 box (%.3g cells), transient, fixed dt=%gs, Jacobi-PCG rtol %.0e, lateral Dirichlet + centre well (SURVEY 8d / BASELINE configs[4] on one GPU)" % (args.ns, p.N, args.dt, args.rtol),
            "cells": p.N, "unknowns": p.n, "nnz": p.nnz, "faces": p.F,
            "pcg_iters_per_step": float(np.mean(iters)), "pcg_iters_total": int(np.sum(iters)),
            "step_form": ("fused: one launch per step does the vector update of step k and the product of step k + 1 (fv_fused.hip), %d of %d steps" % (fused_launches, len(iters)))
                         if fused_launches else "K1 (product) + K2S (vector update and next set-up) per step",
            "resume_runs": True, "residual_refresh_every": 128,  # consecutive run_fixed calls continue the carried residual; a fresh b - A u every 128 steps
            "last_relres": info.relres, "converged": bool(info.converged),
            "device": name, "compute_units": cus,
            "setup_s": {"grid+symbolic": t_symbolic, "assemble": t_assemble, "total": t_setup},
            "device_ms_total": dev_ms,
            "hbm_in_use_gb": (free0 - ctx.mem_info()[0]) / 1e9,
            "lean_setup": bool(p.lean),  # FV_OPT_LEAN_SETUP: the storage forms filled from rows formed on the fly (no faces, no CSR in HBM)
            "other_kernels": kern,
        },
        "roofline": roof,
    }
    if not args.no_multi_iteration:
        try:
            out["config"]["multi_iteration"] = multi_iteration_block(p, args)
        except Exception as e:  # never lose the headline line to a side measurement
            out["config"]["multi_iteration"] = "failed: %r" % (e,)
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.dt, args.rtol)
    if not args.no_lean_block and 7 * p.N < 2**31 - 3:  # the other route of the set-up on the same workload (where a CSR with int32 offsets exists at all)
        other_lean = not p.lean
        p.close()
        key = "lean_setup_block" if other_lean else "csr_route_block"
        try:
            out["config"][key] = lean_block(fv, ctx, args, mins, maxs, ns, dn, src, other_lean)
        except Exception as e:
            out["config"][key] = "failed: %r" % (e,)
    if not args.no_hetero:
        p.close()
        try:
            out["config"]["heterogeneous_K"] = hetero_block(fv, ctx, args)
        except Exception as e:
            out["config"]["heterogeneous_K"] = "failed: %r" % (e,)
    if not args.no_other_configs:
        p.close()  # 45 GB back before the next problems
        try:
            out["config"]["other_baseline_configs"] = other_baseline_configs(fv, ctx)
        except Exception as e:
            out["config"]["other_baseline_configs"] = "failed: %r" % (e,)
    print(json.dumps(out))


def gaussian_face_logK(fv, p, ns, sigma, seed=0):
    """Face log-conductivities of SURVEY 8d's heterogeneous inputs: node log K = log(1e-5) + sigma g, g the unit-variance Gaussian field;
    face value = arithmetic mean of the two node values (grid.jl:27, logtransform = true)."""
    g, where = fv.workloads.smooth_gaussian_field(ns, seed=seed), "host"
    logk = np.log(1e-5) + sigma * g
    del g
    if p.lean or p.F > 1 << 30:  # (no face list to ask for, or 16 F bytes of face ends: the closed form of the grid's face list)
        return fv.workloads.grid_face_means(ns, logk), where
    n1, n2 = np.empty(p.F, np.int64), np.empty(p.F, np.int64)
    p.check(fv.load().fv_problem_get_grid(p.handle, n1.ctypes.data, n2.ctypes.data, None, None))
    Kf = fv.nodehycos2neighborhycos((n1, n2), logk, True)
    return Kf, where


def hetero_face_K(ns, F, N):
    """The bench's heterogeneous conductivity, one value per face: K = 1e-5 exp(g), g a smooth field of unit variance (sums of
    sines of the cell indices), evaluated from the cell the face list reaches the face in (regulargrid emits up to three faces
    per cell, cell by cell: face f belongs to cell ~ f N / F) — sigma = 1 without the 5 GB of face ends on the host."""
    n1, n2, n3 = ns
    K = np.empty(F)
    for lo in range(0, F, 1 << 24):  # (in pieces: 3e8 faces)
        hi = min(lo + (1 << 24), F)
        cell = (np.arange(lo, hi, dtype=np.float64) * (N / F)).astype(np.int64)
        i3 = cell % n3
        i2 = (cell // n3) % n2
        i1 = cell // (n3 * n2)
        g = (np.sin(2 * np.pi * i1 / 97.0) + np.sin(2 * np.pi * i2 / 61.0) + np.sin(2 * np.pi * i3 / 43.0)) / np.sqrt(1.5)
        K[lo:hi] = 1e-5 * np.exp(g)
        del cell, i1, i2, i3, g
    return K


def step_regime_roofline(p, iters, sec_per_step, fused_launches, fused_row_bytes, fused_launch_bytes, moved=None):
    """A roofline-shaped object for the regime a block of fixed-dt steps ran in: the bytes one step must move with
    every array of every launch touched once, over the wall time of a step.
    One-iteration steps through the fused launch: fv_fused_form's bytes.  Steps of several PCG iterations (the carried residual, then
    the loop of DESIGN 4d): K0' (carried set-up: x, x_prev, z or r, D, M^-1 in, r, p out = 64 n) + K1 (the tiled SpMV's form, with p.q)
    + the first vector update (x, z, p, q, M^-1 in, x, z out = 56 n; 49 with M^-1 as a code byte) + per further iteration the loop
    form fv_loop_form reports (fused pass with the lagging x-update + the slim vector update: 105 n, 83 / 76 with codes) — or K1 + K2 + K3 where that loop
    is not taken."""
    n = p.n
    its = np.asarray(iters, dtype=np.float64)
    mean_it = float(its.mean())
    if fused_launches >= 0.8 * len(its) and mean_it <= 1.0:
        bytes_step = float(fused_launch_bytes)
        model = "one fused launch per step (fv_fused_form): %d B per row" % fused_row_bytes
    else:
        form_id, form_name, form_bytes = p.spmv_form()
        loop = p.loop_form()
        if loop in (89, 67):  # one launch per PCG iteration (fv_step_form says what the step moved outside its loop iterations)
            (b_setup, b_first, b_flush), _ = p.step_form()
            bytes_step = float(np.mean(np.where(its >= 1, b_setup + b_first + np.maximum(its - 1, 0) * loop + b_flush, b_setup))) * n
            model = ("per step: set-up %d n (%s) + first pass %d n (direction = z: z, the matrix, a code byte in, w out) + per further iteration ONE launch of %d n "
                     "(z, w, p, x in; z', p', w', x out; the three upper diagonals%s; a code byte — verdict, alpha, beta from the previous launch's sums) + flush %d n; "
                     "weighted with the iteration count of every step" %
                     (b_setup, "the previous step's pending update + the carried residual in one pass" if b_setup == 65 else "carried residual K0'" if b_setup == 64 else "a fresh residual",
                      b_first, loop, " as 16-bit codes" if loop == 67 else "", b_flush))
        elif loop:
            first_upd = 49 if loop in (76, 98) else 56  # (the loop's first vector update is the classic one: x, z, p, q, M^-1 in, x, z out)
            per_it = loop * n
            first = 64 * n + form_bytes + first_upd * n
            upd = first_upd
            bytes_step = float(np.mean(np.where(its >= 1, first + np.maximum(its - 1, 0) * per_it, 64 * n)))
            model = ("per step: carried set-up K0' 64 n + K1 (%s) %d B per row + first vector update %d n, then per further iteration the fused pass (with the lagging x-update) + the "
                     "slim vector update = %d n; weighted with the iteration count of every step" % (form_name, form_bytes // max(n, 1), upd, loop))
        else:
            per_it = form_bytes + 88 * n
            bytes_step = float(np.mean(64 * n + its * per_it))
            model = "per step: carried set-up 64 n + per iteration K1 (%s) + 88 n for K2 + K3" % form_name
    source = "the regime's byte model (below)"
    if moved is not None and moved > 0:  # the library's own running total over the timed steps (fv_step_form): every launch that ran, by the iterations that ran
        model_bytes = bytes_step
        bytes_step = float(moved) / max(len(its), 1)
        source = "fv_step_form's running total over the timed steps (the model below gives %.4g B per row and step)" % (model_bytes / max(n, 1))
    gbs = bytes_step / sec_per_step / 1e9
    stamp = regime_traffic_stamp(p, fused_launches >= 0.8 * len(its) and mean_it <= 1.0, fused_launch_bytes)
    out_extra = {}
    if stamp and stamp.get("whole_step"):  # one launch per step: the PMC bytes of that launch are the step's
        out_extra = {"traffic": stamp["bytes"], "traffic_source": stamp["source"], "frac_traffic": stamp["bytes"] / sec_per_step / 1e9 / HBM_PEAK_GBS}
    elif stamp:
        out_extra = {"loop_launch": {"kernel": stamp["kernel"], "form_bytes": stamp["form_bytes"], "traffic": stamp["bytes"], "traffic_source": stamp["source"],
                                     "ratio": stamp["bytes"] / stamp["form_bytes"]}}
    roof = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None, "bytes_from": source,
            "algorithmic_bytes_per_step": bytes_step, "bytes_per_row_per_step": bytes_step / max(n, 1), "pcg_iters_per_step": mean_it,
            "measured": "wall time of the stepping loop / steps (all launches of a step, polls included)", "bytes_model": model}
    roof.update(out_extra)
    return roof


def regime_traffic_stamp(p, one_launch_steps, fused_launch_bytes):
    """The PMC stamp under profiles/ (spmv_traffic.json) for the launch that dominates a regime, when kernel, form and size match: the fused
    step on chunks with the matrix as doubles (one launch per step), or the one-launch PCG iteration (doubles / codes)."""
    tfile = os.path.join(ROOT, "profiles", "spmv_traffic.json")
    if not os.path.exists(tfile):
        return None
    try:
        tj = json.load(open(tfile))
        if one_launch_steps:
            if p.fused_traversal() != 1:
                return None
            for key in ("fused_doubles_464", "fused_464"):
                t = tj.get(key)
                if t and abs(t["form_bytes"] - fused_launch_bytes) <= 0.005 * fused_launch_bytes:
                    return dict(t, whole_step=True)
            return None
        loop = p.loop_form()
        t = tj.get({89: "ploop_doubles_464", 67: "ploop_coded_464"}.get(loop, ""))
        if t and abs(t["form_bytes"] - loop * p.n) <= 0.005 * loop * p.n:
            return t
    except Exception:
        pass
    return None


def hetero_block(fv, ctx, args):
    """The headline workload with a smooth heterogeneous conductivity (sigma = 1 in log K) instead of one value: SURVEY 8d names both
    for the 10^8-cell configuration.  The operator's diagonals then take as many values as there are faces, so the fused step
    streams them as doubles (73 B per row) where the homogeneous headline reads 16-bit codes (51)."""
    ns = [args.ns] * 3
    mins, maxs = spacing_box(ns)
    dn, src = box_setup(ns)
    p = fv.Problem.regulargrid(mins, maxs, ns, dn, ctx)
    t0 = time.perf_counter()
    K, where = gaussian_face_logK(fv, p, ns, 1.0, seed=0)
    t_field = time.perf_counter() - t0
    p.assemble(K, src, np.full(len(dn), 1e3), None, True)
    del K
    st = p.transient_begin(0.1, None, np.full(p.N, 1e3))
    p.run_fixed(st, args.dt, max(args.warmup, 3), args.rtol, args.maxiter)
    f0 = p.fused_form()[0]
    secs, moved = [], []
    for rep in range(3):
        ctx.synchronize()
        m0 = p.bytes_moved()
        t0 = time.perf_counter()
        iters, info, _ = p.run_fixed(st, args.dt, args.steps, args.rtol, args.maxiter)
        ctx.synchronize()
        secs.append(time.perf_counter() - t0)
        moved.append(p.bytes_moved() - m0)
    sec = float(np.median(secs))
    launches, brow, bl = p.fused_form()
    out = {"workload": "same %d^3 box, node log K = log(1e-5) + 1.0 g with g SURVEY 8d's unit-variance Gaussian field (separable box smoothing of N(0,1) noise, radius 10 cells, 3 passes, seed 0; "
                       "generated on the %s in %.1f s), face K = exp(arithmetic mean of the node values) (grid.jl:27), %d steps x 3 regions" % (args.ns, where, t_field, args.steps),
           "dof_updates_per_s": p.N * args.steps / sec, "ms_per_step": sec / args.steps * 1e3, "ms_per_step_each": [s / args.steps * 1e3 for s in secs],
           "pcg_iters_per_step": float(iters.mean()), "converged": bool(info.converged), "last_relres": info.relres,
           "fused_launches": launches - f0}
    out["roofline"] = step_regime_roofline(p, iters, sec / args.steps, launches - f0, brow, bl, moved[int(np.argsort(secs)[1])])
    # ... and at a time step short enough for this field's stiffest cells to converge in one PCG iteration: the regime of the
    # headline, where the fused step streams the matrix as doubles
    if out["pcg_iters_per_step"] > 1.0:
        dt1, n1s = args.dt / 8.0, min(args.steps, 20)  # (few steps from the initial state again: hundreds of them at this dt end below the tolerance, 0 iterations)
        st = p.transient_begin(0.1, None, np.full(p.N, 1e3))
        p.run_fixed(st, dt1, 8, args.rtol, args.maxiter)
        f1 = p.fused_form()[0]
        secs1, its1, moved1 = [], [], []
        for rep in range(3):
            ctx.synchronize()
            m0 = p.bytes_moved()
            t0 = time.perf_counter()
            it1, info1, _ = p.run_fixed(st, dt1, n1s, args.rtol, args.maxiter)
            ctx.synchronize()
            secs1.append(time.perf_counter() - t0)
            moved1.append(p.bytes_moved() - m0)
            its1.append(float(it1.mean()))
        sec1 = float(np.median(secs1))
        l1, brow1, bl1 = p.fused_form()
        out["one_iteration_regime"] = {"dt": dt1, "steps": n1s, "dof_updates_per_s": p.N * n1s / sec1, "ms_per_step": sec1 / n1s * 1e3,
                                       "ms_per_step_each": [s / n1s * 1e3 for s in secs1], "pcg_iters_per_step": float(np.mean(its1)),
                                       "converged": bool(info1.converged), "fused_launches": l1 - f1,
                                       "roofline": step_regime_roofline(p, it1, sec1 / n1s, (l1 - f1) / 3.0, brow1, bl1, moved1[int(np.argsort(secs1)[1])])}
    p.close()
    return out


def roofline_block(p, prof, iters_per_step, ns, fused_launches=0, nsteps=0):
    """The `roofline` object for the kernel that dominates a step.  In the one-iteration regime on an operator with the tiled
    symmetric form that is the fused step (fused_step_kernel: vector update of step k + product of step k + 1, fv_fused_form's
    bytes); its `spmv` sub-object then gives K1 alone (the PCG SpMV q = (A + D/dt) p with the p.q epilogue, which every other
    regime runs), measured back to back right after the timed regions.  Otherwise the object is K1's:
    achieved / frac use the bytes one launch of the storage form that ran has to move with every array touched once
    (fv_spmv_form: e.g. 32 n of matrix + 16 n of vectors for the symmetric plane-marching form) over the HIP-event average of
    the live launches inside the timed region — a real HBM rate, never above what the hardware does.  SURVEY 8d's CSR
    accounting (12 nnz + 20 n, what a general CSR SpMV of this operator would move) is reported next to it as
    `effective_csr`; it exceeds the real rate by construction for forms that store no column indices / half the matrix.
    `traffic` is the PMC measurement committed under profiles/ for exactly this kernel and size (null when the committed
    figure belongs to another kernel)."""
    form_id, form_name, form_bytes = p.spmv_form()
    csr_bytes = 12 * p.nnz + 20 * p.n
    roof = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
            "kernel": "K1 q=(A+D/dt)p with p.q: %s" % KERNEL_NAMES.get(form_id, form_name), "form": form_name,
            "algorithmic_bytes_per_launch": form_bytes,
            "bytes_model": "bytes the storage form must move, every array once (fv_spmv_form); the CSR accounting of SURVEY 8d is in effective_csr"}
    kern = {}
    if fused_launches and prof and prof["spmv_dot"][1] > 0 and fused_launches >= 0.8 * nsteps:
        _, fbytes_row, fbytes = p.fused_form()
        ms, cnt = prof["spmv_dot"]
        t = ms / cnt * 1e-3
        k1 = dict(roof)
        k1_ms = p.bench_spmv(1.0 / 60.0, 20)
        k1.update(achieved=form_bytes / (k1_ms * 1e-3) / 1e9, frac=form_bytes / (k1_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, avg_launch_ms=k1_ms, launches=20,
                  measured="20 launches back to back after the timed regions (inside the stepping loop the product is part of the fused launch)",
                  effective_csr={"bytes_per_launch": csr_bytes, "GB/s": csr_bytes / (k1_ms * 1e-3) / 1e9, "frac": csr_bytes / (k1_ms * 1e-3) / 1e9 / HBM_PEAK_GBS})
        chunks = p.fused_traversal() == 1
        what = ("x_out = x + alpha z, z' = z + alpha v, the convergence and set-up sums of step k, then q' = (A + D/dt) z' with z'.q' of step k + 1, "
                "stored as v' = -M^-1 (q' - D z'/dt); ")
        roof = {"bound": "hbm", "achieved": fbytes / t / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": fbytes / t / 1e9 / HBM_PEAK_GBS, "traffic": None,
                "kernel": ("fused_chunkd_kernel<512, 4, 0, 3, 2, 2>: " + what + "contiguous chunks of ~3400 rows of a plane (no column halos) marching through the planes, the matrix streamed as doubles: "
                           "z' chunk and the +line diagonal double-buffered in LDS, the +1 diagonal by a wave shift, the +plane diagonal carried in registers, loads issued two to three row pairs ahead; "
                           "diagonal and M^-1 re-derived from the arms (rows next to a Dirichlet cell: the stored diagonal)") if chunks and fbytes_row >= 60 else
                          ("fused_chunk_kernel<512, 5, 0>: " + what + "contiguous chunks of ~5100 rows of a plane (no column halos) marching through the planes, z' chunk "
                           "double-buffered and the 16-bit matrix words of two planes in LDS, +-plane arms in registers, diagonal and M^-1 re-derived from the arms "
                           "(rows next to a Dirichlet cell: out of a table by a per-row code); the first / last plane's products by the same launch where their rows' diagonals fit the table (the bench box), else by spmv_dia_kernel") if chunks else
                          ("fused_step_kernel<16, 0, %s>: " % ("true" if fbytes_row < 60 else "false") + what +
                           "2-D tiles of 16 lines x 128 columns marching through the planes, "
                           "z' tile and U1/U2 ring in LDS, +-plane arms in registers, diagonal and M^-1 re-derived from the arms; first/last plane's products by spmv_dia_kernel"),
                "form": "fused step, v-form (x, z, v in; x_out, z', v' out; 3 upper diagonals%s; storage codes): %d B per row" % (" as one 16-bit word of codes per row (one conductivity: each diagonal takes a handful of values)" if fbytes_row < 60 else "", fbytes_row),
                "algorithmic_bytes_per_launch": fbytes,
                "bytes_model": "every array once (fv_fused_form): 48 n of vectors + 1 n of storage codes + 24 B of matrix per row whose product the kernel forms (2 B where the matrix comes as codes; + 8 where the diagonal is streamed); the unfused pair K1 + K2S moves 41 n + 49 n",
                "avg_launch_ms": ms / cnt, "launches": cnt, "spmv": k1}
        tfile = os.path.join(ROOT, "profiles", "spmv_traffic.json")
        if os.path.exists(tfile):
            try:  # K1's own stamp
                t1 = json.load(open(tfile)).get(str(ns))
                if t1 and t1.get("form") == form_id and abs(t1.get("form_bytes", 0) - form_bytes) <= 0.01 * form_bytes:
                    k1["traffic"] = t1["bytes"]
                    k1["traffic_source"] = t1.get("source")
                    k1["frac_traffic"] = t1["bytes"] / (k1_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
            except Exception:
                pass
        if os.path.exists(tfile):
            try:
                tj = json.load(open(tfile)).get(("fused_%d" if chunks or fbytes_row >= 60 else "fused_tiles_%d") % ns)
                if tj and abs(tj.get("form_bytes", 0) - fbytes) <= 0.01 * fbytes and ("chunk" in tj.get("kernel", "")) == chunks:
                    roof["traffic"] = tj["bytes"]
                    roof["traffic_source"] = tj.get("source")
                    roof["frac_traffic"] = tj["bytes"] / t / 1e9 / HBM_PEAK_GBS
            except Exception:
                pass
        return roof, kern
    if prof and prof["spmv_dot"][1] > 0:
        ms, cnt = prof["spmv_dot"]
        t = ms / cnt * 1e-3
        roof.update(achieved=form_bytes / t / 1e9, frac=form_bytes / t / 1e9 / HBM_PEAK_GBS, avg_launch_ms=ms / cnt, launches=cnt)
        roof["effective_csr"] = {"bytes_per_launch": csr_bytes, "GB/s": csr_bytes / t / 1e9, "frac": csr_bytes / t / 1e9 / HBM_PEAK_GBS}
        # K2 in the one-iteration regime also prepares the next step (pcg_update_spec_kernel); the library says which
        # streams its last launch moved (fv_update_form: 64 B per row, -8 with a uniform storage term, +8 with a dense b')
        fused_bytes = p.update_form()
        fused = fused_bytes > 0 and iters_per_step == 1.0
        for k, bytes_ in (("update", (fused_bytes if fused else 56) * p.n), ("pupdate", 32 * p.n)):
            kms, kc = prof[k]
            if kc:
                kern[k] = {"avg_ms": kms / kc, "launches": kc}
                if kms / kc > 0.05 * (32 * p.n / 5e12 * 1e3):  # the last p-update of a solve is skipped on convergence: no bandwidth figure for no-ops
                    kern[k]["GB/s"] = bytes_ / (kms / kc * 1e-3) / 1e9
                    kern[k]["bytes_per_row"] = bytes_ // p.n
    tfile = os.path.join(ROOT, "profiles", "spmv_traffic.json")
    if os.path.exists(tfile):
        try:
            t = json.load(open(tfile)).get(str(ns))
            # measured for the kernel and the storage form that ran here, not for an earlier one
            if t and t.get("form") == form_id and abs(t.get("form_bytes", 0) - form_bytes) <= 0.01 * form_bytes:
                roof["traffic"] = t["bytes"]
                roof["traffic_source"] = t.get("source")
                if roof.get("avg_launch_ms"):
                    roof["frac_traffic"] = t["bytes"] / (roof["avg_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
        except Exception:
            pass
    return roof, kern


def lean_block(fv, ctx, args, mins, maxs, ns, dn, src, lean=True):
    """The same workload on a problem created the other way: with FV_OPT_LEAN_SETUP (csrc/fv_lean.hip: no face arrays, incident lists or CSR in
    HBM, every storage form filled from rows formed on the fly) or through faces and CSR — the same doubles (tests/test_gpu_lean.py), the same
    kernels, a third of the memory."""
    ctx.synchronize()
    free0 = ctx.mem_info()[0]
    t0 = time.perf_counter()
    p = fv.Problem.regulargrid(mins, maxs, ns, dn, ctx, lean=lean)
    p.assemble(np.array([1e-5]), src, np.full(len(dn), 1e3))
    state = p.transient_begin(0.1, None, np.full(p.N, 1e3))
    t_setup = time.perf_counter() - t0
    if args.warmup > 0:
        p.run_fixed(state, args.dt, args.warmup, args.rtol, args.maxiter)
    regions = []
    for rep in range(max(args.repeats, 1)):  # (as for the headline: the median of the timed regions)
        ctx.synchronize()
        t0 = time.perf_counter()
        iters, info, _ = p.run_fixed(state, args.dt, args.steps, args.rtol, args.maxiter)
        ctx.synchronize()
        regions.append(time.perf_counter() - t0)
    sec = float(np.median(regions))
    in_use = (free0 - ctx.mem_info()[0]) / 1e9
    out = {"workload": "the headline's %d^3 box, dt, tolerance and steps on %s" % (ns[0], "a lean problem (FV_OPT_LEAN_SETUP = 1)" if lean else "a problem that holds faces, incident lists and CSR (FV_OPT_LEAN_SETUP = 0)"),
           "ms_per_step": sec / args.steps * 1e3,
           "value": p.N * args.steps / sec, "pcg_iters_per_step": float(np.mean(iters)), "converged": bool(info.converged), "hbm_in_use_gb": in_use,
           "bytes_per_cell_in_hbm": in_use * 1e9 / p.N, "setup_s": t_setup, "fused_launches": p.fused_form()[0], "traversal": "chunks" if p.fused_traversal() == 1 else "tiles",
           "larger_boxes": "profiles/r05_bench832_lean.json (5.8e8 cells), r05_bench928_lean.json (8.0e8 cells, 5.6e9 non-zeros): the same command with --ns"}
    p.close()
    return out


def multi_iteration_block(p, args):
    """A second measured block on the same operator: dt = 1 h, where a step needs ~10 PCG iterations (the headline's dt =
    60 s converges in one).  Per iteration K1 + K2 + K3 = B_spmv + 88 n bytes (SURVEY 8d's fused floor)."""
    dt, steps = 3600.0, 6
    st = p.new_state()
    st.set_nodes(np.full(p.N, 1e3))
    p.run_fixed(st, dt, 2, args.rtol, args.maxiter)
    p.profile(True)
    p.ctx.synchronize()
    m0 = p.bytes_moved()
    t0 = time.perf_counter()
    iters, info, _ = p.run_fixed(st, dt, steps, args.rtol, args.maxiter)
    p.ctx.synchronize()
    sec = time.perf_counter() - t0
    moved = p.bytes_moved() - m0
    prof = p.profile_get()
    p.profile(False)
    del st
    form_id, form_name, form_bytes = p.spmv_form()
    nit = int(np.sum(iters))
    loop = p.loop_form()  # 89 / 67: ONE launch per iteration (round 5); 105: fused pass (73) + slim vector update z' = z + alpha w (32);
    upd = 0 if loop in (89, 67) else (25 if loop in (76, 98) else 32)  # 83 with the matrix as codes; 7 fewer again with M^-1 as a code byte in the vector update
    per_it_bytes = loop * p.n if loop else form_bytes + 88 * p.n
    ms_it = sec / max(nit, 1) * 1e3
    regime = step_regime_roofline(p, iters, sec / steps, 0, 0, 0, moved)  # every launch of a step: set-up, first pass, the loop
    out = {"workload": "same %d^3 operator, dt=%gs, %d steps" % (args.ns, dt, steps), "pcg_iters_per_step": float(np.mean(iters)), "roofline": regime,
           "converged": bool(info.converged), "ms_per_step": sec / steps * 1e3, "dof_updates_per_s": p.N * steps / sec,
           "ms_per_iteration": ms_it, "bytes_per_iteration": per_it_bytes,
           "bytes_model": (("one launch per iteration: z' = z + alpha w, x += alpha p, p' = z' + beta p, w' = -M^-1 (A + D/dt) p' with the verdict, alpha and beta taken in the launch's "
                            "prologue from the previous launch's sums (%d n: z, w, p, x in, z', p', w', x out, the 3 upper diagonals%s, a code byte); no vector-update launch" % (loop, " as 16-bit codes" if loop == 67 else "")) if upd == 0 else
                           "fused pass x += alpha p, p' = z + beta p, w = -M^-1 (A + D/dt) p' (%d n: z, p, x in, p', w, x out, the 3 upper diagonals%s, storage codes) + vector update z' = z + alpha w (%d n%s)" % (loop - upd, " as 16-bit codes" if loop - upd < 60 else "", upd, ": M^-1 as a code byte" if upd == 25 else "") if loop
                           else "K1 storage form (%s) + 88 n for K2 + K3" % form_name),
           "GB/s": per_it_bytes / (ms_it * 1e-3) / 1e9, "frac_of_peak": per_it_bytes / (ms_it * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "frac_note": "frac_of_peak charges one loop iteration's bytes against wall time / iterations, i.e. it leaves the step's set-up and first pass unaccounted; `roofline` counts every launch of a step",
           "effective_csr_bytes_per_iteration": 12 * p.nnz + 20 * p.n + 88 * p.n, "kernels": {}}
    pass_bytes = (loop - upd) * p.n
    if upd == 0 and loop:  # the one-launch loop's live launches: per step one first pass (fv_step_form's bytes[1]) and iterations - 1 whole ones
        b_first = p.step_form()[0][1]
        pass_bytes = float(np.mean((b_first + np.maximum(np.asarray(iters, dtype=np.float64) - 1, 0) * loop) / np.maximum(iters, 1))) * p.n
    for k, bytes_ in ((("spmv_dot", pass_bytes), ("update", upd * p.n)) if loop else (("spmv_dot", form_bytes), ("update", 56 * p.n), ("pupdate", 32 * p.n))):
        kms, kc = prof[k]
        if kc and bytes_:
            out["kernels"]["fused_pass" if (loop and k == "spmv_dot") else k] = {"avg_ms": kms / kc, "launches": kc, "GB/s": bytes_ / (kms / kc * 1e-3) / 1e9}
    return out


if __name__ == "__main__":
    main()
