"""Multi-GPU leg of bench.py: one process per GPU (torch.distributed.run), the
10^8-cell box split into contiguous row blocks (x-slabs of the structured grid),
halo exchange + PCG all-reduces over RCCL/xGMI inside libfvhip.

torch.distributed (gloo) is only the control plane here: rendezvous, broadcast of
the RCCL unique id, barriers and the max-over-ranks of the timed region."""
import json
import os
import sys
import time

import numpy as np


class Watchdog:
    """A rank stuck in a collective (a peer died, a link is down) never comes back from the library call: after `seconds`
    without a sign of life this thread says where the rank was and ends the PROCESS with a non-zero code, so that the launcher
    (bench.py's own, or torch.distributed.run) takes the other ranks down.  Nothing is re-executed: the process just exits."""

    def __init__(self, rank, seconds):
        import threading

        self.rank, self.seconds, self.stage = rank, seconds, "start"
        self._beat = time.monotonic()
        self._stop = threading.Event()
        self._t = threading.Thread(target=self._run, daemon=True)
        self._t.start()

    def beat(self, stage):
        self.stage, self._beat = stage, time.monotonic()

    def _run(self):
        while not self._stop.wait(1.0):
            if time.monotonic() - self._beat > self.seconds:
                print("bench_dist: rank %d made no progress for %.0f s in stage '%s' (a collective that never completes?); exiting with code 4" %
                      (self.rank, self.seconds, self.stage), file=sys.stderr, flush=True)
                os._exit(4)

    def stop(self):
        self._stop.set()


def run_distributed(fv, args, world, rank):
    import torch
    import torch.distributed as dist

    import bench
    from fvamd import dist as fvdist

    dog = Watchdog(rank, float(os.environ.get("FV_BENCH_COLLECTIVE_TIMEOUT", "300")))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if "FV_BENCH_DEVICE" in os.environ:  # rehearsal on a one-GPU box: several ranks on the same device
        local_rank = int(os.environ["FV_BENCH_DEVICE"])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")  # only reached without the launcher (one-rank rehearsal)
    # gloo and RCCL print banners on stdout while they initialise; the driver wants exactly one JSON line there
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        ctx = fv.Context(local_rank)
        name, cus, mem = ctx.device_info()
        fvdist.comm_init_from_torch(ctx)
        if not fvdist.comm_selftest(ctx):  # ring send/recv + all-reduce over the new communicator, before anything depends on it
            raise RuntimeError("RCCL self-test failed on rank %d: data did not arrive intact" % rank)
    finally:
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)

    dog.beat("communicator up, building the problem")
    ns = [args.ns] * 3
    mins, maxs = bench.spacing_box(ns)
    dn, src = bench.box_setup(ns)
    t_setup = time.perf_counter()
    if os.environ.get("FV_BENCH_GLOBAL_ASSEMBLY") == "1" or world > ns[0]:
        # every rank assembles the (deterministic) global operator on its own GPU, keeps an equal share of the rows
        p = fv.Problem.regulargrid(mins, maxs, ns, dn, ctx)
        bounds, assembly = None, "global operator on every rank"
    else:
        # every rank generates and assembles only the faces of its own x-planes; node arrays are global
        p, bounds = fvdist.slab_problem(mins, maxs, ns, dn, world, rank, ctx)
        assembly = "per-rank slabs"
    p.assemble(np.array([1e-5]), src, np.full(len(dn), 1e3))
    p.transient_begin(0.1, None, np.full(p.N, 1e3))
    N, n_total = p.N, p.n
    blk = fvdist.RowBlock(p, world, rank, bounds)
    p.close()
    t_setup = time.perf_counter() - t_setup

    prob = fv.Problem(blk.handle, ctx)  # a view of the block for the profiling entry points; the block owns the handle
    dog.beat("warm-up steps")
    if args.warmup > 0:
        blk.run_fixed(args.dt, args.warmup, args.rtol, args.maxiter)
    dog.beat("timed region")
    if not args.no_profile:
        prob.profile(2)  # event pairs around the block SpMV only, as in the single-GPU loop (every event is a barrier between two launches)
    ctx.synchronize()
    dist.barrier()
    fused_before = blk.fused_form()[0]
    t0 = time.perf_counter()
    iters, info, dev_ms = blk.run_fixed(args.dt, args.steps, args.rtol, args.maxiter)
    ctx.synchronize()
    dist.barrier()
    sec = time.perf_counter() - t0
    sec_own = sec  # this rank's own clock over the timed region (the headline takes the max over ranks)
    fused_launches, fused_row_bytes, fused_bytes = blk.fused_form()
    fused_launches -= fused_before
    prof = prob.profile_get() if not args.no_profile else None
    prob.profile(False)
    if prof is not None:  # the vector pass: 16 more steps with all event pairs on
        prob.profile(1)
        blk.run_fixed(args.dt, 16, args.rtol, args.maxiter)
        prof["update"] = prob.profile_get()["update"]
        prob.profile(False)
    dog.beat("diagnosis pass")
    # where a step's time goes on this rank (events around the collectives, the halo wait and the two SpMV passes): 16 more
    # steps AFTER the timed region, since every event is a barrier between two launches
    diag = None
    if not args.no_profile:
        fvdist.comm_diag(ctx, True)
        blk.run_fixed(args.dt, 16, args.rtol, args.maxiter)
        ctx.synchronize()
        raw = fvdist.comm_diag_get(ctx)
        fvdist.comm_diag(ctx, False)
        diag = {k + "_ms_per_step": v[0] / 16 for k, v in raw.items()}
        diag.update({k + "_per_step": v[1] / 16 for k, v in raw.items()})
    dist.barrier()
    dog.beat("gathering")
    tmax = torch.tensor([sec], dtype=torch.float64)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    sec = float(tmax[0])

    # roofline of the dominant kernel on this rank's block: the PCG SpMV of the block (pack + interior pass + wait for the
    # halo + boundary pass), HIP events around every live launch set inside the timed region
    form_id, prob_form, form_bytes = prob.spmv_form()
    csr_bytes = 12 * blk.nnz + 20 * blk.nloc  # SURVEY 8d's CSR accounting
    ms_inloop = prof["spmv_dot"][0] / prof["spmv_dot"][1] if prof and prof["spmv_dot"][1] else None
    ms_k2 = prof["update"][0] / prof["update"][1] if prof and prof["update"][1] else None
    try:
        ms_b2b = prob.bench_spmv(1.0 / args.dt, 10)  # 10 back-to-back launches on resident vectors (cache-warm): a ceiling, not the loop's time
    finally:
        prob.handle = None
    ms = ms_inloop if ms_inloop else ms_b2b
    # the one-iteration regime on blocks of whole planes runs the fused step (fv_fused.hip): the event pairs then bracket the
    # whole step on the block — send-row pack, fused launch || halo exchange, boundary products, their conversion
    fused = bool(ms_inloop) and fused_launches >= 0.8 * args.steps
    step_bytes = fused_bytes if fused else form_bytes
    ach = step_bytes / (ms * 1e-3) / 1e9
    gathered = [None] * world
    dist.all_gather_object(gathered, dict(rank=rank, rows=blk.nloc, nnz=blk.nnz, halo=blk.nhalo, send=blk.nsend, spmv_ms_in_loop=ms_inloop,
                                          spmv_ms_back_to_back=ms_b2b, spmv_gbs=ach, update_ms_in_loop=ms_k2, device_ms=dev_ms,
                                          fused_launches=fused_launches, fused_bytes_per_row=fused_row_bytes if fused else None,
                                          wall_s_own=sec_own, diagnosis=diag))
    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        cpu = bench.cpu_baseline(args.dt, args.rtol)  # the other ranks wait at the barrier below
    if rank == 0:
        out = {
            "metric": "DoF-updates/s (cells\u00d7steps) implicit transient; SpMV HBM GB/s vs peak",
            "value": N * args.steps / sec,
            "unit": "DoF-updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": sec / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "synthetic %d^3 box (%.3g cells), transient, fixed dt=%gs, Jacobi-PCG rtol %.0e, contiguous row blocks (x-slabs) over %d GPUs, RCCL halo exchange + one merged 6-double all-reduce per one-iteration step (two per PCG iteration otherwise)" % (args.ns, N, args.dt, args.rtol, world),
                "step_form": "fused step on every row block (fv_fused.hip)" if fused else "K1 + K2S per step",
                "cells": N, "unknowns": n_total, "nnz": int(sum(g["nnz"] for g in gathered)), "assembly": assembly,
                "pcg_iters_per_step": float(np.mean(iters)), "last_relres": info.relres, "converged": bool(info.converged),
                "device": name, "setup_s": t_setup, "per_rank": gathered,
                "per_rank_diagnosis": "ms per step in 16 extra steps with HIP events around every all-reduce (allreduce), the halo exchange on the second stream (halo_exchange), the compute stream's wait for it (halo_wait) and the interior / boundary SpMV passes; a rank that waits for a slower one shows it in allreduce and halo_wait",
            },
            "roofline": {"bound": "hbm", "achieved": ach, "peak": bench.HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / bench.HBM_PEAK_GBS,
                         "traffic": None,
                         "kernel": ("fused step of rank 0's row block: z' of the send rows -> halo exchange || fused_step_kernel (vector update of step k + product of step k + 1 on the interior window) -> boundary products + v-form conversion, per GPU"
                                    if fused else "PCG SpMV of rank 0's row block: pack + interior pass || halo exchange + boundary pass (%s), per GPU" % prob_form),
                         "algorithmic_bytes_per_launch": step_bytes, "avg_launch_ms": ms,
                         "timing": "HIP events around every live launch set inside the timed region (includes the wait for the halo)" if ms_inloop else "back-to-back launches after the timed region",
                         "effective_csr": {"bytes_per_launch": csr_bytes, "GB/s": csr_bytes / (ms * 1e-3) / 1e9, "frac": csr_bytes / (ms * 1e-3) / 1e9 / bench.HBM_PEAK_GBS,
                                           "note": "SURVEY 8d CSR accounting 12 nnz + 20 n; the storage form moves fewer bytes, so this can exceed the hardware rate"}},
        }
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out))
    dog.beat("closing")
    dist.barrier()
    blk.close()
    dist.destroy_process_group()
    dog.stop()
