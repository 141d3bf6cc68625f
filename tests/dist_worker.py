"""World-size-N CPU rehearsal of the multi-GPU protocol (gloo): every rank holds a
row block [local | halo] from finitevolume.jl_amd/partition.py, exchanges halos
with point-to-point messages and reduces the PCG scalars with all-reduce — the
same message pattern libfvhip issues through RCCL.  The arithmetic here is numpy
(checker-side); the product's kernels are exercised by the GPU tests."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def build_case():
    """Lateral-Dirichlet box with a well (the bench workload in miniature), via the oracle."""
    from oracle import fv_oracle as o
    import bench

    ns = [12, 9, 7]
    mins, maxs = bench.spacing_box(ns)
    _, n1, n2, aol, vol = o.regulargrid(mins, maxs, ns, want_coords=False)
    dn, src = bench.box_setup(ns)
    dh = np.full(len(dn), 1e3)
    rng = np.random.default_rng(0)
    K = 1e-5 * np.exp(rng.standard_normal(len(aol)))
    A = o.assembleA(n1, n2, aol, K, src, dn, dh)
    b = o.assembleb(n1, n2, aol, K, src, dn, dh)
    freenode, _ = o.getfreenodes(len(vol), dn)
    D = 0.1 * vol[freenode]
    u0 = np.full(A.n, 1e3) + rng.standard_normal(A.n)
    return A, b, D, u0


class LocalOp:
    def __init__(self, A, D, plan, dist, torch):
        self.pl, self.dist, self.torch = plan, dist, torch
        e0 = plan["entry_lo"]
        self.vals = A.nzval[e0 : e0 + len(plan["colind"])]
        self.rowid = np.repeat(np.arange(plan["nloc"]), np.diff(plan["rowptr"]))
        self.D = D[plan["lo"] : plan["hi"]]
        self.rank = dist.get_rank()
        self.nranks = dist.get_world_size()

    def exchange(self, xloc):
        """Returns [local | halo] after the point-to-point halo exchange."""
        pl, dist, torch = self.pl, self.dist, self.torch
        ext = np.empty(pl["nloc"] + len(pl["halo_cols"]))
        ext[: pl["nloc"]] = xloc
        reqs, recvbufs = [], {}
        off = pl["nloc"]
        for q in range(self.nranks):
            cnt = int(pl["recv_counts"][q])
            if q != self.rank and cnt:
                t = torch.empty(cnt, dtype=torch.float64)
                recvbufs[q] = (t, off)
                reqs.append(dist.irecv(t, src=q))
            off += cnt
        for q in range(self.nranks):
            idx = pl["send_idx"][q]
            if q != self.rank and len(idx):
                reqs.append(dist.isend(torch.from_numpy(np.ascontiguousarray(xloc[idx])), dst=q))
        for r in reqs:
            r.wait()
        for q, (t, o) in recvbufs.items():
            ext[o : o + len(t)] = t.numpy()
        return ext

    def matvec(self, xloc, sigma):
        ext = self.exchange(xloc)
        y = np.zeros(self.pl["nloc"])
        np.add.at(y, self.rowid, self.vals * ext[self.pl["colind"]])
        return y + sigma * self.D * xloc

    def allsum(self, *vals):
        t = self.torch.tensor(vals, dtype=self.torch.float64)
        self.dist.all_reduce(t)
        return t.tolist()


def dist_pcg(op, diagA, rhs, x, sigma, rtol, maxiter):
    """The three-kernel Jacobi-PCG of fv_pcg.hip, scalars via all-reduce."""
    minv = 1.0 / (diagA + sigma * op.D)
    r = rhs - op.matvec(x, sigma)
    p = minv * r
    rz, rr, bb = op.allsum(float(r @ (minv * r)), float(r @ r), float(rhs @ rhs))
    tol2 = rtol * rtol * bb
    it = 0
    while it < maxiter and rr > tol2:
        q = op.matvec(p, sigma)
        (pq,) = op.allsum(float(p @ q))
        alpha = rz / pq
        x = x + alpha * p
        r = r - alpha * q
        rzn, rr = op.allsum(float(r @ (minv * r)), float(r @ r))
        p = minv * r + (rzn / rz) * p
        rz = rzn
        it += 1
    return x, it


def worker(rank, world, port, outdir):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from __graft_entry__ import load_package

        load_package()
        from fvamd import partition

        A, b, D, u0 = build_case()
        rowptr, colind = A.colptr - 1, A.rowval - 1  # symmetric: CSC arrays are the CSR arrays
        pl = partition.plan(rowptr, colind, world, rank)
        # plans must agree pairwise: what I expect from q is what q sends me
        counts = [None] * world
        dist.all_gather_object(counts, (pl["recv_counts"].tolist(), [len(s) for s in pl["send_idx"]]))
        for q in range(world):
            assert counts[q][1][rank] == pl["recv_counts"][q], "send/recv plan mismatch"
        sends = [None] * world
        dist.all_gather_object(sends, [(s + pl["lo"]).tolist() for s in pl["send_idx"]])
        off = 0
        for q in range(world):
            cnt = int(pl["recv_counts"][q])
            assert pl["halo_cols"][off : off + cnt].tolist() == sends[q][rank]
            off += cnt
        op = LocalOp(A, D, pl, dist, torch)
        lo, hi = pl["lo"], pl["hi"]
        diag = A.toscipy().diagonal()[lo:hi]
        # SpMV check
        xg = np.cos(np.arange(A.n) * 0.37)
        y = op.matvec(xg[lo:hi], 0.25)
        ref = A.matvec(xg) + 0.25 * D * xg
        assert np.allclose(y, ref[lo:hi], rtol=1e-13, atol=1e-18)
        # three implicit steps: (D/dt + A) u+ = b + D u/dt
        dt = 3600.0
        u = u0[lo:hi].copy()
        iters = []
        for _ in range(3):
            rhs = b[lo:hi] + op.D * (u / dt)
            u, it = dist_pcg(op, diag, rhs, u, 1.0 / dt, 1e-13, 500)
            iters.append(it)
        parts = [None] * world
        dist.all_gather_object(parts, u.tolist())
        if rank == 0:
            np.save(os.path.join(outdir, "u_dist.npy"), np.concatenate([np.array(p) for p in parts]))
            np.save(os.path.join(outdir, "iters.npy"), np.array(iters))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def serial_reference():
    from oracle import fv_oracle as o

    A, b, D, u0 = build_case()
    dt = 3600.0
    u = u0.copy()
    for _ in range(3):
        rhs = b + D * (u / dt)
        u, ch = o.pcg_jacobi(A, rhs, x0=u, shift=D / dt, tol=1e-13, maxiter=500)
    return u
