"""One-process-per-GPU runs: RCCL communicator bootstrap and row-block problems.

torch.distributed is used only as the control plane (rendezvous, broadcasting the
RCCL unique id, barriers); the data plane — halo exchange and the PCG all-reduces —
is RCCL called directly from libfvhip.so on HIP streams.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import SolveInfo, load, ptr
from .core import ai64
from .core import Problem


def comm_unique_id():
    buf = C.create_string_buffer(_lib.FV_COMM_ID_BYTES)
    _lib.check(load().fv_comm_unique_id(buf))
    return buf.raw


def comm_init(ctx, nranks, rank, uid):
    ctx.check(load().fv_comm_init(ctx.handle, int(nranks), int(rank), C.create_string_buffer(uid, _lib.FV_COMM_ID_BYTES)))


def comm_selftest(ctx, count=1 << 16):
    """fv_comm_selftest: ring send/recv + all-reduce over the RCCL communicator; True when the data arrived intact."""
    ok = C.c_int(0)
    ctx.check(load().fv_comm_selftest(ctx.handle, int(count), C.byref(ok)))
    return bool(ok.value)


def comm_stats(ctx, reset=False):
    """(all-reduces, halo exchanges) issued through the context so far (fv_comm_stats)."""
    a, h = C.c_int64(), C.c_int64()
    ctx.check(load().fv_comm_stats(ctx.handle, C.byref(a), C.byref(h), int(bool(reset))))
    return a.value, h.value


def comm_diag(ctx, enable=True):
    """fv_comm_diag: bracket the collectives, the halo wait and the two SpMV passes of a row block with HIP events."""
    ctx.check(load().fv_comm_diag(ctx.handle, int(bool(enable))))


def comm_diag_get(ctx):
    """-> {name: (total ms, pairs)} for allreduce, halo_exchange, halo_wait, interior_spmv, boundary_spmv (fv_comm_diag_get)."""
    ms = (C.c_double * 5)()
    cnt = (C.c_int64 * 5)()
    ctx.check(load().fv_comm_diag_get(ctx.handle, ms, cnt))
    names = ("allreduce", "halo_exchange", "halo_wait", "interior_spmv", "boundary_spmv")
    return {k: (ms[i], cnt[i]) for i, k in enumerate(names)}


def comm_init_local(ctx, nranks, rank, group_id=0):
    """Loopback transport: `nranks` threads of this process, one context each on the same device (rehearsals/tests)."""
    ctx.check(load().fv_comm_init_local(ctx.handle, int(nranks), int(rank), int(group_id)))


def comm_init_from_torch(ctx):
    """Rank 0 creates the RCCL id; torch.distributed (any backend) broadcasts it."""
    import torch.distributed as dist

    rank, world = dist.get_rank(), dist.get_world_size()
    box = [comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    comm_init(ctx, world, rank, box[0])
    return rank, world


def slab_planes(n1, nranks):
    """Plane ranges of an x-slab decomposition: rank r gets the planes [planes[r], planes[r+1])."""
    return [(r * int(n1)) // int(nranks) for r in range(int(nranks) + 1)]


def slab_face_range(ns, i1_lo, i1_hi):
    """The slab's faces as a range [f0, f1) of the global face list (regulargrid's order): what to cut out of per-face
    conductivities before handing them to the slab problem's assemble."""
    n1, n2, n3 = (int(v) for v in ns)

    def offset(i1):
        return min(i1, n1 - 1) * n2 * n3 + i1 * (n2 - 1) * n3 + i1 * n2 * (n3 - 1)

    return offset(max(int(i1_lo) - 1, 0)), offset(int(i1_hi))


def slab_problem(mins, maxs, ns, dirichletnodes, nranks, rank, ctx=None):
    """The rank's slab problem and the row bounds of all ranks (identical on every rank): assemble it / begin the
    transient with the GLOBAL sources, heads and u0, then RowBlock(problem, nranks, rank, bounds)."""
    planes = slab_planes(ns[0], nranks)
    if planes[rank] == planes[rank + 1]:
        raise ValueError("more ranks than planes")
    p = Problem.regulargrid_slab(mins, maxs, ns, dirichletnodes, planes[rank], planes[rank + 1], ctx)
    plane = int(ns[1]) * int(ns[2])
    bounds = [p.free_rows_before(q * plane) for q in planes]
    return p, bounds


def rank_faces(neighbors, N, dirichletnodes, nranks, rank, bounds=None):
    """Block-local assembly for face-list (unstructured) meshes: the faces a rank needs, i.e. those with an end among the
    free rows [bounds[rank], bounds[rank+1]) — in the order of the global list, so that the rank's rows come out bit for
    bit as in the global operator.  -> (face indices (0-based, ascending), bounds).  bounds=None: equal shares of rows,
    as fv_dist_setup."""
    import numpy as _np

    from .core import _split_neighbors

    n1, n2 = _split_neighbors(neighbors)
    free = _np.ones(int(N), bool)
    free[ai64(dirichletnodes) - 1] = False
    n = int(free.sum())
    if bounds is None:
        bounds = [(r * n) // int(nranks) for r in range(int(nranks) + 1)]
    row = _np.cumsum(free) - 1  # free row of every free node
    lo, hi = bounds[rank], bounds[rank + 1]
    mine = free & (row >= lo) & (row < hi)
    sel = _np.nonzero(mine[n1 - 1] | mine[n2 - 1])[0]
    return sel, [int(b) for b in bounds]


def partial_problem(neighbors, areasoverlengths, N, dirichletnodes, nranks, rank, bounds=None, ctx=None):
    """The rank's problem of a face-list mesh without the global operator: all nodes, only the faces of rank_faces.
    -> (problem, bounds, faces): assemble with conductivities[faces] (or a metaindex[faces]), the global sources, heads
    and u0, then RowBlock(problem, nranks, rank, bounds)."""
    import numpy as _np

    from .core import _split_neighbors

    sel, bounds = rank_faces(neighbors, N, dirichletnodes, nranks, rank, bounds)
    n1, n2 = _split_neighbors(neighbors)
    from ._lib import default_context

    ctx = ctx or default_context()
    before = ctx.get_option(_lib.FV_OPT_REORDER)
    ctx.set_option(_lib.FV_OPT_REORDER, 0)  # a partial operator: a re-numbering computed from a rank's own faces would only cost time
    try:
        p = Problem.create(_np.stack([n1[sel], n2[sel]], axis=1), _np.asarray(areasoverlengths, dtype=_np.float64)[sel], N, dirichletnodes, ctx)
    finally:
        ctx.set_option(_lib.FV_OPT_REORDER, before)
    return p, bounds, sel


class RowBlock:
    """A rank's contiguous range of free rows (fv_dist_setup)."""

    def __init__(self, global_problem, nranks, rank, bounds=None):
        h = _lib.c_prob()
        if bounds is None:
            global_problem.check(load().fv_dist_setup(global_problem.handle, int(nranks), int(rank), C.byref(h)))
        else:
            b = ai64(bounds)
            if len(b) != int(nranks) + 1:
                raise ValueError("bounds needs nranks + 1 entries")
            global_problem.check(load().fv_dist_setup_bounds(global_problem.handle, int(nranks), int(rank), ptr(b), C.byref(h)))
        self.handle, self.ctx = h, global_problem.ctx
        self.nranks, self.rank = int(nranks), int(rank)
        v = [C.c_int64() for _ in range(7)]
        self.ctx.check(load().fv_dist_plan_sizes(h, *[C.byref(x) for x in v]))
        self.lo, self.hi, self.nnz, self.nhalo, self.nsend, self.n_int, self.n_bnd = [x.value for x in v]
        self.nloc = self.hi - self.lo

    def plan(self):
        rp = np.empty(self.nloc + 1, np.int64)
        ci = np.empty(self.nnz, np.int64)
        hc = np.empty(self.nhalo, np.int64)
        rc = np.empty(self.nranks, np.int64)
        sc = np.empty(self.nranks, np.int64)
        si = np.empty(self.nsend, np.int64)
        gb = np.empty(self.n_bnd, np.int64)
        self.ctx.check(load().fv_dist_get_plan(self.handle, ptr(rp), ptr(ci), ptr(hc), ptr(rc), ptr(sc), ptr(si), ptr(gb)))
        return dict(rowptr=rp, colind=ci, halo_cols=hc, recv_counts=rc, send_counts=sc, send_idx=si, groups_bnd=gb)

    def run_fixed(self, dt, nsteps, rtol, maxiter=1000):
        iters = np.zeros(max(int(nsteps), 1), np.int32)
        info = SolveInfo()
        ms = C.c_double()
        self.ctx.check(load().fv_dist_run_fixed(self.handle, float(dt), int(nsteps), float(rtol), int(maxiter), ptr(iters), C.byref(info), C.byref(ms)))
        return iters[: int(nsteps)], info, ms.value

    def fused_form(self):
        """(launches so far, bytes per row, bytes per launch) of the fused step on this block (fv_fused_form)."""
        n, b, t = C.c_int64(), C.c_int32(), C.c_int64()
        self.ctx.check(load().fv_fused_form(self.handle, C.byref(n), C.byref(b), C.byref(t)))
        return n.value, b.value, t.value

    def set_preconditioner(self, kind):
        """"jacobi" (default); "amg": block-Jacobi with the aggregation-AMG cycle of the rank's diagonal block as the
        block solver (no communication inside the preconditioner); "amg_gathered": the coarse levels of the whole operator,
        gathered on every rank (two halo exchanges and one all-reduce per cycle; an iteration count that does not depend on
        the rank count).  Every rank must make the same choice."""
        self.ctx.check(load().fv_precond_set(self.handle, Problem.PRECONDITIONERS[kind]))
        return self

    def solve_steady(self, x0_local=None, rtol=1e-8, maxiter=1000):
        """fv_dist_solve_steady (collective): Jacobi-PCG on A x = b over all ranks -> (the rank's rows of x, info)."""
        x0 = _lib.af64(x0_local) if x0_local is not None else None
        x = np.empty(self.nloc, np.float64)
        info = SolveInfo()
        self.ctx.check(load().fv_dist_solve_steady(self.handle, ptr(x0), float(rtol), int(maxiter), ptr(x), C.byref(info)))
        return x, info

    def step(self, dt, bhat_local=None, rtol=1e-8, maxiter=1000):
        """fv_dist_step (collective): one implicit step of the block's state, optionally with the rank's rows of getb(t)."""
        bh = _lib.af64(bhat_local) if bhat_local is not None else None
        info = SolveInfo()
        self.ctx.check(load().fv_dist_step(self.handle, float(dt), ptr(bh), float(rtol), int(maxiter), C.byref(info)))
        return info

    def run_adaptive(self, t0, tfinal, dt0=1.0, atol=1e-4, rtol=1e-8, maxiter=1000, max_outer=1 << 20):
        """fv_dist_run_adaptive (collective): the default step-doubling stepper on row blocks -> (ts, nsolves, info)."""
        ts = np.empty(int(max_outer) + 1, np.float64)
        nout, nsol = C.c_int64(), C.c_int64()
        info = SolveInfo()
        self.ctx.check(load().fv_dist_run_adaptive(self.handle, float(t0), float(tfinal), float(dt0), float(atol), float(rtol), int(maxiter), int(max_outer), ptr(ts), C.byref(nout), C.byref(nsol), C.byref(info)))
        return ts[: nout.value + 1].copy(), nsol.value, info

    def spmv(self, x_local, sigma=0.0):
        x = _lib.af64(x_local)
        y = np.empty(self.nloc, np.float64)
        self.ctx.check(load().fv_dist_spmv(self.handle, ptr(x), float(sigma), ptr(y)))
        return y

    def spmv_halo(self, x_local, halo_values, sigma=0.0):
        """Interior + boundary SpMV passes with caller-supplied halo values (no communication)."""
        x, h = _lib.af64(x_local), _lib.af64(halo_values)
        y = np.empty(self.nloc, np.float64)
        self.ctx.check(load().fv_dist_spmv_halo(self.handle, ptr(x), ptr(h) if self.nhalo else None, float(sigma), ptr(y)))
        return y

    def state(self):
        u = np.empty(self.nloc, np.float64)
        self.ctx.check(load().fv_dist_state_get(self.handle, ptr(u)))
        return u

    def close(self):
        if getattr(self, "handle", None) and getattr(self.ctx, "handle", None):
            load().fv_problem_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
