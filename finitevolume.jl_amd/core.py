"""Host-side mirror of FiniteVolume.jl's functions for the accelerated path.

Same names, positional orders, return tuples and error behaviour as the Julia
package (/root/reference/src/{FiniteVolume,grid,transient}.jl); all arithmetic is
done by libfvhip.so on the MI355X.  Julia conventions are kept at this surface:
indices are 1-based int64, `neighbors` is an (F, 2) int64 array standing for
Vector{Pair{Int,Int}} (a list of pairs or a (node1, node2) tuple is accepted),
`coords` is 3 x N.

Closures cannot cross the C ABI, so `metaindex` (a callable i -> index, or an
index array) is pre-evaluated over 1:F here, `getb(t)` is evaluated on the host
each step and uploaded, and `callback(t, dt)` is invoked from this side's loop.
"""
import ctypes as C
import math

import numpy as np

from . import _lib
from ._lib import FVError, SolveInfo, af64, ai64, default_context, load, ptr

SQRT_EPS = math.sqrt(np.finfo(np.float64).eps)  # IterativeSolvers' default tol


# ------------------------------------------------------------------ helpers
def _split_neighbors(neighbors):
    if isinstance(neighbors, tuple) and len(neighbors) == 2 and np.ndim(neighbors[0]) == 1:
        return ai64(neighbors[0]), ai64(neighbors[1])
    nb = np.asarray(neighbors, dtype=np.int64)
    if nb.size == 0:
        return np.empty(0, np.int64), np.empty(0, np.int64)
    if nb.ndim != 2 or nb.shape[1] != 2:
        raise TypeError("neighbors must be (F, 2) pairs")
    return np.ascontiguousarray(nb[:, 0]), np.ascontiguousarray(nb[:, 1])


def _metaindex_array(metaindex, F):
    if metaindex is None:
        return None
    if callable(metaindex):
        return ai64([metaindex(i) for i in range(1, F + 1)])
    return ai64(metaindex)


class ConvergenceHistory:
    """What callers read from IterativeSolvers' history: isconverged, data[:resnorm]."""

    def __init__(self, info, resnorm=None):
        self.isconverged = bool(info.converged)
        self.iters = int(info.iters)
        self.mvps = int(info.iters) + 1
        self.relres = float(info.relres)
        self.solve_ms = float(info.solve_ms)
        self.data = {"resnorm": resnorm if resnorm is not None else np.empty(0)}

    def __repr__(self):
        return "ConvergenceHistory(%s after %d iterations, relres=%.3e)" % (
            "converged" if self.isconverged else "not converged", self.iters, self.relres)


class SparseMatrixCSC:
    """Julia's SparseMatrixCSC{Float64,Int64}: 1-based colptr/rowval + nzval."""

    def __init__(self, m, n, colptr, rowval, nzval):
        self.m, self.n = int(m), int(n)
        self.colptr, self.rowval, self.nzval = colptr, rowval, nzval

    @property
    def shape(self):
        return (self.m, self.n)

    def copy(self):
        return SparseMatrixCSC(self.m, self.n, self.colptr.copy(), self.rowval.copy(), self.nzval.copy())

    def toscipy(self):
        import scipy.sparse as sp

        return sp.csc_matrix((self.nzval, self.rowval - 1, self.colptr - 1), shape=(self.m, self.n))

    @staticmethod
    def fromscipy(M):
        M = M.tocsc()
        M.sort_indices()
        return SparseMatrixCSC(M.shape[0], M.shape[1], M.indptr.astype(np.int64) + 1, M.indices.astype(np.int64) + 1, M.data.astype(np.float64))

    @staticmethod
    def fromdense(A):
        import scipy.sparse as sp

        return SparseMatrixCSC.fromscipy(sp.csc_matrix(np.asarray(A, dtype=np.float64)))


# ------------------------------------------------------------------ grid (src/grid.jl)
def regulargrid(mins, maxs, ns, ctx=None):
    """grid.jl:56-110 -> coords (3,N), neighbors (F,2), areasoverlengths, volumes."""
    assert len(mins) == len(maxs)
    assert len(mins) == len(ns)
    if len(mins) != 3:
        raise FVError(_lib.FV_ERR_ARG, "only 3 dimensions supported")
    ctx = ctx or default_context()
    lib = load()
    mins_, maxs_, ns_ = af64(mins), af64(maxs), ai64(ns)
    N, F = C.c_int64(), C.c_int64()
    ctx.check(lib.fv_regulargrid_sizes(ptr(ns_), C.byref(N), C.byref(F)))
    N, F = N.value, F.value
    coords = np.empty((N, 3), np.float64)
    n1 = np.empty(F, np.int64)
    n2 = np.empty(F, np.int64)
    aol = np.empty(F, np.float64)
    vol = np.empty(N, np.float64)
    ctx.check(lib.fv_regulargrid(ctx.handle, ptr(mins_), ptr(maxs_), ptr(ns_), ptr(coords), ptr(n1), ptr(n2), ptr(aol), ptr(vol)))
    return coords.T, np.stack([n1, n2], axis=1), aol, vol


def nodehycos2neighborhycos(neighbors, nodehycos, logtransformhyco=False, ctx=None):
    """grid.jl:14-33; nodehycos is the (n3, n2, n1) array (column-major order == node order)."""
    ctx = ctx or default_context()
    n1, n2 = _split_neighbors(neighbors)
    nh = af64(np.asarray(nodehycos).ravel(order="F"))
    out = np.empty(len(n1), np.float64)
    ctx.check(load().fv_nodehycos2neighborhycos(ctx.handle, len(n1), ptr(n1), ptr(n2), len(nh), ptr(nh), int(bool(logtransformhyco)), ptr(out)))
    return out


# ------------------------------------------------------------------ maps (FiniteVolume.jl:20-44)
def getfreenodes(n, dirichletnodes, ctx=None):
    ctx = ctx or default_context()
    d = ai64(dirichletnodes)
    freenode = np.empty(n, np.uint8)
    n2f = np.empty(n, np.int64)
    nfree = C.c_int64()
    ctx.check(load().fv_getfreenodes(ctx.handle, int(n), len(d), ptr(d), ptr(freenode), ptr(n2f), C.byref(nfree)))
    return freenode.astype(bool), n2f


def getnodei2dirichleti(sources, dirichletnodes, ctx=None):
    ctx = ctx or default_context()
    s, d = af64(sources), ai64(dirichletnodes)
    out = np.empty(len(s), np.int64)
    bad = C.c_int64()
    ctx.check(load().fv_getnodei2dirichleti(ctx.handle, len(s), ptr(s), len(d), ptr(d), ptr(out), C.byref(bad)))
    return out


# ------------------------------------------------------------------ device problem
def _checksum(x):
    """(size, sum, strided sum) of an input array: a cheap fingerprint by which the device-resident adjoint workflow recognises that the
    arguments it is handed are the ones its problem was built from (adjoint.adjointintegrate)."""
    if x is None:
        return None
    a = np.asarray(x)
    if a.dtype == object or a.size == 0:
        return (int(a.size), 0.0, 0.0)
    a = a.reshape(-1)
    return (int(a.size), float(a.sum(dtype=np.float64)), float(a[:: max(1, a.size // 997)].sum(dtype=np.float64)))


class Problem:
    """fv_problem: the mesh + Dirichlet set resident on the GPU (symbolic CSR built once)."""

    def __init__(self, handle, ctx):
        self.handle, self.ctx = handle, ctx
        N, F, n, nnz = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        ctx.check(load().fv_problem_sizes(handle, C.byref(N), C.byref(F), C.byref(n), C.byref(nnz)))
        self.N, self.F, self.n, self.nnz = N.value, F.value, n.value, nnz.value
        self._sources = None
        self._dheads = None
        self.lean = False  # FV_OPT_LEAN_SETUP took effect for this problem (Problem.regulargrid sets it)

    @classmethod
    def create(cls, neighbors, areasoverlengths, N, dirichletnodes, ctx=None):
        ctx = ctx or default_context()
        n1, n2 = _split_neighbors(neighbors)
        aol, d = af64(areasoverlengths), ai64(dirichletnodes)
        if len(aol) < len(n1):
            raise FVError(_lib.FV_ERR_INDEX, "BoundsError: %d areasoverlengths for %d neighbors" % (len(aol), len(n1)))
        h = _lib.c_prob()
        ctx.check(load().fv_problem_create(ctx.handle, int(N), len(n1), ptr(n1), ptr(n2), ptr(aol), len(d), ptr(d), C.byref(h)))
        p = cls(h, ctx)
        p.inputs = {"neighbors": (_checksum(n1), _checksum(n2)), "areasoverlengths": _checksum(aol), "dirichletnodes": _checksum(d)}
        return p

    @classmethod
    def regulargrid(cls, mins, maxs, ns, dirichletnodes, ctx=None, lean=None):
        """fv_problem_create_regulargrid.  lean: FV_OPT_LEAN_SETUP for this problem (True / False; None: the context's setting, by default
        lean only where the CSR's int32 offsets would not hold the operator) — no face arrays, incident lists or CSR in HBM."""
        ctx = ctx or default_context()
        mins_, maxs_, ns_, d = af64(mins), af64(maxs), ai64(ns), ai64(dirichletnodes)
        h = _lib.c_prob()
        before = ctx.get_option(_lib.FV_OPT_LEAN_SETUP)
        if lean is not None:
            ctx.set_option(_lib.FV_OPT_LEAN_SETUP, 1 if lean else 0)
        try:
            ctx.check(load().fv_problem_create_regulargrid(ctx.handle, ptr(mins_), ptr(maxs_), ptr(ns_), len(d), ptr(d), C.byref(h)))
        finally:
            ctx.set_option(_lib.FV_OPT_LEAN_SETUP, before)
        p = cls(h, ctx)
        N = int(ns_[0]) * int(ns_[1]) * int(ns_[2])
        want = before if lean is None else (1 if lean else 0)
        p.lean = bool(N >= 4096 and (want == 1 or (want == 2 and 7 * N > 2**31 - 3)))  # (fv_problem_create_regulargrid's rule)
        return p

    @classmethod
    def regulargrid_slab(cls, mins, maxs, ns, dirichletnodes, i1_lo, i1_hi, ctx=None):
        """One slab of the grid (planes [i1_lo, i1_hi) along the first axis): global node arrays, local faces only.
        For row-block runs whose ranks never hold the global operator (fv_problem_create_regulargrid_slab)."""
        ctx = ctx or default_context()
        mins_, maxs_, ns_, d = af64(mins), af64(maxs), ai64(ns), ai64(dirichletnodes)
        h = _lib.c_prob()
        ctx.check(load().fv_problem_create_regulargrid_slab(ctx.handle, ptr(mins_), ptr(maxs_), ptr(ns_), len(d), ptr(d), int(i1_lo), int(i1_hi), C.byref(h)))
        return cls(h, ctx)

    def free_rows_before(self, node0):
        """Number of free cells whose 0-based node index is below node0."""
        rows = C.c_int64()
        self.check(load().fv_problem_free_rows_before(self.handle, int(node0), C.byref(rows)))
        return rows.value

    @classmethod
    def from_csc(cls, A, ctx=None):
        ctx = ctx or default_context()
        colptr, rowval, nzval = ai64(A.colptr), ai64(A.rowval), af64(A.nzval)
        h = _lib.c_prob()
        ctx.check(load().fv_problem_create_from_csc(ctx.handle, A.n, ptr(colptr), ptr(rowval), ptr(nzval), C.byref(h)))
        return cls(h, ctx)

    def check(self, rc):
        self.ctx.check(rc)

    def free_maps(self):
        freenode = np.empty(self.N, np.uint8)
        n2f = np.empty(self.N, np.int64)
        self.check(load().fv_problem_get_free_maps(self.handle, ptr(freenode), ptr(n2f)))
        return freenode.astype(bool), n2f

    def assemble(self, conductivities, sources, dirichletheads, metaindex=None, logtransformconductivity=False):
        K, s, dh = af64(conductivities), af64(sources), af64(dirichletheads)
        if len(s) != self.N:
            raise FVError(_lib.FV_ERR_ARG, "sources must have one entry per node")
        mi = _metaindex_array(metaindex, self.F)
        bad = C.c_int64()
        self.check(load().fv_assemble(self.handle, len(K), ptr(K), ptr(mi), int(bool(logtransformconductivity)), ptr(s), ptr(dh), C.byref(bad)))
        if not hasattr(self, "inputs"):
            self.inputs = {}
        self.inputs.update(conductivities=_checksum(K), sources=_checksum(s), dirichletheads=_checksum(dh), metaindex=_checksum(mi),
                           logtransformconductivity=bool(logtransformconductivity))
        return self

    def csc(self):
        colptr = np.empty(self.n + 1, np.int64)
        rowval = np.empty(self.nnz, np.int64)
        nzval = np.empty(self.nnz, np.float64)
        self.check(load().fv_get_csc(self.handle, ptr(colptr), ptr(rowval), ptr(nzval)))
        return SparseMatrixCSC(self.n, self.n, colptr, rowval, nzval)

    def b(self):
        b = np.empty(self.n, np.float64)
        self.check(load().fv_get_b(self.handle, ptr(b)))
        return b

    def freenodes2nodes(self, result):
        r = af64(result)
        head = np.empty(self.N, np.float64)
        self.check(load().fv_freenodes2nodes(self.handle, ptr(r), ptr(head)))
        return head

    def solve_steady(self, x0=None, rtol=SQRT_EPS, maxiter=400, want_head=True, want_resnorm=True):
        x0_ = af64(x0) if x0 is not None else None
        head = np.empty(self.N, np.float64) if want_head else None
        res = np.empty(self.n, np.float64)
        hist = np.empty(max(int(maxiter), 1), np.float64) if want_resnorm else None
        info = SolveInfo()
        self.check(load().fv_solve_steady(self.handle, ptr(x0_), float(rtol), int(maxiter), ptr(head), ptr(res), ptr(hist), len(hist) if hist is not None else 0, C.byref(info)))
        ch = ConvergenceHistory(info, hist[: info.resnorm_len].copy() if hist is not None else None)
        return head, res, ch

    PRECONDITIONERS = {"jacobi": 0, "amg": 1, "auto": 2, "amg_gathered": 3}  # (3: row blocks with the coarse levels of the whole operator)

    def set_preconditioner(self, kind):
        """"jacobi" (default) or "amg": the aggregation-AMG V-cycle that stands where the reference uses
        AlgebraicMultigrid.ruge_stuben (FiniteVolume.jl:159-161)."""
        if kind not in self.PRECONDITIONERS:
            raise ValueError("preconditioner must be one of %s" % sorted(self.PRECONDITIONERS))
        self.check(load().fv_precond_set(self.handle, self.PRECONDITIONERS[kind]))
        return self

    def amg_info(self):
        """(rows, nnz) per level of the AMG hierarchy (built on demand)."""
        nl = C.c_int32()
        rows = np.zeros(32, np.int64)
        nnz = np.zeros(32, np.int64)
        self.check(load().fv_amg_info(self.handle, C.byref(nl), ptr(rows), ptr(nnz), 32))
        return rows[: nl.value].copy(), nnz[: nl.value].copy()

    def amg_apply(self, r, sigma=0.0):
        r_ = af64(r)
        z = np.empty(self.n, np.float64)
        self.check(load().fv_amg_apply(self.handle, ptr(r_), float(sigma), ptr(z)))
        return z

    def spmv(self, x, sigma=0.0):
        x_ = af64(x)
        y = np.empty(self.n, np.float64)
        self.check(load().fv_spmv(self.handle, ptr(x_), float(sigma), ptr(y)))
        return y

    def dot(self, a, b):
        a_, b_ = af64(a), af64(b)
        out = C.c_double()
        self.check(load().fv_dot(self.handle, ptr(a_), ptr(b_), C.byref(out)))
        return out.value

    def bench_spmv(self, sigma=0.0, reps=20):
        ms = C.c_double()
        self.check(load().fv_bench_spmv(self.handle, float(sigma), int(reps), C.byref(ms)))
        return ms.value

    SPMV_FORMS = {-1: "none yet", 0: "CSR wave-stream", 1: "sliced-DIA, slice by slice", 2: "sliced-DIA, plane-marching",
                  3: "symmetric plane-marching (diagonal + 3 upper diagonals)",
                  4: "symmetric, tiled traversal (diagonal + 3 upper diagonals, arms through LDS)",
                  5: "SELL-64, 16-bit column offsets (irregular meshes after the locality re-numbering)"}

    def reorder_info(self):
        """fv_problem_reorder_info -> dict(reordered, mean_before, mean_after, seconds)."""
        r, a, b, t = C.c_int32(), C.c_double(), C.c_double(), C.c_double()
        self.check(load().fv_problem_reorder_info(self.handle, C.byref(r), C.byref(a), C.byref(b), C.byref(t)))
        return dict(reordered=bool(r.value), mean_before=a.value, mean_after=b.value, seconds=t.value)

    def spmv_form(self):
        """(form id, form name, bytes one launch of it must move) of the most recent SpMV (fv_spmv_form)."""
        form, nbytes = C.c_int32(), C.c_int64()
        self.check(load().fv_spmv_form(self.handle, C.byref(form), C.byref(nbytes)))
        return form.value, self.SPMV_FORMS.get(form.value, "?"), nbytes.value

    def update_form(self):
        """Bytes per row the most recent K2S launch streams (fv_update_form); 0 before the first."""
        b = C.c_int32()
        self.check(load().fv_update_form(self.handle, C.byref(b)))
        return b.value

    def loop_form(self):
        """Bytes per row and iteration of the most recent many-iteration solve when its passes ran through the fused kernel (113),
        0 for the K1 + K2 + K3 loop (fv_loop_form)."""
        b = C.c_int32()
        self.check(load().fv_loop_form(self.handle, C.byref(b)))
        return b.value

    def step_form(self):
        """((set-up, first pass, flush) bytes per row outside the loop iterations, solves so far) of the most recent solve whose loop ran
        as one launch per iteration (fv_step_form; loop_form() 89 / 67)."""
        b = (C.c_int32 * 3)()
        n, tot = C.c_int64(), C.c_int64()
        self.check(load().fv_step_form(self.handle, b, C.byref(n), C.byref(tot)))
        self.bytes_total = tot.value  # running total of the bytes every Jacobi-PCG solve's launches had to move (differences over a run: its algorithmic bytes)
        return (int(b[0]), int(b[1]), int(b[2])), n.value

    def bytes_moved(self):
        """Running total of the algorithmic bytes of every Jacobi-PCG solve on this problem (fv_step_form's bytes_total)."""
        self.step_form()
        return self.bytes_total

    def fused_form(self):
        """(launches so far, bytes per row of its storage form, bytes per launch) of the fused step of the one-iteration regime
        (fv_fused_form)."""
        n, b, t = C.c_int64(), C.c_int32(), C.c_int64()
        self.check(load().fv_fused_form(self.handle, C.byref(n), C.byref(b), C.byref(t)))
        return n.value, b.value, t.value

    def fused_traversal(self):
        """0: the most recent fused launch walked 2-D tiles (or none has run), 1: contiguous chunks of a plane (fv_fused_traversal)."""
        k = C.c_int32()
        self.check(load().fv_fused_traversal(self.handle, C.byref(k)))
        return k.value

    def profile(self, on=True):
        """True / 1: time K1, K2 and K3 launches; 2: the SpMV (K1) only; False: off."""
        self.check(load().fv_profile_enable(self.handle, int(on)))

    def profile_get(self):
        """{kernel: (total_ms, launches)} measured with HIP events around every PCG launch."""
        out = {}
        for k, name in enumerate(("spmv_dot", "update", "pupdate")):
            ms, cnt = C.c_double(), C.c_int64()
            self.check(load().fv_profile_get(self.handle, k, C.byref(ms), C.byref(cnt)))
            out[name] = (ms.value, cnt.value)
        return out

    # ---- transient
    def transient_begin(self, Ss, volumes, u0_nodes):
        v = af64(volumes) if volumes is not None else None
        u = af64(u0_nodes) if u0_nodes is not None else None
        self.check(load().fv_transient_begin(self.handle, float(Ss), ptr(v), ptr(u)))
        if not hasattr(self, "inputs"):
            self.inputs = {}
        self.inputs.update(Ss=float(Ss), volumes=_checksum(v))
        return DeviceVector(self, 0, owned=False)

    def check_inputs(self, **given):
        """Raise if an argument differs from what this problem was built / assembled / started from (fingerprints recorded by create,
        assemble and transient_begin); arguments the problem has no record of (a regulargrid problem's mesh) are not checked."""
        rec = getattr(self, "inputs", {})
        for name, val in given.items():
            if name not in rec or val is None:
                continue
            if name == "neighbors":
                n1, n2 = _split_neighbors(val)
                got = (_checksum(n1), _checksum(n2))
            elif name in ("Ss", "logtransformconductivity"):
                got = type(rec[name])(val)
            elif name == "metaindex":
                got = _checksum(_metaindex_array(val, self.F))
            elif name == "dirichletnodes":
                got = _checksum(ai64(val))
            else:
                got = _checksum(af64(val))
            if got != rec[name]:
                raise FVError(_lib.FV_ERR_ARG, "%s differs from what the forward run's problem was built from: the device-resident adjoint sweep runs on that "
                                               "problem (re-assemble it, or use the host-closure path)" % name)

    def param_gradient_integral(self, ts, x_knots, lam_knots, scale_by_storage=False, logtransformconductivity=False):
        """fv_param_gradient_integral: exact time integral of (b_p - A_p u)' w for piecewise-linear u, w given at the
        common knots ts (rows of x_knots / lam_knots, free-indexed).  -> per-face K terms, per-face Dirichlet-head
        terms, per-free-row source terms."""
        t = af64(ts)
        X = np.ascontiguousarray(x_knots, dtype=np.float64)
        L = np.ascontiguousarray(lam_knots, dtype=np.float64)
        if X.shape != (len(t), self.n) or L.shape != X.shape:
            raise FVError(_lib.FV_ERR_ARG, "x_knots and lam_knots must be (len(ts), n) arrays")
        fk, fd, rs = np.empty(self.F), np.empty(self.F), np.empty(self.n)
        self.check(load().fv_param_gradient_integral(self.handle, len(t), ptr(t), ptr(X), ptr(L), int(bool(scale_by_storage)),
                                                     int(bool(logtransformconductivity)), ptr(fk), ptr(fd), ptr(rs)))
        return fk, fd, rs

    def param_jacobian_apply(self, x_free, lam_free, scale_by_storage=False, logtransformconductivity=False):
        """fv_param_jacobian_apply: (b_p - A_p u)' w at one time -> per-face K terms, per-face Dirichlet-head terms, per-free-row
        source terms (the action of the pointwise dfdp(u, t, p)' on w)."""
        x, lam = af64(x_free), af64(lam_free)
        if x.shape != (self.n,) or lam.shape != (self.n,):
            raise FVError(_lib.FV_ERR_ARG, "x_free and lam_free must have n entries")
        fk, fd, rs = np.empty(self.F), np.empty(self.F), np.empty(self.n)
        self.check(load().fv_param_jacobian_apply(self.handle, ptr(x), ptr(lam), int(bool(scale_by_storage)), int(bool(logtransformconductivity)),
                                                  ptr(fk), ptr(fd), ptr(rs)))
        return fk, fd, rs

    def new_state(self):
        s = C.c_int32()
        self.check(load().fv_state_alloc(self.handle, C.byref(s)))
        return DeviceVector(self, s.value, owned=True)

    def step(self, src, dst, dt, bhat=None, mode=_lib.FV_STEP_FORWARD, rtol=SQRT_EPS, maxiter=1000):
        bh = af64(bhat) if bhat is not None else None
        info = SolveInfo()
        self.check(load().fv_transient_step(self.handle, src.slot, dst.slot, float(dt), ptr(bh), int(mode), float(rtol), int(maxiter), C.byref(info)))
        return info

    def run_fixed(self, state, dt, nsteps, rtol=SQRT_EPS, maxiter=1000):
        iters = np.zeros(max(int(nsteps), 1), np.int32)
        info = SolveInfo()
        ms = C.c_double()
        self.check(load().fv_transient_run_fixed(self.handle, state.slot, float(dt), int(nsteps), float(rtol), int(maxiter), ptr(iters), C.byref(info), C.byref(ms)))
        return iters[: int(nsteps)], info, ms.value

    def run_adaptive(self, state, t0, tfinal, dt0=1.0, atol=1e-4, rtol=SQRT_EPS, maxiter=1000, max_outer=1 << 20):
        """The default stepper of backwardeulerintegrate (step doubling, transient.jl:78-121,136-154) with constant b,
        entirely on the device: the state is advanced to tfinal.  Returns (ts, nsolves, info), ts as the reference's."""
        ts = np.empty(int(max_outer) + 1, np.float64)
        nout, nsol = C.c_int64(), C.c_int64()
        info = SolveInfo()
        self.check(load().fv_transient_run_adaptive(self.handle, state.slot, float(t0), float(tfinal), float(dt0), float(atol), float(rtol), int(maxiter), int(max_outer), ptr(ts), C.byref(nout), C.byref(nsol), C.byref(info)))
        return ts[: nout.value + 1].copy(), nsol.value, info

    # ---- trajectories in HBM and the adjoint sweep over them (fv_trajectory.hip)
    def new_trajectory(self):
        return Trajectory(self)

    def record(self, trajectory, t0=0.0):
        """While set, run_adaptive pushes its initial state and every outer state, run_fixed every step's state at t0 + k dt
        (fv_trajectory_record); None stops it."""
        self.check(load().fv_trajectory_record(self.handle, trajectory.handle if trajectory is not None else None, float(t0)))

    def adjoint_run(self, u, obs, t0, tfinal, dt0=1.0, adaptive=True, atol=1e-4, rtol=SQRT_EPS, maxiter=1000, max_outer=1 << 20):
        """adjointintegrate (transient.jl:188-205) with the forcing dgdu(u_c, T - t) of getadjointfunctions evaluated on the device from
        the trajectory `u` and the Observation `obs` (fv_adjoint_run).  -> (Trajectory of lambda, outer steps, solves, info)."""
        lam = Trajectory(self)
        nout, nsol = C.c_int64(), C.c_int64()
        info = SolveInfo()
        self.check(load().fv_adjoint_run(self.handle, u.handle, obs.handle, float(t0), float(tfinal), float(dt0), int(bool(adaptive)), float(atol), float(rtol),
                                         int(maxiter), int(max_outer), lam.handle, C.byref(nout), C.byref(nsol), C.byref(info)))
        return lam, nout.value, nsol.value, info

    def param_gradient_integral_traj(self, u, lam, t0, t1, scale_by_storage=False, lam_scale=None, logtransformconductivity=False):
        """fv_param_gradient_integral_traj: the integral of (b_p - A_p u)' w over [t0, t1] with u and lambda read from trajectories in HBM."""
        sc = af64(lam_scale) if lam_scale is not None else None
        fk, fd, rs = np.empty(self.F), np.empty(self.F), np.empty(self.n)
        self.check(load().fv_param_gradient_integral_traj(self.handle, u.handle, lam.handle, float(t0), float(t1), int(bool(scale_by_storage)), ptr(sc),
                                                          int(bool(logtransformconductivity)), ptr(fk), ptr(fd), ptr(rs)))
        return fk, fd, rs

    def close(self):
        if getattr(self, "handle", None) and getattr(self.ctx, "handle", None):
            load().fv_problem_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Trajectory:
    """The stored states of a run — the reference's `us`, `ts` (transient.jl:136-154) — kept in HBM (fv_trajectory): knot k is a
    free-cell vector on the device at time ts[k]."""

    def __init__(self, problem):
        self.problem = problem
        h = C.c_void_p()
        problem.check(load().fv_trajectory_create(problem.handle, C.byref(h)))
        self.handle = h

    def push(self, state, t):
        self.problem.check(load().fv_trajectory_push_state(self.handle, state.slot, float(t)))
        return self

    def push_free(self, u_free, t):
        u = af64(u_free)
        self.problem.check(load().fv_trajectory_push_free(self.handle, ptr(u), float(t)))
        return self

    def __len__(self):
        k = C.c_int64()
        self.problem.check(load().fv_trajectory_size(self.handle, C.byref(k)))
        return k.value

    @property
    def ts(self):
        t = np.empty(max(len(self), 1), np.float64)
        self.problem.check(load().fv_trajectory_times(self.handle, ptr(t), len(t)))
        return t[: len(self)]

    def free_values(self, k):
        out = np.empty(self.problem.n, np.float64)
        self.problem.check(load().fv_trajectory_get_free(self.handle, int(k), ptr(out)))
        return out

    def node_values(self, k):
        out = np.empty(self.problem.N, np.float64)
        self.problem.check(load().fv_trajectory_get_nodes(self.handle, int(k), ptr(out)))
        return out

    def at(self, t):
        """u_c(t) over the free cells (linear between the knots; IndexError outside them, like the interpolant)."""
        out = np.empty(self.problem.n, np.float64)
        try:
            self.problem.check(load().fv_trajectory_eval_free(self.handle, float(t), ptr(out)))
        except FVError as e:
            if "BoundsError" in str(e):
                raise IndexError(str(e)) from None
            raise
        return out

    def reverse_time(self, T):
        self.problem.check(load().fv_trajectory_reverse_time(self.handle, float(T)))
        return self

    def close(self):
        # (safe in either order with the problem's own close: fv_problem_destroy detaches what is still alive — its HBM released, its
        # problem pointer cleared — and fv_trajectory_destroy then frees the handle alone)
        if getattr(self, "handle", None):
            load().fv_trajectory_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Observation:
    """Observation rows (1-based free indices, `obsfreenodes`) with uobs_i(t) and sigma(i, t) as piecewise-linear series over
    `tobs`, on the device (fv_observation): what dgdu and g of transientadjointutils.jl:4-21 read."""

    def __init__(self, problem, obs_free, tobs, uobs, sigma=None):
        self.problem = problem
        idx = np.ascontiguousarray(obs_free, dtype=np.int64)
        t = af64(tobs)
        U = np.ascontiguousarray(uobs, dtype=np.float64).reshape(len(t), len(idx))
        S = np.ascontiguousarray(sigma, dtype=np.float64).reshape(len(t), len(idx)) if sigma is not None else None
        h = C.c_void_p()
        problem.check(load().fv_observation_create(problem.handle, len(idx), ptr(idx), len(t), ptr(t), ptr(U), ptr(S), C.byref(h)))
        self.handle = h

    def integral(self, u, t0, t1):
        """G = the integral of g(u_c, t) over [t0, t1] (transientadjointutils.jl:46-49), exact between the knots."""
        G = C.c_double()
        self.problem.check(load().fv_observation_integral(u.handle, self.handle, float(t0), float(t1), C.byref(G)))
        return G.value

    def close(self):
        if getattr(self, "handle", None):  # (in either order with the problem's close, as Trajectory.close)
            load().fv_observation_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceVector:
    """A state vector over the free cells, resident on the GPU (a slot of its Problem)."""

    def __init__(self, problem, slot, owned=True):
        self.problem, self.slot, self.owned = problem, int(slot), owned

    def free_values(self):
        out = np.empty(self.problem.n, np.float64)
        self.problem.check(load().fv_state_get_free(self.problem.handle, self.slot, ptr(out)))
        return out

    def node_values(self):
        out = np.empty(self.problem.N, np.float64)
        self.problem.check(load().fv_state_get_nodes(self.problem.handle, self.slot, ptr(out)))
        return out

    def set_nodes(self, u_nodes):
        u_ = af64(u_nodes)
        if len(u_) != self.problem.N:
            raise ValueError("set_nodes wants one value per node (%d), got %d" % (self.problem.N, len(u_)))
        self.problem.check(load().fv_state_set_nodes(self.problem.handle, self.slot, ptr(u_)))
        return self

    def set_free(self, u):
        u_ = af64(u)
        self.problem.check(load().fv_state_set_free(self.problem.handle, self.slot, ptr(u_)))
        return self

    def norm2_diff(self, other):
        out = C.c_double()
        self.problem.check(load().fv_state_norm2_diff(self.problem.handle, self.slot, other.slot, C.byref(out)))
        return out.value

    def __array__(self, dtype=None, copy=None):
        return self.free_values()

    def __len__(self):
        return self.problem.n

    def __del__(self):
        try:
            if self.owned and self.problem.handle:
                load().fv_state_free(self.problem.handle, self.slot)
        except Exception:
            pass


# ------------------------------------------------------------------ assembly (FiniteVolume.jl:75-155)
class DeviceMatrix(SparseMatrixCSC):
    """The matrix assembleA returns: a genuine SparseMatrixCSC (host copies of the
    index/value arrays) that also remembers the device problem it was assembled on."""

    def __init__(self, problem, csc):
        super().__init__(csc.m, csc.n, csc.colptr, csc.rowval, csc.nzval)
        self.problem = problem


def renumbered_inputs(neighbors, N, dirichletnodes):
    """(order, rank, neighbors', dirichletnodes') for meshio.locality_order: faces keep their order and orientation."""
    from . import meshio

    n1, n2 = _split_neighbors(neighbors)
    order, rank = meshio.locality_order(n1, n2, N)
    return order, rank, (rank[n1 - 1], rank[n2 - 1]), rank[ai64(dirichletnodes) - 1]


def free_permutation(problem, rank):
    """freenode in the caller's numbering, and for every free cell in the caller's (reference) free order the 0-based
    free index the renumbered device problem gave it."""
    freenode_int, n2f_int = problem.free_maps()
    freenode = freenode_int[rank - 1]
    return freenode, n2f_int[rank[np.nonzero(freenode)[0]] - 1] - 1


def _solvediffusion_reordered(neighbors, areasoverlengths, conductivities, sources, dirichletnodes, dirichletheads, maxiter, rtol, ctx, preconditioner):
    import scipy.sparse as sp

    src = af64(sources)
    order, rank, nb2, dn2 = renumbered_inputs(neighbors, len(src), dirichletnodes)
    p = _assembled_problem(nb2, areasoverlengths, conductivities, src[order - 1], dn2, dirichletheads, None, False, ctx)
    if preconditioner != "jacobi":
        p.set_preconditioner(preconditioner)
    head2, _, ch = p.solve_steady(None, rtol, maxiter)
    freenode, pf = free_permutation(p, rank)
    A2 = p.csc().toscipy().tocsr()
    A = sp.csc_matrix(A2[pf][:, pf])
    A.sort_indices()
    Acsc = SparseMatrixCSC(A.shape[0], A.shape[1], A.indptr.astype(np.int64) + 1, A.indices.astype(np.int64) + 1, A.data.copy())
    return head2[rank - 1], ch, DeviceMatrix(p, Acsc), p.b()[pf], freenode


def _assembled_problem(neighbors, areasoverlengths, conductivities, sources, dirichletnodes, dirichletheads, metaindex, logtransformconductivity, ctx=None):
    p = Problem.create(neighbors, areasoverlengths, len(sources), dirichletnodes, ctx)
    p.assemble(conductivities, sources, dirichletheads, metaindex, logtransformconductivity)
    return p


def assembleA(neighbors, areasoverlengths, conductivities, sources, dirichletnodes, dirichletheads, metaindex=None, logtransformconductivity=False, ctx=None):
    """FiniteVolume.jl:75-108"""
    src = af64(sources)
    # assembleA itself never validates the sources (only assembleb does, :111)
    p = _assembled_problem(neighbors, areasoverlengths, conductivities, np.zeros_like(src), dirichletnodes, dirichletheads, metaindex, logtransformconductivity, ctx)
    return DeviceMatrix(p, p.csc())


def assembleb(neighbors, areasoverlengths, conductivities, sources, dirichletnodes, dirichletheads, metaindex=None, logtransformconductivity=False, ctx=None):
    """FiniteVolume.jl:110-139"""
    p = _assembled_problem(neighbors, areasoverlengths, conductivities, sources, dirichletnodes, dirichletheads, metaindex, logtransformconductivity, ctx)
    return p.b()


def freenodes2nodes(result, sources, dirichletnodes, dirichletheads, ctx=None):
    """FiniteVolume.jl:141-155 -> head, freenode, nodei2freenodei"""
    ctx = ctx or default_context()
    getnodei2dirichleti(sources, dirichletnodes, ctx)  # the reference validates here too (:142)
    N = len(sources)
    p = Problem.create(np.empty((0, 2), np.int64), np.empty(0), N, dirichletnodes, ctx)
    p.assemble(np.empty(0), sources, dirichletheads)
    head = p.freenodes2nodes(result)
    freenode, n2f = p.free_maps()
    return head, freenode, n2f


def solvediffusion(neighbors, areasoverlengths, conductivities, sources, dirichletnodes, dirichletheads, maxiter=400, rtol=SQRT_EPS, ctx=None, preconditioner="auto", reorder=False):
    """FiniteVolume.jl:157-165 -> head, ch, A, b, freenode.

    The reference preconditions CG with Ruge-Stuben AMG.  preconditioner="jacobi": the Jacobi-PCG of
    BASELINE.json's north_star (`maxiter` counts its iterations); "amg": the aggregation-AMG V-cycle
    (fv_precond_set); "auto" (default): Jacobi-PCG for min(maxiter/4, 100) iterations — easy problems never
    pay for a hierarchy — then AMG-PCG from that iterate, the shape of the reference's defaultlinearsolver
    (transient.jl:50-58), so that the reference's maxiter = 400 keeps converging on high-contrast or anisotropic
    problems.  As in the reference, non-convergence is reported through ch.isconverged, not raised.
    reorder=True: the device works on the mesh renumbered by meshio.locality_order (worth a factor ~2 in SpMV rate on
    meshes whose cells are numbered without regard to position); head, A, b and freenode still come back in the
    caller's numbering (A and b are the same numbers, permuted back on the host)."""
    if reorder:
        return _solvediffusion_reordered(neighbors, areasoverlengths, conductivities, sources, dirichletnodes, dirichletheads, maxiter, rtol, ctx, preconditioner)
    p = _assembled_problem(neighbors, areasoverlengths, conductivities, sources, dirichletnodes, dirichletheads, None, False, ctx)
    if preconditioner != "jacobi":
        p.set_preconditioner(preconditioner)
    head, _, ch = p.solve_steady(None, rtol, maxiter)
    freenode, _ = p.free_maps()
    return head, ch, DeviceMatrix(p, p.csc()), p.b(), freenode
