"""CPU ORACLE driver — TEST INFRASTRUCTURE ONLY (see fv_oracle.c header).

ctypes wrappers over oracle/_build/libfvoracle.so plus a restatement, in
Python control flow over numpy vectors, of the reference's steppers
(/root/reference/src/transient.jl:50-174) and high-level entry points
(src/FiniteVolume.jl:141-165).  Indices are Julia's: int64, 1-based.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.  Parity status: pinned by the reference's own
known-answer tests (tests/test_oracle_*.py); the reference (Julia) cannot run
in this image, see DESIGN.md.
"""
import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBPATH = os.path.join(_HERE, "_build", "libfvoracle.so")

FVO_ERR_SOURCE_AT_DIRICHLET = 2


class OracleError(RuntimeError):
    pass


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def _load():
    if not os.path.exists(_LIBPATH):
        build()
    return C.CDLL(_LIBPATH)


_lib = _load()
_lib.fvo_norm2.restype = C.c_double
_lib.fvo_norm2_diff.restype = C.c_double

_i64 = np.int64
_f64 = np.float64


def _pi(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def _pd(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _pb(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def _ai(a):
    return np.ascontiguousarray(a, dtype=_i64)


def _ad(a):
    return np.ascontiguousarray(a, dtype=_f64)


def _check(rc, badnode=None):
    if rc == 0:
        return
    if rc == FVO_ERR_SOURCE_AT_DIRICHLET:
        node = badnode.value if badnode is not None else "?"
        # message text of FiniteVolume.jl:26
        raise OracleError(
            "There cannot be a source at a Dirichlet node, but node %s is a Dirichlet node where a source is located."
            % node
        )
    raise OracleError("oracle error code %d" % rc)


# ---------------------------------------------------------------- grid (a1, a2)
def linrange(a, b, n):
    out = np.empty(n, _f64)
    _check(_lib.fvo_linrange(C.c_double(a), C.c_double(b), C.c_int64(n), _pd(out)))
    return out


def regulargrid(mins, maxs, ns, want_coords=True):
    """grid.jl:56-110 -> coords (3,N), node1, node2 (the Pair halves), aol, volumes."""
    if not (len(mins) == len(maxs) == len(ns)):
        raise AssertionError("length(mins) == length(maxs) == length(ns)")
    if len(mins) != 3:
        raise OracleError("only 3 dimensions supported")
    mins_ = _ad(mins)
    maxs_ = _ad(maxs)
    ns_ = _ai(ns)
    N = C.c_int64()
    F = C.c_int64()
    _check(_lib.fvo_regulargrid_sizes(_pi(ns_), C.byref(N), C.byref(F)))
    N, F = N.value, F.value
    coords = np.empty((N, 3), _f64) if want_coords else None
    node1 = np.empty(F, _i64)
    node2 = np.empty(F, _i64)
    aol = np.empty(F, _f64)
    volumes = np.empty(N, _f64)
    _check(
        _lib.fvo_regulargrid(
            _pd(mins_), _pd(maxs_), _pi(ns_), _pd(coords) if want_coords else None, _pi(node1), _pi(node2), _pd(aol), _pd(volumes)
        )
    )
    return (coords.T if want_coords else None), node1, node2, aol, volumes


def nodehycos2neighborhycos(node1, node2, nodehycos, logtransformhyco=False):
    node1, node2 = _ai(node1), _ai(node2)
    nh = _ad(np.asarray(nodehycos).ravel(order="F"))
    out = np.empty(len(node1), _f64)
    _check(
        _lib.fvo_nodehycos2neighborhycos(
            C.c_int64(len(node1)), _pi(node1), _pi(node2), C.c_int64(len(nh)), _pd(nh), C.c_int(bool(logtransformhyco)), _pd(out)
        )
    )
    return out


# ---------------------------------------------------------------- maps (a3, a4)
def getfreenodes(n, dirichletnodes):
    d = _ai(dirichletnodes)
    freenode = np.empty(n, np.uint8)
    n2f = np.empty(n, _i64)
    nfree = C.c_int64()
    _check(_lib.fvo_getfreenodes(C.c_int64(n), C.c_int64(len(d)), _pi(d), _pb(freenode), _pi(n2f), C.byref(nfree)))
    return freenode.astype(bool), n2f


def getnodei2dirichleti(sources, dirichletnodes):
    s = _ad(sources)
    d = _ai(dirichletnodes)
    out = np.empty(len(s), _i64)
    bad = C.c_int64()
    _check(_lib.fvo_getnodei2dirichleti(C.c_int64(len(s)), _pd(s), C.c_int64(len(d)), _pi(d), _pi(out), C.byref(bad)), bad)
    return out


# ---------------------------------------------------------------- sparse + assembly (a5, a6, a7)
class CSC:
    """Julia SparseMatrixCSC{Float64,Int64}: 1-based colptr/rowval, nzval."""

    def __init__(self, m, n, colptr, rowval, nzval):
        self.m, self.n = int(m), int(n)
        self.colptr, self.rowval, self.nzval = colptr, rowval, nzval

    @property
    def shape(self):
        return (self.m, self.n)

    def copy(self):
        return CSC(self.m, self.n, self.colptr.copy(), self.rowval.copy(), self.nzval.copy())

    def toscipy(self):
        import scipy.sparse as sp

        return sp.csc_matrix((self.nzval, self.rowval - 1, self.colptr - 1), shape=(self.m, self.n))

    def matvec(self, x):
        x = _ad(x)
        y = np.empty(self.m, _f64)
        _lib.fvo_spmv_csc(C.c_int64(self.m), C.c_int64(self.n), _pi(self.colptr), _pi(self.rowval), _pd(self.nzval), _pd(x), _pd(y))
        return y


def sparse(I, J, V, m, n):
    """SparseArrays.sparse(I, J, V, m, n, +)."""
    I, J, V = _ai(I), _ai(J), _ad(V)
    L = len(I)
    colptr = np.empty(n + 1, _i64)
    rowval = np.empty(max(L, 1), _i64)
    nzval = np.empty(max(L, 1), _f64)
    nnz = C.c_int64()
    _check(_lib.fvo_sparse(C.c_int64(L), _pi(I), _pi(J), _pd(V), C.c_int64(m), C.c_int64(n), _pi(colptr), _pi(rowval), _pd(nzval), C.byref(nnz)))
    return CSC(m, n, colptr, rowval[: nnz.value].copy(), nzval[: nnz.value].copy())


def _meta(metaindex, F):
    if metaindex is None:
        return None
    if callable(metaindex):
        return _ai([metaindex(i) for i in range(1, F + 1)])
    return _ai(metaindex)


def assembleA(node1, node2, areasoverlengths, conductivities, sources, dirichletnodes, dirichletheads, metaindex=None, logtransformconductivity=False):
    node1, node2 = _ai(node1), _ai(node2)
    aol, K, d = _ad(areasoverlengths), _ad(conductivities), _ai(dirichletnodes)
    N, F = len(sources), len(node1)
    mi = _meta(metaindex, F)
    colptr = np.empty(N + 1, _i64)
    rowval = np.empty(4 * F + 1, _i64)
    nzval = np.empty(4 * F + 1, _f64)
    nfree = C.c_int64()
    nnz = C.c_int64()
    _check(
        _lib.fvo_assembleA(
            C.c_int64(N), C.c_int64(F), _pi(node1), _pi(node2), _pd(aol), C.c_int64(len(K)), _pd(K),
            _pi(mi) if mi is not None else None, C.c_int(bool(logtransformconductivity)), C.c_int64(len(d)), _pi(d),
            _pi(colptr), _pi(rowval), _pd(nzval), C.byref(nfree), C.byref(nnz),
        )
    )
    n = nfree.value
    return CSC(n, n, colptr[: n + 1].copy(), rowval[: nnz.value].copy(), nzval[: nnz.value].copy())


def assembleb(node1, node2, areasoverlengths, conductivities, sources, dirichletnodes, dirichletheads, metaindex=None, logtransformconductivity=False):
    node1, node2 = _ai(node1), _ai(node2)
    aol, K, d, s, dh = _ad(areasoverlengths), _ad(conductivities), _ai(dirichletnodes), _ad(sources), _ad(dirichletheads)
    N, F = len(s), len(node1)
    mi = _meta(metaindex, F)
    nfree = N - len(np.unique(d))
    b = np.empty(max(nfree, 1), _f64)
    bad = C.c_int64()
    _check(
        _lib.fvo_assembleb(
            C.c_int64(N), C.c_int64(F), _pi(node1), _pi(node2), _pd(aol), C.c_int64(len(K)), _pd(K),
            _pi(mi) if mi is not None else None, C.c_int(bool(logtransformconductivity)), _pd(s), C.c_int64(len(d)), _pi(d), _pd(dh),
            _pd(b), C.byref(bad),
        ),
        bad,
    )
    return b[:nfree]


def freenodes2nodes(result, sources, dirichletnodes, dirichletheads):
    r, s, d, dh = _ad(result), _ad(sources), _ai(dirichletnodes), _ad(dirichletheads)
    head = np.empty(len(s), _f64)
    bad = C.c_int64()
    _check(_lib.fvo_freenodes2nodes(C.c_int64(len(s)), _pd(r), _pd(s), C.c_int64(len(d)), _pi(d), _pd(dh), _pd(head), C.byref(bad)), bad)
    freenode, n2f = getfreenodes(len(s), d)
    return head, freenode, n2f


# ---------------------------------------------------------------- transient pieces (a9, a10)
def freenodei2nodei(nodei2freenodei):
    """The Dict built at transient.jl:158, as a dense 1-based array over free indices."""
    n2f = np.asarray(nodei2freenodei)
    nodes = np.nonzero(n2f > 0)[0]
    f2n = np.empty(len(nodes), _i64)
    f2n[n2f[nodes] - 1] = nodes + 1
    return f2n


def scalebyvolume_b(b, volumes, f2n):
    b = _ad(b)
    v, f2n = _ad(volumes), _ai(f2n)
    _lib.fvo_scalebyvolume_b(C.c_int64(len(b)), _pd(b), _pd(v), _pi(f2n))
    return b


def scalebyvolume_A(A, volumes, f2n):
    v, f2n = _ad(volumes), _ai(f2n)
    _lib.fvo_scalebyvolume_A(C.c_int64(A.n), _pi(A.colptr), _pi(A.rowval), _pd(A.nzval), _pd(v), _pi(f2n))


def diagonalupdate(A, increment):
    if isinstance(A, CSC):
        _lib.fvo_diagonalupdate(C.c_int64(A.n), _pi(A.colptr), _pi(A.rowval), _pd(A.nzval), C.c_double(increment))
    else:  # dense Array{T,2}, transient.jl:1-5
        for i in range(A.shape[0]):
            A[i, i] += increment


# ---------------------------------------------------------------- linear solvers
SQRT_EPS = math.sqrt(np.finfo(np.float64).eps)


class ConvergenceHistory:
    def __init__(self, isconverged, iters, resnorm):
        self.isconverged, self.iters = bool(isconverged), int(iters)
        self.data = {"resnorm": resnorm}


def cg(A, b, x0=None, tol=SQRT_EPS, maxiter=None):
    """IterativeSolvers 0.8.1 cg / cg! (x0=None -> `cg`, initially zero)."""
    b = _ad(b)
    n = A.n
    maxiter = n if maxiter is None else int(maxiter)
    x = np.zeros(n, _f64) if x0 is None else _ad(x0).copy()
    res = np.empty(maxiter + 1, _f64)
    it = C.c_int64()
    conv = C.c_int()
    _check(
        _lib.fvo_cg(
            C.c_int64(n), _pi(A.colptr), _pi(A.rowval), _pd(A.nzval), _pd(b), _pd(x), C.c_double(tol), C.c_int64(maxiter),
            C.c_int(x0 is None), C.byref(it), C.byref(conv), _pd(res),
        )
    )
    return x, ConvergenceHistory(conv.value, it.value, res[: it.value].copy())


def pcg_jacobi(A, b, x0=None, shift=None, tol=SQRT_EPS, maxiter=None):
    """The algorithm the HIP build runs: Jacobi-PCG on (A + diag(shift)) x = b."""
    b = _ad(b)
    n = A.n
    maxiter = n if maxiter is None else int(maxiter)
    x = np.zeros(n, _f64) if x0 is None else _ad(x0).copy()
    sh = _ad(shift) if shift is not None else None
    it = C.c_int64()
    conv = C.c_int()
    rel = C.c_double()
    _check(
        _lib.fvo_pcg_jacobi(
            C.c_int64(n), _pi(A.colptr), _pi(A.rowval), _pd(A.nzval), _pd(sh) if sh is not None else None, _pd(b), _pd(x),
            C.c_double(tol), C.c_int64(maxiter), C.byref(it), C.byref(conv), C.byref(rel),
        )
    )
    ch = ConvergenceHistory(conv.value, it.value, None)
    ch.final_relres = rel.value
    return x, ch


fallbacks = 0  # how many times defaultlinearsolver needed its second stage


def defaultlinearsolver(A, b, x0):
    """transient.jl:50-58.  Stage 1 is the reference's unpreconditioned cg!
    (maxiter=100).  Stage 2 in the reference is Ruge-Stuben-AMG-PCG
    (AlgebraicMultigrid 0.2.2, not restated: parity unpinned); the oracle
    continues with Jacobi-PCG from the partial iterate and counts the event."""
    global fallbacks
    result, ch = cg(A, b, x0=x0, maxiter=100)
    if not ch.isconverged:
        fallbacks += 1
        result, ch = pcg_jacobi(A, b, x0=result, maxiter=100)
    return result


def directlinearsolver(A, b, x0):
    """`A \\ b` — the hook used at test/ode.jl:36 (dense or sparse)."""
    if isinstance(A, CSC):
        import scipy.sparse.linalg as spla

        return spla.splu(A.toscipy()).solve(_ad(b))
    return np.linalg.solve(A, b)


def tightcgsolver(tol=1e-13, maxiter=100000):
    def solver(A, b, x0):
        x, _ = cg(A, b, x0=x0, tol=tol, maxiter=maxiter)
        return x

    return solver


# ---------------------------------------------------------------- steppers (a11-a14)
def _copyA(A):
    return A.copy()


def backwardeuleronestep(rhs, A, b, u_k, dt, linearsolver, atol):
    """transient.jl:65-76 (b already evaluated at the start of the step)."""
    if dt <= 0:
        raise OracleError("time step must be positive")
    rhs[:] = b + u_k / dt
    diagonalupdate(A, 1 / dt)
    onestep = linearsolver(A, rhs, u_k)
    diagonalupdate(A, -1 / dt)
    return onestep


def backwardeulertwostep(rhs, A, getb, u_k, t, dt, linearsolver, atol, onestep=None):
    """transient.jl:78-87"""
    if onestep is None:
        onestep = backwardeuleronestep(rhs, A, getb(t), u_k, dt, linearsolver, atol)
    twostep1 = backwardeuleronestep(rhs, A, getb(t), u_k, 0.5 * dt, linearsolver, atol)
    twostep = backwardeuleronestep(rhs, A, getb(t + 0.5 * dt), twostep1, 0.5 * dt, linearsolver, atol)
    err = float(np.linalg.norm(onestep - twostep))
    if err < atol:
        return twostep, dt, err < atol / 4
    return twostep1, 0.5 * dt, False


def adaptivebackwardeulerstep(rhs, A, getb, u_k, t, dt, linearsolver, atol, callback):
    """transient.jl:89-121"""
    callback(t, dt)
    u_new, laststeptime, increasestepsize = backwardeulertwostep(rhs, A, getb, u_k, t, dt, linearsolver, atol)
    if laststeptime < dt:
        laststepfailed = True
        elapsedtime = 0.0
        u_elapsedtime = u_k
        targetdt = laststeptime
        while elapsedtime < dt:
            callback(t, dt)
            if laststepfailed:
                u_new, laststeptime, increasestepsize = backwardeulertwostep(rhs, A, getb, u_elapsedtime, t + elapsedtime, targetdt, linearsolver, atol, u_new)
            else:
                u_new, laststeptime, increasestepsize = backwardeulertwostep(rhs, A, getb, u_elapsedtime, t + elapsedtime, targetdt, linearsolver, atol)
            if laststeptime == targetdt:
                elapsedtime += laststeptime
                u_elapsedtime = u_new
                if increasestepsize:
                    targetdt = 2 * laststeptime
                laststepfailed = False
            elif laststeptime < targetdt:
                targetdt = laststeptime
                laststepfailed = True
            else:
                raise OracleError("Code is broken -- laststeptime should never be greater than targetdt")
            targetdt = min(targetdt, dt - elapsedtime)
    return u_new, laststeptime, increasestepsize


def fixedbackwardeulerstep(rhs, A, getb, u_k, t, dt, linearsolver, atol, callback):
    """transient.jl:130-134"""
    callback(t, dt)
    u_new = backwardeuleronestep(rhs, A, getb(t), u_k, dt, linearsolver, atol)
    return u_new, dt, False


def backwardeulerintegrate_generic(u0, A, b_or_getb, dt0, t0, tfinal, stepper=adaptivebackwardeulerstep, linearsolver=defaultlinearsolver, atol=1e-4, callback=lambda t, dt: None):
    """transient.jl:123-154 (u0 over the unknowns; A dense ndarray or CSC)."""
    getb = b_or_getb if callable(b_or_getb) else (lambda t: b_or_getb)
    u0 = _ad(u0)
    us = [u0]
    ts = [t0]
    rhs = np.empty_like(u0)
    A = _copyA(A)
    dt = min(dt0, tfinal - t0)
    while ts[-1] < tfinal:
        solution, laststeptime, increasestepsize = stepper(rhs, A, getb, us[-1], ts[-1], dt, linearsolver, atol, callback)
        us.append(solution)
        ts.append(ts[-1] + dt)
        if increasestepsize:
            newdt = min(tfinal - ts[-1], 2 * laststeptime)
        else:
            newdt = min(tfinal - ts[-1], laststeptime)
        dt = newdt
    return us, ts


def backwardeulerintegrate(u0, tspan, Ss, volumes, node1, node2, areasoverlengths, conductivities, sources, dirichletnodes, dirichletheads, metaindex=None, logtransformconductivity=False, getb=None, dt0=1.0, **kwargs):
    """transient.jl:156-174 (both the constant-b and the getb::Function method)."""
    u0 = _ad(u0)
    volumes = _ad(volumes)
    freenodes, n2f = getfreenodes(len(u0), dirichletnodes)
    f2n = freenodei2nodei(n2f)
    if getb is None:
        b = assembleb(node1, node2, areasoverlengths, conductivities, sources, dirichletnodes, dirichletheads, metaindex, logtransformconductivity)
        b = scalebyvolume_b(b, Ss * volumes, f2n)
        getb = lambda t: b  # noqa: E731
    A = assembleA(node1, node2, areasoverlengths, conductivities, sources, dirichletnodes, dirichletheads, metaindex, logtransformconductivity)
    scalebyvolume_A(A, Ss * volumes, f2n)
    u0f = u0[freenodes]
    us, ts = backwardeulerintegrate_generic(u0f, A, getb, dt0, tspan[0], tspan[1], **kwargs)
    us = [freenodes2nodes(x, sources, dirichletnodes, dirichletheads)[0] for x in us]
    return us, ts


def solvediffusion(node1, node2, areasoverlengths, conductivities, sources, dirichletnodes, dirichletheads, maxiter=400, solver="direct", tol=SQRT_EPS):
    """FiniteVolume.jl:157-165.  The reference solves with RS-AMG-PCG (not
    restated); `solver` picks what the oracle uses instead: "direct" (splu),
    "cg" (IterativeSolvers cg from zero) or "pcg" (Jacobi-PCG from zero)."""
    A = assembleA(node1, node2, areasoverlengths, conductivities, sources, dirichletnodes, dirichletheads)
    b = assembleb(node1, node2, areasoverlengths, conductivities, sources, dirichletnodes, dirichletheads)
    if solver == "direct":
        result, ch = directlinearsolver(A, b, None), ConvergenceHistory(True, 0, None)
    elif solver == "cg":
        result, ch = cg(A, b, tol=tol, maxiter=maxiter)
    else:
        result, ch = pcg_jacobi(A, b, tol=tol, maxiter=maxiter)
    head, freenode, _ = freenodes2nodes(result, sources, dirichletnodes, dirichletheads)
    return head, ch, A, b, freenode
