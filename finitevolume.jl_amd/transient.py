"""Implicit (backward-Euler) integration — mirror of /root/reference/src/transient.jl.

The steppers keep the reference's protocol
    stepper!(rhs, A, getb, u_k, t, dt, linearsolver, atol, callback) -> (u_new, laststeptime, increasestepsize)
and its control flow (step doubling, sub-stepping without overshoot).  When `A`
is a device operator the vectors are DeviceVector handles and every linear solve
is the HIP Jacobi-PCG; a user-supplied `linearsolver(A, rhs, x0)` together with a
host matrix keeps the reference's host semantics (the plugin seam of
transient.jl:136, exercised by test/ode.jl:36).
"""
import numpy as np

from . import _lib
from ._lib import FVError
from .core import SQRT_EPS, DeviceMatrix, DeviceVector, Problem, SparseMatrixCSC, _assembled_problem, af64


class DevicePCG:
    """The default `linearsolver`: Jacobi-PCG on the GPU.  Replaces
    defaultlinearsolver (transient.jl:50-58: CG <=100 its, then AMG-PCG <=100 its;
    returns whether or not converged).  `last` holds the latest fv_solve_info."""

    def __init__(self, rtol=SQRT_EPS, maxiter=1000):
        self.rtol, self.maxiter = rtol, maxiter
        self.last = None
        self.total_iters = 0
        self.solves = 0


defaultlinearsolver = DevicePCG()


class DeviceOperator:
    """(I/dt + D^-1 A) of the reference, held on the GPU as A and D separately.
    `adjoint=True` is transpose(A_scaled) (transient.jl:193)."""

    def __init__(self, problem, adjoint=False):
        self.problem, self.adjoint = problem, adjoint

    @property
    def shape(self):
        return (self.problem.n, self.problem.n)

    def transpose(self):
        return DeviceOperator(self.problem, not self.adjoint)

    def copy(self):  # transient.jl:140 copies A defensively; the device operator is never mutated
        return self


def diagonalupdate(A, increment):
    """transient.jl:1-5, 37-48 (host matrices only; the device operator applies the shift in its kernels)"""
    if isinstance(A, SparseMatrixCSC):
        for j in range(A.n):
            lo, hi = A.colptr[j] - 1, A.colptr[j + 1] - 1
            hit = np.nonzero(A.rowval[lo:hi] == j + 1)[0]
            A.nzval[lo + hit] += increment
    else:
        for i in range(A.shape[0]):
            A[i, i] += increment


def scalebyvolume(x, volumes, freenodei2nodei):
    """scalebyvolume! (transient.jl:7-22) for host vectors / SparseMatrixCSC."""
    vols = af64(volumes)
    f2n = freenodei2nodei
    if isinstance(f2n, dict):
        f2n = np.array([f2n[i] for i in range(1, len(f2n) - (1 if -1 in f2n else 0) + 1)], np.int64)
    f2n = np.asarray(f2n, np.int64)
    if isinstance(x, SparseMatrixCSC):
        x.nzval /= vols[f2n[x.rowval - 1] - 1]
    else:
        x /= vols[f2n[: len(x)] - 1]
    return x


def _as_operator(A):
    """Turn the `A` argument of the generic integrator into something steppable."""
    if isinstance(A, DeviceOperator):
        return A
    if isinstance(A, DeviceMatrix) and getattr(A, "_scaled_operator", None) is not None:
        return A._scaled_operator
    if isinstance(A, SparseMatrixCSC):
        csc = A
    elif hasattr(A, "tocsc"):
        csc = SparseMatrixCSC.fromscipy(A)
    else:
        csc = SparseMatrixCSC.fromdense(A)
    p = Problem.from_csc(csc)
    p.transient_begin(1.0, None, None)  # D = I
    return DeviceOperator(p)


# ------------------------------------------------------------------ steppers
def backwardeuleronestep(rhs, A, b, u_k, dt, linearsolver, atol):
    """transient.jl:60-76.  `b` is already evaluated at the start of the step."""
    if dt <= 0:
        raise FVError(_lib.FV_ERR_DT, "time step must be positive")
    if isinstance(A, DeviceOperator):
        if not isinstance(linearsolver, DevicePCG):
            raise TypeError("a device operator is solved by DevicePCG; pass a host matrix to use a custom linearsolver")
        p = A.problem
        dst = p.new_state()
        mode = _lib.FV_STEP_ADJOINT if A.adjoint else _lib.FV_STEP_FORWARD
        info = p.step(u_k, dst, dt, b, mode, linearsolver.rtol, linearsolver.maxiter)
        linearsolver.last = info
        linearsolver.total_iters += info.iters
        linearsolver.solves += 1
        return dst
    # host path: the reference's own sequence
    rhs[:] = b + u_k / dt
    diagonalupdate(A, 1 / dt)
    onestep = linearsolver(A, rhs, u_k)
    diagonalupdate(A, -1 / dt)
    return onestep


def _norm_diff(a, b):
    if isinstance(a, DeviceVector):
        return a.norm2_diff(b)
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)))


def backwardeulertwostep(rhs, A, getb, u_k, t, dt, linearsolver, atol, onestep=None):
    """Step doubling (transient.jl:78-87): one full step against two half steps; the
    two-half-step state is accepted when they agree to atol (and the step may grow when
    they agree to atol/4), otherwise the first half step is handed back."""
    full = onestep if onestep is not None else backwardeuleronestep(rhs, A, getb(t), u_k, dt, linearsolver, atol)
    half = 0.5 * dt
    firsthalf = backwardeuleronestep(rhs, A, getb(t), u_k, half, linearsolver, atol)
    secondhalf = backwardeuleronestep(rhs, A, getb(t + half), firsthalf, half, linearsolver, atol)
    mismatch = _norm_diff(full, secondhalf)
    if mismatch < atol:
        return secondhalf, dt, mismatch < atol / 4
    return firsthalf, half, False


def adaptivebackwardeulerstep(rhs, A, getb, u_k, t, dt, linearsolver, atol, callback):
    """transient.jl:89-121: try the requested dt; when it is rejected, cover it with accepted
    sub-steps — halve on rejection, double when the error is below atol/4, never overshoot —
    and reuse a rejected trial's half-step state as the next trial's full step."""
    callback(t, dt)
    state, taken, grow = backwardeulertwostep(rhs, A, getb, u_k, t, dt, linearsolver, atol)
    if not taken < dt:
        return state, taken, grow
    covered, base, want, reuse = 0.0, u_k, taken, True
    while covered < dt:
        callback(t, dt)
        state, taken, grow = backwardeulertwostep(rhs, A, getb, base, t + covered, want, linearsolver, atol, state if reuse else None)
        if taken == want:
            covered += taken
            base = state
            reuse = False
            if grow:
                want = 2 * taken
        elif taken < want:
            want = taken
            reuse = True
        else:
            raise FVError(_lib.FV_ERR_STATE, "Code is broken -- laststeptime should never be greater than targetdt")
        want = min(want, dt - covered)
    return state, taken, grow


def fixedbackwardeulerstep(rhs, A, getb, u_k, t, dt, linearsolver, atol, callback):
    """transient.jl:130-134"""
    callback(t, dt)
    u_new = backwardeuleronestep(rhs, A, getb(t), u_k, dt, linearsolver, atol)
    return u_new, dt, False


def _nocallback(t, dt):
    return None


def _integrate_generic(u0, A, b_or_getb, dt0, t0, tfinal, stepper=adaptivebackwardeulerstep, linearsolver=defaultlinearsolver, atol=1e-4, callback=_nocallback, _history="host"):
    """transient.jl:123-154.  Returns (us, ts); with a device operator `us` holds
    host copies of every stored step (as the reference keeps them) unless
    _history == "device" (DeviceVector handles, used by the high-level methods)."""
    if callable(b_or_getb):
        getb = b_or_getb
    else:
        bconst = b_or_getb
        getb = lambda t: bconst  # noqa: E731  (transient.jl:123-128)
    host_path = not isinstance(linearsolver, DevicePCG)
    if host_path:
        if isinstance(A, (DeviceOperator,)):
            raise TypeError("custom linearsolver needs a host matrix")
        A = A.copy()  # transient.jl:140
        u0 = af64(u0).copy()
        rhs = np.empty_like(u0)
        us = [u0]
    else:
        A = _as_operator(A)
        if isinstance(u0, DeviceVector):
            first = u0
        else:
            first = A.problem.new_state().set_free(af64(u0))
        rhs = None
        us = [first]
    # the outer loop of transient.jl:141-152: the recorded time advances by the REQUESTED step (also when the stepper
    # sub-stepped), the next request is what the stepper reports it took — doubled if it asks for it — clipped to tfinal
    now, request, ts = t0, min(dt0, tfinal - t0), [t0]
    while now < tfinal:
        state, taken, grow = stepper(rhs, A, getb, us[-1], now, request, linearsolver, atol, callback)
        now += request
        us.append(state)
        ts.append(now)
        request = min(tfinal - now, (2 if grow else 1) * taken)
    if not host_path and _history == "host":
        us = [u.free_values() for u in us]
    return us, ts


def _is_tspan(x):
    return isinstance(x, (tuple, list)) and len(x) == 2 and all(np.isscalar(v) for v in x)


def backwardeulerintegrate(u0, *args, **kwargs):
    """The four methods of transient.jl:123-174, dispatched on the argument shapes:

      backwardeulerintegrate(u0, A, b::Vector | getb::Function, dt0, t0, tfinal; stepper, linearsolver, atol, callback)
      backwardeulerintegrate(u0, tspan, Ss, volumes, neighbors, areasoverlengths, conductivities, sources,
                             dirichletnodes, dirichletheads, metaindex=None, logtransformconductivity=False; dt0=1.0, ...)
      backwardeulerintegrate(u0, tspan, getb::Function, Ss, volumes, neighbors, ... same ...)

    Extra keywords of this build: rtol/maxiter configure the device PCG when
    `linearsolver` is left at its default; keep="last" (high-level methods with constant b) runs the whole
    adaptive integration on the device (fv_transient_run_adaptive) and returns ([u0, u(tfinal)], ts) instead
    of every intermediate state; preconditioner="jacobi" | "amg" | "auto" (default: Jacobi-PCG until a step needs more
    than 50 iterations, then PCG with the AMG V-cycle — large time steps); reorder=True renumbers the mesh for
    locality on the way in (meshio.locality_order) and the states back on the way out."""
    if "stepper_" in kwargs:
        kwargs["stepper"] = kwargs.pop("stepper_")
    rtol = kwargs.pop("rtol", None)
    maxiter = kwargs.pop("maxiter", None)
    auto_solver = (rtol is not None or maxiter is not None) and "linearsolver" not in kwargs
    if auto_solver:
        kwargs["linearsolver"] = DevicePCG(rtol if rtol is not None else SQRT_EPS, maxiter if maxiter is not None else 1000)
    if not _is_tspan(args[0]):
        A, b_or_getb, dt0, t0, tfinal = args
        return _integrate_generic(u0, A, b_or_getb, dt0, t0, tfinal, **kwargs)
    tspan = args[0]
    rest = list(args[1:])
    getb = rest.pop(0) if callable(rest[0]) else None
    Ss, volumes, neighbors, aol, K, sources, dnodes, dheads = rest[:8]
    metaindex = rest[8] if len(rest) > 8 else None
    logt = rest[9] if len(rest) > 9 else False
    if kwargs.pop("reorder", False):
        # the device works on the mesh renumbered by meshio.locality_order; states come back in the caller's numbering
        from .core import renumbered_inputs

        if getb is not None:
            raise ValueError("reorder=True is for the constant-b method (a getb(t) returns vectors in the caller's free order)")
        order, rank, nb2, dn2 = renumbered_inputs(neighbors, len(sources), dnodes)
        us2, ts2 = backwardeulerintegrate(af64(u0)[order - 1], tspan, Ss, af64(volumes)[order - 1], nb2, aol, K, af64(sources)[order - 1], dn2, dheads, metaindex, logt, **kwargs)
        return [u[rank - 1] for u in us2], ts2
    dt0 = kwargs.pop("dt0", 1.0)
    keep = kwargs.pop("keep", "all")
    preconditioner = kwargs.pop("preconditioner", "auto")  # Jacobi-PCG; the AMG V-cycle once a step needs > 50 iterations
    u0 = af64(u0)
    # assembleA + assembleb + scalebyvolume! (transient.jl:157-169) — one device problem
    p = _assembled_problem(neighbors, aol, K, sources, dnodes, dheads, metaindex, logt)
    first = p.transient_begin(Ss, volumes, u0)  # u0[freenodes], transient.jl:170
    if preconditioner != "jacobi":
        p.set_preconditioner(preconditioner)
    if keep == "last":
        # the reference stores every outer step on the host (us); at 10^7-10^8 cells that is the cost of the run.
        # keep="last": the default stepper and solver, constant b, entirely on the device -> ([u0, u(tfinal)], ts)
        if auto_solver:
            kwargs.pop("linearsolver")
        if getb is not None or any(k in kwargs for k in ("stepper", "linearsolver", "callback")):
            raise ValueError('keep="last" runs the default adaptive stepper with the device PCG and a constant b')
        ts, _, info = p.run_adaptive(first, tspan[0], tspan[1], dt0=dt0, atol=kwargs.pop("atol", 1e-4), rtol=rtol if rtol is not None else SQRT_EPS, maxiter=maxiter if maxiter is not None else 1000)
        if kwargs:
            raise TypeError("unexpected keyword arguments %s" % sorted(kwargs))
        return [u0.copy(), first.node_values()], [float(t) for t in ts]
    if keep == "device":
        # every outer state, as the reference keeps them, but in HBM (fv_trajectory): the default stepper or the fixed one, the
        # device PCG, a constant b.  -> (DeviceStates, ts); getcontinuoussolution(us, ts) of it is what the device-resident adjoint
        # sweep (adjointintegrate with a bound forcing) and devicegradientintegral read.
        from .adjoint import DeviceStates

        if auto_solver:
            kwargs.pop("linearsolver")
        stepper = kwargs.pop("stepper", adaptivebackwardeulerstep)
        if getb is not None or any(k in kwargs for k in ("linearsolver", "callback")) or stepper not in (adaptivebackwardeulerstep, fixedbackwardeulerstep):
            raise ValueError('keep="device" runs the adaptive or the fixed stepper with the device PCG and a constant b')
        atol = kwargs.pop("atol", 1e-4)
        if kwargs:
            raise TypeError("unexpected keyword arguments %s" % sorted(kwargs))
        tr = p.new_trajectory()
        r, m = rtol if rtol is not None else SQRT_EPS, maxiter if maxiter is not None else 1000
        p.record(tr, tspan[0])
        try:
            if stepper is adaptivebackwardeulerstep:
                p.run_adaptive(first, tspan[0], tspan[1], dt0=dt0, atol=atol, rtol=r, maxiter=m)
            else:  # fixed: the outer loop of transient.jl:141-152 (the last step clipped to tfinal), one solve per step
                tr.push(first, tspan[0])
                now, request = tspan[0], min(dt0, tspan[1] - tspan[0])
                while now < tspan[1]:
                    p.record(tr, now)
                    it, info, _ = p.run_fixed(first, request, 1, rtol=r, maxiter=m)
                    now += request
                    request = min(tspan[1] - now, request)
        finally:
            p.record(None)
        return DeviceStates(tr), [float(x) for x in tr.ts]
    if keep != "all":
        raise ValueError('keep must be "all", "last" or "device"')
    op = DeviceOperator(p)
    # constant-b method: the assembled b stays on the device (bhat = None); getb method: host closure per step
    us, ts = _integrate_generic(first, op, getb if getb is not None else None, dt0, tspan[0], tspan[1], _history="device", **kwargs)
    us = [u.node_values() for u in us]  # freenodes2nodes, transient.jl:172
    return us, ts
