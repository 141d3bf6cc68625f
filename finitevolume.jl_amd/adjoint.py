"""Adjoint hooks of FiniteVolume.jl kept at the API level
(/root/reference/src/transient.jl:176-216).

The adjoint ODE  dγ/dt = Aᵀγ + [dg/du]ᵀ  is integrated by the SAME implicit stepper
as the forward problem with the transposed operator.  For the assembled operator,
transpose(D⁻¹A) = A D⁻¹, and with w = D⁻¹γ every step is again the SPD solve
(D/dt + A) w⁺ = rhs, so the device operator serves both directions
(fv_transient_step with FV_STEP_ADJOINT).  Interpolation and quadrature stay on the
host, as in the reference (Interpolations / QuadGK there, numpy / scipy here).
"""
import numpy as np

from .core import DeviceMatrix, _assembled_problem, af64
from .transient import DeviceOperator, _integrate_generic, backwardeulerintegrate


def getcontinuoussolution(us, ts, val=None):
    """transient.jl:176-186: piecewise-linear-in-time interpolant of the stored states.
    Returns uc(t) -> vector (Gridded(Linear())); with val=2 an itp(i, t) of the 2-D form."""
    ts = np.asarray(ts, dtype=np.float64)
    U = np.stack([np.asarray(u, dtype=np.float64) for u in us], axis=0)  # (nt, n)
    if np.any(np.diff(ts) <= 0):
        raise ValueError("knot-vectors must be unique and sorted in increasing order")

    def uc(t):
        if t < ts[0] or t > ts[-1]:
            raise IndexError("BoundsError: attempt to interpolate at t = %r outside [%r, %r]" % (t, ts[0], ts[-1]))
        k = int(np.searchsorted(ts, t, side="right")) - 1
        k = min(max(k, 0), len(ts) - 2)
        w = (t - ts[k]) / (ts[k + 1] - ts[k])
        return (1.0 - w) * U[k] + w * U[k + 1]

    if val == 2:
        return lambda i, t: uc(t)[int(i) - 1]
    return uc


def adjointintegrate(*args, **kwargs):
    """transient.jl:188-205, both methods:

      adjointintegrate(getdgdu::Function, tspan, Ss, volumes, neighbors, areasoverlengths, conductivities, sources,
                       dirichletnodes, dirichletheads, metaindex=None, logtransformconductivity=False; kwargs...)
      adjointintegrate(A, getdgdu, tspan; dt0=1.0, kwargs...)     (A = transpose of the scaled operator)

    Returns (lambdas, ts) in terms of λ (reversed in time), vectors over the free cells."""
    if callable(args[0]):
        getdgdu, tspan, Ss, volumes, neighbors, aol, K, sources, dnodes, dheads = args[:10]
        metaindex = args[10] if len(args) > 10 else None
        logt = args[11] if len(args) > 11 else False
        p = _assembled_problem(neighbors, aol, K, sources, dnodes, dheads, metaindex, logt)
        p.transient_begin(Ss, volumes, None)  # scalebyvolume!(A, Ss*volumes, ...), transient.jl:192
        return adjointintegrate(DeviceOperator(p, adjoint=True), getdgdu, tspan, **kwargs)
    A, getdgdu, tspan = args
    dt0 = kwargs.pop("dt0", 1.0)
    if isinstance(A, DeviceOperator):
        n = A.problem.n
    else:
        n = A.shape[1]
    gamma0 = np.zeros(n)
    T = tspan[1]
    gammas, tsgamma = backwardeulerintegrate(gamma0, A, lambda t: getdgdu(T - t), dt0, tspan[0], tspan[1], **kwargs)
    return list(reversed(gammas)), list(reversed([T - t for t in tsgamma]))


def transpose(A):
    """`transpose(A)` of the scaled operator (transient.jl:193) for objects of this build."""
    if isinstance(A, DeviceOperator):
        return A.transpose()
    if isinstance(A, DeviceMatrix) and getattr(A, "_scaled_operator", None) is not None:
        return A._scaled_operator.transpose()
    return np.asarray(A).T


def gradientintegrate(lambdac_or_lambda0, du0dp, dgdp, dfdp_or_integral, tspan, **kwargs):
    """transient.jl:207-216.  With callables (lambdac, dfdp) the integral of dfdp(t)*λ(t) is done
    here by adaptive Gauss-Kronrod (scipy quad_vec, standing in for QuadGK); with vectors
    (lambda0, integrateddfdplambda) it is taken as given."""
    from scipy.integrate import quad_vec

    limit = int(kwargs.get("maxevals", 10**7) // 21 + 1) if "maxevals" in kwargs else 10000
    if callable(lambdac_or_lambda0):
        lambdac, dfdp = lambdac_or_lambda0, dfdp_or_integral
        I2, _ = quad_vec(lambda t: np.asarray(dfdp(t)) @ np.asarray(lambdac(t)), tspan[0], tspan[1], limit=limit)
        lambda0 = np.asarray(lambdac(0))
    else:
        lambda0, I2 = af64(lambdac_or_lambda0), af64(dfdp_or_integral)
    I1, _ = quad_vec(lambda t: np.asarray(dgdp(t), dtype=np.float64), tspan[0], tspan[1], limit=limit)
    return du0dp @ lambda0 + I1 + I2  # du0dp may be a dense array or a scipy sparse matrix (spzeros in the reference)
