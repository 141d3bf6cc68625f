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

from .core import SQRT_EPS as SQRT_EPS_
from .core import DeviceMatrix, Observation, _assembled_problem, af64
from .transient import DeviceOperator, _integrate_generic, backwardeulerintegrate


class LinearInterpolant:
    """uc(t) -> vector: the Gridded(Linear()) interpolant of transient.jl:176-180.  The knots (`ts`) and the stored
    states (`U`, one row per knot) stay readable so quadratures of products can be done exactly."""

    def __init__(self, us, ts):
        self.ts = np.asarray(ts, dtype=np.float64)
        self.U = np.stack([np.asarray(u, dtype=np.float64) for u in us], axis=0)  # (nt, n)
        if np.any(np.diff(self.ts) <= 0):
            raise ValueError("knot-vectors must be unique and sorted in increasing order")

    def __call__(self, t):
        ts = self.ts
        if t < ts[0] or t > ts[-1]:
            raise IndexError("BoundsError: attempt to interpolate at t = %r outside [%r, %r]" % (t, ts[0], ts[-1]))
        k = int(np.searchsorted(ts, t, side="right")) - 1
        k = min(max(k, 0), len(ts) - 2)
        w = (t - ts[k]) / (ts[k + 1] - ts[k])
        return (1.0 - w) * self.U[k] + w * self.U[k + 1]

    def at(self, tq):
        """all components at the times tq -> (len(tq), n)"""
        tq = np.asarray(tq, dtype=np.float64)
        k = np.clip(np.searchsorted(self.ts, tq, side="right") - 1, 0, len(self.ts) - 2)
        w = ((tq - self.ts[k]) / (self.ts[k + 1] - self.ts[k]))[:, None]
        return (1.0 - w) * self.U[k] + w * self.U[k + 1]


class LinearInterpolant2(LinearInterpolant):
    """itp(i, t): the (NoInterp(), Gridded(Linear())) form of transient.jl:182-186 (i is 1-based)."""

    def __call__(self, i, t):
        return LinearInterpolant.__call__(self, t)[int(i) - 1]


class DeviceStates:
    """`us` of a run kept in HBM (backwardeulerintegrate(..., keep="device"), the lambdas of a device adjoint sweep): indexing
    downloads one stored state — node vectors for a forward run (after freenodes2nodes, transient.jl:172), free-cell vectors for
    lambdas (`free=True`)."""

    def __init__(self, trajectory, free=False):
        self.trajectory, self.free = trajectory, free

    def __len__(self):
        return len(self.trajectory)

    def __getitem__(self, k):
        if isinstance(k, slice):
            return [self[i] for i in range(*k.indices(len(self)))]
        k = k + len(self) if k < 0 else k
        return self.trajectory.free_values(k) if self.free else self.trajectory.node_values(k)

    def __iter__(self):
        return (self[k] for k in range(len(self)))


class DeviceSolution:
    """getcontinuoussolution of DeviceStates: u_c(t) interpolated on the device and downloaded when called (node vector, or the
    free-cell vector for lambdas); the adjoint sweep and the gradient integral read the trajectory itself."""

    def __init__(self, states, ts, two=False):
        self.states, self.trajectory, self.two = states, states.trajectory, two
        self.ts = np.asarray(ts, dtype=np.float64)

    def __call__(self, *args):
        t = args[-1]
        free = self.trajectory.at(t)
        v = free if self.states.free else self.trajectory.problem.freenodes2nodes(free)
        return v[int(args[0]) - 1] if self.two else v


def getcontinuoussolution(us, ts, val=None):
    """transient.jl:176-186: piecewise-linear-in-time interpolant of the stored states.
    Returns uc(t) -> vector (Gridded(Linear())); with val=2 an itp(i, t) of the 2-D form."""
    if isinstance(us, DeviceStates):
        return DeviceSolution(us, ts, two=(val == 2))
    return LinearInterpolant2(us, ts) if val == 2 else LinearInterpolant(us, ts)


def adjointintegrate(*args, **kwargs):
    """transient.jl:188-205, both methods:

      adjointintegrate(getdgdu::Function, tspan, Ss, volumes, neighbors, areasoverlengths, conductivities, sources,
                       dirichletnodes, dirichletheads, metaindex=None, logtransformconductivity=False; kwargs...)
      adjointintegrate(A, getdgdu, tspan; dt0=1.0, kwargs...)     (A = transpose of the scaled operator)

    Returns (lambdas, ts) in terms of λ (reversed in time), vectors over the free cells."""
    if isinstance(args[0], BoundForcing) and isinstance(args[0].uc, DeviceSolution):
        # device-resident sweep (fv_adjoint_run): the forward states stay in HBM, the forcing of every solve is a kernel over the
        # observation rows.  The operator is the forward run's own problem (same mesh, parameters and storage term), as
        # adjointintegrate would assemble it again from the same arguments (transient.jl:189-193).
        from .transient import adaptivebackwardeulerstep, fixedbackwardeulerstep

        forcing, tspan = args[0], args[1]
        uc = forcing.uc
        p = uc.trajectory.problem
        # the reference re-assembles A from these arguments (transient.jl:189-193); here the sweep runs on the forward run's problem, so they
        # must be what that problem was built from (fingerprints recorded by Problem.create / assemble / transient_begin)
        names = ("Ss", "volumes", "neighbors", "areasoverlengths", "conductivities", "sources", "dirichletnodes", "dirichletheads", "metaindex", "logtransformconductivity")
        p.check_inputs(**dict(zip(names, args[2:])))
        stepper = kwargs.pop("stepper", kwargs.pop("stepper_", adaptivebackwardeulerstep))
        if stepper not in (adaptivebackwardeulerstep, fixedbackwardeulerstep) or any(k in kwargs for k in ("linearsolver", "callback")):
            raise ValueError("the device-resident adjoint sweep runs the adaptive or the fixed stepper with the device PCG")
        obs = forcing.observation(p)
        lam, _, _, _ = p.adjoint_run(uc.trajectory, obs, tspan[0], tspan[1], dt0=kwargs.pop("dt0", 1.0), adaptive=stepper is adaptivebackwardeulerstep,
                                     atol=kwargs.pop("atol", 1e-4), rtol=kwargs.pop("rtol", SQRT_EPS_), maxiter=kwargs.pop("maxiter", 1000))
        if kwargs:
            raise TypeError("unexpected keyword arguments %s" % sorted(kwargs))
        return DeviceStates(lam, free=True), [float(t) for t in lam.ts]
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

        def dfdp_lambda(t):
            M = dfdp(t)
            if not hasattr(M, "tocsr") and not isinstance(M, DeviceJacobian):  # dense array; scipy sparse matrices and device Jacobians multiply as they are
                M = np.asarray(M)
            return np.asarray(M @ np.asarray(lambdac(t))).ravel()

        I2, _ = quad_vec(dfdp_lambda, tspan[0], tspan[1], limit=limit)
        lambda0 = np.asarray(lambdac(0))
    else:
        lambda0, I2 = af64(lambdac_or_lambda0), af64(dfdp_or_integral)
    I1, _ = quad_vec(lambda t: np.asarray(dgdp(t), dtype=np.float64), tspan[0], tspan[1], limit=limit)
    return du0dp @ lambda0 + I1 + I2  # du0dp may be a dense array or a scipy sparse matrix (spzeros in the reference)


# ------------------------------------------------------------------ src/transientadjointutils.jl (host-side glue, API kept)
def _parameter_jacobians(ueval_free, neighbors, areasoverlengths, conductivities, sources, dirichletnodes, dirichletheads, metaindex, logtransformconductivity):
    """b_p - A_px as an (np x nfree) scipy CSR matrix, p = [conductivities; sources; dirichletheads].

    Stands in for the LinearAdjoints-generated assembleb_p / assembleA_px called at
    transientadjointutils.jl:27-28 (that package is not in the reference tree; the derivative of
    the assembly at FiniteVolume.jl:75-139 is written out here): per face i with conductance
    c_i = K[m]*aol_i or exp(K[m])*aol_i,
      free-free:        d(Ax)_f1/dK_m = dc (x_f1 - x_f2),  d(Ax)_f2/dK_m = dc (x_f2 - x_f1)
      one free end f:   d(Ax)_f/dK_m  = dc x_f ;  db_f/dK_m = dc*dhead ;  db_f/ddhead_j = c_i
      db_f/dsource_node(f) = 1."""
    import scipy.sparse as sp

    from .core import _metaindex_array, _split_neighbors, getfreenodes, getnodei2dirichleti

    n1, n2 = _split_neighbors(neighbors)
    aol, K, dh = af64(areasoverlengths), af64(conductivities), af64(dirichletheads)
    N, F, nK, ndir = len(sources), len(n1), len(K), len(dh)
    freenode, n2f = getfreenodes(N, dirichletnodes)
    n2d = getnodei2dirichleti(np.zeros(N), dirichletnodes)
    nfree = int(freenode.sum())
    mi = _metaindex_array(metaindex, F)
    m = (mi - 1) if mi is not None else np.arange(F)
    c = np.exp(K[m]) * aol if logtransformconductivity else K[m] * aol
    dc = c if logtransformconductivity else aol
    a, b = n1 - 1, n2 - 1
    fa, fb = n2f[a] - 1, n2f[b] - 1  # -2 where Dirichlet
    x = af64(ueval_free)
    rows, cols, vals = [], [], []

    def add(r, cidx, v):
        rows.append(r)
        cols.append(cidx)
        vals.append(v)

    both = freenode[a] & freenode[b]
    add(m[both], fa[both], -dc[both] * (x[fa[both]] - x[fb[both]]))
    add(m[both], fb[both], -dc[both] * (x[fb[both]] - x[fa[both]]))
    onlya = freenode[a] & ~freenode[b]
    add(m[onlya], fa[onlya], dc[onlya] * dh[n2d[b[onlya]] - 1] - dc[onlya] * x[fa[onlya]])
    add(nK + N + n2d[b[onlya]] - 1, fa[onlya], c[onlya])
    onlyb = ~freenode[a] & freenode[b]
    add(m[onlyb], fb[onlyb], dc[onlyb] * dh[n2d[a[onlyb]] - 1] - dc[onlyb] * x[fb[onlyb]])
    add(nK + N + n2d[a[onlyb]] - 1, fb[onlyb], c[onlyb])
    freenodes_idx = np.nonzero(freenode)[0]
    add(nK + freenodes_idx, n2f[freenodes_idx] - 1, np.ones(nfree))
    M = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(nK + N + ndir, nfree)).tocsr()
    return M, freenode, n2f


class DeviceJacobian:
    """dfdp(u, t, p) of transientadjointutils.jl:23-30 — the (np x nfree) matrix (b_p - A_px)' scaled by the storage term — held
    as what it does: `J @ lam` is computed on the device (fv_param_jacobian_apply: one thread per face) and summed over the faces
    of each parameter on the host, without forming the matrix.  What gradientintegrate needs of it (`dfdp(t) * lambdac(t)`,
    transient.jl:208-219)."""

    def __init__(self, problem, ueval_free, maps, scale, logtransformconductivity):
        self.problem, self.ueval, self.maps, self.scale, self.log = problem, ueval_free, maps, scale, logtransformconductivity
        self.shape = (maps["np"], len(ueval_free))

    def __matmul__(self, lam):
        mp = self.maps
        w = af64(lam) * self.scale  # the reference's scaling by the FREE index (see getadjointfunctions)
        face_k, face_dir, row_src = self.problem.param_jacobian_apply(self.ueval, w, False, self.log)
        out = np.zeros(mp["np"])
        out[: mp["nK"]] = np.bincount(mp["m"], weights=face_k, minlength=mp["nK"])
        out[mp["nK"] + mp["free_nodes0"]] = row_src
        for sel, pos in mp["dir_terms"]:
            out[mp["nK"] + mp["N"] :] += np.bincount(pos, weights=face_dir[sel], minlength=mp["ndir"])
        return out

    dot = __matmul__


class BoundForcing:
    """t -> dgdu(uc, t): what the reference's callers write as a closure (`t->dgdu(uc_p, t)`, examples/transientadjoint/ex.jl:118),
    kept as an object so that adjointintegrate can see the solution and the observation series behind it."""

    def __init__(self, forcing, uc):
        self.forcing, self.uc = forcing, uc

    def __call__(self, t):
        return self.forcing(self.uc, t)

    def observation(self, problem):
        return self.forcing.observation(problem)


class ObservationForcing:
    """dgdu of getadjointfunctions (transientadjointutils.jl:13-21): callable as dgdu(u, t) like the reference's closure; `bind(uc)`
    gives the t -> dgdu(uc, t) object the device-resident sweep recognises."""

    def __init__(self, host, sigma, obsfreenodes, uobs, f2n):
        self.host, self.sigma, self.obs, self.uobs, self.f2n = host, sigma, [int(i) for i in obsfreenodes], uobs, f2n
        self._dev = {}

    def __call__(self, u, t):
        return self.host(u, t)

    def bind(self, uc):
        return BoundForcing(self, uc)

    def observation(self, problem):
        """The device form of the observation data: uobs at the observation rows and sigma(i, t) sampled at uobs's knots (sigma is
        taken as linear in t between them — exact for the constant-in-time weights of the reference's example)."""
        key = id(problem)
        if key not in self._dev:
            if not isinstance(self.uobs, LinearInterpolant):
                raise TypeError("the device-resident adjoint needs uobs as a piecewise-linear solution object (getcontinuoussolution)")
            nodes = self.f2n[np.asarray(self.obs, dtype=np.int64) - 1] - 1
            tk = self.uobs.ts
            S = np.array([[self.sigma(i, float(t)) for i in self.obs] for t in tk], dtype=np.float64).reshape(len(tk), len(self.obs))
            self._dev = {key: Observation(problem, self.obs, tk, self.uobs.U[:, nodes], S)}
        return self._dev[key]


def getadjointfunctions(sigma, obsfreenodes, uobs, u0, tspan, Ss, volumes, neighbors, areasoverlengths, conductivities, sources, dirichletnodes, dirichletheads, metaindex=None, logtransformconductivity=False, **kwargs):
    """transientadjointutils.jl:1-55 -> g, dgdu, dfdp, dgdp, du0dp, G.  device=True (keyword): dfdp(u, t, p) returns a
    DeviceJacobian — its product with lambda runs on the GPU — instead of a scipy matrix."""
    on_device = bool(kwargs.pop("device", False))
    import scipy.sparse as sp
    from scipy.integrate import quad

    from .core import getfreenodes

    nK, N, ndir = len(conductivities), len(sources), len(dirichletheads)
    freenodes, n2f = getfreenodes(len(u0), dirichletnodes)
    f2n = np.empty(int(freenodes.sum()), np.int64)  # free index (1-based) -> node (1-based), the Dict of :3
    f2n[n2f[freenodes] - 1] = np.nonzero(freenodes)[0] + 1
    nfree = len(f2n)
    vols = Ss * af64(volumes)

    def g(u, t):
        uo, ue = uobs(t), u(t)
        return float(sum(sigma(i, t) ** 2 * (ue[f2n[i - 1] - 1] - uo[f2n[i - 1] - 1]) ** 2 for i in obsfreenodes))

    def dgdu_host(u, t):
        uo, ue = uobs(t), u(t)
        result = np.zeros(nfree)
        for i in obsfreenodes:
            result[i - 1] = 2 * sigma(i, t) ** 2 * (ue[f2n[i - 1] - 1] - uo[f2n[i - 1] - 1])
        return result

    dgdu = ObservationForcing(dgdu_host, sigma, obsfreenodes, uobs, f2n)

    def split(p):
        p = af64(p)
        return p[:nK], p[nK : nK + N], p[nK + N : nK + N + ndir]

    device_state = {}

    def dfdp_device(u, t, p):
        from .core import Problem, _metaindex_array, _split_neighbors, getnodei2dirichleti

        pK, ps, pd = split(p)
        key = (pK.tobytes(), pd.tobytes())
        if device_state.get("key") != key:  # one problem per parameter vector: the conductances and heads live on the device
            if "problem" in device_state:
                device_state["problem"].close()
            n1, n2 = _split_neighbors(neighbors)
            F = len(n1)
            mi = _metaindex_array(metaindex, F)
            n2d = getnodei2dirichleti(np.zeros(N), dirichletnodes)
            a, b = n1 - 1, n2 - 1
            dir_terms = []
            for fr, di in ((a, b), (b, a)):
                sel = freenodes[fr] & ~freenodes[di]
                dir_terms.append((sel, n2d[di[sel]] - 1))
            device_state.update(key=key, problem=Problem.create(neighbors, areasoverlengths, N, dirichletnodes).assemble(pK, np.zeros(N), pd, metaindex, logtransformconductivity),
                                maps=dict(np=nK + N + ndir, nK=nK, N=N, ndir=ndir, m=(mi - 1) if mi is not None else np.arange(F),
                                          free_nodes0=np.nonzero(freenodes)[0], dir_terms=dir_terms))
        ueval = np.ascontiguousarray(np.asarray(u(t))[freenodes])
        return DeviceJacobian(device_state["problem"], ueval, device_state["maps"], 1.0 / vols[:nfree], logtransformconductivity)

    def dfdp(u, t, p):
        if on_device:
            return dfdp_device(u, t, p)
        pK, ps, pd = split(p)
        ueval = np.asarray(u(t))[freenodes]
        M, _, _ = _parameter_jacobians(ueval, neighbors, areasoverlengths, pK, ps, dirichletnodes, pd, metaindex, logtransformconductivity)
        # scalebyvolume!(transpose(...), Ss*volumes, ...) of transient.jl:25-34 divides the entries of free unknown i by
        # volumes[i] — indexed by the FREE index, without the free->node map (the reference's behaviour, kept as is)
        return M @ sp.diags(1.0 / vols[:nfree])

    dgdpval = np.zeros(nK + N + ndir)

    def dgdp(u, t, p):
        return dgdpval

    du0dp = sp.csr_matrix((nK + N + ndir, nfree))

    def G(p):
        limit = max(300 // 21, 50)  # quadgk(...; maxevals=3*10^2, order=21) in the reference
        if isinstance(p, DeviceSolution):  # the states are in HBM: the exact piecewise integral on the device (fv_observation_integral)
            return dgdu.observation(p.trajectory.problem).integral(p.trajectory, tspan[0], tspan[1])
        if callable(p):
            # the integrand is a polynomial between the knots of the two interpolants: integrate piece by piece
            knots = {float(tspan[0]), float(tspan[1])}
            for f in (p, uobs):
                knots |= {float(t) for t in getattr(f, "ts", ()) if tspan[0] < t < tspan[1]}
            cuts = sorted(knots)
            if len(cuts) > 2:  # knots known: a 6-point Gauss-Legendre rule per piece is exact for these polynomials
                xs, ws = np.polynomial.legendre.leggauss(6)
                return float(sum(0.5 * (b - a) * sum(w * g(p, 0.5 * (a + b) + 0.5 * (b - a) * x) for x, w in zip(xs, ws)) for a, b in zip(cuts[:-1], cuts[1:])))
            return quad(lambda t: g(p, t), tspan[0], tspan[1], limit=limit)[0]  # opaque callables: adaptive Gauss-Kronrod, as QuadGK
        pK, ps, pd = split(p)
        us_p, ts_p = backwardeulerintegrate(u0, tspan, Ss, volumes, neighbors, areasoverlengths, pK, ps, dirichletnodes, pd, metaindex, logtransformconductivity, **kwargs)
        return G(getcontinuoussolution(us_p, ts_p))

    return g, dgdu, dfdp, dgdp, du0dp, G


def devicegradientintegral(uc, lambdas, ts_lambda, tspan, Ss, volumes, neighbors, areasoverlengths, conductivities, sources, dirichletnodes, dirichletheads, metaindex=None, logtransformconductivity=False, scale="reference", ctx=None):
    """The integral over tspan of dfdp(uc, t, p) * lambda(t) — what gradientintegrate(lambdac, du0dp, dgdp, dfdp, tspan)
    adds to the gradient (transient.jl:208-219 with the dfdp of transientadjointutils.jl:23-30) — computed by
    fv_param_gradient_integral: one thread per face walks the merged knots of uc (a LinearInterpolant of node vectors)
    and lambda (free-indexed, piecewise linear over ts_lambda); exact, no quadrature.

    scale: "reference" divides lambda_f by (Ss*volumes)[f] with the FREE index f, as the reference's scalebyvolume!
    call does (see getadjointfunctions); "storage" by the volume of the node behind f; None not at all."""
    from .core import Problem, _metaindex_array, _split_neighbors, getfreenodes, getnodei2dirichleti

    if isinstance(uc, DeviceSolution):
        # u and lambda both in HBM: no state crosses PCIe (fv_param_gradient_integral_traj); lambdas: DeviceStates of the sweep
        if not isinstance(lambdas, DeviceStates):
            raise TypeError("with a device solution the lambdas must be the DeviceStates of the device adjoint sweep")
        p = uc.trajectory.problem
        K, dh = af64(conductivities), af64(dirichletheads)
        nK, N, ndir = len(K), len(sources), len(dh)
        n1, n2 = _split_neighbors(neighbors)
        F = len(n1)
        freenode, n2f = getfreenodes(N, dirichletnodes)
        nfree = int(freenode.sum())
        vols = Ss * af64(volumes)
        if scale == "reference":
            fk, fd, rs = p.param_gradient_integral_traj(uc.trajectory, lambdas.trajectory, tspan[0], tspan[1], False, 1.0 / vols[:nfree], logtransformconductivity)
        elif scale == "storage":
            fk, fd, rs = p.param_gradient_integral_traj(uc.trajectory, lambdas.trajectory, tspan[0], tspan[1], True, None, logtransformconductivity)
        elif scale is None:
            fk, fd, rs = p.param_gradient_integral_traj(uc.trajectory, lambdas.trajectory, tspan[0], tspan[1], False, None, logtransformconductivity)
        else:
            raise ValueError("scale must be 'reference', 'storage' or None")
        mi = _metaindex_array(metaindex, F)
        m = (mi - 1) if mi is not None else np.arange(F)
        out = np.zeros(nK + N + ndir)
        out[:nK] = np.bincount(m, weights=fk, minlength=nK)
        out[nK + np.nonzero(freenode)[0]] = rs
        n2d = getnodei2dirichleti(np.zeros(N), dirichletnodes)
        a, b = n1 - 1, n2 - 1
        for fr, di in ((a, b), (b, a)):
            sel = freenode[fr] & ~freenode[di]
            out[nK + N :] += np.bincount(n2d[di[sel]] - 1, weights=fd[sel], minlength=ndir)
        return out
    if not isinstance(uc, LinearInterpolant):
        raise TypeError("devicegradientintegral needs the piecewise-linear solution object (getcontinuoussolution)")
    K, dh = af64(conductivities), af64(dirichletheads)
    nK, N, ndir = len(K), len(sources), len(dh)
    n1, n2 = _split_neighbors(neighbors)
    F = len(n1)
    freenode, n2f = getfreenodes(N, dirichletnodes)
    nfree = int(freenode.sum())
    lam = LinearInterpolant(lambdas, ts_lambda)
    lo, hi = float(tspan[0]), float(tspan[1])
    knots = np.unique(np.concatenate([lam.ts, uc.ts, [lo, hi]]))
    knots = knots[(knots >= lo) & (knots <= hi)]
    if len(knots) < 2:
        return np.zeros(nK + N + ndir)
    U = uc.at(knots)[:, freenode]
    L = lam.at(knots)
    vols = Ss * af64(volumes)
    if scale == "reference":
        L = L / vols[:nfree]
    elif scale == "storage":
        L = L / vols[freenode]
    elif scale is not None:
        raise ValueError("scale must be 'reference', 'storage' or None")
    # (the Jacobian does not depend on the source values: none are handed over, so none can sit on a Dirichlet node)
    p = Problem.create(neighbors, areasoverlengths, N, dirichletnodes, ctx).assemble(K, np.zeros(N), dh, metaindex, logtransformconductivity)
    try:
        face_k, face_dir, row_src = p.param_gradient_integral(knots, U, L, False, logtransformconductivity)
    finally:
        p.close()
    mi = _metaindex_array(metaindex, F)
    m = (mi - 1) if mi is not None else np.arange(F)
    out = np.zeros(nK + N + ndir)
    out[:nK] = np.bincount(m, weights=face_k, minlength=nK)
    out[nK + np.nonzero(freenode)[0]] = row_src
    n2d = getnodei2dirichleti(np.zeros(N), dirichletnodes)
    a, b = n1 - 1, n2 - 1
    for fr, di in ((a, b), (b, a)):  # faces with exactly one free end: their term goes to the head of the other end
        sel = freenode[fr] & ~freenode[di]
        out[nK + N :] += np.bincount(n2d[di[sel]] - 1, weights=face_dir[sel], minlength=ndir)
    return out


class FVErrorNotSupported(Exception):
    """error("not supported"), FiniteVolume.jl:363"""


def _simpleintegrate(fs, ts):
    """FiniteVolume.jl:262-269 — the trapezoid rule over the stored knots."""
    fs, ts = np.asarray(fs, dtype=np.float64), np.asarray(ts, dtype=np.float64)
    w = np.empty(len(ts))
    w[0], w[-1] = 0.5 * (ts[1] - ts[0]), 0.5 * (ts[-1] - ts[-2])
    w[1:-1] = 0.5 * (ts[2:] - ts[:-2])
    return w @ fs


def _product_integrals(lam, u2, nodes, freeidx, tspan):
    """The integral over tspan of lambda_f(t) * u_node(t) for each (node, free index) pair — integrateproduct of
    FiniteVolume.jl:277-285 (QuadGK there).  Both factors are piecewise linear, so with the knots of both in hand
    Simpson's rule on the merged grid is exact; an opaque u2(i, t) is integrated by Gauss-Kronrod between lambda's knots."""
    lo, hi = float(tspan[0]), float(tspan[1])
    if isinstance(u2, LinearInterpolant):
        knots = np.unique(np.concatenate([lam.ts, u2.ts, [lo, hi]]))
        knots = knots[(knots >= lo) & (knots <= hi)]
        a, b = knots[:-1], knots[1:]
        mid = 0.5 * (a + b)
        out = np.zeros(len(nodes))
        La, Lm, Lb = lam.at(a)[:, freeidx], lam.at(mid)[:, freeidx], lam.at(b)[:, freeidx]
        Ua, Um, Ub = u2.at(a)[:, nodes], u2.at(mid)[:, nodes], u2.at(b)[:, nodes]
        out = ((b - a) / 6.0) @ (La * Ua + 4.0 * Lm * Um + Lb * Ub)
        return out
    from scipy.integrate import quad

    knots = [lo] + [float(t) for t in lam.ts if lo < t < hi] + [hi]
    return np.array([sum(quad(lambda t: lam(t)[f] * u2(n + 1, t), a, b)[0] for a, b in zip(knots[:-1], knots[1:])) for n, f in zip(nodes, freeidx)])


def integratedfdplambda(u2, p, lambdas, ts_lambda, tspan, Ss, volumes, neighbors, areasoverlengths, conductivities, sources, dirichletnodes, dirichletheads, metaindex=None, logtransformconductivity=False, complete=False, device=True):
    """transientadjointutils.jl:57-63 -> FiniteVolume.jl:271-377 (integrateb_pmA_pxlambda): the integral over tspan of
    dfdp(t) * lambda(t) with lambda piecewise linear over ts_lambda and u2 = getcontinuoussolution(us, ts, 2).

    The default follows the reference's hand-unrolled routine term by term, i.e. exactly the terms it carries:
      * sources:          trapz(lambda_f) / (Ss*volumes[node])                               (:327-336)
      * free|Dirichlet faces only: c*dhead*trapz(lambda_f)/(Ss*volumes[f]) into K and c*trapz(lambda_f)/(Ss*volumes[f])
        into the Dirichlet head, plus  + c * integral(lambda_f * u_node) / (Ss*volumes[f])   into K   (:339-360)
        with volumes indexed by the FREE index f there, and the u term entering with a plus sign;
      * no free|free face terms; logtransformconductivity=false raises "not supported"       (:362-363).
    complete=True instead integrates the full Jacobian (b_p - A_px, every face, scaled by D^-1 = 1 / (Ss * volume of
    the node behind each free unknown)) — the quantity gradientintegrate(lambdac, ..., dfdp) integrates: on the device, exactly, when
    u2 is the piecewise-linear solution object (devicegradientintegral); device=False or an opaque u2(i, t) falls back
    to Gauss-Kronrod between lambda's knots with the Jacobian rebuilt on the host at every evaluation."""
    from .core import _metaindex_array, _split_neighbors, getfreenodes, getnodei2dirichleti

    if not logtransformconductivity:
        raise FVErrorNotSupported("not supported")
    nK, N, ndir = len(conductivities), len(sources), len(dirichletheads)
    pv = af64(p)
    pK, ps, pd = pv[:nK], pv[nK : nK + N], pv[nK + N : nK + N + ndir]
    vols = Ss * af64(volumes)
    freenode, n2f = getfreenodes(N, dirichletnodes)
    if complete and device and isinstance(u2, DeviceSolution):
        # u and lambda in HBM (keep="device" + the device-resident sweep): nothing crosses PCIe but the per-face results
        return devicegradientintegral(u2, lambdas, ts_lambda, tspan, Ss, volumes, neighbors, areasoverlengths, pK, ps, dirichletnodes, pd, metaindex, True, scale="storage")
    if isinstance(u2, DeviceSolution) or isinstance(lambdas, DeviceStates):
        raise TypeError("device solutions / lambdas go through complete=True (the whole Jacobian integrated on the device)")
    lam = LinearInterpolant(lambdas, ts_lambda)  # (host copies of every lambda state: only on the paths that need them)
    if complete and device and isinstance(u2, LinearInterpolant):
        # both factors piecewise linear: the device kernel integrates every face exactly (fv_param_gradient_integral)
        return devicegradientintegral(u2, lambdas, ts_lambda, tspan, Ss, volumes, neighbors, areasoverlengths, pK, ps, dirichletnodes, pd, metaindex, True, scale="storage")
    if complete:
        import inspect

        import scipy.sparse as sp
        from scipy.integrate import quad_vec

        if isinstance(u2, LinearInterpolant):
            uc = lambda t: LinearInterpolant.__call__(u2, t)  # noqa: E731
        elif len(inspect.signature(u2).parameters) == 1:
            uc = u2
        else:
            uc = lambda t: np.array([u2(i, t) for i in range(1, N + 1)])  # noqa: E731

        def integrand(t):
            M, _, _ = _parameter_jacobians(np.asarray(uc(t))[freenode], neighbors, areasoverlengths, pK, ps, dirichletnodes, pd, metaindex, True)
            return (M @ sp.diags(1.0 / vols[freenode])) @ lam(t)

        total = np.zeros(nK + N + ndir)
        knots = sorted(set(float(t) for t in ts_lambda if tspan[0] <= t <= tspan[1]) | {float(tspan[0]), float(tspan[1])})
        for a, b in zip(knots[:-1], knots[1:]):
            total += quad_vec(integrand, a, b, limit=20)[0]
        return total

    n1, n2 = _split_neighbors(neighbors)
    aol = af64(areasoverlengths)
    F = len(n1)
    n2d = getnodei2dirichleti(np.zeros(N), dirichletnodes)
    mi = _metaindex_array(metaindex, F)
    m = (mi - 1) if mi is not None else np.arange(F)
    c = np.exp(pK[m]) * aol
    lamint = _simpleintegrate(lambdas, ts_lambda)
    result = np.zeros(nK + N + ndir)
    free_idx = np.nonzero(freenode)[0]
    result[nK + free_idx] += lamint[n2f[free_idx] - 1] / vols[free_idx]
    a, b = n1 - 1, n2 - 1
    for fr, di in ((a, b), (b, a)):  # (free end, Dirichlet end) of the faces with exactly one free end
        sel = freenode[fr] & ~freenode[di]
        f = n2f[fr[sel]] - 1
        d = n2d[di[sel]] - 1
        cs, msel = c[sel], m[sel]
        nodes = fr[sel]
        un, inv = np.unique(nodes, return_inverse=True)  # the reference memoises the product integral per node
        prod = _product_integrals(lam, u2, un, n2f[un] - 1, tspan)[inv] if len(un) else np.zeros(0)
        scale = 1.0 / vols[f]  # volumes[nodei2freenodei[node]]: the free index, as the reference writes it
        np.add.at(result, msel, cs * pd[d] * lamint[f] * scale + cs * prod * scale)
        np.add.at(result, nK + N + d, cs * lamint[f] * scale)
    return result
