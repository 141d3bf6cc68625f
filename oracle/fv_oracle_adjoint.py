"""CPU restatement of the reference's adjoint / gradient routines — TEST INFRASTRUCTURE (the checker), never imported by
the product.  Plain loops, one face / one node at a time, in the reference's order and with its names.

Follows, under /root/reference:
  src/FiniteVolume.jl:262-269     simpleintegrate
  src/FiniteVolume.jl:271-377     integrateb_pmA_pxlambda (the hand-unrolled integral of (b_p - A_px)' lambda; log conductivity only)
  src/transient.jl:25-34          scalebyvolume!(::Transpose, ...)  (divides parent column i by volumes[i]: the FREE index)
  src/transient.jl:176-186        getcontinuoussolution (both forms)
  src/transient.jl:188-205        adjointintegrate (both methods)
  src/transient.jl:207-216        gradientintegrate (both methods)
  src/transientadjointutils.jl:1-55   getadjointfunctions -> g, dgdu, dfdp, dgdp, du0dp, G
  src/transientadjointutils.jl:57-63  integratedfdplambda

Third-party pieces that are NOT in the reference tree, restated from their published behaviour:
  * Interpolations.jl Gridded(Linear()) — piecewise-linear interpolation between the knots, error outside them;
  * QuadGK.quadgk — adaptive Gauss-Kronrod; here scipy.integrate.quad / quad_vec (QUADPACK's adaptive Gauss-Kronrod),
    split at the interpolants' knots where the integrand has kinks;
  * LinearAdjoints 0.1.0 (un-vendored dev checkout, Manifest.toml:183-187) — the macro-generated derivative twins
    assembleb_p / assembleA_px of the primal loops at FiniteVolume.jl:75-139: written out here by differentiating those
    loops entry by entry (parity of these two is pinned only by the finite-difference checks of test/onenodeadjoint.jl:66-75).
Pinned by tests/test_oracle_adjoint_kats.py: test/odeadjoint.jl:34-41, test/onenodeadjoint.jl:45-75, test/theisadjoint.jl:55-84
at the reference's tolerances."""
import bisect
import math

import numpy as np

from . import fv_oracle as o


# ------------------------------------------------------------------ src/FiniteVolume.jl:262-269
def simpleintegrate(fs, ts):
    result = 0.5 * ((ts[1] - ts[0]) * np.asarray(fs[0], float) + (ts[-1] - ts[-2]) * np.asarray(fs[-1], float))
    for i in range(1, len(ts) - 1):
        result = result + 0.5 * (ts[i + 1] - ts[i - 1]) * np.asarray(fs[i], float)
    return result


# ------------------------------------------------------------------ src/transient.jl:176-186
class _Linear:
    def __init__(self, us, ts):
        self.ts = [float(t) for t in ts]
        self.us = [np.asarray(u, float) for u in us]
        for a, b in zip(self.ts[:-1], self.ts[1:]):
            if not b > a:
                raise ValueError("knot-vectors must be unique and sorted in increasing order")

    def _eval(self, t):
        ts = self.ts
        if t < ts[0] or t > ts[-1]:
            raise IndexError("BoundsError: t = %r outside [%r, %r]" % (t, ts[0], ts[-1]))
        k = min(max(bisect.bisect_right(ts, t) - 1, 0), len(ts) - 2)  # the interval [ts[k], ts[k+1]] that holds t
        w = (t - ts[k]) / (ts[k + 1] - ts[k])
        return (1 - w) * self.us[k] + w * self.us[k + 1]

    def __call__(self, t):
        return self._eval(t)


class _Linear2(_Linear):
    def __call__(self, i, t):
        return self._eval(t)[int(i) - 1]


def getcontinuoussolution(us, ts, val=None):
    return _Linear2(us, ts) if val == 2 else _Linear(us, ts)


def _pieces(lo, hi, *knotlists):
    """[lo, hi] cut at the knots inside it: the integrands below are smooth on every piece."""
    cuts = sorted({float(lo), float(hi)} | {float(t) for ks in knotlists for t in ks if lo < t < hi})
    return list(zip(cuts[:-1], cuts[1:]))


_GL_X, _GL_W = np.polynomial.legendre.leggauss(6)  # exact for polynomials of degree <= 11 on each piece


def quadgk(f, lo, hi, knots=()):
    """QuadGK.quadgk(f, lo, hi)[1].  The integrands here are polynomials of low degree between the knots of the
    interpolants (products of piecewise-linear functions, squares of their differences), so a 6-point Gauss-Legendre rule
    on every piece is exact to rounding — what QuadGK's adaptive Gauss-Kronrod rule converges to within its default
    rtol = sqrt(eps).  Works for scalar- and vector-valued f."""
    total = None
    for a, b in _pieces(lo, hi, knots):
        h, c = 0.5 * (b - a), 0.5 * (a + b)
        for x, w in zip(_GL_X, _GL_W):
            v = (w * h) * np.asarray(f(c + h * x), float)
            total = v if total is None else total + v
    return total if total is not None else 0.0


quadgk_vec = quadgk


# ------------------------------------------------------------------ derivative twins of FiniteVolume.jl:75-139
def _conductance(K, aol, i, meta, logt):
    m = i if meta is None else int(meta[i]) - 1
    c = (math.exp(K[m]) if logt else K[m]) * aol[i]
    dc = c if logt else aol[i]  # d c / d K[m]
    return m, c, dc


def assembleb_p(node1, node2, aol, K, sources, dnodes, dheads, metaindex=None, logt=False):
    """d b / d p as a dense (len(p) x nfree) array, p = [conductivities; sources; dirichletheads] — from FiniteVolume.jl:110-139:
    b[f] = sources[node(f)] (:113-120); a face with exactly one Dirichlet end adds c_i * dhead to the free end's entry (:130-136)."""
    N, nK, ndir = len(sources), len(K), len(dheads)
    freenode, n2f = o.getfreenodes(N, dnodes)
    n2d = o.getnodei2dirichleti(np.zeros(N), dnodes)
    nfree = int(freenode.sum())
    J = np.zeros((nK + N + ndir, nfree))
    for node in range(N):
        if freenode[node]:
            J[nK + node, n2f[node] - 1] += 1.0
    for i in range(len(node1)):
        a, b = int(node1[i]) - 1, int(node2[i]) - 1
        m, c, dc = _conductance(K, aol, i, metaindex, logt)
        for fr, di in ((a, b), (b, a)):
            if freenode[fr] and not freenode[di]:
                f, d = n2f[fr] - 1, n2d[di] - 1
                J[m, f] += dc * dheads[d]
                J[nK + N + d, f] += c
    return J


def assembleA_px(x, node1, node2, aol, K, sources, dnodes, dheads, metaindex=None, logt=False):
    """d (A x) / d p as a dense (len(p) x nfree) array, x held fixed — from FiniteVolume.jl:91-104: a free|free face adds
    c to both diagonals and -c to both off-diagonals (:96-99), a face with one free end c to that diagonal (:100-103)."""
    N, nK, ndir = len(sources), len(K), len(dheads)
    freenode, n2f = o.getfreenodes(N, dnodes)
    nfree = int(freenode.sum())
    J = np.zeros((nK + N + ndir, nfree))
    for i in range(len(node1)):
        a, b = int(node1[i]) - 1, int(node2[i]) - 1
        m, c, dc = _conductance(K, aol, i, metaindex, logt)
        if freenode[a] and freenode[b]:
            fa, fb = n2f[a] - 1, n2f[b] - 1
            J[m, fa] += dc * (x[fa] - x[fb])
            J[m, fb] += dc * (x[fb] - x[fa])
        elif freenode[a]:
            J[m, n2f[a] - 1] += dc * x[n2f[a] - 1]
        elif freenode[b]:
            J[m, n2f[b] - 1] += dc * x[n2f[b] - 1]
    return J


# ------------------------------------------------------------------ src/transientadjointutils.jl:1-55
def getadjointfunctions(sigma, obsfreenodes, uobs, u0, tspan, Ss, volumes, node1, node2, aol, conductivities, sources, dnodes, dheads, metaindex=None, logt=False, **kwargs):
    nK, N, ndir = len(conductivities), len(sources), len(dheads)
    freenodes, n2f = o.getfreenodes(len(u0), dnodes)
    f2n = {int(n2f[node]): node + 1 for node in range(len(u0)) if freenodes[node]}  # Dict(zip(values, keys)), :3
    nfree = int(freenodes.sum())
    vols = Ss * np.asarray(volumes, float)

    def g(u, t):
        uobseval, ueval = uobs(t), u(t)
        retval = 0.0
        for i in obsfreenodes:
            retval += sigma(i, t) ** 2 * (ueval[f2n[i] - 1] - uobseval[f2n[i] - 1]) ** 2
        return retval

    def dgdu(u, t):
        uobseval, ueval = uobs(t), u(t)
        result = np.zeros(nfree)
        for i in obsfreenodes:
            result[i - 1] = 2 * sigma(i, t) ** 2 * (ueval[f2n[i] - 1] - uobseval[f2n[i] - 1])
        return result

    def split(p):
        p = np.asarray(p, float)
        return p[:nK], p[nK : nK + N], p[nK + N : nK + N + ndir]

    def dfdp(u, t, p):
        pK, ps, pd = split(p)
        ueval = np.asarray(u(t))[freenodes]
        A_px = assembleA_px(ueval, node1, node2, aol, pK, ps, dnodes, pd, metaindex, logt)
        b_p = assembleb_p(node1, node2, aol, pK, ps, dnodes, pd, metaindex, logt)
        result = b_p - A_px  # transpose(transpose(b_p - A_px) scaled): parent column i (free unknown i) / volumes[i], transient.jl:25-34
        for i in range(nfree):
            result[:, i] /= vols[i]
        return result

    dgdpval = np.zeros(nK + N + ndir)

    def dgdp(u, t, p):
        return dgdpval

    du0dp = np.zeros((nK + N + ndir, nfree))

    def G(p):
        if callable(p):
            knots = getattr(p, "ts", ())
            return quadgk(lambda t: g(p, t), tspan[0], tspan[1], knots)
        pK, ps, pd = split(p)
        us_p, ts_p = o.backwardeulerintegrate(u0, tspan, Ss, volumes, node1, node2, aol, pK, ps, dnodes, pd, metaindex, logt, **kwargs)
        return G(getcontinuoussolution(us_p, ts_p))

    return g, dgdu, dfdp, dgdp, du0dp, G


# ------------------------------------------------------------------ src/FiniteVolume.jl:271-377
def integrateb_pmA_pxlambda(lambdas, ts_lambda, u2, tspan, Ss, volumes, node1, node2, aol, conductivities, sources, dnodes, dheads, metaindex=None, logt=False):
    N, nK, ndir = len(sources), len(conductivities), len(dheads)
    nodei2dirichleti = o.getnodei2dirichleti(np.asarray(sources, float), dnodes)
    freenode, nodei2freenodei = o.getfreenodes(N, dnodes)
    lambda2 = getcontinuoussolution(lambdas, ts_lambda, 2)
    result = np.zeros(nK + N + ndir)
    productdict = {}
    knots_u = getattr(u2, "ts", ())

    def integrateproduct(i):  # i: node, 1-based (:277-285)
        if i not in productdict:
            productdict[i] = quadgk(lambda t: lambda2(nodei2freenodei[i - 1], t) * u2(i, t), tspan[0], tspan[1], list(ts_lambda) + list(knots_u))
        return productdict[i]

    lambdaintegral = simpleintegrate(lambdas, ts_lambda)
    volumes = np.asarray(volumes, float)
    j = 1
    for i in range(1, N + 1):  # :325-335
        if freenode[i - 1]:
            result[nK + i - 1] += lambdaintegral[j - 1] / (Ss * volumes[i - 1])
            j += 1
    if not logt:
        raise RuntimeError("not supported")  # :363
    for i in range(len(node1)):  # :339-360
        n1, n2 = int(node1[i]), int(node2[i])
        m = i if metaindex is None else int(metaindex[i]) - 1
        c = math.exp(conductivities[m]) * aol[i]
        for fr, di in ((n1, n2), (n2, n1)):
            if freenode[fr - 1] and not freenode[di - 1]:
                f = nodei2freenodei[fr - 1]  # free index; the reference takes volumes[f] (not the node's), kept as is
                d = nodei2dirichleti[di - 1]
                result[m] += c * dheads[d - 1] * lambdaintegral[f - 1] / (Ss * volumes[f - 1])
                result[nK + N + d - 1] += c * lambdaintegral[f - 1] / (Ss * volumes[f - 1])
                result[m] += c * integrateproduct(fr) / (Ss * volumes[f - 1])
                break
    return result


def integratedfdplambda(u2, p, lambdas, ts_lambda, tspan, Ss, volumes, node1, node2, aol, conductivities, sources, dnodes, dheads, metaindex=None, logt=False):
    nK, N, ndir = len(conductivities), len(sources), len(dheads)
    p = np.asarray(p, float)
    return integrateb_pmA_pxlambda(lambdas, ts_lambda, u2, tspan, Ss, volumes, node1, node2, aol, p[:nK], p[nK : nK + N], dnodes, p[nK + N : nK + N + ndir], metaindex, logt)


# ------------------------------------------------------------------ src/transient.jl:188-216
def adjointintegrate(*args, dt0=1.0, **kwargs):
    if callable(args[0]):
        getdgdu, tspan, Ss, volumes, node1, node2, aol, K, sources, dnodes, dheads = args[:11]
        metaindex = args[11] if len(args) > 11 else None
        logt = args[12] if len(args) > 12 else False
        A = o.assembleA(node1, node2, aol, K, sources, dnodes, dheads, metaindex, logt)
        freenodes, n2f = o.getfreenodes(len(volumes), dnodes)
        o.scalebyvolume_A(A, Ss * np.asarray(volumes, float), o.freenodei2nodei(n2f))
        cols = np.repeat(np.arange(1, A.n + 1), np.diff(A.colptr))
        At = o.sparse(cols, A.rowval, A.nzval, A.n, A.m)  # transpose(A), :193
        return adjointintegrate(At, getdgdu, tspan, dt0=dt0, **kwargs)
    A, getdgdu, tspan = args
    n = A.n if hasattr(A, "n") else np.asarray(A).shape[1]
    gamma0 = np.zeros(n)
    T = tspan[1]
    gammas, tsgamma = o.backwardeulerintegrate_generic(gamma0, A, lambda t: getdgdu(T - t), dt0, tspan[0], tspan[1], **kwargs)
    return list(reversed(gammas)), list(reversed([T - t for t in tsgamma]))


def gradientintegrate(lambdac_or_lambda0, du0dp, dgdp, dfdp_or_integral, tspan, knots=()):
    if callable(lambdac_or_lambda0):
        lambdac, dfdp = lambdac_or_lambda0, dfdp_or_integral
        I2 = quadgk_vec(lambda t: np.asarray(dfdp(t)) @ np.asarray(lambdac(t)), tspan[0], tspan[1], list(knots) + list(getattr(lambdac, "ts", ())))
        lambda0 = np.asarray(lambdac(0))
    else:
        lambda0, I2 = np.asarray(lambdac_or_lambda0, float), np.asarray(dfdp_or_integral, float)
    I1 = quadgk_vec(lambda t: np.asarray(dgdp(t), float), tspan[0], tspan[1])
    return np.asarray(du0dp) @ lambda0 + I1 + I2
