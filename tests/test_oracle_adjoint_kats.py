"""Pins the oracle's restatement of the adjoint / gradient routines (oracle/fv_oracle_adjoint.py) against the reference's
own known-answer tests for them, at the reference's tolerances.  CPU only."""
import math
import os

import numpy as np
import pytest

from tests import refcases


@pytest.fixture(scope="module")
def oa():
    from oracle import fv_oracle_adjoint

    return fv_oracle_adjoint


def test_simpleintegrate_is_the_trapezoid_rule(oa):
    ts = np.array([0.0, 0.5, 2.0, 2.25])
    fs = [np.array([1.0, t]) for t in ts]
    assert np.allclose(oa.simpleintegrate(fs, ts), [2.25, 0.5 * 2.25**2], rtol=1e-15)  # exact for linear integrands


def test_odeadjoint_closed_form(oracle, oa):
    """test/odeadjoint.jl:1-41: dx/dt = b x, x(0) = a; lambda(t) = (1 - exp(b (T - t))) / b; dG/dp in closed form."""
    a, b, T = 1.0, 2.0, 1.0
    p = [a, b]
    A = np.array([[b]])
    x = lambda t: np.array([a * math.exp(b * t)])  # noqa: E731
    lam = lambda t: (1 - math.exp(b * (T - t))) / b  # noqa: E731
    gradient = np.array([1 / b * (math.exp(b * T) - 1), -a / b**2 * (math.exp(b * T) - 1) + a / b * math.exp(b * T) * T])
    solver = oracle.directlinearsolver
    xs, ts_x = oracle.backwardeulerintegrate_generic(x(0), -A, lambda t: np.zeros(1), 1e-5, 0.0, T, linearsolver=solver, atol=1e-8)
    assert refcases.isapprox(np.concatenate(xs), np.concatenate([x(t) for t in ts_x]), rtol=1e-4)
    lambdas, ts_l = oa.adjointintegrate(-A, lambda t: -np.ones(1), (0.0, T), dt0=1e-5, linearsolver=solver, atol=1e-8)
    assert refcases.isapprox([l[0] for l in lambdas], [lam(t) for t in ts_l], rtol=1e-4)
    xc, lambdac = oa.getcontinuoussolution(xs, ts_x), oa.getcontinuoussolution(lambdas, ts_l)
    dx0dp = np.array([[-1.0], [0.0]])
    dfdp = lambda t: np.array([[0.0], [-xc(t)[0]]])  # noqa: E731
    adj = oa.gradientintegrate(lambdac, dx0dp, lambda t: np.zeros(2), dfdp, (0.0, T), knots=ts_x)
    assert refcases.isapprox(adj, gradient, rtol=1e-4)


def test_onenodeadjoint(oracle, oa):
    """test/onenodeadjoint.jl:45-75: g / dgdu identities, lambda against the closed form, dG/dp against finite differences."""
    c = refcases.onenode(0.0)
    sigma = lambda i, t: 0.01  # noqa: E731
    kw = dict(atol=c["atol"], dt0=c["dt0"], linearsolver=oracle.directlinearsolver)
    mesh = (c["Ss"], c["volumes"], c["node1"], c["node2"], c["aol"])
    us, ts = oracle.backwardeulerintegrate(c["u0"], c["tspan"], *mesh, c["K"], c["sources"], c["dnodes"], c["dheads"], None, True, **kw)
    uobs = oa.getcontinuoussolution(us, ts)
    p0 = np.r_[c["K"] + 1, c["sources"], c["dheads"]]
    us_i, ts_i = oracle.backwardeulerintegrate(c["u0"], c["tspan"], *mesh, c["K"] + 1, c["sources"], c["dnodes"], c["dheads"], None, True, **kw)
    uc_init = oa.getcontinuoussolution(us_i, ts_i)
    freenodes, n2f = oracle.getfreenodes(2, c["dnodes"])
    obsfreenodes = [int(n2f[1])]
    g, dgdu, dfdp, dgdp, du0dp, G = oa.getadjointfunctions(sigma, obsfreenodes, uobs, c["u0"], c["tspan"], *mesh, c["K"], c["sources"], c["dnodes"], c["dheads"], None, True, **kw)
    t1 = c["tspan"][1]
    assert g(uobs, 0.5 * t1) == 0
    assert np.array_equal(dgdu(lambda t: uobs(t) + 1, 0.5 * t1), [2 * sigma(1, 0.5 * t1) ** 2])
    u_init = lambda t: (1 - math.exp(-math.e * t)) / math.e  # noqa: E731
    u_obs = lambda t: 1 - math.exp(-t)  # noqa: E731
    f_an = lambda s: 2 * sigma(1, s) ** 2 * (u_init(s) - u_obs(s))  # noqa: E731
    lambdas, ts_l = oa.adjointintegrate(lambda t: dgdu(uc_init, t), c["tspan"], *mesh, c["K"] + 1, c["sources"], c["dnodes"], c["dheads"], None, True, **kw)
    from scipy.integrate import quad

    gamma = lambda t: math.exp(-math.e * t) * quad(lambda s: math.exp(math.e * s) * f_an(t1 - s), 0, t)[0]  # noqa: E731
    for l, t in zip(lambdas, ts_l):
        want = gamma(t1 - t)
        assert abs(l[0] - want) <= max(1e-7, 1e-4 * max(abs(l[0]), abs(want)))
    lambdac = oa.getcontinuoussolution(lambdas, ts_l)
    dGdp = oa.gradientintegrate(lambdac, du0dp, lambda t: dgdp(uc_init, t, p0), lambda t: dfdp(uc_init, t, p0), c["tspan"], knots=ts_i)
    deltap = 1e-8
    for i in (0, 2, 3):  # importantindices = [1, 3, 4]
        pp, pm = p0.copy(), p0.copy()
        pp[i] += deltap
        pm[i] -= deltap
        x1 = (G(pp) - G(pm)) / (2 * deltap)
        assert abs(x1 - dGdp[i]) <= 1e-2 * max(abs(x1), abs(dGdp[i])), (i, x1, dGdp[i])
    # the hand-unrolled integral of the same quantity carries the source and head terms of this one-face problem exactly
    # (and the conductivity term with the reference's plus sign on the u product, FiniteVolume.jl:346)
    uc2 = oa.getcontinuoussolution(us_i, ts_i, 2)
    idl = oa.integratedfdplambda(uc2, p0, lambdas, ts_l, c["tspan"], *mesh, c["K"] + 1, c["sources"], c["dnodes"], c["dheads"], None, True)
    assert np.allclose(idl[2:], dGdp[2:], rtol=1e-6, atol=1e-14)


def test_theisadjoint_twenty_largest_entries(oracle, oa):
    """test/theisadjoint.jl:1-84: 25 x 25 x 2 cells, uniform log-conductivity mean + 1 observed, mean assumed; the 20
    largest entries of dG/dp from adjointintegrate + integratedfdplambda + gradientintegrate against central differences
    of G at rtol 1e-3."""
    atol, steadyhead, sidelength, thickness = 1e-4, 0.0, 50.0, 10.0
    mins, maxs, ns = [-sidelength, -sidelength, 0.0], [sidelength, sidelength, thickness], [25, 25, 2]
    meanloghyco, Q, Ss = math.log(1e-5), 1e-3, 0.1
    sigma = lambda i, t: 0.03  # noqa: E731
    coords, node1, node2, aol, volumes = oracle.regulargrid(mins, maxs, ns)
    F, N = len(aol), coords.shape[1]
    loghycos = np.full(F, meanloghyco + 1)
    sources = np.zeros(N)
    center = [i for i in range(N) if coords[0, i] == 0 and coords[1, i] == 0]
    sources[center[0]] = sources[center[-1]] = -Q / (2 * len(center) - 2)
    for i in center[1:-1]:
        sources[i] = -2 * Q / (2 * len(center) - 2)
    dn = [i + 1 for i in range(N) if math.hypot(coords[0, i], coords[1, i]) - sidelength >= 0]
    dnodes, dheads = np.array(dn, np.int64), np.full(len(dn), steadyhead)
    u0 = np.full(N, steadyhead)
    tspan = (0.0, 60 * 60 * 24 * 1e1)
    kw = dict(atol=atol, dt0=60.0, linearsolver=oracle.directlinearsolver)
    mesh = (Ss, volumes, node1, node2, aol)
    us, ts = oracle.backwardeulerintegrate(u0, tspan, *mesh, loghycos, sources, dnodes, dheads, None, True, **kw)
    uobs = oa.getcontinuoussolution(us, ts)
    K0 = np.full(F, meanloghyco)
    p0 = np.r_[K0, sources, dheads]
    us_i, ts_i = oracle.backwardeulerintegrate(u0, tspan, *mesh, K0, sources, dnodes, dheads, None, True, **kw)
    uc_init, uc_init2 = oa.getcontinuoussolution(us_i, ts_i), oa.getcontinuoussolution(us_i, ts_i, 2)
    freenodes, n2f = oracle.getfreenodes(N, dnodes)
    obsfreenodes = [int(n2f[i]) for i in center]
    g, dgdu, dfdp, dgdp, du0dp, G = oa.getadjointfunctions(sigma, obsfreenodes, uobs, u0, tspan, *mesh, K0, sources, dnodes, dheads, None, True, **kw)
    lambdas, ts_l = oa.adjointintegrate(lambda t: dgdu(uc_init, t), tspan, *mesh, K0, sources, dnodes, dheads, None, True, **kw)
    idl = oa.integratedfdplambda(uc_init2, p0, lambdas, ts_l, tspan, *mesh, K0, sources, dnodes, dheads, None, True)
    dGdp = oa.gradientintegrate(lambdas[0], np.zeros((len(p0), int(freenodes.sum()))), lambda t: dgdp(uc_init, t, p0), idl, tspan)
    important = np.argsort(-np.abs(dGdp), kind="stable")[:20]
    deltap = 1e-4
    # every finite difference costs two adaptive integrations (~6 s each on one core): the four largest entries by default,
    # all twenty of the reference's test with FV_FULL_KATS=1 (4.5 min; passed in round 2)
    for i in important[: 20 if os.environ.get("FV_FULL_KATS") == "1" else 4]:
        pp, pm = p0.copy(), p0.copy()
        pp[i] += deltap
        pm[i] -= deltap
        x1 = (G(pp) - G(pm)) / (2 * deltap)
        assert abs(x1 - dGdp[i]) <= 1e-3 * max(abs(x1), abs(dGdp[i])), (int(i), x1, dGdp[i])
