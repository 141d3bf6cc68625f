"""The gradient routines of the adjoint workflow against the oracle's restatement of them (oracle/fv_oracle_adjoint.py,
pinned by tests/test_oracle_adjoint_kats.py): the host mirror of getadjointfunctions, the default integratedfdplambda
(the reference's hand-unrolled integrateb_pmA_pxlambda, FiniteVolume.jl:271-377) and the device kernel
fv_param_gradient_integral (the integral of dfdp' lambda over time)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def oa():
    from oracle import fv_oracle_adjoint

    return fv_oracle_adjoint


def _case(fv, metaindex=False):
    mins, maxs, ns = [-50.0, -50.0, 0.0], [50.0, 50.0, 10.0], [7, 6, 4]
    coords, nb, aol, vol = fv.regulargrid(mins, maxs, ns)
    rng = np.random.default_rng(3)
    F, N = len(aol), len(vol)
    if metaindex:
        nK = 11
        meta = rng.integers(1, nK + 1, F)
        K = np.log(1e-5) + 0.5 * rng.standard_normal(nK)
    else:
        meta, K = None, np.log(1e-5) + 0.5 * rng.standard_normal(F)
    left = np.nonzero(coords[0] == mins[0])[0] + 1
    inner = np.array([N // 2 + 3], np.int64)  # a Dirichlet cell inside the domain as well
    dn = np.r_[left, inner].astype(np.int64)
    dh = np.r_[np.full(len(left), 1.0), [0.25]]
    src = np.zeros(N)
    src[N // 3] = 2e-4
    vol = vol * (1 + 0.3 * rng.random(N))  # unequal storage: free-index and node-index volumes differ
    return dict(nb=nb, aol=aol, vol=vol, K=K, meta=meta, dn=dn, dh=dh, src=src, Ss=0.1, N=N, F=F, u0=np.full(N, 0.5), tspan=(0.0, 4.0e4))


@pytest.mark.parametrize("metaindex", [False, True])
def test_adjoint_functions_and_gradient_integrals_match_the_oracle(fv, oracle, oa, metaindex):
    c = _case(fv, metaindex)
    mesh_fv = (c["Ss"], c["vol"], c["nb"], c["aol"])
    mesh_or = (c["Ss"], c["vol"], c["nb"][:, 0], c["nb"][:, 1], c["aol"])
    rest = (c["src"], c["dn"], c["dh"], c["meta"], True)
    kw = dict(stepper=fv.fixedbackwardeulerstep, dt0=5.0e3, rtol=1e-13)
    us, ts = fv.backwardeulerintegrate(c["u0"], c["tspan"], *mesh_fv, c["K"] + 0.3, *rest, **kw)  # "observations"
    us_i, ts_i = fv.backwardeulerintegrate(c["u0"], c["tspan"], *mesh_fv, c["K"], *rest, **kw)
    uobs, uc, uc2 = fv.getcontinuoussolution(us, ts), fv.getcontinuoussolution(us_i, ts_i), fv.getcontinuoussolution(us_i, ts_i, 2)
    freenode, n2f = fv.getfreenodes(c["N"], c["dn"])
    obsfree = [int(n2f[i]) for i in np.nonzero(freenode)[0][[5, 40, 90]]]
    sigma = lambda i, t: 0.03 * (1 + 0.1 * i / 100)  # noqa: E731
    p0 = np.r_[c["K"], c["src"], c["dh"]]
    g, dgdu, dfdp, dgdp, du0dp, G = fv.getadjointfunctions(sigma, obsfree, uobs, c["u0"], c["tspan"], *mesh_fv, c["K"], *rest, **kw)
    og, odgdu, odfdp, odgdp, odu0dp, oG = oa.getadjointfunctions(sigma, obsfree, oa.getcontinuoussolution(us, ts), c["u0"], c["tspan"], *mesh_or, c["K"], c["src"], c["dn"], c["dh"], c["meta"], True)
    ouc, ouc2 = oa.getcontinuoussolution(us_i, ts_i), oa.getcontinuoussolution(us_i, ts_i, 2)
    for t in (0.0, 1.234e4, 4.0e4):
        assert abs(g(uc, t) - og(ouc, t)) <= 1e-13 * abs(og(ouc, t))
        assert np.allclose(dgdu(uc, t), odgdu(ouc, t), rtol=1e-13, atol=0)
        M, oM = dfdp(uc, t, p0), odfdp(ouc, t, p0)
        assert np.abs(np.asarray(M.todense()) - oM).max() <= 1e-12 * np.abs(oM).max()  # the Jacobian the LinearAdjoints twins would give
    assert abs(G(uc) - oG(ouc)) <= 1e-9 * abs(oG(ouc))
    lambdas, ts_l = fv.adjointintegrate(lambda t: dgdu(uc, t), c["tspan"], *mesh_fv, c["K"], *rest, **kw)
    lambdas = [np.asarray(l) for l in lambdas]
    # the reference's hand-unrolled integral (the default of integratedfdplambda), term by term
    idl = fv.integratedfdplambda(uc2, p0, lambdas, ts_l, c["tspan"], *mesh_fv, c["K"], *rest)
    oidl = oa.integratedfdplambda(ouc2, p0, lambdas, ts_l, c["tspan"], *mesh_or, c["K"], c["src"], c["dn"], c["dh"], c["meta"], True)
    assert np.abs(idl - oidl).max() <= 1e-11 * np.abs(oidl).max()
    # the device kernel: the time integral of dfdp(t)' lambda(t) with the reference's scaling, against the oracle's
    # quadrature of its own dfdp (exact on the pieces between the knots of u and lambda)
    olam = oa.getcontinuoussolution(lambdas, ts_l)
    want = oa.quadgk_vec(lambda t: odfdp(ouc, t, p0) @ olam(t), c["tspan"][0], c["tspan"][1], list(ts_l) + list(ts_i))
    got = fv.devicegradientintegral(uc, lambdas, ts_l, c["tspan"], *mesh_fv, c["K"], *rest, scale="reference")
    assert np.abs(got - want).max() <= 1e-11 * np.abs(want).max()
    # and what gradientintegrate(lambdac, du0dp, dgdp, dfdp, tspan) makes of it
    odG = oa.gradientintegrate(olam, odu0dp, lambda t: odgdp(ouc, t, p0), lambda t: odfdp(ouc, t, p0), c["tspan"], knots=ts_i)
    dG = fv.gradientintegrate(lambdas[0], du0dp, lambda t: dgdp(uc, t, p0), got, c["tspan"])
    assert np.abs(dG - odG).max() <= 1e-10 * np.abs(odG).max()
    # the POINTWISE Jacobian on the device (getadjointfunctions(..., device=True): dfdp(u, t, p) as a DeviceJacobian whose product
    # with a vector is fv_param_jacobian_apply) against the oracle's dfdp matrix applied to the same vectors
    _, _, dfdp_dev, _, _, _ = fv.getadjointfunctions(sigma, obsfree, uobs, c["u0"], c["tspan"], *mesh_fv, c["K"], *rest, device=True, **kw)
    rng = np.random.default_rng(4)
    for t in (0.0, 1.234e4, 4.0e4):
        J, oM = dfdp_dev(uc, t, p0), odfdp(ouc, t, p0)
        assert J.shape == oM.shape
        for vec in (np.asarray(olam(t)), rng.standard_normal(oM.shape[1])):
            assert np.abs(J @ vec - oM @ vec).max() <= 1e-12 * np.abs(oM @ vec).max()
    # ... and through gradientintegrate's own quadrature of dfdp(t) * lambdac(t)
    lamc = fv.getcontinuoussolution(lambdas, ts_l)
    dG2 = fv.gradientintegrate(lamc, du0dp, lambda t: dgdp(uc, t, p0), lambda t: dfdp_dev(uc, t, p0), c["tspan"])
    assert np.abs(dG2 - odG).max() <= 1e-6 * np.abs(odG).max()  # (adaptive Gauss-Kronrod on the host against the oracle's exact pieces)
