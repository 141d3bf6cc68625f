"""The device-resident adjoint workflow (fv_trajectory.hip): the states of a forward run kept in HBM
(fv_trajectory — the reference's `us`, `ts`, /root/reference/src/transient.jl:136-154), the observation series on the device
(fv_observation), the adjoint sweep fv_adjoint_run (adjointintegrate, src/transient.jl:188-205, with dgdu of
src/transientadjointutils.jl:13-21 evaluated by a kernel over the observation rows at T - t), the objective G
(transientadjointutils.jl:46-49) and the gradient integral over two trajectories — against the host-closure path of the
mirrored API (which uploads a dense forcing per solve) and against the oracle's own adjointintegrate / getadjointfunctions
(oracle/fv_oracle_adjoint.py, pinned by tests/test_oracle_adjoint_kats.py)."""
import numpy as np
import pytest

from tests.test_gpu_adjoint_oracle import _case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def oa():
    from oracle import fv_oracle_adjoint

    return fv_oracle_adjoint


def relerr(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(b)


def test_trajectory_basics(fv):
    c = _case(fv)
    p = fv.Problem.create(c["nb"], c["aol"], c["N"], c["dn"]).assemble(np.exp(c["K"]), c["src"], c["dh"])
    st = p.transient_begin(c["Ss"], c["vol"], c["u0"])
    tr = p.new_trajectory()
    rng = np.random.default_rng(0)
    knots = [rng.standard_normal(p.n) for _ in range(4)]
    times = [0.0, 1.5, 2.0, 7.25]
    tr.push(st, times[0])
    first = st.free_values()
    for u, t in zip(knots[1:], times[1:]):
        tr.push_free(u, t)
    knots[0] = first
    assert len(tr) == 4 and np.array_equal(tr.ts, times)
    for k in range(4):
        assert np.array_equal(tr.free_values(k), knots[k])
    assert np.array_equal(tr.node_values(2), p.freenodes2nodes(knots[2]))
    for t in (0.0, 0.3, 1.5, 1.75, 6.0, 7.25):
        k = min(max(int(np.searchsorted(times, t, side="right")) - 1, 0), 2)
        w = (t - times[k]) / (times[k + 1] - times[k])
        assert np.array_equal(tr.at(t), (1.0 - w) * knots[k] + w * knots[k + 1])  # the host mirror's formula, bit for bit
    with pytest.raises(IndexError, match="BoundsError"):
        tr.at(7.5)
    with pytest.raises(fv.FVError, match="sorted in increasing order"):
        tr.push_free(knots[0], 7.25)
    tr.reverse_time(10.0)
    assert np.array_equal(tr.ts, [2.75, 8.0, 8.5, 10.0]) and np.array_equal(tr.free_values(0), knots[3]) and np.array_equal(tr.free_values(3), knots[0])
    tr.close()
    p.close()


def test_a_problem_destroyed_before_its_trajectories_detaches_them(fv):
    """Julia does not order finalizers, so fv_problem_destroy may come before fv_trajectory_destroy / fv_observation_destroy
    of its dependents.  The library keeps a list of them: destroying the problem releases their HBM and clears their problem
    pointer; every later call on them is refused (FV_ERR_ARG), and destroying them frees the handles without touching freed memory."""
    import ctypes as C

    lib = fv.load()
    c = _case(fv)
    p = fv.Problem.create(c["nb"], c["aol"], c["N"], c["dn"]).assemble(np.exp(c["K"]), c["src"], c["dh"])
    st = p.transient_begin(c["Ss"], c["vol"], c["u0"])
    tr, tr2 = p.new_trajectory(), p.new_trajectory()
    tr.push(st, 0.0)
    tr.push_free(np.zeros(p.n), 1.0)
    from fvamd.core import Observation

    ob = Observation(p, np.array([1, 2], dtype=np.int64), [0.0, 1.0], np.zeros((2, 2)))
    assert ob.integral(tr, 0.0, 1.0) >= 0.0
    tr2.close()  # (the usual order for one of them)
    p.close()
    n = C.c_int64()
    assert lib.fv_trajectory_size(tr.handle, C.byref(n)) != 0  # refused: the problem is gone
    G = C.c_double()
    assert lib.fv_observation_integral(tr.handle, ob.handle, 0.0, 1.0, C.byref(G)) != 0
    assert lib.fv_trajectory_clear(tr.handle) != 0
    tr.close()
    ob.close()
    assert tr.handle is None and ob.handle is None


@pytest.mark.parametrize("stepper", ["adaptive", "fixed"])
def test_states_recorded_in_hbm_are_the_host_loop_s(fv, stepper):
    """backwardeulerintegrate(..., keep="device") against the loop that downloads every state: the same `ts`, the same states."""
    c = _case(fv)
    mesh = (c["Ss"], c["vol"], c["nb"], c["aol"])
    rest = (c["src"], c["dn"], c["dh"], c["meta"], True)
    kw = dict(dt0=2.0e3, rtol=1e-13)
    if stepper == "fixed":
        kw["stepper"] = fv.fixedbackwardeulerstep
    else:
        kw["atol"] = 1e-7
    us, ts = fv.backwardeulerintegrate(c["u0"], c["tspan"], *mesh, c["K"], *rest, **kw)
    dus, dts = fv.backwardeulerintegrate(c["u0"], c["tspan"], *mesh, c["K"], *rest, keep="device", **kw)
    assert dts == ts and len(dus) == len(us) and len(us) > 8
    for k in (0, 1, len(us) // 2, len(us) - 1):
        assert np.abs(dus[k] - us[k]).max() <= 1e-12 * np.abs(us[k]).max()
    uc, duc = fv.getcontinuoussolution(us, ts), fv.getcontinuoussolution(dus, dts)
    for t in (0.0, 1.1e4, 3.999e4):
        assert np.abs(duc(t) - uc(t)).max() <= 1e-12 * np.abs(uc(t)).max()
    duc2 = fv.getcontinuoussolution(dus, dts, 2)
    assert abs(duc2(17, 1.1e4) - uc(1.1e4)[16]) <= 1e-12 * abs(uc(1.1e4)[16])


@pytest.mark.parametrize("stepper,metaindex", [("fixed", False), ("adaptive", False), ("adaptive", True)])
def test_device_adjoint_sweep_against_the_host_closure_path_and_the_oracle(fv, oracle, oa, stepper, metaindex):
    c = _case(fv, metaindex)
    mesh_fv = (c["Ss"], c["vol"], c["nb"], c["aol"])
    mesh_or = (c["Ss"], c["vol"], c["nb"][:, 0], c["nb"][:, 1], c["aol"])
    rest = (c["src"], c["dn"], c["dh"], c["meta"], True)
    kw = dict(dt0=5.0e3, rtol=1e-13)
    if stepper == "fixed":
        kw["stepper"] = fv.fixedbackwardeulerstep
        okw = dict(stepper=oracle.fixedbackwardeulerstep)
    else:
        kw["atol"] = 1e-6
        okw = dict(atol=1e-6)
    us, ts = fv.backwardeulerintegrate(c["u0"], c["tspan"], *mesh_fv, c["K"] + 0.3, *rest, **kw)  # "observations"
    uobs = fv.getcontinuoussolution(us, ts)
    us_i, ts_i = fv.backwardeulerintegrate(c["u0"], c["tspan"], *mesh_fv, c["K"], *rest, **kw)
    dus, dts = fv.backwardeulerintegrate(c["u0"], c["tspan"], *mesh_fv, c["K"], *rest, keep="device", **kw)
    assert dts == ts_i
    uc, duc = fv.getcontinuoussolution(us_i, ts_i), fv.getcontinuoussolution(dus, dts)
    freenode, n2f = fv.getfreenodes(c["N"], c["dn"])
    obsfree = [int(n2f[i]) for i in np.nonzero(freenode)[0][[5, 40, 90, 91]]]
    sigma = lambda i, t: 0.03 * (1 + 0.1 * i / 100)  # noqa: E731
    g, dgdu, dfdp, dgdp, du0dp, G = fv.getadjointfunctions(sigma, obsfree, uobs, c["u0"], c["tspan"], *mesh_fv, c["K"], *rest, **kw)
    # the objective: the device integral against the host mirror's and the oracle's
    og, odgdu, odfdp, odgdp, odu0dp, oG = oa.getadjointfunctions(sigma, obsfree, oa.getcontinuoussolution(us, ts), c["u0"], c["tspan"], *mesh_or, c["K"], c["src"], c["dn"], c["dh"], c["meta"], True)
    ouc = oa.getcontinuoussolution(us_i, ts_i)
    Gd, Gh, Go = G(duc), G(uc), oG(ouc)
    assert abs(Gd - Gh) <= 1e-11 * abs(Gh) and abs(Gd - Go) <= 1e-6 * abs(Go)  # (the oracle integrates g by its own Gauss-Kronrod between the knots)
    # the sweep: forcing evaluated on the device from the trajectory, against the host closure (a dense upload per solve) ...
    lam_h, ts_h = fv.adjointintegrate(lambda t: dgdu(uc, t), c["tspan"], *mesh_fv, c["K"], *rest, **kw)
    lam_d, ts_d = fv.adjointintegrate(dgdu.bind(duc), c["tspan"], *mesh_fv, c["K"], *rest, **kw)
    assert ts_d == ts_h and len(lam_d) == len(lam_h) > 5
    # the device sweep runs on the forward run's problem: arguments that are not the ones it was built from are refused, not ignored
    with pytest.raises(fv.FVError, match="conductivities differs"):
        fv.adjointintegrate(dgdu.bind(duc), c["tspan"], *mesh_fv, c["K"] + 0.01, *rest, **kw)
    scale = max(np.abs(np.asarray(l)).max() for l in lam_h)
    worst = max(np.abs(lam_d[k] - np.asarray(lam_h[k])).max() for k in range(len(lam_h)))
    assert worst <= 1e-10 * scale, worst / scale
    assert np.array_equal(lam_d[len(lam_d) - 1], np.zeros(len(lam_d[0])))  # lambda(T) = 0
    # ... and against the oracle's adjointintegrate with its own dgdu and a tight CG (an independent run)
    olam, ots = oa.adjointintegrate(lambda t: odgdu(ouc, t), c["tspan"], *mesh_or, c["K"], c["src"], c["dn"], c["dh"], c["meta"], True, dt0=5.0e3,
                                    linearsolver=oracle.tightcgsolver(1e-14), **okw)
    assert [float(t) for t in ots] == ts_d
    oworst = max(np.abs(lam_d[k] - np.asarray(olam[k])).max() for k in range(len(olam)))
    print("%s sweep, %d knots: device vs host closure %.2e, device vs oracle %.2e (of max |lambda| = %.3e)" % (stepper, len(ts_d), worst / scale, oworst / scale, scale))
    assert oworst <= 1e-8 * scale
    # the gradient integral with u and lambda both read from HBM, against the kernel fed with host knots
    p0 = np.r_[c["K"], c["src"], c["dh"]]
    for sc in ("reference", "storage", None):
        want = fv.devicegradientintegral(uc, [np.asarray(l) for l in lam_h], ts_h, c["tspan"], *mesh_fv, c["K"], *rest, scale=sc)
        got = fv.devicegradientintegral(duc, lam_d, ts_d, c["tspan"], *mesh_fv, c["K"], *rest, scale=sc)
        assert np.abs(got - want).max() <= 1e-9 * np.abs(want).max(), sc


def test_device_adjoint_sweep_under_the_renumbering(fv):
    """A face-list mesh numbered badly: the library re-numbers its free cells inside; observation rows, trajectories and lambdas
    cross the ABI in the caller's numbering."""
    import os

    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "fourfractures.npz"))
    m = dict(N=2106, node1=d["node1"], node2=d["node2"], aol=d["areasoverlengths"], K=d["conductivities"], dnodes=d["dirichletnodes"])
    rng = np.random.default_rng(5)
    order = rng.permutation(m["N"])
    rank = np.empty(m["N"], np.int64)
    rank[order] = np.arange(m["N"])
    nb = np.c_[rank[m["node1"] - 1] + 1, rank[m["node2"] - 1] + 1].astype(np.int64)
    dn = np.sort(rank[m["dnodes"] - 1] + 1).astype(np.int64)
    dh = np.full(len(dn), 1.5e6)
    vol = np.exp(rng.uniform(np.log(1e-3), np.log(1e-2), m["N"]))
    K = np.log(m["K"][0]) + 0.2 * rng.standard_normal(len(m["aol"]))
    src = np.zeros(m["N"])
    u0 = np.full(m["N"], 1.5e6)
    mesh = (1e-9, vol, nb, m["aol"])
    rest = (src, dn, dh, None, True)
    kw = dict(dt0=0.5, rtol=1e-13, stepper=fv.fixedbackwardeulerstep)
    tspan = (0.0, 4.0)
    src_obs = src.copy()
    free = np.setdiff1d(np.arange(1, m["N"] + 1), dn)
    src_obs[free[len(free) // 2] - 1] = 1e-9
    us, ts = fv.backwardeulerintegrate(u0, tspan, *mesh, K, src_obs, *rest[1:], **kw)
    uobs = fv.getcontinuoussolution(us, ts)
    us_i, ts_i = fv.backwardeulerintegrate(u0, tspan, *mesh, K, *rest, **kw)
    lib = fv.load()
    assert lib.fv_tune(31, 2) == 0  # the process-wide default of FV_OPT_REORDER: always (as the `forced` fixture of test_gpu_reorder.py)
    try:
        dus, dts = fv.backwardeulerintegrate(u0, tspan, *mesh, K, *rest, keep="device", **kw)
        assert dus.trajectory.problem.reorder_info()["reordered"]
        uc, duc = fv.getcontinuoussolution(us_i, ts_i), fv.getcontinuoussolution(dus, dts)
        freenode, n2f = fv.getfreenodes(m["N"], dn)
        well = free[len(free) // 2]  # the cell whose source makes the "observations" differ, and two more rows
        obsfree = [int(n2f[well - 1])] + [int(n2f[i]) for i in np.nonzero(freenode)[0][[3, 1200]]]
        g, dgdu, dfdp, dgdp, du0dp, G = fv.getadjointfunctions(lambda i, t: 1e-3, obsfree, uobs, u0, tspan, *mesh, K, *rest, **kw)
        lam_d, ts_d = fv.adjointintegrate(dgdu.bind(duc), tspan, *mesh, K, *rest, **kw)
    finally:
        lib.fv_tune(31, 1)
    lam_h, ts_h = fv.adjointintegrate(lambda t: dgdu(uc, t), tspan, *mesh, K, *rest, **kw)
    assert ts_d == ts_h
    scale = max(np.abs(np.asarray(l)).max() for l in lam_h)
    assert scale > 0 and max(np.abs(lam_d[k] - np.asarray(lam_h[k])).max() for k in range(len(lam_h))) <= 1e-9 * scale
    assert abs(G(duc) - G(uc)) <= 1e-10 * abs(G(uc))


def test_theisadjoint_workflow_with_every_state_in_hbm(fv):
    """test/theisadjoint.jl:12-84 with the forward states, the sweep and the gradient integral on the device (keep="device",
    dgdu.bind(uc), integratedfdplambda(complete=True)): the gradient equals the host-closure workflow's and holds up against central
    finite differences of G (G itself integrated on the device) on its largest conductivity and head entries."""
    import math

    atol, steadyhead, side, thick = 1e-4, 0.0, 50.0, 10.0
    mins, maxs, ns = [-side, -side, 0.0], [side, side, thick], [25, 25, 2]
    meanloghyco, Q, Ss = math.log(1e-5), 1e-3, 0.1
    sigma = lambda i, t: 0.03  # noqa: E731
    coords, neighbors, aol, volumes = fv.regulargrid(mins, maxs, ns)
    F, N = len(aol), coords.shape[1]
    center = np.nonzero((coords[0] == 0) & (coords[1] == 0))[0]
    sources = np.zeros(N)
    sources[center] = -2 * Q / (2 * len(center) - 2)
    sources[center[0]] = sources[center[-1]] = -Q / (2 * len(center) - 2)
    dnodes = np.nonzero(np.hypot(coords[0], coords[1]) - side >= 0)[0] + 1
    dheads = np.full(len(dnodes), steadyhead)
    u0 = np.full(N, steadyhead)
    tspan = (0.0, 60 * 60 * 24 * 1e1)
    kw = dict(atol=atol, dt0=60.0)
    mesh = (Ss, volumes, neighbors, aol)
    us, ts = fv.backwardeulerintegrate(u0, tspan, *mesh, np.full(F, meanloghyco + 1), sources, dnodes, dheads, None, True, **kw)
    uobs = fv.getcontinuoussolution(us, ts)
    K0 = np.full(F, meanloghyco)
    p0 = np.r_[K0, sources, dheads]
    rest = (K0, sources, dnodes, dheads, None, True)
    freenodes, n2f = fv.getfreenodes(N, dnodes)
    obsfreenodes = [int(n2f[i]) for i in center]
    g, dgdu, dfdp, dgdp, du0dp, G = fv.getadjointfunctions(sigma, obsfreenodes, uobs, u0, tspan, *mesh, *rest, **kw)
    # host-closure workflow (every state and every forcing through the host)
    us_i, ts_i = fv.backwardeulerintegrate(u0, tspan, *mesh, *rest, **kw)
    uc = fv.getcontinuoussolution(us_i, ts_i)
    lam_h, ts_h = fv.adjointintegrate(lambda t: dgdu(uc, t), tspan, *mesh, *rest, **kw)
    idl_h = fv.integratedfdplambda(fv.getcontinuoussolution(us_i, ts_i, 2), p0, lam_h, ts_h, tspan, *mesh, *rest, complete=True)
    dG_h = fv.gradientintegrate(lam_h[0], du0dp, lambda t: dgdp(uc, t, p0), idl_h, tspan)
    # the same with every state in HBM
    dus, dts = fv.backwardeulerintegrate(u0, tspan, *mesh, *rest, keep="device", **kw)
    assert dts == ts_i and len(dts) > 300
    duc = fv.getcontinuoussolution(dus, dts)
    lam_d, ts_d = fv.adjointintegrate(dgdu.bind(duc), tspan, *mesh, *rest, **kw)
    assert ts_d == ts_h
    idl_d = fv.integratedfdplambda(duc, p0, lam_d, ts_d, tspan, *mesh, *rest, complete=True)
    dG_d = fv.gradientintegrate(lam_d[0], du0dp, lambda t: dgdp(duc, t, p0), idl_d, tspan)
    assert np.abs(dG_d - dG_h).max() <= 1e-9 * np.abs(dG_h).max()
    assert abs(G(duc) - G(uc)) <= 1e-11 * abs(G(uc))

    def Gdev(pv):  # the objective with the forward states of the perturbed parameters kept in HBM as well
        s, t = fv.backwardeulerintegrate(u0, tspan, *mesh, pv[:F], pv[F : F + N], dnodes, pv[F + N :], None, True, keep="device", **kw)
        return G(fv.getcontinuoussolution(s, t))

    deltap = 1e-4
    for i in np.r_[np.argsort(-np.abs(dG_d[:F]), kind="stable")[:3], F + N + np.argsort(-np.abs(dG_d[F + N :]), kind="stable")[:1]]:
        pp, pm = p0.copy(), p0.copy()
        pp[i] += deltap
        pm[i] -= deltap
        x1 = (Gdev(pp) - Gdev(pm)) / (2 * deltap)
        assert abs(x1 - dG_d[i]) <= (2e-3 if i < F else 5e-2) * max(abs(x1), abs(dG_d[i])), (int(i), x1, dG_d[i])
