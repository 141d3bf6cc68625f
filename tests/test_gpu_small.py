"""The single-launch Jacobi-PCG of small systems (fv_small.hip; fv_tune key 61 = largest system it takes, 0 = never) against the
classic loop of separate launches: the same iteration counts, heads to rounding (the sums group differently), the residual
history of a steady solve, implicit steps with the assembled b, a caller's forcing and the adjoint mode, breakdown reported the
same way — and against the oracle's direct solves (/root/reference/src/transient.jl:50-76, src/FiniteVolume.jl:157-165)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def relerr(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(b)


def _box(fv, ns, seed=0):
    mins, maxs = [-50.0, -50.0, 0.0], [50.0, 50.0, 10.0]
    coords, nb, aol, vol = fv.regulargrid(mins, maxs, list(ns))
    rng = np.random.default_rng(seed)
    K = np.exp(fv.nodehycos2neighborhycos(nb, np.log(1e-5) + 0.8 * rng.standard_normal(len(vol)), True))
    dn = (np.nonzero((coords[0] == mins[0]) | (coords[0] == maxs[0]))[0] + 1).astype(np.int64)
    dh = np.where(coords[0][dn - 1] == mins[0], 1.0, 0.0)
    src = np.zeros(len(vol))
    src[len(vol) // 2 + 3] = -1e-4
    return nb, aol, vol, K, dn, dh, src


@pytest.mark.parametrize("ns", [(15, 13, 11), (31, 33, 30), (5, 4, 3)])
def test_single_launch_solver_against_the_classic_loop_and_the_oracle(fv, oracle, ns):
    nb, aol, vol, K, dn, dh, src = _box(fv, ns)
    N = len(vol)
    lib = fv.load()
    out = {}
    try:
        for key in (1 << 15, 0):
            assert lib.fv_tune(61, key) == 0
            p = fv.Problem.create(nb, aol, N, dn).assemble(K, src, dh)
            head, res, ch = p.solve_steady(None, 1e-12, 5000, want_resnorm=True)
            assert ch.isconverged
            st = p.transient_begin(0.1, vol, np.full(N, 0.5))
            its = []
            for dt, k in ((5.0, 3), (4000.0, 2), (0.01, 2)):
                for _ in range(k):
                    info = p.step(st, st, dt, None, 0, 1e-13, 5000)
                    assert info.converged
                    its.append(info.iters)
            u = st.node_values()
            bh = np.sin(np.arange(p.n) * 0.37) * 1e-6  # a caller's forcing (the volume-scaled getb(t) of transient.jl:71), then the adjoint mode
            for mode in (0, 1):
                info = p.step(st, st, 50.0, bh, mode, 1e-13, 5000)
                assert info.converged
                its.append(info.iters)
            out[key] = (head, np.asarray(ch.data["resnorm"]), ch.iters, u, st.node_values(), its)
            p.close()
    finally:
        lib.fv_tune(61, 1 << 15)
    a, b = out[1 << 15], out[0]
    assert a[2] == b[2] and a[5] == b[5], (a[2], b[2], a[5], b[5])
    live = b[1] > 1e-9 * b[1][0]  # (further down — a tiny system terminates exactly — the two recurrences are rounding noise)
    assert np.allclose(a[1][live], b[1][live], rtol=1e-3, atol=0) and len(a[1]) == a[2]
    assert relerr(a[0], b[0]) < 1e-11 and relerr(a[3], b[3]) < 1e-12 and relerr(a[4], b[4]) < 1e-12
    ohead = oracle.solvediffusion(nb[:, 0], nb[:, 1], aol, K, src, dn, dh, solver="direct")[0]
    assert relerr(a[0], ohead) < 1e-8
    ous, _ = oracle.backwardeulerintegrate(np.full(N, 0.5), (0.0, 15.0), 0.1, vol, nb[:, 0], nb[:, 1], aol, K, src, dn, dh, stepper=oracle.fixedbackwardeulerstep, dt0=5.0,
                                          linearsolver=oracle.directlinearsolver)
    p = fv.Problem.create(nb, aol, N, dn).assemble(K, src, dh)
    st = p.transient_begin(0.1, vol, np.full(N, 0.5))
    for _ in range(3):
        p.step(st, st, 5.0, None, 0, 1e-13, 5000)
    assert relerr(st.node_values(), ous[-1]) < 1e-8
    p.close()


def test_single_launch_solver_reports_a_breakdown_like_the_classic_loop(fv):
    """An indefinite operator (a negative conductivity): p.Ap <= 0 ends the solve unconverged, with the classic loop's message."""
    nb, aol, vol, K, dn, dh, src = _box(fv, (9, 8, 7), seed=3)
    K = K.copy()
    K[::2] *= -1.0
    p = fv.Problem.create(nb, aol, len(vol), dn).assemble(K, src, dh)
    head, res, ch = p.solve_steady(None, 1e-12, 200, want_resnorm=False)
    assert not ch.isconverged
    p.close()


def test_step_doubling_attempt_as_one_launch_is_the_three_solves_one_by_one(fv):
    """The adaptive stepper on a small system: the full step, the two half steps and the error norm of an attempt enqueued as ONE
    launch (fv_small_twostep) against the same attempt as three launches and a norm (FV_SMALL_TWOSTEP=0): the same accepted
    times, the same number of solves and, bit for bit, the same states — forward with rejected attempts in it, and the
    device-resident adjoint sweep (time-dependent forcing on the observation rows, weighted norm)."""
    import os

    nb, aol, vol, K, dn, dh, src = _box(fv, (17, 15, 6), seed=3)
    N = len(vol)
    out = {}
    for how in ("1", "0"):
        os.environ["FV_SMALL_TWOSTEP"] = how
        try:
            p = fv.Problem.create(nb, aol, N, dn).assemble(K, src, dh)
            st = p.transient_begin(0.1, vol, np.zeros(N))
            tr = p.new_trajectory()
            p.record(tr)
            ts, nsolves, info = p.run_adaptive(st, 0.0, 4.0e5, dt0=1.0e4, atol=1e-5, rtol=1e-12)  # (dt0 too large for atol: the first attempts are rejected)
            p.record(None)
            assert info.converged and nsolves > 3 * (len(ts) - 1)
            rows = np.array([5, 40, 300, 700])
            from fvamd.core import Observation

            obs = Observation(p, rows, np.array([0.0, 4.0e5]), np.zeros((2, len(rows))))
            lam, nout, nl, info_l = p.adjoint_run(tr, obs, 0.0, 4.0e5, dt0=5.0e4, atol=1e-3, rtol=1e-12)
            assert nl > 3 * nout  # (rejected attempts here too)
            tl = lam.ts
            out[how] = (np.array(ts), nsolves, st.node_values(), np.array(tl), nl, lam.free_values(0), lam.free_values(len(tl) // 2))
            for o in (tr, lam, obs):
                o.close()
            p.close()
        finally:
            os.environ.pop("FV_SMALL_TWOSTEP", None)
    a, b = out["1"], out["0"]
    assert len(a[0]) > 10 and np.array_equal(a[0], b[0]) and a[1] == b[1] and np.array_equal(a[2], b[2])
    assert len(a[3]) > 10 and np.array_equal(a[3], b[3]) and a[4] == b[4] and np.array_equal(a[5], b[5]) and np.array_equal(a[6], b[6])
    assert np.abs(a[5]).max() > 0

