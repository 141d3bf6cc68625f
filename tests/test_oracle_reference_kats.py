"""Pins the CPU oracle against the reference's own known-answer tests
(SURVEY.md §8c items 1-4).  CPU only."""
import math

import numpy as np
import pytest

from tests import refcases


def test_chain4_solvediffusion(oracle):
    c = refcases.chain4()
    for solver in ("direct", "cg", "pcg"):
        h, ch, A, b, freenode = oracle.solvediffusion(c["node1"], c["node2"], c["aol"], c["K"], c["sources"], c["dnodes"], c["dheads"], solver=solver)
        assert refcases.isapprox(h, c["expected"])
        assert ch.isconverged
    assert list(freenode) == [False, True, True, False]
    # both directions listed => every face counted twice (FiniteVolume.jl:94-105)
    assert A.colptr.tolist() == [1, 3, 5] and A.rowval.tolist() == [1, 2, 1, 2]
    assert A.nzval.tolist() == [4.0, -2.0, -2.0, 4.0]
    assert b.tolist() == [2.0, 0.0]


def test_source_at_dirichlet_raises(oracle):
    c = refcases.chain4()
    s = c["sources"].copy()
    s[3] = 1.0
    with pytest.raises(oracle.OracleError, match="There cannot be a source at a Dirichlet node, but node 4"):
        oracle.assembleb(c["node1"], c["node2"], c["aol"], c["K"], s, c["dnodes"], c["dheads"])


def test_ode_diagonal_decay(oracle):
    """test/ode.jl:8-19"""
    v = np.array([1.0, 2.0, 3.0])
    A = oracle.sparse([1, 2, 3], [1, 2, 3], v, 3, 3)
    ys, ts = oracle.backwardeulerintegrate_generic(np.ones(3), A, np.zeros(3), 0.0001, 0.0, 2.0, atol=1e-8)
    assert ts[-1] == 2.0 and len(ts) > 10
    for y, t in zip(ys, ts):
        assert np.allclose(y, np.exp(-v * t), atol=1e-4, rtol=0)


def test_ode_growth(oracle):
    """test/ode.jl:21-29: dy/dt = y + 1"""
    A = oracle.sparse([1], [1], [-1.0], 1, 1)
    ys, ts = oracle.backwardeulerintegrate_generic(np.zeros(1), A, np.ones(1), 0.0001, 0.0, 1.0, atol=1e-8)
    for y, t in zip(ys, ts):
        assert abs(y[0] - (math.exp(t) - 1)) <= 1e-4


def test_ode_dense_oscillator(oracle):
    """test/ode.jl:31-40: dense non-symmetric 2x2 with linearsolver = A \\ b"""
    s7 = math.sqrt(7)

    def y(t, c1=1, c2=2):
        e = math.exp(-t / 4)
        a = np.array([1, 0.75]) * math.cos(s7 * t / 4) - np.array([0, -s7 / 4]) * math.sin(s7 * t / 4)
        b = np.array([1, 0.75]) * math.sin(s7 * t / 4) + np.array([0, -s7 / 4]) * math.cos(s7 * t / 4)
        return c1 * e * a + c2 * e * b

    A = -np.array([[0.5, -1.0], [1.0, -1.0]])
    ys, ts = oracle.backwardeulerintegrate_generic(y(0), A, np.zeros(2), 1e-4, 0.0, 1e2, atol=1e-8, linearsolver=oracle.directlinearsolver)
    assert ts[-1] == 1e2
    for yy, t in zip(ys, ts):
        assert np.linalg.norm(yy - y(t)) <= 1e-4


@pytest.mark.parametrize("loghyco,analytic", [(0.0, lambda t: 1 - math.exp(-t)), (1.0, lambda t: (1 - math.exp(-math.e * t)) / math.e)])
def test_onenode_logconductivity(oracle, loghyco, analytic):
    """test/onenodeadjoint.jl:29-44"""
    c = refcases.onenode(loghyco)
    us, ts = oracle.backwardeulerintegrate(c["u0"], c["tspan"], c["Ss"], c["volumes"], c["node1"], c["node2"], c["aol"], c["K"], c["sources"], c["dnodes"], c["dheads"], None, True, atol=c["atol"], dt0=c["dt0"])
    assert ts[-1] == 1.0
    for u, t in zip(us, ts):
        assert u[0] == 0.0
        assert abs(u[1] - analytic(t)) <= 1e-4 * max(abs(u[1]), abs(analytic(t)))


def test_time_step_must_be_positive(oracle):
    A = oracle.sparse([1], [1], [1.0], 1, 1)
    with pytest.raises(oracle.OracleError, match="time step must be positive"):
        oracle.backwardeuleronestep(np.zeros(1), A, np.zeros(1), np.zeros(1), 0.0, oracle.defaultlinearsolver, 1e-4)
