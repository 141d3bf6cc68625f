"""Oracle vs the documented contracts it restates (Julia stdlib sparse/range,
reference grid enumeration) and vs the reference's data fixtures.  CPU only."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from tests import refcases

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_sparse_contract_sorted_combined_zeros_kept(oracle):
    rng = np.random.default_rng(0)
    m, n, L = 37, 41, 900
    I = rng.integers(1, m + 1, L)
    J = rng.integers(1, n + 1, L)
    V = rng.standard_normal(L)
    V[::7] = 0.0  # explicit zeros must be kept
    A = oracle.sparse(I, J, V, m, n)
    assert A.colptr[0] == 1 and A.colptr[-1] == len(A.rowval) + 1
    for j in range(n):
        rows = A.rowval[A.colptr[j] - 1 : A.colptr[j + 1] - 1]
        assert (np.diff(rows) > 0).all()  # strictly ascending, no repeats
    # structure == set of distinct (i,j) pairs, zeros included
    assert len(A.rowval) == len(set(zip(I.tolist(), J.tolist())))
    # values: left fold in input order
    acc = {}
    for i, j, v in zip(I.tolist(), J.tolist(), V.tolist()):
        acc[(i, j)] = v if (i, j) not in acc else acc[(i, j)] + v
    for j in range(n):
        for k in range(A.colptr[j] - 1, A.colptr[j + 1] - 1):
            assert A.nzval[k] == acc[(int(A.rowval[k]), j + 1)]
    ref = sp.coo_matrix((V, (I - 1, J - 1)), shape=(m, n)).tocsc()
    assert np.allclose(A.toscipy().toarray(), ref.toarray(), rtol=1e-14, atol=1e-14)


def test_sparse_empty_and_out_of_range(oracle):
    A = oracle.sparse([], [], [], 3, 3)
    assert A.colptr.tolist() == [1, 1, 1, 1] and len(A.nzval) == 0
    with pytest.raises(oracle.OracleError):
        oracle.sparse([4], [1], [1.0], 3, 3)


def test_linrange_matches_exact_rationals(oracle):
    from fractions import Fraction

    for a, b, n in [(-50.0, 50.0, 101), (0.0, 10.0, 2), (0.0, 100.0, 464), (-50.0, 50.0, 256), (0.0, 1.0, 11), (-1.0, 2.0, 7)]:
        xs = oracle.linrange(a, b, n)
        for k in sorted({0, 1, min(2, n - 1), n // 2, max(n - 2, 0), n - 1}):
            exact = Fraction(a) + (Fraction(b) - Fraction(a)) * k / (n - 1)
            assert xs[k] == float(exact)  # float(Fraction) is correctly rounded
    assert oracle.linrange(0.0, 1.0, 11)[3] == 0.3  # Julia: range(0, stop=1, length=11)[4] == 0.3


@pytest.mark.parametrize("ns", [(3, 4, 5), (2, 2, 2), (5, 3, 2)])
def test_regulargrid_enumeration(oracle, ns):
    mins, maxs = [-50.0, -50.0, 0.0], [50.0, 50.0, 10.0]
    coords, n1, n2, aol, vol = oracle.regulargrid(mins, maxs, ns)
    N = ns[0] * ns[1] * ns[2]
    F = 3 * N - ns[0] * ns[1] - ns[0] * ns[2] - ns[1] * ns[2]
    assert coords.shape == (3, N) and len(n1) == F and len(vol) == N
    assert (n1 < n2).all()
    # independent restatement of grid.jl:72-108 in pure Python
    xs, ys, zs = (oracle.linrange(mins[d], maxs[d], ns[d]) for d in range(3))
    dx, dy, dz = xs[1] - xs[0], ys[1] - ys[0], zs[1] - zs[0]
    lin = lambda i1, i2, i3: i3 + ns[2] * (i2 - 1) + ns[2] * ns[1] * (i1 - 1)  # noqa: E731
    j = 0
    c = 0
    for i1 in range(1, ns[0] + 1):
        ax = dx * (0.5 if i1 in (1, ns[0]) else 1.0)
        for i2 in range(1, ns[1] + 1):
            ay = dy * (0.5 if i2 in (1, ns[1]) else 1.0)
            for i3 in range(1, ns[2] + 1):
                az = dz * (0.5 if i3 in (1, ns[2]) else 1.0)
                assert vol[c] == ax * ay * az
                c += 1
                li = lin(i1, i2, i3)
                assert tuple(coords[:, li - 1]) == (xs[i1 - 1], ys[i2 - 1], zs[i3 - 1])
                if i1 < ns[0]:
                    assert (n1[j], n2[j], aol[j]) == (li, lin(i1 + 1, i2, i3), ay * az / dx)
                    j += 1
                if i2 < ns[1]:
                    assert (n1[j], n2[j], aol[j]) == (li, lin(i1, i2 + 1, i3), ax * az / dy)
                    j += 1
                if i3 < ns[2]:
                    assert (n1[j], n2[j], aol[j]) == (li, lin(i1, i2, i3 + 1), ax * ay / dz)
                    j += 1
    assert j == F
    # total volume == box volume
    assert np.isclose(vol.sum(), 100 * 100 * 10, rtol=1e-12)


def test_assembly_invariants_box(oracle):
    """symmetric, zero row-sum away from Dirichlet, positive diagonal, max principle."""
    ns = (6, 5, 4)
    coords, n1, n2, aol, vol = oracle.regulargrid([-50.0, -50.0, 0.0], [50.0, 50.0, 10.0], ns)
    rng = np.random.default_rng(1)
    logk = np.log(1e-5) + rng.standard_normal(ns[::-1]).transpose(2, 1, 0).ravel(order="C") * 0  # shape only
    nodek = np.log(1e-5) + 1.5 * rng.standard_normal(len(vol))
    Kf = oracle.nodehycos2neighborhycos(n1, n2, nodek, True)
    assert Kf[0] == 0.5 * (nodek[n1[0] - 1] + nodek[n2[0] - 1])
    left = np.nonzero(coords[0] == -50.0)[0] + 1
    right = np.nonzero(coords[0] == 50.0)[0] + 1
    dn = np.sort(np.r_[left, right]).astype(np.int64)
    dh = np.where(np.isin(dn, left), 1.0, 0.0)
    src = np.zeros(len(vol))
    A = oracle.assembleA(n1, n2, aol, Kf, src, dn, dh, None, True)
    M = A.toscipy()
    assert abs(M - M.T).max() == 0.0
    assert (M.diagonal() > 0).all()
    h, ch, _, b, fn = oracle.solvediffusion(n1, n2, aol, np.exp(Kf), src, dn, dh)
    assert h.min() >= -1e-12 and h.max() <= 1 + 1e-12
    # rows whose cell has no Dirichlet neighbour sum to zero
    isd = np.zeros(len(vol) + 1, bool)
    isd[dn] = True
    touches = np.zeros(len(vol) + 1, bool)
    touches[n1[isd[n2]]] = True
    touches[n2[isd[n1]]] = True
    _, n2f = oracle.getfreenodes(len(vol), dn)
    rows = n2f[(~isd[1:]) & (~touches[1:])] - 1
    rs = np.asarray(M.sum(axis=1)).ravel()
    assert np.abs(rs[rows]).max() <= 1e-18 + 1e-12 * M.diagonal().max()


def test_fourfractures_fixture_vs_pflotran(oracle):
    """loose cross-code sanity (<= ~1 % relative)."""
    d = np.load(os.path.join(GOLDEN, "fourfractures.npz"))
    assert len(d["node1"]) == 6314 and len(d["xs"]) == 2106 and (d["node1"] < d["node2"]).all()
    src = np.zeros(2106)
    h, ch, A, b, fn = oracle.solvediffusion(d["node1"], d["node2"], d["areasoverlengths"], d["conductivities"], src, d["dirichletnodes"], d["dirichletheads"])
    rel = np.linalg.norm(h - d["pflotran_h"]) / np.linalg.norm(d["pflotran_h"])
    assert rel < 1.1e-2
    hf = d["pflotran_h"][fn]
    assert np.linalg.norm(A.matvec(hf) - b) / np.linalg.norm(b) < 1e-2
    deg = np.bincount(np.r_[d["node1"], d["node2"]])[1:]
    assert deg.min() == 3 and deg.max() == 14


@pytest.mark.slow
def test_theis_and_thiem(oracle):
    """test/theis.jl:52-65, the reference's default solvers."""
    c = refcases.theis(oracle.regulargrid)
    assert len(c["u0"]) == 20402 and len(c["aol"]) == 50601 and len(c["dnodes"]) == 4752
    us, ts = oracle.backwardeulerintegrate(c["u0"], c["tspan"], c["Ss"], c["volumes"], c["node1"], c["node2"], c["aol"], c["K"], c["sources"], c["dnodes"], c["dheads"], atol=c["atol"], dt0=c["dt0"])
    assert ts[-1] == c["tspan"][1]
    th = [refcases.theisdrawdown(ts[-1], r, c["T"], c["S"], c["Q"]) for r in c["rs"]]
    assert refcases.isapprox(th, -us[-1][c["goodnodes"]] + c["steadyhead"], atol=1e-4, rtol=2e-2)
    h, ch, A, b, fn = oracle.solvediffusion(c["node1"], c["node2"], c["aol"], c["K"], c["sources"], c["dnodes"], c["dheads"])
    assert (A.n, len(A.nzval)) == (15650, 93108)
    tm = [refcases.thiemdrawdown(r, c["T"], c["Q"], c["sidelength"]) for r in c["rs"]]
    assert refcases.isapprox(tm, -h[c["goodnodes"]] + c["steadyhead"], atol=1e-4, rtol=2e-2)
