"""Inputs of the reference's own known-answer tests, restated as data.

Each builder returns plain numpy inputs (1-based int64 indices, as Julia sees
them) so the same case can be fed to the oracle and to the HIP path.
"""
import math

import numpy as np


def chain4():
    """/root/reference/test/runtests.jl:4-16 — faces listed in both directions."""
    node1 = np.array([1, 2, 2, 3, 3, 4], np.int64)
    node2 = np.array([2, 1, 3, 2, 4, 3], np.int64)
    aol = np.ones(6)
    K = np.ones(6)
    sources = np.zeros(4)
    dnodes = np.array([1, 4], np.int64)
    dheads = np.array([1.0, 0.0])
    expected = np.array([1.0, 2 / 3, 1 / 3, 0.0])
    return dict(node1=node1, node2=node2, aol=aol, K=K, sources=sources, dnodes=dnodes, dheads=dheads, expected=expected)


def W(u):
    """Well function approximation, /root/reference/test/theis.jl:6-12."""
    if u <= 1:
        return -math.log(u) + -0.57721566 + 0.99999193 * u + -0.24991055 * u**2 + 0.05519968 * u**3 + -0.00976004 * u**4 + 0.00107857 * u**5
    return (u**2 + 2.334733 * u + 0.250621) / (u**2 + 3.330657 * u + 1.681534) * math.exp(-u) / u


def theisdrawdown(t, r, T, S, Q):
    return Q * W(r**2 * S / (4 * T * t)) / (4 * math.pi * T)


def thiemdrawdown(r, T, Q, R):
    return Q * math.log(R / r) / (2 * math.pi * T)


def theis(regulargrid, ns=(101, 101, 2)):
    """/root/reference/test/theis.jl:21-50.  `regulargrid` is the grid builder
    under test (oracle's or the product's), returning coords(3,N), node1, node2, aol, volumes."""
    steadyhead = 1e3
    sidelength = 50.0
    thickness = 10.0
    mins = [-sidelength, -sidelength, 0.0]
    maxs = [sidelength, sidelength, thickness]
    k = 1e-5
    Q = 1e-3
    Ss = 0.1
    coords, node1, node2, aol, volumes = regulargrid(mins, maxs, list(ns))
    N = coords.shape[1]
    hycos = np.full(len(aol), k)
    sources = np.zeros(N)
    center = np.nonzero((coords[0] == 0) & (coords[1] == 0))[0]
    nc = len(center)
    sources[center[0]] = -Q / (2 * nc - 2)
    sources[center[-1]] = -Q / (2 * nc - 2)
    sources[center[1:-1]] = -2 * Q / (2 * nc - 2)
    rad = np.sqrt(coords[0] ** 2 + coords[1] ** 2)
    dn = np.nonzero(rad - sidelength >= 0)[0]
    dnodes = (dn + 1).astype(np.int64)
    dheads = np.full(len(dn), steadyhead)
    r0 = 0.1
    good = np.nonzero((coords[2] == thickness) & (coords[1] == 0) & (coords[0] > r0) & (coords[0] <= sidelength / 2))[0]
    return dict(
        coords=coords, node1=node1, node2=node2, aol=aol, volumes=volumes, K=hycos, sources=sources, dnodes=dnodes, dheads=dheads,
        steadyhead=steadyhead, sidelength=sidelength, thickness=thickness, k=k, Q=Q, Ss=Ss, S=Ss * thickness, T=thickness * k,
        goodnodes=good, rs=coords[0, good], tspan=(0.0, 60 * 60 * 24 * 1e1), atol=1e-4, dt0=60.0, u0=np.full(N, steadyhead),
    )


def isapprox(x, y, atol=0.0, rtol=None):
    """Julia isapprox on vectors: norm(x-y) <= max(atol, rtol*max(norm(x), norm(y)))."""
    x, y = np.asarray(x, float), np.asarray(y, float)
    if rtol is None:
        rtol = math.sqrt(np.finfo(float).eps) if atol == 0 else 0.0
    return np.linalg.norm(x - y) <= max(atol, rtol * max(np.linalg.norm(x), np.linalg.norm(y)))


def onenode(loghyco=0.0):
    """/root/reference/test/onenodeadjoint.jl:17-44 (2 cells, 1 face, log-conductivity)."""
    return dict(
        Ss=1.0, volumes=np.array([1.0, 1.0]), node1=np.array([1], np.int64), node2=np.array([2], np.int64), aol=np.array([1.0]),
        K=np.array([loghyco]), sources=np.array([0.0, 1.0]), dnodes=np.array([1], np.int64), dheads=np.array([0.0]),
        u0=np.array([0.0, 0.0]), tspan=(0.0, 1.0), atol=1e-8, dt0=1e-3,
    )
