"""CPU: the host-side seams of the integrator and the adjoint hooks, exercised with a
user-supplied linearsolver and host matrices exactly as the reference's tests do
(test/ode.jl:31-40, test/odeadjoint.jl).  No GPU is touched on this path."""
import math

import numpy as np


def test_odeadjoint_closed_form(fv):
    """test/odeadjoint.jl: d/dp of ∫x dt with dx/dt = b x, x(0) = a"""
    a, b, T = 1.0, 2.0, 1.0
    p = [a, b]
    x = lambda t: np.array([p[0] * math.exp(p[1] * t)])  # noqa: E731
    lam = lambda t: (1 - math.exp(p[1] * (T - t))) / b  # noqa: E731
    grad = np.array([1 / p[1] * (math.exp(p[1] * T) - 1), -p[0] / p[1] ** 2 * (math.exp(p[1] * T) - 1) + p[0] / p[1] * math.exp(p[1] * T) * T])
    A = np.full((1, 1), p[1])
    linearsolver = lambda A, b, x0: np.linalg.solve(A, b)  # noqa: E731
    xs, ts_x = fv.backwardeulerintegrate(x(0), -A, lambda t: np.zeros(1), 1e-5, 0.0, T, linearsolver=linearsolver, atol=1e-8)
    assert ts_x[-1] == T
    X = np.array([v[0] for v in xs])
    Xe = np.array([x(t)[0] for t in ts_x])
    assert np.linalg.norm(X - Xe) <= 1e-4 * max(np.linalg.norm(X), np.linalg.norm(Xe))
    lambdas, ts_l = fv.adjointintegrate(-A, lambda t: -np.ones(1), (0.0, T), dt0=1e-5, linearsolver=linearsolver, atol=1e-8)
    assert ts_l[0] == 0.0 and ts_l[-1] == T  # returned in terms of lambda: time runs forward again
    L = np.array([v[0] for v in lambdas])
    Le = np.array([lam(t) for t in ts_l])
    assert np.linalg.norm(L - Le) <= 1e-4 * max(np.linalg.norm(L), np.linalg.norm(Le))
    xc = fv.getcontinuoussolution(xs, ts_x)
    lambdac = fv.getcontinuoussolution(lambdas, ts_l)
    assert abs(xc(0.5 * (ts_x[3] + ts_x[4]))[0] - 0.5 * (xs[3][0] + xs[4][0])) < 1e-14
    dx0dp = np.array([[-1.0], [0.0]])

    def dfdp(t):
        r = np.zeros((2, 1))
        r[1, 0] = -xc(t)[0]
        return r

    g = fv.gradientintegrate(lambdac, dx0dp, lambda t: np.zeros(2), dfdp, (0.0, T))
    assert np.linalg.norm(g - grad) <= 1e-4 * np.linalg.norm(grad)


def test_diagonalupdate_and_scalebyvolume_host_objects(fv):
    """transient.jl:1-48 on host matrices (the objects a user linearsolver receives)."""
    A = fv.SparseMatrixCSC(3, 3, np.array([1, 3, 5, 6]), np.array([1, 2, 1, 2, 3]), np.array([4.0, -1.0, -1.0, 4.0, 2.0]))
    fv.diagonalupdate(A, 0.5)
    assert A.nzval.tolist() == [4.5, -1.0, -1.0, 4.5, 2.5]
    fv.diagonalupdate(A, -0.5)
    assert A.nzval.tolist() == [4.0, -1.0, -1.0, 4.0, 2.0]
    M = np.eye(2)
    fv.diagonalupdate(M, 2.0)
    assert M.tolist() == [[3.0, 0.0], [0.0, 3.0]]
    vols = np.array([2.0, 10.0, 4.0, 8.0])
    f2n = np.array([1, 3, 4])  # free index -> node
    fv.scalebyvolume(A, vols, f2n)  # row scaling, transient.jl:19
    assert A.nzval.tolist() == [2.0, -0.25, -0.5, 1.0, 0.25]
    b = np.array([2.0, 4.0, 8.0])
    fv.scalebyvolume(b, vols, f2n)
    assert b.tolist() == [1.0, 1.0, 1.0]


def test_getcontinuoussolution_bounds(fv):
    import pytest

    uc = fv.getcontinuoussolution([np.zeros(2), np.ones(2)], [0.0, 2.0])
    assert uc(0.5).tolist() == [0.25, 0.25] and uc(2.0).tolist() == [1.0, 1.0]
    with pytest.raises(IndexError):
        uc(2.5)


def test_integratedfdplambda_rejects_plain_conductivity(fv):
    """FiniteVolume.jl:363 error("not supported")"""
    import pytest

    with pytest.raises(Exception, match="not supported"):
        fv.integratedfdplambda(lambda t: np.zeros(2), np.zeros(4), [np.zeros(1)] * 2, [0.0, 1.0], (0.0, 1.0), 1.0, np.ones(2), np.array([[1, 2]]), np.ones(1), np.ones(1), np.zeros(2), np.array([1]), np.zeros(1), None, False)


def test_uge_mesh_reader_roundtrip(fv, tmp_path):
    """examples/fractures/setupmesh.jl:3-47: CELLS / CONNECTIONS of a .uge file -> neighbours, areas / centre distances,
    volumes; conductivities as the geometric mean of the two cells' fracture values."""
    xs = np.array([0.0, 1.0, 1.0, 3.0])
    ys = np.array([0.0, 0.0, 2.0, 2.0])
    zs = np.array([0.0, 0.0, 0.0, 1.0])
    vol = np.array([0.5, 0.25, 2.0, 1.0])
    conn = [(1, 2, 0.5, 0.0, 0.0, 0.3), (2, 3, 1.0, 1.0, 0.0, 0.7), (3, 4, 2.0, 2.0, 0.5, 1.1)]
    path = tmp_path / "full_mesh_vol_area.uge"
    with open(path, "w") as f:
        f.write("CELLS 4\n")
        for i in range(4):
            f.write("%d %.17g %.17g %.17g %.17g\n" % (i + 1, xs[i], ys[i], zs[i], vol[i]))
        f.write("CONNECTIONS 3\n")
        for c in conn:
            f.write("%d %d %.17g %.17g %.17g %.17g\n" % c)
    m = fv.meshio.read_uge(str(path))
    assert m["node1"].tolist() == [1, 2, 3] and m["node2"].tolist() == [2, 3, 4]
    assert np.array_equal(m["volumes"], vol) and np.array_equal(m["xs"], xs)
    want = np.array([0.3 / 1.0, 0.7 / 2.0, 1.1 / np.sqrt(4.0 + 0.0 + 1.0)])
    assert np.allclose(m["areasoverlengths"], want, rtol=1e-15)
    k = fv.meshio.fracture_conductivities(m["node1"], m["node2"], [1e-12, 4e-12], [1, 1, 2, 2])
    assert np.allclose(k, [1e-12, 2e-12, 4e-12], rtol=1e-15)
    dn, dh = fv.meshio.dirichlet_from_predicate(xs, ys, zs, lambda x, y, z: x == 0.0 or x == 3.0, lambda x, y, z: 2e6 if x < 1 else 1e6)
    assert dn.tolist() == [1, 4] and dh.tolist() == [2e6, 1e6]
    import pytest

    with open(path, "w") as f:
        f.write("CELLS 1\n1 0 0 0 1\nCONNECTIONS 1\n1 2 0 0 0 1\n")
    with pytest.raises(ValueError, match="outside"):
        fv.meshio.read_uge(str(path))


def test_locality_order_and_reorder_mesh(fv):
    """meshio.locality_order / reorder_mesh: a permutation (RCM) that brings connected cells together, faces kept in
    order and re-oriented first < second, per-node arrays permuted, Dirichlet nodes renamed."""
    from tests import workloads

    w = workloads.fractures_like(3, 40, seed=1)
    order, rank = fv.meshio.locality_order(w["node1"], w["node2"], w["N"])
    assert sorted(order.tolist()) == list(range(1, w["N"] + 1)) and np.array_equal(rank[order - 1], np.arange(1, w["N"] + 1))
    m = fv.meshio.reorder_mesh(dict(node1=w["node1"], node2=w["node2"], aol=w["aol"], volumes=w["volumes"], dnodes=w["dnodes"], dheads=w["dheads"]), rank)
    assert (m["node1"] < m["node2"]).all() and np.array_equal(m["aol"], w["aol"]) and np.array_equal(m["dheads"], w["dheads"])
    assert np.array_equal(m["dnodes"], rank[w["dnodes"] - 1])
    assert np.array_equal(m["volumes"][rank - 1], w["volumes"])  # new position of old node i holds its volume
    assert np.array_equal(np.sort(np.stack([m["node1"], m["node2"]], 1), 1), np.sort(np.stack([rank[w["node1"] - 1], rank[w["node2"] - 1]], 1), 1))
    spread = lambda a, b: np.abs(a - b).mean()  # noqa: E731
    assert spread(m["node1"], m["node2"]) < 0.25 * spread(w["node1"], w["node2"])


def test_mesh_jld_reader_on_the_reference_data_files(fv, tmp_path):
    """read_mesh_jld / load_jld (HDF5-subset reader, no HDF5 library) on the two data files of
    examples/fractures/fourfractures, against the same data extracted with h5dump (tests/golden/fourfractures.npz,
    make_fourfractures.py): every array bit for bit."""
    import os

    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    want = np.load(os.path.join(golden, "fourfractures.npz"))
    mesh = fv.meshio.read_mesh_jld(os.path.join(golden, "fourfractures", "mesh.jld"))
    for name in ("xs", "ys", "zs", "areasoverlengths", "conductivities", "dirichletnodes", "dirichletheads", "fractureindices", "node1", "node2"):
        assert mesh[name].dtype == want[name].dtype and np.array_equal(mesh[name], want[name]), name
    assert mesh["neighbors"].shape == (6314, 2) and np.array_equal(mesh["metaindex"], want["fractureindices"][want["node1"] - 1])
    h = fv.meshio.load_jld(os.path.join(golden, "fourfractures", "pflotran_solution.jld"), "h")
    assert np.array_equal(h, want["pflotran_h"])
    everything = fv.meshio.load_jld(os.path.join(golden, "fourfractures", "mesh.jld"))
    assert sorted(everything) == sorted(["xs", "ys", "zs", "neighbors", "areasoverlengths", "fractureindices", "dirichletnodes", "dirichletheads", "conductivities"])
    # errors: a missing variable, a file that is not HDF5, a truncated file
    import pytest

    with pytest.raises(KeyError, match="no variable volumes"):
        fv.meshio.load_jld(os.path.join(golden, "fourfractures", "mesh.jld"), "volumes")
    bad = tmp_path / "bad.jld"
    bad.write_bytes(b"Julia data file (HDF5), version 0.1.1" + b"\0" * 2000)
    with pytest.raises(fv.meshio.JLDFormatError, match="no HDF5 signature"):
        fv.meshio.load_jld(str(bad))
    data = open(os.path.join(golden, "fourfractures", "mesh.jld"), "rb").read()
    bad.write_bytes(data[:60000])
    with pytest.raises(fv.meshio.JLDFormatError):
        fv.meshio.load_jld(str(bad))


def test_fehm_grid_zone_and_stor_readers_on_files_written_from_the_format_descriptions(fv, oracle, tmp_path):
    """examples/watertable/setupmodel.jl:7-19 needs FEHM.parsegrid / parsestor / parsezone (FEHM.jl: outside the tree, no
    sample files): the readers against files laid out as the LaGriT / FEHM manuals describe, from a regulargrid whose
    connections, coefficients and volumes are known; then the solver's own inputs rebuilt from them."""
    import pytest

    mins, maxs, ns = [0.0, 0.0, 0.0], [4.0, 3.0, 2.0], [5, 4, 3]
    coords, n1, n2, aol, vol = oracle.regulargrid(mins, maxs, ns)  # (the device grid generator needs a GPU)
    nb = np.stack([n1, n2], axis=1)
    N = coords.shape[1]
    # .fehmn
    grid = tmp_path / "g.fehmn"
    with open(grid, "w") as f:
        f.write("coor\n%10d\n" % N)
        for i in range(N):
            f.write("%10d %20.12e %20.12e %20.12e\n" % (i + 1, coords[0, i], coords[1, i], coords[2, i]))
        f.write("%10d\nelem\n 8 0\n\nstop\n" % 0)
    assert np.array_equal(fv.meshio.read_fehm_grid(str(grid)), coords)
    # .zone
    top = np.nonzero(coords[2] == maxs[2])[0] + 1
    west = np.nonzero(coords[0] == mins[0])[0] + 1
    zone = tmp_path / "g_outside.zone"
    with open(zone, "w") as f:
        f.write("zone\n00001  top\nnnum\n%10d\n" % len(top))
        for k in range(0, len(top), 10):
            f.write(" ".join("%10d" % v for v in top[k : k + 10]) + "\n")
        f.write("\n00003  left_w\nnnum\n%10d\n" % len(west))
        f.write(" ".join("%10d" % v for v in west) + "\n\nstop\n")
    zonenums, nodes = fv.meshio.read_fehm_zones(str(zone))
    assert zonenums == [1, 3] and np.array_equal(nodes[0], top) and np.array_equal(nodes[1], west)
    # .stor: full symmetric pattern with diagonal, coefficients -A/d compressed to their distinct values
    import scipy.sparse as sp

    a, b = nb[:, 0] - 1, nb[:, 1] - 1
    S = sp.coo_matrix((np.r_[-aol, -aol, np.zeros(N)], (np.r_[a, b, np.arange(N)], np.r_[b, a, np.arange(N)])), shape=(N, N)).tocsr()
    S.sort_indices()
    rowsum = np.asarray(-S.sum(axis=1)).ravel()
    data = S.data.copy()
    diag_pos = np.empty(N, np.int64)
    for i in range(N):
        lo, hi = S.indptr[i], S.indptr[i + 1]
        k = lo + np.searchsorted(S.indices[lo:hi], i)
        data[k] = rowsum[i]
        diag_pos[i] = k
    distinct, inverse = np.unique(data, return_inverse=True)
    ncoef = len(data)
    stor = tmp_path / "g.stor"

    def block(f, values, fmt, per_line=5):
        for k in range(0, len(values), per_line):
            f.write("".join(fmt % v for v in values[k : k + per_line]) + "\n")

    with open(stor, "w") as f:
        f.write("fehmstor ascir8i4 LaGriT Sparse Matrix Voronoi Coefficients\n")
        f.write(" Sun Oct  4 00:00:00 2026 3-D Linear Diffusion Model (matbld3d_astor)\n")
        f.write("%10d%10d%10d%10d%10d\n" % (len(distinct), N, ncoef + N + 1, 1, 7))
        block(f, vol, "%20.12E")
        block(f, np.r_[S.indptr + N + 1, S.indices + 1], "%10d")
        block(f, np.r_[inverse + 1, np.zeros(N + 1, np.int64), diag_pos + N + 2], "%10d")
        block(f, distinct, "%20.12E")
    volumes, aols, neighbors = fv.meshio.read_stor(str(stor))
    assert np.allclose(volumes, vol, rtol=1e-12) and neighbors.shape == (ncoef, 2)
    good = neighbors[:, 0] < neighbors[:, 1]  # setupmodel.jl:13-15
    order = np.lexsort((nb[:, 1], nb[:, 0]))
    assert np.array_equal(neighbors[good], nb[order]) and np.allclose(aols[good], aol[order], rtol=1e-12)
    # malformed files
    bad = tmp_path / "bad.stor"
    bad.write_text(open(stor).read()[:600])
    with pytest.raises(ValueError, match="ends before"):
        fv.meshio.read_stor(str(bad))
    bad.write_text("not a matrix\n\n1 2 3 1\n")
    with pytest.raises(ValueError, match="not a LaGriT stor file"):
        fv.meshio.read_stor(str(bad))
