"""GPU parity: grid, maps and assembly through the C ABI vs the CPU oracle.
Index arrays bit-exact; values bit-exact where the arithmetic is IEEE-exact
(non-log conductivities), <= 2 ulp where exp() is involved."""
import os

import numpy as np
import pytest

from tests import refcases

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _same_csc(A, B, exact_values=True, ulps=0):
    assert A.n == B.n
    assert np.array_equal(A.colptr, B.colptr)
    assert np.array_equal(A.rowval, B.rowval)
    if exact_values:
        assert np.array_equal(A.nzval, B.nzval)
    else:
        assert np.all(np.abs(A.nzval - B.nzval) <= ulps * np.spacing(np.abs(B.nzval)))


@pytest.mark.parametrize("ns", [(2, 2, 2), (3, 4, 5), (101, 101, 2), (17, 9, 33)])
def test_regulargrid_bit_exact(fv, oracle, ns):
    mins, maxs = [-50.0, -50.0, 0.0], [50.0, 50.0, 10.0]
    coords, nb, aol, vol = fv.regulargrid(mins, maxs, list(ns))
    oc, o1, o2, oaol, ovol = oracle.regulargrid(mins, maxs, list(ns))
    assert np.array_equal(coords, oc)
    assert np.array_equal(nb[:, 0], o1) and np.array_equal(nb[:, 1], o2)
    assert np.array_equal(aol, oaol)
    assert np.array_equal(vol, ovol)


def test_regulargrid_irrational_spacing_bit_exact(fv, oracle):
    mins, maxs, ns = [0.0, -1.0, 0.3], [1000.0, 2.0, 100.7], [47, 13, 29]
    coords, nb, aol, vol = fv.regulargrid(mins, maxs, ns)
    oc, o1, o2, oaol, ovol = oracle.regulargrid(mins, maxs, ns)
    assert np.array_equal(coords, oc) and np.array_equal(aol, oaol) and np.array_equal(vol, ovol)


def test_regulargrid_argument_errors(fv):
    with pytest.raises(fv.FVError, match="only 3 dimensions supported"):
        fv.regulargrid([0.0, 0.0], [1.0, 1.0], [2, 2])
    with pytest.raises(fv.FVError):
        fv.regulargrid([0.0, 0.0, 0.0], [1.0, 1.0, 1.0], [1, 2, 2])


@pytest.mark.parametrize("logt", [False, True])
def test_nodehycos2neighborhycos(fv, oracle, logt):
    ns = [5, 6, 7]
    _, nb, _, vol = fv.regulargrid([0.0, 0.0, 0.0], [1.0, 1.0, 1.0], ns)
    rng = np.random.default_rng(3)
    nodek = np.exp(rng.standard_normal((ns[2], ns[1], ns[0])))  # (n3,n2,n1) like the reference
    got = fv.nodehycos2neighborhycos(nb, nodek, logt)
    want = oracle.nodehycos2neighborhycos(nb[:, 0], nb[:, 1], nodek, logt)
    assert np.array_equal(got, want)


def test_free_maps_with_duplicates_and_empty(fv, oracle):
    rng = np.random.default_rng(5)
    N = 5000
    dn = rng.integers(1, N + 1, 700)  # contains repeats
    f, n2f = fv.getfreenodes(N, dn)
    of, on2f = oracle.getfreenodes(N, dn)
    assert np.array_equal(f, of) and np.array_equal(n2f, on2f)
    src = np.zeros(N)
    assert np.array_equal(fv.getnodei2dirichleti(src, dn), oracle.getnodei2dirichleti(src, dn))  # last occurrence wins
    f, n2f = fv.getfreenodes(7, [])
    assert f.all() and n2f.tolist() == [1, 2, 3, 4, 5, 6, 7]
    with pytest.raises(fv.FVError):
        fv.getfreenodes(7, [8])


def test_source_at_dirichlet_message(fv):
    c = refcases.chain4()
    s = c["sources"].copy()
    s[3] = 1.0
    s[0] = 2.0
    nb = np.stack([c["node1"], c["node2"]], 1)
    with pytest.raises(fv.FVError, match="There cannot be a source at a Dirichlet node, but node 1 is a Dirichlet node where a source is located."):
        fv.assembleb(nb, c["aol"], c["K"], s, c["dnodes"], c["dheads"])
    with pytest.raises(fv.FVError, match="node 4 is a Dirichlet"):
        fv.getnodei2dirichleti(s, [4, 1])
    # assembleA alone does not validate (FiniteVolume.jl:75-108 never calls getnodei2dirichleti)
    fv.assembleA(nb, c["aol"], c["K"], s, c["dnodes"], c["dheads"])


def test_chain4_assembly_counts_faces_twice(fv, oracle):
    c = refcases.chain4()
    nb = np.stack([c["node1"], c["node2"]], 1)
    A = fv.assembleA(nb, c["aol"], c["K"], c["sources"], c["dnodes"], c["dheads"])
    assert A.colptr.tolist() == [1, 3, 5] and A.rowval.tolist() == [1, 2, 1, 2] and A.nzval.tolist() == [4.0, -2.0, -2.0, 4.0]
    b = fv.assembleb(nb, c["aol"], c["K"], c["sources"], c["dnodes"], c["dheads"])
    assert b.tolist() == [2.0, 0.0]


def _box(fv, ns, seed=0, sigma=1.5):
    mins, maxs = [-50.0, -50.0, 0.0], [50.0, 50.0, 10.0]
    coords, nb, aol, vol = fv.regulargrid(mins, maxs, list(ns))
    rng = np.random.default_rng(seed)
    logk = np.log(1e-5) + sigma * rng.standard_normal(len(vol))
    left = np.nonzero(coords[0] == mins[0])[0] + 1
    right = np.nonzero(coords[0] == maxs[0])[0] + 1
    dn = np.sort(np.r_[left, right]).astype(np.int64)
    dh = np.where(np.isin(dn, left), 1.0, 0.0)
    return coords, nb, aol, vol, logk, dn, dh


@pytest.mark.parametrize("ns", [(3, 4, 5), (10, 10, 10), (33, 17, 9)])
def test_box_assembly_bit_exact(fv, oracle, ns):
    coords, nb, aol, vol, logk, dn, dh = _box(fv, ns)
    Kf = np.exp(fv.nodehycos2neighborhycos(nb, logk, True))
    src = np.zeros(len(vol))
    src[len(vol) // 2] = 1e-3
    A = fv.assembleA(nb, aol, Kf, src, dn, dh)
    oA = oracle.assembleA(nb[:, 0], nb[:, 1], aol, Kf, src, dn, dh)
    _same_csc(A, oA)
    b = fv.assembleb(nb, aol, Kf, src, dn, dh)
    ob = oracle.assembleb(nb[:, 0], nb[:, 1], aol, Kf, src, dn, dh)
    assert np.array_equal(b, ob)
    # log-transformed conductivities: exp() on the device may differ from libm in the last bits
    lk = fv.nodehycos2neighborhycos(nb, logk, True)
    A = fv.assembleA(nb, aol, lk, src, dn, dh, None, True)
    oA = oracle.assembleA(nb[:, 0], nb[:, 1], aol, lk, src, dn, dh, None, True)
    _same_csc(A, oA, exact_values=False, ulps=8)


def test_random_multigraph_with_repeats_selfloops_metaindex(fv, oracle):
    """Repeated faces, both orientations, self-loops, Dirichlet-Dirichlet faces,
    isolated nodes: sparse()'s combine order must still be reproduced bit for bit."""
    rng = np.random.default_rng(11)
    N, F = 400, 3000
    n1 = rng.integers(1, N - 20, F)  # the last 20 nodes stay isolated
    n2 = rng.integers(1, N - 20, F)
    n2[::50] = n1[::50]  # self-loops
    n1[1::3] = n1[0:-1:3][: len(n1[1::3])]  # repeats
    n2[1::3] = n2[0:-1:3][: len(n2[1::3])]
    aol = rng.random(F) + 0.1
    nK = 37
    K = rng.random(nK) + 0.5
    meta = rng.integers(1, nK + 1, F)
    dn = rng.choice(N, 60, replace=False) + 1
    dh = rng.standard_normal(60)
    src = rng.standard_normal(N)
    src[dn - 1] = 0
    nb = np.stack([n1, n2], 1)
    A = fv.assembleA(nb, aol, K, src, dn, dh, meta)
    oA = oracle.assembleA(n1, n2, aol, K, src, dn, dh, meta)
    _same_csc(A, oA)
    assert np.array_equal(fv.assembleb(nb, aol, K, src, dn, dh, meta), oracle.assembleb(n1, n2, aol, K, src, dn, dh, meta))
    # metaindex as a callable, as the reference's default argument style
    A2 = fv.assembleA(nb, aol, K, src, dn, dh, lambda i: int(meta[i - 1]))
    _same_csc(A2, oA)
    with pytest.raises(fv.FVError):
        fv.assembleA(nb, aol, K, src, dn, dh, np.full(F, nK + 1))
    with pytest.raises(fv.FVError):
        fv.assembleA(np.stack([n1, n2 + N], 1), aol, K, src, dn, dh, meta)


@pytest.mark.parametrize("seed", range(12))
def test_random_multigraph_assembly_bit_exact(fv, oracle, seed):
    """Differential fuzz of assembleA / assembleb against the oracle: random multigraphs (repeated faces, faces in both
    orientations, self-loops, isolated nodes), random Dirichlet sets with repeats (the last head wins), with and without a
    metaindex, plain and log conductivities (plain: every value bit-exact; log: exp() within 8 ulp), random sources on
    free nodes, node counts from 1 up — structure arrays always bit-exact."""
    rng = np.random.default_rng(seed)
    N = int(rng.integers(1, 60)) if seed % 4 else int(rng.integers(200, 3000))
    F = int(rng.integers(0, 6 * N + 1))
    n1 = rng.integers(1, N + 1, F)
    n2 = rng.integers(1, N + 1, F)
    if F > 4:  # repeat some faces, reversed as well
        k = rng.integers(0, F, F // 5)
        n1[k], n2[k] = n1[(k + 1) % F], n2[(k + 1) % F]
        k = rng.integers(0, F, F // 7)
        n1[k], n2[k] = n2[(k + 2) % F].copy(), n1[(k + 2) % F].copy()
    aol = np.exp(rng.standard_normal(F) * 2)
    ndir = int(rng.integers(0, N + 1)) if seed % 5 else N
    dn = rng.integers(1, N + 1, ndir + int(rng.integers(0, 3))).astype(np.int64)
    dh = rng.standard_normal(len(dn)) * 100
    logk = bool(seed % 2)
    if seed % 3 == 0:
        nK = int(rng.integers(1, 9))
        meta = rng.integers(1, nK + 1, F)
    else:
        nK, meta = F, None
    K = rng.standard_normal(nK) if logk else np.exp(rng.standard_normal(nK))
    free = np.ones(N, bool)
    free[dn - 1] = False
    src = np.where(free, rng.standard_normal(N), 0.0)
    nb = np.stack([n1, n2], 1).astype(np.int64)
    A = fv.assembleA(nb, aol, K, src, dn, dh, meta, logk)
    oA = oracle.assembleA(n1, n2, aol, K, src, dn, dh, meta, logk)
    b = fv.assembleb(nb, aol, K, src, dn, dh, meta, logk)
    ob = oracle.assembleb(n1, n2, aol, K, src, dn, dh, meta, logk)
    assert A.n == oA.n == int(free.sum()) and np.array_equal(A.colptr, oA.colptr) and np.array_equal(A.rowval, oA.rowval)
    if logk:
        scale = np.abs(oA.nzval).max() if len(oA.nzval) else 1.0
        assert np.allclose(A.nzval, oA.nzval, rtol=2e-15 * 8, atol=1e-300) or np.abs(A.nzval - oA.nzval).max() <= 1e-14 * scale
        assert np.allclose(b, ob, rtol=1e-12, atol=1e-12 * (np.abs(ob).max() if len(ob) else 1.0))
    else:
        assert np.array_equal(A.nzval, oA.nzval) and np.array_equal(b, ob)


def test_fourfractures_fixture_bit_exact(fv, oracle):
    d = np.load(os.path.join(GOLDEN, "fourfractures.npz"))
    nb = np.stack([d["node1"], d["node2"]], 1)
    src = np.zeros(2106)
    A = fv.assembleA(nb, d["areasoverlengths"], d["conductivities"], src, d["dirichletnodes"], d["dirichletheads"])
    oA = oracle.assembleA(d["node1"], d["node2"], d["areasoverlengths"], d["conductivities"], src, d["dirichletnodes"], d["dirichletheads"])
    _same_csc(A, oA)
    b = fv.assembleb(nb, d["areasoverlengths"], d["conductivities"], src, d["dirichletnodes"], d["dirichletheads"])
    assert np.array_equal(b, oracle.assembleb(d["node1"], d["node2"], d["areasoverlengths"], d["conductivities"], src, d["dirichletnodes"], d["dirichletheads"]))


def test_theis_grid_structure(fv, oracle):
    c = refcases.theis(lambda a, b, n: (lambda r: (r[0], r[1][:, 0], r[1][:, 1], r[2], r[3]))(fv.regulargrid(a, b, n)))
    nb = np.stack([c["node1"], c["node2"]], 1)
    A = fv.assembleA(nb, c["aol"], c["K"], c["sources"], c["dnodes"], c["dheads"])
    assert (A.n, len(A.nzval)) == (15650, 93108)
    oA = oracle.assembleA(c["node1"], c["node2"], c["aol"], c["K"], c["sources"], c["dnodes"], c["dheads"])
    _same_csc(A, oA)


def test_device_generated_grid_problem_matches_uploaded(fv):
    """fv_problem_create_regulargrid (faces never on the host) == fv_problem_create on regulargrid's output."""
    ns = [9, 8, 7]
    coords, nb, aol, vol, logk, dn, dh = _box(fv, ns)
    src = np.zeros(len(vol))
    p1 = fv.Problem.regulargrid([-50.0, -50.0, 0.0], [50.0, 50.0, 10.0], ns, dn)
    p2 = fv.Problem.create(nb, aol, len(vol), dn)
    K = np.array([1e-5])  # nK == 1: one conductivity for every face
    p1.assemble(K, src, dh)
    p2.assemble(np.full(len(aol), 1e-5), src, dh)
    a, b = p1.csc(), p2.csc()
    _same_csc(a, b)
    assert np.array_equal(p1.b(), p2.b())


def test_empty_and_all_dirichlet(fv):
    nb = np.array([[1, 2], [2, 3]], np.int64)
    A = fv.assembleA(nb, np.ones(2), np.ones(2), np.zeros(3), [1, 2, 3], np.zeros(3))
    assert A.n == 0 and A.colptr.tolist() == [1] and len(A.nzval) == 0
    A = fv.assembleA(np.empty((0, 2), np.int64), np.empty(0), np.empty(0), np.zeros(3), [], [])
    assert A.n == 3 and A.colptr.tolist() == [1, 1, 1, 1]
    head, freenode, n2f = fv.freenodes2nodes(np.array([5.0]), np.zeros(3), [1, 3], [7.0, 9.0])
    assert head.tolist() == [7.0, 5.0, 9.0] and freenode.tolist() == [False, True, False] and n2f.tolist() == [-1, 1, -1]


@pytest.mark.parametrize("name", ["oracle_box_3x4x5.npz", "oracle_box_10x10x10.npz"])
def test_committed_oracle_fixtures_grid_maps_assembly_and_heads(fv, name):
    """The device against the COMMITTED oracle fixtures (tests/golden/make_oracle_fixtures.py) — no oracle call: the
    files are the numbers a session with a Julia runtime will diff the real package against.  Grid, free maps, colptr / rowval /
    nzval / b bit for bit; the heads of the 10^3 box within 1e-8 of the fixture's direct solve."""
    g = np.load(os.path.join(GOLDEN, name))
    coords, nb, aol, vol = fv.regulargrid(list(g["mins"]), list(g["maxs"]), [int(v) for v in g["ns"]])
    assert np.array_equal(coords, g["coords"]) and np.array_equal(nb[:, 0], g["node1"]) and np.array_equal(nb[:, 1], g["node2"])
    assert np.array_equal(aol, g["areasoverlengths"]) and np.array_equal(vol, g["volumes"])
    f, n2f = fv.getfreenodes(len(vol), g["dirichletnodes"])
    assert np.array_equal(f, g["freenode"]) and np.array_equal(n2f, g["nodei2freenodei"])
    args = (nb, aol, g["conductivities"], g["sources"], g["dirichletnodes"], g["dirichletheads"])
    A = fv.assembleA(*args)
    assert np.array_equal(A.colptr, g["colptr"]) and np.array_equal(A.rowval, g["rowval"]) and np.array_equal(A.nzval, g["nzval"])
    assert np.array_equal(fv.assembleb(*args), g["b"])
    if "head_direct" in g.files:
        head, ch, A2, b2, freenode = fv.solvediffusion(*args)
        assert ch.isconverged
        p = fv.Problem.create(nb, aol, len(vol), g["dirichletnodes"]).assemble(g["conductivities"], g["sources"], g["dirichletheads"])
        tight, _, ch2 = p.solve_steady(None, 1e-13, 5000)
        p.close()
        assert np.linalg.norm(tight - g["head_direct"]) <= 1e-8 * np.linalg.norm(g["head_direct"])
        assert np.linalg.norm(head - g["head_direct"]) <= 1e-6 * np.linalg.norm(g["head_direct"])  # (the reference's default tolerance, sqrt(eps))
