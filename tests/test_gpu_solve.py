"""GPU parity: SpMV, reductions, steady Jacobi-PCG and implicit stepping through
the C ABI vs the CPU oracle.  Floating-point bar (BASELINE.json north_star): heads
within 1e-8 relative of the CPU reference, with the device PCG run to rtol 1e-12
and the oracle solved directly (SURVEY.md §7 risk 2)."""
import math
import os

import numpy as np
import pytest

from tests import refcases

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
HEAD_RTOL = 1e-8


def relerr(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(b)


def _box(fv, ns, seed=0, sigma=1.5):
    mins, maxs = [-50.0, -50.0, 0.0], [50.0, 50.0, 10.0]
    coords, nb, aol, vol = fv.regulargrid(mins, maxs, list(ns))
    rng = np.random.default_rng(seed)
    logk = np.log(1e-5) + sigma * rng.standard_normal(len(vol))
    K = np.exp(fv.nodehycos2neighborhycos(nb, logk, True))
    left = np.nonzero(coords[0] == mins[0])[0] + 1
    right = np.nonzero(coords[0] == maxs[0])[0] + 1
    dn = np.sort(np.r_[left, right]).astype(np.int64)
    dh = np.where(np.isin(dn, left), 1.0, 0.0)
    return coords, nb, aol, vol, K, dn, dh


@pytest.mark.parametrize("ns", [(3, 4, 5), (20, 19, 18), (64, 33, 7)])
def test_spmv_and_dot_match_oracle(fv, oracle, ns):
    coords, nb, aol, vol, K, dn, dh = _box(fv, ns)
    src = np.zeros(len(vol))
    p = fv.Problem.create(nb, aol, len(vol), dn).assemble(K, src, dh)
    oA = oracle.assembleA(nb[:, 0], nb[:, 1], aol, K, src, dn, dh)
    rng = np.random.default_rng(1)
    x = rng.standard_normal(p.n)
    y = p.spmv(x)
    oy = oA.matvec(x)
    assert np.allclose(y, oy, rtol=0, atol=1e-14 * np.abs(oA.nzval).max() * np.abs(x).max() * 8)
    # shifted operator (A + sigma*D) x
    p.transient_begin(0.1, vol, np.zeros(len(vol)))
    freenode, _ = p.free_maps()
    D = 0.1 * vol[freenode]
    ys = p.spmv(x, sigma=0.25)
    assert np.allclose(ys, oy + 0.25 * D * x, rtol=1e-13, atol=1e-22)
    z = rng.standard_normal(p.n)
    assert math.isclose(p.dot(x, z), float(np.dot(x, z)), rel_tol=1e-12, abs_tol=1e-12)


def test_spmv_long_and_short_rows(fv, oracle):
    """Rows longer than the 8-lane group (hub nodes) and single-entry rows."""
    rng = np.random.default_rng(2)
    N = 300
    hub = np.full(120, 1)
    n1 = np.r_[hub, rng.integers(2, N + 1, 500)]
    n2 = np.r_[np.arange(2, 122), rng.integers(2, N + 1, 500)]
    keep = n1 != n2
    n1, n2 = n1[keep], n2[keep]
    aol = rng.random(len(n1)) + 0.5
    K = rng.random(len(n1)) + 0.5
    dn = np.array([N], np.int64)
    dh = np.array([1.0])
    src = np.zeros(N)
    nb = np.stack([n1, n2], 1)
    for lpr_case in range(1):
        p = fv.Problem.create(nb, aol, N, dn).assemble(K, src, dh)
        oA = oracle.assembleA(n1, n2, aol, K, src, dn, dh)
        x = rng.standard_normal(p.n)
        assert np.allclose(p.spmv(x), oA.matvec(x), rtol=1e-13, atol=1e-13)


def test_chain4_solvediffusion(fv):
    """test/runtests.jl:4-16"""
    c = refcases.chain4()
    nb = np.stack([c["node1"], c["node2"]], 1)
    h, ch, A, b, freenode = fv.solvediffusion(nb, c["aol"], c["K"], c["sources"], c["dnodes"], c["dheads"])
    assert refcases.isapprox(h, c["expected"])
    assert ch.isconverged and list(freenode) == [False, True, True, False]
    assert len(ch.data["resnorm"]) == ch.iters


@pytest.mark.parametrize("ns,sigma", [((10, 10, 10), 3.0), ((24, 20, 16), 1.0), ((40, 40, 12), 0.0)])
def test_box_steady_heads_vs_direct(fv, oracle, ns, sigma):
    """examples/box_model BCs (ex.jl:29-37) scaled down; heads within 1e-8 of a direct solve."""
    coords, nb, aol, vol, K, dn, dh = _box(fv, ns, sigma=sigma)
    src = np.zeros(len(vol))
    head, ch, A, b, freenode = fv.solvediffusion(nb, aol, K, src, dn, dh, maxiter=20000, rtol=1e-13)
    assert ch.isconverged
    ohead = oracle.solvediffusion(nb[:, 0], nb[:, 1], aol, K, src, dn, dh, solver="direct")[0]
    assert relerr(head, ohead) < HEAD_RTOL
    assert head.min() >= -1e-9 and head.max() <= 1 + 1e-9  # ex_piml_data.jl:49-51
    # true residual of the returned solution
    r = A.toscipy() @ head[freenode] - b
    assert np.linalg.norm(r) / np.linalg.norm(b) < 1e-11
    # residual history is monotone-ish and ends below tolerance
    assert ch.data["resnorm"][-1] <= 1e-13 * np.linalg.norm(b) * 1.0000001


def test_steady_maxiter_reports_nonconvergence(fv):
    coords, nb, aol, vol, K, dn, dh = _box(fv, (12, 12, 12), sigma=3.0)
    head, ch, *_ = fv.solvediffusion(nb, aol, K, np.zeros(len(vol)), dn, dh, maxiter=3, rtol=1e-14, preconditioner="jacobi")
    assert not ch.isconverged and ch.iters == 3


def test_fourfractures_steady_vs_direct_and_pflotran(fv, oracle):
    d = np.load(os.path.join(GOLDEN, "fourfractures.npz"))
    nb = np.stack([d["node1"], d["node2"]], 1)
    src = np.zeros(2106)
    head, ch, A, b, fn = fv.solvediffusion(nb, d["areasoverlengths"], d["conductivities"], src, d["dirichletnodes"], d["dirichletheads"], maxiter=20000, rtol=1e-13, preconditioner="jacobi")
    assert ch.isconverged
    ohead = oracle.solvediffusion(d["node1"], d["node2"], d["areasoverlengths"], d["conductivities"], src, d["dirichletnodes"], d["dirichletheads"], solver="direct")[0]
    assert relerr(head, ohead) < HEAD_RTOL
    assert relerr(head, d["pflotran_h"]) < 1.1e-2  # loose cross-code check, SURVEY §8c item 5
    # AMG on the irregular fracture mesh (aol spread over decades): the reference's maxiter = 400 suffices
    head_a, ch_a, *_ = fv.solvediffusion(nb, d["areasoverlengths"], d["conductivities"], src, d["dirichletnodes"], d["dirichletheads"], maxiter=400, rtol=1e-13, preconditioner="amg")
    assert ch_a.isconverged and ch_a.iters < ch.iters
    assert relerr(head_a, ohead) < HEAD_RTOL


def test_fractures_example_from_the_saved_mesh_file(fv, oracle):
    """examples/fractures/ex.jl:9-14 as written: JLD.load of mesh.jld, per-fracture conductivities, solvediffusion — from the reference's own data file (tests/golden/fourfractures/mesh.jld)."""
    m = fv.meshio.read_mesh_jld(os.path.join(GOLDEN, "fourfractures", "mesh.jld"))
    N = len(m["xs"])
    src = np.zeros(N)
    kf = np.array([1e-12, 3e-12, 5e-13, 2e-12])  # one conductivity per fracture
    head, ch, A, b, fn = fv.solvediffusion(m["neighbors"], m["areasoverlengths"], kf[m["metaindex"] - 1], src, m["dirichletnodes"], m["dirichletheads"], maxiter=400, rtol=1e-13)
    assert ch.isconverged
    ohead = oracle.solvediffusion(m["node1"], m["node2"], m["areasoverlengths"], kf[m["metaindex"] - 1], src, m["dirichletnodes"], m["dirichletheads"], solver="direct")[0]
    assert relerr(head, ohead) < HEAD_RTOL
    # the same through the metaindex of the assembly (assembleA / assembleb take it: FiniteVolume.jl:75, 106)
    p = fv.Problem.create(m["neighbors"], m["areasoverlengths"], N, m["dirichletnodes"]).assemble(kf, src, m["dirichletheads"], m["metaindex"])
    assert np.array_equal(p.csc().nzval, A.nzval) and np.array_equal(p.b(), b)
    # the file's own per-connection conductivities: the run of ex.jl itself
    head, ch, *_ = fv.solvediffusion(m["neighbors"], m["areasoverlengths"], m["conductivities"], src, m["dirichletnodes"], m["dirichletheads"], maxiter=400, rtol=1e-13)
    h = fv.meshio.load_jld(os.path.join(GOLDEN, "fourfractures", "pflotran_solution.jld"), "h")
    assert ch.isconverged and relerr(head, h) < 1.1e-2


def test_fixed_steps_vs_oracle_direct(fv, oracle):
    coords, nb, aol, vol, K, dn, dh = _box(fv, (14, 12, 10), sigma=1.0)
    N = len(vol)
    src = np.zeros(N)
    src[N // 2] = -1e-4
    u0 = np.full(N, 0.5)
    us, ts = fv.backwardeulerintegrate(u0, (0.0, 500.0), 0.1, vol, nb, aol, K, src, dn, dh, stepper=fv.fixedbackwardeulerstep, dt0=100.0, rtol=1e-13)
    ous, ots = oracle.backwardeulerintegrate(u0, (0.0, 500.0), 0.1, vol, nb[:, 0], nb[:, 1], aol, K, src, dn, dh, stepper=oracle.fixedbackwardeulerstep, dt0=100.0, linearsolver=oracle.directlinearsolver)
    assert ts == ots and len(us) == 6
    for u, ou in zip(us, ous):
        assert relerr(u, ou) < HEAD_RTOL
    # the same through the all-device loop (fv_transient_run_fixed)
    p = fv.Problem.create(nb, aol, N, dn).assemble(K, src, dh)
    st = p.transient_begin(0.1, vol, u0)
    iters, info, ms = p.run_fixed(st, 100.0, 5, rtol=1e-13)
    assert relerr(st.node_values(), ous[-1]) < HEAD_RTOL and (iters > 0).all() and info.converged


def test_profile_levels_time_the_kernels_they_name_and_leave_the_heads_alone(fv):
    """fv_profile_enable: 1 = event pairs around K1, K2 / K2S and K3, 2 = around K1 only (the bench's timed region),
    0 = none; the heads of a run are the same bits at every level."""
    coords, nb, aol, vol, K, dn, dh = _box(fv, (24, 20, 16), sigma=1.0)
    N = len(vol)
    src = np.zeros(N)
    src[N // 2] = -1e-4
    heads, counts = [], []
    for level in (0, 1, 2):
        p = fv.Problem.create(nb, aol, N, dn).assemble(K, src, dh)
        st = p.transient_begin(0.1, vol, np.full(N, 0.5))
        p.profile(level)
        iters, info, ms = p.run_fixed(st, 100.0, 6, rtol=1e-10)
        counts.append((p.profile_get(), int(iters.sum())))
        heads.append(st.node_values())
        p.profile(False)
    assert np.array_equal(heads[0], heads[1]) and np.array_equal(heads[0], heads[2])
    off, full, k1 = counts
    assert all(v == (0.0, 0) for v in off[0].values())
    assert full[0]["spmv_dot"][1] >= full[1] and full[0]["update"][1] > 0 and full[0]["spmv_dot"][0] > 0
    assert k1[0]["spmv_dot"][1] == full[0]["spmv_dot"][1] and k1[0]["update"] == (0.0, 0) and k1[0]["pupdate"] == (0.0, 0)


def test_storage_term_as_codes_or_one_double_gives_the_same_bits(fv):
    """fv_tune key 35: K2S takes D = Ss * volumes as one double when it is the same on every row, as one-byte codes into a
    table when it takes a few values (regulargrid's own volumes: halved on the faces of the box, quartered on its edges),
    and as its stream otherwise: 8, 7 or 0 bytes per row fewer, the same run bit for bit."""
    coords, nb, aol, vol, K, dn, dh = _box(fv, (32, 24, 21), sigma=1.0)  # odd n
    N = len(vol)
    src = np.zeros(N)
    src[N // 2] = -1e-4
    lib = fv.load()
    assert 2 <= len(np.unique(vol)) <= 8
    volumes = {"even": np.full(N, vol.max()), "grid": vol, "many": vol * (1.0 + 1e-3 * (np.arange(N) % 50))}
    out = {}
    try:
        assert lib.fv_tune(36, 0) == 0
        for name, v in volumes.items():
            for switch in (1, 0):
                assert lib.fv_tune(35, switch) == 0
                p = fv.Problem.create(nb, aol, N, dn).assemble(K, src, dh)
                st = p.transient_begin(0.1, v, np.full(N, 0.5))
                iters, info, ms = p.run_fixed(st, 0.0009765625, 40, rtol=1e-13)  # one PCG iteration per step: every step is a K2S
                assert info.converged and (iters == 1).all()
                out[(name, switch)] = (st.node_values(), p.update_form())
    finally:
        lib.fv_tune(35, 1)
        lib.fv_tune(36, 1)
    assert [out[(name, 1)][1] for name in volumes] == [56, 57, 64] and all(out[(name, 0)][1] == 64 for name in volumes)
    for name in volumes:
        assert np.array_equal(out[(name, 1)][0], out[(name, 0)][0])


def test_zform_update_keeps_one_vector_between_one_iteration_steps(fv, oracle):
    """fv_tune key 36: between two one-iteration steps the Jacobi-scaled residual (= the first direction) is the only
    vector K2S leaves; r is taken from it (56 instead of 64 bytes per row).  Same iterations, heads equal to rounding
    and within the solver tolerance of the oracle's direct solves; a step that does not converge in its one iteration, a
    step that is converged at its set-up and a change of dt in between all find the residual they need."""
    coords, nb, aol, vol, K, dn, dh = _box(fv, (15, 13, 11), sigma=1.0)  # odd n: scalar tails
    N = len(vol)
    src = np.zeros(N)
    src[N // 2] = -1e-4
    u0 = np.full(N, 0.5)
    dt = 0.0009765625
    ous, _ = oracle.backwardeulerintegrate(u0, (0.0, 40 * dt), 0.1, vol, nb[:, 0], nb[:, 1], aol, K, src, dn, dh, stepper=oracle.fixedbackwardeulerstep, dt0=dt, linearsolver=oracle.directlinearsolver)
    lib = fv.load()
    out = {}
    try:
        for zf in (1, 0):
            assert lib.fv_tune(36, zf) == 0
            p = fv.Problem.create(nb, aol, N, dn).assemble(K, src, dh)
            st = p.transient_begin(0.1, vol, u0)
            it1, info, _ = p.run_fixed(st, dt, 40, rtol=1e-13)
            form = p.update_form()
            h1 = st.node_values()
            assert info.converged and relerr(h1, ous[-1]) < HEAD_RTOL
            it2, info, _ = p.run_fixed(st, 40.0, 5, rtol=1e-13)      # the first step speculates and misses
            it3, info, _ = p.run_fixed(st, dt, 12, rtol=1e-13)
            it4, info, _ = p.run_fixed(st, dt, 30, rtol=1e-3)        # loose tolerance: steps converged at their set-up
            assert info.converged
            out[zf] = (h1, st.node_values(), np.r_[it1, it2, it3, it4], form)
    finally:
        lib.fv_tune(36, 1)
    assert out[1][3] == 49 and out[0][3] == 57  # regulargrid volumes: D comes as codes (-7) in both
    assert (out[1][2][:40] == 1).all() and (out[1][2][40:45] > 1).all() and (out[1][2][-30:] == 0).any()
    assert np.abs(out[1][2] - out[0][2]).max() <= 1
    assert relerr(out[1][0], out[0][0]) < 1e-12 and relerr(out[1][1], out[0][1]) < 1e-9


def test_new_storage_term_on_the_same_assembly_reaches_every_copy_of_the_folded_matrix(fv):
    """fv_transient_begin with another Ss on an assembled problem: the shift folded into the solver's copies of the matrix
    (CSR values, lane-major, symmetric) and the storage codes follow — the second run equals a fresh problem's."""
    mins, maxs = [-50.0, -50.0, 0.0], [50.0, 50.0, 10.0]
    ns = [36, 182, 186]  # free rows: 34 planes of 180 x 186 = 33 480 rows, 1.14e6 in all: the symmetric (tiled) form
    coords, nb, aol, vol = fv.regulargrid(mins, maxs, ns)
    N = len(vol)
    rng = np.random.default_rng(3)
    K = np.exp(fv.nodehycos2neighborhycos(nb, np.log(1e-5) + rng.standard_normal(N), True))
    dn = np.nonzero((coords[0] == mins[0]) | (coords[0] == maxs[0]) | (coords[1] == mins[1]) | (coords[1] == maxs[1]))[0] + 1
    dh = np.full(len(dn), 1.0)
    src = np.zeros(N)
    inner = np.setdiff1d(np.arange(1, N + 1), dn)
    src[inner[len(inner) // 2] - 1] = -1e-3
    u0 = np.full(N, 1.0)

    def run(p, Ss):
        st = p.transient_begin(Ss, vol, u0)
        it, info, _ = p.run_fixed(st, 0.5, 12, rtol=1e-13)  # the same dt in every run: the copies are keyed on sigma = 1 / dt
        assert info.converged
        return st.node_values(), it.copy(), p.spmv_form()[0]

    p = fv.Problem.regulargrid(mins, maxs, ns, dn).assemble(K, src, dh)
    first = run(p, 0.1)
    second = run(p, 0.37)  # same assembly, same dt: only D changed
    fresh = run(fv.Problem.regulargrid(mins, maxs, ns, dn).assemble(K, src, dh), 0.37)
    assert second[2] == fresh[2] == 4
    assert np.array_equal(second[1], fresh[1]) and np.array_equal(second[0], fresh[0])
    assert relerr(first[0], second[0]) > 1e-9


def test_time_dependent_getb_method(fv, oracle):
    """transient.jl:165-174: caller supplies the volume-scaled b(t)."""
    coords, nb, aol, vol, K, dn, dh = _box(fv, (8, 8, 6), sigma=0.5)
    N = len(vol)
    src = np.zeros(N)
    u0 = np.zeros(N)
    ob = oracle.assembleb(nb[:, 0], nb[:, 1], aol, K, src, dn, dh)
    freenode, n2f = oracle.getfreenodes(N, dn)
    bhat = ob / (0.1 * vol[freenode])
    getb = lambda t: bhat * (1 + 0.5 * math.sin(t / 50.0))  # noqa: E731
    us, ts = fv.backwardeulerintegrate(u0, (0.0, 200.0), getb, 0.1, vol, nb, aol, K, src, dn, dh, stepper=fv.fixedbackwardeulerstep, dt0=50.0, rtol=1e-13)
    ous, ots = oracle.backwardeulerintegrate(u0, (0.0, 200.0), 0.1, vol, nb[:, 0], nb[:, 1], aol, K, src, dn, dh, getb=getb, stepper=oracle.fixedbackwardeulerstep, dt0=50.0, linearsolver=oracle.directlinearsolver)
    assert ts == ots
    assert relerr(us[-1], ous[-1]) < HEAD_RTOL


@pytest.mark.parametrize("loghyco,analytic", [(0.0, lambda t: 1 - math.exp(-t)), (1.0, lambda t: (1 - math.exp(-math.e * t)) / math.e)])
def test_onenode_logconductivity_adaptive(fv, loghyco, analytic):
    """test/onenodeadjoint.jl:29-44 with the default (adaptive) stepper"""
    c = refcases.onenode(loghyco)
    nb = np.stack([c["node1"], c["node2"]], 1)
    us, ts = fv.backwardeulerintegrate(c["u0"], c["tspan"], c["Ss"], c["volumes"], nb, c["aol"], c["K"], c["sources"], c["dnodes"], c["dheads"], None, True, atol=c["atol"], dt0=c["dt0"])
    assert ts[-1] == 1.0 and len(ts) > 5
    for u, t in zip(us, ts):
        assert u[0] == 0.0
        assert abs(u[1] - analytic(t)) <= 1e-4 * max(abs(u[1]), abs(analytic(t)))


def test_ode_diagonal_decay_generic_entry(fv):
    """test/ode.jl:8-19 through the generic integrator: sparse SPD A uploaded as CSC"""
    v = np.array([1.0, 2.0, 3.0])
    A = fv.SparseMatrixCSC(3, 3, np.array([1, 2, 3, 4]), np.array([1, 2, 3]), v.copy())
    ys, ts = fv.backwardeulerintegrate(np.ones(3), A, np.zeros(3), 0.0001, 0.0, 2.0, atol=1e-8)
    assert ts[-1] == 2.0
    for y, t in zip(ys, ts):
        assert np.allclose(y, np.exp(-v * t), atol=1e-4, rtol=0)


def test_ode_growth_generic_entry(fv):
    """test/ode.jl:21-29"""
    A = fv.SparseMatrixCSC(1, 1, np.array([1, 2]), np.array([1]), np.array([-1.0]))
    ys, ts = fv.backwardeulerintegrate(np.zeros(1), A, np.ones(1), 0.0001, 0.0, 1.0, atol=1e-8)
    for y, t in zip(ys, ts):
        assert abs(y[0] - (math.exp(t) - 1)) <= 1e-4


def test_ode_dense_user_linearsolver_seam(fv):
    """test/ode.jl:31-40: dense non-symmetric A with linearsolver = A \\ b stays on the host seam"""
    s7 = math.sqrt(7)

    def y(t, c1=1, c2=2):
        e = math.exp(-t / 4)
        a = np.array([1, 0.75]) * math.cos(s7 * t / 4) - np.array([0, -s7 / 4]) * math.sin(s7 * t / 4)
        b = np.array([1, 0.75]) * math.sin(s7 * t / 4) + np.array([0, -s7 / 4]) * math.cos(s7 * t / 4)
        return c1 * e * a + c2 * e * b

    A = -np.array([[0.5, -1.0], [1.0, -1.0]])
    calls = []

    def linearsolver(A, b, x0):
        calls.append(1)
        return np.linalg.solve(A, b)

    ys, ts = fv.backwardeulerintegrate(y(0), A, np.zeros(2), 1e-4, 0.0, 1e2, atol=1e-8, linearsolver=linearsolver)
    assert ts[-1] == 1e2 and len(calls) > 10
    for yy, t in zip(ys, ts):
        assert np.linalg.norm(yy - y(t)) <= 1e-4


def test_nonsymmetric_matrix_is_refused_by_the_device_solver(fv):
    """The device path is a conjugate gradient on the caller's CSC arrays read as CSR: a non-symmetric A (test/ode.jl:31-40's
    oscillator; transpose(D^-1 A) with unequal volumes) must not be solved as A' silently — it is refused with a message that
    points at the host linearsolver seam; symmetric-to-rounding input passes."""
    A = fv.SparseMatrixCSC.fromdense(-np.array([[0.5, -1.0], [1.0, -1.0]]))
    with pytest.raises(fv.FVError, match="not symmetric.*linearsolver"):
        fv.backwardeulerintegrate(np.ones(2), A, np.zeros(2), 1e-2, 0.0, 1.0)
    # structurally unsymmetric: an entry without a partner
    B = fv.SparseMatrixCSC(2, 2, np.array([1, 3, 4]), np.array([1, 2, 2]), np.array([2.0, -1.0, 2.0]))
    with pytest.raises(fv.FVError, match="not symmetric"):
        fv.Problem.from_csc(B)
    # D^-1 A with unequal volumes, transposed: what adjointintegrate(A, ...) would be handed for a scaled host matrix
    L = np.array([[2.0, -1.0, 0.0], [-1.0, 2.0, -1.0], [0.0, -1.0, 2.0]])
    with pytest.raises(fv.FVError, match="not symmetric"):
        fv.Problem.from_csc(fv.SparseMatrixCSC.fromdense((np.diag(1.0 / np.array([1.0, 2.0, 3.0])) @ L).T))
    C = L.copy()
    C[0, 1] *= 1 + 1e-15  # rounding-level asymmetry is what a scaled symmetric matrix looks like
    p = fv.Problem.from_csc(fv.SparseMatrixCSC.fromdense(C))
    assert p.n == 3


def test_adaptive_run_that_cannot_reach_tfinal_fails_loudly(fv):
    """fv_transient_run_adaptive stops after max_outer outer steps: it must not hand u(t < tfinal) back as u(tfinal)."""
    c = refcases.onenode(0.0)
    nb = np.stack([c["node1"], c["node2"]], 1)
    p = fv.Problem.create(nb, c["aol"], len(c["sources"]), c["dnodes"])
    p.assemble(c["K"], c["sources"], c["dheads"], None, True)
    st = p.transient_begin(c["Ss"], c["volumes"], c["u0"])
    with pytest.raises(fv.FVError, match="max_outer.*< tfinal"):
        p.run_adaptive(st, 0.0, 10.0, dt0=1e-3, atol=1e-6, max_outer=5)
    ts, nsolves, info = p.run_adaptive(st, 0.0, 0.5, dt0=1e-2, atol=1e-4)
    assert ts[-1] == 0.5


def test_nonpositive_dt_raises(fv):
    A = fv.SparseMatrixCSC(1, 1, np.array([1, 2]), np.array([1]), np.array([1.0]))
    with pytest.raises(fv.FVError, match="time step must be positive"):
        fv.backwardeulerintegrate(np.zeros(1), A, np.ones(1), 0.0, 0.0, 1.0)
    p = fv.Problem.from_csc(A)
    st = p.transient_begin(1.0, None, None)
    with pytest.raises(fv.FVError, match="time step must be positive"):
        p.step(st, st, -1.0)


def test_callback_invoked_per_step(fv):
    c = refcases.onenode(0.0)
    nb = np.stack([c["node1"], c["node2"]], 1)
    seen = []
    us, ts = fv.backwardeulerintegrate(c["u0"], (0.0, 0.5), c["Ss"], c["volumes"], nb, c["aol"], c["K"], c["sources"], c["dnodes"], c["dheads"], None, True, stepper=fv.fixedbackwardeulerstep, dt0=0.1, callback=lambda t, dt: seen.append((t, dt)))
    assert len(seen) == len(ts) - 1 and seen[0] == (0.0, 0.1)


@pytest.mark.slow
def test_theis_and_thiem(fv, oracle):
    """test/theis.jl:52-65 with the reference's settings (atol=1e-4, dt0=60, 10 days)."""
    grid = lambda a, b, n: (lambda r: (r[0], r[1][:, 0], r[1][:, 1], r[2], r[3]))(fv.regulargrid(a, b, n))  # noqa: E731
    c = refcases.theis(grid)
    nb = np.stack([c["node1"], c["node2"]], 1)
    usteady, ch, A, b, freenode = fv.solvediffusion(nb, c["aol"], c["K"], c["sources"], c["dnodes"], c["dheads"], maxiter=20000, preconditioner="jacobi")
    solver = fv.DevicePCG(rtol=1e-12, maxiter=2000)
    us, ts = fv.backwardeulerintegrate(c["u0"], c["tspan"], c["Ss"], c["volumes"], nb, c["aol"], c["K"], c["sources"], c["dnodes"], c["dheads"], atol=c["atol"], dt0=c["dt0"], linearsolver=solver)
    assert ts[-1] == c["tspan"][1]
    th = [refcases.theisdrawdown(ts[-1], r, c["T"], c["S"], c["Q"]) for r in c["rs"]]
    assert refcases.isapprox(th, -us[-1][c["goodnodes"]] + c["steadyhead"], atol=1e-4, rtol=2e-2)
    tm = [refcases.thiemdrawdown(r, c["T"], c["Q"], c["sidelength"]) for r in c["rs"]]
    assert refcases.isapprox(tm, -usteady[c["goodnodes"]] + c["steadyhead"], atol=1e-4, rtol=2e-2)
    # against the oracle with (near-)exact solves: same step sequence, heads within 1e-8
    ous, ots = oracle.backwardeulerintegrate(c["u0"], c["tspan"], c["Ss"], c["volumes"], c["node1"], c["node2"], c["aol"], c["K"], c["sources"], c["dnodes"], c["dheads"], atol=c["atol"], dt0=c["dt0"], linearsolver=oracle.tightcgsolver(1e-14))
    assert len(ts) == len(ots) and np.allclose(ts, ots, rtol=0, atol=0)
    assert relerr(us[-1], ous[-1]) < HEAD_RTOL
    assert solver.solves > 3000
    # the same adaptive run without leaving the device (fv_transient_run_adaptive): same outer time grid, same heads
    p = fv.Problem.create(nb, c["aol"], len(c["volumes"]), c["dnodes"]).assemble(c["K"], c["sources"], c["dheads"])
    st = p.transient_begin(c["Ss"], c["volumes"], c["u0"])
    ts_d, nsolves, info = p.run_adaptive(st, c["tspan"][0], c["tspan"][1], dt0=c["dt0"], atol=c["atol"], rtol=1e-12, maxiter=2000)
    assert info.converged and len(ts_d) == len(ts) and np.array_equal(ts_d, np.array(ts))
    assert nsolves == solver.solves
    assert relerr(st.node_values(), us[-1]) < 1e-10
    us2, ts2 = fv.backwardeulerintegrate(c["u0"], c["tspan"], c["Ss"], c["volumes"], nb, c["aol"], c["K"], c["sources"], c["dnodes"], c["dheads"], atol=c["atol"], dt0=c["dt0"], rtol=1e-12, maxiter=2000, keep="last")
    assert len(us2) == 2 and ts2 == ts and np.array_equal(us2[0], c["u0"]) and relerr(us2[1], us[-1]) < 1e-10
    # the steady solve with the AMG V-cycle in the preconditioner's seat, as the reference runs it (maxiter = 400)
    usteady_a, ch_a, *_ = fv.solvediffusion(nb, c["aol"], c["K"], c["sources"], c["dnodes"], c["dheads"], preconditioner="amg")
    assert ch_a.isconverged and ch_a.iters < ch.iters and ch_a.iters <= 60
    assert refcases.isapprox(tm, -usteady_a[c["goodnodes"]] + c["steadyhead"], atol=1e-4, rtol=2e-2)


# ------------------------------------------------------------------ multi-GPU plan and single-rank RCCL path
def _dist_case(fv, ns=(12, 9, 7)):
    import bench

    mins, maxs = bench.spacing_box(list(ns))
    dn, src = bench.box_setup(list(ns))
    p = fv.Problem.regulargrid(mins, maxs, list(ns), dn)
    rng = np.random.default_rng(0)
    K = 1e-5 * np.exp(rng.standard_normal(p.F))
    p.assemble(K, src, np.full(len(dn), 1e3))
    u0 = np.full(p.N, 1e3)
    u0 += rng.standard_normal(p.N)
    st = p.transient_begin(0.1, None, u0)
    return p, st


@pytest.mark.parametrize("nranks", [2, 3, 8])
def test_device_partition_plan_matches_host_planner(fv, nranks):
    from fvamd import dist, partition

    p, st = _dist_case(fv)
    A = p.csc()
    rowptr, colind = A.colptr - 1, A.rowval - 1
    u_global = st.free_values()
    for rank in range(nranks):
        blk = dist.RowBlock(p, nranks, rank)
        want = partition.plan(rowptr, colind, nranks, rank)
        got = blk.plan()
        assert (blk.lo, blk.hi) == (want["lo"], want["hi"])
        assert np.array_equal(got["rowptr"], want["rowptr"])
        assert np.array_equal(got["colind"], want["colind"])
        assert np.array_equal(got["halo_cols"], want["halo_cols"])
        assert np.array_equal(got["recv_counts"], want["recv_counts"])
        assert got["send_counts"].tolist() == [len(s) for s in want["send_idx"]]
        assert np.array_equal(got["send_idx"], np.concatenate(want["send_idx"]) if blk.nsend else np.empty(0, np.int64))
        bgroups = np.unique(np.nonzero(want["boundary_rows"])[0] // 64)
        assert np.array_equal(got["groups_bnd"], bgroups)
        assert blk.n_int + blk.n_bnd == (blk.nloc + 63) // 64
        assert np.array_equal(blk.state(), u_global[blk.lo : blk.hi])


def test_single_rank_rccl_path_matches_plain_run(fv):
    """nranks = 1 through the distributed driver (RCCL communicator of one rank:
    all-reduce + empty halo) must reproduce the single-GPU loop."""
    from fvamd import dist

    p, st = _dist_case(fv, (16, 12, 10))
    ctx = p.ctx
    dist.comm_init(ctx, 1, 0, dist.comm_unique_id())
    blk = dist.RowBlock(p, 1, 0)
    rng = np.random.default_rng(3)
    x = rng.standard_normal(p.n)
    assert np.allclose(blk.spmv(x, 0.01), p.spmv(x, 0.01), rtol=1e-14, atol=1e-20)
    # 40 steps: crosses a refresh of the carried residual (every 32 steps) in both drivers
    it_d, info_d, _ = blk.run_fixed(3600.0, 40, 1e-12)
    it_s, info_s, _ = p.run_fixed(st, 3600.0, 40, 1e-12)
    assert info_d.converged and info_s.converged
    assert np.array_equal(it_d, it_s)
    assert relerr(blk.state(), st.free_values()) < 1e-12
    # one-iteration regime (tiny dt): both drivers prepare step k+1 inside step k's vector update; then a large step,
    # where that speculation misses
    for dt, nsteps in ((2.0**-10, 40), (3600.0, 3)):
        it_d, info_d, _ = blk.run_fixed(dt, nsteps, 1e-12)
        it_s, info_s, _ = p.run_fixed(st, dt, nsteps, 1e-12)
        assert info_d.converged and info_s.converged
        assert np.array_equal(it_d, it_s), (dt, it_d, it_s)
        assert relerr(blk.state(), st.free_values()) < 1e-12
        if dt < 1:
            assert (it_s == 1).all()
    fv.load().fv_comm_destroy(ctx.handle)


# ------------------------------------------------------------------ adjoint on the device operator
def test_onenode_adjoint_lambda(fv):
    """test/onenodeadjoint.jl:51-66: adjointintegrate re-enters the hot path with transpose(A)."""
    from scipy.integrate import quad

    sigma2 = 0.01**2
    c = refcases.onenode(0.0)
    nb = np.stack([c["node1"], c["node2"]], 1)
    uobs = lambda t: 1 - math.exp(-t)  # noqa: E731  (observations: analytic solution for loghyco = 0)
    # forward model at the perturbed parameters (loghyco + 1)
    us, ts = fv.backwardeulerintegrate(c["u0"], c["tspan"], c["Ss"], c["volumes"], nb, c["aol"], c["K"] + 1, c["sources"], c["dnodes"], c["dheads"], None, True, atol=c["atol"], dt0=c["dt0"])
    uc = fv.getcontinuoussolution(us, ts)
    dgdu = lambda t: np.array([2 * sigma2 * (uc(t)[1] - uobs(t))])  # noqa: E731  (transientadjointutils.jl:13-21, one observed free node)
    lambdas, ts_l = fv.adjointintegrate(dgdu, c["tspan"], c["Ss"], c["volumes"], nb, c["aol"], c["K"] + 1, c["sources"], c["dnodes"], c["dheads"], None, True, atol=c["atol"], dt0=c["dt0"])
    u_init = lambda s: (1 - math.exp(-math.e * s)) / math.e  # noqa: E731
    f = lambda s: 2 * sigma2 * (u_init(s) - uobs(s))  # noqa: E731
    T = c["tspan"][1]
    gamma = lambda t: math.exp(-math.e * t) * quad(lambda s: math.exp(math.e * s) * f(T - s), 0, t)[0]  # noqa: E731
    assert ts_l[0] == 0.0 and ts_l[-1] == T and len(ts_l) > 5
    for lam, t in zip(lambdas, ts_l):
        want = gamma(T - t)
        assert abs(lam[0] - want) <= max(1e-7, 1e-4 * max(abs(lam[0]), abs(want)))


def test_adjoint_step_matches_transposed_scaled_operator(fv, oracle):
    """FV_STEP_ADJOINT solves (I/dt + A D^-1) g+ = g/dt + bhat: checked against a dense solve with unequal volumes."""
    rng = np.random.default_rng(7)
    ns = (5, 4, 3)
    coords, nb, aol, vol = fv.regulargrid([0.0, 0.0, 0.0], [4.0, 3.0, 2.0], list(ns))
    K = rng.random(len(aol)) + 0.5
    dn = np.array([1, len(vol)], np.int64)
    dh = np.array([1.0, 0.0])
    vol = vol * (1 + rng.random(len(vol)))  # make D non-uniform so that A D^-1 is NOT symmetric
    p = fv.Problem.create(nb, aol, len(vol), dn).assemble(K, np.zeros(len(vol)), dh)
    st = p.transient_begin(0.3, vol, None)
    g0 = rng.standard_normal(p.n)
    bhat = rng.standard_normal(p.n)
    st.set_free(g0)
    dst = p.new_state()
    dt = 0.7
    p.step(st, dst, dt, bhat, mode=1, rtol=1e-14, maxiter=500)
    A = p.csc().toscipy().toarray()
    freenode, _ = p.free_maps()
    D = 0.3 * vol[freenode]
    M = np.eye(p.n) / dt + A @ np.diag(1 / D)  # transpose of (I/dt + D^-1 A)
    want = np.linalg.solve(M, g0 / dt + bhat)
    assert relerr(dst.free_values(), want) < 1e-11
    # and the forward mode against (I/dt + D^-1 A) u+ = u/dt + bhat
    p.step(st, dst, dt, bhat, mode=0, rtol=1e-14, maxiter=500)
    Mf = np.eye(p.n) / dt + np.diag(1 / D) @ A
    assert relerr(dst.free_values(), np.linalg.solve(Mf, g0 / dt + bhat)) < 1e-11


@pytest.mark.parametrize("ns,nranks", [((12, 9, 7), 3), ((40, 30, 20), 2), ((40, 30, 20), 5)])
def test_virtual_ranks_interior_and_boundary_passes(fv, ns, nranks):
    """Every rank of an N-way partition rehearsed on ONE GPU: the block's interior + boundary SpMV passes
    (sliced-DIA and CSR subsets) with the halo values a peer would have sent == the global SpMV's slice."""
    from fvamd import dist

    p, st = _dist_case(fv, ns)
    rng = np.random.default_rng(11)
    x = rng.standard_normal(p.n)
    for sigma in (0.0, 0.37):
        y_global = p.spmv(x, sigma)
        for rank in range(nranks):
            blk = dist.RowBlock(p, nranks, rank)
            plan = blk.plan()
            y = blk.spmv_halo(x[blk.lo : blk.hi], x[plan["halo_cols"]], sigma)
            assert np.allclose(y, y_global[blk.lo : blk.hi], rtol=1e-13, atol=1e-18), (rank, sigma)
            assert blk.n_bnd > 0 or nranks == 1


def test_onenode_adjoint_gradient_vs_finite_differences(fv):
    """test/onenodeadjoint.jl:51-75: getadjointfunctions + adjointintegrate + gradientintegrate vs central FD of G."""
    sigma = lambda i, t: 0.01  # noqa: E731
    c = refcases.onenode(0.0)
    nb = np.stack([c["node1"], c["node2"]], 1)
    kw = dict(atol=c["atol"], dt0=c["dt0"])
    us, ts = fv.backwardeulerintegrate(c["u0"], c["tspan"], c["Ss"], c["volumes"], nb, c["aol"], c["K"], c["sources"], c["dnodes"], c["dheads"], None, True, **kw)
    uobs = fv.getcontinuoussolution(us, ts)
    p0 = np.r_[c["K"] + 1, c["sources"], c["dheads"]]
    us_i, ts_i = fv.backwardeulerintegrate(c["u0"], c["tspan"], c["Ss"], c["volumes"], nb, c["aol"], c["K"] + 1, c["sources"], c["dnodes"], c["dheads"], None, True, **kw)
    uc_init = fv.getcontinuoussolution(us_i, ts_i)
    freenodes, n2f = fv.getfreenodes(2, c["dnodes"])
    obsfreenodes = [int(n2f[1])]
    g, dgdu, dfdp, dgdp, du0dp, G = fv.getadjointfunctions(sigma, obsfreenodes, uobs, c["u0"], c["tspan"], c["Ss"], c["volumes"], nb, c["aol"], c["K"], c["sources"], c["dnodes"], c["dheads"], None, True, **kw)
    assert g(uobs, 0.5) == 0
    shifted = lambda t: uobs(t) + 1  # noqa: E731
    assert dgdu(shifted, 0.5).tolist() == [2 * 0.01**2]
    lambdas, ts_l = fv.adjointintegrate(lambda t: dgdu(uc_init, t), c["tspan"], c["Ss"], c["volumes"], nb, c["aol"], c["K"] + 1, c["sources"], c["dnodes"], c["dheads"], None, True, **kw)
    lambdac = fv.getcontinuoussolution(lambdas, ts_l)
    dGdp = fv.gradientintegrate(lambdac, du0dp, lambda t: dgdp(uc_init, t, p0), lambda t: dfdp(uc_init, t, p0), c["tspan"], maxevals=300, order=21)
    # the same integral through integratedfdplambda: complete=True integrates the whole Jacobian; the default keeps the
    # reference's hand-unrolled terms (FiniteVolume.jl:271-377), which agree on the source and Dirichlet-head entries
    uc_init2 = fv.getcontinuoussolution(us_i, ts_i, 2)
    idl = fv.integratedfdplambda(uc_init2, p0, lambdas, ts_l, c["tspan"], c["Ss"], c["volumes"], nb, c["aol"], c["K"], c["sources"], c["dnodes"], c["dheads"], None, True, complete=True)
    dGdp2 = fv.gradientintegrate(lambdas[0], du0dp, lambda t: dgdp(uc_init, t, p0), idl, c["tspan"])
    # (complete=True ran on the device — fv_param_gradient_integral; the host quadrature of the same integral agrees)
    idl_host = fv.integratedfdplambda(uc_init2, p0, lambdas, ts_l, c["tspan"], c["Ss"], c["volumes"], nb, c["aol"], c["K"], c["sources"], c["dnodes"], c["dheads"], None, True, complete=True, device=False)
    assert np.allclose(idl, idl_host, rtol=1e-8, atol=1e-12 * np.abs(idl_host).max())
    idl_ref = fv.integratedfdplambda(uc_init2, p0, lambdas, ts_l, c["tspan"], c["Ss"], c["volumes"], nb, c["aol"], c["K"], c["sources"], c["dnodes"], c["dheads"], None, True)
    assert np.allclose(idl_ref[[2, 3]], idl[[2, 3]], rtol=1e-6, atol=0)
    # ... and carry the u term of the conductivity entry with the opposite sign (dhead = 0 here: that is the whole entry)
    assert abs(idl_ref[0] + idl[0]) <= 1e-6 * abs(idl[0])
    deltap = 1e-6
    for i in (0, 2, 3):  # importantindices = [1, 3, 4] of the reference (1-based)
        pp, pm = p0.copy(), p0.copy()
        pp[i] += deltap
        pm[i] -= deltap
        fd = (G(pp) - G(pm)) / (2 * deltap)
        assert abs(fd - dGdp[i]) <= 1e-2 * max(abs(fd), abs(dGdp[i])), (i, fd, dGdp[i])
        assert abs(dGdp2[i] - dGdp[i]) <= 1e-3 * abs(dGdp[i]) + 1e-14


def test_transientadjoint_heterogeneous_field_many_observations(fv):
    """test/transientadjoint.jl at a reduced size (31 x 31 x 2; the Matern field of GaussianRandomFields, not available,
    replaced by smoothed noise): forward run in a heterogeneous log-conductivity field, noisy observations at 60 free
    nodes, adjoint run forced by dgdu, gradient by the device integral; the reference prints the ratio to one-sided
    finite differences of G for the 20 largest entries without asserting it — here the 6 largest are asserted."""
    atol, steadyhead, side, thick = 1e-2, 1e3, 50.0, 10.0
    mins, maxs, ns = [-side, -side, 0.0], [side, side, thick], [31, 31, 2]
    meanloghyco, Q, Ss, sig = math.log(1e-5), 1e-3, 0.1, 0.03
    coords, neighbors, aol, volumes = fv.regulargrid(mins, maxs, ns)
    F, N = len(aol), coords.shape[1]
    rng = np.random.default_rng(0)
    field = rng.standard_normal(ns)
    for _ in range(3):  # smoothed noise with a correlation length of a few cells, unit variance
        for ax in range(2):
            field = (np.roll(field, 1, ax) + field + np.roll(field, -1, ax)) / 3.0
    nodeloghycos = meanloghyco + (field / field.std()).ravel()
    loghycos = fv.nodehycos2neighborhycos(neighbors, nodeloghycos, True)
    center = np.nonzero((coords[0] == 0) & (coords[1] == 0))[0]
    sources = np.zeros(N)
    sources[center] = -Q / (2 * len(center) - 2)
    dmask = np.hypot(coords[0], coords[1]) - side >= 0
    dnodes = np.nonzero(dmask)[0] + 1
    dheads = np.full(len(dnodes), steadyhead)
    u0 = np.full(N, steadyhead)
    tspan = (0.0, 60 * 60 * 24 * 1e1)
    meta = lambda i: i  # noqa: E731
    kw = dict(atol=atol, dt0=60.0)
    us, ts = fv.backwardeulerintegrate(u0, tspan, Ss, volumes, neighbors, aol, loghycos, sources, dnodes, dheads, meta, True, **kw)
    uobs = fv.getcontinuoussolution([u + sig * rng.standard_normal(N) for u in us], ts)
    K0 = np.full(F, meanloghyco)
    p0 = np.r_[K0, sources, dheads]
    us_i, ts_i = fv.backwardeulerintegrate(u0, tspan, Ss, volumes, neighbors, aol, K0, sources, dnodes, dheads, meta, True, **kw)
    uc_init = fv.getcontinuoussolution(us_i, ts_i)
    freenodes, n2f = fv.getfreenodes(N, dnodes)
    nfree = int(freenodes.sum())
    obsfreenodes = (rng.permutation(nfree)[:60] + 1).tolist()
    g, dgdu, dfdp, dgdp, du0dp, G = fv.getadjointfunctions(lambda i, t: sig, obsfreenodes, uobs, u0, tspan, Ss, volumes, neighbors, aol, K0, sources, dnodes, dheads, meta, True, **kw)
    assert g(uobs, 0.5 * tspan[1]) == 0
    want = np.zeros(nfree)
    want[np.array(obsfreenodes) - 1] = 2 * sig**2
    assert np.allclose(dgdu(lambda t: uobs(t) + 1, 0.5 * tspan[1]), want, rtol=1e-9, atol=0)  # (u + 1 - u rounds at 1e3)
    lambdas, ts_l = fv.adjointintegrate(lambda t: dgdu(uc_init, t), tspan, Ss, volumes, neighbors, aol, K0, sources, dnodes, dheads, meta, True, **kw)
    integral = fv.devicegradientintegral(uc_init, lambdas, ts_l, tspan, Ss, volumes, neighbors, aol, K0, sources, dnodes, dheads, meta, True, scale="storage")
    dGdp = fv.gradientintegrate(lambdas[0], du0dp, lambda t: dgdp(uc_init, t, p0), integral, tspan)
    assert dGdp.shape == p0.shape and np.isfinite(dGdp).all()
    # measured ratios adjoint / finite difference: 0.977 .. 1.030 (the stepper's atol is 1e-2 here, as in the reference's test)
    G0 = G(p0)
    deltap = 1e-6
    for i in np.argsort(-np.abs(dGdp), kind="stable")[:6]:
        pp = p0.copy()
        pp[i] += deltap
        x1 = (G(pp) - G0) / deltap
        print("entry", int(i), "one-sided FD", x1, "adjoint", dGdp[i], "ratio", dGdp[i] / x1)
        assert abs(x1 - dGdp[i]) <= 5e-2 * max(abs(x1), abs(dGdp[i])), (int(i), x1, dGdp[i])


@pytest.mark.parametrize("logk", [True, False])
def test_device_gradient_integral_vs_host_simpson_of_the_jacobians(fv, logk):
    """fv_param_gradient_integral (one thread per face walking the knots) against Simpson's rule — exact for the
    piecewise-quadratic integrand — over the host-built Jacobians b_p - A_px: box with a log-normal field, a
    metaindex sharing parameters between faces, interior Dirichlet cells, a repeated Dirichlet node, a self-loop face,
    unevenly spaced knots that differ between u and lambda, a tspan cutting through intervals."""
    from fvamd import adjoint

    coords, nb, aol, vol, K, dn, dh = _box(fv, (9, 7, 6), sigma=1.0)
    N, F = len(vol), len(aol)
    rng = np.random.default_rng(3)
    nb = np.vstack([nb, [[5, 5]]])  # a face from a node to itself: no contribution
    aol = np.r_[aol, 1.0]
    F += 1
    dn = np.r_[dn, [N // 2, N // 2 + 1, dn[0]]]  # interior Dirichlet cells; dn[0] repeated: the last head wins
    dh = np.r_[dh, [0.3, -0.2, 7.0]]
    nK = 11
    mi = rng.integers(1, nK + 1, F)
    Kp = rng.standard_normal(nK) * 0.5 + (np.log(1e-3) if logk else 0.0)
    if not logk:
        Kp = np.exp(Kp) * 1e-3
    src = rng.standard_normal(N) * 1e-4
    Ss = 0.2
    freenode, n2f = fv.getfreenodes(N, dn)
    src[~freenode] = 0.0
    nfree = int(freenode.sum())
    ts_u = np.cumsum(np.r_[0.0, rng.uniform(0.5, 2.0, 7)])
    ts_l = np.cumsum(np.r_[0.0, rng.uniform(0.3, 3.0, 5)])
    ts_l *= ts_u[-1] / ts_l[-1]
    us = [rng.standard_normal(N) for _ in ts_u]
    lambdas = [rng.standard_normal(nfree) for _ in ts_l]
    uc = fv.getcontinuoussolution(us, ts_u)
    lam = adjoint.LinearInterpolant(lambdas, ts_l)
    tspan = (0.37, 0.93 * ts_u[-1])
    for scale, w in (("reference", 1.0 / (Ss * vol[:nfree])), ("storage", 1.0 / (Ss * vol[freenode])), (None, np.ones(nfree))):
        got = fv.devicegradientintegral(uc, lambdas, ts_l, tspan, Ss, vol, nb, aol, Kp, src, dn, dh, mi, logk, scale=scale)

        def f(t):
            M, _, _ = adjoint._parameter_jacobians(uc(t)[freenode], nb, aol, Kp, src, dn, dh, mi, logk)
            return M @ (lam(t) * w)

        knots = np.unique(np.r_[ts_u, ts_l, tspan])
        knots = knots[(knots >= tspan[0]) & (knots <= tspan[1])]
        want = sum((b - a) / 6.0 * (f(a) + 4.0 * f(0.5 * (a + b)) + f(b)) for a, b in zip(knots[:-1], knots[1:]))
        assert got.shape == want.shape == (nK + N + len(dh),)
        assert np.abs(got - want).max() <= 1e-12 * np.abs(want).max(), scale
        assert np.count_nonzero(got[:nK]) == nK and np.count_nonzero(got[nK + N :]) > 0
    # many knots: several passes over the time series inside the library (chunks share a knot), same answer
    p = fv.Problem.create(nb, aol, N, dn).assemble(Kp, src, dh, mi, logk)
    kn = np.linspace(0.0, 5.0, 41)
    X, L = rng.standard_normal((41, nfree)), rng.standard_normal((41, nfree))
    whole = p.param_gradient_integral(kn, X, L, False, logk)
    parts = [p.param_gradient_integral(kn[a : b + 1], X[a : b + 1], L[a : b + 1], False, logk) for a, b in ((0, 13), (13, 14), (14, 40))]
    fv.load().fv_tune(20, 5)  # ... and with the library itself taking 5 knots per pass (10 passes)
    try:
        passes = p.param_gradient_integral(kn, X, L, False, logk)
    finally:
        fv.load().fv_tune(20, 0)
    for k in range(3):
        assert np.allclose(whole[k], sum(part[k] for part in parts), rtol=1e-12, atol=1e-14 * np.abs(whole[k]).max())
        assert np.allclose(whole[k], passes[k], rtol=1e-12, atol=1e-14 * np.abs(whole[k]).max())
    with pytest.raises(fv.FVError, match="needs fv_transient_begin"):
        p.param_gradient_integral(kn, X, L, True, logk)
    with pytest.raises(fv.FVError, match="knots must not decrease"):
        p.param_gradient_integral(kn[::-1].copy(), X, L, False, logk)
    p.transient_begin(Ss, vol, np.zeros(N))
    scaled = p.param_gradient_integral(kn, X, L, True, logk)
    direct = p.param_gradient_integral(kn, X, L / (Ss * vol[freenode]), False, logk)
    for k in range(3):
        assert np.allclose(scaled[k], direct[k], rtol=1e-12, atol=1e-14 * np.abs(direct[k]).max())


def test_parameter_jacobians_vs_finite_differences_of_the_oracle_assembly(fv, oracle):
    """b_p - A_px (the LinearAdjoints-generated pair called at transientadjointutils.jl:27-28) against central
    finite differences of the oracle's assembleb - assembleA*x, with a metaindex, both conductivity forms."""
    import sys

    adj = sys.modules[fv.__name__ + ".adjoint"]
    rng = np.random.default_rng(7)
    # 3 x 4 lattice of nodes, faces along both directions, three conductivity zones, two Dirichlet nodes
    n1, n2 = [], []
    for i in range(3):
        for j in range(4):
            k = i * 4 + j + 1
            if j < 3:
                n1.append(k), n2.append(k + 1)
            if i < 2:
                n1.append(k), n2.append(k + 4)
    n1, n2 = np.array(n1), np.array(n2)
    F, N = len(n1), 12
    aol = rng.uniform(0.5, 2.0, F)
    meta = rng.integers(1, 4, F)
    dn = np.array([1, 7])
    dh = np.array([3.0, -1.5])
    src = rng.normal(size=N)
    src[dn - 1] = 0.0  # FiniteVolume.jl:26: no source at a Dirichlet node
    x = rng.normal(size=N - 2)
    nb = np.stack([n1, n2], 1)
    for logk in (False, True):
        K = rng.uniform(0.2, 1.5, 3)
        p0 = np.r_[K, src, dh]

        def resid(p):
            Kp, sp_, dp = p[:3], p[3 : 3 + N], p[3 + N :]
            A = oracle.assembleA(n1, n2, aol, Kp, sp_, dn, dp, meta, logk).toscipy()
            return oracle.assembleb(n1, n2, aol, Kp, sp_, dn, dp, meta, logk) - A @ x

        M, freenode, n2f = adj._parameter_jacobians(x, nb, aol, K, src, dn, dh, meta, logk)
        M = M.toarray()
        assert M.shape == (3 + N + 2, N - 2)
        h = 1e-6
        for i in range(len(p0)):
            if i - 3 in dn - 1:  # a Dirichlet node's source slot cannot be perturbed (it must stay zero)
                continue
            pp, pm = p0.copy(), p0.copy()
            pp[i] += h
            pm[i] -= h
            fd = (resid(pp) - resid(pm)) / (2 * h)
            assert np.allclose(M[i], fd, rtol=1e-7, atol=1e-8), (logk, i)
        # Dirichlet nodes' own source entries have no effect
        assert not M[3 + dn - 1].any()


def test_theisadjoint_gradient_vs_finite_differences(fv):
    """test/theisadjoint.jl:12-84: 25 x 25 x 2 grid, forward x2 + adjoint on the device stepper, integratedfdplambda,
    gradient against central finite differences of G on the 20 largest entries (rtol 1e-3 there and here)."""
    atol, steadyhead, side, thick = 1e-4, 0.0, 50.0, 10.0
    mins, maxs, ns = [-side, -side, 0.0], [side, side, thick], [25, 25, 2]
    meanloghyco, Q, Ss = math.log(1e-5), 1e-3, 0.1
    sigma = lambda i, t: 0.03  # noqa: E731
    coords, neighbors, aol, volumes = fv.regulargrid(mins, maxs, ns)
    F, N = len(aol), coords.shape[1]
    loghycos = np.full(F, meanloghyco + 1)
    center = np.nonzero((coords[0] == 0) & (coords[1] == 0))[0]
    assert len(center) == 2
    sources = np.zeros(N)
    sources[center] = -2 * Q / (2 * len(center) - 2)
    sources[center[0]] = sources[center[-1]] = -Q / (2 * len(center) - 2)
    dmask = np.hypot(coords[0], coords[1]) - side >= 0
    dnodes = np.nonzero(dmask)[0] + 1
    dheads = np.full(len(dnodes), steadyhead)
    u0 = np.full(N, steadyhead)
    tspan = (0.0, 60 * 60 * 24 * 1e1)
    meta = lambda i: i  # noqa: E731
    kw = dict(atol=atol, dt0=60.0)
    us, ts = fv.backwardeulerintegrate(u0, tspan, Ss, volumes, neighbors, aol, loghycos, sources, dnodes, dheads, meta, True, **kw)
    uobs = fv.getcontinuoussolution(us, ts)
    K0 = np.full(F, meanloghyco)
    p0 = np.r_[K0, sources, dheads]
    us_i, ts_i = fv.backwardeulerintegrate(u0, tspan, Ss, volumes, neighbors, aol, K0, sources, dnodes, dheads, meta, True, **kw)
    uc_init = fv.getcontinuoussolution(us_i, ts_i)
    uc_init2 = fv.getcontinuoussolution(us_i, ts_i, 2)
    freenodes, n2f = fv.getfreenodes(N, dnodes)
    obsfreenodes = [int(n2f[i]) for i in center]
    g, dgdu, dfdp, dgdp, du0dp, G = fv.getadjointfunctions(sigma, obsfreenodes, uobs, u0, tspan, Ss, volumes, neighbors, aol, K0, sources, dnodes, dheads, meta, True, **kw)
    lambdas, ts_l = fv.adjointintegrate(lambda t: dgdu(uc_init, t), tspan, Ss, volumes, neighbors, aol, K0, sources, dnodes, dheads, meta, True, **kw)
    idl = fv.integratedfdplambda(uc_init2, p0, lambdas, ts_l, tspan, Ss, volumes, neighbors, aol, K0, sources, dnodes, dheads, meta, True)
    dGdp = fv.gradientintegrate(lambdas[0], du0dp, lambda t: dgdp(uc_init, t, p0), idl, tspan, maxevals=300, order=21)
    assert dGdp.shape == (F + N + len(dnodes),)
    important = np.argsort(-np.abs(dGdp), kind="stable")[:20]
    deltap = 1e-4
    for i in important:
        pp, pm = p0.copy(), p0.copy()
        pp[i] += deltap
        pm[i] -= deltap
        x1 = (G(pp) - G(pm)) / (2 * deltap)
        assert abs(x1 - dGdp[i]) <= 1e-3 * max(abs(x1), abs(dGdp[i])), (int(i), x1, dGdp[i])
    # the whole Jacobian integrated on the device (fv_param_gradient_integral): the same source and head entries, and
    # conductivity entries that hold up against finite differences too (the hand-unrolled routine above, followed term
    # by term from the reference, has no free|free face terms, so its conductivity entries do not)
    idl_dev = fv.integratedfdplambda(uc_init2, p0, lambdas, ts_l, tspan, Ss, volumes, neighbors, aol, K0, sources, dnodes, dheads, meta, True, complete=True)
    dGdp_dev = fv.gradientintegrate(lambdas[0], du0dp, lambda t: dgdp(uc_init, t, p0), idl_dev, tspan)
    assert np.allclose(dGdp_dev[F : F + N], dGdp[F : F + N], rtol=1e-9, atol=1e-12 * np.abs(dGdp).max())
    for i in np.r_[np.argsort(-np.abs(dGdp_dev[:F]), kind="stable")[:4], F + N + np.argsort(-np.abs(dGdp_dev[F + N :]), kind="stable")[:2]]:
        pp, pm = p0.copy(), p0.copy()
        pp[i] += deltap
        pm[i] -= deltap
        x1 = (G(pp) - G(pm)) / (2 * deltap)
        print("conductivity / head entry", int(i), "FD", x1, "device integral", dGdp_dev[i], "hand-unrolled", dGdp[i])
        # measured: conductivities 1.7e-4 apart; the head entry 2.4 % (a head change is a step at t = 0, resolved to the
        # stepper's atol), where the hand-unrolled value is a factor 2 off (volumes taken by free index)
        assert abs(x1 - dGdp_dev[i]) <= (2e-3 if i < F else 5e-2) * max(abs(x1), abs(dGdp_dev[i])), (int(i), x1, dGdp_dev[i])


@pytest.mark.parametrize("schedule", ["several_iterations", "one_iteration", "one_then_several"])
def test_run_fixed_residual_carry_matches_fresh_residuals_and_oracle(fv, oracle, schedule):
    """fv_transient_run_fixed: carrying the residual from step to step (fv_tune key 7: refresh period) and preparing the
    next step inside a one-iteration step's vector update (key 8) give the heads of the run that recomputes b' - A u
    every step, and the oracle's (direct solves), over two calls of 30 + 40 steps that cross refresh boundaries.
    dt = 40: several PCG iterations per step (never speculates); dt = 2^-10: one per step (the bench's regime, every step
    speculates); 2^-10 then 40: the first step of the second call speculates and misses (pcg_pupdate_kernel<true>)."""
    coords, nb, aol, vol, K, dn, dh = _box(fv, (15, 13, 11), sigma=1.0)  # odd n: exercises the scalar tails
    N = len(vol)
    src = np.zeros(N)
    src[N // 2] = -1e-4
    u0 = np.full(N, 0.5)
    dts = {"several_iterations": (40.0, 40.0), "one_iteration": (0.0009765625, 0.0009765625), "one_then_several": (0.0009765625, 40.0)}[schedule]
    rtol = 1e-13
    ous1, _ = oracle.backwardeulerintegrate(u0, (0.0, 30 * dts[0]), 0.1, vol, nb[:, 0], nb[:, 1], aol, K, src, dn, dh, stepper=oracle.fixedbackwardeulerstep, dt0=dts[0], linearsolver=oracle.directlinearsolver)
    ous2, _ = oracle.backwardeulerintegrate(ous1[-1], (0.0, 40 * dts[1]), 0.1, vol, nb[:, 0], nb[:, 1], aol, K, src, dn, dh, stepper=oracle.fixedbackwardeulerstep, dt0=dts[1], linearsolver=oracle.directlinearsolver)
    lib = fv.load()
    heads, its = {}, {}
    try:
        for refresh, spec in ((0, 0), (32, 1), (32, 0), (5, 1)):
            assert lib.fv_tune(7, refresh) == 0 and lib.fv_tune(8, spec) == 0
            p = fv.Problem.create(nb, aol, N, dn).assemble(K, src, dh)
            st = p.transient_begin(0.1, vol, u0)
            a, info, _ = p.run_fixed(st, dts[0], 30, rtol=rtol)
            assert info.converged and relerr(st.node_values(), ous1[-1]) < HEAD_RTOL
            b, info, _ = p.run_fixed(st, dts[1], 40, rtol=rtol)  # the second call starts from a fresh residual again
            assert info.converged and relerr(st.node_values(), ous2[-1]) < HEAD_RTOL
            heads[(refresh, spec)], its[(refresh, spec)] = st.node_values(), np.r_[a, b]
            # another state of the same problem is not disturbed by the hidden ping-pong vector
            st2 = p.new_state()
            st2.set_nodes(u0)
            p.run_fixed(st2, dts[0], 3, rtol=rtol)
            assert relerr(st2.node_values(), ous1[3]) < HEAD_RTOL
    finally:
        lib.fv_tune(7, 128)
        lib.fv_tune(8, 1)
    base = its[(0, 0)]
    if schedule == "several_iterations":
        assert (base > 1).all()
    elif schedule == "one_iteration":
        assert (base == 1).all()
    else:
        assert (base[:30] == 1).all() and (base[30:] > 1).all()
    for key in ((32, 1), (32, 0), (5, 1)):
        assert relerr(heads[key], heads[(0, 0)]) < 1e-11
        assert np.abs(its[key] - base).max() <= 1
    if schedule == "one_iteration":
        # bursts of unpolled steps (fv_tune key 13): off, short, long; and a chain broken on the device (fault
        # injection, key 14: the 4th step of the first burst is treated as not converged and resumed by the host)
        try:
            for chain, brk in ((0, -1), (3, -1), (16, -1), (8, 3), (8, 0), (8, 7)):
                assert lib.fv_tune(13, chain) == 0 and lib.fv_tune(14, brk) == 0
                p = fv.Problem.create(nb, aol, N, dn).assemble(K, src, dh)
                st = p.transient_begin(0.1, vol, u0)
                a, info, _ = p.run_fixed(st, dts[0], 30, rtol=rtol)
                b, info, _ = p.run_fixed(st, dts[1], 40, rtol=rtol)
                assert info.converged and relerr(st.node_values(), ous2[-1]) < HEAD_RTOL, (chain, brk)
                assert relerr(st.node_values(), heads[(0, 0)]) < 1e-11
                it = np.r_[a, b]
                if brk < 0:
                    assert (it == 1).all()
                else:  # every burst breaks at that index: some steps take a second iteration, nothing else changes
                    assert set(np.unique(it)) == {1, 2} and 2 <= (it == 2).sum() <= 35, (chain, brk, it)
        finally:
            lib.fv_tune(13, 8)
            lib.fv_tune(14, -1)


@pytest.mark.parametrize("dt", [0.0009765625, 40.0])
def test_run_fixed_in_chunks_goes_on_where_the_previous_call_stopped(fv, dt):
    """Stepping in chunks (to look at the state in between, as a driver that stores every k-th step does) costs what one
    long call costs: a call picks up the residual, the prepared set-up and the refresh count the previous call on the same
    slot left (fv_tune key 33) — the same iterations and bit for bit the same state as one call of 70 steps — unless
    something else touched the state or solved in between, in which case it starts afresh (and still lands on the same
    heads to the solver tolerance)."""
    coords, nb, aol, vol, K, dn, dh = _box(fv, (15, 13, 11), sigma=1.0)
    N = len(vol)
    src = np.zeros(N)
    src[N // 2] = -1e-4
    u0 = np.full(N, 0.5)
    rtol = 1e-13

    def run(chunks, between=None, resume=1):
        fv.load().fv_tune(33, resume)
        try:
            p = fv.Problem.create(nb, aol, N, dn).assemble(K, src, dh)
            st = p.transient_begin(0.1, vol, u0)
            its = []
            for k, n in enumerate(chunks):
                it, info, _ = p.run_fixed(st, dt, n, rtol=rtol)
                assert info.converged
                its.append(it.copy())
                if between is not None and k + 1 < len(chunks):
                    between(p, st)
            return st.node_values(), np.concatenate(its)
        finally:
            fv.load().fv_tune(33, 1)

    whole, its_whole = run([70])
    chunked, its_chunked = run([7, 2, 29, 32], between=lambda p, st: st.node_values())  # reading the state does not disturb anything
    assert np.array_equal(its_whole, its_chunked) and np.array_equal(whole, chunked)
    single, its_single = run([7, 1, 30, 32])  # a one-step call takes the plain path (no ping-pong, unfolded shift): rounding differs
    assert relerr(single, whole) < 1e-11 and np.abs(its_single - its_whole).max() <= 1
    fresh, its_fresh = run([7, 2, 29, 32], resume=0)  # every call starts from a fresh residual
    assert relerr(fresh, whole) < 1e-11 and np.abs(its_fresh - its_whole).max() <= 1
    # a write to the state between two calls must be noticed
    poked, _ = run([35, 35], between=lambda p, st: st.set_nodes(st.node_values()))
    assert relerr(poked, whole) < 1e-11
    # ... and so must another solve on the same problem (it overwrites the residual the run left)
    other, _ = run([35, 35], between=lambda p, st: p.run_fixed(p.new_state().set_nodes(u0), 2 * dt, 2, rtol=rtol))
    assert relerr(other, whole) < 1e-11


def test_run_fixed_at_steady_state_takes_zero_iterations(fv):
    """A state that already solves every step (no sources, u0 = the uniform Dirichlet head): each solve converges on
    entry, the ping-pong must leave the state where it is."""
    coords, nb, aol, vol, K, dn, dh = _box(fv, (9, 8, 7), sigma=0.0)
    N = len(vol)
    dh = np.full(len(dn), 2.5)
    p = fv.Problem.create(nb, aol, N, dn).assemble(K, np.zeros(N), dh)
    st = p.transient_begin(0.1, vol, np.full(N, 2.5))
    iters, info, _ = p.run_fixed(st, 10.0, 6, rtol=1e-10)
    assert (iters == 0).all() and info.converged
    assert np.array_equal(st.node_values(), np.full(N, 2.5))


# ------------------------------------------------------------------ aggregation-AMG preconditioner
def _aniso_box(fv, ns, sigma=3.0):
    """the box_model geometry (100 x 100 x 10 m: flat cells, z couplings ~100x the lateral ones) with a smooth log-K field"""
    from tests import workloads

    mins, maxs = [-50.0, -50.0, 0.0], [50.0, 50.0, 10.0]
    coords, nb, aol, vol = fv.regulargrid(mins, maxs, list(ns))
    logk = np.log(1e-5) + sigma * workloads.smooth_gaussian_field(ns, seed=0, radius_cells=(4, 4, 4))
    K = fv.nodehycos2neighborhycos(nb, logk, True)
    dn, dh = workloads.box_model_dirichlet(ns)
    return nb, aol, vol, K, dn, dh


def test_amg_preconditioned_solve_matches_direct_and_beats_jacobi(fv, oracle):
    ns = (24, 22, 20)
    nb, aol, vol, K, dn, dh = _aniso_box(fv, ns)
    N = len(vol)
    src = np.zeros(N)
    p = fv.Problem.create(nb, aol, N, dn).assemble(K, src, dh, None, True)
    head_j, res_j, ch_j = p.solve_steady(None, 1e-10, 20000)
    p.set_preconditioner("amg")
    rows, nnz = p.amg_info()
    assert rows[0] == p.n and nnz[0] == p.nnz and len(rows) >= 2
    assert (np.diff(rows) < 0).all() and rows[-1] <= 2048
    head_a, res_a, ch_a = p.solve_steady(None, 1e-10, 400)
    assert ch_j.isconverged and ch_a.isconverged
    assert ch_a.iters * 5 < ch_j.iters, (ch_a.iters, ch_j.iters)
    ohead = oracle.solvediffusion(nb[:, 0], nb[:, 1], aol, np.exp(K), src, dn, dh, solver="direct")[0]
    assert relerr(head_a, ohead) < HEAD_RTOL and relerr(head_j, ohead) < HEAD_RTOL
    hist = ch_a.data["resnorm"]
    assert len(hist) == ch_a.iters and hist[-1] <= 1e-10 * np.linalg.norm(p.b()) * 1.0000001
    # bit-reproducible: no floating-point atomics in the set-up or the cycle
    p2 = fv.Problem.create(nb, aol, N, dn).assemble(K, src, dh, None, True).set_preconditioner("amg")
    head_b, res_b, ch_b = p2.solve_steady(None, 1e-10, 400)
    assert np.array_equal(res_a, res_b) and np.array_equal(hist, ch_b.data["resnorm"])
    # the public wrapper: "amg", and the default "auto" (100 Jacobi iterations, then the V-cycle from that iterate:
    # one residual history, converged inside the reference's maxiter = 400)
    head_w, ch_w, *_ = fv.solvediffusion(nb, aol, np.exp(K), src, dn, dh, maxiter=400, rtol=1e-10, preconditioner="amg")
    assert ch_w.isconverged and relerr(head_w, ohead) < HEAD_RTOL
    head_d, ch_d, *_ = fv.solvediffusion(nb, aol, np.exp(K), src, dn, dh, maxiter=400, rtol=1e-10)
    assert ch_d.isconverged and 100 < ch_d.iters < 400 and relerr(head_d, ohead) < HEAD_RTOL
    assert len(ch_d.data["resnorm"]) == ch_d.iters and np.isfinite(ch_d.data["resnorm"]).all()
    assert ch_d.data["resnorm"][-1] <= 1e-10 * np.linalg.norm(p.b()) * 1.0000001
    _, ch_j400, *_ = fv.solvediffusion(nb, aol, np.exp(K), src, dn, dh, maxiter=400, rtol=1e-10, preconditioner="jacobi")
    assert not ch_j400.isconverged


def test_amg_cycle_is_symmetric_positive_definite(fv):
    ns = (20, 12, 9)
    nb, aol, vol, K, dn, dh = _aniso_box(fv, ns, sigma=2.0)
    N = len(vol)
    p = fv.Problem.create(nb, aol, N, dn).assemble(K, np.zeros(N), dh, None, True)
    p.transient_begin(0.1, vol, np.zeros(N))
    rng = np.random.default_rng(5)
    for sigma in (0.0, 1e-3):
        x, y = rng.standard_normal(p.n), rng.standard_normal(p.n)
        Mx, My = p.amg_apply(x, sigma), p.amg_apply(y, sigma)
        assert abs(y @ Mx - x @ My) <= 1e-12 * (np.linalg.norm(x) * np.linalg.norm(My))
        assert x @ Mx > 0 and y @ My > 0


def test_amg_tiny_problem_uses_the_exact_inverse(fv, oracle):
    """n <= 1024: the whole 'hierarchy' is the dense inverse, PCG converges in one iteration"""
    c = refcases.chain4()
    p = fv.Problem.create(np.stack([c["node1"], c["node2"]], 1), c["aol"], 4, c["dnodes"]).assemble(c["K"], c["sources"], c["dheads"])
    p.set_preconditioner("amg")
    head, res, ch = p.solve_steady(None, 1e-12, 10)
    assert ch.isconverged and ch.iters <= 2
    assert np.allclose(head, [1, 2 / 3, 1 / 3, 0], atol=1e-14)


def test_amg_in_implicit_steps_matches_jacobi_steps(fv):
    """the hierarchy carries the storage term (P^T D P is diagonal): large-dt implicit steps through run_fixed"""
    ns = (18, 16, 12)
    nb, aol, vol, K, dn, dh = _aniso_box(fv, ns, sigma=2.0)
    N = len(vol)
    src = np.zeros(N)
    out = {}
    for kind in ("jacobi", "amg", "auto"):
        p = fv.Problem.create(nb, aol, N, dn).assemble(K, src, dh, None, True).set_preconditioner(kind)
        st = p.transient_begin(0.1, vol, np.zeros(N))
        iters, info, _ = p.run_fixed(st, 3.0e4, 6, rtol=1e-12, maxiter=20000)
        assert info.converged
        out[kind] = (st.node_values(), iters)
    assert relerr(out["amg"][0], out["jacobi"][0]) < 1e-9
    assert out["amg"][1].sum() * 3 < out["jacobi"][1].sum()
    # "auto": the first step runs Jacobi-PCG, needs more than 50 iterations, and the V-cycle takes over from the second
    assert relerr(out["auto"][0], out["jacobi"][0]) < 1e-9
    assert out["auto"][1][0] == out["jacobi"][1][0] > 50 and (out["auto"][1][1:] <= out["amg"][1][1:] + 1).all()


# ------------------------------------------------------------------ multi-rank protocol on one GPU (loopback transport)
def _run_ranks_in_threads(fv, nranks, group_id, make_problem, schedule, rtol, by_rank=False):
    """One host thread per rank, each with its own context on device 0 and the loopback transport (fv_comm_init_local):
    the complete row-block driver — plan, pack, halo exchange, interior/boundary passes, reductions, carry-over,
    speculation — with real kernels; only the wire differs from RCCL."""
    import threading

    from fvamd import dist

    out, errors = [None] * nranks, []

    def worker(rank):
        try:
            ctx = fv.Context(0)
            dist.comm_init_local(ctx, nranks, rank, group_id)
            p = make_problem(ctx) if not by_rank else make_problem(ctx, rank)
            p, bounds = p if isinstance(p, tuple) else (p, None)
            blk = dist.RowBlock(p, nranks, rank, bounds)
            p.close()
            its = []
            dist.comm_diag(ctx, True)  # (events around the collectives and passes: must not change a bit of the result)
            for dt, nsteps in schedule:
                it, info, _ = blk.run_fixed(dt, nsteps, rtol)
                assert info.converged
                its.append(it.copy())
            diag = dist.comm_diag_get(ctx)
            dist.comm_diag(ctx, False)
            assert diag["allreduce"][1] > 0 and diag["halo_exchange"][1] == diag["halo_wait"][1] > 0 and diag["interior_spmv"][1] > 0, diag
            assert all(ms >= 0.0 for ms, _ in diag.values())
            out[rank] = (blk.lo, blk.hi, blk.state(), np.concatenate(its))
            blk.close()
            fv.load().fv_comm_destroy(ctx.handle)
        except BaseException as e:  # noqa: BLE001  (a failing rank would leave the others waiting at a barrier)
            errors.append((rank, repr(e)))

    threads = [threading.Thread(target=worker, args=(r,), daemon=True) for r in range(nranks)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not errors, errors
    assert all(not t.is_alive() for t in threads), "a rank did not finish (deadlock in the protocol?)"
    return out


@pytest.mark.parametrize("nranks", [2, 3])
def test_block_jacobi_amg_on_row_blocks(fv, oracle, nranks):
    """FV_PRECOND_AMG on row blocks: every rank builds the aggregation-AMG hierarchy of its own diagonal block and the
    V-cycle preconditions the distributed PCG (block-Jacobi, nothing travels inside the preconditioner).  High-contrast box:
    the heads are the oracle's direct solve, far fewer iterations than Jacobi-PCG, also in implicit steps with a large dt."""
    import threading

    from fvamd import dist

    coords, nb, aol, vol, K, dn, dh = _box(fv, (24, 20, 16), sigma=2.0)
    N = len(vol)
    src = np.zeros(N)
    u0 = np.zeros(N)
    ohead = oracle.solvediffusion(nb[:, 0], nb[:, 1], aol, K, src, dn, dh, solver="direct")[0]
    ous, _ = oracle.backwardeulerintegrate(u0, (0.0, 3 * 3.0e4), 0.1, vol, nb[:, 0], nb[:, 1], aol, K, src, dn, dh, stepper=oracle.fixedbackwardeulerstep, dt0=3.0e4, linearsolver=oracle.directlinearsolver)
    res = {}
    for kind in ("jacobi", "amg"):
        out, errors = [None] * nranks, []

        def worker(rank):
            try:
                ctx = fv.Context(0)
                dist.comm_init_local(ctx, nranks, rank, 1100 + 10 * nranks + (kind == "amg"))
                p = fv.Problem.create(nb, aol, N, dn, ctx).assemble(K, src, dh)
                p.transient_begin(0.1, vol, u0)
                blk = dist.RowBlock(p, nranks, rank).set_preconditioner(kind)
                p.close()
                x, info = blk.solve_steady(None, 1e-12, 20000)
                assert info.converged
                its, info2, _ = blk.run_fixed(3.0e4, 3, 1e-12, 20000)
                assert info2.converged
                out[rank] = (blk.lo, blk.hi, x, info.iters, blk.state(), its.copy())
                blk.close()
                fv.load().fv_comm_destroy(ctx.handle)
            except BaseException as e:  # noqa: BLE001
                errors.append((rank, repr(e)))

        threads = [threading.Thread(target=worker, args=(r,), daemon=True) for r in range(nranks)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=600)
        assert not errors, errors
        n = out[-1][1]
        xs, us = np.empty(n), np.empty(n)
        for lo, hi, x, it, u, its in out:
            xs[lo:hi], us[lo:hi] = x, u
        res[kind] = (xs, out[0][3], us, out[0][5])
    freenode = np.ones(N, bool)
    freenode[dn - 1] = False
    for kind in res:
        assert relerr(res[kind][0], ohead[freenode]) < HEAD_RTOL, kind
        assert relerr(res[kind][2], ous[-1][freenode]) < HEAD_RTOL, kind
    print("steady iterations: Jacobi-PCG", res["jacobi"][1], "block-Jacobi AMG-PCG", res["amg"][1], "; steps", res["jacobi"][3], res["amg"][3])
    assert res["amg"][1] * 4 < res["jacobi"][1] and res["amg"][3].sum() * 3 < res["jacobi"][3].sum()


def test_gathered_amg_levels_keep_the_iteration_count_of_one_gpu(fv, oracle):
    """FV_PRECOND_AMG_GATHERED on row blocks: aggregation stays rank-local on level 0, level 1 is the Galerkin product of the
    whole operator gathered on every rank, the levels below are built and applied by every rank for itself.  High-contrast box, 1
    / 2 / 3 / 5 / 8 loopback ranks: the iteration count of the steady solve stays within a few iterations of the one-GPU solve
    (block-Jacobi AMG: it grows with the rank count), heads = the oracle's direct solve; an implicit step with a large dt too."""
    import threading

    from fvamd import dist

    coords, nb, aol, vol, K, dn, dh = _box(fv, (40, 32, 24), sigma=2.0)
    N = len(vol)
    src = np.zeros(N)
    u0 = np.zeros(N)
    ohead = oracle.solvediffusion(nb[:, 0], nb[:, 1], aol, K, src, dn, dh, solver="direct")[0]
    ous, _ = oracle.backwardeulerintegrate(u0, (0.0, 2 * 3.0e4), 0.1, vol, nb[:, 0], nb[:, 1], aol, K, src, dn, dh, stepper=oracle.fixedbackwardeulerstep, dt0=3.0e4, linearsolver=oracle.directlinearsolver)
    freenode = np.ones(N, bool)
    freenode[dn - 1] = False
    p1 = fv.Problem.create(nb, aol, N, dn).assemble(K, src, dh)
    p1.set_preconditioner("amg")
    _, _, ch1 = p1.solve_steady(None, 1e-12, 400)
    assert ch1.isconverged
    p1.close()
    its = {}
    for kind, nranks in (("amg_gathered", 2), ("amg_gathered", 3), ("amg_gathered", 5), ("amg_gathered", 8), ("amg", 3), ("amg", 5)):
        out, errors = [None] * nranks, []

        def worker(rank):
            try:
                ctx = fv.Context(0)
                dist.comm_init_local(ctx, nranks, rank, 1300 + 10 * nranks + (kind == "amg"))
                p = fv.Problem.create(nb, aol, N, dn, ctx).assemble(K, src, dh)
                p.transient_begin(0.1, vol, u0)
                blk = dist.RowBlock(p, nranks, rank).set_preconditioner(kind)
                p.close()
                x, info = blk.solve_steady(None, 1e-12, 2000)
                assert info.converged
                sits, info2, _ = blk.run_fixed(3.0e4, 2, 1e-12, 2000)
                assert info2.converged
                out[rank] = (blk.lo, blk.hi, x, info.iters, blk.state(), sits.copy())
                blk.close()
                fv.load().fv_comm_destroy(ctx.handle)
            except BaseException as e:  # noqa: BLE001
                errors.append((rank, repr(e)))

        threads = [threading.Thread(target=worker, args=(r,), daemon=True) for r in range(nranks)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=600)
        assert not errors, errors
        n = out[-1][1]
        xs, us = np.empty(n), np.empty(n)
        for lo, hi, x, it, u, sits in out:
            xs[lo:hi], us[lo:hi] = x, u
            assert it == out[0][3] and np.array_equal(sits, out[0][5])  # every rank counts the same iterations
        assert relerr(xs, ohead[freenode]) < HEAD_RTOL, (kind, nranks)
        assert relerr(us, ous[-1][freenode]) < HEAD_RTOL, (kind, nranks)
        its[(kind, nranks)] = (out[0][3], out[0][5].tolist())
    print("steady AMG-PCG iterations: one GPU", ch1.iters, "; row blocks", its)
    # (not the same hierarchy as on one GPU — no aggregate crosses a rank boundary —, so not the same count to the iteration; but it
    # does not grow with the rank count, as the block-Jacobi one does)
    got = [its[("amg_gathered", nranks)][0] for nranks in (2, 3, 5, 8)]
    assert max(got) <= ch1.iters + 3 and max(got) - min(got) <= 12, (ch1.iters, its)
    assert min(its[("amg", 3)][0], its[("amg", 5)][0]) > max(got) + 15, its


@pytest.mark.parametrize("nranks", [1, 3])
def test_one_reduction_pcg_on_row_blocks(fv, oracle, nranks):
    """fv_tune key 34: the Chronopoulos-Gear form of the PCG in the many-iteration regime of the row-block driver — one
    3-double all-reduce per iteration instead of two all-reduces — against the classic form: same heads (and the oracle's
    direct solve), iteration counts within a few, and the collective pattern counted (fv_comm_stats)."""
    import threading

    from fvamd import dist

    coords, nb, aol, vol, K, dn, dh = _box(fv, (14, 11, 9), sigma=1.0)
    N = len(vol)
    src = np.zeros(N)
    src[N // 2] = -3e-4
    u0 = np.full(N, 0.5)
    ohead = oracle.solvediffusion(nb[:, 0], nb[:, 1], aol, K, src, dn, dh, solver="direct")[0]
    ous, _ = oracle.backwardeulerintegrate(u0, (0.0, 5 * 400.0), 0.1, vol, nb[:, 0], nb[:, 1], aol, K, src, dn, dh, stepper=oracle.fixedbackwardeulerstep, dt0=400.0, linearsolver=oracle.directlinearsolver)
    results = {}
    for form in (0, 1):
        out, errors = [None] * nranks, []

        def worker(rank):
            try:
                ctx = fv.Context(0)
                dist.comm_init_local(ctx, nranks, rank, 900 + 10 * form + nranks)
                p = fv.Problem.create(nb, aol, N, dn, ctx).assemble(K, src, dh)
                p.transient_begin(0.1, vol, u0)
                blk = dist.RowBlock(p, nranks, rank)
                p.close()
                dist.comm_stats(ctx, reset=True)
                x, info = blk.solve_steady(None, 1e-12, 5000)
                ar_steady, _ = dist.comm_stats(ctx, reset=True)
                its, info2, _ = blk.run_fixed(400.0, 5, 1e-12, 5000)
                ar_steps, _ = dist.comm_stats(ctx, reset=True)
                assert info.converged and info2.converged
                out[rank] = (blk.lo, blk.hi, x, info.iters, ar_steady, blk.state(), its.copy(), ar_steps)
                blk.close()
                fv.load().fv_comm_destroy(ctx.handle)
            except BaseException as e:  # noqa: BLE001
                errors.append((rank, repr(e)))

        assert fv.load().fv_tune(34, form) == 0
        try:
            threads = [threading.Thread(target=worker, args=(r,), daemon=True) for r in range(nranks)]
            for t in threads:
                t.start()
            for t in threads:
                t.join(timeout=300)
        finally:
            fv.load().fv_tune(34, 0)
        assert not errors, errors
        n = out[-1][1]
        xs, us = np.empty(n), np.empty(n)
        for lo, hi, x, it, ar, u, its, ar2 in out:
            xs[lo:hi], us[lo:hi] = x, u
        results[form] = (xs, out[0][3], out[0][4], us, out[0][6], out[0][7])
    freenode = np.ones(N, bool)
    freenode[dn - 1] = False
    for form in (0, 1):
        xs, it, ar, us, its, ar2 = results[form]
        assert relerr(xs, ohead[freenode]) < HEAD_RTOL and relerr(us, ous[-1][freenode]) < HEAD_RTOL, form
    it0, ar0, its0, ars0 = results[0][1], results[0][2], results[0][4], results[0][5]
    it1, ar1, its1, ars1 = results[1][1], results[1][2], results[1][4], results[1][5]
    assert it0 > 20 and abs(it1 - it0) <= 3 and np.abs(its1.astype(int) - its0.astype(int)).max() <= 2
    # classic: set-up (1) + two per iteration (the last iteration's second one included);  one-reduction: set-up (1) + delta0 (1)
    # + one per iteration enqueued (iterations are enqueued in chunks, so a few surplus no-op rounds may follow convergence)
    assert 2 * it0 <= ar0 <= 2 * it0 + 2 * 32 + 2
    assert it1 + 2 <= ar1 <= it1 + 32 + 2
    assert ar1 < 0.6 * ar0 and ars1 < 0.7 * ars0
    assert relerr(results[1][0], results[0][0]) < 1e-9 and relerr(results[1][3], results[0][3]) < 1e-9


@pytest.mark.parametrize("nranks", [2, 3])
def test_row_blocks_steady_solve_adaptive_stepper_and_host_forcing(fv, oracle, nranks):
    """Row blocks beyond the fixed-dt run (loopback transport, one thread per rank): fv_dist_solve_steady against the
    single-GPU steady solve and the oracle's direct solve; fv_dist_run_adaptive against fv_transient_run_adaptive (same `ts`,
    same number of solves — every rank takes the same step-doubling decisions from all-reduced error norms); fv_dist_step
    with a caller's volume-scaled forcing against fv_transient_step with the same bhat."""
    import threading

    from fvamd import dist

    coords, nb, aol, vol, K, dn, dh = _box(fv, (14, 11, 9), sigma=1.0)
    N = len(vol)
    src = np.zeros(N)
    src[N // 2] = -3e-4
    u0 = np.full(N, 0.5) + 0.01 * np.random.default_rng(2).standard_normal(N)
    Ss = 0.1

    def make(ctx):
        p = fv.Problem.create(nb, aol, N, dn, ctx).assemble(K, src, dh)
        p.transient_begin(Ss, vol, u0)
        return p

    ref = make(fv.default_context())
    st = fv.DeviceVector(ref, 0, owned=False)
    head_ref, x_ref, ch = ref.solve_steady(None, 1e-13, 5000, want_resnorm=False)
    ohead = oracle.solvediffusion(nb[:, 0], nb[:, 1], aol, K, src, dn, dh, solver="direct")[0]
    assert ch.isconverged and relerr(head_ref, ohead) < HEAD_RTOL
    ts_ref, nsolves_ref, _ = ref.run_adaptive(st, 0.0, 4.0e3, dt0=50.0, atol=1e-5, rtol=1e-13)
    u_adapt = st.free_values()
    freenode, _ = ref.free_maps()
    bhat = 1e-3 * np.random.default_rng(3).standard_normal(ref.n)
    st.set_nodes(u0)
    ref.step(st, st, 120.0, bhat, rtol=1e-13)
    u_bhat = st.free_values()
    ref.close()
    out, errors = [None] * nranks, []

    def worker(rank):
        try:
            ctx = fv.Context(0)
            dist.comm_init_local(ctx, nranks, rank, 700 + nranks)
            p = make(ctx)
            blk = dist.RowBlock(p, nranks, rank)
            p.close()
            x, info = blk.solve_steady(None, 1e-13, 5000)
            assert info.converged
            ts, nsolves, info = blk.run_adaptive(0.0, 4.0e3, dt0=50.0, atol=1e-5, rtol=1e-13)
            ua = blk.state()
            blk.close()
            # a fresh block for the forced step (the state restarts from u0)
            p = make(ctx)
            blk = dist.RowBlock(p, nranks, rank)
            p.close()
            info = blk.step(120.0, bhat[blk.lo : blk.hi], rtol=1e-13)
            assert info.converged
            out[rank] = (blk.lo, blk.hi, x, ts, nsolves, ua, blk.state())
            blk.close()
            fv.load().fv_comm_destroy(ctx.handle)
        except BaseException as e:  # noqa: BLE001
            errors.append((rank, repr(e)))

    threads = [threading.Thread(target=worker, args=(r,), daemon=True) for r in range(nranks)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    assert all(not t.is_alive() for t in threads)
    xs, ua, ub = np.empty_like(x_ref), np.empty_like(x_ref), np.empty_like(x_ref)
    for lo, hi, x, ts, nsolves, a, b in out:
        xs[lo:hi], ua[lo:hi], ub[lo:hi] = x, a, b
        assert np.array_equal(ts, ts_ref) and nsolves == nsolves_ref
    assert relerr(xs, x_ref) < 1e-10
    assert relerr(ua, u_adapt) < 1e-9
    assert relerr(ub, u_bhat) < 1e-11


@pytest.mark.parametrize("nranks", [2, 3, 16])  # 16: blocks thinner than a grid plane, i.e. more than two peers per rank
def test_multi_rank_driver_over_the_loopback_transport(fv, nranks):
    import bench

    ns = [14, 9, 7]
    mins, maxs = bench.spacing_box(ns)
    dn, src = bench.box_setup(ns)

    def make_problem(ctx):
        p = fv.Problem.regulargrid(mins, maxs, ns, dn, ctx)
        K = 1e-5 * np.exp(np.random.default_rng(0).standard_normal(p.F))
        p.assemble(K, src, np.full(len(dn), 1e3))
        u0 = np.full(p.N, 1e3) + np.random.default_rng(1).standard_normal(p.N)
        p.transient_begin(0.1, None, u0)
        return p

    # several iterations per step, then the one-iteration regime (speculative set-up, 5-scalar reduction), then a
    # large step again (the speculation misses)
    schedule = [(3600.0, 6), (2.0**-10, 40), (3600.0, 3)]
    ref = make_problem(fv.default_context())
    st = fv.DeviceVector(ref, 0, owned=False)
    ref_its = np.concatenate([ref.run_fixed(st, dt, k, 1e-12)[0] for dt, k in schedule])
    want = st.free_values()
    out = _run_ranks_in_threads(fv, nranks, 100 + nranks, make_problem, schedule, 1e-12)
    got = np.empty_like(want)
    for lo, hi, state, its in out:
        got[lo:hi] = state
        assert np.array_equal(its, ref_its), (its, ref_its)  # reductions differ in order only: same iteration counts
    assert relerr(got, want) < 1e-12
    # the bursts of unpolled steps break (fault injection) at the same place on every rank and are resumed: at the first
    # step of a burst, in the middle, at the last one (which reduces its own sums) — with a step's five sums all-reduced
    # together with the next step's p.q (fv_tune 22, the default) and separately: same bits either way
    lib = fv.load()
    baseline = [o[2].copy() for o in out]
    for brk in (2, 0, 7):
        states = {}
        for merged in (1, 0):
            try:
                lib.fv_tune(14, brk)
                lib.fv_tune(22, merged)
                res = _run_ranks_in_threads(fv, nranks, 300 + 10 * brk + merged + 100 * nranks, make_problem, schedule, 1e-12)
            finally:
                lib.fv_tune(14, -1)
                lib.fv_tune(22, 1)
            for lo, hi, state, its in res:
                got[lo:hi] = state
                assert np.array_equal(its, res[0][3]) and (its[6:46] == 2).sum() >= 2 and set(np.unique(its[6:46])) == {1, 2}
            assert relerr(got, want) < 1e-12
            states[merged] = [r[2] for r in res]
        assert all(np.array_equal(a, b) for a, b in zip(states[0], states[1])), brk
    # ... and without injected breaks the merged reduction changes no bit either
    lib.fv_tune(22, 0)
    try:
        res = _run_ranks_in_threads(fv, nranks, 900 + nranks, make_problem, schedule, 1e-12)
    finally:
        lib.fv_tune(22, 1)
    assert all(np.array_equal(a, r[2]) for a, r in zip(baseline, res))


@pytest.mark.parametrize("nranks", [2, 5])
def test_slab_assembled_row_blocks_equal_the_globally_assembled_ones(fv, nranks):
    """Each rank assembles only the faces of its own planes (fv_problem_create_regulargrid_slab) and takes its rows with
    fv_dist_setup_bounds: same plan, same matrix rows, bit-identical run as blocks cut from the global operator."""
    import bench
    from fvamd import dist

    ns = [11, 6, 5]
    mins, maxs = bench.spacing_box(ns)
    dn, src = bench.box_setup(ns)
    dn = np.union1d(dn, [1 + 3 * 30 + 7, 1 + 4 * 30, 1 + 4 * 30 + 1])  # Dirichlet cells inside, two at the start of a plane
    dh = np.full(len(dn), 1e3)
    ref = fv.Problem.regulargrid(mins, maxs, ns, dn)
    Kg = 1e-5 * np.exp(np.random.default_rng(0).standard_normal(ref.F))
    u0 = np.full(ref.N, 1e3) + np.random.default_rng(1).standard_normal(ref.N)
    planes = dist.slab_planes(ns[0], nranks)
    bounds_ref = [ref.free_rows_before(q * 30) for q in planes]
    free, n2f = ref.free_maps()
    assert bounds_ref == [int(free[: q * 30].sum()) for q in planes] and bounds_ref[-1] == ref.n

    def global_blocks(ctx, rank):
        p = fv.Problem.regulargrid(mins, maxs, ns, dn, ctx)
        p.assemble(Kg, src, dh)
        p.transient_begin(0.1, None, u0)
        return p, bounds_ref

    plans = {}

    def slab_blocks(ctx, rank):
        p, bounds = dist.slab_problem(mins, maxs, ns, dn, nranks, rank, ctx)
        assert bounds == bounds_ref
        f0, f1 = dist.slab_face_range(ns, planes[rank], planes[rank + 1])
        assert p.F == f1 - f0 and p.N == ref.N and p.n == ref.n
        p.assemble(Kg[f0:f1], src, dh)
        p.transient_begin(0.1, None, u0)
        blk = dist.RowBlock(p, nranks, rank, bounds)  # a second one, for the plan only (no communication)
        rng = np.random.default_rng(rank)
        plans[rank] = (blk.plan(), blk.spmv_halo(rng.standard_normal(blk.nloc), rng.standard_normal(blk.nhalo), 0.5))
        blk.close()
        return p, bounds

    schedule = [(3600.0, 4), (2.0**-10, 20)]
    a = _run_ranks_in_threads(fv, nranks, 500 + nranks, global_blocks, schedule, 1e-12, by_rank=True)
    b = _run_ranks_in_threads(fv, nranks, 520 + nranks, slab_blocks, schedule, 1e-12, by_rank=True)
    ref.assemble(Kg, src, dh)
    ref.transient_begin(0.1, None, u0)
    for rank, (x, y) in enumerate(zip(a, b)):
        assert x[:2] == y[:2] == (bounds_ref[rank], bounds_ref[rank + 1])
        assert np.array_equal(x[2], y[2]) and np.array_equal(x[3], y[3])
        want = dist.RowBlock(ref, nranks, rank, bounds_ref)
        for u, v in zip(want.plan(), plans[rank][0]):
            assert np.array_equal(u, v)
        rng = np.random.default_rng(rank)
        assert np.array_equal(want.spmv_halo(rng.standard_normal(want.nloc), rng.standard_normal(want.nhalo), 0.5), plans[rank][1])
        want.close()
    # a rank whose rows are not inside its slab is refused
    p, bounds = dist.slab_problem(mins, maxs, ns, dn, nranks, 0)
    p.assemble(1e-5, src, dh)
    p.transient_begin(0.1, None, u0)
    with pytest.raises(fv.FVError, match="not inside the slab"):
        dist.RowBlock(p, nranks, 1, bounds)
    with pytest.raises(fv.FVError, match="bounds must run"):
        dist.RowBlock(p, nranks, 0, [0] + bounds[1:-1] + [bounds[-1] - 1])


@pytest.mark.parametrize("dt,rtol", [(60.0, 1e-5), (20.0, 3e-5), (600.0, 1e-4)])
def test_bursts_with_steps_already_converged_at_their_set_up(fv, dt, rtol):
    """Loose tolerances: some steps of a fixed-dt run need no iteration at all (the carried residual is already within the
    tolerance), also in the middle of a burst of unpolled steps, where the host flips the state vectors without looking —
    the device hands the iterate over unchanged and counts 0 iterations: same bits and the same iteration counts as the
    run that polls after every step.  (Before the fix the burst lost an update there: 4.6e-5 relative at rtol 1e-5.)
    The row-block driver on two ranks sees the same steps."""
    import bench

    ns = [30, 26, 22]
    mins, maxs = bench.spacing_box(ns)
    dn, src = bench.box_setup(ns)
    src[:] = 0.0
    lib = fv.load()

    def make_problem(ctx):
        p = fv.Problem.regulargrid(mins, maxs, ns, dn, ctx)
        p.assemble(np.array([1e-5]), src, np.full(len(dn), 1e3))
        p.transient_begin(0.1, None, np.full(p.N, 1e3) + np.random.default_rng(0).standard_normal(p.N))
        return p

    out = {}
    try:
        for chain in (0, 8):
            lib.fv_tune(13, chain)
            p = make_problem(fv.default_context())
            st = fv.DeviceVector(p, 0, owned=False)
            its = np.concatenate([p.run_fixed(st, dt, 40, rtol)[0] for _ in range(6)])
            out[chain] = (st.free_values(), its)
            p.close()
    finally:
        lib.fv_tune(13, 8)
    assert np.array_equal(out[0][1], out[8][1]) and np.array_equal(out[0][0], out[8][0])
    assert (out[8][1] == 0).sum() > 100 and (out[8][1] >= 1).sum() >= 2  # the mix this test is about
    res = _run_ranks_in_threads(fv, 2, 950 + int(dt), make_problem, [(dt, 40)] * 6, rtol)
    got = np.concatenate([r[2] for r in res])
    for lo, hi, state, its in res:
        assert np.array_equal(its, res[0][3])
    # reductions are summed in another order on two ranks: a step on the edge of the tolerance may fall on the other side
    assert np.abs(res[0][3].astype(int) - out[8][1].astype(int)).sum() <= 8
    assert relerr(got, out[8][0]) < 50 * rtol


@pytest.mark.parametrize("tight", [False, True])
def test_bursts_against_step_by_step_polling_on_random_schedules(fv, tight):
    """tools/burst_fuzz.py: random boxes and random schedules of (dt, steps, rtol) — 0-, 1- and many-iteration steps mixed —
    run with a poll after every step, with bursts of 8 and of 3 unpolled steps, and with the bursts' merged launches off:
    states and iteration counts must agree bit for bit."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("burst_fuzz", os.path.join(os.path.dirname(os.path.dirname(GOLDEN)), "tools", "burst_fuzz.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run(fv, 8, tight, verbose=False) == 0


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5])
def test_row_block_bursts_against_step_by_step_polling_on_random_schedules(fv, seed):
    """The same differential check for the row-block driver (2 or 3 ranks over the loopback transport): with the rank count
    fixed, a poll after every step, bursts with the merged collective and bursts with the two separate collectives must
    give the same bits and the same iteration counts on every rank."""
    import bench

    rng = np.random.default_rng(1000 + seed)
    nranks = 2 + seed % 2
    ns = [int(rng.integers(12, 30)), int(rng.integers(10, 24)), int(rng.integers(8, 20))]
    mins, maxs = bench.spacing_box(ns)
    dn, src = bench.box_setup(ns)
    if seed % 3 == 0:
        src[:] = 0.0
    tight = seed % 2 == 1
    dts, rtols = ([2.0**-10, 2.0**-8, 1.0, 3600.0], [1e-8, 1e-10, 1e-12]) if tight else ([2.0**-8, 1.0, 20.0, 60.0, 600.0], [1e-3, 1e-5, 3e-5, 1e-8])
    schedule = [(float(rng.choice(dts)), int(rng.integers(1, 40))) for _ in range(int(rng.integers(3, 7)))]
    rtol = float(rng.choice(rtols))

    def make_problem(ctx):
        p = fv.Problem.regulargrid(mins, maxs, ns, dn, ctx)
        K = 1e-5 * np.exp(np.random.default_rng(100 + seed).standard_normal(p.F))
        p.assemble(K, src, np.full(len(dn), 1e3))
        p.transient_begin(0.1, None, np.full(p.N, 1e3) + np.random.default_rng(200 + seed).standard_normal(p.N))
        return p

    lib = fv.load()
    out = {}
    try:
        for k, (chain, merged) in enumerate(((0, 1), (8, 1), (8, 0))):
            lib.fv_tune(13, chain)
            lib.fv_tune(22, merged)
            out[(chain, merged)] = _run_ranks_in_threads(fv, nranks, 2000 + 10 * seed + k, make_problem, schedule, rtol)
    finally:
        lib.fv_tune(13, 8)
        lib.fv_tune(22, 1)
    ref = out[(0, 1)]
    for key, res in out.items():
        for a, b in zip(ref, res):
            assert np.array_equal(a[3], b[3]), (key, schedule, rtol, a[3], b[3])
            assert np.array_equal(a[2], b[2]), (key, schedule, rtol)


@pytest.mark.parametrize("case", ["fourfractures", "multigraph"])
def test_rank_local_assembly_of_a_face_list_mesh(fv, case):
    """Unstructured meshes without the global operator on every rank: a rank builds its problem from the faces incident to
    its rows only (dist.partial_problem) — plan, block product and the whole run bit-identical to blocks cut from the
    globally assembled operator (the real DFN mesh, 3 ranks; a random multigraph with repeated faces, 4 ranks)."""
    from fvamd import dist

    rng = np.random.default_rng(7)
    if case == "fourfractures":
        d = np.load(os.path.join(GOLDEN, "fourfractures.npz"))
        nb = np.stack([d["node1"], d["node2"]], 1)
        aol, dn, dh = d["areasoverlengths"], d["dirichletnodes"], d["dirichletheads"]
        N, nranks = 2106, 3
        K = d["conductivities"] * np.exp(rng.standard_normal(len(aol)))
        vol = np.exp(rng.standard_normal(N)) * 1e-3
        u0 = np.full(N, 1.5e6) + 1e3 * rng.standard_normal(N)
        Ss, schedule, rtol = 1e-9, [(50.0, 5), (0.5, 12)], 1e-12
    else:
        N, F, nranks = 1500, 6000, 4
        nb = np.stack([rng.integers(1, N + 1, F), rng.integers(1, N + 1, F)], 1)
        nb[rng.integers(0, F, 500)] = nb[rng.integers(0, F, 500)]  # repeated faces
        aol = np.exp(rng.standard_normal(F))
        dn = rng.choice(N, 200, replace=False) + 1
        dh = rng.standard_normal(200)
        K = np.exp(rng.standard_normal(F))
        vol = np.exp(rng.standard_normal(N))
        u0 = rng.standard_normal(N)
        Ss, schedule, rtol = 1.0, [(0.5, 4), (0.001, 12)], 1e-12
    src = np.zeros(N)

    def global_blocks(ctx, rank):
        p = fv.Problem.create(nb, aol, N, dn, ctx).assemble(K, src, dh)
        p.transient_begin(Ss, vol, u0)
        return p

    probes = {}

    def local_blocks(ctx, rank):
        p, bounds, faces = dist.partial_problem(nb, aol, N, dn, nranks, rank, None, ctx)
        assert 0 < len(faces) < len(aol) and p.F == len(faces)
        p.assemble(K[faces], src, dh)
        p.transient_begin(Ss, vol, u0)
        blk = dist.RowBlock(p, nranks, rank, bounds)
        r2 = np.random.default_rng(rank)
        probes[rank] = (blk.plan(), blk.spmv_halo(r2.standard_normal(blk.nloc), r2.standard_normal(blk.nhalo), 0.3))
        blk.close()
        return p, bounds

    a = _run_ranks_in_threads(fv, nranks, 3000 + nranks, global_blocks, schedule, rtol, by_rank=True)
    b = _run_ranks_in_threads(fv, nranks, 3010 + nranks, local_blocks, schedule, rtol, by_rank=True)
    ref = fv.Problem.create(nb, aol, N, dn).assemble(K, src, dh)
    ref.transient_begin(Ss, vol, u0)
    for rank, (x, y) in enumerate(zip(a, b)):
        assert x[:2] == y[:2] and np.array_equal(x[2], y[2]) and np.array_equal(x[3], y[3])
        want = dist.RowBlock(ref, nranks, rank)
        for u, v in zip(want.plan(), probes[rank][0]):
            assert np.array_equal(u, v)
        r2 = np.random.default_rng(rank)
        assert np.array_equal(want.spmv_halo(r2.standard_normal(want.nloc), r2.standard_normal(want.nhalo), 0.3), probes[rank][1])
        want.close()


def test_gathered_amg_on_rank_local_blocks_of_an_irregular_mesh(fv):
    """FV_PRECOND_AMG_GATHERED where no rank ever holds the whole operator: a random multigraph (repeated faces, rows of very
    different length, cells without a face), every rank assembling only the faces of its own rows (dist.partial_problem), 3 ranks.
    The solution is the one-GPU solution; the iteration count stays near the one-GPU AMG's although most couplings of this graph cross
    rank boundaries."""
    import threading

    from fvamd import dist

    rng = np.random.default_rng(21)  # (the mesh of test_amg_on_a_random_multigraph_with_isolated_and_zero_rows)
    N, F, nranks = 6000, 26000, 3
    n1 = rng.integers(1, N - 30, F)  # the last 30 cells have no face at all
    n2 = rng.integers(1, N - 30, F)
    n2[::40] = n1[::40]
    nb = np.stack([n1, n2], 1)
    aol = np.exp(rng.uniform(-3, 3, F))
    K = np.exp(rng.normal(0.0, 1.0, F))
    dn = rng.choice(N - 30, 300, replace=False) + 1
    dh = rng.uniform(0.0, 2.0, 300)
    src = 1e-2 * rng.standard_normal(N)
    src[dn - 1] = 0
    src[N - 30 :] = 0
    ref = fv.Problem.create(nb, aol, N, dn).assemble(K, src, dh)
    A, b = ref.csc().toscipy().tocsr(), ref.b()
    live = A.diagonal() > 0
    ref.set_preconditioner("amg")
    _, x1, ch1 = ref.solve_steady(None, 1e-12, 400)
    assert ch1.isconverged
    out, errors = [None] * nranks, []

    def worker(rank):
        try:
            ctx = fv.Context(0)
            dist.comm_init_local(ctx, nranks, rank, 1490)
            p, bounds, faces = dist.partial_problem(nb, aol, N, dn, nranks, rank, None, ctx)
            p.assemble(K[faces], src, dh)
            p.transient_begin(1.0, np.ones(N), np.zeros(N))
            blk = dist.RowBlock(p, nranks, rank, bounds).set_preconditioner("amg_gathered")
            p.close()
            x, info = blk.solve_steady(None, 1e-12, 400)
            out[rank] = (blk.lo, blk.hi, x, info.iters, info.converged)
            blk.close()
            fv.load().fv_comm_destroy(ctx.handle)
        except BaseException as e:  # noqa: BLE001
            errors.append((rank, repr(e)))

    threads = [threading.Thread(target=worker, args=(r,), daemon=True) for r in range(nranks)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not errors, errors
    xs = np.empty(out[-1][1])
    for lo, hi, x, it, conv in out:
        xs[lo:hi] = x
        assert conv and it == out[0][3]
    r = A @ xs - b
    assert np.linalg.norm(r[live]) <= 1e-10 * np.linalg.norm(b)
    assert relerr(xs[live], x1[live]) < 1e-8
    print("irregular mesh, AMG-PCG iterations: one GPU", ch1.iters, "; gathered levels on 3 rank-local blocks", out[0][3])
    # (a random graph cut into row ranges: two thirds of a row's neighbours belong to other ranks and cannot be aggregated with it)
    assert out[0][3] <= ch1.iters * 3 // 2


def test_amg_on_a_random_multigraph_with_isolated_and_zero_rows(fv, oracle):
    """AMG set-up on an irregular operator: repeated faces, self-loops (zero contribution), rows of very different
    length, free nodes whose only neighbours are Dirichlet (no couplings: the smoother alone must solve them), a
    few thousand rows so that real coarsening happens."""
    rng = np.random.default_rng(21)
    N, F = 6000, 26000
    n1 = rng.integers(1, N - 30, F)  # the last 30 nodes have no face at all
    n2 = rng.integers(1, N - 30, F)
    n2[::40] = n1[::40]
    aol = np.exp(rng.uniform(-3, 3, F))
    K = np.exp(rng.normal(0.0, 1.0, F))
    dn = rng.choice(N - 30, 300, replace=False) + 1
    dh = rng.uniform(0.0, 2.0, 300)
    src = 1e-2 * rng.standard_normal(N)
    src[dn - 1] = 0
    src[N - 30 :] = 0  # the isolated nodes: 0 * x = 0 stays solvable for CG
    nb = np.stack([n1, n2], 1)
    p = fv.Problem.create(nb, aol, N, dn).assemble(K, src, dh)
    A, b = p.csc().toscipy().tocsr(), p.b()
    deg = np.diff(A.indptr)
    assert deg.max() > 3 * max(deg[deg > 0].min(), 1)
    p.set_preconditioner("amg")
    rows, nnz = p.amg_info()
    assert len(rows) >= 2 and rows[-1] <= 2048
    head, res, ch = p.solve_steady(None, 1e-12, 400)
    assert ch.isconverged
    live = A.diagonal() > 0
    r = A @ res - b
    assert np.linalg.norm(r[live]) <= 1e-10 * np.linalg.norm(b)
    p.set_preconditioner("jacobi")
    head_j, res_j, ch_j = p.solve_steady(None, 1e-12, 20000)
    assert ch_j.isconverged and ch.iters < ch_j.iters
    assert relerr(res[live], res_j[live]) < 1e-8


@pytest.mark.parametrize("case", ["box", "multigraph"])
def test_amg_galerkin_by_row_merge_is_the_sorted_product_bit_for_bit(fv, case):
    """The Galerkin products of the set-up by merging the member rows of each aggregate (amg_merge_kernel: a stable rank sort of a
    coarse row's few dozen entries in LDS; the multigraph's hubs take the launch with the larger capacity) against the global
    stable radix sort it replaces (FV_AMG_GALERKIN=sort): the same level sizes, and the V-cycle applied to the same vector gives the
    same bits — same sums in the same order."""
    rng = np.random.default_rng(3)
    if case == "box":
        nb, aol, vol, K, dn, dh = _aniso_box(fv, (40, 36, 30), sigma=2.5)
        N = len(vol)
    else:
        N, F = 30000, 140000
        n1 = rng.integers(1, N, F)
        n2 = rng.integers(1, N, F)
        hub = rng.choice(N, 12, replace=False) + 1  # a dozen hubs with ~600 faces each: coarse rows beyond the small launch's 64 entries
        n1[: 12 * 600] = np.repeat(hub, 600)
        n1[12 * 600 : 12 * 600 + 5000] = hub[0]  # ... and one with 5 600: beyond the large launch's 2 048 too, that pass falls back to the sort
        aol = np.exp(rng.uniform(-2, 2, F))
        K = np.exp(rng.normal(0.0, 1.0, F))
        dn = rng.choice(N, 500, replace=False) + 1
        dh = rng.uniform(0.0, 2.0, 500)
        nb = np.stack([n1, n2], 1)
    out = []
    for how in ("sort", "merge"):
        os.environ["FV_AMG_GALERKIN"] = how
        try:
            p = fv.Problem.create(nb, aol, N, dn).assemble(K, np.zeros(N), dh, None, case == "box")
            p.set_preconditioner("amg")
            rows, nnz = p.amg_info()
            z = p.amg_apply(np.random.default_rng(9).standard_normal(p.n), 0.0)
            out.append((rows.tolist(), nnz.tolist(), z))
            p.close()
        finally:
            os.environ.pop("FV_AMG_GALERKIN", None)
    assert len(out[0][0]) >= 3 and out[0][0] == out[1][0] and out[0][1] == out[1][1], (out[0][:2], out[1][:2])
    assert np.isfinite(out[0][2]).all() and np.array_equal(out[0][2], out[1][2])


def test_run_adaptive_degenerate_spans(fv):
    coords, nb, aol, vol, K, dn, dh = _box(fv, (6, 5, 4), sigma=0.5)
    N = len(vol)
    p = fv.Problem.create(nb, aol, N, dn).assemble(K, np.zeros(N), dh)
    st = p.transient_begin(0.1, vol, np.full(N, 0.25))
    ts, nsolves, info = p.run_adaptive(st, 3.0, 3.0, dt0=1.0)  # empty span: no step, state untouched
    assert ts.tolist() == [3.0] and nsolves == 0 and np.array_equal(st.node_values()[~np.isin(np.arange(1, N + 1), dn)], np.full(N - len(dn), 0.25))
    ts, nsolves, info = p.run_adaptive(st, 0.0, 1e-3, dt0=5.0, atol=1e-4, rtol=1e-12)  # dt0 clipped to the span
    assert ts[0] == 0.0 and ts[-1] == 1e-3 and nsolves >= 3 and info.converged
    with pytest.raises(fv.FVError):
        p.run_adaptive(st, 0.0, 1.0, dt0=0.0)


def test_reordered_mesh_gives_the_same_heads(fv):
    """meshio.locality_order is a renaming of the cells: heads of the re-ordered problem, mapped back, are the heads."""
    from tests import workloads

    w = workloads.fractures_like(4, 50, seed=2)
    order, rank = fv.meshio.locality_order(w["node1"], w["node2"], w["N"])
    m = fv.meshio.reorder_mesh(dict(node1=w["node1"], node2=w["node2"], aol=w["aol"], volumes=w["volumes"], dnodes=w["dnodes"], dheads=w["dheads"]), rank)
    src = np.zeros(w["N"])
    h0, ch0, *_ = fv.solvediffusion(np.stack([w["node1"], w["node2"]], 1), w["aol"], w["K"], src, w["dnodes"], w["dheads"], maxiter=20000, rtol=1e-13, preconditioner="jacobi")
    h1, ch1, *_ = fv.solvediffusion(np.stack([m["node1"], m["node2"]], 1), m["aol"], w["K"], src, m["dnodes"], m["dheads"], maxiter=20000, rtol=1e-13, preconditioner="jacobi")
    assert ch0.isconverged and ch1.isconverged
    assert relerr(h1[rank - 1], h0) < 1e-9
    # the same through reorder=True: head, A, b, freenode in the caller's numbering; A and b bit for bit what the
    # un-reordered assembly gives (per-entry sums run in face order whatever the cells are called)
    nb = np.stack([w["node1"], w["node2"]], 1)
    h2, ch2, A2, b2, fn2 = fv.solvediffusion(nb, w["aol"], w["K"], src, w["dnodes"], w["dheads"], maxiter=20000, rtol=1e-13, preconditioner="jacobi", reorder=True)
    _, _, A0, b0, fn0 = fv.solvediffusion(nb, w["aol"], w["K"], src, w["dnodes"], w["dheads"], maxiter=5, preconditioner="jacobi")
    assert ch2.isconverged and relerr(h2, h0) < 1e-9
    assert np.array_equal(fn2, fn0) and np.array_equal(b2, b0)
    assert np.array_equal(A2.colptr, A0.colptr) and np.array_equal(A2.rowval, A0.rowval) and np.array_equal(A2.nzval, A0.nzval)
    u0 = np.full(w["N"], 1.5e6)
    us_a, ts_a = fv.backwardeulerintegrate(u0, (0.0, 3.0), 1e-9, w["volumes"], nb, w["aol"], w["K"], src, w["dnodes"], w["dheads"], stepper=fv.fixedbackwardeulerstep, dt0=1.0, rtol=1e-13)
    us_b, ts_b = fv.backwardeulerintegrate(u0, (0.0, 3.0), 1e-9, w["volumes"], nb, w["aol"], w["K"], src, w["dnodes"], w["dheads"], stepper=fv.fixedbackwardeulerstep, dt0=1.0, rtol=1e-13, reorder=True)
    assert ts_a == ts_b and all(relerr(a, b) < 1e-10 for a, b in zip(us_a, us_b))
