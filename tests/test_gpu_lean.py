"""GPU: FV_OPT_LEAN_SETUP — a regulargrid problem without face arrays, incident lists and CSR in HBM (csrc/fv_lean.hip): b, the
diagonal and every storage form of the solver filled from rows formed on the fly.  The claim is "the same doubles as the CSR
route", so every check here is bit for bit against the same problem created with the option off: b, products with random
vectors in every form (plain and with the shift folded in), whole transient runs (heads, iteration counts, the kernels that ran).
What needs the faces or the CSR must fail loudly."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MINS, MAXS = [-50.0, -50.0, 0.0], [50.0, 50.0, 10.0]


def _dirichlet(ns, kind):
    n1, n2, n3 = ns
    idx = np.arange(n1 * n2 * n3).reshape(n1, n2, n3)
    if kind == "xfaces":
        d = np.r_[idx[0].ravel(), idx[-1].ravel()]
    elif kind == "lateral":
        m = np.zeros((n1, n2, n3), bool)
        m[:, 0, :] = m[:, -1, :] = m[:, :, 0] = m[:, :, -1] = True
        d = idx[m]
    elif kind == "xy":  # the x- and the y-faces
        m = np.zeros((n1, n2, n3), bool)
        m[0] = m[-1] = True
        m[:, 0, :] = m[:, -1, :] = True
        d = idx[m]
    elif kind == "well":  # the x-faces and one column of cells through the box
        d = np.r_[idx[0].ravel(), idx[-1].ravel(), idx[1:-1, n2 // 2, n3 // 3]]
    else:
        raise ValueError(kind)
    return (np.unique(d) + 1).astype(np.int64)


def _conductivities(F, kind, rng):
    if kind == "uniform":
        return np.array([1e-5]), None, False
    if kind == "faces":
        return np.exp(np.log(1e-5) + 0.7 * rng.standard_normal(F)), None, False
    if kind == "log":
        return np.log(1e-5) + 0.7 * rng.standard_normal(F), None, True
    if kind == "meta":
        return np.exp(np.log(1e-5) + rng.standard_normal(17)), rng.integers(1, 18, F).astype(np.int64), False
    raise ValueError(kind)


def _pair(fv, ns, dkind, kkind, seed):
    rng = np.random.default_rng(seed)
    dn = _dirichlet(ns, dkind)
    ps = [fv.Problem.regulargrid(MINS, MAXS, list(ns), dn, lean=lean) for lean in (False, True)]
    assert (ps[0].N, ps[0].F, ps[0].n, ps[0].nnz) == (ps[1].N, ps[1].F, ps[1].n, ps[1].nnz)
    K, meta, logt = _conductivities(ps[0].F, kkind, rng)
    src = np.zeros(ps[0].N)
    free = np.setdiff1d(np.arange(1, ps[0].N + 1), dn)
    src[free[rng.integers(0, len(free), 5)] - 1] = -1e-3 * rng.random(5)
    dh = 1000.0 + rng.random(len(dn))
    for p in ps:
        p.assemble(K, src, dh, metaindex=meta, logtransformconductivity=logt)
    return ps, rng, dn


@pytest.mark.parametrize("kkind", ["faces", "log", "meta"])
def test_lean_assembly_against_the_oracle_directly(fv, oracle, kkind):
    """assembleA / assembleb (/root/reference/src/FiniteVolume.jl:75-155) as restated by the oracle on regulargrid's own face list, against the matrix a
    lean problem writes out from its rows and its b: colptr, rowval, nzval and b bit for bit — no CSR-route problem in between."""
    ns = [17, 16, 18]
    rng = np.random.default_rng(21)
    _, n1, n2, aol, vol = oracle.regulargrid(MINS, MAXS, ns, want_coords=False)
    dn = _dirichlet(ns, "lateral")
    K, meta, logt = _conductivities(len(aol), kkind, rng)
    src = np.zeros(len(vol))
    src[np.setdiff1d(np.arange(len(vol)), dn - 1)[::97]] = rng.standard_normal(len(np.setdiff1d(np.arange(len(vol)), dn - 1)[::97]))
    dh = 1000.0 + rng.random(len(dn))
    mi = None if meta is None else (lambda i: int(meta[i - 1]))
    Ao = oracle.assembleA(n1, n2, aol, K, src, dn, dh, metaindex=mi, logtransformconductivity=logt)
    bo = oracle.assembleb(n1, n2, aol, K, src, dn, dh, metaindex=mi, logtransformconductivity=logt)
    p = fv.Problem.regulargrid(MINS, MAXS, ns, dn, lean=True)
    assert p.lean and p.nnz == len(Ao.nzval)
    p.assemble(K, src, dh, metaindex=meta, logtransformconductivity=logt)
    A = p.csc()
    assert np.array_equal(A.colptr, Ao.colptr) and np.array_equal(A.rowval, Ao.rowval)
    if logt:  # (the device's exp and the host's differ in the last place: the CSR route's own tests allow the same)
        assert np.allclose(A.nzval, Ao.nzval, rtol=1e-14, atol=0) and np.allclose(p.b(), bo, rtol=1e-13, atol=0)
    else:
        assert np.array_equal(A.nzval, Ao.nzval) and np.array_equal(p.b(), bo)
    p.close()


SHAPES = [((20, 18, 70), "xfaces"), ((20, 18, 70), "lateral"), ((12, 30, 66), "xy"), ((36, 182, 186), "xfaces"), ((34, 184, 188), "lateral")]


@pytest.mark.parametrize("ns,dkind", SHAPES)
@pytest.mark.parametrize("kkind", ["uniform", "faces", "log", "meta"])
def test_lean_assembly_and_products_are_the_csr_route_bit_for_bit(fv, ns, dkind, kkind):
    """assembleb's vector, and A x / (A + sigma D) x for random x — in the sliced-DIA form (small boxes) and the symmetric tiled form
    (boxes of > 2^20 rows) — come out with the same bits as from the problem that holds faces and CSR."""
    if kkind in ("log", "meta") and ns[0] > 30:
        pytest.skip("the large boxes take the two plain kinds")
    (p0, p1), rng, dn = _pair(fv, ns, dkind, kkind, seed=sum(ns) + 7 * len(dkind) + len(kkind))
    assert np.array_equal(p0.b(), p1.b())
    A0, A1 = p0.csc(), p1.csc()  # (assembleA's matrix: a lean problem writes it out from its rows, a window at a time)
    assert np.array_equal(A0.colptr, A1.colptr) and np.array_equal(A0.rowval, A1.rowval) and np.array_equal(A0.nzval, A1.nzval)
    del A0, A1
    x = rng.standard_normal(p0.n)
    y0, y1 = p0.spmv(x), p1.spmv(x)
    assert p0.spmv_form()[0] == p1.spmv_form()[0] and np.array_equal(y0, y1)
    u0 = 1000.0 + rng.random(p0.N)
    for p in (p0, p1):
        p.transient_begin(0.1, None, u0)
    for sigma in (1.0 / 60.0, 0.37):
        assert np.array_equal(p0.spmv(x, sigma), p1.spmv(x, sigma)), sigma
    # a second assembly (other conductivities, other heads) refills every form
    K2 = np.array([3e-5])
    for p in (p0, p1):
        p.assemble(K2, np.zeros(p.N), np.full(len(dn), 7.0))
    assert np.array_equal(p0.b(), p1.b()) and np.array_equal(p0.spmv(x, 0.37), p1.spmv(x, 0.37))
    for p in (p0, p1):
        p.close()


@pytest.mark.parametrize("dkind,kkind", [("lateral", "uniform"), ("xfaces", "faces"), ("xy", "uniform")])
def test_lean_transient_runs_are_the_csr_route_bit_for_bit(fv, dkind, kkind):
    """Fixed-dt runs through every regime of the stepping loop (one-iteration fused steps, the many-iteration loop, zero-iteration
    steps, a new dt: every form refilled with another folded shift): heads, iteration counts and the forms that ran are those of
    the problem with faces and CSR."""
    ns = (34, 184, 188) if dkind != "xy" else (36, 130, 260)
    (p0, p1), rng, dn = _pair(fv, ns, dkind, kkind, seed=5)
    # smooth heads (one PCG iteration per small step, as in tests/test_gpu_fused.py): Dirichlet heads on the same smooth field
    c = np.indices(ns).reshape(3, -1).astype(np.float64)
    field = 1000.0 + 0.5 * np.sin(c[0] / 5.0) * np.cos(c[1] / 40.0) + 0.001 * c[2]
    K, meta, logt = _conductivities(p0.F, kkind, np.random.default_rng(6))
    src = np.zeros(p0.N)
    src[np.setdiff1d(np.arange(p0.N), dn - 1)[p0.n // 2]] = -1e-3
    for p in (p0, p1):
        p.assemble(K, src, field[dn - 1] + 0.25, metaindex=meta, logtransformconductivity=logt)
    smooth = field + 1e-3 * rng.random(p0.N)
    DT = 2.0**-10  # far below the diffusion time of a cell: the one-iteration regime
    sched = [(DT, 14, 1e-11), (40.0, 3, 1e-12), (DT, 9, 1e-11), (DT, 6, 1e-3), (300.0, 4, 1e-12), (DT / 2, 7, 1e-12)]
    out = []
    for p in (p0, p1):
        st = p.transient_begin(0.1, None, smooth)
        its = []
        for dt, nsteps, rtol in sched:
            it, info, _ = p.run_fixed(st, dt, nsteps, rtol=rtol, maxiter=2000)
            assert info.converged
            its.append(it.copy())
        out.append((st.node_values(), np.concatenate(its), p.fused_form(), p.spmv_form()[0], p.fused_traversal(), p.loop_form()))
    a, b = out
    assert a[2][0] > 10 and a[2] == b[2] and a[3:] == b[3:], (a[2:], b[2:])
    assert np.array_equal(a[1], b[1]) and (a[1] > 1).any()
    assert np.array_equal(a[0], b[0])
    # a steady solve on the same problems (no shift)
    h0, _, c0 = p0.solve_steady(None, 1e-9, 20000, want_resnorm=False)
    h1, _, c1 = p1.solve_steady(None, 1e-9, 20000, want_resnorm=False)
    assert c0.isconverged and c0.iters == c1.iters and np.array_equal(h0, h1)
    for p in (p0, p1):
        p.close()


def test_lean_problem_rebuilds_what_needs_faces_for_the_call(fv):
    ns = (20, 18, 70)
    (p0, p1), rng, dn = _pair(fv, ns, "xfaces", "uniform", seed=2)
    lib = fv.load()
    grids = []
    for p in (p0, p1):  # (the volumes are kept; the face list is generated again when asked for)
        n1, n2, aol, vol = np.empty(p.F, np.int64), np.empty(p.F, np.int64), np.empty(p.F), np.empty(p.N)
        p.check(lib.fv_problem_get_grid(p.handle, n1.ctypes.data, n2.ctypes.data, aol.ctypes.data, vol.ctypes.data))
        grids.append((n1, n2, aol, vol))
    assert all(np.array_equal(a, b) for a, b in zip(*grids))
    xs, ls = rng.standard_normal(p0.n), rng.standard_normal(p0.n)
    outs = []
    for p in (p0, p1):  # (the parameter gradients read face arrays generated for the call)
        p.transient_begin(0.1, None, np.full(p.N, 1000.0))
        outs.append(p.param_jacobian_apply(xs, ls, scale_by_storage=True))
    assert all(np.array_equal(a, b) for a, b in zip(*outs))
    ctx = fv.default_context()
    assert ctx.get_option(fv._lib.FV_OPT_LEAN_SETUP) == 2  # the default: lean only where the CSR would not fit
    for p in (p0, p1):
        p.close()


def test_lean_adaptive_stepper_and_trajectory_are_the_csr_routes(fv):
    """The step-doubling stepper (a new dt, i.e. a new folded shift and every form refilled, at almost every step) with its states recorded in HBM:
    the same outer steps, times and states as on the problem with faces and CSR."""
    (p0, p1), rng, dn = _pair(fv, (24, 40, 70), "lateral", "faces", seed=12)  # (> 32768 rows: below, the CSR route has its single-launch solver)
    res = []
    for p in (p0, p1):
        st = p.transient_begin(0.1, None, 1000.0 + rng.random(p.N) * 0 + np.linspace(0.0, 2.0, p.N))
        tr = p.new_trajectory()
        p.record(tr, 0.0)
        out = p.run_adaptive(st, 0.0, 2000.0, dt0=1.0, atol=1e-3, rtol=1e-10, maxiter=2000)
        p.record(None)
        res.append((st.free_values(), np.asarray(tr.ts), tr.free_values(len(tr) - 1), out))
        tr.close()
        p.close()
    a, b = res
    assert len(a[1]) > 5 and np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2])


def test_row_blocks_cut_from_a_lean_problem(fv):
    """fv_dist_setup on a lean global problem: it lends itself a CSR for the call, the rank's rows are cut out of it, and it goes back.  Two ranks (host
    threads, the loopback transport) with whole planes each: states, iteration counts and the forms that ran equal those of blocks cut from the CSR route."""
    import threading

    from fvamd import dist

    ns = (34, 184, 188)
    dn = _dirichlet(ns, "lateral")
    N = ns[0] * ns[1] * ns[2]
    rng = np.random.default_rng(17)
    c = np.indices(ns).reshape(3, -1).astype(np.float64)
    field = 1000.0 + 0.5 * np.sin(c[0] / 5.0) * np.cos(c[1] / 40.0) + 0.001 * c[2]
    src = np.zeros(N)
    u0 = field + 1e-3 * rng.random(N)
    lib = fv.load()
    results = {}
    for gid, lean in ((31, False), (32, True)):
        out, errors = [None, None], []

        def worker(rank, lean=lean, gid=gid, out=out, errors=errors):
            try:
                ctx = fv.Context(0)
                dist.comm_init_local(ctx, 2, rank, gid)
                p = fv.Problem.regulargrid(MINS, MAXS, list(ns), dn, ctx, lean=lean)
                p.assemble(np.array([1e-5]), src, field[dn - 1] + 0.25)
                p.transient_begin(0.1, None, u0)
                d3 = p.n // ns[0]
                blk = dist.RowBlock(p, 2, rank, np.array([0, 16, 34]) * d3)
                assert p.lean == lean and p.spmv(np.ones(p.n)).shape == (p.n,)  # (the lean problem is whole again after the loan)
                p.close()
                its = []
                for dt, nsteps, rtol in ((2.0**-10, 12, 1e-11), (40.0, 3, 1e-12)):
                    it, info, _ = blk.run_fixed(dt, nsteps, rtol, maxiter=2000)
                    assert info.converged
                    its.append(it.copy())
                out[rank] = (blk.lo, blk.hi, blk.state(), np.concatenate(its), blk.fused_form())
                blk.close()
                lib.fv_comm_destroy(ctx.handle)
            except BaseException as e:  # noqa: BLE001
                errors.append((rank, repr(e)))

        threads = [threading.Thread(target=worker, args=(r,), daemon=True) for r in range(2)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=300)
        assert not errors, errors
        assert all(not t.is_alive() for t in threads)
        results[lean] = out
    for r in range(2):
        a, b = results[False][r], results[True][r]
        assert a[:2] == b[:2] and np.array_equal(a[3], b[3]) and a[4] == b[4] and np.array_equal(a[2], b[2]), (r, a[3], b[3], a[4], b[4])


def test_lean_amg_is_the_csr_routes_hierarchy(fv):
    """The aggregation-AMG preconditioner (where the reference uses AlgebraicMultigrid: src/FiniteVolume.jl:159-161) on a lean problem: level 0's CSR
    is written out from the rows for the set-up alone and given back; the cycles run level 0 through the problem's own product.  Same hierarchy, same
    iterates: level sizes, iteration counts and heads of a steady solve and of AMG-preconditioned time steps equal the CSR route's bit for bit."""
    (p0, p1), rng, dn = _pair(fv, (36, 182, 186), "xfaces", "faces", seed=9)
    res = []
    for p in (p0, p1):
        p.set_preconditioner("amg")
        rows, nnz = p.amg_info()
        h, _, ch = p.solve_steady(None, 1e-10, 500, want_resnorm=False)
        st = p.transient_begin(0.1, None, np.full(p.N, 1000.0))
        it, info, _ = p.run_fixed(st, 86400.0, 3, rtol=1e-10, maxiter=500)
        assert ch.isconverged and info.converged
        res.append((rows, nnz, ch.iters, h, it.copy(), st.free_values()))
        p.close()
    a, b = res
    assert len(a[0]) >= 3 and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert a[2] == b[2] and np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4]) and np.array_equal(a[5], b[5])


@pytest.mark.parametrize("scattered", [False, True])
def test_lean_problem_with_dirichlet_cells_inside_the_box(fv, scattered):
    """Dirichlet cells inside the box (a column of them through it, or cells sprinkled over it) leave 64-row groups with more than eight distinct
    column offsets: the sliced-DIA form hands such groups to the CSR kernel, and a lean problem has no CSR — it forms and applies their rows on the
    spot (lean_rows_spmv_kernel).  Same operator, other summation kernel: products and transient runs agree with the CSR route to rounding."""
    ns = (24, 40, 70) if scattered else (40, 96, 260)
    rng = np.random.default_rng(3)
    N = ns[0] * ns[1] * ns[2]
    dn = np.sort(rng.choice(N, N // 15, replace=False) + 1).astype(np.int64) if scattered else _dirichlet(ns, "well")
    K = np.exp(np.log(1e-5) + 0.5 * rng.standard_normal(3 * N))  # (one per face: the first F count)
    src = np.zeros(N)
    dh = 1000.0 + rng.random(len(dn))
    x = None
    out = []
    for lean in (False, True):
        p = fv.Problem.regulargrid(MINS, MAXS, list(ns), dn, lean=lean)
        p.assemble(K[: p.F], src, dh)
        x = rng.standard_normal(p.n) if x is None else x
        y0 = p.spmv(x)
        st = p.transient_begin(0.1, None, 1000.0 + rng.random(N) * 0 + np.linspace(0.0, 1.0, N))
        y1 = p.spmv(x, 0.05)
        it, info, _ = p.run_fixed(st, 30.0, 6, rtol=1e-11, maxiter=4000)
        assert info.converged
        out.append((y0, y1, it.copy(), st.free_values(), p.spmv_form()[0], p.b()))
        p.close()
    a, b = out
    scale = np.abs(a[0]).max()
    assert np.array_equal(a[5], b[5]) and a[4] == b[4]
    assert np.abs(a[0] - b[0]).max() < 1e-13 * scale and np.abs(a[1] - b[1]).max() < 1e-13 * np.abs(a[1]).max()
    assert np.abs(a[2].astype(int) - b[2].astype(int)).max() <= 1 and np.abs(a[3] - b[3]).max() < 1e-9 * 1e3
