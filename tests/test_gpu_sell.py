"""GPU parity of the SELL-64 form with 16-bit column offsets (fv_spmv.hip: spmv_sell_kernel, FV_SPMV_SELL) that serves the 64-row
groups of irregular face-list meshes after the locality re-numbering — the `A * x` inside cg! of /root/reference/src/FiniteVolume.jl:161
and src/transient.jl:52 on the reference's DFN meshes (examples/fractures) — against the oracle's CSR product and against the
library's own CSR wave-stream kernel (fv_tune 54 = 0): products, dots, fixed-dt runs; groups that do not fit the form (rows of
more than 32 entries, neighbours further than 32767 rows away) stay with the CSR kernel inside the same launch set."""
import numpy as np
import pytest

from tests import workloads

pytestmark = pytest.mark.gpu


def relerr(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(b)


def _mesh(nfrac=4, m=320, seed=3, hubs=0, far=0):
    """fractures-like mesh (cells numbered at random inside each fracture); hubs: cells tied to 40 others (rows too long for the
    form); far: extra ties between the first and the last fracture (neighbours far apart even after the re-numbering)."""
    w = workloads.fractures_like(nfrac, m, seed=seed)
    rng = np.random.default_rng(seed + 100)
    n1, n2, aol = [w["node1"]], [w["node2"]], [w["aol"]]
    per = m * m
    for h in range(hubs):
        c = int(rng.integers(1, w["N"] + 1))
        others = rng.choice(np.arange(1, w["N"] + 1), 40, replace=False)
        others = others[others != c]
        n1.append(np.minimum(c, others))
        n2.append(np.maximum(c, others))
        aol.append(np.exp(rng.uniform(np.log(3e-9), np.log(3e-5), len(others))))
    if far:
        a = rng.choice(np.arange(1, per + 1), far, replace=False)
        b = (nfrac - 1) * per + rng.choice(np.arange(1, per + 1), far, replace=False)
        n1.append(a)
        n2.append(b)
        aol.append(np.exp(rng.uniform(np.log(3e-9), np.log(3e-5), far)))
    n1, n2, aol = np.concatenate(n1), np.concatenate(n2), np.concatenate(aol)
    w = dict(w, node1=n1.astype(np.int64), node2=n2.astype(np.int64), aol=aol, K=np.full(len(aol), 1e-12))
    return w


def _problem(fv, w):
    p = fv.Problem.create((w["node1"], w["node2"]), w["aol"], w["N"], w["dnodes"])
    p.assemble(w["K"], np.zeros(w["N"]), w["dheads"])
    return p


@pytest.mark.parametrize("hubs,far", [(0, 0), (6, 0), (0, 200), (5, 50)])
def test_sell_products_against_the_oracle_and_the_csr_kernel(fv, oracle, hubs, far):
    w = _mesh(hubs=hubs, far=far)
    lib = fv.load()
    p = _problem(fv, w)
    assert p.reorder_info()["reordered"]
    oA = oracle.assembleA(w["node1"], w["node2"], w["aol"], w["K"], np.zeros(w["N"]), w["dnodes"], w["dheads"])
    rng = np.random.default_rng(0)
    x = rng.standard_normal(p.n)
    want = oA.matvec(x)
    y = p.spmv(x)
    form = p.spmv_form()
    assert form[0] == 5, form  # FV_SPMV_SELL
    assert relerr(y, want) < 1e-14
    # the form's bytes: 10 per stored entry + 16 n + 5 per group (+ the CSR share of the groups left out), below the CSR accounting
    assert form[2] < 12 * p.nnz + 20 * p.n
    # with the storage term (unfolded: sigma D x added in the kernel) and against the CSR kernel bit for bit where the sums have the same order
    p.transient_begin(1e-9, w["volumes"], np.full(w["N"], 1.5e6))
    ys = p.spmv(x, 3.0)
    lib.fv_tune(54, 0)
    try:
        q = _problem(fv, w)
        q.transient_begin(1e-9, w["volumes"], np.full(w["N"], 1.5e6))
        yc = q.spmv(x)
        ysc = q.spmv(x, 3.0)
        assert q.spmv_form()[0] == 0
        q.close()
    finally:
        lib.fv_tune(54, 1)
    assert relerr(y, yc) < 1e-15 and relerr(ys, ysc) < 1e-15
    print("hubs %d far %d: SELL %d B per row against %d of the CSR accounting" % (hubs, far, form[2] // p.n, (12 * p.nnz + 20 * p.n) // p.n))
    p.close()


@pytest.mark.parametrize("hubs,far", [(0, 0), (5, 50)])
def test_sell_fixed_dt_runs_and_steady_solve_match_the_csr_path(fv, hubs, far):
    """Transient steps in the one-iteration regime and with several iterations per step, and a steady Jacobi-PCG solve: same
    iteration counts as with the CSR kernel, heads to rounding."""
    w = _mesh(hubs=hubs, far=far)
    lib = fv.load()
    out = {}
    for sell in (1, 0):
        lib.fv_tune(54, sell)
        try:
            p = _problem(fv, w)
            st = p.transient_begin(1e-9, w["volumes"], np.full(w["N"], 1.5e6))
            its = []
            for dt, k in ((1.0, 12), (1e4, 3), (1.0, 9)):
                it, info, _ = p.run_fixed(st, dt, k, rtol=1e-12, maxiter=5000)
                assert info.converged
                its.append(it.copy())
            heads = st.node_values()
            form = p.spmv_form()[0]
            head_s, res, ch = p.solve_steady(None, 1e-10, 20000, want_resnorm=False)
            assert ch.isconverged
            out[sell] = (heads, np.concatenate(its), form, head_s, ch.iters)
            p.close()
        finally:
            lib.fv_tune(54, 1)
    assert out[1][2] == 5 and out[0][2] == 0
    assert np.array_equal(out[1][1], out[0][1]), (out[1][1], out[0][1])
    assert relerr(out[1][0], out[0][0]) < 1e-12
    assert abs(out[1][4] - out[0][4]) <= 2 and relerr(out[1][3], out[0][3]) < 1e-8


@pytest.mark.parametrize("hubs,far,uniform_vol", [(0, 0, False), (5, 50, False), (0, 0, True)])
def test_fused_step_on_the_sell_form_against_the_unfused_pair_and_the_csr_path(fv, oracle, hubs, far, uniform_vol):
    """fused_sell_step_kernel in the bursts of a fixed-dt run (fv_tune 55): same iteration counts as the SpMV + K2S pair on the same
    form and on the CSR form, heads to rounding; the storage term as a stream (arbitrary cell volumes) and as codes (one volume);
    groups outside the form take the classic product + conversion; an injected chain break falls back to further iterations; the
    oracle's direct solves for the same steps."""
    w = _mesh(nfrac=3, m=300, hubs=hubs, far=far)
    if uniform_vol:
        w = dict(w, volumes=np.full(w["N"], 3e-3))
    lib = fv.load()
    sched = [(1.0, 26, 1e-12), (1e4, 2, 1e-12), (1.0, 11, 1e-12)]

    def run(tune):
        for k, v in tune:
            assert lib.fv_tune(k, v) == 0
        try:
            p = _problem(fv, w)
            st = p.transient_begin(1e-9, w["volumes"], np.full(w["N"], 1.5e6))
            its = [p.run_fixed(st, dt, k, rtol=rtol, maxiter=5000)[0].copy() for dt, k, rtol in sched]
            out = (st.node_values(), np.concatenate(its), p.fused_form(), p.spmv_form()[0])
            p.close()
        finally:
            for k, v in tune:
                lib.fv_tune(k, {14: -1}.get(k, 1))
        return out

    fused = run(())
    pair = run(((55, 0),))
    csr = run(((54, 0),))
    assert fused[3] == 5 and pair[3] == 5 and csr[3] == 0
    assert fused[2][0] >= 20 and pair[2][0] == 0 and csr[2][0] == 0, (fused[2], pair[2], csr[2])
    per_row = fused[2][1]
    assert (115 <= per_row <= 126) if uniform_vol else (122 <= per_row <= 134), fused[2]
    assert np.array_equal(fused[1], pair[1]) and np.array_equal(fused[1], csr[1]), (fused[1], pair[1], csr[1])
    assert relerr(fused[0], pair[0]) < 1e-12 and relerr(fused[0], csr[0]) < 1e-12
    for brk in (0, 4):
        b = run(((14, brk),))
        c = run(((14, brk), (55, 0)))
        assert b[2][0] > 0 and (b[1] > 1).sum() >= 2 and np.array_equal(b[1] > 1, c[1] > 1), (brk, b[1], c[1])
        assert relerr(b[0], fused[0]) < 1e-11
    # the oracle: the first 8 steps with its own CG at 1e-14
    u0 = np.full(w["N"], 1.5e6)
    ous, _ = oracle.backwardeulerintegrate(u0, (0.0, 8.0), 1e-9, w["volumes"], w["node1"], w["node2"], w["aol"], w["K"], np.zeros(w["N"]), w["dnodes"], w["dheads"],
                                           stepper=oracle.fixedbackwardeulerstep, dt0=1.0, linearsolver=oracle.tightcgsolver(1e-14))
    p = _problem(fv, w)
    st = p.transient_begin(1e-9, w["volumes"], u0)
    p.run_fixed(st, 1.0, 8, rtol=1e-12, maxiter=5000)
    assert relerr(st.node_values(), ous[-1]) < 1e-8
    print("fused step on SELL: %d B per row; change over 8 steps vs oracle %.2e" % (per_row, relerr(st.node_values() - u0, ous[-1] - u0)))
    p.close()


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_fused_sell_step_on_random_schedules_against_the_unfused_pair(fv, seed):
    """Differential fuzz: random sequences of (dt, steps, rtol) on a fractures-like mesh with hub rows and far ties, an injected
    chain break on some seeds, the fused step on the SELL form on and off (fv_tune 55): same iteration counts, heads to rounding."""
    rng = np.random.default_rng(900 + seed)
    w = _mesh(nfrac=3, m=280, seed=20 + seed, hubs=int(rng.integers(0, 4)), far=int(rng.integers(0, 60)))
    lib = fv.load()
    sched = [(float(rng.choice([1.0, 1.0, 3.0, 0.25, 1e4])), int(rng.integers(1, 24)), float(rng.choice([1e-12, 1e-12, 1e-9, 1e-4]))) for _ in range(int(rng.integers(3, 6)))]
    tune = ((14, int(rng.integers(0, 8))),) if seed % 2 else ()

    def run(extra):
        for k, v in tune + extra:
            assert lib.fv_tune(k, v) == 0
        try:
            p = _problem(fv, w)
            st = p.transient_begin(1e-9, w["volumes"], np.full(w["N"], 1.5e6))
            its = [p.run_fixed(st, dt, k, rtol=rtol, maxiter=5000)[0].copy() for dt, k, rtol in sched]
            out = (st.node_values(), np.concatenate(its), p.fused_form()[0])
            p.close()
        finally:
            for k, v in tune + extra:
                lib.fv_tune(k, {14: -1}.get(k, 1))
        return out

    a, b = run(()), run(((55, 0),))
    assert b[2] == 0
    assert np.array_equal(a[1], b[1]), (sched, tune, a[1], b[1])
    assert relerr(a[0], b[0]) < 1e-11, (sched, tune)
