"""GPU parity of the fused step of the one-iteration regime (fv_fused.hip: the vector update of one fixed-dt step and the
product of the next in one pass; /root/reference/src/transient.jl:60-76,130-134) through the C ABI: against the oracle's
direct solves (1e-8 relative, the north-star bar), and against the unfused K1 + K2S chain it replaces (same iteration
counts, heads to rounding), on boxes whose free rows take the tiled symmetric form: tile edges inside lines and planes,
derived and streamed diagonals, a chain that breaks, steps that are converged at their set-up, runs in chunks."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def relerr(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(b)


def _problem(fv, ns, lateral=False, seed=0, sigma=0.7, uniform_k=False):
    """x-min / x-max planes Dirichlet (every interior row has a zero row sum: diagonal from the arms); lateral: the four
    lateral faces instead (every slice holds a row next to a Dirichlet cell: streamed diagonal)."""
    mins, maxs = [-50.0, -50.0, 0.0], [50.0, 50.0, 10.0]
    coords, nb, aol, vol = fv.regulargrid(mins, maxs, list(ns))
    N = len(vol)
    rng = np.random.default_rng(seed)
    if uniform_k:
        K = np.full(len(aol), 1e-5)
    else:
        K = np.exp(fv.nodehycos2neighborhycos(nb, np.log(1e-5) + sigma * rng.standard_normal(N), True))
    if lateral:
        dn = np.nonzero((coords[1] == mins[1]) | (coords[1] == maxs[1]) | (coords[2] == mins[2]) | (coords[2] == maxs[2]))[0] + 1
    else:
        dn = np.nonzero((coords[0] == mins[0]) | (coords[0] == maxs[0]))[0] + 1
    dn = dn.astype(np.int64)
    smooth = 1000.0 + 0.5 * np.sin(coords[0] / 17.0) * np.cos(coords[1] / 23.0) + 0.1 * coords[2]  # (smooth heads: one PCG iteration per small step)
    dh = smooth[dn - 1] + 0.25
    src = np.zeros(N)
    inner = np.setdiff1d(np.arange(1, N + 1), dn)
    src[inner[len(inner) // 2] - 1] = -1e-3
    u0 = smooth + 1e-3 * rng.random(N)
    return mins, maxs, coords, nb, aol, vol, K, dn, dh, src, u0


def _run(fv, case, fused, schedule, Ss=0.1, tune=()):
    mins, maxs, coords, nb, aol, vol, K, dn, dh, src, u0 = case
    lib = fv.load()
    lib.fv_tune(41, 1 if fused else 0)
    for k, v in tune:
        assert lib.fv_tune(k, v) == 0
    try:
        p = fv.Problem.create(nb, aol, len(vol), dn).assemble(K, src, dh)
        st = p.transient_begin(Ss, vol, u0)
        its = []
        for dt, nsteps, rtol in schedule:
            it, info, _ = p.run_fixed(st, dt, nsteps, rtol=rtol, maxiter=2000)
            assert info.converged
            its.append(it.copy())
        out = (st.node_values(), np.concatenate(its), p.fused_form(), p.spmv_form()[0], p.fused_traversal(), p.loop_form())
        p.close()
    finally:
        lib.fv_tune(41, 1)
        for k, v in tune:
            lib.fv_tune(k, {14: -1, 13: 8, 46: 1, 49: 1, 60: 1, 63: 1}.get(k, 0))
    return out


# The symmetric (tiled) form wants > 2^20 rows in planes of >= 32768 rows, >= 90 % of the slices regular (the first and last plane
# never are).  Free rows: 34 planes x 182 lines x 186 (two column tiles, the second 58 wide; 23 line tiles, the last with 6
# lines; planes of 33 852 rows: slices straddle them); 34 x 165 x 200 (a second tile of 72 columns, a last line tile of 5); and
# 34 x 182 x 186 again with the four lateral faces Dirichlet (every slice holds rows next to a Dirichlet cell).
BOX, BOX2, BOX3 = (36, 182, 186), (36, 165, 200), (34, 184, 188)
BOX4 = (36, 58, 602)  # lines longer than the chunk kernels' block: their halo rows take three rounds of its threads
BOX5 = (36, 40, 900)  # ... four rounds (lines of up to 1024 rows: the tiled product's own limit)
DT = 2.0**-10  # far below the diffusion time of a cell: the one-iteration regime


@pytest.mark.parametrize("ns,lateral", [(BOX, False), (BOX2, False), (BOX3, True)])
def test_fused_steps_against_the_unfused_chain_and_the_oracle(fv, oracle, ns, lateral):
    case = _problem(fv, ns, lateral=lateral)
    mins, maxs, coords, nb, aol, vol, K, dn, dh, src, u0 = case
    dt, nsteps = DT, 21
    fused = _run(fv, case, True, [(dt, nsteps, 1e-11)])
    plain = _run(fv, case, False, [(dt, nsteps, 1e-11)])
    assert fused[3] == 4 and plain[3] == 4  # the tiled symmetric form serves the operator
    assert fused[2][0] >= nsteps - 4 and plain[2][0] == 0  # the fused launches ran (all but the run's first steps)
    n_free = int((np.ones(len(vol), bool)).sum() - len(dn))
    assert fused[2][1] in ((73, 81) if lateral else (73,)) and 0.9 * 73 * n_free <= fused[2][2] <= 81 * n_free  # (some slices stream the diagonal)
    assert np.array_equal(fused[1], plain[1]) and (fused[1][2:] == 1).all()
    assert relerr(fused[0], plain[0]) < 1e-12
    # the oracle: the same steps, its CG run to 1e-14
    ous, ots = oracle.backwardeulerintegrate(u0, (0.0, dt * nsteps), 0.1, vol, nb[:, 0], nb[:, 1], aol, K, src, dn, dh, stepper=oracle.fixedbackwardeulerstep,
                                             dt0=dt, linearsolver=oracle.tightcgsolver(1e-14))
    assert len(ous) == nsteps + 1
    assert relerr(fused[0], ous[-1]) < 1e-8
    print("change over the run, fused vs oracle: %.2e (unfused: %.2e)" % (relerr(fused[0] - u0, ous[-1] - u0), relerr(plain[0] - u0, ous[-1] - u0)))


def test_fused_chain_that_breaks_falls_back_to_further_iterations(fv):
    """fv_tune key 14 treats one chained step of every burst as not converged: the launch behind it turns into the fall-back
    (that step's residual and next direction), the host resumes the step at its second iteration."""
    case = _problem(fv, BOX)
    sched = [(DT, 30, 1e-11)]
    ref = _run(fv, case, False, sched)
    for brk in (0, 3, 7):
        got = _run(fv, case, True, sched, tune=((14, brk),))
        plain = _run(fv, case, False, sched, tune=((14, brk),))
        assert got[2][0] > 0
        assert (got[1] >= 1).all() and (got[1] > 1).sum() >= 2 and np.array_equal(got[1] > 1, plain[1] > 1)
        assert relerr(got[0], ref[0]) < 1e-11


def test_fused_steps_converged_at_their_set_up_and_changing_tolerances(fv):
    """A loose tolerance after tight steps: the steps count 0 iterations and hand the state over unchanged; a tight one
    again: they iterate again.  Time steps that need several iterations in between leave and re-enter the fused regime."""
    case = _problem(fv, BOX, seed=3)
    sched = [(DT, 12, 1e-11), (DT, 20, 1e-3), (DT, 11, 1e-12), (40.0, 5, 1e-10), (DT, 19, 1e-11), (DT / 4, 9, 1e-6)]
    fused = _run(fv, case, True, sched)
    plain = _run(fv, case, False, sched)
    assert fused[2][0] > 30
    assert (fused[1][12:32] == 0).all() and (fused[1][34:43] == 1).all() and (fused[1][43:48] > 1).all()
    assert np.array_equal(fused[1], plain[1])
    assert relerr(fused[0], plain[0]) < 1e-11


def test_fused_runs_in_chunks_and_burst_lengths(fv):
    """One run of 40 steps = runs of 7 + 1 + 13 + 2 + 17 (each call goes on where the previous one stopped, fv_problem::resume),
    whatever the burst length; all-fused runs are the same bits however they are cut."""
    case = _problem(fv, BOX, seed=5)
    dt, rtol = DT, 1e-11
    whole = _run(fv, case, True, [(dt, 40, rtol)])
    plain = _run(fv, case, False, [(dt, 40, rtol)])
    for tune in ((), ((13, 3),), ((13, 32),)):
        cut = _run(fv, case, True, [(dt, 7, rtol), (dt, 1, rtol), (dt, 13, rtol), (dt, 2, rtol), (dt, 17, rtol)], tune=tune)
        assert np.array_equal(cut[1], whole[1])
        assert relerr(cut[0], whole[0]) < 1e-12
    assert relerr(whole[0], plain[0]) < 1e-12


def test_fused_step_with_uniform_storage_and_profile_events(fv):
    """One storage value (no code stream): a slab of equal cells; and the profile events of a burst are all recorded."""
    case = list(_problem(fv, BOX, seed=7, uniform_k=True))
    case[5] = np.full(len(case[5]), case[5].max())  # equal volumes: D takes one value
    case = tuple(case)
    fused = _run(fv, case, True, [(DT, 20, 1e-11)])
    plain = _run(fv, case, False, [(DT, 20, 1e-11)])
    assert fused[2][0] > 10 and np.array_equal(fused[1], plain[1]) and relerr(fused[0], plain[0]) < 1e-12
    mins, maxs, coords, nb, aol, vol, K, dn, dh, src, u0 = case
    p = fv.Problem.create(nb, aol, len(vol), dn).assemble(K, src, dh)
    st = p.transient_begin(0.1, vol, u0)
    p.profile(1)
    it, info, _ = p.run_fixed(st, DT, 20, rtol=1e-11)
    prof = p.profile_get()
    assert info.converged and prof["spmv_dot"][1] >= 18 and prof["spmv_dot"][0] > 0
    p.close()


def test_many_iteration_loop_through_the_fused_kernel(fv, oracle):
    """fv_tune key 46: with several PCG iterations per step the direction update and the product run as one pass of the fused
    kernel (z = M^-1 r kept instead of r between the passes).  Same iteration counts as the K1 + K2 + K3 loop, heads to
    rounding, the oracle's heads within 1e-8; the carried residual of the next step and a steady solve (no shift, residual
    history, warm restart) find what they need."""
    case = _problem(fv, BOX, seed=11)
    mins, maxs, coords, nb, aol, vol, K, dn, dh, src, u0 = case
    sched = [(40.0, 6, 1e-12), (DT, 9, 1e-11), (300.0, 4, 1e-12)]
    on = _run(fv, case, True, sched)
    off = _run(fv, case, True, sched, tune=((46, 0),))
    assert (on[1][:6] > 3).all() and (on[1][-4:] > 3).all()
    assert np.abs(on[1].astype(int) - off[1].astype(int)).max() <= 1 and relerr(on[0], off[0]) < 1e-11
    t = 0.0
    u = u0
    for dt, steps, _ in sched:
        ous, ots = oracle.backwardeulerintegrate(u, (t, t + dt * steps), 0.1, vol, nb[:, 0], nb[:, 1], aol, K, src, dn, dh, stepper=oracle.fixedbackwardeulerstep,
                                                 dt0=dt, linearsolver=oracle.tightcgsolver(1e-14))
        u, t = ous[-1], ots[-1]
    assert relerr(on[0], u) < 1e-8 and relerr(on[0] - u0, u - u0) < 1e-6
    # steady: Jacobi-PCG on A x = b, history and restart from a partial iterate
    lib = fv.load()
    out = {}
    try:
        for key in (1, 0):
            assert lib.fv_tune(46, key) == 0
            p = fv.Problem.create(nb, aol, len(vol), dn).assemble(K, src, dh)
            head, res, ch = p.solve_steady(None, 1e-10, 150, want_resnorm=True)
            assert not ch.isconverged and ch.iters == 150
            head2, res2, ch2 = p.solve_steady(res, 1e-10, 20000, want_resnorm=False)
            assert ch2.isconverged
            out[key] = (np.asarray(ch.data["resnorm"]), head2, ch2.iters, p.spmv_form()[0])
            p.close()
    finally:
        lib.fv_tune(46, 1)
    assert out[1][3] == 4 and len(out[1][0]) == 150
    assert np.allclose(out[1][0], out[0][0], rtol=1e-6, atol=0) and abs(out[1][2] - out[0][2]) <= max(3, out[0][2] // 50)
    assert np.abs(out[1][1] - out[0][1]).max() <= 1e-6 * np.abs(out[0][1]).max()  # both are rtol 1e-10 solves


def test_matrix_as_codes_gives_the_same_bits(fv):
    """fv_tune key 49: with one conductivity on a regular grid each upper diagonal of the operator takes a handful of values and
    the fused kernel reads one 16-bit word of codes per row instead of three doubles — the same doubles out of the tables: heads
    and iteration counts bit for bit those of the run that streams the matrix, in the one-iteration regime and in the
    many-iteration loop; a heterogeneous field keeps the doubles."""
    case = _problem(fv, BOX, seed=13, uniform_k=True)
    sched = [(DT, 17, 1e-11), (40.0, 4, 1e-12), (DT, 10, 1e-11)]
    coded = _run(fv, case, True, sched, tune=((60, 0),))  # (the same 2-D tiles on both sides: the chunk traversal groups the partial sums differently)
    plain = _run(fv, case, True, sched, tune=((49, 0),))
    assert coded[2][0] > 15 and plain[2][0] == coded[2][0]
    assert coded[2][2] < 0.75 * plain[2][2]  # bytes per launch of the storage form: 2 instead of 24 of matrix per row
    assert np.array_equal(coded[1], plain[1]) and np.array_equal(coded[0], plain[0])
    hetero = _problem(fv, BOX, seed=13)
    a = _run(fv, hetero, True, [(DT, 12, 1e-11)])
    b = _run(fv, hetero, True, [(DT, 12, 1e-11)], tune=((49, 0),))
    assert a[2][2] == b[2][2] and np.array_equal(a[0], b[0])


# ------------------------------------------------------------------ the chunk traversal (round 4; fused_chunk_kernel, fv_tune key 60)
@pytest.mark.parametrize("ns,lateral", [(BOX, False), (BOX2, False), (BOX3, True), (BOX4, False), (BOX4, True), (BOX5, False)])
def test_chunk_traversal_against_the_tiles_and_the_oracle(fv, oracle, ns, lateral):
    """With the matrix as codes the fused step walks contiguous chunks of a plane's rows instead of 2-D tiles (no column halos,
    the diagonal of rows next to a Dirichlet cell out of a table by a per-row code, everything that needs a diagonal formed when
    the row's plane is the centre plane).  Same Jacobi-PCG step: identical iteration counts, heads to rounding of the tiles' run
    (partial sums group differently) — one-iteration steps, a many-iteration stretch (the loop through the same kernel), loose
    steps (zero iterations), an injected chain break; the oracle's heads within 1e-8."""
    case = _problem(fv, ns, lateral=lateral, seed=21, uniform_k=True)
    mins, maxs, coords, nb, aol, vol, K, dn, dh, src, u0 = case
    sched = [(DT, 14, 1e-11), (40.0, 3, 1e-12), (DT, 9, 1e-11), (DT, 6, 1e-3), (DT, 7, 1e-12)]
    tiles = _run(fv, case, True, sched, tune=((60, 0),))
    assert tiles[4] == 0 and tiles[2][1] == 51 and tiles[2][0] > 20
    for variant in (1, 2):  # 1: the first / last plane's products formed by the chunk kernel too; 2: those planes by the slice-by-slice launch
        got = _run(fv, case, True, sched, tune=((60, variant),))
        assert got[4] == 1 and got[2][1] == 51 and got[2][0] == tiles[2][0], (variant, got[2], got[4])
        assert np.array_equal(got[1], tiles[1]), (variant, got[1], tiles[1])
        assert relerr(got[0], tiles[0]) < 1e-12, (variant, relerr(got[0], tiles[0]))
    chunks = _run(fv, case, True, sched)
    assert chunks[4] == 1
    for brk in (0, 5):
        a = _run(fv, case, True, sched[:1], tune=((14, brk),))
        b = _run(fv, case, True, sched[:1], tune=((14, brk), (60, 0)))
        assert a[4] == 1 and b[4] == 0 and np.array_equal(a[1], b[1]) and (a[1] > 1).sum() >= 1 and relerr(a[0], b[0]) < 1e-11
    # the many-iteration loop alone: a steady-ish long step from the start state
    la = _run(fv, case, True, [(300.0, 4, 1e-12)])
    lb = _run(fv, case, True, [(300.0, 4, 1e-12)], tune=((60, 0),))
    assert la[4] == 1 and lb[4] == 0 and la[5] == 67 and lb[5] in (76, 83) and np.abs(la[1].astype(int) - lb[1].astype(int)).max() <= 1 and relerr(la[0], lb[0]) < 1e-11
    # the oracle: the same schedule, its CG run to 1e-14 (every step its own solve)
    t, u = 0.0, u0
    for dt, steps, _ in sched:
        ous, ots = oracle.backwardeulerintegrate(u, (t, t + dt * steps), 0.1, vol, nb[:, 0], nb[:, 1], aol, K, src, dn, dh, stepper=oracle.fixedbackwardeulerstep,
                                                 dt0=dt, linearsolver=oracle.tightcgsolver(1e-14))
        u, t = ous[-1], ots[-1]
    print("chunks vs tiles %.2e, chunks vs oracle %.2e (change over the run %.2e)" % (relerr(chunks[0], tiles[0]), relerr(chunks[0], u), relerr(chunks[0] - u0, u - u0)))
    assert relerr(chunks[0], u) < 1e-8


@pytest.mark.parametrize("ns,lateral", [(BOX, False), (BOX2, False), (BOX3, True), (BOX4, False), (BOX4, True), (BOX5, False)])
def test_chunk_traversal_with_the_matrix_as_doubles_against_the_tiles_and_the_oracle(fv, oracle, ns, lateral):
    """Round 5: a heterogeneous conductivity (one value per face, /root/reference/src/FiniteVolume.jl:75-108) has no matrix codes; its
    fused step / pass walks the same chunks with the three upper diagonals streamed as doubles (fused_chunkd_kernel: U2 through an
    LDS ring, U1 by a wave shift, U3 carried in registers, the diagonal of a row next to a Dirichlet cell from the stored one).  Same
    Jacobi-PCG step as the 2-D tiles: identical iteration counts, heads to rounding (partial sums group differently), 73 B per row —
    one-iteration steps, the many-iteration loop, zero-iteration steps, an injected chain break; the oracle's heads within 1e-8."""
    case = _problem(fv, ns, lateral=lateral, seed=31)
    mins, maxs, coords, nb, aol, vol, K, dn, dh, src, u0 = case
    sched = [(DT, 14, 1e-11), (40.0, 3, 1e-12), (DT, 9, 1e-11), (DT, 6, 1e-3), (DT, 7, 1e-12)]
    tiles = _run(fv, case, True, sched, tune=((60, 0),))
    assert tiles[4] == 0 and tiles[2][1] in (73, 81) and tiles[2][0] > 20
    for variant in (1, 2):  # 1: the first / last plane's products formed by the chunk kernel too; 2: those planes by the slice-by-slice launch
        got = _run(fv, case, True, sched, tune=((60, variant),))
        assert got[4] == 1 and got[2][1] == tiles[2][1] and got[2][0] == tiles[2][0], (variant, got[2], got[4])
        assert np.array_equal(got[1], tiles[1]), (variant, got[1], tiles[1])
        assert relerr(got[0], tiles[0]) < 1e-12, (variant, relerr(got[0], tiles[0]))
    chunks = _run(fv, case, True, sched)
    assert chunks[4] == 1
    for brk in (0, 5):
        a = _run(fv, case, True, sched[:1], tune=((14, brk),))
        b = _run(fv, case, True, sched[:1], tune=((14, brk), (60, 0)))
        assert a[4] == 1 and b[4] == 0 and np.array_equal(a[1], b[1]) and (a[1] > 1).sum() >= 1 and relerr(a[0], b[0]) < 1e-11
    la = _run(fv, case, True, [(300.0, 4, 1e-12)])
    lb = _run(fv, case, True, [(300.0, 4, 1e-12)], tune=((60, 0),))
    assert la[4] == 1 and lb[4] == 0 and la[5] == 89 and lb[5] == 105 and np.abs(la[1].astype(int) - lb[1].astype(int)).max() <= 1 and relerr(la[0], lb[0]) < 1e-11
    t, u = 0.0, u0
    for dt, steps, _ in sched:
        ous, ots = oracle.backwardeulerintegrate(u, (t, t + dt * steps), 0.1, vol, nb[:, 0], nb[:, 1], aol, K, src, dn, dh, stepper=oracle.fixedbackwardeulerstep,
                                                 dt0=dt, linearsolver=oracle.tightcgsolver(1e-14))
        u, t = ous[-1], ots[-1]
    print("chunks (doubles) vs tiles %.2e, vs oracle %.2e (change over the run %.2e)" % (relerr(chunks[0], tiles[0]), relerr(chunks[0], u), relerr(chunks[0] - u0, u - u0)))
    assert relerr(chunks[0], u) < 1e-8


@pytest.mark.parametrize("lateral,uniform_k", [(False, False), (True, False), (True, True)])
def test_one_launch_iterations_against_the_pass_and_update_pair_and_the_oracle(fv, oracle, lateral, uniform_k):
    """Round 5 (fv_tune key 63): on whole regular boxes a PCG iteration of the many-iteration loop is ONE launch — the launch takes the
    verdict on the iterate and alpha, beta from sums the previous launch left (the next iterate's r.z and r.r as polynomials in the
    step length), applies z' = z + alpha w and x += alpha p and forms the next direction and product; the last update is flushed when the
    loop has stopped.  Same Jacobi-PCG iteration and stopping rule (/root/reference/src/transient.jl:50-58): iteration counts within one of
    the pass + update pair's, heads to rounding, the oracle's heads within 1e-8; a steady solve with its residual history, a solve that
    runs out of iterations and a warm restart; steps that are converged at their set-up."""
    case = _problem(fv, BOX3 if lateral else BOX, lateral=lateral, seed=41, uniform_k=uniform_k)
    want_on, want_off = (67, (76, 83)) if uniform_k else (89, (105,))  # (the matrix as 16-bit codes: 67 B per row and iteration)
    mins, maxs, coords, nb, aol, vol, K, dn, dh, src, u0 = case
    sched = [(40.0, 6, 1e-12), (DT, 5, 1e-11), (300.0, 4, 1e-12), (40.0, 3, 1e-3), (3.0, 5, 1e-10)]
    on = _run(fv, case, True, sched)
    off = _run(fv, case, True, sched, tune=((63, 0),))
    assert (on[1][:6] > 3).all() and np.abs(on[1].astype(int) - off[1].astype(int)).max() <= 1, (on[1], off[1])
    assert relerr(on[0], off[0]) < 1e-11, relerr(on[0], off[0])
    t, u = 0.0, u0
    for dt, steps, _ in sched[:3]:
        ous, ots = oracle.backwardeulerintegrate(u, (t, t + dt * steps), 0.1, vol, nb[:, 0], nb[:, 1], aol, K, src, dn, dh, stepper=oracle.fixedbackwardeulerstep,
                                                 dt0=dt, linearsolver=oracle.tightcgsolver(1e-14))
        u, t = ous[-1], ots[-1]
    # loose steps behind several-iteration steps: the step after a deferred flush is converged at its set-up, and the one after that must find the
    # residual where the flush + set-up kernel left it (tools/ploop_fuzz.py, seed 7)
    loose = [(600.0, 8, 1e-3), (1.0, 5, 1e-8), (DT, 6, 1e-12), (60.0, 8, 1e-3), (4000.0, 3, 1e-12)]
    lon = _run(fv, case, True, loose)
    loff = _run(fv, case, True, loose, tune=((63, 0),))
    assert np.abs(lon[1].astype(int) - loff[1].astype(int)).max() <= 1 and relerr(lon[0], loff[0]) < 1e-10, (lon[1], loff[1], relerr(lon[0], loff[0]))
    tight = _run(fv, case, True, sched[:3])
    assert relerr(tight[0], u) < 1e-8 and relerr(tight[0] - u0, u - u0) < 1e-6
    tight_off = _run(fv, case, True, sched[:3], tune=((63, 0),))  # (the loop forms of a schedule that ends with many-iteration steps)
    assert tight[5] == want_on and tight_off[5] in want_off, (tight[5], tight_off[5])
    lib = fv.load()
    out = {}
    try:
        for key in (1, 0):
            assert lib.fv_tune(63, key) == 0
            p = fv.Problem.create(nb, aol, len(vol), dn).assemble(K, src, dh)
            p.solve_steady(None, 1e-10, 3)  # (the first product of a problem establishes the storage form the loop asks for)
            head, res, ch = p.solve_steady(None, 1e-10, 150, want_resnorm=True)
            assert not ch.isconverged and ch.iters == 150
            head2, res2, ch2 = p.solve_steady(res, 1e-10, 20000, want_resnorm=False)
            assert ch2.isconverged
            out[key] = (np.asarray(ch.data["resnorm"]), head2, ch2.iters, p.loop_form())
            p.close()
    finally:
        lib.fv_tune(63, 1)
    assert out[1][3] == want_on and out[0][3] in want_off and len(out[1][0]) == 150
    assert np.allclose(out[1][0], out[0][0], rtol=1e-6, atol=0), np.abs(out[1][0] / out[0][0] - 1).max()
    assert abs(out[1][2] - out[0][2]) <= max(3, out[0][2] // 50)
    assert np.abs(out[1][1] - out[0][1]).max() <= 1e-6 * np.abs(out[0][1]).max()  # both are rtol 1e-10 solves


def test_many_iteration_loop_leaves_a_row_with_a_zero_diagonal_alone(fv):
    """a free cell all of whose faces carry conductance 0 has a zero diagonal in a steady solve, M^-1 = 0 by definition
    (fv_pcg.hip: jacobi diagonal), and the loop that keeps z = M^-1 r and recovers r as z / M^-1 would form 0 / 0 there.  The loop
    through the fused kernel is therefore gated on M^-1 > 0 like the z-form K2S; the classic K1 + K2 + K3 loop runs and leaves the
    row where it started, as it does with fv_tune 46 = 0 — same residual history, same heads, nothing NaN."""
    case = _problem(fv, BOX, seed=17)
    mins, maxs, coords, nb, aol, vol, K, dn, dh, src, u0 = case
    N = len(vol)
    n1, n2, n3 = BOX
    cell = ((n1 // 2) * n2 + n2 // 2) * n3 + n3 // 2 + 1  # an interior cell, 1-based
    K = K.copy()
    K[(nb[:, 0] == cell) | (nb[:, 1] == cell)] = 0.0
    src = np.zeros(N)
    lib = fv.load()
    out = {}
    try:
        for key in (1, 0):
            assert lib.fv_tune(46, key) == 0
            p = fv.Problem.create(nb, aol, N, dn).assemble(K, src, dh)
            head, res, ch = p.solve_steady(None, 1e-9, 400, want_resnorm=True)
            out[key] = (head, np.asarray(ch.data["resnorm"]), ch.iters, p.loop_form(), p.spmv_form()[0])
            p.close()
    finally:
        lib.fv_tune(46, 1)
    assert out[1][4] == 4  # the tiled symmetric form serves the operator: the loop through the fused kernel WOULD be taken
    assert out[1][3] == 0  # ... and is not: a row with M^-1 = 0
    assert np.isfinite(out[1][0]).all() and np.isfinite(out[1][1]).all()
    assert out[1][0][cell - 1] == 0.0  # the isolated row stays at the x0 = 0 it started from
    assert out[1][2] == out[0][2] and np.array_equal(out[1][1], out[0][1]) and np.array_equal(out[1][0], out[0][0])


# ------------------------------------------------------------------ the fused step on row blocks (loopback transport)
def _run_row_blocks(fv, case, nranks, group_id, schedule, planes_per_rank, Ss=0.1, tune=()):
    """One host thread per rank (own context on device 0, fv_comm_init_local), whole planes per rank: the row-block driver
    with the fused step inside its bursts.  -> per rank (lo, hi, state, iterations, fused form)."""
    import threading

    from fvamd import dist

    mins, maxs, coords, nb, aol, vol, K, dn, dh, src, u0 = case
    lib = fv.load()
    for k, v in tune:
        assert lib.fv_tune(k, v) == 0
    out, errors = [None] * nranks, []

    def worker(rank):
        try:
            ctx = fv.Context(0)
            dist.comm_init_local(ctx, nranks, rank, group_id)
            p = fv.Problem.create(nb, aol, len(vol), dn, ctx).assemble(K, src, dh)
            p.transient_begin(Ss, vol, u0)
            d3 = p.n // sum(planes_per_rank)
            assert d3 * sum(planes_per_rank) == p.n
            bounds = np.concatenate([[0], np.cumsum(planes_per_rank)]) * d3
            blk = dist.RowBlock(p, nranks, rank, bounds)
            p.close()
            its = []
            for dt, nsteps, rtol in schedule:
                it, info, _ = blk.run_fixed(dt, nsteps, rtol, maxiter=2000)
                assert info.converged
                its.append(it.copy())
            out[rank] = (blk.lo, blk.hi, blk.state(), np.concatenate(its), blk.fused_form())
            blk.close()
            lib.fv_comm_destroy(ctx.handle)
        except BaseException as e:  # noqa: BLE001  (a failing rank would leave the others waiting at a barrier)
            errors.append((rank, repr(e)))

    try:
        threads = [threading.Thread(target=worker, args=(r,), daemon=True) for r in range(nranks)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=300)
    finally:
        for k, v in tune:
            lib.fv_tune(k, {14: -1, 13: 8, 49: 1, 50: 1, 60: 1}.get(k, 0))
    assert not errors, errors
    assert all(not t.is_alive() for t in threads), "a rank did not finish (deadlock in the protocol?)"
    return out


def _gather(out, n):
    got = np.empty(n)
    for lo, hi, state, _, _ in out:
        got[lo:hi] = state
    return got


# two blocks of 34 planes, three of 34 / 36 / 34 (each > 2^20 rows: the tiled symmetric form serves a block's interior window)
@pytest.mark.parametrize("planes", [(34, 34), (34, 36, 34)])
def test_fused_steps_on_row_blocks_against_one_gpu(fv, planes):
    """fv_dist_run_fixed with the fused step in its bursts (per step: one 6-double collective, z' of the send rows formed
    from z and v before the launch, the boundary groups' products after the halo): same iteration counts on every rank as
    the single-GPU fused run, heads to rounding; the unfused row-block run (fv_tune 50 = 0) agrees as well."""
    ns = (sum(planes) + 2, 182, 186)
    case = _problem(fv, ns)
    sched = [(DT, 21, 1e-11), (1.0, 2, 1e-11), (DT, 12, 1e-11)]
    one = _run(fv, case, True, sched)
    free = np.ones(len(case[5]), bool)
    free[case[7] - 1] = False
    want = one[0][free]
    nranks = len(planes)
    res = _run_row_blocks(fv, case, nranks, 7000 + nranks, sched, planes)
    for lo, hi, state, its, form in res:
        assert np.array_equal(its, one[1]), (its, one[1])
        assert form[0] >= 21 + 12 - 10 and form[1] == 73, form  # the fused launches ran on every rank
    assert relerr(_gather(res, len(want)), want) < 1e-12
    plain = _run_row_blocks(fv, case, nranks, 7100 + nranks, sched, planes, tune=((50, 0),))
    assert all(r[4][0] == 0 for r in plain) and all(np.array_equal(r[3], one[1]) for r in plain)
    assert relerr(_gather(plain, len(want)), want) < 1e-12
    print("row blocks, fused vs unfused run: %.2e; vs one GPU: %.2e" % (relerr(_gather(res, len(want)), _gather(plain, len(want))), relerr(_gather(res, len(want)), want)))


def test_fused_row_block_chain_that_breaks_and_steps_converged_at_their_set_up(fv):
    """The injected break (fv_tune 14) at a burst's first, a middle and its last step, on every rank at the same place; a loose
    tolerance after tight steps (zero-iteration steps: alpha = 0 on every rank); uniform conductivity: the matrix as codes."""
    planes = (34, 34)
    ns = (sum(planes) + 2, 182, 186)
    case = _problem(fv, ns, seed=2)
    sched = [(DT, 30, 1e-11)]
    ref = _run_row_blocks(fv, case, 2, 7200, sched, planes, tune=((50, 0),))
    want = _gather(ref, ref[-1][1])
    for brk in (0, 3, 7):
        got = _run_row_blocks(fv, case, 2, 7210 + brk, sched, planes, tune=((14, brk),))
        plain = _run_row_blocks(fv, case, 2, 7220 + brk, sched, planes, tune=((14, brk), (50, 0)))
        for g, q in zip(got, plain):
            assert g[4][0] > 0 and q[4][0] == 0
            assert (g[3] >= 1).all() and (g[3] > 1).sum() >= 2 and np.array_equal(g[3] > 1, q[3] > 1), (brk, g[3], q[3])
            assert np.array_equal(g[3], got[0][3])
        assert relerr(_gather(got, len(want)), want) < 1e-11
    sched2 = [(DT, 12, 1e-11), (DT, 20, 1e-3), (DT, 10, 1e-12)]
    a = _run_row_blocks(fv, case, 2, 7300, sched2, planes)
    b = _run_row_blocks(fv, case, 2, 7301, sched2, planes, tune=((50, 0),))
    for x, y in zip(a, b):
        assert np.array_equal(x[3], y[3]) and (x[3][12:32] == 0).sum() >= 15 and x[4][0] > 0
    assert relerr(_gather(a, len(want)), _gather(b, len(want))) < 1e-12
    cu = _problem(fv, ns, seed=2, uniform_k=True)
    c = _run_row_blocks(fv, cu, 2, 7310, sched, planes)
    d = _run_row_blocks(fv, cu, 2, 7311, sched, planes, tune=((50, 0),))
    assert all(r[4][1] == 51 for r in c), [r[4] for r in c]  # 73 - 24 + 2: the three upper diagonals as one 16-bit word
    assert all(np.array_equal(x[3], y[3]) for x, y in zip(c, d))
    assert relerr(_gather(c, len(want)), _gather(d, len(want))) < 1e-12


def test_time_steps_that_are_not_powers_of_two_keep_the_derived_diagonal(fv):
    """The shifted diagonal is re-derived bit for bit as -(sum of the arms) + round(D / dt) only if the fold formed it with two
    roundings as well (not one FMA): with a heterogeneous conductivity and dt = 1e-3 (1 / dt inexact) every slice away from the
    Dirichlet planes must still do without the diagonal stream — 73 B per row in the fused step, 41 in K1 — and agree with the run
    that streams it (fv_tune 37 = 0)."""
    case = _problem(fv, BOX, seed=5)
    sched = [(1.0e-3, 24, 1e-11), (3.0e-3, 10, 1e-11)]
    a = _run(fv, case, True, sched)
    n_free = len(case[5]) - len(case[7])
    assert a[2][0] >= 20 and a[2][1] == 73, a[2]
    lib = fv.load()
    lib.fv_tune(37, 0)
    try:
        b = _run(fv, case, True, sched)
    finally:
        lib.fv_tune(37, 1)
    assert b[2][1] == 81 and np.array_equal(a[1], b[1])
    assert relerr(a[0], b[0]) < 1e-13
    p = fv.Problem.create(case[3], case[4], len(case[5]), case[7]).assemble(case[6], case[9], case[8])
    st = p.transient_begin(0.1, case[5], case[10])
    p.run_fixed(st, 1.0e-3, 4, rtol=1e-11, maxiter=100)
    form = p.spmv_form()
    assert form[0] == 4 and form[2] < 43 * n_free, form  # 24 (three upper diagonals) + 1 (storage code) + 16 (x, y) per row
    p.close()


def test_ranks_agree_on_the_fused_step_before_they_use_it(fv):
    """A block of two planes cannot run the fused step, its neighbour of 34 planes could: the two forms issue their collectives
    and halo exchanges in different orders, so the ranks agree at the start of every fv_dist_run_fixed call (one all-reduce) and
    both keep K1 + K2S — no launch of the fused kernel anywhere, no hang, the single-GPU heads."""
    planes = (34, 2)
    ns = (sum(planes) + 2, 182, 186)
    case = _problem(fv, ns, seed=4)
    sched = [(DT, 19, 1e-11), (DT, 6, 1e-11)]
    one = _run(fv, case, True, sched)
    free = np.ones(len(case[5]), bool)
    free[case[7] - 1] = False
    want = one[0][free]
    res = _run_row_blocks(fv, case, 2, 7400, sched, planes)
    assert all(r[4][0] == 0 for r in res), [r[4] for r in res]
    assert all(np.array_equal(r[3], one[1]) for r in res)
    assert relerr(_gather(res, len(want)), want) < 1e-12


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_fused_row_blocks_on_random_schedules_against_the_unfused_driver(fv, seed):
    """Differential fuzz of the row-block driver with the fused step: random sequences of (dt, steps, rtol) — one-iteration runs,
    runs with several iterations per step, loose tolerances (zero-iteration steps), short and long calls, an injected chain break
    on some seeds — with fv_tune 50 on and off: the same iteration counts on every rank, heads to rounding."""
    rng = np.random.default_rng(4200 + seed)
    planes = (34, 34) if seed % 2 == 0 else (34, 35, 34)
    ns = (sum(planes) + 2, 182, 186)
    case = _problem(fv, ns, seed=10 + seed)
    sched = []
    for _ in range(int(rng.integers(3, 6))):
        dt = float(rng.choice([DT, DT, 3 * DT, 0.7e-3, 1.0]))
        sched.append((dt, int(rng.integers(1, 26)), float(rng.choice([1e-11, 1e-11, 1e-9, 1e-3]))))
    tune = ((14, int(rng.integers(0, 8))),) if seed >= 2 else ()
    nranks = len(planes)
    a = _run_row_blocks(fv, case, nranks, 7500 + 10 * seed, sched, planes, tune=tune)
    b = _run_row_blocks(fv, case, nranks, 7501 + 10 * seed, sched, planes, tune=tune + ((50, 0),))
    n = a[-1][1]
    for x, y in zip(a, b):
        assert np.array_equal(x[3], y[3]), (sched, tune, x[3], y[3])
        assert y[4][0] == 0
    assert relerr(_gather(a, n), _gather(b, n)) < 1e-11, (sched, tune)
    # (the fused launches run wherever a call holds a burst of one-iteration steps: certainly in a tight run of 8 small steps or more)
    assert any(x[4][0] > 0 for x in a) or not any(dt < 1.0 and k >= 8 and rtol <= 1e-11 for dt, k, rtol in sched), (sched, [x[4] for x in a])


def test_minv_as_codes_in_the_many_iteration_loop_gives_the_same_bits(fv):
    """fv_tune key 59: where the Jacobi diagonal takes at most 16 distinct values (one conductivity on a regular grid) the vector
    pass of the many-iteration loop reads M^-1 as a code byte — the same doubles out of the table: iteration counts and heads bit
    for bit, 84 instead of 91 B per row and iteration; a heterogeneous field keeps the stream."""
    case = _problem(fv, BOX, seed=17, uniform_k=True)
    mins, maxs, coords, nb, aol, vol, K, dn, dh, src, u0 = case

    def run(case, key):
        lib = fv.load()
        assert lib.fv_tune(59, key) == 0 and lib.fv_tune(63, 0) == 0  # (the pass + vector update pair: the one-launch iteration of round 5 reads no M^-1 at all)
        try:
            p = fv.Problem.create(case[3], case[4], len(case[5]), case[7]).assemble(case[6], case[9], case[8])
            st = p.transient_begin(0.1, case[5], case[10])
            its = [p.run_fixed(st, dt, k, rtol=1e-12, maxiter=3000)[0].copy() for dt, k in ((40.0, 5), (7.0, 4), (300.0, 3))]
            out = (st.node_values(), np.concatenate(its), p.loop_form())
            p.close()
        finally:
            lib.fv_tune(59, 1)
            lib.fv_tune(63, 1)
        return out

    a, b = run(case, 1), run(case, 0)
    assert (a[1] > 3).all() and a[2] == 76 and b[2] == 83, (a[1], a[2], b[2])
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0])
    hetero = _problem(fv, BOX, seed=17)
    c, d = run(hetero, 1), run(hetero, 0)
    assert c[2] == 105 and d[2] == 105 and np.array_equal(c[0], d[0])
