"""Locality re-numbering of the free cells inside the library (fv_problem_create, face-list meshes): forced on
(fv_tune(31, 2)) for small cases it must be invisible at the C ABI — fv_get_csc / fv_get_b / the free maps bit for bit the
oracle's, every free-indexed vector in the caller's numbering, heads as before — and on the 5M-cell fractures-like mesh
(cells numbered at random inside each fracture) it must kick in by itself and pay."""
import math
import os

import ctypes as C

import numpy as np
import pytest

from tests import refcases, workloads
from tests import test_gpu_assembly as tga
from tests import test_gpu_solve as tgs

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture
def forced(fv):
    lib = fv.load()
    assert lib.fv_tune(31, 2) == 0
    yield lib
    lib.fv_tune(31, 1)


@pytest.mark.parametrize("seed", range(12))
def test_assembly_fuzz_is_bit_exact_under_the_renumbering(fv, oracle, forced, seed):
    tga.test_random_multigraph_assembly_bit_exact(fv, oracle, seed)


def test_fixtures_are_bit_exact_under_the_renumbering(fv, oracle, forced):
    tga.test_fourfractures_fixture_bit_exact(fv, oracle)
    tga.test_random_multigraph_with_repeats_selfloops_metaindex(fv, oracle)
    tga.test_theis_grid_structure(fv, oracle)
    tga.test_empty_and_all_dirichlet(fv)


def test_solves_and_adjoint_workflows_under_the_renumbering(fv, oracle, forced):
    """Steady and transient heads against the oracle, a time-dependent getb (free-indexed host vectors per step), the
    adjoint step (free-indexed bhat), the device gradient integral (free-indexed knots in, per-row terms out), the AMG
    cycle through fv_amg_apply — all through the mirrored reference API, which only ever sees the caller's numbering."""
    tgs.test_fourfractures_steady_vs_direct_and_pflotran(fv, oracle)
    tgs.test_fixed_steps_vs_oracle_direct(fv, oracle)
    tgs.test_time_dependent_getb_method(fv, oracle)
    tgs.test_onenode_adjoint_lambda(fv)
    tgs.test_adjoint_step_matches_transposed_scaled_operator(fv, oracle)
    tgs.test_device_gradient_integral_vs_host_simpson_of_the_jacobians(fv, True)
    tgs.test_amg_cycle_is_symmetric_positive_definite(fv)
    tgs.test_amg_on_a_random_multigraph_with_isolated_and_zero_rows(fv, oracle)
    tgs.test_spmv_long_and_short_rows(fv, oracle)


def test_the_numbering_really_changed_and_vectors_cross_in_the_callers_order(fv, oracle, forced):
    d = np.load(os.path.join(GOLDEN, "fourfractures.npz"))
    nb = np.stack([d["node1"], d["node2"]], 1)
    rng = np.random.default_rng(5)
    shuffle = rng.permutation(2106)  # number the cells at random, as a DFN generator would
    rank = np.empty(2106, np.int64)
    rank[shuffle] = np.arange(2106)
    nb2 = rank[nb - 1] + 1
    dn2 = rank[d["dirichletnodes"] - 1] + 1
    p = fv.Problem.create(nb2, d["areasoverlengths"], 2106, dn2)
    info = p.reorder_info()
    assert info["reordered"] and info["mean_after"] * 4 < info["mean_before"], info
    src = np.zeros(2106)
    p.assemble(d["conductivities"], src, d["dirichletheads"])
    oA = oracle.assembleA(nb2[:, 0], nb2[:, 1], d["areasoverlengths"], d["conductivities"], src, dn2, d["dirichletheads"])
    A = p.csc()
    assert np.array_equal(A.colptr, oA.colptr) and np.array_equal(A.rowval, oA.rowval) and np.array_equal(A.nzval, oA.nzval)
    assert np.array_equal(p.b(), oracle.assembleb(nb2[:, 0], nb2[:, 1], d["areasoverlengths"], d["conductivities"], src, dn2, d["dirichletheads"]))
    freenode, n2f = p.free_maps()
    ofree, on2f = oracle.getfreenodes(2106, dn2)
    assert np.array_equal(freenode, ofree) and np.array_equal(n2f, on2f)
    # free-indexed vectors: y = A x in the caller's numbering, state round trip, node views
    import scipy.sparse as sp

    As = sp.csc_matrix((oA.nzval, oA.rowval - 1, oA.colptr - 1), shape=(oA.n, oA.n)).tocsr()
    x = rng.standard_normal(p.n)
    y = p.spmv(x)
    assert np.abs(y - As @ x).max() <= 1e-13 * np.abs(As @ x).max()
    st = p.transient_begin(1e-9, np.ones(2106), np.full(2106, 1.5e6))
    st.set_free(x)
    assert np.array_equal(st.free_values(), x)
    nodes = st.node_values()
    assert np.array_equal(nodes[freenode], x) and np.array_equal(nodes[dn2 - 1], d["dirichletheads"])
    # the same mesh with and without the library's re-numbering: identical structure seen from outside, heads to rounding
    forced.fv_tune(31, 0)
    q = fv.Problem.create(nb2, d["areasoverlengths"], 2106, dn2)
    assert not q.reorder_info()["reordered"]
    q.assemble(d["conductivities"], src, d["dirichletheads"])
    sq = q.transient_begin(1e-9, np.ones(2106), np.full(2106, 1.5e6))
    st.set_nodes(np.full(2106, 1.5e6))
    ia, _, _ = p.run_fixed(st, 50.0, 12, 1e-13, 5000)
    ib, _, _ = q.run_fixed(sq, 50.0, 12, 1e-13, 5000)
    ha, hb = st.node_values(), sq.node_values()
    assert np.abs(ha - hb).max() <= 1e-9 * np.abs(hb).max()
    # row blocks are ranges of the CALLER's numbering: cut from the re-numbered problem (through a canonical view of its
    # rows, fv_dist_setup) they are the blocks of the un-numbered one — plan, block products with injected halos, state
    from fvamd import dist

    for nranks in (2, 3):
        for rank in range(nranks):
            ba, bb = dist.RowBlock(p, nranks, rank), dist.RowBlock(q, nranks, rank)
            pa, pb = ba.plan(), bb.plan()
            assert (ba.lo, ba.hi, ba.nnz, ba.nhalo, ba.nsend) == (bb.lo, bb.hi, bb.nnz, bb.nhalo, bb.nsend)
            for key in pa:
                assert np.array_equal(pa[key], pb[key]), key
            xl, hl = rng.standard_normal(ba.nloc), rng.standard_normal(ba.nhalo)
            assert np.array_equal(ba.spmv_halo(xl, hl, 0.02), bb.spmv_halo(xl, hl, 0.02))
            assert np.abs(ba.state() - bb.state()).max() <= 1e-9 * np.abs(bb.state()).max()  # (the two 12-step runs above)
            ba.close()
            bb.close()
    # ... and the first free row at or after a node is a rank among the free nodes whatever the numbering inside
    for node in (0, 1, 17, 1000, 2105, 2106):
        ra, rb = C.c_int64(), C.c_int64()
        p.check(forced.fv_problem_free_rows_before(p.handle, node, C.byref(ra)))
        q.check(forced.fv_problem_free_rows_before(q.handle, node, C.byref(rb)))
        assert ra.value == rb.value == int(ofree[:node].sum())


def test_fractures_like_mesh_is_renumbered_by_the_library_and_runs_faster(fv):
    """configs[3]: 5M cells numbered at random inside each fracture.  fv_problem_create re-numbers them by itself (no host
    pre-processing by the caller); against the same mesh with the re-numbering switched off: same heads, and the SpMV's
    gather of x coalesces (>= 1.4x on the transient step)."""
    import time

    w = workloads.fractures_like(20, 500, seed=0)
    lib = fv.load()
    res = {}
    for mode in (1, 0):
        assert lib.fv_tune(31, mode) == 0
        try:
            p = fv.Problem.create((w["node1"], w["node2"]), w["aol"], w["N"], w["dnodes"])
        finally:
            lib.fv_tune(31, 1)
        info = p.reorder_info()
        assert info["reordered"] == bool(mode), info
        p.assemble(w["K"], np.zeros(w["N"]), w["dheads"])
        st = p.transient_begin(1e-9, w["volumes"], np.full(w["N"], 1.5e6))
        p.run_fixed(st, 1.0, 3, 1e-10, 5000)
        p.ctx.synchronize()
        t0 = time.perf_counter()
        it, inf, _ = p.run_fixed(st, 1.0, 50, 1e-10, 5000)
        p.ctx.synchronize()
        res[mode] = dict(sec=time.perf_counter() - t0, head=st.node_values(), iters=it.copy(), info=info, spmv_ms=p.bench_spmv(1.0, 10))
        assert inf.converged
        p.close()
    print("fractures-like 5M: re-numbered %.4f s (SpMV %.3f ms), as numbered %.4f s (SpMV %.3f ms); %r" % (res[1]["sec"], res[1]["spmv_ms"], res[0]["sec"], res[0]["spmv_ms"], res[1]["info"]))
    assert np.abs(res[1]["head"] - res[0]["head"]).max() <= 1e-8 * np.abs(res[0]["head"]).max()
    assert res[1]["sec"] * 1.4 <= res[0]["sec"]


def test_device_renumbering_is_the_host_routine_s_order(fv, forced):
    """fv_reorder.hip builds the reverse Cuthill-McKee order level by level on the device; it must be the order of the host
    routine it replaces (fv_tune key 47 = 0), cell for cell: the mean |i - j| over the faces after the re-numbering — an exact
    rational of integers on both sides — is compared on a shuffled fourfractures mesh, on multigraphs with repeated faces,
    self-loops, isolated cells and several components, and on the 5M-cell mesh (where the time is printed)."""
    d = np.load(os.path.join(GOLDEN, "fourfractures.npz"))
    rng = np.random.default_rng(9)
    cases = []
    shuffle = rng.permutation(2106)
    rank = np.empty(2106, np.int64)
    rank[shuffle] = np.arange(2106)
    cases.append((np.stack([rank[d["node1"] - 1] + 1, rank[d["node2"] - 1] + 1], 1), 2106, rank[d["dirichletnodes"] - 1] + 1))
    for seed in range(6):
        r = np.random.default_rng(100 + seed)
        N = int(r.integers(50, 4000))
        parts = []
        for lo, hi in ((1, N // 3), (N // 3 + 5, 2 * N // 3), (2 * N // 3 + 3, N)):  # three components and a few cells nobody touches
            m = 3 * (hi - lo)
            parts.append(np.stack([r.integers(lo, hi + 1, m), r.integers(lo, hi + 1, m)], 1))
        nb = np.concatenate(parts)
        nb = np.concatenate([nb, nb[: len(nb) // 10]])  # repeated faces (and self-loops from the random draw)
        r.shuffle(nb)
        dn = np.unique(r.integers(1, N + 1, max(1, N // 40)))
        cases.append((nb, N, dn))
    lib = forced
    for nb, N, dn in cases:
        got = {}
        for dev in (1, 0):
            assert lib.fv_tune(47, dev) == 0
            try:
                p = fv.Problem.create(nb, np.ones(len(nb)), N, dn.astype(np.int64))
                got[dev] = p.reorder_info()
                p.close()
            finally:
                lib.fv_tune(47, 1)
        assert got[1]["reordered"] and got[0]["reordered"]
        assert got[1]["mean_before"] == got[0]["mean_before"] and got[1]["mean_after"] == got[0]["mean_after"], (N, got)
    lib.fv_tune(31, 1)
    w = workloads.fractures_like(20, 500, seed=0)
    got = {}
    for dev in (1, 0):
        assert lib.fv_tune(47, dev) == 0
        try:
            import time

            t0 = time.perf_counter()
            p = fv.Problem.create((w["node1"], w["node2"]), w["aol"], w["N"], w["dnodes"])
            got[dev] = dict(p.reorder_info(), create_s=time.perf_counter() - t0)
            p.close()
        finally:
            lib.fv_tune(47, 1)
    print("fractures-like 5M: fv_problem_create %.3f s with the device re-numbering (%.3f s of it), %.3f s with the host routine (%.3f s)" %
          (got[1]["create_s"], got[1]["seconds"], got[0]["create_s"], got[0]["seconds"]))
    assert got[1]["reordered"] and got[1]["mean_after"] == got[0]["mean_after"]
    assert got[1]["seconds"] < 0.5 * got[0]["seconds"]
