"""CPU: the oracle against the fixtures it generated (tests/golden/make_oracle_fixtures.py) — index arrays of
regulargrid / getfreenodes / assembleA bit for bit, heads of the 10^3 box, the Theis run's outer steps and final heads.  These files
are what a session with a Julia runtime diffs the real package against; here they pin the oracle against regressions."""
import os

import numpy as np

from tests import refcases

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _box(oracle, g):
    coords, n1, n2, aol, vol = oracle.regulargrid(list(g["mins"]), list(g["maxs"]), [int(v) for v in g["ns"]])
    assert np.array_equal(coords, g["coords"]) and np.array_equal(n1, g["node1"]) and np.array_equal(n2, g["node2"])
    assert np.array_equal(aol, g["areasoverlengths"]) and np.array_equal(vol, g["volumes"])
    freenode, n2f = oracle.getfreenodes(len(vol), g["dirichletnodes"])
    assert np.array_equal(np.asarray(freenode, bool), g["freenode"]) and np.array_equal(np.asarray(n2f), g["nodei2freenodei"])
    args = (n1, n2, aol, g["conductivities"], g["sources"], g["dirichletnodes"], g["dirichletheads"])
    A = oracle.assembleA(*args)
    b = oracle.assembleb(*args)
    assert np.array_equal(A.colptr, g["colptr"]) and np.array_equal(A.rowval, g["rowval"]) and np.array_equal(A.nzval, g["nzval"]) and np.array_equal(b, g["b"])
    return args


def test_box_3x4x5_index_arrays_and_values(oracle):
    _box(oracle, np.load(os.path.join(GOLD, "oracle_box_3x4x5.npz")))


def test_box_10x10x10_index_arrays_values_and_heads(oracle):
    g = np.load(os.path.join(GOLD, "oracle_box_10x10x10.npz"))
    args = _box(oracle, g)
    head, ch, A, b, freenode = oracle.solvediffusion(*args, solver="direct")
    assert np.allclose(head, g["head_direct"], rtol=1e-13, atol=0)
    assert np.abs(g["head_cg_1e14"] - g["head_direct"]).max() <= 1e-11  # (the two solvers of the fixture agree)
    assert head.min() >= 0.0 and head.max() <= 1.0  # examples/box_model/ex_piml_data.jl:49-51


def test_theis_outer_steps_and_final_heads(oracle):
    g = np.load(os.path.join(GOLD, "oracle_theis_101x101x2.npz"))
    c = refcases.theis(oracle.regulargrid)
    assert np.array_equal(c["goodnodes"] + 1, g["goodnodes"]) and np.array_equal(c["rs"], g["rs"])
    us, ts = oracle.backwardeulerintegrate(c["u0"], c["tspan"], c["Ss"], c["volumes"], c["node1"], c["node2"], c["aol"], c["K"], c["sources"], c["dnodes"], c["dheads"],
                                           None, False, atol=c["atol"], dt0=c["dt0"])
    assert np.array_equal(np.asarray(ts), g["ts"])
    assert np.allclose(np.asarray(us[-1])[c["goodnodes"]], g["head_final_goodnodes"], rtol=1e-13, atol=0)
    # the fixture against the analytic curves the reference's own test uses (test/theis.jl:55-65)
    theis = np.array([c["steadyhead"] - refcases.theisdrawdown(86400.0 * 10, r, c["T"], c["S"], c["Q"]) for r in g["rs"]])
    thiem = np.array([c["steadyhead"] - refcases.thiemdrawdown(r, c["T"], c["Q"], c["sidelength"]) for r in g["rs"]])
    assert refcases.isapprox(g["head_final_goodnodes"], thiem, atol=1e-4, rtol=2e-2)
    assert theis.shape == thiem.shape
