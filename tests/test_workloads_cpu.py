"""CPU: the synthetic inputs' helpers against the oracle's restatement of the reference functions they stand in for."""
import numpy as np


def test_grid_face_means_are_nodehycos2neighborhycos_on_the_grids_face_list(oracle):
    from tests.workloads import grid_face_means

    for ns in ([3, 4, 5], [2, 2, 2], [6, 2, 3]):
        _, n1, n2, aol, vol = oracle.regulargrid([0.0, 0.0, 0.0], [1.0, 2.0, 3.0], ns, want_coords=False)
        v = np.random.default_rng(sum(ns)).standard_normal(len(vol))
        want = oracle.nodehycos2neighborhycos(n1, n2, v, True)  # /root/reference/src/grid.jl:27
        got = grid_face_means(ns, v)
        assert got.shape == want.shape and np.array_equal(got, want)
