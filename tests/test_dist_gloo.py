"""CPU (gloo, world_size 2 and 3): the row-block plan + halo/all-reduce protocol of
the multi-GPU path reproduce the serial solution."""
import socket
import tempfile

import numpy as np
import pytest

from tests import dist_worker


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_row_ranges_and_plan_shapes(fv):
    from fvamd import partition

    assert partition.row_ranges(10, 3).tolist() == [0, 3, 6, 10]
    A, b, D, u0 = dist_worker.build_case()
    rowptr, colind = A.colptr - 1, A.rowval - 1
    total = 0
    for r in range(4):
        pl = partition.plan(rowptr, colind, 4, r)
        total += pl["nloc"]
        assert pl["rowptr"][0] == 0 and pl["rowptr"][-1] == len(pl["colind"])
        assert pl["colind"].max() < pl["nloc"] + len(pl["halo_cols"])
        assert (np.diff(pl["halo_cols"]) > 0).all()
        assert pl["recv_counts"].sum() == len(pl["halo_cols"]) and pl["recv_counts"][r] == 0
        # slabs of a structured grid only talk to their neighbours
        assert all(pl["recv_counts"][q] == 0 for q in range(4) if abs(q - r) > 1)
    assert total == A.n


@pytest.mark.parametrize("world", [2, 3])
def test_distributed_protocol_matches_serial(world):
    import torch.multiprocessing as mp

    with tempfile.TemporaryDirectory() as td:
        mp.spawn(dist_worker.worker, args=(world, _free_port(), td), nprocs=world, join=True)
        u = np.load(td + "/u_dist.npy")
        iters = np.load(td + "/iters.npy")
    ref = dist_worker.serial_reference()
    assert np.linalg.norm(u - ref) / np.linalg.norm(ref) < 1e-11
    assert (iters > 3).all()
