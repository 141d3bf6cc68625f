#!/usr/bin/env python3
"""Experiment: does a brick (3-D blocked) numbering of the cells speed up the SpMV?
Builds the same box twice through the generic fv_problem_create path: natural node
numbering vs nodes renumbered brick by brick (B^3 cells contiguous)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
import bench
fv = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 320
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ns = [n, n, n]
mins, maxs = bench.spacing_box(ns)
coords, nb, aol, vol = fv.regulargrid(mins, maxs, ns)
del coords
dn, src = bench.box_setup(ns)
N = n ** 3
def run(perm_nodes, label):
    # perm_nodes[old_node0] = new_node0
    if perm_nodes is None:
        n1, n2, d, v = nb[:, 0], nb[:, 1], dn, vol
    else:
        n1 = perm_nodes[nb[:, 0] - 1] + 1; n2 = perm_nodes[nb[:, 1] - 1] + 1; d = perm_nodes[dn - 1] + 1
        v = np.empty_like(vol); v[perm_nodes] = vol
    p = fv.Problem.create((n1, n2), aol, N, d)
    p.assemble(np.array([1e-5]), np.zeros(N), np.full(len(d), 1e3))
    p.transient_begin(0.1, v, np.full(N, 1e3))
    ms = min(p.bench_spmv(1 / 60.0, 10) for _ in range(3))
    bytes_ = 12 * p.nnz + 20 * p.n
    print("%-28s n=%d nnz=%d  %.3f ms  %.0f GB/s (12nnz+20n, shift folded)" % (label, p.n, p.nnz, ms, bytes_ / ms / 1e6))
    p.close()
run(None, "natural numbering")
i = np.arange(N, dtype=np.int64)
i3 = i % n; i2 = (i // n) % n; i1 = i // (n * n)
for BB in (B, 2 * B):
    b1, b2, b3 = i1 // BB, i2 // BB, i3 // BB
    nbk = (n + BB - 1) // BB
    assert n % BB == 0
    key = (((b1 * nbk + b2) * nbk + b3) * BB + (i1 % BB)) * BB * BB + (i2 % BB) * BB + (i3 % BB)
    run(key, "brick %d^3 numbering" % BB)
