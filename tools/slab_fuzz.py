"""Differential check of the per-rank slab assembly: random grids, random Dirichlet sets (whole planes, scattered cells,
repeats), random rank counts — every rank's block cut from its own slab problem must equal, bit for bit, the block cut from
the globally assembled operator with the same bounds: plan arrays and a block product with random vectors."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from __graft_entry__ import load_package  # noqa: E402

fv = load_package()
from fvamd import dist  # noqa: E402

bad = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    rng = np.random.default_rng(seed)
    ns = [int(rng.integers(2, 14)), int(rng.integers(2, 9)), int(rng.integers(2, 9))]
    N = ns[0] * ns[1] * ns[2]
    plane = ns[1] * ns[2]
    mins, maxs = [0.0, 0.0, 0.0], [float(ns[0]), float(ns[1]) * 0.7, float(ns[2]) * 1.3]
    nodes = []
    if rng.random() < 0.5:
        nodes += list(range(1, plane + 1))  # first plane
    if rng.random() < 0.4:
        q = int(rng.integers(0, ns[0]))
        nodes += list(range(q * plane + 1, (q + 1) * plane + 1))  # some whole plane
    nodes += list(rng.integers(1, N + 1, int(rng.integers(0, max(2, N // 6)))))
    if not nodes:
        nodes = [int(rng.integers(1, N + 1))]
    dn = np.array(nodes, dtype=np.int64)
    rng.shuffle(dn)
    dh = rng.standard_normal(len(dn))
    nranks = int(rng.integers(1, min(ns[0], 6) + 1))
    ref = fv.Problem.regulargrid(mins, maxs, ns, dn)
    Kg = np.exp(rng.standard_normal(ref.F))
    free, _ = ref.free_maps()
    src = np.where(free, rng.standard_normal(N), 0.0)
    u0 = rng.standard_normal(N)
    ref.assemble(Kg, src, dh)
    ref.transient_begin(0.3, None, u0)
    planes = dist.slab_planes(ns[0], nranks)
    ok = True
    for rank in range(nranks):
        p, bounds = dist.slab_problem(mins, maxs, ns, dn, nranks, rank)
        f0, f1 = dist.slab_face_range(ns, planes[rank], planes[rank + 1])
        ok &= p.F == f1 - f0
        p.assemble(Kg[f0:f1], src, dh)
        p.transient_begin(0.3, None, u0)
        a = dist.RowBlock(p, nranks, rank, bounds)
        b = dist.RowBlock(ref, nranks, rank, bounds)
        for u, v in zip(a.plan(), b.plan()):
            ok &= bool(np.array_equal(u, v))
        r2 = np.random.default_rng(seed * 100 + rank)
        xl, xh = r2.standard_normal(a.nloc), r2.standard_normal(a.nhalo)
        ok &= bool(np.array_equal(a.spmv_halo(xl, xh, 0.7), b.spmv_halo(xl, xh, 0.7)))
        ok &= bool(np.array_equal(a.state(), b.state()))
        a.close()
        b.close()
        p.close()
    ref.close()
    bad += not ok
    print("seed %d ns %s ranks %d dirichlet %d free %d: %s" % (seed, ns, nranks, len(np.unique(dn)), int(free.sum()), "ok" if ok else "MISMATCH"), flush=True)
print("mismatches:", bad)
