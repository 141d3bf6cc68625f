"""Synthetic inputs for the BASELINE.json configurations.  Data only:
numpy arrays in the reference's conventions (1-based int64 indices)."""
import numpy as np


def smooth_gaussian_field(ns, seed=0, radius_cells=(10, 10, 10), passes=3):
    """Stand-in for the reference's Matern field (GaussianRandomFields is not available):
    unit-variance field from separable box smoothing of N(0,1) noise, shape (n3,n2,n1) like
    the reference's node arrays; returned flattened in node order."""
    n1, n2, n3 = ns
    rng = np.random.default_rng(seed)
    g = rng.standard_normal((n1, n2, n3)).astype(np.float64)
    for axis, r in enumerate(radius_cells):
        r = int(max(1, min(r, ns[axis] // 2)))
        n = ns[axis]
        for _ in range(passes):
            # running sums, padded with their edge values: window sum of cells [i - r, i + r] = cp[i + 2 r + 1] - cp[i] (windows
            # clipped at the box's faces count the face cell repeatedly, as np.pad(mode="edge") of the sums implies)
            c = np.cumsum(g, axis=axis)
            pad = [(0, 0)] * 3
            pad[axis] = (r + 1, r)
            cp = np.pad(c, pad, mode="edge")
            del c
            hi = [slice(None)] * 3
            lo = [slice(None)] * 3
            hi[axis] = slice(2 * r + 1, 2 * r + 1 + n)
            lo[axis] = slice(0, n)
            g = cp[tuple(hi)] - cp[tuple(lo)]
            del cp
            g /= 2 * r + 1
    g -= g.mean()
    g /= g.std()
    return g.reshape(-1)  # C order of (n1,n2,n3) == node order i3 + n3*(i2 + n2*i1)


def grid_face_means(ns, nodevalues):
    """0.5 (v[node1] + v[node2]) for every face of regulargrid(ns) in the face list's order (a cell's x-, y-, z-face, cells in node order:
    /root/reference/src/grid.jl:72-105) — nodehycos2neighborhycos(neighbors, v, true) (grid.jl:27) without the face list, for grids
    whose 2 F face ends would not fit the host comfortably (or a lean problem, which keeps none)."""
    n1, n2, n3 = ns
    v = np.asarray(nodevalues, np.float64).reshape(n1, n2, n3)
    vals = np.empty((n1, n2, n3, 3))
    have = np.zeros((n1, n2, n3, 3), bool)
    vals[:-1, :, :, 0] = 0.5 * (v[:-1] + v[1:])
    have[:-1, :, :, 0] = True
    vals[:, :-1, :, 1] = 0.5 * (v[:, :-1] + v[:, 1:])
    have[:, :-1, :, 1] = True
    vals[:, :, :-1, 2] = 0.5 * (v[:, :, :-1] + v[:, :, 1:])
    have[:, :, :-1, 2] = True
    return vals[have]


def box_model_dirichlet(ns):
    """Dirichlet 1.0 on x = xmin, 0.0 on x = xmax (examples/box_model/ex.jl:29-37)."""
    n1, n2, n3 = ns
    plane = n2 * n3
    left = np.arange(plane) + 1
    right = (n1 - 1) * plane + np.arange(plane) + 1
    dn = np.r_[left, right].astype(np.int64)
    dh = np.r_[np.ones(plane), np.zeros(plane)]
    return dn, dh


def fractures_like(nfrac=20, m=500, seed=0):
    """DFN-style irregular connectivity: `nfrac` triangular lattices (interior degree 6) of
    m x m nodes, node order randomly permuted inside each fracture, pairs of fractures tied
    node-to-node along an intersection line (degree up to ~14).  aol log-uniform in
    [3e-9, 3e-5], K = 1e-12, volumes log-uniform over a decade, heads 2e6 / 1e6 on the two
    extreme lattice columns of every fracture (examples/fractures/setupmesh.jl:55)."""
    rng = np.random.default_rng(seed)
    per = m * m
    N = nfrac * per
    ii, jj = np.meshgrid(np.arange(m), np.arange(m), indexing="ij")
    lid = (ii * m + jj).astype(np.int64)
    e1 = [lid[:-1, :].ravel(), lid[:, :-1].ravel(), lid[:-1, 1:].ravel()]
    e2 = [lid[1:, :].ravel(), lid[:, 1:].ravel(), lid[1:, :-1].ravel()]
    a_loc, b_loc = np.concatenate(e1), np.concatenate(e2)
    n1, n2, dn, dh = [], [], [], []
    perms = []
    for f in range(nfrac):
        perm = rng.permutation(per).astype(np.int64)
        perms.append(perm)
        off = f * per
        n1.append(off + perm[a_loc])
        n2.append(off + perm[b_loc])
        dn.append(off + perm[lid[:, 0]])
        dh.append(np.full(m, 2e6))
        dn.append(off + perm[lid[:, m - 1]])
        dh.append(np.full(m, 1e6))
    for f in range(nfrac - 1):  # intersection lines: row of fracture f tied to a column of fracture f+1
        g = f + 1
        r = int(rng.integers(m // 4, 3 * m // 4))
        c = int(rng.integers(m // 4, 3 * m // 4))
        for shift in range(3):  # a few parallel ties -> nodes of degree up to 6 + 2*3
            n1.append(f * per + perms[f][lid[min(r + shift, m - 1), 1 : m - 1]])
            n2.append(g * per + perms[g][lid[1 : m - 1, c]])
    n1 = np.concatenate(n1)
    n2 = np.concatenate(n2)
    lo, hi = np.minimum(n1, n2), np.maximum(n1, n2)
    order = np.lexsort((hi, lo))  # the reference meshes list faces sorted with first < second
    n1, n2 = lo[order] + 1, hi[order] + 1
    F = len(n1)
    aol = np.exp(rng.uniform(np.log(3e-9), np.log(3e-5), F))
    K = np.full(F, 1e-12)
    vol = np.exp(rng.uniform(np.log(1e-3), np.log(1e-2), N))
    dn = np.concatenate(dn) + 1
    dh = np.concatenate(dh)
    return dict(N=N, node1=n1.astype(np.int64), node2=n2.astype(np.int64), aol=aol, K=K, volumes=vol, dnodes=dn.astype(np.int64), dheads=dh)
