"""Mesh ingestion for the reference's DFN workflow (host-side glue, as in the reference:
/root/reference/examples/fractures/setupmesh.jl:3-47).  PFLOTRAN "unstructured explicit" grids (.uge):

    CELLS <n>
    <id> <x> <y> <z> <volume>            (n lines)
    CONNECTIONS <m>
    <id1> <id2> <x> <y> <z> <area>       (m lines; x y z = face centre)

give everything the solver wants: neighbours, areas/lengths (length = distance between the two cell centres,
setupmesh.jl:42-43) and — unlike the mesh.jld the reference saves — the cell volumes the transient path needs.
The saved meshes themselves (`mesh.jld`, setupmesh.jl:46; read back at examples/fractures/ex.jl:9) are read by
`read_mesh_jld` through the HDF5-subset reader in jldio.py."""
import numpy as np

from .jldio import JLDFormatError, load_jld  # noqa: F401  (re-exported)

_MESH_JLD_FIELDS = ("xs", "ys", "zs", "neighbors", "areasoverlengths", "fractureindices", "dirichletnodes", "dirichletheads", "conductivities")


def read_mesh_jld(path):
    """The variables examples/fractures/ex.jl:9 loads from a mesh.jld, as a dict: xs, ys, zs, neighbors ((m, 2) int64,
    1-based), node1/node2 (its columns), areasoverlengths, fractureindices, dirichletnodes (1-based), dirichletheads,
    conductivities, and metaindex — the 1-based fracture of each connection's first node (ex.jl:11-13)."""
    mesh = dict(zip(_MESH_JLD_FIELDS, load_jld(path, *_MESH_JLD_FIELDS)))
    nb = mesh["neighbors"]
    if nb.ndim != 2 or nb.shape[1] != 2:
        raise JLDFormatError("%s: neighbors is not a list of pairs" % path)
    N = len(mesh["xs"])
    if len(nb) and (nb.min() < 1 or nb.max() > N):
        raise ValueError("%s: connection refers to a cell outside 1..%d" % (path, N))
    mesh["node1"], mesh["node2"] = nb[:, 0].copy(), nb[:, 1].copy()
    mesh["metaindex"] = mesh["fractureindices"][mesh["node1"] - 1]
    return mesh


def read_uge(path):
    """-> dict(xs, ys, zs, volumes, node1, node2, areas, areasoverlengths); node indices 1-based int64."""
    with open(path) as f:
        head = f.readline().split()
        if len(head) < 2 or head[0].upper() != "CELLS":
            raise ValueError("%s: expected 'CELLS <n>' on the first line" % path)
        ncells = int(head[1])
        cells = np.loadtxt(f, dtype=np.float64, max_rows=ncells, ndmin=2)
        if cells.shape != (ncells, 5):
            raise ValueError("%s: expected %d cell lines of 5 columns, got %s" % (path, ncells, cells.shape))
        head = f.readline().split()
        if len(head) < 2 or head[0].upper() != "CONNECTIONS":
            raise ValueError("%s: expected 'CONNECTIONS <m>' after the cells" % path)
        nconn = int(head[1])
        conn = np.loadtxt(f, dtype=np.float64, max_rows=nconn, ndmin=2)
        if conn.shape[0] != nconn or conn.shape[1] < 6:
            raise ValueError("%s: expected %d connection lines of 6 columns, got %s" % (path, nconn, conn.shape))
    ids = cells[:, 0].astype(np.int64)
    if not np.array_equal(ids, np.arange(1, ncells + 1)):
        raise ValueError("%s: cell ids must run 1..n in order" % path)
    xs, ys, zs, volumes = cells[:, 1].copy(), cells[:, 2].copy(), cells[:, 3].copy(), cells[:, 4].copy()
    node1, node2 = conn[:, 0].astype(np.int64), conn[:, 1].astype(np.int64)
    if nconn and (min(node1.min(), node2.min()) < 1 or max(node1.max(), node2.max()) > ncells):
        raise ValueError("%s: connection refers to a cell outside 1..%d" % (path, ncells))
    areas = conn[:, -1].copy()  # connectiondata[:, end], setupmesh.jl:41
    a, b = node1 - 1, node2 - 1
    lengths = np.sqrt((xs[a] - xs[b]) ** 2 + (ys[a] - ys[b]) ** 2 + (zs[a] - zs[b]) ** 2)
    return dict(xs=xs, ys=ys, zs=zs, volumes=volumes, node1=node1, node2=node2, areas=areas, areasoverlengths=areas / lengths)


def fracture_conductivities(node1, node2, fractureconductivities, fractureindices):
    """Per-connection conductivity = geometric mean of the two cells' fracture conductivities (setupmesh.jl:36-39).
    fractureindices: 1-based fracture id per cell; fractureconductivities: one value per fracture."""
    k = np.asarray(fractureconductivities, dtype=np.float64)
    fi = np.asarray(fractureindices, dtype=np.int64)
    n1, n2 = np.asarray(node1, dtype=np.int64), np.asarray(node2, dtype=np.int64)
    return np.sqrt(k[fi[n1 - 1] - 1] * k[fi[n2 - 1] - 1])


def dirichlet_from_predicate(xs, ys, zs, isdirichletnode, dirichlethead):
    """dirichletnodes (1-based) and heads from the two closures setupmesh.jl:45-46 takes."""
    nodes = np.array([i + 1 for i in range(len(xs)) if isdirichletnode(xs[i], ys[i], zs[i])], dtype=np.int64)
    heads = np.array([dirichlethead(xs[i - 1], ys[i - 1], zs[i - 1]) for i in nodes], dtype=np.float64)
    return nodes, heads


def locality_order(node1, node2, N):
    """A node order that puts connected cells next to each other (reverse Cuthill-McKee of the connectivity graph).
    DFN meshes often number the cells of a fracture in an order unrelated to their position; every x element the SpMV
    gathers is then its own cache line and the kernel runs at a third of its rate on a well-ordered mesh.  The solver
    keeps whatever order it is given (its exports are checked index for index against the reference), so re-ordering
    is a pre-processing step, like the rest of this module: returns `order` (new position -> old node, 1-based) and
    `rank` (old node -> new position, 1-based)."""
    import scipy.sparse as sp
    from scipy.sparse.csgraph import reverse_cuthill_mckee

    a = np.asarray(node1, dtype=np.int64) - 1
    b = np.asarray(node2, dtype=np.int64) - 1
    keep = a != b
    g = sp.coo_matrix((np.ones(int(keep.sum()), np.int8), (a[keep], b[keep])), shape=(N, N)).tocsr()
    g = (g + g.T).tocsr()
    order0 = np.asarray(reverse_cuthill_mckee(g, symmetric_mode=True), dtype=np.int64)
    rank0 = np.empty(N, np.int64)
    rank0[order0] = np.arange(N)
    return order0 + 1, rank0 + 1


def reorder_mesh(mesh, rank):
    """The mesh dict of read_uge (or any dict with node1/node2 and per-node arrays) renumbered by `rank` (old -> new,
    1-based).  Faces keep their order (so per-face arrays stay valid) but are re-oriented to first < second, as the
    reference's meshes list them; per-node arrays (length N) are permuted; `dirichletnodes` are renamed."""
    rank = np.asarray(rank, dtype=np.int64)
    N = len(rank)
    out = dict(mesh)
    n1, n2 = rank[np.asarray(mesh["node1"], dtype=np.int64) - 1], rank[np.asarray(mesh["node2"], dtype=np.int64) - 1]
    out["node1"], out["node2"] = np.minimum(n1, n2), np.maximum(n1, n2)
    order = np.empty(N, np.int64)
    order[rank - 1] = np.arange(N)
    for k, v in mesh.items():
        if k in ("node1", "node2"):
            continue
        if k in ("dirichletnodes", "dnodes"):
            out[k] = rank[np.asarray(v, dtype=np.int64) - 1]
        elif isinstance(v, np.ndarray) and v.ndim == 1 and len(v) == N:
            out[k] = v[order]
    return out
