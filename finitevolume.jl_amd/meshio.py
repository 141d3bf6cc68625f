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


# ---------------------------------------------------------------------------------------------------------------------
# FEHM / LaGriT files of examples/watertable/setupmodel.jl:7-16 (FEHM.parsegrid, FEHM.parsestor, FEHM.parsezone).
# FEHM.jl is a dependency outside the reference tree and no .fehmn/.stor/.zone file ships with it, so these readers
# follow the published file formats (LaGriT "stor" sparse-matrix file, FEHM `coor`/`zone` macros) and are UNPINNED:
# tested against files written by the tests from the same descriptions, not against FEHM.jl's output.
def _tokens(f):
    for line in f:
        for tok in line.split():
            yield tok


def read_fehm_grid(path):
    """FEHM.parsegrid: the `coor` block of a .fehmn grid file -> coords (3, N), as setupmodel.jl:7-10 slices it."""
    with open(path) as f:
        for line in f:
            if line.strip().lower().startswith("coor"):
                break
        else:
            raise ValueError("%s: no 'coor' block" % path)
        n = int(f.readline().split()[0])
        rows = np.loadtxt(f, dtype=np.float64, max_rows=n, ndmin=2)
    if rows.shape[0] != n or rows.shape[1] < 4:
        raise ValueError("%s: expected %d coordinate lines 'id x y z', got %s" % (path, n, rows.shape))
    if not np.array_equal(rows[:, 0].astype(np.int64), np.arange(1, n + 1)):
        raise ValueError("%s: node ids must run 1..n in order" % path)
    return np.ascontiguousarray(rows[:, 1:4].T)


def read_fehm_zones(path):
    """FEHM.parsezone: a `zone` file with `nnum` node lists -> (zonenums, nodesinzones): zone numbers and one int64
    array of 1-based nodes per zone (setupmodel.jl:15-19)."""
    zonenums, nodes = [], []
    with open(path) as f:
        lines = [ln.strip() for ln in f]
    i = 0
    while i < len(lines) and not lines[i].lower().startswith("zone"):
        i += 1
    if i == len(lines):
        raise ValueError("%s: no 'zone' macro" % path)
    i += 1
    while i < len(lines):
        if not lines[i]:
            i += 1
            continue
        if lines[i].lower().startswith("stop"):
            break
        zonenums.append(int(lines[i].split()[0]))
        i += 1
        if i >= len(lines) or not lines[i].lower().startswith("nnum"):
            raise ValueError("%s: zone %d is not given as an 'nnum' node list" % (path, zonenums[-1]))
        i += 1
        count = int(lines[i].split()[0])
        vals = [int(t) for t in lines[i].split()[1:]]
        i += 1
        while len(vals) < count:
            vals.extend(int(t) for t in lines[i].split())
            i += 1
        if len(vals) != count:
            raise ValueError("%s: zone %d lists %d nodes, announced %d" % (path, zonenums[-1], len(vals), count))
        nodes.append(np.array(vals, dtype=np.int64))
    return zonenums, nodes


def read_stor(path):
    """FEHM.parsestor: a LaGriT ASCII .stor file -> (volumes, areasoverlengths, neighbors) with one entry per stored
    matrix entry (both directions of every connection and the diagonal, as in the file; setupmodel.jl:12-14 keeps
    those with node1 < node2).  neighbors is (m, 2) int64, 1-based; areasoverlengths = |A_ij / d_ij|.

    Layout (LaGriT manual, "stor file format"): title; date; `NWRITTEN NEQ NCOEF+NEQ+1 NUM_AREA_COEF [NCON_MAX]`;
    NEQ Voronoi volumes; NEQ+1 row pointers (offset by NEQ+1) followed by the NCOEF column indices; NCOEF indices into
    the written coefficients followed by NEQ+1 zeros; NEQ diagonal pointers; NWRITTEN x NUM_AREA_COEF coefficients (the
    scalar A_ij/d_ij, negative, is the last — or only — component)."""
    with open(path) as f:
        title = f.readline()
        if "stor" not in title.lower():
            raise ValueError("%s: not a LaGriT stor file (title line %r)" % (path, title.strip()))
        if "asc" not in title.lower():
            raise ValueError("%s: only the ASCII form of the stor file is read" % path)
        f.readline()
        head = f.readline().split()
        nwritten, neq, ncoef_tot, nareacoef = (int(t) for t in head[:4])
        ncoef = ncoef_tot - neq - 1
        if min(nwritten, neq, ncoef) < 0 or nareacoef not in (1, 3, 4):
            raise ValueError("%s: implausible matrix parameters %s" % (path, head))
        tok = _tokens(f)
        try:
            volumes = np.array([float(next(tok)) for _ in range(neq)])
            rowptr = np.array([int(next(tok)) for _ in range(neq + 1)], dtype=np.int64)
            cols = np.array([int(next(tok)) for _ in range(ncoef)], dtype=np.int64)
            coefidx = np.array([int(next(tok)) for _ in range(ncoef)], dtype=np.int64)
            for _ in range(neq + 1 + neq):  # the zero padding and the diagonal pointers
                next(tok)
            coefs = np.array([float(next(tok)) for _ in range(nwritten * nareacoef)]).reshape(nareacoef, nwritten)
        except StopIteration:
            raise ValueError("%s: file ends before the announced arrays do" % path) from None
    start = rowptr - (neq + 1)  # 0-based start of every row in `cols`
    if start[0] != 0 or start[-1] != ncoef or np.any(np.diff(start) < 0):
        raise ValueError("%s: row pointers do not describe %d entries in %d rows" % (path, ncoef, neq))
    if ncoef and (cols.min() < 1 or cols.max() > neq or coefidx.min() < 1 or coefidx.max() > nwritten):
        raise ValueError("%s: column or coefficient index out of range" % path)
    rows = np.repeat(np.arange(1, neq + 1, dtype=np.int64), np.diff(start))
    aol = np.abs(coefs[-1][coefidx - 1])
    return volumes, aol, np.stack([rows, cols], axis=1)
