"""Row-block partition of the assembled operator across the GPUs of one node.

The reference has nothing distributed; this is the build's own
multi-GPU design: free rows are split into contiguous ranges, one per rank;
each rank keeps its row block of the CSR with columns renumbered
[local | halo], the halo being the sorted remote columns it references.  Per
SpMV every rank sends each peer the rows that peer references (ascending global
order == the receiver's halo order) and receives its own halo the same way.

This module is the host-side (numpy) statement of the plan.  libfvhip builds the
same plan on the device (fv_dist_setup); tests check the two against each other,
and the gloo world-size-2 CPU test runs the exchange protocol on top of it.
0-based indices throughout (device convention).
"""
import numpy as np


def row_ranges(n, nranks):
    """Contiguous, nearly equal row ranges: rank r owns [bounds[r], bounds[r+1])."""
    return np.array([(r * n) // nranks for r in range(nranks + 1)], dtype=np.int64)


def plan(rowptr, colind, nranks, rank):
    """Partition plan of rank `rank` for the n x n CSR (rowptr, colind), 0-based.

    Returns a dict with
      lo, hi              owned global row range
      rowptr, colind      local CSR, columns: < nloc local, >= nloc halo slot
      entry_lo            offset of the local entries in the global value array
      halo_cols           global column of every halo slot (ascending)
      recv_counts[q]      halo slots owned by peer q (contiguous, in peer order)
      send_idx[q]         local rows peer q needs, ascending
      boundary_rows       bool mask: local rows that reference a halo slot
    """
    rowptr = np.asarray(rowptr, np.int64)
    colind = np.asarray(colind, np.int64)
    n = len(rowptr) - 1
    bounds = row_ranges(n, nranks)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    nloc = hi - lo
    e0, e1 = int(rowptr[lo]), int(rowptr[hi])
    cols = colind[e0:e1]
    remote = (cols < lo) | (cols >= hi)
    halo_cols = np.unique(cols[remote])
    local_col = np.where(remote, nloc + np.searchsorted(halo_cols, cols), cols - lo)
    owner = np.searchsorted(bounds, halo_cols, side="right") - 1
    recv_counts = np.bincount(owner, minlength=nranks).astype(np.int64) if len(halo_cols) else np.zeros(nranks, np.int64)
    send_idx = []
    for q in range(nranks):
        if q == rank:
            send_idx.append(np.empty(0, np.int64))
            continue
        qc = colind[int(rowptr[bounds[q]]) : int(rowptr[bounds[q + 1]])]
        mine = qc[(qc >= lo) & (qc < hi)]
        send_idx.append(np.unique(mine) - lo)
    lrp = rowptr[lo : hi + 1] - e0
    rowlen = np.diff(lrp)
    rowid = np.repeat(np.arange(nloc), rowlen)
    boundary = np.zeros(nloc, bool)
    boundary[rowid[remote]] = True
    return dict(lo=lo, hi=hi, nloc=nloc, rowptr=lrp, colind=local_col.astype(np.int64), entry_lo=e0, halo_cols=halo_cols,
                recv_counts=recv_counts, send_idx=send_idx, boundary_rows=boundary, bounds=bounds)
