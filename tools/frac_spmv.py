import os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
from __graft_entry__ import load_package
from tests import workloads
fv = load_package()
w = workloads.fractures_like(20, 500, seed=0)
p = fv.Problem.create((w["node1"], w["node2"]), w["aol"], w["N"], w["dnodes"])
p.assemble(w["K"], np.zeros(w["N"]), w["dheads"])
p.transient_begin(1e-9, w["volumes"], np.full(w["N"], 1.5e6))
ms = min(p.bench_spmv(1.0, 20) for _ in range(3))
b = 12 * p.nnz + 20 * p.n
print("fractures-like: n %d nnz %d SpMV %.4f ms -> %.0f GB/s (CSR accounting)" % (p.n, p.nnz, ms, b / ms / 1e6))
# the same mesh with the natural (unpermuted) node order inside each fracture would be the locality upper bound
