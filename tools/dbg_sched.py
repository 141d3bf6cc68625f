import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package
import test_gpu_fused as T
fv = load_package()
case = T._problem(fv, T.BOX, seed=3)
sched = [(T.DT, 12, 1e-11), (T.DT, 20, 1e-3)]
out = T._run(fv, case, True, sched)
print("fused16", out[1].tolist(), out[2], flush=True)
