"""Same-process A/B of the sliced-DIA value layout (packed / padded) x SpMV kernel (plane-marching with and without the 16-byte windows, slice by slice) on the bench operator, plus the elimination diagnosis of the marching kernel (fv_tune 17)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402
import bench  # noqa: E402

fv = load_package()
ns = [int(sys.argv[1]) if len(sys.argv) > 1 else 464] * 3
mins, maxs = bench.spacing_box(ns)
dn, src = bench.box_setup(ns)
lib = fv.load()
probs = {}
for packed in (0, 1):  # the layout is fixed when a problem's DIA copy is built: one problem per layout, same process
    lib.fv_tune(11, packed)
    p = fv.Problem.regulargrid(mins, maxs, ns, dn)
    p.assemble(np.array([1e-5]), src, np.full(len(dn), 1e3))
    p.transient_begin(0.1, None, np.full(p.N, 1e3))
    p.bench_spmv(1 / 60.0, 2)
    probs[packed] = p
res = {}
for r in range(5):
    for packed in (0, 1):
        for march in (1, 4, 0):  # 4: marching with separate centre + edge loads instead of the 16-byte window
            lib.fv_tune(9, 2 if march else 0)
            lib.fv_tune(18, 0 if march == 4 else 1)
            res.setdefault((packed, march), []).append(probs[packed].bench_spmv(1 / 60.0, 10))
        lib.fv_tune(18, 1)
    for dbg in (1, 2, 3):  # diagnosis (wrong results): marching kernel without in-plane arm loads / masked edge loads / both
        lib.fv_tune(9, 1)
        lib.fv_tune(17, dbg)
        res.setdefault((1, 10 + dbg), []).append(probs[1].bench_spmv(1 / 60.0, 10))
        lib.fv_tune(17, 0)
for (packed, march), t in sorted(res.items()):
    print("FV_BAND=%s values %s, %s: median %.3f ms min %.3f ms" % (os.environ.get("FV_BAND", "default"), "packed" if packed else "padded to 8 blocks", {1: "plane-marching", 4: "plane-marching, centre + edge loads (no window)", 0: "slice by slice", 11: "marching WITHOUT in-plane arm loads (diagnosis)", 12: "marching WITHOUT masked edge loads (diagnosis)", 13: "marching without both (diagnosis)"}[march], float(np.median(t)), min(t)), flush=True)
