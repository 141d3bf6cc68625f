"""The bench's heterogeneous-conductivity block alone (bench.hetero_block: sigma = 1 field on the 464^3 box, dt = 60 s — two or three PCG
iterations per step — and the time step at which its steps take one), for rocprofv3 / PMC passes.  usage: python tools/hetero_rate.py [--ns 464] [--steps 20]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--ns", type=int, default=464)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--warmup", type=int, default=5)
ap.add_argument("--dt", type=float, default=60.0)
ap.add_argument("--rtol", type=float, default=1e-10)
ap.add_argument("--maxiter", type=int, default=2000)
args = ap.parse_args()
fv = load_package()
print(json.dumps(bench.hetero_block(fv, fv.default_context(), args)))
