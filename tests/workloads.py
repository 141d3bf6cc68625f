"""The synthetic BASELINE.json inputs live in the package (finitevolume.jl_amd/workloads.py) so that bench.py does not
import from tests/; re-exported here for the tests."""
import importlib.util
import os
import sys

_pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "finitevolume.jl_amd", "workloads.py")
_spec = importlib.util.spec_from_file_location("fv_workloads", _pkg)
_mod = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_mod)
smooth_gaussian_field = _mod.smooth_gaussian_field
box_model_dirichlet = _mod.box_model_dirichlet
fractures_like = _mod.fractures_like
grid_face_means = _mod.grid_face_means
