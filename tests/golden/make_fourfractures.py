"""Converts the reference's own data fixture
/root/reference/examples/fractures/fourfractures/{mesh,pflotran_solution}.jld
(HDF5/JLD written by Julia 0.6) into tests/golden/fourfractures.npz.

Data only (inputs + the PFLOTRAN cross-code head vector); run once in the build
container, where /root/reference and /opt/conda/bin/h5dump exist:
    python tests/golden/make_fourfractures.py
"""
import os
import re
import subprocess

import numpy as np

SRC = "/root/reference/examples/fractures/fourfractures"
H5DUMP = "/opt/conda/bin/h5dump"
HERE = os.path.dirname(os.path.abspath(__file__))


def dump(fname, dset, dtype):
    """Text dump (works for the compound Pair{Int,Int} too); %.17g keeps doubles exact."""
    txt = subprocess.check_output([H5DUMP, "-d", dset, "-y", "-w", "0", "-m", "%.17g", os.path.join(SRC, fname)], text=True)
    body = txt[txt.index("DATA {") + 6 : txt.rindex("}")]
    toks = re.findall(r"[-+]?(?:\d+\.?\d*(?:[eE][-+]?\d+)?|nan|inf)", body)
    return np.array([float(t) for t in toks]).astype(dtype)


def main():
    nb = dump("mesh.jld", "/neighbors", np.int64).reshape(-1, 2)
    data = dict(
        node1=nb[:, 0].copy(),
        node2=nb[:, 1].copy(),
        areasoverlengths=dump("mesh.jld", "/areasoverlengths", np.float64),
        conductivities=dump("mesh.jld", "/conductivities", np.float64),
        dirichletnodes=dump("mesh.jld", "/dirichletnodes", np.int64),
        dirichletheads=dump("mesh.jld", "/dirichletheads", np.float64),
        fractureindices=dump("mesh.jld", "/fractureindices", np.int64),
        xs=dump("mesh.jld", "/xs", np.float64),
        ys=dump("mesh.jld", "/ys", np.float64),
        zs=dump("mesh.jld", "/zs", np.float64),
        pflotran_h=dump("pflotran_solution.jld", "/h", np.float64),
    )
    assert data["node1"].shape == (6314,) and data["xs"].shape == (2106,) and data["pflotran_h"].shape == (2106,)
    np.savez_compressed(os.path.join(HERE, "fourfractures.npz"), **data)
    print({k: v.shape for k, v in data.items()})


if __name__ == "__main__":
    main()
