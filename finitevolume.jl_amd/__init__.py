"""finitevolume.jl_amd — MI355X-native implementation of FiniteVolume.jl's hot path.

The directory name carries a dot, so import it through the loader:

    from __graft_entry__ import load_package
    fv = load_package()          # registers the package as `fvamd`
    coords, neighbors, aol, volumes = fv.regulargrid(mins, maxs, ns)
    head, ch, A, b, freenode = fv.solvediffusion(neighbors, aol, K, sources, dnodes, dheads)

Function names, argument orders and return tuples are those of the Julia package
(nothing is exported there either: callers write FiniteVolume.f).
"""
from ._lib import Context, FVError, default_context, load  # noqa: F401
from .core import (  # noqa: F401
    SQRT_EPS,
    ConvergenceHistory,
    DeviceMatrix,
    DeviceVector,
    Problem,
    SparseMatrixCSC,
    assembleA,
    assembleb,
    freenodes2nodes,
    getfreenodes,
    getnodei2dirichleti,
    nodehycos2neighborhycos,
    regulargrid,
    solvediffusion,
)
from .adjoint import adjointintegrate, devicegradientintegral, getadjointfunctions, getcontinuoussolution, gradientintegrate, integratedfdplambda, transpose  # noqa: F401
from .transient import (  # noqa: F401
    DeviceOperator,
    DevicePCG,
    adaptivebackwardeulerstep,
    backwardeulerintegrate,
    backwardeuleronestep,
    backwardeulertwostep,
    defaultlinearsolver,
    diagonalupdate,
    fixedbackwardeulerstep,
    scalebyvolume,
)
from . import meshio, workloads  # noqa: F401,E402
