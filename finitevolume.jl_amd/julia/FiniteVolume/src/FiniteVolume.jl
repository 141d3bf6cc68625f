# A package named FiniteVolume (the reference's name and UUID, /root/reference/Project.toml) over libfvhip.so, so that the
# reference's examples and tests — which say `import FiniteVolume` and call `FiniteVolume.f(...)` (test/theis.jl:1-3,31,52,54;
# examples/fractures/ex.jl:1) — resolve to the HIP path without an edited line:
#
#     julia> import Pkg; Pkg.develop(path = "finitevolume.jl_amd/julia/FiniteVolume")    # instead of the reference checkout
#     julia> include("test/theis.jl")                                                       # unchanged
#
# Everything lives in FiniteVolumeHIP.jl (one file up); this module binds each of its names under FiniteVolume.  The
# reference exports nothing either (callers qualify every name), so nothing is exported here.
# Written without a Julia runtime (none exists in the build image): checked statically by tests/test_cabi_exports.py.
module FiniteVolume

include(joinpath(@__DIR__, "..", "..", "FiniteVolumeHIP.jl"))

for name in names(FiniteVolumeHIP; all = true)
	text = string(name)
	(startswith(text, "#") || name in (:eval, :include, :FiniteVolumeHIP, :__init__)) && continue
	@eval const $name = FiniteVolumeHIP.$name
end

end # module
