# FiniteVolumeHIP.jl — Julia-side shim over libfvhip.so (include/fvhip.h).
#
# Keeps FiniteVolume.jl's function names, positional orders and return tuples for the
# accelerated path (src/FiniteVolume.jl:20-165, src/grid.jl:14-33,56-110,
# src/transient.jl:123-174), so the package's examples run unchanged after
#
#     import FiniteVolumeHIP; const FiniteVolume = FiniteVolumeHIP
#
# NOTE: written without a Julia runtime (none exists in the build image, SURVEY.md
# §8c) as a 1:1 mirror of the tested Python binding finitevolume.jl_amd/{_lib,core,
# transient}.py.  Every ccall below names a symbol declared in include/fvhip.h.
module FiniteVolumeHIP

import Libdl
import SparseArrays

const libfvhip = get(ENV, "FVHIP_LIB", joinpath(@__DIR__, "..", "libfvhip.so"))

struct SolveInfo            # fv_solve_info
	converged::Int32
	iters::Int32
	relres::Float64
	bnorm::Float64
	solve_ms::Float64
	resnorm_len::Int64
end

# What callers read from IterativeSolvers' ConvergenceHistory (FiniteVolume.jl:161,164)
struct ConvergenceHistory
	isconverged::Bool
	iters::Int
	data::Dict{Symbol, Any}
end

mutable struct Context
	handle::Ptr{Cvoid}
	function Context(device::Integer=0)
		h = Ref{Ptr{Cvoid}}(C_NULL)
		rc = ccall((:fv_ctx_create, libfvhip), Cint, (Cint, Ref{Ptr{Cvoid}}), device, h)
		rc == 0 || error(unsafe_string(ccall((:fv_last_error, libfvhip), Cstring, (Ptr{Cvoid},), C_NULL)))
		ctx = new(h[])
		finalizer(c->ccall((:fv_ctx_destroy, libfvhip), Cvoid, (Ptr{Cvoid},), c.handle), ctx)
		return ctx
	end
end

const defaultctx = Ref{Union{Nothing, Context}}(nothing)
function context()
	if defaultctx[] === nothing
		defaultctx[] = Context(0)
	end
	return defaultctx[]
end

function check(ctx::Context, rc)
	rc == 0 && return nothing
	# the library returns the reference's own message text for its validation errors
	error(unsafe_string(ccall((:fv_last_error, libfvhip), Cstring, (Ptr{Cvoid},), ctx.handle)))
end

mutable struct Problem      # fv_problem: mesh + Dirichlet set + CSR operator on the GPU
	handle::Ptr{Cvoid}
	ctx::Context
	N::Int
	F::Int
	n::Int
	nnz::Int
	function Problem(handle, ctx)
		N = Ref{Int64}(0); F = Ref{Int64}(0); n = Ref{Int64}(0); nnz = Ref{Int64}(0)
		check(ctx, ccall((:fv_problem_sizes, libfvhip), Cint, (Ptr{Cvoid}, Ref{Int64}, Ref{Int64}, Ref{Int64}, Ref{Int64}), handle, N, F, n, nnz))
		p = new(handle, ctx, N[], F[], n[], nnz[])
		finalizer(q->ccall((:fv_problem_destroy, libfvhip), Cvoid, (Ptr{Cvoid},), q.handle), p)
		return p
	end
end

splitneighbors(neighbors::Array{Pair{Int, Int}, 1}) = (Int64[first(p) for p in neighbors], Int64[last(p) for p in neighbors])

# ---------------------------------------------------------------- src/grid.jl
function regulargrid(mins, maxs, ns)
	@assert length(mins) == length(maxs)
	@assert length(mins) == length(ns)
	length(mins) == 3 || error("only 3 dimensions supported")
	ctx = context()
	ns64 = Int64[ns...]
	N = Ref{Int64}(0); F = Ref{Int64}(0)
	check(ctx, ccall((:fv_regulargrid_sizes, libfvhip), Cint, (Ptr{Int64}, Ref{Int64}, Ref{Int64}), ns64, N, F))
	coords = Array{Float64}(undef, 3, N[])
	n1 = Array{Int64}(undef, F[]); n2 = Array{Int64}(undef, F[])
	areasoverlengths = Array{Float64}(undef, F[])
	volumes = Array{Float64}(undef, N[])
	check(ctx, ccall((:fv_regulargrid, libfvhip), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Int64}, Ptr{Float64}, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Ptr{Float64}),
		ctx.handle, Float64[mins...], Float64[maxs...], ns64, coords, n1, n2, areasoverlengths, volumes))
	neighbors = [n1[i]=>n2[i] for i = 1:F[]]
	return coords, neighbors, areasoverlengths, volumes
end

function nodehycos2neighborhycos(neighbors, nodehycos, logtransformhyco=false)
	ctx = context()
	n1, n2 = splitneighbors(neighbors)
	nh = Float64[nodehycos...]   # (n3, n2, n1) column-major == node order, grid.jl:18-23
	out = Array{Float64}(undef, length(n1))
	check(ctx, ccall((:fv_nodehycos2neighborhycos, libfvhip), Cint, (Ptr{Cvoid}, Int64, Ptr{Int64}, Ptr{Int64}, Int64, Ptr{Float64}, Cint, Ptr{Float64}),
		ctx.handle, length(n1), n1, n2, length(nh), nh, logtransformhyco ? 1 : 0, out))
	return out
end

# ---------------------------------------------------------------- src/FiniteVolume.jl:20-44
function getfreenodes(n, dirichletnodes)
	ctx = context()
	freenode = Array{UInt8}(undef, n)
	nodei2freenodei = Array{Int64}(undef, n)
	nfree = Ref{Int64}(0)
	dn = Int64[dirichletnodes...]
	check(ctx, ccall((:fv_getfreenodes, libfvhip), Cint, (Ptr{Cvoid}, Int64, Int64, Ptr{Int64}, Ptr{UInt8}, Ptr{Int64}, Ref{Int64}),
		ctx.handle, n, length(dn), dn, freenode, nodei2freenodei, nfree))
	return freenode .!= 0, nodei2freenodei
end

function getnodei2dirichleti(sources, dirichletnodes)
	ctx = context()
	out = Array{Int64}(undef, length(sources))
	bad = Ref{Int64}(0)
	dn = Int64[dirichletnodes...]
	check(ctx, ccall((:fv_getnodei2dirichleti, libfvhip), Cint, (Ptr{Cvoid}, Int64, Ptr{Float64}, Int64, Ptr{Int64}, Ptr{Int64}, Ref{Int64}),
		ctx.handle, length(sources), Float64[sources...], length(dn), dn, out, bad))
	return out
end

# ---------------------------------------------------------------- assembly, FiniteVolume.jl:75-155
function createproblem(neighbors::Array{Pair{Int, Int}, 1}, areasoverlengths::Vector, numnodes::Int, dirichletnodes::Array{Int, 1})
	ctx = context()
	n1, n2 = splitneighbors(neighbors)
	h = Ref{Ptr{Cvoid}}(C_NULL)
	check(ctx, ccall((:fv_problem_create, libfvhip), Cint, (Ptr{Cvoid}, Int64, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Int64, Ptr{Int64}, Ref{Ptr{Cvoid}}),
		ctx.handle, numnodes, length(n1), n1, n2, Float64[areasoverlengths...], length(dirichletnodes), Int64[dirichletnodes...], h))
	return Problem(h[], ctx)
end

function assemble!(p::Problem, conductivities::Vector, sources::Vector, dirichletheads::Vector, metaindex, logtransformconductivity::Bool)
	# closures cannot cross the C ABI: metaindex.(1:F) is evaluated here (SURVEY.md §7 risk 6)
	mi = metaindex === nothing ? Ptr{Int64}(C_NULL) : Int64[metaindex(i) for i = 1:p.F]
	bad = Ref{Int64}(0)
	check(p.ctx, ccall((:fv_assemble, libfvhip), Cint, (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Int64}, Cint, Ptr{Float64}, Ptr{Float64}, Ref{Int64}),
		p.handle, length(conductivities), Float64[conductivities...], mi, logtransformconductivity ? 1 : 0, Float64[sources...], Float64[dirichletheads...], bad))
	return p
end

function getcsc(p::Problem)
	colptr = Array{Int64}(undef, p.n + 1); rowval = Array{Int64}(undef, p.nnz); nzval = Array{Float64}(undef, p.nnz)
	check(p.ctx, ccall((:fv_get_csc, libfvhip), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}), p.handle, colptr, rowval, nzval))
	return SparseArrays.SparseMatrixCSC(p.n, p.n, colptr, rowval, nzval)   # a genuine SparseMatrixCSC{Float64,Int64}
end

function getb(p::Problem)
	b = Array{Float64}(undef, p.n)
	check(p.ctx, ccall((:fv_get_b, libfvhip), Cint, (Ptr{Cvoid}, Ptr{Float64}), p.handle, b))
	return b
end

identitymetaindex(f) = f === nothing || f === identity

function assembleA(neighbors::Array{Pair{Int, Int}, 1}, areasoverlengths::Vector, conductivities::Vector, sources::Vector, dirichletnodes::Array{Int, 1}, dirichletheads::Vector, metaindex=nothing, logtransformconductivity::Bool=false)
	p = createproblem(neighbors, areasoverlengths, length(sources), dirichletnodes)
	# assembleA itself never validates the sources (only assembleb does, FiniteVolume.jl:111)
	assemble!(p, conductivities, zeros(length(sources)), dirichletheads, identitymetaindex(metaindex) ? nothing : metaindex, logtransformconductivity)
	return getcsc(p)
end

function assembleb(neighbors::Array{Pair{Int, Int}, 1}, areasoverlengths::Vector, conductivities::Vector, sources::Vector, dirichletnodes::Array{Int, 1}, dirichletheads::Vector, metaindex=nothing, logtransformconductivity::Bool=false)
	p = createproblem(neighbors, areasoverlengths, length(sources), dirichletnodes)
	assemble!(p, conductivities, sources, dirichletheads, identitymetaindex(metaindex) ? nothing : metaindex, logtransformconductivity)
	return getb(p)
end

function freenodes2nodes(result, sources, dirichletnodes, dirichletheads)
	getnodei2dirichleti(sources, dirichletnodes)   # validation, FiniteVolume.jl:142
	p = createproblem(Pair{Int, Int}[], Float64[], length(sources), Int[dirichletnodes...])
	assemble!(p, Float64[], Float64[sources...], Float64[dirichletheads...], nothing, false)
	head = Array{Float64}(undef, length(sources))
	check(p.ctx, ccall((:fv_freenodes2nodes, libfvhip), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), p.handle, Float64[result...], head))
	freenode, nodei2freenodei = getfreenodes(length(sources), dirichletnodes)
	return head, freenode, nodei2freenodei
end

# solvediffusion, FiniteVolume.jl:157-165.  The reference's RS-AMG-PCG becomes, by default, Jacobi-PCG for up to 100
# iterations followed by PCG with an aggregation-AMG V-cycle from that iterate (`maxiter` counts both phases).
# Non-convergence is reported in ch, as in the reference.
function solvediffusion(neighbors::Array{Pair{Int, Int}, 1}, areasoverlengths::Vector, conductivities::Vector, sources::Vector, dirichletnodes::Array{Int, 1}, dirichletheads::Vector; maxiter=400, rtol=sqrt(eps(Float64)), preconditioner=:auto)
	p = createproblem(neighbors, areasoverlengths, length(sources), dirichletnodes)
	assemble!(p, conductivities, sources, dirichletheads, nothing, false)
	# :amg = the aggregation-AMG V-cycle in the seat of AlgebraicMultigrid.ruge_stuben (FiniteVolume.jl:159-161)
	# :auto = Jacobi-PCG first, then AMG-PCG from that iterate (the shape of defaultlinearsolver, transient.jl:50-58)
	preconditioner in (:jacobi, :amg, :auto) || error("preconditioner must be :jacobi, :amg or :auto")
	check(p.ctx, ccall((:fv_precond_set, libfvhip), Cint, (Ptr{Cvoid}, Cint), p.handle, preconditioner == :amg ? 1 : (preconditioner == :auto ? 2 : 0)))
	head = Array{Float64}(undef, p.N)
	resnorm = Array{Float64}(undef, max(maxiter, 1))
	info = Ref(SolveInfo(0, 0, 0.0, 0.0, 0.0, 0))
	check(p.ctx, ccall((:fv_solve_steady, libfvhip), Cint, (Ptr{Cvoid}, Ptr{Float64}, Float64, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Ref{SolveInfo}),
		p.handle, C_NULL, rtol, maxiter, head, C_NULL, resnorm, length(resnorm), info))
	ch = ConvergenceHistory(info[].converged != 0, info[].iters, Dict{Symbol, Any}(:resnorm=>resnorm[1:info[].resnorm_len]))
	freenode, _ = getfreenodes(p.N, dirichletnodes)
	return head, ch, getcsc(p), getb(p), freenode
end

# ---------------------------------------------------------------- transient, src/transient.jl:60-174
struct DeviceVector          # a state slot of a Problem
	problem::Problem
	slot::Int32
end

function newstate(p::Problem)
	s = Ref{Int32}(0)
	check(p.ctx, ccall((:fv_state_alloc, libfvhip), Cint, (Ptr{Cvoid}, Ref{Int32}), p.handle, s))
	return DeviceVector(p, s[])
end
freestate(v::DeviceVector) = v.slot == 0 ? nothing : check(v.problem.ctx, ccall((:fv_state_free, libfvhip), Cint, (Ptr{Cvoid}, Int32), v.problem.handle, v.slot))

function nodevalues(v::DeviceVector)
	out = Array{Float64}(undef, v.problem.N)
	check(v.problem.ctx, ccall((:fv_state_get_nodes, libfvhip), Cint, (Ptr{Cvoid}, Int32, Ptr{Float64}), v.problem.handle, v.slot, out))
	return out
end

function normdiff(a::DeviceVector, b::DeviceVector)
	out = Ref{Float64}(0.0)
	check(a.problem.ctx, ccall((:fv_state_norm2_diff, libfvhip), Cint, (Ptr{Cvoid}, Int32, Int32, Ref{Float64}), a.problem.handle, a.slot, b.slot, out))
	return out[]
end

# backwardeuleronestep!, transient.jl:60-76 — b === nothing uses the assembled b on the device,
# otherwise b is the volume-scaled vector getb(t) of the reference
function backwardeuleronestep!(p::Problem, b, u_k::DeviceVector, dt; rtol=sqrt(eps(Float64)), maxiter=1000, mode=0)
	dt <= 0 && error("time step must be positive")
	dst = newstate(p)
	info = Ref(SolveInfo(0, 0, 0.0, 0.0, 0.0, 0))
	check(p.ctx, ccall((:fv_transient_step, libfvhip), Cint, (Ptr{Cvoid}, Int32, Int32, Float64, Ptr{Float64}, Cint, Float64, Int64, Ref{SolveInfo}),
		p.handle, u_k.slot, dst.slot, dt, b === nothing ? C_NULL : Float64[b...], mode, rtol, maxiter, info))
	return dst
end

# Step doubling (backwardeulertwostep!, transient.jl:78-87): one full step against two half steps;
# accept the two-half-step state when they agree to atol, otherwise hand back the first half step.
function backwardeulertwostep!(p::Problem, getb::Function, u_k, t, dt, atol, full=nothing; kwargs...)
	if full === nothing
		full = backwardeuleronestep!(p, getb(t), u_k, dt; kwargs...)
	end
	half = 0.5 * dt
	firsthalf = backwardeuleronestep!(p, getb(t), u_k, half; kwargs...)
	secondhalf = backwardeuleronestep!(p, getb(t + half), firsthalf, half; kwargs...)
	mismatch = normdiff(full, secondhalf)
	mismatch < atol && return secondhalf, dt, mismatch < atol / 4
	return firsthalf, half, false
end

# adaptivebackwardeulerstep! (transient.jl:89-121): try the requested dt; when it is rejected, cover
# it with accepted sub-steps (halving on rejection, doubling when the error is below atol/4, never
# overshooting), reusing a rejected trial's half-step state as the next trial's full step.
function adaptivebackwardeulerstep!(p::Problem, getb::Function, u_k, t, dt, atol, callback; kwargs...)
	callback(t, dt)
	state, taken, grow = backwardeulertwostep!(p, getb, u_k, t, dt, atol; kwargs...)
	taken < dt || return state, taken, grow
	covered = 0.0
	base = u_k
	want = taken
	reuse = true
	while covered < dt
		callback(t, dt)
		state, taken, grow = backwardeulertwostep!(p, getb, base, t + covered, want, atol, reuse ? state : nothing; kwargs...)
		if taken == want
			covered += taken
			base = state
			reuse = false
			grow && (want = 2 * taken)
		elseif taken < want
			want = taken
			reuse = true
		else
			error("Code is broken -- laststeptime should never be greater than targetdt")
		end
		want = min(want, dt - covered)
	end
	return state, taken, grow
end

function fixedbackwardeulerstep!(p::Problem, getb::Function, u_k, t, dt, atol, callback; kwargs...)
	callback(t, dt)
	return backwardeuleronestep!(p, getb(t), u_k, dt; kwargs...), dt, false
end

# backwardeulerintegrate, transient.jl:156-174 (constant-b and getb::Function methods)
function backwardeulerintegrate(u0, tspan, Ss::Number, volumes::Vector, neighbors::Array{Pair{Int, Int}, 1}, areasoverlengths::Vector, conductivities::Vector, sources::Vector, dirichletnodes::Array{Int, 1}, dirichletheads::Vector, metaindex=nothing, logtransformconductivity=false; kwargs...)
	return backwardeulerintegrate(u0, tspan, t->nothing, Ss, volumes, neighbors, areasoverlengths, conductivities, sources, dirichletnodes, dirichletheads, metaindex, logtransformconductivity; kwargs...)
end

function backwardeulerintegrate(u0, tspan, getb::Function, Ss::Number, volumes::Vector, neighbors::Array{Pair{Int, Int}, 1}, areasoverlengths::Vector, conductivities::Vector, sources::Vector, dirichletnodes::Array{Int, 1}, dirichletheads::Vector, metaindex=nothing, logtransformconductivity=false; dt0=1.0, stepper! =adaptivebackwardeulerstep!, atol=1e-4, callback=(t, dt)->nothing, rtol=sqrt(eps(Float64)), maxiter=1000)
	p = createproblem(neighbors, areasoverlengths, length(sources), dirichletnodes)
	assemble!(p, conductivities, sources, dirichletheads, identitymetaindex(metaindex) ? nothing : metaindex, logtransformconductivity)
	check(p.ctx, ccall((:fv_transient_begin, libfvhip), Cint, (Ptr{Cvoid}, Float64, Ptr{Float64}, Ptr{Float64}), p.handle, Ss, Float64[volumes...], Float64[u0...]))
	us = DeviceVector[DeviceVector(p, Int32(0))]
	ts = [tspan[1]]
	dt = min(dt0, tspan[2] - tspan[1])
	while ts[end] < tspan[2]   # transient.jl:142-152
		solution, laststeptime, increasestepsize = stepper!(p, getb, us[end], ts[end], dt, atol, callback; rtol=rtol, maxiter=maxiter)
		push!(us, solution)
		push!(ts, ts[end] + dt)
		dt = increasestepsize ? min(tspan[2] - ts[end], 2 * laststeptime) : min(tspan[2] - ts[end], laststeptime)
	end
	result = map(nodevalues, us)   # freenodes2nodes, transient.jl:172
	foreach(freestate, us)
	return result, ts
end

# The same integration (default stepper, constant b) entirely on the device: only u(tfinal) comes back, with the
# reference's `ts`.  For grids where the reference's per-step history (`us`) would not fit the host.
function backwardeulerintegrate_last(u0, tspan, Ss::Number, volumes::Vector, neighbors::Array{Pair{Int, Int}, 1}, areasoverlengths::Vector, conductivities::Vector, sources::Vector, dirichletnodes::Array{Int, 1}, dirichletheads::Vector, metaindex=nothing, logtransformconductivity=false; dt0=1.0, atol=1e-4, rtol=sqrt(eps(Float64)), maxiter=1000, maxsteps=1 << 20)
	p = createproblem(neighbors, areasoverlengths, length(sources), dirichletnodes)
	assemble!(p, conductivities, sources, dirichletheads, identitymetaindex(metaindex) ? nothing : metaindex, logtransformconductivity)
	check(p.ctx, ccall((:fv_transient_begin, libfvhip), Cint, (Ptr{Cvoid}, Float64, Ptr{Float64}, Ptr{Float64}), p.handle, Ss, Float64[volumes...], Float64[u0...]))
	ts = Array{Float64}(undef, maxsteps + 1)
	nouter = Ref{Int64}(0)
	nsolves = Ref{Int64}(0)
	info = Ref(SolveInfo(0, 0, 0.0, 0.0, 0.0, 0))
	check(p.ctx, ccall((:fv_transient_run_adaptive, libfvhip), Cint, (Ptr{Cvoid}, Int32, Float64, Float64, Float64, Float64, Float64, Int64, Int64, Ptr{Float64}, Ref{Int64}, Ref{Int64}, Ref{SolveInfo}),
		p.handle, Int32(0), tspan[1], tspan[2], dt0, atol, rtol, maxiter, maxsteps, ts, nouter, nsolves, info))
	return nodevalues(DeviceVector(p, Int32(0))), ts[1:nouter[] + 1]
end

# ---------------------------------------------------------------- adjoint hooks, transient.jl:176-216
# getcontinuoussolution: the piecewise-linear-in-time interpolants (Interpolations.jl in the reference; written out here)
function getcontinuoussolution(us::Vector{T}, ts::Vector) where {T <: AbstractArray}
	return t->begin
		(t < ts[1] || t > ts[end]) && throw(BoundsError(ts, t))
		lininterp(us, ts, t)
	end
end

function getcontinuoussolution(us::Vector{T}, ts::Vector, ::Type{Val{2}}) where {T <: AbstractArray}
	return (i, t)->begin
		(t < ts[1] || t > ts[end]) && throw(BoundsError(ts, t))
		lininterp(us, ts, t)[i]
	end
end

function freevalues(v::DeviceVector)
	out = Array{Float64}(undef, v.problem.n)
	check(v.problem.ctx, ccall((:fv_state_get_free, libfvhip), Cint, (Ptr{Cvoid}, Int32, Ptr{Float64}), v.problem.handle, v.slot, out))
	return out
end

# adjointintegrate(getdgdu, tspan, Ss, volumes, ...) — transient.jl:188-205.  gamma(t) = lambda(T - t) obeys
# dgamma/dt = transpose(D^-1 A) gamma + dgdu(T - t); with w = D^-1 gamma every implicit step is the same SPD solve as a
# forward step, which is what FV_STEP_ADJOINT (mode = 1) of fv_transient_step does.  The steppers above are reused as
# they are; states stay on the device, lambda comes back free-indexed and reversed in time like the reference's.
function adjointintegrate(getdgdu::Function, tspan, Ss::Number, volumes::Vector, neighbors::Array{Pair{Int, Int}, 1}, areasoverlengths::Vector, conductivities::Vector, sources::Vector, dirichletnodes::Array{Int, 1}, dirichletheads::Vector, metaindex=nothing, logtransformconductivity=false; dt0=1.0, stepper! =adaptivebackwardeulerstep!, atol=1e-4, callback=(t, dt)->nothing, rtol=sqrt(eps(Float64)), maxiter=1000)
	p = createproblem(neighbors, areasoverlengths, length(sources), dirichletnodes)
	assemble!(p, conductivities, zeros(length(sources)), dirichletheads, identitymetaindex(metaindex) ? nothing : metaindex, logtransformconductivity)
	check(p.ctx, ccall((:fv_transient_begin, libfvhip), Cint, (Ptr{Cvoid}, Float64, Ptr{Float64}, Ptr{Float64}), p.handle, Ss, Float64[volumes...], Ptr{Float64}(C_NULL)))
	check(p.ctx, ccall((:fv_state_set_free, libfvhip), Cint, (Ptr{Cvoid}, Int32, Ptr{Float64}), p.handle, Int32(0), zeros(p.n)))   # gamma0 = 0
	T = tspan[2]
	getb = t->getdgdu(T - t)
	gammas = DeviceVector[DeviceVector(p, Int32(0))]
	tsgamma = [tspan[1]]
	dt = min(dt0, tspan[2] - tspan[1])
	while tsgamma[end] < tspan[2]
		solution, laststeptime, increasestepsize = stepper!(p, getb, gammas[end], tsgamma[end], dt, atol, callback; rtol=rtol, maxiter=maxiter, mode=1)
		push!(gammas, solution)
		push!(tsgamma, tsgamma[end] + dt)
		dt = increasestepsize ? min(tspan[2] - tsgamma[end], 2 * laststeptime) : min(tspan[2] - tsgamma[end], laststeptime)
	end
	result = map(freevalues, gammas)
	foreach(freestate, gammas)
	return reverse(result), reverse(T .- tsgamma)
end

# gradientintegrate(lambda0, du0dp, dgdp, integrateddfdplambda, tspan) — transient.jl:213-216; the integral of dgdp by
# the composite Simpson rule over 64 panels (QuadGK in the reference; dgdp is identically zero in its workflows)
function gradientintegrate(lambda0::Vector, du0dp, dgdp, integrateddfdplambda::Vector, tspan; kwargs...)
	m = 64
	h = (tspan[2] - tspan[1]) / m
	I1 = (dgdp(tspan[1]) + dgdp(tspan[2])) * (h / 3)
	for k = 1:m - 1
		I1 += dgdp(tspan[1] + k * h) * ((isodd(k) ? 4 : 2) * h / 3)
	end
	return du0dp * lambda0 + I1 + integrateddfdplambda
end

# ---------------------------------------------------------------- gradients, transientadjointutils.jl:57-63
# integratedfdplambda with the reference's argument list (u2: the getcontinuoussolution(us, ts, 2) object is replaced by
# the stored states themselves, `us`/`ts_u`): the integral over tspan of dfdp(t)' * lambda(t) with the COMPLETE Jacobian
# b_p - A_p u (every face; D^-1 by node volume), exact for the piecewise-linear u and lambda (fv_param_gradient_integral).
function lininterp(vs::Vector, ts::Vector, t)
	k = clamp(searchsortedlast(ts, t), 1, length(ts) - 1)
	w = (t - ts[k]) / (ts[k + 1] - ts[k])
	return (1 - w) * vs[k] + w * vs[k + 1]
end

function integratedfdplambda(us::Vector, ts_u::Vector, p::Vector, lambdas::Vector, ts_lambda::Vector, tspan, Ss::Number, volumes::Vector, neighbors::Array{Pair{Int, Int}, 1}, areasoverlengths::Vector, conductivities::Vector, sources::Vector, dirichletnodes::Array{Int, 1}, dirichletheads::Vector, metaindex=nothing, logtransformconductivity::Bool=false)
	nK, N, nd = length(conductivities), length(sources), length(dirichletheads)
	pK, pd = p[1:nK], p[nK + N + 1:nK + N + nd]
	prob = createproblem(neighbors, areasoverlengths, N, dirichletnodes)
	assemble!(prob, pK, zeros(N), pd, identitymetaindex(metaindex) ? nothing : metaindex, logtransformconductivity)
	check(prob.ctx, ccall((:fv_transient_begin, libfvhip), Cint, (Ptr{Cvoid}, Float64, Ptr{Float64}, Ptr{Float64}), prob.handle, Ss, Float64[volumes...], zeros(N)))
	freenodes, nodei2freenodei = getfreenodes(N, dirichletnodes)
	knots = sort(unique(filter(t->tspan[1] <= t <= tspan[2], vcat(ts_u, ts_lambda, [tspan[1], tspan[2]]))))
	X = hcat([lininterp(us, ts_u, t)[freenodes] for t in knots]...)      # n x nt, column-major = knot after knot
	L = hcat([lininterp(lambdas, ts_lambda, t) for t in knots]...)
	facek = Array{Float64}(undef, prob.F); facedir = Array{Float64}(undef, prob.F); rowsrc = Array{Float64}(undef, prob.n)
	check(prob.ctx, ccall((:fv_param_gradient_integral, libfvhip), Cint, (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Cint, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
		prob.handle, length(knots), knots, X, L, 1, logtransformconductivity ? 1 : 0, facek, facedir, rowsrc))
	result = zeros(nK + N + nd)
	nodei2dirichleti = getnodei2dirichleti(sources, dirichletnodes)
	for (i, (node1, node2)) in enumerate(neighbors)
		result[metaindex === nothing ? i : metaindex(i)] += facek[i]
		if freenodes[node1] != freenodes[node2]
			result[nK + N + nodei2dirichleti[freenodes[node1] ? node2 : node1]] += facedir[i]
		end
	end
	for node = 1:N
		if freenodes[node]
			result[nK + node] = rowsrc[nodei2freenodei[node]]
		end
	end
	return result
end

end # module
