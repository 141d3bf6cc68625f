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
# transient,adjoint}.py.  Every ccall below names a symbol declared in include/fvhip.h, and
# every function of the reference's call surface (tests/golden/reference_api.json: name and
# positional arities taken from src/*.jl) is defined here with that arity — both checked by
# tests/test_cabi_exports.py.
module FiniteVolumeHIP

import Libdl
import LinearAlgebra
import SparseArrays

const libfvhip = get(ENV, "FVHIP_LIB", joinpath(@__DIR__, "..", "libfvhip.so"))
const FVHIP_ABI_VERSION = 4   # of include/fvhip.h this shim was written against
function __init__()
	have = ccall((:fv_abi_version, libfvhip), Cint, ())
	have == FVHIP_ABI_VERSION || error("libfvhip.so speaks ABI version $have, FiniteVolumeHIP.jl expects $FVHIP_ABI_VERSION: rebuild it (make -C finitevolume.jl_amd/csrc)")
end

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

# Per-context options (include/fvhip.h): FV_OPT_REORDER — locality re-numbering of face-list meshes (0 never / 1 auto / 2 always);
# FV_OPT_LEAN_SETUP — regular-grid problems without face arrays, incident lists and CSR in HBM (0 never / 1 every grid of >= 4096 cells /
# 2 [default] grids whose CSR would not fit int32 offsets: 8e8 cells on one GPU).  Read when a problem is created in the context.
const FV_OPT_REORDER = 1
const FV_OPT_LEAN_SETUP = 2
function setoption!(option::Integer, value::Integer; ctx::Context=context())
	check(ctx, ccall((:fv_ctx_set_option, libfvhip), Cint, (Ptr{Cvoid}, Cint, Cint), ctx.handle, option, value))
	return nothing
end
function getoption(option::Integer; ctx::Context=context())
	v = Ref{Cint}(0)
	check(ctx, ccall((:fv_ctx_get_option, libfvhip), Cint, (Ptr{Cvoid}, Cint, Ref{Cint}), ctx.handle, option, v))
	return Int(v[])
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

# The problem of regulargrid(mins, maxs, ns, ...) (src/grid.jl:56-110) without the face list crossing the boundary: the grid is generated on the
# device, and — FV_OPT_LEAN_SETUP — kept as a closed form where the face arrays and the CSR would not fit (beyond 3.06e8 cells) or are not wanted.
function createproblem(mins::Vector, maxs::Vector, ns::Vector, dirichletnodes::Array{Int, 1})
	ctx = context()
	h = Ref{Ptr{Cvoid}}(C_NULL)
	check(ctx, ccall((:fv_problem_create_regulargrid, libfvhip), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Int64}, Int64, Ptr{Int64}, Ref{Ptr{Cvoid}}),
		ctx.handle, Float64[mins...], Float64[maxs...], Int64[ns...], length(dirichletnodes), Int64[dirichletnodes...], h))
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

# ---------------------------------------------------------------- transient, src/transient.jl:1-174
# Two kinds of operator go through the same steppers, with the reference's stepper protocol
#     stepper!(rhs, A, getb, u_k, t, dt, linearsolver, atol, callback) -> (u_new, laststeptime, increasestepsize):
#   * a DeviceOperator (the assembled, volume-scaled operator resident on the GPU; states are DeviceVector slots) with the
#     DevicePCG "linear solver" — the default of every method below;
#   * a host matrix (Matrix or SparseMatrixCSC) with a user `linearsolver(A, rhs, x0) -> x`, e.g. (A, b, x0)->A \ b
#     (test/ode.jl:36): the reference's own host sequence, nothing touches the GPU.
struct DeviceVector          # a state slot of a Problem
	problem::Problem
	slot::Int32
end

struct DeviceOperator        # (D/dt + A) of a Problem after fv_transient_begin; adjoint: the transposed scaled operator
	problem::Problem
	adjoint::Bool
end
LinearAlgebra.transpose(A::DeviceOperator) = DeviceOperator(A.problem, !A.adjoint)
Base.size(A::DeviceOperator, i::Integer) = A.problem.n
Base.copy(A::DeviceOperator) = A             # transient.jl:140 copies A because it edits the diagonal; the device shift is not stored

struct DevicePCG             # the `linearsolver` of device operators: Jacobi-PCG (or the AMG V-cycle, fv_precond_set)
	rtol::Float64
	maxiter::Int
end
DevicePCG(; rtol=sqrt(eps(Float64)), maxiter=1000) = DevicePCG(rtol, maxiter)

function newstate(p::Problem)
	s = Ref{Int32}(0)
	check(p.ctx, ccall((:fv_state_alloc, libfvhip), Cint, (Ptr{Cvoid}, Ref{Int32}), p.handle, s))
	return DeviceVector(p, s[])
end
freestate(v::DeviceVector) = v.slot == 0 ? nothing : check(v.problem.ctx, ccall((:fv_state_free, libfvhip), Cint, (Ptr{Cvoid}, Int32), v.problem.handle, v.slot))
freestate(v) = nothing

function nodevalues(v::DeviceVector)
	out = Array{Float64}(undef, v.problem.N)
	check(v.problem.ctx, ccall((:fv_state_get_nodes, libfvhip), Cint, (Ptr{Cvoid}, Int32, Ptr{Float64}), v.problem.handle, v.slot, out))
	return out
end

function freevalues(v::DeviceVector)
	out = Array{Float64}(undef, v.problem.n)
	check(v.problem.ctx, ccall((:fv_state_get_free, libfvhip), Cint, (Ptr{Cvoid}, Int32, Ptr{Float64}), v.problem.handle, v.slot, out))
	return out
end

function setfree!(v::DeviceVector, u)
	check(v.problem.ctx, ccall((:fv_state_set_free, libfvhip), Cint, (Ptr{Cvoid}, Int32, Ptr{Float64}), v.problem.handle, v.slot, Float64[u...]))
	return v
end

function statedistance(a::DeviceVector, b::DeviceVector)   # norm(onestep - twostep), transient.jl:81
	out = Ref{Float64}(0.0)
	check(a.problem.ctx, ccall((:fv_state_norm2_diff, libfvhip), Cint, (Ptr{Cvoid}, Int32, Int32, Ref{Float64}), a.problem.handle, a.slot, b.slot, out))
	return out[]
end
statedistance(a::AbstractVector, b::AbstractVector) = LinearAlgebra.norm(a - b)

# diagonalupdate!, transient.jl:1-5,37-48 (host matrices; the device operator carries its shift in the kernels)
function diagonalupdate!(A::Array{T, 2}, increment) where {T}
	for k = 1:size(A, 1)
		A[k, k] += increment
	end
end

function diagonalupdate!(A::SparseArrays.SparseMatrixCSC, increment)
	rv = SparseArrays.rowvals(A)
	nz = SparseArrays.nonzeros(A)
	for col = 1:size(A, 2), k in SparseArrays.nzrange(A, col)
		rv[k] == col && (nz[k] += increment)
	end
end

# scalebyvolume!, transient.jl:7-35: b and the rows of A divided by the storage of the cell behind each free unknown;
# the Transpose method divides parent column i by volumes[i] — by the free index, as the reference does
function scalebyvolume!(b::Vector, volumes, freenodei2nodei)
	for k = 1:length(b)
		b[k] /= volumes[freenodei2nodei[k]]
	end
end

function scalebyvolume!(A::SparseArrays.SparseMatrixCSC, volumes, freenodei2nodei)
	rv = SparseArrays.rowvals(A)
	nz = SparseArrays.nonzeros(A)
	for col = 1:size(A, 2), k in SparseArrays.nzrange(A, col)
		nz[k] /= volumes[freenodei2nodei[rv[k]]]
	end
end

function scalebyvolume!(A::LinearAlgebra.Transpose{T, S}, volumes, freenodei2nodei) where {T, S <: SparseArrays.SparseMatrixCSC}
	nz = SparseArrays.nonzeros(A.parent)
	for col = 1:size(A.parent, 2), k in SparseArrays.nzrange(A.parent, col)
		nz[k] /= volumes[col]
	end
end

# backwardeuleronestep!, transient.jl:60-76.  Device operator: b === nothing means the assembled b resident on the device,
# otherwise b is the volume-scaled vector getb(t) of the reference (free-indexed, host).
function backwardeuleronestep!(rhs, A::DeviceOperator, b, u_k::DeviceVector, dt, linearsolver::DevicePCG, atol)
	dt <= 0 && error("time step must be positive")
	p = A.problem
	dst = newstate(p)
	info = Ref(SolveInfo(0, 0, 0.0, 0.0, 0.0, 0))
	check(p.ctx, ccall((:fv_transient_step, libfvhip), Cint, (Ptr{Cvoid}, Int32, Int32, Float64, Ptr{Float64}, Cint, Float64, Int64, Ref{SolveInfo}),
		p.handle, u_k.slot, dst.slot, dt, b === nothing ? Ptr{Float64}(C_NULL) : Float64[b...], A.adjoint ? 1 : 0, linearsolver.rtol, linearsolver.maxiter, info))
	return dst
end

function backwardeuleronestep!(rhs, A, b, u_k, dt, linearsolver, atol)   # host matrix + user linearsolver
	dt <= 0 && error("time step must be positive")
	@. rhs = b + u_k / dt
	diagonalupdate!(A, 1 / dt)
	onestep = linearsolver(A, rhs, u_k)
	diagonalupdate!(A, -1 / dt)
	return onestep
end

backwardeuleronestep!(rhs, A, getb::Function, u_k, t, dt, linearsolver, atol) = backwardeuleronestep!(rhs, A, getb(t), u_k, dt, linearsolver, atol)

# Step doubling (backwardeulertwostep!, transient.jl:78-87): one full step against two half steps; accept the
# two-half-step state when they agree to atol, otherwise hand back the first half step.
function backwardeulertwostep!(rhs, A, getb, u_k, t, dt, linearsolver, atol, full=nothing)
	if full === nothing
		full = backwardeuleronestep!(rhs, A, getb(t), u_k, dt, linearsolver, atol)
	end
	half = 0.5 * dt
	firsthalf = backwardeuleronestep!(rhs, A, getb(t), u_k, half, linearsolver, atol)
	secondhalf = backwardeuleronestep!(rhs, A, getb(t + half), firsthalf, half, linearsolver, atol)
	mismatch = statedistance(full, secondhalf)
	mismatch < atol && return secondhalf, dt, mismatch < atol / 4
	return firsthalf, half, false
end

# adaptivebackwardeulerstep! (transient.jl:89-121): try the requested dt; when it is rejected, cover it with accepted
# sub-steps (halving on rejection, doubling when the error is below atol/4, never overshooting), reusing a rejected
# trial's half-step state as the next trial's full step.
function adaptivebackwardeulerstep!(rhs, A, getb, u_k, t, dt, linearsolver, atol, callback)
	callback(t, dt)
	state, taken, grow = backwardeulertwostep!(rhs, A, getb, u_k, t, dt, linearsolver, atol)
	taken < dt || return state, taken, grow
	covered = 0.0
	base = u_k
	want = taken
	reuse = true
	while covered < dt
		callback(t, dt)
		state, taken, grow = backwardeulertwostep!(rhs, A, getb, base, t + covered, want, linearsolver, atol, reuse ? state : nothing)
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

function fixedbackwardeulerstep!(rhs, A, getb, u_k, t, dt, linearsolver, atol, callback)
	callback(t, dt)
	return backwardeuleronestep!(rhs, A, getb(t), u_k, dt, linearsolver, atol), dt, false
end

nocallback(t, dt) = nothing

# A symmetric host matrix as a device operator (D = I): the generic entry with the default solver
function deviceoperator(A::Union{SparseArrays.SparseMatrixCSC, Matrix})
	ctx = context()
	S = SparseArrays.SparseMatrixCSC{Float64, Int64}(SparseArrays.sparse(A))
	h = Ref{Ptr{Cvoid}}(C_NULL)
	# (refused with a message that points at `linearsolver` when A is not symmetric: the device solver is a CG)
	check(ctx, ccall((:fv_problem_create_from_csc, libfvhip), Cint, (Ptr{Cvoid}, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Ref{Ptr{Cvoid}}),
		ctx.handle, size(S, 2), S.colptr, S.rowval, S.nzval, h))
	p = Problem(h[], ctx)
	check(ctx, ccall((:fv_transient_begin, libfvhip), Cint, (Ptr{Cvoid}, Float64, Ptr{Float64}, Ptr{Float64}), p.handle, 1.0, Ptr{Float64}(C_NULL), Ptr{Float64}(C_NULL)))
	return DeviceOperator(p, false)
end
deviceoperator(A::DeviceOperator) = A
deviceoperator(A::LinearAlgebra.Transpose) = deviceoperator(SparseArrays.sparse(A))

# backwardeulerintegrate(u0, A, b | getb, dt0, t0, tfinal; ...), transient.jl:123-154.  Returns (us, ts) with host vectors
# over the unknowns of A, as the reference (history = :device keeps DeviceVector handles for the methods built on it).
function backwardeulerintegrate(u0, A, b::Vector, dt0, t0, tfinal; kwargs...)
	return backwardeulerintegrate(u0, A, t->b, dt0, t0, tfinal; kwargs...)
end

function backwardeulerintegrate(u0, A, getb::Function, dt0, t0, tfinal; stepper! =adaptivebackwardeulerstep!, linearsolver=DevicePCG(), atol=1e-4, callback=nocallback, history=:host)
	device = linearsolver isa DevicePCG
	if device
		A = deviceoperator(A)
		first = u0 isa DeviceVector ? u0 : setfree!(newstate(A.problem), u0)
		rhs = nothing
	else
		A isa DeviceOperator && error("a user linearsolver needs a host matrix")
		A = copy(A)                        # transient.jl:140: the steps edit the diagonal
		first = Float64[u0...]
		rhs = similar(first)
	end
	# the outer loop of transient.jl:141-152: the recorded time advances by the REQUESTED step (also when the stepper
	# sub-stepped); the next request is what the stepper reports it took, doubled if it asks for it, clipped to tfinal
	us, ts = Any[first], [t0]
	now, request = t0, min(dt0, tfinal - t0)
	while now < tfinal
		state, taken, grow = stepper!(rhs, A, getb, us[end], now, request, linearsolver, atol, callback)
		now += request
		push!(us, state)
		push!(ts, now)
		request = min(tfinal - now, (grow ? 2 : 1) * taken)
	end
	if device && history == :host
		result = map(freevalues, us)
		foreach(freestate, us)
		return result, ts
	end
	return us, ts
end

function transientproblem(u0, Ss, volumes, neighbors, areasoverlengths, conductivities, sources, dirichletnodes, dirichletheads, metaindex, logtransformconductivity)
	p = createproblem(neighbors, areasoverlengths, length(sources), dirichletnodes)
	assemble!(p, conductivities, sources, dirichletheads, identitymetaindex(metaindex) ? nothing : metaindex, logtransformconductivity)
	# assembleA + assembleb + scalebyvolume!, transient.jl:157-169: D = Ss * volumes[free] lives beside the operator
	check(p.ctx, ccall((:fv_transient_begin, libfvhip), Cint, (Ptr{Cvoid}, Float64, Ptr{Float64}, Ptr{Float64}), p.handle, Ss, Float64[volumes...], u0 === nothing ? Ptr{Float64}(C_NULL) : Float64[u0...]))
	return p
end

# backwardeulerintegrate(u0, tspan, [getb,] Ss, volumes, neighbors, ...), transient.jl:156-174.
# keep = :all stores every outer step like the reference (100 steps of 10^7 cells are 8 GB of device slots, then of host
# memory); keep = :last runs the default stepper with a constant b entirely on the device and returns ([u0, u(tfinal)], ts).
function backwardeulerintegrate(u0, tspan, Ss::Number, volumes::Vector, neighbors::Array{Pair{Int, Int}, 1}, areasoverlengths::Vector, conductivities::Vector, sources::Vector, dirichletnodes::Array{Int, 1}, dirichletheads::Vector, metaindex=nothing, logtransformconductivity=false; dt0=1.0, keep=:all, atol=1e-4, rtol=sqrt(eps(Float64)), maxiter=1000, maxsteps=1 << 20, kwargs...)
	if keep == :last
		isempty(kwargs) || error("keep = :last runs the default adaptive stepper with the device PCG and a constant b")
		p = transientproblem(u0, Ss, volumes, neighbors, areasoverlengths, conductivities, sources, dirichletnodes, dirichletheads, metaindex, logtransformconductivity)
		ts = Array{Float64}(undef, maxsteps + 1)
		nouter = Ref{Int64}(0)
		nsolves = Ref{Int64}(0)
		info = Ref(SolveInfo(0, 0, 0.0, 0.0, 0.0, 0))
		check(p.ctx, ccall((:fv_transient_run_adaptive, libfvhip), Cint, (Ptr{Cvoid}, Int32, Float64, Float64, Float64, Float64, Float64, Int64, Int64, Ptr{Float64}, Ref{Int64}, Ref{Int64}, Ref{SolveInfo}),
			p.handle, Int32(0), tspan[1], tspan[2], dt0, atol, rtol, maxiter, maxsteps, ts, nouter, nsolves, info))
		return [Float64[u0...], nodevalues(DeviceVector(p, Int32(0)))], ts[1:nouter[] + 1]
	end
	keep == :all || error("keep must be :all or :last")
	return backwardeulerintegrate(u0, tspan, t->nothing, Ss, volumes, neighbors, areasoverlengths, conductivities, sources, dirichletnodes, dirichletheads, metaindex, logtransformconductivity; dt0=dt0, atol=atol, rtol=rtol, maxiter=maxiter, kwargs...)
end

function backwardeulerintegrate(u0, tspan, getb::Function, Ss::Number, volumes::Vector, neighbors::Array{Pair{Int, Int}, 1}, areasoverlengths::Vector, conductivities::Vector, sources::Vector, dirichletnodes::Array{Int, 1}, dirichletheads::Vector, metaindex=nothing, logtransformconductivity=false; dt0=1.0, rtol=sqrt(eps(Float64)), maxiter=1000, linearsolver=DevicePCG(rtol, maxiter), kwargs...)
	linearsolver isa DevicePCG || error("the assembled operator lives on the device: pass a host matrix to backwardeulerintegrate(u0, A, b, dt0, t0, tfinal; linearsolver=...) to use a custom linearsolver")
	p = transientproblem(u0, Ss, volumes, neighbors, areasoverlengths, conductivities, sources, dirichletnodes, dirichletheads, metaindex, logtransformconductivity)
	us, ts = backwardeulerintegrate(DeviceVector(p, Int32(0)), DeviceOperator(p, false), getb, dt0, tspan[1], tspan[2]; linearsolver=linearsolver, history=:device, kwargs...)
	result = map(nodevalues, us)   # freenodes2nodes, transient.jl:172
	foreach(freestate, us)
	return result, ts
end

backwardeulerintegrate_last(args...; kwargs...) = (r = backwardeulerintegrate(args...; keep=:last, kwargs...); (r[1][2], r[2]))

# ---------------------------------------------------------------- adjoint hooks, transient.jl:176-216
# getcontinuoussolution: the piecewise-linear-in-time interpolants (Interpolations.jl in the reference).  Callable objects
# that keep their knots, so that integrals of products of two of them can be done exactly.
struct LinearInterpolant{D} <: Function   # callable, and a Function for the reference's ::Function signatures
	us::Vector
	ts::Vector{Float64}
end

function lininterp(vs::Vector, ts::Vector, t)
	(t < ts[1] || t > ts[end]) && throw(BoundsError(ts, t))
	k = clamp(searchsortedlast(ts, t), 1, length(ts) - 1)
	w = (t - ts[k]) / (ts[k + 1] - ts[k])
	return (1 - w) * vs[k] + w * vs[k + 1]
end
(uc::LinearInterpolant{1})(t) = lininterp(uc.us, uc.ts, t)
(uc::LinearInterpolant{2})(i, t) = lininterp(uc.us, uc.ts, t)[i]

getcontinuoussolution(us::Vector{T}, ts::Vector) where {T <: AbstractArray} = LinearInterpolant{1}(us, Float64[ts...])
getcontinuoussolution(us::Vector{T}, ts::Vector, ::Type{Val{2}}) where {T <: AbstractArray} = LinearInterpolant{2}(us, Float64[ts...])

# adjointintegrate(getdgdu, tspan, Ss, volumes, ...) — transient.jl:188-199: the transposed scaled operator on the device
# (with w = D^-1 gamma every implicit step is the same SPD solve as a forward step: FV_STEP_ADJOINT of fv_transient_step)
function adjointintegrate(getdgdu::Function, tspan, Ss::Number, volumes::Vector, neighbors::Array{Pair{Int, Int}, 1}, areasoverlengths::Vector, conductivities::Vector, sources::Vector, dirichletnodes::Array{Int, 1}, dirichletheads::Vector, metaindex=nothing, logtransformconductivity=false; rtol=sqrt(eps(Float64)), maxiter=1000, kwargs...)
	p = transientproblem(nothing, Ss, volumes, neighbors, areasoverlengths, conductivities, zeros(length(sources)), dirichletnodes, dirichletheads, metaindex, logtransformconductivity)
	return adjointintegrate(DeviceOperator(p, true), getdgdu, tspan; linearsolver=DevicePCG(rtol, maxiter), kwargs...)
end

# adjointintegrate(A, getdgdu, tspan) — transient.jl:201-205: gamma(t) = lambda(T - t) obeys dgamma/dt = A gamma + dgdu(T - t)
# with A the transposed operator; returned in terms of lambda (reversed in time).  A host matrix takes the host path when a
# `linearsolver` is given (A is rarely symmetric), a symmetric one without it is uploaded.
function adjointintegrate(A, getdgdu::Function, tspan; dt0=1.0, kwargs...)
	gamma0 = zeros(size(A, 2))
	T = tspan[2]
	gammas, tsgamma = backwardeulerintegrate(gamma0, A, t->getdgdu(T - t), dt0, tspan[1], tspan[2]; kwargs...)
	return reverse(gammas), reverse(T .- tsgamma)
end

# Integral of a vector-valued f over [lo, hi] cut at `knots` (where interpolants have kinks): 6-point Gauss-Legendre on
# every piece, halved until two levels agree to rtol — QuadGK.quadgk in the reference (transient.jl:208,214)
const GLX = (0.2386191860831969, 0.6612093864662645, 0.9324695142031521)
const GLW = (0.46791393457269104, 0.3607615730481386, 0.17132449237917036)
function gausspiece(f, a, b)
	c, h = 0.5 * (a + b), 0.5 * (b - a)
	acc = nothing
	for k = 1:3
		v = (f(c - h * GLX[k]) + f(c + h * GLX[k])) * (GLW[k] * h)
		acc = acc === nothing ? v : acc + v
	end
	return acc
end
function integratepiece(f, a, b, rtol, depth)
	whole = gausspiece(f, a, b)
	m = 0.5 * (a + b)
	halves = gausspiece(f, a, m) + gausspiece(f, m, b)
	(depth >= 12 || LinearAlgebra.norm(whole - halves) <= rtol * max(LinearAlgebra.norm(halves), eps(Float64))) && return halves
	return integratepiece(f, a, m, rtol, depth + 1) + integratepiece(f, m, b, rtol, depth + 1)
end
function integratevector(f, lo, hi, knots=Float64[]; rtol=sqrt(eps(Float64)))
	cuts = sort(unique(vcat([lo, hi], filter(t->lo < t < hi, knots))))
	total = nothing
	for k = 1:length(cuts) - 1
		v = integratepiece(f, cuts[k], cuts[k + 1], rtol, 0)
		total = total === nothing ? v : total + v
	end
	return total
end
knotsof(f) = f isa LinearInterpolant ? f.ts : Float64[]

# gradientintegrate, transient.jl:207-216 (both methods)
function gradientintegrate(lambdac::Function, du0dp, dgdp, dfdp::Function, tspan; kwargs...)
	I2 = integratevector(t->dfdp(t) * lambdac(t), tspan[1], tspan[2])
	return gradientintegrate(lambdac(0), du0dp, dgdp, I2, tspan; kwargs...)
end
function gradientintegrate(lambdac::LinearInterpolant{1}, du0dp, dgdp, dfdp::Function, tspan; kwargs...)
	I2 = integratevector(t->dfdp(t) * lambdac(t), tspan[1], tspan[2], lambdac.ts)
	return gradientintegrate(lambdac(0), du0dp, dgdp, I2, tspan; kwargs...)
end
function gradientintegrate(lambda0::Vector, du0dp, dgdp, integrateddfdplambda::Vector, tspan; kwargs...)
	I1 = integratevector(t->dgdp(t), tspan[1], tspan[2])
	return du0dp * lambda0 + I1 + integrateddfdplambda
end

# ---------------------------------------------------------------- src/transientadjointutils.jl
# d(b - A x)/dp for p = [conductivities; sources; dirichletheads] at fixed x (free-indexed), as a sparse (np x nfree) matrix:
# what the LinearAdjoints-generated assembleb_p / assembleA_px of the reference (transientadjointutils.jl:27-28) return,
# written out from the assembly loops at FiniteVolume.jl:75-139.
function parameterjacobian(x, neighbors, areasoverlengths, conductivities, sources, dirichletnodes, dirichletheads, metaindex, logtransformconductivity)
	nK, N, nd = length(conductivities), length(sources), length(dirichletheads)
	freenode, nodei2freenodei = getfreenodes(N, dirichletnodes)
	nodei2dirichleti = getnodei2dirichleti(zeros(N), dirichletnodes)
	I = Int[]; J = Int[]; V = Float64[]
	entry(i, j, v) = (push!(I, i); push!(J, j); push!(V, v))
	for node = 1:N
		freenode[node] && entry(nK + node, nodei2freenodei[node], 1.0)
	end
	for (i, (node1, node2)) in enumerate(neighbors)
		m = identitymetaindex(metaindex) ? i : metaindex(i)
		c = logtransformconductivity ? exp(conductivities[m]) * areasoverlengths[i] : conductivities[m] * areasoverlengths[i]
		dc = logtransformconductivity ? c : areasoverlengths[i]
		if freenode[node1] && freenode[node2]
			f1, f2 = nodei2freenodei[node1], nodei2freenodei[node2]
			entry(m, f1, -dc * (x[f1] - x[f2]))
			entry(m, f2, -dc * (x[f2] - x[f1]))
		elseif freenode[node1] || freenode[node2]
			fr, di = freenode[node1] ? (node1, node2) : (node2, node1)
			f, d = nodei2freenodei[fr], nodei2dirichleti[di]
			entry(m, f, dc * dirichletheads[d] - dc * x[f])
			entry(nK + N + d, f, c)
		end
	end
	return SparseArrays.sparse(I, J, V, nK + N + nd, sum(freenode))
end

function getadjointfunctions(sigma, obsfreenodes, uobs, u0, tspan, Ss::Number, volumes::Vector, neighbors::Array{Pair{Int, Int}, 1}, areasoverlengths::Vector, conductivities::Vector, sources::Vector, dirichletnodes::Array{Int, 1}, dirichletheads::Vector, metaindex=nothing, logtransformconductivity=false; kwargs...)
	nK, N, nd = length(conductivities), length(sources), length(dirichletheads)
	freenodes, nodei2freenodei = getfreenodes(length(u0), dirichletnodes)
	freenodei2nodei = Dict(nodei2freenodei[node]=>node for node = 1:length(u0) if freenodes[node])
	nfree = sum(freenodes)
	function g(u, t)
		uo, ue = uobs(t), u(t)
		return sum(sigma(i, t)^2 * (ue[freenodei2nodei[i]] - uo[freenodei2nodei[i]])^2 for i in obsfreenodes)
	end
	obsnodes = [freenodei2nodei[i] for i in obsfreenodes]   # the observed free unknowns as node indices
	function dgdu(u, t)   # transientadjointutils.jl:13-21: d/du of the weighted misfit, non-zero on the observed unknowns only
		misfit = u(t)[obsnodes] .- uobs(t)[obsnodes]
		weights = [sigma(i, t)^2 for i in obsfreenodes]
		gradient = zeros(nfree)
		gradient[obsfreenodes] .= 2 .* weights .* misfit
		return gradient
	end
	splitp(p) = (p[1:nK], p[nK + 1:nK + N], p[nK + N + 1:nK + N + nd])
	function dfdp(u, t, p)
		pK, ps, pd = splitp(p)
		M = parameterjacobian(u(t)[freenodes], neighbors, areasoverlengths, pK, ps, dirichletnodes, pd, metaindex, logtransformconductivity)
		# scalebyvolume!(transpose(b_p - A_px), Ss * volumes, ...), transientadjointutils.jl:29: the entries of free unknown i
		# divided by (Ss * volumes)[i] — the FREE index, as the reference's Transpose method does (transient.jl:25-34)
		return M * SparseArrays.spdiagm(0=>1 ./ (Ss .* volumes[1:nfree]))
	end
	dgdpval = zeros(nK + N + nd)
	dgdp(u, t, p) = dgdpval
	du0dp = SparseArrays.spzeros(nK + N + nd, nfree)
	function G(p::Vector)
		pK, ps, pd = splitp(p)
		us_p, ts_p = backwardeulerintegrate(u0, tspan, Ss, volumes, neighbors, areasoverlengths, pK, ps, dirichletnodes, pd, metaindex, logtransformconductivity; kwargs...)
		return G(getcontinuoussolution(us_p, ts_p))
	end
	G(uc_p) = integratevector(t->[g(uc_p, t)], tspan[1], tspan[2], vcat(knotsof(uc_p), knotsof(uobs)))[1]
	return g, dgdu, dfdp, dgdp, du0dp, G
end

# simpleintegrate, FiniteVolume.jl:262-269: the trapezoid rule over the stored knots
function simpleintegrate(fs, ts)
	result = 0.5 * ((ts[2] - ts[1]) * fs[1] + (ts[end] - ts[end - 1]) * fs[end])
	for k = 2:length(ts) - 1
		result += 0.5 * (ts[k + 1] - ts[k - 1]) * fs[k]
	end
	return result
end

# integratedfdplambda(u2, p, lambdas, ts_lambda, tspan, Ss, volumes, neighbors, ...), transientadjointutils.jl:57-63 ->
# FiniteVolume.jl:271-377.  u2 = getcontinuoussolution(us, ts, Val{2}).
#   complete = false (default): the reference's hand-unrolled terms, exactly the ones it carries — sources; for faces with
#     one Dirichlet end the head and conductivity terms with volumes taken by FREE index and the u product entering with a
#     plus sign; no free|free face terms; "not supported" without the log transform;
#   complete = true: the whole Jacobian b_p - A_p u, every face, D^-1 by node volume, integrated exactly on the device
#     (fv_param_gradient_integral) — the quantity gradientintegrate(lambdac, du0dp, dgdp, dfdp, tspan) integrates.
function integratedfdplambda(u2, p::Vector, lambdas::Vector, ts_lambda::Vector, tspan, Ss::Number, volumes::Vector, neighbors::Array{Pair{Int, Int}, 1}, areasoverlengths::Vector, conductivities::Vector, sources::Vector, dirichletnodes::Array{Int, 1}, dirichletheads::Vector, metaindex=nothing, logtransformconductivity=false; complete=false)
	nK, N, nd = length(conductivities), length(sources), length(dirichletheads)
	pK, pd = p[1:nK], p[nK + N + 1:nK + N + nd]
	freenodes, nodei2freenodei = getfreenodes(N, dirichletnodes)
	nodei2dirichleti = getnodei2dirichleti(zeros(N), dirichletnodes)
	result = zeros(nK + N + nd)
	mi(i) = identitymetaindex(metaindex) ? i : metaindex(i)
	if complete
		u2 isa LinearInterpolant || error("complete = true needs the stored solution: pass getcontinuoussolution(us, ts, Val{2})")
		prob = transientproblem(nothing, Ss, volumes, neighbors, areasoverlengths, pK, zeros(N), dirichletnodes, pd, metaindex, logtransformconductivity)
		knots = sort(unique(filter(t->tspan[1] <= t <= tspan[2], vcat(u2.ts, ts_lambda, [tspan[1], tspan[2]]))))
		X = hcat([lininterp(u2.us, u2.ts, t)[freenodes] for t in knots]...)      # n x nt, column-major = knot after knot
		L = hcat([lininterp(lambdas, ts_lambda, t) for t in knots]...)
		facek = Array{Float64}(undef, prob.F); facedir = Array{Float64}(undef, prob.F); rowsrc = Array{Float64}(undef, prob.n)
		check(prob.ctx, ccall((:fv_param_gradient_integral, libfvhip), Cint, (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Cint, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
			prob.handle, length(knots), knots, X, L, 1, logtransformconductivity ? 1 : 0, facek, facedir, rowsrc))
		for (i, (node1, node2)) in enumerate(neighbors)
			result[mi(i)] += facek[i]
			if freenodes[node1] != freenodes[node2]
				result[nK + N + nodei2dirichleti[freenodes[node1] ? node2 : node1]] += facedir[i]
			end
		end
		for node = 1:N
			freenodes[node] && (result[nK + node] = rowsrc[nodei2freenodei[node]])
		end
		return result
	end
	logtransformconductivity || error("not supported")
	lambda2 = getcontinuoussolution(lambdas, ts_lambda, Val{2})
	lambdaintegral = simpleintegrate(lambdas, ts_lambda)
	products = Dict{Int, Float64}()                  # the reference memoises the product integral per node
	productintegral(node) = get!(products, node) do
		integratevector(t->[lambda2(nodei2freenodei[node], t) * u2(node, t)], tspan[1], tspan[2], vcat(ts_lambda, knotsof(u2)))[1]
	end
	for node = 1:N
		freenodes[node] && (result[nK + node] += lambdaintegral[nodei2freenodei[node]] / (Ss * volumes[node]))
	end
	for (i, (node1, node2)) in enumerate(neighbors)
		freenodes[node1] == freenodes[node2] && continue
		fr, di = freenodes[node1] ? (node1, node2) : (node2, node1)
		f, d = nodei2freenodei[fr], nodei2dirichleti[di]
		c = exp(pK[mi(i)]) * areasoverlengths[i]
		storage = Ss * volumes[f]                     # by the free index, as the reference writes it
		result[mi(i)] += c * pd[d] * lambdaintegral[f] / storage + c * productintegral(fr) / storage
		result[nK + N + d] += c * lambdaintegral[f] / storage
	end
	return result
end

# ---------------------------------------------------------------- the adjoint workflow with every state in HBM (ABI 3)
# examples/transientadjoint/ex.jl:100-123 runs nine forward + adjoint pairs per objective call; the reference keeps `us` on the
# host and evaluates the forcing t -> dgdu(uc_p, t) through its interpolant at every solve (transient.jl:188-205,
# transientadjointutils.jl:13-21).  Here the states stay where they were computed (fv_trajectory), the observation series live on
# the device (fv_observation) and the sweep is one library call (fv_adjoint_run).  Usage, in the shape of the example:
#     uc   = devicesolution(u0, tspan, Ss, volumes, neighbors, areasoverlengths, conductivities, sources, dirichletnodes, dirichletheads, i->i, true; atol=atol, dt0=dt0)
#     obs  = deviceobservation(uc.problem, obsfreenodes, tobs, uobs_at_obs, sigma_at_obs)      # series at the observation rows, one row per knot of tobs
#     G    = objectiveintegral(uc, obs, tspan)                                                   # G(uc_p) of transientadjointutils.jl:46-49
#     lam  = adjointintegrate(uc, obs, tspan; atol=atol, dt0=dt0)                                # a DeviceSolution of lambda
#     idl  = integratedfdplambda(uc, lam, tspan, 1 ./ (Ss .* volumes)[1:nfree], true)            # per-face / per-row terms of the integral of dfdp' lambda
mutable struct DeviceTrajectory       # fv_trajectory: (time, free-cell vector in HBM) knots of a run — the reference's `us`, `ts`
	problem::Problem
	handle::Ptr{Cvoid}
	function DeviceTrajectory(p::Problem)
		h = Ref{Ptr{Cvoid}}(C_NULL)
		check(p.ctx, ccall((:fv_trajectory_create, libfvhip), Cint, (Ptr{Cvoid}, Ref{Ptr{Cvoid}}), p.handle, h))
		tr = new(p, h[])
		finalizer(t -> (t.handle == C_NULL || ccall((:fv_trajectory_destroy, libfvhip), Cint, (Ptr{Cvoid},), t.handle); t.handle = C_NULL), tr)
		return tr
	end
end

function knottimes(tr::DeviceTrajectory)
	n = Ref{Int64}(0)
	check(tr.problem.ctx, ccall((:fv_trajectory_size, libfvhip), Cint, (Ptr{Cvoid}, Ref{Int64}), tr.handle, n))
	ts = Array{Float64}(undef, n[])
	check(tr.problem.ctx, ccall((:fv_trajectory_times, libfvhip), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64), tr.handle, ts, n[]))
	return ts
end

function knotnodevalues(tr::DeviceTrajectory, k::Integer)     # us[k] after freenodes2nodes (transient.jl:172); k is 1-based
	u = Array{Float64}(undef, tr.problem.N)
	check(tr.problem.ctx, ccall((:fv_trajectory_get_nodes, libfvhip), Cint, (Ptr{Cvoid}, Int64, Ptr{Float64}), tr.handle, k - 1, u))
	return u
end

function knotfreevalues(tr::DeviceTrajectory, k::Integer)
	u = Array{Float64}(undef, tr.problem.n)
	check(tr.problem.ctx, ccall((:fv_trajectory_get_free, libfvhip), Cint, (Ptr{Cvoid}, Int64, Ptr{Float64}), tr.handle, k - 1, u))
	return u
end

struct DeviceSolution                # getcontinuoussolution of a trajectory: uc(t) over the free cells, interpolated on the device
	trajectory::DeviceTrajectory
	problem::Problem
end
function (uc::DeviceSolution)(t)
	u = Array{Float64}(undef, uc.problem.n)
	check(uc.problem.ctx, ccall((:fv_trajectory_eval_free, libfvhip), Cint, (Ptr{Cvoid}, Float64, Ptr{Float64}), uc.trajectory.handle, t, u))
	return u
end

# backwardeulerintegrate(...) of transient.jl:156-163 with `us` kept in HBM: the default stepper, the device PCG, a constant b
function devicesolution(u0, tspan, Ss::Number, volumes::Vector, neighbors::Array{Pair{Int, Int}, 1}, areasoverlengths::Vector, conductivities::Vector, sources::Vector, dirichletnodes::Array{Int, 1}, dirichletheads::Vector, metaindex=nothing, logtransformconductivity=false; dt0=1.0, atol=1e-4, rtol=sqrt(eps(Float64)), maxiter=1000, maxsteps=1 << 20)
	p = transientproblem(u0, Ss, volumes, neighbors, areasoverlengths, conductivities, sources, dirichletnodes, dirichletheads, metaindex, logtransformconductivity)
	tr = DeviceTrajectory(p)
	check(p.ctx, ccall((:fv_trajectory_record, libfvhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Float64), p.handle, tr.handle, tspan[1]))
	ts = Array{Float64}(undef, maxsteps + 1)
	nouter = Ref{Int64}(0)
	nsolves = Ref{Int64}(0)
	info = Ref(SolveInfo(0, 0, 0.0, 0.0, 0.0, 0))
	rc = ccall((:fv_transient_run_adaptive, libfvhip), Cint, (Ptr{Cvoid}, Int32, Float64, Float64, Float64, Float64, Float64, Int64, Int64, Ptr{Float64}, Ref{Int64}, Ref{Int64}, Ref{SolveInfo}),
		p.handle, Int32(0), tspan[1], tspan[2], dt0, atol, rtol, maxiter, maxsteps, ts, nouter, nsolves, info)
	ccall((:fv_trajectory_record, libfvhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Float64), p.handle, C_NULL, 0.0)
	check(p.ctx, rc)
	return DeviceSolution(tr, p)
end

mutable struct DeviceObservation      # fv_observation: obsfreenodes with uobs_i(t), sigma(i, t) as series over tobs (rows: knots)
	problem::Problem
	handle::Ptr{Cvoid}
end
function deviceobservation(p::Problem, obsfreenodes, tobs::Vector, uobs::Matrix, sigma::Union{Nothing, Matrix}=nothing)
	size(uobs) == (length(tobs), length(obsfreenodes)) || error("uobs must be length(tobs) x length(obsfreenodes)")
	h = Ref{Ptr{Cvoid}}(C_NULL)
	U = Float64[uobs[k, j] for j = 1:size(uobs, 2), k = 1:size(uobs, 1)]   # one row per knot in memory
	S = sigma === nothing ? Ptr{Float64}(C_NULL) : Float64[sigma[k, j] for j = 1:size(sigma, 2), k = 1:size(sigma, 1)]
	check(p.ctx, ccall((:fv_observation_create, libfvhip), Cint, (Ptr{Cvoid}, Int64, Ptr{Int64}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Ptr{Cvoid}}),
		p.handle, length(obsfreenodes), Int64[obsfreenodes...], length(tobs), Float64[tobs...], U, S, h))
	o = DeviceObservation(p, h[])
	finalizer(x -> (x.handle == C_NULL || ccall((:fv_observation_destroy, libfvhip), Cint, (Ptr{Cvoid},), x.handle); x.handle = C_NULL), o)
	return o
end

function objectiveintegral(uc::DeviceSolution, obs::DeviceObservation, tspan)    # G(uc_p), transientadjointutils.jl:46-49
	G = Ref{Float64}(0.0)
	check(uc.problem.ctx, ccall((:fv_observation_integral, libfvhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Float64, Float64, Ref{Float64}), uc.trajectory.handle, obs.handle, tspan[1], tspan[2], G))
	return G[]
end

# adjointintegrate(t->dgdu(uc_p, t), tspan, ...) of transient.jl:188-205 with dgdu evaluated on the device at T - t
function adjointintegrate(uc::DeviceSolution, obs::DeviceObservation, tspan; dt0=1.0, atol=1e-4, rtol=sqrt(eps(Float64)), maxiter=1000, maxsteps=1 << 20, fixedstep=false)
	p = uc.problem
	lam = DeviceTrajectory(p)
	nouter = Ref{Int64}(0)
	nsolves = Ref{Int64}(0)
	info = Ref(SolveInfo(0, 0, 0.0, 0.0, 0.0, 0))
	check(p.ctx, ccall((:fv_adjoint_run, libfvhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Float64, Float64, Cint, Float64, Float64, Int64, Int64, Ptr{Cvoid}, Ref{Int64}, Ref{Int64}, Ref{SolveInfo}),
		p.handle, uc.trajectory.handle, obs.handle, tspan[1], tspan[2], dt0, fixedstep ? 0 : 1, atol, rtol, maxiter, maxsteps, lam.handle, nouter, nsolves, info))
	return DeviceSolution(lam, p)
end

# the integral over tspan of dfdp(uc, t, p)' * lambda(t), per face (conductivity and Dirichlet-head terms) and per free row (sources):
# integratedfdplambda / gradientintegrate's I2 (transient.jl:208-219) with both factors read from HBM
function integratedfdplambda(uc::DeviceSolution, lam::DeviceSolution, tspan, lambdascale::Union{Nothing, Vector}, logtransformconductivity::Bool; scalebystorage::Bool=false)
	p = uc.problem
	facek = Array{Float64}(undef, p.F)
	facedir = Array{Float64}(undef, p.F)
	rowsrc = Array{Float64}(undef, p.n)
	check(p.ctx, ccall((:fv_param_gradient_integral_traj, libfvhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Float64, Cint, Ptr{Float64}, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
		p.handle, uc.trajectory.handle, lam.trajectory.handle, tspan[1], tspan[2], scalebystorage ? 1 : 0, lambdascale === nothing ? Ptr{Float64}(C_NULL) : Float64[lambdascale...], logtransformconductivity ? 1 : 0, facek, facedir, rowsrc))
	return facek, facedir, rowsrc
end

end # module
