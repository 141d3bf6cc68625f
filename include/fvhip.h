/*
 * fvhip.h — C ABI of libfvhip.so, the MI355X (gfx950) implementation of
 * FiniteVolume.jl's hot path: grid/connectivity generation, Dirichlet
 * elimination + sparse assembly into CSR, the steady solve and the implicit
 * (backward-Euler) transient step, solved by Jacobi-preconditioned CG built
 * from hand-written HIP kernels.
 *
 * The reference (madsjulia/FiniteVolume.jl) is pure Julia and exposes no FFI:
 * its "interface" is a set of Julia functions.  Each entry point below names
 * the reference function(s) it replaces (file:line under /root/reference) —
 * the Julia shim (finitevolume.jl_amd/julia/FiniteVolumeHIP.jl) and the Python
 * mirror (finitevolume.jl_amd/) `ccall`/ctypes exactly these symbols.
 *
 * Conventions
 *   - Index arrays crossing the boundary are int64 and 1-BASED, as Julia's
 *     `Int` arrays are (neighbors' Pair halves, dirichletnodes, metaindex,
 *     colptr/rowval).  On the device they are int32 0-based.
 *   - Reals are double (Float64).
 *   - Every pointer argument may address host memory or device memory of the
 *     context's GPU; the library copies with hipMemcpyDefault during the call
 *     and never retains caller pointers.  Outputs go to caller-allocated
 *     buffers whose lengths come from the *_sizes queries.
 *   - Every function returns an int status (FV_OK == 0).  On failure
 *     fv_last_error() returns a message; for the reference's own validation
 *     errors the message text is the reference's (e.g. FiniteVolume.jl:26).
 *     No C++ exception or abort crosses the boundary.
 *   - One host thread per context; calls are synchronous at return.
 *     Not re-entrant on the same handle.
 */
#ifndef FVHIP_H
#define FVHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FVHIP_ABI_VERSION 4 /* 4: fv_step_form (the one-launch PCG iteration); 3: fv_fused_traversal, fv_trajectory_*, fv_observation_*, fv_adjoint_run, fv_param_gradient_integral_traj; 2: fv_ctx_set_option, fv_fused_form; fv_tune left the public header; fv_transient_run_adaptive fails when max_outer runs out */

enum {
    FV_OK = 0,
    FV_ERR_ARG = 1,
    FV_ERR_SOURCE_AT_DIRICHLET = 2, /* FiniteVolume.jl:25-27 */
    FV_ERR_INDEX = 3,               /* Julia BoundsError analogue */
    FV_ERR_NOMEM = 4,
    FV_ERR_HIP = 5,
    FV_ERR_STATE = 6,     /* call order (e.g. solve before assemble) */
    FV_ERR_DT = 7,        /* "time step must be positive", transient.jl:68-70 */
    FV_ERR_TOO_LARGE = 8, /* exceeds int32 device indexing */
    FV_ERR_COMM = 9       /* RCCL failure */
};

typedef struct fv_ctx fv_ctx;         /* one GPU + its streams (+ RCCL communicator) */
typedef struct fv_problem fv_problem; /* device-resident mesh, CSR operator, vectors */
typedef struct fv_trajectory fv_trajectory;   /* the stored states of a run (the reference's `us`, `ts`), kept in HBM */
typedef struct fv_observation fv_observation; /* observation rows with their uobs(t), sigma(t) series on the device */

/* What IterativeSolvers' ConvergenceHistory carries for the callers of
 * solvediffusion (FiniteVolume.jl:161,164: ch.isconverged, ch.data[:resnorm]). */
typedef struct fv_solve_info {
    int32_t converged;   /* ||r||_2 <= rtol*||b||_2 reached */
    int32_t iters;       /* PCG iterations performed */
    double relres;       /* final ||r||_2 / ||b||_2 (recurrence residual) */
    double bnorm;        /* ||b||_2 of the system solved */
    double solve_ms;     /* device time of the solve, HIP events */
    int64_t resnorm_len; /* entries written to the caller's resnorm buffer */
} fv_solve_info;

/* ---------------------------------------------------------------- lifecycle */
int fv_abi_version(void);
int fv_ctx_create(int device, fv_ctx **out);
void fv_ctx_destroy(fv_ctx *ctx);
int fv_ctx_synchronize(fv_ctx *ctx);
/* hipMemGetInfo of the context's device: how much of the 288 GB a problem occupies. */
int fv_device_mem_info(fv_ctx *ctx, int64_t *free_bytes, int64_t *total_bytes);
/* ctx may be NULL: last error of the calling thread outside any context. */
const char *fv_last_error(fv_ctx *ctx);
int fv_device_info(fv_ctx *ctx, char *name, int name_cap, int *compute_units, int64_t *total_mem_bytes);

/* ---------------------------------------------------------------- a1/a2: src/grid.jl */
/* regulargrid(mins, maxs, ns) sizes — grid.jl:61,69 */
int fv_regulargrid_sizes(const int64_t ns[3], int64_t *N, int64_t *F);
/* regulargrid — grid.jl:56-110.  coords is 3 x N column-major (may be NULL);
 * node1/node2 are the halves of neighbors::Vector{Pair{Int,Int}}. */
int fv_regulargrid(fv_ctx *ctx, const double mins[3], const double maxs[3], const int64_t ns[3], double *coords,
                   int64_t *node1, int64_t *node2, double *areasoverlengths, double *volumes);
/* nodehycos2neighborhycos — grid.jl:14-33 (nodehycos in (n3,n2,n1) column-major order == node order) */
int fv_nodehycos2neighborhycos(fv_ctx *ctx, int64_t F, const int64_t *node1, const int64_t *node2, int64_t N,
                               const double *nodehycos, int logtransformhyco, double *neighborhycos);

/* ---------------------------------------------------------------- a3/a4: src/FiniteVolume.jl:20-44 */
/* getfreenodes(n, dirichletnodes) -> freenode::Vector{Bool} (as u8), nodei2freenodei */
int fv_getfreenodes(fv_ctx *ctx, int64_t N, int64_t ndir, const int64_t *dirichletnodes, uint8_t *freenode,
                    int64_t *nodei2freenodei, int64_t *nfree);
/* getnodei2dirichleti(sources, dirichletnodes); *badnode receives the offending
 * node when FV_ERR_SOURCE_AT_DIRICHLET is returned. */
int fv_getnodei2dirichleti(fv_ctx *ctx, int64_t N, const double *sources, int64_t ndir, const int64_t *dirichletnodes,
                           int64_t *nodei2dirichleti, int64_t *badnode);

/* ---------------------------------------------------------------- problem = mesh + Dirichlet set */
/* Uploads the connectivity and builds, on the device, the free-node maps and
 * the symbolic CSR of assembleA (FiniteVolume.jl:75-108: rows sorted, repeats
 * merged) once; fv_assemble then only refills values. */
int fv_problem_create(fv_ctx *ctx, int64_t N, int64_t F, const int64_t *node1, const int64_t *node2,
                      const double *areasoverlengths, int64_t ndir, const int64_t *dirichletnodes, fv_problem **out);
/* Same, with regulargrid's connectivity generated directly on the device
 * (no F-sized host arrays: the 464^3 configuration has 3e8 faces). */
int fv_problem_create_regulargrid(fv_ctx *ctx, const double mins[3], const double maxs[3], const int64_t ns[3],
                                  int64_t ndir, const int64_t *dirichletnodes, fv_problem **out);
/* A general symmetric operator given as SparseMatrixCSC (1-based), for the
 * generic integrator entry backwardeulerintegrate(u0, A, b, dt0, t0, tfinal)
 * (transient.jl:123-154) and for the adjoint's transpose(A) (transient.jl:193):
 * all n unknowns are free, D = I.  The arrays are taken as CSR as they stand, so the matrix must be symmetric (checked on
 * the device, 1e-12 relative): FV_ERR_ARG with a message naming the first unmatched entry otherwise — the reference's
 * own non-symmetric cases (test/ode.jl:36) go through a host `linearsolver`, never through the device PCG. */
int fv_problem_create_from_csc(fv_ctx *ctx, int64_t n, const int64_t *colptr, const int64_t *rowval,
                               const double *nzval, fv_problem **out);
void fv_problem_destroy(fv_problem *p);
/* Whether fv_problem_create re-numbered the free cells for locality (reverse Cuthill-McKee; what the users of the reference
 * do by hand for DFN meshes numbered at random, examples/fractures), the mean |i - j| over the faces between free cells
 * before and after, and the seconds it took.  Purely internal: every free-indexed array crosses this ABI in the caller's
 * numbering (rank among the free nodes, FiniteVolume.jl:32-44), fv_get_csc / fv_get_b bit for bit as without it. */
int fv_problem_reorder_info(fv_problem *p, int32_t *reordered, double *mean_before, double *mean_after, double *seconds);
/* N cells, F faces, n free cells, nnz stored entries of A */
int fv_problem_sizes(fv_problem *p, int64_t *N, int64_t *F, int64_t *n, int64_t *nnz);
int fv_problem_get_free_maps(fv_problem *p, uint8_t *freenode, int64_t *nodei2freenodei);
/* grid arrays of a regulargrid-created problem (any pointer may be NULL) */
int fv_problem_get_grid(fv_problem *p, int64_t *node1, int64_t *node2, double *areasoverlengths, double *volumes);

/* ---------------------------------------------------------------- a5/a6/a7: assembly */
/* assembleA + assembleb (FiniteVolume.jl:75-139) in one pass over the faces.
 * K has nK entries; metaindex is NULL (the default i->i; nK == F, or nK == 1
 * meaning one conductivity for every face) or F 1-based indices into K
 * (`metaindex.(1:F)` pre-evaluated by the caller: closures cannot cross a C ABI).
 * Values are combined in face order, as sparse(I,J,V,n,n,+) does. */
int fv_assemble(fv_problem *p, int64_t nK, const double *conductivities, const int64_t *metaindex,
                int logtransformconductivity, const double *sources, const double *dirichletheads, int64_t *badnode);
/* The assembled A as SparseMatrixCSC{Float64,Int64} arrays (colptr n+1, rowval nnz,
 * nzval nnz; 1-based).  A is symmetric, so these are also its CSR arrays. */
int fv_get_csc(fv_problem *p, int64_t *colptr, int64_t *rowval, double *nzval);
int fv_get_b(fv_problem *p, double *b);
/* freenodes2nodes (FiniteVolume.jl:141-155): scatter n free values to N cells, fill Dirichlet heads */
int fv_freenodes2nodes(fv_problem *p, const double *result_free, double *head_nodes);

/* ---------------------------------------------------------------- a8: steady solve */
/* solvediffusion (FiniteVolume.jl:157-165): A x = b from x0 (NULL = zeros) by
 * Jacobi-PCG (replaces the reference's RS-AMG-PCG).  head_nodes
 * (N) and/or result_free (n) may be NULL.  resnorm (capacity resnorm_cap, may be
 * NULL) receives ||r||_2 after every iteration, like ch.data[:resnorm]. */
int fv_solve_steady(fv_problem *p, const double *x0_free, double rtol, int64_t maxiter, double *head_nodes,
                    double *result_free, double *resnorm, int64_t resnorm_cap, fv_solve_info *info);

/* ---------------------------------------------------------------- a9-a14: transient */
/* scalebyvolume! (transient.jl:7-22) folded into the operator: D = Ss*volumes[free].
 * volumes == NULL uses the grid's own volumes (regulargrid-created problems) or 1.
 * State slot 0 := u0[free] (transient.jl:170).  u0 may be NULL (zeros). */
int fv_transient_begin(fv_problem *p, double Ss, const double *volumes, const double *u0_nodes);
/* device-resident state vectors over the free cells */
int fv_state_alloc(fv_problem *p, int32_t *slot);
int fv_state_free(fv_problem *p, int32_t slot);
int fv_state_set_nodes(fv_problem *p, int32_t slot, const double *u_nodes);
int fv_state_set_free(fv_problem *p, int32_t slot, const double *u_free);
int fv_state_get_nodes(fv_problem *p, int32_t slot, double *u_nodes); /* freenodes2nodes, transient.jl:172 */
int fv_state_get_free(fv_problem *p, int32_t slot, double *u_free);
int fv_state_copy(fv_problem *p, int32_t src, int32_t dst);
/* ||u_a - u_b||_2 — the step-doubling error, transient.jl:81 */
int fv_state_norm2_diff(fv_problem *p, int32_t a, int32_t b, double *out);

#define FV_STEP_FORWARD 0 /* (I/dt + D^-1 A) u+ = u/dt + bhat          transient.jl:65-76 */
#define FV_STEP_ADJOINT 1 /* (I/dt + A D^-1) g+ = g/dt + bhat          transient.jl:188-205 (transpose(A)) */
/* backwardeuleronestep! (transient.jl:60-76): one implicit step of size dt from
 * slot src into slot dst (src == dst allowed), initial guess u_src.  Solved in
 * the equivalent SPD form (D/dt + A) u+ = D u/dt + b.  bhat_free is the
 * volume-scaled right-hand side getb(t) of the reference (n values) or NULL to
 * use the assembled b.  dt <= 0 -> FV_ERR_DT. */
int fv_transient_step(fv_problem *p, int32_t src, int32_t dst, double dt, const double *bhat_free, int mode,
                      double rtol, int64_t maxiter, fv_solve_info *info);
/* fixedbackwardeulerstep! x nsteps (transient.jl:130-154) without leaving the
 * device: slot is advanced in place; iters_per_step (nsteps, may be NULL). */
int fv_transient_run_fixed(fv_problem *p, int32_t slot, double dt, int64_t nsteps, double rtol, int64_t maxiter,
                           int32_t *iters_per_step, fv_solve_info *last_info, double *total_ms);
/* backwardeulerintegrate with the default stepper adaptivebackwardeulerstep! (transient.jl:78-121,136-154) and a
 * constant b, without leaving the device: step doubling (one step of dt against two of dt/2, accept below atol, grow
 * x2 below atol/4, on failure halve and sub-step to the requested time without overshoot), final step clipped to
 * tfinal.  The slot is advanced from t0 to tfinal; if max_outer outer steps do not get there the call returns FV_ERR_STATE
 * with the slot at u(t), t = ts_out[*n_outer] < tfinal (the reference never stops short, transient.jl:143-152); ts_out
 * (max_outer + 1 entries) receives the reference's `ts`: t0 and the time after every outer step; *n_outer their
 * count - 1, *n_solves the number of linear solves. */
int fv_transient_run_adaptive(fv_problem *p, int32_t slot, double t0, double tfinal, double dt0, double atol, double rtol,
                              int64_t maxiter, int64_t max_outer, double *ts_out, int64_t *n_outer, int64_t *n_solves,
                              fv_solve_info *last_info);

/* ---------------------------------------------------------------- parameter gradients (adjoint workflow)
 * The time integral of dfdp(t)' * lambda(t) over [ts[0], ts[nt-1]] with dfdp = (b_p - A_p u), optionally scaled by
 * D^-1 = 1 / (Ss volumes) as transientadjointutils.jl:23-30 does — the quantity gradientintegrate integrates
 * (transient.jl:208-219) and FiniteVolume.jl:271-377 (integrateb_pmA_pxlambda) unrolls by hand.  u (x_knots) and lambda
 * (lam_knots) are given at nt common knots, free-indexed, knot after knot (nt * n doubles each, host memory), and taken
 * as linear in between, so the result is exact.  Returned per face / per free row; the caller sums them into
 * p = [conductivities; sources; dirichletheads]:
 *   face_k[i]    -> conductivities[metaindex(i)]   (includes dc/dK: areasoverlengths, or the conductance itself when
 *                                                   logtransform != 0)
 *   face_dir[i]  -> dirichletheads[position of face i's Dirichlet end]   (0 for faces without exactly one)
 *   row_src[f]   -> sources[node of free row f]
 * Uses the conductances and heads of the last fv_assemble; scale_by_storage needs fv_transient_begin. */
int fv_param_gradient_integral(fv_problem *p, int64_t nt, const double *ts, const double *x_knots, const double *lam_knots,
                               int scale_by_storage, int logtransform, double *face_k, double *face_dir, double *row_src);
/* The same per-face / per-row terms at ONE time: the action of the pointwise parameter Jacobian dfdp(u, t, p)' = (b_p - A_p u)' D^-1
 * (src/transientadjointutils.jl:23-30: assembleb_p - assembleA_px, then scalebyvolume!) on a free-indexed vector lam, for callers that
 * hand dfdp(t) * lambda(t) to their own quadrature (src/transient.jl:208-219).  x_free, lam_free: n doubles each; outputs as above. */
int fv_param_jacobian_apply(fv_problem *p, const double *x_free, const double *lam_free, int scale_by_storage, int logtransform,
                            double *face_k, double *face_dir, double *row_src);

/* ---------------------------------------------------------------- kernel-level entry points (parity tests, roofline) */
/* y = (A + sigma*D) x on n free unknowns */
int fv_spmv(fv_problem *p, const double *x_free, double sigma, double *y_free);
/* reps back-to-back launches of the PCG SpMV kernel on resident vectors;
 * average launch duration from HIP events on the launch stream. */
int fv_bench_spmv(fv_problem *p, double sigma, int32_t reps, double *avg_ms);
int fv_dot(fv_problem *p, const double *a_free, const double *b_free, double *out);
/* Time every PCG kernel launch with HIP event pairs on the launch stream (for the
 * roofline figures of bench.py).  kernel: 0 = SpMV+dot, 1 = x/r update, 2 = p update. */
/* ---------------------------------------------------------------- preconditioner of the PCG
 * FV_PRECOND_JACOBI (default): diagonal scaling fused into the PCG's vector kernels.
 * FV_PRECOND_AMG: one cycle of an aggregation-based algebraic multigrid per iteration — the role
 * AlgebraicMultigrid.ruge_stuben + aspreconditioner play at FiniteVolume.jl:159-161 (solvediffusion).  Inside the library's
 * own PCG loop the first two coarse levels are solved by two flexible-CG steps each (K-cycle; the PCG around it is then the
 * flexible variant), on row blocks with the rank's own inner products; fv_amg_apply applies the plain V(1,1) cycle, a fixed symmetric operator.  The hierarchy
 * is built on the device at the first solve after fv_assemble (and again after the next fv_assemble /
 * fv_transient_begin); it carries the storage term, so shifted solves (implicit steps) can use it as well.
 */
#define FV_PRECOND_JACOBI 0
#define FV_PRECOND_AMG 1
/* steady solves: Jacobi-PCG for min(maxiter/4, 100) iterations, then AMG-PCG from that iterate for the rest — the
 * shape of the reference's defaultlinearsolver (transient.jl:50-58); implicit steps: Jacobi until one step needs more
 * than 50 iterations, the V-cycle from the next step on (large time steps). */
#define FV_PRECOND_AUTO 2
/* Row blocks of a distributed run (fv_dist_setup / fv_dist_setup_local): FV_PRECOND_AMG builds every rank's hierarchy on its own
 * diagonal block — block-Jacobi, no communication inside the preconditioner, an iteration count that grows with the rank count.
 * FV_PRECOND_AMG_GATHERED keeps the aggregation rank-local on level 0 but forms level 1 as the Galerkin product of the WHOLE
 * operator (the role of the one global hierarchy of AlgebraicMultigrid.ruge_stuben at FiniteVolume.jl:159-161), gathered on every
 * rank, and every rank builds and applies the levels below it for itself: two halo exchanges and one all-reduce of the level-1
 * right-hand side per cycle, replicated coarse work, and an iteration count that does not depend on the rank count.  On a whole
 * problem (one rank) it is FV_PRECOND_AMG.  Every rank must make the same choice, and building the hierarchy is collective: it happens
 * inside the first solve after fv_assemble / fv_transient_begin (or inside fv_amg_info), which every rank must then enter. */
#define FV_PRECOND_AMG_GATHERED 3
int fv_precond_set(fv_problem *p, int kind);
/* theta: strength threshold of the matching (0.10); omega: Jacobi damping of the smoother (0.85); passes: pairwise
 * passes per level (2 -> aggregates of ~4-5); rounds: handshake rounds per pass (10).  Process-wide. */
int fv_amg_configure(double theta, double omega, int passes, int rounds);
/* Builds the hierarchy if needed; rows[l], nnz[l] for l < min(*nlevels, cap). */
int fv_amg_info(fv_problem *p, int32_t *nlevels, int64_t *rows, int64_t *nnz, int32_t cap);
/* z = M^-1 r: one cycle on host vectors over the free cells (tests: symmetry, definiteness). */
int fv_amg_apply(fv_problem *p, const double *r_free, double sigma, double *z_free);

/* Per-context options (a context belongs to one host thread: nothing here is shared between contexts).
 *   FV_OPT_REORDER  locality re-numbering of the free cells of face-list meshes at fv_problem_create (fv_problem_reorder_info;
 *                   the reference's one printed throughput is such a mesh, examples/fractures/ex.jl:9-15): 0 never, 1 when the
 *                   mesh is numbered far worse than its size needs and the new order at least halves the mean distance between
 *                   the two cells of a face [default], 2 always.  Read when a problem is created in this context. */
#define FV_OPT_REORDER 1
/*   FV_OPT_LEAN_SETUP  fv_problem_create_regulargrid without face arrays, incident lists and canonical CSR in HBM: a regulargrid mesh is a
 *                   closed form (src/grid.jl:56-110), so fv_assemble computes b and the diagonal from rows formed on the fly and the
 *                   solver's storage forms are filled the same way — bit for bit the values assembleA (src/FiniteVolume.jl:75-108) gives.
 *                   ~190 B of HBM per cell instead of ~490 during a transient run, and no int32 CSR offsets: 8e8 cells on one GPU where
 *                   the CSR ends at 3e8.  Transient and steady Jacobi-PCG solves, fv_spmv, states and trajectories work as always, and
 *                   fv_get_csc writes assembleA's matrix out from the rows (a window at a time; the same arrays as from the CSR route);
 *                   fv_problem_get_grid and the parameter gradients generate the face arrays for the duration of the call (F < 2^31).
 *                   The AMG preconditioner writes level 0's CSR out for its set-up and gives it back (operators of < 2^31 entries).
 *                   fv_dist_setup lends the problem a CSR for the call and cuts the rank's rows out of it (same limit).  Dirichlet cells inside
 *                   the box (not only on its faces) are served: the few 64-row groups around them, which the CSR route hands to its CSR
 *                   kernel, have their rows formed and applied on the spot.  0 never, 1 every grid of >= 4096 cells, 2 [default]
 *                   grids whose CSR would not fit int32 offsets (7 N > 2^31: before, FV_ERR_TOO_LARGE).  Read when a problem is created. */
#define FV_OPT_LEAN_SETUP 2
int fv_ctx_set_option(fv_ctx *ctx, int option, int value);
int fv_ctx_get_option(fv_ctx *ctx, int option, int *value);
/* on = 1: HIP event pairs around every K1 / K2 / K3 launch of the PCG loop; on = 2: around K1 (the SpMV) only — an event
 * between two launches is a barrier (~10 us each at 464^3), so the timed region of the bench uses 2; 0: off. */
int fv_profile_enable(fv_problem *p, int on);
int fv_profile_get(fv_problem *p, int kernel, double *total_ms, int64_t *launches);
/* Storage form the most recent SpMV of this problem ran in (the `A * x` inside cg!, src/FiniteVolume.jl:161 and
 * src/transient.jl:52) and the bytes one launch of it has to move when every array is touched once:
 *   FV_SPMV_CSR      wave-stream CSR: 12 nnz + 20 n (SURVEY 8d's accounting)
 *   FV_SPMV_DIA      sliced-DIA, slice by slice: 8 B per stored lane-major value (zeros included) + 16 n + slice metadata
 *   FV_SPMV_DIA_MARCH  the same values, plane-marching traversal (x arms from registers)
 *   FV_SPMV_SYM_MARCH  symmetric plane-marching: 32 n (diagonal + 3 upper diagonals; 24 n + a code byte where the diagonal is
 *                      re-derived from the arms, fv_tune key 37) + 16 n
 *   FV_SPMV_SYM_TILE   the same arrays, tiled traversal (blocks own 1024 rows of a plane and march through planes, arms through
 *                      LDS): where the free rows are a regular box numbered like regulargrid's (fv_tune key 38)
 *   FV_SPMV_SELL     SELL-64 with 16-bit column offsets (irregular meshes whose locality re-numbering keeps every neighbour
 *                    within 32767 rows): 10 B per stored entry (64 x the longest row of each 64-row group; the diagonal first)
 *                    + 16 n + 5 B per group; groups that do not fit (rows of more than 32 entries, a far neighbour) stay CSR
 * Slices left to another form (CSR groups of an irregular part, slices the symmetric kernel hands to the slice kernel)
 * are counted in their own form, pro rata by slice count.  *form = -1 before the first SpMV. */
#define FV_SPMV_CSR 0
#define FV_SPMV_DIA 1
#define FV_SPMV_DIA_MARCH 2
#define FV_SPMV_SYM_MARCH 3
#define FV_SPMV_SYM_TILE 4
#define FV_SPMV_SELL 5
int fv_spmv_form(fv_problem *p, int32_t *form, int64_t *bytes_per_launch);
/* Bytes per row the most recent K2S launch (the fused vector update of a fixed-dt step in the one-iteration regime:
 * x += alpha p, r -= alpha q, the convergence sums and the next step's set-up) streams: 56 in the z-form (fv_tune key 36:
 * x_in, q, M^-1, D, z in and x_out, z' out), 64 otherwise (x_in, q, M^-1, D, r in and x_out, r, p' out); 7 / 8 fewer when the
 * storage term comes as codes / one double (key 35), 8 more when a dense b' is streamed too.  0 before the first such launch. */
int fv_update_form(fv_problem *p, int32_t *bytes_per_row);
/* The fused step of the one-iteration regime (fv_tune key 41; replaces the K1 + K2S pair behind src/transient.jl:60-76 when a
 * fixed-dt run's steps converge in one PCG iteration and the operator has the tiled symmetric form): launches so far on this
 * problem and the bytes per row its storage form moves with every array touched once — x, z, v in and x_out, z', v' out (48),
 * three upper diagonals (24), one storage code byte = 73 (+ 8 where the diagonal is streamed); *bytes_per_launch: the same
 * summed over the operator (rows whose product the slice-by-slice launch forms carry no matrix bytes here); 0 launches / 0
 * bytes when it has not run.  Where the matrix comes as 16-bit codes: 51.  On the SELL form of an irregular mesh (FV_SPMV_SELL):
 * 48 + the storage term (8 as a stream, 1 as codes) + 10 per stored entry, ~129 on a DFN mesh.  On row blocks the same launch runs
 * inside fv_dist_run_fixed's bursts. */
int fv_fused_form(fv_problem *p, int64_t *launches, int32_t *bytes_per_row, int64_t *bytes_per_launch);
/* The PCG loop of the most recent solve with several iterations (cg! of src/transient.jl:52 / src/FiniteVolume.jl:161): 0 = K1 + K2 +
 * K3 per iteration (the SpMV form's bytes + 88 per row); 113 = the direction update and the product as one pass of the fused kernel
 * (z = M^-1 r and p in, p' and q out, three upper diagonals, a code byte: 57) + the vector update in the z-form (x, z, p, q, M^-1 in,
 * x, z out: 56); 91 with the matrix as codes, 7 fewer again where M^-1 takes few distinct values and comes as a code byte.
 * Round 5: 89 (67 with the matrix as codes) = ONE launch per iteration on whole regular boxes — z, w, p, x in and z', p', w', x out (64), the
 * three upper diagonals (24 or 2), a code byte; the launch takes the verdict on the iterate and alpha, beta from sums the previous launch
 * left (the next iterate's r.z and r.r as polynomials in the step length), so that no vector-update launch and no M^-1 stream remain. */
int fv_loop_form(fv_problem *p, int32_t *bytes_per_row);
/* What the most recent solve with the one-launch loop (fv_loop_form 89 / 67) moved per row OUTSIDE its loop iterations: bytes[0] the
 * set-up (65 = the previous step's pending update and the carried residual in one pass, pcg_carry_flush_kernel; 64 = the carried
 * set-up K0' alone; 0 = another set-up), bytes[1] the first pass (direction = z: z, the matrix, a code byte in, w out: 41, or 19 with
 * the matrix as codes), bytes[2] the flush of the last update (48: z, w, p, x in, z, x out; 0 when the next step's set-up applies it);
 * *solves: solves on this problem whose loop ran that way.  A step of m iterations moves bytes[0] + bytes[1] + (m - 1) x fv_loop_form +
 * bytes[2] per row.  *bytes_total: a running total over EVERY Jacobi-PCG solve on this problem, whatever its regime — the bytes its launches had to
 * move with every array of every launch touched once (set-ups, products by their storage form, vector updates, fused steps, loop launches by
 * the iterations that ran); differences of it over a timed run give that run's algorithmic bytes (bench.py's regime rooflines).  AMG solves and
 * row blocks are not counted.  A measurement aid: the step is backwardeuleronestep! / cg! of src/transient.jl:50-76 either way. */
int fv_step_form(fv_problem *p, int32_t bytes[3], int64_t *solves, int64_t *bytes_total);
/* How the most recent fused launch (fv_fused_form / fv_loop_form) walked the planes of the operator: 0 = 2-D tiles of 16 lines x 128
 * columns (or it has not run), 1 = contiguous chunks of a plane's rows (no column halos; where the matrix comes as codes and the
 * rows whose diagonal does not follow from their arms share at most 15 values).  A measurement aid like the two above: the step it
 * reports on is backwardeuleronestep! / cg! of src/transient.jl:50-76 either way. */
int fv_fused_traversal(fv_problem *p, int32_t *kind);

/* ---------------------------------------------------------------- trajectories in HBM and the adjoint sweep over them
 * The reference keeps every outer state of a run on the host (`us`, `ts`: src/transient.jl:136-154), interpolates it linearly in
 * time (getcontinuoussolution, :176-180) and evaluates the adjoint's forcing dgdu(u_c, T - t) — non-zero on the observation rows
 * only (src/transientadjointutils.jl:13-21) — through that interpolant at every solve of adjointintegrate (:188-205).  These entry
 * points keep the states where they were computed: an fv_trajectory is a list of (time, free-cell vector in HBM), an
 * fv_observation the observation rows with uobs_i(t) and sigma(i, t) as piecewise-linear series on the device, and
 * fv_adjoint_run integrates the adjoint ODE with the reference's stepper without a host vector per solve.  A trajectory belongs
 * to its problem (destroy it first); knot times must increase. */
int fv_trajectory_create(fv_problem *p, fv_trajectory **out);                       /* needs fv_transient_begin */
int fv_trajectory_destroy(fv_trajectory *tr);
int fv_trajectory_clear(fv_trajectory *tr);
int fv_trajectory_push_state(fv_trajectory *tr, int32_t slot, double t);            /* append a copy of a state vector (push!(us, ...), transient.jl:144) */
int fv_trajectory_push_free(fv_trajectory *tr, const double *u_free, double t);     /* ... of a host vector over the free cells */
int fv_trajectory_size(fv_trajectory *tr, int64_t *nknots);
int fv_trajectory_times(fv_trajectory *tr, double *ts, int64_t cap);                /* `ts` */
int fv_trajectory_get_free(fv_trajectory *tr, int64_t k, double *u_free);           /* us[k+1] before freenodes2nodes */
int fv_trajectory_get_nodes(fv_trajectory *tr, int64_t k, double *u_nodes);         /* us[k+1] after it (transient.jl:172) */
int fv_trajectory_eval_free(fv_trajectory *tr, double t, double *u_free);           /* u_c(t), transient.jl:176-180; BoundsError outside the knots */
int fv_trajectory_reverse_time(fv_trajectory *tr, double T);                        /* knots reversed, t -> T - t (transient.jl:204) */
/* While a trajectory is set, fv_transient_run_adaptive pushes its initial state and the state of every outer step, and
 * fv_transient_run_fixed the state of every step at t0 + k dt (the initial state is the caller's to push); NULL stops it. */
int fv_trajectory_record(fv_problem *p, fv_trajectory *tr, double t0);
/* obs_free: 1-based free-cell indices (obsfreenodes); tobs[nt] increasing; uobs / sigma: nt x nobs, one row per knot (sigma NULL: 1). */
int fv_observation_create(fv_problem *p, int64_t nobs, const int64_t *obs_free, int64_t nt, const double *tobs, const double *uobs,
                          const double *sigma, fv_observation **out);
int fv_observation_destroy(fv_observation *o);
/* G = int_t0^t1 sum_i sigma(i,t)^2 (u_i(t) - uobs_i(t))^2 dt (g and G of transientadjointutils.jl:4-12, 46-49), exact for the
 * piecewise-linear series: a 6-point Gauss-Legendre rule between consecutive knots of either. */
int fv_observation_integral(fv_trajectory *u, fv_observation *o, double t0, double t1, double *G);
/* adjointintegrate(t -> dgdu(u_c, t), (t0, tfinal), ...) of src/transient.jl:188-205 with dgdu of transientadjointutils.jl:13-21:
 * gamma(0) = 0, d gamma/dt = transpose(D^-1 A) gamma + dgdu(T - t), the adaptive stepper (adaptive != 0; atol as in
 * backwardeulerintegrate) or the fixed one, first step dt0; lambda_out (empty on entry) receives lambda(t) = gamma(T - t) at
 * ascending times, T = tfinal.  Uses the operator of the last fv_assemble and the storage term of fv_transient_begin. */
int fv_adjoint_run(fv_problem *p, fv_trajectory *u, fv_observation *o, double t0, double tfinal, double dt0, int adaptive, double atol,
                   double rtol, int64_t maxiter, int64_t max_outer, fv_trajectory *lambda_out, int64_t *n_outer, int64_t *n_solves,
                   fv_solve_info *last_info);
/* fv_param_gradient_integral over [t0, t1] with u and lambda read from trajectories (their merged knots, both interpolated on
 * the device).  scale_by_storage: lambda_f / (Ss volume of the node behind f); or lam_scale_free[f] (host, free-indexed) as the
 * factor of lambda_f — the reference's scalebyvolume! of the transposed Jacobian divides by volumes[f] with the FREE index f
 * (src/transient.jl:25-34); both NULL / 0: no scaling. */
int fv_param_gradient_integral_traj(fv_problem *p, fv_trajectory *u, fv_trajectory *lam, double t0, double t1, int scale_by_storage,
                                    const double *lam_scale_free, int logtransform, double *face_k, double *face_dir, double *row_src);

/* ---------------------------------------------------------------- multi-GPU (RCCL over xGMI) */
#define FV_COMM_ID_BYTES 128
int fv_comm_unique_id(char id[FV_COMM_ID_BYTES]);
int fv_comm_init(fv_ctx *ctx, int nranks, int rank, const char id[FV_COMM_ID_BYTES]);
int fv_comm_destroy(fv_ctx *ctx);
/* All-reduces and halo exchanges issued through this context so far (also counted with one rank, where nothing travels);
 * reset != 0 zeroes the counters.  For tests of the collective pattern: two all-reduces per PCG iteration in the classic
 * form, one in the one-reduction form (fv_tune key 34), one per step in the one-iteration regime. */
int fv_comm_stats(fv_ctx *ctx, int64_t *allreduces, int64_t *halo_exchanges, int reset);
/* Where a distributed step's time goes on this rank: enable = 1 brackets, from now on, every all-reduce [0], every halo exchange
 * (on the second stream) [1], the compute stream's wait for the halo [2] and the interior [3] / boundary [4] SpMV passes of a row
 * block with HIP events and zeroes the sums; fv_comm_diag_get returns milliseconds and pair counts per category so far.  A rank
 * that waits for a slower one shows it in [0] and [2].  An event between two launches is a barrier: run a few steps with it
 * after the timed region, not inside it.  enable = 0 stops recording (the sums stay readable). */
int fv_comm_diag(fv_ctx *ctx, int enable);
int fv_comm_diag_get(fv_ctx *ctx, double total_ms[5], int64_t counts[5]);
/* Health check of the RCCL transport, to run once after fv_comm_init on every rank: a ring of ncclSend/ncclRecv (the halo
 * exchange's call pattern, on the halo stream) and an ncclAllReduce of `count` doubles; *ok = 1 when the data arrived. */
int fv_comm_selftest(fv_ctx *ctx, int64_t count, int *ok);

/* Row-block partition for one-process-per-GPU runs (no reference counterpart:
 * FiniteVolume.jl is single-process).  Call on a global problem after fv_assemble
 * and fv_transient_begin: rank `rank` of `nranks` gets a NEW problem holding its
 * contiguous range of free rows ([r*n/nranks, (r+1)*n/nranks)), columns renumbered
 * [local | halo]; the global problem can be destroyed afterwards.  State slot 0 of
 * the block is the rank's slice of the global slot 0. */
int fv_dist_setup(fv_problem *global, int nranks, int rank, fv_problem **block);
/* Row ranges chosen by the caller: rank r owns the free rows [bounds[r], bounds[r+1]) (bounds[0] = 0, bounds[nranks] = n,
 * the same array on every rank). */
int fv_dist_setup_bounds(fv_problem *global_problem, int nranks, int rank, const int64_t *bounds, fv_problem **block);
/* Row-block runs WITHOUT the global operator on every GPU: the problem of one slab of a structured grid — global node
 * arrays, but only the faces of the cells in the planes [i1_lo, i1_hi) (and the x-faces into plane i1_lo).  After
 * fv_assemble / fv_transient_begin (called with the global sources, heads, u0) its free rows of those planes are bit
 * for bit the global operator's; pass it to fv_dist_setup_bounds with bounds on plane boundaries, which
 * fv_problem_free_rows_before(problem, i1 * ns[1] * ns[2], &rows) converts into row numbers.  Per-face conductivities are
 * indexed by the slab's own faces (a contiguous range of the global face list); a single value or a metaindex work as
 * before. */
int fv_problem_create_regulargrid_slab(fv_ctx *ctx, const double mins[3], const double maxs[3], const int64_t ns[3], int64_t ndir,
                                       const int64_t *dirichletnodes, int64_t i1_lo, int64_t i1_hi, fv_problem **out);
int fv_problem_free_rows_before(fv_problem *p, int64_t node0, int64_t *rows);
int fv_dist_plan_sizes(fv_problem *block, int64_t *lo, int64_t *hi, int64_t *nnz_loc, int64_t *nhalo, int64_t *nsend,
                       int64_t *n_interior_groups, int64_t *n_boundary_groups);
/* The plan, 0-based (device convention): local CSR, global column of every halo slot,
 * per-rank receive/send counts, the concatenated send list (local rows) and the
 * 64-row groups that touch the halo.  Any pointer may be NULL. */
int fv_dist_get_plan(fv_problem *block, int64_t *rowptr_loc, int64_t *colind_loc, int64_t *halo_cols, int64_t *recv_counts,
                     int64_t *send_counts, int64_t *send_idx, int64_t *groups_bnd);
/* fixedbackwardeulerstep! x nsteps on the row block; halo exchange by grouped
 * ncclSend/ncclRecv on a second stream (overlapping the interior SpMV) and the PCG
 * scalars by ncclAllReduce.  Collective: every rank must call it with the same arguments. */
int fv_dist_run_fixed(fv_problem *block, double dt, int64_t nsteps, double rtol, int64_t maxiter, int32_t *iters_per_step,
                      fv_solve_info *last_info, double *total_ms);
/* Beyond the fixed-dt run, all collective (every rank calls them with the same scalars):
 * solvediffusion on row blocks (FiniteVolume.jl:157-165): Jacobi-PCG on A x = b from x0_local (NULL = 0), x_local = the
 * rank's rows of the solution;  one implicit step of the block's state with a caller's forcing, bhat_local = the rank's
 * rows of the volume-scaled getb(t) (NULL = the assembled b; transient.jl:60-76,165-174);  the default adaptive stepper
 * (step doubling, transient.jl:78-121,136-154, one extra all-reduce per trial for norm(onestep - twostep)) with the
 * semantics of fv_transient_run_adaptive. */
int fv_dist_solve_steady(fv_problem *block, const double *x0_local, double rtol, int64_t maxiter, double *x_local, fv_solve_info *info);
int fv_dist_step(fv_problem *block, double dt, const double *bhat_local, double rtol, int64_t maxiter, fv_solve_info *info);
int fv_dist_run_adaptive(fv_problem *block, double t0, double tfinal, double dt0, double atol, double rtol, int64_t maxiter,
                         int64_t max_outer, double *ts_out, int64_t *n_outer, int64_t *n_solves, fv_solve_info *last_info);
int fv_dist_spmv(fv_problem *block, const double *x_local, double sigma, double *y_local); /* collective */
/* Same kernels (interior pass + boundary pass) with the halo values supplied by the caller instead of
 * received from peers: lets one GPU rehearse any rank of an N-way partition.  Not a collective. */
int fv_dist_spmv_halo(fv_problem *block, const double *x_local, const double *halo_values, double sigma, double *y_local);
int fv_dist_state_get(fv_problem *block, double *u_local);

#ifdef __cplusplus
}
#endif
#endif /* FVHIP_H */
