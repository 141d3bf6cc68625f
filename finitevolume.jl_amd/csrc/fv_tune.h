// Private entry points of libfvhip.so: exported for the tools and tests of this repository, not declared in include/fvhip.h and not
// part of the drop-in boundary.
#pragma once
#ifdef __cplusplus
extern "C" {
#endif
/* Process-wide selectors for differential tests and A/B measurements (defaults in brackets).  Round 5 folded the panel into 18 keys (the
 * members of a family are bits of one key); round 4 had pruned it: of the 55 keys of round 3 the launch-shape, streaming-hint and diagnosis experiments are gone — frozen at their measured
 * best, the tools that drove them deleted, their logs kept under profiles/ — among them the two keys whose settings gave wrong
 * results by design (17, 29).  EVERY value of EVERY key below gives correct results: each names an alternative kernel, storage form
 * or policy that a test compares with the default.  Not synchronised: set it while no call of the library is in flight; tests and
 * tools reset what they set.  The same settings can be given as FV_TUNE="key=value,key=value" in the environment (read when the
 * first context is created).  What a caller may legitimately choose is per context (fv_ctx_set_option) or per problem
 * (fv_precond_set), in include/fvhip.h.
 *   7: fixed-dt runs carry the residual from step to step and recompute it from scratch every `value` steps [128]; 0 = every step
 *      computes its initial residual with an SpMV
 *   8: after a one-iteration step the vector update also prepares the next step's set-up (K2S, and with it the fused step) [1]
 *   9: plane-marching SpMV kernels on operators with a plane stride: 0 never, 1 when x outgrows the last-level cache, 2 always [1]
 *  13: one-iteration steps of a fixed-dt run enqueued per device poll; < 2 = poll after every step [8]
 *  14: fault injection for the tests of key 13: the chained step with this index of every burst is treated as not converged [-1]
 *  20: knots per device pass of fv_param_gradient_integral, 0 = as many as fit 2 GiB [0] (tests force several passes)
 *  21: a one-rank row-block run issues its all-reduces through RCCL anyway (tests of the call path on one GPU) [0]
 *  22: bursts take a step's verdict and the next step's scalars in one launch; row blocks all-reduce a step's five sums together
 *      with the next step's p.q (one 6-double collective per step, not two) [1]
 *  25: print the next N choices between the SpMV kernels to stderr (also FV_TRACE_SPMV=N in the environment) [0]
 *  27: the richest SpMV form a structured operator may take: 0 CSR wave-stream, 1 + sliced-DIA slice by slice, 2 + plane-marching
 *      sliced-DIA, 3 + symmetric plane-marching (diagonal + three upper diagonals), 4 + its tiled traversal [4]
 *  31: the locality re-numbering of face-list meshes — units digit: process-wide default of FV_OPT_REORDER (fv_ctx_set_option) for contexts that
 *      have not set it: 0 never, 1 when the mesh is numbered far worse than its size needs and the new order at least halves the mean face
 *      distance, 2 always; tens digit: 1 = computed by the host routine instead of fv_reorder.hip (the same order) [1]
 *  33: a fixed-dt run goes on from the residual, the prepared set-up and the refresh count the previous call on the same slot left [1]
 *  34: PCG of the row-block driver: 0 = the classic two-reduction form north_star names, 1 = Chronopoulos-Gear's one-reduction form [0]
 *  35: bits [7] — 1: the storage term Ss * volumes as one-byte codes (at most 16 distinct values) or one double instead of its stream;
 *      2: K2S in the z-form (only z = M^-1 r kept between two one-iteration steps); 4: zero row sum — slices whose stored diagonal is, bit for
 *      bit, minus the sum of the row's six arms (+ the folded sigma D by the row's storage code) are computed without the diagonal stream
 *  41: the fused family, bits [127] — 1: the fused step (fv_fused_form) in bursts of one-iteration steps (0 = K1 + K2S); 2: the many-iteration PCG
 *      loop through the fused kernel (fv_loop_form; 0 = K1 + K2 + K3); 4: the three upper diagonals as one 16-bit word of codes per row where each
 *      takes at most 32 distinct values (0 = doubles); 8: the fused step on row blocks (fv_dist_run_fixed); 16: the fused step on the SELL form
 *      (irregular meshes); 32: M^-1 as one-byte codes in the vector pass of the many-iteration loop; 64: a PCG iteration as ONE launch on whole
 *      regular boxes (round 5, fv_loop_form 89 / 67; 0 = the pass + vector-update pair)
 *  54: SELL-64 with 16-bit column offsets (FV_SPMV_SELL) for the groups the CSR wave-stream kernel would serve [1]
 *  60: the fused step / pass on contiguous chunks of a plane (fused_chunk_kernel, fused_chunkd_kernel): 1 = with the first / last plane's products
 *      formed by it too [1], 2 = those planes by the slice-by-slice launch, 0 = the 2-D tiles
 *  61: systems of at most this many rows are solved by the single-launch kernel of fv_small.hip [32768]; 0 = never
 * 18 keys (round 4: 27; round 3: 55).  The tests and tools still name the members of keys 31 / 35 / 41 by the numbers they had as keys of their own
 * (36, 37, 46, 47, 49, 50, 55, 59, 63): the Python binding translates them into the bits above (finitevolume.jl_amd/_lib.py, legacy_tune).
 *
 * Environment variables the library reads besides FV_TUNE (diagnostics and one differential switch, none changes a result):
 *   FV_TRACE_SPMV=N / FV_TRACE_FUSED=1 / FV_TRACE_REORDER=1  print kernel choices, fused-step eligibility, re-numbering decisions to stderr
 *   FV_TRACE_ALLOC=1     every device allocation / release that takes more than 0.1 ms, to stderr
 *   FV_AMG_VERBOSE=1     the AMG set-up phase by phase, to stderr
 *   FV_SMALL_TWOSTEP=0   small systems: the three solves of a step-doubling attempt one by one instead of in one launch (the same bits: tested)
 *   FV_AMG_KCYCLE=k      AMG K-cycle on the coarse levels 1 .. k (the PCG around it becomes flexible) [2]; 0 = V-cycle (round 5: was fv_tune key 52)
 *   FV_AMG_GALERKIN=sort the AMG's Galerkin products by the global stable sort instead of the row merge (the same bits: tested)
 *   FV_BAND=rows         band height of the CSR stream kernel's traversal order (experiments)
 *   FV_PLACE=0           the vectors a step writes from plain allocations instead of chosen by their write class (fv_place.hip; timing only)
 *   FV_ALLOC_SKEW=bytes  stagger consecutive large arrays inside their allocations by (k mod 16) x bytes (an experiment of rounds 2 and 5: no effect) */
int fv_tune(int key, int value);
/* Test infrastructure, like fv_tune: the loopback transport for rehearsals of the row-block driver on ONE device (RCCL refuses two ranks on one
 * GPU).  nranks host threads of one process, each with its own context on the same device, join the group `group_id`; halos then move by
 * device-to-device copies and the reductions are summed on the host in rank order.  Same plan, kernels and call sequence as the RCCL path
 * (fv_comm_init in include/fvhip.h).  Every rank's thread must make the same sequence of distributed calls.  Not part of the drop-in boundary
 * (round 5: moved out of the public header). */
struct fv_ctx;
int fv_comm_init_local(struct fv_ctx *ctx, int nranks, int rank, int group_id);
#ifdef __cplusplus
}
#endif
