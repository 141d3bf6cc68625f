// Private (experimental) entry points of libfvhip.so: exported for the tools and tests of this repository, not declared in
// include/fvhip.h and not part of the drop-in boundary.
#pragma once
#ifdef __cplusplus
extern "C" {
#endif
/* Process-wide kernel selection for A/B measurements (defaults in brackets).  NOT part of the public interface (include/fvhip.h):
 * an experimenter's panel for the tools and tests of this repository.  Not synchronised: set it while no call of the library is in
 * flight; every default is the measured best; tests and tools reset what they set.  The same settings can be given without any
 * call as FV_TUNE="key=value,key=value" in the environment (read when the first context is created).
 *   0: CSR SpMV form, 2 = wave-private CSR-stream [2], 1 = lanes-per-row;  1: unroll of the lanes-per-row form (2, 4, 8)
 *   2: plane-blocked group order [1];  3: fold sigma*D into a diagonal copy for fixed-dt runs [1]
 *   4: non-temporal matrix streams [1];  5: fuse the step set-up into the first SpMV's epilogue [0]
 *   6: sliced-DIA SpMV for grid-like 64-row slices [1]
 *   7: fixed-dt runs carry the residual from step to step and recompute it from scratch every `value` steps [128];
 *      0 = every step computes its initial residual with an SpMV
 *   8: in such runs, after a one-iteration step, the first vector update of a step also prepares the next step's
 *      set-up, so a step is SpMV + one fused vector pass [1]
 *   9: plane-marching sliced-DIA SpMV on operators with a plane stride (structured grids): 0 never, 1 when the x
 *      vector outgrows the last-level cache (key 19), 2 always [1]
 *  10: segments per XCD of that kernel, 0 = chosen per operator [0]
 *  11: sliced-DIA values packed (1) or padded to 8 blocks per slice (0); read when the DIA copy is built [1]
 *  12: the fused vector pass of key 8 takes the sparse b's share of |rhs|^2 from a gather over b's support: 1 = by extra
 *      blocks of the same launch, 2 = inside the vector blocks, 0 = b is streamed like the other vectors [1]
 *  13: one-iteration steps of a fixed-dt run enqueued per device poll (a step that needs more iterations stops the
 *      chain on the device and is resumed by the host); < 2 = poll after every step [8]
 *  14: fault injection for the tests of key 13: the chained step with this index of every burst is treated as not
 *      converged [-1]
 *  17: diagnosis switches of the marching kernel (bit 0: no in-plane arm loads, bit 1: no plane-arm edge loads);
 *      results are wrong when set [0]
 *  18: marching kernel: one 16-byte window access per step instead of centre + two edge loads when stride mod 64 <= 32 [1]
 *  19: MiB of x above which key 9 = 1 takes the marching kernel (below, x stays in the 256 MB infinity cache and the
 *      slice-by-slice kernel is faster; inside the stepping loop the crossover is at ~2e7 rows) [160]
 *  20: knots per device pass of fv_param_gradient_integral, 0 = as many as fit 2 GiB [0]
 *  21: a one-rank row-block run issues its all-reduces through RCCL anyway (tests of the call path on one GPU) [0]
 *  22: bursts of unpolled steps take a step's verdict and the next step's scalars in one launch; a row-block run also
 *      all-reduces a step's five sums together with the next step's p.q (one 6-double collective per step, not two) [1]
 *  25: print the next N choices between the two sliced-DIA kernels to stderr (also FV_TRACE_SPMV=N in the environment) [0]
 *  26: streaming hints of the fused vector pass of key 8: bit 0 = its read-once inputs bypass the caches, bit 1 = its
 *      x and r outputs too (the next SpMV's input stays cacheable), 7 = the search direction as well [3]
 *  27: symmetric plane-marching SpMV (stored diagonal + three upper diagonals, the lower arms read from the upper
 *      arrays) wherever the plane-marching kernel of key 9 runs and the operator is a symmetric 7-point one [1]
 *  28: streaming hints of that kernel: bit 0 = diagonal and plane-diagonal streams, bit 1 = the two in-plane upper
 *      diagonals (re-read as lower arms), bit 2 = the y store [4]
 *  29: diagnosis switches of that kernel (bit 0: no in-plane x arm loads, bit 1: no in-plane lower-value loads, bit 2: no
 *      window shuffles; results are wrong when set), bit 3: load the +-1 arms instead of taking them from the neighbouring
 *      lanes [0]
 *  30: blocks per CU the SpMV grids are sized for (the symmetric kernel takes 6 when this is left at 8) [8]
 *  31: process-wide default of FV_OPT_REORDER (fv_ctx_set_option) for contexts that have not set it: 0 never,
 *      1 when the mesh is numbered far worse than its size needs and the new order at least halves the mean distance
 *      between the two cells of a face, 2 always; read when the problem is created [1]
 *  32: experiment: large device arrays are handed out staggered by k x `value` bytes inside their allocations, so that the
 *      streams of a vector kernel do not start on the same HBM channels (no effect beyond run-to-run noise measured) [0]
 *  33: a fixed-dt run (fv_transient_run_fixed / fv_dist_run_fixed) goes on from the residual, the prepared set-up and the
 *      refresh count the previous call on the same slot left, when dt, assembly and storage are unchanged and nothing else
 *      has touched the state or solved in between: stepping in chunks then costs what one long call costs [1]
 *  34: PCG of the row-block driver in the many-iteration regime: 0 = the classic form north_star names (two all-reduces per
 *      iteration: p.q, then r.M^-1 r with r.r), 1 = the one-reduction form of Chronopoulos and Gear (one 3-double all-reduce
 *      per iteration; 96 instead of 88 bytes of vector traffic per row and a recurrence for A p) [0]
 *  35: K2S takes the storage term Ss * volumes as one-byte codes into a table when it has at most 16 distinct values (a
 *      regular grid with a scalar Ss: the cell volume and its half, quarter and eighth on the faces, edges and corners of
 *      the box), as one double when it has one, instead of streaming it: 7 or 8 bytes per row fewer; 0 = always stream [1]
 *  36: K2S in the z-form: between two one-iteration steps only the Jacobi-scaled residual z = M^-1 r (which is the next
 *      step's first direction) is kept, and r is taken from it as z / M^-1 where it is needed: 56 instead of 64 bytes per
 *      row; needs M^-1 > 0 on every row; 0 = keep r and z [1]
 *  37: zero row sum in the symmetric plane-marching SpMV: slices in which every row's stored diagonal is, bit for bit, minus
 *      the sum of its six off-diagonals in assembly order (plus the folded sigma D, taken by the row's storage code) — rows
 *      without a Dirichlet neighbour — are computed without the diagonal stream: 40 (41) instead of 48 bytes per row;
 *      0 = always stream the diagonal [1]
 *  38: the tiled traversal of the symmetric form (FV_SPMV_SYM_TILE) where it applies; 0 = always the plane-marching kernel [1]
 *  39, 40: experiments on the tiled kernel's launch: resident blocks per CU its grid is sized for [2], segments of planes per
 *      tile column (0 = chosen to fill whole rounds of the resident blocks) [0]
 *  41: the fused step (fv_fused_form) in bursts of one-iteration steps where it applies; 0 = always K1 + K2S [1]
 *  42, 43: experiments on its launch: resident blocks per CU for 8-line tiles [2], segments of planes per tile (0 = chosen) [0]
 *  44: lines per tile of the fused step: 16 (blocks of 1024 threads, one per CU) or 8 (512 threads, two per CU) [16]
 *  46: the many-iteration PCG loop through the fused kernel too (fv_loop_form) [1]
 *  47: the locality re-numbering of FV_OPT_REORDER computed on the device (fv_reorder.hip) [1]; 0 = by the host routine
 *  48: experiment: blocks of the device re-numbering's walk (0 = a sixteenth of the CUs, at most 16) [0]
 *  49: the fused kernel reads the three upper diagonals as one 16-bit word of codes per row where each takes at most 32 distinct
 *      values (a homogeneous conductivity on a regular grid): 2 instead of 24 bytes of matrix per row; 0 = always the doubles [1]
 *  50: the fused step on row blocks (fv_dist_run_fixed) too; 0 = row blocks keep the K1 + K2S pair [1]
 *  51: CUs per XCD a row block's fused launch leaves to the halo exchange's kernel [1]
 *  52: AMG K-cycle: the coarse levels 1 .. value are solved by two flexible-CG steps preconditioned by the cycle below them, the PCG
 *      around the cycle becomes flexible [2]; 0 = V-cycle
 *  53: AMG coarse levels with at least this many rows run the wave-stream CSR kernel [65536]; 0 = always the lanes-per-row kernel
 *  54: SELL-64 with 16-bit column offsets (FV_SPMV_SELL) for the 64-row groups the CSR wave-stream kernel would serve [1]
 *  55: the fused step on the SELL form (irregular meshes) [1]
 *  56: experiment: resident blocks per CU the SELL step's grid is sized for [4]
 *  58: experiment: resident blocks per CU the SELL SpMV's grid is sized for [8]
 *  59: M^-1 as one-byte codes in the vector pass of the many-iteration loop where it takes at most 16 distinct values [1]
 *  45: streaming-hint experiments on the fused step (bit 0: z' stored non-temporally, 1: v' too, 2: x / v plain loads, 3: x_out
 *      plain store, 4: matrix plain loads) [0] */
int fv_tune(int key, int value);
#ifdef __cplusplus
}
#endif
