// The fused time step of the one-iteration regime (fv_fused.hip) as the PCG driver sees it.
#pragma once
#include "fv_internal.h"

constexpr int FV_FUSED_PARTS = 4096; // per-block partial sums of one quantity (fused launch + the slice-by-slice launch behind it)

// per-block partial sums a fused launch leaves for the next one (two sets, by step parity: a launch reads the previous
// set in its first microseconds while its blocks write the other one at their end)
struct FusedSums {
    double *arz, *arr;       // the finished step's r.M^-1 r, r.r                 (nvec pieces)
    double *srz, *srr, *sbb; // the next step's set-up: rho.z, rho.rho (nvec), rhs.rhs (nbb)
    double *pq;              // z'.q' of the next step's product                 (npq)
    int nvec, nbb, npq;
    double *t2;              // the one-launch PCG iteration (fv_ploop_pass): its seventh sum, q.q
};

extern int g_fused, g_fused_dist, g_fused_dist_spare, g_fused_sell;
// fv_spmv.hip: the SELL form's grid, and the classic product (with partial x.y) of the groups outside it
int fv_sell_grid(fv_problem *p);
int fv_spmv_sell_rest(fv_problem *p, const double *x, double *y, const double *vals, double *partials, int *nparts);
bool fv_fused_streams_storage(fv_problem *p);
bool fv_fused_applicable(fv_problem *p, double sigma);
int fv_fused_prepare(fv_problem *p);
FusedSums fv_fused_sums(fv_problem *p, int parity);
int fv_fused_enter(fv_problem *p, double sigma);
int fv_fused_step(fv_problem *p, const double *x, double *x_next, double sigma, double dt, double rtol, int chain_index, int mode, const FusedSums &in,
                  bool force_prev_unconverged, const double *folded, int64_t bsupport, FusedSums *out_sums, const double *red = nullptr);
// row blocks (red: the six all-reduced sums, see fused_step_kernel): the z' the neighbours need into the send buffer
// before the launch; the boundary groups' classic products (left in p->qv2) into the v-form after it
int fv_fused_pack(fv_problem *p, const double *red, int mode, int chain_index, bool force_prev_unconverged, double rtol);
int fv_fused_convert_groups(fv_problem *p, const int32_t *groups, int64_t count, double sigma);
// fv_spmv.hip: y = A x (values with the shift folded in) over the slices the symmetric form leaves to the slice-by-slice
// kernel, partial x.y per block; use_done: a no-op once the solve's done flag is set
// vform_sigma != 0: y receives v = -M^-1 (q - vform_sigma D x) instead of q (the fused step's v-form), the partial sums are of x.q
// wform: y receives w = -M^-1 q (the many-iteration loop's form of the product)
int fv_spmv_rest(fv_problem *p, const double *x, double *y, const double *vals, double *partials, int *nparts, bool use_done = false, double vform_sigma = 0.0,
                 bool wform = false, bool irregular_only = false);
bool fv_fused_iteration_applicable(fv_problem *p, double sigma, bool folded);
// x: the iterate, updated in place by x += alpha_last * p (the lagging update of the previous iteration) when xapply is set
int fv_fused_iteration(fv_problem *p, int it, const double *folded, const double *part_rz, const double *part_rr, int nvec, int *npq, double *x, bool xapply);
// The one-launch PCG iteration (MODE 3 of the chunk kernels): see kf_ploop_prologue in fv_fused.hip
bool fv_ploop_applicable(fv_problem *p, double sigma, bool folded);
int fv_ploop_pass(fv_problem *p, int j, const double *folded, const double *z, const double *w, const double *pold, const double *xin, double *xout,
                  double *znew, double *pnew, double *wnew);
