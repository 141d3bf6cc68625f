// Interface between the SpMV translation unit (fv_spmv.hip) and the solvers built on it (fv_pcg.hip).
#pragma once
#include "fv_internal.h"

// Optional epilogue that turns the first SpMV of an implicit step (q = A u) into the whole PCG set-up
// (see pcg_init_kernel<true>): r = b' - q, p = M^-1 r and the three partial sums, without writing q.
struct StepInitEpilogue {
    const double *bprime; // b' (may be null = 0)
    const double *D;
    const double *diagA;
    double *minv, *r, *pv;
    double *part_rz, *part_rr, *part_bb;
    double sigma, dt;
    int b_times_D, compute_minv;
    int q_shifted; // the SpMV used the shifted operator: r = rhs - q instead of b' - q
};

enum { SPMV_PLAIN = 0, SPMV_DOT = 1, SPMV_INIT = 2 };

// A subset of the operator's 64-row groups (interior / boundary part of a row block), split by storage form.
struct GroupSubset {
    const int32_t *dia = nullptr; // DIA slices of the subset
    int64_t ndia = 0;
    const int32_t *csr = nullptr; // its CSR groups
    int64_t ncsr = 0;
    // when the subset's groups are exactly the slices [win_lo, win_hi): lets the plane-marching kernel do its DIA part
    int64_t win_lo = 0, win_hi = 0;
};

// y = (A + sigma*D) x over the whole operator or a subset of its row groups.  mode SPMV_DOT also leaves per-block
// partials of x.y in `partials`; SPMV_INIT runs the step set-up epilogue instead of writing y.  `vals_override`: value
// array to use instead of p->vals (the folded copy).  *nparts = number of partials written.
int spmv_apply(fv_problem *p, const double *x, double *y, double sigma, const double *vals_override, int mode, double *partials,
               const StepInitEpilogue *epi_in, bool use_done, int *nparts, const GroupSubset *subset = nullptr);
// the folded value array for this sigma (built if needed), or nullptr when folding is not possible / switched off
int ensure_folded(fv_problem *p, double sigma, const double **out);
int fv_build_dia(fv_problem *p); // sliced-DIA copy of the grid-like slices (idempotent)

extern int g_spmv_form, g_fuse_init, g_fold_shift, g_use_dia; // fv_tune knobs the PCG driver looks at
