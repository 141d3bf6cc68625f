/* Threaded CPU baseline for bench.py's `cpu_baseline` leg ONLY (test infrastructure like fv_oracle.c: nothing the product
 * ships or calls).  The same unpreconditioned CG as fvo_cg (IterativeSolvers 0.8.1 cg!, /root/reference/src/transient.jl:50-58 calls
 * it per time step) spread over the host's cores with OpenMP: SURVEY 8(d) / BASELINE.md ask for the CPU path "additionally on all host
 * cores with the count printed".  Not a parity oracle — the dot products sum in thread order — and not the reference's own
 * threading either (Julia's SparseArrays mul! and IterativeSolvers run on one thread): it says what the host could do at best.
 * The product uses the symmetry of the UNSCALED matrix (assembleA, FiniteVolume.jl:96-99: CSC arrays == CSR arrays): row i gathers
 * sum_k nzval[k] x[rowval[k]] over column i's entries and divides by Ss V_i (the row scaling of scalebyvolume!), so that rows can be
 * shared out without atomics.  */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

typedef int64_t i64;

int fvo_mt_threads(void) { return omp_get_max_threads(); }

static void spmv_sym(i64 n, const i64 *colptr, const i64 *rowval, const double *nzval, const double *dvol, const double *x, double *y)
{
#pragma omp parallel for schedule(static)
    for (i64 i = 0; i < n; i++) {
        double s = 0.0;
        for (i64 k = colptr[i] - 1; k < colptr[i + 1] - 1; k++)
            s += nzval[k] * x[rowval[k] - 1];
        y[i] = s / dvol[i]; /* scalebyvolume! (transient.jl:7-22): row i of the matrix divided by Ss V_i */
    }
}

static double dot(i64 n, const double *a, const double *b)
{
    double s = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : s)
    for (i64 i = 0; i < n; i++)
        s += a[i] * b[i];
    return s;
}

/* steps of backwardeuleronestep! (transient.jl:60-76) with a fixed dt on (D^-1 A + I/dt) — A as assembled (symmetric, 1-based CSC),
 * dvol = Ss V of the free rows, b already scaled by volume, the shift applied on the fly — from the state u (in place):
 * rhs = b + u/dt, CG from x0 = u to tol. */
int fvo_mt_fixed_steps(i64 n, const i64 *colptr, const i64 *rowval, const double *nzval, const double *dvol, const double *b, double *u, double dt, i64 nsteps,
                       double tol, i64 maxiter, i64 *iters_total, int nthreads)
{
    if (nthreads > 0)
        omp_set_num_threads(nthreads);
    double *p = calloc((size_t)n + 1, sizeof(double)), *r = malloc(sizeof(double) * ((size_t)n + 1)), *c = malloc(sizeof(double) * ((size_t)n + 1)),
           *rhs = malloc(sizeof(double) * ((size_t)n + 1));
    if (!p || !r || !c || !rhs)
        return 4;
    const double sh = 1.0 / dt;
    i64 total = 0;
    for (i64 s = 0; s < nsteps; s++) {
#pragma omp parallel for schedule(static)
        for (i64 i = 0; i < n; i++) {
            rhs[i] = b[i] + u[i] / dt;
            p[i] = 0.0;
        }
        spmv_sym(n, colptr, rowval, nzval, dvol, u, c);
#pragma omp parallel for schedule(static)
        for (i64 i = 0; i < n; i++)
            r[i] = rhs[i] - (c[i] + sh * u[i]);
        double residual = sqrt(dot(n, r, r)), prev = 1.0;
        const double reltol = sqrt(dot(n, rhs, rhs)) * tol;
        i64 it = 0;
        while (!(it >= maxiter || residual <= reltol)) {
            const double beta = residual * residual / (prev * prev);
#pragma omp parallel for schedule(static)
            for (i64 i = 0; i < n; i++)
                p[i] = r[i] + beta * p[i];
            spmv_sym(n, colptr, rowval, nzval, dvol, p, c);
#pragma omp parallel for schedule(static)
            for (i64 i = 0; i < n; i++)
                c[i] += sh * p[i];
            const double alpha = residual * residual / dot(n, p, c);
#pragma omp parallel for schedule(static)
            for (i64 i = 0; i < n; i++) {
                u[i] += alpha * p[i];
                r[i] -= alpha * c[i];
            }
            prev = residual;
            residual = sqrt(dot(n, r, r));
            it++;
        }
        total += it;
    }
    *iters_total = total;
    free(p); free(r); free(c); free(rhs);
    return 0;
}
