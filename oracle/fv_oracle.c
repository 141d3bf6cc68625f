/*
 * fv_oracle.c — CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C restatement of the arithmetic on FiniteVolume.jl's hot path
 * (grid -> Dirichlet elimination -> COO -> CSC assembly -> implicit step
 * linear algebra).  Single thread, 1-based int64 indices at the interface,
 * exactly as the Julia reference sees them.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / reported CPU baseline.  The
 * product (libfvhip.so + finitevolume.jl_amd/) never links, imports or calls
 * anything in oracle/.
 *
 * Parity status: the reference is Julia and no Julia runtime exists in the
 * build container, so the reference itself cannot be run.
 * This restatement is pinned by the reference's own known-answer tests
 * (test/runtests.jl:4-16, test/ode.jl, test/onenodeadjoint.jl:17-44,
 * test/theis.jl:21-65) and by the documented contracts of the Julia stdlib
 * (SparseArrays.sparse, range).  Third-party arithmetic restated from the
 * published algorithm at the pinned version (Manifest.toml):
 *   - SparseArrays.sparse(I,J,V,m,n,+)  (Julia stdlib)      -> fvo_sparse
 *   - IterativeSolvers 0.8.1 cg!/cg                           -> fvo_cg
 *   - AlgebraicMultigrid 0.2.2: NOT restated (parity unpinned; the build
 *     replaces it with Jacobi, fvo_pcg_jacobi, as BASELINE.json asks).
 *
 * Build:  make -C oracle      (gcc -O2 -ffp-contract=off; Julia never fuses
 *                              a*b+c, so neither may this file)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define FVO_OK 0
#define FVO_ERR_ARG 1
#define FVO_ERR_SOURCE_AT_DIRICHLET 2 /* FiniteVolume.jl:25-27 */
#define FVO_ERR_INDEX 3
#define FVO_ERR_NOMEM 4

typedef int64_t i64;

/* ------------------------------------------------------------------ */
/* a1. regulargrid — src/grid.jl:56-110                                */
/* ------------------------------------------------------------------ */

/* Julia `range(a; stop=b, length=n)` (grid.jl:62-64).  For Float64 end points
 * Base builds a StepRangeLen in twice precision (base/twiceprecision.jl), so
 * element k is the correctly rounded value of a + k*(b-a)/(n-1).  Restated
 * here with binary128 arithmetic and one final rounding.  */
int fvo_linrange(double a, double b, i64 n, double *out)
{
    if (n < 2)
        return FVO_ERR_ARG; /* grid.jl:65 reads xs[2] */
    for (i64 k = 0; k < n; k++) {
        __float128 v = ((__float128)a * (__float128)(n - 1 - k) + (__float128)b * (__float128)k) /
                       (__float128)(n - 1);
        out[k] = (double)v;
    }
    out[0] = a;
    out[n - 1] = b;
    return FVO_OK;
}

int fvo_regulargrid_sizes(const i64 ns[3], i64 *N, i64 *F)
{
    if (ns[0] < 2 || ns[1] < 2 || ns[2] < 2)
        return FVO_ERR_ARG;
    *N = ns[0] * ns[1] * ns[2];
    /* grid.jl:69 */
    *F = 3 * (*N) - ns[0] * ns[1] - ns[0] * ns[2] - ns[1] * ns[2];
    return FVO_OK;
}

/* coords: 3 x N column-major (coords[3*c + d]); node1/node2: the Pair halves. */
int fvo_regulargrid(const double mins[3], const double maxs[3], const i64 ns[3], double *coords, i64 *node1,
                    i64 *node2, double *aol, double *volumes)
{
    i64 N, F;
    int rc = fvo_regulargrid_sizes(ns, &N, &F);
    if (rc)
        return rc;
    double *xs = malloc(sizeof(double) * (size_t)ns[0]);
    double *ys = malloc(sizeof(double) * (size_t)ns[1]);
    double *zs = malloc(sizeof(double) * (size_t)ns[2]);
    if (!xs || !ys || !zs)
        return FVO_ERR_NOMEM;
    fvo_linrange(mins[0], maxs[0], ns[0], xs);
    fvo_linrange(mins[1], maxs[1], ns[1], ys);
    fvo_linrange(mins[2], maxs[2], ns[2], zs);
    const double dx = xs[1] - xs[0]; /* grid.jl:65-67 */
    const double dy = ys[1] - ys[0];
    const double dz = zs[1] - zs[0];
    i64 j = 0, c = 0;
    for (i64 i1 = 1; i1 <= ns[0]; i1++) {
        double areadx = xs[1] - xs[0];
        if (i1 == 1 || i1 == ns[0])
            areadx *= 0.5;
        for (i64 i2 = 1; i2 <= ns[1]; i2++) {
            double aready = ys[1] - ys[0];
            if (i2 == 1 || i2 == ns[1])
                aready *= 0.5;
            for (i64 i3 = 1; i3 <= ns[2]; i3++) {
                double areadz = zs[1] - zs[0];
                if (i3 == 1 || i3 == ns[2])
                    areadz *= 0.5;
                /* linearindex, grid.jl:60; the loop visits cells in that order */
                const i64 li = i3 + ns[2] * (i2 - 1) + ns[2] * ns[1] * (i1 - 1);
                volumes[c++] = areadx * aready * areadz; /* grid.jl:87 */
                if (coords) {
                    coords[3 * (li - 1) + 0] = xs[i1 - 1];
                    coords[3 * (li - 1) + 1] = ys[i2 - 1];
                    coords[3 * (li - 1) + 2] = zs[i3 - 1];
                }
                if (i1 < ns[0]) { /* grid.jl:91-95 */
                    node1[j] = li;
                    node2[j] = li + ns[2] * ns[1];
                    aol[j] = aready * areadz / dx;
                    j++;
                }
                if (i2 < ns[1]) { /* grid.jl:96-100 */
                    node1[j] = li;
                    node2[j] = li + ns[2];
                    aol[j] = areadx * areadz / dy;
                    j++;
                }
                if (i3 < ns[2]) { /* grid.jl:101-105 */
                    node1[j] = li;
                    node2[j] = li + 1;
                    aol[j] = areadx * aready / dz;
                    j++;
                }
            }
        }
    }
    free(xs);
    free(ys);
    free(zs);
    return (j == F) ? FVO_OK : FVO_ERR_INDEX;
}

/* a2. nodehycos2neighborhycos — src/grid.jl:14-33.  nodehycos is the
 * (n3,n2,n1) column-major array, i.e. linear position == node index. */
int fvo_nodehycos2neighborhycos(i64 F, const i64 *node1, const i64 *node2, i64 N, const double *nodehycos,
                                int logtransform, double *out)
{
    for (i64 i = 0; i < F; i++) {
        if (node1[i] < 1 || node1[i] > N || node2[i] < 1 || node2[i] > N)
            return FVO_ERR_INDEX;
        const double k1 = nodehycos[node1[i] - 1], k2 = nodehycos[node2[i] - 1];
        out[i] = logtransform ? 0.5 * (k1 + k2) : sqrt(k1 * k2); /* grid.jl:27,29 */
    }
    return FVO_OK;
}

/* ------------------------------------------------------------------ */
/* a3/a4. free-node and Dirichlet maps — src/FiniteVolume.jl:20-44     */
/* ------------------------------------------------------------------ */

int fvo_getfreenodes(i64 n, i64 ndir, const i64 *dirichletnodes, uint8_t *freenode, i64 *nodei2freenodei,
                     i64 *nfree)
{
    for (i64 i = 0; i < n; i++)
        freenode[i] = 1;
    for (i64 i = 0; i < ndir; i++) {
        if (dirichletnodes[i] < 1 || dirichletnodes[i] > n)
            return FVO_ERR_INDEX;
        freenode[dirichletnodes[i] - 1] = 0;
    }
    i64 j = 1;
    for (i64 i = 0; i < n; i++) {
        if (freenode[i])
            nodei2freenodei[i] = j++;
        else
            nodei2freenodei[i] = -1;
    }
    *nfree = j - 1;
    return FVO_OK;
}

int fvo_getnodei2dirichleti(i64 n, const double *sources, i64 ndir, const i64 *dirichletnodes,
                            i64 *nodei2dirichleti, i64 *badnode)
{
    for (i64 i = 0; i < n; i++)
        nodei2dirichleti[i] = -1;
    for (i64 i = 0; i < ndir; i++) {
        const i64 node = dirichletnodes[i];
        if (node < 1 || node > n)
            return FVO_ERR_INDEX;
        nodei2dirichleti[node - 1] = i + 1;
        if (sources[node - 1] != 0) { /* FiniteVolume.jl:25-27 */
            if (badnode)
                *badnode = node;
            return FVO_ERR_SOURCE_AT_DIRICHLET;
        }
    }
    return FVO_OK;
}

/* ------------------------------------------------------------------ */
/* SparseArrays.sparse(I, J, V, m, n, +) — Julia stdlib, called at     */
/* FiniteVolume.jl:107.  Counting sort into an unsorted-row CSR, sweep  */
/* combining repeats in INPUT order (left fold), transpose to CSC with  */
/* ascending row indices per column; explicit zeros are kept.           */
/* Output arrays are 1-based.  rowval/nzval need capacity coolen.       */
/* ------------------------------------------------------------------ */
int fvo_sparse(i64 coolen, const i64 *I, const i64 *J, const double *V, i64 m, i64 n, i64 *colptr, i64 *rowval,
               double *nzval, i64 *nnz_out)
{
    i64 *csrrowptr = calloc((size_t)m + 2, sizeof(i64));
    i64 *csrcolval = malloc(sizeof(i64) * (size_t)(coolen ? coolen : 1));
    double *csrnzval = malloc(sizeof(double) * (size_t)(coolen ? coolen : 1));
    i64 *klasttouch = calloc((size_t)n + 1, sizeof(i64));
    if (!csrrowptr || !csrcolval || !csrnzval || !klasttouch)
        return FVO_ERR_NOMEM;
    /* all index variables below hold Julia's 1-based values */
    for (i64 k = 0; k < coolen; k++) {
        if (I[k] < 1 || I[k] > m || J[k] < 1 || J[k] > n) {
            free(csrrowptr); free(csrcolval); free(csrnzval); free(klasttouch);
            return FVO_ERR_INDEX;
        }
        csrrowptr[I[k] + 1 - 1]++; /* csrrowptr[Ik+1] += 1 */
    }
    i64 countsum = 1;
    csrrowptr[0] = 1;
    for (i64 i = 2; i <= m + 1; i++) {
        i64 overwritten = csrrowptr[i - 1];
        csrrowptr[i - 1] = countsum;
        countsum += overwritten;
    }
    for (i64 k = 0; k < coolen; k++) {
        i64 csrk = csrrowptr[I[k] + 1 - 1];
        csrrowptr[I[k] + 1 - 1] = csrk + 1;
        csrcolval[csrk - 1] = J[k];
        csrnzval[csrk - 1] = V[k];
    }
    for (i64 j = 0; j <= n; j++)
        colptr[j] = 0;
    i64 writek = 1, newcsrrowptri = 1, origcsrrowptri = 1;
    i64 origcsrrowptrip1 = (m >= 1) ? csrrowptr[1] : 1;
    for (i64 i = 1; i <= m; i++) {
        for (i64 readk = origcsrrowptri; readk <= origcsrrowptrip1 - 1; readk++) {
            const i64 j = csrcolval[readk - 1];
            if (klasttouch[j] < newcsrrowptri) {
                klasttouch[j] = writek;
                if (writek != readk) {
                    csrcolval[writek - 1] = j;
                    csrnzval[writek - 1] = csrnzval[readk - 1];
                }
                writek++;
                colptr[j + 1 - 1]++;
            } else {
                const i64 klt = klasttouch[j];
                csrnzval[klt - 1] = csrnzval[klt - 1] + csrnzval[readk - 1]; /* combine(old, new) */
            }
        }
        newcsrrowptri = writek;
        origcsrrowptri = origcsrrowptrip1;
        if (origcsrrowptrip1 != writek)
            csrrowptr[i + 1 - 1] = writek;
        if (i < m)
            origcsrrowptrip1 = csrrowptr[i + 2 - 1];
    }
    countsum = 1;
    colptr[0] = 1;
    for (i64 j = 2; j <= n + 1; j++) {
        i64 overwritten = colptr[j - 1];
        colptr[j - 1] = countsum;
        countsum += overwritten;
    }
    const i64 cscnnz = countsum - 1;
    for (i64 i = 1; i <= m; i++) {
        for (i64 csrk = csrrowptr[i - 1]; csrk <= csrrowptr[i + 1 - 1] - 1; csrk++) {
            const i64 j = csrcolval[csrk - 1];
            const double x = csrnzval[csrk - 1];
            const i64 csck = colptr[j + 1 - 1];
            colptr[j + 1 - 1] = csck + 1;
            rowval[csck - 1] = i;
            nzval[csck - 1] = x;
        }
    }
    /* the write cursors left colptr[j+1] == start of column j+1: already final */
    *nnz_out = cscnnz;
    free(csrrowptr); free(csrcolval); free(csrnzval); free(klasttouch);
    return FVO_OK;
}

/* ------------------------------------------------------------------ */
/* a5. assembleA — src/FiniteVolume.jl:75-108                           */
/* metaindex: NULL (identity, the default `i->i`) or F 1-based indices. */
/* Returns CSC (1-based).  rowval/nzval capacity: 4*F.                  */
/* ------------------------------------------------------------------ */
static inline double fvo_face_c(const double *K, const i64 *metaindex, const double *aol, i64 i, int logt)
{
    const double k = K[(metaindex ? metaindex[i] : i + 1) - 1];
    return logt ? exp(k) * aol[i] : k * aol[i]; /* FiniteVolume.jl:83 / :96 */
}

int fvo_assembleA(i64 N, i64 F, const i64 *node1, const i64 *node2, const double *aol, i64 nK, const double *K,
                  const i64 *metaindex, int logtransform, i64 ndir, const i64 *dirichletnodes, i64 *colptr,
                  i64 *rowval, double *nzval, i64 *nfree_out, i64 *nnz_out)
{
    uint8_t *freenode = malloc((size_t)N + 1);
    i64 *n2f = malloc(sizeof(i64) * ((size_t)N + 1));
    i64 *I = malloc(sizeof(i64) * (size_t)(4 * F + 1));
    i64 *J = malloc(sizeof(i64) * (size_t)(4 * F + 1));
    double *V = malloc(sizeof(double) * (size_t)(4 * F + 1));
    if (!freenode || !n2f || !I || !J || !V)
        return FVO_ERR_NOMEM;
    i64 nfree;
    int rc = fvo_getfreenodes(N, ndir, dirichletnodes, freenode, n2f, &nfree);
    i64 len = 0;
    for (i64 i = 0; i < F && !rc; i++) {
        const i64 a = node1[i], b = node2[i];
        if (a < 1 || a > N || b < 1 || b > N) { rc = FVO_ERR_INDEX; break; }
        if (metaindex && (metaindex[i] < 1 || metaindex[i] > nK)) { rc = FVO_ERR_INDEX; break; }
        if (!metaindex && i >= nK) { rc = FVO_ERR_INDEX; break; }
        if (freenode[a - 1] && freenode[b - 1]) { /* :95-99 */
            const double c = fvo_face_c(K, metaindex, aol, i, logtransform);
            I[len] = n2f[a - 1]; J[len] = n2f[a - 1]; V[len++] = c;
            I[len] = n2f[a - 1]; J[len] = n2f[b - 1]; V[len++] = -c;
            I[len] = n2f[b - 1]; J[len] = n2f[b - 1]; V[len++] = c;
            I[len] = n2f[b - 1]; J[len] = n2f[a - 1]; V[len++] = -c;
        } else if (freenode[a - 1]) { /* :100-101 */
            const double c = fvo_face_c(K, metaindex, aol, i, logtransform);
            I[len] = n2f[a - 1]; J[len] = n2f[a - 1]; V[len++] = c;
        } else if (freenode[b - 1]) { /* :102-103 */
            const double c = fvo_face_c(K, metaindex, aol, i, logtransform);
            I[len] = n2f[b - 1]; J[len] = n2f[b - 1]; V[len++] = c;
        }
    }
    if (!rc)
        rc = fvo_sparse(len, I, J, V, nfree, nfree, colptr, rowval, nzval, nnz_out);
    *nfree_out = nfree;
    free(freenode); free(n2f); free(I); free(J); free(V);
    return rc;
}

/* a6. assembleb — src/FiniteVolume.jl:110-139 */
int fvo_assembleb(i64 N, i64 F, const i64 *node1, const i64 *node2, const double *aol, i64 nK, const double *K,
                  const i64 *metaindex, int logtransform, const double *sources, i64 ndir,
                  const i64 *dirichletnodes, const double *dirichletheads, double *b, i64 *badnode)
{
    uint8_t *freenode = malloc((size_t)N + 1);
    i64 *n2f = malloc(sizeof(i64) * ((size_t)N + 1));
    i64 *n2d = malloc(sizeof(i64) * ((size_t)N + 1));
    if (!freenode || !n2f || !n2d)
        return FVO_ERR_NOMEM;
    i64 nfree;
    int rc = fvo_getnodei2dirichleti(N, sources, ndir, dirichletnodes, n2d, badnode);
    if (!rc)
        rc = fvo_getfreenodes(N, ndir, dirichletnodes, freenode, n2f, &nfree);
    if (!rc) {
        i64 j = 0;
        for (i64 i = 0; i < N; i++)
            if (freenode[i])
                b[j++] = sources[i];
        for (i64 i = 0; i < F; i++) {
            const i64 a = node1[i], c2 = node2[i];
            if (a < 1 || a > N || c2 < 1 || c2 > N) { rc = FVO_ERR_INDEX; break; }
            if (metaindex && (metaindex[i] < 1 || metaindex[i] > nK)) { rc = FVO_ERR_INDEX; break; }
            if (freenode[a - 1] && !freenode[c2 - 1]) /* :131-132 */
                b[n2f[a - 1] - 1] += fvo_face_c(K, metaindex, aol, i, logtransform) * dirichletheads[n2d[c2 - 1] - 1];
            else if (!freenode[a - 1] && freenode[c2 - 1]) /* :133-134 */
                b[n2f[c2 - 1] - 1] += fvo_face_c(K, metaindex, aol, i, logtransform) * dirichletheads[n2d[a - 1] - 1];
        }
    }
    free(freenode); free(n2f); free(n2d);
    return rc;
}

/* a7. freenodes2nodes — src/FiniteVolume.jl:141-155 */
int fvo_freenodes2nodes(i64 N, const double *result, const double *sources, i64 ndir, const i64 *dirichletnodes,
                        const double *dirichletheads, double *head, i64 *badnode)
{
    uint8_t *freenode = malloc((size_t)N + 1);
    i64 *n2f = malloc(sizeof(i64) * ((size_t)N + 1));
    i64 *n2d = malloc(sizeof(i64) * ((size_t)N + 1));
    if (!freenode || !n2f || !n2d)
        return FVO_ERR_NOMEM;
    i64 nfree;
    int rc = fvo_getnodei2dirichleti(N, sources, ndir, dirichletnodes, n2d, badnode);
    if (!rc)
        rc = fvo_getfreenodes(N, ndir, dirichletnodes, freenode, n2f, &nfree);
    if (!rc) {
        i64 sofar = 0;
        for (i64 i = 0; i < N; i++)
            head[i] = freenode[i] ? result[sofar++] : dirichletheads[n2d[i] - 1];
    }
    free(freenode); free(n2f); free(n2d);
    return rc;
}

/* ------------------------------------------------------------------ */
/* a9/a10. scalebyvolume! / diagonalupdate! — src/transient.jl:7-48    */
/* ------------------------------------------------------------------ */

/* vols: Ss*volumes (length N); f2n: free index -> node index, 1-based (n) */
void fvo_scalebyvolume_b(i64 n, double *b, const double *vols, const i64 *f2n)
{
    for (i64 i = 0; i < n; i++)
        b[i] /= vols[f2n[i] - 1]; /* transient.jl:9 */
}

void fvo_scalebyvolume_A(i64 n, const i64 *colptr, const i64 *rowval, double *nzval, const double *vols,
                         const i64 *f2n)
{
    for (i64 i = 1; i <= n; i++)
        for (i64 j = colptr[i - 1]; j <= colptr[i] - 1; j++)
            nzval[j - 1] /= vols[f2n[rowval[j - 1] - 1] - 1]; /* transient.jl:19 — row scaling */
}

void fvo_diagonalupdate(i64 n, const i64 *colptr, const i64 *rowval, double *nzval, double increment)
{
    for (i64 i = 1; i <= n; i++)
        for (i64 j = colptr[i - 1]; j <= colptr[i] - 1; j++)
            if (rowval[j - 1] == i)
                nzval[j - 1] += increment; /* transient.jl:43-45 */
}

/* ------------------------------------------------------------------ */
/* Linear algebra used by the solvers                                   */
/* ------------------------------------------------------------------ */

/* y = A*x for SparseMatrixCSC, column-oriented accumulation as Julia's mul! */
void fvo_spmv_csc(i64 m, i64 n, const i64 *colptr, const i64 *rowval, const double *nzval, const double *x,
                  double *y)
{
    for (i64 i = 0; i < m; i++)
        y[i] = 0.0;
    for (i64 col = 1; col <= n; col++) {
        const double xc = x[col - 1];
        for (i64 k = colptr[col - 1]; k <= colptr[col] - 1; k++)
            y[rowval[k - 1] - 1] += nzval[k - 1] * xc;
    }
}

static double fvo_dot(i64 n, const double *a, const double *b)
{
    double s = 0.0;
    for (i64 i = 0; i < n; i++)
        s += a[i] * b[i];
    return s;
}

double fvo_norm2(i64 n, const double *a) { return sqrt(fvo_dot(n, a, a)); }

double fvo_norm2_diff(i64 n, const double *a, const double *b)
{
    double s = 0.0;
    for (i64 i = 0; i < n; i++) {
        const double d = a[i] - b[i];
        s += d * d;
    }
    return sqrt(s);
}

/* IterativeSolvers 0.8.1 `cg!(x, A, b; tol, maxiter)` without preconditioner
 * (the CGIterable), called at transient.jl:52.  Stops when
 * ||r||_2 <= tol*||b||_2 or after maxiter iterations.  initially_zero
 * mirrors `cg(A, b)` (x0 = 0, no initial mat-vec).  resnorm (capacity
 * maxiter+1, may be NULL) receives the residual norm after each iteration. */
int fvo_cg(i64 n, const i64 *colptr, const i64 *rowval, const double *nzval, const double *b, double *x,
           double tol, i64 maxiter, int initially_zero, i64 *iters_out, int *converged_out, double *resnorm)
{
    double *u = calloc((size_t)n + 1, sizeof(double));
    double *r = malloc(sizeof(double) * ((size_t)n + 1));
    double *c = malloc(sizeof(double) * ((size_t)n + 1));
    if (!u || !r || !c)
        return FVO_ERR_NOMEM;
    memcpy(r, b, sizeof(double) * (size_t)n);
    double residual, reltol;
    if (initially_zero) {
        residual = fvo_norm2(n, b);
        reltol = residual * tol;
    } else {
        fvo_spmv_csc(n, n, colptr, rowval, nzval, x, c);
        for (i64 i = 0; i < n; i++)
            r[i] -= c[i];
        residual = fvo_norm2(n, r);
        reltol = fvo_norm2(n, b) * tol;
    }
    double prev_residual = 1.0;
    i64 it = 0;
    while (!(it >= maxiter || residual <= reltol)) {
        const double beta = residual * residual / (prev_residual * prev_residual);
        for (i64 i = 0; i < n; i++)
            u[i] = r[i] + beta * u[i];
        fvo_spmv_csc(n, n, colptr, rowval, nzval, u, c);
        const double alpha = residual * residual / fvo_dot(n, u, c);
        for (i64 i = 0; i < n; i++)
            x[i] += alpha * u[i];
        for (i64 i = 0; i < n; i++)
            r[i] -= alpha * c[i];
        prev_residual = residual;
        residual = fvo_norm2(n, r);
        if (resnorm)
            resnorm[it] = residual;
        it++;
    }
    *iters_out = it;
    *converged_out = residual <= reltol;
    free(u); free(r); free(c);
    return FVO_OK;
}

/* Jacobi-preconditioned CG on  (A + diag(shift)) x = b  — the algorithm the
 * HIP build runs (IterativeSolvers' PCGIterable with Pl = diag).  A is the
 * symmetric CSC/CSR matrix, shift may be NULL.  Starts from x. */
int fvo_pcg_jacobi(i64 n, const i64 *colptr, const i64 *rowval, const double *nzval, const double *shift,
                   const double *b, double *x, double tol, i64 maxiter, i64 *iters_out, int *converged_out,
                   double *final_relres)
{
    double *u = calloc((size_t)n + 1, sizeof(double));
    double *r = malloc(sizeof(double) * ((size_t)n + 1));
    double *c = malloc(sizeof(double) * ((size_t)n + 1));
    double *minv = malloc(sizeof(double) * ((size_t)n + 1));
    if (!u || !r || !c || !minv)
        return FVO_ERR_NOMEM;
    for (i64 i = 1; i <= n; i++) {
        double d = shift ? shift[i - 1] : 0.0;
        for (i64 k = colptr[i - 1]; k <= colptr[i] - 1; k++)
            if (rowval[k - 1] == i)
                d += nzval[k - 1];
        minv[i - 1] = 1.0 / d;
    }
    fvo_spmv_csc(n, n, colptr, rowval, nzval, x, c);
    for (i64 i = 0; i < n; i++)
        r[i] = b[i] - (c[i] + (shift ? shift[i] * x[i] : 0.0));
    double residual = fvo_norm2(n, r);
    const double bnorm = fvo_norm2(n, b);
    const double reltol = bnorm * tol;
    double rho = 1.0;
    i64 it = 0;
    while (!(it >= maxiter || residual <= reltol)) {
        for (i64 i = 0; i < n; i++)
            c[i] = minv[i] * r[i];
        const double rho_prev = rho;
        rho = fvo_dot(n, c, r);
        const double beta = rho / rho_prev;
        for (i64 i = 0; i < n; i++)
            u[i] = c[i] + beta * u[i];
        fvo_spmv_csc(n, n, colptr, rowval, nzval, u, c);
        if (shift)
            for (i64 i = 0; i < n; i++)
                c[i] += shift[i] * u[i];
        const double alpha = rho / fvo_dot(n, u, c);
        for (i64 i = 0; i < n; i++)
            x[i] += alpha * u[i];
        for (i64 i = 0; i < n; i++)
            r[i] -= alpha * c[i];
        residual = fvo_norm2(n, r);
        it++;
    }
    *iters_out = it;
    *converged_out = residual <= reltol;
    if (final_relres)
        *final_relres = bnorm > 0 ? residual / bnorm : residual;
    free(u); free(r); free(c); free(minv);
    return FVO_OK;
}
