// CSR SpMV and the fused Jacobi-preconditioned conjugate gradient.
//
// Replaces IterativeSolvers.cg!/cg + AlgebraicMultigrid at
// /root/reference/src/FiniteVolume.jl:160-161 and src/transient.jl:50-58.
// The operator is (A + sigma*D): A the assembled symmetric CSR, D = Ss*volumes on
// the free cells, sigma = 1/dt (0 for the steady solve), i.e. the SPD form of the
// reference's (I/dt + D^-1 A).
//
// Everything here is HBM-bandwidth bound (0.13 flop/B), so no MFMA: the kernels
// are built for coalesced streaming of vals/colind, L2-served gathers of x and
// wave64 shuffle + LDS reductions.  One PCG iteration is three launches:
//   K1 spmv_dot   q = (A + sigma D) p ; partial p.q per block      12 nnz + 28 n  B
//   K2 update     alpha = rz/pq ; x += alpha p ; r -= alpha q ;
//                 partial r.M^-1 r and r.r per block                56 n B
//   K3 pupdate    beta = rz'/rz ; p = M^-1 r + beta p ; scalars     32 n B
// Scalars never visit the host inside the loop: every block re-reduces the <= 2048
// per-block partials of the previous kernel in a fixed order (deterministic, no
// atomics), and a device-side `done` flag turns surplus launches into no-ops, so
// the host polls convergence only once per chunk of iterations.
#include "fv_internal.h"

// ------------------------------------------------------------------ reductions
__device__ inline double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        v += __shfl_xor(v, off, 64);
    return v;
}

// all threads of the 256-thread block get the sum; smem: 4 doubles
__device__ inline double block_sum(double v, double *smem)
{
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads(); // smem may still be read by a previous call
    if (lane == 0)
        smem[wave] = v;
    __syncthreads();
    return (smem[0] + smem[1]) + (smem[2] + smem[3]);
}

__device__ inline double reduce_partials(const double *__restrict__ part, int count, double *smem)
{
    double v = 0.0;
    for (int i = threadIdx.x; i < count; i += FV_BLOCK)
        v += part[i];
    return block_sum(v, smem);
}

// ------------------------------------------------------------------ SpMV
// LPR lanes cooperate on one row (8 for the 7-point stencil: a wave64 covers 8
// consecutive rows, whose ~56 stored entries are contiguous in vals/colind, so a
// wave load instruction is one coalesced 448-byte burst).  Each thread keeps U
// independent rows in flight to cover HBM latency.  Blocks sweep the matrix in
// passes of G*RPB consecutive rows so that concurrently running blocks work on
// neighbouring rows (x re-reads stay in L2 / Infinity Cache); within a pass the
// eight XCDs get contiguous sub-windows (blockIdx & 7 labels the XCD share).
template <int LPR, int U, bool DOT>
__global__ __launch_bounds__(FV_BLOCK) void spmv_kernel(int64_t n, const int32_t *__restrict__ rowptr,
                                                         const int32_t *__restrict__ colind, const double *__restrict__ vals,
                                                         const double *__restrict__ x, double *__restrict__ y,
                                                         const double *__restrict__ shift, double sigma,
                                                         double *__restrict__ partials, const PcgScalars *__restrict__ scal)
{
    __shared__ double smem[4];
    if (scal && scal->done)
        return;
    constexpr int ROWS_SUB = FV_BLOCK / LPR; // rows per unrolled sub-pass
    constexpr int RPB = ROWS_SUB * U;        // rows per block per pass
    const int G = gridDim.x;
    const int slot = (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3);
    const int sub = threadIdx.x % LPR;
    const int rib = threadIdx.x / LPR;
    double dacc = 0.0;
    for (int64_t base = (int64_t)slot * RPB; base < n; base += (int64_t)G * RPB) {
        int32_t k[U], e[U];
        double sum[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int64_t row = base + u * ROWS_SUB + rib;
            if (row < n) {
                k[u] = rowptr[row] + sub;
                e[u] = rowptr[row + 1];
            } else {
                k[u] = 0;
                e[u] = 0;
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            double s = 0.0;
            if (k[u] < e[u])
                s = vals[k[u]] * x[colind[k[u]]];
            sum[u] = s;
        }
#pragma unroll
        for (int u = 0; u < U; u++) // rows longer than LPR
            for (int32_t kk = k[u] + LPR; kk < e[u]; kk += LPR)
                sum[u] += vals[kk] * x[colind[kk]];
#pragma unroll
        for (int u = 0; u < U; u++) {
            double s = sum[u];
#pragma unroll
            for (int off = LPR / 2; off > 0; off >>= 1)
                s += __shfl_xor(s, off, LPR);
            const int64_t row = base + u * ROWS_SUB + rib;
            if (sub == 0 && row < n) {
                const double xr = (shift || DOT) ? x[row] : 0.0;
                if (shift)
                    s += sigma * shift[row] * xr;
                y[row] = s;
                if (DOT)
                    dacc += xr * s;
            }
        }
    }
    if (DOT) {
        const double t = block_sum(dacc, smem);
        if (threadIdx.x == 0)
            partials[blockIdx.x] = t;
    }
}

// lanes per row from the mean row length: 8 covers the 7-point stencil in one pass
static int spmv_lpr(const fv_problem *p)
{
    const double avg = p->n > 0 ? (double)p->nnz / (double)p->n : 0.0;
    return avg >= 11.0 ? 16 : (avg <= 4.0 ? 4 : 8);
}

int fv_spmv_grid(fv_problem *p)
{
    // a multiple of 8 (XCD shares), at most one partial per block
    const int lpr = spmv_lpr(p);
    const int rpb = (FV_BLOCK / lpr) * 2;
    int64_t g = (p->n + rpb - 1) / rpb;
    g = ((g + 7) / 8) * 8;
    if (g > FV_MAX_PARTIALS)
        g = FV_MAX_PARTIALS;
    if (g < 8)
        g = 8;
    return (int)g;
}

// y = (A + sigma*D) x ; partials != NULL also emits per-block partial sums of x.y.
// use_done: honour the PCG early-exit flag.
static int spmv_launch_impl(fv_problem *p, const double *x, double *y, double sigma, double *partials, bool use_done)
{
    fv_ctx *ctx = p->ctx;
    const int G = fv_spmv_grid(p);
    const double *shift = (sigma != 0.0) ? p->D.p : nullptr;
    const PcgScalars *scal = use_done ? p->scal.p : nullptr;
    const int lpr = spmv_lpr(p);
#define FV_SPMV_CASE(L)                                                                                                 \
    if (partials)                                                                                                       \
        hipLaunchKernelGGL((spmv_kernel<L, 2, true>), dim3(G), dim3(FV_BLOCK), 0, ctx->stream, p->n, p->rowptr.p,       \
                           p->colind.p, p->vals.p, x, y, shift, sigma, partials, scal);                                 \
    else                                                                                                                \
        hipLaunchKernelGGL((spmv_kernel<L, 2, false>), dim3(G), dim3(FV_BLOCK), 0, ctx->stream, p->n, p->rowptr.p,      \
                           p->colind.p, p->vals.p, x, y, shift, sigma, partials, scal);
    if (lpr == 4) {
        FV_SPMV_CASE(4)
    } else if (lpr == 8) {
        FV_SPMV_CASE(8)
    } else {
        FV_SPMV_CASE(16)
    }
#undef FV_SPMV_CASE
    FV_LAUNCH_CHECK(ctx);
    return FV_OK;
}

int fv_spmv_launch(fv_problem *p, const double *x, double *y, double sigma, double *partials_or_null)
{
    return spmv_launch_impl(p, x, y, sigma, partials_or_null, false);
}

// ------------------------------------------------------------------ PCG vector kernels
__device__ inline int64_t vec_stride() { return (int64_t)gridDim.x * FV_BLOCK; }

// r = rhs - q (q = (A + sigma D) x0, or absent when x0 = 0); M^-1 = 1/(diag(A) + sigma D);
// p = M^-1 r; per-block partials of r.M^-1 r, r.r, rhs.rhs
__global__ __launch_bounds__(FV_BLOCK) void pcg_init_kernel(int64_t n, const double *__restrict__ rhs, const double *__restrict__ q,
                                                             const double *__restrict__ diagA, const double *__restrict__ D,
                                                             double sigma, double *__restrict__ r, double *__restrict__ pv,
                                                             double *__restrict__ minv, double *__restrict__ part_rz,
                                                             double *__restrict__ part_rr, double *__restrict__ part_bb)
{
    __shared__ double smem[4];
    double arz = 0.0, arr = 0.0, abb = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x; i < n; i += vec_stride()) {
        const double bi = rhs[i];
        const double ri = q ? bi - q[i] : bi;
        const double d = D ? diagA[i] + sigma * D[i] : diagA[i];
        const double mi = 1.0 / d;
        const double zi = mi * ri;
        r[i] = ri;
        minv[i] = mi;
        pv[i] = zi;
        arz += ri * zi;
        arr += ri * ri;
        abb += bi * bi;
    }
    const double t0 = block_sum(arz, smem);
    const double t1 = block_sum(arr, smem);
    const double t2 = block_sum(abb, smem);
    if (threadIdx.x == 0) {
        part_rz[blockIdx.x] = t0;
        part_rr[blockIdx.x] = t1;
        part_bb[blockIdx.x] = t2;
    }
}

__global__ __launch_bounds__(FV_BLOCK) void pcg_init_finalize_kernel(const double *__restrict__ part_rz,
                                                                      const double *__restrict__ part_rr,
                                                                      const double *__restrict__ part_bb, int nparts, double rtol,
                                                                      PcgScalars *__restrict__ scal)
{
    __shared__ double smem[4];
    const double rz = reduce_partials(part_rz, nparts, smem);
    const double rr = reduce_partials(part_rr, nparts, smem);
    const double bb = reduce_partials(part_bb, nparts, smem);
    if (threadIdx.x == 0) {
        scal->rz[0] = rz;
        scal->rz[1] = 0.0;
        scal->rr = rr;
        scal->bnorm2 = bb;
        scal->tol2 = rtol * rtol * bb; // stop when ||r|| <= rtol*||b||  (IterativeSolvers' reltol)
        scal->pq = 0.0;
        scal->iters = 0;
        scal->done = (rr <= scal->tol2) ? 1 : 0;
    }
}

// K2
__global__ __launch_bounds__(FV_BLOCK) void pcg_update_kernel(int64_t n, int it, double *__restrict__ x, double *__restrict__ r,
                                                               const double *__restrict__ pv, const double *__restrict__ q,
                                                               const double *__restrict__ minv, const double *__restrict__ part_pq,
                                                               int npq, PcgScalars *__restrict__ scal, double *__restrict__ part_rz,
                                                               double *__restrict__ part_rr)
{
    __shared__ double smem[4];
    if (scal->done)
        return;
    const double pq = reduce_partials(part_pq, npq, smem);
    if (!(pq > 0.0)) { // breakdown: not positive definite, or NaN
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            scal->pq = pq;
            scal->done = 2;
        }
        return;
    }
    const double alpha = scal->rz[it & 1] / pq;
    double arz = 0.0, arr = 0.0;
    const int64_t n2 = n >> 1;
    double2 *x2 = reinterpret_cast<double2 *>(x);
    double2 *r2 = reinterpret_cast<double2 *>(r);
    const double2 *p2 = reinterpret_cast<const double2 *>(pv);
    const double2 *q2 = reinterpret_cast<const double2 *>(q);
    const double2 *m2 = reinterpret_cast<const double2 *>(minv);
    for (int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x; i < n2; i += vec_stride()) {
        double2 xv = x2[i], rv = r2[i];
        const double2 pvv = p2[i], qv = q2[i], mv = m2[i];
        xv.x += alpha * pvv.x;
        xv.y += alpha * pvv.y;
        rv.x -= alpha * qv.x;
        rv.y -= alpha * qv.y;
        x2[i] = xv;
        r2[i] = rv;
        arz += rv.x * (mv.x * rv.x) + rv.y * (mv.y * rv.y);
        arr += rv.x * rv.x + rv.y * rv.y;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const int64_t i = n - 1;
        const double xi = x[i] + alpha * pv[i];
        const double ri = r[i] - alpha * q[i];
        x[i] = xi;
        r[i] = ri;
        arz += ri * (minv[i] * ri);
        arr += ri * ri;
    }
    const double t0 = block_sum(arz, smem);
    const double t1 = block_sum(arr, smem);
    if (threadIdx.x == 0) {
        part_rz[blockIdx.x] = t0;
        part_rr[blockIdx.x] = t1;
        if (blockIdx.x == 0)
            scal->pq = pq;
    }
}

// K3
__global__ __launch_bounds__(FV_BLOCK) void pcg_pupdate_kernel(int64_t n, int it, const double *__restrict__ r,
                                                                const double *__restrict__ minv, double *__restrict__ pv,
                                                                const double *__restrict__ part_rz, const double *__restrict__ part_rr,
                                                                int nparts, PcgScalars *__restrict__ scal, double *__restrict__ hist,
                                                                int64_t hist_cap)
{
    __shared__ double smem[4];
    if (scal->done)
        return;
    const double rzn = reduce_partials(part_rz, nparts, smem);
    const double rrn = reduce_partials(part_rr, nparts, smem);
    const double beta = rzn / scal->rz[it & 1];
    const int64_t n2 = n >> 1;
    const double2 *r2 = reinterpret_cast<const double2 *>(r);
    const double2 *m2 = reinterpret_cast<const double2 *>(minv);
    double2 *p2 = reinterpret_cast<double2 *>(pv);
    for (int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x; i < n2; i += vec_stride()) {
        const double2 rv = r2[i], mv = m2[i];
        double2 pvv = p2[i];
        pvv.x = mv.x * rv.x + beta * pvv.x;
        pvv.y = mv.y * rv.y + beta * pvv.y;
        p2[i] = pvv;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const int64_t i = n - 1;
        pv[i] = minv[i] * r[i] + beta * pv[i];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        scal->rz[(it + 1) & 1] = rzn;
        scal->rr = rrn;
        scal->iters = it + 1;
        if (hist && it < hist_cap)
            hist[it] = sqrt(rrn);
        if (rrn <= scal->tol2)
            scal->done = 1;
    }
}

static int vec_grid(int64_t n)
{
    int64_t g = (n / 2 + FV_BLOCK - 1) / FV_BLOCK;
    if (g < 1)
        g = 1;
    if (g > FV_MAX_PARTIALS)
        g = FV_MAX_PARTIALS;
    return (int)g;
}

int fv_pcg_prepare(fv_problem *p)
{
    fv_ctx *ctx = p->ctx;
    if (p->r.p)
        return FV_OK;
    const size_t n = (size_t)p->n + 2; // +2: the double2 tail never reads past the allocation
    FV_TRY(p->r.alloc(ctx, n));
    FV_TRY(p->pvec.alloc(ctx, n));
    FV_TRY(p->q.alloc(ctx, n));
    FV_TRY(p->minv.alloc(ctx, n));
    FV_TRY(p->rhs.alloc(ctx, n));
    FV_TRY(p->tmp.alloc(ctx, n));
    FV_TRY(p->part_pq.alloc(ctx, FV_MAX_PARTIALS));
    FV_TRY(p->part_rz.alloc(ctx, FV_MAX_PARTIALS));
    FV_TRY(p->part_rr.alloc(ctx, FV_MAX_PARTIALS));
    FV_TRY(p->part_bb.alloc(ctx, FV_MAX_PARTIALS));
    FV_TRY(p->scal.alloc(ctx, 1));
    FV_TRY(p->scal.zero(ctx));
    FV_TRY(p->pvec.zero(ctx));
    FV_TRY(p->q.zero(ctx));
    return FV_OK;
}

int fv_pcg_solve(fv_problem *p, double *x, const double *rhs, double sigma, bool x0_zero, double rtol, int64_t maxiter,
                 fv_solve_info *info, bool time_it)
{
    fv_ctx *ctx = p->ctx;
    FV_TRY(fv_pcg_prepare(p));
    if (maxiter < 0)
        maxiter = 0;
    if (maxiter > 0x7ffffff0LL)
        maxiter = 0x7ffffff0LL;
    if (sigma != 0.0 && !p->D.p) {
        fv_set_error(ctx, "fv_pcg_solve: shifted operator requested before fv_transient_begin");
        return FV_ERR_STATE;
    }
    const int64_t n = p->n;
    const int Gs = fv_spmv_grid(p);
    const int Gv = vec_grid(n);
    if (time_it)
        FV_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    const double *Dp = (sigma != 0.0) ? p->D.p : nullptr;
    if (x0_zero) {
        FV_HIP(ctx, hipMemsetAsync(x, 0, (size_t)n * sizeof(double), ctx->stream));
        hipLaunchKernelGGL(pcg_init_kernel, dim3(Gv), dim3(FV_BLOCK), 0, ctx->stream, n, rhs, (const double *)nullptr, p->diagA.p, Dp,
                           sigma, p->r.p, p->pvec.p, p->minv.p, p->part_rz.p, p->part_rr.p, p->part_bb.p);
    } else {
        FV_TRY(spmv_launch_impl(p, x, p->q.p, sigma, nullptr, false));
        hipLaunchKernelGGL(pcg_init_kernel, dim3(Gv), dim3(FV_BLOCK), 0, ctx->stream, n, rhs, (const double *)p->q.p, p->diagA.p, Dp,
                           sigma, p->r.p, p->pvec.p, p->minv.p, p->part_rz.p, p->part_rr.p, p->part_bb.p);
    }
    FV_LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(pcg_init_finalize_kernel, dim3(1), dim3(FV_BLOCK), 0, ctx->stream, p->part_rz.p, p->part_rr.p, p->part_bb.p,
                       Gv, rtol, p->scal.p);
    FV_LAUNCH_CHECK(ctx);
    PcgScalars *hs = reinterpret_cast<PcgScalars *>(ctx->pinned);
    int64_t it = 0;
    int64_t chunk = 4;
    bool polled = false;
    while (it < maxiter) {
        const int64_t m = (maxiter - it < chunk) ? (maxiter - it) : chunk;
        for (int64_t k = 0; k < m; k++) {
            const int iter = (int)(it + k);
            FV_TRY(spmv_launch_impl(p, p->pvec.p, p->q.p, sigma, p->part_pq.p, true));
            hipLaunchKernelGGL(pcg_update_kernel, dim3(Gv), dim3(FV_BLOCK), 0, ctx->stream, n, iter, x, p->r.p, p->pvec.p, p->q.p,
                               p->minv.p, p->part_pq.p, Gs, p->scal.p, p->part_rz.p, p->part_rr.p);
            hipLaunchKernelGGL(pcg_pupdate_kernel, dim3(Gv), dim3(FV_BLOCK), 0, ctx->stream, n, iter, p->r.p, p->minv.p, p->pvec.p,
                               p->part_rz.p, p->part_rr.p, Gv, p->scal.p, p->hist.p, p->hist_cap);
        }
        FV_LAUNCH_CHECK(ctx);
        it += m;
        FV_HIP(ctx, hipMemcpyAsync(hs, p->scal.p, sizeof(PcgScalars), hipMemcpyDeviceToHost, ctx->stream));
        FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        polled = true;
        if (hs->done)
            break;
        if (chunk < 32)
            chunk *= 2;
    }
    if (time_it)
        FV_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    if (!polled) {
        FV_HIP(ctx, hipMemcpyAsync(hs, p->scal.p, sizeof(PcgScalars), hipMemcpyDeviceToHost, ctx->stream));
        FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (info) {
        info->converged = hs->done == 1;
        info->iters = hs->iters;
        info->bnorm = sqrt(hs->bnorm2);
        info->relres = hs->bnorm2 > 0 ? sqrt(hs->rr / hs->bnorm2) : sqrt(hs->rr);
        info->solve_ms = 0.0;
        info->resnorm_len = 0;
        if (time_it) {
            float ms = 0.f;
            FV_HIP(ctx, hipEventSynchronize(ctx->ev1));
            FV_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
            info->solve_ms = ms;
        }
    }
    if (hs->done == 2) {
        fv_set_error(ctx, "PCG breakdown: p.Ap = %g is not positive (operator not SPD?)", hs->pq);
    }
    return FV_OK;
}

// ------------------------------------------------------------------ small vector utilities
__global__ __launch_bounds__(FV_BLOCK) void dot_kernel(int64_t n, const double *__restrict__ a, const double *__restrict__ b,
                                                        double *__restrict__ part)
{
    __shared__ double smem[4];
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x; i < n; i += vec_stride())
        acc += a[i] * b[i];
    const double t = block_sum(acc, smem);
    if (threadIdx.x == 0)
        part[blockIdx.x] = t;
}

__global__ __launch_bounds__(FV_BLOCK) void diff2_kernel(int64_t n, const double *__restrict__ a, const double *__restrict__ b,
                                                          double *__restrict__ part)
{
    __shared__ double smem[4];
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x; i < n; i += vec_stride()) {
        const double d = a[i] - b[i];
        acc += d * d;
    }
    const double t = block_sum(acc, smem);
    if (threadIdx.x == 0)
        part[blockIdx.x] = t;
}

__global__ __launch_bounds__(FV_BLOCK) void final_sum_kernel(const double *__restrict__ part, int nparts, double *__restrict__ out)
{
    __shared__ double smem[4];
    const double t = reduce_partials(part, nparts, smem);
    if (threadIdx.x == 0)
        *out = t;
}

static int reduce_to_host(fv_problem *p, int G, double *out_host)
{
    fv_ctx *ctx = p->ctx;
    hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(FV_BLOCK), 0, ctx->stream, p->part_bb.p, G, p->part_bb.p + (FV_MAX_PARTIALS - 1));
    FV_LAUNCH_CHECK(ctx);
    double *h = reinterpret_cast<double *>(ctx->pinned);
    FV_HIP(ctx, hipMemcpyAsync(h, p->part_bb.p + (FV_MAX_PARTIALS - 1), sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *out_host = *h;
    return FV_OK;
}

int fv_dot_device(fv_problem *p, const double *a, const double *b, double *out_host)
{
    FV_TRY(fv_pcg_prepare(p));
    int G = vec_grid(p->n);
    if (G > FV_MAX_PARTIALS - 1)
        G = FV_MAX_PARTIALS - 1;
    hipLaunchKernelGGL(dot_kernel, dim3(G), dim3(FV_BLOCK), 0, p->ctx->stream, p->n, a, b, p->part_bb.p);
    FV_LAUNCH_CHECK(p->ctx);
    return reduce_to_host(p, G, out_host);
}

int fv_norm2_diff_device(fv_problem *p, const double *a, const double *b, double *out_host)
{
    FV_TRY(fv_pcg_prepare(p));
    int G = vec_grid(p->n);
    if (G > FV_MAX_PARTIALS - 1)
        G = FV_MAX_PARTIALS - 1;
    hipLaunchKernelGGL(diff2_kernel, dim3(G), dim3(FV_BLOCK), 0, p->ctx->stream, p->n, a, b, p->part_bb.p);
    FV_LAUNCH_CHECK(p->ctx);
    double s = 0.0;
    FV_TRY(reduce_to_host(p, G, &s));
    *out_host = sqrt(s);
    return FV_OK;
}
