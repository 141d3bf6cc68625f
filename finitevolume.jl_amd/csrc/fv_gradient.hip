// Parameter gradients of the transient system on the device.
//
// The adjoint workflow integrates dfdp(t)' * lambda(t) over time (transientadjointutils.jl:23-30, 57-63;
// FiniteVolume.jl:271-377 is the reference's hand-unrolled version, "this loop is slow" at :339) with
// dfdp = D^-1 (b_p - A_p u), p = [conductivities; sources; dirichletheads].  u and lambda are piecewise linear between
// the stored time steps, every entry of (b_p - A_p u)' w is bilinear in (u, w) or linear in w, so on common knots the
// integral has a closed form per interval [t_k, t_k+1] of length h:
//     int x w dt = h ((x_k w_k + x_k+1 w_k+1) / 3 + (x_k w_k+1 + x_k+1 w_k) / 6),      int w dt = h (w_k + w_k+1) / 2.
// One thread per face walks the knots with the face's two (u, w) pairs in registers; the per-parameter sums over faces
// (metaindex, Dirichlet position) are index bookkeeping the caller does on the per-face results, deterministically.
//   face both free (rows a, b):  d/dK_m :  - dc (u_a - u_b)(w_a - w_b)
//   one free end f, head H:      d/dK_m :  + dc (H - u_f) w_f ;   d/dH : + c w_f
//   free row f:                  d/dsource(node of f) : w_f
// with w = lambda ./ (Ss volumes) when the D^-1 scaling is requested, c the assembled conductance and dc = dc/dK
// (areasoverlengths, or c itself for log-conductivities).
#include "fv_internal.h"

int g_gradient_knots_per_pass = 0; // fv_tune key 20: 0 = as many as fit ~2 GiB (tests force several passes)

namespace {

struct Bilinear {
    double xx = 0.0; // int x w dt
    double w1 = 0.0; // int w dt
    __device__ void interval(double h, double x0, double w0, double x1, double w1_)
    {
        xx += h * ((x0 * w0 + x1 * w1_) * (1.0 / 3.0) + (x0 * w1_ + x1 * w0) * (1.0 / 6.0));
        w1 += h * 0.5 * (w0 + w1_);
    }
};

__global__ __launch_bounds__(FV_BLOCK) void gradient_face_kernel(int64_t F, int64_t n, int knots, const double *__restrict__ ts,
                                                                  const double *__restrict__ X, const double *__restrict__ L,
                                                                  const double *__restrict__ D, const int32_t *__restrict__ node1,
                                                                  const int32_t *__restrict__ node2, const int32_t *__restrict__ nodemap,
                                                                  const double *__restrict__ cond, const double *__restrict__ aol,
                                                                  const double *__restrict__ dheads, int logtransform, int accumulate,
                                                                  double *__restrict__ face_k, double *__restrict__ face_dir)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i >= F)
        return;
    const int32_t na = node1[i], nb = node2[i];
    const int32_t a = nodemap[na], b = nodemap[nb];
    double gk = 0.0, gd = 0.0;
    if (na != nb && (a >= 0 || b >= 0)) {
        const double c = cond[i];
        const double dc = logtransform ? c : aol[i];
        Bilinear acc;
        if (a >= 0 && b >= 0) {
            const double sa = D ? 1.0 / D[a] : 1.0, sb = D ? 1.0 / D[b] : 1.0;
            double x0 = X[a] - X[b], w0 = L[a] * sa - L[b] * sb;
            for (int k = 1; k < knots; k++) {
                const double *Xk = X + (int64_t)k * n, *Lk = L + (int64_t)k * n;
                const double x1 = Xk[a] - Xk[b], w1 = Lk[a] * sa - Lk[b] * sb;
                acc.interval(ts[k] - ts[k - 1], x0, w0, x1, w1);
                x0 = x1;
                w0 = w1;
            }
            gk = -dc * acc.xx;
        } else {
            const int32_t f = a >= 0 ? a : b;
            const double H = dheads[-(a >= 0 ? b : a) - 1];
            const double s = D ? 1.0 / D[f] : 1.0;
            double x0 = H - X[f], w0 = L[f] * s;
            for (int k = 1; k < knots; k++) {
                const double x1 = H - X[(int64_t)k * n + f], w1 = L[(int64_t)k * n + f] * s;
                acc.interval(ts[k] - ts[k - 1], x0, w0, x1, w1);
                x0 = x1;
                w0 = w1;
            }
            gk = dc * acc.xx;
            gd = c * acc.w1;
        }
    }
    face_k[i] = accumulate ? face_k[i] + gk : gk;
    face_dir[i] = accumulate ? face_dir[i] + gd : gd;
}

__global__ __launch_bounds__(FV_BLOCK) void gradient_row_kernel(int64_t n, int knots, const double *__restrict__ ts, const double *__restrict__ L,
                                                                 const double *__restrict__ D, int accumulate, double *__restrict__ row_src)
{
    const int64_t f = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (f >= n)
        return;
    const double s = D ? 1.0 / D[f] : 1.0;
    double acc = 0.0, w0 = L[f] * s;
    for (int k = 1; k < knots; k++) {
        const double w1 = L[(int64_t)k * n + f] * s;
        acc += (ts[k] - ts[k - 1]) * 0.5 * (w0 + w1);
        w0 = w1;
    }
    row_src[f] = accumulate ? row_src[f] + acc : acc;
}

// The same terms at ONE time: (b_p - A_p u)' w per face / free row — the action of the pointwise Jacobian dfdp(u, t, p)' on a vector
// (transientadjointutils.jl:23-30: assembleb_p - assembleA_px, scaled by the storage term), for callers that run their own
// quadrature over dfdp(t) * lambda(t) (transient.jl:208-219).
__global__ __launch_bounds__(FV_BLOCK) void jacobian_face_kernel(int64_t F, const double *__restrict__ X, const double *__restrict__ L,
                                                                  const double *__restrict__ D, const int32_t *__restrict__ node1,
                                                                  const int32_t *__restrict__ node2, const int32_t *__restrict__ nodemap,
                                                                  const double *__restrict__ cond, const double *__restrict__ aol,
                                                                  const double *__restrict__ dheads, int logtransform, double *__restrict__ face_k,
                                                                  double *__restrict__ face_dir)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i >= F)
        return;
    const int32_t na = node1[i], nb = node2[i];
    const int32_t a = nodemap[na], b = nodemap[nb];
    double gk = 0.0, gd = 0.0;
    if (na != nb && (a >= 0 || b >= 0)) {
        const double c = cond[i];
        const double dc = logtransform ? c : aol[i];
        if (a >= 0 && b >= 0) {
            const double sa = D ? 1.0 / D[a] : 1.0, sb = D ? 1.0 / D[b] : 1.0;
            gk = -dc * ((X[a] - X[b]) * (L[a] * sa - L[b] * sb));
        } else {
            const int32_t f = a >= 0 ? a : b;
            const double H = dheads[-(a >= 0 ? b : a) - 1];
            const double w = L[f] * (D ? 1.0 / D[f] : 1.0);
            gk = dc * ((H - X[f]) * w);
            gd = c * w;
        }
    }
    face_k[i] = gk;
    face_dir[i] = gd;
}

__global__ __launch_bounds__(FV_BLOCK) void jacobian_row_kernel(int64_t n, const double *__restrict__ L, const double *__restrict__ D,
                                                                 double *__restrict__ row_src)
{
    const int64_t f = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (f < n)
        row_src[f] = L[f] * (D ? 1.0 / D[f] : 1.0);
}

} // namespace

extern "C" int fv_param_jacobian_apply(fv_problem *p, const double *x_free, const double *lam_free, int scale_by_storage, int logtransform,
                                       double *face_k, double *face_dir, double *row_src)
{
    if (!p || !x_free || !lam_free || !face_k || !face_dir || !row_src)
        return FV_ERR_ARG;
    fv_ctx *ctx = p->ctx;
    if (!p->assembled || p->from_csc || p->dist) {
        fv_set_error(ctx, "fv_param_jacobian_apply: needs a mesh problem after fv_assemble (not a CSC import or a row block)");
        return FV_ERR_STATE;
    }
    if (scale_by_storage && !p->transient_ready) {
        fv_set_error(ctx, "fv_param_jacobian_apply: the D^-1 scaling needs fv_transient_begin (Ss, volumes)");
        return FV_ERR_STATE;
    }
    FV_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t n = p->n, F = p->F;
    DevBuf<double> X, L, gk, gd, gs;
    FV_TRY(X.alloc(ctx, (size_t)n));
    FV_TRY(L.alloc(ctx, (size_t)n));
    FV_TRY(gk.alloc(ctx, (size_t)F));
    FV_TRY(gd.alloc(ctx, (size_t)F));
    FV_TRY(gs.alloc(ctx, (size_t)n));
    FV_TRY(fv_free_in(p, X.p, x_free)); // (the caller's numbering of the free cells)
    FV_TRY(fv_free_in(p, L.p, lam_free));
    const double *D = scale_by_storage ? p->D.p : nullptr;
    FaceArrays fa; // (a lean problem's face arrays exist for the duration of this call)
    FV_TRY(fv_face_arrays(p, fa));
    if (F > 0)
        hipLaunchKernelGGL(jacobian_face_kernel, dim3(fv_blocks(F)), dim3(FV_BLOCK), 0, ctx->stream, F, (const double *)X.p, (const double *)L.p, D,
                           fa.node1, fa.node2, (const int32_t *)p->nodemap.p, fa.cond, fa.aol, (const double *)p->dheads.p, logtransform ? 1 : 0, gk.p, gd.p);
    if (n > 0)
        hipLaunchKernelGGL(jacobian_row_kernel, dim3(fv_blocks(n)), dim3(FV_BLOCK), 0, ctx->stream, n, (const double *)L.p, D, gs.p);
    FV_LAUNCH_CHECK(ctx);
    FV_HIP(ctx, hipMemcpyAsync(face_k, gk.p, (size_t)F * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    FV_HIP(ctx, hipMemcpyAsync(face_dir, gd.p, (size_t)F * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return fv_free_out(p, row_src, gs.p);
}

// one pass over kc knots already on the device (X, L: kc vectors of n doubles in the internal numbering, ts_dev their times)
int fv_param_gradient_integral_device(fv_problem *p, int64_t kc, const double *ts_dev, const double *X, const double *L, const double *D,
                                      int logtransform, int accumulate, double *gk, double *gd, double *gs)
{
    fv_ctx *ctx = p->ctx;
    const int64_t n = p->n, F = p->F;
    FaceArrays fa;
    FV_TRY(fv_face_arrays(p, fa));
    if (F > 0)
        hipLaunchKernelGGL(gradient_face_kernel, dim3(fv_blocks(F)), dim3(FV_BLOCK), 0, ctx->stream, F, n, (int)kc, ts_dev, X, L, D, fa.node1, fa.node2,
                           (const int32_t *)p->nodemap.p, fa.cond, fa.aol, (const double *)p->dheads.p, logtransform ? 1 : 0, accumulate, gk, gd);
    if (n > 0)
        hipLaunchKernelGGL(gradient_row_kernel, dim3(fv_blocks(n)), dim3(FV_BLOCK), 0, ctx->stream, n, (int)kc, ts_dev, L, D, accumulate, gs);
    FV_LAUNCH_CHECK(ctx);
    if (p->lean)
        FV_HIP(ctx, hipStreamSynchronize(ctx->stream)); // (the generated face arrays go with this call)
    return FV_OK;
}

extern "C" int fv_param_gradient_integral(fv_problem *p, int64_t nt, const double *ts, const double *x_knots, const double *lam_knots,
                                          int scale_by_storage, int logtransform, double *face_k, double *face_dir, double *row_src)
{
    if (!p || nt < 2 || !ts || !x_knots || !lam_knots || !face_k || !face_dir || !row_src)
        return FV_ERR_ARG;
    fv_ctx *ctx = p->ctx;
    if (!p->assembled || p->from_csc || p->dist) {
        fv_set_error(ctx, "fv_param_gradient_integral: needs a mesh problem after fv_assemble (not a CSC import or a row block)");
        return FV_ERR_STATE;
    }
    if (scale_by_storage && !p->transient_ready) {
        fv_set_error(ctx, "fv_param_gradient_integral: the D^-1 scaling needs fv_transient_begin (Ss, volumes)");
        return FV_ERR_STATE;
    }
    for (int64_t k = 1; k < nt; k++)
        if (!(ts[k] >= ts[k - 1])) {
            fv_set_error(ctx, "fv_param_gradient_integral: knots must not decrease (ts[%lld] = %g after %g)", (long long)k, ts[k], ts[k - 1]);
            return FV_ERR_ARG;
        }
    FV_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t n = p->n, F = p->F;
    // knots per pass: both series together stay below ~2 GiB on the device; consecutive passes share one knot
    int64_t chunk = (int64_t)(1ull << 27) / (n > 0 ? n : 1);
    if (g_gradient_knots_per_pass >= 2)
        chunk = g_gradient_knots_per_pass;
    if (chunk < 2)
        chunk = 2;
    if (chunk > nt)
        chunk = nt;
    DevBuf<double> X, L, T, gk, gd, gs;
    FV_TRY(X.alloc(ctx, (size_t)(chunk * n)));
    FV_TRY(L.alloc(ctx, (size_t)(chunk * n)));
    FV_TRY(T.alloc(ctx, (size_t)chunk));
    FV_TRY(gk.alloc(ctx, (size_t)F));
    FV_TRY(gd.alloc(ctx, (size_t)F));
    FV_TRY(gs.alloc(ctx, (size_t)n));
    const double *D = scale_by_storage ? p->D.p : nullptr;
    FaceArrays fa;
    FV_TRY(fv_face_arrays(p, fa));
    int accumulate = 0;
    for (int64_t k0 = 0; k0 + 1 < nt; k0 += chunk - 1) {
        const int64_t kc = nt - k0 < chunk ? nt - k0 : chunk;
        FV_TRY(fv_free_in(p, X.p, x_knots + k0 * n, kc)); // the knots arrive in the caller's numbering of the free cells
        FV_TRY(fv_free_in(p, L.p, lam_knots + k0 * n, kc));
        FV_HIP(ctx, hipMemcpyAsync(T.p, ts + k0, (size_t)kc * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        if (F > 0)
            hipLaunchKernelGGL(gradient_face_kernel, dim3(fv_blocks(F)), dim3(FV_BLOCK), 0, ctx->stream, F, n, (int)kc, (const double *)T.p,
                               (const double *)X.p, (const double *)L.p, D, fa.node1, fa.node2, (const int32_t *)p->nodemap.p, fa.cond, fa.aol,
                               (const double *)p->dheads.p, logtransform ? 1 : 0, accumulate, gk.p, gd.p);
        if (n > 0)
            hipLaunchKernelGGL(gradient_row_kernel, dim3(fv_blocks(n)), dim3(FV_BLOCK), 0, ctx->stream, n, (int)kc, (const double *)T.p,
                               (const double *)L.p, D, accumulate, gs.p);
        FV_LAUNCH_CHECK(ctx);
        FV_HIP(ctx, hipStreamSynchronize(ctx->stream)); // the host buffers of the next pass may be the caller's next slice
        accumulate = 1;
    }
    FV_HIP(ctx, fv_memcpy_sync(ctx, face_k, gk.p, (size_t)F * sizeof(double), hipMemcpyDeviceToHost));
    FV_HIP(ctx, fv_memcpy_sync(ctx, face_dir, gd.p, (size_t)F * sizeof(double), hipMemcpyDeviceToHost));
    FV_TRY(fv_free_out(p, row_src, gs.p));
    return FV_OK;
}

FV_WARM_TU(gradient) // (fv_ctx_create loads every code object of the library up front: fv_warm_modules, fv_ctx.hip)
