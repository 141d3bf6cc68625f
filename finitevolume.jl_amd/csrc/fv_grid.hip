// regulargrid / nodehycos2neighborhycos on the device.
// Reference: /root/reference/src/grid.jl:14-33, 56-110.
// Compiled with -ffp-contract=off: the reference's products/quotients are
// individually rounded, so volumes and areasoverlengths come out bit-identical.
#include "fv_internal.h"

// One thread per cell c = i3 + n3*(i2 + n2*i1) (grid.jl:60, 0-based here).  The
// reference appends a cell's x-, y-, z-face in that order while looping i1,i2,i3
// (grid.jl:72-105); the number of faces emitted by all earlier cells has the closed
// form below, so every cell knows where its faces go and no scan is needed.
__global__ __launch_bounds__(FV_BLOCK) void regulargrid_kernel(int64_t n1, int64_t n2, int64_t n3, const double *__restrict__ xs,
                                                                const double *__restrict__ ys, const double *__restrict__ zs,
                                                                int32_t *__restrict__ node1, int32_t *__restrict__ node2,
                                                                double *__restrict__ aol, double *__restrict__ volumes,
                                                                double *__restrict__ coords)
{
    const int64_t N = n1 * n2 * n3;
    const int64_t c = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (c >= N)
        return;
    const int64_t i3 = c % n3;
    const int64_t i2 = (c / n3) % n2;
    const int64_t i1 = c / (n3 * n2);
    const double dx = xs[1] - xs[0], dy = ys[1] - ys[0], dz = zs[1] - zs[0]; // grid.jl:65-67
    double areadx = dx, aready = dy, areadz = dz;
    if (i1 == 0 || i1 == n1 - 1)
        areadx *= 0.5;
    if (i2 == 0 || i2 == n2 - 1)
        aready *= 0.5;
    if (i3 == 0 || i3 == n3 - 1)
        areadz *= 0.5;
    if (volumes)
        volumes[c] = areadx * aready * areadz; // grid.jl:87
    if (coords) {
        coords[3 * c + 0] = xs[i1];
        coords[3 * c + 1] = ys[i2];
        coords[3 * c + 2] = zs[i3];
    }
    if (!node1)
        return;
    const int64_t X = (i1 < n1 - 1) ? c : (n1 - 1) * n2 * n3;
    const int64_t Y = i1 * (n2 - 1) * n3 + (i2 < n2 - 1 ? i2 * n3 + i3 : (n2 - 1) * n3);
    const int64_t Z = (i1 * n2 + i2) * (n3 - 1) + (i3 < n3 - 1 ? i3 : n3 - 1);
    int64_t j = X + Y + Z;
    if (i1 < n1 - 1) {
        node1[j] = (int32_t)c;
        node2[j] = (int32_t)(c + n3 * n2);
        aol[j] = aready * areadz / dx;
        j++;
    }
    if (i2 < n2 - 1) {
        node1[j] = (int32_t)c;
        node2[j] = (int32_t)(c + n3);
        aol[j] = areadx * areadz / dy;
        j++;
    }
    if (i3 < n3 - 1) {
        node1[j] = (int32_t)c;
        node2[j] = (int32_t)(c + 1);
        aol[j] = areadx * aready / dz;
        j++;
    }
}

int fv_grid_generate_device(fv_ctx *ctx, const double mins[3], const double maxs[3], const int64_t ns[3], int32_t *node1,
                            int32_t *node2, double *aol, double *volumes, double *coords)
{
    std::vector<double> ax[3];
    if (fv_grid_axes(mins, maxs, ns, ax) != FV_OK) {
        fv_set_error(ctx, "regulargrid needs ns[d] >= 2 in every dimension");
        return FV_ERR_ARG;
    }
    const int64_t N = ns[0] * ns[1] * ns[2];
    if (N > 0x7fffffffLL) {
        fv_set_error(ctx, "grid of %lld cells exceeds the int32 device index range", (long long)N);
        return FV_ERR_TOO_LARGE;
    }
    DevBuf<double> dax[3];
    for (int d = 0; d < 3; d++) {
        FV_TRY(dax[d].alloc(ctx, ax[d].size()));
        FV_HIP(ctx, hipMemcpyAsync(dax[d].p, ax[d].data(), ax[d].size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    }
    hipLaunchKernelGGL(regulargrid_kernel, dim3(fv_blocks(N)), dim3(FV_BLOCK), 0, ctx->stream, ns[0], ns[1], ns[2], dax[0].p,
                       dax[1].p, dax[2].p, node1, node2, aol, volumes, coords);
    FV_LAUNCH_CHECK(ctx);
    FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FV_OK;
}

extern "C" int fv_regulargrid_sizes(const int64_t ns[3], int64_t *N, int64_t *F)
{
    if (!ns || !N || !F)
        return FV_ERR_ARG;
    if (ns[0] < 2 || ns[1] < 2 || ns[2] < 2) {
        fv_set_error(nullptr, "regulargrid needs ns[d] >= 2 in every dimension");
        return FV_ERR_ARG;
    }
    *N = ns[0] * ns[1] * ns[2];
    *F = 3 * (*N) - ns[0] * ns[1] - ns[0] * ns[2] - ns[1] * ns[2]; // grid.jl:69
    return FV_OK;
}

__global__ __launch_bounds__(FV_BLOCK) void widen_kernel(const int32_t *__restrict__ src, int64_t *__restrict__ dst, int64_t n,
                                                          int64_t add)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i < n)
        dst[i] = (int64_t)src[i] + add;
}

__global__ __launch_bounds__(FV_BLOCK) void narrow_kernel(const int64_t *__restrict__ src, int32_t *__restrict__ dst, int64_t n,
                                                           int64_t lo, int64_t hi, int *__restrict__ bad)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i < n) {
        const int64_t v = src[i];
        if (v < lo || v > hi) {
            *bad = 1;
            dst[i] = 0;
        } else
            dst[i] = (int32_t)(v - lo);
    }
}

int fv_widen_indices(fv_ctx *ctx, const int32_t *src, int64_t *dst, int64_t n, int64_t add)
{
    if (n <= 0)
        return FV_OK;
    hipLaunchKernelGGL(widen_kernel, dim3(fv_blocks(n)), dim3(FV_BLOCK), 0, ctx->stream, src, dst, n, add);
    FV_LAUNCH_CHECK(ctx);
    return FV_OK;
}

// dst[i] = src[i] - lo after checking lo <= src[i] <= hi; *bad (host) set if any index is out of range
int fv_narrow_indices(fv_ctx *ctx, const int64_t *src, int32_t *dst, int64_t n, int64_t lo, int64_t hi, int *bad)
{
    *bad = 0;
    if (n <= 0)
        return FV_OK;
    DevBuf<int> dbad;
    FV_TRY(dbad.alloc(ctx, 1));
    FV_TRY(dbad.zero(ctx));
    hipLaunchKernelGGL(narrow_kernel, dim3(fv_blocks(n)), dim3(FV_BLOCK), 0, ctx->stream, src, dst, n, lo, hi, dbad.p);
    FV_LAUNCH_CHECK(ctx);
    FV_HIP(ctx, hipMemcpyAsync(bad, dbad.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FV_OK;
}

// Stage a caller array (host or device) of n elements into a fresh device buffer.
template <class T>
static int stage_in(fv_ctx *ctx, const T *src, int64_t n, DevBuf<T> &buf)
{
    FV_TRY(buf.alloc(ctx, (size_t)n));
    if (n > 0)
        FV_HIP(ctx, hipMemcpyAsync(buf.p, src, (size_t)n * sizeof(T), hipMemcpyDefault, ctx->stream));
    return FV_OK;
}

extern "C" int fv_regulargrid(fv_ctx *ctx, const double mins[3], const double maxs[3], const int64_t ns[3], double *coords,
                              int64_t *node1, int64_t *node2, double *areasoverlengths, double *volumes)
{
    if (!ctx || !mins || !maxs || !ns)
        return FV_ERR_ARG;
    FV_HIP(ctx, hipSetDevice(ctx->device));
    int64_t N, F;
    FV_TRY(fv_regulargrid_sizes(ns, &N, &F));
    DevBuf<int32_t> d1, d2;
    DevBuf<double> daol, dvol, dco;
    const bool want_faces = node1 || node2 || areasoverlengths;
    if (want_faces) {
        FV_TRY(d1.alloc(ctx, (size_t)F));
        FV_TRY(d2.alloc(ctx, (size_t)F));
        FV_TRY(daol.alloc(ctx, (size_t)F));
    }
    if (volumes)
        FV_TRY(dvol.alloc(ctx, (size_t)N));
    if (coords)
        FV_TRY(dco.alloc(ctx, (size_t)(3 * N)));
    FV_TRY(fv_grid_generate_device(ctx, mins, maxs, ns, want_faces ? d1.p : nullptr, d2.p, daol.p, volumes ? dvol.p : nullptr,
                                   coords ? dco.p : nullptr));
    if (want_faces) {
        DevBuf<int64_t> wide;
        FV_TRY(wide.alloc(ctx, (size_t)F));
        if (node1) {
            FV_TRY(fv_widen_indices(ctx, d1.p, wide.p, F, 1));
            FV_TRY(fv_copy(ctx, node1, wide.p, (size_t)F * sizeof(int64_t)));
        }
        if (node2) {
            FV_TRY(fv_widen_indices(ctx, d2.p, wide.p, F, 1));
            FV_TRY(fv_copy(ctx, node2, wide.p, (size_t)F * sizeof(int64_t)));
        }
        if (areasoverlengths)
            FV_TRY(fv_copy(ctx, areasoverlengths, daol.p, (size_t)F * sizeof(double)));
    }
    if (volumes)
        FV_TRY(fv_copy(ctx, volumes, dvol.p, (size_t)N * sizeof(double)));
    if (coords)
        FV_TRY(fv_copy(ctx, coords, dco.p, (size_t)(3 * N) * sizeof(double)));
    return FV_OK;
}

__global__ __launch_bounds__(FV_BLOCK) void neighborhycos_kernel(int64_t F, const int32_t *__restrict__ node1,
                                                                  const int32_t *__restrict__ node2,
                                                                  const double *__restrict__ nodehycos, int logt,
                                                                  double *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i >= F)
        return;
    const double k1 = nodehycos[node1[i]], k2 = nodehycos[node2[i]];
    out[i] = logt ? 0.5 * (k1 + k2) : sqrt(k1 * k2); // grid.jl:27,29
}

extern "C" int fv_nodehycos2neighborhycos(fv_ctx *ctx, int64_t F, const int64_t *node1, const int64_t *node2, int64_t N,
                                          const double *nodehycos, int logtransformhyco, double *neighborhycos)
{
    if (!ctx || F < 0 || N < 0 || (F > 0 && (!node1 || !node2 || !neighborhycos)) || (N > 0 && !nodehycos))
        return FV_ERR_ARG;
    FV_HIP(ctx, hipSetDevice(ctx->device));
    if (F == 0)
        return FV_OK;
    DevBuf<int64_t> w;
    DevBuf<int32_t> d1, d2;
    DevBuf<double> dk, dout;
    FV_TRY(d1.alloc(ctx, (size_t)F));
    FV_TRY(d2.alloc(ctx, (size_t)F));
    int bad1 = 0, bad2 = 0;
    FV_TRY(stage_in(ctx, node1, F, w));
    FV_TRY(fv_narrow_indices(ctx, w.p, d1.p, F, 1, N, &bad1));
    FV_TRY(stage_in(ctx, node2, F, w));
    FV_TRY(fv_narrow_indices(ctx, w.p, d2.p, F, 1, N, &bad2));
    if (bad1 || bad2) {
        fv_set_error(ctx, "BoundsError: neighbor index outside 1:%lld", (long long)N);
        return FV_ERR_INDEX;
    }
    FV_TRY(stage_in(ctx, nodehycos, N, dk));
    FV_TRY(dout.alloc(ctx, (size_t)F));
    hipLaunchKernelGGL(neighborhycos_kernel, dim3(fv_blocks(F)), dim3(FV_BLOCK), 0, ctx->stream, F, d1.p, d2.p, dk.p,
                       logtransformhyco, dout.p);
    FV_LAUNCH_CHECK(ctx);
    return fv_copy(ctx, neighborhycos, dout.p, (size_t)F * sizeof(double));
}
