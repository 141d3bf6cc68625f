// K1-shaped streaming: per slice 7 lane-major value blocks + one x block in, one y block out,
// with 8-byte (W=1) or 16-byte (W=2) accesses per lane.  hipcc --offload-arch=gfx950 -O3 membw6.hip -o membw6
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int W, bool NT>
__global__ __launch_bounds__(256, 8) void k(long nsl, const double *__restrict__ vals, const double *__restrict__ x, double *__restrict__ y)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long stride = (long)gridDim.x * 4;
    for (long s = (long)blockIdx.x * 4 + wave; s < nsl; s += stride) {
        double acc[W];
#pragma unroll
        for (int w = 0; w < W; w++) acc[w] = 0.0;
        const double *xb = x + s * 64 * W + lane * W;
        double xv[W];
#pragma unroll
        for (int w = 0; w < W; w++) xv[w] = xb[w];
#pragma unroll
        for (int kk = 0; kk < 7; kk++) {
            const double *vb = vals + (s * 7 + kk) * 64 * W + lane * W;
#pragma unroll
            for (int w = 0; w < W; w++) {
                const double v = NT ? __builtin_nontemporal_load(vb + w) : vb[w];
                acc[w] += v * xv[w];
            }
        }
        double *yb = y + s * 64 * W + lane * W;
#pragma unroll
        for (int w = 0; w < W; w++) {
            if (NT) __builtin_nontemporal_store(acc[w], yb + w); else yb[w] = acc[w];
        }
    }
}
int main()
{
    const long n = 99038016;
    double *vals, *x, *y;
    CK(hipMalloc(&vals, (size_t)n * 7 * 8 + 4096));
    CK(hipMalloc(&x, (size_t)n * 8 + 4096));
    CK(hipMalloc(&y, (size_t)n * 8 + 4096));
    CK(hipMemset(vals, 0, (size_t)n * 7 * 8));
    CK(hipMemset(x, 0, (size_t)n * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double bytes = (double)n * 9 * 8;
    for (int rep = 0; rep < 2; rep++)
        for (int var = 0; var < 4; var++) {
            const int W = (var & 1) ? 2 : 1; const bool nt = var & 2;
            const long nsl = n / (64 * W);
            float best = 1e9f;
            for (int it = 0; it < 6; it++) {
                CK(hipEventRecord(e0));
                if (W == 1 && !nt) hipLaunchKernelGGL((k<1, false>), dim3(2048), dim3(256), 0, 0, nsl, vals, x, y);
                if (W == 2 && !nt) hipLaunchKernelGGL((k<2, false>), dim3(2048), dim3(256), 0, 0, nsl, vals, x, y);
                if (W == 1 && nt) hipLaunchKernelGGL((k<1, true>), dim3(2048), dim3(256), 0, 0, nsl, vals, x, y);
                if (W == 2 && nt) hipLaunchKernelGGL((k<2, true>), dim3(2048), dim3(256), 0, 0, nsl, vals, x, y);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            printf("W=%d (%2d B/lane) %s: %.3f ms  %.0f GB/s\n", W, 8 * W, nt ? "nt" : "  ", best, bytes / best / 1e6);
        }
    return 0;
}
