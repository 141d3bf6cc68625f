// Does the in-order vmcnt (stores counted with loads) serialise the y store into every pass?
// MODE 0: store right after the row sums (as the SpMV does)   MODE 1: store of the PREVIOUS pass issued
// after this pass's loads   MODE 2: no store
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
template <int MODE>
__global__ __launch_bounds__(256) void k(const double *__restrict__ vals, const int *__restrict__ cols, double *__restrict__ y, long ngroups)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long gstride = (long)gridDim.x * 4;
    double prev = 0; long prevrow = -1;
    for (long g = (long)blockIdx.x * 4 + wave; g < ngroups; g += gstride) {
        const long k0 = g * 448, k1 = k0 + 448;
        double2 v[4]; int2 c[4];
#pragma unroll
        for (int it = 0; it < 4; it++) { const long j = k0 + 2 * (lane + it * 64); if (j < k1) { v[it] = *reinterpret_cast<const double2 *>(vals + j); c[it] = *reinterpret_cast<const int2 *>(cols + j); } }
        if (MODE == 1 && prevrow >= 0) y[prevrow] = prev;
        double sum = 0;
#pragma unroll
        for (int it = 0; it < 4; it++) { const long j = k0 + 2 * (lane + it * 64); if (j < k1) sum += v[it].x * c[it].x + v[it].y * c[it].y; }
        if (MODE == 0) y[g * 64 + lane] = sum;
        else if (MODE == 1) { prev = sum; prevrow = g * 64 + lane; }
        else if (sum == 1.2345e300) y[0] = sum;
    }
    if (MODE == 1 && prevrow >= 0) y[prevrow] = prev;
}
template <class F> static double timeit(F f, int reps)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipEventRecord(e0)); for (int i = 0; i < reps; i++) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps;
}
int main()
{
    const long rows = 99000000 / 64 * 64, ngroups = rows / 64, nnz = rows * 7;
    double *vals, *y; int *cols;
    CK(hipMalloc(&vals, (nnz + 8) * 8)); CK(hipMalloc(&cols, (nnz + 8) * 4)); CK(hipMalloc(&y, rows * 8));
    CK(hipMemset(vals, 0, (nnz + 8) * 8)); CK(hipMemset(cols, 0, (nnz + 8) * 4));
    const double b = nnz * 12.0, by = rows * 8.0;
    for (int grid : {2048, 8192}) {
        double t0 = timeit([&] { hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, vals, cols, y, ngroups); }, 5);
        double t1 = timeit([&] { hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, vals, cols, y, ngroups); }, 5);
        double t2 = timeit([&] { hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, vals, cols, y, ngroups); }, 5);
        printf("grid %d: store now %.3f ms %.0f GB/s | store deferred %.3f ms %.0f GB/s | no store %.3f ms %.0f GB/s\n", grid, t0, (b + by) / t0 / 1e6, t1,
               (b + by) / t1 / 1e6, t2, b / t2 / 1e6);
    }
    return 0;
}
