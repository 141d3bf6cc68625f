// Which load shape limits the wave-stream skeleton?  (vals 16 B/lane always)
//  MODE 0: vals only                     MODE 1: + cols as int2 (8 B/lane, 4 loads)
//  MODE 2: + cols as int4 (16 B/lane, 2 loads)   MODE 3: MODE 2 without the y store
//  MODE 4: + cols as ushort2-like 4 B/lane (4 loads)   (the 16-bit local index stream)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
template <int MODE>
__global__ __launch_bounds__(256) void k(const double *__restrict__ vals, const int *__restrict__ cols, double *__restrict__ y, long ngroups)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long gstride = (long)gridDim.x * 4;
    for (long g = (long)blockIdx.x * 4 + wave; g < ngroups; g += gstride) {
        const long k0 = g * 448, k1 = k0 + 448;
        double2 v[4];
        double sum = 0;
#pragma unroll
        for (int it = 0; it < 4; it++) { const long j = k0 + 2 * (lane + it * 64); if (j < k1) v[it] = *reinterpret_cast<const double2 *>(vals + j); }
        if (MODE == 1) {
            int2 c[4];
#pragma unroll
            for (int it = 0; it < 4; it++) { const long j = k0 + 2 * (lane + it * 64); if (j < k1) c[it] = *reinterpret_cast<const int2 *>(cols + j); }
#pragma unroll
            for (int it = 0; it < 4; it++) { const long j = k0 + 2 * (lane + it * 64); if (j < k1) sum += v[it].x * c[it].x + v[it].y * c[it].y; }
        } else if (MODE == 2 || MODE == 3) {
            int4 c[2];
#pragma unroll
            for (int it = 0; it < 2; it++) { const long j = k0 + 4 * (lane + it * 64); if (j < k1) c[it] = *reinterpret_cast<const int4 *>(cols + j); }
#pragma unroll
            for (int it = 0; it < 4; it++) { const long j = k0 + 2 * (lane + it * 64); if (j < k1) sum += v[it].x + v[it].y; }
#pragma unroll
            for (int it = 0; it < 2; it++) { const long j = k0 + 4 * (lane + it * 64); if (j < k1) sum += c[it].x + c[it].y + c[it].z + c[it].w; }
        } else if (MODE == 4) {
            const unsigned short *c16 = reinterpret_cast<const unsigned short *>(cols);
            unsigned c[4];
#pragma unroll
            for (int it = 0; it < 4; it++) { const long j = k0 + 2 * (lane + it * 64); if (j < k1) c[it] = *reinterpret_cast<const unsigned *>(c16 + j); }
#pragma unroll
            for (int it = 0; it < 4; it++) { const long j = k0 + 2 * (lane + it * 64); if (j < k1) sum += v[it].x * (c[it] & 0xffff) + v[it].y * (c[it] >> 16); }
        } else {
#pragma unroll
            for (int it = 0; it < 4; it++) { const long j = k0 + 2 * (lane + it * 64); if (j < k1) sum += v[it].x + v[it].y; }
        }
        if (MODE != 3) y[g * 64 + lane] = sum;
        else if (sum == 1.2345e300) y[0] = sum;
    }
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
    const int grid = 2048;
    const double bv = nnz * 8.0, by = rows * 8.0;
    double t0 = timeit([&] { hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, vals, cols, y, ngroups); }, 5);
    double t1 = timeit([&] { hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, vals, cols, y, ngroups); }, 5);
    double t2 = timeit([&] { hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, vals, cols, y, ngroups); }, 5);
    double t3 = timeit([&] { hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, vals, cols, y, ngroups); }, 5);
    double t4 = timeit([&] { hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 0, 0, vals, cols, y, ngroups); }, 5);
    printf("vals only            %.3f ms %.0f GB/s\n", t0, (bv + by) / t0 / 1e6);
    printf("vals + cols int2     %.3f ms %.0f GB/s\n", t1, (bv + nnz * 4.0 + by) / t1 / 1e6);
    printf("vals + cols int4     %.3f ms %.0f GB/s\n", t2, (bv + nnz * 4.0 + by) / t2 / 1e6);
    printf("vals + cols int4 -y  %.3f ms %.0f GB/s\n", t3, (bv + nnz * 4.0) / t3 / 1e6);
    printf("vals + idx16 (4B/ln) %.3f ms %.0f GB/s\n", t4, (bv + nnz * 2.0 + by) / t4 / 1e6);
    return 0;
}
