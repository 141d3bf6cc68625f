// How does a small write stream affect read bandwidth?  Store y only for every M-th group.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
template <bool NT>
__global__ __launch_bounds__(256) void k(const double *__restrict__ vals, const int *__restrict__ cols, double *__restrict__ y, long ngroups, int M)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long gstride = (long)gridDim.x * 4;
    for (long g = (long)blockIdx.x * 4 + wave; g < ngroups; g += gstride) {
        const long k0 = g * 448, k1 = k0 + 448;
        double2 v[4]; int2 c[4];
#pragma unroll
        for (int it = 0; it < 4; it++) { const long j = k0 + 2 * (lane + it * 64); if (j < k1) { v[it] = *reinterpret_cast<const double2 *>(vals + j); c[it] = *reinterpret_cast<const int2 *>(cols + j); } }
        double sum = 0;
#pragma unroll
        for (int it = 0; it < 4; it++) { const long j = k0 + 2 * (lane + it * 64); if (j < k1) sum += v[it].x * c[it].x + v[it].y * c[it].y; }
        if (M > 0 && (g % M) == 0) { if (NT) __builtin_nontemporal_store(sum, y + g * 64 + lane); else y[g * 64 + lane] = sum; }
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
    const double b = nnz * 12.0;
    for (int M : {0, 64, 16, 4, 2, 1}) {
        double t0 = timeit([&] { hipLaunchKernelGGL(k<false>, dim3(2048), dim3(256), 0, 0, vals, cols, y, ngroups, M); }, 5);
        double t1 = timeit([&] { hipLaunchKernelGGL(k<true>, dim3(2048), dim3(256), 0, 0, vals, cols, y, ngroups, M); }, 5);
        const double by = M ? rows * 8.0 / M : 0;
        printf("store every %2d-th group: plain %.3f ms %.0f GB/s | nt %.3f ms %.0f GB/s   (write share %.2f%%)\n", M, t0, (b + by) / t0 / 1e6, t1, (b + by) / t1 / 1e6,
               100 * by / (b + by));
    }
    return 0;
}
