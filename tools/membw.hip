// Standalone HBM streaming micro-benchmark (read-only sum, copy) to find the practical
// ceiling on this box for the access shapes the PCG kernels use.  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int U, bool NT>
__global__ __launch_bounds__(256) void read_sum(const double2 *__restrict__ a, size_t n2, double *out)
{
    double s = 0;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n2; i += U * stride) {
        double2 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (NT) { v[u].x = __builtin_nontemporal_load(&a[i + u * stride].x); v[u].y = __builtin_nontemporal_load(&a[i + u * stride].y); }
            else v[u] = a[i + u * stride];
        }
#pragma unroll
        for (int u = 0; u < U; u++) s += v[u].x + v[u].y;
    }
    for (; i < n2; i += stride) s += a[i].x + a[i].y;
    if (s == 1.2345e300) out[0] = s;
}

template <int U>
__global__ __launch_bounds__(256) void copy_k(const double2 *__restrict__ a, double2 *__restrict__ b, size_t n2)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n2; i += U * stride) {
        double2 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = a[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; u++) b[i + u * stride] = v[u];
    }
    for (; i < n2; i += stride) b[i] = a[i];
}

// contiguous chunk per block (each block streams its own contiguous range)
__global__ __launch_bounds__(256) void read_sum_chunk(const double2 *__restrict__ a, size_t n2, double *out)
{
    const size_t per = (n2 + gridDim.x - 1) / gridDim.x;
    const size_t lo = (size_t)blockIdx.x * per, hi = lo + per < n2 ? lo + per : n2;
    double s = 0;
    for (size_t i = lo + threadIdx.x; i < hi; i += 256 * 4) {
        double2 v0 = a[i], v1 = (i + 256 < hi) ? a[i + 256] : double2{0, 0}, v2 = (i + 512 < hi) ? a[i + 512] : double2{0, 0},
                v3 = (i + 768 < hi) ? a[i + 768] : double2{0, 0};
        s += v0.x + v0.y + v1.x + v1.y + v2.x + v2.y + v3.x + v3.y;
    }
    if (s == 1.2345e300) out[0] = s;
}

template <class F>
static double timeit(F f, int reps)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f();
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; i++) f();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main()
{
    const size_t bytes = (size_t)6 << 30; // 6 GiB per array
    const size_t n2 = bytes / 16;
    double2 *a, *b; double *out;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&out, 8));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
    for (int grid : {1024, 2048, 4096, 8192, 65536}) {
        double t1 = timeit([&] { hipLaunchKernelGGL((read_sum<1, false>), dim3(grid), dim3(256), 0, 0, a, n2, out); }, 5);
        double t4 = timeit([&] { hipLaunchKernelGGL((read_sum<4, false>), dim3(grid), dim3(256), 0, 0, a, n2, out); }, 5);
        double t8 = timeit([&] { hipLaunchKernelGGL((read_sum<8, false>), dim3(grid), dim3(256), 0, 0, a, n2, out); }, 5);
        double tn = timeit([&] { hipLaunchKernelGGL((read_sum<4, true>), dim3(grid), dim3(256), 0, 0, a, n2, out); }, 5);
        double tc = timeit([&] { hipLaunchKernelGGL(read_sum_chunk, dim3(grid), dim3(256), 0, 0, a, n2, out); }, 5);
        double tcp = timeit([&] { hipLaunchKernelGGL((copy_k<4>), dim3(grid), dim3(256), 0, 0, a, b, n2); }, 5);
        printf("grid %6d: read U1 %.0f  U4 %.0f  U8 %.0f  U4nt %.0f  chunk %.0f GB/s | copy U4 %.0f GB/s (r+w)\n", grid, bytes / t1 / 1e6,
               bytes / t4 / 1e6, bytes / t8 / 1e6, bytes / tn / 1e6, bytes / tc / 1e6, 2.0 * bytes / tcp / 1e6);
    }
    return 0;
}
