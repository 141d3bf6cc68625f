// Structural micro-benchmarks: which ingredient of the wave-stream SpMV costs bandwidth?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// MODE 0: stream vals(16B)+cols(8B) per lane, 4 pairs in flight, sum in registers, one 8B store per lane per group
// MODE 1: + products through LDS and per-lane serial row sums (7 entries per row)
// MODE 2: MODE 1 + per-group dependent "rowptr" loads (prefetched one group ahead)
template <int MODE>
__global__ __launch_bounds__(256) void k(const double *__restrict__ vals, const int *__restrict__ cols, const int *__restrict__ rowptr,
                                         double *__restrict__ y, long ngroups)
{
    __shared__ double prod_all[4][514];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double *prod = prod_all[wave];
    const long gstride = (long)gridDim.x * 4;
    long g = (long)blockIdx.x * 4 + wave;
    int s = 0, e = 0;
    if (MODE == 2 && g < ngroups) { s = rowptr[g * 64 + lane]; e = rowptr[g * 64 + lane + 1]; }
    for (; g < ngroups; g += gstride) {
        int k0 = (int)(g * 448), k1 = k0 + 448;
        int my_s = k0 + lane * 7, my_e = my_s + 7;
        if (MODE == 2) {
            my_s = s; my_e = e;
            k0 = __builtin_amdgcn_readfirstlane(my_s);
            k1 = __builtin_amdgcn_readlane(my_e, 63);
            s = 0; e = 0;
            if (g + gstride < ngroups) { s = rowptr[(g + gstride) * 64 + lane]; e = rowptr[(g + gstride) * 64 + lane + 1]; }
        }
        double2 v[4]; int2 c[4];
#pragma unroll
        for (int it = 0; it < 4; it++) {
            const int j = k0 + 2 * (lane + it * 64);
            if (j < k1) { v[it] = *reinterpret_cast<const double2 *>(vals + j); c[it] = *reinterpret_cast<const int2 *>(cols + j); }
        }
        double sum = 0;
        if (MODE == 0) {
#pragma unroll
            for (int it = 0; it < 4; it++) { const int j = k0 + 2 * (lane + it * 64); if (j < k1) sum += v[it].x * c[it].x + v[it].y * c[it].y; }
        } else {
#pragma unroll
            for (int it = 0; it < 4; it++) {
                const int j = k0 + 2 * (lane + it * 64);
                if (j < k1) { double2 pr; pr.x = v[it].x * c[it].x; pr.y = v[it].y * c[it].y; *reinterpret_cast<double2 *>(prod + (j - k0)) = pr; }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (int kk = my_s - k0; kk < my_e - k0; kk++) sum += prod[kk];
            __builtin_amdgcn_wave_barrier();
        }
        y[g * 64 + lane] = sum;
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
    double *vals, *y; int *cols, *rowptr;
    CK(hipMalloc(&vals, (nnz + 8) * 8)); CK(hipMalloc(&cols, (nnz + 8) * 4)); CK(hipMalloc(&rowptr, (rows + 8) * 4)); CK(hipMalloc(&y, rows * 8));
    CK(hipMemset(vals, 0, (nnz + 8) * 8)); CK(hipMemset(cols, 0, (nnz + 8) * 4));
    int *h = (int *)malloc((rows + 1) * 4); for (long i = 0; i <= rows; i++) h[i] = (int)(i * 7);
    CK(hipMemcpy(rowptr, h, (rows + 1) * 4, hipMemcpyHostToDevice));
    const double bytes0 = nnz * 12.0 + rows * 8.0, bytes2 = bytes0 + rows * 4.0;
    for (int grid : {2048, 4096, 8192}) {
        double t0 = timeit([&] { hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, vals, cols, rowptr, y, ngroups); }, 5);
        double t1 = timeit([&] { hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, vals, cols, rowptr, y, ngroups); }, 5);
        double t2 = timeit([&] { hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, vals, cols, rowptr, y, ngroups); }, 5);
        printf("grid %5d: stream-only %.3f ms %.0f GB/s | +LDS row sums %.3f ms %.0f GB/s | +rowptr chain %.3f ms %.0f GB/s\n", grid, t0,
               bytes0 / t0 / 1e6, t1, bytes0 / t1 / 1e6, t2, bytes2 / t2 / 1e6);
    }
    return 0;
}
