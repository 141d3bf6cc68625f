// Streaming micro-benchmarks behind the bandwidth statements of DESIGN.md (one file, one binary; not part of the product):
//   hipcc --offload-arch=gfx950 -O3 tools/membw.hip -o tools/membw && tools/membw <suite> [args of that suite]
// suites:
//   ceiling   read-only sum and copy ceilings of the box for the access shapes the PCG kernels use (was membw.hip)
//   skeleton  which ingredient of the wave-stream CSR SpMV costs bandwidth: LDS products, dependent row pointers (membw2)
//   colshape  which load shape of the column-index stream limits that skeleton (membw3)
//   store     does the in-order vmcnt serialise the y store into every pass (membw4)
//   writes    how a small write stream affects read bandwidth: y stored for every M-th group only (membw5)
//   k1shape   K1-shaped streaming: 7 lane-major value blocks + x in, y out, 8- or 16-byte accesses per lane (membw6)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace ceiling {
// Standalone HBM streaming micro-benchmark (read-only sum, copy) to find the practical
// ceiling on this box for the access shapes the PCG kernels use.  Not part of the product.
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

int run(int argc, char **argv)
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
#undef CK
} // namespace ceiling

namespace skeleton {
// Structural micro-benchmarks: which ingredient of the wave-stream SpMV costs bandwidth?
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

int run(int argc, char **argv)
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
#undef CK
} // namespace skeleton

namespace colshape {
// Which load shape limits the wave-stream skeleton?  (vals 16 B/lane always)
//  MODE 0: vals only                     MODE 1: + cols as int2 (8 B/lane, 4 loads)
//  MODE 2: + cols as int4 (16 B/lane, 2 loads)   MODE 3: MODE 2 without the y store
//  MODE 4: + cols as ushort2-like 4 B/lane (4 loads)   (the 16-bit local index stream)
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
int run(int argc, char **argv)
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
#undef CK
} // namespace colshape

namespace store {
// Does the in-order vmcnt (stores counted with loads) serialise the y store into every pass?
// MODE 0: store right after the row sums (as the SpMV does)   MODE 1: store of the PREVIOUS pass issued
// after this pass's loads   MODE 2: no store
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
int run(int argc, char **argv)
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
#undef CK
} // namespace store

namespace writes {
// How does a small write stream affect read bandwidth?  Store y only for every M-th group.
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
int run(int argc, char **argv)
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
#undef CK
} // namespace writes

namespace k1shape {
// K1-shaped streaming: per slice 7 lane-major value blocks + one x block in, one y block out,
// with 8-byte (W=1) or 16-byte (W=2) accesses per lane.  hipcc --offload-arch=gfx950 -O3 membw6.hip -o membw6
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
int run(int argc, char **argv)
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
#undef CK
} // namespace k1shape

int main(int argc, char **argv)
{
    const char *s = argc > 1 ? argv[1] : "";
#define SUITE(name) if (!strcmp(s, #name)) return name::run(argc - 1, argv + 1)
    SUITE(ceiling);
    SUITE(skeleton);
    SUITE(colshape);
    SUITE(store);
    SUITE(writes);
    SUITE(k1shape);
    printf("usage: membw ceiling|skeleton|colshape|store|writes|k1shape [args]\n");
    return 2;
}
