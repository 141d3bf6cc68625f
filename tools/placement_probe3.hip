// An array's "write class" (tools/placement_probe2.hip: 5.2 or 5.7 TB/s, a property of the allocation) under other traversals:
// contiguous chunk per block (2048 / 256 blocks), pieces of 8 KB dealt round-robin to the blocks, chunks in a permuted order.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// mode 0: block g writes [g per, (g + 1) per); mode 1: pieces of 512 double2 round-robin; mode 2: like 0 with the chunk index bit-reversed-ish (g * 1021 mod G)
__global__ __launch_bounds__(512) void write_kernel(long n2, double2 *__restrict__ a, int mode)
{
    const long G = gridDim.x;
    if (mode == 1) {
        for (long i = (long)blockIdx.x * 512 + threadIdx.x; i < n2; i += G * 512)
            a[i] = make_double2(1.0, 2.0);
        return;
    }
    const long g = mode == 2 ? ((long)blockIdx.x * 1021) % G : blockIdx.x;
    const long per = (n2 + G - 1) / G;
    const long lo = g * per, hi = lo + per < n2 ? lo + per : n2;
    for (long i = lo + threadIdx.x; i < hi; i += 512)
        a[i] = make_double2(1.0, 2.0);
}
static double rate(long n, double *v, int mode, int grid, int reps = 6)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int r = 0; r < reps + 1; r++) {
        if (r == 1)
            CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(write_kernel, dim3(grid), dim3(512), 0, 0, n / 2, (double2 *)v, mode);
    }
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
    return 8.0 * n / (ms / reps) / 1e9;
}
int main()
{
    const long n = 99038016L;
    const size_t bytes = (size_t)n * 8;
    std::vector<void *> keep;
    printf("write TB/s: chunks x 2048 | chunks x 256 | chunks x 1024 | chunks x 4093 | round-robin pieces x 2048 | permuted chunks x 2048 | half the array, chunks x 2048\n");
    for (int inc = 0; inc < 16; inc++) {
        double *v = nullptr;
        CHECK(hipMalloc((void **)&v, bytes + 4096));
        CHECK(hipMemset(v, 0, bytes));
        printf("array at %p: %.2f | %.2f | %.2f | %.2f | %.2f | %.2f | %.2f\n", (void *)v, rate(n, v, 0, 2048), rate(n, v, 0, 256), rate(n, v, 0, 1024), rate(n, v, 0, 4093),
               rate(n, v, 1, 2048), rate(n, v, 2, 2048), rate(n / 2, v, 0, 2048));
        if (inc % 3 == 2)
            CHECK(hipFree(v));
        else
            keep.push_back(v);
        void *extra = nullptr;
        CHECK(hipMalloc(&extra, (size_t)(37 + 11 * inc) << 20));
        keep.push_back(extra);
    }
    for (void *e : keep)
        CHECK(hipFree(e));
    return 0;
}
