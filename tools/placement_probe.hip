// Does the relative placement of the arrays a six-stream kernel walks (three in, three out, the same index at the same time: the fused
// step's access shape without its arms) decide its speed?  (a) arrays carved out of ONE allocation at offsets k x (size + gap), gap swept;
// (b) arrays from separate hipMalloc calls, several incarnations with other allocations made and released in between.
//   hipcc --offload-arch=gfx950 -O3 tools/placement_probe.hip -o /tmp/placement_probe && /tmp/placement_probe [rows]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                  \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

__global__ __launch_bounds__(512) void six_kernel(long n2, const double2 *__restrict__ a, const double2 *__restrict__ b, const double2 *__restrict__ c,
                                                  double2 *__restrict__ oa, double2 *__restrict__ ob, double2 *__restrict__ oc, double s)
{
    // contiguous chunks per block, like the chunk traversal: block g walks [g * per, (g + 1) * per)
    const long per = (n2 + gridDim.x - 1) / gridDim.x;
    const long lo = (long)blockIdx.x * per, hi = lo + per < n2 ? lo + per : n2;
    for (long i = lo + threadIdx.x; i < hi; i += 512) {
        const double2 x = a[i], y = b[i], z = c[i];
        oa[i] = make_double2(x.x + s * y.x, x.y + s * y.y);
        ob[i] = make_double2(y.x + s * z.x, y.y + s * z.y);
        oc[i] = make_double2(z.x + s * x.x, z.y + s * x.y);
    }
}

__global__ __launch_bounds__(512) void read_kernel(long n2, const double2 *__restrict__ a, double *__restrict__ out)
{
    const long per = (n2 + gridDim.x - 1) / gridDim.x;
    const long lo = (long)blockIdx.x * per, hi = lo + per < n2 ? lo + per : n2;
    double acc = 0.0;
    for (long i = lo + threadIdx.x; i < hi; i += 512) {
        const double2 x = a[i];
        acc += x.x + x.y;
    }
    if (acc == 1.2345e300)
        out[0] = acc;
}
__global__ __launch_bounds__(512) void write_kernel(long n2, double2 *__restrict__ a)
{
    const long per = (n2 + gridDim.x - 1) / gridDim.x;
    const long lo = (long)blockIdx.x * per, hi = lo + per < n2 ? lo + per : n2;
    for (long i = lo + threadIdx.x; i < hi; i += 512)
        a[i] = make_double2(1.0, 2.0);
}
static double time_one(long n, double *v, bool write, int reps)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int grid = 256 * 8;
    for (int r = 0; r < reps + 1; r++) {
        if (r == 1)
            CHECK(hipEventRecord(e0));
        if (write)
            hipLaunchKernelGGL(write_kernel, dim3(grid), dim3(512), 0, 0, n / 2, (double2 *)v);
        else
            hipLaunchKernelGGL(read_kernel, dim3(grid), dim3(512), 0, 0, n / 2, (const double2 *)v, v);
    }
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
    return ms / reps;
}

static double time_six(long n, double *v[6], int reps)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int grid = 256 * 8;
    for (int w = 0; w < 2; w++)
        hipLaunchKernelGGL(six_kernel, dim3(grid), dim3(512), 0, 0, n / 2, (const double2 *)v[0], (const double2 *)v[1], (const double2 *)v[2], (double2 *)v[3],
                           (double2 *)v[4], (double2 *)v[5], 0.5);
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < reps; r++)
        hipLaunchKernelGGL(six_kernel, dim3(grid), dim3(512), 0, 0, n / 2, (const double2 *)v[0], (const double2 *)v[1], (const double2 *)v[2], (double2 *)v[3],
                           (double2 *)v[4], (double2 *)v[5], 0.5);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
    return ms / reps;
}

int main(int argc, char **argv)
{
    const long n = argc > 1 ? atol(argv[1]) : 99038016L; // rows of the 464^3 bench box
    const size_t bytes = (size_t)n * 8;
    printf("six streams of %ld doubles (%.1f MB each), 48 B per row: ms per launch, TB/s\n", n, bytes / 1e6);
    // (a) one allocation
    const size_t gaps[] = {0, 256, 4096, 65536, 1 << 20, (2 << 20) + 4096, (16 << 20) + 65536, 3 << 20, 5 << 20, (1 << 20) + 256};
    size_t maxgap = 0;
    for (size_t g : gaps)
        maxgap = g > maxgap ? g : maxgap;
    char *slab = nullptr;
    CHECK(hipMalloc((void **)&slab, 6 * (bytes + maxgap + (2 << 20))));
    CHECK(hipMemset(slab, 0, 6 * (bytes + maxgap + (2 << 20))));
    for (int round = 0; round < 2; round++)
        for (size_t g : gaps) {
            const size_t stride = (bytes + g + 255) / 256 * 256;
            double *v[6];
            for (int k = 0; k < 6; k++)
                v[k] = (double *)(slab + (size_t)k * stride);
            const double ms = time_six(n, v, 10);
            printf("slab  gap %9zu: %.4f ms  %.2f TB/s\n", g, ms, 48.0 * n / ms / 1e9);
        }
    CHECK(hipFree(slab));
    // (b) separate allocations, the allocator's state shifted between incarnations
    std::vector<void *> keep;
    for (int inc = 0; inc < 14; inc++) {
        double *v[6];
        for (int k = 0; k < 6; k++) {
            CHECK(hipMalloc((void **)&v[k], bytes + 4096));
            CHECK(hipMemset(v[k], 0, bytes));
        }
        const double ms = time_six(n, v, 10);
        printf("separate, incarnation %d: %.4f ms  %.2f TB/s;", inc, ms, 48.0 * n / ms / 1e9);
        printf("  read alone TB/s:");
        for (int k = 0; k < 6; k++)
            printf(" %.2f", 8.0 * n / time_one(n, v[k], false, 5) / 1e9);
        printf("  write alone:");
        for (int k = 0; k < 6; k++)
            printf(" %.2f", 8.0 * n / time_one(n, v[k], true, 5) / 1e9);
        {   // the same six arrays in another pairing of inputs and outputs
            double *w[6] = {v[3], v[4], v[5], v[0], v[1], v[2]};
            printf("  swapped roles: %.4f ms", time_six(n, w, 10));
            double *u[6] = {v[0], v[2], v[4], v[1], v[3], v[5]};
            printf("  interleaved roles: %.4f ms\n", time_six(n, u, 10));
        }
        for (int k = 0; k < 6; k++)
            CHECK(hipFree(v[k]));
        void *extra = nullptr;
        CHECK(hipMalloc(&extra, (size_t)(37 + 11 * inc) << 20));
        keep.push_back(extra);
    }
    for (void *e : keep)
        CHECK(hipFree(e));
    return 0;
}
