// Write / read rate of a 792 MB window as a function of where it lies inside one 6 GB allocation (and of the allocation's address),
// and of separately allocated arrays with their addresses: what makes an array "write-slow"?
//   hipcc --offload-arch=gfx950 -O3 tools/placement_probe2.hip -o tools/placement_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <time.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

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
static double rate(long n, double *v, bool write, int reps = 6)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int r = 0; r < reps + 1; r++) {
        if (r == 1)
            CHECK(hipEventRecord(e0));
        if (write)
            hipLaunchKernelGGL(write_kernel, dim3(2048), dim3(512), 0, 0, n / 2, (double2 *)v);
        else
            hipLaunchKernelGGL(read_kernel, dim3(2048), dim3(512), 0, 0, n / 2, (const double2 *)v, v);
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
    const size_t bytes = (size_t)n * 8, slabsz = (size_t)6 << 30;
    char *slab = nullptr;
    CHECK(hipMalloc((void **)&slab, slabsz));
    CHECK(hipMemset(slab, 0, slabsz));
    printf("slab at %p\n", (void *)slab);
    for (size_t off = 0; off + bytes <= slabsz; off += (size_t)128 << 20)
        printf("window at +%5zu MB: write %.2f  read %.2f TB/s\n", off >> 20, rate(n, (double *)(slab + off), true), rate(n, (double *)(slab + off), false));
    // smaller windows: 64 MB pieces across the first 2 GB
    const long m = (64L << 20) / 8;
    printf("64 MB windows, write TB/s:");
    for (size_t off = 0; off < ((size_t)2 << 30); off += (size_t)64 << 20)
        printf(" %.2f", rate(m, (double *)(slab + off), true, 20));
    printf("\n");
    CHECK(hipFree(slab));
    {   // ONE array measured again and again: is the class a property of the allocation, or of the moment?
        double *v = nullptr, *w = nullptr;
        CHECK(hipMalloc((void **)&v, bytes + 4096));
        CHECK(hipMalloc((void **)&w, bytes + 4096));
        CHECK(hipMemset(v, 0, bytes));
        CHECK(hipMemset(w, 0, bytes));
        printf("two fixed arrays (%p, %p), write TB/s over time (pairs):\n", (void *)v, (void *)w);
        for (int t = 0; t < 240; t++) {
            printf(" %.2f/%.2f", rate(n, v, true, 4), rate(n, w, true, 4));
            if (t % 12 == 11)
                printf("\n");
            if (t % 40 == 39) { // an idle pause now and then
                CHECK(hipDeviceSynchronize());
                struct timespec ts = {0, 200000000};
                nanosleep(&ts, nullptr);
            }
        }
        CHECK(hipFree(v));
        CHECK(hipFree(w));
    }
    std::vector<void *> keep;
    for (int inc = 0; inc < 16; inc++) {
        double *v = nullptr;
        CHECK(hipMalloc((void **)&v, bytes + 4096));
        CHECK(hipMemset(v, 0, bytes));
        printf("separate array at %p: write %.2f  read %.2f;  64 MB pieces write:", (void *)v, rate(n, v, true), rate(n, v, false));
        for (size_t off = 0; off + ((size_t)64 << 20) <= bytes; off += (size_t)64 << 20)
            printf(" %.2f", rate(m, (double *)((char *)v + off), true, 20));
        printf("\n");
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
