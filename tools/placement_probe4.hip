// Would choosing the arrays by their write class pay?  16 arrays of the bench's size from separate allocations, each probed with a
// 256-block chunked write; the six-stream kernel (three in, three out, 255 blocks marching like the fused step, and 2048 blocks) on the six
// fastest, the six slowest and the first six — both directions of the ping-pong.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <functional>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(512) void write_kernel(long n2, double2 *__restrict__ a)
{
    const long per = (n2 + gridDim.x - 1) / gridDim.x;
    const long lo = (long)blockIdx.x * per, hi = lo + per < n2 ? lo + per : n2;
    for (long i = lo + threadIdx.x; i < hi; i += 512)
        a[i] = make_double2(0.0, 0.0);
}
__global__ __launch_bounds__(512) void six_kernel(long n2, const double2 *__restrict__ a, const double2 *__restrict__ b, const double2 *__restrict__ c,
                                                  double2 *__restrict__ oa, double2 *__restrict__ ob, double2 *__restrict__ oc, double s)
{
    const long per = (n2 + gridDim.x - 1) / gridDim.x;
    const long lo = (long)blockIdx.x * per, hi = lo + per < n2 ? lo + per : n2;
    for (long i = lo + threadIdx.x; i < hi; i += 512) {
        const double2 x = a[i], y = b[i], z = c[i];
        oa[i] = make_double2(x.x + s * y.x, x.y + s * y.y);
        ob[i] = make_double2(y.x + s * z.x, y.y + s * z.y);
        oc[i] = make_double2(z.x + s * x.x, z.y + s * x.y);
    }
}
static float timed(int reps, const std::function<void()> &launch)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    launch();
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < reps; r++)
        launch();
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
    const long n = argc > 1 ? atol(argv[1]) : 99038016L;
    const size_t bytes = (size_t)n * 8;
    const int M = 16;
    std::vector<std::pair<double, double *>> arr;
    std::vector<void *> keep;
    for (int k = 0; k < M; k++) {
        double *v = nullptr;
        CHECK(hipMalloc((void **)&v, bytes + 4096));
        CHECK(hipMemset(v, 0, bytes));
        arr.push_back({0.0, v});
        void *extra = nullptr;
        CHECK(hipMalloc(&extra, (size_t)(37 + 11 * k) << 20));
        keep.push_back(extra);
    }
    for (int round = 0; round < 3; round++) { // every array touched before any is measured; three rounds: is the class stable?
        printf("write probe round %d:", round);
        for (auto &a : arr) {
            double *v = a.second;
            const float ms = timed(4, [&] { hipLaunchKernelGGL(write_kernel, dim3(256), dim3(512), 0, 0, n / 2, (double2 *)v); });
            a.first = 8.0 * n / ms / 1e9;
            printf(" %.2f", a.first);
        }
        printf("\n");
    }
    printf("write probe (256 blocks), TB/s, in allocation order:");
    for (auto &a : arr)
        printf(" %.2f", a.first);
    printf("\n");
    auto six = [&](const char *name, std::vector<double *> v) {
        for (int grid : {255, 2048}) {
            const float fwd = timed(10, [&] { hipLaunchKernelGGL(six_kernel, dim3(grid), dim3(512), 0, 0, n / 2, (const double2 *)v[0], (const double2 *)v[1], (const double2 *)v[2], (double2 *)v[3], (double2 *)v[4], (double2 *)v[5], 0.5); });
            const float bwd = timed(10, [&] { hipLaunchKernelGGL(six_kernel, dim3(grid), dim3(512), 0, 0, n / 2, (const double2 *)v[3], (const double2 *)v[4], (const double2 *)v[5], (double2 *)v[0], (double2 *)v[1], (double2 *)v[2], 0.5); });
            printf("%-14s %4d blocks: %.4f / %.4f ms (mean %.4f = %.2f TB/s)\n", name, grid, fwd, bwd, 0.5 * (fwd + bwd), 48.0 * n / (0.5 * (fwd + bwd)) / 1e9);
        }
    };
    std::vector<double *> first;
    for (int k = 0; k < 6; k++)
        first.push_back(arr[k].second);
    six("first six", first);
    std::sort(arr.begin(), arr.end());
    std::vector<double *> slow, fast;
    for (int k = 0; k < 6; k++) {
        slow.push_back(arr[k].second);
        fast.push_back(arr[M - 1 - k].second);
    }
    six("six slowest", slow);
    six("six fastest", fast);
    six("six slowest", slow);
    six("six fastest", fast);
    return 0;
}
