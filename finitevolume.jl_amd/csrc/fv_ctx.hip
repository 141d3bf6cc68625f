// Context, error reporting and the device exclusive scan used by the map and
// CSR-pointer builders.
#include "fv_internal.h"
#include <cstdlib>

static thread_local std::string g_err;

static thread_local unsigned long long g_err_seq = 0, g_err_noctx_seq = 0; // per host thread, like g_err
void fv_set_error(fv_ctx *ctx, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err_seq++;
    if (ctx) {
        ctx->err = buf;
        ctx->err_seq = g_err_seq;
    } else
        g_err_noctx_seq = g_err_seq;
    g_err = buf;
}

extern "C" int fv_abi_version(void) { return FVHIP_ABI_VERSION; }

// the context's own last message, unless a helper without a context (argument checks of the grid functions, ...) has
// reported something since: then that message is the one the failing call produced
extern "C" const char *fv_last_error(fv_ctx *ctx) { return (ctx && ctx->err_seq > g_err_noctx_seq) ? ctx->err.c_str() : g_err.c_str(); }

__global__ void fv_warm_ctx_kernel() {}
void fv_warm_amg(hipStream_t);
void fv_warm_assembly(hipStream_t);
void fv_warm_dist(hipStream_t);
void fv_warm_fused(hipStream_t);
void fv_warm_gradient(hipStream_t);
void fv_warm_grid(hipStream_t);
void fv_warm_lean(hipStream_t);
void fv_warm_pcg(hipStream_t);
void fv_warm_place(hipStream_t);
void fv_warm_reorder(hipStream_t);
void fv_warm_small(hipStream_t);
void fv_warm_spmv(hipStream_t);
void fv_warm_trajectory(hipStream_t);
void fv_warm_transient(hipStream_t);

// every code object of the library loaded before the first call that needs one (FV_WARM_TU, fv_internal.h); once per process and device
static int fv_warm_modules(fv_ctx *ctx)
{
    static bool done[64] = {};
    if (ctx->device < 64) {
        if (done[ctx->device])
            return FV_OK;
        done[ctx->device] = true;
    }
    hipLaunchKernelGGL(fv_warm_ctx_kernel, dim3(1), dim3(64), 0, ctx->stream);
    fv_warm_amg(ctx->stream);
    fv_warm_assembly(ctx->stream);
    fv_warm_dist(ctx->stream);
    fv_warm_fused(ctx->stream);
    fv_warm_gradient(ctx->stream);
    fv_warm_grid(ctx->stream);
    fv_warm_lean(ctx->stream);
    fv_warm_pcg(ctx->stream);
    fv_warm_place(ctx->stream);
    fv_warm_reorder(ctx->stream);
    fv_warm_small(ctx->stream);
    fv_warm_spmv(ctx->stream);
    fv_warm_trajectory(ctx->stream);
    fv_warm_transient(ctx->stream);
    FV_LAUNCH_CHECK(ctx);
    FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FV_OK;
}

static int ctx_init(fv_ctx *ctx)
{
    FV_HIP(ctx, hipSetDevice(ctx->device));
    hipDeviceProp_t prop;
    FV_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
    ctx->num_cus = prop.multiProcessorCount;
    ctx->total_mem = (int64_t)prop.totalGlobalMem;
    ctx->name = prop.name;
    if (ctx->name.empty())
        ctx->name = prop.gcnArchName; // some ROCm builds leave the marketing name blank
    FV_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    FV_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
    FV_HIP(ctx, hipEventCreate(&ctx->ev0));
    FV_HIP(ctx, hipEventCreate(&ctx->ev1));
    FV_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_halo, hipEventDisableTiming));
    FV_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_comp, hipEventDisableTiming));
    ctx->pinned_bytes = 4096;
    FV_HIP(ctx, hipHostMalloc(&ctx->pinned, ctx->pinned_bytes, hipHostMallocDefault));
    FV_TRY(fv_warm_modules(ctx));
    return FV_OK;
}

extern "C" void fv_ctx_destroy(fv_ctx *ctx);

extern "C" int fv_ctx_set_option(fv_ctx *ctx, int option, int value)
{
    if (!ctx)
        return FV_ERR_ARG;
    if (option == FV_OPT_REORDER && value >= 0 && value <= 2) {
        ctx->opt_reorder = value;
        return FV_OK;
    }
    if (option == FV_OPT_LEAN_SETUP && value >= 0 && value <= 2) {
        ctx->opt_lean = value;
        return FV_OK;
    }
    fv_set_error(ctx, "fv_ctx_set_option: unknown option %d or value %d out of range", option, value);
    return FV_ERR_ARG;
}

extern int g_reorder; // fv_assembly.hip
extern "C" int fv_ctx_get_option(fv_ctx *ctx, int option, int *value)
{
    if (!ctx || !value)
        return FV_ERR_ARG;
    if (option == FV_OPT_REORDER) {
        *value = ctx->opt_reorder >= 0 ? ctx->opt_reorder : g_reorder;
        return FV_OK;
    }
    if (option == FV_OPT_LEAN_SETUP) {
        *value = ctx->opt_lean;
        return FV_OK;
    }
    fv_set_error(ctx, "fv_ctx_get_option: unknown option %d", option);
    return FV_ERR_ARG;
}

extern "C" int fv_ctx_create(int device, fv_ctx **out)
{
    if (!out)
        return FV_ERR_ARG;
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        fv_set_error(nullptr, "no HIP device visible (hipGetDeviceCount: %s); libfvhip has no CPU fallback",
                     hipGetErrorString(e));
        return FV_ERR_HIP;
    }
    if (device < 0 || device >= count) {
        fv_set_error(nullptr, "device %d out of range (have %d)", device, count);
        return FV_ERR_ARG;
    }
    {
        // FV_TUNE="key=value,...": the experimenter's panel (fv_tune.h) without a call, read once per process
        static bool env_read = false;
        if (!env_read) {
            env_read = true;
            if (const char *e = getenv("FV_PLACE")) // 0: the loop's vectors from plain allocations (fv_place.hip)
                g_place = atoi(e) != 0;
            if (const char *e = getenv("FV_ALLOC_SKEW")) // bytes by which consecutive large arrays are staggered inside their allocations (DevBuf, fv_internal.h)
                g_alloc_skew_bytes = atoi(e) > 0 ? atoi(e) / 256 * 256 : 0;
            if (const char *e = getenv("FV_TUNE")) {
                const char *s = e;
                while (*s) {
                    char *end = nullptr;
                    const long k = strtol(s, &end, 10);
                    if (end == s || *end != '=')
                        break;
                    s = end + 1;
                    const long v = strtol(s, &end, 10);
                    if (end == s)
                        break;
                    if (fv_tune((int)k, (int)v) != FV_OK)
                        fprintf(stderr, "[fvhip] FV_TUNE: %ld=%ld refused\n", k, v);
                    s = (*end == ',') ? end + 1 : end;
                    if (*end != ',' && *end != 0)
                        break;
                }
            }
        }
    }
    fv_ctx *ctx = new fv_ctx();
    ctx->device = device;
    const int rc = ctx_init(ctx);
    if (rc != FV_OK) { // hand the message over to the context-free slot and release whatever was created
        fv_set_error(nullptr, "%s", ctx->err.c_str());
        fv_ctx_destroy(ctx);
        return rc;
    }
    *out = ctx;
    return FV_OK;
}

extern "C" void fv_ctx_destroy(fv_ctx *ctx)
{
    if (!ctx)
        return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    fv_comm_destroy(ctx);
    for (auto &v : ctx->diag_ev)
        for (hipEvent_t e : v)
            (void)hipEventDestroy(e);
    if (ctx->pinned)
        (void)hipHostFree(ctx->pinned);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->ev_halo) (void)hipEventDestroy(ctx->ev_halo);
    if (ctx->ev_comp) (void)hipEventDestroy(ctx->ev_comp);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
    delete ctx;
}

extern "C" int fv_ctx_synchronize(fv_ctx *ctx)
{
    if (!ctx)
        return FV_ERR_ARG;
    FV_HIP(ctx, hipSetDevice(ctx->device));
    FV_HIP(ctx, hipDeviceSynchronize());
    return FV_OK;
}

extern "C" int fv_device_info(fv_ctx *ctx, char *name, int name_cap, int *compute_units, int64_t *total_mem_bytes)
{
    if (!ctx)
        return FV_ERR_ARG;
    if (name && name_cap > 0)
        snprintf(name, (size_t)name_cap, "%s", ctx->name.c_str());
    if (compute_units)
        *compute_units = ctx->num_cus;
    if (total_mem_bytes)
        *total_mem_bytes = ctx->total_mem;
    return FV_OK;
}

extern "C" int fv_device_mem_info(fv_ctx *ctx, int64_t *free_bytes, int64_t *total_bytes)
{
    if (!ctx)
        return FV_ERR_ARG;
    FV_HIP(ctx, hipSetDevice(ctx->device));
    size_t f = 0, t = 0;
    FV_HIP(ctx, hipMemGetInfo(&f, &t));
    if (free_bytes)
        *free_bytes = (int64_t)f;
    if (total_bytes)
        *total_bytes = (int64_t)t;
    return FV_OK;
}

// ------------------------------------------------------------------ exclusive scan
// Three-phase hierarchical scan: tiles of 2048 int32 (256 threads x 8), tile
// totals scanned recursively in int64, then added back.  Deterministic.
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = FV_BLOCK * SCAN_ITEMS;

template <class Tin>
__global__ __launch_bounds__(FV_BLOCK) void scan_tile_kernel(const Tin *__restrict__ in, int64_t n,
                                                              int64_t *__restrict__ local, int64_t *__restrict__ tile_sums)
{
    __shared__ int64_t wave_tot[FV_BLOCK / 64];
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    int64_t v[SCAN_ITEMS];
    int64_t tsum = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        const int64_t i = base + k;
        v[k] = (i < n) ? (int64_t)in[i] : 0;
        tsum += v[k];
    }
    // inclusive scan of tsum across the wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int64_t inc = tsum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int64_t o = __shfl_up(inc, off, 64);
        if (lane >= off)
            inc += o;
    }
    if (lane == 63)
        wave_tot[wave] = inc;
    __syncthreads();
    int64_t woff = 0;
    for (int w = 0; w < wave; w++)
        woff += wave_tot[w];
    int64_t run = woff + inc - tsum; // exclusive prefix of this thread within the tile
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        const int64_t i = base + k;
        if (i < n)
            local[i] = run;
        run += v[k];
    }
    if (threadIdx.x == FV_BLOCK - 1)
        tile_sums[blockIdx.x] = run;
}

template <class Tout>
__global__ __launch_bounds__(FV_BLOCK) void scan_add_kernel(const int64_t *__restrict__ local,
                                                             const int64_t *__restrict__ tile_off, int64_t n,
                                                             Tout *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i < n)
        out[i] = (Tout)(local[i] + tile_off[i / SCAN_TILE]);
}

// in-place-capable recursive scan on int64 data: data[i] <- exclusive prefix, returns total (host)
static int scan_i64_inplace(fv_ctx *ctx, int64_t *data, int64_t n, int64_t *total)
{
    if (n <= 0) {
        *total = 0;
        return FV_OK;
    }
    const int64_t ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    DevBuf<int64_t> local, sums;
    FV_TRY(local.alloc(ctx, (size_t)n));
    FV_TRY(sums.alloc(ctx, (size_t)ntiles));
    hipLaunchKernelGGL(scan_tile_kernel<int64_t>, dim3((unsigned)ntiles), dim3(FV_BLOCK), 0, ctx->stream, data, n, local.p,
                       sums.p);
    FV_LAUNCH_CHECK(ctx);
    if (ntiles > 1) {
        FV_TRY(scan_i64_inplace(ctx, sums.p, ntiles, total));
    } else {
        FV_HIP(ctx, hipMemcpyAsync(total, sums.p, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
        FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        FV_HIP(ctx, hipMemsetAsync(sums.p, 0, sizeof(int64_t), ctx->stream));
    }
    hipLaunchKernelGGL(scan_add_kernel<int64_t>, dim3(fv_blocks(n)), dim3(FV_BLOCK), 0, ctx->stream, local.p, sums.p, n,
                       data);
    FV_LAUNCH_CHECK(ctx);
    FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FV_OK;
}

int fv_exclusive_scan_i32(fv_ctx *ctx, const int32_t *in, int32_t *out, int64_t n, int64_t *total)
{
    if (n <= 0) {
        *total = 0;
        int32_t z = 0;
        FV_HIP(ctx, hipMemcpyAsync(out, &z, sizeof z, hipMemcpyHostToDevice, ctx->stream));
        FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return FV_OK;
    }
    const int64_t ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    DevBuf<int64_t> local, sums;
    FV_TRY(local.alloc(ctx, (size_t)n));
    FV_TRY(sums.alloc(ctx, (size_t)ntiles));
    hipLaunchKernelGGL(scan_tile_kernel<int32_t>, dim3((unsigned)ntiles), dim3(FV_BLOCK), 0, ctx->stream, in, n, local.p,
                       sums.p);
    FV_LAUNCH_CHECK(ctx);
    FV_TRY(scan_i64_inplace(ctx, sums.p, ntiles, total));
    if (*total > 0x7fffffffLL) {
        fv_set_error(ctx, "scan total %lld exceeds the int32 device index range", (long long)*total);
        return FV_ERR_TOO_LARGE;
    }
    hipLaunchKernelGGL(scan_add_kernel<int32_t>, dim3(fv_blocks(n)), dim3(FV_BLOCK), 0, ctx->stream, local.p, sums.p, n,
                       out);
    FV_LAUNCH_CHECK(ctx);
    const int32_t t32 = (int32_t)*total;
    FV_HIP(ctx, hipMemcpyAsync(out + n, &t32, sizeof t32, hipMemcpyHostToDevice, ctx->stream));
    FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FV_OK;
}

// ------------------------------------------------------------------ stream compaction
__global__ __launch_bounds__(FV_BLOCK) void compact_kernel(int64_t n, const int32_t *__restrict__ flag, const int32_t *__restrict__ scan,
                                                            int32_t *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i < n && flag[i])
        out[scan[i]] = (int32_t)i;
}

// out[] <- ascending indices i with flag[i] != 0 (flags must be 0/1); *count = how many
int fv_compact_flags(fv_ctx *ctx, const int32_t *flag, int64_t n, int32_t *out, int64_t *count)
{
    DevBuf<int32_t> scan;
    FV_TRY(scan.alloc(ctx, (size_t)n + 1));
    FV_TRY(fv_exclusive_scan_i32(ctx, flag, scan.p, n, count));
    if (*count > 0 && out) {
        hipLaunchKernelGGL(compact_kernel, dim3(fv_blocks(n)), dim3(FV_BLOCK), 0, ctx->stream, n, flag, scan.p, out);
        FV_LAUNCH_CHECK(ctx);
        FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return FV_OK;
}

// ------------------------------------------------------------------ scratch pool of a set-up phase (see fv_internal.h)
#include <chrono>
#include <map>
#include <unordered_map>
namespace {
struct DevPool {
    int depth = 0;
    std::multimap<size_t, void *> idle;          // released blocks by capacity
    std::unordered_map<void *, size_t> capacity; // blocks handed out while the pool is on
};
thread_local DevPool t_pool;
// FV_TRACE_ALLOC=1: every device allocation / release that takes more than 0.1 ms goes to stderr (where a set-up phase loses its time)
bool trace_alloc()
{
    static const bool on = getenv("FV_TRACE_ALLOC") != nullptr;
    return on;
}
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
} // namespace

hipError_t fv_dev_malloc(void **p, size_t bytes)
{
    DevPool &pool = t_pool;
    if (pool.depth > 0) {
        auto it = pool.idle.lower_bound(bytes);
        if (it != pool.idle.end() && it->first <= 4 * bytes + ((size_t)1 << 20)) {
            *p = it->second;
            pool.capacity[*p] = it->first;
            pool.idle.erase(it);
            return hipSuccess;
        }
    }
    const double t0 = trace_alloc() ? now_s() : 0.0;
    const hipError_t e = hipMalloc(p, bytes);
    if (trace_alloc() && now_s() - t0 > 1e-4)
        fprintf(stderr, "[alloc] hipMalloc %.1f MB: %.0f us\n", (double)bytes / 1048576.0, (now_s() - t0) * 1e6);
    if (e == hipSuccess && pool.depth > 0)
        pool.capacity[*p] = bytes;
    return e;
}

void fv_dev_free(void *p)
{
    DevPool &pool = t_pool;
    if (pool.depth > 0) {
        auto it = pool.capacity.find(p);
        if (it != pool.capacity.end()) {
            pool.idle.emplace(it->second, p);
            pool.capacity.erase(it);
            return;
        }
    }
    const double t0 = trace_alloc() ? now_s() : 0.0;
    (void)hipFree(p);
    if (trace_alloc() && now_s() - t0 > 1e-4)
        fprintf(stderr, "[alloc] hipFree: %.0f us\n", (now_s() - t0) * 1e6);
}

void fv_pool_begin() { t_pool.depth++; }

void fv_pool_end()
{
    DevPool &pool = t_pool;
    if (pool.depth <= 0 || --pool.depth > 0)
        return;
    for (auto &kv : pool.idle)
        (void)hipFree(kv.second);
    pool.idle.clear();
    pool.capacity.clear(); // (blocks still in use are their owners' from here on)
}
