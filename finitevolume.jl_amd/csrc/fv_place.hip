// Where the vectors a stepping loop writes in every step live.  An array of several hundred MB from hipMalloc has a "write class" that
// belongs to the allocation for as long as it lives: written by the same kernel, most arrays take 5.5 TB/s and some 4.0-4.8 (the same
// array again and again: the same figure; reads do not tell them apart) — tools/placement_probe2.hip, placement_probe4.hip,
// profiles/r05_placement_probe*.log.  A fused step whose three output vectors are of the slow kind runs 10-15 % longer: the run-to-run
// spread of the bench between processes.  So the vectors written in every step (the states, z / p / w and their ping-pong partners, v)
// are chosen: each request looks at a few candidates, times a chunked write of zeros into each, and takes the fastest; the others wait
// for the next request — the vectors that are only read (M^-1, rhs, scratch) take the slowest.  Nothing depends on it but time.
#include "fv_internal.h"

int g_place = 1; // FV_PLACE=0: plain allocations (fv_ctx_create reads the environment once)

__global__ __launch_bounds__(512) void place_write_kernel(long n2, double2 *__restrict__ a)
{
    const long per = (n2 + gridDim.x - 1) / gridDim.x;
    const long lo = (long)blockIdx.x * per, hi = lo + per < n2 ? lo + per : n2;
    for (long i = lo + threadIdx.x; i < hi; i += 512)
        a[i] = make_double2(0.0, 0.0);
}

// room for the stagger of the large arrays inside their allocations (FV_ALLOC_SKEW, DevBuf::alloc): the caller adds fv_vec_skew()
static size_t place_skew_room() { return g_alloc_skew_bytes > 0 ? (size_t)16 * (size_t)g_alloc_skew_bytes : 0; }
size_t fv_vec_skew(size_t bytes)
{
    return (g_alloc_skew_bytes > 0 && bytes >= ((size_t)1 << 22)) ? (size_t)(g_alloc_skew_count++ % 16) * (size_t)g_alloc_skew_bytes : 0;
}

constexpr size_t PLACE_MIN_BYTES = (size_t)256 << 20; // below this the last-level cache takes the writes and the classes do not show

// one candidate: allocated, touched, timed (bytes per second of a 256-block chunked write, the fused kernels' grid)
static int place_candidate(fv_problem *p, size_t count)
{
    fv_ctx *ctx = p->ctx;
    void *base = nullptr;
    const hipError_t e = hipMalloc(&base, count * sizeof(double) + place_skew_room());
    if (e != hipSuccess) {
        fv_set_error(ctx, "hipMalloc of %zu bytes failed: %s", count * sizeof(double), hipGetErrorString(e));
        return FV_ERR_NOMEM;
    }
    const long n2 = (long)(count / 2);
    hipLaunchKernelGGL(place_write_kernel, dim3(256), dim3(512), 0, ctx->stream, n2, (double2 *)base); // first touch
    FV_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    for (int r = 0; r < 3; r++)
        hipLaunchKernelGGL(place_write_kernel, dim3(256), dim3(512), 0, ctx->stream, n2, (double2 *)base);
    FV_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    FV_HIP(ctx, hipEventSynchronize(ctx->ev1));
    float ms = 0.0f;
    FV_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    fv_problem::PlacedSpare s;
    s.base = base;
    s.count = count;
    s.rate = ms > 0.0f ? 3.0 * (double)count * 8.0 / ((double)ms * 1e-3) : 0.0;
    if (s.rate > p->place_best)
        p->place_best = s.rate;
    p->place_probes++;
    p->spares.push_back(s);
    return FV_OK;
}

// count doubles for a vector that is written in every step (hot) or only read there; *base_out: what hipMalloc returned (free with hipFree)
int fv_vec_alloc_raw(fv_problem *p, size_t count, bool hot, void **base_out)
{
    fv_ctx *ctx = p->ctx;
    *base_out = nullptr;
    if (count == 0)
        count = 1;
    bool placed = g_place && count * sizeof(double) >= PLACE_MIN_BYTES;
    if (placed) { // candidates are a luxury of free memory: with less than eight such vectors' worth left (1e9 cells on one GPU) every request takes what
                  // it gets, and the candidates at hand go back first
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) != hipSuccess || fr < 8 * count * sizeof(double)) {
            placed = false;
            bool mine = false;
            for (auto &s : p->spares)
                mine = mine || s.count == count;
            if (!mine)
                fv_vec_release_spares(p);
        }
    }
    // candidates of another size (the problem was re-dimensioned: never happens today) are of no use to this request
    auto usable = [&](const fv_problem::PlacedSpare &s) { return s.count == count; };
    if (placed && hot) {
        int made = 0;
        auto npool = [&]() {
            int c = 0;
            for (auto &s : p->spares)
                c += usable(s) ? 1 : 0;
            return c;
        };
        while (npool() < 3 && made < 3) {
            if (place_candidate(p, count) != FV_OK)
                break; // (no memory for another candidate: what is there must do)
            made++;
        }
        auto fastest = [&]() {
            int best = -1;
            for (size_t i = 0; i < p->spares.size(); i++)
                if (usable(p->spares[i]) && (best < 0 || p->spares[i].rate > p->spares[(size_t)best].rate))
                    best = (int)i;
            return best;
        };
        int f = fastest();
        while (f >= 0 && p->spares[(size_t)f].rate < 0.94 * p->place_best && made < 5) { // only slow ones at hand: look a little further
            if (place_candidate(p, count) != FV_OK)
                break;
            made++;
            f = fastest();
        }
        if (f >= 0) {
            *base_out = p->spares[(size_t)f].base;
            p->spares.erase(p->spares.begin() + f);
            return FV_OK;
        }
    } else if (!p->spares.empty()) { // read-only in the loop: the slowest candidate at hand, if any (no memory for new candidates: the fastest for a vector the loop writes)
        int s = -1;
        for (size_t i = 0; i < p->spares.size(); i++)
            if (usable(p->spares[i]) && (s < 0 || (hot ? p->spares[i].rate > p->spares[(size_t)s].rate : p->spares[i].rate < p->spares[(size_t)s].rate)))
                s = (int)i;
        if (s >= 0) {
            *base_out = p->spares[(size_t)s].base;
            p->spares.erase(p->spares.begin() + s);
            return FV_OK;
        }
    }
    const hipError_t e = hipMalloc(base_out, count * sizeof(double) + place_skew_room());
    if (e != hipSuccess) {
        *base_out = nullptr;
        fv_set_error(ctx, "hipMalloc of %zu bytes failed: %s", count * sizeof(double), hipGetErrorString(e));
        return FV_ERR_NOMEM;
    }
    return FV_OK;
}

int fv_vec_alloc(fv_problem *p, DevBuf<double> &buf, size_t count, bool hot)
{
    buf.release();
    if (count == 0)
        count = 1;
    void *base = nullptr;
    FV_TRY(fv_vec_alloc_raw(p, count, hot, &base));
    buf.base = base;
    buf.p = reinterpret_cast<double *>(static_cast<char *>(base) + fv_vec_skew(count * sizeof(double)));
    buf.n = count;
    return FV_OK;
}

// candidates nobody asked for (end of a run: every vector of the loop exists by then; fv_problem_destroy)
void fv_vec_release_spares(fv_problem *p)
{
    for (auto &s : p->spares)
        (void)hipFree(s.base);
    p->spares.clear();
}

FV_WARM_TU(place)
