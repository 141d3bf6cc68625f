// Dirichlet elimination maps, symbolic CSR of assembleA, numeric fill of A and b.
// Reference: /root/reference/src/FiniteVolume.jl:20-44 (maps), :75-108 (assembleA),
// :110-139 (assembleb), :141-155 (freenodes2nodes).
//
// The reference pushes COO triplets face by face and lets SparseArrays.sparse
// sort rows and sum repeats in input order.  Here the same matrix is built
// without atomics on values: a one-time symbolic phase lists, per free row, the
// incident (face, end) pairs in face order and the sorted distinct columns; the
// numeric phase is one thread per row folding its contributions left to right,
// which reproduces sparse()'s combine order bit for bit.
// Compiled with -ffp-contract=off (Julia does not fuse a*b+c).
#include "fv_internal.h"

#include <chrono>
#include <cmath>

constexpr uint32_t SLOT_DIRICHLET = 0x7fffffffu;
constexpr uint32_t SLOT_FIRST = 0x80000000u;

// ------------------------------------------------------------------ maps (a3, a4)
__global__ __launch_bounds__(FV_BLOCK) void fill_i32_kernel(int32_t *p, int64_t n, int32_t v)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i < n)
        p[i] = v;
}

// nodei2dirichleti[node] = i, later entries overwrite earlier ones (FiniteVolume.jl:24)
__global__ __launch_bounds__(FV_BLOCK) void dirpos_kernel(const int32_t *__restrict__ dn, int64_t ndir, int32_t *__restrict__ dirpos)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i < ndir)
        atomicMax(&dirpos[dn[i]], (int32_t)i);
}

__global__ __launch_bounds__(FV_BLOCK) void freemask_kernel(const int32_t *__restrict__ dirpos, int64_t N, int32_t *__restrict__ mask)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i < N)
        mask[i] = dirpos[i] < 0 ? 1 : 0;
}

__global__ __launch_bounds__(FV_BLOCK) void nodemap_kernel(const int32_t *__restrict__ dirpos, const int32_t *__restrict__ rank,
                                                            int64_t N, int32_t *__restrict__ nodemap, int32_t *__restrict__ f2n)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i >= N)
        return;
    if (dirpos[i] < 0) {
        nodemap[i] = rank[i];
        f2n[rank[i]] = (int32_t)i;
    } else
        nodemap[i] = -(dirpos[i] + 1);
}

int fv_build_maps(fv_problem *p, const int64_t *dirichletnodes)
{
    fv_ctx *ctx = p->ctx;
    const int64_t N = p->N, ndir = p->ndir;
    DevBuf<int32_t> dirpos, mask, rank, dn;
    FV_TRY(dirpos.alloc(ctx, (size_t)N));
    FV_TRY(mask.alloc(ctx, (size_t)N));
    FV_TRY(rank.alloc(ctx, (size_t)N + 1));
    hipLaunchKernelGGL(fill_i32_kernel, dim3(fv_blocks(N)), dim3(FV_BLOCK), 0, ctx->stream, dirpos.p, N, -1);
    FV_LAUNCH_CHECK(ctx);
    if (ndir > 0) {
        DevBuf<int64_t> w;
        FV_TRY(w.alloc(ctx, (size_t)ndir));
        FV_HIP(ctx, hipMemcpyAsync(w.p, dirichletnodes, (size_t)ndir * sizeof(int64_t), hipMemcpyDefault, ctx->stream));
        FV_TRY(dn.alloc(ctx, (size_t)ndir));
        int bad = 0;
        FV_TRY(fv_narrow_indices(ctx, w.p, dn.p, ndir, 1, N, &bad));
        if (bad) {
            fv_set_error(ctx, "BoundsError: dirichletnodes entry outside 1:%lld", (long long)N);
            return FV_ERR_INDEX;
        }
        hipLaunchKernelGGL(dirpos_kernel, dim3(fv_blocks(ndir)), dim3(FV_BLOCK), 0, ctx->stream, dn.p, ndir, dirpos.p);
        FV_LAUNCH_CHECK(ctx);
    }
    hipLaunchKernelGGL(freemask_kernel, dim3(fv_blocks(N)), dim3(FV_BLOCK), 0, ctx->stream, dirpos.p, N, mask.p);
    FV_LAUNCH_CHECK(ctx);
    int64_t nfree = 0;
    FV_TRY(fv_exclusive_scan_i32(ctx, mask.p, rank.p, N, &nfree));
    p->n = nfree;
    FV_TRY(p->nodemap.alloc(ctx, (size_t)N));
    FV_TRY(p->f2n.alloc(ctx, (size_t)nfree));
    hipLaunchKernelGGL(nodemap_kernel, dim3(fv_blocks(N)), dim3(FV_BLOCK), 0, ctx->stream, dirpos.p, rank.p, N, p->nodemap.p,
                       p->f2n.p);
    FV_LAUNCH_CHECK(ctx);
    FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FV_OK;
}

// ------------------------------------------------------------------ locality re-numbering of the free cells
int g_reorder_device = 1; // fv_tune key 47: the re-numbering is computed on the device (0: by the host routine, one core)
int g_reorder = 1; // fv_tune key 31: 0 never, 1 when the mesh is numbered badly and the re-numbering helps, 2 always (tests)

__global__ __launch_bounds__(FV_BLOCK) void apply_perm_kernel(int64_t N, const int32_t *__restrict__ perm, int32_t *__restrict__ nodemap,
                                                               int32_t *__restrict__ f2n, int32_t *__restrict__ iperm)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i >= N)
        return;
    const int32_t c = nodemap[i]; // canonical free index
    if (c >= 0) {
        const int32_t r = perm[c];
        nodemap[i] = r;
        f2n[r] = (int32_t)i;
        iperm[r] = c;
    }
}

// dst[perm[i] + k n] = src[i + k n] (SCATTER = in) or dst[i + k n] = src[perm[i] + k n] (out)
template <bool SCATTER>
__global__ __launch_bounds__(FV_BLOCK) void permute_kernel(int64_t n, int64_t count, const int32_t *__restrict__ perm, const double *__restrict__ src,
                                                            double *__restrict__ dst)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i >= n)
        return;
    const int64_t r = perm[i];
    for (int64_t k = 0; k < count; k++) {
        if (SCATTER)
            dst[k * n + r] = src[k * n + i];
        else
            dst[k * n + i] = src[k * n + r];
    }
}

int fv_free_in(fv_problem *p, double *dst_dev, const double *src, int64_t count)
{
    fv_ctx *ctx = p->ctx;
    const size_t bytes = (size_t)(p->n * count) * sizeof(double);
    if (bytes == 0)
        return FV_OK;
    if (!p->reordered) {
        FV_HIP(ctx, hipMemcpyAsync(dst_dev, src, bytes, hipMemcpyDefault, ctx->stream));
        return FV_OK;
    }
    if (p->stage.n < (size_t)(p->n * count))
        FV_TRY(p->stage.alloc(ctx, (size_t)(p->n * count)));
    FV_HIP(ctx, hipMemcpyAsync(p->stage.p, src, bytes, hipMemcpyDefault, ctx->stream));
    hipLaunchKernelGGL(permute_kernel<true>, dim3(fv_blocks(p->n)), dim3(FV_BLOCK), 0, ctx->stream, p->n, count, (const int32_t *)p->perm.p,
                       (const double *)p->stage.p, dst_dev);
    FV_LAUNCH_CHECK(ctx);
    return FV_OK;
}

int fv_free_out(fv_problem *p, double *dst, const double *src_dev, int64_t count)
{
    fv_ctx *ctx = p->ctx;
    const size_t bytes = (size_t)(p->n * count) * sizeof(double);
    if (bytes == 0)
        return FV_OK;
    if (!p->reordered)
        return fv_copy(ctx, dst, src_dev, bytes);
    if (p->stage.n < (size_t)(p->n * count))
        FV_TRY(p->stage.alloc(ctx, (size_t)(p->n * count)));
    hipLaunchKernelGGL(permute_kernel<false>, dim3(fv_blocks(p->n)), dim3(FV_BLOCK), 0, ctx->stream, p->n, count, (const int32_t *)p->perm.p, src_dev,
                       p->stage.p);
    FV_LAUNCH_CHECK(ctx);
    return fv_copy(ctx, dst, p->stage.p, bytes);
}

// Looks at how far apart the two cells of a face are numbered; when that is far worse than a mesh of this size needs
// (mean |i - j| above 2 n^(2/3), the scale of a well-numbered 3-D grid) a reverse Cuthill-McKee order of the free cells is
// computed (host, fv_host_locality_order) and adopted if it at least halves that distance.  The fractures-like 5M-cell
// mesh of the bench — cells numbered at random inside each fracture, as DFN generators leave them — goes from a mean
// distance of ~8e4 to ~5e2, its SpMV from 2.2 to 5 TB/s (profiles/r02_reorder_fractures.log).  Called between
// fv_build_maps and fv_build_symbolic: everything built afterwards is in the new numbering.
static int fv_reorder_free(fv_problem *p)
{
    fv_ctx *ctx = p->ctx;
    const int64_t n = p->n, F = p->F, N = p->N;
    const int mode = ctx->opt_reorder >= 0 ? ctx->opt_reorder : g_reorder; // FV_OPT_REORDER of the context, else the process default
    if (mode == 0 || n < 2 || F < 1 || (mode == 1 && n < 65536))
        return FV_OK;
    const auto t0 = std::chrono::steady_clock::now();
    if (g_reorder_device) { // the same order, built in HBM (fv_reorder.hip); the host routine below stays as the reference and the fall-back
        DevBuf<int32_t> dperm;
        FV_TRY(dperm.alloc(ctx, (size_t)n));
        bool adopted = false, handled = false;
        FV_TRY(fv_device_locality_order(p, mode, dperm.p, &adopted, &handled, &p->reorder_mean_before, &p->reorder_mean_after));
        if (handled) {
            if (!adopted)
                return FV_OK;
            FV_TRY(p->iperm.alloc(ctx, (size_t)n));
            p->perm.swap(dperm);
            hipLaunchKernelGGL(apply_perm_kernel, dim3(fv_blocks(N)), dim3(FV_BLOCK), 0, ctx->stream, N, (const int32_t *)p->perm.p, p->nodemap.p, p->f2n.p,
                               p->iperm.p);
            FV_LAUNCH_CHECK(ctx);
            FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
            p->reordered = true;
            p->reorder_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            return FV_OK;
        }
    }
    std::vector<int32_t> a((size_t)F), b((size_t)F), map((size_t)N);
    FV_HIP(ctx, fv_memcpy_sync(ctx, a.data(), p->node1.p, (size_t)F * sizeof(int32_t), hipMemcpyDeviceToHost));
    FV_HIP(ctx, fv_memcpy_sync(ctx, b.data(), p->node2.p, (size_t)F * sizeof(int32_t), hipMemcpyDeviceToHost));
    FV_HIP(ctx, fv_memcpy_sync(ctx, map.data(), p->nodemap.p, (size_t)N * sizeof(int32_t), hipMemcpyDeviceToHost));
    int64_t m = 0;
    double sum = 0.0;
    for (int64_t k = 0; k < F; k++) { // faces between two free cells, in canonical free indices
        const int32_t fa = map[(size_t)a[(size_t)k]], fb = map[(size_t)b[(size_t)k]];
        if (fa >= 0 && fb >= 0 && fa != fb) {
            a[(size_t)m] = fa;
            b[(size_t)m] = fb;
            sum += fabs((double)fa - (double)fb);
            m++;
        }
    }
    if (m == 0)
        return FV_OK;
    p->reorder_mean_before = sum / (double)m;
    if (mode == 1 && p->reorder_mean_before <= 2.0 * pow((double)n, 2.0 / 3.0))
        return FV_OK; // numbered like a grid (or better): nothing to gain
    std::vector<int32_t> perm((size_t)n);
    double before = 0.0, after = 0.0;
    FV_TRY(fv_host_locality_order(n, m, a.data(), b.data(), perm.data(), &before, &after));
    p->reorder_mean_after = after;
    if (mode == 1 && after * 2.0 >= before)
        return FV_OK;
    FV_TRY(p->perm.alloc(ctx, (size_t)n));
    FV_TRY(p->iperm.alloc(ctx, (size_t)n));
    FV_HIP(ctx, fv_memcpy_sync(ctx, p->perm.p, perm.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(apply_perm_kernel, dim3(fv_blocks(N)), dim3(FV_BLOCK), 0, ctx->stream, N, (const int32_t *)p->perm.p, p->nodemap.p, p->f2n.p,
                       p->iperm.p);
    FV_LAUNCH_CHECK(ctx);
    FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    p->reordered = true;
    p->reorder_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return FV_OK;
}

extern "C" int fv_problem_reorder_info(fv_problem *p, int32_t *reordered, double *mean_before, double *mean_after, double *seconds)
{
    if (!p)
        return FV_ERR_ARG;
    if (reordered)
        *reordered = p->reordered ? 1 : 0;
    if (mean_before)
        *mean_before = p->reorder_mean_before;
    if (mean_after)
        *mean_after = p->reorder_mean_after;
    if (seconds)
        *seconds = p->reorder_seconds;
    return FV_OK;
}

__global__ __launch_bounds__(FV_BLOCK) void export_maps_kernel(const int32_t *__restrict__ nodemap, const int32_t *__restrict__ iperm, int64_t N,
                                                                uint8_t *__restrict__ freenode, int64_t *__restrict__ n2f)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i >= N)
        return;
    const int32_t m = nodemap[i];
    freenode[i] = m >= 0;
    n2f[i] = m >= 0 ? (int64_t)(iperm ? iperm[m] : m) + 1 : -1; // FiniteVolume.jl:35-41: the rank among the free nodes
}

__global__ __launch_bounds__(FV_BLOCK) void export_dirmap_kernel(const int32_t *__restrict__ nodemap, int64_t N,
                                                                  int64_t *__restrict__ n2d)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i < N)
        n2d[i] = nodemap[i] >= 0 ? -1 : (int64_t)(-nodemap[i]); // 1-based position, FiniteVolume.jl:21-24
}

static int export_free_maps(fv_problem *p, uint8_t *freenode, int64_t *nodei2freenodei)
{
    fv_ctx *ctx = p->ctx;
    DevBuf<uint8_t> df;
    DevBuf<int64_t> dm;
    FV_TRY(df.alloc(ctx, (size_t)p->N));
    FV_TRY(dm.alloc(ctx, (size_t)p->N));
    hipLaunchKernelGGL(export_maps_kernel, dim3(fv_blocks(p->N)), dim3(FV_BLOCK), 0, ctx->stream, (const int32_t *)p->nodemap.p,
                       p->reordered ? (const int32_t *)p->iperm.p : (const int32_t *)nullptr, p->N, df.p, dm.p);
    FV_LAUNCH_CHECK(ctx);
    if (freenode)
        FV_TRY(fv_copy(ctx, freenode, df.p, (size_t)p->N));
    if (nodei2freenodei)
        FV_TRY(fv_copy(ctx, nodei2freenodei, dm.p, (size_t)p->N * sizeof(int64_t)));
    return FV_OK;
}

extern "C" int fv_problem_get_free_maps(fv_problem *p, uint8_t *freenode, int64_t *nodei2freenodei)
{
    if (!p)
        return FV_ERR_ARG;
    FV_HIP(p->ctx, hipSetDevice(p->ctx->device));
    return export_free_maps(p, freenode, nodei2freenodei);
}

extern "C" int fv_getfreenodes(fv_ctx *ctx, int64_t N, int64_t ndir, const int64_t *dirichletnodes, uint8_t *freenode,
                               int64_t *nodei2freenodei, int64_t *nfree)
{
    if (!ctx || N < 0 || ndir < 0 || (ndir > 0 && !dirichletnodes))
        return FV_ERR_ARG;
    FV_HIP(ctx, hipSetDevice(ctx->device));
    fv_problem tmp;
    tmp.ctx = ctx;
    tmp.N = N;
    tmp.ndir = ndir;
    FV_TRY(fv_build_maps(&tmp, dirichletnodes));
    if (nfree)
        *nfree = tmp.n;
    return export_free_maps(&tmp, freenode, nodei2freenodei);
}

// first i (in dirichletnodes order) whose node carries a source: FiniteVolume.jl:22-27
__global__ __launch_bounds__(FV_BLOCK) void source_check_kernel(const int32_t *__restrict__ dn, int64_t ndir,
                                                                 const double *__restrict__ sources, int32_t *__restrict__ firstbad)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i < ndir && sources[dn[i]] != 0)
        atomicMin(firstbad, (int32_t)i);
}

// dn_dev: 0-based dirichlet nodes on device.  Returns FV_ERR_SOURCE_AT_DIRICHLET with the reference's message.
static int check_sources(fv_ctx *ctx, const int32_t *dn_dev, int64_t ndir, const double *sources_dev, int64_t *badnode)
{
    if (ndir <= 0)
        return FV_OK;
    DevBuf<int32_t> fb;
    FV_TRY(fb.alloc(ctx, 1));
    const int32_t big = 0x7fffffff;
    FV_HIP(ctx, hipMemcpyAsync(fb.p, &big, sizeof big, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(source_check_kernel, dim3(fv_blocks(ndir)), dim3(FV_BLOCK), 0, ctx->stream, dn_dev, ndir, sources_dev, fb.p);
    FV_LAUNCH_CHECK(ctx);
    int32_t first = big;
    FV_HIP(ctx, hipMemcpyAsync(&first, fb.p, sizeof first, hipMemcpyDeviceToHost, ctx->stream));
    FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (first != big) {
        int32_t node0 = 0;
        FV_HIP(ctx, fv_memcpy_sync(ctx, &node0, dn_dev + first, sizeof node0, hipMemcpyDeviceToHost));
        const long long node = (long long)node0 + 1;
        if (badnode)
            *badnode = node;
        fv_set_error(ctx,
                     "There cannot be a source at a Dirichlet node, but node %lld is a Dirichlet node where a source is located.",
                     node);
        return FV_ERR_SOURCE_AT_DIRICHLET;
    }
    return FV_OK;
}

extern "C" int fv_getnodei2dirichleti(fv_ctx *ctx, int64_t N, const double *sources, int64_t ndir, const int64_t *dirichletnodes,
                                      int64_t *nodei2dirichleti, int64_t *badnode)
{
    if (!ctx || N < 0 || ndir < 0 || (N > 0 && !sources) || (ndir > 0 && !dirichletnodes) || !nodei2dirichleti)
        return FV_ERR_ARG;
    FV_HIP(ctx, hipSetDevice(ctx->device));
    fv_problem tmp;
    tmp.ctx = ctx;
    tmp.N = N;
    tmp.ndir = ndir;
    FV_TRY(fv_build_maps(&tmp, dirichletnodes));
    DevBuf<int64_t> w, out;
    DevBuf<int32_t> dn;
    DevBuf<double> ds;
    FV_TRY(ds.alloc(ctx, (size_t)N));
    FV_HIP(ctx, hipMemcpyAsync(ds.p, sources, (size_t)N * sizeof(double), hipMemcpyDefault, ctx->stream));
    if (ndir > 0) {
        FV_TRY(w.alloc(ctx, (size_t)ndir));
        FV_HIP(ctx, hipMemcpyAsync(w.p, dirichletnodes, (size_t)ndir * sizeof(int64_t), hipMemcpyDefault, ctx->stream));
        FV_TRY(dn.alloc(ctx, (size_t)ndir));
        int bad = 0;
        FV_TRY(fv_narrow_indices(ctx, w.p, dn.p, ndir, 1, N, &bad));
        FV_TRY(check_sources(ctx, dn.p, ndir, ds.p, badnode));
    }
    FV_TRY(out.alloc(ctx, (size_t)N));
    hipLaunchKernelGGL(export_dirmap_kernel, dim3(fv_blocks(N)), dim3(FV_BLOCK), 0, ctx->stream, tmp.nodemap.p, N, out.p);
    FV_LAUNCH_CHECK(ctx);
    return fv_copy(ctx, nodei2dirichleti, out.p, (size_t)N * sizeof(int64_t));
}

// ------------------------------------------------------------------ symbolic phase
// every face puts one incident entry on each free end: FiniteVolume.jl:95-104
__global__ __launch_bounds__(FV_BLOCK) void inc_count_kernel(int64_t F, const int32_t *__restrict__ node1,
                                                              const int32_t *__restrict__ node2,
                                                              const int32_t *__restrict__ nodemap, int32_t *__restrict__ cnt)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i >= F)
        return;
    const int32_t fa = nodemap[node1[i]], fb = nodemap[node2[i]];
    if (fa >= 0)
        atomicAdd(&cnt[fa], 1);
    if (fb >= 0)
        atomicAdd(&cnt[fb], 1);
}

__global__ __launch_bounds__(FV_BLOCK) void inc_fill_kernel(int64_t F, const int32_t *__restrict__ node1,
                                                             const int32_t *__restrict__ node2, const int32_t *__restrict__ nodemap,
                                                             const int32_t *__restrict__ incptr, int32_t *__restrict__ cursor,
                                                             uint32_t *__restrict__ inc_face)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i >= F)
        return;
    const int32_t fa = nodemap[node1[i]], fb = nodemap[node2[i]];
    if (fa >= 0)
        inc_face[incptr[fa] + atomicAdd(&cursor[fa], 1)] = ((uint32_t)i << 1);
    if (fb >= 0)
        inc_face[incptr[fb] + atomicAdd(&cursor[fb], 1)] = ((uint32_t)i << 1) | 1u;
}

__device__ inline int32_t other_free_index(uint32_t code, const int32_t *node1, const int32_t *node2, const int32_t *nodemap)
{
    const uint32_t f = code >> 1;
    return nodemap[(code & 1u) ? node1[f] : node2[f]]; // end 0: this row is node1, the other end is node2
}

// One thread per free row: put the incident list into face order (the atomic fill
// above is unordered) and leave the row's sorted distinct columns in tmpcol.
__global__ __launch_bounds__(FV_BLOCK) void row_structure_kernel(int64_t n, const int32_t *__restrict__ node1,
                                                                  const int32_t *__restrict__ node2,
                                                                  const int32_t *__restrict__ nodemap,
                                                                  const int32_t *__restrict__ incptr, uint32_t *__restrict__ inc_face,
                                                                  int32_t *__restrict__ tmpcol, int32_t *__restrict__ rowcnt)
{
    const int64_t r = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (r >= n)
        return;
    const int32_t s = incptr[r], e = incptr[r + 1];
    for (int32_t a = s + 1; a < e; a++) { // insertion sort by (face, end)
        const uint32_t v = inc_face[a];
        int32_t b = a - 1;
        while (b >= s && inc_face[b] > v) {
            inc_face[b + 1] = inc_face[b];
            b--;
        }
        inc_face[b + 1] = v;
    }
    int32_t *col = tmpcol + (int64_t)s + r; // room for d + 1 candidates
    int32_t m = 0;
    if (e > s)
        col[m++] = (int32_t)r; // every incident face adds to the diagonal
    for (int32_t a = s; a < e; a++) {
        const int32_t fo = other_free_index(inc_face[a], node1, node2, nodemap);
        if (fo >= 0)
            col[m++] = fo;
    }
    for (int32_t a = 1; a < m; a++) {
        const int32_t v = col[a];
        int32_t b = a - 1;
        while (b >= 0 && col[b] > v) {
            col[b + 1] = col[b];
            b--;
        }
        col[b + 1] = v;
    }
    int32_t u = 0;
    for (int32_t a = 0; a < m; a++)
        if (a == 0 || col[a] != col[u - 1])
            col[u++] = col[a];
    rowcnt[r] = u;
}

__global__ __launch_bounds__(FV_BLOCK) void row_slots_kernel(int64_t n, const int32_t *__restrict__ node1,
                                                              const int32_t *__restrict__ node2, const int32_t *__restrict__ nodemap,
                                                              const int32_t *__restrict__ incptr, const uint32_t *__restrict__ inc_face,
                                                              const int32_t *__restrict__ tmpcol, const int32_t *__restrict__ rowptr,
                                                              int32_t *__restrict__ colind, int32_t *__restrict__ diagpos,
                                                              uint32_t *__restrict__ inc_slot, uint8_t *__restrict__ mark)
{
    const int64_t r = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (r >= n)
        return;
    const int32_t s = incptr[r], e = incptr[r + 1];
    const int32_t rs = rowptr[r], len = rowptr[r + 1] - rs;
    const int32_t *col = tmpcol + (int64_t)s + r;
    int32_t dp = -1;
    for (int32_t k = 0; k < len; k++) {
        colind[rs + k] = col[k];
        if (col[k] == (int32_t)r)
            dp = rs + k;
    }
    diagpos[r] = dp;
    for (int32_t a = s; a < e; a++) {
        const int32_t fo = other_free_index(inc_face[a], node1, node2, nodemap);
        if (fo < 0) {
            inc_slot[a] = SLOT_DIRICHLET;
            continue;
        }
        int32_t lo = 0, hi = len; // lower_bound in the sorted row
        while (lo < hi) {
            const int32_t mid = (lo + hi) >> 1;
            if (col[mid] < fo)
                lo = mid + 1;
            else
                hi = mid;
        }
        uint32_t code = (uint32_t)lo;
        if (!mark[rs + lo]) {
            mark[rs + lo] = 1;
            code |= SLOT_FIRST;
        }
        inc_slot[a] = code;
    }
}

int fv_build_symbolic(fv_problem *p)
{
    fv_ctx *ctx = p->ctx;
    const int64_t n = p->n, F = p->F;
    DevBuf<int32_t> cnt, tmpcol, rowcnt;
    FV_TRY(cnt.alloc(ctx, (size_t)n));
    FV_TRY(cnt.zero(ctx));
    if (F > 0) {
        hipLaunchKernelGGL(inc_count_kernel, dim3(fv_blocks(F)), dim3(FV_BLOCK), 0, ctx->stream, F, p->node1.p, p->node2.p,
                           p->nodemap.p, cnt.p);
        FV_LAUNCH_CHECK(ctx);
    }
    FV_TRY(p->incptr.alloc(ctx, (size_t)n + 1));
    FV_TRY(fv_exclusive_scan_i32(ctx, cnt.p, p->incptr.p, n, &p->E));
    FV_TRY(p->inc_face.alloc(ctx, (size_t)p->E));
    FV_TRY(p->inc_slot.alloc(ctx, (size_t)p->E));
    FV_TRY(cnt.zero(ctx));
    if (F > 0) {
        hipLaunchKernelGGL(inc_fill_kernel, dim3(fv_blocks(F)), dim3(FV_BLOCK), 0, ctx->stream, F, p->node1.p, p->node2.p,
                           p->nodemap.p, p->incptr.p, cnt.p, p->inc_face.p);
        FV_LAUNCH_CHECK(ctx);
    }
    FV_TRY(tmpcol.alloc(ctx, (size_t)(p->E + n)));
    FV_TRY(rowcnt.alloc(ctx, (size_t)n));
    if (n > 0) {
        hipLaunchKernelGGL(row_structure_kernel, dim3(fv_blocks(n)), dim3(FV_BLOCK), 0, ctx->stream, n, p->node1.p, p->node2.p,
                           p->nodemap.p, p->incptr.p, p->inc_face.p, tmpcol.p, rowcnt.p);
        FV_LAUNCH_CHECK(ctx);
    }
    FV_TRY(p->rowptr.alloc(ctx, (size_t)n + 1));
    FV_TRY(fv_exclusive_scan_i32(ctx, rowcnt.p, p->rowptr.p, n, &p->nnz));
    FV_TRY(p->colind.alloc(ctx, (size_t)p->nnz + 2)); // +2: the stream SpMV reads entry pairs
    FV_TRY(p->colind.zero(ctx));
    FV_TRY(p->diagpos.alloc(ctx, (size_t)n));
    DevBuf<uint8_t> mark;
    FV_TRY(mark.alloc(ctx, (size_t)p->nnz));
    FV_TRY(mark.zero(ctx));
    if (n > 0) {
        hipLaunchKernelGGL(row_slots_kernel, dim3(fv_blocks(n)), dim3(FV_BLOCK), 0, ctx->stream, n, p->node1.p, p->node2.p,
                           p->nodemap.p, p->incptr.p, p->inc_face.p, tmpcol.p, p->rowptr.p, p->colind.p, p->diagpos.p,
                           p->inc_slot.p, mark.p);
        FV_LAUNCH_CHECK(ctx);
    }
    FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FV_OK;
}

// ------------------------------------------------------------------ numeric phase (a5, a6)
// c_i = K[metaindex(i)] * aol[i]  or  exp(K[metaindex(i)]) * aol[i]   (FiniteVolume.jl:83, :96)
__global__ __launch_bounds__(FV_BLOCK) void conductance_kernel(int64_t F, int64_t nK, const double *__restrict__ K,
                                                                const int64_t *__restrict__ metaindex,
                                                                const double *__restrict__ aol, int logt,
                                                                double *__restrict__ cond, int *__restrict__ bad)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i >= F)
        return;
    int64_t m = metaindex ? metaindex[i] - 1 : (nK == 1 ? 0 : i);
    if (m < 0 || m >= nK) {
        *bad = 1;
        m = 0;
    }
    const double k = K[m];
    cond[i] = logt ? exp(k) * aol[i] : k * aol[i];
}

// One thread per free row.  Contributions are folded in face order, which is the
// order sparse(I,J,V,n,n,+) combines repeated (row,col) pairs in, and the order
// assembleb's `b[...] +=` statements run in.
__global__ __launch_bounds__(FV_BLOCK) void assemble_rows_kernel(
    int64_t n, const int32_t *__restrict__ incptr, const uint32_t *__restrict__ inc_face, const uint32_t *__restrict__ inc_slot,
    const int32_t *__restrict__ node1, const int32_t *__restrict__ node2, const int32_t *__restrict__ nodemap,
    const int32_t *__restrict__ rowptr, const int32_t *__restrict__ diagpos, const int32_t *__restrict__ f2n,
    const double *__restrict__ cond, const double *__restrict__ sources, const double *__restrict__ dheads,
    double *__restrict__ vals, double *__restrict__ b, double *__restrict__ diagA)
{
    const int64_t r = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (r >= n)
        return;
    const int32_t s = incptr[r], e = incptr[r + 1];
    const int32_t rs = rowptr[r], dp = diagpos[r];
    double dacc = 0.0, bacc = sources[f2n[r]]; // FiniteVolume.jl:113-120
    for (int32_t a = s; a < e; a++) {
        const uint32_t code = inc_face[a];
        const uint32_t f = code >> 1;
        const double c = cond[f];
        dacc = (a == s) ? c : dacc + c; // (row,row,+c): FiniteVolume.jl:96,98,101,103
        const uint32_t sl = inc_slot[a];
        if (sl == SLOT_DIRICHLET) { // exactly one Dirichlet end: FiniteVolume.jl:131-134
            const int32_t other = nodemap[(code & 1u) ? node1[f] : node2[f]];
            bacc += c * dheads[-other - 1];
        } else { // (row,other,-c): FiniteVolume.jl:97,99
            const int32_t k = rs + (int32_t)(sl & 0x7fffffffu);
            if (k == dp)
                dacc = dacc + (-c); // self-loop face
            else
                vals[k] = (sl & SLOT_FIRST) ? -c : vals[k] + (-c);
        }
    }
    if (dp >= 0)
        vals[dp] = dacc;
    diagA[r] = dacc;
    b[r] = bacc;
}

int fv_face_arrays(fv_problem *p, FaceArrays &fa)
{
    fv_ctx *ctx = p->ctx;
    if (!p->lean) {
        fa.node1 = p->node1.p;
        fa.node2 = p->node2.p;
        fa.cond = p->cond.p;
        fa.aol = p->aol.p;
        return FV_OK;
    }
    if (p->F > 0x7fffffffLL) {
        fv_set_error(ctx, "F=%lld exceeds the int32 range of the face generator", (long long)p->F);
        return FV_ERR_TOO_LARGE;
    }
    if (!p->assembled || !p->lean_K.p) {
        fv_set_error(ctx, "call fv_assemble first");
        return FV_ERR_STATE;
    }
    FV_TRY(fa.t1.alloc(ctx, (size_t)p->F));
    FV_TRY(fa.t2.alloc(ctx, (size_t)p->F));
    FV_TRY(fa.ta.alloc(ctx, (size_t)p->F));
    FV_TRY(fa.tc.alloc(ctx, (size_t)p->F));
    FV_TRY(fv_grid_generate_device(ctx, p->lean_mins, p->lean_maxs, p->ns, fa.t1.p, fa.t2.p, fa.ta.p, nullptr, nullptr));
    DevBuf<int> dbad;
    FV_TRY(dbad.alloc(ctx, 1));
    FV_TRY(dbad.zero(ctx));
    if (p->F > 0)
        hipLaunchKernelGGL(conductance_kernel, dim3(fv_blocks(p->F)), dim3(FV_BLOCK), 0, ctx->stream, p->F, p->lean_nK, (const double *)p->lean_K.p,
                           (const int64_t *)p->lean_meta.p, (const double *)fa.ta.p, p->lean_logt, fa.tc.p, dbad.p);
    FV_LAUNCH_CHECK(ctx);
    fa.node1 = fa.t1.p;
    fa.node2 = fa.t2.p;
    fa.aol = fa.ta.p;
    fa.cond = fa.tc.p;
    return FV_OK;
}

// metaindex entries outside 1:nK?
__global__ __launch_bounds__(FV_BLOCK) void meta_range_kernel(int64_t F, int64_t nK, const int64_t *__restrict__ metaindex, int *__restrict__ bad)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i < F && (metaindex[i] < 1 || metaindex[i] > nK))
        *bad = 1;
}

// fv_assemble of a lean problem: the conductivities stay as handed over (the rows are formed from them whenever a storage form
// is filled), b and the diagonal come from one pass over the rows (fv_lean.hip)
static int lean_assemble(fv_problem *p, int64_t nK, const double *conductivities, const int64_t *metaindex, int logtransformconductivity,
                         const double *sources, const double *dirichletheads, int64_t *badnode)
{
    fv_ctx *ctx = p->ctx;
    if (nK < 1) {
        fv_set_error(ctx, "BoundsError: no conductivities");
        return FV_ERR_INDEX;
    }
    DevBuf<double> dsrc;
    FV_TRY(dsrc.alloc(ctx, (size_t)p->N));
    FV_HIP(ctx, hipMemcpyAsync(dsrc.p, sources, (size_t)p->N * sizeof(double), hipMemcpyDefault, ctx->stream));
    if (p->ndir > 0)
        FV_HIP(ctx, hipMemcpyAsync(p->dheads.p, dirichletheads, (size_t)p->ndir * sizeof(double), hipMemcpyDefault, ctx->stream));
    FV_TRY(check_sources(ctx, p->dnodes0.p, p->ndir, dsrc.p, badnode));
    const int64_t keep = (metaindex || nK == 1) ? nK : p->F; // (without a metaindex the first F entries are the faces')
    if ((int64_t)p->lean_K.n != keep)
        FV_TRY(p->lean_K.alloc(ctx, (size_t)keep));
    FV_HIP(ctx, hipMemcpyAsync(p->lean_K.p, conductivities, (size_t)keep * sizeof(double), hipMemcpyDefault, ctx->stream));
    if (metaindex) {
        if ((int64_t)p->lean_meta.n != p->F)
            FV_TRY(p->lean_meta.alloc(ctx, (size_t)p->F));
        FV_HIP(ctx, hipMemcpyAsync(p->lean_meta.p, metaindex, (size_t)p->F * sizeof(int64_t), hipMemcpyDefault, ctx->stream));
        DevBuf<int> dbad;
        FV_TRY(dbad.alloc(ctx, 1));
        FV_TRY(dbad.zero(ctx));
        hipLaunchKernelGGL(meta_range_kernel, dim3(fv_blocks(p->F)), dim3(FV_BLOCK), 0, ctx->stream, p->F, nK, (const int64_t *)p->lean_meta.p, dbad.p);
        FV_LAUNCH_CHECK(ctx);
        int bad = 0;
        FV_TRY(fv_copy(ctx, &bad, dbad.p, sizeof bad));
        if (bad) {
            p->lean_meta.release();
            p->assembled = false;
            fv_set_error(ctx, "BoundsError: metaindex outside 1:%lld", (long long)nK);
            return FV_ERR_INDEX;
        }
    } else
        p->lean_meta.release();
    p->lean_nK = nK;
    p->lean_logt = logtransformconductivity ? 1 : 0;
    FV_TRY(fv_lean_assemble(p, dsrc.p));
    FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    p->assembled = true;
    p->assemble_epoch++;
    return FV_OK;
}

extern "C" int fv_assemble(fv_problem *p, int64_t nK, const double *conductivities, const int64_t *metaindex,
                           int logtransformconductivity, const double *sources, const double *dirichletheads, int64_t *badnode)
{
    if (!p || nK < 0 || (nK > 0 && !conductivities) || (p->N > 0 && !sources) || (p->ndir > 0 && !dirichletheads))
        return FV_ERR_ARG;
    fv_ctx *ctx = p->ctx;
    FV_HIP(ctx, hipSetDevice(ctx->device));
    if (p->from_csc) {
        fv_set_error(ctx, "fv_assemble: this problem was created from a CSC matrix and has no faces");
        return FV_ERR_STATE;
    }
    if (!metaindex && nK != 1 && nK < p->F) {
        fv_set_error(ctx, "BoundsError: %lld conductivities for %lld faces", (long long)nK, (long long)p->F);
        return FV_ERR_INDEX;
    }
    if (p->lean)
        return lean_assemble(p, nK, conductivities, metaindex, logtransformconductivity, sources, dirichletheads, badnode);
    DevBuf<double> dK, dsrc;
    DevBuf<int64_t> dmeta;
    FV_TRY(dK.alloc(ctx, (size_t)nK));
    if (nK > 0)
        FV_HIP(ctx, hipMemcpyAsync(dK.p, conductivities, (size_t)nK * sizeof(double), hipMemcpyDefault, ctx->stream));
    FV_TRY(dsrc.alloc(ctx, (size_t)p->N));
    FV_HIP(ctx, hipMemcpyAsync(dsrc.p, sources, (size_t)p->N * sizeof(double), hipMemcpyDefault, ctx->stream));
    if (p->ndir > 0)
        FV_HIP(ctx, hipMemcpyAsync(p->dheads.p, dirichletheads, (size_t)p->ndir * sizeof(double), hipMemcpyDefault, ctx->stream));
    if (metaindex && p->F > 0) {
        FV_TRY(dmeta.alloc(ctx, (size_t)p->F));
        FV_HIP(ctx, hipMemcpyAsync(dmeta.p, metaindex, (size_t)p->F * sizeof(int64_t), hipMemcpyDefault, ctx->stream));
    }
    // getnodei2dirichleti's validation (FiniteVolume.jl:25-27), run by assembleb at :111
    FV_TRY(check_sources(ctx, p->dnodes0.p, p->ndir, dsrc.p, badnode));
    if (p->F > 0) {
        DevBuf<int> dbad;
        FV_TRY(dbad.alloc(ctx, 1));
        FV_TRY(dbad.zero(ctx));
        hipLaunchKernelGGL(conductance_kernel, dim3(fv_blocks(p->F)), dim3(FV_BLOCK), 0, ctx->stream, p->F, nK, dK.p,
                           metaindex ? dmeta.p : nullptr, p->aol.p, logtransformconductivity, p->cond.p, dbad.p);
        FV_LAUNCH_CHECK(ctx);
        int bad = 0;
        FV_HIP(ctx, hipMemcpyAsync(&bad, dbad.p, sizeof bad, hipMemcpyDeviceToHost, ctx->stream));
        FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (bad) {
            fv_set_error(ctx, "BoundsError: metaindex outside 1:%lld", (long long)nK);
            return FV_ERR_INDEX;
        }
    }
    if (p->n > 0) {
        hipLaunchKernelGGL(assemble_rows_kernel, dim3(fv_blocks(p->n)), dim3(FV_BLOCK), 0, ctx->stream, p->n, p->incptr.p,
                           p->inc_face.p, p->inc_slot.p, p->node1.p, p->node2.p, p->nodemap.p, p->rowptr.p, p->diagpos.p,
                           p->f2n.p, p->cond.p, dsrc.p, p->dheads.p, p->vals.p, p->b.p, p->diagA.p);
        FV_LAUNCH_CHECK(ctx);
    }
    FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    p->assembled = true;
    p->assemble_epoch++;
    return FV_OK;
}

// ------------------------------------------------------------------ problem lifecycle
static int finish_problem(fv_problem *p, const int64_t *dirichletnodes, bool face_list = false)
{
    fv_ctx *ctx = p->ctx;
    FV_TRY(fv_build_maps(p, dirichletnodes));
    if (face_list) // a mesh given as a face list may be numbered any way; regulargrid's numbering is the structured kernels' own
        FV_TRY(fv_reorder_free(p));
    // keep the Dirichlet nodes (0-based, caller order) for the source validation
    FV_TRY(p->dnodes0.alloc(ctx, (size_t)p->ndir));
    if (p->ndir > 0) {
        DevBuf<int64_t> w;
        FV_TRY(w.alloc(ctx, (size_t)p->ndir));
        FV_HIP(ctx, hipMemcpyAsync(w.p, dirichletnodes, (size_t)p->ndir * sizeof(int64_t), hipMemcpyDefault, ctx->stream));
        int bad = 0;
        FV_TRY(fv_narrow_indices(ctx, w.p, p->dnodes0.p, p->ndir, 1, p->N, &bad));
    }
    FV_TRY(fv_build_symbolic(p));
    FV_TRY(p->vals.alloc(ctx, (size_t)p->nnz + 2));
    FV_TRY(p->vals.zero(ctx));
    FV_TRY(p->b.alloc(ctx, (size_t)p->n));
    FV_TRY(p->diagA.alloc(ctx, (size_t)p->n));
    FV_TRY(p->cond.alloc(ctx, (size_t)p->F));
    FV_TRY(p->dheads.alloc(ctx, (size_t)p->ndir));
    return FV_OK;
}

extern "C" int fv_problem_create(fv_ctx *ctx, int64_t N, int64_t F, const int64_t *node1, const int64_t *node2,
                                 const double *areasoverlengths, int64_t ndir, const int64_t *dirichletnodes, fv_problem **out)
{
    if (!ctx || !out || N < 0 || F < 0 || ndir < 0 || (F > 0 && (!node1 || !node2 || !areasoverlengths)) ||
        (ndir > 0 && !dirichletnodes))
        return FV_ERR_ARG;
    *out = nullptr;
    FV_HIP(ctx, hipSetDevice(ctx->device));
    if (N > 0x7fffffffLL || F > 0x7fffffffLL) {
        fv_set_error(ctx, "N=%lld / F=%lld exceed the int32 device index range", (long long)N, (long long)F);
        return FV_ERR_TOO_LARGE;
    }
    fv_problem *p = new fv_problem();
    p->ctx = ctx;
    p->N = N;
    p->F = F;
    p->ndir = ndir;
    int rc = FV_OK;
    do {
        if ((rc = p->node1.alloc(ctx, (size_t)F)) || (rc = p->node2.alloc(ctx, (size_t)F)) || (rc = p->aol.alloc(ctx, (size_t)F)))
            break;
        if (F > 0) {
            DevBuf<int64_t> w;
            if ((rc = w.alloc(ctx, (size_t)F)))
                break;
            int bad1 = 0, bad2 = 0;
            if (hipMemcpyAsync(w.p, node1, (size_t)F * sizeof(int64_t), hipMemcpyDefault, ctx->stream) != hipSuccess) {
                fv_set_error(ctx, "copy of neighbors failed");
                rc = FV_ERR_HIP;
                break;
            }
            if ((rc = fv_narrow_indices(ctx, w.p, p->node1.p, F, 1, N, &bad1)))
                break;
            if (hipMemcpyAsync(w.p, node2, (size_t)F * sizeof(int64_t), hipMemcpyDefault, ctx->stream) != hipSuccess) {
                fv_set_error(ctx, "copy of neighbors failed");
                rc = FV_ERR_HIP;
                break;
            }
            if ((rc = fv_narrow_indices(ctx, w.p, p->node2.p, F, 1, N, &bad2)))
                break;
            if (bad1 || bad2) {
                fv_set_error(ctx, "BoundsError: neighbor index outside 1:%lld", (long long)N);
                rc = FV_ERR_INDEX;
                break;
            }
            if ((rc = fv_copy(ctx, p->aol.p, areasoverlengths, (size_t)F * sizeof(double))))
                break;
        }
        rc = finish_problem(p, dirichletnodes, true);
    } while (0);
    if (rc != FV_OK) {
        delete p;
        return rc;
    }
    *out = p;
    return FV_OK;
}

extern "C" int fv_problem_create_regulargrid(fv_ctx *ctx, const double mins[3], const double maxs[3], const int64_t ns[3],
                                             int64_t ndir, const int64_t *dirichletnodes, fv_problem **out)
{
    if (!ctx || !out || !mins || !maxs || !ns || ndir < 0 || (ndir > 0 && !dirichletnodes))
        return FV_ERR_ARG;
    *out = nullptr;
    FV_HIP(ctx, hipSetDevice(ctx->device));
    int64_t N, F;
    FV_TRY(fv_regulargrid_sizes(ns, &N, &F));
    // FV_OPT_LEAN_SETUP: no face arrays, no incident lists, no CSR — the solver's storage forms come straight from the grid (fv_lean.hip).
    // 1: wherever the sliced forms apply (>= 4096 cells); 2 [default]: where the CSR's int32 offsets would not hold the operator (nnz <= 7 N)
    const bool lean = N >= 4096 && N <= 0x7fffffffLL && (ctx->opt_lean == 1 || (ctx->opt_lean == 2 && (F > 0x7fffffffLL || 7 * N > 0x7fffffffLL - 2)));
    if (F > 0x7fffffffLL && !lean) {
        fv_set_error(ctx, "F=%lld exceeds the int32 device index range", (long long)F);
        return FV_ERR_TOO_LARGE;
    }
    fv_problem *p = new fv_problem();
    p->ctx = ctx;
    p->N = N;
    p->F = F;
    p->ndir = ndir;
    p->from_grid = true;
    for (int d = 0; d < 3; d++)
        p->ns[d] = ns[d];
    int rc;
    if (lean) {
        p->lean = true;
        if ((rc = p->gridvol.alloc(ctx, (size_t)N)) ||
            (rc = fv_grid_generate_device(ctx, mins, maxs, ns, nullptr, nullptr, nullptr, p->gridvol.p, nullptr)) ||
            (rc = fv_lean_finish(p, dirichletnodes, mins, maxs))) {
            delete p;
            return rc;
        }
        *out = p;
        return FV_OK;
    }
    if ((rc = p->node1.alloc(ctx, (size_t)F)) || (rc = p->node2.alloc(ctx, (size_t)F)) || (rc = p->aol.alloc(ctx, (size_t)F)) ||
        (rc = p->gridvol.alloc(ctx, (size_t)N)) ||
        (rc = fv_grid_generate_device(ctx, mins, maxs, ns, p->node1.p, p->node2.p, p->aol.p, p->gridvol.p, nullptr)) ||
        (rc = finish_problem(p, dirichletnodes))) {
        delete p;
        return rc;
    }
    *out = p;
    return FV_OK;
}

// The problem of ONE slab of a structured grid, for row-block runs that never hold the global operator: node arrays
// (maps, volumes) are global, but only the faces emitted by the cells of the planes [i1_lo - 1, i1_hi) exist — every face
// incident to a cell of the planes [i1_lo, i1_hi), in the order of the global face list.  After fv_assemble the free rows of
// those planes are complete (bit for bit the rows of the global operator); the others are empty or partial and must not
// be used: hand the problem to fv_dist_setup_bounds with bounds that follow the same planes.
extern "C" int fv_problem_create_regulargrid_slab(fv_ctx *ctx, const double mins[3], const double maxs[3], const int64_t ns[3],
                                                  int64_t ndir, const int64_t *dirichletnodes, int64_t i1_lo, int64_t i1_hi,
                                                  fv_problem **out)
{
    if (!ctx || !out || !mins || !maxs || !ns || ndir < 0 || (ndir > 0 && !dirichletnodes))
        return FV_ERR_ARG;
    *out = nullptr;
    FV_HIP(ctx, hipSetDevice(ctx->device));
    int64_t N, F;
    FV_TRY(fv_regulargrid_sizes(ns, &N, &F));
    if (i1_lo < 0 || i1_hi > ns[0] || i1_lo >= i1_hi) {
        fv_set_error(ctx, "slab planes [%lld, %lld) outside the grid's %lld", (long long)i1_lo, (long long)i1_hi, (long long)ns[0]);
        return FV_ERR_ARG;
    }
    const int64_t gen_lo = i1_lo > 0 ? i1_lo - 1 : 0; // the x-faces into plane i1_lo are emitted by the cells of the plane before it
    const int64_t Fl = fv_grid_face_offset(ns, i1_hi) - fv_grid_face_offset(ns, gen_lo);
    if (Fl > 0x7fffffffLL) {
        fv_set_error(ctx, "F=%lld exceeds the int32 device index range", (long long)Fl);
        return FV_ERR_TOO_LARGE;
    }
    fv_problem *p = new fv_problem();
    p->ctx = ctx;
    p->N = N;
    p->F = Fl;
    p->ndir = ndir;
    p->from_grid = true;
    p->slab_lo = i1_lo;
    p->slab_hi = i1_hi;
    for (int d = 0; d < 3; d++)
        p->ns[d] = ns[d];
    int rc;
    if ((rc = p->node1.alloc(ctx, (size_t)Fl)) || (rc = p->node2.alloc(ctx, (size_t)Fl)) || (rc = p->aol.alloc(ctx, (size_t)Fl)) ||
        (rc = p->gridvol.alloc(ctx, (size_t)N)) ||
        (rc = fv_grid_generate_device(ctx, mins, maxs, ns, p->node1.p, p->node2.p, p->aol.p, p->gridvol.p, nullptr, gen_lo, i1_hi)) ||
        (rc = finish_problem(p, dirichletnodes))) {
        delete p;
        return rc;
    }
    *out = p;
    return FV_OK;
}

// number of free cells with a (0-based) node index below `node`: the first free row of the plane that starts there
extern "C" int fv_problem_free_rows_before(fv_problem *p, int64_t node, int64_t *rows)
{
    if (!p || !rows || node < 0 || node > p->N)
        return FV_ERR_ARG;
    fv_ctx *ctx = p->ctx;
    FV_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<int32_t> chunk(4096);
    for (int64_t at = node; at < p->N; at += (int64_t)chunk.size()) {
        const int64_t m = p->N - at < (int64_t)chunk.size() ? p->N - at : (int64_t)chunk.size();
        FV_HIP(ctx, fv_memcpy_sync(ctx, chunk.data(), p->nodemap.p + at, (size_t)m * sizeof(int32_t), hipMemcpyDeviceToHost));
        for (int64_t k = 0; k < m; k++)
            if (chunk[(size_t)k] >= 0) { // the first free cell at or after `node`: its (canonical) free index is the answer
                int32_t idx = chunk[(size_t)k];
                if (p->reordered) // nodemap holds the internal index: callers only ever see the rank among the free nodes
                    FV_HIP(ctx, fv_memcpy_sync(ctx, &idx, p->iperm.p + idx, sizeof idx, hipMemcpyDeviceToHost));
                *rows = idx;
                return FV_OK;
            }
    }
    *rows = p->n;
    return FV_OK;
}

extern "C" void fv_problem_destroy(fv_problem *p)
{
    if (!p)
        return;
    (void)hipSetDevice(p->ctx->device);
    (void)hipDeviceSynchronize();
    fv_detach_dependents(p);
    delete p;
}

extern "C" int fv_problem_sizes(fv_problem *p, int64_t *N, int64_t *F, int64_t *n, int64_t *nnz)
{
    if (!p)
        return FV_ERR_ARG;
    if (N) *N = p->N;
    if (F) *F = p->F;
    if (n) *n = p->n;
    if (nnz) *nnz = p->nnz;
    return FV_OK;
}

extern "C" int fv_problem_get_grid(fv_problem *p, int64_t *node1, int64_t *node2, double *areasoverlengths, double *volumes)
{
    if (!p)
        return FV_ERR_ARG;
    fv_ctx *ctx = p->ctx;
    FV_HIP(ctx, hipSetDevice(ctx->device));
    if (p->from_csc) {
        fv_set_error(ctx, "fv_problem_get_grid: problem has no faces");
        return FV_ERR_STATE;
    }
    if (p->lean && (node1 || node2 || areasoverlengths)) { // the face list is not kept: generated again, handed out, released
        if (p->F > 0x7fffffffLL) {
            fv_set_error(ctx, "fv_problem_get_grid: F=%lld exceeds the int32 range of the face generator", (long long)p->F);
            return FV_ERR_TOO_LARGE;
        }
        DevBuf<int32_t> t1, t2;
        DevBuf<double> ta;
        DevBuf<int64_t> w;
        FV_TRY(t1.alloc(ctx, (size_t)p->F));
        FV_TRY(t2.alloc(ctx, (size_t)p->F));
        FV_TRY(ta.alloc(ctx, (size_t)p->F));
        FV_TRY(fv_grid_generate_device(ctx, p->lean_mins, p->lean_maxs, p->ns, t1.p, t2.p, ta.p, nullptr, nullptr));
        FV_TRY(w.alloc(ctx, (size_t)p->F));
        if (node1) {
            FV_TRY(fv_widen_indices(ctx, t1.p, w.p, p->F, 1));
            FV_TRY(fv_copy(ctx, node1, w.p, (size_t)p->F * sizeof(int64_t)));
        }
        if (node2) {
            FV_TRY(fv_widen_indices(ctx, t2.p, w.p, p->F, 1));
            FV_TRY(fv_copy(ctx, node2, w.p, (size_t)p->F * sizeof(int64_t)));
        }
        if (areasoverlengths)
            FV_TRY(fv_copy(ctx, areasoverlengths, ta.p, (size_t)p->F * sizeof(double)));
        node1 = node2 = nullptr;
        areasoverlengths = nullptr;
    }
    if (node1 || node2) {
        DevBuf<int64_t> w;
        FV_TRY(w.alloc(ctx, (size_t)p->F));
        if (node1) {
            FV_TRY(fv_widen_indices(ctx, p->node1.p, w.p, p->F, 1));
            FV_TRY(fv_copy(ctx, node1, w.p, (size_t)p->F * sizeof(int64_t)));
        }
        if (node2) {
            FV_TRY(fv_widen_indices(ctx, p->node2.p, w.p, p->F, 1));
            FV_TRY(fv_copy(ctx, node2, w.p, (size_t)p->F * sizeof(int64_t)));
        }
    }
    if (areasoverlengths)
        FV_TRY(fv_copy(ctx, areasoverlengths, p->aol.p, (size_t)p->F * sizeof(double)));
    if (volumes) {
        if (!p->from_grid) {
            fv_set_error(ctx, "fv_problem_get_grid: volumes are only known for regulargrid-created problems");
            return FV_ERR_STATE;
        }
        FV_TRY(fv_copy(ctx, volumes, p->gridvol.p, (size_t)p->N * sizeof(double)));
    }
    return FV_OK;
}

// ------------------------------------------------------------------ exports
// A re-numbered problem exports the matrix in the caller's (canonical) numbering: column i of the CSC is internal row
// perm[i], its entries go back through iperm and are put in ascending canonical order — the same arrays, bit for bit, the
// un-numbered problem would have produced (values are folded per entry in face order, which no numbering changes).
__global__ __launch_bounds__(FV_BLOCK) void canon_len_kernel(int64_t n, const int32_t *__restrict__ perm, const int32_t *__restrict__ rowptr,
                                                              int32_t *__restrict__ len)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i < n) {
        const int32_t r = perm[i];
        len[i] = rowptr[r + 1] - rowptr[r];
    }
}

__global__ __launch_bounds__(FV_BLOCK) void canon_rows_kernel(int64_t n, const int32_t *__restrict__ perm, const int32_t *__restrict__ iperm,
                                                               const int32_t *__restrict__ rowptr, const int32_t *__restrict__ colind,
                                                               const double *__restrict__ vals, const int32_t *__restrict__ start,
                                                               int64_t *__restrict__ colptr_out, int64_t *__restrict__ rowval_out,
                                                               double *__restrict__ nzval_out)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i > n)
        return;
    if (colptr_out)
        colptr_out[i] = (int64_t)start[i] + 1;
    if (i == n || !rowval_out)
        return;
    const int32_t r = perm[i], s0 = rowptr[r], len = rowptr[r + 1] - s0;
    const int64_t o = start[i];
    for (int32_t k = 0; k < len; k++) { // insertion sort by canonical column (rows are short)
        const int64_t c = (int64_t)iperm[colind[s0 + k]] + 1;
        const double v = vals ? vals[s0 + k] : 0.0;
        int32_t j = k;
        while (j > 0 && rowval_out[o + j - 1] > c) {
            rowval_out[o + j] = rowval_out[o + j - 1];
            if (nzval_out)
                nzval_out[o + j] = nzval_out[o + j - 1];
            j--;
        }
        rowval_out[o + j] = c;
        if (nzval_out)
            nzval_out[o + j] = v;
    }
}

static int get_csc_canonical(fv_problem *p, int64_t *colptr, int64_t *rowval, double *nzval)
{
    fv_ctx *ctx = p->ctx;
    const int64_t n = p->n, nnz = p->nnz;
    DevBuf<int32_t> len, start;
    DevBuf<int64_t> cp, rv;
    DevBuf<double> nz;
    FV_TRY(len.alloc(ctx, (size_t)n));
    FV_TRY(start.alloc(ctx, (size_t)n + 1));
    hipLaunchKernelGGL(canon_len_kernel, dim3(fv_blocks(n)), dim3(FV_BLOCK), 0, ctx->stream, n, (const int32_t *)p->perm.p, (const int32_t *)p->rowptr.p,
                       len.p);
    FV_LAUNCH_CHECK(ctx);
    int64_t total = 0;
    FV_TRY(fv_exclusive_scan_i32(ctx, len.p, start.p, n, &total));
    FV_TRY(cp.alloc(ctx, (size_t)n + 1));
    FV_TRY(rv.alloc(ctx, (size_t)nnz));
    if (nzval)
        FV_TRY(nz.alloc(ctx, (size_t)nnz));
    hipLaunchKernelGGL(canon_rows_kernel, dim3(fv_blocks(n + 1)), dim3(FV_BLOCK), 0, ctx->stream, n, (const int32_t *)p->perm.p,
                       (const int32_t *)p->iperm.p, (const int32_t *)p->rowptr.p, (const int32_t *)p->colind.p,
                       nzval ? (const double *)p->vals.p : (const double *)nullptr, (const int32_t *)start.p, cp.p, rv.p, nzval ? nz.p : (double *)nullptr);
    FV_LAUNCH_CHECK(ctx);
    if (colptr)
        FV_TRY(fv_copy(ctx, colptr, cp.p, (size_t)(n + 1) * sizeof(int64_t)));
    if (rowval)
        FV_TRY(fv_copy(ctx, rowval, rv.p, (size_t)nnz * sizeof(int64_t)));
    if (nzval)
        FV_TRY(fv_copy(ctx, nzval, nz.p, (size_t)nnz * sizeof(double)));
    return FV_OK;
}

extern "C" int fv_get_csc(fv_problem *p, int64_t *colptr, int64_t *rowval, double *nzval)
{
    if (!p)
        return FV_ERR_ARG;
    fv_ctx *ctx = p->ctx;
    FV_HIP(ctx, hipSetDevice(ctx->device));
    if (nzval && !p->assembled) {
        fv_set_error(ctx, "fv_get_csc: call fv_assemble first");
        return FV_ERR_STATE;
    }
    if (p->lean) // (written out from the rows formed on the fly, a window at a time: fv_lean.hip)
        return fv_lean_get_csc(p, colptr, rowval, nzval);
    if (p->reordered)
        return get_csc_canonical(p, colptr, rowval, nzval);
    DevBuf<int64_t> w;
    if (colptr) {
        FV_TRY(w.alloc(ctx, (size_t)p->n + 1));
        FV_TRY(fv_widen_indices(ctx, p->rowptr.p, w.p, p->n + 1, 1));
        FV_TRY(fv_copy(ctx, colptr, w.p, (size_t)(p->n + 1) * sizeof(int64_t)));
    }
    if (rowval) {
        FV_TRY(w.alloc(ctx, (size_t)p->nnz));
        FV_TRY(fv_widen_indices(ctx, p->colind.p, w.p, p->nnz, 1));
        FV_TRY(fv_copy(ctx, rowval, w.p, (size_t)p->nnz * sizeof(int64_t)));
    }
    if (nzval)
        FV_TRY(fv_copy(ctx, nzval, p->vals.p, (size_t)p->nnz * sizeof(double)));
    return FV_OK;
}

extern "C" int fv_get_b(fv_problem *p, double *b)
{
    if (!p || !b)
        return FV_ERR_ARG;
    FV_HIP(p->ctx, hipSetDevice(p->ctx->device));
    if (!p->assembled) {
        fv_set_error(p->ctx, "fv_get_b: call fv_assemble first");
        return FV_ERR_STATE;
    }
    return fv_free_out(p, b, p->b.p);
}

// freenodes2nodes: FiniteVolume.jl:146-153
__global__ __launch_bounds__(FV_BLOCK) void scatter_nodes_kernel(int64_t N, const int32_t *__restrict__ nodemap,
                                                                  const double *__restrict__ ufree, const double *__restrict__ dheads,
                                                                  double *__restrict__ head)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i >= N)
        return;
    const int32_t m = nodemap[i];
    head[i] = m >= 0 ? ufree[m] : dheads[-m - 1];
}

__global__ __launch_bounds__(FV_BLOCK) void gather_free_kernel(int64_t n, const int32_t *__restrict__ f2n,
                                                                const double *__restrict__ unodes, double *__restrict__ ufree)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i < n)
        ufree[i] = unodes[f2n[i]];
}

int fv_scatter_nodes(fv_problem *p, const double *ufree_dev, double *head_dev)
{
    if (p->N > 0) {
        hipLaunchKernelGGL(scatter_nodes_kernel, dim3(fv_blocks(p->N)), dim3(FV_BLOCK), 0, p->ctx->stream, p->N, p->nodemap.p,
                           ufree_dev, p->dheads.p, head_dev);
        FV_LAUNCH_CHECK(p->ctx);
    }
    return FV_OK;
}

int fv_gather_free(fv_problem *p, const double *unodes_dev, double *ufree_dev)
{
    if (p->n > 0) {
        hipLaunchKernelGGL(gather_free_kernel, dim3(fv_blocks(p->n)), dim3(FV_BLOCK), 0, p->ctx->stream, p->n, p->f2n.p, unodes_dev,
                           ufree_dev);
        FV_LAUNCH_CHECK(p->ctx);
    }
    return FV_OK;
}

extern "C" int fv_freenodes2nodes(fv_problem *p, const double *result_free, double *head_nodes)
{
    if (!p || !result_free || !head_nodes)
        return FV_ERR_ARG;
    fv_ctx *ctx = p->ctx;
    FV_HIP(ctx, hipSetDevice(ctx->device));
    if (!p->assembled) {
        fv_set_error(ctx, "fv_freenodes2nodes: call fv_assemble first (Dirichlet heads unknown)");
        return FV_ERR_STATE;
    }
    DevBuf<double> uf, hd;
    FV_TRY(uf.alloc(ctx, (size_t)p->n));
    FV_TRY(hd.alloc(ctx, (size_t)p->N));
    FV_TRY(fv_free_in(p, uf.p, result_free));
    FV_TRY(fv_scatter_nodes(p, uf.p, hd.p));
    return fv_copy(ctx, head_nodes, hd.p, (size_t)p->N * sizeof(double));
}

// ------------------------------------------------------------------ general symmetric operator from CSC
__global__ __launch_bounds__(FV_BLOCK) void csc_diag_kernel(int64_t n, const int32_t *__restrict__ rowptr,
                                                             const int32_t *__restrict__ colind, const double *__restrict__ vals,
                                                             int32_t *__restrict__ diagpos, double *__restrict__ diagA)
{
    const int64_t r = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (r >= n)
        return;
    int32_t dp = -1;
    for (int32_t k = rowptr[r]; k < rowptr[r + 1]; k++)
        if (colind[k] == (int32_t)r)
            dp = k;
    diagpos[r] = dp;
    diagA[r] = dp >= 0 ? vals[dp] : 0.0;
}

__global__ __launch_bounds__(FV_BLOCK) void iota_kernel(int32_t *p, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i < n)
        p[i] = (int32_t)i;
}

// The CSC arrays of the caller are used as they are as CSR arrays, which is the same operator only for a symmetric matrix,
// and the PCG needs a symmetric one anyway: every stored (i, j) must have a stored (j, i) with the same value (to
// rounding: 1e-12 relative).  first_bad: the smallest offending entry index, or 0x7fffffff.
__global__ __launch_bounds__(FV_BLOCK) void csc_symmetry_kernel(int64_t n, const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx,
                                                                 const double *__restrict__ vals, int32_t *__restrict__ first_bad)
{
    const int64_t c = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (c >= n)
        return;
    for (int32_t k = ptr[c]; k < ptr[c + 1]; k++) {
        const int32_t r = idx[k];
        if (r == c)
            continue;
        int32_t lo = ptr[r], hi = ptr[r + 1] - 1; // rowval is ascending inside a column (SparseMatrixCSC's contract)
        bool found = false;
        double w = 0.0;
        while (lo <= hi) {
            const int32_t mid = lo + ((hi - lo) >> 1);
            const int32_t v = idx[mid];
            if (v == (int32_t)c) {
                found = true;
                w = vals[mid];
                break;
            }
            if (v < (int32_t)c)
                lo = mid + 1;
            else
                hi = mid - 1;
        }
        if (!found) // unsorted input: fall back to a scan
            for (int32_t j = ptr[r]; j < ptr[r + 1]; j++)
                if (idx[j] == (int32_t)c) {
                    found = true;
                    w = vals[j];
                    break;
                }
        const double v0 = vals[k];
        const bool same = found ? fabs(v0 - w) <= 1e-12 * (fabs(v0) + fabs(w)) : v0 == 0.0;
        if (!same)
            atomicMin(first_bad, k);
    }
}

extern "C" int fv_problem_create_from_csc(fv_ctx *ctx, int64_t n, const int64_t *colptr, const int64_t *rowval,
                                          const double *nzval, fv_problem **out)
{
    if (!ctx || !out || n < 0 || !colptr)
        return FV_ERR_ARG;
    *out = nullptr;
    FV_HIP(ctx, hipSetDevice(ctx->device));
    if (n > 0x7fffffffLL)
        return FV_ERR_TOO_LARGE;
    int64_t last = 0;
    FV_HIP(ctx, hipMemcpy(&last, colptr + n, sizeof last, hipMemcpyDefault));
    const int64_t nnz = last - 1;
    if (nnz < 0 || nnz > 0x7fffffffLL || (nnz > 0 && (!rowval || !nzval))) {
        fv_set_error(ctx, "fv_problem_create_from_csc: bad colptr / nnz=%lld", (long long)nnz);
        return FV_ERR_ARG;
    }
    fv_problem *p = new fv_problem();
    p->ctx = ctx;
    p->N = p->n = n;
    p->nnz = nnz;
    p->from_csc = true;
    int rc = FV_OK;
    do {
        if ((rc = p->rowptr.alloc(ctx, (size_t)n + 1)) || (rc = p->colind.alloc(ctx, (size_t)nnz + 2)) ||
            (rc = p->colind.zero(ctx)) || (rc = p->vals.alloc(ctx, (size_t)nnz + 2)) || (rc = p->vals.zero(ctx)) || (rc = p->diagpos.alloc(ctx, (size_t)n)) ||
            (rc = p->diagA.alloc(ctx, (size_t)n)) || (rc = p->b.alloc(ctx, (size_t)n)) || (rc = p->nodemap.alloc(ctx, (size_t)n)) ||
            (rc = p->f2n.alloc(ctx, (size_t)n)) || (rc = p->dheads.alloc(ctx, 1)))
            break;
        DevBuf<int64_t> w;
        int bad = 0;
        if ((rc = w.alloc(ctx, (size_t)(nnz > n + 1 ? nnz : n + 1))))
            break;
        if (hipMemcpyAsync(w.p, colptr, (size_t)(n + 1) * sizeof(int64_t), hipMemcpyDefault, ctx->stream) != hipSuccess) {
            rc = FV_ERR_HIP;
            break;
        }
        if ((rc = fv_narrow_indices(ctx, w.p, p->rowptr.p, n + 1, 1, nnz + 1, &bad)))
            break;
        if (!bad && nnz > 0) {
            if (hipMemcpyAsync(w.p, rowval, (size_t)nnz * sizeof(int64_t), hipMemcpyDefault, ctx->stream) != hipSuccess) {
                rc = FV_ERR_HIP;
                break;
            }
            if ((rc = fv_narrow_indices(ctx, w.p, p->colind.p, nnz, 1, n, &bad)))
                break;
        }
        if (bad) {
            fv_set_error(ctx, "fv_problem_create_from_csc: index out of range");
            rc = FV_ERR_INDEX;
            break;
        }
        if (nnz > 0 && (rc = fv_copy(ctx, p->vals.p, nzval, (size_t)nnz * sizeof(double))))
            break;
        if ((rc = p->b.zero(ctx)))
            break;
        if (n > 0) {
            hipLaunchKernelGGL(csc_diag_kernel, dim3(fv_blocks(n)), dim3(FV_BLOCK), 0, ctx->stream, n, p->rowptr.p, p->colind.p,
                               p->vals.p, p->diagpos.p, p->diagA.p);
            hipLaunchKernelGGL(iota_kernel, dim3(fv_blocks(n)), dim3(FV_BLOCK), 0, ctx->stream, p->nodemap.p, n);
            hipLaunchKernelGGL(iota_kernel, dim3(fv_blocks(n)), dim3(FV_BLOCK), 0, ctx->stream, p->f2n.p, n);
        }
        if (n > 0 && nnz > 0) {
            DevBuf<int32_t> firstbad;
            if ((rc = firstbad.alloc(ctx, 1)))
                break;
            int32_t h = 0x7fffffff;
            if (hipMemcpyAsync(firstbad.p, &h, sizeof h, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
                rc = FV_ERR_HIP;
                break;
            }
            hipLaunchKernelGGL(csc_symmetry_kernel, dim3(fv_blocks(n)), dim3(FV_BLOCK), 0, ctx->stream, n, (const int32_t *)p->rowptr.p,
                               (const int32_t *)p->colind.p, (const double *)p->vals.p, firstbad.p);
            if (hipMemcpyAsync(&h, firstbad.p, sizeof h, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                hipStreamSynchronize(ctx->stream) != hipSuccess) {
                rc = FV_ERR_HIP;
                break;
            }
            if (h != 0x7fffffff) {
                fv_set_error(ctx, "fv_problem_create_from_csc: the matrix is not symmetric (stored entry %d has no equal transposed partner); "
                                  "the device solver is a conjugate gradient and needs a symmetric positive definite operator — "
                                  "pass a host linearsolver such as (A, b, x0) -> A \\ b for this matrix", h + 1);
                rc = FV_ERR_ARG;
                break;
            }
        }
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) {
            rc = FV_ERR_HIP;
            break;
        }
    } while (0);
    if (rc != FV_OK) {
        delete p;
        return rc;
    }
    p->assembled = true;
    *out = p;
    return FV_OK;
}

FV_WARM_TU(assembly) // (fv_ctx_create loads every code object of the library up front: fv_warm_modules, fv_ctx.hip)
