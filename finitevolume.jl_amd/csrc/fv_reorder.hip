// Locality re-numbering of the free cells on the device: reverse Cuthill-McKee over the graph whose edges are the faces
// between two free cells — the same order, cell for cell, as the host routine it replaces (fv_host_locality_order:
// breadth-first from the unseen cell of lowest (degree, index), a cell's unseen neighbours appended by ascending (degree,
// index), the whole order reversed), so nothing else changes; what changes is where the time goes: the host version copies
// the face list back and walks 5e6 cells on one core (0.6 s for the fractures-like mesh), this one keeps everything in HBM.
//
// The reference's users meet such meshes first (DFN generators number the cells of a fracture at random; the one
// throughput the reference prints is for such a mesh, /root/reference/examples/fractures/ex.jl:9-15).
//
// Cuthill-McKee is sequential in its definition but not in its result: a cell is claimed by the neighbour that is dequeued
// first, i.e. by the neighbour with the smallest position in the previous level, so a level can be built from the previous
// one in data-parallel passes: (a) every frontier cell offers its position to its unseen neighbours (atomicMin on
// integers), (b) every frontier cell counts the neighbours it won, (c) a scan gives each its slot range, (d) it writes them
// there sorted by (degree, index).  The graphs at hand are narrow and deep (thousands of levels of a few thousand cells), so
// a per-level launch sequence would cost more in launches than in work, and one block is bound by the latency of its dependent
// loads: ONE launch of a few blocks (an eighth of the CUs at most, all resident) walks all levels of all components, the
// blocks meeting at a grid barrier between the passes (bounded spin: a block that waits too long makes everybody leave and the
// host routine runs instead; every loop condition is the same value in every block).  Integer atomics only: the order is
// deterministic.
#include "fv_internal.h"
#include "fv_device.h"
#include <cstdlib>

int g_reorder_blocks = 0; // (frozen) blocks of the device walk, 0 = a sixteenth of the CUs, at most 16

namespace {

constexpr int RB = 1024; // threads of the one block
constexpr int NB = 8;    // neighbours handled per stage of the walk

__global__ __launch_bounds__(FV_BLOCK) void edge_ends_kernel(int64_t F, const int32_t *__restrict__ node1, const int32_t *__restrict__ node2,
                                                              const int32_t *__restrict__ nodemap, int32_t *__restrict__ ea, int32_t *__restrict__ eb,
                                                              int32_t *__restrict__ deg, unsigned long long *__restrict__ dist_sum,
                                                              unsigned long long *__restrict__ used)
{
    const int64_t k = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    unsigned long long d = 0, u = 0;
    if (k < F) {
        const int32_t a = nodemap[node1[k]], b = nodemap[node2[k]]; // canonical free indices (negative: a Dirichlet cell)
        const bool edge = a >= 0 && b >= 0 && a != b;
        ea[k] = edge ? a : -1;
        eb[k] = edge ? b : -1;
        if (edge) {
            atomicAdd(deg + a, 1);
            atomicAdd(deg + b, 1);
            d = (unsigned long long)(a > b ? a - b : b - a);
            u = 1;
        }
    }
    // exact integer sums (|a - b| < 2^31, < 2^31 faces): the same number whatever the order
    for (int off = 32; off > 0; off >>= 1) {
        d += __shfl_xor(d, off, 64);
        u += __shfl_xor(u, off, 64);
    }
    if ((threadIdx.x & 63) == 0 && u) {
        atomicAdd(dist_sum, d);
        atomicAdd(used, u);
    }
}

__global__ __launch_bounds__(FV_BLOCK) void adj_fill_kernel(int64_t F, const int32_t *__restrict__ ea, const int32_t *__restrict__ eb,
                                                             int32_t *__restrict__ cursor, int32_t *__restrict__ adj)
{
    const int64_t k = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (k >= F || ea[k] < 0)
        return;
    adj[atomicAdd(cursor + ea[k], 1)] = eb[k];
    adj[atomicAdd(cursor + eb[k], 1)] = ea[k];
}

// neighbours of every cell in ascending order (rows are short; the fill order above depends on the atomics' timing)
__global__ __launch_bounds__(FV_BLOCK) void adj_sort_kernel(int64_t n, const int32_t *__restrict__ ptr, int32_t *__restrict__ adj)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i >= n)
        return;
    const int32_t s = ptr[i], e = ptr[i + 1];
    for (int32_t k = s + 1; k < e; k++) {
        const int32_t v = adj[k];
        int32_t j = k;
        while (j > s && adj[j - 1] > v) {
            adj[j] = adj[j - 1];
            j--;
        }
        adj[j] = v;
    }
}

struct Bfs {
    int32_t n, n_iso;
    const int32_t *ptr, *adj;
    int32_t *level, *ppos, *order, *cnt, *cnt2; // cnt2: degree of the cell at every position of the order (sort keys)
    int32_t *blocksum;                          // per block: children won by its share of the frontier
    unsigned long long *best;                   // start cell of the next component: min (degree << 32 | index) over the unseen
    unsigned int *bar;                          // [0] arrivals at the grid barrier (monotonic), [1] abort flag
    int32_t *status;                            // [0]: cells ordered, [1]: components, [2]: levels, [3]: 1 = gave up
    int32_t max_components;
};

// Everything the walk writes is read by other CUs: loads and stores of those arrays go to L2 (agent scope), never through a
// CU's own cache (MI355X_MICROARCH.md: a CU's vector L1 is not refreshed by other CUs' stores).  ptr / adj are read-only.
__device__ inline int32_t ldg(const int32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void stg(int32_t *p, int32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Barrier over the blocks of the launch (all resident: the grid is a small fraction of the CUs).  A monotonic arrival
// counter; the spin is bounded — a block that waits too long raises the abort flag and everybody leaves (the host routine
// then computes the order), so the grid always drains.  Returns false when the walk has been given up.
__device__ inline bool grid_barrier(const Bfs &g, unsigned int &epoch)
{
    __threadfence();
    __syncthreads();
    __shared__ int ok;
    if (threadIdx.x == 0) {
        epoch += gridDim.x;
        __hip_atomic_fetch_add(g.bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int good = 1;
        long long spins = 0;
        while (__hip_atomic_load(g.bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
            if (__hip_atomic_load(g.bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) || ++spins > (1ll << 21)) {
                __hip_atomic_store(g.bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                good = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        if (__hip_atomic_load(g.bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            good = 0;
        ok = good;
    }
    __syncthreads();
    return ok != 0;
}

// block-wide exclusive scan of one value per thread; returns the thread's offset, *total = the sum (all threads)
__device__ inline int32_t block_scan(int32_t v, int32_t *wsum, int32_t *total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int32_t x = v;
    for (int off = 1; off < 64; off <<= 1) {
        const int32_t y = __shfl_up(x, off, 64);
        if (lane >= off)
            x += y;
    }
    __syncthreads();
    if (lane == 63)
        wsum[wave] = x;
    __syncthreads();
    int32_t base = 0, all = 0;
    for (int w = 0; w < RB / 64; w++) {
        const int32_t s = wsum[w];
        if (w < wave)
            base += s;
        all += s;
    }
    *total = all;
    return base + x - v;
}

#define FV_BFS_SYNC()                  \
    do {                               \
        if (!grid_barrier(g, epoch))   \
            return;                    \
    } while (0)

__global__ __launch_bounds__(RB) void bfs_order_kernel(Bfs g)
{
    __shared__ int32_t wsum[RB / 64];
    const int tid = (int)threadIdx.x, nb = (int)gridDim.x, blk = (int)blockIdx.x;
    const int32_t n = g.n;
    unsigned int epoch = 0;
    int32_t done = 0, comps = 0, levels = 0;
    // cells without any edge: components of their own, first in the order (lowest degree), by index — block 0, the others wait
    if (g.n_iso > 0) {
        if (blk == 0) {
            int32_t at = 0;
            for (int32_t base = 0; base < n; base += RB) {
                const int32_t i = base + tid;
                const int32_t iso = (i < n && g.ptr[i + 1] == g.ptr[i]) ? 1 : 0;
                int32_t total;
                const int32_t off = block_scan(iso, wsum, &total);
                if (iso) {
                    stg(g.order + at + off, i);
                    stg(g.level + i, 0);
                }
                at += total;
            }
        }
        done = comps = g.n_iso;
        FV_BFS_SYNC();
    }
    while (done < n) {
        if (comps >= g.max_components) {
            if (blk == 0 && tid == 0)
                g.status[3] = 1; // the host routine takes over (a mesh of very many small components)
            return;
        }
        // the unseen cell of lowest (degree, index) starts the next component
        unsigned long long mine = ~0ull;
        for (int32_t i = blk * RB + tid; i < n; i += nb * RB)
            if (ldg(g.level + i) < 0) {
                const unsigned long long key = ((unsigned long long)(uint32_t)(g.ptr[i + 1] - g.ptr[i]) << 32) | (uint32_t)i;
                mine = key < mine ? key : mine;
            }
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long o = __shfl_xor(mine, off, 64);
            mine = o < mine ? o : mine;
        }
        if ((tid & 63) == 0 && mine != ~0ull)
            atomicMin(g.best, mine);
        FV_BFS_SYNC();
        const int32_t start = (int32_t)(__hip_atomic_load(g.best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0xffffffffull);
        if (blk == 0 && tid == 0) {
            stg(g.order + done, start);
            stg(g.level + start, 0);
        }
        FV_BFS_SYNC(); // (everybody has read `best`; the start cell is in place)
        if (blk == 0 && tid == 0)
            __hip_atomic_store(g.best, ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        comps++;
        int32_t lo = done, hi = done + 1, cur = 0;
        while (lo < hi) {
            // this block's share of the frontier: a contiguous range, so that children come out in the order of their parents
            const int32_t len = hi - lo, share = (len + nb - 1) / nb;
            const int32_t mylo = lo + blk * share < hi ? lo + blk * share : hi, myhi = mylo + share < hi ? mylo + share : hi;
            // (a) every frontier cell offers its position to its unseen neighbours (eight neighbours per stage: the walk is
            // bound by the latency of dependent loads, every load of a stage is issued before the first is waited for)
            for (int32_t q = mylo + tid; q < myhi; q += RB) {
                const int32_t u = ldg(g.order + q);
                for (int32_t k = g.ptr[u], e = g.ptr[u + 1]; k < e; k += NB) {
                    int32_t v[NB], lv[NB];
#pragma unroll
                    for (int j = 0; j < NB; j++)
                        v[j] = k + j < e ? g.adj[k + j] : -1;
#pragma unroll
                    for (int j = 0; j < NB; j++)
                        lv[j] = v[j] >= 0 ? ldg(g.level + v[j]) : 0;
#pragma unroll
                    for (int j = 0; j < NB; j++)
                        if (v[j] >= 0 && (lv[j] < 0 || lv[j] == cur + 1)) {
                            if (lv[j] < 0)
                                stg(g.level + v[j], cur + 1); // (every writer writes the same value)
                            atomicMin(g.ppos + v[j], q);
                        }
                }
            }
            FV_BFS_SYNC();
            // (b) how many it won; offsets inside the block's share, the share's total to blocksum
            int32_t carry = 0;
            for (int32_t base = mylo; base < myhi; base += RB) {
                const int32_t q = base + tid;
                int32_t c = 0;
                if (q < myhi) {
                    const int32_t u = ldg(g.order + q);
                    int32_t prev = -1;
                    for (int32_t k = g.ptr[u], e = g.ptr[u + 1]; k < e; k += NB) {
                        int32_t v[NB], lv[NB], pp[NB];
#pragma unroll
                        for (int j = 0; j < NB; j++)
                            v[j] = k + j < e ? g.adj[k + j] : -1;
#pragma unroll
                        for (int j = 0; j < NB; j++) {
                            lv[j] = v[j] >= 0 ? ldg(g.level + v[j]) : 0;
                            pp[j] = v[j] >= 0 ? ldg(g.ppos + v[j]) : -1;
                        }
#pragma unroll
                        for (int j = 0; j < NB; j++) {
                            if (v[j] >= 0 && v[j] != prev && lv[j] == cur + 1 && pp[j] == q)
                                c++;
                            prev = v[j] >= 0 ? v[j] : prev;
                        }
                    }
                }
                int32_t total;
                const int32_t off = block_scan(c, wsum, &total);
                if (q < myhi)
                    stg(g.cnt + q, carry + off);
                carry += total;
            }
            if (tid == 0)
                stg(g.blocksum + blk, carry);
            FV_BFS_SYNC();
            // (c) where the share's children start; (d) its cells write theirs by ascending (degree, index)
            int32_t before = 0, all = 0;
            for (int b = 0; b < nb; b++) {
                const int32_t sb = ldg(g.blocksum + b);
                before += b < blk ? sb : 0;
                all += sb;
            }
            for (int32_t q = mylo + tid; q < myhi; q += RB) {
                const int32_t u = ldg(g.order + q);
                const int32_t slot = hi + before + ldg(g.cnt + q);
                int32_t m = 0, prev = -1;
                for (int32_t k = g.ptr[u], e = g.ptr[u + 1]; k < e; k += NB) {
                    int32_t v[NB], lv[NB], pp[NB], dg[NB];
#pragma unroll
                    for (int j = 0; j < NB; j++)
                        v[j] = k + j < e ? g.adj[k + j] : -1;
#pragma unroll
                    for (int j = 0; j < NB; j++) {
                        lv[j] = v[j] >= 0 ? ldg(g.level + v[j]) : 0;
                        pp[j] = v[j] >= 0 ? ldg(g.ppos + v[j]) : -1;
                        dg[j] = v[j] >= 0 ? g.ptr[v[j] + 1] - g.ptr[v[j]] : 0;
                    }
#pragma unroll
                    for (int j = 0; j < NB; j++) {
                        if (v[j] >= 0 && v[j] != prev && lv[j] == cur + 1 && pp[j] == q) {
                            // insertion into the sorted run order[slot, slot + m): (degree, index) ascending; the run's degrees in cnt2
                            int32_t i = m;
                            while (i > 0) {
                                const int32_t w = ldg(g.order + slot + i - 1);
                                const int32_t dw = ldg(g.cnt2 + slot + i - 1);
                                if (dw < dg[j] || (dw == dg[j] && w < v[j]))
                                    break;
                                stg(g.order + slot + i, w);
                                stg(g.cnt2 + slot + i, dw);
                                i--;
                            }
                            stg(g.order + slot + i, v[j]);
                            stg(g.cnt2 + slot + i, dg[j]);
                            m++;
                        }
                        prev = v[j] >= 0 ? v[j] : prev;
                    }
                }
            }
            FV_BFS_SYNC();
            lo = hi;
            hi += all;
            cur++;
            levels++;
        }
        done = hi;
    }
    if (blk == 0 && tid == 0) {
        g.status[0] = done;
        g.status[1] = comps;
        g.status[2] = levels;
    }
}
#undef FV_BFS_SYNC

__global__ __launch_bounds__(FV_BLOCK) void count_isolated_kernel(int64_t n, const int32_t *__restrict__ ptr, int32_t *__restrict__ count)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    const int iso = i < n && ptr[i + 1] == ptr[i];
    const unsigned long long m = __ballot(iso);
    if ((threadIdx.x & 63) == 0 && m)
        atomicAdd(count, (int32_t)__popcll(m));
}

__global__ __launch_bounds__(FV_BLOCK) void reverse_order_kernel(int64_t n, const int32_t *__restrict__ order, int32_t *__restrict__ perm)
{
    const int64_t k = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (k < n)
        perm[order[k]] = (int32_t)(n - 1 - k);
}

__global__ __launch_bounds__(FV_BLOCK) void edge_dist_kernel(int64_t F, const int32_t *__restrict__ ea, const int32_t *__restrict__ eb,
                                                              const int32_t *__restrict__ perm, unsigned long long *__restrict__ dist_sum)
{
    const int64_t k = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    unsigned long long d = 0;
    if (k < F && ea[k] >= 0) {
        const int32_t a = perm[ea[k]], b = perm[eb[k]];
        d = (unsigned long long)(a > b ? a - b : b - a);
    }
    for (int off = 32; off > 0; off >>= 1)
        d += __shfl_xor(d, off, 64);
    if ((threadIdx.x & 63) == 0 && d)
        atomicAdd(dist_sum, d);
}

} // namespace

// mean |i - j| over the faces between two free cells, as numbered (device; exact)
// perm_dev (n int32, device) receives perm[canonical] = new index when *adopted; want: 1 = adopt only if numbered badly and
// the new order at least halves the mean distance (fv_reorder_free's rule), 2 = always.
// Returns FV_OK with *handled = 0 when the graph has too many components for the one-block walk (the host routine then runs).
int fv_device_locality_order(fv_problem *p, int want, int32_t *perm_dev, bool *adopted, bool *handled, double *mean_before, double *mean_after)
{
    fv_ctx *ctx = p->ctx;
    const int64_t n = p->n, F = p->F;
    *adopted = false;
    *handled = true;
    *mean_before = *mean_after = 0.0;
    DevBuf<int32_t> ea, eb, deg, ptr, cursor, adj, level, ppos, order, cnt, cnt2, status;
    DevBuf<unsigned long long> sums;
    FV_TRY(ea.alloc(ctx, (size_t)F));
    FV_TRY(eb.alloc(ctx, (size_t)F));
    FV_TRY(deg.alloc(ctx, (size_t)n + 1));
    FV_TRY(deg.zero(ctx));
    FV_TRY(sums.alloc(ctx, 3));
    FV_TRY(sums.zero(ctx));
    hipLaunchKernelGGL(edge_ends_kernel, dim3(fv_blocks(F)), dim3(FV_BLOCK), 0, ctx->stream, F, (const int32_t *)p->node1.p, (const int32_t *)p->node2.p,
                       (const int32_t *)p->nodemap.p, ea.p, eb.p, deg.p, sums.p, sums.p + 1);
    FV_LAUNCH_CHECK(ctx);
    unsigned long long hs[3] = {0, 0, 0};
    FV_TRY(fv_copy(ctx, hs, sums.p, 2 * sizeof(unsigned long long)));
    if (hs[1] == 0)
        return FV_OK; // no face between two free cells
    *mean_before = (double)hs[0] / (double)hs[1];
    if (want == 1 && *mean_before <= 2.0 * pow((double)n, 2.0 / 3.0))
        return FV_OK; // numbered like a grid (or better): nothing to gain
    if (2 * hs[1] >= 0x7fffffffull) {
        *handled = false;
        return FV_OK;
    }
    FV_TRY(ptr.alloc(ctx, (size_t)n + 1));
    int64_t total = 0;
    FV_TRY(fv_exclusive_scan_i32(ctx, deg.p, ptr.p, n, &total));
    FV_TRY(cursor.alloc(ctx, (size_t)n + 1));
    FV_HIP(ctx, hipMemcpyAsync(cursor.p, ptr.p, ((size_t)n + 1) * sizeof(int32_t), hipMemcpyDeviceToDevice, ctx->stream));
    FV_TRY(adj.alloc(ctx, (size_t)total));
    hipLaunchKernelGGL(adj_fill_kernel, dim3(fv_blocks(F)), dim3(FV_BLOCK), 0, ctx->stream, F, (const int32_t *)ea.p, (const int32_t *)eb.p, cursor.p, adj.p);
    hipLaunchKernelGGL(adj_sort_kernel, dim3(fv_blocks(n)), dim3(FV_BLOCK), 0, ctx->stream, n, (const int32_t *)ptr.p, adj.p);
    FV_LAUNCH_CHECK(ctx);
    FV_TRY(level.alloc(ctx, (size_t)n));
    FV_TRY(ppos.alloc(ctx, (size_t)n));
    FV_TRY(order.alloc(ctx, (size_t)n));
    FV_TRY(cnt.alloc(ctx, (size_t)n));
    FV_TRY(cnt2.alloc(ctx, (size_t)n));
    FV_TRY(status.alloc(ctx, 4));
    FV_HIP(ctx, hipMemsetAsync(level.p, 0xff, (size_t)n * sizeof(int32_t), ctx->stream)); // -1
    FV_HIP(ctx, hipMemsetAsync(ppos.p, 0x7f, (size_t)n * sizeof(int32_t), ctx->stream));  // 0x7f7f7f7f: above every position
    FV_TRY(status.zero(ctx));
    Bfs g{};
    g.n = (int32_t)n;
    g.ptr = ptr.p;
    g.adj = adj.p;
    g.level = level.p;
    g.ppos = ppos.p;
    g.order = order.p;
    g.cnt = cnt.p;
    g.cnt2 = cnt2.p;
    g.status = status.p;
    g.max_components = 1 << 12;
    // the walk's blocks meet at a grid barrier: few enough of them to be resident together, and few enough for the barrier to stay
    // cheap — a level of a DFN mesh is ~2 000 cells: the 5M-cell mesh re-numbers in 0.128 / 0.095 / 0.086 / 0.110 / 0.142 / 0.175 s
    // with 4 / 8 / 16 / 32 / 48 / 64 blocks
    int nblk = ctx->num_cus / 16;
    nblk = nblk < 1 ? 1 : (nblk > 16 ? 16 : nblk);
    if (g_reorder_blocks > 0 && g_reorder_blocks <= 64)
        nblk = g_reorder_blocks;
    DevBuf<int32_t> blocksum, niso;
    DevBuf<unsigned long long> best;
    DevBuf<unsigned int> bar;
    FV_TRY(blocksum.alloc(ctx, 64));
    FV_TRY(blocksum.zero(ctx));
    FV_TRY(best.alloc(ctx, 1));
    FV_HIP(ctx, hipMemsetAsync(best.p, 0xff, sizeof(unsigned long long), ctx->stream));
    FV_TRY(bar.alloc(ctx, 2));
    FV_TRY(bar.zero(ctx));
    FV_TRY(niso.alloc(ctx, 1));
    FV_TRY(niso.zero(ctx));
    hipLaunchKernelGGL(count_isolated_kernel, dim3(fv_blocks(n)), dim3(FV_BLOCK), 0, ctx->stream, n, (const int32_t *)ptr.p, niso.p);
    FV_LAUNCH_CHECK(ctx);
    int32_t h_iso = 0;
    FV_TRY(fv_copy(ctx, &h_iso, niso.p, sizeof h_iso));
    g.n_iso = h_iso;
    g.blocksum = blocksum.p;
    g.best = best.p;
    g.bar = bar.p;
    hipLaunchKernelGGL(bfs_order_kernel, dim3(nblk), dim3(RB), 0, ctx->stream, g);
    FV_LAUNCH_CHECK(ctx);
    int32_t st[4] = {0, 0, 0, 0};
    FV_TRY(fv_copy(ctx, st, status.p, sizeof st));
    if (getenv("FV_TRACE_REORDER"))
        fprintf(stderr, "[fvhip] device re-numbering: %d blocks, %d cells ordered of %lld, %d components, %d levels%s\n", nblk, st[0], (long long)n, st[1], st[2],
                st[0] != (int32_t)n ? " -> host routine" : "");
    if (st[0] != (int32_t)n) { // very many components, or the walk was given up: the host routine computes the order
        *handled = false;
        return FV_OK;
    }
    hipLaunchKernelGGL(reverse_order_kernel, dim3(fv_blocks(n)), dim3(FV_BLOCK), 0, ctx->stream, n, (const int32_t *)order.p, perm_dev);
    hipLaunchKernelGGL(edge_dist_kernel, dim3(fv_blocks(F)), dim3(FV_BLOCK), 0, ctx->stream, F, (const int32_t *)ea.p, (const int32_t *)eb.p,
                       (const int32_t *)perm_dev, sums.p + 2);
    FV_LAUNCH_CHECK(ctx);
    FV_TRY(fv_copy(ctx, hs + 2, sums.p + 2, sizeof(unsigned long long)));
    *mean_after = (double)hs[2] / (double)hs[1];
    *adopted = !(want == 1 && *mean_after * 2.0 >= *mean_before);
    return FV_OK;
}

FV_WARM_TU(reorder) // (fv_ctx_create loads every code object of the library up front: fv_warm_modules, fv_ctx.hip)
