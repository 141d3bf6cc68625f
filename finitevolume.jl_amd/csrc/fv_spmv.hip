// SpMV of the assembled operator (A + sigma*D): CSR wave-stream, sliced-DIA and plane-marching sliced-DIA forms,
// their set-up (group order, DIA copy, folded diagonal) and the dispatcher the PCG calls (spmv_apply).
// The conjugate gradient built on top is in fv_pcg.hip.
//
// Replaces the `A * x` inside IterativeSolvers.cg!/cg at /root/reference/src/FiniteVolume.jl:160-161 and
// src/transient.jl:50-58.  The operator is (A + sigma*D): A the assembled symmetric CSR, D = Ss*volumes on the free
// cells, sigma = 1/dt (0 for the steady solve), i.e. the SPD form of the reference's (I/dt + D^-1 A).
//
// Everything here is HBM-bandwidth bound (0.13 flop/B), so no MFMA: the kernels are built for coalesced streaming of the
// matrix, L2- or register-served re-use of x, and wave64 shuffle + LDS reductions for the p.q epilogue.
#include "fv_internal.h"
#include "fv_device.h"
#include "fv_spmv.h"

#include <cstdlib>

// ------------------------------------------------------------------ SpMV
// LPR lanes cooperate on one row (8 for the 7-point stencil: a wave64 covers 8
// consecutive rows, whose ~56 stored entries are contiguous in vals/colind, so a
// wave load instruction is one coalesced 448-byte burst).  Each thread keeps U
// independent rows in flight to cover HBM latency.  Blocks sweep the matrix in
// passes of G*RPB consecutive rows so that concurrently running blocks work on
// neighbouring rows (x re-reads stay in L2 / Infinity Cache); within a pass the
// eight XCDs get contiguous sub-windows (blockIdx & 7 labels the XCD share).
template <int LPR, int U, bool DOT>
__global__ __launch_bounds__(FV_BLOCK) void spmv_kernel(int64_t n, const int32_t *__restrict__ rowptr,
                                                         const int32_t *__restrict__ colind, const double *__restrict__ vals,
                                                         const double *__restrict__ x, double *__restrict__ y,
                                                         const double *__restrict__ shift, double sigma,
                                                         double *__restrict__ partials, const PcgScalars *__restrict__ scal)
{
    __shared__ double smem[4];
    if (scal && scal->done)
        return;
    constexpr int ROWS_SUB = FV_BLOCK / LPR; // rows per unrolled sub-pass
    constexpr int RPB = ROWS_SUB * U;        // rows per block per pass
    const int G = gridDim.x;
    const int slot = (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3);
    const int sub = threadIdx.x % LPR;
    const int rib = threadIdx.x / LPR;
    double dacc = 0.0;
    for (int64_t base = (int64_t)slot * RPB; base < n; base += (int64_t)G * RPB) {
        int32_t k[U], e[U];
        double sum[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int64_t row = base + u * ROWS_SUB + rib;
            if (row < n) {
                k[u] = rowptr[row] + sub;
                e[u] = rowptr[row + 1];
            } else {
                k[u] = 0;
                e[u] = 0;
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            double s = 0.0;
            if (k[u] < e[u])
                s = vals[k[u]] * x[colind[k[u]]];
            sum[u] = s;
        }
#pragma unroll
        for (int u = 0; u < U; u++) // rows longer than LPR
            for (int32_t kk = k[u] + LPR; kk < e[u]; kk += LPR)
                sum[u] += vals[kk] * x[colind[kk]];
#pragma unroll
        for (int u = 0; u < U; u++) {
            double s = sum[u];
#pragma unroll
            for (int off = LPR / 2; off > 0; off >>= 1)
                s += __shfl_xor(s, off, LPR);
            const int64_t row = base + u * ROWS_SUB + rib;
            if (sub == 0 && row < n) {
                const double xr = (shift || DOT) ? x[row] : 0.0;
                if (shift)
                    s += sigma * shift[row] * xr;
                y[row] = s;
                if (DOT)
                    dacc += xr * s;
            }
        }
    }
    if (DOT) {
        const double t = block_sum(dacc, smem);
        if (threadIdx.x == 0)
            partials[blockIdx.x] = t;
    }
}

// ------------------------------------------------------------------ SpMV, wave-private CSR-stream (the production form)
// Each WAVE owns one group of 64 consecutive rows per pass, hence one contiguous
// range of vals/colind.  That range is streamed with lane-contiguous 16-byte
// (vals) and 8-byte (colind) loads that do not depend on the individual row
// pointers; the products v*x[col] are staged in the wave's own LDS tile and each
// lane then sums its row in column order.  rowptr, colind, vals, y (and D) are
// fully coalesced HBM streams; only the x gather is irregular.  There is no block
// barrier in the loop: the 32 waves of a CU drift through their load / gather /
// reduce phases independently and cover each other's latency, and the row
// pointers of the next pass are prefetched one pass ahead.
//
// Traffic, not latency, bounds this kernel (measured: time = L2-miss bytes / ~5
// TB/s for every variant), so the traversal matters: `order` lists the row groups
// band by band and, inside a band, plane after plane (see build_group_order), and
// every XCD sweeps its own contiguous part of that list.  A group's +plane x
// lines are then still in that XCD's 4 MiB L2 when the same band of the next
// plane needs them as centre and -plane arms, instead of being fetched 3 times.
// Optional epilogue that turns the first SpMV of an implicit step (q = A u) into the whole PCG set-up
// (see pcg_init_kernel<true>): r = b' - q, p = M^-1 r and the three partial sums, without writing q.

template <int WT, bool DOT, bool NT, bool INIT = false>
__global__ __launch_bounds__(FV_BLOCK) void spmv_wstream_kernel(int64_t n, const int32_t *__restrict__ rowptr,
                                                                 const int32_t *__restrict__ colind, const double *__restrict__ vals,
                                                                 const double *__restrict__ x, double *__restrict__ y,
                                                                 const double *__restrict__ shift, double sigma,
                                                                 double *__restrict__ partials, const PcgScalars *__restrict__ scal,
                                                                 const int32_t *__restrict__ order, int64_t npos, StepInitEpilogue epi = {})
{
    constexpr int NIT = WT / 128; // entry pairs per lane
    constexpr int WPB = FV_BLOCK / 64;
    __shared__ double prod_all[WPB][WT + 2];
    __shared__ double smem[4];
    if (scal && scal->done)
        return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double *prod = prod_all[wave];
    const int64_t ngroups = npos; // positions to visit: all 64-row groups, or the entries of `order`
    // position space: XCD share xs of [0, ngroups) is swept in passes of (G/8)*WPB consecutive positions
    const int64_t per_xcd = (ngroups + 7) >> 3;
    const int64_t pstride = (int64_t)(gridDim.x >> 3) * WPB;
    const int64_t xbase = (int64_t)(blockIdx.x & 7) * per_xcd;
    const int64_t xend = (xbase + per_xcd < ngroups) ? xbase + per_xcd : ngroups;
    int64_t pos = xbase + (int64_t)(blockIdx.x >> 3) * WPB + wave;
    double dacc = 0.0, arr = 0.0, abb = 0.0; // INIT: dacc = r.M^-1 r
    int64_t group = 0;
    int32_t s = 0, e = 0;
    if (pos < xend) {
        group = order ? order[pos] : pos;
        const int64_t row = (group << 6) + lane;
        if (row < n) {
            s = rowptr[row];
            e = rowptr[row + 1];
        }
    }
    for (; pos < xend; pos += pstride) {
        const int64_t r0 = group << 6;
        const int nr = (int)((n - r0 < 64) ? (n - r0) : 64);
        const int32_t my_s = s, my_e = e;
        const int32_t k0 = __builtin_amdgcn_readfirstlane(my_s);
        const int32_t k1 = __builtin_amdgcn_readlane(my_e, nr - 1);
        const int32_t ka = k0 & ~1; // 16-byte aligned start of the streamed range
        // prefetch the next group's row pointers
        s = 0;
        e = 0;
        if (pos + pstride < xend) {
            group = order ? order[pos + pstride] : pos + pstride;
            const int64_t nrow = (group << 6) + lane;
            if (nrow < n) {
                s = rowptr[nrow];
                e = rowptr[nrow + 1];
            }
        }
        const int64_t row = r0 + lane;
        double sum = 0.0;
        if (k1 - ka <= WT) {
            double2 v[NIT];
            int2 c[NIT];
#pragma unroll
            for (int it = 0; it < NIT; it++) {
                const int32_t j = ka + 2 * (lane + it * 64);
                if (j < k1) { // vals/colind carry two padding entries past nnz
                    if (NT) { // read-once streams: ask the caches not to keep them, so the x lines survive in L2
                        v[it].x = __builtin_nontemporal_load(vals + j);
                        v[it].y = __builtin_nontemporal_load(vals + j + 1);
                        c[it].x = __builtin_nontemporal_load(colind + j);
                        c[it].y = __builtin_nontemporal_load(colind + j + 1);
                    } else {
                        v[it] = *reinterpret_cast<const double2 *>(vals + j);
                        c[it] = *reinterpret_cast<const int2 *>(colind + j);
                    }
                }
            }
#pragma unroll
            for (int it = 0; it < NIT; it++) {
                const int32_t j = ka + 2 * (lane + it * 64);
                if (j < k1) {
                    double2 pr;
                    pr.x = v[it].x * x[c[it].x];
                    pr.y = v[it].y * x[c[it].y];
                    *reinterpret_cast<double2 *>(prod + (j - ka)) = pr;
                }
            }
            // the wave's own LDS writes must land before other lanes read them
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (int32_t k = my_s - ka, ke = my_e - ka; k < ke; k++)
                sum += prod[k];
            __builtin_amdgcn_wave_barrier(); // reads done before the next pass overwrites the tile
        } else { // rare: more than WT entries in 64 rows; each lane walks its own row
            for (int32_t k = my_s; k < my_e; k++)
                sum += vals[k] * x[colind[k]];
        }
        if (INIT) {
            if (lane < nr) { // sum = (A u)_row
                const double di = epi.D[row];
                double bi = epi.bprime ? epi.bprime[row] : 0.0;
                if (epi.b_times_D)
                    bi *= di;
                const double rhsv = bi + di * (x[row] / epi.dt);
                const double ri = epi.q_shifted ? rhsv - sum : bi - sum;
                bi = rhsv;
                double mi;
                if (epi.compute_minv) {
                    {
                        const double dd = epi.diagA[row] + epi.sigma * di;
                        mi = dd > 0.0 ? 1.0 / dd : 0.0;
                    }
                    epi.minv[row] = mi;
                } else
                    mi = epi.minv[row];
                const double zi = mi * ri;
                epi.r[row] = ri;
                epi.pv[row] = zi;
                dacc += ri * zi;
                arr += ri * ri;
                abb += bi * bi;
            }
        } else if (lane < nr) {
            const double xr = (shift || DOT) ? x[row] : 0.0;
            if (shift)
                sum += sigma * shift[row] * xr;
            if (NT)
                __builtin_nontemporal_store(sum, y + row);
            else
                y[row] = sum;
            if (DOT)
                dacc += xr * sum;
        }
    }
    if (INIT) {
        const double t0 = block_sum(dacc, smem);
        const double t1 = block_sum(arr, smem);
        const double t2 = block_sum(abb, smem);
        if (threadIdx.x == 0) {
            epi.part_rz[blockIdx.x] = t0;
            epi.part_rr[blockIdx.x] = t1;
            epi.part_bb[blockIdx.x] = t2;
        }
    } else if (DOT) {
        const double tsum = block_sum(dacc, smem);
        if (threadIdx.x == 0)
            partials[blockIdx.x] = tsum;
    }
}

// Tuning knobs (fv_tune) for in-process A/B: 0 = SpMV form (1 lanes-per-row, 2 wave stream),
// 1 = unroll of the lanes-per-row form, 2 = use the plane-blocked traversal order (0/1),
// 3 = use the diagonal-folded shifted matrix copy in fixed-dt runs (0/1), 4 = non-temporal streaming loads (0/1),
// 5 = fuse the PCG set-up of an implicit step into its first SpMV (0/1), 6 = sliced-DIA form for grid-like slices (0/1)
int g_spmv_form = 2;
static int g_spmv_unroll = 2;
static int g_use_order = 1;
static int g_nt = 1;
int g_fuse_init = 0; // measured: with the sliced-DIA SpMV the separate set-up kernel is ~3 % faster than the fused epilogue
int g_use_dia = 1;
static int g_dia_packed = 1; // sliced-DIA values packed (sl_noff blocks per slice) or padded to DIA_K blocks (fv_tune key 11; read when the DIA copy is built)
static int g_march_dbg = 0; // diagnosis switches of the marching kernel (fv_tune key 17; results are wrong when set)
static int g_march_wide = 1; // marching kernel: 16-byte window accesses instead of centre + two edge loads when stride mod 64 <= 32 (fv_tune key 18)
static int g_trace_spmv = getenv("FV_TRACE_SPMV") ? atoi(getenv("FV_TRACE_SPMV")) : 0;
static int g_march = 1;      // plane-marching sliced-DIA kernel on structured grids (fv_tune key 9): 0 never, 1 when x outgrows the last-level cache, 2 always
static int g_march_min_mb = 160; // ... i.e. when the x vector exceeds this many MiB (fv_tune key 19; MI355X has 256 MB of infinity cache, which the step's other streams share: inside the stepping loop the crossover is at ~2e7 rows)
static int g_march_segs = 0; // segments per XCD of the marching kernel (fv_tune key 10; 0 = chosen per operator)
int g_fold_shift = 1;
extern int g_carry_refresh, g_carry_speculate; // fv_transient.hip
extern int g_sparse_b, g_chain_test_break;     // fv_pcg.hip
extern int g_chain_steps;                      // fv_transient.hip

// blocks that are all resident at 8 waves per SIMD: 8 per CU (2048 on the 256-CU MI355X; fewer on a partitioned device)
static int g_resident_blocks = FV_MAX_PARTIALS;
static void set_resident_blocks(const fv_ctx *ctx)
{
    int64_t b = (int64_t)ctx->num_cus * 8 / 8 * 8;
    if (b < 8)
        b = 8;
    g_resident_blocks = b > FV_MAX_PARTIALS ? FV_MAX_PARTIALS : (int)b;
}

extern int g_gradient_knots_per_pass; // fv_gradient.hip
extern int g_comm_single_rank_collectives; // fv_comm.hip
extern int g_defer_reduce, g_k2s_nt;        // fv_pcg.hip

extern "C" int fv_tune(int key, int value)
{
    if (key == 0 && (value == 1 || value == 2))
        g_spmv_form = value;
    else if (key == 1 && (value == 2 || value == 4 || value == 8))
        g_spmv_unroll = value;
    else if (key == 2 && (value == 0 || value == 1))
        g_use_order = value;
    else if (key == 3 && (value == 0 || value == 1))
        g_fold_shift = value;
    else if (key == 4 && (value == 0 || value == 1))
        g_nt = value;
    else if (key == 5 && (value == 0 || value == 1))
        g_fuse_init = value;
    else if (key == 6 && (value == 0 || value == 1))
        g_use_dia = value;
    else if (key == 7 && value >= 0)
        g_carry_refresh = value;
    else if (key == 8 && (value == 0 || value == 1))
        g_carry_speculate = value;
    else if (key == 9 && value >= 0 && value <= 2)
        g_march = value;
    else if (key == 19 && value >= 0)
        g_march_min_mb = value;
    else if (key == 20 && (value == 0 || value >= 2))
        g_gradient_knots_per_pass = value;
    else if (key == 21 && (value == 0 || value == 1))
        g_comm_single_rank_collectives = value;
    else if (key == 22 && (value == 0 || value == 1))
        g_defer_reduce = value;
    else if (key == 25 && value >= 0)
        g_trace_spmv = value;
    else if (key == 26 && ((value >= 0 && value <= 3) || value == 7))
        g_k2s_nt = value;
    else if (key == 10 && value >= 0 && value <= 16)
        g_march_segs = value;
    else if (key == 11 && (value == 0 || value == 1))
        g_dia_packed = value;
    else if (key == 12 && value >= 0 && value <= 2)
        g_sparse_b = value;
    else if (key == 17 && value >= 0 && value <= 3)
        g_march_dbg = value;
    else if (key == 18 && (value == 0 || value == 1))
        g_march_wide = value;
    else if (key == 13 && value >= 0 && value <= 32)
        g_chain_steps = value;
    else if (key == 14 && value >= -1 && value < 32)
        g_chain_test_break = value;
    else
        return FV_ERR_ARG;
    return FV_OK;
}

// lanes per row from the mean row length: 8 covers the 7-point stencil in one pass
static int spmv_lpr(const fv_problem *p)
{
    const double avg = p->n > 0 ? (double)p->nnz / (double)p->n : 0.0;
    return avg >= 11.0 ? 16 : (avg <= 4.0 ? 4 : 8);
}

constexpr int STREAM_RB = 256; // rows per block per pass of the wave-stream form (4 waves x 64)

int fv_spmv_grid(fv_problem *p)
{
    // a multiple of 8 (XCD shares), at most one partial per block
    int64_t g;
    if (g_spmv_form >= 2)
        g = (p->n + STREAM_RB - 1) / STREAM_RB;
    else {
        const int rpb = (FV_BLOCK / spmv_lpr(p)) * g_spmv_unroll;
        g = (p->n + rpb - 1) / rpb;
    }
    g = ((g + 7) / 8) * 8;
    if (g > FV_MAX_PARTIALS)
        g = FV_MAX_PARTIALS;
    if (g < 8)
        g = 8;
    return (int)g;
}

// A matrix whose rows mostly reach `stride` rows ahead (the +i1 neighbour of a
// structured grid) is traversed band by band: for each band of BAND in-plane row
// offsets, plane after plane.  Returned as a list of 64-row group ids.
static int build_group_order(fv_problem *p)
{
    fv_ctx *ctx = p->ctx;
    p->order_built = true;
    const int64_t n = p->n;
    if (n < (1 << 20) || p->nnz == 0)
        return FV_OK; // small: x stays cache-resident anyway
    // estimate the far stride from the middle row, then count how many rows agree
    int32_t rp[2] = {0, 0};
    FV_HIP(ctx, hipMemcpy(rp, p->rowptr.p + n / 2, sizeof rp, hipMemcpyDeviceToHost));
    if (rp[1] <= rp[0])
        return FV_OK;
    int32_t lastcol = 0;
    FV_HIP(ctx, hipMemcpy(&lastcol, p->colind.p + (rp[1] - 1), sizeof lastcol, hipMemcpyDeviceToHost));
    const int64_t stride = (int64_t)lastcol - n / 2;
    if (stride < 32768 || stride > n / 4)
        return FV_OK; // near-diagonal band (natural order is fine) or no plane structure
    extern int fv_count_far_stride(fv_problem *, int64_t, int64_t *);
    int64_t agree = 0;
    FV_TRY(fv_count_far_stride(p, stride, &agree));
    if (agree < (n - stride) * 8 / 10)
        return FV_OK;
    if (p->dist) { // a row block is traversed by its interior / boundary lists; the stride lets its interior window march
        p->order_stride = stride;
        return FV_OK;
    }
    int64_t BAND = 8192; // rows per band: ~18 grid lines of the 464^3 box; FV_BAND overrides (experiments)
    if (const char *e = getenv("FV_BAND"))
        BAND = atoll(e) > 0 ? (atoll(e) + 63) / 64 * 64 : BAND;
    const int64_t ngroups = (n + 63) >> 6;
    std::vector<int32_t> order;
    order.reserve((size_t)ngroups);
    const int64_t nplanes = (n + stride - 1) / stride;
    for (int64_t b0 = 0; b0 < stride; b0 += BAND) {
        const int64_t b1 = (b0 + BAND < stride) ? b0 + BAND : stride;
        for (int64_t pl = 0; pl < nplanes; pl++) {
            const int64_t lo = pl * stride + b0;
            int64_t hi = pl * stride + b1;
            if (hi > n)
                hi = n;
            if (lo >= hi)
                continue;
            for (int64_t g = (lo + 63) >> 6; (g << 6) < hi; g++) // groups whose first row lies in [lo, hi)
                order.push_back((int32_t)g);
        }
    }
    if ((int64_t)order.size() != ngroups) { // group 0 starts at row 0 in band 0: every group is counted exactly once
        fv_set_error(ctx, "internal: group order covers %zu of %lld groups", order.size(), (long long)ngroups);
        return FV_ERR_STATE;
    }
    FV_TRY(p->group_order.alloc(ctx, (size_t)ngroups));
    FV_HIP(ctx, hipMemcpy(p->group_order.p, order.data(), (size_t)ngroups * sizeof(int32_t), hipMemcpyHostToDevice));
    p->order_stride = stride;
    return FV_OK;
}

__global__ __launch_bounds__(FV_BLOCK) void far_stride_kernel(int64_t n, const int32_t *__restrict__ rowptr,
                                                               const int32_t *__restrict__ colind, int64_t stride,
                                                               unsigned long long *__restrict__ count)
{
    const int64_t r = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    bool hit = false;
    if (r < n) {
        const int32_t e = rowptr[r + 1];
        hit = e > rowptr[r] && (int64_t)colind[e - 1] - r == stride;
    }
    __shared__ int wcount[FV_BLOCK / 64];
    const unsigned long long m = __ballot(hit);
    if ((threadIdx.x & 63) == 0)
        wcount[threadIdx.x >> 6] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        const int c = wcount[0] + wcount[1] + wcount[2] + wcount[3];
        if (c)
            atomicAdd(count, (unsigned long long)c);
    }
}

int fv_count_far_stride(fv_problem *p, int64_t stride, int64_t *agree)
{
    fv_ctx *ctx = p->ctx;
    DevBuf<unsigned long long> cnt;
    FV_TRY(cnt.alloc(ctx, 1));
    FV_TRY(cnt.zero(ctx));
    hipLaunchKernelGGL(far_stride_kernel, dim3(fv_blocks(p->n)), dim3(FV_BLOCK), 0, ctx->stream, p->n, p->rowptr.p, p->colind.p, stride,
                       cnt.p);
    FV_LAUNCH_CHECK(ctx);
    unsigned long long h = 0;
    FV_HIP(ctx, hipMemcpyAsync(&h, cnt.p, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *agree = (int64_t)h;
    return FV_OK;
}

// y = (A + sigma*D) x ; partials != NULL also emits per-block partial sums of x.y.
// use_done: honour the PCG early-exit flag.  vals_override: a value array with the
// shift already folded into the diagonal (then sigma must be passed as 0).
// order_override/npos_override: visit only the listed 64-row groups (distributed interior / boundary passes).
static int spmv_launch_impl(fv_problem *p, const double *x, double *y, double sigma, double *partials, bool use_done,
                            const double *vals_override = nullptr, const int32_t *order_override = nullptr,
                            int64_t npos_override = -1, int *grid_out = nullptr)
{
    fv_ctx *ctx = p->ctx;
    set_resident_blocks(ctx);
    if (!p->order_built)
        FV_TRY(build_group_order(p));
    int G = fv_spmv_grid(p);
    if (npos_override >= 0) { // a grid sized for the listed groups (4 per block), multiple of 8
        int64_t g = ((npos_override + 3) / 4 + 7) / 8 * 8;
        if (g < 8)
            g = 8;
        if (g < G)
            G = (int)g;
    }
    if (grid_out)
        *grid_out = G;
    const double *shift = (sigma != 0.0) ? p->D.p : nullptr;
    const PcgScalars *scal = use_done ? p->scal.p : nullptr;
    const double *vals = vals_override ? vals_override : p->vals.p;
    const int32_t *order = (g_use_order && p->group_order.p) ? p->group_order.p : nullptr;
    int64_t npos = (p->n + 63) >> 6;
    if (npos_override >= 0) {
        order = order_override;
        npos = npos_override;
    }
    const bool stream_form = g_spmv_form == 2 || npos_override >= 0;
#define FV_SPMV_ARGS p->n, p->rowptr.p, p->colind.p, vals, x, y, shift, sigma, partials, scal
    if (stream_form) {
        if (partials && g_nt)
            hipLaunchKernelGGL((spmv_wstream_kernel<512, true, true>), dim3(G), dim3(FV_BLOCK), 0, ctx->stream, FV_SPMV_ARGS, order, npos);
        else if (partials)
            hipLaunchKernelGGL((spmv_wstream_kernel<512, true, false>), dim3(G), dim3(FV_BLOCK), 0, ctx->stream, FV_SPMV_ARGS, order, npos);
        else if (g_nt)
            hipLaunchKernelGGL((spmv_wstream_kernel<512, false, true>), dim3(G), dim3(FV_BLOCK), 0, ctx->stream, FV_SPMV_ARGS, order, npos);
        else
            hipLaunchKernelGGL((spmv_wstream_kernel<512, false, false>), dim3(G), dim3(FV_BLOCK), 0, ctx->stream, FV_SPMV_ARGS, order, npos);
    } else {
        const int lpr = spmv_lpr(p);
#define FV_SPMV_CASE(L, UU)                                                                                              \
    if (partials)                                                                                                        \
        hipLaunchKernelGGL((spmv_kernel<L, UU, true>), dim3(G), dim3(FV_BLOCK), 0, ctx->stream, FV_SPMV_ARGS);            \
    else                                                                                                                 \
        hipLaunchKernelGGL((spmv_kernel<L, UU, false>), dim3(G), dim3(FV_BLOCK), 0, ctx->stream, FV_SPMV_ARGS);
        if (lpr == 4) {
            FV_SPMV_CASE(4, 2)
        } else if (lpr == 16) {
            FV_SPMV_CASE(16, 2)
        } else if (g_spmv_unroll == 8) {
            FV_SPMV_CASE(8, 8)
        } else if (g_spmv_unroll == 4) {
            FV_SPMV_CASE(8, 4)
        } else {
            FV_SPMV_CASE(8, 2)
        }
#undef FV_SPMV_CASE
    }
#undef FV_SPMV_ARGS
    FV_LAUNCH_CHECK(ctx);
    return FV_OK;
}

// ------------------------------------------------------------------ sliced-DIA form of the grid-like part
// A 64-row slice of a grid-structured matrix has only a handful of distinct column
// offsets (col - row): 7 for the 7-point stencil, also across line ends and next to
// Dirichlet cells, where some rows merely lack some of them.  For every slice with at
// most DIA_K distinct offsets the values are kept a second time lane-major,
//      sval[(slice_pos*DIA_K + k)*64 + lane] = A[row, row + off_k]   (0 where absent),
// and the SpMV needs no column indices, no row pointers, no LDS and no cross-lane
// reduction: lane = row, every value load and every x load of a step is one contiguous
// 512-byte access.  Entry traffic drops from 12 to 8 bytes.  Slices with more offsets
// (irregular meshes, rows longer than DIA_K) stay with the CSR wave-stream kernel, which
// then runs over the list of remaining 64-row groups.  The terms of a row are summed in
// ascending column order, exactly like the CSR kernels.
constexpr int DIA_K = 8;

__global__ __launch_bounds__(FV_BLOCK) void dia_pattern_kernel(int64_t n, const int32_t *__restrict__ rowptr,
                                                                const int32_t *__restrict__ colind, uint8_t *__restrict__ sl_noff,
                                                                int32_t *__restrict__ sl_off, int32_t *__restrict__ is_dia,
                                                                int32_t *__restrict__ is_csr)
{
    constexpr int WPB = FV_BLOCK / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t nslices = (n + 63) >> 6;
    const int64_t sl = (int64_t)blockIdx.x * WPB + wave;
    if (sl >= nslices)
        return;
    const int64_t row = (sl << 6) + lane;
    int32_t o[DIA_K];
    int len = 0;
    int32_t k0 = 0;
    if (row < n) {
        k0 = rowptr[row];
        len = rowptr[row + 1] - k0;
    }
#pragma unroll
    for (int k = 0; k < DIA_K; k++)
        o[k] = (k < len && k < DIA_K) ? (int32_t)((int64_t)colind[k0 + k] - row) : 0x7fffffff;
    const bool toolong = __any(len > DIA_K);
    int32_t last = -0x7fffffff - 1;
    int count = 0;
    bool ok = !toolong;
    int32_t found[DIA_K];
    while (ok) { // distinct offsets of the slice in ascending order
        int32_t m = 0x7fffffff;
#pragma unroll
        for (int k = 0; k < DIA_K; k++)
            if (o[k] > last && o[k] < m)
                m = o[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const int32_t t = __shfl_xor(m, off, 64);
            m = t < m ? t : m;
        }
        if (m == 0x7fffffff)
            break;
        if (count == DIA_K) {
            ok = false;
            break;
        }
#pragma unroll
        for (int k = 0; k < DIA_K; k++)
            if (k == count)
                found[k] = m;
        count++;
        last = m;
    }
    if (count == 0)
        ok = false; // nothing stored in this slice: leave it to the CSR kernel (which writes the zeros)
    if (lane == 0) {
        sl_noff[sl] = ok ? (uint8_t)count : 0;
        is_dia[sl] = ok ? 1 : 0;
        is_csr[sl] = ok ? 0 : 1;
    }
    if (ok && lane < DIA_K) {
        int32_t v = 0;
#pragma unroll
        for (int k = 0; k < DIA_K; k++)
            if (k == lane && k < count)
                v = found[k];
        sl_off[sl * DIA_K + lane] = v;
    }
}

// sval <- the (possibly diagonal-folded) CSR values, lane-major per slice
__global__ __launch_bounds__(FV_BLOCK) void dia_fill_kernel(int64_t n, int64_t ndia, const int32_t *__restrict__ dia_list,
                                                             const uint8_t *__restrict__ sl_noff, const int32_t *__restrict__ sl_off,
                                                             const int32_t *__restrict__ rowptr, const int32_t *__restrict__ colind,
                                                             const double *__restrict__ vals, const int32_t *__restrict__ dia_pos,
                                                             double *__restrict__ sval)
{
    constexpr int WPB = FV_BLOCK / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t pos = (int64_t)blockIdx.x * WPB + wave;
    if (pos >= ndia)
        return;
    const int64_t sl = dia_list[pos];
    const int64_t base = dia_pos[sl];
    const int64_t row = (sl << 6) + lane;
    const int noff = sl_noff[sl];
    int32_t ptr = 0, end = 0;
    if (row < n) {
        ptr = rowptr[row];
        end = rowptr[row + 1];
    }
    for (int k = 0; k < noff; k++) { // rows are short (<= DIA_K) and, in a row block, not necessarily sorted (halo columns)
        const int32_t off = sl_off[sl * DIA_K + k];
        double v = 0.0;
        for (int32_t j = ptr; j < end; j++)
            if ((int64_t)colind[j] - row == off) {
                v = vals[j];
                break;
            }
        sval[(base + k) * 64 + lane] = v;
    }
}

template <bool DOT, bool NT, bool INIT>
__global__ __launch_bounds__(FV_BLOCK) void spmv_dia_kernel(int64_t n, int64_t ncols, int64_t ndia, const int32_t *__restrict__ dia_list,
                                                             const int32_t *__restrict__ dia_pos, const uint8_t *__restrict__ sl_noff,
                                                             const int32_t *__restrict__ sl_off,
                                                             const double *__restrict__ sval, const double *__restrict__ x,
                                                             double *__restrict__ y, const double *__restrict__ shift, double sigma,
                                                             double *__restrict__ partials, const PcgScalars *__restrict__ scal,
                                                             StepInitEpilogue epi)
{
    constexpr int WPB = FV_BLOCK / 64;
    __shared__ double smem[4];
    if (scal && scal->done)
        return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t per_xcd = (ndia + 7) >> 3;
    const int64_t pstride = (int64_t)(gridDim.x >> 3) * WPB;
    const int64_t xbase = (int64_t)(blockIdx.x & 7) * per_xcd;
    const int64_t xend = (xbase + per_xcd < ndia) ? xbase + per_xcd : ndia;
    double dacc = 0.0, arr = 0.0, abb = 0.0;
    for (int64_t pos = xbase + (int64_t)(blockIdx.x >> 3) * WPB + wave; pos < xend; pos += pstride) {
        const int64_t sl = dia_list[pos];
        const int64_t row = (sl << 6) + lane;
        const int noff = __builtin_amdgcn_readfirstlane((int)sl_noff[sl]);
        const int32_t offs = (lane < DIA_K) ? sl_off[sl * DIA_K + lane] : 0;
        const double *sv = sval + (int64_t)dia_pos[sl] * 64 + lane; // dia_pos: start of the slice's values in units of 64 doubles (packed, no padding)
        double v[DIA_K], xv[DIA_K];
#pragma unroll
        for (int k = 0; k < DIA_K; k++) {
            v[k] = 0.0;
            xv[k] = 0.0;
            if (k < noff) {
                const int32_t off = __builtin_amdgcn_readlane(offs, k);
                int64_t c = row + off;
                c = c < 0 ? 0 : (c >= ncols ? ncols - 1 : c); // absent entries (value 0) near the ends may point outside; ncols = n + halo slots
                v[k] = NT ? __builtin_nontemporal_load(sv + k * 64) : sv[k * 64];
                xv[k] = x[c];
            }
        }
        double sum = 0.0;
#pragma unroll
        for (int k = 0; k < DIA_K; k++)
            if (k < noff)
                sum += v[k] * xv[k];
        if (row < n) {
            if (INIT) {
                const double di = epi.D[row];
                double bi = epi.bprime ? epi.bprime[row] : 0.0;
                if (epi.b_times_D)
                    bi *= di;
                const double rhs = bi + di * (x[row] / epi.dt);
                const double ri = epi.q_shifted ? rhs - sum : bi - sum;
                double mi;
                if (epi.compute_minv) {
                    {
                        const double dd = epi.diagA[row] + epi.sigma * di;
                        mi = dd > 0.0 ? 1.0 / dd : 0.0;
                    }
                    epi.minv[row] = mi;
                } else
                    mi = epi.minv[row];
                const double zi = mi * ri;
                epi.r[row] = ri;
                epi.pv[row] = zi;
                dacc += ri * zi;
                arr += ri * ri;
                abb += rhs * rhs;
            } else {
                const double xr = (shift || DOT) ? x[row] : 0.0;
                if (shift)
                    sum += sigma * shift[row] * xr;
                if (NT)
                    __builtin_nontemporal_store(sum, y + row);
                else
                    y[row] = sum;
                if (DOT)
                    dacc += xr * sum;
            }
        }
    }
    if (INIT) {
        const double t0 = block_sum(dacc, smem);
        const double t1 = block_sum(arr, smem);
        const double t2 = block_sum(abb, smem);
        if (threadIdx.x == 0) {
            epi.part_rz[blockIdx.x] = t0;
            epi.part_rr[blockIdx.x] = t1;
            epi.part_bb[blockIdx.x] = t2;
        }
    } else if (DOT) {
        const double tsum = block_sum(dacc, smem);
        if (threadIdx.x == 0)
            partials[blockIdx.x] = tsum;
    }
}

// Sliced-DIA SpMV for operators with a plane stride (structured grids), marching along the plane direction.
// A wave owns a "pencil": the slices s0, s0 + step, s0 + 2 step, ... with step = (stride - shift) / 64 and
// shift = stride mod 64, i.e. the same 64 in-plane positions (moving by `shift` rows per plane) of consecutive planes.
// Then the -plane arm of the current slice is the previous slice's centre and the +plane arm is the next slice's
// centre, both moved by `shift` lanes: they are taken from registers (wave shuffle), only the `shift` lanes that
// fall off the end are loaded.  Every x line is therefore fetched once for the three plane-direction uses, however
// short-lived it is in L2 (a 4 MiB L2 turns over in ~6 us at this rate, far less than the time between planes in any
// slice-by-slice traversal).  Work items are (pencil, segment of `seglen` plane steps); XCD k owns the segments
// [k m, (k+1) m) and its resident waves march through neighbouring pencils of one segment together, so the in-plane
// arms (+-1, +-line) are shared through L2 as before.  Only slices inside [win_lo, win_hi) are computed (a row block's
// interior pass); the others are merely walked through.
// One 16-byte access per lane covering the 128 x elements [64 slice - 32, 64 slice + 96): lane l holds elements 2l, 2l+1 of
// that window.  Lanes whose elements would fall outside [0, ncols) are clamped: they hold misplaced but finite values that
// only ever meet absent (zero) matrix entries.
__device__ inline double2 march_window(const double *__restrict__ x, int64_t slice, int lane, int32_t ncols)
{
    int32_t i = (int32_t)(slice << 6) - 32 + 2 * lane;
    const int32_t hi = (ncols - 2) & ~1;
    i = i < 0 ? 0 : (i > hi ? hi : i);
    return *reinterpret_cast<const double2 *>(x + i);
}
// element j (0..127, per lane) of such a window
__device__ inline double march_window_elem(double2 w, int j)
{
    const double a = __shfl(w.x, j >> 1, 64), b = __shfl(w.y, j >> 1, 64);
    return (j & 1) ? b : a;
}

template <bool DOT, bool NT, bool WIDE>
__global__ __launch_bounds__(FV_BLOCK, 8) void spmv_dia_march_kernel(int64_t n, int64_t ncols, int64_t nslices, int64_t step, int shift, int64_t stride,
                                                                   int seglen, int segs_per_xcd, int dbg, int64_t win_lo, int64_t win_hi,
                                                                   const int32_t *__restrict__ dia_pos,
                                                                   const uint8_t *__restrict__ sl_noff, const int32_t *__restrict__ sl_off,
                                                                   const double *__restrict__ sval, const double *__restrict__ x,
                                                                   double *__restrict__ y, const double *__restrict__ dshift, double sigma,
                                                                   double *__restrict__ partials, const PcgScalars *__restrict__ scal)
{
    constexpr int WPB = FV_BLOCK / 64;
    __shared__ double smem[4];
    if (scal && scal->done)
        return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int xcd = (int)(blockIdx.x & 7);
    const int64_t wstride = (int64_t)(gridDim.x >> 3) * WPB;    // waves of this XCD
    const int64_t nitems = (int64_t)segs_per_xcd * step;         // (segment, pencil) pairs of this XCD
    double dacc = 0.0;
    for (int64_t item = (int64_t)(blockIdx.x >> 3) * WPB + wave; item < nitems; item += wstride) {
        const int64_t seg = (int64_t)xcd * segs_per_xcd + item / step;
        const int64_t pc = item % step;
        int64_t sl = pc + seg * seglen * step;
        if (sl >= nslices)
            continue;
        // rows and columns fit int32 (device indices are int32): keeps the address arithmetic in one register
        const int32_t nc32 = (int32_t)ncols, st32 = (int32_t)stride;
        double prevc = 0.0, curc, nextc;
        // WIDE (0 < shift <= 32): one window access per step brings the next centre AND the `shift` elements on either side
        // of it that the plane arms need beyond the register shuffle: m0 / m1 = the -plane arm of this / the next slice
        // (taken from the windows of the slices before them), parm = the +plane arm of this slice (from the next window)
        double m0 = 0.0, m1 = 0.0, parm = 0.0;
        bool have_m0 = false;
        if (WIDE) {
            const double2 w0 = march_window(x, sl, lane, nc32);
            curc = march_window_elem(w0, 32 + lane);
            m1 = march_window_elem(w0, 32 - shift + lane);
            if (sl - step >= 0) {
                const double2 wp = march_window(x, sl - step, lane, nc32);
                m0 = march_window_elem(wp, 32 - shift + lane);
                have_m0 = true;
            }
        } else {
            const int32_t r0 = (int32_t)(sl << 6) + lane;
            curc = r0 < nc32 ? x[r0] : 0.0;
        }
        bool have_prev = false;
        // the slice's metadata is fetched one step ahead too, so that a step waits for one memory round trip
        // (values + arms), not three (pattern -> offsets -> values)
        int nx_noff = (int)sl_noff[sl];
        int32_t nx_offs = (lane < DIA_K) ? sl_off[sl * DIA_K + lane] : 0;
        int32_t nx_pos = dia_pos[sl];
        for (int k = 0; k < seglen && sl < nslices; k++, sl += step) {
            const int32_t row = (int32_t)(sl << 6) + lane;
            const int64_t nsl = sl + step;
            const bool have_next = nsl < nslices;
            const int noff = __builtin_amdgcn_readfirstlane(nx_noff);
            const int32_t offs = nx_offs;
            const int64_t pos = __builtin_amdgcn_readfirstlane(nx_pos);
            double mnew = 0.0;
            {
                if (WIDE) {
                    nextc = 0.0;
                    if (have_next) {
                        const double2 w = march_window(x, nsl, lane, nc32);
                        nextc = march_window_elem(w, 32 + lane);
                        parm = march_window_elem(w, 32 + shift + lane);
                        mnew = march_window_elem(w, 32 - shift + lane);
                    }
                } else {
                    const int32_t rn = (int32_t)(nsl << 6) + lane;
                    nextc = (have_next && rn < nc32) ? x[rn] : 0.0;
                }
                if (have_next && k + 1 < seglen) {
                    nx_noff = (int)sl_noff[nsl];
                    nx_offs = (lane < DIA_K) ? sl_off[nsl * DIA_K + lane] : 0;
                    nx_pos = dia_pos[nsl];
                }
            }
            if (noff > 0 && sl >= win_lo && sl < win_hi) { // the window: all slices, or the interior ones of a row block
                const double *sv = sval + pos * 64 + lane;
                double sum = 0.0;
#pragma unroll
                for (int j = 0; j < DIA_K; j++) {
                    if (j < noff) {
                        const int32_t off = __builtin_amdgcn_readlane(offs, j);
                        const double v = NT ? __builtin_nontemporal_load(sv + j * 64) : sv[j * 64];
                        double xv;
                        if (off == 0)
                            xv = curc;
                        else if (WIDE && off == -st32 && have_m0)
                            xv = m0;
                        else if (WIDE && off == st32 && have_next)
                            xv = parm;
                        else if (!WIDE && off == -st32 && have_prev) {
                            // lanes >= shift: the previous slice's centre, `shift` lanes down; the first `shift` lanes: loaded.
                            // Every lane issues the load (the others re-read the slice's first row) so that there is no
                            // divergent branch — a masked load inside one costs 6 % of the kernel (exec-mask bookkeeping
                            // serialises the loads in flight)
                            const double sh = __shfl(prevc, (lane - shift) & 63, 64);
                            int32_t c = lane < shift ? row - st32 : row - lane;
                            c = c < 0 ? 0 : c;
                            const double ld = (dbg & 2) ? 0.0 : x[c];
                            xv = lane < shift ? ld : sh;
                        } else if (!WIDE && off == st32 && have_next) {
                            const double sh = __shfl(nextc, (lane + shift) & 63, 64);
                            int32_t c = lane + shift >= 64 ? row + st32 : row - lane + 63;
                            c = c >= nc32 ? nc32 - 1 : c;
                            const double ld = (dbg & 2) ? 0.0 : x[c];
                            xv = lane + shift >= 64 ? ld : sh;
                        } else {
                            if (dbg & 1) // diagnosis only (wrong results): what the in-plane arm loads cost
                                xv = curc;
                            else {
                                int32_t c = row + off; // |off| <= stride < n/4: no overflow
                                c = c < 0 ? 0 : (c >= nc32 ? nc32 - 1 : c);
                                xv = x[c];
                            }
                        }
                        sum += v * xv;
                    }
                }
                if (row < n) {
                    if (dshift)
                        sum += sigma * dshift[row] * curc;
                    if (NT)
                        __builtin_nontemporal_store(sum, y + row);
                    else
                        y[row] = sum;
                    if (DOT)
                        dacc += curc * sum;
                }
            }
            prevc = curc;
            curc = nextc;
            have_prev = true;
            if (WIDE) {
                m0 = m1;
                m1 = mnew;
                have_m0 = true;
            }
        }
    }
    if (DOT) {
        const double tsum = block_sum(dacc, smem);
        if (threadIdx.x == 0)
            partials[blockIdx.x] = tsum;
    }
}

// values are packed: slice i of the list holds sl_noff lane-major blocks of 64 doubles, one after the other
__global__ __launch_bounds__(FV_BLOCK) void dia_len_kernel(int64_t ndia, const int32_t *__restrict__ dia_list, const uint8_t *__restrict__ sl_noff,
                                                            int pad_to, int32_t *__restrict__ len)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i < ndia)
        len[i] = pad_to > 0 ? pad_to : sl_noff[dia_list[i]];
}

__global__ __launch_bounds__(FV_BLOCK) void dia_pos_kernel(int64_t ndia, const int32_t *__restrict__ dia_list, const int32_t *__restrict__ start,
                                                            int32_t *__restrict__ dia_pos)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i < ndia)
        dia_pos[dia_list[i]] = start[i];
}

// traversal order of the DIA slices: the plane-blocked group order (build_group_order) restricted to the DIA slices
__global__ __launch_bounds__(FV_BLOCK) void dia_order_flag_kernel(int64_t ng, const int32_t *__restrict__ order, const uint8_t *__restrict__ sl_noff,
                                                                   int32_t *__restrict__ flag)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i < ng)
        flag[i] = sl_noff[order[i]] > 0;
}

__global__ __launch_bounds__(FV_BLOCK) void dia_order_gather_kernel(int64_t m, const int32_t *__restrict__ idx, const int32_t *__restrict__ order,
                                                                     int32_t *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i < m)
        out[i] = order[idx[i]];
}

int fv_build_dia(fv_problem *p)
{
    fv_ctx *ctx = p->ctx;
    p->dia_built = true;
    p->ndia = 0;
    p->ncsr_groups = (p->n + 63) >> 6;
    const int64_t ns = (p->n + 63) >> 6;
    if (ns == 0 || p->nnz == 0 || p->n < 4096)
        return FV_OK; // tiny problems are launch-bound: one kernel is better than two
    DevBuf<int32_t> fd, fc;
    FV_TRY(p->sl_noff.alloc(ctx, (size_t)ns));
    FV_TRY(p->sl_off.alloc(ctx, (size_t)ns * DIA_K));
    FV_TRY(fd.alloc(ctx, (size_t)ns));
    FV_TRY(fc.alloc(ctx, (size_t)ns));
    hipLaunchKernelGGL(dia_pattern_kernel, dim3(fv_blocks(ns, FV_BLOCK / 64)), dim3(FV_BLOCK), 0, ctx->stream, p->n, p->rowptr.p, p->colind.p,
                       p->sl_noff.p, p->sl_off.p, fd.p, fc.p);
    FV_LAUNCH_CHECK(ctx);
    FV_TRY(p->dia_list.alloc(ctx, (size_t)ns));
    FV_TRY(p->csr_list.alloc(ctx, (size_t)ns));
    FV_TRY(fv_compact_flags(ctx, fd.p, ns, p->dia_list.p, &p->ndia));
    FV_TRY(fv_compact_flags(ctx, fc.p, ns, p->csr_list.p, &p->ncsr_groups));
    if (p->ndia * 2 < ns) { // mostly irregular: keep the pure CSR form
        p->ndia = 0;
        p->ncsr_groups = ns;
        p->dia_list.release();
        p->csr_list.release();
        p->sl_noff.release();
        p->sl_off.release();
        return FV_OK;
    }
    {
        DevBuf<int32_t> len, start;
        FV_TRY(len.alloc(ctx, (size_t)p->ndia));
        FV_TRY(start.alloc(ctx, (size_t)p->ndia + 1));
        hipLaunchKernelGGL(dia_len_kernel, dim3(fv_blocks(p->ndia)), dim3(FV_BLOCK), 0, ctx->stream, p->ndia, (const int32_t *)p->dia_list.p,
                           (const uint8_t *)p->sl_noff.p, g_dia_packed ? 0 : DIA_K, len.p);
        FV_LAUNCH_CHECK(ctx);
        int64_t nblocks = 0; // blocks of 64 doubles in all
        FV_TRY(fv_exclusive_scan_i32(ctx, len.p, start.p, p->ndia, &nblocks));
        FV_TRY(p->dia_vals.alloc(ctx, (size_t)nblocks * 64 + 64));
        FV_TRY(p->dia_pos.alloc(ctx, (size_t)ns));
        hipLaunchKernelGGL(dia_pos_kernel, dim3(fv_blocks(p->ndia)), dim3(FV_BLOCK), 0, ctx->stream, p->ndia, (const int32_t *)p->dia_list.p,
                           (const int32_t *)start.p, p->dia_pos.p);
        FV_LAUNCH_CHECK(ctx);
        FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    p->dia_epoch = -1;
    if (p->group_order.p) { // walk the slices band by band, plane after plane: the +-plane x arms are then re-used while still in L2
        DevBuf<int32_t> flag, idx;
        FV_TRY(flag.alloc(ctx, (size_t)ns));
        FV_TRY(idx.alloc(ctx, (size_t)ns));
        hipLaunchKernelGGL(dia_order_flag_kernel, dim3(fv_blocks(ns)), dim3(FV_BLOCK), 0, ctx->stream, ns, (const int32_t *)p->group_order.p,
                           (const uint8_t *)p->sl_noff.p, flag.p);
        FV_LAUNCH_CHECK(ctx);
        int64_t cnt = 0;
        FV_TRY(fv_compact_flags(ctx, flag.p, ns, idx.p, &cnt));
        if (cnt == p->ndia) {
            FV_TRY(p->dia_list_ord.alloc(ctx, (size_t)cnt));
            hipLaunchKernelGGL(dia_order_gather_kernel, dim3(fv_blocks(cnt)), dim3(FV_BLOCK), 0, ctx->stream, cnt, (const int32_t *)idx.p,
                               (const int32_t *)p->group_order.p, p->dia_list_ord.p);
            FV_LAUNCH_CHECK(ctx);
            FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        }
    }
    return FV_OK;
}

// lane-major values for the current assembly; src_tag identifies the source array (0 = plain, else the folded sigma)
static int ensure_dia_vals(fv_problem *p, const double *src, double src_tag)
{
    fv_ctx *ctx = p->ctx;
    if (p->dia_epoch == p->assemble_epoch && p->dia_tag == src_tag)
        return FV_OK;
    hipLaunchKernelGGL(dia_fill_kernel, dim3(fv_blocks(p->ndia, FV_BLOCK / 64)), dim3(FV_BLOCK), 0, ctx->stream, p->n, p->ndia, p->dia_list.p,
                       p->sl_noff.p, p->sl_off.p, p->rowptr.p, p->colind.p, src, (const int32_t *)p->dia_pos.p, p->dia_vals.p);
    FV_LAUNCH_CHECK(ctx);
    p->dia_epoch = p->assemble_epoch;
    p->dia_tag = src_tag;
    return FV_OK;
}


static int stream_grid(int64_t npos)
{
    int64_t g = ((npos + 3) / 4 + 7) / 8 * 8; // 4 groups per block and pass, multiple of 8 (XCD shares)
    if (g < 8)
        g = 8;
    if (g > g_resident_blocks)
        g = g_resident_blocks;
    return (int)g;
}

// One launch of the CSR wave-stream kernel over `npos` 64-row groups (all of them in `order`, or the listed ones).
static int launch_wstream(fv_problem *p, int G, const double *vals, const double *x, double *y, const double *shift, double sigma,
                          int mode, double *partials, const PcgScalars *scal, const int32_t *order, int64_t npos, const StepInitEpilogue &epi)
{
    fv_ctx *ctx = p->ctx;
#define FV_WS(D_, N_, I_)                                                                                                            \
    hipLaunchKernelGGL((spmv_wstream_kernel<512, D_, N_, I_>), dim3(G), dim3(FV_BLOCK), 0, ctx->stream, p->n, p->rowptr.p, p->colind.p, \
                       vals, x, y, shift, sigma, partials, scal, order, npos, epi)
    if (mode == SPMV_INIT) {
        if (g_nt)
            FV_WS(false, true, true);
        else
            FV_WS(false, false, true);
    } else if (mode == SPMV_DOT) {
        if (g_nt)
            FV_WS(true, true, false);
        else
            FV_WS(true, false, false);
    } else {
        if (g_nt)
            FV_WS(false, true, false);
        else
            FV_WS(false, false, false);
    }
#undef FV_WS
    FV_LAUNCH_CHECK(ctx);
    return FV_OK;
}

static StepInitEpilogue offset_epilogue(StepInitEpilogue e, int off)
{
    e.part_rz += off;
    e.part_rr += off;
    e.part_bb += off;
    return e;
}

// y = (A + sigma*D) x over the whole operator.  mode SPMV_DOT also leaves per-block partials of x.y in
// `partials`; SPMV_INIT runs the step set-up epilogue instead of writing y.  `vals_override`: value array
// with the shift already folded into the diagonal (sigma must then be 0).  *nparts = partials written.
// Subset of the operator's 64-row groups for the distributed interior / boundary passes.

int spmv_apply(fv_problem *p, const double *x, double *y, double sigma, const double *vals_override, int mode, double *partials,
               const StepInitEpilogue *epi_in, bool use_done, int *nparts, const GroupSubset *subset)
{
    fv_ctx *ctx = p->ctx;
    set_resident_blocks(ctx);
    if (!p->order_built)
        FV_TRY(build_group_order(p));
    if (!p->dia_built)
        FV_TRY(fv_build_dia(p));
    const double *shift = (sigma != 0.0) ? p->D.p : nullptr;
    const PcgScalars *scal = use_done ? p->scal.p : nullptr;
    const double *vals = vals_override ? vals_override : p->vals.p;
    StepInitEpilogue epi = epi_in ? *epi_in : StepInitEpilogue{};
    const int64_t ngroups = (p->n + 63) >> 6;
    if (g_spmv_form != 2) { // lanes-per-row form (A/B only): no epilogues
        if (mode == SPMV_INIT) {
            fv_set_error(ctx, "internal: fused set-up needs the stream form");
            return FV_ERR_STATE;
        }
        int G = 0;
        FV_TRY(spmv_launch_impl(p, x, y, sigma, mode == SPMV_DOT ? partials : nullptr, use_done, vals_override, nullptr, -1, &G));
        if (nparts)
            *nparts = G;
        return FV_OK;
    }
    if (subset && !(g_use_dia && p->ndia > 0)) { // subset of a pure-CSR operator: everything is in subset->csr
        const int G = stream_grid(subset->ncsr);
        if (subset->ncsr > 0)
            FV_TRY(launch_wstream(p, G, vals, x, y, shift, sigma, mode, partials, scal, subset->csr, subset->ncsr, epi));
        if (nparts)
            *nparts = subset->ncsr > 0 ? G : 0;
        return FV_OK;
    }
    if (g_use_dia && p->ndia > 0) {
        FV_TRY(ensure_dia_vals(p, vals, vals_override ? p->shifted_sigma : 0.0));
        const int32_t *dlist = subset ? subset->dia : ((g_use_order && p->dia_list_ord.p) ? p->dia_list_ord.p : p->dia_list.p);
        const int64_t dcount = subset ? subset->ndia : p->ndia;
        const int32_t *clist = subset ? subset->csr : p->csr_list.p;
        const int64_t ccount = subset ? subset->ncsr : p->ncsr_groups;
        const int GA = dcount > 0 ? stream_grid(dcount) : 0;
#define FV_DIA(D_, N_, I_)                                                                                                        \
    hipLaunchKernelGGL((spmv_dia_kernel<D_, N_, I_>), dim3(GA), dim3(FV_BLOCK), 0, ctx->stream, p->n, p->n + p->nhalo, dcount, dlist, p->dia_pos.p, p->sl_noff.p, \
                       p->sl_off.p, p->dia_vals.p, x, y, shift, sigma, partials, scal, epi)
        // structured grids: plane-marching form over the whole DIA part (not for subsets or the fused set-up)
        // ... when it pays: while x (8 bytes per column) stays in the 256 MB last-level cache the -plane / +plane arms of the
        // slice-by-slice kernel come from there and the marching kernel's static partition only costs (short pencils, uneven
        // XCD shares): back-to-back launches, 177 vs 216 us on 1.2e7 rows, 352 vs 364 on 2.5e7, 505 vs 539 on 3.4e7, 777 vs 758 on
        // 5e7, ~2000 vs 1600 on 1e8 (tools/march_vs_dia.py, profiles/r01_march_vs_dia.log); inside the stepping loop, where the
        // vector pass between two SpMVs evicts x, ms per step slices vs marching: 216^3 0.306 / 0.330, 256^3 0.509 / 0.517,
        // 280^3 0.710 / 0.705, 320^3 1.067 / 1.056, 380^3 1.785 / 1.738 (tools/step_ab.py 9 0 2, profiles/r01_step_ab_march.log)
        const bool march_pays = g_march == 2 || (p->n + p->nhalo) * (int64_t)sizeof(double) > (int64_t)g_march_min_mb * 1048576;
        const bool march = g_march && march_pays && mode != SPMV_INIT && p->order_stride >= 4096 && dcount > 0 && (!subset || subset->win_hi > subset->win_lo);
        int GM = 0;
        if (g_trace_spmv > 0) { // fv_tune key 25 / FV_TRACE_SPMV: the next N kernel choices to stderr
            g_trace_spmv--;
            fprintf(stderr, "[fvhip] sliced-DIA SpMV: %s kernel, n %lld (+%lld halo), %lld slices%s, plane stride %lld, window [%lld, %lld)\n",
                    march ? "plane-marching" : "slice-by-slice", (long long)p->n, (long long)p->nhalo, (long long)dcount,
                    subset ? " (subset)" : "", (long long)p->order_stride, subset ? (long long)subset->win_lo : 0LL,
                    subset ? (long long)subset->win_hi : 0LL);
        }
        if (march) {
            const int64_t ns = (p->n + 63) >> 6;
            const int sh = (int)(p->order_stride % 64);
            const int64_t step = (p->order_stride - sh) / 64;
            const int64_t nk = (ns + step - 1) / step;                     // plane steps of the longest pencil
            // m segments per XCD: a static partition pays for a partly filled last round of the XCD's resident waves, short
            // segments pay for their start-up loads: the smallest m whose m * step (pencil, segment) items fill >= 95 % of
            // whole rounds, else the best filling one
            int segs_per_xcd = g_march_segs;
            if (segs_per_xcd <= 0) {
                double best = 0.0;
                for (int m = 1; m <= 8; m++) {
                    const int64_t items = (int64_t)m * step;
                    int64_t gg = ((items + 3) / 4) * 8;
                    if (gg > g_resident_blocks)
                        gg = g_resident_blocks;
                    const int64_t waves = gg / 8 * 4;
                    const double eff = (double)items / (double)(((items + waves - 1) / waves) * waves);
                    if (eff > best) {
                        best = eff;
                        segs_per_xcd = m;
                    }
                    if (eff >= 0.95)
                        break;
                }
            }
            const int seglen = (int)((nk + 8 * segs_per_xcd - 1) / (8 * segs_per_xcd));
            const int64_t per_xcd = (int64_t)segs_per_xcd * step;
            int64_t g = ((per_xcd + 3) / 4) * 8;
            if (g > g_resident_blocks)
                g = g_resident_blocks;
            GM = (int)g;
#define FV_MARCH_W(D_, N_, W_)                                                                                                                 \
    hipLaunchKernelGGL((spmv_dia_march_kernel<D_, N_, W_>), dim3(GM), dim3(FV_BLOCK), 0, ctx->stream, p->n, p->n + p->nhalo, ns, step, sh, p->order_stride, \
                       seglen, segs_per_xcd, g_march_dbg, subset ? subset->win_lo : (int64_t)0, subset ? subset->win_hi : ns, (const int32_t *)p->dia_pos.p, (const uint8_t *)p->sl_noff.p, (const int32_t *)p->sl_off.p,        \
                       (const double *)p->dia_vals.p, x, y, shift, sigma, partials, scal)
#define FV_MARCH(D_, N_)                                                                                                                      \
    do {                                                                                                                                      \
        if (g_march_wide && sh > 0 && sh <= 32)                                                                                               \
            FV_MARCH_W(D_, N_, true);                                                                                                         \
        else                                                                                                                                  \
            FV_MARCH_W(D_, N_, false);                                                                                                        \
    } while (0)
            if (mode == SPMV_DOT) {
                if (g_nt)
                    FV_MARCH(true, true);
                else
                    FV_MARCH(true, false);
            } else {
                if (g_nt)
                    FV_MARCH(false, true);
                else
                    FV_MARCH(false, false);
            }
#undef FV_MARCH_W
#undef FV_MARCH
        }
        if (dcount > 0 && !march) {
        if (mode == SPMV_INIT) {
            if (g_nt)
                FV_DIA(false, true, true);
            else
                FV_DIA(false, false, true);
        } else if (mode == SPMV_DOT) {
            if (g_nt)
                FV_DIA(true, true, false);
            else
                FV_DIA(true, false, false);
        } else {
            if (g_nt)
                FV_DIA(false, true, false);
            else
                FV_DIA(false, false, false);
        }
        }
#undef FV_DIA
        FV_LAUNCH_CHECK(ctx);
        const int GD = march ? GM : GA; // partials written by the DIA part
        int GB = 0;
        if (ccount > 0) {
            GB = stream_grid(ccount);
            FV_TRY(launch_wstream(p, GB, vals, x, y, shift, sigma, mode, partials ? partials + GD : nullptr, scal, clist, ccount,
                                  offset_epilogue(epi, GD)));
        }
        if (nparts)
            *nparts = GD + GB;
        return FV_OK;
    }
    const int G = stream_grid(ngroups);
    const int32_t *order = (g_use_order && p->group_order.p) ? p->group_order.p : nullptr;
    FV_TRY(launch_wstream(p, G, vals, x, y, shift, sigma, mode, partials, scal, order, ngroups, epi));
    if (nparts)
        *nparts = G;
    return FV_OK;
}


int fv_spmv_launch(fv_problem *p, const double *x, double *y, double sigma, double *partials_or_null, bool fold, int *npartials)
{
    const double *folded = nullptr;
    if (fold && sigma != 0.0)
        FV_TRY(ensure_folded(p, sigma, &folded));
    return spmv_apply(p, x, y, folded ? 0.0 : sigma, folded, partials_or_null ? SPMV_DOT : SPMV_PLAIN, partials_or_null, nullptr, false, npartials);
}

// vals_shifted = vals with sigma*D added to every stored diagonal entry
__global__ __launch_bounds__(FV_BLOCK) void fold_shift_kernel(int64_t n, const int32_t *__restrict__ diagpos, const double *__restrict__ D,
                                                               double sigma, double *__restrict__ vals_shifted, int *__restrict__ missing)
{
    const int64_t r = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (r >= n)
        return;
    const int32_t dp = diagpos[r];
    if (dp >= 0)
        vals_shifted[dp] += sigma * D[r];
    else
        *missing = 1;
}


// Returns the folded value array for this sigma (building it if needed), or nullptr when folding is not possible.
int ensure_folded(fv_problem *p, double sigma, const double **out)
{
    fv_ctx *ctx = p->ctx;
    *out = nullptr;
    if (!g_fold_shift || p->fold_ok == 0 || p->nnz == 0)
        return FV_OK;
    if (p->vals_shifted.p && p->shifted_sigma == sigma && p->shifted_epoch == p->assemble_epoch) {
        *out = p->vals_shifted.p;
        return FV_OK;
    }
    if (!p->vals_shifted.p)
        FV_TRY(p->vals_shifted.alloc(ctx, (size_t)p->nnz + 2));
    FV_HIP(ctx, hipMemcpyAsync(p->vals_shifted.p, p->vals.p, ((size_t)p->nnz + 2) * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    DevBuf<int> miss;
    FV_TRY(miss.alloc(ctx, 1));
    FV_TRY(miss.zero(ctx));
    hipLaunchKernelGGL(fold_shift_kernel, dim3(fv_blocks(p->n)), dim3(FV_BLOCK), 0, ctx->stream, p->n, p->diagpos.p, p->D.p, sigma,
                       p->vals_shifted.p, miss.p);
    FV_LAUNCH_CHECK(ctx);
    int h = 0;
    FV_HIP(ctx, hipMemcpyAsync(&h, miss.p, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (h) { // a free row without a stored diagonal cannot carry the shift
        p->fold_ok = 0;
        p->vals_shifted.release();
        return FV_OK;
    }
    p->fold_ok = 1;
    p->shifted_sigma = sigma;
    p->shifted_epoch = p->assemble_epoch;
    *out = p->vals_shifted.p;
    return FV_OK;
}

