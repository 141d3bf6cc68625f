// CSR SpMV and the fused Jacobi-preconditioned conjugate gradient.
//
// Replaces IterativeSolvers.cg!/cg + AlgebraicMultigrid at
// /root/reference/src/FiniteVolume.jl:160-161 and src/transient.jl:50-58.
// The operator is (A + sigma*D): A the assembled symmetric CSR, D = Ss*volumes on
// the free cells, sigma = 1/dt (0 for the steady solve), i.e. the SPD form of the
// reference's (I/dt + D^-1 A).
//
// Everything here is HBM-bandwidth bound (0.13 flop/B), so no MFMA: the kernels
// are built for coalesced streaming of vals/colind, L2-served gathers of x and
// wave64 shuffle + LDS reductions.  One PCG iteration is three launches:
//   K1 spmv_dot   q = (A + sigma D) p ; partial p.q per block      12 nnz + 28 n  B
//   K2 update     alpha = rz/pq ; x += alpha p ; r -= alpha q ;
//                 partial r.M^-1 r and r.r per block                56 n B
//   K3 pupdate    beta = rz'/rz ; p = M^-1 r + beta p ; scalars     32 n B
// Scalars never visit the host inside the loop: every block re-reduces the <= 2048
// per-block partials of the previous kernel in a fixed order (deterministic, no
// atomics), and a device-side `done` flag turns surplus launches into no-ops, so
// the host polls convergence only once per chunk of iterations.
#include "fv_internal.h"
#include "fv_device.h"

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
struct StepInitEpilogue {
    const double *bprime; // b' (may be null = 0)
    const double *D;
    const double *diagA;
    double *minv, *r, *pv;
    double *part_rz, *part_rr, *part_bb;
    double sigma, dt;
    int b_times_D, compute_minv;
    int q_shifted; // the SpMV used the shifted operator: r = rhs - q instead of b' - q
};

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
                    mi = 1.0 / (epi.diagA[row] + epi.sigma * di);
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
static int g_spmv_form = 2;
static int g_spmv_unroll = 2;
static int g_use_order = 1;
static int g_nt = 1;
static int g_fuse_init = 0; // measured: with the sliced-DIA SpMV the separate set-up kernel is ~3 % faster than the fused epilogue
static int g_use_dia = 1;
static int g_dia_packed = 1; // sliced-DIA values packed (sl_noff blocks per slice) or padded to DIA_K blocks (fv_tune key 11; read when the DIA copy is built)
static int g_march = 1;      // plane-marching sliced-DIA kernel on structured grids (fv_tune key 9)
static int g_march_segs = 0; // segments per XCD of the marching kernel (fv_tune key 10; 0 = chosen per operator)
int g_fold_shift = 1;
extern int g_carry_refresh, g_carry_speculate; // fv_transient.hip

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
    else if (key == 9 && (value == 0 || value == 1))
        g_march = value;
    else if (key == 10 && value >= 0 && value <= 16)
        g_march_segs = value;
    else if (key == 11 && (value == 0 || value == 1))
        g_dia_packed = value;
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
                    mi = 1.0 / (epi.diagA[row] + epi.sigma * di);
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
template <bool DOT, bool NT>
__global__ __launch_bounds__(FV_BLOCK, 8) void spmv_dia_march_kernel(int64_t n, int64_t ncols, int64_t nslices, int64_t step, int shift, int64_t stride,
                                                                   int seglen, int segs_per_xcd, int64_t win_lo, int64_t win_hi,
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
        {
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
            {
                const int32_t rn = (int32_t)(nsl << 6) + lane;
                nextc = (have_next && rn < nc32) ? x[rn] : 0.0;
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
                        else if (off == -st32 && have_prev) {
                            xv = __shfl(prevc, (lane - shift) & 63, 64);
                            if (lane < shift) {
                                int32_t c = row - st32;
                                c = c < 0 ? 0 : c;
                                xv = x[c];
                            }
                        } else if (off == st32 && have_next) {
                            xv = __shfl(nextc, (lane + shift) & 63, 64);
                            if (lane + shift >= 64) {
                                int32_t c = row + st32;
                                c = c >= nc32 ? nc32 - 1 : c;
                                xv = x[c];
                            }
                        } else {
                            int32_t c = row + off; // |off| <= stride < n/4: no overflow
                            c = c < 0 ? 0 : (c >= nc32 ? nc32 - 1 : c);
                            xv = x[c];
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

static int build_dia(fv_problem *p)
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

enum { SPMV_PLAIN = 0, SPMV_DOT = 1, SPMV_INIT = 2 };

static int stream_grid(int64_t npos)
{
    int64_t g = ((npos + 3) / 4 + 7) / 8 * 8; // 4 groups per block and pass, multiple of 8 (XCD shares)
    if (g < 8)
        g = 8;
    if (g > FV_MAX_PARTIALS)
        g = FV_MAX_PARTIALS;
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
struct GroupSubset {
    const int32_t *dia = nullptr; // DIA slices of the subset
    int64_t ndia = 0;
    const int32_t *csr = nullptr; // its CSR groups
    int64_t ncsr = 0;
    // when the subset's groups are exactly the slices [win_lo, win_hi): lets the plane-marching kernel do its DIA part
    int64_t win_lo = 0, win_hi = 0;
};

static int spmv_apply(fv_problem *p, const double *x, double *y, double sigma, const double *vals_override, int mode, double *partials,
                      const StepInitEpilogue *epi_in, bool use_done, int *nparts, const GroupSubset *subset = nullptr)
{
    fv_ctx *ctx = p->ctx;
    if (!p->order_built)
        FV_TRY(build_group_order(p));
    if (!p->dia_built)
        FV_TRY(build_dia(p));
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
        const bool march = g_march && mode != SPMV_INIT && p->order_stride >= 4096 && dcount > 0 && (!subset || subset->win_hi > subset->win_lo);
        int GM = 0;
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
                    if (gg > FV_MAX_PARTIALS)
                        gg = FV_MAX_PARTIALS;
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
            if (g > FV_MAX_PARTIALS)
                g = FV_MAX_PARTIALS;
            GM = (int)g;
#define FV_MARCH(D_, N_)                                                                                                                      \
    hipLaunchKernelGGL((spmv_dia_march_kernel<D_, N_>), dim3(GM), dim3(FV_BLOCK), 0, ctx->stream, p->n, p->n + p->nhalo, ns, step, sh, p->order_stride, \
                       seglen, segs_per_xcd, subset ? subset->win_lo : (int64_t)0, subset ? subset->win_hi : ns, (const int32_t *)p->dia_pos.p, (const uint8_t *)p->sl_noff.p, (const int32_t *)p->sl_off.p,        \
                       (const double *)p->dia_vals.p, x, y, shift, sigma, partials, scal)
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

static int ensure_folded(fv_problem *p, double sigma, const double **out);

int fv_spmv_launch(fv_problem *p, const double *x, double *y, double sigma, double *partials_or_null, bool fold, int *npartials)
{
    const double *folded = nullptr;
    if (fold && sigma != 0.0)
        FV_TRY(ensure_folded(p, sigma, &folded));
    return spmv_apply(p, x, y, folded ? 0.0 : sigma, folded, partials_or_null ? SPMV_DOT : SPMV_PLAIN, partials_or_null, nullptr, false, npartials);
}

// ------------------------------------------------------------------ PCG vector kernels

// r = rhs - q with q = (A + sigma D) x0 (absent when x0 = 0), for an explicit
// right-hand side; or, for an implicit time step from the state x0 itself,
//      rhs = b' + D x0/dt ,  q = A x0 (unshifted)  =>  r = b' - q
// (the D x0/dt terms of rhs and of the shifted operator cancel, so the step needs
// neither a materialised rhs nor the shift in its first SpMV).  b' is the assembled
// b, or D*bhat when the caller supplies the volume-scaled bhat of the reference.
// Also M^-1 = 1/(diag(A) + sigma D) (skipped when cached), p = M^-1 r and the
// per-block partials of r.M^-1 r, r.r, rhs.rhs.
template <bool IMPLICIT>
__global__ __launch_bounds__(FV_BLOCK) void pcg_init_kernel(int64_t n, const double *__restrict__ rhs, const double *__restrict__ q,
                                                             const double *__restrict__ diagA, const double *__restrict__ D,
                                                             double sigma, double dt, int b_times_D, const double *__restrict__ x0,
                                                             int compute_minv, int q_shifted, double *__restrict__ r, double *__restrict__ pv,
                                                             double *__restrict__ minv, double *__restrict__ part_rz,
                                                             double *__restrict__ part_rr, double *__restrict__ part_bb)
{
    __shared__ double smem[4];
    double arz = 0.0, arr = 0.0, abb = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x; i < n; i += vec_stride()) {
        double bi = rhs ? rhs[i] : 0.0; // explicit rhs, or b' of the implicit step (absent = 0)
        double ri;
        if (IMPLICIT) {
            const double di = D[i];
            if (b_times_D)
                bi *= di;
            const double rhsv = bi + di * (x0[i] / dt); // the right-hand side this step solves for
            ri = q_shifted ? rhsv - q[i] : bi - q[i];  // q = (A + sigma D) x0 or A x0
            bi = rhsv;
        } else
            ri = q ? bi - q[i] : bi;
        double mi;
        if (compute_minv) {
            const double d = D ? diagA[i] + sigma * D[i] : diagA[i];
            mi = 1.0 / d;
            minv[i] = mi;
        } else
            mi = minv[i];
        const double zi = mi * ri;
        r[i] = ri;
        pv[i] = zi;
        arz += ri * zi;
        arr += ri * ri;
        abb += bi * bi;
    }
    const double t0 = block_sum(arz, smem);
    const double t1 = block_sum(arr, smem);
    const double t2 = block_sum(abb, smem);
    if (threadIdx.x == 0) {
        part_rz[blockIdx.x] = t0;
        part_rr[blockIdx.x] = t1;
        part_bb[blockIdx.x] = t2;
    }
}

// K0' — set-up of an implicit step from the previous step's final residual (PcgSystem::carry_prev): with the same
// operator and b', rhs_new - rhs_old = sigma D (x - x_prev), so r0 = r_final + sigma D (x - x_prev) needs no SpMV.
// Everything else as pcg_init_kernel<true>: p = M^-1 r0 and the partials of r.M^-1 r, r.r, rhs.rhs with
// rhs = b' + D x/dt.  Streams: r, D, x, x_prev, b', M^-1 in; r, p out (64 B per row).
__global__ __launch_bounds__(FV_BLOCK) void pcg_carry_init_kernel(int64_t n, const double *__restrict__ bprime, const double *__restrict__ D,
                                                                   double dt, const double *__restrict__ x, const double *__restrict__ xprev,
                                                                   const double *__restrict__ minv, double *__restrict__ r,
                                                                   double *__restrict__ pv, double *__restrict__ part_rz,
                                                                   double *__restrict__ part_rr, double *__restrict__ part_bb)
{
    __shared__ double smem[4];
    double arz = 0.0, arr = 0.0, abb = 0.0;
    const int64_t n2 = n >> 1;
    const double2 *b2 = reinterpret_cast<const double2 *>(bprime);
    const double2 *D2 = reinterpret_cast<const double2 *>(D);
    const double2 *x2 = reinterpret_cast<const double2 *>(x);
    const double2 *o2 = reinterpret_cast<const double2 *>(xprev);
    const double2 *m2 = reinterpret_cast<const double2 *>(minv);
    double2 *r2 = reinterpret_cast<double2 *>(r);
    double2 *p2 = reinterpret_cast<double2 *>(pv);
    for (int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x; i < n2; i += vec_stride()) {
        const double2 bv = bprime ? b2[i] : make_double2(0.0, 0.0);
        const double2 dv = D2[i], xv = x2[i], ov = o2[i], mv = m2[i];
        double2 rv = r2[i];
        rv.x += dv.x * ((xv.x - ov.x) / dt);
        rv.y += dv.y * ((xv.y - ov.y) / dt);
        const double hx = bv.x + dv.x * (xv.x / dt), hy = bv.y + dv.y * (xv.y / dt);
        const double zx = mv.x * rv.x, zy = mv.y * rv.y;
        r2[i] = rv;
        p2[i] = make_double2(zx, zy);
        arz += rv.x * zx + rv.y * zy;
        arr += rv.x * rv.x + rv.y * rv.y;
        abb += hx * hx + hy * hy;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const int64_t i = n - 1;
        const double ri = r[i] + D[i] * ((x[i] - xprev[i]) / dt);
        const double hi = (bprime ? bprime[i] : 0.0) + D[i] * (x[i] / dt);
        const double zi = minv[i] * ri;
        r[i] = ri;
        pv[i] = zi;
        arz += ri * zi;
        arr += ri * ri;
        abb += hi * hi;
    }
    const double t0 = block_sum(arz, smem);
    const double t1 = block_sum(arr, smem);
    const double t2 = block_sum(abb, smem);
    if (threadIdx.x == 0) {
        part_rz[blockIdx.x] = t0;
        part_rr[blockIdx.x] = t1;
        part_bb[blockIdx.x] = t2;
    }
}

__global__ __launch_bounds__(FV_BLOCK) void pcg_init_finalize_kernel(const double *__restrict__ part_rz,
                                                                      const double *__restrict__ part_rr,
                                                                      const double *__restrict__ part_bb, int nparts, double rtol,
                                                                      PcgScalars *__restrict__ scal)
{
    __shared__ double smem[4];
    const double rz = reduce_partials(part_rz, nparts, smem);
    const double rr = reduce_partials(part_rr, nparts, smem);
    const double bb = reduce_partials(part_bb, nparts, smem);
    if (threadIdx.x == 0) {
        scal->rz[0] = rz;
        scal->rz[1] = 0.0;
        scal->rr = rr;
        scal->bnorm2 = bb;
        scal->tol2 = rtol * rtol * bb; // stop when ||r|| <= rtol*||b||  (IterativeSolvers' reltol)
        scal->pq = 0.0;
        scal->iters = 0;
        scal->done = (rr <= scal->tol2) ? 1 : 0;
    }
}

// K2.  SPLIT: the iterate is read from xin and written to x (first iteration of a ping-pong step), else in place.
template <bool SPLIT>
__global__ __launch_bounds__(FV_BLOCK) void pcg_update_kernel(int64_t n, int it, const double *xin, double *x, double *__restrict__ r,
                                                               const double *__restrict__ pv, const double *__restrict__ q,
                                                               const double *__restrict__ minv, const double *__restrict__ part_pq,
                                                               int npq, PcgScalars *__restrict__ scal, double *__restrict__ part_rz,
                                                               double *__restrict__ part_rr)
{
    __shared__ double smem[4];
    if (scal->done)
        return;
    const double pq = reduce_partials(part_pq, npq, smem);
    if (!(pq > 0.0)) { // breakdown: not positive definite, or NaN
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            scal->pq = pq;
            scal->done = 2;
        }
        return;
    }
    const double alpha = scal->rz[it & 1] / pq;
    double arz = 0.0, arr = 0.0;
    const int64_t n2 = n >> 1;
    double2 *x2 = reinterpret_cast<double2 *>(x);
    const double2 *xi2 = SPLIT ? reinterpret_cast<const double2 *>(xin) : x2;
    double2 *r2 = reinterpret_cast<double2 *>(r);
    const double2 *p2 = reinterpret_cast<const double2 *>(pv);
    const double2 *q2 = reinterpret_cast<const double2 *>(q);
    const double2 *m2 = reinterpret_cast<const double2 *>(minv);
    for (int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x; i < n2; i += vec_stride()) {
        double2 xv = xi2[i], rv = r2[i];
        const double2 pvv = p2[i], qv = q2[i], mv = m2[i];
        xv.x += alpha * pvv.x;
        xv.y += alpha * pvv.y;
        rv.x -= alpha * qv.x;
        rv.y -= alpha * qv.y;
        x2[i] = xv;
        r2[i] = rv;
        arz += rv.x * (mv.x * rv.x) + rv.y * (mv.y * rv.y);
        arr += rv.x * rv.x + rv.y * rv.y;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const int64_t i = n - 1;
        const double xi = (SPLIT ? xin[i] : x[i]) + alpha * pv[i];
        const double ri = r[i] - alpha * q[i];
        x[i] = xi;
        r[i] = ri;
        arz += ri * (minv[i] * ri);
        arr += ri * ri;
    }
    const double t0 = block_sum(arz, smem);
    const double t1 = block_sum(arr, smem);
    if (threadIdx.x == 0) {
        part_rz[blockIdx.x] = t0;
        part_rr[blockIdx.x] = t1;
        if (blockIdx.x == 0)
            scal->pq = pq;
    }
}

// K2S — the first K2 of a fixed-dt step that is expected to converge in this iteration (the previous step did):
// besides alpha, x_out = x_in + alpha p, r -= alpha q and the partials of the convergence test, it prepares the NEXT
// step the way pcg_carry_init_kernel would: r0' = r + D (x_out - x_in)/dt is what it leaves in r, p' = M^-1 r0' goes to
// pnext, and the partials of r0'.M^-1 r0', r0'.r0', rhs'.rhs' (rhs' = b' + D x_out/dt) to the spec_* arrays.  If the
// step does converge here, the next step starts straight at its K1 (80 B/row instead of 56 + 64).  If it does not,
// pcg_pupdate_kernel<true> takes the D (x_out - x_in)/dt term out of r again before it builds the next direction.
__global__ __launch_bounds__(FV_BLOCK) void pcg_update_spec_kernel(int64_t n, const double *__restrict__ xin, double *__restrict__ xout,
                                                                    double *__restrict__ r, const double *__restrict__ pv,
                                                                    const double *__restrict__ q, const double *__restrict__ minv,
                                                                    const double *__restrict__ D, const double *__restrict__ bprime, double dt,
                                                                    const double *__restrict__ part_pq, int npq, PcgScalars *__restrict__ scal,
                                                                    double *__restrict__ part_rz, double *__restrict__ part_rr,
                                                                    double *__restrict__ pnext, double *__restrict__ spec_rz,
                                                                    double *__restrict__ spec_rr, double *__restrict__ spec_bb)
{
    __shared__ double smem[4];
    if (scal->done)
        return;
    const double pq = reduce_partials(part_pq, npq, smem);
    if (!(pq > 0.0)) {
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            scal->pq = pq;
            scal->done = 2;
        }
        return;
    }
    const double alpha = scal->rz[0] / pq;
    double arz = 0.0, arr = 0.0, srz = 0.0, srr = 0.0, sbb = 0.0;
    const int64_t n2 = n >> 1;
    const double2 *xi2 = reinterpret_cast<const double2 *>(xin);
    double2 *xo2 = reinterpret_cast<double2 *>(xout);
    double2 *r2 = reinterpret_cast<double2 *>(r);
    const double2 *p2 = reinterpret_cast<const double2 *>(pv);
    const double2 *q2 = reinterpret_cast<const double2 *>(q);
    const double2 *m2 = reinterpret_cast<const double2 *>(minv);
    const double2 *D2 = reinterpret_cast<const double2 *>(D);
    const double2 *b2 = reinterpret_cast<const double2 *>(bprime);
    double2 *pn2 = reinterpret_cast<double2 *>(pnext);
    for (int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x; i < n2; i += vec_stride()) {
        const double2 xv = xi2[i], pvv = p2[i], qv = q2[i], mv = m2[i], dv = D2[i];
        const double2 bv = bprime ? b2[i] : make_double2(0.0, 0.0);
        double2 rv = r2[i];
        const double xnx = xv.x + alpha * pvv.x, xny = xv.y + alpha * pvv.y;
        rv.x -= alpha * qv.x;
        rv.y -= alpha * qv.y;
        arz += rv.x * (mv.x * rv.x) + rv.y * (mv.y * rv.y);
        arr += rv.x * rv.x + rv.y * rv.y;
        // the next step's set-up, on the increment as stored (x_out - x_in after rounding, like K0')
        const double cx = rv.x + dv.x * ((xnx - xv.x) / dt), cy = rv.y + dv.y * ((xny - xv.y) / dt);
        const double hx = bv.x + dv.x * (xnx / dt), hy = bv.y + dv.y * (xny / dt);
        const double zx = mv.x * cx, zy = mv.y * cy;
        xo2[i] = make_double2(xnx, xny);
        r2[i] = make_double2(cx, cy);
        pn2[i] = make_double2(zx, zy);
        srz += cx * zx + cy * zy;
        srr += cx * cx + cy * cy;
        sbb += hx * hx + hy * hy;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const int64_t i = n - 1;
        const double xn = xin[i] + alpha * pv[i];
        const double ri = r[i] - alpha * q[i];
        arz += ri * (minv[i] * ri);
        arr += ri * ri;
        const double c = ri + D[i] * ((xn - xin[i]) / dt);
        const double h = (bprime ? bprime[i] : 0.0) + D[i] * (xn / dt);
        const double z = minv[i] * c;
        xout[i] = xn;
        r[i] = c;
        pnext[i] = z;
        srz += c * z;
        srr += c * c;
        sbb += h * h;
    }
    const double t0 = block_sum(arz, smem);
    const double t1 = block_sum(arr, smem);
    const double t2 = block_sum(srz, smem);
    const double t3 = block_sum(srr, smem);
    const double t4 = block_sum(sbb, smem);
    if (threadIdx.x == 0) {
        part_rz[blockIdx.x] = t0;
        part_rr[blockIdx.x] = t1;
        spec_rz[blockIdx.x] = t2;
        spec_rr[blockIdx.x] = t3;
        spec_bb[blockIdx.x] = t4;
        if (blockIdx.x == 0)
            scal->pq = pq;
    }
}

// K3.  UNSPEC: the K2 before it was pcg_update_spec_kernel and the step did not converge there: r carries the next
// step's D (x_out - x_in)/dt term, which is taken out again here (same expression, same operands).
template <bool UNSPEC>
__global__ __launch_bounds__(FV_BLOCK) void pcg_pupdate_kernel(int64_t n, int it, double *__restrict__ r, const double *__restrict__ minv,
                                                                double *__restrict__ pv, const double *__restrict__ part_rz,
                                                                const double *__restrict__ part_rr, int nparts, PcgScalars *__restrict__ scal,
                                                                double *__restrict__ hist, int64_t hist_cap, const double *__restrict__ xin,
                                                                const double *__restrict__ xout, const double *__restrict__ D, double dt)
{
    __shared__ double smem[4];
    if (scal->done)
        return;
    const double rzn = reduce_partials(part_rz, nparts, smem);
    const double rrn = reduce_partials(part_rr, nparts, smem);
    const double beta = rzn / scal->rz[it & 1];
    const bool converged = rrn <= scal->tol2; // same value in every block: the search direction is not needed any more
    const int64_t n2 = converged ? 0 : (n >> 1);
    double2 *r2 = reinterpret_cast<double2 *>(r);
    const double2 *m2 = reinterpret_cast<const double2 *>(minv);
    double2 *p2 = reinterpret_cast<double2 *>(pv);
    const double2 *xi2 = reinterpret_cast<const double2 *>(xin);
    const double2 *xo2 = reinterpret_cast<const double2 *>(xout);
    const double2 *D2 = reinterpret_cast<const double2 *>(D);
    for (int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x; i < n2; i += vec_stride()) {
        double2 rv = r2[i];
        const double2 mv = m2[i];
        if (UNSPEC) {
            const double2 a = xi2[i], b = xo2[i], dv = D2[i];
            rv.x -= dv.x * ((b.x - a.x) / dt);
            rv.y -= dv.y * ((b.y - a.y) / dt);
            r2[i] = rv;
        }
        double2 pvv = p2[i];
        pvv.x = mv.x * rv.x + beta * pvv.x;
        pvv.y = mv.y * rv.y + beta * pvv.y;
        p2[i] = pvv;
    }
    if (!converged && (n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const int64_t i = n - 1;
        double ri = r[i];
        if (UNSPEC) {
            ri -= D[i] * ((xout[i] - xin[i]) / dt);
            r[i] = ri;
        }
        pv[i] = minv[i] * ri + beta * pv[i];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        scal->rz[(it + 1) & 1] = rzn;
        scal->rr = rrn;
        scal->iters = it + 1;
        if (hist && it < hist_cap)
            hist[it] = sqrt(rrn);
        if (converged)
            scal->done = 1;
    }
}

int fv_pcg_prepare(fv_problem *p)
{
    fv_ctx *ctx = p->ctx;
    if (p->r.p)
        return FV_OK;
    const size_t n = (size_t)p->n + (size_t)p->nhalo + FV_VEC_PAD; // slack for double2 tails and whole x lines; halo slots of a row block
    FV_TRY(p->r.alloc(ctx, n));
    FV_TRY(p->pvec.alloc(ctx, n));
    FV_TRY(p->q.alloc(ctx, n));
    FV_TRY(p->minv.alloc(ctx, n));
    FV_TRY(p->rhs.alloc(ctx, n));
    FV_TRY(p->tmp.alloc(ctx, n));
    // two launches (sliced-DIA part + CSR part) may each leave up to FV_MAX_PARTIALS partials
    FV_TRY(p->part_pq.alloc(ctx, 4 * FV_MAX_PARTIALS)); // distributed: interior + boundary pass, each DIA + CSR
    FV_TRY(p->part_rz.alloc(ctx, 2 * FV_MAX_PARTIALS));
    FV_TRY(p->part_rr.alloc(ctx, 2 * FV_MAX_PARTIALS));
    FV_TRY(p->part_bb.alloc(ctx, 2 * FV_MAX_PARTIALS));
    FV_TRY(p->scal.alloc(ctx, 1));
    FV_TRY(p->scal.zero(ctx));
    FV_TRY(p->pvec.zero(ctx));
    FV_TRY(p->q.zero(ctx));
    return FV_OK;
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

extern int g_fold_shift;

// Returns the folded value array for this sigma (building it if needed), or nullptr when folding is not possible.
static int ensure_folded(fv_problem *p, double sigma, const double **out)
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

int fv_pcg_solve(fv_problem *p, double *x, const PcgSystem &sys, double rtol, int64_t maxiter, fv_solve_info *info, bool time_it)
{
    fv_ctx *ctx = p->ctx;
    FV_TRY(fv_pcg_prepare(p));
    if (maxiter < 0)
        maxiter = 0;
    if (maxiter > 0x7ffffff0LL)
        maxiter = 0x7ffffff0LL;
    const double sigma = sys.sigma;
    if ((sigma != 0.0 || sys.implicit_step) && !p->D.p) {
        fv_set_error(ctx, "fv_pcg_solve: shifted operator requested before fv_transient_begin");
        return FV_ERR_STATE;
    }
    const int64_t n = p->n;
    const int Gv = vec_grid(n);
    const double *folded = nullptr;
    if (sys.fold_shift && sigma != 0.0)
        FV_TRY(ensure_folded(p, sigma, &folded));
    const double sig_mv = folded ? 0.0 : sigma; // the SpMV's own shift is off when the diagonal already carries it
    if (time_it)
        FV_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    const double *Dp = (sigma != 0.0) ? p->D.p : nullptr;
    // the Jacobi diagonal only depends on (sigma, assembled values): keep it across the steps of a fixed-dt run
    const int compute_minv = !(p->minv_valid && p->minv_sigma == sigma && p->minv_epoch == p->assemble_epoch);
    p->minv_valid = true;
    p->minv_sigma = sigma;
    p->minv_epoch = p->assemble_epoch;
    int Ginit = Gv; // number of per-block partials the set-up produced
    // speculative set-up left behind by the previous step's K2S (pcg_update_spec_kernel): r, p' and the partials are ready
    const bool use_spec = sys.use_spec && p->spec_valid && sys.implicit_step && !compute_minv;
    p->spec_valid = false;
    // ... and whether this step's first K2 should prepare the next step the same way
    const bool speculate = sys.speculate && sys.x_next && sys.implicit_step && !sys.b_times_D && !compute_minv && !g_fuse_init &&
                           p->precond == FV_PRECOND_JACOBI && p->last_iters == 1 && maxiter > 0;
    if (speculate && !p->pnext.p)
        FV_TRY(p->pnext.alloc(ctx, (size_t)n + (size_t)p->nhalo + FV_VEC_PAD));
    const double *in_rz = p->part_rz.p, *in_rr = p->part_rr.p, *in_bb = p->part_bb.p;
    if (use_spec) {
        std::swap(p->pvec.p, p->pnext.p);
        std::swap(p->pvec.n, p->pnext.n);
        in_rz += FV_MAX_PARTIALS;
        in_rr += FV_MAX_PARTIALS;
        in_bb += FV_MAX_PARTIALS;
    } else if (sys.implicit_step && sys.carry_prev && !compute_minv && !sys.b_times_D) {
        hipLaunchKernelGGL(pcg_carry_init_kernel, dim3(Gv), dim3(FV_BLOCK), 0, ctx->stream, n, sys.rhs, (const double *)p->D.p, sys.dt,
                           (const double *)x, sys.carry_prev, (const double *)p->minv.p, p->r.p, p->pvec.p, p->part_rz.p, p->part_rr.p,
                           p->part_bb.p);
    } else if (sys.implicit_step && g_fuse_init && g_spmv_form == 2) {
        // the whole set-up in the epilogue of ONE SpMV: with the folded matrix q = (A + sigma D) x0 and
        // r0 = rhs - q; otherwise q = A x0 (plain) and r0 = b' - q (the D x0/dt terms cancel)
        StepInitEpilogue epi{};
        epi.bprime = sys.rhs;
        epi.D = p->D.p;
        epi.diagA = p->diagA.p;
        epi.minv = p->minv.p;
        epi.r = p->r.p;
        epi.pv = p->pvec.p;
        epi.part_rz = p->part_rz.p;
        epi.part_rr = p->part_rr.p;
        epi.part_bb = p->part_bb.p;
        epi.sigma = sigma;
        epi.dt = sys.dt;
        epi.b_times_D = (int)sys.b_times_D;
        epi.compute_minv = compute_minv;
        epi.q_shifted = folded ? 1 : 0;
        FV_TRY(spmv_apply(p, x, nullptr, 0.0, folded, SPMV_INIT, nullptr, &epi, false, &Ginit));
    } else if (sys.implicit_step) {
        // q = A x0 (plain) or, when the folded matrix is in use, (A + sigma D) x0 — never alternate between the two
        // value arrays inside a run (the lane-major copy would be rebuilt every time)
        FV_TRY(spmv_apply(p, x, p->q.p, 0.0, folded, SPMV_PLAIN, nullptr, nullptr, false, nullptr));
        hipLaunchKernelGGL(pcg_init_kernel<true>, dim3(Gv), dim3(FV_BLOCK), 0, ctx->stream, n, sys.rhs, (const double *)p->q.p, p->diagA.p,
                           (const double *)p->D.p, sigma, sys.dt, (int)sys.b_times_D, (const double *)x, compute_minv, folded ? 1 : 0, p->r.p,
                           p->pvec.p, p->minv.p, p->part_rz.p, p->part_rr.p, p->part_bb.p);
    } else if (sys.x0_zero) {
        FV_HIP(ctx, hipMemsetAsync(x, 0, (size_t)n * sizeof(double), ctx->stream));
        hipLaunchKernelGGL(pcg_init_kernel<false>, dim3(Gv), dim3(FV_BLOCK), 0, ctx->stream, n, sys.rhs, (const double *)nullptr, p->diagA.p,
                           Dp, sigma, 0.0, 0, (const double *)nullptr, compute_minv, 0, p->r.p, p->pvec.p, p->minv.p, p->part_rz.p,
                           p->part_rr.p, p->part_bb.p);
    } else {
        FV_TRY(spmv_apply(p, x, p->q.p, sig_mv, folded, SPMV_PLAIN, nullptr, nullptr, false, nullptr));
        hipLaunchKernelGGL(pcg_init_kernel<false>, dim3(Gv), dim3(FV_BLOCK), 0, ctx->stream, n, sys.rhs, (const double *)p->q.p, p->diagA.p,
                           Dp, sigma, 0.0, 0, (const double *)nullptr, compute_minv, 0, p->r.p, p->pvec.p, p->minv.p, p->part_rz.p,
                           p->part_rr.p, p->part_bb.p);
    }
    FV_LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(pcg_init_finalize_kernel, dim3(1), dim3(FV_BLOCK), 0, ctx->stream, in_rz, in_rr, in_bb, Ginit, rtol, p->scal.p);
    FV_LAUNCH_CHECK(ctx);
    PcgScalars *hs = reinterpret_cast<PcgScalars *>(ctx->pinned);
    int64_t it = 0;
    constexpr int64_t MAX_CHUNK = 32;
    bool polled = false;
    if (p->precond == FV_PRECOND_AMG) {
        if (sys.x_next) {
            fv_set_error(ctx, "fv_pcg_solve: the ping-pong state is a Jacobi-path feature");
            return FV_ERR_STATE;
        }
        FV_TRY(fv_amg_pcg_loop(p, x, sigma, folded != nullptr, maxiter, hs));
        polled = true;
        maxiter = 0; // skip the Jacobi loop below
    }
    // Launches past convergence are no-ops (the done flag), but they still cost a few microseconds each and
    // show up as zero-work kernels in traces, so the first chunk is sized by the previous solve on this
    // problem (consecutive time steps need about the same number of iterations) and later chunks double.
    int64_t chunk = p->last_iters > 0 ? p->last_iters : 1;
    if (chunk > MAX_CHUNK)
        chunk = MAX_CHUNK;
    if (p->profile && p->prof_ev.empty()) {
        p->prof_ev.resize((size_t)(6 * MAX_CHUNK));
        for (hipEvent_t &e : p->prof_ev)
            FV_HIP(ctx, hipEventCreate(&e));
    }
#define FV_PROF(idx)                                                                                            \
    if (p->profile)                                                                                             \
    FV_HIP(ctx, hipEventRecord(p->prof_ev[(size_t)(6 * k + (idx))], ctx->stream))
    while (it < maxiter) {
        const int64_t m = (maxiter - it < chunk) ? (maxiter - it) : chunk;
        int32_t iters_before = 0;
        if (p->profile && polled)
            iters_before = hs->iters;
        for (int64_t k = 0; k < m; k++) {
            const int iter = (int)(it + k);
            FV_PROF(0);
            int npq = 0;
            FV_TRY(spmv_apply(p, p->pvec.p, p->q.p, sig_mv, folded, SPMV_DOT, p->part_pq.p, nullptr, true, &npq));
            FV_PROF(1);
            FV_PROF(2);
            const bool spec = iter == 0 && speculate;
            if (spec)
                hipLaunchKernelGGL(pcg_update_spec_kernel, dim3(Gv), dim3(FV_BLOCK), 0, ctx->stream, n, (const double *)x, sys.x_next, p->r.p,
                                   (const double *)p->pvec.p, (const double *)p->q.p, (const double *)p->minv.p, (const double *)p->D.p,
                                   sys.rhs, sys.dt, (const double *)p->part_pq.p, npq, p->scal.p, p->part_rz.p, p->part_rr.p, p->pnext.p,
                                   p->part_rz.p + FV_MAX_PARTIALS, p->part_rr.p + FV_MAX_PARTIALS, p->part_bb.p + FV_MAX_PARTIALS);
            else if (iter == 0 && sys.x_next)
                hipLaunchKernelGGL(pcg_update_kernel<true>, dim3(Gv), dim3(FV_BLOCK), 0, ctx->stream, n, iter, (const double *)x, sys.x_next,
                                   p->r.p, p->pvec.p, p->q.p, p->minv.p, p->part_pq.p, npq, p->scal.p, p->part_rz.p, p->part_rr.p);
            else
                hipLaunchKernelGGL(pcg_update_kernel<false>, dim3(Gv), dim3(FV_BLOCK), 0, ctx->stream, n, iter, (const double *)nullptr,
                                   sys.x_next ? sys.x_next : x, p->r.p, p->pvec.p, p->q.p, p->minv.p, p->part_pq.p, npq, p->scal.p,
                                   p->part_rz.p, p->part_rr.p);
            FV_PROF(3);
            FV_PROF(4);
            if (spec)
                hipLaunchKernelGGL(pcg_pupdate_kernel<true>, dim3(Gv), dim3(FV_BLOCK), 0, ctx->stream, n, iter, p->r.p, (const double *)p->minv.p,
                                   p->pvec.p, (const double *)p->part_rz.p, (const double *)p->part_rr.p, Gv, p->scal.p, p->hist.p, p->hist_cap,
                                   (const double *)x, (const double *)sys.x_next, (const double *)p->D.p, sys.dt);
            else
                hipLaunchKernelGGL(pcg_pupdate_kernel<false>, dim3(Gv), dim3(FV_BLOCK), 0, ctx->stream, n, iter, p->r.p, (const double *)p->minv.p,
                                   p->pvec.p, (const double *)p->part_rz.p, (const double *)p->part_rr.p, Gv, p->scal.p, p->hist.p, p->hist_cap,
                                   (const double *)nullptr, (const double *)nullptr, (const double *)nullptr, 0.0);
            FV_PROF(5);
        }
        FV_LAUNCH_CHECK(ctx);
        it += m;
        FV_HIP(ctx, hipMemcpyAsync(hs, p->scal.p, sizeof(PcgScalars), hipMemcpyDeviceToHost, ctx->stream));
        FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        polled = true;
        if (p->profile) { // only launches that did real work (not the post-convergence no-ops)
            const int64_t live = (int64_t)hs->iters - iters_before;
            for (int64_t k = 0; k < m && k < live; k++)
                for (int c = 0; c < 3; c++) {
                    float ms = 0.f;
                    FV_HIP(ctx, hipEventElapsedTime(&ms, p->prof_ev[(size_t)(6 * k + 2 * c)], p->prof_ev[(size_t)(6 * k + 2 * c + 1)]));
                    p->prof_ms[c] += ms;
                    p->prof_launches[c]++;
                }
        }
        if (hs->done)
            break;
        if (chunk < MAX_CHUNK)
            chunk *= 2;
    }
#undef FV_PROF
    if (time_it)
        FV_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    if (!polled) {
        FV_HIP(ctx, hipMemcpyAsync(hs, p->scal.p, sizeof(PcgScalars), hipMemcpyDeviceToHost, ctx->stream));
        FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    p->last_iters = hs->iters;
    p->spec_valid = speculate && hs->done == 1 && hs->iters == 1; // the K2S ran and the step converged in it
    if (info) {
        info->converged = hs->done == 1;
        info->iters = hs->iters;
        info->bnorm = sqrt(hs->bnorm2);
        info->relres = hs->bnorm2 > 0 ? sqrt(hs->rr / hs->bnorm2) : sqrt(hs->rr);
        info->solve_ms = 0.0;
        info->resnorm_len = 0;
        if (time_it) {
            float ms = 0.f;
            FV_HIP(ctx, hipEventSynchronize(ctx->ev1));
            FV_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
            info->solve_ms = ms;
        }
    }
    if (hs->done == 2) {
        fv_set_error(ctx, "PCG breakdown: p.Ap = %g is not positive (operator not SPD?)", hs->pq);
    }
    return FV_OK;
}

// ------------------------------------------------------------------ small vector utilities
__global__ __launch_bounds__(FV_BLOCK) void dot_kernel(int64_t n, const double *__restrict__ a, const double *__restrict__ b,
                                                        double *__restrict__ part)
{
    __shared__ double smem[4];
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x; i < n; i += vec_stride())
        acc += a[i] * b[i];
    const double t = block_sum(acc, smem);
    if (threadIdx.x == 0)
        part[blockIdx.x] = t;
}

__global__ __launch_bounds__(FV_BLOCK) void diff2_kernel(int64_t n, const double *__restrict__ a, const double *__restrict__ b,
                                                          double *__restrict__ part)
{
    __shared__ double smem[4];
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x; i < n; i += vec_stride()) {
        const double d = a[i] - b[i];
        acc += d * d;
    }
    const double t = block_sum(acc, smem);
    if (threadIdx.x == 0)
        part[blockIdx.x] = t;
}

__global__ __launch_bounds__(FV_BLOCK) void final_sum_kernel(const double *__restrict__ part, int nparts, double *__restrict__ out)
{
    __shared__ double smem[4];
    const double t = reduce_partials(part, nparts, smem);
    if (threadIdx.x == 0)
        *out = t;
}

static int reduce_to_host(fv_problem *p, int G, double *out_host)
{
    fv_ctx *ctx = p->ctx;
    hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(FV_BLOCK), 0, ctx->stream, p->part_bb.p, G, p->part_bb.p + (FV_MAX_PARTIALS - 1));
    FV_LAUNCH_CHECK(ctx);
    double *h = reinterpret_cast<double *>(ctx->pinned);
    FV_HIP(ctx, hipMemcpyAsync(h, p->part_bb.p + (FV_MAX_PARTIALS - 1), sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *out_host = *h;
    return FV_OK;
}

int fv_dot_device(fv_problem *p, const double *a, const double *b, double *out_host)
{
    FV_TRY(fv_pcg_prepare(p));
    int G = vec_grid(p->n);
    if (G > FV_MAX_PARTIALS - 1)
        G = FV_MAX_PARTIALS - 1;
    hipLaunchKernelGGL(dot_kernel, dim3(G), dim3(FV_BLOCK), 0, p->ctx->stream, p->n, a, b, p->part_bb.p);
    FV_LAUNCH_CHECK(p->ctx);
    return reduce_to_host(p, G, out_host);
}

int fv_norm2_diff_device(fv_problem *p, const double *a, const double *b, double *out_host)
{
    FV_TRY(fv_pcg_prepare(p));
    int G = vec_grid(p->n);
    if (G > FV_MAX_PARTIALS - 1)
        G = FV_MAX_PARTIALS - 1;
    hipLaunchKernelGGL(diff2_kernel, dim3(G), dim3(FV_BLOCK), 0, p->ctx->stream, p->n, a, b, p->part_bb.p);
    FV_LAUNCH_CHECK(p->ctx);
    double s = 0.0;
    FV_TRY(reduce_to_host(p, G, &s));
    *out_host = sqrt(s);
    return FV_OK;
}

// ------------------------------------------------------------------ distributed fixed-step run (row blocks over RCCL)
// Same three kernels per iteration as the single-GPU solve; the differences are
//   - p's halo slots are refreshed before every SpMV: pack kernel -> grouped
//     ncclSend/ncclRecv on the second stream, overlapped with the SpMV over the
//     interior row groups; the boundary groups run after the halo has landed;
//   - the per-block partials are summed to device scalars and all-reduced
//     (p.q: 1 double; r.M^-1 r and r.r: one 2-double message), and K2/K3 read the
//     reduced scalars instead of re-reducing partials.
// Every rank sees bit-identical scalars, so all ranks take the same branches and
// enqueue the same collectives.
__global__ __launch_bounds__(FV_BLOCK) void dist_pack_kernel(int64_t nsend, const int32_t *__restrict__ idx, const double *__restrict__ x,
                                                              double *__restrict__ buf)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i < nsend)
        buf[i] = x[idx[i]];
}

static int dist_exchange_begin(fv_problem *p, double *xext)
{
    fv_ctx *ctx = p->ctx;
    fv_dist *d = p->dist;
    if (d->nranks <= 1)
        return FV_OK;
    if (d->nsend > 0) {
        hipLaunchKernelGGL(dist_pack_kernel, dim3(fv_blocks(d->nsend)), dim3(FV_BLOCK), 0, ctx->stream, d->nsend, d->send_idx.p, xext,
                           d->sendbuf.p);
        FV_LAUNCH_CHECK(ctx);
    }
    FV_HIP(ctx, hipEventRecord(ctx->ev_comp, ctx->stream));
    FV_HIP(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_comp, 0));
    FV_TRY(fv_comm_halo_exchange(ctx, d, d->sendbuf.p, xext + p->n, ctx->stream2));
    FV_HIP(ctx, hipEventRecord(ctx->ev_halo, ctx->stream2));
    return FV_OK;
}

static int dist_exchange_wait(fv_problem *p)
{
    if (p->dist->nranks <= 1)
        return FV_OK;
    FV_HIP(p->ctx, hipStreamWaitEvent(p->ctx->stream, p->ctx->ev_halo, 0));
    return FV_OK;
}

// y = (A + sigma D) x on the row block; with want_dot the local x.y lands in red[0] (not yet all-reduced)
// split a list of 64-row groups into those stored as DIA slices and those left to the CSR kernel
__global__ __launch_bounds__(FV_BLOCK) void list_form_flags_kernel(int64_t m, const int32_t *__restrict__ list, const uint8_t *__restrict__ sl_noff,
                                                                    int32_t *__restrict__ fd, int32_t *__restrict__ fc)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i >= m)
        return;
    const int dia = sl_noff ? (sl_noff[list[i]] > 0) : 0;
    fd[i] = dia;
    fc[i] = !dia;
}

__global__ __launch_bounds__(FV_BLOCK) void list_gather_kernel(int64_t m, const int32_t *__restrict__ idx, const int32_t *__restrict__ list,
                                                                int32_t *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i < m)
        out[i] = list[idx[i]];
}

static int split_list(fv_problem *p, const int32_t *list, int64_t m, DevBuf<int32_t> &out_dia, int64_t *ndia, DevBuf<int32_t> &out_csr,
                      int64_t *ncsr)
{
    fv_ctx *ctx = p->ctx;
    *ndia = 0;
    *ncsr = 0;
    FV_TRY(out_dia.alloc(ctx, (size_t)m));
    FV_TRY(out_csr.alloc(ctx, (size_t)m));
    if (m <= 0)
        return FV_OK;
    DevBuf<int32_t> fd, fc, idx;
    FV_TRY(fd.alloc(ctx, (size_t)m));
    FV_TRY(fc.alloc(ctx, (size_t)m));
    FV_TRY(idx.alloc(ctx, (size_t)m));
    const uint8_t *noff = (g_use_dia && p->ndia > 0) ? p->sl_noff.p : nullptr;
    hipLaunchKernelGGL(list_form_flags_kernel, dim3(fv_blocks(m)), dim3(FV_BLOCK), 0, ctx->stream, m, list, noff, fd.p, fc.p);
    FV_LAUNCH_CHECK(ctx);
    FV_TRY(fv_compact_flags(ctx, fd.p, m, idx.p, ndia));
    if (*ndia > 0)
        hipLaunchKernelGGL(list_gather_kernel, dim3(fv_blocks(*ndia)), dim3(FV_BLOCK), 0, ctx->stream, *ndia, idx.p, list, out_dia.p);
    FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    FV_TRY(fv_compact_flags(ctx, fc.p, m, idx.p, ncsr));
    if (*ncsr > 0)
        hipLaunchKernelGGL(list_gather_kernel, dim3(fv_blocks(*ncsr)), dim3(FV_BLOCK), 0, ctx->stream, *ncsr, idx.p, list, out_csr.p);
    FV_LAUNCH_CHECK(ctx);
    FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FV_OK;
}

static int dist_build_split(fv_problem *p)
{
    fv_dist *d = p->dist;
    if (!p->dia_built)
        FV_TRY(build_dia(p));
    FV_TRY(split_list(p, d->groups_int.p, d->n_int, d->int_dia, &d->n_int_dia, d->int_csr, &d->n_int_csr));
    FV_TRY(split_list(p, d->groups_bnd.p, d->n_bnd, d->bnd_dia, &d->n_bnd_dia, d->bnd_csr, &d->n_bnd_csr));
    // x-slab partitions of a structured grid: the interior groups are one contiguous run of slices
    d->int_lo = d->int_hi = 0;
    if (d->n_int > 0) {
        int32_t first = 0, last = 0;
        FV_HIP(p->ctx, hipMemcpy(&first, d->groups_int.p, sizeof first, hipMemcpyDeviceToHost));
        FV_HIP(p->ctx, hipMemcpy(&last, d->groups_int.p + (d->n_int - 1), sizeof last, hipMemcpyDeviceToHost));
        if ((int64_t)last - first + 1 == d->n_int) {
            d->int_lo = first;
            d->int_hi = (int64_t)last + 1;
        }
    }
    d->split_built = true;
    return FV_OK;
}

// y = (A + sigma D) x on the row block; with want_dot the local x.y lands in red[0] (not yet all-reduced).
// skip_exchange: the halo slots of xext were filled by the caller (single-GPU rehearsal of the boundary pass).
static int dist_spmv(fv_problem *p, double *xext, double *y, double sigma, const double *folded, bool want_dot, bool use_done,
                     bool skip_exchange = false)
{
    fv_ctx *ctx = p->ctx;
    fv_dist *d = p->dist;
    if (!d->split_built)
        FV_TRY(dist_build_split(p));
    const int mode = want_dot ? SPMV_DOT : SPMV_PLAIN;
    GroupSubset interior, boundary;
    interior.dia = d->int_dia.p;
    interior.ndia = d->n_int_dia;
    interior.csr = d->int_csr.p;
    interior.ncsr = d->n_int_csr;
    interior.win_lo = d->int_lo;
    interior.win_hi = d->int_hi;
    boundary.dia = d->bnd_dia.p;
    boundary.ndia = d->n_bnd_dia;
    boundary.csr = d->bnd_csr.p;
    boundary.ncsr = d->n_bnd_csr;
    int na = 0, nb = 0;
    if (!skip_exchange)
        FV_TRY(dist_exchange_begin(p, xext));
    FV_TRY(spmv_apply(p, xext, y, sigma, folded, mode, want_dot ? p->part_pq.p : nullptr, nullptr, use_done, &na, &interior));
    if (!skip_exchange)
        FV_TRY(dist_exchange_wait(p));
    if (d->n_bnd > 0)
        FV_TRY(spmv_apply(p, xext, y, sigma, folded, mode, want_dot ? p->part_pq.p + na : nullptr, nullptr, use_done, &nb, &boundary));
    if (want_dot) {
        hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(FV_BLOCK), 0, ctx->stream, (const double *)p->part_pq.p, na + nb, d->red.p);
        FV_LAUNCH_CHECK(ctx);
    }
    return FV_OK;
}

// up to five partial-sum arrays reduced by one launch (block k sums array k into out[k])
struct SumSet {
    const double *a[5];
};
__global__ __launch_bounds__(FV_BLOCK) void final_sum_multi_kernel(SumSet set, int nparts, double *__restrict__ out)
{
    __shared__ double smem[4];
    const double t = reduce_partials(set.a[blockIdx.x], nparts, smem);
    if (threadIdx.x == 0)
        out[blockIdx.x] = t;
}

// x_next / carry_prev / speculate / use_spec: the ping-pong state, the residual carry-over and the speculative set-up
// of PcgSystem, same meaning.  All-reduce buffer: red[0] p.q; red[1..2] r.M^-1 r, r.r; red[3] rhs.rhs of a regular
// set-up, or red[3..5] the next step's r.M^-1 r, r.r, rhs.rhs left by pcg_update_spec_kernel.
static int dist_step(fv_problem *p, double *u, double dt, double rtol, int64_t maxiter, fv_solve_info *info, double *x_next = nullptr,
                     const double *carry_prev = nullptr, bool speculate_in = false, bool use_spec_in = false)
{
    fv_ctx *ctx = p->ctx;
    fv_dist *d = p->dist;
    const int64_t n = p->n;
    const int Gv = vec_grid(n);
    const double sigma = 1.0 / dt;
    const double *folded = nullptr;
    FV_TRY(ensure_folded(p, sigma, &folded));
    const double sig_mv = folded ? 0.0 : sigma;
    const int compute_minv = !(p->minv_valid && p->minv_sigma == sigma && p->minv_epoch == p->assemble_epoch);
    p->minv_valid = true;
    p->minv_sigma = sigma;
    p->minv_epoch = p->assemble_epoch;
    double *red = d->red.p;
    const bool use_spec = use_spec_in && p->spec_valid && !compute_minv;
    p->spec_valid = false;
    const bool speculate = speculate_in && x_next && !compute_minv && p->last_iters == 1 && maxiter > 0;
    if (speculate && !p->pnext.p)
        FV_TRY(p->pnext.alloc(ctx, (size_t)n + (size_t)p->nhalo + FV_VEC_PAD));
    if (use_spec) {
        // r, p' and the all-reduced set-up scalars (red[3..5]) were left by the previous step's K2S
        std::swap(p->pvec.p, p->pnext.p);
        std::swap(p->pvec.n, p->pnext.n);
        hipLaunchKernelGGL(pcg_init_finalize_kernel, dim3(1), dim3(FV_BLOCK), 0, ctx->stream, (const double *)(red + 3), (const double *)(red + 4),
                           (const double *)(red + 5), 1, rtol, p->scal.p);
        FV_LAUNCH_CHECK(ctx);
    } else {
        if (carry_prev && !compute_minv) {
            // r0 = r_final + sigma D (u - u_prev): purely local, no halo of u needed
            hipLaunchKernelGGL(pcg_carry_init_kernel, dim3(Gv), dim3(FV_BLOCK), 0, ctx->stream, n, (const double *)p->b.p, (const double *)p->D.p,
                               dt, (const double *)u, carry_prev, (const double *)p->minv.p, p->r.p, p->pvec.p, p->part_rz.p, p->part_rr.p,
                               p->part_bb.p);
        } else {
            // q = (A + sigma D) u with the matrix the iterations use (folded when available), r0 = rhs - q
            FV_TRY(dist_spmv(p, u, p->q.p, sig_mv, folded, false, false));
            hipLaunchKernelGGL(pcg_init_kernel<true>, dim3(Gv), dim3(FV_BLOCK), 0, ctx->stream, n, (const double *)p->b.p, (const double *)p->q.p,
                               p->diagA.p, (const double *)p->D.p, sigma, dt, 0, (const double *)u, compute_minv, 1, p->r.p, p->pvec.p,
                               p->minv.p, p->part_rz.p, p->part_rr.p, p->part_bb.p);
        }
        SumSet init{};
        init.a[0] = p->part_rz.p;
        init.a[1] = p->part_rr.p;
        init.a[2] = p->part_bb.p;
        hipLaunchKernelGGL(final_sum_multi_kernel, dim3(3), dim3(FV_BLOCK), 0, ctx->stream, init, Gv, red + 1);
        FV_LAUNCH_CHECK(ctx);
        FV_TRY(fv_comm_allreduce_sum(ctx, d, red + 1, 3, ctx->stream));
        hipLaunchKernelGGL(pcg_init_finalize_kernel, dim3(1), dim3(FV_BLOCK), 0, ctx->stream, (const double *)(red + 1), (const double *)(red + 2),
                           (const double *)(red + 3), 1, rtol, p->scal.p);
        FV_LAUNCH_CHECK(ctx);
    }
    PcgScalars *hs = reinterpret_cast<PcgScalars *>(ctx->pinned);
    int64_t it = 0;
    int64_t chunk = p->last_iters > 0 ? p->last_iters : 1;
    if (chunk > 32)
        chunk = 32;
    while (it < maxiter) {
        const int64_t m = (maxiter - it < chunk) ? (maxiter - it) : chunk;
        for (int64_t k = 0; k < m; k++) {
            const int iter = (int)(it + k);
            const bool spec = iter == 0 && speculate;
            FV_TRY(dist_spmv(p, p->pvec.p, p->q.p, sig_mv, folded, true, true));
            FV_TRY(fv_comm_allreduce_sum(ctx, d, red, 1, ctx->stream));
            SumSet sums{};
            sums.a[0] = p->part_rz.p;
            sums.a[1] = p->part_rr.p;
            if (spec) {
                hipLaunchKernelGGL(pcg_update_spec_kernel, dim3(Gv), dim3(FV_BLOCK), 0, ctx->stream, n, (const double *)u, x_next, p->r.p,
                                   (const double *)p->pvec.p, (const double *)p->q.p, (const double *)p->minv.p, (const double *)p->D.p,
                                   (const double *)p->b.p, dt, (const double *)red, 1, p->scal.p, p->part_rz.p, p->part_rr.p, p->pnext.p,
                                   p->part_rz.p + FV_MAX_PARTIALS, p->part_rr.p + FV_MAX_PARTIALS, p->part_bb.p + FV_MAX_PARTIALS);
                sums.a[2] = p->part_rz.p + FV_MAX_PARTIALS;
                sums.a[3] = p->part_rr.p + FV_MAX_PARTIALS;
                sums.a[4] = p->part_bb.p + FV_MAX_PARTIALS;
            } else if (iter == 0 && x_next)
                hipLaunchKernelGGL(pcg_update_kernel<true>, dim3(Gv), dim3(FV_BLOCK), 0, ctx->stream, n, iter, (const double *)u, x_next, p->r.p,
                                   p->pvec.p, p->q.p, p->minv.p, (const double *)red, 1, p->scal.p, p->part_rz.p, p->part_rr.p);
            else
                hipLaunchKernelGGL(pcg_update_kernel<false>, dim3(Gv), dim3(FV_BLOCK), 0, ctx->stream, n, iter, (const double *)nullptr,
                                   x_next ? x_next : u, p->r.p, p->pvec.p, p->q.p, p->minv.p, (const double *)red, 1, p->scal.p,
                                   p->part_rz.p, p->part_rr.p);
            const int nsums = spec ? 5 : 2;
            hipLaunchKernelGGL(final_sum_multi_kernel, dim3(nsums), dim3(FV_BLOCK), 0, ctx->stream, sums, Gv, red + 1);
            FV_LAUNCH_CHECK(ctx);
            FV_TRY(fv_comm_allreduce_sum(ctx, d, red + 1, nsums, ctx->stream));
            if (spec)
                hipLaunchKernelGGL(pcg_pupdate_kernel<true>, dim3(Gv), dim3(FV_BLOCK), 0, ctx->stream, n, iter, p->r.p, (const double *)p->minv.p,
                                   p->pvec.p, (const double *)(red + 1), (const double *)(red + 2), 1, p->scal.p, (double *)nullptr, (int64_t)0,
                                   (const double *)u, (const double *)x_next, (const double *)p->D.p, dt);
            else
                hipLaunchKernelGGL(pcg_pupdate_kernel<false>, dim3(Gv), dim3(FV_BLOCK), 0, ctx->stream, n, iter, p->r.p, (const double *)p->minv.p,
                                   p->pvec.p, (const double *)(red + 1), (const double *)(red + 2), 1, p->scal.p, (double *)nullptr, (int64_t)0,
                                   (const double *)nullptr, (const double *)nullptr, (const double *)nullptr, 0.0);
            FV_LAUNCH_CHECK(ctx);
        }
        it += m;
        FV_HIP(ctx, hipMemcpyAsync(hs, p->scal.p, sizeof(PcgScalars), hipMemcpyDeviceToHost, ctx->stream));
        FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (hs->done)
            break;
        if (chunk < 32)
            chunk *= 2;
    }
    if (it == 0) {
        FV_HIP(ctx, hipMemcpyAsync(hs, p->scal.p, sizeof(PcgScalars), hipMemcpyDeviceToHost, ctx->stream));
        FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    p->last_iters = hs->iters;
    p->spec_valid = speculate && hs->done == 1 && hs->iters == 1;
    if (info) {
        info->converged = hs->done == 1;
        info->iters = hs->iters;
        info->bnorm = sqrt(hs->bnorm2);
        info->relres = hs->bnorm2 > 0 ? sqrt(hs->rr / hs->bnorm2) : sqrt(hs->rr);
        info->solve_ms = 0.0;
        info->resnorm_len = 0;
    }
    return FV_OK;
}

extern "C" int fv_dist_run_fixed(fv_problem *p, double dt, int64_t nsteps, double rtol, int64_t maxiter, int32_t *iters_per_step,
                                 fv_solve_info *last_info, double *total_ms)
{
    if (!p || !p->dist || nsteps < 0)
        return FV_ERR_ARG;
    fv_ctx *ctx = p->ctx;
    FV_HIP(ctx, hipSetDevice(ctx->device));
    if (!(dt > 0)) {
        fv_set_error(ctx, "time step must be positive");
        return FV_ERR_DT;
    }
    hipEvent_t e0, e1;
    FV_HIP(ctx, hipEventCreate(&e0));
    FV_HIP(ctx, hipEventCreate(&e1));
    FV_HIP(ctx, hipEventRecord(e0, ctx->stream));
    fv_solve_info inf = {};
    int rc = FV_OK;
    // ping-pong state + residual carry-over, as in fv_transient_run_fixed (identical decisions on every rank:
    // they depend only on the step index and on the all-reduced iteration count)
    const int64_t refresh = g_carry_refresh;
    const bool pingpong = refresh > 0 && nsteps >= 2;
    if (pingpong && p->pingpong_slot < 0)
        rc = fv_slot_new(p, &p->pingpong_slot);
    double *u = p->slots[0];
    double *alt = (pingpong && rc == FV_OK) ? p->slots[(size_t)p->pingpong_slot] : nullptr;
    const double *prev = nullptr;
    for (int64_t s = 0; s < nsteps && rc == FV_OK; s++) {
        const bool carry = prev != nullptr && (s % refresh) != 0;
        rc = dist_step(p, u, dt, rtol, maxiter, &inf, alt, carry ? prev : nullptr, pingpong && g_carry_speculate && s + 1 < nsteps, carry);
        if (iters_per_step)
            iters_per_step[s] = inf.iters;
        if (pingpong && rc == FV_OK) {
            prev = u;
            if (inf.iters > 0) {
                double *t = u;
                u = alt;
                alt = t;
            }
        }
    }
    if (pingpong && rc == FV_OK) {
        p->slots[0] = u;
        p->slots[(size_t)p->pingpong_slot] = alt;
    }
    if (rc == FV_OK) {
        float ms = 0.f;
        if (hipEventRecord(e1, ctx->stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess ||
            hipEventElapsedTime(&ms, e0, e1) != hipSuccess) {
            fv_set_error(ctx, "event timing failed");
            rc = FV_ERR_HIP;
        }
        if (total_ms)
            *total_ms = ms;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (last_info)
        *last_info = inf;
    return rc;
}

// one distributed SpMV on host data (tests): y_local = (A + sigma D) x, x_local has nloc entries
extern "C" int fv_dist_spmv(fv_problem *p, const double *x_local, double sigma, double *y_local)
{
    if (!p || !p->dist || !x_local || !y_local)
        return FV_ERR_ARG;
    fv_ctx *ctx = p->ctx;
    FV_HIP(ctx, hipSetDevice(ctx->device));
    FV_TRY(fv_pcg_prepare(p));
    FV_HIP(ctx, hipMemcpyAsync(p->tmp.p, x_local, (size_t)p->n * sizeof(double), hipMemcpyDefault, ctx->stream));
    FV_TRY(dist_spmv(p, p->tmp.p, p->rhs.p, sigma, nullptr, false, false));
    return fv_copy(ctx, y_local, p->rhs.p, (size_t)p->n * sizeof(double));
}

// Rehearsal of the interior + boundary passes on one GPU: the caller supplies the halo values a peer
// would have sent (nhalo doubles, in halo-slot order); no communication happens.
extern "C" int fv_dist_spmv_halo(fv_problem *p, const double *x_local, const double *halo_values, double sigma, double *y_local)
{
    if (!p || !p->dist || !x_local || !y_local || (p->nhalo > 0 && !halo_values))
        return FV_ERR_ARG;
    fv_ctx *ctx = p->ctx;
    FV_HIP(ctx, hipSetDevice(ctx->device));
    FV_TRY(fv_pcg_prepare(p));
    FV_HIP(ctx, hipMemcpyAsync(p->tmp.p, x_local, (size_t)p->n * sizeof(double), hipMemcpyDefault, ctx->stream));
    if (p->nhalo > 0)
        FV_HIP(ctx, hipMemcpyAsync(p->tmp.p + p->n, halo_values, (size_t)p->nhalo * sizeof(double), hipMemcpyDefault, ctx->stream));
    FV_TRY(dist_spmv(p, p->tmp.p, p->rhs.p, sigma, nullptr, true, false, true));
    return fv_copy(ctx, y_local, p->rhs.p, (size_t)p->n * sizeof(double));
}

// local slice of the state (slot 0), nloc values
extern "C" int fv_dist_state_get(fv_problem *p, double *u_local)
{
    if (!p || !p->dist || !u_local)
        return FV_ERR_ARG;
    FV_HIP(p->ctx, hipSetDevice(p->ctx->device));
    return fv_copy(p->ctx, u_local, p->slots[0], (size_t)p->n * sizeof(double));
}
