// Jacobi-PCG of a SMALL system in one launch (round 4).
//
// The reference's own workloads are small: test/theis.jl:21-54 is 15 650 unknowns and ~3 300 solves, the inversions of
// examples/ run 51 x 51 x 2 cells.  At that size the kernels of the PCG loop are a few microseconds each and a solve is bound by
// its launches and its host polls: 13 launches and 3 synchronous copies per one-iteration solve, 69 us per solve on the Theis
// problem (profiles/r04_small_end.json) where the kernels themselves add up to ~45.  Here the whole solve — the step's initial
// residual b' - A x0, the Jacobi diagonal, every iteration of defaultlinearsolver's CG (/root/reference/src/transient.jl:50-58;
// IterativeSolvers' stopping rule ||r|| <= rtol ||rhs||) with the preconditioner north_star names — is ONE persistent kernel of at
// most 32 resident blocks of 1024 threads that meet at a grid barrier between the phases (three per iteration: after the product's p.q, after the
// vector update's sums, after the new direction).  The operator is the canonical CSR the assembly built, the shift sigma D is
// applied on the fly; the vectors (a few hundred KB) live in the L2.  Same arithmetic as the classic loop's kernels row by row; the
// sums are grouped by this kernel's blocks.  The barrier's spin is bounded: if the blocks should ever fail to meet (they are all
// resident by construction) a flag is raised, every block leaves and the host reports the error — the grid always drains.
#include "fv_internal.h"
#include "fv_device.h"

#include <cmath>

int g_small_n = 1 << 15; // fv_tune key 61: systems of at most this many rows are solved by the single-launch kernel (0 = never)

namespace {

constexpr int SM_MAXG = 32;
constexpr int SM_BLOCK = 1024; // few, fat blocks: the barrier's cost grows with the number of blocks that meet (8 blocks: ~3.5 us, 62: ~14)

struct SmallArgs {
    int64_t n, rows_per_block;
    const int32_t *rowptr, *colind;
    const double *vals, *diagA, *D, *rhs;
    double sigma, dt, rtol;
    int implicit, b_times_D, x0_zero, G;
    double *x, *r, *pv, *q, *minv;
    const double *x0; // the initial guess (= x, or the state a step starts from: the kernel then writes x without a copy in front of it)
    uint32_t seq;
    double *part; // 4 x SM_MAXG: p.q | r.z | r.r | rhs.rhs
    PcgScalars *scal;
    int64_t maxiter;
    double *hist;
    int64_t hist_cap;
    uint32_t *bar; // [0] arrivals (monotonic across launches), [1] failure flag
    uint32_t bar_base;
};

struct GridBarrier {
    uint32_t *bar;
    uint32_t target;
    int G;
    bool failed;
    __device__ void sync()
    {
        __syncthreads();
        if (threadIdx.x == 0) {
            target += (uint32_t)G;
            __threadfence(); // this block's writes are visible device-wide before it is counted
            atomicAdd(bar, 1u);
            long spins = 0;
            while ((int32_t)(__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
                if (__hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) || ++spins > 4000000L) {
                    atomicExch(bar + 1, 1u);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            __threadfence(); // ... and the other blocks' writes are visible to this one (the CU's own cache is invalidated)
        }
        __syncthreads();
        if (__hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            failed = true;
    }
};

__device__ inline double small_block_sum(double v, double *smem) // all threads get the sum; smem: SM_BLOCK / 64 doubles; fixed order
{
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0)
        smem[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < SM_BLOCK / 64; w++)
        t += smem[w];
    return t;
}

// every block reduces the same G per-block values in the same order
__device__ inline double small_total(const double *part, int G)
{
    double s = 0.0;
    for (int g = 0; g < G; g++)
        s += __hip_atomic_load(part + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return s;
}

// one solve; false: the blocks did not meet at a barrier (nothing of this solve's results may be used)
__device__ __forceinline__ bool small_solve(const SmallArgs &a, GridBarrier &gb, double *smem)
{
    const int64_t lo = (int64_t)blockIdx.x * a.rows_per_block, hi = lo + a.rows_per_block < a.n ? lo + a.rows_per_block : a.n;
    double *part_pq = a.part, *part_rz = a.part + SM_MAXG, *part_rr = a.part + 2 * SM_MAXG, *part_bb = a.part + 3 * SM_MAXG;
    const bool shift = a.sigma != 0.0 && a.D != nullptr;
    auto product = [&](const double *v, int64_t i) -> double { // ((A + sigma D) v)_i, entries in column order like every other form
        double s = 0.0;
        const int32_t e = a.rowptr[i + 1];
        for (int32_t k = a.rowptr[i]; k < e; k++)
            s += a.vals[k] * v[a.colind[k]];
        if (shift)
            s += a.sigma * a.D[i] * v[i];
        return s;
    };
    // ---- set-up (pcg_init_kernel): r0, M^-1, p = M^-1 r0, the three sums
    double arz = 0.0, arr = 0.0, abb = 0.0;
    for (int64_t i = lo + threadIdx.x; i < hi; i += SM_BLOCK) {
        double bi = a.rhs ? a.rhs[i] : 0.0, ri;
        if (a.implicit) { // rhs = b' + D x0/dt, r0 = b' - A x0: the D x0/dt terms of rhs and of the shifted operator cancel
            const double di = a.D[i];
            if (a.b_times_D)
                bi *= di;
            double q0 = 0.0;
            const int32_t e = a.rowptr[i + 1];
            for (int32_t k = a.rowptr[i]; k < e; k++)
                q0 += a.vals[k] * a.x0[a.colind[k]];
            const double rhsv = bi + di * (a.x0[i] / a.dt);
            ri = bi - q0;
            bi = rhsv;
        } else
            ri = a.x0_zero ? bi : bi - product(a.x0, i);
        const double d = shift ? a.diagA[i] + a.sigma * a.D[i] : a.diagA[i];
        const double mi = d > 0.0 ? 1.0 / d : 0.0; // (a free cell without any face has an empty row: left where it is)
        a.minv[i] = mi;
        const double zi = mi * ri;
        a.r[i] = ri;
        a.pv[i] = zi;
        arz += ri * zi;
        arr += ri * ri;
        abb += bi * bi;
    }
    {
        const double t0 = small_block_sum(arz, smem), t1 = small_block_sum(arr, smem), t2 = small_block_sum(abb, smem);
        if (threadIdx.x == 0) {
            part_rz[blockIdx.x] = t0;
            part_rr[blockIdx.x] = t1;
            part_bb[blockIdx.x] = t2;
        }
    }
    gb.sync();
    if (gb.failed)
        return false;
    double rz = small_total(part_rz, a.G), rr = small_total(part_rr, a.G);
    const double bb = small_total(part_bb, a.G);
    const double tol2 = a.rtol * a.rtol * bb;
    int done = rr <= tol2 ? 1 : 0;
    int64_t it = 0;
    double pq = 0.0;
    while (!done && it < a.maxiter) {
        // ---- K1: q = (A + sigma D) p, p.q
        double apq = 0.0;
        for (int64_t i = lo + threadIdx.x; i < hi; i += SM_BLOCK) {
            const double qi = product(a.pv, i);
            a.q[i] = qi;
            apq += a.pv[i] * qi;
        }
        {
            const double t = small_block_sum(apq, smem);
            if (threadIdx.x == 0)
                part_pq[blockIdx.x] = t;
        }
        gb.sync();
        if (gb.failed)
            return false;
        pq = small_total(part_pq, a.G);
        if (!(pq > 0.0)) { // breakdown: not positive definite, or NaN
            done = 2;
            break;
        }
        const double alpha = rz / pq;
        // ---- K2: x += alpha p; r -= alpha q; sums r.M^-1 r, r.r
        arz = arr = 0.0;
        for (int64_t i = lo + threadIdx.x; i < hi; i += SM_BLOCK) {
            a.x[i] = (it == 0 ? (a.x0_zero ? 0.0 : a.x0[i]) : a.x[i]) + alpha * a.pv[i]; // (the first update reads the initial guess where it is)
            const double ri = a.r[i] - alpha * a.q[i];
            a.r[i] = ri;
            arz += ri * (a.minv[i] * ri);
            arr += ri * ri;
        }
        {
            const double t0 = small_block_sum(arz, smem), t1 = small_block_sum(arr, smem);
            if (threadIdx.x == 0) {
                part_rz[blockIdx.x] = t0;
                part_rr[blockIdx.x] = t1;
            }
        }
        gb.sync();
        if (gb.failed)
            return false;
        const double rzn = small_total(part_rz, a.G);
        rr = small_total(part_rr, a.G);
        if (blockIdx.x == 0 && threadIdx.x == 0 && a.hist && it < a.hist_cap)
            a.hist[it] = sqrt(rr);
        it++;
        if (rr <= tol2) {
            rz = rzn;
            done = 1;
            break;
        }
        // ---- K3: p = M^-1 r + beta p
        const double beta = rzn / rz;
        rz = rzn;
        for (int64_t i = lo + threadIdx.x; i < hi; i += SM_BLOCK)
            a.pv[i] = a.minv[i] * a.r[i] + beta * a.pv[i];
        gb.sync(); // the next product reads the other blocks' p
        if (gb.failed)
            return false;
    }
    if (it == 0 && (a.x0_zero || a.x0 != a.x)) // converged where it started: x is the initial guess
        for (int64_t i = lo + threadIdx.x; i < hi; i += SM_BLOCK)
            a.x[i] = a.x0_zero ? 0.0 : a.x0[i];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        PcgScalars *s = a.scal;
        s->small_seq = a.seq;
        s->rz[0] = s->rz[1] = rz;
        s->rr = rr;
        s->tol2 = tol2;
        s->bnorm2 = bb;
        s->pq = pq;
        s->iters = (int32_t)it;
        s->done = done;
        s->chain_step = 0;
        s->zero_mask = 0;
        s->xlag = -1;
        s->pad_ = (int32_t)gb.target; // the barrier's count at the end of this launch: the next launch's base
    }
    return true;
}

__global__ __launch_bounds__(SM_BLOCK) void pcg_small_kernel(SmallArgs a)
{
    __shared__ double smem[SM_BLOCK / 64];
    GridBarrier gb{a.bar, a.bar_base, a.G, false};
    small_solve(a, gb, smem);
}

// The step-doubling attempt of the adaptive stepper (transient.jl:72-91) as ONE launch: the full step, the two half steps — each the
// solve above, the second half step starting from the first one's result — and the squared error norm || w .* (full - two halves) ||^2,
// left in the last solve's scalar block (tol2x[0]).  first = 1: the full step is there already (the half step of a rejected attempt).
struct Small3Args {
    SmallArgs s[3];
    int first;
    const double *w; // weight of the error norm (null: none)
};

__global__ __launch_bounds__(SM_BLOCK) void pcg_small3_kernel(Small3Args a3)
{
    __shared__ double smem[SM_BLOCK / 64];
    GridBarrier gb{a3.s[0].bar, a3.s[0].bar_base, a3.s[0].G, false};
    for (int k = a3.first; k < 3; k++) {
        if (!small_solve(a3.s[k], gb, smem))
            return;
        gb.sync(); // every block's rows of this solve's x are final (also where a solve converged at its start and copied its guess)
        if (gb.failed)
            return;
    }
    const SmallArgs &a = a3.s[2];
    const int64_t lo = (int64_t)blockIdx.x * a.rows_per_block, hi = lo + a.rows_per_block < a.n ? lo + a.rows_per_block : a.n;
    const double *one = a3.s[0].x, *two = a.x;
    double acc = 0.0;
    for (int64_t i = lo + threadIdx.x; i < hi; i += SM_BLOCK) {
        const double d = a3.w ? a3.w[i] * one[i] - a3.w[i] * two[i] : one[i] - two[i]; // (the expression of wdiff_kernel / diff2_kernel: the same accept / reject decisions)
        acc += d * d;
    }
    const double t = small_block_sum(acc, smem);
    if (threadIdx.x == 0)
        a.part[blockIdx.x] = t;
    gb.sync();
    if (gb.failed)
        return;
    const double err2 = small_total(a.part, a.G);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        a.scal->tol2x[0] = err2;
        a.scal->pad_ = (int32_t)gb.target;
        __threadfence();
        a.scal->small_seq = a.seq + 1u; // (the attempt as a whole got through: the host looks for this number)
    }
}

} // namespace

// Blocks of the single-launch kernels that can be resident at once: what the occupancy query says the kernel's own registers and LDS allow per
// CU (the CU count alone says nothing about that), times the CUs; the grid barrier needs every block resident.  A GPU shared with
// other processes can still fall short: the barrier's spin is bounded (~0.2 s) and the launch then reports FV_ERR_STATE.
static int small_resident_blocks(fv_ctx *ctx, const void *kernel)
{
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, SM_BLOCK, 0) != hipSuccess || per_cu < 1)
        per_cu = 1;
    return per_cu * ctx->num_cus;
}

bool fv_pcg_small_takes(const fv_problem *p, const PcgSystem &sys)
{
    const int64_t n = p->n;
    if (g_small_n <= 0 || n <= 0 || n > g_small_n || p->dist || p->nhalo > 0 || sys.x_next || sys.carry_prev || sys.speculate || sys.use_spec || sys.chain_index >= 0 ||
        sys.resume_it > 0 || p->profile || !p->rowptr.p || !p->colind.p || !p->vals.p || !p->diagA.p)
        return false;
    return (sys.implicit_step ? fv_step_precond(p) : p->precond) != FV_PRECOND_AMG;
}

// *handled = false: not a case for this kernel (the classic loop runs).  x: initial guess in (or sys.x0_src), solution out.
int fv_pcg_small(fv_problem *p, double *x, const PcgSystem &sys, double rtol, int64_t maxiter, fv_solve_info *info, bool time_it, bool *handled)
{
    fv_ctx *ctx = p->ctx;
    *handled = false;
    const int64_t n = p->n;
    if (!fv_pcg_small_takes(p, sys)) {
        if (sys.x0_src) { // (step_impl asks the same predicate before it leaves the copy out)
            fv_set_error(ctx, "internal: an implicit step left its copy to the single-launch solver, which does not take the system");
            return FV_ERR_STATE;
        }
        return FV_OK;
    }
    if (!p->small_part.p) {
        FV_TRY(p->small_part.alloc(ctx, (size_t)4 * SM_MAXG));
        FV_TRY(p->small_bar.alloc(ctx, 2));
        FV_TRY(p->small_bar.zero(ctx));
        p->small_bar_base = 0;
    }
    SmallArgs a{};
    a.n = n;
    // one row per thread where 32 blocks of 1024 threads allow it: a thread's rows are walked one after the other, each a chain of three dependent
    // loads (row pointer, column, vector entry) — with eight rows per thread a pass over 15 650 rows took 16 us, the whole solve 46
    int G = (int)((n + SM_BLOCK - 1) / SM_BLOCK);
    G = G < 1 ? 1 : (G > SM_MAXG ? SM_MAXG : G);
    if (G > small_resident_blocks(ctx, reinterpret_cast<const void *>(&pcg_small_kernel))) // (every block must be resident for the barrier)
        G = small_resident_blocks(ctx, reinterpret_cast<const void *>(&pcg_small_kernel));
    a.G = G;
    a.rows_per_block = (n + G - 1) / G;
    a.rowptr = p->rowptr.p;
    a.colind = p->colind.p;
    a.vals = p->vals.p;
    a.diagA = p->diagA.p;
    a.D = sys.sigma != 0.0 || sys.implicit_step ? p->D.p : nullptr;
    a.rhs = sys.rhs;
    a.sigma = sys.sigma;
    a.dt = sys.dt;
    a.rtol = rtol;
    a.implicit = sys.implicit_step ? 1 : 0;
    a.b_times_D = sys.b_times_D ? 1 : 0;
    a.x0_zero = sys.x0_zero ? 1 : 0;
    a.x = x;
    a.x0 = sys.x0_src ? sys.x0_src : x;
    a.seq = ++p->small_seq;
    a.r = p->r.p;
    a.pv = p->pvec.p;
    a.q = p->q.p;
    a.minv = p->minv.p;
    a.part = p->small_part.p;
    a.scal = p->scal.p;
    a.maxiter = maxiter;
    a.hist = p->hist.p;
    a.hist_cap = p->hist_cap;
    a.bar = p->small_bar.p;
    a.bar_base = p->small_bar_base;
    if (time_it)
        FV_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    hipLaunchKernelGGL(pcg_small_kernel, dim3(G), dim3(SM_BLOCK), 0, ctx->stream, a);
    FV_LAUNCH_CHECK(ctx);
    if (time_it)
        FV_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    // one copy back: the scalar block carries the launch's number (written last, by a launch that got through its barriers) and the
    // barrier's count for the next launch
    PcgScalars *hs = reinterpret_cast<PcgScalars *>(ctx->pinned);
    FV_HIP(ctx, hipMemcpyAsync(hs, p->scal.p, sizeof(PcgScalars), hipMemcpyDeviceToHost, ctx->stream));
    FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (hs->small_seq != a.seq) {
        FV_HIP(ctx, hipMemsetAsync(p->small_bar.p, 0, 2 * sizeof(uint32_t), ctx->stream));
        p->small_bar_base = 0;
        fv_set_error(ctx, "fv_pcg_small: the blocks of the single-launch solver did not meet at their grid barrier");
        return FV_ERR_STATE;
    }
    p->small_bar_base = (uint32_t)hs->pad_;
    // what the classic loop leaves behind it
    p->minv_valid = true;
    p->minv_sigma = sys.sigma;
    p->minv_epoch = p->assemble_epoch;
    p->z_where = 0;
    p->spec_valid = false;
    p->vready = false;
    p->last_iters = hs->iters;
    p->loop_bytes = 0;
    p->fused_chunked = false;
    p->small_solves++;
    if (info) {
        info->converged = hs->done == 1;
        info->iters = hs->iters;
        info->bnorm = std::sqrt(hs->bnorm2);
        info->relres = hs->bnorm2 > 0 ? std::sqrt(hs->rr / hs->bnorm2) : std::sqrt(hs->rr);
        info->solve_ms = 0.0;
        info->resnorm_len = 0;
        if (time_it) {
            float ms = 0.f;
            FV_HIP(ctx, hipEventSynchronize(ctx->ev1));
            FV_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
            info->solve_ms = ms;
        }
    }
    if (hs->done == 2)
        fv_set_error(ctx, "PCG breakdown: p.Ap = %g is not positive (operator not SPD?)", hs->pq); // (reported like the classic loop: info->converged = 0)
    *handled = true;
    return FV_OK;
}

bool fv_small_twostep_takes(const fv_problem *p, int mode, double dt)
{
    // =0 (differential tests): the solves one by one.  Read at every attempt on purpose — the test flips it inside one process —; a getenv is ~0.1 us
    // beside an attempt of ~100 (ADVICE r4 asked for a cache: it would freeze the switch)
    const char *e = getenv("FV_SMALL_TWOSTEP");
    if ((e && atoi(e) == 0) || mode == FV_STEP_ADJOINT || !(dt > 0) || !p->D.p)
        return false;
    PcgSystem sys;
    sys.implicit_step = true;
    sys.sigma = 1.0 / dt;
    sys.dt = dt;
    return fv_pcg_small_takes(p, sys);
}

// The three solves and the error norm of one step-doubling attempt (adaptive_twostep, fv_transient.hip) in one launch, for the
// systems fv_pcg_small takes.  rhs[k]: the forcing of solve k as step_impl would pass it (b' of an implicit step; null = none);
// have_onestep: the full step is in `onestep` already.  *handled = false: not a case for it (the caller runs the three solves one by one).
int fv_small_twostep(fv_problem *p, int mode, const double *const rhs[3], bool b_times_D, double *uk, double dt, double *onestep, bool have_onestep,
                     double *two1, double *two, const double *weight, double rtol, int64_t maxiter, fv_solve_info *info, double *err, bool *handled)
{
    fv_ctx *ctx = p->ctx;
    *handled = false;
    if (!fv_small_twostep_takes(p, mode, dt))
        return FV_OK;
    FV_TRY(fv_pcg_prepare(p));
    if (!p->small_part.p) {
        FV_TRY(p->small_part.alloc(ctx, (size_t)4 * SM_MAXG));
        FV_TRY(p->small_bar.alloc(ctx, 2));
        FV_TRY(p->small_bar.zero(ctx));
        p->small_bar_base = 0;
    }
    if (!p->small_scal3.p)
        FV_TRY(p->small_scal3.alloc(ctx, 3));
    const int64_t n = p->n;
    int G = (int)((n + SM_BLOCK - 1) / SM_BLOCK);
    G = G < 1 ? 1 : (G > SM_MAXG ? SM_MAXG : G);
    if (G > small_resident_blocks(ctx, reinterpret_cast<const void *>(&pcg_small3_kernel)))
        G = small_resident_blocks(ctx, reinterpret_cast<const void *>(&pcg_small3_kernel));
    Small3Args a3{};
    const uint32_t seq = p->small_seq + 1;
    p->small_seq += 2;
    double *const xs[3] = {onestep, two1, two};
    const double *const x0s[3] = {uk, uk, two1};
    const double dts[3] = {dt, 0.5 * dt, 0.5 * dt};
    for (int k = 0; k < 3; k++) {
        SmallArgs &a = a3.s[k];
        a.n = n;
        a.G = G;
        a.rows_per_block = (n + G - 1) / G;
        a.rowptr = p->rowptr.p;
        a.colind = p->colind.p;
        a.vals = p->vals.p;
        a.diagA = p->diagA.p;
        a.D = p->D.p;
        a.rhs = rhs[k];
        a.sigma = 1.0 / dts[k];
        a.dt = dts[k];
        a.rtol = rtol;
        a.implicit = 1;
        a.b_times_D = b_times_D ? 1 : 0;
        a.x0_zero = 0;
        a.x = xs[k];
        a.x0 = x0s[k];
        a.seq = seq;
        a.r = p->r.p;
        a.pv = p->pvec.p;
        a.q = p->q.p;
        a.minv = p->minv.p;
        a.part = p->small_part.p;
        a.scal = p->small_scal3.p + k;
        a.maxiter = maxiter;
        a.hist = nullptr;
        a.hist_cap = 0;
        a.bar = p->small_bar.p;
        a.bar_base = p->small_bar_base;
    }
    a3.first = have_onestep ? 1 : 0;
    a3.w = weight;
    hipLaunchKernelGGL(pcg_small3_kernel, dim3(G), dim3(SM_BLOCK), 0, ctx->stream, a3);
    FV_LAUNCH_CHECK(ctx);
    PcgScalars *hs = reinterpret_cast<PcgScalars *>(ctx->pinned);
    static_assert(3 * sizeof(PcgScalars) <= 1024, "three scalar blocks fit the pinned page's first kilobyte");
    FV_HIP(ctx, hipMemcpyAsync(hs, p->small_scal3.p, 3 * sizeof(PcgScalars), hipMemcpyDeviceToHost, ctx->stream));
    FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (hs[2].small_seq != seq + 1u) {
        FV_HIP(ctx, hipMemsetAsync(p->small_bar.p, 0, 2 * sizeof(uint32_t), ctx->stream));
        p->small_bar_base = 0;
        fv_set_error(ctx, "fv_small_twostep: the blocks of the single-launch stepper did not meet at their grid barrier");
        return FV_ERR_STATE;
    }
    p->small_bar_base = (uint32_t)hs[2].pad_;
    // what three classic solves leave behind them
    p->minv_valid = true;
    p->minv_sigma = 1.0 / dts[2];
    p->minv_epoch = p->assemble_epoch;
    p->z_where = 0;
    p->spec_valid = false;
    p->vready = false;
    p->last_iters = hs[2].iters;
    p->loop_bytes = 0;
    p->fused_chunked = false;
    p->small_solves += 3 - a3.first;
    p->resume.ok = false;
    for (int k = a3.first; k < 3; k++) {
        if (hs[k].done == 2)
            fv_set_error(ctx, "PCG breakdown: p.Ap = %g is not positive (operator not SPD?)", hs[k].pq);
        if (p->precond == FV_PRECOND_AUTO && !p->auto_steps_amg && hs[k].iters > FV_AUTO_SWITCH_ITERS)
            p->auto_steps_amg = true; // (as step_impl after each solve: this operator wants the V-cycle from the next step on)
    }
    if (info) { // (of the last solve, as the one-by-one path leaves it)
        info->converged = hs[2].done == 1;
        info->iters = hs[2].iters;
        info->bnorm = std::sqrt(hs[2].bnorm2);
        info->relres = hs[2].bnorm2 > 0 ? std::sqrt(hs[2].rr / hs[2].bnorm2) : std::sqrt(hs[2].rr);
        info->solve_ms = 0.0;
        info->resnorm_len = 0;
    }
    *err = std::sqrt(hs[2].tol2x[0]);
    *handled = true;
    return FV_OK;
}

FV_WARM_TU(small) // (fv_ctx_create loads every code object of the library up front: fv_warm_modules, fv_ctx.hip)
