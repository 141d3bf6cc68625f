// Aggregation-based algebraic multigrid preconditioner for the PCG (SPD M-matrix-like operators).
//
// Stands where the reference calls AlgebraicMultigrid.ruge_stuben + aspreconditioner
// (/root/reference/src/FiniteVolume.jl:159-161): one V(1,1) cycle per CG iteration.  The
// construction is not Ruge-Stuben but plain (unsmoothed) aggregation, chosen because every phase
// of it is a data-parallel pass over CSR rows with no sequential colouring:
//
//   setup, per level: three passes of pairwise matching ("handshake": every unmatched row names its
//     strongest unmatched neighbour, mutual choices become pairs; leftovers join the pair of their
//     strongest matched neighbour), each followed by the Galerkin product with the piecewise-constant
//     prolongation, which for such a P is a merge of the member rows with columns renamed — one
//     thread per coarse row, sorted by column, duplicates summed in a fixed order.  No floating-point
//     atomics anywhere: the hierarchy and every cycle are bit-reproducible.
//   cycle: x = w D^-1 b ; t = A x ; b_c = P^T (b - t) ; recurse ; x += P x_c ; t = A x ;
//     x += w D^-1 (b - t).  Same w before and after, so the preconditioner is symmetric.  The coarsest
//     level (<= 1024 rows) is solved with an explicit inverse computed once by Gauss-Jordan on the device.
//   The storage term of the implicit step is carried along exactly: P^T D P is diagonal (sums of D over
//     the aggregates), so level l applies A_l + sigma D_l and only the Jacobi diagonals and the coarsest
//     inverse depend on sigma.
//
// Everything is HBM-bound streaming like the rest of the library; level 0 uses the problem's own SpMV
// (sliced-DIA / CSR-stream), the coarse levels a lanes-per-row CSR kernel.
#include "fv_internal.h"
#include "fv_device.h"

#include <rocprim/device/device_radix_sort.hpp>

#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace {

struct AmgLevel {
    int64_t n = 0, nnz = 0, nc = 0;
    // operator (level 0: views of the problem's arrays)
    const int32_t *rowptr = nullptr, *colind = nullptr;
    const double *vals = nullptr, *D = nullptr;
    DevBuf<int32_t> o_rowptr, o_colind;
    DevBuf<double> o_vals, o_D;
    DevBuf<double> diag, dinv;
    // transfer to the next level
    DevBuf<int32_t> agg;         // n: coarse index, -1 = not represented below (row without couplings)
    DevBuf<int32_t> memptr, mem; // nc+1 / members of every aggregate, ascending
    // cycle workspace (levels >= 1: u = the iterate before the last smoothing pass, whose fused product writes the level's result)
    DevBuf<double> x, b, t, u;
    // K-cycle workspace (levels solved by two flexible-CG steps, amg_kcycle): v1 = A c1, r1, c2, v2 = A c2, partial sums
    DevBuf<double> kv1, kr1, kc2, kv2, kpart;
};

} // namespace

struct fv_amg {
    std::vector<AmgLevel *> lev;
    DevBuf<double> inv, inv2; // coarsest inverse (nco x nco), row-major, and the second copy its Gauss-Jordan steps alternate with
    int64_t nco = 0;
    bool dense = false;
    double sigma = NAN;       // the sigma dinv / inv were built for
    int64_t epoch = -1;       // assemble_epoch of the hierarchy
    int64_t storage_epoch = -1; // which fv_transient_begin the aggregated D belongs to
    bool kcycle = false;      // this application of the cycle may use the K-cycle (the PCG loop around it is flexible); fv_amg_apply keeps the linear V-cycle
    bool fold = false;        // level-0 SpMVs use the folded value array (fixed-dt runs), as the PCG around them
    bool gathered = false;    // row blocks, FV_PRECOND_AMG_GATHERED: levels >= 1 are those of the WHOLE operator, replicated on every rank
    DevBuf<double> z;         // preconditioned residual of the PCG
    DevBuf<int32_t> loc_rowptr, loc_colind; // row blocks: the rank's diagonal block as the level-0 structure
    DevBuf<double> loc_vals;
    ~fv_amg()
    {
        for (AmgLevel *l : lev)
            delete l;
    }
};

void fv_amg_free(fv_amg *a) { delete a; }

// tunables (fv_amg_configure)
// Defaults since round 3 (tools/amg_sweep.py on the 256^3 sigma = 3 box, profiles/r03_amg_sweep*.log): theta 0.25 -> 0.10, omega 2/3 ->
// 0.85, rounds 6 -> 10: 148 -> 97 PCG iterations, solve 0.25 -> 0.13 s at the same set-up time and operator complexity.
static double g_theta = 0.10; // a coupling is strong when -a_ij >= theta * max_k(-a_ik)
static double g_omega = 0.85; // Jacobi damping of the smoother
static int g_passes = 2;      // pairwise passes per level (aggregates of ~4: the K-cycle keeps the extra levels near their two-grid rate)
static int g_rounds = 10;     // handshake rounds per pass
static int g_coarse_max = 1024;
static int g_coarse_sweeps = 12; // Jacobi sweeps on a coarsest level too large for the dense inverse
int g_amg_stream = 65536;       // (frozen; round 3 measured it) coarse levels with at least this many rows use the wave-stream CSR kernel (0: never)
// FV_AMG_KCYCLE in the environment (round 5: was fv_tune key 52; read at every use, so a sweep may change it between solves): coarse levels
// 1 .. k are solved by two flexible-CG steps preconditioned by the cycle below them (K-cycle; 0 = V-cycle) [2]
static int amg_kcycle_levels()
{
    const char *e = getenv("FV_AMG_KCYCLE");
    return e ? atoi(e) : 2;
}

extern "C" int fv_amg_configure(double theta, double omega, int passes, int rounds)
{
    if (!(theta > 0 && theta <= 1) || !(omega > 0 && omega < 1) || passes < 1 || passes > 4 || rounds < 1 || rounds > 16)
        return FV_ERR_ARG;
    g_theta = theta;
    g_omega = omega;
    g_passes = passes;
    g_rounds = rounds;
    return FV_OK;
}

// ------------------------------------------------------------------ setup kernels
// (AMG_ALONE: an unmatched row that found no strong unmatched neighbour in some round — it cannot find one later, so the later rounds
// neither scan it again nor offer it to its neighbours; the leftover rule treats it like any unmatched row)
constexpr int32_t AMG_UNMATCHED = -1, AMG_ISOLATED = -2, AMG_ALONE = -3;

// Preference of row i for neighbour j: the octave of the coupling strength first, then a hash that is symmetric in
// (i, j).  Ranking by the raw strength makes the handshake crawl: wherever strengths vary monotonically every row names
// its uphill neighbour and only the top of each chain pairs up per round; with octaves + a symmetric hash two
// neighbours often rank each other first, and 6 rounds match almost everything (which also keeps the aggregates of
// high-conductivity regions from snowballing through the leftover rule).
__device__ inline uint64_t amg_pref(double s, int32_t i, int32_t j)
{
    const uint32_t lo = (uint32_t)(i < j ? i : j), hi = (uint32_t)(i < j ? j : i);
    uint32_t h = lo * 0x9E3779B1u ^ hi * 0x85EBCA77u;
    h ^= h >> 15;
    h *= 0x2C1B3C6Du;
    h ^= h >> 12;
    h *= 0x297A2D39u;
    h ^= h >> 15;
    const uint64_t octave = ((uint64_t)__double_as_longlong(s) >> 52) & 0x7ffu;
    return (octave << 32) | h;
}

// every unmatched row names its preferred strong unmatched neighbour (strong: -a_ij >= theta * the row's strongest
// coupling), -1 if there is none; rows without any negative off-diagonal are marked isolated
// cut[i] = theta * the row's strongest coupling (0: no negative off-diagonal at all) — the same in every round of a pass, so the
// rounds read the row once instead of twice
__global__ __launch_bounds__(FV_BLOCK) void amg_cut_kernel(int64_t n, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ colind,
                                                            const double *__restrict__ vals, double theta, double *__restrict__ cut)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i >= n)
        return;
    double maxoff = 0.0;
    for (int32_t k = rowptr[i]; k < rowptr[i + 1]; k++) {
        const int32_t j = colind[k];
        if (j != i && j < n && -vals[k] > maxoff)
            maxoff = -vals[k];
    }
    cut[i] = theta * maxoff;
}

__global__ __launch_bounds__(FV_BLOCK) void amg_pick_kernel(int64_t n, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ colind,
                                                             const double *__restrict__ vals, const int32_t *__restrict__ partner,
                                                             int32_t *__restrict__ cand, const double *__restrict__ cutv)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i >= n)
        return;
    if (partner[i] != AMG_UNMATCHED) {
        cand[i] = -1;
        return;
    }
    const double cut = cutv[i];
    if (!(cut > 0.0)) {
        cand[i] = AMG_ISOLATED;
        return;
    }
    uint64_t bestp = 0;
    int32_t best = -1;
    for (int32_t k = rowptr[i]; k < rowptr[i + 1]; k++) {
        const int32_t j = colind[k];
        if (j == i || j >= n)
            continue;
        const double s = -vals[k];
        if (s >= cut && s > 0.0 && partner[j] == AMG_UNMATCHED) {
            const uint64_t pr = amg_pref(s, (int32_t)i, j);
            if (best < 0 || pr > bestp || (pr == bestp && j < best)) {
                bestp = pr;
                best = j;
            }
        }
    }
    cand[i] = best;
}

__global__ __launch_bounds__(FV_BLOCK) void amg_match_kernel(int64_t n, const int32_t *__restrict__ cand, int32_t *__restrict__ partner)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i >= n || partner[i] != AMG_UNMATCHED)
        return;
    const int32_t c = cand[i];
    if (c == AMG_ISOLATED)
        partner[i] = AMG_ISOLATED;
    else if (c < 0)
        partner[i] = AMG_ALONE; // no strong unmatched neighbour now, so none later
    else if (cand[c] == (int32_t)i)
        partner[i] = c;
}

// root of every row's aggregate: pairs -> the smaller index; leftovers join the pair of their strongest matched
// neighbour (if strong enough) or stay alone; isolated rows get -1.  isroot feeds the numbering scan.
__global__ __launch_bounds__(FV_BLOCK) void amg_root_kernel(int64_t n, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ colind,
                                                             const double *__restrict__ vals, const int32_t *__restrict__ partner,
                                                             int32_t *__restrict__ rootof, int32_t *__restrict__ isroot, double theta)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i >= n)
        return;
    const int32_t pi = partner[i];
    int32_t root;
    if (pi >= 0)
        root = pi < (int32_t)i ? pi : (int32_t)i;
    else if (pi == AMG_ISOLATED)
        root = -1;
    else {
        double maxoff = 0.0, bestv = 0.0;
        int32_t best = -1;
        for (int32_t k = rowptr[i]; k < rowptr[i + 1]; k++) {
            const int32_t j = colind[k];
            if (j == i || j >= n)
                continue;
            const double s = -vals[k];
            if (s > maxoff)
                maxoff = s;
            if (s > 0.0 && partner[j] >= 0 && (s > bestv || (s == bestv && j < best))) {
                bestv = s;
                best = j;
            }
        }
        if (best >= 0 && bestv >= theta * maxoff) {
            const int32_t pj = partner[best];
            root = pj < best ? pj : best;
        } else
            root = (int32_t)i;
    }
    rootof[i] = root;
    isroot[i] = root == (int32_t)i;
}

__global__ __launch_bounds__(FV_BLOCK) void amg_number_kernel(int64_t n, const int32_t *__restrict__ rootof, const int32_t *__restrict__ cidx,
                                                               int32_t *__restrict__ agg)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i < n)
        agg[i] = rootof[i] < 0 ? -1 : cidx[rootof[i]];
}

// agg_out[i] = next[agg_out[i]] (composition of two aggregation maps; -1 stays -1)
__global__ __launch_bounds__(FV_BLOCK) void amg_compose_kernel(int64_t n, int32_t *__restrict__ agg, const int32_t *__restrict__ next)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i < n) {
        const int32_t a = agg[i];
        agg[i] = a < 0 ? -1 : next[a];
    }
}

__global__ __launch_bounds__(FV_BLOCK) void amg_count_kernel(int64_t n, const int32_t *__restrict__ agg, int32_t *__restrict__ cnt)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i < n && agg[i] >= 0)
        atomicAdd(&cnt[agg[i]], 1);
}

// sort keys of the member lists: the aggregate, rows without one last
__global__ __launch_bounds__(FV_BLOCK) void amg_member_keys_kernel(int64_t n, const int32_t *__restrict__ agg, uint32_t nc, uint32_t *__restrict__ key,
                                                                    int32_t *__restrict__ idx)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i < n) {
        key[i] = agg[i] < 0 ? nc : (uint32_t)agg[i];
        idx[i] = (int32_t)i;
    }
}

// Galerkin product with a piecewise-constant P, step 1: every stored entry (i, j, v) becomes (agg[i] * nc + agg[j], v);
// entries of rows or columns without an aggregate get the largest key and fall off the end of the sort
__global__ __launch_bounds__(FV_BLOCK) void amg_expand_kernel(int64_t n, int64_t ncols, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ colind,
                                                               const int32_t *__restrict__ agg, uint64_t nc, uint64_t *__restrict__ key)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i >= n)
        return;
    const int32_t I = agg[i];
    for (int32_t k = rowptr[i]; k < rowptr[i + 1]; k++) {
        const int32_t j = colind[k];
        const int32_t c = (j < ncols) ? agg[j] : -1;
        key[k] = (I < 0 || c < 0) ? nc * nc : (uint64_t)I * nc + (uint64_t)c;
    }
}

// step 2, after the stable sort: head[k] = 1 where a new (row, column) starts
__global__ __launch_bounds__(FV_BLOCK) void amg_heads_kernel(int64_t m, const uint64_t *__restrict__ key, uint64_t invalid, int32_t *__restrict__ head)
{
    const int64_t k = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (k < m)
        head[k] = (key[k] != invalid && (k == 0 || key[k] != key[k - 1])) ? 1 : 0;
}

// step 3: one thread per run sums it front to back (= the order of the fine entries: the sort is stable), writes the
// coarse entry and counts it for its row
__global__ __launch_bounds__(FV_BLOCK) void amg_runs_kernel(int64_t m, const uint64_t *__restrict__ key, const double *__restrict__ val,
                                                             const int32_t *__restrict__ head, const int32_t *__restrict__ pos, uint64_t nc,
                                                             int32_t *__restrict__ colind_c, double *__restrict__ vals_c,
                                                             int32_t *__restrict__ rowcnt)
{
    const int64_t k = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (k >= m || !head[k])
        return;
    const uint64_t kk = key[k];
    double s = val[k];
    for (int64_t q = k + 1; q < m && key[q] == kk; q++)
        s += val[q];
    const int32_t o = pos[k];
    colind_c[o] = (int32_t)(kk % nc);
    vals_c[o] = s;
    atomicAdd(&rowcnt[(int32_t)(kk / nc)], 1);
}

// The same product without the global sort, for the aggregates of one pairwise pass (one to three members, a few dozen entries):
// LANES lanes per coarse row stage its members' entries in LDS with their columns renamed, in fine order (members ascending, each
// row front to back), rank them by (coarse column, position) — a stable sort of at most CAP entries — and sum every run front to
// back: the sums and their order are those of the stable radix sort above, bit for bit.  COUNT pass: distinct columns per coarse
// row; FILL pass: the entries at rowptr_c.  A coarse row with more than CAP entries (or more members than lanes) is left to the launch
// with the larger CAP, one workgroup per row, which works through the list `big` (stat = {rows on the list, largest entry count}).
template <int LANES, int CAP, bool FILL, bool BIG>
__global__ __launch_bounds__(FV_BLOCK) void amg_merge_kernel(int64_t nc, int64_t n, const int32_t *__restrict__ memptr, const int32_t *__restrict__ mem,
                                                              const int32_t *__restrict__ rowptr, const int32_t *__restrict__ colind,
                                                              const double *__restrict__ vals, const int32_t *__restrict__ agg, int32_t *__restrict__ rowcnt,
                                                              const int32_t *__restrict__ rowptr_c, int32_t *__restrict__ colind_c, double *__restrict__ vals_c,
                                                              int32_t *__restrict__ big, int32_t *__restrict__ stat)
{
    constexpr int GROUPS = FV_BLOCK / LANES;
    __shared__ uint32_t s_col[GROUPS][CAP], s_scol[GROUPS][CAP];
    __shared__ double s_val[FILL ? GROUPS : 1][FILL ? CAP : 1], s_sval[FILL ? GROUPS : 1][FILL ? CAP : 1];
    __shared__ int32_t s_rs[BIG ? 1 : GROUPS][LANES], s_off[BIG ? 1 : GROUPS][LANES + 1]; // (used by the small launch only)
    const int g = threadIdx.x / LANES, sub = threadIdx.x % LANES;
    const int64_t I = BIG ? (int64_t)big[blockIdx.x] : (int64_t)blockIdx.x * GROUPS + g;
    int E = 0;
    int32_t m0 = 0, m1 = 0;
    if (I < nc) {
        m0 = memptr[I];
        m1 = memptr[I + 1];
    }
    if (BIG) {
        for (int32_t m = m0; m < m1; m++) {
            const int32_t row = mem[m];
            E += rowptr[row + 1] - rowptr[row];
        }
    } else {
        // the members' row pointers side by side (one lane each), their entry counts scanned over the group: all lanes then walk
        // the concatenated entries with independent loads instead of member after member
        const int nm = m1 - m0;
        int32_t rs = 0, len = 0;
        if (sub < nm && nm <= LANES) {
            const int32_t row = mem[m0 + sub];
            rs = rowptr[row];
            len = rowptr[row + 1] - rs;
        }
        int incl = len;
#pragma unroll
        for (int o = 1; o < LANES; o <<= 1) {
            const int t = __shfl_up(incl, o, LANES);
            if (sub >= o)
                incl += t;
        }
        E = nm <= LANES ? __shfl(incl, LANES - 1, LANES) : CAP + 1; // (more members than lanes: the larger launch)
        s_rs[g][sub] = rs;
        s_off[g][sub] = incl - len;
        if (sub == 0)
            s_off[g][LANES] = E;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (E > CAP) { // (uniform over the group)
        if (!FILL && sub == 0) {
            if (!BIG)
                big[atomicAdd(&stat[0], 1)] = (int32_t)I;
            atomicMax(&stat[1], E);
        }
        E = 0;
        m1 = m0;
    }
    if (BIG) {
        int at = 0;
        for (int32_t m = m0; m < m1; m++) {
            const int32_t row = mem[m];
            const int32_t s = rowptr[row], e = rowptr[row + 1];
            for (int32_t k = s + sub; k < e; k += LANES) {
                const int32_t j = colind[k];
                const int32_t c = j < n ? agg[j] : -1;
                s_col[g][at + (k - s)] = (uint32_t)c; // (no aggregate: 0xffffffff, sorts behind everything)
                if (FILL)
                    s_val[g][at + (k - s)] = vals[k];
            }
            at += e - s;
        }
        __syncthreads();
    } else {
        const int nm = m1 - m0;
        for (int e = sub; e < E; e += LANES) {
            int m = 0;
            for (int q = 1; q < nm; q++)
                m = (e >= s_off[g][q]) ? q : m;
            const int32_t k = s_rs[g][m] + (e - s_off[g][m]);
            const int32_t j = colind[k];
            const int32_t c = j < n ? agg[j] : -1;
            s_col[g][e] = (uint32_t)c;
            if (FILL)
                s_val[g][e] = vals[k];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    for (int e = sub; e < E; e += LANES) {
        const uint32_t c = s_col[g][e];
        int rank = 0;
        for (int q = 0; q < E; q++) {
            const uint32_t cq = s_col[g][q];
            rank += (cq < c) || (cq == c && q < e);
        }
        s_scol[g][rank] = c;
        if (FILL)
            s_sval[g][rank] = s_val[g][e];
    }
    if (BIG)
        __syncthreads();
    else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    int heads = 0;
    for (int e = sub; e < E; e += LANES) {
        const uint32_t c = s_scol[g][e];
        if (c == 0xffffffffu || (e > 0 && s_scol[g][e - 1] == c))
            continue;
        heads++;
        if (FILL) {
            int pos = 0; // distinct columns before this one
            for (int q = 1; q <= e; q++)
                pos += s_scol[g][q] != s_scol[g][q - 1];
            double v = s_sval[g][e];
            for (int q = e + 1; q < E && s_scol[g][q] == c; q++)
                v += s_sval[g][q];
            const int32_t o = rowptr_c[I] + pos;
            colind_c[o] = (int32_t)c;
            vals_c[o] = v;
        }
    }
    if (!FILL) {
        if (BIG) {
            __shared__ int s_heads;
            if (threadIdx.x == 0)
                s_heads = 0;
            __syncthreads();
            if (heads)
                atomicAdd(&s_heads, heads);
            __syncthreads();
            if (threadIdx.x == 0 && E > 0)
                rowcnt[I] = s_heads;
        } else {
#pragma unroll
            for (int off = LANES / 2; off > 0; off >>= 1)
                heads += __shfl_xor(heads, off, LANES);
            if (sub == 0 && E > 0)
                rowcnt[I] = heads;
        }
    }
}

__global__ __launch_bounds__(FV_BLOCK) void amg_aggD_kernel(int64_t nc, const int32_t *__restrict__ memptr, const int32_t *__restrict__ mem,
                                                             const double *__restrict__ D, double *__restrict__ Dc)
{
    const int64_t I = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (I >= nc)
        return;
    double s = 0.0;
    for (int32_t k = memptr[I]; k < memptr[I + 1]; k++)
        s += D[mem[k]];
    Dc[I] = s;
}

__global__ __launch_bounds__(FV_BLOCK) void amg_diag_kernel(int64_t n, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ colind,
                                                             const double *__restrict__ vals, double *__restrict__ diag)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i >= n)
        return;
    double d = 0.0;
    for (int32_t k = rowptr[i]; k < rowptr[i + 1]; k++)
        if (colind[k] == (int32_t)i)
            d += vals[k];
    diag[i] = d;
}

__global__ __launch_bounds__(FV_BLOCK) void amg_dinv_kernel(int64_t n, const double *__restrict__ diag, const double *__restrict__ D, double sigma,
                                                             double *__restrict__ dinv)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i < n) {
        const double d = diag[i] + (D ? sigma * D[i] : 0.0);
        dinv[i] = d > 0.0 ? 1.0 / d : 0.0;
    }
}

// ------------------------------------------------------------------ cycle kernels
// y = (A + sigma D) x, LPR lanes per row (coarse levels: irregular rows of ~10-30 entries)
template <int LPR>
__global__ __launch_bounds__(FV_BLOCK) void amg_spmv_kernel(int64_t n, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ colind,
                                                             const double *__restrict__ vals, const double *__restrict__ x,
                                                             double *__restrict__ y, const double *__restrict__ D, double sigma)
{
    const int64_t row = (int64_t)blockIdx.x * (FV_BLOCK / LPR) + threadIdx.x / LPR;
    const int sub = threadIdx.x % LPR;
    double s = 0.0;
    if (row < n) {
        const int32_t e = rowptr[row + 1];
        for (int32_t k = rowptr[row] + sub; k < e; k += LPR)
            s += vals[k] * x[colind[k]];
    }
#pragma unroll
    for (int off = LPR / 2; off > 0; off >>= 1)
        s += __shfl_xor(s, off, LPR);
    if (sub == 0 && row < n) {
        if (D)
            s += sigma * D[row] * x[row];
        y[row] = s;
    }
}

// ---- products of the coarse levels with the consumer of the product folded in.  A cycle visits levels 1 and 2 several times per
// PCG iteration; every vector pass it does not launch is a launch and 16-40 bytes per row less.  What follows the row sum:
//   PLAIN   y = A x
//   SMOOTH  xout = x + omega dinv (b - A x)            (the cycle's last smoothing pass; xout is another vector than x: Jacobi)
//   KDOT2   y = A x, partial sums of x.y and x.b        (K-cycle: v1 = A c1, rho1, alpha1)
//   KDOT3   y = A x, partial sums of x.v1, x.y, x.r1    (K-cycle: v2 = A c2, gamma, beta, alpha2)
enum { AMG_PLAIN = 0, AMG_SMOOTH = 1, AMG_KDOT2 = 2, AMG_KDOT3 = 3 };
constexpr int AMG_KG = FV_MAX_PARTIALS; // stride of the K-cycle's partial sums: one per block of the launch that forms them

struct AmgOp {
    int64_t n;
    const int32_t *rowptr, *colind;
    const double *vals;
    const double *x;
    const double *D; // storage term of the level (null: none), times sigma
    double sigma;
    double *y;
    const double *b, *dinv;
    double omega;
    double *xout;
    const double *v1, *r1;
    double *part;
};

template <int EPI>
__device__ inline void amg_row_done(const AmgOp &a, int64_t row, double s, double &acc0, double &acc1, double &acc2)
{
    const double xr = (EPI != AMG_PLAIN || a.D) ? a.x[row] : 0.0;
    if (a.D)
        s += a.sigma * a.D[row] * xr;
    if (EPI == AMG_SMOOTH) {
        a.xout[row] = xr + a.omega * a.dinv[row] * (a.b[row] - s);
        return;
    }
    a.y[row] = s;
    if (EPI == AMG_KDOT2) {
        acc0 += xr * s;
        acc1 += xr * a.b[row];
    }
    if (EPI == AMG_KDOT3) {
        acc0 += xr * a.v1[row];
        acc1 += xr * s;
        acc2 += xr * a.r1[row];
    }
}

// (measured, round 4: letting the block that finishes last add the partials up — so that the K-cycle's vector kernels read five numbers
// instead of reducing 2 048 partials each — needs a device-scope release fence in every block, i.e. a write-back of the XCD's L2 with
// the product's output in it: the product went from 55 to 284 us.  The partials stay.)
template <int EPI>
__device__ inline void amg_block_done(const AmgOp &a, double acc0, double acc1, double acc2, double *smem)
{
    if (EPI == AMG_KDOT2) {
        const double t0 = block_sum(acc0, smem);
        const double t1 = block_sum(acc1, smem);
        if (threadIdx.x == 0) {
            a.part[blockIdx.x] = t0;
            a.part[AMG_KG + blockIdx.x] = t1;
        }
    }
    if (EPI == AMG_KDOT3) {
        const double t0 = block_sum(acc0, smem);
        const double t1 = block_sum(acc1, smem);
        const double t2 = block_sum(acc2, smem);
        if (threadIdx.x == 0) {
            a.part[2 * AMG_KG + blockIdx.x] = t0;
            a.part[3 * AMG_KG + blockIdx.x] = t1;
            a.part[4 * AMG_KG + blockIdx.x] = t2;
        }
    }
}

// Large levels: a wave takes 64 consecutive rows, streams their entries (one contiguous range of vals / colind) with lane-contiguous
// 16- and 8-byte loads, leaves the products in its own LDS tile and every lane sums its row in column order (the form of
// spmv_wstream_kernel, fv_spmv.hip); every XCD walks its own contiguous share of the row groups.
template <int WT, int EPI>
__global__ __launch_bounds__(FV_BLOCK) void amg_stream_kernel(AmgOp a)
{
    constexpr int NIT = WT / 128, WPB = FV_BLOCK / 64;
    __shared__ double prod_all[WPB][WT + 2];
    __shared__ double smem[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double *prod = prod_all[wave];
    const int64_t ngroups = (a.n + 63) >> 6;
    const int64_t per_xcd = (ngroups + 7) >> 3;
    const int64_t pstride = (int64_t)(gridDim.x >> 3) * WPB;
    const int64_t xbase = (int64_t)(blockIdx.x & 7) * per_xcd;
    const int64_t xend = (xbase + per_xcd < ngroups) ? xbase + per_xcd : ngroups;
    double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0;
    int32_t s = 0, e = 0;
    int64_t pos = xbase + (int64_t)(blockIdx.x >> 3) * WPB + wave;
    if (pos < xend && (pos << 6) + lane < a.n) {
        s = a.rowptr[(pos << 6) + lane];
        e = a.rowptr[(pos << 6) + lane + 1];
    }
    for (; pos < xend; pos += pstride) {
        const int64_t r0 = pos << 6;
        const int nr = (int)((a.n - r0 < 64) ? (a.n - r0) : 64);
        const int32_t my_s = s, my_e = e;
        const int32_t k0 = __builtin_amdgcn_readfirstlane(my_s);
        const int32_t k1 = __builtin_amdgcn_readlane(my_e, nr - 1);
        const int32_t ka = k0 & ~1;
        s = 0;
        e = 0;
        if (pos + pstride < xend && ((pos + pstride) << 6) + lane < a.n) { // the next group's row pointers, one pass ahead
            s = a.rowptr[((pos + pstride) << 6) + lane];
            e = a.rowptr[((pos + pstride) << 6) + lane + 1];
        }
        double sum = 0.0;
        if (k1 - ka <= WT) {
            double2 v[NIT];
            int2 c[NIT];
#pragma unroll
            for (int it = 0; it < NIT; it++) {
                const int32_t j = ka + 2 * (lane + it * 64);
                if (j < k1) { // (vals / colind carry two padding entries past nnz)
                    v[it] = *reinterpret_cast<const double2 *>(a.vals + j);
                    c[it] = *reinterpret_cast<const int2 *>(a.colind + j);
                }
            }
#pragma unroll
            for (int it = 0; it < NIT; it++) {
                const int32_t j = ka + 2 * (lane + it * 64);
                if (j < k1) {
                    double2 pr;
                    pr.x = v[it].x * a.x[c[it].x];
                    pr.y = v[it].y * a.x[c[it].y];
                    *reinterpret_cast<double2 *>(prod + (j - ka)) = pr;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (int32_t k = my_s - ka, ke = my_e - ka; k < ke; k++)
                sum += prod[k];
            __builtin_amdgcn_wave_barrier(); // reads done before the next pass overwrites the tile
        } else { // more than WT entries in 64 rows: every lane walks its own row
            for (int32_t k = my_s; k < my_e; k++)
                sum += a.vals[k] * a.x[a.colind[k]];
        }
        if (lane < nr)
            amg_row_done<EPI>(a, r0 + lane, sum, acc0, acc1, acc2);
    }
    amg_block_done<EPI>(a, acc0, acc1, acc2, smem);
}

// Small levels (launch-bound either way): LPR lanes per row, grid-stride
template <int LPR, int EPI>
__global__ __launch_bounds__(FV_BLOCK) void amg_rows_kernel(AmgOp a)
{
    __shared__ double smem[4];
    constexpr int RPB = FV_BLOCK / LPR;
    const int sub = threadIdx.x % LPR;
    double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0;
    for (int64_t row0 = (int64_t)blockIdx.x * RPB; row0 < a.n; row0 += (int64_t)gridDim.x * RPB) {
        const int64_t row = row0 + threadIdx.x / LPR;
        double s = 0.0;
        if (row < a.n) {
            const int32_t e = a.rowptr[row + 1];
            for (int32_t k = a.rowptr[row] + sub; k < e; k += LPR)
                s += a.vals[k] * a.x[a.colind[k]];
        }
#pragma unroll
        for (int off = LPR / 2; off > 0; off >>= 1)
            s += __shfl_xor(s, off, LPR);
        if (sub == 0 && row < a.n)
            amg_row_done<EPI>(a, row, s, acc0, acc1, acc2);
    }
    amg_block_done<EPI>(a, acc0, acc1, acc2, smem);
}

__global__ __launch_bounds__(FV_BLOCK) void amg_smooth0_kernel(int64_t n, const double *__restrict__ dinv, const double *__restrict__ b, double omega,
                                                                double *__restrict__ x)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i < n)
        x[i] = omega * dinv[i] * b[i];
}

__global__ __launch_bounds__(FV_BLOCK) void amg_smooth_kernel(int64_t n, const double *__restrict__ dinv, const double *__restrict__ b,
                                                               const double *__restrict__ t, double omega, double *__restrict__ x)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i < n)
        x[i] += omega * dinv[i] * (b[i] - t[i]);
}

// the last smoothing pass of the top level, with the PCG's r.z folded in (b is r, x is z): one partial per block
__global__ __launch_bounds__(FV_BLOCK) void amg_smooth_dot_kernel(int64_t n, const double *__restrict__ dinv, const double *__restrict__ b,
                                                                   const double *__restrict__ t, double omega, double *__restrict__ x,
                                                                   double *__restrict__ part, const double *__restrict__ q, double *__restrict__ part_q)
{
    __shared__ double smem[4];
    double acc = 0.0, accq = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x; i < n; i += vec_stride()) {
        const double bi = b[i];
        const double xi = x[i] + omega * dinv[i] * (bi - t[i]);
        x[i] = xi;
        acc += bi * xi;
        if (q)
            accq += q[i] * xi; // flexible PCG (a K-cycle is not a fixed linear operator): z_new.q for beta = -alpha z_new.q / (r.z)_old
    }
    const double s = block_sum(acc, smem);
    const double sq = q ? block_sum(accq, smem) : 0.0;
    if (threadIdx.x == 0) {
        part[blockIdx.x] = s;
        if (q)
            part_q[blockIdx.x] = sq;
    }
}

// b_c[I] = sum over the members of (b - t): LPA lanes per aggregate, fixed lane partition + shuffle tree, so the sum order is fixed
template <int LPA>
__global__ __launch_bounds__(FV_BLOCK) void amg_restrict_kernel(int64_t nc, const int32_t *__restrict__ memptr, const int32_t *__restrict__ mem,
                                                                 const double *__restrict__ b, const double *__restrict__ t,
                                                                 double *__restrict__ bc, const double *__restrict__ dinv_c, double omega,
                                                                 double *__restrict__ uc)
{
    const int64_t I = (int64_t)blockIdx.x * (FV_BLOCK / LPA) + threadIdx.x / LPA;
    const int sub = threadIdx.x % LPA;
    double s = 0.0;
    if (I < nc) {
        const int32_t e = memptr[I + 1];
        for (int32_t k = memptr[I] + sub; k < e; k += LPA) {
            const int32_t m = mem[k];
            s += b[m] - t[m];
        }
    }
#pragma unroll
    for (int off = LPA / 2; off > 0; off >>= 1)
        s += __shfl_xor(s, off, LPA);
    if (sub == 0 && I < nc) {
        bc[I] = s;
        if (uc) // the coarse cycle's first smoothing pass from a zero iterate, while b_c is in a register
            uc[I] = omega * dinv_c[I] * s;
    }
}

__global__ __launch_bounds__(FV_BLOCK) void amg_prolong_kernel(int64_t n, const int32_t *__restrict__ agg, const double *__restrict__ xc,
                                                                double *__restrict__ x)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i < n) {
        const int32_t a = agg[i];
        if (a >= 0)
            x[i] += xc[a];
    }
}

// (the preconditioner's role: AlgebraicMultigrid.ruge_stuben + aspreconditioner at /root/reference/src/FiniteVolume.jl:159-161 and
// src/transient.jl:54; iteration-level parity with it is unpinned by the survey's own decision)
// ------------------------------------------------------------------ K-cycle (Notay's AGMG): a coarse system A_l e = b is solved by
// two steps of flexible CG preconditioned by the cycle of level l instead of by one application of that cycle:
//   c1 = cycle(b), v1 = A c1, rho1 = c1.v1, alpha1 = c1.b, r1 = b - (alpha1 / rho1) v1,
//   c2 = cycle(r1), v2 = A c2, gamma = c2.v1, beta = c2.v2, alpha2 = c2.r1, rho2 = beta - gamma^2 / rho1,
//   e = (alpha1 / rho1 - gamma alpha2 / (rho1 rho2)) c1 + (alpha2 / rho2) c2.
// Unsmoothed aggregation loses a factor per level in a V-cycle; the K-cycle keeps every level near its two-grid rate.  All scalars
// stay on the device: the dot kernels leave per-block partials, the consumers reduce them in their first microseconds.
__global__ __launch_bounds__(FV_BLOCK) void amg_kres_kernel(int64_t n, const double *__restrict__ b, const double *__restrict__ v1,
                                                             const double *__restrict__ part, int nparts, double *__restrict__ r1,
                                                             const double *__restrict__ dinv, double omega, double *__restrict__ u)
{
    __shared__ double smem[4];
    const double rho1 = reduce_partials(part, nparts, smem);
    const double alpha1 = reduce_partials(part + AMG_KG, nparts, smem);
    const double f = rho1 > 0.0 ? alpha1 / rho1 : 0.0;
    for (int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * FV_BLOCK) {
        const double ri = b[i] - f * v1[i];
        r1[i] = ri;
        u[i] = omega * dinv[i] * ri; // (the second cycle's first smoothing pass)
    }
}

// x holds c1 on entry, the level's correction on exit
__global__ __launch_bounds__(FV_BLOCK) void amg_kcomb_kernel(int64_t n, const double *__restrict__ c2, const double *__restrict__ part, int nparts,
                                                              double *__restrict__ x)
{
    __shared__ double smem[4];
    const double rho1 = reduce_partials(part, nparts, smem);
    const double alpha1 = reduce_partials(part + AMG_KG, nparts, smem);
    const double gamma = reduce_partials(part + 2 * AMG_KG, nparts, smem);
    const double beta = reduce_partials(part + 3 * AMG_KG, nparts, smem);
    const double alpha2 = reduce_partials(part + 4 * AMG_KG, nparts, smem);
    double f1 = 1.0, f2 = 0.0; // breakdown of the first step (rho1 <= 0: c1 = 0 or an indefinite cycle): the plain cycle's c1
    if (rho1 > 0.0) {
        const double rho2 = beta - gamma * gamma / rho1;
        f1 = alpha1 / rho1;
        if (rho2 > 0.0) {
            f1 -= gamma * alpha2 / (rho1 * rho2);
            f2 = alpha2 / rho2;
        }
    }
    for (int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * FV_BLOCK)
        x[i] = f1 * x[i] + f2 * c2[i];
}

// ------------------------------------------------------------------ coarsest level: explicit inverse by Gauss-Jordan
__global__ __launch_bounds__(FV_BLOCK) void amg_dense_fill_kernel(int64_t n, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ colind,
                                                                   const double *__restrict__ vals, const double *__restrict__ D, double sigma,
                                                                   double *__restrict__ M)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i >= n)
        return;
    for (int32_t k = rowptr[i]; k < rowptr[i + 1]; k++)
        M[i * n + colind[k]] += vals[k]; // one thread per row: no race
    if (D)
        M[i * n + i] += sigma * D[i];
}

// Block Gauss-Jordan inversion (no pivoting: the matrix is SPD), GJB pivots per launch, from one copy of the matrix into the other.
// With K the pivot block, P = M[K,K], R = M[K,:], C = M[:,K]:   M'[K,K] = P^-1,  M'[K,j] = P^-1 R[:,j],  M'[i,K] = -C[i,:] P^-1,
// M'[i,j] = M[i,j] - C[i,:] P^-1 R[:,j]  (i, j outside K) — the sweep of the scalar algorithm, GJB pivots at a time; after the last
// block M' is the inverse.  Every workgroup inverts the small P for itself in LDS (its first wave, one entry per lane).
constexpr int GJB = 8;
__global__ __launch_bounds__(FV_BLOCK) void amg_gj_block_kernel(int64_t n, int64_t k0, int B, const double *__restrict__ M, double *__restrict__ Mo)
{
    __shared__ double P[GJB][GJB];
    if (threadIdx.x < GJB * GJB) {
        const int a = threadIdx.x / GJB, b = threadIdx.x % GJB;
        P[a][b] = (a < B && b < B) ? M[(k0 + a) * n + (k0 + b)] : (a == b ? 1.0 : 0.0);
    }
    __syncthreads();
    if (threadIdx.x < GJB * GJB) { // the first wave inverts the block by the scalar sweep, one entry per lane (padding rows: identity)
        const int a = threadIdx.x / GJB, b = threadIdx.x % GJB;
        for (int k = 0; k < GJB; k++) {
            const double ip = 1.0 / P[k][k], pak = P[a][k], pkb = P[k][b], pab = P[a][b];
            const double v = (a == k) ? ((b == k) ? ip : pkb * ip) : ((b == k) ? -pak * ip : pab - pak * pkb * ip);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier(); // (every lane has read the old block)
            P[a][b] = v;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
    __syncthreads();
    const int64_t j = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    const int64_t i = blockIdx.y;
    if (j >= n)
        return;
    const bool iK = i >= k0 && i < k0 + B, jK = j >= k0 && j < k0 + B;
    double v;
    if (iK && jK)
        v = P[i - k0][j - k0];
    else if (iK) { // P^-1 R[:, j]
        v = 0.0;
        for (int b = 0; b < B; b++)
            v += P[i - k0][b] * M[(k0 + b) * n + j];
    } else if (jK) { // -C[i, :] P^-1
        v = 0.0;
        for (int a = 0; a < B; a++)
            v -= M[i * n + (k0 + a)] * P[a][j - k0];
    } else {
        v = M[i * n + j];
        for (int a = 0; a < B; a++) {
            double w = 0.0; // (P^-1 R[:, j])_a
            for (int b = 0; b < B; b++)
                w += P[a][b] * M[(k0 + b) * n + j];
            v -= M[i * n + (k0 + a)] * w;
        }
    }
    Mo[i * n + j] = v;
}

// y = Minv b, one wave per row
__global__ __launch_bounds__(FV_BLOCK) void amg_gemv_kernel(int64_t n, const double *__restrict__ M, const double *__restrict__ b, double *__restrict__ y)
{
    const int64_t row = (int64_t)blockIdx.x * (FV_BLOCK / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    double s = 0.0;
    if (row < n)
        for (int64_t j = lane; j < n; j += 64)
            s += M[row * n + j] * b[j];
    s = wave_sum(s);
    if (lane == 0 && row < n)
        y[row] = s;
}

// ------------------------------------------------------------------ host side: hierarchy
// one product of level L with the epilogue `epi` (the AmgOp fields of that epilogue filled in by the caller); *grid: blocks launched
// (= partial sums per quantity of the K-cycle epilogues)
static int amg_level_op(fv_ctx *ctx, const AmgLevel *L, int epi, AmgOp op, double sigma, int *grid = nullptr)
{
    op.n = L->n;
    op.rowptr = L->rowptr;
    op.colind = L->colind;
    op.vals = L->vals;
    op.D = (sigma != 0.0) ? L->D : nullptr;
    op.sigma = sigma;
    const double avg = L->n > 0 ? (double)L->nnz / (double)L->n : 0.0;
    const dim3 blk(FV_BLOCK);
    int G;
#define FV_AMG_OP(KERNEL)                                                                                \
    switch (epi) {                                                                                       \
    case AMG_SMOOTH: hipLaunchKernelGGL((KERNEL, AMG_SMOOTH>), dim3(G), blk, 0, ctx->stream, op); break; \
    case AMG_KDOT2: hipLaunchKernelGGL((KERNEL, AMG_KDOT2>), dim3(G), blk, 0, ctx->stream, op); break;   \
    case AMG_KDOT3: hipLaunchKernelGGL((KERNEL, AMG_KDOT3>), dim3(G), blk, 0, ctx->stream, op); break;   \
    default: hipLaunchKernelGGL((KERNEL, AMG_PLAIN>), dim3(G), blk, 0, ctx->stream, op); break;          \
    }
    // (measured, round 4: the same products with the values rounded to single precision — a third of the matrix bytes less — take
    // the same time: on these irregular rows the kernel is bound by its 8-byte gathers of x, not by the streams)
    if (g_amg_stream && L->o_vals.p && L->n >= g_amg_stream && avg <= 14.0) {
        // 4 row groups per block and pass, a multiple of 8 blocks (XCD shares), at most 8 blocks per CU
        int64_t g = ((((L->n + 63) >> 6) + 3) / 4 + 7) / 8 * 8;
        const int64_t cap = (int64_t)ctx->num_cus * 8 / 8 * 8;
        if (g > cap)
            g = cap;
        if (g > AMG_KG)
            g = AMG_KG;
        if (g < 8)
            g = 8;
        G = (int)g;
        FV_AMG_OP(amg_stream_kernel<1024)
    } else if (avg > 12.0) {
        const unsigned nb = fv_blocks(L->n, FV_BLOCK / 16);
        G = nb < (unsigned)AMG_KG ? (int)nb : AMG_KG;
        FV_AMG_OP(amg_rows_kernel<16)
    } else {
        const unsigned nb = fv_blocks(L->n, FV_BLOCK / 8);
        G = nb < (unsigned)AMG_KG ? (int)nb : AMG_KG;
        FV_AMG_OP(amg_rows_kernel<8)
    }
#undef FV_AMG_OP
    FV_LAUNCH_CHECK(ctx);
    if (grid)
        *grid = G;
    return FV_OK;
}

static int amg_level_spmv(fv_ctx *ctx, const AmgLevel *L, const double *x, double *y, double sigma)
{
    AmgOp op{};
    op.x = x;
    op.y = y;
    return amg_level_op(ctx, L, AMG_PLAIN, op, sigma);
}

// One pairwise pass on the CSR (n, rowptr, colind, vals): agg (n entries) and the number of aggregates.
static int amg_pairwise(fv_ctx *ctx, int64_t n, const int32_t *rowptr, const int32_t *colind, const double *vals, DevBuf<int32_t> &agg,
                        int64_t *nc)
{
    DevBuf<int32_t> partner, cand, rootof, isroot, cidx;
    FV_TRY(partner.alloc(ctx, (size_t)n));
    FV_TRY(cand.alloc(ctx, (size_t)n));
    FV_TRY(rootof.alloc(ctx, (size_t)n));
    FV_TRY(isroot.alloc(ctx, (size_t)n));
    FV_TRY(cidx.alloc(ctx, (size_t)n + 1));
    FV_TRY(agg.alloc(ctx, (size_t)n));
    FV_HIP(ctx, hipMemsetAsync(partner.p, 0xff, (size_t)n * sizeof(int32_t), ctx->stream)); // AMG_UNMATCHED
    const dim3 g(fv_blocks(n)), b(FV_BLOCK);
    // (measured, round 4: walking a compacted list of the still unmatched rows in the later rounds is SLOWER — 1.15 ms against 0.51 ms
    // per round at 16.6 M rows with a fifth of them on the list: the rows of a wave are then no longer neighbours in colind / vals)
    DevBuf<double> cut;
    FV_TRY(cut.alloc(ctx, (size_t)n));
    hipLaunchKernelGGL(amg_cut_kernel, g, b, 0, ctx->stream, n, rowptr, colind, vals, g_theta, cut.p);
    for (int r = 0; r < g_rounds; r++) {
        hipLaunchKernelGGL(amg_pick_kernel, g, b, 0, ctx->stream, n, rowptr, colind, vals, (const int32_t *)partner.p, cand.p, (const double *)cut.p);
        hipLaunchKernelGGL(amg_match_kernel, g, b, 0, ctx->stream, n, (const int32_t *)cand.p, partner.p);
        FV_LAUNCH_CHECK(ctx);
    }
    hipLaunchKernelGGL(amg_root_kernel, g, b, 0, ctx->stream, n, rowptr, colind, vals, (const int32_t *)partner.p, rootof.p, isroot.p, g_theta);
    FV_LAUNCH_CHECK(ctx);
    FV_TRY(fv_exclusive_scan_i32(ctx, isroot.p, cidx.p, n, nc));
    hipLaunchKernelGGL(amg_number_kernel, g, b, 0, ctx->stream, n, (const int32_t *)rootof.p, (const int32_t *)cidx.p, agg.p);
    FV_LAUNCH_CHECK(ctx);
    FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FV_OK;
}

static unsigned amg_bits(uint64_t maxval)
{
    unsigned b = 1;
    while (b < 64 && (maxval >> b) != 0)
        b++;
    return b;
}

// stable LSD radix sort of (key, value) pairs (rocPRIM): the one library primitive of the set-up.  Stability is what
// keeps every later sum in a fixed order.
template <class K, class V>
static int amg_sort_pairs(fv_ctx *ctx, const K *kin, K *kout, const V *vin, V *vout, size_t count, unsigned end_bit)
{
    size_t bytes = 0;
    FV_HIP(ctx, rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, vin, vout, count, 0u, end_bit, ctx->stream));
    DevBuf<char> tmp;
    FV_TRY(tmp.alloc(ctx, bytes));
    FV_HIP(ctx, rocprim::radix_sort_pairs((void *)tmp.p, bytes, kin, kout, vin, vout, count, 0u, end_bit, ctx->stream));
    FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FV_OK;
}

// members of every aggregate, ascending (CSR-like: memptr, mem)
static int amg_members(fv_ctx *ctx, int64_t n, const int32_t *agg, int64_t nc, DevBuf<int32_t> &memptr, DevBuf<int32_t> &mem)
{
    DevBuf<int32_t> cnt, idx;
    DevBuf<uint32_t> key, key2;
    FV_TRY(cnt.alloc(ctx, (size_t)nc + 1));
    FV_TRY(cnt.zero(ctx));
    FV_TRY(memptr.alloc(ctx, (size_t)nc + 1));
    hipLaunchKernelGGL(amg_count_kernel, dim3(fv_blocks(n)), dim3(FV_BLOCK), 0, ctx->stream, n, agg, cnt.p);
    FV_LAUNCH_CHECK(ctx);
    int64_t total = 0;
    FV_TRY(fv_exclusive_scan_i32(ctx, cnt.p, memptr.p, nc, &total));
    FV_TRY(key.alloc(ctx, (size_t)n));
    FV_TRY(key2.alloc(ctx, (size_t)n));
    FV_TRY(idx.alloc(ctx, (size_t)n));
    FV_TRY(mem.alloc(ctx, (size_t)n)); // the first `total` entries are the lists; rows without an aggregate trail behind
    hipLaunchKernelGGL(amg_member_keys_kernel, dim3(fv_blocks(n)), dim3(FV_BLOCK), 0, ctx->stream, n, agg, (uint32_t)nc, key.p, idx.p);
    FV_LAUNCH_CHECK(ctx);
    return amg_sort_pairs<uint32_t, int32_t>(ctx, key.p, key2.p, idx.p, mem.p, (size_t)n, amg_bits((uint64_t)nc));
}

// Galerkin coarse operator P^T A P for the aggregation `agg`: CSR with sorted columns + aggregated storage diagonal
// ncols: columns below it have an aggregate in agg (the rows' own count; a row block with its halo slots for the gathered levels)
static int amg_galerkin(fv_ctx *ctx, int64_t n, int64_t nnz, const int32_t *rowptr, const int32_t *colind, const double *vals, const double *D,
                        const int32_t *agg, int64_t nc, const int32_t *memptr, const int32_t *mem, DevBuf<int32_t> &rowptr_c,
                        DevBuf<int32_t> &colind_c, DevBuf<double> &vals_c, DevBuf<double> &Dc, int64_t *nnz_c, int64_t ncols = -1)
{
    if (ncols < 0)
        ncols = n;
    DevBuf<uint64_t> key, key2;
    DevBuf<double> val2;
    DevBuf<int32_t> head, pos, rowcnt;
    // the merge of the member rows (amg_merge_kernel); FV_AMG_GALERKIN=sort keeps the global sort (the tests compare the two, bit for bit)
    const char *how = getenv("FV_AMG_GALERKIN");
    if (!(how && !strcmp(how, "sort"))) {
        constexpr int CAP_S = 64, CAP_B = 2048;
        DevBuf<int32_t> big, stat;
        FV_TRY(rowcnt.alloc(ctx, (size_t)nc + 1));
        FV_TRY(rowcnt.zero(ctx));
        FV_TRY(big.alloc(ctx, (size_t)nc));
        FV_TRY(stat.alloc(ctx, 2));
        FV_TRY(stat.zero(ctx));
        const dim3 gs(fv_blocks(nc, FV_BLOCK / 16)), blk(FV_BLOCK);
        hipLaunchKernelGGL((amg_merge_kernel<16, CAP_S, false, false>), gs, blk, 0, ctx->stream, nc, ncols, memptr, mem, rowptr, colind, vals, agg, rowcnt.p,
                           (const int32_t *)nullptr, (int32_t *)nullptr, (double *)nullptr, big.p, stat.p);
        FV_LAUNCH_CHECK(ctx);
        int32_t hstat[2] = {0, 0};
        FV_HIP(ctx, hipMemcpyAsync(hstat, stat.p, sizeof hstat, hipMemcpyDeviceToHost, ctx->stream));
        FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (hstat[1] <= CAP_B) {
            const int nbig = hstat[0];
            if (nbig > 0) {
                hipLaunchKernelGGL((amg_merge_kernel<FV_BLOCK, CAP_B, false, true>), dim3(nbig), blk, 0, ctx->stream, nc, ncols, memptr, mem, rowptr, colind, vals,
                                   agg, rowcnt.p, (const int32_t *)nullptr, (int32_t *)nullptr, (double *)nullptr, big.p, stat.p);
                FV_LAUNCH_CHECK(ctx);
                // (an aggregate of more members than the small launch has lanes is passed on without its entry count: the larger
                // launch has counted it now)
                FV_HIP(ctx, hipMemcpyAsync(hstat, stat.p, sizeof hstat, hipMemcpyDeviceToHost, ctx->stream));
                FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
            }
        }
        if (hstat[1] <= CAP_B) {
            const int nbig = hstat[0];
            FV_TRY(rowptr_c.alloc(ctx, (size_t)nc + 1));
            FV_TRY(fv_exclusive_scan_i32(ctx, rowcnt.p, rowptr_c.p, nc, nnz_c));
            FV_TRY(colind_c.alloc(ctx, (size_t)*nnz_c + 2)); // (+ two zero entries: the wave-stream kernel reads pairs)
            FV_TRY(vals_c.alloc(ctx, (size_t)*nnz_c + 2));
            FV_HIP(ctx, hipMemsetAsync(colind_c.p + *nnz_c, 0, 2 * sizeof(int32_t), ctx->stream));
            FV_HIP(ctx, hipMemsetAsync(vals_c.p + *nnz_c, 0, 2 * sizeof(double), ctx->stream));
            hipLaunchKernelGGL((amg_merge_kernel<16, CAP_S, true, false>), gs, blk, 0, ctx->stream, nc, ncols, memptr, mem, rowptr, colind, vals, agg,
                               (int32_t *)nullptr, (const int32_t *)rowptr_c.p, colind_c.p, vals_c.p, big.p, stat.p);
            if (nbig > 0)
                hipLaunchKernelGGL((amg_merge_kernel<FV_BLOCK, CAP_B, true, true>), dim3(nbig), blk, 0, ctx->stream, nc, ncols, memptr, mem, rowptr, colind, vals,
                                   agg, (int32_t *)nullptr, (const int32_t *)rowptr_c.p, colind_c.p, vals_c.p, big.p, stat.p);
            FV_LAUNCH_CHECK(ctx);
            if (D) {
                FV_TRY(Dc.alloc(ctx, (size_t)nc));
                hipLaunchKernelGGL(amg_aggD_kernel, dim3(fv_blocks(nc)), dim3(FV_BLOCK), 0, ctx->stream, nc, memptr, mem, D, Dc.p);
                FV_LAUNCH_CHECK(ctx);
            }
            FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
            return FV_OK;
        }
        rowcnt.release(); // an aggregate with more than CAP_B entries: the sort takes the whole pass
    }
    FV_TRY(key.alloc(ctx, (size_t)nnz));
    FV_TRY(key2.alloc(ctx, (size_t)nnz));
    FV_TRY(val2.alloc(ctx, (size_t)nnz));
    const uint64_t unc = (uint64_t)nc;
    hipLaunchKernelGGL(amg_expand_kernel, dim3(fv_blocks(n)), dim3(FV_BLOCK), 0, ctx->stream, n, ncols, rowptr, colind, agg, unc, key.p);
    FV_LAUNCH_CHECK(ctx);
    FV_TRY((amg_sort_pairs<uint64_t, double>(ctx, key.p, key2.p, vals, val2.p, (size_t)nnz, amg_bits(unc * unc))));
    key.release();
    FV_TRY(head.alloc(ctx, (size_t)nnz));
    FV_TRY(pos.alloc(ctx, (size_t)nnz + 1));
    hipLaunchKernelGGL(amg_heads_kernel, dim3(fv_blocks(nnz)), dim3(FV_BLOCK), 0, ctx->stream, nnz, (const uint64_t *)key2.p, unc * unc, head.p);
    FV_LAUNCH_CHECK(ctx);
    FV_TRY(fv_exclusive_scan_i32(ctx, head.p, pos.p, nnz, nnz_c));
    FV_TRY(colind_c.alloc(ctx, (size_t)*nnz_c + 2)); // (+ two zero entries: the wave-stream kernel reads pairs)
    FV_TRY(vals_c.alloc(ctx, (size_t)*nnz_c + 2));
    FV_HIP(ctx, hipMemsetAsync(colind_c.p + *nnz_c, 0, 2 * sizeof(int32_t), ctx->stream));
    FV_HIP(ctx, hipMemsetAsync(vals_c.p + *nnz_c, 0, 2 * sizeof(double), ctx->stream));
    FV_TRY(rowcnt.alloc(ctx, (size_t)nc + 1));
    FV_TRY(rowcnt.zero(ctx));
    hipLaunchKernelGGL(amg_runs_kernel, dim3(fv_blocks(nnz)), dim3(FV_BLOCK), 0, ctx->stream, nnz, (const uint64_t *)key2.p, (const double *)val2.p,
                       (const int32_t *)head.p, (const int32_t *)pos.p, unc, colind_c.p, vals_c.p, rowcnt.p);
    FV_LAUNCH_CHECK(ctx);
    FV_TRY(rowptr_c.alloc(ctx, (size_t)nc + 1));
    int64_t check = 0;
    FV_TRY(fv_exclusive_scan_i32(ctx, rowcnt.p, rowptr_c.p, nc, &check));
    if (check != *nnz_c) {
        fv_set_error(ctx, "AMG Galerkin product: row counts (%lld) and entries (%lld) disagree", (long long)check, (long long)*nnz_c);
        return FV_ERR_STATE;
    }
    if (D) {
        FV_TRY(Dc.alloc(ctx, (size_t)nc));
        hipLaunchKernelGGL(amg_aggD_kernel, dim3(fv_blocks(nc)), dim3(FV_BLOCK), 0, ctx->stream, nc, memptr, mem, D, Dc.p);
        FV_LAUNCH_CHECK(ctx);
    }
    FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FV_OK;
}

static bool amg_verbose() { return getenv("FV_AMG_VERBOSE") != nullptr; }
static double amg_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// Row blocks of a distributed operator: the hierarchy is built on the rank's own diagonal block (the entries whose column is
// one of its rows) — block-Jacobi with the V-cycle as the block solver, no communication inside the preconditioner; the
// couplings across ranks stay with the PCG's SpMV around it.
__global__ __launch_bounds__(FV_BLOCK) void amg_local_count_kernel(int64_t n, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ colind,
                                                                    int32_t *__restrict__ cnt)
{
    const int64_t r = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (r >= n)
        return;
    int32_t c = 0;
    for (int32_t k = rowptr[r]; k < rowptr[r + 1]; k++)
        c += colind[k] < (int32_t)n;
    cnt[r] = c;
}

__global__ __launch_bounds__(FV_BLOCK) void amg_local_fill_kernel(int64_t n, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ colind,
                                                                   const double *__restrict__ vals, const int32_t *__restrict__ start,
                                                                   int32_t *__restrict__ ci, double *__restrict__ va)
{
    const int64_t r = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (r >= n)
        return;
    int32_t o = start[r];
    for (int32_t k = rowptr[r]; k < rowptr[r + 1]; k++)
        if (colind[k] < (int32_t)n) {
            ci[o] = colind[k];
            va[o] = vals[k];
            o++;
        }
}

// ------------------------------------------------------------------ gathered coarse levels for row blocks (FV_PRECOND_AMG_GATHERED)
// Block-Jacobi AMG (above) needs no communication but its iteration count grows with the number of ranks: nothing couples the blocks
// below level 0.  Here every rank still aggregates its own rows only (no aggregate crosses a rank boundary), but the level-1 operator
// is the Galerkin product of the WHOLE operator: a rank forms its rows of it from its full rows — halo columns renamed by the
// aggregate ids their owners sent through the ordinary halo exchange —, the ranks' pieces are summed into one CSR that every rank
// holds (all-reduces of disjoint segments), and the hierarchy below is built from it by every rank for itself: identical everywhere,
// no communication below level 1.  Per cycle: two halo exchanges (the two level-0 products are those of the whole operator) and one
// all-reduce of the level-1 right-hand side (each rank restricts its own rows; the other rows' sums are zero).  The coarse work is
// replicated, not divided — what this buys is an iteration count that no longer depends on the rank count.
__global__ __launch_bounds__(FV_BLOCK) void amg_global_ids_kernel(int64_t n, int64_t next, const int32_t *__restrict__ agg, double off, double *__restrict__ g)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i < next)
        g[i] = (i < n && agg[i] >= 0) ? off + (double)agg[i] : -1.0;
}

__global__ __launch_bounds__(FV_BLOCK) void amg_to_int_kernel(int64_t m, const double *__restrict__ g, int32_t *__restrict__ a)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i < m)
        a[i] = (int32_t)g[i];
}

__global__ __launch_bounds__(FV_BLOCK) void amg_row_counts_kernel(int64_t nc, const int32_t *__restrict__ rowptr, double *__restrict__ cnt)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i < nc)
        cnt[i] = (double)(rowptr[i + 1] - rowptr[i]);
}

// a rank's rows [lo, hi) of the level-1 operator into their places of the gathered arrays (columns as doubles: they travel through
// the same all-reduce as the values)
__global__ __launch_bounds__(FV_BLOCK) void amg_scatter_piece_kernel(int64_t lo, int64_t hi, const int32_t *__restrict__ prp, const int32_t *__restrict__ pci,
                                                                      const double *__restrict__ pva, const int32_t *__restrict__ grp,
                                                                      double *__restrict__ gc, double *__restrict__ gv)
{
    const int64_t I = lo + (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (I >= hi)
        return;
    const int32_t s = prp[I], e = prp[I + 1], o = grp[I];
    for (int32_t k = s; k < e; k++) {
        gc[o + (k - s)] = (double)pci[k];
        gv[o + (k - s)] = pva[k];
    }
}

// Every rank's verdict on a stretch of local work (allocations, scans) before the next collective: a rank that failed alone would return while
// the others wait in that collective for ever.  One double through the all-reduce the ranks share; the failing rank keeps its own
// message, the others learn that somebody failed.
static int amg_agree(fv_problem *p, int rc)
{
    fv_ctx *ctx = p->ctx;
    fv_dist *d = p->dist;
    const double flag = rc != FV_OK ? 1.0 : 0.0;
    double total = 0.0;
    int rc2 = FV_OK;
    if (hipMemcpyAsync(d->red.p, &flag, sizeof flag, hipMemcpyHostToDevice, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess)
        rc2 = FV_ERR_HIP;
    if (rc2 == FV_OK)
        rc2 = fv_comm_allreduce_sum(ctx, d, d->red.p, 1, ctx->stream);
    if (rc2 == FV_OK && fv_memcpy_sync(ctx, &total, d->red.p, sizeof total, hipMemcpyDeviceToHost) != hipSuccess)
        rc2 = FV_ERR_HIP;
    if (rc != FV_OK)
        return rc;
    if (rc2 != FV_OK)
        return rc2;
    if (total > 0.0) {
        fv_set_error(ctx, "gathered AMG level: the set-up failed on another rank");
        return FV_ERR_STATE;
    }
    return FV_OK;
}

// agg_loc: the rank's own aggregation of its n rows (local ids 0 .. nc_loc - 1, -1 = none).  On return L carries the transfer to the
// gathered level (global ids) and C is that level.
static int amg_gather_level1(fv_problem *p, AmgLevel *L, DevBuf<int32_t> &agg_loc, int64_t nc_loc, AmgLevel *C)
{
    fv_ctx *ctx = p->ctx;
    fv_dist *d = p->dist;
    const int64_t n = p->n, next = p->n + p->nhalo;
    const int R = d->nranks;
    // every rank's number of aggregates -> my offset, the size of the level
    DevBuf<double> counts;
    auto phase0 = [&]() -> int {
        FV_TRY(counts.alloc(ctx, (size_t)R));
        FV_TRY(counts.zero(ctx));
        const double mine = (double)nc_loc;
        FV_HIP(ctx, hipMemcpyAsync(counts.p + d->rank, &mine, sizeof mine, hipMemcpyHostToDevice, ctx->stream));
        FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return FV_OK;
    };
    FV_TRY(amg_agree(p, phase0()));
    FV_TRY(fv_comm_allreduce_sum(ctx, d, counts.p, R, ctx->stream));
    std::vector<double> hc((size_t)R);
    FV_HIP(ctx, fv_memcpy_sync(ctx, hc.data(), counts.p, (size_t)R * sizeof(double), hipMemcpyDeviceToHost));
    int64_t off = 0, ncg = 0;
    for (int r = 0; r < R; r++) {
        if (r < d->rank)
            off += (int64_t)hc[(size_t)r];
        ncg += (int64_t)hc[(size_t)r];
    }
    if (ncg <= 0 || ncg >= 0x7fffffffLL) { // (the same sum on every rank: all return together)
        fv_set_error(ctx, "gathered AMG level: %lld aggregates over all ranks", (long long)ncg);
        return FV_ERR_STATE;
    }
    // global aggregate ids of my rows and, through the halo exchange, of my halo columns
    DevBuf<double> gid;
    DevBuf<int32_t> agg_ext;
    auto phase1 = [&]() -> int {
        FV_TRY(gid.alloc(ctx, (size_t)next + FV_VEC_PAD));
        FV_TRY(agg_ext.alloc(ctx, (size_t)next));
        hipLaunchKernelGGL(amg_global_ids_kernel, dim3(fv_blocks(next)), dim3(FV_BLOCK), 0, ctx->stream, n, next, (const int32_t *)agg_loc.p, (double)off, gid.p);
        FV_LAUNCH_CHECK(ctx);
        return FV_OK;
    };
    FV_TRY(amg_agree(p, phase1()));
    FV_TRY(fv_dist_exchange(p, gid.p));
    // my rows of P^T A P, in the row space of the whole level (the rows of other ranks stay empty); row lengths of the whole level
    DevBuf<int32_t> prp, pci;
    DevBuf<double> pva, pD;
    int64_t pnnz = 0;
    DevBuf<double> dcnt;
    DevBuf<int32_t> icnt;
    auto phase2 = [&]() -> int {
        hipLaunchKernelGGL(amg_to_int_kernel, dim3(fv_blocks(next)), dim3(FV_BLOCK), 0, ctx->stream, next, (const double *)gid.p, agg_ext.p);
        FV_LAUNCH_CHECK(ctx);
        gid.release();
        FV_TRY(amg_members(ctx, n, agg_ext.p, ncg, L->memptr, L->mem));
        FV_TRY(amg_galerkin(ctx, n, p->nnz, p->rowptr.p, p->colind.p, p->vals.p, L->D, agg_ext.p, ncg, L->memptr.p, L->mem.p, prp, pci, pva, pD, &pnnz, next));
        FV_TRY(dcnt.alloc(ctx, (size_t)ncg));
        FV_TRY(icnt.alloc(ctx, (size_t)ncg));
        hipLaunchKernelGGL(amg_row_counts_kernel, dim3(fv_blocks(ncg)), dim3(FV_BLOCK), 0, ctx->stream, ncg, (const int32_t *)prp.p, dcnt.p);
        FV_LAUNCH_CHECK(ctx);
        return FV_OK;
    };
    FV_TRY(amg_agree(p, phase2()));
    FV_TRY(fv_comm_allreduce_sum(ctx, d, dcnt.p, (int)ncg, ctx->stream));
    int64_t gnnz = 0;
    DevBuf<double> gc;
    auto phase3 = [&]() -> int {
        hipLaunchKernelGGL(amg_to_int_kernel, dim3(fv_blocks(ncg)), dim3(FV_BLOCK), 0, ctx->stream, ncg, (const double *)dcnt.p, icnt.p);
        FV_LAUNCH_CHECK(ctx);
        FV_TRY(C->o_rowptr.alloc(ctx, (size_t)ncg + 1));
        FV_TRY(fv_exclusive_scan_i32(ctx, icnt.p, C->o_rowptr.p, ncg, &gnnz));
        if (gnnz <= 0 || gnnz >= 0x7fffffffLL - 2) {
            fv_set_error(ctx, "gathered AMG level: %lld entries", (long long)gnnz);
            return FV_ERR_STATE;
        }
        dcnt.release();
        icnt.release();
        // entries: every rank writes its rows' segments, the sum over the ranks is the level
        FV_TRY(gc.alloc(ctx, (size_t)gnnz));
        FV_TRY(gc.zero(ctx));
        FV_TRY(C->o_vals.alloc(ctx, (size_t)gnnz + 2));
        FV_TRY(C->o_vals.zero(ctx));
        FV_TRY(C->o_colind.alloc(ctx, (size_t)gnnz + 2));
        FV_TRY(C->o_colind.zero(ctx));
        if (nc_loc > 0)
            hipLaunchKernelGGL(amg_scatter_piece_kernel, dim3(fv_blocks(nc_loc)), dim3(FV_BLOCK), 0, ctx->stream, off, off + nc_loc, (const int32_t *)prp.p,
                               (const int32_t *)pci.p, (const double *)pva.p, (const int32_t *)C->o_rowptr.p, gc.p, C->o_vals.p);
        FV_LAUNCH_CHECK(ctx);
        return FV_OK;
    };
    FV_TRY(amg_agree(p, phase3()));
    FV_TRY(fv_comm_allreduce_sum(ctx, d, gc.p, (int)gnnz, ctx->stream));
    FV_TRY(fv_comm_allreduce_sum(ctx, d, C->o_vals.p, (int)gnnz, ctx->stream));
    hipLaunchKernelGGL(amg_to_int_kernel, dim3(fv_blocks(gnnz)), dim3(FV_BLOCK), 0, ctx->stream, gnnz, (const double *)gc.p, C->o_colind.p);
    FV_LAUNCH_CHECK(ctx);
    if (L->D) {
        C->o_D.swap(pD); // (sums over my members; zero on the rows of other ranks)
        FV_TRY(fv_comm_allreduce_sum(ctx, d, C->o_D.p, (int)ncg, ctx->stream));
    }
    FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    C->n = ncg;
    C->nnz = gnnz;
    C->rowptr = C->o_rowptr.p;
    C->colind = C->o_colind.p;
    C->vals = C->o_vals.p;
    C->D = L->D ? C->o_D.p : nullptr;
    L->nc = ncg;
    L->agg.swap(agg_ext);
    if (amg_verbose())
        fprintf(stderr, "[amg] rank %d: gathered level 1: %lld rows (mine %lld from %lld), %lld entries\n", d->rank, (long long)ncg, (long long)nc_loc,
                (long long)off, (long long)gnnz);
    return FV_OK;
}

static int amg_build_pooled(fv_problem *p)
{
    fv_ctx *ctx = p->ctx;
    const double t_start = amg_now();
    if (p->nhalo && !p->dist) {
        fv_set_error(ctx, "the AMG preconditioner needs a whole operator or a row block set up by fv_dist_setup");
        return FV_ERR_STATE;
    }
    fv_amg_free(p->amg);
    p->amg = new fv_amg();
    fv_amg *a = p->amg;
    AmgLevel *L = new AmgLevel();
    a->lev.push_back(L);
    L->n = p->n;
    if (p->dist) {
        DevBuf<int32_t> cnt;
        FV_TRY(cnt.alloc(ctx, (size_t)p->n));
        FV_TRY(a->loc_rowptr.alloc(ctx, (size_t)p->n + 1));
        hipLaunchKernelGGL(amg_local_count_kernel, dim3(fv_blocks(p->n)), dim3(FV_BLOCK), 0, ctx->stream, p->n, (const int32_t *)p->rowptr.p,
                           (const int32_t *)p->colind.p, cnt.p);
        FV_LAUNCH_CHECK(ctx);
        int64_t nloc = 0;
        FV_TRY(fv_exclusive_scan_i32(ctx, cnt.p, a->loc_rowptr.p, p->n, &nloc));
        FV_TRY(a->loc_colind.alloc(ctx, (size_t)nloc + 2));
        FV_TRY(a->loc_vals.alloc(ctx, (size_t)nloc + 2));
        hipLaunchKernelGGL(amg_local_fill_kernel, dim3(fv_blocks(p->n)), dim3(FV_BLOCK), 0, ctx->stream, p->n, (const int32_t *)p->rowptr.p,
                           (const int32_t *)p->colind.p, (const double *)p->vals.p, (const int32_t *)a->loc_rowptr.p, a->loc_colind.p, a->loc_vals.p);
        FV_LAUNCH_CHECK(ctx);
        L->nnz = nloc;
        L->rowptr = a->loc_rowptr.p;
        L->colind = a->loc_colind.p;
        L->vals = a->loc_vals.p;
    } else if (p->lean) { // level 0's structure for the set-up alone (matching, Galerkin products, diagonal): written out from the rows, given back below
        FV_TRY(fv_lean_csr32(p, a->loc_rowptr, a->loc_colind, a->loc_vals));
        L->nnz = p->nnz;
        L->rowptr = a->loc_rowptr.p;
        L->colind = a->loc_colind.p;
        L->vals = a->loc_vals.p;
    } else {
        L->nnz = p->nnz;
        L->rowptr = p->rowptr.p;
        L->colind = p->colind.p;
        L->vals = p->vals.p;
    }
    L->D = p->D.p; // may be null (steady solve before fv_transient_begin)
    a->gathered = p->dist && p->dist->nranks > 1 && p->amg_gathered;
    for (int depth = 0; depth < 24; depth++) {
        L = a->lev.back();
        // (gathered levels: whether there is a level 1 is not for one rank's block size to decide — every rank builds it)
        const bool gather_here = a->gathered && depth == 0;
        if (L->n <= g_coarse_max && !gather_here)
            break;
        // passes of pairwise aggregation, each on the Galerkin operator of the previous one
        DevBuf<int32_t> cur_rowptr, cur_colind, agg_total;
        DevBuf<double> cur_vals, cur_D;
        const int32_t *rp = L->rowptr, *ci = L->colind;
        const double *va = L->vals, *Dp = L->D;
        int64_t ncur = L->n, nnz_cur = L->nnz;
        bool stalled = false;
        // coarse levels of fewer than a million rows are launch-bound: one pass more there (aggregates of ~8) makes the hierarchy a
        // level shorter at the same iteration count (256^3 sigma = 3: 7 -> 6 levels, 37 iterations either way, solve 96.1 -> 92.7 ms)
        const int npasses = (a->lev.size() > 1 && L->n < 1000000 && g_passes < 4) ? g_passes + 1 : g_passes; // (never the operator itself)
        for (int pass = 0; pass < npasses; pass++) {
            DevBuf<int32_t> agg, memptr, mem, nrp, nci;
            DevBuf<double> nva, nD;
            int64_t nc = 0, nnzc = 0;
            const double tp0 = amg_now();
            FV_TRY(amg_pairwise(ctx, ncur, rp, ci, va, agg, &nc));
            if (amg_verbose())
                fprintf(stderr, "[amg] level %zu pass %d: %lld rows -> %lld aggregates (matching %.3f s)\n", a->lev.size() - 1, pass, (long long)ncur,
                        (long long)nc, amg_now() - tp0);
            if (gather_here && pass == 0 && (nc == 0 || ncur <= 1)) { // (a block without couplings: no aggregates of its own, the others' level all the same)
                agg_total.swap(agg);
                ncur = nc;
                break;
            }
            if ((nc == 0 || nc > (int64_t)(0.95 * (double)ncur)) && !(gather_here && pass == 0)) {
                stalled = pass == 0;
                break;
            }
            FV_TRY(amg_members(ctx, ncur, agg.p, nc, memptr, mem));
            const double tg0 = amg_now();
            FV_TRY(amg_galerkin(ctx, ncur, nnz_cur, rp, ci, va, Dp, agg.p, nc, memptr.p, mem.p, nrp, nci, nva, nD, &nnzc));
            if (amg_verbose())
                fprintf(stderr, "[amg]   Galerkin: nnz %lld -> %lld (%.3f s)\n", (long long)nnz_cur, (long long)nnzc, amg_now() - tg0);
            if (pass == 0) {
                agg_total.swap(agg);
            } else {
                hipLaunchKernelGGL(amg_compose_kernel, dim3(fv_blocks(L->n)), dim3(FV_BLOCK), 0, ctx->stream, L->n, agg_total.p,
                                   (const int32_t *)agg.p);
                FV_LAUNCH_CHECK(ctx);
                FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
            }
            cur_rowptr.swap(nrp);
            cur_colind.swap(nci);
            cur_vals.swap(nva);
            cur_D.swap(nD);
            rp = cur_rowptr.p;
            ci = cur_colind.p;
            va = cur_vals.p;
            Dp = L->D ? cur_D.p : nullptr;
            ncur = nc;
            nnz_cur = nnzc;
            if (ncur <= g_coarse_max)
                break;
        }
        if (gather_here) {
            AmgLevel *C = new AmgLevel();
            a->lev.push_back(C); // (owned by the hierarchy whatever happens next)
            FV_TRY(amg_gather_level1(p, L, agg_total, ncur, C));
            FV_TRY(C->x.alloc(ctx, (size_t)C->n));
            FV_TRY(C->b.alloc(ctx, (size_t)C->n));
            FV_TRY(C->u.alloc(ctx, (size_t)C->n));
            continue;
        }
        if (stalled || ncur == L->n || !agg_total.p)
            break; // cannot coarsen further: this level is the coarsest
        AmgLevel *C = new AmgLevel();
        C->n = ncur;
        C->nnz = nnz_cur;
        C->o_rowptr.swap(cur_rowptr);
        C->o_colind.swap(cur_colind);
        C->o_vals.swap(cur_vals);
        C->o_D.swap(cur_D);
        C->rowptr = C->o_rowptr.p;
        C->colind = C->o_colind.p;
        C->vals = C->o_vals.p;
        C->D = L->D ? C->o_D.p : nullptr;
        L->nc = ncur;
        L->agg.swap(agg_total);
        FV_TRY(amg_members(ctx, L->n, L->agg.p, L->nc, L->memptr, L->mem));
        FV_TRY(C->x.alloc(ctx, (size_t)C->n));
        FV_TRY(C->b.alloc(ctx, (size_t)C->n));
        FV_TRY(C->u.alloc(ctx, (size_t)C->n));
        a->lev.push_back(C);
    }
    for (AmgLevel *l : a->lev) {
        FV_TRY(l->diag.alloc(ctx, (size_t)l->n));
        FV_TRY(l->dinv.alloc(ctx, (size_t)l->n));
        FV_TRY(l->t.alloc(ctx, (size_t)l->n + FV_VEC_PAD));
        hipLaunchKernelGGL(amg_diag_kernel, dim3(fv_blocks(l->n)), dim3(FV_BLOCK), 0, ctx->stream, l->n, l->rowptr, l->colind, l->vals, l->diag.p);
        FV_LAUNCH_CHECK(ctx);
    }
    if (p->lean && a->lev.size() > 1) { // the cycles run level 0 through the problem's own product: its CSR has done its work
        a->loc_rowptr.release();
        a->loc_colind.release();
        a->loc_vals.release();
        a->lev[0]->rowptr = a->lev[0]->colind = nullptr;
        a->lev[0]->vals = nullptr;
    }
    AmgLevel *last = a->lev.back();
    a->nco = last->n;
    a->dense = last->n <= 2 * (int64_t)g_coarse_max && last->n > 0; // also a tiny problem as a whole: the "cycle" is then the exact inverse
    if (a->dense) {
        FV_TRY(a->inv.alloc(ctx, (size_t)(a->nco * a->nco)));
        FV_TRY(a->inv2.alloc(ctx, (size_t)(a->nco * a->nco)));
    }
    FV_TRY(a->z.alloc(ctx, (size_t)p->n + (size_t)p->nhalo + FV_VEC_PAD)); // (a row block's halo slots stay zero: the cycle is block-local)
    FV_TRY(a->z.zero(ctx));
    FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    a->epoch = p->assemble_epoch;
    a->storage_epoch = p->storage_epoch;
    a->sigma = NAN;
    if (amg_verbose())
        fprintf(stderr, "[amg] hierarchy of %zu levels, coarsest %lld rows (%s), %.3f s\n", a->lev.size(), (long long)a->nco,
                a->dense ? "dense inverse" : "Jacobi sweeps", amg_now() - t_start);
    return FV_OK;
}

// the set-up's scratch arrays (sort buffers, flags, scans: hundreds of them, shrinking level by level) come out of a pool
static int amg_build(fv_problem *p)
{
    fv_pool_begin();
    const int rc = amg_build_pooled(p);
    fv_pool_end();
    return rc;
}

// Jacobi diagonals of every level and the coarsest inverse for this sigma
static int amg_set_sigma(fv_problem *p, double sigma)
{
    fv_ctx *ctx = p->ctx;
    fv_amg *a = p->amg;
    if (a->sigma == sigma)
        return FV_OK;
    const double t_sig = amg_now();
    if (sigma != 0.0 && !a->lev[0]->D) {
        fv_set_error(ctx, "AMG hierarchy was built without the storage term; call fv_transient_begin before selecting it for shifted solves");
        return FV_ERR_STATE;
    }
    for (AmgLevel *l : a->lev) {
        hipLaunchKernelGGL(amg_dinv_kernel, dim3(fv_blocks(l->n)), dim3(FV_BLOCK), 0, ctx->stream, l->n, (const double *)l->diag.p,
                           sigma != 0.0 ? l->D : (const double *)nullptr, sigma, l->dinv.p);
        FV_LAUNCH_CHECK(ctx);
    }
    if (a->dense) {
        const AmgLevel *l = a->lev.back();
        const int64_t m = a->nco;
        FV_HIP(ctx, hipMemsetAsync(a->inv.p, 0, (size_t)(m * m) * sizeof(double), ctx->stream));
        hipLaunchKernelGGL(amg_dense_fill_kernel, dim3(fv_blocks(m)), dim3(FV_BLOCK), 0, ctx->stream, m, l->rowptr, l->colind, l->vals,
                           sigma != 0.0 ? l->D : (const double *)nullptr, sigma, a->inv.p);
        FV_LAUNCH_CHECK(ctx);
        const dim3 g2(fv_blocks(m), (unsigned)m);
        double *src = a->inv.p, *dst = a->inv2.p;
        for (int64_t k = 0; k < m; k += GJB) {
            hipLaunchKernelGGL(amg_gj_block_kernel, g2, dim3(FV_BLOCK), 0, ctx->stream, m, k, (int)(m - k < GJB ? m - k : GJB), (const double *)src, dst);
            std::swap(src, dst);
        }
        FV_LAUNCH_CHECK(ctx);
        if (src != a->inv.p) // (an odd number of steps: the inverse sits in the second copy)
            a->inv.swap(a->inv2);
    }
    FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    a->sigma = sigma;
    if (amg_verbose())
        fprintf(stderr, "[amg] diagonals + coarsest inverse for sigma = %g: %.3f s\n", sigma, amg_now() - t_sig);
    return FV_OK;
}

int fv_amg_prepare(fv_problem *p, double sigma)
{
    if (!p->amg || p->amg->epoch != p->assemble_epoch || p->amg->storage_epoch != p->storage_epoch)
        FV_TRY(amg_build(p));
    return amg_set_sigma(p, sigma);
}

// level 0 uses the problem's own SpMV; on a row block the product with the rank's diagonal block (x's halo slots are zero)
static int amg_top_spmv(fv_problem *p, const double *x, double *t, double sigma)
{
    if (p->dist && p->amg->gathered) // the whole operator's rows: x's halo slots are fetched from their owners
        return fv_dist_full_spmv(p, const_cast<double *>(x), t, sigma, p->amg->fold);
    if (p->dist)
        return fv_dist_local_spmv(p, const_cast<double *>(x), t, sigma, p->amg->fold);
    return fv_spmv_launch(p, x, t, sigma, nullptr, p->amg->fold);
}

static int amg_cycle(fv_problem *p, size_t l, const double *b, double *x, double sigma, double *dot_part, const double *dot_q, double *dot_part_q,
                     bool first_ready);

// x_l ~ A_l^-1 b_l by two flexible-CG steps preconditioned by the cycle of level l (l >= 1, not the coarsest).  Whoever formed b has
// left omega dinv b in the level's u (the restriction above it); the inner products ride on the two products with A_l.
static int amg_kcycle(fv_problem *p, size_t l, const double *b, double *x, double sigma)
{
    fv_ctx *ctx = p->ctx;
    AmgLevel *L = p->amg->lev[l];
    if (!L->kpart.p) {
        FV_TRY(L->kv1.alloc(ctx, (size_t)L->n));
        FV_TRY(L->kr1.alloc(ctx, (size_t)L->n));
        FV_TRY(L->kc2.alloc(ctx, (size_t)L->n));
        FV_TRY(L->kv2.alloc(ctx, (size_t)L->n));
        FV_TRY(L->kpart.alloc(ctx, (size_t)5 * AMG_KG));
    }
    const unsigned nb = fv_blocks(L->n);
    const dim3 g(nb < 256u ? nb : 256u), blk(FV_BLOCK);
    int G = 0;
    FV_TRY(amg_cycle(p, l, b, x, sigma, nullptr, nullptr, nullptr, true)); // c1
    AmgOp op{};
    op.x = x;
    op.y = L->kv1.p;
    op.b = b;
    op.part = L->kpart.p;
    FV_TRY(amg_level_op(ctx, L, AMG_KDOT2, op, sigma, &G)); // v1 = A c1, rho1 = c1.v1, alpha1 = c1.b
    hipLaunchKernelGGL(amg_kres_kernel, g, blk, 0, ctx->stream, L->n, b, (const double *)L->kv1.p, (const double *)L->kpart.p, G, L->kr1.p,
                       (const double *)L->dinv.p, g_omega, L->u.p);
    FV_LAUNCH_CHECK(ctx);
    FV_TRY(amg_cycle(p, l, L->kr1.p, L->kc2.p, sigma, nullptr, nullptr, nullptr, true)); // c2
    op = AmgOp{};
    op.x = L->kc2.p;
    op.y = L->kv2.p;
    op.v1 = L->kv1.p;
    op.r1 = L->kr1.p;
    op.part = L->kpart.p;
    FV_TRY(amg_level_op(ctx, L, AMG_KDOT3, op, sigma, &G)); // v2 = A c2, gamma = c2.v1, beta = c2.v2, alpha2 = c2.r1
    hipLaunchKernelGGL(amg_kcomb_kernel, g, blk, 0, ctx->stream, L->n, (const double *)L->kc2.p, (const double *)L->kpart.p, G, x);
    FV_LAUNCH_CHECK(ctx);
    return FV_OK;
}

// x_l = V(b_l) on level l (x, b: the level's vectors; level 0: the caller's)
// dot_part (top level only): per-block partials of b.x, i.e. the PCG's r.z, written by the last smoothing pass
// dot_q / dot_part_q (with dot_part): also the partials of x.dot_q (flexible PCG: z.q)
// first_ready: the first smoothing pass from a zero iterate, omega dinv b, is already there — in x on level 0 (left by the PCG's vector
// update), in the level's u below (left by the restriction or the K-cycle's residual kernel)
static int amg_cycle(fv_problem *p, size_t l, const double *b, double *x, double sigma, double *dot_part = nullptr, const double *dot_q = nullptr,
                     double *dot_part_q = nullptr, bool first_ready = false)
{
    fv_ctx *ctx = p->ctx;
    fv_amg *a = p->amg;
    AmgLevel *L = a->lev[l];
    const dim3 g(fv_blocks(L->n)), blk(FV_BLOCK);
    if (l + 1 == a->lev.size()) { // coarsest
        if (a->dense) {
            hipLaunchKernelGGL(amg_gemv_kernel, dim3(fv_blocks(L->n, FV_BLOCK / 64)), blk, 0, ctx->stream, L->n, (const double *)a->inv.p, b, x);
            FV_LAUNCH_CHECK(ctx);
            return FV_OK;
        }
        // no dense inverse (single level, or coarsening stalled high): symmetric Jacobi sweeps
        hipLaunchKernelGGL(amg_smooth0_kernel, g, blk, 0, ctx->stream, L->n, (const double *)L->dinv.p, b, g_omega, x);
        for (int s = 1; s < g_coarse_sweeps; s++) {
            if (l == 0)
                FV_TRY(amg_top_spmv(p, x, L->t.p, sigma));
            else
                FV_TRY(amg_level_spmv(ctx, L, x, L->t.p, sigma));
            hipLaunchKernelGGL(amg_smooth_kernel, g, blk, 0, ctx->stream, L->n, (const double *)L->dinv.p, b, (const double *)L->t.p, g_omega, x);
        }
        FV_LAUNCH_CHECK(ctx);
        return FV_OK;
    }
    AmgLevel *C = a->lev[l + 1];
    const bool c_last = l + 2 == a->lev.size();
    const bool c_kcycle = a->kcycle && (int)(l + 1) <= amg_kcycle_levels() && !c_last;
    double *u = l == 0 ? x : L->u.p; // the iterate before the last smoothing pass
    if (!first_ready) {
        hipLaunchKernelGGL(amg_smooth0_kernel, g, blk, 0, ctx->stream, L->n, (const double *)L->dinv.p, b, g_omega, u);
        FV_LAUNCH_CHECK(ctx);
    }
    if (l == 0)
        FV_TRY(amg_top_spmv(p, u, L->t.p, sigma));
    else
        FV_TRY(amg_level_spmv(ctx, L, u, L->t.p, sigma));
    // (4 lanes per aggregate: the aggregates of two pairwise passes have ~5 members)
    const bool sum_ranks = l == 0 && a->gathered; // every rank has restricted its own rows: the level's right-hand side is their sum
    hipLaunchKernelGGL(amg_restrict_kernel<4>, dim3(fv_blocks(C->n, FV_BLOCK / 4)), blk, 0, ctx->stream, C->n, (const int32_t *)L->memptr.p,
                       (const int32_t *)L->mem.p, b, (const double *)L->t.p, C->b.p, (const double *)C->dinv.p, g_omega,
                       (c_last || sum_ranks) ? (double *)nullptr : C->u.p);
    FV_LAUNCH_CHECK(ctx);
    if (sum_ranks) {
        FV_TRY(fv_comm_allreduce_sum(ctx, p->dist, C->b.p, (int)C->n, ctx->stream));
        if (!c_last) {
            hipLaunchKernelGGL(amg_smooth0_kernel, dim3(fv_blocks(C->n)), blk, 0, ctx->stream, C->n, (const double *)C->dinv.p, (const double *)C->b.p,
                               g_omega, C->u.p);
            FV_LAUNCH_CHECK(ctx);
        }
    }
    if (c_kcycle)
        FV_TRY(amg_kcycle(p, l + 1, C->b.p, C->x.p, sigma));
    else
        FV_TRY(amg_cycle(p, l + 1, C->b.p, C->x.p, sigma, nullptr, nullptr, nullptr, !c_last));
    hipLaunchKernelGGL(amg_prolong_kernel, g, blk, 0, ctx->stream, L->n, (const int32_t *)L->agg.p, (const double *)C->x.p, u);
    FV_LAUNCH_CHECK(ctx);
    if (l > 0) { // x = u + omega dinv (b - A u) in the product's own launch
        AmgOp op{};
        op.x = u;
        op.b = b;
        op.dinv = L->dinv.p;
        op.omega = g_omega;
        op.xout = x;
        return amg_level_op(ctx, L, AMG_SMOOTH, op, sigma);
    }
    FV_TRY(amg_top_spmv(p, x, L->t.p, sigma));
    if (dot_part)
        hipLaunchKernelGGL(amg_smooth_dot_kernel, dim3(vec_grid(L->n)), blk, 0, ctx->stream, L->n, (const double *)L->dinv.p, b,
                           (const double *)L->t.p, g_omega, x, dot_part, dot_q, dot_part_q);
    else
        hipLaunchKernelGGL(amg_smooth_kernel, g, blk, 0, ctx->stream, L->n, (const double *)L->dinv.p, b, (const double *)L->t.p, g_omega, x);
    FV_LAUNCH_CHECK(ctx);
    return FV_OK;
}

// kcycle: the caller's PCG is flexible (row blocks: dist_amg_loop); fv_amg_apply keeps the fixed linear V-cycle
bool fv_amg_kcycle_available(fv_problem *p) { return amg_kcycle_levels() > 0; }
int fv_amg_apply_device(fv_problem *p, const double *r, double *z, double sigma, bool kcycle)
{
    FV_TRY(fv_amg_prepare(p, sigma));
    p->amg->fold = false;
    p->amg->kcycle = kcycle && amg_kcycle_levels() > 0 && p->amg->lev.size() > 2;
    return amg_cycle(p, 0, r, z, sigma);
}

// ------------------------------------------------------------------ PCG driven by the V-cycle
// The set-up (r0, ||rhs||^2, ||r0||^2, tol2, done) has been left in the workspace by fv_pcg_solve's own set-up path.
__global__ __launch_bounds__(FV_BLOCK) void amg_dot_kernel(int64_t n, const double *__restrict__ a, const double *__restrict__ b,
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

// first direction: rz0 = r.z, p = z
__global__ __launch_bounds__(FV_BLOCK) void amg_pcg_start_kernel(int64_t n, const double *__restrict__ z, double *__restrict__ pv,
                                                                  const double *__restrict__ part_rz, int nparts, PcgScalars *__restrict__ scal)
{
    __shared__ double smem[4];
    const double rz = reduce_partials(part_rz, nparts, smem);
    for (int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x; i < n; i += vec_stride())
        pv[i] = z[i];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        scal->rz[0] = rz;
        scal->iters = 0;
    }
}

// x += alpha p ; r -= alpha q ; partial r.r
__global__ __launch_bounds__(FV_BLOCK) void amg_pcg_update_kernel(int64_t n, int it, double *__restrict__ x, double *__restrict__ r,
                                                                   const double *__restrict__ pv, const double *__restrict__ q,
                                                                   const double *__restrict__ part_pq, int npq, PcgScalars *__restrict__ scal,
                                                                   double *__restrict__ part_rr, const double *__restrict__ dinv, double omega,
                                                                   double *__restrict__ z0)
{
    __shared__ double smem[4];
    const double pq = reduce_partials(part_pq, npq, smem);
    if (!(pq > 0.0)) {
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            scal->pq = pq;
            scal->done = 2;
        }
        return;
    }
    const double alpha = scal->rz[it & 1] / pq;
    double arr = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x; i < n; i += vec_stride()) {
        const double ri = r[i] - alpha * q[i];
        x[i] += alpha * pv[i];
        r[i] = ri;
        if (z0) // the cycle's first smoothing pass from a zero iterate, while the new residual is in a register
            z0[i] = omega * dinv[i] * ri;
        arr += ri * ri;
    }
    const double t = block_sum(arr, smem);
    if (threadIdx.x == 0) {
        part_rr[blockIdx.x] = t;
        if (blockIdx.x == 0)
            scal->pq = pq;
    }
}

__global__ __launch_bounds__(FV_BLOCK) void amg_pcg_check_kernel(int it, const double *__restrict__ part_rr, int nparts, PcgScalars *__restrict__ scal,
                                                                  double *__restrict__ hist, int64_t hist_cap)
{
    __shared__ double smem[4];
    if (scal->done)
        return;
    const double rr = reduce_partials(part_rr, nparts, smem);
    if (threadIdx.x == 0) {
        scal->rr = rr;
        scal->iters = it + 1;
        if (hist && it < hist_cap)
            hist[it] = sqrt(rr);
        if (rr <= scal->tol2)
            scal->done = 1;
    }
}

// beta = r.z / (r.z)_old ; p = z + beta p.  part_zq (flexible PCG, for a preconditioner that is not a fixed linear operator):
// beta = z.(r - r_old) / (r.z)_old with r_old = r + alpha q, i.e. -alpha z.q / (r.z)_old — the same number when z = M r exactly.
__global__ __launch_bounds__(FV_BLOCK) void amg_pcg_direction_kernel(int64_t n, int it, const double *__restrict__ z, double *__restrict__ pv,
                                                                      const double *__restrict__ part_rz, int nparts,
                                                                      PcgScalars *__restrict__ scal, const double *__restrict__ part_zq)
{
    __shared__ double smem[4];
    const double rzn = reduce_partials(part_rz, nparts, smem);
    double beta = rzn / scal->rz[it & 1];
    if (part_zq) {
        const double zq = reduce_partials(part_zq, nparts, smem);
        beta = -zq / scal->pq; // (alpha = (r.z)_old / p.q)
    }
    for (int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x; i < n; i += vec_stride())
        pv[i] = z[i] + beta * pv[i];
    if (blockIdx.x == 0 && threadIdx.x == 0)
        scal->rz[(it + 1) & 1] = rzn;
}

int fv_amg_pcg_loop(fv_problem *p, double *x, double sigma, bool fold, int64_t maxiter, PcgScalars *hs)
{
    fv_ctx *ctx = p->ctx;
    const int64_t n = p->n;
    const int Gv = vec_grid(n);
    FV_TRY(fv_amg_prepare(p, sigma));
    fv_amg *a = p->amg;
    a->fold = fold;
    FV_HIP(ctx, hipMemcpyAsync(hs, p->scal.p, sizeof(PcgScalars), hipMemcpyDeviceToHost, ctx->stream));
    FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (hs->done || maxiter <= 0)
        return FV_OK;
    const bool fused_dot = a->lev.size() > 1; // a single-level "hierarchy" (tiny problem) ends in the dense solve, not in a smoothing pass
    a->kcycle = amg_kcycle_levels() > 0 && a->lev.size() > 2;
    const bool flexible = a->kcycle && fused_dot;
    FV_TRY(amg_cycle(p, 0, p->r.p, a->z.p, sigma, fused_dot ? p->part_rz.p : nullptr));
    if (!fused_dot)
        hipLaunchKernelGGL(amg_dot_kernel, dim3(Gv), dim3(FV_BLOCK), 0, ctx->stream, n, (const double *)p->r.p, (const double *)a->z.p, p->part_rz.p);
    hipLaunchKernelGGL(amg_pcg_start_kernel, dim3(Gv), dim3(FV_BLOCK), 0, ctx->stream, n, (const double *)a->z.p, p->pvec.p,
                       (const double *)p->part_rz.p, Gv, p->scal.p);
    FV_LAUNCH_CHECK(ctx);
    for (int64_t it = 0; it < maxiter; it++) {
        int npq = 0;
        FV_TRY(fv_spmv_launch(p, p->pvec.p, p->q.p, sigma, p->part_pq.p, fold, &npq));
        hipLaunchKernelGGL(amg_pcg_update_kernel, dim3(Gv), dim3(FV_BLOCK), 0, ctx->stream, n, (int)it, x, p->r.p, (const double *)p->pvec.p,
                           (const double *)p->q.p, (const double *)p->part_pq.p, npq, p->scal.p, p->part_rr.p,
                           fused_dot ? (const double *)a->lev[0]->dinv.p : (const double *)nullptr, g_omega, fused_dot ? a->z.p : (double *)nullptr);
        hipLaunchKernelGGL(amg_pcg_check_kernel, dim3(1), dim3(FV_BLOCK), 0, ctx->stream, (int)it, (const double *)p->part_rr.p, Gv, p->scal.p,
                           p->hist.p, p->hist_cap);
        FV_LAUNCH_CHECK(ctx);
        FV_HIP(ctx, hipMemcpyAsync(hs, p->scal.p, sizeof(PcgScalars), hipMemcpyDeviceToHost, ctx->stream));
        FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (hs->done)
            break;
        FV_TRY(amg_cycle(p, 0, p->r.p, a->z.p, sigma, fused_dot ? p->part_rz.p : nullptr, flexible ? (const double *)p->q.p : nullptr, p->part_bb.p,
                         fused_dot));
        if (!fused_dot)
            hipLaunchKernelGGL(amg_dot_kernel, dim3(Gv), dim3(FV_BLOCK), 0, ctx->stream, n, (const double *)p->r.p, (const double *)a->z.p,
                               p->part_rz.p);
        hipLaunchKernelGGL(amg_pcg_direction_kernel, dim3(Gv), dim3(FV_BLOCK), 0, ctx->stream, n, (int)it, (const double *)a->z.p, p->pvec.p,
                           (const double *)p->part_rz.p, Gv, p->scal.p, flexible ? (const double *)p->part_bb.p : nullptr);
        FV_LAUNCH_CHECK(ctx);
    }
    return FV_OK;
}

// ------------------------------------------------------------------ C ABI
extern "C" int fv_precond_set(fv_problem *p, int kind)
{
    if (!p || (kind != FV_PRECOND_JACOBI && kind != FV_PRECOND_AMG && kind != FV_PRECOND_AUTO && kind != FV_PRECOND_AMG_GATHERED))
        return FV_ERR_ARG;
    const bool gathered = kind == FV_PRECOND_AMG_GATHERED;
    if (gathered)
        kind = FV_PRECOND_AMG; // (the same solver paths; the hierarchy differs: amg_build_pooled)
    // (validation first, state afterwards: a refused call leaves the problem's preconditioner and hierarchy as they were — ADVICE r4)
    if (kind != FV_PRECOND_JACOBI && p->nhalo && !p->dist) {
        fv_set_error(p->ctx, "the AMG preconditioner needs a whole operator or a row block set up by fv_dist_setup");
        return FV_ERR_STATE;
    }
    if (kind == FV_PRECOND_AUTO && p->dist) {
        fv_set_error(p->ctx, "row blocks take FV_PRECOND_JACOBI or FV_PRECOND_AMG (every rank must make the same choice: no automatic switch)");
        return FV_ERR_ARG;
    }
    if (p->amg && p->amg_gathered != gathered) { // another hierarchy
        fv_amg_free(p->amg);
        p->amg = nullptr;
    }
    p->amg_gathered = gathered;
    p->precond = kind;
    p->auto_steps_amg = false;
    return FV_OK;
}

extern "C" int fv_amg_info(fv_problem *p, int32_t *nlevels, int64_t *rows, int64_t *nnz, int32_t cap)
{
    if (!p || !nlevels)
        return FV_ERR_ARG;
    fv_ctx *ctx = p->ctx;
    FV_HIP(ctx, hipSetDevice(ctx->device));
    if (!p->assembled) {
        fv_set_error(ctx, "fv_amg_info: call fv_assemble first");
        return FV_ERR_STATE;
    }
    FV_TRY(fv_pcg_prepare(p));
    FV_TRY(fv_amg_prepare(p, 0.0));
    *nlevels = (int32_t)p->amg->lev.size();
    for (int32_t l = 0; l < *nlevels && l < cap; l++) {
        if (rows)
            rows[l] = p->amg->lev[(size_t)l]->n;
        if (nnz)
            nnz[l] = p->amg->lev[(size_t)l]->nnz;
    }
    return FV_OK;
}

// z = M^-1 r on host vectors (tests: symmetry and definiteness of the preconditioner)
extern "C" int fv_amg_apply(fv_problem *p, const double *r_free, double sigma, double *z_free)
{
    if (!p || !r_free || !z_free)
        return FV_ERR_ARG;
    fv_ctx *ctx = p->ctx;
    FV_HIP(ctx, hipSetDevice(ctx->device));
    if (!p->assembled) {
        fv_set_error(ctx, "fv_amg_apply: call fv_assemble first");
        return FV_ERR_STATE;
    }
    FV_TRY(fv_pcg_prepare(p));
    FV_TRY(fv_free_in(p, p->tmp.p, r_free));
    if (p->nhalo > 0) // a row block: the cycle's level-0 products read the iterate's halo slots, which must hold zeros (block-Jacobi)
        FV_HIP(ctx, hipMemsetAsync(p->rhs.p + p->n, 0, (size_t)p->nhalo * sizeof(double), ctx->stream));
    FV_TRY(fv_amg_apply_device(p, p->tmp.p, p->rhs.p, sigma));
    return fv_free_out(p, z_free, p->rhs.p);
}

FV_WARM_TU(amg) // (fv_ctx_create loads every code object of the library up front: fv_warm_modules, fv_ctx.hip)
