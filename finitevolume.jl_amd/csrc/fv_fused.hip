// The fused time step of the one-iteration regime ("KF"): the vector update of step k (what K2S does: alpha, x_out, the
// convergence sums, the next step's set-up) and the SpMV of step k + 1 (what K1 does: q' = (A + sigma D) z', partial z'.q')
// in ONE pass over the symmetric arrays, 2-D tiles of a plane marching through the planes.
//
// Replaces, for fixed-dt runs whose steps converge in one PCG iteration, the K1 + K2S pair behind
// /root/reference/src/transient.jl:60-76 (backwardeuleronestep!: rhs = b + u/dt, solve (A + I/dt) x = rhs from x0 = u) as
// driven by fixedbackwardeulerstep! (transient.jl:130-134); semantics unchanged: the same Jacobi-PCG iteration, the same
// stopping rule ||r|| <= rtol ||rhs||, the same fall-back to further iterations when a step does not converge in one.
//
// The v-form.  The residual of an implicit step's system at its start state is rho(x) = b - A x whatever the time step,
// so across one-iteration steps  rho' = rho - alpha A z  with z = M^-1 rho the step's direction: the next direction is
//     z' = z + alpha v ,   v = -M^-1 (A z) = -M^-1 (q - sigma D z) ,
// and the launch that forms q = (A + sigma D) z stores v INSTEAD of q.  The SpMV of the next step needs z' on the rows
// around a tile; with v those halo rows cost two streams and one FMA (no diagonal, no storage term, no division), which
// is what makes the fusion pay: a row is read once (x, z, v, the three upper diagonals, one storage code byte) and written
// once (x_out, z', v'), 73 bytes instead of the 90 of K1 + K2S.  Per own row, with d = the shifted diagonal (re-derived
// from the six arms where the row sum is zero, streamed elsewhere) and sD = sigma D of the row:
//     x_out = x + alpha z ;  z' = z + alpha v ;  rho' = d z'  (next step's residual at x_out) ;
//     r = rho' - sD (x_out - x)  (this step's residual after its iteration: the convergence test) ;
//     sums: r.M^-1 r, r.r | rho'.z', rho'.rho', |sD x_out (+ b)|^2 | z'.q'
// A launch checks the PREVIOUS step's verdict first (all blocks reduce the same partial sums): if that step did not
// converge in its one iteration the launch turns into the fall-back (r and the next direction of that step, done = 3)
// exactly as pcg_chain_boundary_kernel does for the unfused chain, and the host resumes that step at iteration 2.
//
// Traversal: a block of TL x TW / 2 threads owns TL lines x TW columns of a plane, two consecutive columns per thread
// (16-byte accesses).  Per plane step: loads of plane p + 2; update of plane p + 1 (needs the U1 / U2 tiles of that plane
// for its diagonal); product of plane p from the z' tile in LDS (+-1, +-line arms), registers (+-plane arms) and the LDS
// ring of U1 / U2 tiles; one barrier.  LDS: z' tile double-buffered, U1 / U2 in a ring of three plane slots.
#include "fv_internal.h"
#include "fv_device.h"
#include "fv_spmv.h"
#include "fv_fused.h"

int g_fused = 1;       // fv_tune key 41: 0 = never use the fused step
static const int g_fused_blocks = 2; // (round 3's experiment key 42, frozen) resident blocks per CU the grid is sized for (8-line tiles)
static const int g_fused_segs = 0;   // (experiment key 43, frozen) segments of planes per tile, 0 = chosen to fill whole rounds
int g_fused_codes = 1;  // fv_tune key 49: the matrix as 16-bit codes per row where its diagonals take few distinct values (0: always the doubles)
int g_fused_iter = 1;   // fv_tune key 46: the many-iteration loop through the fused kernel too (direction update + product in one pass, z kept instead of r)
static const int g_fused_nt = 0;     // (experiment key 45, frozen: every bit lost its A/B) bit 0 = z' stored non-temporally, bit 1 = v' too, bit 2 = x / v loaded with plain loads, bit 3 = x_out stored plainly, bit 4 = matrix loaded with plain loads
int g_fused_dist_spare = 1; // (key 51, frozen) CUs per XCD a row block's fused launch leaves to the halo exchange
static const int g_fused_sell_blocks = 4; // (experiment key 56, frozen) resident blocks per CU the SELL step's grid is sized for
int g_fused_sell = 1;  // fv_tune key 55: the fused step on the SELL form (irregular meshes) too
int g_fused_dist = 1;  // fv_tune key 50: the fused step on row blocks too (0: row blocks keep the K1 + K2S pair)
int g_fused_chunk = 1;  // fv_tune key 60: the coded fused step / pass on CHUNKS of a plane (fused_chunk_kernel: no column halos) where it applies;
                        // 0 = always the 2-D tiles
static const int g_fused_lines = 16; // (key 44, frozen) lines per tile, 8 (blocks of 512 threads, two per CU) or 16 (1024 threads, one per CU: fewer halo rows per
                        // own row; 464^3, same process: 1.351 against 1.438 ms per step, the K1 + K2S pair 1.696)

namespace {

constexpr int KF_TW = 128; // columns of a tile; lines: 8 (512 threads, two blocks per CU) or 16 (1024 threads, one)
constexpr int KF_NSUM = 6;

struct KfArgs {
    // geometry: lines of nz rows, planes of d3 rows, nplanes = one past the last plane whose product this kernel forms
    int32_t nz, L, d3, nplanes, P, seglen, tilesC, tiles, nsegs;
    int32_t chunk; // fused_chunk_kernel: rows of a plane per work item (tiles = chunks per plane)
    uint32_t front; // doubles of zero padding in front of row 0 of the symmetric arrays (dg, u1, u2, u3 point at row 0)
    int32_t pfirst; // ... its first product plane: 1, or 0 when the first and the last plane are centre planes too (kc_ends; nplanes is then P)
    // the symmetric arrays (row 0 pointers: zero-padded in front and behind), flags per 64-row slice, storage codes
    const double *dg, *u1, *u2, *u3;
    const uint8_t *ok, *code;
    const uint16_t *mcode; // CODED: per row, codes of U1 | U2 << 5 | U3 << 10 into mt; bit 15: the row's slice is one whose products the symmetric kernels form
    const uint8_t *kcode;  // fused_chunk_kernel: per row, storage code | diagonal code << 4 (0: the diagonal follows from the arms, k: kdiag.v[k])
    StorageTable kdiag;
    MatrixTables mt;
    StorageTable sD; // sigma x the distinct values of D
    // vectors
    const double *x, *z, *v;
    double *xout, *znext, *vnext;
    const double *w; // MODE 3 (the whole PCG iteration in one launch): w = -M^-1 q of the previous launch; z' = z + ax w goes to zout
    double *zout;
    int first;       // MODE 3: the first pass of a solve — the direction is z itself, nothing but w' is stored
    // the assembled b on its support (sparse): share of |rhs|^2
    const int32_t *bidx;
    const double *b;
    int64_t bm;
    // scalars / control
    PcgScalars *scal;
    FusedSums in, out;
    int nt;          // streaming hints (bit mask; frozen at the measured best since round 4)
    int xapply;      // MODE 1: apply the lagging x-update of the previous iteration, x += scal->alpha_last * p (p = a.v, the old direction)
    int mode;        // 0: scalars of this step already in scal (set-up finalised by an earlier launch); 1: merged boundary
    int chain_index; // index of this step in its burst
    int force_prev_unconverged;
    double rtol;
    const double *red; // row blocks: the six sums all-reduced over the ranks — [0] z.q, [1] r.M^-1 r, [2] r.r of the previous step, [3..5] this
                       // step's rho.z, rho.rho, rhs.rhs — instead of the previous launch's partial sums (null: single GPU)
    double *hist;     // MODE 1: residual history (may be null) and its capacity; chain_index = the iteration the vector pass finished
    int64_t hist_cap;
    // fall-back of the previous step (mode 1): see pcg_chain_boundary_kernel
    int64_t n;
    double *r, *pold;
    const double *minv, *D, *xprev;
    double dt;
};

// all threads get the sum of v over the block; red: NT / 64 doubles
template <int NT>
__device__ inline double kf_block_sum(double v, double *red)
{
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0)
        red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < NT / 64; w++)
        t += red[w];
    return t;
}
template <int NT>
__device__ inline double kf_reduce(const double *__restrict__ part, int count, double *red)
{
    double v = 0.0;
    for (int i = threadIdx.x; i < count; i += NT)
        v += part[i];
    return kf_block_sum<NT>(v, red);
}

struct VRow {
    double xn, zn, r, c, h, mv;
};
__device__ __forceinline__ VRow vrow(double xin, double z, double v, double d, double sD, double alpha)
{
    VRow o;
    o.mv = 1.0 / d;
    o.xn = xin + alpha * z;
    o.zn = z + alpha * v;
    o.c = d * o.zn;                // rho': the next system's residual at x_out
    o.r = o.c - sD * (o.xn - xin); // the finished step's residual (increment as stored, like pcg_carry_init_kernel)
    o.h = sD * o.xn;
    return o;
}

// The scalar part of a fused step (MODE 0), shared by the tiled kernel and the SELL kernel: every block takes the same decisions
// from values no block of this launch writes (the partial sums of the previous launch or the all-reduced sums of a row block, scal
// fields written by earlier launches) — the previous chained step's verdict and, if it did not converge, its fall-back (that
// step's residual and next direction, done = 3); this step's set-up scalars, zero-iteration steps, breakdown; alpha.
// false: the launch stops here.  red: NT / 64 doubles of LDS.
template <int NT>
__device__ __forceinline__ bool kf_step_prologue(const KfArgs &a, double *red, double &alpha)
{
    const int tid = (int)threadIdx.x;
    PcgScalars *scal = a.scal;
    const int d0 = *reinterpret_cast<volatile int32_t *>(&scal->done);
    if (d0 == 2)
        return false;
    if (d0 == 3 && !(a.mode == 1 && *reinterpret_cast<volatile int32_t *>(&scal->chain_step) == a.chain_index - 1))
        return false; // the chain broke at an earlier step (the exception: block 0 of THIS launch has just said so)
    double rz0;
    bool zero_iteration;
    // a sum of the previous launch: its partial pieces, or (row blocks) the value all-reduced over the ranks
    auto total = [&](int which, const double *part, int count) -> double { return a.red ? a.red[which] : kf_reduce<NT>(part, count, red); };
    if (a.mode == 1) {
        // the previous chained step's verdict
        const double rrn = total(2, a.in.arr, a.in.nvec);
        const bool converged = rrn <= scal->tol2x[(a.chain_index - 1) & 1] && !a.force_prev_unconverged;
        if (!converged) {
            // as pcg_chain_boundary_kernel: that step's residual without the next step's storage term, p = M^-1 r + beta p
            const double rzn = total(1, a.in.arz, a.in.nvec);
            const double beta = rzn / scal->rz[0];
            const int64_t n2 = a.n >> 1;
            double2 *r2 = reinterpret_cast<double2 *>(a.r);
            const double2 *m2 = reinterpret_cast<const double2 *>(a.minv);
            double2 *p2 = reinterpret_cast<double2 *>(a.pold);
            const double2 *xi2 = reinterpret_cast<const double2 *>(a.xprev);
            const double2 *xo2 = reinterpret_cast<const double2 *>(a.x);
            const double2 *D2 = reinterpret_cast<const double2 *>(a.D);
            const double2 *z2 = reinterpret_cast<const double2 *>(a.z);
            for (int64_t i = (int64_t)blockIdx.x * NT + tid; i < n2; i += (int64_t)gridDim.x * NT) {
                const double2 mv = m2[i], zv = z2[i], xa = xi2[i], xb = xo2[i], dv = D2[i];
                double2 rv = make_double2(zv.x / mv.x, zv.y / mv.y);
                rv.x -= dv.x * ((xb.x - xa.x) / a.dt);
                rv.y -= dv.y * ((xb.y - xa.y) / a.dt);
                r2[i] = rv;
                double2 pv = p2[i];
                pv.x = mv.x * rv.x + beta * pv.x;
                pv.y = mv.y * rv.y + beta * pv.y;
                p2[i] = pv;
            }
            if ((a.n & 1) && blockIdx.x == 0 && tid == 0) {
                const int64_t i = a.n - 1;
                double ri = a.z[i] / a.minv[i];
                ri -= a.D[i] * ((a.x[i] - a.xprev[i]) / a.dt);
                a.r[i] = ri;
                a.pold[i] = a.minv[i] * ri + beta * a.pold[i];
            }
            if (blockIdx.x == 0 && tid == 0) {
                scal->rz[1] = rzn;
                scal->rr = rrn;
                scal->iters = 1;
                scal->chain_step = a.chain_index - 1;
                __threadfence();
                scal->done = 3;
            }
            return false;
        }
        rz0 = total(3, a.in.srz, a.in.nvec);
        const double rr0 = total(4, a.in.srr, a.in.nvec);
        const double bb = total(5, a.in.sbb, a.in.nbb);
        const double tol2 = a.rtol * a.rtol * bb;
        zero_iteration = rr0 <= tol2;
        if (blockIdx.x == 0 && tid == 0) {
            scal->rz[0] = rz0;
            scal->rz[1] = 0.0;
            scal->rr = rr0;
            scal->bnorm2 = bb;
            scal->tol2 = tol2;
            scal->tol2x[a.chain_index & 1] = tol2;
            scal->iters = 0;
            scal->done = zero_iteration ? 1 : 0;
        }
    } else {
        rz0 = scal->rz[0];
        zero_iteration = d0 == 1; // converged at its set-up (pcg_init_finalize_kernel said so)
    }
    if (!zero_iteration) {
        const double pq = total(0, a.in.pq, a.in.npq);
        if (!(pq > 0.0)) { // breakdown: not positive definite, or NaN
            if (blockIdx.x == 0 && tid == 0) {
                scal->pq = pq;
                scal->done = 2;
            }
            return false;
        }
        alpha = rz0 / pq;
        if (blockIdx.x == 0 && tid == 0)
            scal->pq = pq;
    } else if (blockIdx.x == 0 && tid == 0)
        scal->zero_mask |= 1u << a.chain_index; // the step counts 0 iterations: alpha = 0 hands the state over unchanged
    return true;
}

// The scalar part of ONE-LAUNCH PCG iterations (MODE 3 of the chunk kernels; fv_ploop_pass; round 5).  Launch j >= 1 of a solve finds
// what launch j - 1 left: the iterate's TRUE sums r.z and r.r (r = d z, formed where the diagonal is at hand) and five sums over that
// launch's product q = (A + sigma D) p:  p.q,  z.q,  q.M^-1 q,  r.q,  q.q.  With the step length alpha = r.z / p.q the NEXT iterate's
// sums are polynomials in alpha — z' = z - alpha M^-1 q, r' = r - alpha q:
//     r'.z' = r.z - 2 alpha z.q + alpha^2 q.M^-1 q ,     r'.r' = r.r - 2 alpha r.q + alpha^2 q.q
// — so this launch has the verdict on iterate j and beta = r'.z' / r.z BEFORE its pass, and the pass itself forms z' = z + alpha w
// (w = -M^-1 q stored by the previous launch), x += alpha p, p' = z' + beta p and the next product: no vector-update launch, no
// M^-1 stream, 89 B per row and iteration where the pass + update pair moved 105 (67 instead of 76 with the matrix as codes).
// Same Jacobi-PCG iteration and stopping rule (/root/reference/src/transient.jl:50-58); every launch re-bases on the true sums of
// the iterate it reads, so the polynomial's rounding (relative eps x r.r / r'.r': 1e-9 when an iteration gains four digits) never
// accumulates.  a.first: the first pass of a solve (alpha = beta = 0: the direction is z itself; the set-up has given its verdict).
// false: the launch stops here (converged, broken down, or already done).
template <int NT>
__device__ __forceinline__ bool kf_ploop_prologue(const KfArgs &a, double *red, double &ax, double &beta)
{
    const int tid = (int)threadIdx.x;
    PcgScalars *scal = a.scal;
    if (*reinterpret_cast<volatile int32_t *>(&scal->done))
        return false;
    ax = 0.0;
    beta = 0.0;
    if (a.first)
        return true;
    const int np = a.in.npq;
    const double pq = kf_reduce<NT>(a.in.pq, np, red);
    if (!(pq > 0.0)) { // breakdown: not positive definite, or NaN
        if (blockIdx.x == 0 && tid == 0) {
            scal->pq = pq;
            scal->done = 2;
        }
        return false;
    }
    const double rzb = kf_reduce<NT>(a.in.arz, np, red), rrb = kf_reduce<NT>(a.in.arr, np, red);
    const double s1 = kf_reduce<NT>(a.in.srz, np, red), s2 = kf_reduce<NT>(a.in.srr, np, red);
    const double t1 = kf_reduce<NT>(a.in.sbb, np, red), t2 = kf_reduce<NT>(a.in.t2, np, red);
    const double alpha = rzb / pq;
    double rzn = rzb + alpha * (alpha * s2 - 2.0 * s1), rrn = rrb + alpha * (alpha * t2 - 2.0 * t1);
    if (!(rrn > 0.0))
        rrn = 0.0; // (cancellation to below the rounding of r.r: the iteration gained more than eight digits)
    const bool converged = rrn <= scal->tol2;
    if (blockIdx.x == 0 && tid == 0) {
        scal->rz[(a.chain_index + 1) & 1] = rzn;
        scal->rr = rrn;
        scal->pq = pq;
        scal->iters = a.chain_index + 1;
        scal->alpha_last = alpha; // the update this launch applies (or, if it stops here, the flush kernel: fv_ploop_flush)
        if (a.hist && a.chain_index < a.hist_cap)
            a.hist[a.chain_index] = sqrt(rrn);
        if (converged)
            scal->done = 1;
    }
    if (converged)
        return false;
    ax = alpha;
    beta = rzn > 0.0 ? rzn / rzb : 0.0; // (r'.z' lost to cancellation: restart the directions from z')
    return true;
}

// MODE 0: the fused step.  MODE 1: one pass of the many-iteration regime — the direction update of PCG iteration `it` and its
// product: p' = z + beta p (z = M^-1 r, kept instead of r between the passes), q = (A + sigma D) p', partial p'.q; the scalar
// work of K3 (beta, the convergence verdict, the residual history) in the prologue.  Same traversal, same halo trick (a halo
// row is z + beta p: two streams, one FMA); no x, no storage term, no vector sums.  a.z = z, a.v = p (old), a.znext = p', a.vnext = q.
// CODED: the three upper diagonals come as one 16-bit word per row — three 5-bit codes into tables of their distinct values
// (a homogeneous conductivity on a regular grid: a handful of values per direction; fv_matrix_codes) — instead of three doubles:
// 2 instead of 24 bytes of matrix per row, the same doubles out of the tables, so nothing else changes.
template <int TL, int MODE, bool CODED>
__global__ __launch_bounds__(TL * 64, 4) void fused_step_kernel(KfArgs a)
{
    constexpr int TW = KF_TW, NT = TL * TW / 2, HC = TW / 2;
    constexpr int ZS = TW + 4, U1S = TW + 2, U2S = TW; // LDS row strides: own column tc sits at 2 + tc (z', U1) so that pairs stay 16-byte aligned
    constexpr int ZT = (TL + 2) * ZS, U1T = TL * U1S, U2T = (TL + 1) * U2S;
    constexpr int NH = 2 * TW + 2 * TL; // threads with a halo row: line above, line below, column left, column right
    __shared__ __align__(16) double zs[2 * ZT];
    __shared__ __align__(16) double u1s[3 * U1T];
    __shared__ __align__(16) double u2s[3 * U2T];
    __shared__ double tab[FV_STORAGE_CODES];
    __shared__ double mtab[CODED ? 3 * FV_MATRIX_CODES : 1];
    __shared__ double red[NT / 64];
    const int tid = (int)threadIdx.x;
    PcgScalars *scal = a.scal;
    // ---------------------------------------------------------------- scalars: every block takes the same decisions from
    // values no block of this launch writes (the partial sums of the previous launch, scal fields written by earlier ones)
    double alpha = 0.0;
    if (MODE == 1) {
        if (*reinterpret_cast<volatile int32_t *>(&scal->done))
            return;
        // K3's scalars (pcg_pupdate_kernel): beta from the sums the vector pass left, the verdict on the iteration that pass finished
        const double rzn = kf_reduce<NT>(a.in.arz, a.in.nvec, red);
        const double rrn = kf_reduce<NT>(a.in.arr, a.in.nvec, red);
        const bool converged = rrn <= scal->tol2;
        if (blockIdx.x == 0 && tid == 0) {
            scal->rz[(a.chain_index + 1) & 1] = rzn;
            scal->rr = rrn;
            scal->iters = a.chain_index + 1;
            if (a.hist && a.chain_index < a.hist_cap)
                a.hist[a.chain_index] = sqrt(rrn);
            if (converged)
                scal->done = 1;
        }
        if (converged)
            return;
        alpha = rzn / scal->rz[a.chain_index & 1]; // (beta: it plays alpha's part in z + alpha v)
    }
    if (MODE == 0) {
        if (!kf_step_prologue<NT>(a, red, alpha))
            return;
    }
    // MODE 1: x += ax * p_old, the x-update of the previous iteration, which lags one pass behind (its direction is this pass's a.v)
    const bool XU = MODE == 0 || a.xapply != 0;
    const double ax = MODE == 1 && a.xapply ? scal->alpha_last : 0.0;
    // ---------------------------------------------------------------- the pass
    if (tid < FV_STORAGE_CODES)
        tab[tid] = a.sD.v[tid];
    if (CODED && tid < 3 * FV_MATRIX_CODES)
        mtab[tid] = a.mt.v[tid];
    const int32_t nz = a.nz, d3 = a.d3;
    const int tl = tid / HC, tc = 2 * (tid % HC);
    const int xcd = (int)(blockIdx.x & 7);
    const int64_t items = (int64_t)a.tiles * a.nsegs, per_xcd = (items + 7) / 8;
    double acc[KF_NSUM] = {0, 0, 0, 0, 0, 0};
    const int zo = (tl + 1) * ZS + 2 + tc, u1o = tl * U1S + 2 + tc, u2o = (tl + 1) * U2S + tc; // own positions in the tiles
    for (int64_t j = (int64_t)(blockIdx.x >> 3); j < per_xcd; j += (int64_t)(gridDim.x >> 3)) {
        const int64_t item = (int64_t)xcd * per_xcd + j;
        if (item >= items)
            break;
        const int32_t seg = (int32_t)(item / a.tiles), tile = (int32_t)(item % a.tiles);
        const int32_t l0 = (tile / a.tilesC) * TL, c0 = (tile % a.tilesC) * TW;
        const int32_t p0 = 1 + seg * a.seglen, p1 = (p0 + a.seglen < a.nplanes) ? p0 + a.seglen : a.nplanes;
        if (p0 >= p1)
            continue;
        // the vector part (x_out, z', the five vector sums) of the plane before the first / after the last product plane
        // belongs to the first / last segment: planes 0 and P - 1 are nobody's product planes here (their rows are next to
        // Dirichlet cells: the slice-by-slice launch forms their products)
        const bool vec_first = p0 == 1, vec_last = p1 == a.nplanes && a.nplanes == a.P - 1;
        const bool own = (l0 + tl < a.L) && (c0 + tc < nz);
        const uint32_t o = (uint32_t)((l0 + tl) * nz + c0 + tc);
        bool hv = false;
        uint32_t ho = 0;
        int hz = 0, hu = -1;
        bool hu_is_u2 = false;
        if (tid < TW) { // the line above the tile
            hv = l0 >= 1 && c0 + tid < nz;
            ho = (uint32_t)((l0 - 1) * nz + c0 + tid);
            hz = 2 + tid;
            hu = tid;
            hu_is_u2 = true;
        } else if (tid < 2 * TW) { // the line below
            const int t = tid - TW;
            hv = l0 + TL < a.L && c0 + t < nz;
            ho = (uint32_t)((l0 + TL) * nz + c0 + t);
            hz = (TL + 1) * ZS + 2 + t;
        } else if (tid < 2 * TW + TL) { // the column to the left
            const int t = tid - 2 * TW;
            hv = c0 >= 1 && l0 + t < a.L;
            ho = (uint32_t)((l0 + t) * nz + c0 - 1);
            hz = (t + 1) * ZS + 1;
            hu = t * U1S + 1;
        } else if (tid < NH) { // the column to the right
            const int t = tid - 2 * TW - TL;
            hv = c0 + TW < nz && l0 + t < a.L;
            ho = (uint32_t)((l0 + t) * nz + c0 + TW);
            hz = (t + 1) * ZS + 2 + TW;
        }
        const bool ht = tid < NH;
        if (!hv)
            ho = own ? o : 0u; // (never used; keeps the address in range)
        // one register for the halo role: z' slot | U slot << 12 | exists << 24 | has a U value << 25 | that value is U2's << 26
        const uint32_t hdesc = (uint32_t)hz | ((uint32_t)(hu >= 0 ? hu : 0) << 12) | ((uint32_t)hv << 24) | ((uint32_t)(hu >= 0) << 25) |
                               ((uint32_t)hu_is_u2 << 26);
#define KF_HZ ((int)(hdesc & 4095u))
#define KF_HU ((int)((hdesc >> 12) & 4095u))
#define KF_HV ((hdesc >> 24) & 1u)
#define KF_HHU ((hdesc >> 25) & 1u)
#define KF_HU2 ((hdesc >> 26) & 1u)
        // plane bases are uniform (scalar registers), the lane part is one 32-bit byte offset: "saddr + voffset" accesses
        uint32_t ob = (own ? o : 0u) * 8u, hb = ho * 8u; // rows outside the plane read row 0 of it: finite, never used
        auto PB = [&](const void *arr, int32_t pl, int esz) -> const char * {
            return reinterpret_cast<const char *>(arr) + (uint64_t)((int64_t)pl * d3) * (uint64_t)esz;
        };
        auto P2 = [&](const double *arr, int32_t pl) -> double2 { return *reinterpret_cast<const double2 *>(PB(arr, pl, 8) + ob); };
        auto P2nt = [&](const double *arr, int32_t pl) -> double2 {
            const double *b = reinterpret_cast<const double *>(PB(arr, pl, 8) + ob);
            return make_double2(__builtin_nontemporal_load(b), __builtin_nontemporal_load(b + 1));
        };
        auto C2 = [&](int32_t pl) -> uint32_t { // storage codes of the thread's two rows (one value: code 0)
            return a.code ? (uint32_t) * reinterpret_cast<const uint16_t *>(PB(a.code, pl, 1) + (ob >> 3)) : 0u;
        };
        auto SD = [&](uint32_t c) -> double2 { return make_double2(tab[c & 255u], tab[c >> 8]); }; // sigma D of the two rows
        auto H1 = [&](const double *arr, int32_t pl) -> double { return *reinterpret_cast<const double *>(PB(arr, pl, 8) + hb); };
        // the matrix values of the thread's two rows of plane pl: which = 0, 1, 2 for U1, U2, U3 (as doubles, or out of the tables)
        auto MW = [&](int32_t pl) -> uint32_t { return *reinterpret_cast<const uint32_t *>(PB(a.mcode, pl, 2) + (ob >> 2)); };
        auto MV = [&](uint32_t w, int which) -> double2 {
            return make_double2(mtab[which * FV_MATRIX_CODES + ((w >> (5 * which)) & 31u)], mtab[which * FV_MATRIX_CODES + ((w >> (16 + 5 * which)) & 31u)]);
        };
        auto MU = [&](const double *arr, int which, int32_t pl, bool plain) -> double2 {
            if (CODED)
                return MV(MW(pl), which);
            return plain ? P2(arr, pl) : P2nt(arr, pl);
        };
        auto MH = [&](bool u2, int32_t pl) -> double { // a halo row's U2 (line above) or U1 (column to the left)
            if (CODED) {
                const uint32_t w = *reinterpret_cast<const uint16_t *>(PB(a.mcode, pl, 2) + (hb >> 2));
                return mtab[(u2 ? 1 : 0) * FV_MATRIX_CODES + ((w >> (u2 ? 5 : 0)) & 31u)];
            }
            return H1(u2 ? a.u2 : a.u1, pl);
        };
        auto ST2 = [&](double *arr, int32_t pl, double2 val) { *reinterpret_cast<double2 *>(const_cast<char *>(PB(arr, pl, 8)) + ob) = val; };
        auto ST2nt = [&](double *arr, int32_t pl, double2 val) {
            double *q = reinterpret_cast<double *>(const_cast<char *>(PB(arr, pl, 8)) + ob);
            __builtin_nontemporal_store(val.x, q);
            __builtin_nontemporal_store(val.y, q + 1);
        };
        __syncthreads(); // tab; the previous item's last LDS reads
        // ---------------- prologue: z' of plane p0 - 1 (registers), z' of plane p0 (tile + halo), U tiles of p0 and p0 + 1
        double2 Zm, Pc, Mc; // Mc: M^-1 of the centre plane's rows (Cc: their storage codes); Pc: the -plane and diagonal terms of their products
        uint32_t Cc;
        int flc;                // ... and their slice flags (bit 0: this kernel forms the product, bit 1: diagonal from the arms)
        int zb = p0 & 1;        // half of zs that holds the centre plane's z'
        int s0 = 0, s1 = 1, s2 = 2; // ring slots of planes p, p + 1, p + 2
        {
            const double2 vv = P2(a.v, p0 - 1), zz = P2(a.z, p0 - 1);
            Zm = make_double2(zz.x + alpha * vv.x, zz.y + alpha * vv.y);
            if (MODE == 1 && vec_first && own) {
                ST2(a.znext, 0, Zm);
                if (XU) {
                    const double2 xi = P2(a.x, 0);
                    ST2(a.xout, 0, make_double2(xi.x + ax * vv.x, xi.y + ax * vv.y));
                }
            }
            if (MODE == 0 && vec_first && own) { // plane 0: its vector part
                const double2 xi = P2(a.x, 0), dd = P2(a.dg, 0), ss = SD(C2(0));
                const VRow ra = vrow(xi.x, zz.x, vv.x, dd.x, ss.x, alpha), rb = vrow(xi.y, zz.y, vv.y, dd.y, ss.y, alpha);
                ST2(a.xout, 0, make_double2(ra.xn, rb.xn));
                ST2(a.znext, 0, Zm);
                acc[0] += ra.r * (ra.mv * ra.r) + rb.r * (rb.mv * rb.r);
                acc[1] += ra.r * ra.r + rb.r * rb.r;
                acc[2] += ra.c * ra.zn + rb.c * rb.zn;
                acc[3] += ra.c * ra.c + rb.c * rb.c;
                acc[4] += ra.h * ra.h + rb.h * rb.h;
            }
        }
        double2 Zc0;
        {
            const double2 xi = MODE == 0 ? P2(a.x, p0) : make_double2(0.0, 0.0), vv = P2(a.v, p0), zz = P2(a.z, p0), dd = P2(a.dg, p0);
            // (the stored diagonal: bit for bit the derived one where bit 1 of the flag is set, symdia_rowsum_kernel)
            Cc = C2(p0);
            const double2 ss = SD(Cc);
            const VRow ra = vrow(xi.x, zz.x, vv.x, own ? dd.x : 1.0, ss.x, alpha), rb = vrow(xi.y, zz.y, vv.y, own ? dd.y : 1.0, ss.y, alpha);
            Zc0 = make_double2(ra.zn, rb.zn);
            Mc = make_double2(ra.mv, rb.mv);
            if (MODE == 1 && own) {
                ST2(a.znext, p0, Zc0);
                if (XU) {
                    const double2 xp = P2(a.x, p0);
                    ST2(a.xout, p0, make_double2(xp.x + ax * vv.x, xp.y + ax * vv.y));
                }
            }
            if (MODE == 0 && own) {
                ST2(a.xout, p0, make_double2(ra.xn, rb.xn));
                ST2(a.znext, p0, Zc0);
                acc[0] += ra.r * (ra.mv * ra.r) + rb.r * (rb.mv * rb.r);
                acc[1] += ra.r * ra.r + rb.r * rb.r;
                acc[2] += ra.c * ra.zn + rb.c * rb.zn;
                acc[3] += ra.c * ra.c + rb.c * rb.c;
                acc[4] += ra.h * ra.h + rb.h * rb.h;
            }
            const double2 A3m = MU(a.u3, 2, p0 - 1, false);
            Pc = make_double2(A3m.x * Zm.x + dd.x * Zc0.x, A3m.y * Zm.y + dd.y * Zc0.y);
        }
        flc = own ? (int)a.ok[((int64_t)p0 * d3 + o) >> 6] : 0;
        const uint32_t w0 = CODED ? MW(p0) : 0u, w1 = CODED ? MW(p0 + 1) : 0u;
        double2 A3c = CODED ? MV(w0, 2) : P2nt(a.u3, p0), A3n = CODED ? MV(w1, 2) : P2nt(a.u3, p0 + 1);
        *reinterpret_cast<double2 *>(zs + zb * ZT + zo) = Zc0;
        *reinterpret_cast<double2 *>(u1s + s0 * U1T + u1o) = CODED ? MV(w0, 0) : P2nt(a.u1, p0);
        *reinterpret_cast<double2 *>(u2s + s0 * U2T + u2o) = CODED ? MV(w0, 1) : P2nt(a.u2, p0);
        *reinterpret_cast<double2 *>(u1s + s1 * U1T + u1o) = CODED ? MV(w1, 0) : P2nt(a.u1, p0 + 1);
        *reinterpret_cast<double2 *>(u2s + s1 * U2T + u2o) = CODED ? MV(w1, 1) : P2nt(a.u2, p0 + 1);
        if (ht) {
            double zn = 0.0, ua = 0.0, ub = 0.0;
            if (hv) {
                zn = H1(a.z, p0) + alpha * H1(a.v, p0);
                if (hu >= 0) {
                    ua = MH(hu_is_u2, p0);
                    ub = MH(hu_is_u2, p0 + 1);
                }
            }
            zs[zb * ZT + hz] = zn;
            if (hu >= 0) {
                (hu_is_u2 ? u2s + s0 * U2T : u1s + s0 * U1T)[hu] = ua;
                (hu_is_u2 ? u2s + s1 * U2T : u1s + s1 * U1T)[hu] = ub;
            }
        }
        double2 Xa = XU ? P2nt(a.x, p0 + 1) : make_double2(0.0, 0.0), Va = P2nt(a.v, p0 + 1), Za = P2(a.z, p0 + 1);
        uint32_t Ca = C2(p0 + 1);
        int fla = own ? (int)a.ok[((int64_t)(p0 + 1) * d3 + o) >> 6] : 0;
        __syncthreads();
        for (int32_t p = p0; p < p1; p++) {
            // (opaque to the optimiser, or it keeps one 64-bit lane address per stream alive across the loop instead of
            // scalar plane base + this one offset)
            asm volatile("" : "+v"(ob), "+v"(hb));
            // ---- own streams of plane p + 2, halo streams of plane p + 1
            const bool more = p + 2 <= p1, inseg = p + 1 < p1;
            const bool vec_n = inseg || vec_last; // plane p + 1's vector part is ours
            double2 Xb = make_double2(0.0, 0.0), Vb = Xb, Zb = Xb, V1b = Xb, V2b = Xb, A3b = Xb;
            uint32_t Cb = 0, Wb = 0; // (CODED: the matrix word of plane p + 2, decoded when it is needed)
            int flb = 0;
            double hq = 0.0, hzv = 0.0, hub = 0.0;
            if (more) {
                if (XU)
                    Xb = (a.nt & 4) ? P2(a.x, p + 2) : P2nt(a.x, p + 2);
                Vb = (a.nt & 4) ? P2(a.v, p + 2) : P2nt(a.v, p + 2);
                Zb = P2(a.z, p + 2);
                Cb = C2(p + 2);
                if (CODED)
                    Wb = MW(p + 2);
                else {
                    V1b = (a.nt & 16) ? P2(a.u1, p + 2) : P2nt(a.u1, p + 2);
                    V2b = (a.nt & 16) ? P2(a.u2, p + 2) : P2nt(a.u2, p + 2);
                    A3b = (a.nt & 16) ? P2(a.u3, p + 2) : P2nt(a.u3, p + 2);
                }
                flb = own ? (int)a.ok[((int64_t)(p + 2) * d3 + o) >> 6] : 0;
            }
            if (ht && KF_HV) {
                if (inseg) {
                    hq = H1(a.v, p + 1);
                    hzv = H1(a.z, p + 1);
                }
                if (more && KF_HHU)
                    hub = MH(KF_HU2 != 0, p + 2);
            }
            // ---- update of plane p + 1
            const double *u1n = u1s + s1 * U1T, *u2n = u2s + s1 * U2T;
            double2 Dn;
            const double2 Sa = SD(Ca);
            if (fla & 2) { // zero row sum: the diagonal from the six arms, in the order the assembly added them (+ sigma D)
                const double2 V1n = *reinterpret_cast<const double2 *>(u1n + u1o), V2n = *reinterpret_cast<const double2 *>(u2n + u2o);
                const double v1m0n = u1n[u1o - 1];
                const double2 V2mn = *reinterpret_cast<const double2 *>(u2n + u2o - U2S);
                double so = A3c.x + V2mn.x;
                so += v1m0n;
                so += A3n.x;
                so += V2n.x;
                so += V1n.x;
                Dn.x = -so + Sa.x;
                so = A3c.y + V2mn.y;
                so += V1n.x;
                so += A3n.y;
                so += V2n.y;
                so += V1n.y;
                Dn.y = -so + Sa.y;
            } else
                Dn = P2(a.dg, p + 1);
            const VRow ua = vrow(Xa.x, Za.x, Va.x, own ? Dn.x : 1.0, Sa.x, alpha), ub = vrow(Xa.y, Za.y, Va.y, own ? Dn.y : 1.0, Sa.y, alpha);
            const double2 Zn = make_double2(ua.zn, ub.zn);
            const double2 Zc = *reinterpret_cast<const double2 *>(zs + zb * ZT + zo);
            const double2 Pn = make_double2(A3c.x * Zc.x + Dn.x * Zn.x, A3c.y * Zc.y + Dn.y * Zn.y);
            if (MODE == 1 && vec_n && own) {
                ST2(a.znext, p + 1, Zn);
                if (XU)
                    ST2(a.xout, p + 1, make_double2(Xa.x + ax * Va.x, Xa.y + ax * Va.y));
            }
            if (MODE == 0 && vec_n && own) {
                double *xo = reinterpret_cast<double *>(const_cast<char *>(PB(a.xout, p + 1, 8)) + ob);
                if (a.nt & 8)
                    *reinterpret_cast<double2 *>(xo) = make_double2(ua.xn, ub.xn);
                else {
                    __builtin_nontemporal_store(ua.xn, xo);
                    __builtin_nontemporal_store(ub.xn, xo + 1);
                }
                if (a.nt & 1)
                    ST2nt(a.znext, p + 1, Zn);
                else
                    ST2(a.znext, p + 1, Zn);
                acc[0] += ua.r * (ua.mv * ua.r) + ub.r * (ub.mv * ub.r);
                acc[1] += ua.r * ua.r + ub.r * ub.r;
                acc[2] += ua.c * ua.zn + ub.c * ub.zn;
                acc[3] += ua.c * ua.c + ub.c * ub.c;
                acc[4] += ua.h * ua.h + ub.h * ub.h;
            }
            if (inseg)
                *reinterpret_cast<double2 *>(zs + (zb ^ 1) * ZT + zo) = Zn;
            // ---- product of plane p; stored as v' = -M^-1 (q' - sigma D z')
            {
                const double *zrow = zs + zb * ZT + zo, *u1c = u1s + s0 * U1T, *u2c = u2s + s0 * U2T;
                const double x1m0 = zrow[-1], x1p1 = zrow[2];
                const double2 x2m = *reinterpret_cast<const double2 *>(zrow - ZS), x2p = *reinterpret_cast<const double2 *>(zrow + ZS);
                const double2 V1c = *reinterpret_cast<const double2 *>(u1c + u1o), V2c = *reinterpret_cast<const double2 *>(u2c + u2o);
                const double v1m0 = u1c[u1o - 1];
                const double2 V2m = *reinterpret_cast<const double2 *>(u2c + u2o - U2S);
                double t0 = Pc.x, t1 = Pc.y;
                t0 += V2m.x * x2m.x;
                t1 += V2m.y * x2m.y;
                t0 += v1m0 * x1m0;
                t1 += V1c.x * Zc.x;
                t0 += V1c.x * Zc.y;
                t1 += V1c.y * x1p1;
                t0 += V2c.x * x2p.x;
                t1 += V2c.y * x2p.y;
                t0 += A3c.x * Zn.x;
                t1 += A3c.y * Zn.y;
                if (MODE == 1 && (flc & 1)) {
                    ST2nt(a.vnext, p, make_double2(-(Mc.x * t0), -(Mc.y * t1))); // w = -M^-1 q: the vector pass reads it once (z' = z + alpha w)
                    acc[5] += Zc.x * t0 + Zc.y * t1;
                }
                if (MODE == 0 && (flc & 1)) {
                    const double2 Sc = SD(Cc);
                    const double2 vn = make_double2(-(Mc.x * (t0 - Sc.x * Zc.x)), -(Mc.y * (t1 - Sc.y * Zc.y)));
                    if (a.nt & 2)
                        ST2nt(a.vnext, p, vn);
                    else
                        ST2(a.vnext, p, vn);
                    acc[5] += Zc.x * t0 + Zc.y * t1;
                }
            }
            // ---- halo z' of plane p + 1, U tiles of plane p + 2
            if (ht && inseg)
                zs[(zb ^ 1) * ZT + KF_HZ] = KF_HV ? hzv + alpha * hq : 0.0;
            if (more) {
                if (CODED) {
                    V1b = MV(Wb, 0);
                    V2b = MV(Wb, 1);
                    A3b = MV(Wb, 2);
                }
                *reinterpret_cast<double2 *>(u1s + s2 * U1T + u1o) = V1b;
                *reinterpret_cast<double2 *>(u2s + s2 * U2T + u2o) = V2b;
                if (ht && KF_HHU)
                    (KF_HU2 ? u2s + s2 * U2T : u1s + s2 * U1T)[KF_HU] = hub;
            }
            __syncthreads();
            Pc = Pn;
            Mc = make_double2(ua.mv, ub.mv);
            Cc = Ca;
            flc = fla;
            A3c = A3n;
            A3n = A3b;
            Xa = Xb;
            Va = Vb;
            Za = Zb;
            Ca = Cb;
            fla = flb;
            zb ^= 1;
            const int st = s0;
            s0 = s1;
            s1 = s2;
            s2 = st;
        }
#undef KF_HZ
#undef KF_HU
#undef KF_HV
#undef KF_HHU
#undef KF_HU2
    }
    // the assembled b's share of |rhs|^2 over its support: |h + b|^2 = h.h + b (2 h + b), h = sigma D x_out, with x_out of
    // these few rows formed again from x and z
    double sgather = 0.0;
    if (MODE == 0 && a.bm > 0) {
        __syncthreads();
        for (int64_t k = (int64_t)blockIdx.x * NT + tid; k < a.bm; k += (int64_t)gridDim.x * NT) {
            const int32_t i = a.bidx[k];
            const double bi = a.b[i];
            const double xn = a.x[i] + alpha * a.z[i];
            const double sd = a.code ? tab[a.code[i]] : tab[0];
            sgather += bi * (2.0 * (sd * xn) + bi);
        }
    }
    const int G = (int)gridDim.x;
    for (int k = MODE == 1 ? KF_NSUM - 1 : 0; k < KF_NSUM; k++) {
        const double t = kf_block_sum<NT>(acc[k], red);
        if (tid == 0)
            (k == 0 ? a.out.arz : k == 1 ? a.out.arr : k == 2 ? a.out.srz : k == 3 ? a.out.srr : k == 4 ? a.out.sbb : a.out.pq)[blockIdx.x] = t;
    }
    if (MODE == 0 && a.bm > 0) {
        const double t = kf_block_sum<NT>(sgather, red);
        if (tid == 0)
            a.out.sbb[G + blockIdx.x] = t;
    }
}

// ------------------------------------------------------------------ the same step / pass on CHUNKS of a plane (round 4)
// The 2-D tiles above pay for their column halos: a halo element left or right of a tile is 8 bytes out of a 128-byte line that
// belongs to the neighbouring tile, and the PMC counted 1.15 x the form's bytes at 464^3 with the matrix as codes.
// Lines are consecutive in memory (row = line x nz + column), so a CONTIGUOUS range of a plane's rows — C of them, whatever the
// line length — needs no column halo at all: the -1 / +1 neighbours of a row are its neighbours in the range (across a line end the
// matrix entry is zero), the -line / +line neighbours sit nz rows before / behind, and the halo of a chunk is the nz rows in
// front of it and the nz rows behind it, read as whole lines of consecutive doubles.  With the matrix as 16-bit codes the LDS
// that the U1 / U2 ring of doubles took holds a chunk of C = 6144 rows where the tile had 2048: halo rows 2 nz / C = 15 % of the
// own rows at 16 bytes each (z, v) instead of 12.5 % at 16 plus 1.6 % at 256 (column halos: whole lines for one value) —
// 2.4 instead of ~6.5 extra bytes per row of 51.
// What differs from the tile kernel besides the traversal: (1) every quantity that needs a row's diagonal — M^-1, the residual
// sums, the product's diagonal term — is formed when the row's plane is the CENTRE plane of a step (its codes are then in the
// ring slot being read), so the ring has two plane slots, not three, and holds the 16-bit words themselves; a row carries
// x_out - x from the step that formed x_out to the step that needs it.  (2) A thread owns NP pairs of rows, NT x 2 rows apart.
// (3) Slices whose diagonal is streamed (rows next to a Dirichlet cell) get it prefetched one plane step ahead.
// LDS: z' double-buffered 2 x (C + 2 nz) doubles, code ring 2 x (C + nz) words, tables: 139 KB at nz = 464, C = 6128.
// Results: the same arithmetic per row in the same order as the tile kernel; partial sums group differently (other blocks).
template <int NT, int NP, int MODE, int HR = 2> // HR: rounds in which the block's NT threads cover the 2 nz halo rows (2 nz <= HR x NT; 2: lines of up to 512 rows, 3: 768)
__global__ __launch_bounds__(NT) void fused_chunk_kernel(KfArgs a)
{
    extern __shared__ __align__(16) unsigned char kc_lds[];
    const int tid = (int)threadIdx.x;
    constexpr int NTB = 0; // the streaming hints (KfArgs::nt's bits) at their frozen values, as compile-time constants: with run-time tests the compiler split every
                           // 16-byte store into two 8-byte ones (round 5: 0.8965 -> 0.8702 ms per step at 464^3, profiles/r05_step_ab_coded_addressing.log)
    const int32_t nz = a.nz, d3 = a.d3, C = a.chunk;
    const int ZL = C + 2 * nz, WL = C + nz;
    double *zs = reinterpret_cast<double *>(kc_lds);          // 2 x ZL: [nz rows before | C own | nz rows behind]
    uint16_t *ws = reinterpret_cast<uint16_t *>(zs + 2 * ZL); // 2 x WL: [nz rows before | C own], slot = plane parity
    double *tab = reinterpret_cast<double *>(ws + 2 * WL); // sigma D by the low nibble of a row's code byte
    double *dtab = tab + FV_STORAGE_CODES;                 // the diagonal of a row whose code byte has a high nibble k > 0: dtab[k]
    double *mtab = dtab + FV_STORAGE_CODES;
    double *red = mtab + 3 * FV_MATRIX_CODES;
    PcgScalars *scal = a.scal;
    double alpha = 0.0;
    if (MODE == 1) { // K3's scalars, as in fused_step_kernel
        if (*reinterpret_cast<volatile int32_t *>(&scal->done))
            return;
        const double rzn = kf_reduce<NT>(a.in.arz, a.in.nvec, red);
        const double rrn = kf_reduce<NT>(a.in.arr, a.in.nvec, red);
        const bool converged = rrn <= scal->tol2;
        if (blockIdx.x == 0 && tid == 0) {
            scal->rz[(a.chain_index + 1) & 1] = rzn;
            scal->rr = rrn;
            scal->iters = a.chain_index + 1;
            if (a.hist && a.chain_index < a.hist_cap)
                a.hist[a.chain_index] = sqrt(rrn);
            if (converged)
                scal->done = 1;
        }
        if (converged)
            return;
        alpha = rzn / scal->rz[a.chain_index & 1];
    }
    if (MODE == 0) {
        if (!kf_step_prologue<NT>(a, red, alpha))
            return;
    }
    double ax3 = 0.0;
    if (MODE == 3) { // the whole PCG iteration in this launch (kf_ploop_prologue): alpha here plays beta's part, ax3 is the step length
        if (!kf_ploop_prologue<NT>(a, red, ax3, alpha))
            return;
    }
    // MODE 1: x += ax * p_old, the x-update of the previous iteration, which lags one pass behind (its direction is this pass's a.v)
    const bool XU = MODE == 0 || (MODE == 1 && a.xapply != 0) || (MODE == 3 && !a.first);
    const bool LW = MODE == 3 && !a.first; // MODE 3: w and the old direction are read (not in the first pass of a solve: the direction is z itself)
    double ax = MODE == 1 && a.xapply ? scal->alpha_last : (MODE == 3 ? ax3 : 0.0);
    if (MODE == 1 || MODE == 3) {
        const long long ab = __double_as_longlong(ax);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)ab), hi = __builtin_amdgcn_readfirstlane((uint32_t)((unsigned long long)ab >> 32));
        ax = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    }
    { // alpha is the same number in every lane: kept in scalar registers
        const long long ab = __double_as_longlong(alpha);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)ab), hi = __builtin_amdgcn_readfirstlane((uint32_t)((unsigned long long)ab >> 32));
        alpha = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    }
    if (tid < FV_STORAGE_CODES) {
        tab[tid] = a.sD.v[tid];
        dtab[tid] = a.kdiag.v[tid];
    }
    if (tid < 3 * FV_MATRIX_CODES)
        mtab[tid] = a.mt.v[tid];
    const double *T1 = mtab, *T2 = mtab + FV_MATRIX_CODES, *T3 = mtab + 2 * FV_MATRIX_CODES;
    const int xcd = (int)(blockIdx.x & 7);
    const int64_t items = (int64_t)a.tiles * a.nsegs, per_xcd = (items + 7) / 8;
    double acc[KF_NSUM] = {0, 0, 0, 0, 0, 0}, acc6 = 0.0;
    for (int64_t j = (int64_t)(blockIdx.x >> 3); j < per_xcd; j += (int64_t)(gridDim.x >> 3)) {
        const int64_t item = (int64_t)xcd * per_xcd + j;
        if (item >= items)
            break;
        const int32_t seg = (int32_t)(item / a.tiles), chunk = (int32_t)(item % a.tiles);
        const int32_t cs = chunk * C, Cl = (cs + C <= d3) ? C : d3 - cs; // the chunk's rows of a plane: [cs, cs + Cl)
        const int32_t p0 = a.pfirst + seg * a.seglen, p1 = (p0 + a.seglen < a.nplanes) ? p0 + a.seglen : a.nplanes;
        if (p0 >= p1 || Cl <= 0)
            continue;
        // pfirst = 0 (the end planes are centre planes like the others: their rows' diagonals come out of the table): no plane is
        // anybody's vector-part-only plane; plane -1 and plane P do not exist — their z' count as zero beside a zero matrix entry
        const bool vec_first = a.pfirst == 1 && p0 == 1, vec_last = a.pfirst == 1 && p1 == a.nplanes && a.nplanes == a.P - 1;
        const int64_t nrows8 = (int64_t)a.n * 8; // a halo row outside the vectors (in front of plane 0, behind plane P - 1) is not read
        bool own[NP];
        uint32_t ob[NP]; // byte offset of the pair's doubles inside a plane
#pragma unroll
        for (int k = 0; k < NP; k++) {
            const int lr = 2 * (tid + NT * k);
            own[k] = lr < Cl;
            ob[k] = (uint32_t)(cs + (own[k] ? lr : 0)) * 8u; // rows beyond the chunk read its first row: finite, never used
        }
        // halo roles: h < nz the row cs - nz + h (z' slot h, code slot h), else the row cs + Cl + (h - nz) (z' slot nz + Cl + h - nz)
        int32_t hb[HR];
        int hs[HR];
        bool hv[HR], hbefore[HR];
#pragma unroll
        for (int r = 0; r < HR; r++) {
            const int h = tid + NT * r;
            hv[r] = h < 2 * nz;
            hbefore[r] = h < nz;
            const int32_t grow = !hv[r] ? cs : (hbefore[r] ? cs - nz + h : cs + Cl + (h - nz));
            hb[r] = grow * 8;
            hs[r] = hbefore[r] ? h : nz + Cl + (h - nz);
        }
        auto PB = [&](const void *arr, int32_t pl, int esz) -> const char * {
            return reinterpret_cast<const char *>(arr) + (uint64_t)((int64_t)pl * d3) * (uint64_t)esz;
        };
        auto P2 = [&](const double *arr, int32_t pl, int k) -> double2 { return *reinterpret_cast<const double2 *>(PB(arr, pl, 8) + ob[k]); };
        auto P2nt = [&](const double *arr, int32_t pl, int k) -> double2 {
            const double *b = reinterpret_cast<const double *>(PB(arr, pl, 8) + ob[k]);
            return make_double2(__builtin_nontemporal_load(b), __builtin_nontemporal_load(b + 1));
        };
        auto C2 = [&](int32_t pl, int k) -> uint32_t { // the pair's code bytes: storage code | diagonal code << 4 (fv_chunk_codes)
            return (uint32_t) * reinterpret_cast<const uint16_t *>(PB(a.kcode, pl, 1) + (ob[k] >> 3));
        };
        auto SD = [&](uint32_t c) -> double2 { return make_double2(tab[c & 15u], tab[(c >> 8) & 15u]); };
        auto MW = [&](int32_t pl, int k) -> uint32_t { return *reinterpret_cast<const uint32_t *>(PB(a.mcode, pl, 2) + (ob[k] >> 2)); };
        auto HIN = [&](int32_t pl, int r) -> bool { // the halo row exists (always, unless the end planes are centre planes)
            const int64_t off = (int64_t)pl * d3 * 8 + (int64_t)hb[r];
            return off >= 0 && off < nrows8;
        };
        auto H1 = [&](const double *arr, int32_t pl, int r) -> double { return *reinterpret_cast<const double *>(PB(arr, pl, 8) + (int64_t)hb[r]); };
        auto HZ = [&](int32_t pl, int r) -> double { // the direction on a halo row, formed from its streams (the row exists)
            if (MODE == 3)
                return LW ? (H1(a.z, pl, r) + ax * H1(a.w, pl, r)) + alpha * H1(a.v, pl, r) : H1(a.z, pl, r);
            return H1(a.z, pl, r) + alpha * H1(a.v, pl, r);
        };
        auto HW = [&](int32_t pl, int r) -> uint16_t { return *reinterpret_cast<const uint16_t *>(PB(a.mcode, pl, 2) + (int64_t)(hb[r] >> 2)); };
        auto ST2 = [&](double *arr, int32_t pl, int k, double2 val) { *reinterpret_cast<double2 *>(const_cast<char *>(PB(arr, pl, 8)) + ob[k]) = val; };
        auto ST2nt = [&](double *arr, int32_t pl, int k, double2 val) {
            double *q = reinterpret_cast<double *>(const_cast<char *>(PB(arr, pl, 8)) + ob[k]);
            __builtin_nontemporal_store(val.x, q);
            __builtin_nontemporal_store(val.y, q + 1);
        };
        __syncthreads(); // the tables; the previous item's last LDS reads
        // ---------------- prologue: z' of plane p0 - 1 (registers), plane p0 (LDS, with halo and codes), batch p0 + 1 in flight
        double2 Zm[NP], DX[NP], Za[NP], Va[NP], Xa[NP], Qa[NP]; // (MODE 3: DX carries z' of the plane that is the centre plane next, Qa is the batch's w)
        uint32_t Mc[NP], Ma[NP], Wa[NP]; // M: the pair's code bytes | (centre plane only) U3 codes of the plane before << 18; W: its matrix words (bit 15: this kernel forms the row's product)
        int zb = p0 & 1;
#pragma unroll
        for (int k = 0; k < NP; k++) {
            const int32_t pm = p0 > 0 ? p0 - 1 : 0; // (p0 = 0: there is no plane before it; what is loaded here is not used)
            double2 zz = P2(a.z, pm, k);
            const double2 vv = (MODE == 3 && !LW) ? make_double2(0.0, 0.0) : P2(a.v, pm, k);
            if (LW) { // z' = z + ax w first, then the direction
                const double2 ww = P2(a.w, pm, k);
                zz = make_double2(zz.x + ax * ww.x, zz.y + ax * ww.y);
            }
            Zm[k] = p0 > 0 ? make_double2(zz.x + alpha * vv.x, zz.y + alpha * vv.y) : make_double2(0.0, 0.0);
            if (MODE == 1 && vec_first && own[k]) {
                ST2(a.znext, 0, k, Zm[k]);
                if (XU) {
                    const double2 xi = P2(a.x, 0, k);
                    ST2(a.xout, 0, k, make_double2(xi.x + ax * vv.x, xi.y + ax * vv.y));
                }
            }
            if (MODE == 0 && vec_first && own[k]) { // plane 0: its whole vector part, with the stored diagonal
                const double2 xi = P2(a.x, 0, k), dd = P2(a.dg, 0, k), ss = SD(C2(0, k));
                const VRow ra = vrow(xi.x, zz.x, vv.x, dd.x, ss.x, alpha), rb = vrow(xi.y, zz.y, vv.y, dd.y, ss.y, alpha);
                ST2(a.xout, 0, k, make_double2(ra.xn, rb.xn));
                ST2(a.znext, 0, k, Zm[k]);
                acc[0] += ra.r * (ra.mv * ra.r) + rb.r * (rb.mv * rb.r);
                acc[1] += ra.r * ra.r + rb.r * rb.r;
                acc[2] += ra.c * ra.zn + rb.c * rb.zn;
                acc[3] += ra.c * ra.c + rb.c * rb.c;
                acc[4] += ra.h * ra.h + rb.h * rb.h;
            }
            const uint32_t wm = MW(pm, k);
            const uint32_t u3m = ((wm >> 10) & 31u) | (((wm >> 26) & 31u) << 5);
            const double2 v0 = (MODE == 3 && !LW) ? make_double2(0.0, 0.0) : P2(a.v, p0, k);
            double2 z0 = P2(a.z, p0, k);
            if (LW) {
                const double2 w0 = P2(a.w, p0, k);
                z0 = make_double2(z0.x + ax * w0.x, z0.y + ax * w0.y);
            }
            const double2 Zc0 = make_double2(z0.x + alpha * v0.x, z0.y + alpha * v0.y);
            const uint32_t c0 = C2(p0, k);
            DX[k] = MODE == 3 ? z0 : make_double2(0.0, 0.0);
            if (LW && own[k]) { // (the first pass of a solve stores neither: z' is z, the direction is z)
                ST2(a.zout, p0, k, z0);
                ST2(a.znext, p0, k, Zc0);
                const double2 xi = P2(a.x, p0, k);
                ST2(a.xout, p0, k, make_double2(xi.x + ax * v0.x, xi.y + ax * v0.y));
            }
            if (MODE == 1 && own[k]) {
                ST2(a.znext, p0, k, Zc0);
                if (XU) {
                    const double2 xi = P2(a.x, p0, k);
                    ST2(a.xout, p0, k, make_double2(xi.x + ax * v0.x, xi.y + ax * v0.y));
                }
            }
            if (MODE == 0) {
                const double2 xi = P2(a.x, p0, k);
                const double2 xn = make_double2(xi.x + alpha * z0.x, xi.y + alpha * z0.y);
                DX[k] = make_double2(xn.x - xi.x, xn.y - xi.y);
                if (own[k]) {
                    const double2 ss = SD(c0);
                    ST2(a.xout, p0, k, xn);
                    ST2(a.znext, p0, k, Zc0);
                    const double hx = ss.x * xn.x, hy = ss.y * xn.y;
                    acc[4] += hx * hx + hy * hy;
                }
            }
            if (own[k]) {
                const int lr = 2 * (tid + NT * k);
                *reinterpret_cast<double2 *>(zs + zb * ZL + nz + lr) = Zc0;
                *reinterpret_cast<uint32_t *>(ws + (p0 & 1) * WL + nz + lr) = MW(p0, k);
            }
            Mc[k] = c0 | (u3m << 18);
        }
#pragma unroll
        for (int r = 0; r < HR; r++)
            if (hv[r]) {
                const bool in = HIN(p0, r);
                zs[zb * ZL + hs[r]] = in ? HZ(p0, r) : 0.0;
                if (hbefore[r])
                    ws[(p0 & 1) * WL + hs[r]] = in ? HW(p0, r) : (uint16_t)0;
            }
        const int32_t pn = p0 + 1 < a.P ? p0 + 1 : p0; // (a one-plane segment at the very end: nothing behind it)
#pragma unroll
        for (int k = 0; k < NP; k++) {
            Za[k] = P2(a.z, pn, k);
            Va[k] = MODE == 3 ? (LW ? P2(a.v, pn, k) : make_double2(0.0, 0.0)) : P2nt(a.v, pn, k);
            Qa[k] = LW ? P2(a.w, pn, k) : make_double2(0.0, 0.0);
            Xa[k] = XU ? P2nt(a.x, pn, k) : make_double2(0.0, 0.0);
            Wa[k] = MW(pn, k);
            Ma[k] = C2(pn, k);
        }
        double hz[HR], hq[HR];
        uint16_t hw[HR];
#pragma unroll
        for (int r = 0; r < HR; r++) {
            hz[r] = hq[r] = 0.0;
            hw[r] = 0;
            if (hv[r] && p0 + 1 < p1 && HIN(p0 + 1, r)) {
                hz[r] = H1(a.z, p0 + 1, r);
                if (MODE != 3 || LW)
                    hq[r] = H1(a.v, p0 + 1, r);
                if (LW)
                    hz[r] += ax * H1(a.w, p0 + 1, r); // (z' of the row: one register for both)
                if (hbefore[r])
                    hw[r] = HW(p0 + 1, r);
            }
        }
        __syncthreads();
        for (int32_t p = p0; p < p1; p++) {
#pragma unroll
            for (int k = 0; k < NP; k++)
                asm volatile("" : "+v"(ob[k])); // (opaque: one 32-bit offset per pair beside the scalar plane bases, no 64-bit lane addresses kept alive)
            const bool inseg = p + 1 < p1, more = p + 2 <= p1 && p + 2 < a.P;
            const bool nextp = p + 1 < a.P;          // there is a plane behind the centre plane
            const bool vec_n = inseg || vec_last;   // plane p + 1's vector part is ours
            const bool lastplane = vec_last && p + 1 == a.P - 1; // ... and it never becomes a centre plane: its diagonal terms now, from the stored diagonal
            const double *zc = zs + zb * ZL + nz;
            double *zn_ = zs + (zb ^ 1) * ZL + nz;
            const uint16_t *wc_ = ws + (p & 1) * WL + nz;
            uint16_t *wn_ = ws + ((p + 1) & 1) * WL + nz;
#pragma unroll
            for (int k = 0; k < NP; k++) {
                const int lr = own[k] ? 2 * (tid + NT * k) : 0; // (pairs beyond the chunk compute on its first rows; nothing of theirs is kept)
                // ---- (a) plane p + 1: z', x_out; its z' and codes into the other LDS slots
                if (LW)
                    Za[k] = make_double2(Za[k].x + ax * Qa[k].x, Za[k].y + ax * Qa[k].y); // z' = z + ax w
                const double2 Zn = nextp ? make_double2(Za[k].x + alpha * Va[k].x, Za[k].y + alpha * Va[k].y) : make_double2(0.0, 0.0);
                double2 dxn = MODE == 3 ? Za[k] : make_double2(0.0, 0.0);
                const uint32_t Mn = Ma[k];
                if (LW && inseg && own[k]) {
                    ST2(a.zout, p + 1, k, Za[k]);
                    ST2(a.xout, p + 1, k, make_double2(Xa[k].x + ax * Va[k].x, Xa[k].y + ax * Va[k].y));
                }
                if (MODE == 0) {
                    const double2 xn = make_double2(Xa[k].x + alpha * Za[k].x, Xa[k].y + alpha * Za[k].y);
                    dxn = make_double2(xn.x - Xa[k].x, xn.y - Xa[k].y);
                    if (vec_n && own[k]) {
                        if (NTB & 8) // (plain stores of x_out measure 3.6 % faster here than the streaming ones the tile kernel prefers:
                            ST2nt(a.xout, p + 1, k, xn); // 0.949 against 0.984 ms per step, profiles/r04_step_ab_chunk_hints.log)
                        else
                            ST2(a.xout, p + 1, k, xn);
                        const double2 sa = SD(Mn);
                        const double hx = sa.x * xn.x, hy = sa.y * xn.y;
                        acc[4] += hx * hx + hy * hy;
                        if (lastplane) {
                            const double2 dd = P2(a.dg, p + 1, k);
                            const double cx = dd.x * Zn.x, cy = dd.y * Zn.y;
                            const double rx = cx - sa.x * dxn.x, ry = cy - sa.y * dxn.y;
                            acc[0] += rx * ((1.0 / dd.x) * rx) + ry * ((1.0 / dd.y) * ry);
                            acc[1] += rx * rx + ry * ry;
                            acc[2] += cx * Zn.x + cy * Zn.y;
                            acc[3] += cx * cx + cy * cy;
                        }
                    }
                }
                if (MODE == 1 && XU && vec_n && own[k]) // (before (b) refills the registers: p_old of plane p + 1 is Va)
                    ST2(a.xout, p + 1, k, make_double2(Xa[k].x + ax * Va[k].x, Xa[k].y + ax * Va[k].y));
                if (vec_n && own[k] && (MODE != 3 || LW)) {
                    if (NTB & 1)
                        ST2nt(a.znext, p + 1, k, Zn);
                    else
                        ST2(a.znext, p + 1, k, Zn);
                }
                if (inseg && own[k]) {
                    *reinterpret_cast<double2 *>(zn_ + lr) = Zn;
                    *reinterpret_cast<uint32_t *>(wn_ + lr) = Wa[k];
                }
                // ---- (b) the batch of plane p + 2 into the registers (a) has just emptied
                if (more) {
                    Za[k] = P2(a.z, p + 2, k);
                    if (MODE == 3) { // (w and the old direction are read by the neighbouring chunks' halo rows as well: cacheable)
                        if (LW) {
                            Va[k] = P2(a.v, p + 2, k);
                            Qa[k] = P2(a.w, p + 2, k);
                        }
                    } else
                        Va[k] = (NTB & 4) ? P2(a.v, p + 2, k) : P2nt(a.v, p + 2, k);
                    if (XU)
                        Xa[k] = (NTB & 4) ? P2(a.x, p + 2, k) : P2nt(a.x, p + 2, k);
                    Wa[k] = MW(p + 2, k);
                    Ma[k] = C2(p + 2, k);
                }
                // ---- (c) centre plane p: its diagonal, the residual sums of its rows, its product
                const double *zrow = zc + lr;
                const uint16_t *wrow = wc_ + lr;
                const double2 Zc = *reinterpret_cast<const double2 *>(zrow);
                const double x1m0 = zrow[-1], x1p1 = zrow[2];
                const double2 x2m = *reinterpret_cast<const double2 *>(zrow - nz), x2p = *reinterpret_cast<const double2 *>(zrow + nz);
                const uint32_t wc = *reinterpret_cast<const uint32_t *>(wrow), wl = *reinterpret_cast<const uint32_t *>(wrow - nz);
                const uint32_t wm1 = wrow[-1];
                const uint32_t M = Mc[k];
                const double2 V1c = make_double2(T1[wc & 31u], T1[(wc >> 16) & 31u]), V2c = make_double2(T2[(wc >> 5) & 31u], T2[(wc >> 21) & 31u]);
                const double2 A3c = make_double2(T3[(wc >> 10) & 31u], T3[(wc >> 26) & 31u]);
                const double v1m0 = T1[wm1 & 31u];
                const double2 V2m = make_double2(T2[(wl >> 5) & 31u], T2[(wl >> 21) & 31u]);
                const double2 A3m = p > 0 ? make_double2(T3[(M >> 18) & 31u], T3[(M >> 23) & 31u]) : make_double2(0.0, 0.0);
                const double2 Sc = SD(M);
                // the diagonal: zero row sum — from the six arms, in the order the assembly added them (+ sigma D) —, or, on a row
                // whose stored diagonal is something else (a Dirichlet neighbour), out of the table by the row's code
                double2 d;
                {
                    double so = A3m.x + V2m.x;
                    so += v1m0;
                    so += A3c.x;
                    so += V2c.x;
                    so += V1c.x;
                    d.x = -so + Sc.x;
                    so = A3m.y + V2m.y;
                    so += V1c.x;
                    so += A3c.y;
                    so += V2c.y;
                    so += V1c.y;
                    d.y = -so + Sc.y;
                    if (M & 0xf0f0u) {
                        if (M & 0xf0u)
                            d.x = dtab[(M >> 4) & 15u];
                        if (M & 0xf000u)
                            d.y = dtab[(M >> 12) & 15u];
                    }
                }
                double t0 = A3m.x * Zm[k].x + d.x * Zc.x, t1 = A3m.y * Zm[k].y + d.y * Zc.y;
                t0 += V2m.x * x2m.x;
                t1 += V2m.y * x2m.y;
                t0 += v1m0 * x1m0;
                t1 += V1c.x * Zc.x;
                t0 += V1c.x * Zc.y;
                t1 += V1c.y * x1p1;
                t0 += V2c.x * x2p.x;
                t1 += V2c.y * x2p.y;
                t0 += A3c.x * Zn.x;
                t1 += A3c.y * Zn.y;
                if (MODE == 1 && own[k] && (wc & 0x8000u)) {
                    const double wx = 1.0 / d.x;
                    double wy = wx;
                    if (__double_as_longlong(d.y) != __double_as_longlong(d.x))
                        wy = 1.0 / d.y;
                    ST2nt(a.vnext, p, k, make_double2(-(wx * t0), -(wy * t1))); // w = -M^-1 q: the vector pass reads it once (z' = z + alpha w)
                    acc[5] += Zc.x * t0 + Zc.y * t1;
                }
                if (MODE == 3 && own[k]) { // (every row's product is this kernel's: the host runs the mode on whole regular boxes only)
                    const double wx = 1.0 / d.x;
                    double wy = wx;
                    if (__double_as_longlong(d.y) != __double_as_longlong(d.x))
                        wy = 1.0 / d.y;
                    const double2 zo = DX[k];
                    ST2(a.vnext, p, k, make_double2(-(wx * t0), -(wy * t1))); // w' = -M^-1 q' (the next launch's halo rows read it too: cacheable)
                    const double rx = d.x * zo.x, ry = d.y * zo.y; // the residual of the row
                    acc[0] += rx * zo.x + ry * zo.y;               // r.z and r.r of the iterate this launch formed: the next launch's base
                    acc[1] += rx * rx + ry * ry;
                    acc[2] += zo.x * t0 + zo.y * t1;               // z.q, q.M^-1 q, r.q, q.q: the next iterate's r.z and r.r as polynomials in the step length
                    acc[3] += t0 * (wx * t0) + t1 * (wy * t1);
                    acc[4] += rx * t0 + ry * t1;
                    acc6 += t0 * t0 + t1 * t1;
                    acc[5] += Zc.x * t0 + Zc.y * t1;
                }
                if (MODE == 0 && own[k]) {
                    // (the two rows of a pair almost always share their diagonal — one conductivity, interior rows —: one division then)
                    const double mx = 1.0 / d.x;
                    double my = mx;
                    if (__double_as_longlong(d.y) != __double_as_longlong(d.x))
                        my = 1.0 / d.y;
                    if (wc & 0x8000u) {
                        const double2 vn = make_double2(-(mx * (t0 - Sc.x * Zc.x)), -(my * (t1 - Sc.y * Zc.y)));
                        if (NTB & 2)
                            ST2nt(a.vnext, p, k, vn);
                        else
                            ST2(a.vnext, p, k, vn);
                        acc[5] += Zc.x * t0 + Zc.y * t1;
                    }
                    const double cx = d.x * Zc.x, cy = d.y * Zc.y;
                    const double rx = cx - Sc.x * DX[k].x, ry = cy - Sc.y * DX[k].y;
                    acc[0] += rx * (mx * rx) + ry * (my * ry);
                    acc[1] += rx * rx + ry * ry;
                    acc[2] += cx * Zc.x + cy * Zc.y;
                    acc[3] += cx * cx + cy * cy;
                }
                // ---- the pair's state for the next step; the streamed diagonal of plane p + 1 where it is not derived
                Zm[k] = Zc;
                DX[k] = dxn;
                Mc[k] = (Mn & 0xffffu) | ((((wc >> 10) & 31u) | (((wc >> 26) & 31u) << 5)) << 18);
            }
            // ---- halo of plane p + 1 into the other slots; the halo of plane p + 2 in flight
#pragma unroll
            for (int r = 0; r < HR; r++) {
                if (hv[r] && inseg) {
                    zs[(zb ^ 1) * ZL + hs[r]] = hz[r] + alpha * hq[r];
                    if (hbefore[r])
                        ws[((p + 1) & 1) * WL + hs[r]] = hw[r];
                }
                if (hv[r] && p + 2 < p1) {
                    const bool in = HIN(p + 2, r);
                    hz[r] = hq[r] = 0.0;
                    hw[r] = 0;
                    if (in) {
                    hz[r] = H1(a.z, p + 2, r);
                    if (MODE != 3 || LW)
                        hq[r] = H1(a.v, p + 2, r);
                    if (LW)
                        hz[r] += ax * H1(a.w, p + 2, r);
                    if (hbefore[r])
                        hw[r] = HW(p + 2, r);
                    }
                }
            }
            __syncthreads();
            zb ^= 1;
        }
    }
    double sgather = 0.0;
    if (MODE == 0 && a.bm > 0) { // the assembled b's share of |rhs|^2 over its support, as in fused_step_kernel
        __syncthreads();
        for (int64_t k = (int64_t)blockIdx.x * NT + tid; k < a.bm; k += (int64_t)gridDim.x * NT) {
            const int32_t i = a.bidx[k];
            const double bi = a.b[i];
            const double xn = a.x[i] + alpha * a.z[i];
            const double sd = a.code ? tab[a.code[i]] : tab[0];
            sgather += bi * (2.0 * (sd * xn) + bi);
        }
    }
    const int G = (int)gridDim.x;
    for (int k = MODE == 1 ? KF_NSUM - 1 : 0; k < KF_NSUM; k++) {
        const double t = kf_block_sum<NT>(acc[k], red);
        if (tid == 0)
            (k == 0 ? a.out.arz : k == 1 ? a.out.arr : k == 2 ? a.out.srz : k == 3 ? a.out.srr : k == 4 ? a.out.sbb : a.out.pq)[blockIdx.x] = t;
    }
    if (MODE == 3) {
        const double t = kf_block_sum<NT>(acc6, red);
        if (tid == 0)
            a.out.t2[blockIdx.x] = t;
    }
    if (MODE == 0 && a.bm > 0) {
        const double t = kf_block_sum<NT>(sgather, red);
        if (tid == 0)
            a.out.sbb[G + blockIdx.x] = t;
    }
}

// ------------------------------------------------------------------ chunks of a plane with the matrix as DOUBLES (round 5)
// A heterogeneous conductivity (one value per face: /root/reference/src/FiniteVolume.jl:75-108, every input the reference runs)
// has no matrix codes, and the 2-D tiles that served it paid 13 % of extra traffic for their column halos.  The same chunk
// traversal with the three upper diagonals streamed as doubles: what the 16-bit words did through the LDS ring is split three ways —
//   U2 (the +line diagonal; the -line arm of a row is U2 of the row nz in front of it, another wave's): an LDS ring of doubles
//      with two plane slots, [nz rows before | C own], written for plane p + 1 while plane p's is read, like z';
//   U1 (+1; the -1 arm is U1 of the row in front, the neighbouring LANE's second row): registers + a DPP wave shift, the wave's
//      edge element by a scalar load — no LDS;
//   U3 (+plane; the -plane arm is the own row's value one plane step earlier): registers only, carried from step to step.
// LDS per row: z' 2 x 8 + U2 2 x 8 = 32 B -> chunks of up to 4 400 rows at nz = 464 (49 per plane): halo rows 2 nz / C = 21 % of the
// own rows at 16 B (z, v) + nz U2 values = 4.2 extra bytes per row of 73 (the tiles: 9.5).  U1 / U3 of a plane are loaded one plane
// step ahead of their use (they are needed when the plane is the centre plane), z / v / x / U2 / the code byte two steps ahead (the
// plane's z' must be in LDS one step before).  The code byte (chunk_code_stream_kernel): storage code | bit 4 the row's diagonal is
// not minus the sum of its arms and is loaded from the stored diagonal (rows next to a Dirichlet cell: the first and the last line
// of a plane, the end planes) | bit 5 this traversal forms the row's product.  Same arithmetic per row in the same order as the
// tile kernel and the coded chunk kernel: the same bits per row; partial sums group differently.
__device__ __forceinline__ double kd_from_below(double v, double edge) // v of the lane below (DPP wave_shr:1), lane 0 gets `edge`
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(edge), __double2loint(v), 0x138, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(edge), __double2hiint(v), 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

template <int NT, int NP, int MODE, int PD, int PU, int HR = 2, bool BIG = false> // (HR as in fused_chunk_kernel; BIG: vectors of more than 4 GiB)
__global__ __launch_bounds__(NT) void fused_chunkd_kernel(KfArgs a)
{
    // Rolling prefetch: the loads a pair needs (its batch z, v, x, U2, code byte for step (a); U1, U3 and the edge element for (c))
    // are issued PD pairs ahead of their use, across the step boundary — not a whole plane step ahead: PD + 1 pairs' worth of
    // loaded-but-unused registers instead of NP pairs' worth, which is what lets 5 pairs of doubles fit 256 registers
    static_assert(PD <= NP && PU <= PD, "prefetch distance");
    extern __shared__ __align__(16) unsigned char kc_lds[];
    const int tid = (int)threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int32_t nz = a.nz, d3 = a.d3, C = a.chunk;
    const int ZL = C + 2 * nz, WL = C + nz;
    double *zs = reinterpret_cast<double *>(kc_lds); // 2 x ZL: [nz rows before | C own | nz rows behind]
    double *u2s = zs + 2 * ZL;                       // 2 x WL: [nz rows before | C own], slot = plane parity
    double *tab = u2s + 2 * WL;                      // sigma D by the low nibble of a row's code byte
    double *red = tab + FV_STORAGE_CODES;
    PcgScalars *scal = a.scal;
    double alpha = 0.0;
    if (MODE == 1) { // K3's scalars, as in fused_step_kernel
        if (*reinterpret_cast<volatile int32_t *>(&scal->done))
            return;
        const double rzn = kf_reduce<NT>(a.in.arz, a.in.nvec, red);
        const double rrn = kf_reduce<NT>(a.in.arr, a.in.nvec, red);
        const bool converged = rrn <= scal->tol2;
        if (blockIdx.x == 0 && tid == 0) {
            scal->rz[(a.chain_index + 1) & 1] = rzn;
            scal->rr = rrn;
            scal->iters = a.chain_index + 1;
            if (a.hist && a.chain_index < a.hist_cap)
                a.hist[a.chain_index] = sqrt(rrn);
            if (converged)
                scal->done = 1;
        }
        if (converged)
            return;
        alpha = rzn / scal->rz[a.chain_index & 1];
    }
    if (MODE == 0) {
        if (!kf_step_prologue<NT>(a, red, alpha))
            return;
    }
    double ax3 = 0.0;
    if (MODE == 3) { // the whole PCG iteration in this launch: see kf_ploop_prologue (alpha here plays beta's part, ax3 is the step length)
        if (!kf_ploop_prologue<NT>(a, red, ax3, alpha))
            return;
    }
    const bool XU = MODE == 0 || (MODE == 1 && a.xapply != 0) || (MODE == 3 && !a.first);
    const bool LW = MODE == 3 && !a.first; // w and the old direction are read (not in the first pass of a solve: alpha = beta = 0, the direction is z itself)
    double ax = MODE == 1 && a.xapply ? scal->alpha_last : (MODE == 3 ? ax3 : 0.0);
    if (MODE == 1 || MODE == 3) {
        const long long ab = __double_as_longlong(ax);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)ab), hi = __builtin_amdgcn_readfirstlane((uint32_t)((unsigned long long)ab >> 32));
        ax = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    }
    {
        const long long ab = __double_as_longlong(alpha);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)ab), hi = __builtin_amdgcn_readfirstlane((uint32_t)((unsigned long long)ab >> 32));
        alpha = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    }
    if (tid < FV_STORAGE_CODES)
        tab[tid] = a.sD.v[tid];
    const int xcd = (int)(blockIdx.x & 7);
    const int64_t items = (int64_t)a.tiles * a.nsegs, per_xcd = (items + 7) / 8;
    double acc[KF_NSUM] = {0, 0, 0, 0, 0, 0}, acc6 = 0.0;
    for (int64_t j = (int64_t)(blockIdx.x >> 3); j < per_xcd; j += (int64_t)(gridDim.x >> 3)) {
        const int64_t item = (int64_t)xcd * per_xcd + j;
        if (item >= items)
            break;
        const int32_t seg = (int32_t)(item / a.tiles), chunk = (int32_t)(item % a.tiles);
        const int32_t cs = chunk * C, Cl = (cs + C <= d3) ? C : d3 - cs; // the chunk's rows of a plane: [cs, cs + Cl)
        const int32_t p0 = a.pfirst + seg * a.seglen, p1 = (p0 + a.seglen < a.nplanes) ? p0 + a.seglen : a.nplanes;
        if (p0 >= p1 || Cl <= 0)
            continue;
        const bool vec_first = a.pfirst == 1 && p0 == 1, vec_last = a.pfirst == 1 && p1 == a.nplanes && a.nplanes == a.P - 1;
        const int64_t nrows8 = (int64_t)a.n * 8; // a halo row outside the vectors (in front of plane 0, behind plane P - 1) is not read
        bool own[NP];
        uint32_t ob[NP];  // byte offset of the pair's doubles inside a plane
        int32_t erow[NP]; // the row in front of the first row of this wave's pairs (wave-uniform; -1 in front of the plane: the arrays are padded)
#pragma unroll
        for (int k = 0; k < NP; k++) {
            const int lr = 2 * (tid + NT * k);
            own[k] = lr < Cl;
            ob[k] = (uint32_t)(cs + (own[k] ? lr : 0)) * 8u; // rows beyond the chunk read its first row: finite, never used
            const int lr0 = 2 * (wave * 64 + NT * k);
            erow[k] = cs + (lr0 < Cl ? lr0 : 0) - 1;
        }
        int32_t hb[HR];
        int hs[HR];
        bool hv[HR], hbefore[HR];
#pragma unroll
        for (int r = 0; r < HR; r++) {
            const int h = tid + NT * r;
            hv[r] = h < 2 * nz;
            hbefore[r] = h < nz;
            const int32_t grow = !hv[r] ? cs : (hbefore[r] ? cs - nz + h : cs + Cl + (h - nz));
            hb[r] = grow * 8;
            hs[r] = hbefore[r] ? h : nz + Cl + (h - nz);
        }
        // Addressing: array base (a kernel argument: scalar registers) + ONE 32-bit byte offset per access = the pair's offset inside a
        // plane + the plane's offset (a scalar) [+ the front padding of the symmetric arrays, whose rows -d3 .. -1 read zeros].  No
        // per-plane 64-bit bases: eleven arrays x three planes of them cost ~90 spilled scalar registers.  Every vector is < 4 GiB (the
        // host checks when it builds the symmetric copy); offsets are taken modulo 2^32, plane -1 of a padded array included.
        // BIG: ... counted from the plane in front of the item's segment (the 32-bit offsets span a segment, not the array — vectors of more
        // than 4 GiB are served as long as (planes of a segment + 3) x d3 x 8 bytes stay below 2^32, which kc_plan sees to; the rebased array
        // starts are scalar 64-bit sums, formed once per item; the one-launch iteration then spills 8 B per lane, hence a variant of its own)
        const uint32_t d3b = (uint32_t)d3 * 8u, fb = a.front * 8u;
        const int32_t pb = BIG ? (p0 > 0 ? p0 - 1 : 0) : 0; // (below 4 GiB per vector the offsets count from the arrays' starts, as they always did)
        const uint64_t rb8 = BIG ? (uint64_t)((int64_t)pb * d3) * 8u : 0u;
        const char *u1f = reinterpret_cast<const char *>(a.u1 - a.front) + rb8, *u2f = reinterpret_cast<const char *>(a.u2 - a.front) + rb8,
                   *u3f = reinterpret_cast<const char *>(a.u3 - a.front) + rb8, *dgf = reinterpret_cast<const char *>(a.dg - a.front) + rb8;
        const uint8_t *kcode_s = a.kcode + (rb8 >> 3);
        auto RB = [&](const void *arr) -> const char * { return reinterpret_cast<const char *>(arr) + rb8; };
        auto RBW = [&](void *arr) -> char * { return reinterpret_cast<char *>(arr) + rb8; };
        auto OFF = [&](int32_t pl) -> uint32_t { return (uint32_t)(pl - pb) * d3b; };
        auto LD2 = [&](const char *base, uint32_t off) -> double2 { return *reinterpret_cast<const double2 *>(base + off); };
        auto LD2nt = [&](const char *base, uint32_t off) -> double2 {
            const double *b = reinterpret_cast<const double *>(base + off);
            return make_double2(__builtin_nontemporal_load(b), __builtin_nontemporal_load(b + 1));
        };
        auto P2 = [&](const double *arr, int32_t pl, int k) -> double2 { return LD2(RB(arr), ob[k] + OFF(pl)); };
        auto P2nt = [&](const double *arr, int32_t pl, int k) -> double2 { return LD2nt(RB(arr), ob[k] + OFF(pl)); };
        auto PUL = [&](const char *basef, int32_t pl, int k) -> double2 { return LD2nt(basef, fb + ob[k] + OFF(pl)); }; // a padded array (its start + front)
        auto C2 = [&](int32_t pl, int k) -> uint32_t { return (uint32_t) * reinterpret_cast<const uint16_t *>(kcode_s + ((ob[k] + OFF(pl)) >> 3)); };
        auto SD = [&](uint32_t c) -> double2 { return make_double2(tab[c & 15u], tab[(c >> 8) & 15u]); };
        auto E1L = [&](int32_t pl, int k) -> double {
            // (a wave-uniform address, loaded as a vector all the same: through the constant address space — a scalar load — the launch measured
            // 1.8 % SLOWER: scalar loads return out of order, so the wait for it also waits for every LDS operation in flight)
            const uint32_t off = (uint32_t)__builtin_amdgcn_readfirstlane((int)(fb + OFF(pl) + (uint32_t)erow[k] * 8u));
            return *reinterpret_cast<const double *>(u1f + off);
        };
        auto HIN = [&](int32_t pl, int r) -> bool {
            const int64_t off = (int64_t)pl * d3 * 8 + (int64_t)hb[r];
            return off >= 0 && off < nrows8;
        };
        auto H1 = [&](const double *arr, int32_t pl, int r) -> double { return *reinterpret_cast<const double *>(RB(arr) + ((uint32_t)hb[r] + OFF(pl))); };
        auto HZ = [&](int32_t pl, int r) -> double { // the direction on a halo row, formed from its streams (the row exists)
            if (MODE == 3)
                return LW ? (H1(a.z, pl, r) + ax * H1(a.w, pl, r)) + alpha * H1(a.v, pl, r) : H1(a.z, pl, r);
            return H1(a.z, pl, r) + alpha * H1(a.v, pl, r);
        };
        auto HU = [&](int32_t pl, int r) -> double { return *reinterpret_cast<const double *>(u2f + (fb + (uint32_t)hb[r] + OFF(pl))); };
        auto ST2 = [&](double *arr, int32_t pl, int k, double2 val) { *reinterpret_cast<double2 *>(RBW(arr) + (ob[k] + OFF(pl))) = val; };
        auto ST2nt = [&](double *arr, int32_t pl, int k, double2 val) {
            double *q = reinterpret_cast<double *>(RBW(arr) + (ob[k] + OFF(pl)));
            __builtin_nontemporal_store(val.x, q);
            __builtin_nontemporal_store(val.y, q + 1);
        };
        __syncthreads(); // the table; the previous item's last LDS reads
        // ---------------- prologue: z' of plane p0 - 1 and its U3 (registers), plane p0 (z' and U2 in LDS with their halos, U1 / U3 in
        // registers), the batch of plane p0 + 1 in flight
        double2 Zm[NP], DX[NP], A3m[NP], A3c[NP], U1c[NP], Za[NP], Va[NP], Xa[NP], U2a[NP], Wa[NP]; // (MODE 3: DX carries z' of the plane that is the centre plane next, Wa is the batch's w)
        double E1[NP];
        uint32_t Mc[NP], Ma[NP];
        int zb = p0 & 1;
#pragma unroll
        for (int k = 0; k < NP; k++) {
            const int32_t pm = p0 > 0 ? p0 - 1 : 0; // (p0 = 0: there is no plane before it; what is loaded here is not used)
            double2 zz = P2(a.z, pm, k);
            const double2 vv = (MODE == 3 && !LW) ? make_double2(0.0, 0.0) : P2(a.v, pm, k);
            if (LW) { // z' = z + ax w first, then the direction
                const double2 ww = P2(a.w, pm, k);
                zz = make_double2(zz.x + ax * ww.x, zz.y + ax * ww.y);
            }
            Zm[k] = p0 > 0 ? make_double2(zz.x + alpha * vv.x, zz.y + alpha * vv.y) : make_double2(0.0, 0.0);
            if (MODE == 1 && vec_first && own[k]) {
                ST2(a.znext, 0, k, Zm[k]);
                if (XU) {
                    const double2 xi = P2(a.x, 0, k);
                    ST2(a.xout, 0, k, make_double2(xi.x + ax * vv.x, xi.y + ax * vv.y));
                }
            }
            if (MODE == 0 && vec_first && own[k]) { // plane 0: its whole vector part, with the stored diagonal
                const double2 xi = P2(a.x, 0, k), dd = LD2(dgf, fb + ob[k] + OFF(0)), ss = SD(C2(0, k));
                const VRow ra = vrow(xi.x, zz.x, vv.x, dd.x, ss.x, alpha), rb = vrow(xi.y, zz.y, vv.y, dd.y, ss.y, alpha);
                ST2(a.xout, 0, k, make_double2(ra.xn, rb.xn));
                ST2(a.znext, 0, k, Zm[k]);
                acc[0] += ra.r * (ra.mv * ra.r) + rb.r * (rb.mv * rb.r);
                acc[1] += ra.r * ra.r + rb.r * rb.r;
                acc[2] += ra.c * ra.zn + rb.c * rb.zn;
                acc[3] += ra.c * ra.c + rb.c * rb.c;
                acc[4] += ra.h * ra.h + rb.h * rb.h;
            }
            A3m[k] = PUL(u3f, p0 - 1, k); // (p0 = 0: the zero padding in front of the array)
            const double2 v0 = (MODE == 3 && !LW) ? make_double2(0.0, 0.0) : P2(a.v, p0, k);
            double2 z0 = P2(a.z, p0, k);
            if (LW) {
                const double2 w0 = P2(a.w, p0, k);
                z0 = make_double2(z0.x + ax * w0.x, z0.y + ax * w0.y);
            }
            const double2 Zc0 = make_double2(z0.x + alpha * v0.x, z0.y + alpha * v0.y);
            const uint32_t c0 = C2(p0, k);
            DX[k] = MODE == 3 ? z0 : make_double2(0.0, 0.0);
            if (LW && own[k]) { // (the first pass of a solve stores neither: z' is z, the direction is z)
                ST2(a.zout, p0, k, z0);
                ST2(a.znext, p0, k, Zc0);
                const double2 xi = P2(a.x, p0, k);
                ST2(a.xout, p0, k, make_double2(xi.x + ax * v0.x, xi.y + ax * v0.y));
            }
            if (MODE == 1 && own[k]) {
                ST2(a.znext, p0, k, Zc0);
                if (XU) {
                    const double2 xi = P2(a.x, p0, k);
                    ST2(a.xout, p0, k, make_double2(xi.x + ax * v0.x, xi.y + ax * v0.y));
                }
            }
            if (MODE == 0) {
                const double2 xi = P2(a.x, p0, k);
                const double2 xn = make_double2(xi.x + alpha * z0.x, xi.y + alpha * z0.y);
                DX[k] = make_double2(xn.x - xi.x, xn.y - xi.y);
                if (own[k]) {
                    const double2 ss = SD(c0);
                    ST2(a.xout, p0, k, xn);
                    ST2(a.znext, p0, k, Zc0);
                    const double hx = ss.x * xn.x, hy = ss.y * xn.y;
                    acc[4] += hx * hx + hy * hy;
                }
            }
            if (own[k]) {
                const int lr = 2 * (tid + NT * k);
                *reinterpret_cast<double2 *>(zs + zb * ZL + nz + lr) = Zc0;
                *reinterpret_cast<double2 *>(u2s + (p0 & 1) * WL + nz + lr) = PUL(u2f, p0, k);
            }
            if (k < PU) { // (the other pairs' by the rolling prefetch of the first step)
                U1c[k] = PUL(u1f, p0, k);
                A3c[k] = PUL(u3f, p0, k);
                E1[k] = E1L(p0, k);
            }
            Mc[k] = c0;
        }
#pragma unroll
        for (int r = 0; r < HR; r++)
            if (hv[r]) {
                zs[zb * ZL + hs[r]] = HIN(p0, r) ? HZ(p0, r) : 0.0;
                if (hbefore[r])
                    u2s[(p0 & 1) * WL + hs[r]] = HU(p0, r); // (in front of plane 0: the array's zero padding)
            }
        const int32_t pn = p0 + 1 < a.P ? p0 + 1 : p0; // (a one-plane segment at the very end: nothing behind it)
        auto issue_batch = [&](int k, int32_t pl) {
            Za[k] = P2(a.z, pl, k);
            if (MODE == 3) { // (w and the old direction are read by the neighbouring chunks' halo rows as well: cacheable)
                if (LW) {
                    Va[k] = P2(a.v, pl, k);
                    Wa[k] = P2(a.w, pl, k);
                }
            } else
                Va[k] = P2nt(a.v, pl, k); // (read once: streaming loads; z stays cacheable for the halo rows of the neighbouring chunks)
            if (XU)
                Xa[k] = P2nt(a.x, pl, k);
            U2a[k] = PUL(u2f, pl, k);
            Ma[k] = C2(pl, k);
        };
        auto issue_u13 = [&](int k, int32_t pl) {
            A3c[k] = PUL(u3f, pl, k);
            U1c[k] = PUL(u1f, pl, k);
            E1[k] = E1L(pl, k);
        };
#pragma unroll
        for (int k = 0; k < NP; k++) {
            Xa[k] = Va[k] = Wa[k] = make_double2(0.0, 0.0);
            if (k < PD)
                issue_batch(k, pn);
        }
        double hz[HR], hq[HR], hu[HR];
        constexpr int KH = NP >= 3 ? NP - 3 : 0; // the pair behind whose section a step issues the halo loads of the plane after next
        auto issue_halo = [&](int32_t pl) { // consumed at the end of the step in which plane pl - 1 is the centre plane
#pragma unroll
            for (int r = 0; r < HR; r++) {
                hz[r] = hq[r] = hu[r] = 0.0;
                if (hv[r] && pl < p1) {
                    if (HIN(pl, r)) {
                        hz[r] = H1(a.z, pl, r);
                        if (MODE != 3 || LW)
                            hq[r] = H1(a.v, pl, r);
                        if (LW)
                            hz[r] += ax * H1(a.w, pl, r); // (z' of the row: one register for both)
                    }
                    if (hbefore[r])
                        hu[r] = HU(pl, r);
                }
            }
        };
        issue_halo(p0 + 1);
        __syncthreads();
        for (int32_t p = p0; p < p1; p++) {
#pragma unroll
            for (int k = 0; k < NP; k++)
                asm volatile("" : "+v"(ob[k])); // (opaque: one 32-bit offset per pair beside the scalar plane bases, no 64-bit lane addresses kept alive)
            const bool inseg = p + 1 < p1, more = p + 2 <= p1 && p + 2 < a.P;
            const bool nextp = p + 1 < a.P;        // there is a plane behind the centre plane
            const bool vec_n = inseg || vec_last; // plane p + 1's vector part is ours
            const bool lastplane = vec_last && p + 1 == a.P - 1; // ... and it never becomes a centre plane: its diagonal terms now, from the stored diagonal
            const double *zc = zs + zb * ZL + nz;
            double *zn_ = zs + (zb ^ 1) * ZL + nz;
            const double *u2c = u2s + (p & 1) * WL + nz;
            double *u2n = u2s + ((p + 1) & 1) * WL + nz;
#pragma unroll
            for (int k = 0; k < NP; k++) {
                const int lr = own[k] ? 2 * (tid + NT * k) : 0; // (pairs beyond the chunk compute on its first rows; nothing of theirs is kept)
                // ---- (a) plane p + 1: z', x_out; its z' and U2 into the other LDS slots
                if (LW)
                    Za[k] = make_double2(Za[k].x + ax * Wa[k].x, Za[k].y + ax * Wa[k].y); // z' = z + ax w
                const double2 Zn = nextp ? make_double2(Za[k].x + alpha * Va[k].x, Za[k].y + alpha * Va[k].y) : make_double2(0.0, 0.0);
                double2 dxn = MODE == 3 ? Za[k] : make_double2(0.0, 0.0);
                const uint32_t Mn = Ma[k];
                if (LW && inseg && own[k]) {
                    ST2(a.zout, p + 1, k, Za[k]);
                    ST2(a.xout, p + 1, k, make_double2(Xa[k].x + ax * Va[k].x, Xa[k].y + ax * Va[k].y));
                }
                if (MODE == 0) {
                    const double2 xn = make_double2(Xa[k].x + alpha * Za[k].x, Xa[k].y + alpha * Za[k].y);
                    dxn = make_double2(xn.x - Xa[k].x, xn.y - Xa[k].y);
                    if (vec_n && own[k]) {
                        ST2(a.xout, p + 1, k, xn);
                        const double2 sa = SD(Mn);
                        const double hx = sa.x * xn.x, hy = sa.y * xn.y;
                        acc[4] += hx * hx + hy * hy;
                        if (lastplane) {
                            const double2 dd = LD2(dgf, fb + ob[k] + OFF(p + 1));
                            const double cx = dd.x * Zn.x, cy = dd.y * Zn.y;
                            const double rx = cx - sa.x * dxn.x, ry = cy - sa.y * dxn.y;
                            acc[0] += rx * ((1.0 / dd.x) * rx) + ry * ((1.0 / dd.y) * ry);
                            acc[1] += rx * rx + ry * ry;
                            acc[2] += cx * Zn.x + cy * Zn.y;
                            acc[3] += cx * cx + cy * cy;
                        }
                    }
                }
                if (MODE == 1 && XU && vec_n && own[k]) // (p_old of plane p + 1 is Va)
                    ST2(a.xout, p + 1, k, make_double2(Xa[k].x + ax * Va[k].x, Xa[k].y + ax * Va[k].y));
                if (vec_n && own[k] && (MODE != 3 || LW))
                    ST2(a.znext, p + 1, k, Zn);
                if (inseg && own[k]) {
                    *reinterpret_cast<double2 *>(zn_ + lr) = Zn;
                    *reinterpret_cast<double2 *>(u2n + lr) = U2a[k];
                }
                // ---- (c) centre plane p: its diagonal, the residual sums of its rows, its product
                const double *zrow = zc + lr;
                const double2 Zc = *reinterpret_cast<const double2 *>(zrow);
                const double x1m0 = zrow[-1], x1p1 = zrow[2];
                const double2 x2m = *reinterpret_cast<const double2 *>(zrow - nz), x2p = *reinterpret_cast<const double2 *>(zrow + nz);
                const double2 V2c = *reinterpret_cast<const double2 *>(u2c + lr), V2m = *reinterpret_cast<const double2 *>(u2c + lr - nz);
                const double2 V1c = U1c[k], A3k = A3c[k], A3p = A3m[k];
                const double v1m0 = kd_from_below(V1c.y, E1[k]); // (outside every divergent branch: the DPP shift reads the neighbouring lane)
                const uint32_t M = Mc[k];
                const double2 Sc = SD(M);
                // the diagonal: zero row sum — from the six arms, in the order the assembly added them (+ sigma D) —, or, on a row
                // whose stored diagonal is something else (a Dirichlet neighbour), the stored one
                double2 d;
                {
                    double so = A3p.x + V2m.x;
                    so += v1m0;
                    so += A3k.x;
                    so += V2c.x;
                    so += V1c.x;
                    d.x = -so + Sc.x;
                    so = A3p.y + V2m.y;
                    so += V1c.x;
                    so += A3k.y;
                    so += V2c.y;
                    so += V1c.y;
                    d.y = -so + Sc.y;
                    if (M & 0x1010u) {
                        const double2 dd = LD2(dgf, fb + ob[k] + OFF(p));
                        if (M & 0x10u)
                            d.x = dd.x;
                        if (M & 0x1000u)
                            d.y = dd.y;
                    }
                }
                double t0 = A3p.x * Zm[k].x + d.x * Zc.x, t1 = A3p.y * Zm[k].y + d.y * Zc.y;
                t0 += V2m.x * x2m.x;
                t1 += V2m.y * x2m.y;
                t0 += v1m0 * x1m0;
                t1 += V1c.x * Zc.x;
                t0 += V1c.x * Zc.y;
                t1 += V1c.y * x1p1;
                t0 += V2c.x * x2p.x;
                t1 += V2c.y * x2p.y;
                t0 += A3k.x * Zn.x;
                t1 += A3k.y * Zn.y;
                if (MODE == 1 && own[k] && (M & 0x20u)) {
                    const double wx = 1.0 / d.x, wy = 1.0 / d.y;
                    ST2nt(a.vnext, p, k, make_double2(-(wx * t0), -(wy * t1))); // w = -M^-1 q: the vector pass reads it once (z' = z + alpha w)
                    acc[5] += Zc.x * t0 + Zc.y * t1;
                }
                if (MODE == 3 && own[k]) { // (every row's product is this kernel's: the host runs the mode on whole regular boxes only)
                    const double wx = 1.0 / d.x, wy = 1.0 / d.y;
                    const double2 zo = DX[k];
                    ST2(a.vnext, p, k, make_double2(-(wx * t0), -(wy * t1))); // w' = -M^-1 q' (the next launch's halo rows read it too: cacheable)
                    const double rx = d.x * zo.x, ry = d.y * zo.y; // the residual of the row
                    acc[0] += rx * zo.x + ry * zo.y;               // r.z and r.r of the iterate this launch formed: the next launch's base
                    acc[1] += rx * rx + ry * ry;
                    acc[2] += zo.x * t0 + zo.y * t1;               // z.q, q.M^-1 q, r.q, q.q: the next iterate's r.z and r.r as polynomials in the step length
                    acc[3] += t0 * (wx * t0) + t1 * (wy * t1);
                    acc[4] += rx * t0 + ry * t1;
                    acc6 += t0 * t0 + t1 * t1;
                    acc[5] += Zc.x * t0 + Zc.y * t1;
                }
                if (MODE == 0 && own[k]) {
                    const double mx = 1.0 / d.x, my = 1.0 / d.y;
                    if (M & 0x20u) {
                        const double2 vn = make_double2(-(mx * (t0 - Sc.x * Zc.x)), -(my * (t1 - Sc.y * Zc.y)));
                        ST2(a.vnext, p, k, vn);
                        acc[5] += Zc.x * t0 + Zc.y * t1;
                    }
                    const double cx = d.x * Zc.x, cy = d.y * Zc.y;
                    const double rx = cx - Sc.x * DX[k].x, ry = cy - Sc.y * DX[k].y;
                    acc[0] += rx * (mx * rx) + ry * (my * ry);
                    acc[1] += rx * rx + ry * ry;
                    acc[2] += cx * Zc.x + cy * Zc.y;
                    acc[3] += cx * cx + cy * cy;
                }
                // ---- the pair's state for the next step
                Zm[k] = Zc;
                DX[k] = dxn;
                Mc[k] = Mn;
                A3m[k] = A3k;
                // ---- (b) the loads of the pair PD positions ahead: U1 / U3 of its centre plane, the batch of the plane behind that.  (Issued BEFORE this
                // pair's product instead — so that they travel during (c); any wait drains the wave's whole queue once stores are in flight —: no
                // difference at 464^3, 1.274 against 1.276 / 1.253 ms per step in separate processes, and 28 B per lane of spills.)
                if (k + PU < NP)
                    issue_u13(k + PU, p); // ... a pair of this same step
                else if (inseg)
                    issue_u13(k + PU - NP, p + 1); // ... of the next step
                if (k + PD < NP) {
                    if (nextp)
                        issue_batch(k + PD, p + 1);
                } else if (more)
                    issue_batch(k + PD - NP, p + 2);
                if (k == KH) { // the halo of plane p + 1 into the other slots (nobody reads those before the barrier), then the halo of plane p + 2 in flight
#pragma unroll
                    for (int r = 0; r < HR; r++)
                        if (hv[r] && inseg) {
                            zs[(zb ^ 1) * ZL + hs[r]] = hz[r] + alpha * hq[r];
                            if (hbefore[r])
                                u2s[((p + 1) & 1) * WL + hs[r]] = hu[r];
                        }
                    issue_halo(p + 2);
                }
                __builtin_amdgcn_sched_barrier(0); // (the scheduler must not hoist these loads further up: their registers are the point)
            }
            __syncthreads();
            zb ^= 1;
        }
    }
    double sgather = 0.0;
    if (MODE == 0 && a.bm > 0) { // the assembled b's share of |rhs|^2 over its support, as in fused_step_kernel
        __syncthreads();
        for (int64_t k = (int64_t)blockIdx.x * NT + tid; k < a.bm; k += (int64_t)gridDim.x * NT) {
            const int32_t i = a.bidx[k];
            const double bi = a.b[i];
            const double xn = a.x[i] + alpha * a.z[i];
            const double sd = a.code ? tab[a.code[i]] : tab[0];
            sgather += bi * (2.0 * (sd * xn) + bi);
        }
    }
    const int G = (int)gridDim.x;
    for (int k = MODE == 1 ? KF_NSUM - 1 : 0; k < KF_NSUM; k++) {
        const double t = kf_block_sum<NT>(acc[k], red);
        if (tid == 0)
            (k == 0 ? a.out.arz : k == 1 ? a.out.arr : k == 2 ? a.out.srz : k == 3 ? a.out.srr : k == 4 ? a.out.sbb : a.out.pq)[blockIdx.x] = t;
    }
    if (MODE == 3) {
        const double t = kf_block_sum<NT>(acc6, red);
        if (tid == 0)
            a.out.t2[blockIdx.x] = t;
    }
    if (MODE == 0 && a.bm > 0) {
        const double t = kf_block_sum<NT>(sgather, red);
        if (tid == 0)
            a.out.sbb[G + blockIdx.x] = t;
    }
}

// ------------------------------------------------------------------ the fused step on the SELL form (irregular meshes)
// The same step (kf_step_prologue, vrow, the same six sums) for operators stored as SELL-64 with 16-bit column offsets
// (fv_spmv.hip): one wave per 64-row group, a lane per row.  A neighbour's z' is z + alpha v of that row — two gathers that hit the
// cache lines the re-numbering keeps close — so a row streams x, z, v, its matrix blocks and sigma D (a code byte where the
// storage term takes few values) in and x_out, z', v' out: 74 + 56 = 130 B per row against 89 + 49 for the SpMV + K2S pair.
struct KsArgs {
    int64_t count;
    const int32_t *list, *ptr;
    const uint8_t *w8;
    const double *sv;
    const int16_t *sd;
    const double *D; // the storage term as a stream (null: codes / one value through a.code and a.sD)
    double sigma;
    // the groups that stay with the CSR kernel: their vector part is this kernel's, their products follow (fv_spmv_sell_rest)
    const int32_t *rest;
    int64_t nrest;
    const double *diagA; // ... whose shifted diagonal is diagA + sigma D, as the Jacobi preconditioner formed it
};

__global__ __launch_bounds__(FV_BLOCK) void fused_sell_step_kernel(KfArgs a, KsArgs s)
{
    constexpr int NT = FV_BLOCK, WPB = FV_BLOCK / 64;
    __shared__ double red[NT / 64];
    __shared__ double tab[FV_STORAGE_CODES];
    const int tid = (int)threadIdx.x;
    double alpha = 0.0;
    if (!kf_step_prologue<NT>(a, red, alpha))
        return;
    if (tid < FV_STORAGE_CODES)
        tab[tid] = a.sD.v[tid];
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    const int64_t per_xcd = (s.count + 7) >> 3;
    const int64_t pstride = (int64_t)(gridDim.x >> 3) * WPB;
    const int64_t xbase = (int64_t)(blockIdx.x & 7) * per_xcd;
    const int64_t xend = (xbase + per_xcd < s.count) ? xbase + per_xcd : s.count;
    double acc[KF_NSUM] = {0, 0, 0, 0, 0, 0};
    for (int64_t pos = xbase + (int64_t)(blockIdx.x >> 3) * WPB + wave; pos < xend; pos += pstride) {
        const int64_t g = s.list[pos], row = (g << 6) + lane;
        const int w = s.w8[g];
        const bool live = row < a.n;
        const int64_t r0 = live ? row : 0;
        const double *mv = s.sv + (int64_t)s.ptr[g] * 64 + lane;
        const int16_t *md = s.sd + (int64_t)s.ptr[g] * 64 + lane;
        const double *zr = a.z + r0, *vr = a.v + r0;
        const double xi = __builtin_nontemporal_load(a.x + r0), zi = zr[0], vi = vr[0];
        const double sD = s.D ? s.sigma * __builtin_nontemporal_load(s.D + r0) : (a.code ? tab[a.code[r0]] : tab[0]);
        const double d = live ? __builtin_nontemporal_load(mv) : 1.0; // the (shifted) diagonal first
        double sum = 0.0;
        int k = 1;
        // (three entries at a time, loads first: written out by hand — a generic predicated loop over the entries, also with two or
        // four groups per wave and pass, measured 0.17-0.19 ms per step on the 5M-cell mesh against 0.136; six at a time: no gain)
        for (; k + 3 <= w; k += 3) {
            const double a0 = __builtin_nontemporal_load(mv + (k + 0) * 64), a1 = __builtin_nontemporal_load(mv + (k + 1) * 64);
            const double a2 = __builtin_nontemporal_load(mv + (k + 2) * 64);
            const int d0 = __builtin_nontemporal_load(md + (k + 0) * 64), d1 = __builtin_nontemporal_load(md + (k + 1) * 64);
            const int d2 = __builtin_nontemporal_load(md + (k + 2) * 64);
            const double z0 = zr[d0] + alpha * vr[d0], z1 = zr[d1] + alpha * vr[d1], z2 = zr[d2] + alpha * vr[d2];
            sum += a0 * z0;
            sum += a1 * z1;
            sum += a2 * z2;
        }
        for (; k < w; k++) {
            const int dk = __builtin_nontemporal_load(md + k * 64);
            sum += __builtin_nontemporal_load(mv + k * 64) * (zr[dk] + alpha * vr[dk]);
        }
        if (live) {
            const VRow o = vrow(xi, zi, vi, d, sD, alpha);
            const double q = d * o.zn + sum;
            __builtin_nontemporal_store(o.xn, a.xout + row);
            const double vn = -(o.mv * (q - sD * o.zn));
            if (a.nt & 1)
                __builtin_nontemporal_store(o.zn, a.znext + row);
            else
                a.znext[row] = o.zn;
            if (a.nt & 2)
                __builtin_nontemporal_store(vn, a.vnext + row);
            else
                a.vnext[row] = vn;
            acc[0] += o.r * (o.mv * o.r);
            acc[1] += o.r * o.r;
            acc[2] += o.c * o.zn;
            acc[3] += o.c * o.c;
            acc[4] += o.h * o.h;
            acc[5] += o.zn * q;
        }
    }
    for (int64_t pos = (int64_t)blockIdx.x * WPB + wave; pos < s.nrest; pos += (int64_t)gridDim.x * WPB) {
        const int64_t row = ((int64_t)s.rest[pos] << 6) + lane;
        if (row < a.n) {
            const double sD = a.D[row] * s.sigma;
            const VRow o = vrow(a.x[row], a.z[row], a.v[row], s.diagA[row] + sD, sD, alpha);
            a.xout[row] = o.xn;
            a.znext[row] = o.zn;
            acc[0] += o.r * (o.mv * o.r);
            acc[1] += o.r * o.r;
            acc[2] += o.c * o.zn;
            acc[3] += o.c * o.c;
            acc[4] += o.h * o.h;
        }
    }
    // the assembled b's share of |rhs|^2 over its support (as in the tiled kernel)
    double sgather = 0.0;
    if (a.bm > 0) {
        for (int64_t k = (int64_t)blockIdx.x * NT + tid; k < a.bm; k += (int64_t)gridDim.x * NT) {
            const int32_t i = a.bidx[k];
            const double bi = a.b[i];
            const double xn = a.x[i] + alpha * a.z[i];
            const double sd = s.D ? s.sigma * s.D[i] : (a.code ? tab[a.code[i]] : tab[0]);
            sgather += bi * (2.0 * (sd * xn) + bi);
        }
    }
    const int G = (int)gridDim.x;
    for (int k = 0; k < KF_NSUM; k++) {
        const double t = kf_block_sum<NT>(acc[k], red);
        if (tid == 0)
            (k == 0 ? a.out.arz : k == 1 ? a.out.arr : k == 2 ? a.out.srz : k == 3 ? a.out.srr : k == 4 ? a.out.sbb : a.out.pq)[blockIdx.x] = t;
    }
    if (a.bm > 0) {
        const double t = kf_block_sum<NT>(sgather, red);
        if (tid == 0)
            a.out.sbb[G + blockIdx.x] = t;
    }
}

// v = -M^-1 (q - sigma D z): from a product formed the classic way (entry into the fused regime, the slices the fused
// kernel leaves to the slice-by-slice launch)
__global__ __launch_bounds__(FV_BLOCK) void q_to_v_kernel(int64_t n, const double *__restrict__ q, const double *__restrict__ z,
                                                           const double *__restrict__ minv, const double *__restrict__ D, double sigma,
                                                           double *__restrict__ v)
{
    for (int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x; i < n; i += vec_stride())
        v[i] = -(minv[i] * (q[i] - (sigma * D[i]) * z[i]));
}
__global__ __launch_bounds__(FV_BLOCK) void q_to_v_slices_kernel(int64_t n, int64_t nsl, const int32_t *__restrict__ slices, const double *__restrict__ z,
                                                                  const double *__restrict__ minv, const double *__restrict__ D, double sigma,
                                                                  double *__restrict__ v)
{
    const int64_t k = ((int64_t)blockIdx.x * FV_BLOCK + threadIdx.x) >> 6;
    if (k >= nsl)
        return;
    const int64_t i = ((int64_t)slices[k] << 6) + (threadIdx.x & 63);
    if (i < n)
        v[i] = -(minv[i] * (v[i] - (sigma * D[i]) * z[i])); // (in place: the slice-by-slice launch left q there)
}

// Row blocks: z' = z + alpha v of the rows the neighbouring ranks need, straight into the send buffer BEFORE the fused
// launch, so that the halo exchange travels while that launch runs (a v-form halo row costs two loads and one FMA here
// as well).  Every thread takes the launch's own decisions from the same all-reduced sums (fused_step_kernel's prologue
// with a.red): whenever the launch will stop short of its pass, nothing is packed (the exchange then ships stale, finite
// values nobody uses).
__global__ __launch_bounds__(FV_BLOCK) void fused_pack_kernel(int64_t nsend, const int32_t *__restrict__ idx, const double *__restrict__ z,
                                                               const double *__restrict__ v, const double *__restrict__ red,
                                                               const PcgScalars *__restrict__ scal, int mode, int chain_index, int force_prev_unconverged,
                                                               double rtol, double *__restrict__ buf)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i >= nsend)
        return;
    const int d0 = *reinterpret_cast<const volatile int32_t *>(&scal->done);
    if (d0 == 2 || d0 == 3)
        return;
    double rz0;
    bool zero_iteration;
    if (mode == 1) {
        const bool converged = red[2] <= scal->tol2x[(chain_index - 1) & 1] && !force_prev_unconverged;
        if (!converged)
            return;
        rz0 = red[3];
        const double tol2 = rtol * rtol * red[5];
        zero_iteration = red[4] <= tol2;
    } else {
        rz0 = scal->rz[0];
        zero_iteration = d0 == 1;
    }
    double alpha = 0.0;
    if (!zero_iteration) {
        const double pq = red[0];
        if (!(pq > 0.0))
            return;
        alpha = rz0 / pq;
    }
    const int32_t r = idx[i];
    buf[i] = z[r] + alpha * v[r];
}

} // namespace

// Can the chained step of this problem run fused?  (The symmetric tiled form serves the operator — the classic K1 of the
// step that enters the regime has just established that —, the shift is folded, D comes as codes, M^-1 > 0.)
// ... on the SELL form (irregular meshes): the classic K1 has just run in that form with the shift folded
static bool kf_sell(const fv_problem *p) { return p->sell_state == 1 && p->last_form == FV_SPMV_SELL; }
bool fv_fused_streams_storage(fv_problem *p) { return kf_sell(p); } // (the SELL kernel takes sigma D as a stream where there are no codes)

bool fv_fused_applicable(fv_problem *p, double sigma)
{
    if (g_fused && g_fused_sell && kf_sell(p))
        return !p->dist && p->nhalo == 0 && sigma != 0.0 && p->sell_vals_epoch == p->assemble_epoch && p->sell_tag == sigma &&
               fv_sell_grid(p) <= FV_FUSED_PARTS / 2;
    if (!g_fused || (!p->dist && p->nhalo > 0) || p->sym_state != 1 || p->last_form != FV_SPMV_SYM_TILE)
        return false;
    if (p->sym_epoch != p->assemble_epoch || p->sym_tag != sigma || sigma == 0.0)
        return false;
    if (p->dcode_n <= 0 || p->dcode_epoch != p->storage_epoch)
        return false;
    if (p->dist) {
        // a row block: whole planes, the symmetric form on its interior window, every other interior slice and the boundary
        // slices by the slice-by-slice / CSR launches behind the fused one (their partial sums must fit behind its own)
        const fv_dist *d = p->dist;
        if (!g_fused_dist || !d->split_built || d->int_hi <= d->int_lo || d->n_int_csr > 0)
            return false;
        auto grid_bound = [](int64_t groups) -> int64_t { return groups > 0 ? (groups / 4 + 16 < FV_MAX_PARTIALS ? groups / 4 + 16 : FV_MAX_PARTIALS) : 0; };
        if (2 * (int64_t)p->ctx->num_cus + grid_bound(p->sym_nrest) + grid_bound(d->n_bnd_dia) + grid_bound(d->n_bnd_csr) > FV_FUSED_PARTS)
            return false;
    }
    const int64_t nz = p->sym_d[1], d3 = p->sym_d[2];
    if (p->sym_d[0] != 1 || nz < 64 || nz % 2 || d3 % 2 || d3 % nz || p->n % d3 || p->n / d3 < 3)
        return false;
    if ((d3 / nz + 2) * (int64_t)(KF_TW + 4) >= 4096 * 64) // (the halo descriptor's 12-bit LDS slots always fit; the guard is on the grid)
        return false;
    return true;
}

int fv_fused_prepare(fv_problem *p)
{
    fv_ctx *ctx = p->ctx;
    if (p->fz_part.p)
        return FV_OK;
    const size_t n = (size_t)p->n + (size_t)p->nhalo + FV_VEC_PAD;
    FV_TRY(fv_vec_alloc(p, p->qv, n, true));
    FV_TRY(fv_vec_alloc(p, p->qv2, n, true));
    FV_TRY(p->qv.zero(ctx));
    FV_TRY(p->qv2.zero(ctx));
    FV_TRY(p->fz_part.alloc(ctx, (size_t)2 * 7 * FV_FUSED_PARTS));
    FV_TRY(p->fz_part.zero(ctx));
    return FV_OK;
}

FusedSums fv_fused_sums(fv_problem *p, int parity)
{
    FusedSums s{};
    double *base = p->fz_part.p + (size_t)parity * 7 * FV_FUSED_PARTS;
    s.arz = base;
    s.arr = base + FV_FUSED_PARTS;
    s.srz = base + 2 * FV_FUSED_PARTS;
    s.srr = base + 3 * FV_FUSED_PARTS;
    s.pq = base + 4 * FV_FUSED_PARTS;
    s.sbb = base + 5 * FV_FUSED_PARTS; // two blocks of FV_FUSED_PARTS: the vector part, then the sparse-b gather
    return s;
}

// v of the direction z (p->pvec) from its classic product q (p->q): entry into the fused regime
int fv_fused_enter(fv_problem *p, double sigma)
{
    fv_ctx *ctx = p->ctx;
    FV_TRY(fv_fused_prepare(p));
    hipLaunchKernelGGL(q_to_v_kernel, dim3(vec_grid(p->n)), dim3(FV_BLOCK), 0, ctx->stream, p->n, (const double *)p->q.p, (const double *)p->pvec.p,
                       (const double *)p->minv.p, (const double *)p->D.p, sigma, p->qv.p);
    FV_LAUNCH_CHECK(ctx);
    return FV_OK;
}

// geometry, grid and the symmetric arrays of a launch; returns the grid size
static int kf_setup(fv_problem *p, KfArgs &a)
{
    fv_ctx *ctx = p->ctx;
    const int64_t nz = p->sym_d[1], d3 = p->sym_d[2];
    a.nz = (int32_t)nz;
    a.d3 = (int32_t)d3;
    a.L = (int32_t)(d3 / nz);
    a.P = (int32_t)(p->n / d3);
    a.nplanes = a.P - 1;
    const int TLr = g_fused_lines == 16 ? 16 : 8;
    a.tilesC = (int32_t)((nz + KF_TW - 1) / KF_TW);
    a.tiles = a.tilesC * (int32_t)((a.L + TLr - 1) / TLr);
    int resident = ctx->num_cus * (TLr == 16 ? 1 : g_fused_blocks) / 8 * 8;
    // a row block's launch runs while the halo exchange is in flight: its blocks fill their CUs (registers, LDS), so one CU
    // per XCD is left to the transport's kernel — or that kernel, enqueued first, would push a block into a second round
    if (p->dist && p->dist->nranks > 1 && resident >= 64)
        resident -= 8 * g_fused_dist_spare;
    if (resident < 8)
        resident = 8;
    if (resident > FV_FUSED_PARTS)
        resident = FV_FUSED_PARTS;
    int nsegs = 1;
    {
        int64_t best = -1;
        for (int m = 1; m <= 64 && m <= a.nplanes - 1; m++) {
            const int64_t rounds = ((int64_t)a.tiles * m + resident - 1) / resident;
            const int64_t cost = rounds * ((a.nplanes - 1 + m - 1) / m + 3);
            if (best < 0 || cost < best) {
                best = cost;
                nsegs = m;
            }
        }
        if (g_fused_segs > 0 && g_fused_segs <= a.nplanes - 1)
            nsegs = g_fused_segs;
    }
    a.nsegs = nsegs;
    a.seglen = (a.nplanes - 1 + nsegs - 1) / nsegs;
    int64_t g = (((int64_t)a.tiles * nsegs + 7) / 8) * 8;
    if (g > resident)
        g = resident;
    const int GF = (int)g;
    const double *dg = p->sym_vals.p + p->sym_front;
    a.dg = dg;
    a.front = (uint32_t)p->sym_front;
    a.u1 = dg + p->sym_ld;
    a.u2 = dg + 2 * p->sym_ld;
    a.u3 = dg + 3 * p->sym_ld;
    a.ok = p->sym_ok.p;
    return GF;
}

// the matrix as codes where the symmetric copy has them (fv_matrix_codes, fv_spmv.hip) and the switch is on
static bool kf_codes(fv_problem *p, KfArgs &a)
{
    if (!g_fused_codes || p->sym_mcode_n <= 0 || !p->sym_mcode.p)
        return false;
    a.mcode = p->sym_mcode.p;
    a.mt = p->sym_mtab;
    return true;
}

// ---- the chunk kernels' plan: variant (threads, pairs per thread), rows per chunk, chunks per plane, segments of planes, grid, LDS
constexpr size_t KC_LDS_MAX = 160 * 1024;
struct KcPlan {
    int nt, np, grid, hr;
    bool big; // vectors of more than 4 GiB (fv_problem::sym_big): the doubles kernel's offsets count from the item's segment
    size_t lds;
    bool doubles; // fused_chunkd_kernel (the matrix as doubles) instead of fused_chunk_kernel (as 16-bit codes)
};
static size_t kc_lds_bytes(int64_t C, int64_t nz, int nt, bool doubles)
{
    if (doubles) // z' and U2 double-buffered, the storage table, the reduction scratch
        return (size_t)(16 * (C + 2 * nz) + 16 * (C + nz) + 8 * (FV_STORAGE_CODES + nt / 64));
    return (size_t)(16 * (C + 2 * nz) + 4 * (C + nz) + 8 * (2 * FV_STORAGE_CODES + 3 * FV_MATRIX_CODES + nt / 64));
}
// false: the 2-D tiles serve the launch (lines longer than the block, chunks that would not fit the LDS, no code bytes)
static bool kc_plan(fv_problem *p, KfArgs &a, KcPlan &pl, bool coded, int mode = 0)
{
    if (!g_fused_chunk || !p->kc_code.p || p->kc_state != (coded ? 1 : 2) || (coded && !a.mcode))
        return false;

    a.kcode = p->kc_code.p;
    a.kdiag = p->kc_dtab;
    a.pfirst = 1;
    if (p->kc_ends && !p->dist) { // the first and the last plane are centre planes too: no slice-by-slice launch behind the kernel for them
        a.pfirst = 0;
        a.nplanes = a.P;
    }
    // 512 threads x 5 pairs of rows per thread: the variant whose state fits the register file without spills — 247 VGPRs at two waves per SIMD.
    // Measured before the others were removed (464^3, one process, ms per step): tiles 1.070, (512, 5) 0.951, (512, 6) 0.995 with 27 spilled
    // registers, (1024, 2) 1.43, (1024, 3) 2.00, (768, 4) 1.44 (profiles/r04_step_ab_chunks*.log).
    const int64_t nz = a.nz, d3 = a.d3;
    // (the kernel with the matrix as doubles carries more state per pair: four fit 256 registers, five spill.  With four halo rounds — lines of
    // 769 .. 1024 rows — five pairs spill in the coded kernel too, 40 B per lane in the step and 88 in the one-launch iteration: 928^3, alternating
    // runs, the step 6.99 ms with five pairs and 7.20 with four, an iteration's launches 90.5 against 87.9 per six steps: profiles/r05_hr4_pairs.log)
    const int nt = 512, np = (coded && !(2 * nz > 3 * nt && mode == 3)) ? 5 : 4;
    // the 2 nz halo rows are covered in two rounds of the block (three for the coded kernel on lines of 513 .. 768 rows: 640^3, the largest box one
    // GPU holds with 32-bit indices); longer lines stay with the tiles
    pl.hr = (int)((2 * nz + nt - 1) / nt);
    if (pl.hr < 2)
        pl.hr = 2;
    if (pl.hr > 4)
        return false;
    const int64_t fixed = (int64_t)kc_lds_bytes(0, nz, nt, !coded);
    int64_t cmax = ((int64_t)KC_LDS_MAX - fixed) / (coded ? 20 : 32) / 16 * 16;
    if (cmax > 2 * (int64_t)nt * np)
        cmax = 2 * (int64_t)nt * np;
    if (cmax < 2048 || cmax < 2 * nz) // (a chunk shorter than two lines: the halo would outweigh it)
        return false;
    fv_ctx *ctx = p->ctx;
    int resident = ctx->num_cus / 8 * 8; // one block per CU (its LDS)
    if (p->dist && p->dist->nranks > 1 && resident >= 64)
        resident -= 8 * g_fused_dist_spare;
    if (resident < 8)
        resident = 8;
    if (resident > FV_FUSED_PARTS)
        resident = FV_FUSED_PARTS;
    // chunks per plane K (equal chunks of C rows, C a multiple of 16) and segments of planes m: rounds x (planes per segment + 3)
    // x (rows per chunk + its halo's weight), smallest first
    const int64_t kmin = (d3 + cmax - 1) / cmax;
    double best = -1.0;
    int64_t bestK = kmin, bestC = cmax;
    int bestm = 1;
    // (round 5: searching up to 4 kmin + 8 chunks — shorter chunks, longer segments, which this cost model prefers on small operators — measured WORSE:
    // 216^3 0.126 -> 0.138 ms per step, 320^3 0.318 -> 0.328, 464^3 unchanged; the model underrates what a chunk costs besides its rows)
    for (int64_t K = kmin; K <= kmin + 8; K++) {
        const int64_t Cr = ((d3 + K - 1) / K + 15) / 16 * 16;
        if (Cr > cmax || Cr < 2 * nz)
            continue;
        const int64_t Ke = (d3 + Cr - 1) / Cr;
        const int nprod = a.nplanes - a.pfirst; // product planes
        for (int m = 1; m <= 64 && m <= nprod; m++) {
            // (fused_chunkd_kernel's 32-bit byte offsets span a segment with the plane in front of it and the two behind, and the symmetric arrays' front padding)
            if (!coded && p->sym_big && (((int64_t)(nprod + m - 1) / m + 4) * d3 + (int64_t)p->sym_front + 2 * nz) * 8 >= ((int64_t)1 << 32))
                continue;
            const int64_t rounds = (Ke * m + resident - 1) / resident;
            const double cost = (double)rounds * (double)((nprod + m - 1) / m + 3) * ((double)Cr + 0.6 * (double)nz);
            if (best < 0.0 || cost < best) {
                best = cost;
                bestK = Ke;
                bestC = Cr;
                bestm = m;
            }
        }
    }
    if (best < 0.0)
        return false;
    a.chunk = (int32_t)bestC;
    a.tiles = (int32_t)bestK;
    a.tilesC = 1;
    a.nsegs = bestm;
    a.seglen = (a.nplanes - a.pfirst + bestm - 1) / bestm;
    int64_t g = (((int64_t)a.tiles * a.nsegs + 7) / 8) * 8;
    if (g > resident)
        g = resident;
    pl.nt = nt;
    pl.np = np;
    pl.grid = (int)g;
    pl.lds = kc_lds_bytes(bestC, nz, nt, !coded);
    pl.doubles = !coded;
    pl.big = p->sym_big;
    return true;
}
template <int NT, int NP, int MODE, bool DOUBLES, int PD = 2, int PU = 2, int HR = 2, bool BIG = false>
static int kc_launch_one(fv_ctx *ctx, const KfArgs &a, const KcPlan &pl)
{
    static bool raised = false; // (per instantiation: the dynamic LDS limit of the kernel, above the 64 KB default)
    void (*kern)(KfArgs);
    if constexpr (DOUBLES)
        kern = &fused_chunkd_kernel<NT, NP, MODE, PD, PU, HR, BIG>;
    else
        kern = &fused_chunk_kernel<NT, NP, MODE, HR>;
    if (!raised) {
        FV_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)KC_LDS_MAX));
        raised = true;
    }
    hipLaunchKernelGGL(kern, dim3(pl.grid), dim3(NT), pl.lds, ctx->stream, a);
    return FV_OK;
}
template <int MODE>
static int kc_launch(fv_ctx *ctx, const KfArgs &a, const KcPlan &pl)
{
    if (pl.doubles) {
        // 512 threads x 4 pairs, the batch three pairs ahead, U1 / U3 two, the wave's edge element by a vector load: measured 464^3, one process, ms
        // per step (tiles 1.531): (2,2) 1.376, (3,3) 1.373, (4,4) 1.378, (2,1) 1.347, (3,2) 1.335, (1,1) 1.344; 5 pairs spill (60-124 B per lane):
        // 1.384-1.477; the edge element by a scalar load (constant address space): +1.8 %; 64-bit plane bases: +0.1 % (profiles/r05_hetero_ab_*.log)
        if (pl.big) { // (vectors of more than 4 GiB: 5.4e8 rows and more)
            if (pl.hr == 4)
                return kc_launch_one<512, 4, MODE, true, 3, 2, 4, true>(ctx, a, pl);
            if (pl.hr == 3)
                return kc_launch_one<512, 4, MODE, true, 3, 2, 3, true>(ctx, a, pl);
            return kc_launch_one<512, 4, MODE, true, 3, 2, 2, true>(ctx, a, pl);
        }
        if (pl.hr == 4)
            return kc_launch_one<512, 4, MODE, true, 3, 2, 4>(ctx, a, pl);
        if (pl.hr == 3)
            return kc_launch_one<512, 4, MODE, true, 3, 2, 3>(ctx, a, pl);
        return kc_launch_one<512, 4, MODE, true, 3, 2>(ctx, a, pl);
    }
    // the coded kernel: streaming hints as compile-time constants (stores stay 16-byte instructions): 464^3, one process, ms per step 0.8965 -> 0.8702;
    // with 32-bit offsets instead of 64-bit plane bases on top the median of five rounds was WORSE (0.965 / 0.941, minimum 0.893 / 0.874: two
    // modes), so its bases stay (profiles/r05_step_ab_coded_addressing.log)
    if (pl.hr == 4) // (lines of 769 .. 1024 rows: five pairs per thread spill 40-88 B per lane with four halo rounds, four pairs do not)
        return pl.np == 5 ? kc_launch_one<512, 5, MODE, false, 2, 2, 4>(ctx, a, pl) : kc_launch_one<512, 4, MODE, false, 2, 2, 4>(ctx, a, pl);
    if (pl.hr == 3)
        return kc_launch_one<512, 5, MODE, false, 2, 2, 3>(ctx, a, pl);
    return kc_launch_one<512, 5, MODE, false>(ctx, a, pl);
}

// the SELL variant of fv_fused_step (same contract)
static int fused_sell_step(fv_problem *p, const double *x, double *x_next, double sigma, double dt, double rtol, int chain_index, int mode,
                           const FusedSums &in, bool force_prev_unconverged, const double *folded, int64_t bsupport, FusedSums *out_sums)
{
    fv_ctx *ctx = p->ctx;
    KfArgs a{};
    KsArgs s{};
    const bool coded = p->dcode_n > 0 && p->dcode_epoch == p->storage_epoch;
    a.code = coded && p->dcode_n > 1 ? p->dcode.p : nullptr;
    for (int k = 0; k < FV_STORAGE_CODES; k++)
        a.sD.v[k] = coded ? sigma * p->dtable.v[k] : 0.0;
    a.x = x;
    a.z = p->pvec.p;
    a.v = p->qv.p;
    a.xout = x_next;
    a.znext = p->pnext.p;
    a.vnext = p->qv2.p;
    a.bidx = bsupport > 0 ? p->bnz_idx.p : nullptr;
    a.b = p->b.p;
    a.bm = bsupport > 0 ? bsupport : 0;
    a.scal = p->scal.p;
    a.in = in;
    a.mode = mode;
    a.nt = g_fused_nt;
    a.chain_index = chain_index;
    a.force_prev_unconverged = force_prev_unconverged ? 1 : 0;
    a.rtol = rtol;
    a.n = p->n;
    a.r = p->r.p;
    a.pold = p->pnext.p;
    a.minv = p->minv.p;
    a.D = p->D.p;
    a.xprev = x_next;
    a.dt = dt;
    // (78 registers: six waves per SIMD at most, and four blocks per CU measure best — 5M-cell mesh, ms per step at 8 / 6 / 5 / 4 / 3 / 2: 0.161 / 0.146 / 0.142 / 0.136 / 0.142 / 0.174; a grid beyond the resident blocks runs a second, thin round of its static shares)
    int G = fv_sell_grid(p);
    const int resident = p->ctx->num_cus * g_fused_sell_blocks / 8 * 8;
    if (G > resident && resident >= 8)
        G = resident;
    FusedSums out = fv_fused_sums(p, chain_index & 1);
    out.nvec = G;
    out.nbb = a.bm > 0 ? 2 * G : G;
    a.out = out;
    s.count = p->sell_n;
    s.list = p->sell_list.p;
    s.ptr = p->sell_ptr.p;
    s.w8 = p->sell_w.p;
    s.sv = p->sell_vals.p;
    s.sd = p->sell_dcol.p;
    s.D = coded ? nullptr : p->D.p;
    s.sigma = sigma;
    s.rest = p->sell_rest.p;
    s.nrest = p->sell_nrest;
    s.diagA = p->diagA.p;
    hipLaunchKernelGGL(fused_sell_step_kernel, dim3(G), dim3(FV_BLOCK), 0, ctx->stream, a, s);
    FV_LAUNCH_CHECK(ctx);
    int GR = 0;
    if (p->sell_nrest > 0) { // the groups outside the form: classic product of z' into v', then the v-form
        FV_TRY(fv_spmv_sell_rest(p, p->pnext.p, p->qv2.p, folded, out.pq + G, &GR));
        hipLaunchKernelGGL(q_to_v_slices_kernel, dim3(fv_blocks(p->sell_nrest * 64)), dim3(FV_BLOCK), 0, ctx->stream, p->n, p->sell_nrest,
                           (const int32_t *)p->sell_rest.p, (const double *)p->pnext.p, (const double *)p->minv.p, (const double *)p->D.p, sigma, p->qv2.p);
        FV_LAUNCH_CHECK(ctx);
    }
    out.npq = G + GR;
    *out_sums = out;
    const int64_t matrix = p->sell_blocks * 640 + 5 * p->sell_n, stor = coded ? (a.code ? 1 : 0) : 8;
    p->fused_bytes_launch = (48 + stor) * p->n + matrix;
    p->fused_bytes = (int32_t)(p->fused_bytes_launch / (p->n > 0 ? p->n : 1));
    return FV_OK;
}

// One fused launch (+ the slice-by-slice launch for the slices the symmetric form leaves out): step `chain_index` of a
// burst.  x -> x_next, p->pvec (z) -> p->pnext (z'), p->qv (v) -> p->qv2 (v'); sums of parity `chain_index & 1`.
int fv_fused_step(fv_problem *p, const double *x, double *x_next, double sigma, double dt, double rtol, int chain_index, int mode, const FusedSums &in,
                  bool force_prev_unconverged, const double *folded, int64_t bsupport, FusedSums *out_sums, const double *red)
{
    fv_ctx *ctx = p->ctx;
    if (kf_sell(p))
        return fused_sell_step(p, x, x_next, sigma, dt, rtol, chain_index, mode, in, force_prev_unconverged, folded, bsupport, out_sums);
    KfArgs a{};
    a.red = red;
    const int GF = kf_setup(p, a);
    const int TLr = g_fused_lines == 16 ? 16 : 8;
    a.code = p->dcode_n > 1 ? p->dcode.p : nullptr;
    for (int k = 0; k < FV_STORAGE_CODES; k++)
        a.sD.v[k] = sigma * p->dtable.v[k];
    a.x = x;
    a.z = p->pvec.p;
    a.v = p->qv.p;
    a.xout = x_next;
    a.znext = p->pnext.p;
    a.vnext = p->qv2.p;
    a.bidx = bsupport > 0 ? p->bnz_idx.p : nullptr;
    a.b = p->b.p;
    a.bm = bsupport > 0 ? bsupport : 0;
    a.scal = p->scal.p;
    a.in = in;
    FusedSums out = fv_fused_sums(p, chain_index & 1);
    a.mode = mode;
    a.nt = g_fused_nt;
    a.chain_index = chain_index;
    a.force_prev_unconverged = force_prev_unconverged ? 1 : 0;
    a.rtol = rtol;
    a.n = p->n;
    a.r = p->r.p;
    a.pold = p->pnext.p; // the previous step's direction (the host has swapped the two direction vectors for this step)
    a.minv = p->minv.p;
    a.D = p->D.p;
    a.xprev = x_next;    // ... and the state that step started from
    a.dt = dt;
    out.nvec = GF;
    out.nbb = a.bm > 0 ? 2 * GF : GF;
    // (the sparse-b partials sit right behind the vector part's: sbb[GF .. 2 GF))
    a.out = out;
    const bool coded = kf_codes(p, a);
    KcPlan kc{};
    const bool chunks = kc_plan(p, a, kc, coded);
    if (chunks) {
        out.nvec = kc.grid; // (the plan has its own grid: one partial sum of each kind per block)
        out.nbb = a.bm > 0 ? 2 * kc.grid : kc.grid;
        a.out = out;
        FV_TRY(kc_launch<0>(ctx, a, kc));
    } else if (TLr == 16) {
        if (coded)
            hipLaunchKernelGGL((fused_step_kernel<16, 0, true>), dim3(GF), dim3(1024), 0, ctx->stream, a);
        else
            hipLaunchKernelGGL((fused_step_kernel<16, 0, false>), dim3(GF), dim3(1024), 0, ctx->stream, a);
    } else {
        if (coded)
            hipLaunchKernelGGL((fused_step_kernel<8, 0, true>), dim3(GF), dim3(512), 0, ctx->stream, a);
        else
            hipLaunchKernelGGL((fused_step_kernel<8, 0, false>), dim3(GF), dim3(512), 0, ctx->stream, a);
    }
    FV_LAUNCH_CHECK(ctx);
    // the slices the symmetric form leaves out (first / last plane, irregular ones): classic product of z' into v', then v-form
    int GR = 0;
    const int GK = chunks ? kc.grid : GF;
    const bool ends = chunks && a.pfirst == 0;
    if (ends ? p->sym_nrest_irr > 0 : p->sym_nrest > 0) {
        FV_TRY(fv_spmv_rest(p, p->pnext.p, p->qv2.p, folded, out.pq + GK, &GR, false, sigma, false, ends)); // (stored in the v-form by the launch itself)
    }
    out.npq = GK + GR;
    p->fused_chunked = chunks;
    *out_sums = out;
    {
        // every array once: x, z, v in and x_out, z', v' out on all rows (48) + a code byte; the three upper diagonals on the rows
        // whose product this kernel forms (24), + the stored diagonal where it is not re-derived (8)
        const int64_t nok = (p->dist ? p->dist->int_hi - p->dist->int_lo : p->ndia) - (ends ? p->sym_nrest_irr : p->sym_nrest), nder = ends ? nok : p->sym_nderived; // (chunks: no diagonal stream at all)
        const int mb = coded ? 2 : 24;
        p->fused_bytes = (nder * 2 >= nok ? 73 : 81) - 24 + mb;
        p->fused_bytes_launch = (48 + (a.code ? 1 : 0)) * p->n + mb * 64 * nok + 8 * 64 * (nok - nder);
        if (chunks && kc.doubles) // the code byte on every row, the stored diagonal on the rows that do not derive theirs
            p->fused_bytes_launch = 49 * p->n + 24 * 64 * nok + 8 * p->kc_nstream;
    }
    return FV_OK;
}

// Row blocks: the send buffer of the halo exchange of z' (p->pnext), formed from z (p->pvec), v (p->qv) and the all-reduced
// sums in `red` before the fused launch of the same step (fused_pack_kernel).
int fv_fused_pack(fv_problem *p, const double *red, int mode, int chain_index, bool force_prev_unconverged, double rtol)
{
    fv_ctx *ctx = p->ctx;
    fv_dist *d = p->dist;
    if (!d || d->nsend <= 0)
        return FV_OK;
    hipLaunchKernelGGL(fused_pack_kernel, dim3(fv_blocks(d->nsend)), dim3(FV_BLOCK), 0, ctx->stream, d->nsend, (const int32_t *)d->send_idx.p,
                       (const double *)p->pvec.p, (const double *)p->qv.p, red, (const PcgScalars *)p->scal.p, mode, chain_index,
                       force_prev_unconverged ? 1 : 0, rtol, d->sendbuf.p);
    FV_LAUNCH_CHECK(ctx);
    return FV_OK;
}

// Row blocks: the products of the listed 64-row groups (boundary slices), left as q' in the NEW v array (p->qv2) by the
// classic launches, into v' = -M^-1 (q' - sigma D z') with z' = p->pnext.
int fv_fused_convert_groups(fv_problem *p, const int32_t *groups, int64_t count, double sigma)
{
    fv_ctx *ctx = p->ctx;
    if (count <= 0)
        return FV_OK;
    hipLaunchKernelGGL(q_to_v_slices_kernel, dim3(fv_blocks(count * 64)), dim3(FV_BLOCK), 0, ctx->stream, p->n, count, groups, (const double *)p->pnext.p,
                       (const double *)p->minv.p, (const double *)p->D.p, sigma, p->qv2.p);
    FV_LAUNCH_CHECK(ctx);
    return FV_OK;
}

// Can the many-iteration loop run its passes through the fused kernel (MODE 1)?  The tiled symmetric form serves the
// operator with the shift folded into its copy (or no shift at all).
bool fv_fused_iteration_applicable(fv_problem *p, double sigma, bool folded)
{
    if (!g_fused || !g_fused_iter || p->dist || p->nhalo > 0 || p->sym_state != 1 || p->last_form != FV_SPMV_SYM_TILE)
        return false;
    if (p->sym_epoch != p->assemble_epoch || p->sym_tag != (folded ? sigma : 0.0) || (sigma != 0.0 && !folded))
        return false;
    const int64_t nz = p->sym_d[1], d3 = p->sym_d[2];
    return p->sym_d[0] == 1 && nz >= 64 && nz % 2 == 0 && d3 % 2 == 0 && d3 % nz == 0 && p->n % d3 == 0 && p->n / d3 >= 3;
}

// One pass of the many-iteration regime: the scalars of K3 for iteration `it` (from the sums the vector pass left in
// part_rz / part_rr), p' = z + beta p into p->pnext (z = p->r, which holds M^-1 r between the passes; p = p->pvec), q = (A + sigma D) p'
// into p->q, partial p'.q into part_pq (*npq pieces, the slice-by-slice launch's behind the kernel's).
int fv_fused_iteration(fv_problem *p, int it, const double *folded, const double *part_rz, const double *part_rr, int nvec, int *npq, double *x, bool xapply)
{
    fv_ctx *ctx = p->ctx;
    KfArgs a{};
    const int GF = kf_setup(p, a);
    const int TLr = g_fused_lines == 16 ? 16 : 8;
    a.code = nullptr;
    for (int k = 0; k < FV_STORAGE_CODES; k++)
        a.sD.v[k] = 0.0;
    // the diagonal of a slice that re-derives it carries the folded shift by the row's storage code (sym_shift, as in K1)
    if (p->sym_shift_mode) {
        a.sD = p->sym_shift;
        a.code = p->sym_shift_mode == 1 ? p->dcode.p : nullptr;
    }
    a.z = p->r.p;
    a.v = p->pvec.p;
    a.znext = p->pnext.p;
    a.vnext = p->q.p; // receives w = -M^-1 q
    a.x = x;
    a.xout = x;
    a.xapply = xapply ? 1 : 0;
    a.scal = p->scal.p;
    a.in = FusedSums{};
    a.in.arz = const_cast<double *>(part_rz);
    a.in.arr = const_cast<double *>(part_rr);
    a.in.nvec = nvec;
    a.out = FusedSums{};
    a.out.pq = p->part_pq.p;
    a.chain_index = it;
    a.hist = p->hist.p;
    a.hist_cap = p->hist_cap;
    a.n = p->n;
    const bool coded = kf_codes(p, a);
    KcPlan kc{};
    const bool chunks = kc_plan(p, a, kc, coded);
    if (chunks)
        FV_TRY(kc_launch<1>(ctx, a, kc));
    else if (TLr == 16) {
        if (coded)
            hipLaunchKernelGGL((fused_step_kernel<16, 1, true>), dim3(GF), dim3(1024), 0, ctx->stream, a);
        else
            hipLaunchKernelGGL((fused_step_kernel<16, 1, false>), dim3(GF), dim3(1024), 0, ctx->stream, a);
    } else {
        if (coded)
            hipLaunchKernelGGL((fused_step_kernel<8, 1, true>), dim3(GF), dim3(512), 0, ctx->stream, a);
        else
            hipLaunchKernelGGL((fused_step_kernel<8, 1, false>), dim3(GF), dim3(512), 0, ctx->stream, a);
    }
    // per row and iteration: z, p, x in and p', w, x out (48) + the three upper diagonals (24, or 2 as codes) + a storage code byte in the
    // pass; z, w in, z' out (24) + M^-1 (8, or a code byte: fv_loop_form subtracts 7) in the vector update
    p->loop_bytes = coded ? 83 : 105;
    FV_LAUNCH_CHECK(ctx);
    int GR = 0;
    const int GK = chunks ? kc.grid : GF;
    const bool ends = chunks && a.pfirst == 0;
    if (ends ? p->sym_nrest_irr > 0 : p->sym_nrest > 0)
        FV_TRY(fv_spmv_rest(p, p->pnext.p, p->q.p, folded ? folded : p->vals.p, p->part_pq.p + GK, &GR, true, 0.0, true, ends)); // (stored as w as well)
    *npq = GK + GR;
    p->fused_chunked = chunks;
    return FV_OK;
}

// ---- the one-launch PCG iteration (MODE 3 of the chunk kernels; kf_ploop_prologue)
int g_ploop = 1; // fv_tune key 63: the many-iteration PCG loop as one launch per iteration where it applies (0: the pass + vector update pair)
// Whole regular boxes only: every plane a centre plane of the chunk traversal (kc_ends), no slice left to the slice-by-slice kernel
// — the five sums over the product must come from ONE kernel — and a chunk plan that fits.
bool fv_ploop_applicable(fv_problem *p, double sigma, bool folded)
{
    if (!g_ploop || !fv_fused_iteration_applicable(p, sigma, folded) || !p->kc_ends || p->sym_nrest_irr > 0 || (p->kc_state != 1 && p->kc_state != 2))
        return false;
    KfArgs a{};
    kf_setup(p, a);
    KcPlan kc{};
    return kc_plan(p, a, kc, kf_codes(p, a), 3) && a.pfirst == 0;
}

// Launch j of a solve's loop (j = 0: the first pass, direction = z, nothing but w' stored).  z, w, pold: iterate j - 1's scaled residual,
// the previous launch's w = -M^-1 q and direction; xin -> xout: x += alpha p (may be the same array); znew, pnew, wnew receive iterate
// j's.  Sums: the previous launch's (set j - 1 & 1) in, this launch's out.
int fv_ploop_pass(fv_problem *p, int j, const double *folded, const double *z, const double *w, const double *pold, const double *xin, double *xout,
                  double *znew, double *pnew, double *wnew)
{
    fv_ctx *ctx = p->ctx;
    FV_TRY(fv_fused_prepare(p));
    KfArgs a{};
    kf_setup(p, a);
    a.code = nullptr;
    for (int k = 0; k < FV_STORAGE_CODES; k++)
        a.sD.v[k] = 0.0;
    if (p->sym_shift_mode) { // the diagonal a row re-derives carries the folded shift by the row's storage code (as in fv_fused_iteration)
        a.sD = p->sym_shift;
        a.code = p->sym_shift_mode == 1 ? p->dcode.p : nullptr;
    }
    a.first = j == 0 ? 1 : 0;
    a.z = z;
    a.w = w;
    a.v = pold;
    a.x = xin;
    a.xout = xout;
    a.zout = znew;
    a.znext = pnew;
    a.vnext = wnew;
    a.scal = p->scal.p;
    a.chain_index = j - 1;
    a.hist = p->hist.p;
    a.hist_cap = p->hist_cap;
    a.n = p->n;
    const bool coded = kf_codes(p, a);
    KcPlan kc{};
    if (!kc_plan(p, a, kc, coded, 3) || a.pfirst != 0) {
        fv_set_error(ctx, "internal: the one-launch PCG iteration on an operator it does not serve");
        return FV_ERR_STATE;
    }
    FusedSums in = fv_fused_sums(p, (j + 1) & 1), out = fv_fused_sums(p, j & 1);
    in.t2 = in.sbb + FV_FUSED_PARTS;
    out.t2 = out.sbb + FV_FUSED_PARTS;
    in.npq = in.nvec = kc.grid;
    a.in = in;
    a.out = out;
    FV_TRY(kc_launch<3>(ctx, a, kc));
    FV_LAUNCH_CHECK(ctx);
    (void)folded;
    p->loop_bytes = coded ? 67 : 89; // z, w, p, x in and z', p', w', x out (64) + the three upper diagonals (24, or 2 as codes) + the code byte
    p->fused_chunked = true;
    p->ploop_grid = kc.grid;
    return FV_OK;
}

FV_WARM_TU(fused) // (fv_ctx_create loads every code object of the library up front: fv_warm_modules, fv_ctx.hip)
