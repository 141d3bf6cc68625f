// The fused Jacobi-preconditioned conjugate gradient and its row-block (multi-GPU) driver.
//
// Replaces IterativeSolvers.cg!/cg + AlgebraicMultigrid at
// /root/reference/src/FiniteVolume.jl:160-161 and src/transient.jl:50-58.
// The operator is (A + sigma*D): A the assembled symmetric CSR, D = Ss*volumes on
// the free cells, sigma = 1/dt (0 for the steady solve), i.e. the SPD form of the
// reference's (I/dt + D^-1 A).  The SpMV forms live in fv_spmv.hip.
//
// Everything here is HBM-bandwidth bound, so no MFMA.  One PCG iteration is three launches:
//   K1 spmv_dot   q = (A + sigma D) p ; partial p.q per block      12 nnz + 20 n  B (CSR accounting)
//   K2 update     alpha = rz/pq ; x += alpha p ; r -= alpha q ;
//                 partial r.M^-1 r and r.r per block                56 n B
//   K3 pupdate    beta = rz'/rz ; p = M^-1 r + beta p ; scalars     32 n B
// Scalars never visit the host inside the loop: every block re-reduces the <= 2048
// per-block partials of the previous kernel in a fixed order (deterministic, no
// atomics), and a device-side `done` flag turns surplus launches into no-ops, so
// the host polls convergence only once per chunk of iterations.
#include "fv_internal.h"
#include "fv_device.h"
#include "fv_spmv.h"
#include "fv_fused.h"
#include <cstring>
#include <cstdlib>

extern int g_carry_refresh, g_carry_speculate, g_chain_steps, g_resume_runs; // fv_transient.hip
int g_defer_reduce = 1; // fv_tune key 22: bursts of chained steps take a step's verdict and the next step's scalars in one launch; row-block runs also merge their two all-reduces (see dist_step)
int g_sparse_b = 1; // (frozen) K2S leaves the b' stream out when b' is sparse
int g_chain_test_break = -1; // fv_tune key 14 (tests): the chained step with this index of every burst is treated as not converged

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
            mi = d > 0.0 ? 1.0 / d : 0.0; // a free cell without any face has an empty row: leave it where it is (0 * x = b must hold)
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
// zsrc: the previous step's final residual is not in r but, Jacobi-scaled, in zsrc (a z-form K2S left it there: see
// pcg_update_spec_kernel); it is r = zsrc / M^-1, the value every consumer of such a state takes.
__global__ __launch_bounds__(FV_BLOCK) void pcg_carry_init_kernel(int64_t n, const double *__restrict__ bprime, const double *__restrict__ D,
                                                                   double dt, const double *__restrict__ x, const double *__restrict__ xprev,
                                                                   const double *__restrict__ minv, double *__restrict__ r,
                                                                   double *pv, double *__restrict__ part_rz,
                                                                   double *__restrict__ part_rr, double *__restrict__ part_bb,
                                                                   const double *zsrc = nullptr)
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
        double2 rv;
        if (zsrc) {
            const double2 zv = reinterpret_cast<const double2 *>(zsrc)[i];
            rv = make_double2(zv.x / mv.x, zv.y / mv.y);
        } else
            rv = r2[i];
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
        const double ri = (zsrc ? zsrc[i] / minv[i] : r[i]) + D[i] * ((x[i] - xprev[i]) / dt);
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
                                                                      PcgScalars *__restrict__ scal, int nparts_bb = -1, int chained = 0)
{
    __shared__ double smem[4];
    if (chained && scal->done != 1)
        return; // the step before this one (not polled by the host) did not converge in its one iteration: the chain stops
    const double rz = reduce_partials(part_rz, nparts, smem);
    const double rr = reduce_partials(part_rr, nparts, smem);
    const double bb = reduce_partials(part_bb, nparts_bb >= 0 ? nparts_bb : nparts, smem); // rhs.rhs may come in more pieces (sparse b part)
    if (threadIdx.x == 0) {
        scal->rz[0] = rz;
        scal->rz[1] = 0.0;
        scal->rr = rr;
        scal->bnorm2 = bb;
        scal->tol2 = rtol * rtol * bb; // stop when ||r|| <= rtol*||b||  (IterativeSolvers' reltol)
        scal->tol2x[0] = scal->tol2x[1] = scal->tol2;
        if (!chained)
            scal->zero_mask = 0u; // a burst starts with a step set up here without the chained flag
        scal->pq = 0.0;
        scal->iters = 0;
        scal->done = (rr <= scal->tol2) ? 1 : 0;
        scal->xlag = -1; // no x-update is pending at a set-up.  (A freshly zeroed block says 0 = "iteration 0's update is pending": a solve that is converged
                         // at its set-up while its chunk of launches enqueued a w-form update then sent pcg_xflush_kernel through a null lag_p — a fault
                         // tools/ploop_fuzz.py found in round 5, in the pair path of round 4)
    }
}

__device__ inline double2 nt_load2(const double2 *p)
{
    const double *q = reinterpret_cast<const double *>(p);
    return make_double2(__builtin_nontemporal_load(q), __builtin_nontemporal_load(q + 1));
}
__device__ inline void nt_store2(double2 *p, double2 v)
{
    double *q = reinterpret_cast<double *>(p);
    __builtin_nontemporal_store(v.x, q);
    __builtin_nontemporal_store(v.y, q + 1);
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

// K2 of the many-iteration loop when its passes run through the fused kernel (fv_fused_iteration): between the passes the
// residual is kept Jacobi-scaled, z = M^-1 r, in the array r (the direction update p' = z + beta p and the halo rows of the
// next product then need z and p only); the residual itself is z / M^-1 here, as in K2S's z-form.  r_is_z = 0: the array
// still holds r (the first iteration after a set-up).  x += alpha p; r' = r - alpha q; the array receives z' = M^-1 r'.
template <bool SPLIT>
__global__ __launch_bounds__(FV_BLOCK) void pcg_update_z_kernel(int64_t n, int it, int r_is_z, const double *xin, double *x, double *__restrict__ r,
                                                                 const double *__restrict__ pv, const double *__restrict__ q,
                                                                 const double *__restrict__ minv, const double *__restrict__ part_pq,
                                                                 int npq, PcgScalars *__restrict__ scal, double *__restrict__ part_rz,
                                                                 double *__restrict__ part_rr, const uint8_t *__restrict__ mcode, StorageTable mtab)
{
    __shared__ double smem[4];
    __shared__ double mvtab[FV_STORAGE_CODES]; // mcode: M^-1 as one-byte codes into its few distinct values (fv_minv_codes) instead of its stream
    if (scal->done)
        return;
    if (mcode && threadIdx.x < FV_STORAGE_CODES)
        mvtab[threadIdx.x] = mtab.v[threadIdx.x];
    const double pq = reduce_partials(part_pq, npq, smem); // (its barriers publish mvtab)
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
        const double2 pvv = p2[i], qv = nt_load2(q2 + i);
        double2 mv;
        if (mcode) {
            const uint32_t c = reinterpret_cast<const uint16_t *>(mcode)[i];
            mv = make_double2(mvtab[c & 255u], mvtab[c >> 8]);
        } else
            mv = m2[i];
        if (r_is_z)
            rv = make_double2(rv.x / mv.x, rv.y / mv.y);
        xv.x += alpha * pvv.x;
        xv.y += alpha * pvv.y;
        rv.x -= alpha * qv.x;
        rv.y -= alpha * qv.y;
        const double2 zv = make_double2(mv.x * rv.x, mv.y * rv.y);
        x2[i] = xv;
        r2[i] = zv;
        arz += rv.x * zv.x + rv.y * zv.y;
        arr += rv.x * rv.x + rv.y * rv.y;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const int64_t i = n - 1;
        const double xi = (SPLIT ? xin[i] : x[i]) + alpha * pv[i];
        const double ri = (r_is_z ? r[i] / minv[i] : r[i]) - alpha * q[i];
        x[i] = xi;
        r[i] = minv[i] * ri;
        arz += ri * (minv[i] * ri);
        arr += ri * ri;
    }
    const double t0 = block_sum(arz, smem);
    const double t1 = block_sum(arr, smem);
    if (threadIdx.x == 0) {
        part_rz[blockIdx.x] = t0;
        part_rr[blockIdx.x] = t1;
        if (blockIdx.x == 0) {
            scal->pq = pq;
            scal->xlag = -1; // (this kernel updates x itself: nothing lags behind it)
        }
    }
}

// The vector update of that loop from its second iteration on: the pass before it (fv_fused_iteration) has left w = -M^-1 q in
// place of q and will apply x += alpha p itself one pass later (it reads that direction anyway), so what is left here is
//     z' = z + alpha w ;  r' = z' / M^-1 ;  sums r'.z', r'.r'
// — z, w in, z' out, M^-1 as its stream or as a code byte: 25 / 32 B per row where pcg_update_z_kernel moves 49 / 56.  Block 0
// records the update that now lags (alpha, the direction it belongs to, its iteration) for the next pass / pcg_xflush_kernel.
__global__ __launch_bounds__(FV_BLOCK) void pcg_update_w_kernel(int64_t n, int it, double *__restrict__ z, const double *__restrict__ w,
                                                                 const double *__restrict__ pv, const double *__restrict__ minv,
                                                                 const double *__restrict__ part_pq, int npq, PcgScalars *__restrict__ scal,
                                                                 double *__restrict__ part_rz, double *__restrict__ part_rr,
                                                                 const uint8_t *__restrict__ mcode, StorageTable mtab)
{
    __shared__ double smem[4];
    __shared__ double mvtab[FV_STORAGE_CODES];
    if (scal->done)
        return;
    if (mcode && threadIdx.x < FV_STORAGE_CODES)
        mvtab[threadIdx.x] = mtab.v[threadIdx.x];
    const double pq = reduce_partials(part_pq, npq, smem); // (its barriers publish mvtab)
    if (!(pq > 0.0)) { // breakdown: not positive definite, or NaN
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            scal->pq = pq;
            scal->done = 2;
            scal->xlag = -1; // (the pass in front of this launch has applied the update that was lagging: pcg_xflush_kernel must not apply it again)
        }
        return;
    }
    const double alpha = scal->rz[it & 1] / pq;
    double arz = 0.0, arr = 0.0;
    const int64_t n2 = n >> 1;
    double2 *z2 = reinterpret_cast<double2 *>(z);
    const double2 *w2 = reinterpret_cast<const double2 *>(w);
    const double2 *m2 = reinterpret_cast<const double2 *>(minv);
    for (int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x; i < n2; i += vec_stride()) {
        double2 zv = z2[i];
        const double2 wv = nt_load2(w2 + i);
        double2 mv;
        if (mcode) {
            const uint32_t c = reinterpret_cast<const uint16_t *>(mcode)[i];
            mv = make_double2(mvtab[c & 255u], mvtab[c >> 8]);
        } else
            mv = m2[i];
        zv.x += alpha * wv.x;
        zv.y += alpha * wv.y;
        const double2 rv = make_double2(zv.x / mv.x, zv.y / mv.y);
        z2[i] = zv;
        arz += rv.x * zv.x + rv.y * zv.y;
        arr += rv.x * rv.x + rv.y * rv.y;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const int64_t i = n - 1;
        const double zi = z[i] + alpha * w[i], ri = zi / minv[i];
        z[i] = zi;
        arz += ri * zi;
        arr += ri * ri;
    }
    const double t0 = block_sum(arz, smem);
    const double t1 = block_sum(arr, smem);
    if (threadIdx.x == 0) {
        part_rz[blockIdx.x] = t0;
        part_rr[blockIdx.x] = t1;
        if (blockIdx.x == 0) {
            scal->pq = pq;
            scal->alpha_last = alpha;
            scal->lag_p = pv;
            scal->xlag = it;
        }
    }
}

// x += alpha_last * lag_p when an x-update of that loop is still outstanding (the loop has stopped: converged, out of iterations, broken
// down — the host cannot know at which launch), then pcg_xflush_clear_kernel marks it done.
__global__ __launch_bounds__(FV_BLOCK) void pcg_xflush_kernel(int64_t n, double *__restrict__ x, const PcgScalars *__restrict__ scal)
{
    if (scal->xlag < 0 || !scal->lag_p)
        return;
    const double alpha = scal->alpha_last;
    const double *__restrict__ pv = scal->lag_p;
    for (int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x; i < n; i += vec_stride())
        x[i] += alpha * pv[i];
}
__global__ void pcg_xflush_clear_kernel(PcgScalars *__restrict__ scal) { scal->xlag = -1; }

// The code byte of pcg_carry_flush_kernel: the row's storage code (0 where D takes one value) | bit 7: the assembled b is not zero there
__global__ __launch_bounds__(FV_BLOCK) void carry_code_kernel(int64_t n, const uint8_t *__restrict__ dcode, const double *__restrict__ b, uint8_t *__restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x; i < n; i += vec_stride())
        out[i] = (uint8_t)((dcode ? (dcode[i] & 15) : 0) | ((b && b[i] != 0.0) ? 0x80 : 0));
}

// K0' for a step that follows a loop of one-launch iterations in the same fixed-dt run (fv_ploop_pass): the previous step's last update
// is still pending — z_m = z + alpha w, x_m = x + alpha p, alpha in the scalar block — and is applied HERE, in the pass that forms the
// new step's set-up the way pcg_carry_init_kernel does (same arithmetic): r0 = z_m / M^-1 + D (x_m - x_prev) / dt, z0 = M^-1 r0, the
// sums r0.z0, r0.r0, rhs.rhs with rhs = b + D x_m / dt.  z, w, p, x, x_prev, M^-1 in (48) + a code byte (the storage code; bit 7: b is
// not zero on the row, where it is then read), x_m and z0 out (16): 65 B per row where the flush (48) and K0' (64) moved 112.
__global__ __launch_bounds__(FV_BLOCK) void pcg_carry_flush_kernel(int64_t n, const double *z, const double *w, const double *pp, double *x,
                                                                    const double *__restrict__ xprev, const double *__restrict__ minv,
                                                                    const uint8_t *__restrict__ code, StorageTable dtab, const double *__restrict__ b, double dt,
                                                                    const PcgScalars *__restrict__ scal, double *pv, double *__restrict__ part_rz,
                                                                    double *__restrict__ part_rr, double *__restrict__ part_bb)
{
    __shared__ double smem[4];
    __shared__ double tab[FV_STORAGE_CODES];
    if (threadIdx.x < FV_STORAGE_CODES)
        tab[threadIdx.x] = dtab.v[threadIdx.x];
    __syncthreads();
    const double alpha = scal->alpha_last;
    double arz = 0.0, arr = 0.0, abb = 0.0;
    const int64_t n2 = n >> 1;
    const double2 *z2 = reinterpret_cast<const double2 *>(z), *w2 = reinterpret_cast<const double2 *>(w), *q2 = reinterpret_cast<const double2 *>(pp);
    const double2 *o2 = reinterpret_cast<const double2 *>(xprev), *m2 = reinterpret_cast<const double2 *>(minv);
    double2 *x2 = reinterpret_cast<double2 *>(x), *p2 = reinterpret_cast<double2 *>(pv);
    for (int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x; i < n2; i += vec_stride()) {
        const double2 zv0 = z2[i], wv = nt_load2(w2 + i), pvv = q2[i], xv0 = x2[i], ov = nt_load2(o2 + i), mv = nt_load2(m2 + i);
        const uint32_t c = reinterpret_cast<const uint16_t *>(code)[i];
        const double2 dv = make_double2(tab[c & 15u], tab[(c >> 8) & 15u]);
        double2 bv = make_double2(0.0, 0.0);
        if (c & 0x8080u)
            bv = reinterpret_cast<const double2 *>(b)[i];
        const double2 zv = make_double2(zv0.x + alpha * wv.x, zv0.y + alpha * wv.y);
        const double2 xv = make_double2(xv0.x + alpha * pvv.x, xv0.y + alpha * pvv.y);
        double2 rv = make_double2(zv.x / mv.x, zv.y / mv.y);
        rv.x += dv.x * ((xv.x - ov.x) / dt);
        rv.y += dv.y * ((xv.y - ov.y) / dt);
        const double hx = bv.x + dv.x * (xv.x / dt), hy = bv.y + dv.y * (xv.y / dt);
        const double zx = mv.x * rv.x, zy = mv.y * rv.y;
        x2[i] = xv;
        p2[i] = make_double2(zx, zy);
        arz += rv.x * zx + rv.y * zy;
        arr += rv.x * rv.x + rv.y * rv.y;
        abb += hx * hx + hy * hy;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const int64_t i = n - 1;
        const double di = tab[code[i] & 15u];
        const double zi0 = z[i] + alpha * w[i], xi = x[i] + alpha * pp[i];
        const double ri = zi0 / minv[i] + di * ((xi - xprev[i]) / dt);
        const double hi = ((code[i] & 0x80u) ? b[i] : 0.0) + di * (xi / dt);
        const double zi = minv[i] * ri;
        x[i] = xi;
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

// The end of a loop of one-launch iterations (fv_ploop_pass; kf_ploop_prologue in fv_fused.hip): the one update that is still pending —
// z' = z + alpha w, x' = x + alpha p.  have_alpha: the launch that found the iterate converged has left alpha in the scalar block
// (it stopped before its pass); else the loop ran out of iterations: alpha, the iterate's r.z and r.r (as polynomials in alpha) and
// the verdict on it from the last launch's sums, as the next launch's prologue would have taken them.
__global__ __launch_bounds__(FV_BLOCK) void pcg_ploop_flush_kernel(int64_t n, const double *z, const double *__restrict__ w, const double *__restrict__ pv,
                                                                    const double *xin, double *xout, double *zout, PcgScalars *__restrict__ scal,
                                                                    FusedSums in, int have_alpha, int it, double *__restrict__ hist, int64_t hist_cap)
{
    __shared__ double smem[4];
    double alpha;
    if (have_alpha)
        alpha = scal->alpha_last;
    else {
        if (scal->done)
            return;
        const int np = in.npq;
        const double pq = reduce_partials(in.pq, np, smem);
        if (!(pq > 0.0)) {
            if (blockIdx.x == 0 && threadIdx.x == 0) {
                scal->pq = pq;
                scal->done = 2;
            }
            return;
        }
        const double rzb = reduce_partials(in.arz, np, smem), rrb = reduce_partials(in.arr, np, smem);
        const double s1 = reduce_partials(in.srz, np, smem), s2 = reduce_partials(in.srr, np, smem);
        const double t1 = reduce_partials(in.sbb, np, smem), t2 = reduce_partials(in.t2, np, smem);
        alpha = rzb / pq;
        const double rzn = rzb + alpha * (alpha * s2 - 2.0 * s1);
        double rrn = rrb + alpha * (alpha * t2 - 2.0 * t1);
        if (!(rrn > 0.0))
            rrn = 0.0;
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            scal->rz[it & 1] = rzn;
            scal->rr = rrn;
            scal->pq = pq;
            scal->iters = it;
            scal->alpha_last = alpha;
            if (hist && it - 1 < hist_cap)
                hist[it - 1] = sqrt(rrn);
            if (rrn <= scal->tol2)
                scal->done = 1;
        }
    }
    const int64_t n2 = n >> 1;
    const double2 *z2 = reinterpret_cast<const double2 *>(z), *w2 = reinterpret_cast<const double2 *>(w), *p2 = reinterpret_cast<const double2 *>(pv);
    const double2 *xi2 = reinterpret_cast<const double2 *>(xin);
    double2 *xo2 = reinterpret_cast<double2 *>(xout), *zo2 = reinterpret_cast<double2 *>(zout);
    for (int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x; i < n2; i += vec_stride()) {
        const double2 zv = z2[i], wv = nt_load2(w2 + i), pvv = nt_load2(p2 + i), xv = xi2[i];
        zo2[i] = make_double2(zv.x + alpha * wv.x, zv.y + alpha * wv.y);
        xo2[i] = make_double2(xv.x + alpha * pvv.x, xv.y + alpha * pvv.y);
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const int64_t i = n - 1;
        zout[i] = z[i] + alpha * w[i];
        xout[i] = xin[i] + alpha * pv[i];
    }
}

// K3's scalars alone (the verdict on iteration `it`), for the end of a chunk of that loop: the next pass's prologue would
// take it, but the host polls first.  The pass that follows repeats the same values.
__global__ __launch_bounds__(FV_BLOCK) void pcg_verdict_kernel(int it, const double *__restrict__ part_rz, const double *__restrict__ part_rr, int nparts,
                                                                PcgScalars *__restrict__ scal, double *__restrict__ hist, int64_t hist_cap)
{
    __shared__ double smem[4];
    if (scal->done)
        return;
    const double rzn = reduce_partials(part_rz, nparts, smem);
    const double rrn = reduce_partials(part_rr, nparts, smem);
    if (threadIdx.x == 0) {
        scal->rz[(it + 1) & 1] = rzn;
        scal->rr = rrn;
        scal->iters = it + 1;
        if (hist && it < hist_cap)
            hist[it] = sqrt(rrn);
        if (rrn <= scal->tol2)
            scal->done = 1;
    }
}

// the sparse assembled b of K2S: its support (rows next to a Dirichlet cell or with a source) and where the gather
// blocks leave their partial sums; nblocks = 0: b' is streamed by the vector blocks (or there is none)
struct SparseRhs {
    int64_t m = 0;
    const int32_t *idx = nullptr;
    const double *b = nullptr;
    double *part = nullptr;
    int nblocks = 0;     // extra blocks at the end of the launch that do the gather, or
    int in_vector = 0;   // every vector block gathers its share of the support after its stream (partials: one per block)
};

// K2S — the first K2 of a fixed-dt step that is expected to converge in this iteration (the previous step did):
// besides alpha, x_out = x_in + alpha p, r -= alpha q and the partials of the convergence test, it prepares the NEXT
// step the way pcg_carry_init_kernel would: r0' = r + D (x_out - x_in)/dt is what it leaves in r, p' = M^-1 r0' goes to
// pnext, and the partials of r0'.M^-1 r0', r0'.r0', rhs'.rhs' (rhs' = b' + D x_out/dt) to the spec_* arrays.  If the
// step does converge here, the next step starts straight at its K1.  Streams: x_in, q, M^-1, D, r in (p is M^-1 r0 on the first
// iteration of a step, so it is re-formed in registers and only the sparse-b gather blocks read the stored one); x_out, r, p'
// out: 64 B per row (+8 when a dense b' is streamed too) instead of K2's 56 + K0''s 64.  If it does not,
// pcg_pupdate_kernel<true> takes the D (x_out - x_in)/dt term out of r again before it builds the next direction.
// The storage term D comes as its stream, as a one-byte code per row into a small table, or as one double (StorageArg,
// storage_form below): 8, 1 or 0 B per row.
//
// ZF, the z-form (fv_tune key 36): at the first iteration of a step the direction IS the Jacobi-scaled residual, p = z = M^-1 r0,
// so a state between two one-iteration steps needs one vector, not two: this kernel takes r0 := z / M^-1 from the direction
// it is handed (whoever set the step up: z in instead of r in), and leaves only p' = M^-1 r0' for the next step — no r
// stream out: 56 B per row (x_in, q, M^-1, D, z in; x_out, p' out).  r is written again by whoever needs it after such a step: the K3 / boundary launch of a step that did not
// converge here, or the next step's pcg_carry_init_kernel (zsrc).  The sums of the next step's set-up are taken on r0'
// as computed here (before the product with M^-1), which differs from p' / M^-1 in the last bit at most.  Needs M^-1 > 0
// on every row (minv_positive).
template <int NT, bool ZF> // streaming hints: bit 0 = the read-once inputs x_in, q, M^-1, D, r; bit 1 = x_out and r (p' stays cacheable: the next K1 reads it)
__global__ __launch_bounds__(FV_BLOCK) void pcg_update_spec_kernel(int64_t n, const double *__restrict__ xin, double *__restrict__ xout,
                                                                    double *__restrict__ r, const double *__restrict__ pv,
                                                                    const double *__restrict__ q, const double *__restrict__ minv,
                                                                    StorageArg sa, const double *__restrict__ bprime,
                                                                    double dt, const double *__restrict__ part_pq, int npq,
                                                                    PcgScalars *__restrict__ scal,
                                                                    double *__restrict__ part_rz, double *__restrict__ part_rr,
                                                                    double *__restrict__ pnext, double *__restrict__ spec_rz,
                                                                    double *__restrict__ spec_rr, double *__restrict__ spec_bb,
                                                                    SparseRhs sb, int chain_index)
{
    __shared__ double smem[4];
    __shared__ double dtab[FV_STORAGE_CODES];
    const double *__restrict__ D = sa.D;
    const double Dc = sa.tab.v[0];
    if (sa.code) {
        if (threadIdx.x < FV_STORAGE_CODES)
            dtab[threadIdx.x] = sa.tab.v[threadIdx.x];
        __syncthreads();
    }
    if (scal->done) {
        // A chained step that was already converged at its set-up (the carried residual is within the tolerance): the host
        // flips the two state vectors after every chained step without looking, so the iterate is handed over unchanged.
        // r, p' and the set-up sums stay as they are — they describe exactly this state — and the step counts 0 iterations.
        if (scal->done == 1 && chain_index >= 0 && scal->iters == 0) {
            const int gmain0 = (int)gridDim.x - sb.nblocks;
            if ((int)blockIdx.x < gmain0) {
                // ... and the direction of the set-up (= the scaled residual) goes into the other direction vector too, which the
                // host swaps in at the next chained step without looking: from here on both hold it
                const int64_t n2c = n >> 1;
                const double2 *xi2c = reinterpret_cast<const double2 *>(xin);
                double2 *xo2c = reinterpret_cast<double2 *>(xout);
                const double2 *pv2c = reinterpret_cast<const double2 *>(pv);
                double2 *pn2c = reinterpret_cast<double2 *>(pnext);
                for (int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x; i < n2c; i += (int64_t)gmain0 * FV_BLOCK) {
                    xo2c[i] = xi2c[i];
                    pn2c[i] = pv2c[i];
                }
                if (blockIdx.x == 0 && threadIdx.x == 0) {
                    if (n & 1) {
                        xout[n - 1] = xin[n - 1];
                        pnext[n - 1] = pv[n - 1];
                    }
                    scal->zero_mask |= 1u << chain_index;
                }
            }
        }
        return;
    }
    const double pq = reduce_partials(part_pq, npq, smem);
    if (!(pq > 0.0)) {
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            scal->pq = pq;
            scal->done = 2;
        }
        return;
    }
    const double alpha = scal->rz[0] / pq;
    const int gmain = (int)gridDim.x - sb.nblocks; // the last sb.nblocks blocks of the launch do the sparse-b gather instead
    if ((int)blockIdx.x >= gmain) {
        // rhs'.rhs' = h.h + sum over b's support of b'(2 h + b') with h = D x_out/dt: x_out of these few rows is formed
        // here from x_in and p (the vector blocks of this launch write it, in no particular order)
        const int gb = (int)blockIdx.x - gmain;
        double acc = 0.0;
        for (int64_t k = (int64_t)gb * FV_BLOCK + threadIdx.x; k < sb.m; k += (int64_t)sb.nblocks * FV_BLOCK) {
            const int32_t i = sb.idx[k];
            const double bi = sb.b[i];
            const double xn = xin[i] + alpha * pv[i];
            acc += bi * (2.0 * ((D ? D[i] : sa.code ? dtab[sa.code[i]] : Dc) * (xn / dt)) + bi);
        }
        const double t = block_sum(acc, smem);
        if (threadIdx.x == 0)
            sb.part[gb] = t;
        return;
    }
    const int64_t stride = (int64_t)gmain * FV_BLOCK;
    double arz = 0.0, arr = 0.0, srz = 0.0, srr = 0.0, sbb = 0.0;
    const int64_t n2 = n >> 1;
    const double2 *xi2 = reinterpret_cast<const double2 *>(xin);
    double2 *xo2 = reinterpret_cast<double2 *>(xout);
    double2 *r2 = reinterpret_cast<double2 *>(r);
    const double2 *q2 = reinterpret_cast<const double2 *>(q);
    const double2 *m2 = reinterpret_cast<const double2 *>(minv);
    const double2 *D2 = reinterpret_cast<const double2 *>(D);
    const double2 *b2 = reinterpret_cast<const double2 *>(bprime);
    double2 *pn2 = reinterpret_cast<double2 *>(pnext);
    for (int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x; i < n2; i += stride) {
        double2 xv, pvv, qv, mv, dv = make_double2(Dc, Dc), rv;
        if (sa.code) {
            const unsigned c = reinterpret_cast<const uint16_t *>(sa.code)[i];
            dv = make_double2(dtab[c & 255u], dtab[c >> 8]);
        }
        if (NT & 1) {
            xv = nt_load2(xi2 + i);
            qv = nt_load2(q2 + i);
            mv = nt_load2(m2 + i);
            if (D)
                dv = nt_load2(D2 + i);
            rv = nt_load2((ZF ? reinterpret_cast<const double2 *>(pv) : r2) + i);
        } else {
            xv = xi2[i];
            qv = q2[i];
            mv = m2[i];
            if (D)
                dv = D2[i];
            rv = (ZF ? reinterpret_cast<const double2 *>(pv) : r2)[i];
        }
        if (ZF) {
            // the direction p = z is what was streamed in; the residual it stands for is z / M^-1
            pvv = rv;
            rv = make_double2(pvv.x / mv.x, pvv.y / mv.y);
        } else
            // this is the first iteration of its step, so the direction is p = M^-1 r0 and r still holds r0: the product is
            // formed again (the same multiplication that produced the stored p, bit for bit) instead of streaming p in
            pvv = make_double2(mv.x * rv.x, mv.y * rv.y);
        const double2 bv = bprime ? b2[i] : make_double2(0.0, 0.0);
        const double xnx = xv.x + alpha * pvv.x, xny = xv.y + alpha * pvv.y;
        rv.x -= alpha * qv.x;
        rv.y -= alpha * qv.y;
        arz += rv.x * (mv.x * rv.x) + rv.y * (mv.y * rv.y);
        arr += rv.x * rv.x + rv.y * rv.y;
        // the next step's set-up, on the increment as stored (x_out - x_in after rounding, like K0')
        const double cx = rv.x + dv.x * ((xnx - xv.x) / dt), cy = rv.y + dv.y * ((xny - xv.y) / dt);
        const double hx = bv.x + dv.x * (xnx / dt), hy = bv.y + dv.y * (xny / dt);
        const double zx = mv.x * cx, zy = mv.y * cy;
        if (NT & 2) {
            nt_store2(xo2 + i, make_double2(xnx, xny));
            if (!ZF)
                nt_store2(r2 + i, make_double2(cx, cy));
        } else {
            xo2[i] = make_double2(xnx, xny);
            if (!ZF)
                r2[i] = make_double2(cx, cy);
        }
        pn2[i] = make_double2(zx, zy);
        srz += cx * zx + cy * zy;
        srr += cx * cx + cy * cy;
        sbb += hx * hx + hy * hy;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const int64_t i = n - 1;
        const double r0 = ZF ? pv[i] / minv[i] : r[i];
        const double xn = xin[i] + alpha * (ZF ? pv[i] : minv[i] * r0);
        const double ri = r0 - alpha * q[i];
        const double di = D ? D[i] : sa.code ? dtab[sa.code[i]] : Dc;
        arz += ri * (minv[i] * ri);
        arr += ri * ri;
        const double c = ri + di * ((xn - xin[i]) / dt);
        const double h = (bprime ? bprime[i] : 0.0) + di * (xn / dt);
        const double z = minv[i] * c;
        xout[i] = xn;
        if (!ZF)
            r[i] = c;
        pnext[i] = z;
        srz += c * z;
        srr += c * c;
        sbb += h * h;
    }
    double sgather = 0.0;
    if (sb.in_vector)
        for (int64_t k = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x; k < sb.m; k += stride) {
            const int32_t i = sb.idx[k];
            const double bi = sb.b[i];
            const double xn = xin[i] + alpha * pv[i];
            sgather += bi * (2.0 * ((D ? D[i] : sa.code ? dtab[sa.code[i]] : Dc) * (xn / dt)) + bi);
        }
    const double t0 = block_sum(arz, smem);
    const double t1 = block_sum(arr, smem);
    const double t2 = block_sum(srz, smem);
    const double t3 = block_sum(srr, smem);
    const double t4 = block_sum(sbb, smem);
    const double t5 = sb.in_vector ? block_sum(sgather, smem) : 0.0;
    if (threadIdx.x == 0) {
        part_rz[blockIdx.x] = t0;
        part_rr[blockIdx.x] = t1;
        spec_rz[blockIdx.x] = t2;
        spec_rr[blockIdx.x] = t3;
        spec_bb[blockIdx.x] = t4;
        if (sb.in_vector)
            sb.part[blockIdx.x] = t5;
        if (blockIdx.x == 0)
            scal->pq = pq;
    }
}

int g_k2s_nt = 3; // (frozen at the measured best) streaming hints of K2S (see pcg_update_spec_kernel): the read-once streams bypass the caches, so that p' and x survive for the next K1 (-8 % per step at 216^3, -10 % on a 1.2e7-row block, -1.5 % at 464^3)
int g_zform = 1; // fv_tune key 36: K2S in the z-form (see pcg_update_spec_kernel) where it applies
static auto k2s_kernel(bool zf) -> decltype(&pcg_update_spec_kernel<0, false>)
{
    switch (g_k2s_nt) {
    case 1:
        return zf ? pcg_update_spec_kernel<1, true> : pcg_update_spec_kernel<1, false>;
    case 2:
        return zf ? pcg_update_spec_kernel<2, true> : pcg_update_spec_kernel<2, false>;
    case 3:
        return zf ? pcg_update_spec_kernel<3, true> : pcg_update_spec_kernel<3, false>;
    case 7:
        return zf ? pcg_update_spec_kernel<7, true> : pcg_update_spec_kernel<7, false>;
    default:
        return zf ? pcg_update_spec_kernel<0, true> : pcg_update_spec_kernel<0, false>;
    }
}

// a gather of three scattered doubles per entry is latency-bound: enough blocks to keep every CU busy (64 blocks took 53 us
// for the 1.7e6 entries of the 464^3 box)
constexpr int FV_SPARSE_B_BLOCKS = 1024;
static int sparse_b_grid(int64_t support)
{
    if (support <= 0)
        return 0;
    const int64_t g = (support + FV_BLOCK - 1) / FV_BLOCK;
    return (int)(g < FV_SPARSE_B_BLOCKS ? g : FV_SPARSE_B_BLOCKS);
}

// the sparse-b argument of K2S and the number of extra partials it leaves behind the vector blocks' (g_sparse_b: 1 = extra
// gather blocks, 2 = inside the vector blocks)
static SparseRhs sparse_b_arg(fv_problem *p, int64_t support, int Gv, int *extra_blocks, int *extra_partials)
{
    SparseRhs sb;
    *extra_blocks = *extra_partials = 0;
    if (support <= 0)
        return sb;
    sb.m = support;
    sb.idx = p->bnz_idx.p;
    sb.b = p->b.p;
    sb.part = p->part_bb.p + FV_VEC_PARTIALS + Gv;
    if (g_sparse_b == 2) {
        sb.in_vector = 1;
        *extra_partials = Gv;
    } else {
        sb.nblocks = sparse_b_grid(support);
        *extra_blocks = *extra_partials = sb.nblocks;
    }
    return sb;
}

__global__ __launch_bounds__(FV_BLOCK) void nonzero_flag_kernel(int64_t n, const double *__restrict__ v, int32_t *__restrict__ flag)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i < n)
        flag[i] = v[i] != 0.0;
}

// support of the assembled b (rebuilt after every fv_assemble); *count < 0: b is not sparse enough to bother
static int ensure_b_support(fv_problem *p, int64_t *count)
{
    fv_ctx *ctx = p->ctx;
    if (p->bnz_epoch != p->assemble_epoch) {
        DevBuf<int32_t> flag;
        FV_TRY(flag.alloc(ctx, (size_t)p->n));
        hipLaunchKernelGGL(nonzero_flag_kernel, dim3(fv_blocks(p->n)), dim3(FV_BLOCK), 0, ctx->stream, p->n, (const double *)p->b.p, flag.p);
        FV_LAUNCH_CHECK(ctx);
        FV_TRY(p->bnz_idx.alloc(ctx, (size_t)p->n));
        FV_TRY(fv_compact_flags(ctx, flag.p, p->n, p->bnz_idx.p, &p->bnz_count));
        p->bnz_epoch = p->assemble_epoch;
    }
    *count = (p->bnz_count * 8 <= p->n) ? p->bnz_count : -1;
    return FV_OK;
}

// How K2S gets the storage term D = Ss * volumes.  On a regular grid with a scalar Ss it takes a handful of values (the
// cell volume, halved on the faces of the box, quartered on its edges, an eighth at its corners), on other meshes with even
// volumes one: then K2S streams a one-byte code per row (or nothing) instead of the double, and looks the value up in a
// table of at most FV_STORAGE_CODES doubles in LDS — the same doubles, so nothing else changes.  Built once per D
// (fv_transient_begin bumps storage_epoch; a row block's D is a slice copied at set-up): each pass over D codes the rows
// whose value is in the table and one row that is not offers its value as the next entry; a mesh with more than
// FV_STORAGE_CODES distinct values keeps the stream.
int g_uniform_storage = 1; // fv_tune key 35: 0 = K2S always streams D
__global__ __launch_bounds__(FV_BLOCK) void storage_code_kernel(int64_t n, const double *__restrict__ D, StorageTable tab, int ntab,
                                                                 uint8_t *__restrict__ code, int32_t *__restrict__ claim,
                                                                 double *__restrict__ offered)
{
    bool offered_one = false;
    for (int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x; i < n; i += vec_stride()) {
        const long long bits = __double_as_longlong(D[i]);
        int c = -1;
        for (int k = 0; k < ntab; k++)
            if (__double_as_longlong(tab.v[k]) == bits)
                c = k;
        if (c >= 0)
            code[i] = (uint8_t)c;
        else if (!offered_one) {
            offered_one = true; // one try per thread: whoever gets the claim decides the next table entry
            if (*reinterpret_cast<volatile int32_t *>(claim) == 0 && atomicCAS(claim, 0, 1) == 0)
                *offered = D[i];
        }
    }
}
int fv_storage_form(fv_problem *p, StorageArg *out, int *bytes_saved, bool ignore_switch)
{
    fv_ctx *ctx = p->ctx;
    if (p->dcode_epoch != p->storage_epoch || p->dcode_ptr != p->D.p) {
        p->dcode_n = 0; // 0: keep the stream
        // two samples first: a mesh with uneven volumes shows more than FV_STORAGE_CODES values at once
        bool few = p->n > 0 && p->D.p;
        if (few) {
            const size_t m = (size_t)(p->n < 2048 ? p->n : 2048);
            std::vector<double> h(2 * m);
            FV_TRY(fv_copy(ctx, h.data(), p->D.p, m * sizeof(double)));
            FV_TRY(fv_copy(ctx, h.data() + m, p->D.p + ((size_t)p->n - m) / 2, m * sizeof(double)));
            std::vector<uint64_t> bits(2 * m);
            memcpy(bits.data(), h.data(), 2 * m * sizeof(double));
            std::sort(bits.begin(), bits.end());
            few = std::unique(bits.begin(), bits.end()) - bits.begin() <= FV_STORAGE_CODES;
        }
        if (few) {
            FV_TRY(p->dcode.alloc(ctx, (size_t)p->n + 16));
            FV_TRY(p->dcode.zero(ctx));
            DevBuf<int32_t> claim;
            DevBuf<double> offered;
            FV_TRY(claim.alloc(ctx, 1));
            FV_TRY(offered.alloc(ctx, 1));
            int ntab = 0;
            for (;;) {
                FV_TRY(claim.zero(ctx));
                hipLaunchKernelGGL(storage_code_kernel, dim3(vec_grid(p->n)), dim3(FV_BLOCK), 0, ctx->stream, p->n, (const double *)p->D.p,
                                   p->dtable, ntab, p->dcode.p, claim.p, offered.p);
                FV_LAUNCH_CHECK(ctx);
                int32_t h = 0;
                FV_TRY(fv_copy(ctx, &h, claim.p, sizeof h));
                if (!h) { // every row has its code
                    p->dcode_n = ntab;
                    break;
                }
                if (ntab == FV_STORAGE_CODES)
                    break; // too many distinct values
                FV_TRY(fv_copy(ctx, &p->dtable.v[ntab], offered.p, sizeof(double)));
                ntab++;
            }
            if (p->dcode_n == 0)
                p->dcode.release();
        }
        p->dcode_epoch = p->storage_epoch;
        p->dcode_ptr = p->D.p;
    }
    *out = StorageArg{};
    out->D = p->D.p;
    *bytes_saved = 0;
    if ((g_uniform_storage || ignore_switch) && p->dcode_n > 0) {
        out->D = nullptr;
        out->tab = p->dtable;
        out->code = p->dcode_n > 1 ? p->dcode.p : nullptr; // one value: no stream at all
        *bytes_saved = p->dcode_n > 1 ? 7 : 8;
    }
    return FV_OK;
}

// M^-1 (the Jacobi diagonal's reciprocal) as one-byte codes where it takes few distinct values — a homogeneous conductivity on a
// regular grid: the interior rows share one diagonal, the rows on faces / edges / next to Dirichlet cells a handful more — for the
// vector pass of the many-iteration loop (pcg_update_z_kernel): 1 instead of 8 bytes per row, the same doubles out of the table.
// Built like the storage codes (fv_storage_form), for the M^-1 of (minv_sigma, minv_epoch, storage_epoch).
int g_minv_codes = 1; // fv_tune key 59
static int fv_minv_codes(fv_problem *p, const uint8_t **code, StorageTable *tab)
{
    fv_ctx *ctx = p->ctx;
    *code = nullptr;
    if (!g_minv_codes || !p->minv.p || p->n < 2)
        return FV_OK;
    if (!(p->mvcode_epoch == p->minv_epoch && p->mvcode_sigma == p->minv_sigma && p->mvcode_sepoch == p->storage_epoch && p->mvcode_ptr == p->minv.p)) {
        p->mvcode_n = 0;
        const size_t m = (size_t)(p->n < 2048 ? p->n : 2048);
        std::vector<double> h(2 * m);
        FV_TRY(fv_copy(ctx, h.data(), p->minv.p, m * sizeof(double)));
        FV_TRY(fv_copy(ctx, h.data() + m, p->minv.p + ((size_t)p->n - m) / 2, m * sizeof(double)));
        std::vector<uint64_t> bits(2 * m);
        memcpy(bits.data(), h.data(), 2 * m * sizeof(double));
        std::sort(bits.begin(), bits.end());
        if (std::unique(bits.begin(), bits.end()) - bits.begin() <= FV_STORAGE_CODES) {
            if (!p->mvcode.p || p->mvcode.n < (size_t)p->n + 16)
                FV_TRY(p->mvcode.alloc(ctx, (size_t)p->n + 16));
            FV_TRY(p->mvcode.zero(ctx));
            DevBuf<int32_t> claim;
            DevBuf<double> offered;
            FV_TRY(claim.alloc(ctx, 1));
            FV_TRY(offered.alloc(ctx, 1));
            int ntab = 0;
            for (;;) {
                FV_TRY(claim.zero(ctx));
                hipLaunchKernelGGL(storage_code_kernel, dim3(vec_grid(p->n)), dim3(FV_BLOCK), 0, ctx->stream, p->n, (const double *)p->minv.p, p->mvtable, ntab,
                                   p->mvcode.p, claim.p, offered.p);
                FV_LAUNCH_CHECK(ctx);
                int32_t hc = 0;
                FV_TRY(fv_copy(ctx, &hc, claim.p, sizeof hc));
                if (!hc) {
                    p->mvcode_n = ntab;
                    break;
                }
                if (ntab == FV_STORAGE_CODES)
                    break;
                FV_TRY(fv_copy(ctx, &p->mvtable.v[ntab], offered.p, sizeof(double)));
                ntab++;
            }
        }
        p->mvcode_epoch = p->minv_epoch;
        p->mvcode_sigma = p->minv_sigma;
        p->mvcode_sepoch = p->storage_epoch;
        p->mvcode_ptr = p->minv.p;
    }
    if (p->mvcode_n > 0) {
        *code = p->mvcode.p;
        *tab = p->mvtable;
    }
    return FV_OK;
}

// The z-form divides by M^-1: every row must have one (a free cell without faces and without storage has M^-1 = 0).
// M^-1 = 1 / (diagA + sigma D) is positive and finite on every row for EVERY sigma > 0 when no row has a negative part and each has
// a positive one — a property of the assembly and the storage term, not of the time step: checked once per (assembly, storage)
// and not once per sigma (an adaptive run changes dt at almost every solve; on the Theis problem this check, a memset and a
// synchronous 4-byte copy, came to 2 of 13 launches per solve).  Bit 0: some row fails for sigma > 0; bit 1: for sigma = 0.
__global__ __launch_bounds__(FV_BLOCK) void minv_bad_kernel(int64_t n, const double *__restrict__ diagA, const double *__restrict__ D,
                                                             int32_t *__restrict__ bad)
{
    int b = 0;
    for (int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x; i < n; i += vec_stride()) {
        const double a = diagA[i], d = D ? D[i] : 0.0;
        const bool pos_a = a > 1.0e-300 && a < 1.0e300, pos_d = d > 1.0e-300 && d < 1.0e300;
        if (!((a >= 0.0 && a < 1.0e300) && (d >= 0.0 && d < 1.0e300) && (pos_a || pos_d)))
            b |= 1;
        if (!pos_a)
            b |= 2;
    }
    b |= __shfl_xor(b, 32, 64);
    b |= __shfl_xor(b, 16, 64);
    b |= __shfl_xor(b, 8, 64);
    b |= __shfl_xor(b, 4, 64);
    b |= __shfl_xor(b, 2, 64);
    b |= __shfl_xor(b, 1, 64);
    if (b && (threadIdx.x & 63) == 0)
        atomicOr(bad, b);
}
static int minv_positive(fv_problem *p, bool *ok)
{
    fv_ctx *ctx = p->ctx;
    if (!(p->zf_minv_epoch == p->assemble_epoch && p->zf_storage_epoch == p->storage_epoch)) {
        DevBuf<int32_t> flag;
        FV_TRY(flag.alloc(ctx, 1));
        FV_TRY(flag.zero(ctx));
        hipLaunchKernelGGL(minv_bad_kernel, dim3(vec_grid(p->n)), dim3(FV_BLOCK), 0, ctx->stream, p->n, (const double *)p->diagA.p,
                           p->transient_ready ? (const double *)p->D.p : (const double *)nullptr, flag.p);
        FV_LAUNCH_CHECK(ctx);
        int32_t h = 3;
        FV_TRY(fv_copy(ctx, &h, flag.p, sizeof h));
        p->zf_minv_bits = h;
        p->zf_minv_epoch = p->assemble_epoch;
        p->zf_storage_epoch = p->storage_epoch;
    }
    // (minv_sigma: the shift of the Jacobi diagonal in use — fv_pcg_solve sets it before anyone asks)
    *ok = p->minv_sigma > 0.0 ? !(p->zf_minv_bits & 1) : (p->minv_sigma == 0.0 && !(p->zf_minv_bits & 2));
    return FV_OK;
}

// r = z / M^-1 from the direction vector that holds the scaled residual of the state (fv_problem::z_where), for a consumer
// that wants r in its own array
__global__ __launch_bounds__(FV_BLOCK) void unscale_kernel(int64_t n, const double *__restrict__ z, const double *__restrict__ minv,
                                                            double *__restrict__ r)
{
    for (int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x; i < n; i += vec_stride())
        r[i] = z[i] / minv[i];
}
static int residual_to_r(fv_problem *p)
{
    if (p->z_where == 0)
        return FV_OK;
    fv_ctx *ctx = p->ctx;
    const double *z = p->z_where == 1 ? p->pvec.p : p->z_where == 2 ? p->pnext.p : p->r.p; // (4: in r itself, the many-iteration loop's z-form)
    hipLaunchKernelGGL(unscale_kernel, dim3(vec_grid(p->n)), dim3(FV_BLOCK), 0, ctx->stream, p->n, z, (const double *)p->minv.p, p->r.p);
    FV_LAUNCH_CHECK(ctx);
    p->z_where = 0;
    return FV_OK;
}

extern "C" int fv_fused_form(fv_problem *p, int64_t *launches, int32_t *bytes_per_row, int64_t *bytes_per_launch)
{
    if (!p || !launches || !bytes_per_row || !bytes_per_launch)
        return FV_ERR_ARG;
    *launches = p->fused_launches;
    *bytes_per_row = p->fused_bytes;
    *bytes_per_launch = p->fused_bytes_launch;
    return FV_OK;
}

extern "C" int fv_fused_traversal(fv_problem *p, int32_t *kind)
{
    if (!p || !kind)
        return FV_ERR_ARG;
    *kind = p->fused_chunked ? 1 : 0;
    return FV_OK;
}

extern "C" int fv_loop_form(fv_problem *p, int32_t *bytes_per_row)
{
    if (!p || !bytes_per_row)
        return FV_ERR_ARG;
    *bytes_per_row = p->loop_bytes - ((p->loop_bytes > 0 && p->loop_minv_coded) ? 7 : 0); // (M^-1 as a code byte in the vector pass)
    return FV_OK;
}

extern "C" int fv_step_form(fv_problem *p, int32_t bytes[3], int64_t *solves, int64_t *bytes_total)
{
    if (!p || !bytes)
        return FV_ERR_ARG;
    for (int k = 0; k < 3; k++)
        bytes[k] = p->ploop_bytes[k];
    if (solves)
        *solves = p->ploop_solves;
    if (bytes_total)
        *bytes_total = p->bytes_total;
    return FV_OK;
}

extern "C" int fv_update_form(fv_problem *p, int32_t *bytes_per_row)
{
    if (!p || !bytes_per_row)
        return FV_ERR_ARG;
    *bytes_per_row = p->k2s_bytes;
    return FV_OK;
}

// K3.  UNSPEC: the K2 before it was pcg_update_spec_kernel and the step did not converge there: r carries the next
// step's D (x_out - x_in)/dt term, which is taken out again here (same expression, same operands).
template <bool UNSPEC>
__global__ __launch_bounds__(FV_BLOCK) void pcg_pupdate_kernel(int64_t n, int it, double *__restrict__ r, const double *__restrict__ minv,
                                                                double *__restrict__ pv, const double *__restrict__ part_rz,
                                                                const double *__restrict__ part_rr, int nparts, PcgScalars *__restrict__ scal,
                                                                double *__restrict__ hist, int64_t hist_cap, const double *__restrict__ xin,
                                                                const double *__restrict__ xout, const double *__restrict__ D, double dt,
                                                                int chain_index = -1, int force_unconverged = 0,
                                                                const double *__restrict__ zsrc = nullptr)
{
    // zsrc (UNSPEC only): the K2S was a z-form one: what it left is p' = M^-1 r0' in zsrc and nothing in r
    __shared__ double smem[4];
    {
        // (done = 3 with THIS step's index was written by block 0 of this very launch: a block that starts after block 0 has finished
        // — a grid that is not resident all at once: other processes or ranks on the device — must still do its share of the
        // fall-back; it takes the same decision from the same sums)
        const int d0 = *reinterpret_cast<volatile int32_t *>(&scal->done);
        if (d0 && !(d0 == 3 && chain_index >= 0 && *reinterpret_cast<volatile int32_t *>(&scal->chain_step) == chain_index))
            return;
    }
    const double rzn = reduce_partials(part_rz, nparts, smem);
    const double rrn = reduce_partials(part_rr, nparts, smem);
    const double beta = rzn / scal->rz[it & 1];
    // same value in every block: the search direction is not needed any more (force_unconverged: fault injection for the
    // chained-step tests, fv_tune key 14 — constant forcing never makes a later step need MORE iterations by itself)
    const bool converged = rrn <= scal->tol2 && !force_unconverged;
    const int64_t n2 = converged ? 0 : (n >> 1);
    double2 *r2 = reinterpret_cast<double2 *>(r);
    const double2 *m2 = reinterpret_cast<const double2 *>(minv);
    double2 *p2 = reinterpret_cast<double2 *>(pv);
    const double2 *xi2 = reinterpret_cast<const double2 *>(xin);
    const double2 *xo2 = reinterpret_cast<const double2 *>(xout);
    const double2 *D2 = reinterpret_cast<const double2 *>(D);
    for (int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x; i < n2; i += vec_stride()) {
        double2 rv;
        const double2 mv = m2[i];
        if (UNSPEC && zsrc) {
            const double2 zv = reinterpret_cast<const double2 *>(zsrc)[i];
            rv = make_double2(zv.x / mv.x, zv.y / mv.y);
        } else
            rv = r2[i];
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
        double ri = (UNSPEC && zsrc) ? zsrc[i] / minv[i] : r[i];
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
        else if (chain_index >= 0) { // unpolled chain: stop here; the host resumes this step at iteration it + 1
            scal->chain_step = chain_index;
            __threadfence();
            scal->done = 3;
        }
    }
}

// Row-block bursts with merged collectives: the verdict on the PREVIOUS chained step (pcg_pupdate_kernel<true> on that
// step's vectors) and the scalars of the step that has just done its K1 (pcg_init_finalize_kernel) in one launch, after
// the 6-double all-reduce: red[1..2] the previous step's r.M^-1 r and r.r, red[3..5] this step's r.M^-1 r, r.r, rhs.rhs.
// Every block takes the same verdict from values no block of this launch writes (red, the previous step's tol2 by index
// parity); block 0 then either closes the chain (done = 3) or writes the new step's scalars, which the other blocks never
// read on that path.
struct BoundarySums { // the previous step's r.M^-1 r, r.r (nprev pieces each) and the new step's r.M^-1 r, r.r (nnext), rhs.rhs (nnext_bb)
    const double *prev_rz, *prev_rr, *next_rz, *next_rr, *next_bb;
    int nprev, nnext, nnext_bb;
};
__global__ __launch_bounds__(FV_BLOCK) void pcg_chain_boundary_kernel(int64_t n, double *__restrict__ r, const double *__restrict__ minv,
                                                                       double *__restrict__ pv, BoundarySums sums, double rtol,
                                                                       PcgScalars *scal, const double *__restrict__ xin,
                                                                       const double *__restrict__ xout, const double *__restrict__ D, double dt,
                                                                       int prev_index, int force_unconverged,
                                                                       const double *__restrict__ zsrc = nullptr)
{
    __shared__ double smem[4];
    {
        const int d0 = *reinterpret_cast<volatile int32_t *>(&scal->done); // (as in pcg_pupdate_kernel: block 0 of this launch may have closed the chain already)
        if (d0 && !(d0 == 3 && *reinterpret_cast<volatile int32_t *>(&scal->chain_step) == prev_index))
            return;
    }
    const double rzn = reduce_partials(sums.prev_rz, sums.nprev, smem);
    const double rrn = reduce_partials(sums.prev_rr, sums.nprev, smem);
    const bool converged = rrn <= scal->tol2x[prev_index & 1] && !force_unconverged;
    if (!converged) { // as pcg_pupdate_kernel<true>: take the next step's D (x_out - x_in)/dt out of r again, p = z + beta p
        const double beta = rzn / scal->rz[0];
        const int64_t n2 = n >> 1;
        double2 *r2 = reinterpret_cast<double2 *>(r);
        const double2 *m2 = reinterpret_cast<const double2 *>(minv);
        double2 *p2 = reinterpret_cast<double2 *>(pv);
        const double2 *xi2 = reinterpret_cast<const double2 *>(xin);
        const double2 *xo2 = reinterpret_cast<const double2 *>(xout);
        const double2 *D2 = reinterpret_cast<const double2 *>(D);
        for (int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x; i < n2; i += vec_stride()) {
            double2 rv;
            const double2 mv = m2[i];
            if (zsrc) { // z-form K2S before this launch: see pcg_pupdate_kernel
                const double2 zv = reinterpret_cast<const double2 *>(zsrc)[i];
                rv = make_double2(zv.x / mv.x, zv.y / mv.y);
            } else
                rv = r2[i];
            const double2 a = xi2[i], b = xo2[i], dv = D2[i];
            rv.x -= dv.x * ((b.x - a.x) / dt);
            rv.y -= dv.y * ((b.y - a.y) / dt);
            r2[i] = rv;
            double2 pvv = p2[i];
            pvv.x = mv.x * rv.x + beta * pvv.x;
            pvv.y = mv.y * rv.y + beta * pvv.y;
            p2[i] = pvv;
        }
        if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
            const int64_t i = n - 1;
            double ri = zsrc ? zsrc[i] / minv[i] : r[i];
            ri -= D[i] * ((xout[i] - xin[i]) / dt);
            r[i] = ri;
            pv[i] = minv[i] * ri + beta * pv[i];
        }
    }
    if (blockIdx.x != 0)
        return;
    if (!converged) {
        if (threadIdx.x == 0) {
            scal->rz[1] = rzn;
            scal->rr = rrn;
            scal->iters = 1;
            scal->chain_step = prev_index;
            __threadfence();
            scal->done = 3;
        }
        return;
    }
    const double rz0 = reduce_partials(sums.next_rz, sums.nnext, smem);
    const double rr0 = reduce_partials(sums.next_rr, sums.nnext, smem);
    const double bb = reduce_partials(sums.next_bb, sums.nnext_bb, smem);
    if (threadIdx.x == 0) {
        const double tol2 = rtol * rtol * bb;
        scal->rz[0] = rz0;
        scal->rz[1] = 0.0;
        scal->rr = rr0;
        scal->bnorm2 = bb;
        scal->tol2 = tol2;
        scal->tol2x[(prev_index + 1) & 1] = tol2;
        scal->pq = 0.0;
        scal->iters = 0;
        scal->done = (rr0 <= tol2) ? 1 : 0;
    }
}

int fv_pcg_prepare(fv_problem *p)
{
    fv_ctx *ctx = p->ctx;
    if (p->r.p)
        return FV_OK;
    const size_t n = (size_t)p->n + (size_t)p->nhalo + FV_VEC_PAD; // slack for double2 tails and whole x lines; halo slots of a row block
    // (the vectors every step writes first — fv_place.hip —: the ones that are only read take what they leave)
    FV_TRY(fv_vec_alloc(p, p->pvec, n, true));
    FV_TRY(fv_vec_alloc(p, p->q, n, true));
    FV_TRY(fv_vec_alloc(p, p->r, n, true));
    FV_TRY(fv_vec_alloc(p, p->minv, n, false));
    FV_TRY(fv_vec_alloc(p, p->rhs, n, false));
    FV_TRY(fv_vec_alloc(p, p->tmp, n, false));
    // two launches (sliced-DIA part + CSR part) may each leave up to FV_MAX_PARTIALS partials
    FV_TRY(p->part_pq.alloc(ctx, 4 * FV_MAX_PARTIALS)); // distributed: interior + boundary pass, each DIA + CSR
    FV_TRY(p->part_rz.alloc(ctx, 2 * FV_VEC_PARTIALS));
    FV_TRY(p->part_rr.alloc(ctx, 2 * FV_VEC_PARTIALS));
    FV_TRY(p->part_bb.alloc(ctx, 3 * FV_VEC_PARTIALS)); // + the pieces of K2S's sparse-b gather blocks behind the speculative half
    FV_TRY(p->scal.alloc(ctx, 1));
    FV_TRY(p->scal.zero(ctx));
    FV_TRY(p->pvec.zero(ctx));
    FV_TRY(p->q.zero(ctx));
    return FV_OK;
}

static inline int fv_step_precond_of(const fv_problem *p, const PcgSystem &sys) { return sys.implicit_step ? fv_step_precond(p) : p->precond; }

// Bytes the most recent product's storage form moves per launch (fv_spmv_form), for fv_problem::bytes_total
static int64_t k1_form_bytes(fv_problem *p)
{
    int32_t form = 0;
    int64_t b = 0;
    return fv_spmv_form(p, &form, &b) == FV_OK ? b : 0;
}

int fv_ploop_flush_pending(fv_problem *p)
{
    fv_problem::PlPending &pd = p->pl_pending;
    if (!pd.valid)
        return FV_OK;
    fv_ctx *ctx = p->ctx;
    pd.valid = false;
    hipLaunchKernelGGL(pcg_ploop_flush_kernel, dim3(vec_grid(p->n)), dim3(FV_BLOCK), 0, ctx->stream, p->n, pd.z, pd.w, pd.pp, pd.xin, pd.x, p->r.p, p->scal.p,
                       FusedSums{}, 1, 0, (double *)nullptr, (int64_t)0);
    FV_LAUNCH_CHECK(ctx);
    p->z_where = 4;
    return FV_OK;
}

// the code byte of pcg_carry_flush_kernel (cached per storage and assembly epoch); *ok = false: D has too many distinct values
static int ensure_carry_codes(fv_problem *p, bool *ok, StorageTable *tab)
{
    fv_ctx *ctx = p->ctx;
    StorageArg sa{};
    int saved = 0;
    FV_TRY(fv_storage_form(p, &sa, &saved, true));
    *ok = sa.D == nullptr;
    if (!*ok)
        return FV_OK;
    *tab = sa.tab;
    if (p->vcode_sepoch != p->storage_epoch || p->vcode_aepoch != p->assemble_epoch || !p->vcode.p) {
        if (!p->vcode.p)
            FV_TRY(p->vcode.alloc(ctx, (size_t)p->n + 16));
        hipLaunchKernelGGL(carry_code_kernel, dim3(vec_grid(p->n)), dim3(FV_BLOCK), 0, ctx->stream, p->n, sa.code, (const double *)p->b.p, p->vcode.p);
        FV_LAUNCH_CHECK(ctx);
        p->vcode_sepoch = p->storage_epoch;
        p->vcode_aepoch = p->assemble_epoch;
    }
    return FV_OK;
}

int fv_pcg_solve(fv_problem *p, double *x, const PcgSystem &sys, double rtol, int64_t maxiter, fv_solve_info *info, bool time_it)
{
    fv_ctx *ctx = p->ctx;
    p->resume.ok = false; // any solve overwrites what a finished fixed-dt run left for its successor (the run sets it again at its end)
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
    { // small systems: the whole solve in one launch (fv_small.hip)
        bool handled = false;
        FV_TRY(fv_pcg_small(p, x, sys, rtol, maxiter, info, time_it, &handled));
        if (handled)
            return FV_OK;
    }
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
    const bool chained = sys.chain_index >= 0; // caller guarantees: spec_valid, speculation on, Jacobi, one-iteration regime
    const bool resume = sys.resume_it > 0;
    // speculative set-up left behind by the previous step's K2S (pcg_update_spec_kernel): r, p' and the partials are ready
    const bool use_spec = !resume && sys.use_spec && p->spec_valid && sys.implicit_step && !compute_minv;
    p->spec_valid = false;
    if (chained && !use_spec) {
        fv_set_error(ctx, "internal: chained step without a prepared set-up");
        return FV_ERR_STATE;
    }
    // ... and whether this step's first K2 should prepare the next step the same way
    const bool speculate = !resume && sys.speculate && sys.x_next && sys.implicit_step && !sys.b_times_D && !compute_minv && !g_fuse_init &&
                           fv_step_precond(p) != FV_PRECOND_AMG && p->last_iters == 1 && maxiter > 0;
    if (speculate && !p->pnext.p)
        FV_TRY(fv_vec_alloc(p, p->pnext, (size_t)n + (size_t)p->nhalo + FV_VEC_PAD, true));
    // K2S without the b' stream when b' (the assembled b) is sparse: its share of rhs.rhs comes from a gather over its support
    int64_t bsupport = -1;
    if (speculate && g_sparse_b && sys.rhs == p->b.p)
        FV_TRY(ensure_b_support(p, &bsupport));
    int Gx = 0, Gs = 0; // extra blocks of the K2S launch, extra rhs.rhs partials it leaves
    const SparseRhs sbarg = sparse_b_arg(p, bsupport, Gv, &Gx, &Gs);
    StorageArg sarg{};
    int Dsaved = 0;
    if (speculate)
        FV_TRY(fv_storage_form(p, &sarg, &Dsaved, false));
    bool zf = false; // this step's K2S in the z-form
    if (speculate && g_zform)
        FV_TRY(minv_positive(p, &zf));
    // The fused step of the one-iteration regime (fv_fused.hip) where the chained step can run it; was_vready: the previous
    // step was such a launch and has left v, the product's sums and the set-up sums of THIS step in its own arrays
    const bool was_vready = p->vready && use_spec;
    p->vready = false;
    const bool fused = chained && speculate && zf && folded && (sarg.D == nullptr || fv_fused_streams_storage(p)) && (bsupport >= 0 || !sys.rhs) &&
                       sys.x_next && fv_fused_applicable(p, sigma);
    {
        static int trace = getenv("FV_TRACE_FUSED") ? atoi(getenv("FV_TRACE_FUSED")) : 0;
        if (trace > 0 && sys.implicit_step) {
            trace--;
            fprintf(stderr, "[fvhip] step: chained %d (index %d) speculate %d use_spec %d zf %d folded %d D-stream %d b-support %lld x_next %d applicable %d last_iters %lld -> fused %d (v ready %d)\n",
                    (int)chained, sys.chain_index, (int)speculate, (int)use_spec, (int)zf, folded ? 1 : 0, sarg.D ? 1 : 0, (long long)bsupport, sys.x_next ? 1 : 0,
                    (int)fv_fused_applicable(p, sigma), (long long)p->last_iters, (int)fused, (int)was_vready);
        }
    }
    if (chained && sys.chain_index > 0 && was_vready && !fused) {
        fv_set_error(ctx, "internal: a burst left the fused regime in its middle");
        return FV_ERR_STATE;
    }
    FusedSums fin{};
    if (was_vready) {
        fin = fv_fused_sums(p, p->vready_parity);
        fin.nvec = p->vready_counts[0];
        fin.nbb = p->vready_counts[1];
        fin.npq = p->vready_counts[2];
    }
    const double *in_rz = p->part_rz.p, *in_rr = p->part_rr.p, *in_bb = p->part_bb.p;
    int in_nbb = -1;
    // a pending update of the previous step's loop (fv_problem::pl_pending): this step's carried set-up applies it, anything else flushes it first
    bool take_pending = false;
    StorageTable vtab{};
    if (p->pl_pending.valid) {
        bool codes = false;
        if (!resume && !use_spec && sys.implicit_step && sys.carry_prev && !compute_minv && !sys.b_times_D && x == p->pl_pending.x && maxiter > 0 &&
            sys.rhs == p->b.p && fv_step_precond_of(p, sys) != FV_PRECOND_AMG && fv_ploop_applicable(p, sigma, folded != nullptr))
            FV_TRY(ensure_carry_codes(p, &codes, &vtab));
        take_pending = codes;
        if (!take_pending) {
            FV_TRY(fv_ploop_flush_pending(p));
            p->bytes_total += 48 * n;
        }
    }
    const bool take_pending_acct = take_pending;
    if (resume) {
        // nothing to set up: r, p and the scalars are those of the interrupted solve
    } else if (use_spec) {
        in_nbb = Gv + p->spec_extra_bb;
        p->pvec.swap(p->pnext);
        if (p->z_where == 1 || p->z_where == 2)
            p->z_where = 3 - p->z_where; // the scaled residual a z-form K2S left moves with its vector
        in_rz += FV_VEC_PARTIALS;
        in_rr += FV_VEC_PARTIALS;
        in_bb += FV_VEC_PARTIALS;
        if (was_vready) { // ... a fused launch left the set-up sums in its own arrays
            in_rz = fin.srz;
            in_rr = fin.srr;
            in_bb = fin.sbb;
            Ginit = fin.nvec;
            in_nbb = fin.nbb;
        }
        if (!zf)
            FV_TRY(residual_to_r(p)); // the first K2 of this step reads r
    } else if (take_pending) {
        // the previous step's last update and this step's carried set-up in one pass; only z0 (pvec) is written: the loop that follows
        // is the one-launch kind (it reads nothing else)
        const fv_problem::PlPending pd = p->pl_pending;
        p->pl_pending.valid = false;
        hipLaunchKernelGGL(pcg_carry_flush_kernel, dim3(Gv), dim3(FV_BLOCK), 0, ctx->stream, n, pd.z, pd.w, pd.pp, x, sys.carry_prev, (const double *)p->minv.p,
                           (const uint8_t *)p->vcode.p, vtab, (const double *)p->b.p, sys.dt, (const PcgScalars *)p->scal.p, p->pvec.p, p->part_rz.p,
                           p->part_rr.p, p->part_bb.p);
        p->z_where = 0;
    } else if (sys.implicit_step && sys.carry_prev && !compute_minv && !sys.b_times_D) {
        const double *zsrc = p->z_where == 1 ? p->pvec.p : p->z_where == 2 ? p->pnext.p : p->z_where == 4 ? p->r.p : nullptr;
        hipLaunchKernelGGL(pcg_carry_init_kernel, dim3(Gv), dim3(FV_BLOCK), 0, ctx->stream, n, sys.rhs, (const double *)p->D.p, sys.dt,
                           (const double *)x, sys.carry_prev, (const double *)p->minv.p, p->r.p, p->pvec.p, p->part_rz.p, p->part_rr.p,
                           p->part_bb.p, zsrc);
        p->z_where = 0;
    } else if (sys.implicit_step && g_fuse_init && g_spmv_form == 2 && !p->lean) { // (lean: the set-up epilogue runs in the seven-diagonal slices, whose values a lean problem does not fill)
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
    // fv_step_form's running total: the bytes this solve's launches must move with every array of every launch touched once — the
    // carried set-ups by their streams (64 / 65 n), any other set-up as a product of the form that ran + a 56 n vector pass
    int64_t acct = 0;
    if (resume || use_spec)
        acct = 0;
    else if (take_pending_acct)
        acct = 65 * n;
    else if (sys.implicit_step && sys.carry_prev && !compute_minv && !sys.b_times_D)
        acct = 64 * n;
    else if (sys.x0_zero && !sys.implicit_step)
        acct = 48 * n;
    else
        acct = k1_form_bytes(p) + 56 * n;
    if (!resume && !use_spec)
        p->z_where = 0; // every other set-up has written r
    if (take_pending_acct)
        p->z_where = 1; // ... but pcg_carry_flush_kernel writes z0 = M^-1 r0 (pvec) alone: should the step be converged at its set-up, the next
                        // carried set-up finds the residual there (found by tools/ploop_fuzz.py: loose steps behind a several-iteration step read a stale r)
    if (!resume)
        if (!(chained && sys.chain_index > 0 && (g_defer_reduce || was_vready))) // ... unless the previous chained step's boundary launch wrote them (or this step's fused launch will)
            hipLaunchKernelGGL(pcg_init_finalize_kernel, dim3(1), dim3(FV_BLOCK), 0, ctx->stream, in_rz, in_rr, in_bb, Ginit, rtol, p->scal.p,
                               in_nbb, chained && sys.chain_index > 0 ? 1 : 0);
    FV_LAUNCH_CHECK(ctx);
    PcgScalars *hs = reinterpret_cast<PcgScalars *>(ctx->pinned);
    int64_t it = resume ? sys.resume_it : 0;
    constexpr int64_t MAX_CHUNK = 32;
    bool polled = false;
    if ((sys.implicit_step ? fv_step_precond(p) : p->precond) == FV_PRECOND_AMG) {
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
    if (chained) {
        chunk = 1;
        maxiter = 1;
    }
    const int64_t kprof = chained ? sys.chain_index : 0; // profiling slots of this launch set
    if (p->profile && p->prof_ev.empty()) {
        p->prof_ev.resize((size_t)(6 * MAX_CHUNK));
        for (hipEvent_t &e : p->prof_ev)
            FV_HIP(ctx, hipEventCreate(&e));
    }
#define FV_PROF(idx)                                                                                            \
    if (p->profile && ((idx) < 2 || p->profile_level == 1))                                                     \
    FV_HIP(ctx, hipEventRecord(p->prof_ev[(size_t)(6 * (k + kprof) + (idx))], ctx->stream))
    // ---- the many-iteration loop as ONE launch per iteration (fv_ploop_pass, fv_fused.hip) on operators it serves: launch j takes the
    // verdict on iterate j and alpha, beta from the sums launch j - 1 left, applies the update of iteration j - 1 and forms the next
    // direction and product; the last update is flushed when the loop has stopped.  z_j in zb[j & 1] (z_0: the set-up's pvec), p_j in
    // pb[j & 1], w_j in wb[j & 1]; x in place (its first update reads the old state and writes x_next where the state ping-pongs).
    bool ploop = false;
    if (!resume && !use_spec && !speculate && !chained && maxiter > 0 && fv_step_precond_of(p, sys) != FV_PRECOND_AMG && !p->dist && !sys.x0_src) {
        bool mpos = false;
        FV_TRY(minv_positive(p, &mpos));
        ploop = mpos && fv_ploop_applicable(p, sigma, folded != nullptr);
    }
    const bool took_pending = take_pending;
    if (take_pending && !ploop) {
        fv_set_error(ctx, "internal: a pending update was taken by a set-up whose loop cannot use it");
        return FV_ERR_STATE;
    }
    if (ploop) {
        const size_t nv = (size_t)n + (size_t)p->nhalo + FV_VEC_PAD;
        if (!p->pnext.p)
            FV_TRY(fv_vec_alloc(p, p->pnext, nv, true));
        if (!p->zalt.p) {
            FV_TRY(fv_vec_alloc(p, p->zalt, nv, true));
            FV_TRY(p->zalt.zero(ctx));
        }
        if (!p->walt.p) {
            FV_TRY(fv_vec_alloc(p, p->walt, nv, true));
            FV_TRY(p->walt.zero(ctx));
        }
        double *zb[2] = {p->r.p, p->zalt.p}, *pb[2] = {p->pvec.p, p->pnext.p}, *wb[2] = {p->q.p, p->walt.p};
        double *xdst = sys.x_next ? sys.x_next : x;
        auto zof = [&](int64_t j) -> double * { return j == 0 ? pb[0] : zb[j & 1]; };
        int64_t j = 0; // the next launch
        while (j < maxiter) {
            int64_t m = chunk + (j == 0 ? 1 : 0); // (the launch that finds the last iterate converged does no pass: one more than the iterations expected)
            if (m > MAX_CHUNK)
                m = MAX_CHUNK;
            if (m > maxiter - j)
                m = maxiter - j;
            for (int64_t k = 0; k < m; k++, j++) {
                FV_PROF(0);
                if (j == 0)
                    FV_TRY(fv_ploop_pass(p, 0, folded, pb[0], nullptr, nullptr, x, xdst, nullptr, nullptr, wb[0]));
                else
                    FV_TRY(fv_ploop_pass(p, (int)j, folded, zof(j - 1), wb[(j - 1) & 1], pb[(j - 1) & 1], j == 1 ? x : xdst, xdst, zb[j & 1], pb[j & 1], wb[j & 1]));
                FV_PROF(1);
            }
            FV_HIP(ctx, hipMemcpyAsync(hs, p->scal.p, sizeof(PcgScalars), hipMemcpyDeviceToHost, ctx->stream));
            FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
            polled = true;
            if (p->profile) { // only launches that did real work
                const int64_t live = hs->done == 1 ? (int64_t)hs->iters - (j - m) : m;
                for (int64_t k = 0; k < m && k < live; k++) {
                    float ms = 0.f;
                    FV_HIP(ctx, hipEventElapsedTime(&ms, p->prof_ev[(size_t)(6 * k)], p->prof_ev[(size_t)(6 * k + 1)]));
                    p->prof_ms[0] += ms;
                    p->prof_launches[0]++;
                }
            }
            if (hs->done)
                break;
            if (chunk < MAX_CHUNK)
                chunk *= 2;
        }
        // the pending update: iterate `it` = z_{it-1} + alpha w_{it-1}, x likewise
        int64_t itf = -1;
        int have_alpha = 0;
        if (hs->done == 1 && hs->iters >= 1) {
            itf = hs->iters;
            have_alpha = 1;
        } else if (hs->done == 0 && j >= 1)
            itf = j; // out of iterations: every launch ran its pass
        bool deferred = false;
        if (itf >= 1 && have_alpha && sys.defer_flush && sys.implicit_step && xdst != x && sys.rhs == p->b.p && !sys.b_times_D) { // (the flush + set-up kernel reads the assembled b by its support)
            // the next call is the carried step behind this one: its set-up applies the update (pcg_carry_flush_kernel)
            bool codes = false;
            StorageTable tmp{};
            FV_TRY(ensure_carry_codes(p, &codes, &tmp));
            if (codes) {
                fv_problem::PlPending &pd = p->pl_pending;
                pd.valid = true;
                pd.z = zof(itf - 1);
                pd.w = wb[(itf - 1) & 1];
                pd.pp = pb[(itf - 1) & 1];
                pd.xin = itf == 1 ? x : xdst;
                pd.x = xdst;
                deferred = itf >= 2; // (a one-iteration step's update reads the old state and writes the other vector: not an in-place update, flush it)
                if (!deferred)
                    pd.valid = false;
            }
        }
        if (itf >= 1 && !deferred) {
            FusedSums fin2 = fv_fused_sums(p, (int)((itf - 1) & 1));
            fin2.t2 = fin2.sbb + FV_FUSED_PARTS;
            fin2.npq = p->ploop_grid;
            hipLaunchKernelGGL(pcg_ploop_flush_kernel, dim3(Gv), dim3(FV_BLOCK), 0, ctx->stream, n, (const double *)zof(itf - 1), (const double *)wb[(itf - 1) & 1],
                               (const double *)pb[(itf - 1) & 1], (const double *)(itf == 1 ? x : xdst), xdst, p->r.p, p->scal.p, fin2, have_alpha, (int)itf,
                               p->hist.p, p->hist_cap);
            FV_LAUNCH_CHECK(ctx);
            if (!have_alpha) {
                FV_HIP(ctx, hipMemcpyAsync(hs, p->scal.p, sizeof(PcgScalars), hipMemcpyDeviceToHost, ctx->stream));
                FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
            }
        }
        if (hs->iters >= 1 || (j >= 1 && hs->done != 1)) { // (a solve that sat converged at its set-up: its launches stopped at their prologues, nothing moved)
            const int64_t full = hs->done == 1 ? (int64_t)hs->iters - 1 : j - 1; // launches beyond the first that ran their pass
            acct += (int64_t)(p->loop_bytes == 67 ? 19 : 41) * n + (full > 0 ? full : 0) * (int64_t)p->loop_bytes * n + ((itf >= 1 && !deferred) ? 48 * n : 0);
        }
        p->ploop_solves++;
        p->ploop_bytes[0] = took_pending ? 65 : ((sys.implicit_step && sys.carry_prev && !compute_minv && !sys.b_times_D) ? 64 : 0);
        p->ploop_bytes[1] = p->loop_bytes == 67 ? 19 : 41;
        p->ploop_bytes[2] = (itf >= 1 && !deferred) ? 48 : 0;
        p->loop_minv_coded = false;
        maxiter = 0; // (skip the loop below)
    }
    int zloop = 0, zr_is_z = 0; // the many-iteration loop through the fused kernel: 0 no, 1 to be decided after the first product, 2 yes
    bool wloop_first = true, wloop_lag = false; // ... its first vector update is the classic one; afterwards x lags one pass behind
    while (it < maxiter) {
        const int64_t m = (maxiter - it < chunk) ? (maxiter - it) : chunk;
        int32_t iters_before = 0;
        if (p->profile && polled)
            iters_before = hs->iters;
        if (fused) {
            const int64_t k = 0;
            FV_TRY(fv_fused_prepare(p));
            if (!was_vready) { // entry: the product the classic way, then v from it; the sums of this step's set-up are where K2S left them
                int npq0 = 0;
                FV_TRY(spmv_apply(p, p->pvec.p, p->q.p, sig_mv, folded, SPMV_DOT, p->part_pq.p, nullptr, true, &npq0));
                FV_TRY(fv_fused_enter(p, sigma));
                fin.arz = p->part_rz.p;
                fin.arr = p->part_rr.p;
                fin.srz = const_cast<double *>(in_rz);
                fin.srr = const_cast<double *>(in_rr);
                fin.sbb = const_cast<double *>(in_bb);
                fin.pq = p->part_pq.p;
                fin.nvec = Gv;
                fin.nbb = in_nbb >= 0 ? in_nbb : Gv;
                fin.npq = npq0;
            }
            const int fmode = (was_vready && sys.chain_index > 0) ? 1 : 0;
            FusedSums fout{};
            FV_PROF(0);
            FV_TRY(fv_fused_step(p, x, sys.x_next, sigma, sys.dt, rtol, sys.chain_index, fmode, fin,
                                 fmode == 1 && sys.chain_index - 1 == g_chain_test_break, folded, bsupport, &fout));
            FV_PROF(1);
            FV_PROF(2);
            FV_PROF(3);
            FV_PROF(4);
            if (!sys.chain_more) // the burst's last step: its verdict now, so that the host's poll sees a finished state
                hipLaunchKernelGGL(pcg_pupdate_kernel<true>, dim3(Gv), dim3(FV_BLOCK), 0, ctx->stream, n, 0, p->r.p, (const double *)p->minv.p,
                                   p->pvec.p, (const double *)fout.arz, (const double *)fout.arr, fout.nvec, p->scal.p, p->hist.p, p->hist_cap,
                                   (const double *)x, (const double *)sys.x_next, (const double *)p->D.p, sys.dt, sys.chain_index,
                                   (sys.chain_index == g_chain_test_break) ? 1 : 0, (const double *)p->pnext.p);
            FV_PROF(5);
            FV_LAUNCH_CHECK(ctx);
            p->qv.swap(p->qv2);
            p->vready = true;
            p->vready_parity = sys.chain_index & 1;
            p->vready_counts[0] = fout.nvec;
            p->vready_counts[1] = fout.nbb;
            p->vready_counts[2] = fout.npq;
            p->fused_launches++;
            p->bytes_total += acct + p->fused_bytes_launch + (was_vready ? 0 : k1_form_bytes(p) + 32 * n);
            p->last_iters = 1;
            p->spec_valid = true;
            p->z_where = 2;
            return FV_OK;
        }
        for (int64_t k = 0; k < m; k++) {
            const int iter = (int)(it + k);
            // The many-iteration loop through the fused kernel: the direction update of K3 and the product in one pass, the
            // residual kept as z = M^-1 r in the array r between the passes (pcg_update_z_kernel).  Entered at a solve's first
            // iteration (the set-up left r and p = M^-1 r; the classic K1 forms that product and establishes the storage form).
            if (iter == 0 && !resume && !speculate && !chained && fv_step_precond_of(p, sys) != FV_PRECOND_AMG && !p->dist)
                zloop = 1; // (decided for good once the first product has shown the form: see below)
            FV_PROF(0);
            int npq = 0;
            if (zloop == 2) {
                // (from the loop's second pass on the x-update of the iteration before lags one pass behind: this pass applies it)
                FV_TRY(fv_fused_iteration(p, iter - 1, folded, p->part_rz.p, p->part_rr.p, Gv, &npq, sys.x_next ? sys.x_next : x, wloop_lag));
                p->pvec.swap(p->pnext);
            } else
                FV_TRY(spmv_apply(p, p->pvec.p, p->q.p, sig_mv, folded, SPMV_DOT, p->part_pq.p, nullptr, true, &npq));
            FV_PROF(1);
            FV_PROF(2);
            if (zloop == 1) {
                // the loop keeps z = M^-1 r and recovers r as z / M^-1: every row needs M^-1 > 0 (a free row whose diagonal is
                // zero — all conductances 0, no storage — has M^-1 = 0 by definition and must stay with the classic loop,
                // which leaves that row alone).  Cached per Jacobi diagonal like the z-form K2S's check.
                bool mpos = false;
                FV_TRY(minv_positive(p, &mpos));
                zloop = mpos && fv_fused_iteration_applicable(p, sigma, folded != nullptr) ? 2 : 0;
                if (zloop == 2 && !p->pnext.p) // the pass writes the new direction beside the old one (halo rows of other tiles still read it)
                    FV_TRY(fv_vec_alloc(p, p->pnext, (size_t)n + (size_t)p->nhalo + FV_VEC_PAD, true));
            }
            if (zloop == 2) {
                const uint8_t *mvc = nullptr;
                StorageTable mvt{};
                FV_TRY(fv_minv_codes(p, &mvc, &mvt)); // (cached per Jacobi diagonal)
                if (wloop_first) { // the loop's first iteration: the classic product left q, this update moves x itself
                    if (iter == 0 && sys.x_next)
                        hipLaunchKernelGGL(pcg_update_z_kernel<true>, dim3(Gv), dim3(FV_BLOCK), 0, ctx->stream, n, iter, zr_is_z, (const double *)x, sys.x_next,
                                           p->r.p, p->pvec.p, p->q.p, p->minv.p, p->part_pq.p, npq, p->scal.p, p->part_rz.p, p->part_rr.p, mvc, mvt);
                    else
                        hipLaunchKernelGGL(pcg_update_z_kernel<false>, dim3(Gv), dim3(FV_BLOCK), 0, ctx->stream, n, iter, zr_is_z, (const double *)nullptr,
                                           sys.x_next ? sys.x_next : x, p->r.p, p->pvec.p, p->q.p, p->minv.p, p->part_pq.p, npq, p->scal.p,
                                           p->part_rz.p, p->part_rr.p, mvc, mvt);
                    wloop_first = false;
                } else { // the pass left w = -M^-1 q and owes x its update: z' = z + alpha w here, x += alpha p in the next pass
                    hipLaunchKernelGGL(pcg_update_w_kernel, dim3(Gv), dim3(FV_BLOCK), 0, ctx->stream, n, iter, p->r.p, (const double *)p->q.p,
                                       (const double *)p->pvec.p, (const double *)p->minv.p, (const double *)p->part_pq.p, npq, p->scal.p, p->part_rz.p,
                                       p->part_rr.p, mvc, mvt);
                    wloop_lag = true;
                }
                p->loop_minv_coded = mvc != nullptr;
                zr_is_z = 1;
                FV_PROF(3);
                FV_PROF(4);
                if (k == m - 1) // the chunk's last iteration: its verdict now, for the host's poll (the next pass would take it otherwise)
                    hipLaunchKernelGGL(pcg_verdict_kernel, dim3(1), dim3(FV_BLOCK), 0, ctx->stream, iter, (const double *)p->part_rz.p,
                                       (const double *)p->part_rr.p, Gv, p->scal.p, p->hist.p, p->hist_cap);
                FV_PROF(5);
                continue;
            }
            const bool spec = iter == 0 && speculate;
            if (spec) {
                hipLaunchKernelGGL(k2s_kernel(zf), dim3(Gv + Gx), dim3(FV_BLOCK), 0, ctx->stream, n, (const double *)x, sys.x_next, p->r.p,
                                   (const double *)p->pvec.p, (const double *)p->q.p, (const double *)p->minv.p,
                                   sarg,
                                   bsupport >= 0 ? (const double *)nullptr : sys.rhs, sys.dt, (const double *)p->part_pq.p, npq, p->scal.p,
                                   p->part_rz.p, p->part_rr.p, p->pnext.p, p->part_rz.p + FV_VEC_PARTIALS, p->part_rr.p + FV_VEC_PARTIALS,
                                   p->part_bb.p + FV_VEC_PARTIALS, sbarg, chained ? sys.chain_index : -1);
                p->spec_extra_bb = Gs;
                p->k2s_bytes = (zf ? 56 : 64) - Dsaved + (bsupport >= 0 || !sys.rhs ? 0 : 8);
            }
            else if (iter == 0 && sys.x_next)
                hipLaunchKernelGGL(pcg_update_kernel<true>, dim3(Gv), dim3(FV_BLOCK), 0, ctx->stream, n, iter, (const double *)x, sys.x_next,
                                   p->r.p, p->pvec.p, p->q.p, p->minv.p, p->part_pq.p, npq, p->scal.p, p->part_rz.p, p->part_rr.p);
            else
                hipLaunchKernelGGL(pcg_update_kernel<false>, dim3(Gv), dim3(FV_BLOCK), 0, ctx->stream, n, iter, (const double *)nullptr,
                                   sys.x_next ? sys.x_next : x, p->r.p, p->pvec.p, p->q.p, p->minv.p, p->part_pq.p, npq, p->scal.p,
                                   p->part_rz.p, p->part_rr.p);
            FV_PROF(3);
            FV_PROF(4);
            if (spec && chained && sys.chain_more && g_defer_reduce) {
                // this step's verdict and, if it converged, the next chained step's scalars from the set-up K2S left
                const BoundarySums bs{p->part_rz.p, p->part_rr.p, p->part_rz.p + FV_VEC_PARTIALS, p->part_rr.p + FV_VEC_PARTIALS,
                                      p->part_bb.p + FV_VEC_PARTIALS, Gv, Gv, Gv + Gs};
                // (every block takes the verdict from the same sums; only a chain that breaks — rare — has vector work here, so the
                // grid is small: 2048 blocks re-reducing five arrays cost 14 us per step on the 5M-cell mesh, 128 cost 4)
                hipLaunchKernelGGL(pcg_chain_boundary_kernel, dim3(Gv < 128 ? Gv : 128), dim3(FV_BLOCK), 0, ctx->stream, n, p->r.p, (const double *)p->minv.p,
                                   p->pvec.p, bs, rtol, p->scal.p, (const double *)x, (const double *)sys.x_next, (const double *)p->D.p, sys.dt,
                                   sys.chain_index, (sys.chain_index == g_chain_test_break) ? 1 : 0, zf ? (const double *)p->pnext.p : nullptr);
            } else if (spec)
                hipLaunchKernelGGL(pcg_pupdate_kernel<true>, dim3(Gv), dim3(FV_BLOCK), 0, ctx->stream, n, iter, p->r.p, (const double *)p->minv.p,
                                   p->pvec.p, (const double *)p->part_rz.p, (const double *)p->part_rr.p, Gv, p->scal.p, p->hist.p, p->hist_cap,
                                   (const double *)x, (const double *)sys.x_next, (const double *)p->D.p, sys.dt, sys.chain_index,
                                   (chained && sys.chain_index == g_chain_test_break) ? 1 : 0, zf ? (const double *)p->pnext.p : nullptr);
            else
                hipLaunchKernelGGL(pcg_pupdate_kernel<false>, dim3(Gv), dim3(FV_BLOCK), 0, ctx->stream, n, iter, p->r.p, (const double *)p->minv.p,
                                   p->pvec.p, (const double *)p->part_rz.p, (const double *)p->part_rr.p, Gv, p->scal.p, p->hist.p, p->hist_cap,
                                   (const double *)nullptr, (const double *)nullptr, (const double *)nullptr, 0.0);
            FV_PROF(5);
        }
        FV_LAUNCH_CHECK(ctx);
        it += m;
        if (chained) { // the caller polls once per burst (fv_pcg_chain_poll)
            p->bytes_total += acct + k1_form_bytes(p) + (int64_t)p->k2s_bytes * n;
            p->last_iters = 1;
            p->spec_valid = true;
            p->z_where = zf ? 2 : 0; // unless the chain stops on the device (the poll then says so)
            return FV_OK;
        }
        FV_HIP(ctx, hipMemcpyAsync(hs, p->scal.p, sizeof(PcgScalars), hipMemcpyDeviceToHost, ctx->stream));
        FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        polled = true;
        if (p->profile) { // only launches that did real work (not the post-convergence no-ops)
            const int64_t live = (int64_t)hs->iters - iters_before;
            for (int64_t k = 0; k < m && k < live; k++)
                for (int c = 0; c < (p->profile_level == 1 ? 3 : 1); c++) {
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
    if (wloop_lag) { // the last x-update of the loop through the fused kernel, wherever the loop stopped
        hipLaunchKernelGGL(pcg_xflush_kernel, dim3(Gv), dim3(FV_BLOCK), 0, ctx->stream, n, sys.x_next ? sys.x_next : x, (const PcgScalars *)p->scal.p);
        hipLaunchKernelGGL(pcg_xflush_clear_kernel, dim3(1), dim3(1), 0, ctx->stream, p->scal.p);
        FV_LAUNCH_CHECK(ctx);
    }
    if (time_it)
        FV_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    if (!polled) {
        FV_HIP(ctx, hipMemcpyAsync(hs, p->scal.p, sizeof(PcgScalars), hipMemcpyDeviceToHost, ctx->stream));
        FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    p->last_iters = hs->iters;
    if (!ploop && fv_step_precond_of(p, sys) != FV_PRECOND_AMG) { // (the V-cycle's launches are not modelled: fv_step_form's total leaves AMG solves out)
        const int64_t its = hs->iters - (resume ? sys.resume_it : 0), k1b = k1_form_bytes(p);
        if (its > 0) {
            if (zloop == 2) {
                const int64_t lb = p->loop_bytes - (p->loop_minv_coded ? 7 : 0);
                acct += k1b + (p->loop_minv_coded ? 49 : 56) * n + (its - 1) * lb * n + (wloop_lag ? 24 * n : 0);
            } else
                acct += its * (k1b + 56 * n) + (its - 1) * 32 * n + ((speculate && !resume) ? ((int64_t)p->k2s_bytes - 56) * n : 0);
        }
    }
    p->bytes_total += acct;
    if (zloop != 2 && !ploop)
        p->loop_bytes = 0; // (else what fv_fused_iteration reported: 113, or 91 with the matrix as codes)
    if ((zloop == 2 || ploop) && hs->iters >= 1) // (a solve that was converged at its set-up has launched no-ops only: r is still r)
        p->z_where = 4; // the array r holds M^-1 r (residual_to_r / the next step's carried set-up take it from there)
    p->spec_valid = speculate && hs->done == 1 && hs->iters == 1; // the K2S ran and the step converged in it
    if (speculate && zf && hs->iters >= 1)
        p->z_where = p->spec_valid ? 2 : 0; // a z-form K2S ran: its p' stands for the residual, unless the K3 behind it had to go on (it wrote r)
    if (use_spec && hs->done == 1 && hs->iters == 0) {
        // converged at its set-up: nothing ran, so the prepared set-up (direction, sums) is still exactly the next step's — put the
        // direction vectors back and keep it, as a burst does, instead of deriving a new one from the rounded residual
        p->pvec.swap(p->pnext);
        if (p->z_where == 1 || p->z_where == 2)
            p->z_where = 3 - p->z_where;
        p->spec_valid = true;
        p->last_iters = 1; // still the one-iteration regime, as after a chained step (the next step speculates in either mode)
    }
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

// After a burst of `nsteps` chained steps: one poll.  *completed = steps that converged in their one iteration; when it is
// < nsteps, step *completed is interrupted after its first iteration (done flag cleared here so that it can be resumed).
int fv_pcg_chain_poll(fv_problem *p, int nsteps, int *completed, fv_solve_info *info, uint32_t *zero_mask)
{
    fv_ctx *ctx = p->ctx;
    PcgScalars *hs = reinterpret_cast<PcgScalars *>(ctx->pinned);
    FV_HIP(ctx, hipMemcpyAsync(hs, p->scal.p, sizeof(PcgScalars), hipMemcpyDeviceToHost, ctx->stream));
    FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const int ndone = hs->done == 3 ? hs->chain_step : nsteps;
    if (p->profile && !p->prof_ev.empty())
        for (int k = 0; k < nsteps && k <= ndone && k < 32; k++)
            for (int c = 0; c < (p->profile_level != 1 ? 1 : (p->dist ? 2 : 3)); c++) { // the row-block driver times the SpMV and K2 / K2S only
                if (c == 2 && k == ndone)
                    continue; // the p-update of an interrupted step did real work; it is counted when the step resumes
                float ms = 0.f;
                FV_HIP(ctx, hipEventElapsedTime(&ms, p->prof_ev[(size_t)(6 * k + 2 * c)], p->prof_ev[(size_t)(6 * k + 2 * c + 1)]));
                p->prof_ms[c] += ms;
                p->prof_launches[c]++;
            }
    *completed = ndone;
    if (zero_mask)
        *zero_mask = hs->zero_mask;
    if (hs->done == 3) {
        p->spec_valid = false;
        p->vready = false;
        p->z_where = 0; // the launch that stopped the chain wrote r
        p->last_iters = 2; // at least
        // (a device-side memset: an asynchronous copy out of a variable on this stack frame may read it after the frame is gone — the
        // flag then stayed set now and then, the resumed step's launches were no-ops and it ran into maxiter; found by the fuzz of the
        // row-block driver, whose three host threads shift the timing)
        FV_HIP(ctx, hipMemsetAsync(&p->scal.p->done, 0, sizeof(int32_t), ctx->stream));
    } else if (info) {
        info->converged = hs->done == 1;
        info->iters = hs->iters;
        info->bnorm = sqrt(hs->bnorm2);
        info->relres = hs->bnorm2 > 0 ? sqrt(hs->rr / hs->bnorm2) : sqrt(hs->rr);
        info->solve_ms = 0.0;
        info->resnorm_len = 0;
    }
    if (hs->done == 2)
        fv_set_error(ctx, "PCG breakdown: p.Ap = %g is not positive (operator not SPD?)", hs->pq);
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
    hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(FV_BLOCK), 0, ctx->stream, p->part_bb.p, G, p->part_bb.p + (FV_VEC_PARTIALS - 1));
    FV_LAUNCH_CHECK(ctx);
    double *h = reinterpret_cast<double *>(ctx->pinned);
    FV_HIP(ctx, hipMemcpyAsync(h, p->part_bb.p + (FV_VEC_PARTIALS - 1), sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *out_host = *h;
    return FV_OK;
}

int fv_dot_device(fv_problem *p, const double *a, const double *b, double *out_host)
{
    FV_TRY(fv_pcg_prepare(p));
    int G = vec_grid(p->n);
    if (G > FV_VEC_PARTIALS - 1)
        G = FV_VEC_PARTIALS - 1;
    hipLaunchKernelGGL(dot_kernel, dim3(G), dim3(FV_BLOCK), 0, p->ctx->stream, p->n, a, b, p->part_bb.p);
    FV_LAUNCH_CHECK(p->ctx);
    return reduce_to_host(p, G, out_host);
}

int fv_norm2_diff_device(fv_problem *p, const double *a, const double *b, double *out_host)
{
    FV_TRY(fv_pcg_prepare(p));
    int G = vec_grid(p->n);
    if (G > FV_VEC_PARTIALS - 1)
        G = FV_VEC_PARTIALS - 1;
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

// packed: the send buffer has been filled by the caller (the fused step forms the values it ships, fv_fused_pack)
static int dist_exchange_begin(fv_problem *p, double *xext, bool packed = false)
{
    fv_ctx *ctx = p->ctx;
    fv_dist *d = p->dist;
    if (d->nranks <= 1)
        return FV_OK;
    if (d->nsend > 0 && !packed) {
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
    FV_TRY(fv_diag_mark(p->ctx, 2, p->ctx->stream)); // (diagnosis: how long the compute stream stalls here)
    FV_HIP(p->ctx, hipStreamWaitEvent(p->ctx->stream, p->ctx->ev_halo, 0));
    FV_TRY(fv_diag_mark(p->ctx, 2, p->ctx->stream));
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
        FV_TRY(fv_build_dia(p));
    FV_TRY(split_list(p, d->groups_int.p, d->n_int, d->int_dia, &d->n_int_dia, d->int_csr, &d->n_int_csr));
    FV_TRY(split_list(p, d->groups_bnd.p, d->n_bnd, d->bnd_dia, &d->n_bnd_dia, d->bnd_csr, &d->n_bnd_csr));
    // x-slab partitions of a structured grid: the interior groups are one contiguous run of slices
    d->int_lo = d->int_hi = 0;
    if (d->n_int > 0) {
        int32_t first = 0, last = 0;
        FV_HIP(p->ctx, fv_memcpy_sync(p->ctx, &first, d->groups_int.p, sizeof first, hipMemcpyDeviceToHost));
        FV_HIP(p->ctx, fv_memcpy_sync(p->ctx, &last, d->groups_int.p + (d->n_int - 1), sizeof last, hipMemcpyDeviceToHost));
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
// npq_out: leave the partial sums of x.y in part_pq (their count goes to *npq_out) instead of reducing them into red[0].
static int dist_spmv(fv_problem *p, double *xext, double *y, double sigma, const double *folded, bool want_dot, bool use_done,
                     bool skip_exchange = false, int *npq_out = nullptr)
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
    FV_TRY(fv_diag_mark(ctx, 3, ctx->stream));
    FV_TRY(spmv_apply(p, xext, y, sigma, folded, mode, want_dot ? p->part_pq.p : nullptr, nullptr, use_done, &na, &interior));
    FV_TRY(fv_diag_mark(ctx, 3, ctx->stream));
    if (!skip_exchange)
        FV_TRY(dist_exchange_wait(p));
    if (d->n_bnd > 0) {
        FV_TRY(fv_diag_mark(ctx, 4, ctx->stream));
        FV_TRY(spmv_apply(p, xext, y, sigma, folded, mode, want_dot ? p->part_pq.p + na : nullptr, nullptr, use_done, &nb, &boundary));
        FV_TRY(fv_diag_mark(ctx, 4, ctx->stream));
    }
    if (want_dot && npq_out)
        *npq_out = na + nb;
    else if (want_dot) {
        hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(FV_BLOCK), 0, ctx->stream, (const double *)p->part_pq.p, na + nb, d->red.p);
        FV_LAUNCH_CHECK(ctx);
    }
    return FV_OK;
}


// up to five partial-sum arrays reduced by one launch (block k sums array k into out[k])
struct SumSet {
    const double *a[6];
    int extra[6]; // partials beyond (or, negative, short of) the common count: the sparse-b pieces of rhs.rhs; p.q's own count
};
__global__ __launch_bounds__(FV_BLOCK) void final_sum_multi_kernel(SumSet set, int nparts, double *__restrict__ out)
{
    __shared__ double smem[4];
    const double t = reduce_partials(set.a[blockIdx.x], nparts + set.extra[blockIdx.x], smem);
    if (threadIdx.x == 0)
        out[blockIdx.x] = t;
}

// x_next / carry_prev / speculate / use_spec: the ping-pong state, the residual carry-over and the speculative set-up
// of PcgSystem, same meaning.  All-reduce buffer: red[0] p.q; red[1..2] r.M^-1 r, r.r; red[3] rhs.rhs of a regular
// set-up, or red[3..5] the next step's r.M^-1 r, r.r, rhs.rhs left by pcg_update_spec_kernel.
//
// Bursts of chained steps (chain_index >= 0) with g_defer_reduce: a step's five sums (its own r.M^-1 r, r.r and the next
// step's set-up scalars) are NOT all-reduced at its end but together with the next step's p.q — one 6-double collective
// per step instead of a 1- and a 5-double one.  The next step's K1 only needs p' (left by K2S), so it runs before the
// previous step's convergence is known; that verdict (pcg_pupdate_kernel<true> of the PREVIOUS step, with that step's
// vectors) and the new step's scalars follow the merged all-reduce.  A step that did not converge stops the chain there
// as before: K1 of the step after it has run for nothing, everything later is skipped by the done flag.  The last step
// of a burst reduces its sums itself, so the host poll sees a finished state.
int fv_dist_local_spmv(fv_problem *p, double *x, double *y, double sigma, bool fold, bool want_dot)
{
    const double *folded = nullptr;
    if (fold && sigma != 0.0)
        FV_TRY(ensure_folded(p, sigma, &folded));
    return dist_spmv(p, x, y, folded ? 0.0 : sigma, folded, want_dot, false, true); // interior + boundary passes, no exchange
}

// the product with the whole operator's rows of this block: halo exchange of x, interior + boundary passes (the gathered AMG's level 0)
int fv_dist_full_spmv(fv_problem *p, double *xext, double *y, double sigma, bool fold)
{
    const double *folded = nullptr;
    if (fold && sigma != 0.0)
        FV_TRY(ensure_folded(p, sigma, &folded));
    return dist_spmv(p, xext, y, folded ? 0.0 : sigma, folded, false, false);
}

// halo slots of xext (n + nhalo doubles) filled from their owners; returns when they are in place
int fv_dist_exchange(fv_problem *p, double *xext)
{
    FV_TRY(dist_exchange_begin(p, xext));
    FV_TRY(dist_exchange_wait(p));
    FV_HIP(p->ctx, hipStreamSynchronize(p->ctx->stream));
    return FV_OK;
}

// ------------------------------------------------------------------ PCG with the block-Jacobi AMG V-cycle on row blocks
// z = V_local(r) on every rank's diagonal block (fv_amg.hip; no communication), the PCG around it as on one GPU with its
// three sums all-reduced: p.q, r.r (the stopping test, before the next V-cycle is spent) and r.z.
__global__ __launch_bounds__(FV_BLOCK) void dist_amg_direction_kernel(int64_t n, int first, const double *__restrict__ z, double *__restrict__ pv,
                                                                       const double *__restrict__ rz_new, int it, PcgScalars *__restrict__ scal,
                                                                       const double *__restrict__ zq, const double *__restrict__ pq)
{
    if (scal->done)
        return;
    const double rzn = *rz_new;
    // zq / pq (the K-cycle is not a fixed linear operator: flexible PCG): beta = z.(r - r_old) / (r.z)_old = -z.q / p.q
    const double beta = first ? 0.0 : (zq ? -*zq / *pq : rzn / scal->rz[it & 1]);
    for (int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x; i < n; i += vec_stride())
        pv[i] = first ? z[i] : z[i] + beta * pv[i];
    if (blockIdx.x == 0 && threadIdx.x == 0)
        scal->rz[first ? 0 : ((it + 1) & 1)] = rzn;
}

__global__ void dist_amg_check_kernel(const double *__restrict__ rr, int it, PcgScalars *__restrict__ scal)
{
    if (threadIdx.x != 0 || blockIdx.x != 0 || scal->done)
        return;
    scal->rr = *rr;
    scal->iters = it + 1;
    if (*rr <= scal->tol2)
        scal->done = 1;
}

static int dist_amg_loop(fv_problem *p, double *x, double sigma, double sig_mv, const double *folded, int64_t maxiter, PcgScalars *hs)
{
    fv_ctx *ctx = p->ctx;
    fv_dist *d = p->dist;
    const int64_t n = p->n;
    const int Gv = vec_grid(n);
    double *red = d->red.p;
    FV_TRY(fv_amg_prepare(p, sigma));
    if (!p->cg_u.p) { // z: n + halo + pad, halo slots zero for ever (the V-cycle's level-0 products are block-local)
        FV_TRY(p->cg_u.alloc(ctx, (size_t)n + (size_t)p->nhalo + FV_VEC_PAD));
        FV_TRY(p->cg_u.zero(ctx));
        FV_TRY(p->cg_scal.alloc(ctx, 4));
    }
    double *z = p->cg_u.p;
    FV_HIP(ctx, hipMemcpyAsync(hs, p->scal.p, sizeof(PcgScalars), hipMemcpyDeviceToHost, ctx->stream));
    FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (hs->done || maxiter <= 0)
        return FV_OK;
    // the rank's cycle as a K-cycle (fv_amg.hip; its inner products are the rank's own: block-Jacobi stays communication-free):
    // the PCG around it is then the flexible variant, z.q travelling with r.z in the same collective
    const bool flexible = fv_amg_kcycle_available(p);
    auto precondition = [&](int first, int it) -> int {
        FV_TRY(fv_amg_apply_device(p, p->r.p, z, sigma, flexible));
        hipLaunchKernelGGL(dot_kernel, dim3(Gv), dim3(FV_BLOCK), 0, ctx->stream, n, (const double *)p->r.p, (const double *)z, p->part_rz.p);
        hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(FV_BLOCK), 0, ctx->stream, (const double *)p->part_rz.p, Gv, red + 4);
        if (flexible && !first) {
            hipLaunchKernelGGL(dot_kernel, dim3(Gv), dim3(FV_BLOCK), 0, ctx->stream, n, (const double *)p->q.p, (const double *)z, p->part_bb.p);
            hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(FV_BLOCK), 0, ctx->stream, (const double *)p->part_bb.p, Gv, red + 5);
        }
        FV_LAUNCH_CHECK(ctx);
        FV_TRY(fv_comm_allreduce_sum(ctx, d, red + 4, (flexible && !first) ? 2 : 1, ctx->stream));
        hipLaunchKernelGGL(dist_amg_direction_kernel, dim3(Gv), dim3(FV_BLOCK), 0, ctx->stream, n, first, (const double *)z, p->pvec.p,
                           (const double *)(red + 4), it, p->scal.p, (flexible && !first) ? (const double *)(red + 5) : (const double *)nullptr,
                           (const double *)red);
        FV_LAUNCH_CHECK(ctx);
        return FV_OK;
    };
    FV_TRY(precondition(1, 0));
    for (int64_t it = 0; it < maxiter; it++) {
        FV_TRY(dist_spmv(p, p->pvec.p, p->q.p, sig_mv, folded, true, true)); // red[0] = the local p.q
        FV_TRY(fv_comm_allreduce_sum(ctx, d, red, 1, ctx->stream));
        hipLaunchKernelGGL(pcg_update_kernel<false>, dim3(Gv), dim3(FV_BLOCK), 0, ctx->stream, n, (int)it, (const double *)nullptr, x, p->r.p, p->pvec.p,
                           p->q.p, p->minv.p, (const double *)red, 1, p->scal.p, p->part_rz.p, p->part_rr.p);
        hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(FV_BLOCK), 0, ctx->stream, (const double *)p->part_rr.p, Gv, red + 2);
        FV_LAUNCH_CHECK(ctx);
        FV_TRY(fv_comm_allreduce_sum(ctx, d, red + 2, 1, ctx->stream));
        hipLaunchKernelGGL(dist_amg_check_kernel, dim3(1), dim3(64), 0, ctx->stream, (const double *)(red + 2), (int)it, p->scal.p);
        FV_LAUNCH_CHECK(ctx);
        FV_HIP(ctx, hipMemcpyAsync(hs, p->scal.p, sizeof(PcgScalars), hipMemcpyDeviceToHost, ctx->stream));
        FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (hs->done)
            break;
        FV_TRY(precondition(0, (int)it));
    }
    return FV_OK;
}

// ------------------------------------------------------------------ one-reduction PCG (Chronopoulos & Gear) on row blocks
// The classic loop needs two global sums per iteration one after the other (p.q before alpha, r.M^-1 r before beta): two
// all-reduce latencies on the wire.  With u = M^-1 r, w = A u and the recurrences p = u + beta p, s = w + beta s (= A p)
// both sums of an iteration — gamma = r.u and delta = w.u, plus r.r for the stopping test — are known after ONE SpMV and
// travel in one 3-double all-reduce:  beta = gamma'/gamma,  alpha = gamma' / (delta - beta gamma'/alpha_prev).
// Per iteration: one fused vector pass (96 B/row: p, s, x, r, u, w, M^-1 in; p, s, x, r, u out), the block SpMV w = A u with
// its halo exchange, one reduction launch, one all-reduce, one scalar launch.  fv_tune key 34; the classic form stays the
// default (north_star names it; on one GPU it moves 8 B/row less).
int g_cg_one_reduction = 0;

struct CgcgScalars { // cg_scal: alpha, beta, gamma (= r.u of the iterate the current direction was built from)
    double alpha, beta, gamma;
};

__global__ __launch_bounds__(FV_BLOCK) void cgcg_vector_kernel(int64_t n, int first, double *__restrict__ x, double *__restrict__ r,
                                                                double *__restrict__ u, const double *__restrict__ w, double *__restrict__ pv,
                                                                double *__restrict__ s, const double *__restrict__ minv,
                                                                const CgcgScalars *__restrict__ cg, const PcgScalars *__restrict__ scal,
                                                                double *__restrict__ part_g, double *__restrict__ part_rr)
{
    __shared__ double smem[4];
    if (scal->done)
        return;
    const double alpha = cg->alpha, beta = first ? 0.0 : cg->beta;
    double ag = 0.0, arr = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x; i < n; i += vec_stride()) {
        const double pi = first ? u[i] : u[i] + beta * pv[i];
        const double si = first ? w[i] : w[i] + beta * s[i];
        pv[i] = pi;
        s[i] = si;
        x[i] += alpha * pi;
        const double ri = r[i] - alpha * si;
        r[i] = ri;
        const double ui = minv[i] * ri;
        u[i] = ui;
        ag += ri * ui;
        arr += ri * ri;
    }
    const double t0 = block_sum(ag, smem);
    const double t1 = block_sum(arr, smem);
    if (threadIdx.x == 0) {
        part_g[blockIdx.x] = t0;
        part_rr[blockIdx.x] = t1;
    }
}

// red[0] = delta = w.u, red[1] = gamma' = r.u, red[2] = r.r (all-reduced).  init: the sums before the first iteration
// (delta0 only; gamma0 = scal->rz[0] from the set-up): alpha0 = gamma0 / delta0.
__global__ void cgcg_scalar_kernel(const double *__restrict__ red, int init, int it, CgcgScalars *__restrict__ cg, PcgScalars *__restrict__ scal,
                                   double *__restrict__ hist, int64_t hist_cap)
{
    if (threadIdx.x != 0 || blockIdx.x != 0 || scal->done)
        return;
    const double delta = red[0];
    if (init) {
        if (!(delta > 0.0)) {
            scal->pq = delta;
            scal->done = 2;
            return;
        }
        cg->gamma = scal->rz[0];
        cg->alpha = cg->gamma / delta;
        cg->beta = 0.0;
        scal->pq = delta;
        return;
    }
    const double gn = red[1], rr = red[2];
    scal->rr = rr;
    scal->iters = it + 1;
    if (hist && it < hist_cap)
        hist[it] = sqrt(rr);
    if (rr <= scal->tol2) {
        scal->done = 1;
        return;
    }
    const double beta = gn / cg->gamma;
    const double denom = delta - beta * gn / cg->alpha;
    if (!(denom > 0.0)) { // breakdown: not positive definite, or the recurrences have drifted
        scal->pq = denom;
        scal->done = 2;
        return;
    }
    cg->beta = beta;
    cg->alpha = gn / denom;
    cg->gamma = gn;
    scal->pq = denom;
}

// What a row-block solve is asked for beyond the fixed-dt step with the assembled b: an explicit system
// (A + sigma D) x = rhs (the steady solve: sigma = 0, rhs = b), or an implicit step whose forcing is a caller's
// volume-scaled vector (b' = D * bhat, the getb(t) of transient.jl:165-174).
struct DistSystem {
    bool explicit_system = false;
    double sigma = 0.0;          // explicit_system only
    const double *rhs = nullptr; // explicit: the right-hand side; implicit: bhat (null = the assembled b)
};

static int dist_step(fv_problem *p, double *u, double dt, double rtol, int64_t maxiter, fv_solve_info *info, double *x_next = nullptr,
                     const double *carry_prev = nullptr, bool speculate_in = false, bool use_spec_in = false, int chain_index = -1,
                     int resume_it = 0, bool last_in_burst = true, const DistSystem *ds = nullptr)
{
    fv_ctx *ctx = p->ctx;
    fv_dist *d = p->dist;
    const int64_t n = p->n;
    const int Gv = vec_grid(n);
    const bool explicit_sys = ds && ds->explicit_system;
    const double sigma = explicit_sys ? ds->sigma : 1.0 / dt;
    const double *bprime = (ds && ds->rhs) ? ds->rhs : (const double *)p->b.p;
    const int b_times_D = (ds && !explicit_sys && ds->rhs) ? 1 : 0;
    const double *folded = nullptr;
    if (sigma != 0.0)
        FV_TRY(ensure_folded(p, sigma, &folded));
    const double sig_mv = folded ? 0.0 : sigma;
    const int compute_minv = !(p->minv_valid && p->minv_sigma == sigma && p->minv_epoch == p->assemble_epoch);
    p->minv_valid = true;
    p->minv_sigma = sigma;
    p->minv_epoch = p->assemble_epoch;
    double *red = d->red.p;
    const bool chained = chain_index >= 0, resume = resume_it > 0; // bursts of unpolled steps, as in fv_pcg_solve
    const bool use_spec = !resume && use_spec_in && p->spec_valid && !compute_minv;
    p->spec_valid = false;
    if (chained && !use_spec) {
        fv_set_error(ctx, "internal: chained step without a prepared set-up");
        return FV_ERR_STATE;
    }
    const bool speculate = !resume && speculate_in && x_next && !compute_minv && p->last_iters == 1 && maxiter > 0;
    if (speculate && !p->pnext.p)
        FV_TRY(fv_vec_alloc(p, p->pnext, (size_t)n + (size_t)p->nhalo + FV_VEC_PAD, true));
    int64_t bsupport = -1; // the block's b is as sparse as the global one: its share of rhs.rhs by the gather blocks of K2S
    if (speculate && g_sparse_b)
        FV_TRY(ensure_b_support(p, &bsupport));
    int Gx = 0, Gs = 0; // extra blocks of the K2S launch, extra rhs.rhs partials it leaves
    const SparseRhs sbarg = sparse_b_arg(p, bsupport, Gv, &Gx, &Gs);
    StorageArg sarg{};
    int Dsaved = 0;
    if (speculate)
        FV_TRY(fv_storage_form(p, &sarg, &Dsaved, false));
    bool zf = false; // this step's K2S in the z-form (each rank decides for its own rows: the arithmetic is per row)
    if (speculate && g_zform)
        FV_TRY(minv_positive(p, &zf));
    // The fused step (fv_fused.hip) on the row block, decided at a burst's first step for the whole burst; was_vready: the
    // previous step was such a launch (its v, its six local sums are in the fused step's own arrays)
    const bool was_vready = p->vready && use_spec;
    p->vready = false;
    bool fused = false;
    if (chained && speculate && x_next && carry_prev) { // (these four are the same on every rank)
        if (chain_index == 0) {
            if (!d->split_built)
                FV_TRY(dist_build_split(p));
            // what this rank could do — its rows' M^-1, its storage codes, its block's shape — and what ALL ranks can: the fused
            // step and the K1 + K2S pair issue their collectives and halo exchanges in different orders, so the ranks must not
            // mix them.  Agreed once per fv_dist_run_fixed call (one 1-double all-reduce and a host read-back).
            const bool local = zf && folded && sarg.D == nullptr && bsupport >= 0 && fv_fused_applicable(p, sigma);
            // Agreed at the first burst of a call, again at every burst while the answer is "no" (a rank whose storage form was not
            // established yet at the first burst no longer pins the whole run to the unfused pair), and every 16th burst while it is
            // "yes" (all ranks then fall back together should one of them lose the form).  One 1-double all-reduce and
            // a host read-back each time — every rank reaches this point at the same bursts, so the collective always matches.
            const bool ask = d->fused_agreed < 0 || d->fused_agreed == 0 || (d->fused_bursts % 16) == 0;
            d->fused_bursts++;
            if (ask) {
                bool all = local;
                if (d->nranks > 1) {
                    double *h = reinterpret_cast<double *>(static_cast<char *>(ctx->pinned) + 2048);
                    h[0] = local ? 0.0 : 1.0;
                    FV_HIP(ctx, hipMemcpyAsync(red + 7, h, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
                    FV_TRY(fv_comm_allreduce_sum(ctx, d, red + 7, 1, ctx->stream));
                    FV_HIP(ctx, hipMemcpyAsync(h, red + 7, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
                    FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
                    all = h[0] == 0.0;
                }
                d->fused_agreed = all ? 1 : 0;
            }
            if (d->fused_agreed == 1 && !local) {
                fv_set_error(ctx, "internal: the ranks agreed on the fused step at the start of this run and this rank can no longer run it");
                return FV_ERR_STATE;
            }
            p->burst_fused = d->fused_agreed == 1;
        }
        fused = p->burst_fused;
    }
    if (chained && chain_index > 0 && was_vready != fused) {
        fv_set_error(ctx, "internal: a burst changed between the fused step and the K1 + K2S pair in its middle");
        return FV_ERR_STATE;
    }
    const double *prev_z = nullptr; // use_spec: where the previous step's z-form K2S left its p' (nullptr: it wrote r)
    const bool defer_in = chained && chain_index > 0 && g_defer_reduce && speculate && carry_prev;   // red[1..5]: the previous step's local sums
    const bool defer_out = chained && !last_in_burst && g_defer_reduce && speculate; // leave this step's sums to the next one
    if (resume) {
        // r, p and the scalars are those of the interrupted solve
    } else if (use_spec) {
        // r, p' and the all-reduced set-up scalars (red[3..5]) were left by the previous step's K2S
        p->pvec.swap(p->pnext);
        if (p->z_where)
            p->z_where = 3 - p->z_where;
        if (!zf)
            FV_TRY(residual_to_r(p)); // the first K2 of this step reads r
        prev_z = p->z_where == 1 ? p->pvec.p : nullptr;
        if (!defer_in && !(fused && chain_index > 0)) { // (a fused launch past a burst's first takes its scalars from the merged collective itself)
            hipLaunchKernelGGL(pcg_init_finalize_kernel, dim3(1), dim3(FV_BLOCK), 0, ctx->stream, (const double *)(red + 3),
                               (const double *)(red + 4), (const double *)(red + 5), 1, rtol, p->scal.p, -1, chained && chain_index > 0 ? 1 : 0);
            FV_LAUNCH_CHECK(ctx);
        }
    } else {
        if (carry_prev && !compute_minv && !ds) {
            // r0 = r_final + sigma D (u - u_prev): purely local, no halo of u needed
            const double *zsrc = p->z_where == 1 ? p->pvec.p : p->z_where == 2 ? p->pnext.p : nullptr;
            hipLaunchKernelGGL(pcg_carry_init_kernel, dim3(Gv), dim3(FV_BLOCK), 0, ctx->stream, n, (const double *)p->b.p, (const double *)p->D.p,
                               dt, (const double *)u, carry_prev, (const double *)p->minv.p, p->r.p, p->pvec.p, p->part_rz.p, p->part_rr.p,
                               p->part_bb.p, zsrc);
        } else if (explicit_sys) {
            // q = (A + sigma D) x, r0 = rhs - q
            FV_TRY(dist_spmv(p, u, p->q.p, sig_mv, folded, false, false));
            hipLaunchKernelGGL(pcg_init_kernel<false>, dim3(Gv), dim3(FV_BLOCK), 0, ctx->stream, n, bprime, (const double *)p->q.p, p->diagA.p,
                               sigma != 0.0 ? (const double *)p->D.p : (const double *)nullptr, sigma, 0.0, 0, (const double *)nullptr, compute_minv, 0,
                               p->r.p, p->pvec.p, p->minv.p, p->part_rz.p, p->part_rr.p, p->part_bb.p);
        } else {
            // q = (A + sigma D) u with the matrix the iterations use (folded when available), r0 = rhs - q
            FV_TRY(dist_spmv(p, u, p->q.p, sig_mv, folded, false, false));
            hipLaunchKernelGGL(pcg_init_kernel<true>, dim3(Gv), dim3(FV_BLOCK), 0, ctx->stream, n, bprime, (const double *)p->q.p,
                               p->diagA.p, (const double *)p->D.p, sigma, dt, b_times_D, (const double *)u, compute_minv, 1, p->r.p, p->pvec.p,
                               p->minv.p, p->part_rz.p, p->part_rr.p, p->part_bb.p);
        }
        p->z_where = 0; // each of these set-ups has written r
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
    int64_t it = resume ? resume_it : 0;
    int64_t chunk = p->last_iters > 0 ? p->last_iters : 1;
    if (chunk > 32)
        chunk = 32;
    if (chained) {
        chunk = 1;
        maxiter = 1;
    }
    if (fv_step_precond(p) == FV_PRECOND_AMG && !resume) { // the block-Jacobi V-cycle: plain in-place solves (no ping-pong, no carried residual)
        if (x_next || chained || speculate) {
            fv_set_error(ctx, "internal: the AMG path of the row-block driver steps in place");
            return FV_ERR_STATE;
        }
        FV_TRY(dist_amg_loop(p, u, sigma, sig_mv, folded, maxiter, hs));
        FV_HIP(ctx, hipMemcpyAsync(hs, p->scal.p, sizeof(PcgScalars), hipMemcpyDeviceToHost, ctx->stream));
        FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        p->last_iters = hs->iters;
        p->spec_valid = false;
        if (info) {
            info->converged = hs->done == 1;
            info->iters = hs->iters;
            info->bnorm = sqrt(hs->bnorm2);
            info->relres = hs->bnorm2 > 0 ? sqrt(hs->rr / hs->bnorm2) : sqrt(hs->rr);
            info->solve_ms = 0.0;
            info->resnorm_len = 0;
        }
        if (hs->done == 2)
            fv_set_error(ctx, "PCG breakdown: p.Ap = %g is not positive (operator not SPD?)", hs->pq);
        return FV_OK;
    }
    if (g_cg_one_reduction && !speculate && !chained && !resume && maxiter > 0) {
        // ---- one-reduction form: see cgcg_vector_kernel.  The iterate lives in x_next when the caller ping-pongs.
        double *xx = u;
        if (x_next) {
            FV_HIP(ctx, hipMemcpyAsync(x_next, u, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
            xx = x_next;
        }
        if (!p->cg_u.p) {
            FV_TRY(p->cg_u.alloc(ctx, (size_t)n + (size_t)p->nhalo + FV_VEC_PAD));
            FV_TRY(p->cg_u.zero(ctx));
            FV_TRY(p->cg_scal.alloc(ctx, 4));
        }
        if (!p->pnext.p)
            FV_TRY(fv_vec_alloc(p, p->pnext, (size_t)n + (size_t)p->nhalo + FV_VEC_PAD, true));
        CgcgScalars *cg = reinterpret_cast<CgcgScalars *>(p->cg_scal.p);
        double *uu = p->cg_u.p, *ss = p->pnext.p, *ww = p->q.p;
        // u0 = M^-1 r0 is what the set-up left in pvec; w0 = A u0, delta0 = w0.u0
        FV_HIP(ctx, hipMemcpyAsync(uu, p->pvec.p, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
        FV_TRY(dist_spmv(p, uu, ww, sig_mv, folded, true, true));
        FV_TRY(fv_comm_allreduce_sum(ctx, d, red, 1, ctx->stream));
        hipLaunchKernelGGL(cgcg_scalar_kernel, dim3(1), dim3(64), 0, ctx->stream, (const double *)red, 1, 0, cg, p->scal.p, (double *)nullptr, (int64_t)0);
        FV_LAUNCH_CHECK(ctx);
        int64_t it1 = 0, chunk1 = p->last_iters > 0 ? p->last_iters : 1;
        if (chunk1 > 32)
            chunk1 = 32;
        while (it1 < maxiter) {
            const int64_t m = (maxiter - it1 < chunk1) ? (maxiter - it1) : chunk1;
            for (int64_t k = 0; k < m; k++) {
                const int iter = (int)(it1 + k);
                hipLaunchKernelGGL(cgcg_vector_kernel, dim3(Gv), dim3(FV_BLOCK), 0, ctx->stream, n, iter == 0 ? 1 : 0, xx, p->r.p, uu, (const double *)ww,
                                   p->pvec.p, ss, (const double *)p->minv.p, (const CgcgScalars *)cg, (const PcgScalars *)p->scal.p, p->part_rz.p,
                                   p->part_rr.p);
                FV_LAUNCH_CHECK(ctx);
                FV_TRY(dist_spmv(p, uu, ww, sig_mv, folded, true, true)); // red[0] = the local w.u
                SumSet two{};
                two.a[0] = p->part_rz.p;
                two.a[1] = p->part_rr.p;
                hipLaunchKernelGGL(final_sum_multi_kernel, dim3(2), dim3(FV_BLOCK), 0, ctx->stream, two, Gv, red + 1);
                FV_LAUNCH_CHECK(ctx);
                FV_TRY(fv_comm_allreduce_sum(ctx, d, red, 3, ctx->stream)); // THE collective of the iteration: delta, gamma', r.r
                hipLaunchKernelGGL(cgcg_scalar_kernel, dim3(1), dim3(64), 0, ctx->stream, (const double *)red, 0, iter, cg, p->scal.p, (double *)nullptr,
                                   (int64_t)0);
                FV_LAUNCH_CHECK(ctx);
            }
            it1 += m;
            FV_HIP(ctx, hipMemcpyAsync(hs, p->scal.p, sizeof(PcgScalars), hipMemcpyDeviceToHost, ctx->stream));
            FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (hs->done)
                break;
            if (chunk1 < 32)
                chunk1 *= 2;
        }
        p->last_iters = hs->iters;
        p->spec_valid = false;
        if (info) {
            info->converged = hs->done == 1;
            info->iters = hs->iters;
            info->bnorm = sqrt(hs->bnorm2);
            info->relres = hs->bnorm2 > 0 ? sqrt(hs->rr / hs->bnorm2) : sqrt(hs->rr);
            info->solve_ms = 0.0;
            info->resnorm_len = 0;
        }
        if (hs->done == 2)
            fv_set_error(ctx, "PCG breakdown in the one-reduction form: %g is not positive (operator not SPD, or drift of the recurrences)", hs->pq);
        return FV_OK;
    }
    // per-kernel timing (fv_profile_enable), as in fv_pcg_solve: event pairs around the block SpMV (pack, interior pass,
    // the wait for the halo, boundary pass) and around K2 / K2S of every iteration; the reductions, collectives and K3 /
    // boundary launches in between are not timed
    const int64_t kprof = chained ? chain_index : 0;
    if (p->profile && p->prof_ev.empty()) {
        p->prof_ev.resize((size_t)(6 * 32));
        for (hipEvent_t &e : p->prof_ev)
            FV_HIP(ctx, hipEventCreate(&e));
    }
#define FV_PROF(idx)                                                                                            \
    if (p->profile && ((idx) < 2 || p->profile_level == 1))                                                     \
    FV_HIP(ctx, hipEventRecord(p->prof_ev[(size_t)(6 * (k + kprof) + (idx))], ctx->stream))
    if (fused && !resume && maxiter > 0) {
        // ---- the fused step on a row block.  Per step ONE 6-double collective (the previous launch's local sums: z.q, that
        // step's r.M^-1 r and r.r, this step's set-up scalars), then: z' of the rows the neighbours need straight into the send
        // buffer (two loads and an FMA per row: the v-form), the halo exchange of z' on its way while the fused launch runs
        // over the block (vector part of all rows, products of the interior window), then the boundary groups' products from
        // the halo and their conversion to the v-form.  The launch's prologue takes verdict, alpha and the fall-back from the
        // all-reduced sums, so every rank takes the same decisions.
        const int64_t k = 0;
        FV_TRY(fv_fused_prepare(p));
        FusedSums fin{};
        int fmode = 0;
        FV_PROF(0);
        if (!was_vready) { // entry: the product the classic way, v from it
            FV_TRY(dist_spmv(p, p->pvec.p, p->q.p, sig_mv, folded, true, true));
            FV_TRY(fv_comm_allreduce_sum(ctx, d, red, 1, ctx->stream));
            FV_TRY(fv_fused_enter(p, sigma));
        } else {
            fin = fv_fused_sums(p, p->vready_parity);
            fin.nvec = p->vready_counts[0];
            fin.nbb = p->vready_counts[1];
            fin.npq = p->vready_counts[2];
            SumSet six{};
            six.a[0] = fin.pq;
            six.extra[0] = fin.npq - fin.nvec;
            six.a[1] = fin.arz;
            six.a[2] = fin.arr;
            six.a[3] = fin.srz;
            six.a[4] = fin.srr;
            six.a[5] = fin.sbb;
            six.extra[5] = fin.nbb - fin.nvec;
            // (a burst's first step: the five vector sums were all-reduced and judged at the end of the previous burst)
            const int nsum = chain_index > 0 ? 6 : 1;
            hipLaunchKernelGGL(final_sum_multi_kernel, dim3(nsum), dim3(FV_BLOCK), 0, ctx->stream, six, fin.nvec, red);
            FV_LAUNCH_CHECK(ctx);
            FV_TRY(fv_comm_allreduce_sum(ctx, d, red, nsum, ctx->stream));
            fmode = chain_index > 0 ? 1 : 0;
        }
        const bool force_prev = fmode == 1 && chain_index - 1 == g_chain_test_break;
        FV_TRY(fv_fused_pack(p, red, fmode, chain_index, force_prev, rtol));
        FV_TRY(dist_exchange_begin(p, p->pnext.p, true));
        FusedSums fout{};
        FV_TRY(fv_diag_mark(ctx, 3, ctx->stream));
        FV_TRY(fv_fused_step(p, u, x_next, sigma, dt, rtol, chain_index, fmode, fin, force_prev, folded, bsupport, &fout, red));
        FV_TRY(fv_diag_mark(ctx, 3, ctx->stream));
        FV_TRY(dist_exchange_wait(p));
        if (d->n_bnd > 0) {
            GroupSubset boundary;
            boundary.dia = d->bnd_dia.p;
            boundary.ndia = d->n_bnd_dia;
            boundary.csr = d->bnd_csr.p;
            boundary.ncsr = d->n_bnd_csr;
            int nb = 0;
            FV_TRY(fv_diag_mark(ctx, 4, ctx->stream));
            if (d->n_bnd_csr == 0) { // slabs of a structured grid: sliced-DIA boundary slices, stored in the v-form by the launch itself
                StepInitEpilogue vform{};
                vform.q_shifted = 2;
                vform.minv = p->minv.p;
                vform.D = p->D.p;
                vform.sigma = sigma;
                FV_TRY(spmv_apply(p, p->pnext.p, p->qv2.p, sig_mv, folded, SPMV_DOT, fout.pq + fout.npq, &vform, false, &nb, &boundary));
            } else {
                FV_TRY(spmv_apply(p, p->pnext.p, p->qv2.p, sig_mv, folded, SPMV_DOT, fout.pq + fout.npq, nullptr, false, &nb, &boundary));
                FV_TRY(fv_fused_convert_groups(p, d->bnd_dia.p, d->n_bnd_dia, sigma));
                FV_TRY(fv_fused_convert_groups(p, d->bnd_csr.p, d->n_bnd_csr, sigma));
            }
            FV_TRY(fv_diag_mark(ctx, 4, ctx->stream));
            fout.npq += nb;
            if (fout.npq > FV_FUSED_PARTS) {
                fv_set_error(ctx, "internal: %d partial sums of the fused step's product on a row block", fout.npq);
                return FV_ERR_STATE;
            }
        }
        FV_PROF(1);
        FV_PROF(2);
        FV_PROF(3);
        if (last_in_burst) { // its five vector sums and its verdict now, so that the host's poll sees a finished state
            SumSet five{};
            five.a[0] = fout.arz;
            five.a[1] = fout.arr;
            five.a[2] = fout.srz;
            five.a[3] = fout.srr;
            five.a[4] = fout.sbb;
            five.extra[4] = fout.nbb - fout.nvec;
            hipLaunchKernelGGL(final_sum_multi_kernel, dim3(5), dim3(FV_BLOCK), 0, ctx->stream, five, fout.nvec, red + 1);
            FV_LAUNCH_CHECK(ctx);
            FV_TRY(fv_comm_allreduce_sum(ctx, d, red + 1, 5, ctx->stream));
            hipLaunchKernelGGL(pcg_pupdate_kernel<true>, dim3(Gv), dim3(FV_BLOCK), 0, ctx->stream, n, 0, p->r.p, (const double *)p->minv.p, p->pvec.p,
                               (const double *)(red + 1), (const double *)(red + 2), 1, p->scal.p, (double *)nullptr, (int64_t)0, (const double *)u,
                               (const double *)x_next, (const double *)p->D.p, dt, chain_index, (chain_index == g_chain_test_break) ? 1 : 0,
                               (const double *)p->pnext.p);
            FV_LAUNCH_CHECK(ctx);
        }
        p->qv.swap(p->qv2);
        p->vready = true;
        p->vready_parity = chain_index & 1;
        p->vready_counts[0] = fout.nvec;
        p->vready_counts[1] = fout.nbb;
        p->vready_counts[2] = fout.npq;
        p->fused_launches++;
        p->last_iters = 1;
        p->spec_valid = true;
        p->z_where = 2;
        return FV_OK;
    }
    while (it < maxiter) {
        const int64_t m = (maxiter - it < chunk) ? (maxiter - it) : chunk;
        for (int64_t k = 0; k < m; k++) {
            const int iter = (int)(it + k);
            const bool spec = iter == 0 && speculate;
            int npq = 0;
            FV_PROF(0);
            FV_TRY(dist_spmv(p, p->pvec.p, p->q.p, sig_mv, folded, true, true, false, (defer_in && iter == 0) ? &npq : nullptr));
            FV_PROF(1);
            if (defer_in && iter == 0) {
                // one reduction launch for the six sums of the merged collective: this step's p.q and the five the previous
                // step's K2S left in its partial arrays (nothing has written them since)
                SumSet six{};
                six.a[0] = p->part_pq.p;
                six.extra[0] = npq - Gv;
                six.a[1] = p->part_rz.p;
                six.a[2] = p->part_rr.p;
                six.a[3] = p->part_rz.p + FV_VEC_PARTIALS;
                six.a[4] = p->part_rr.p + FV_VEC_PARTIALS;
                six.a[5] = p->part_bb.p + FV_VEC_PARTIALS;
                six.extra[5] = Gs;
                hipLaunchKernelGGL(final_sum_multi_kernel, dim3(6), dim3(FV_BLOCK), 0, ctx->stream, six, Gv, red);
                FV_LAUNCH_CHECK(ctx);
                // p.q of this step with the five sums the previous step left un-reduced; then, in one launch, that step's
                // verdict (on its own vectors: its p is this step's pnext, its iterate went from carry_prev to u) and this
                // step's scalars
                FV_TRY(fv_comm_allreduce_sum(ctx, d, red, 6, ctx->stream));
                const BoundarySums bs{red + 1, red + 2, red + 3, red + 4, red + 5, 1, 1, 1};
                hipLaunchKernelGGL(pcg_chain_boundary_kernel, dim3(Gv < 128 ? Gv : 128), dim3(FV_BLOCK), 0, ctx->stream, n, p->r.p, (const double *)p->minv.p,
                                   p->pnext.p, bs, rtol, p->scal.p, carry_prev, (const double *)u, (const double *)p->D.p, dt,
                                   chain_index - 1, (chain_index - 1 == g_chain_test_break) ? 1 : 0, prev_z);
                FV_LAUNCH_CHECK(ctx);
            } else
                FV_TRY(fv_comm_allreduce_sum(ctx, d, red, 1, ctx->stream));
            SumSet sums{};
            sums.a[0] = p->part_rz.p;
            sums.a[1] = p->part_rr.p;
            FV_PROF(2);
            if (spec) {
                hipLaunchKernelGGL(k2s_kernel(zf), dim3(Gv + Gx), dim3(FV_BLOCK), 0, ctx->stream, n, (const double *)u, x_next, p->r.p,
                                   (const double *)p->pvec.p, (const double *)p->q.p, (const double *)p->minv.p,
                                   sarg,
                                   bsupport >= 0 ? (const double *)nullptr : (const double *)p->b.p, dt, (const double *)red, 1, p->scal.p,
                                   p->part_rz.p, p->part_rr.p, p->pnext.p, p->part_rz.p + FV_VEC_PARTIALS, p->part_rr.p + FV_VEC_PARTIALS,
                                   p->part_bb.p + FV_VEC_PARTIALS, sbarg, chained ? chain_index : -1);
                p->k2s_bytes = (zf ? 56 : 64) - Dsaved + (bsupport >= 0 ? 0 : 8);
                sums.a[2] = p->part_rz.p + FV_VEC_PARTIALS;
                sums.a[3] = p->part_rr.p + FV_VEC_PARTIALS;
                sums.a[4] = p->part_bb.p + FV_VEC_PARTIALS;
                sums.extra[4] = Gs;
            } else if (iter == 0 && x_next)
                hipLaunchKernelGGL(pcg_update_kernel<true>, dim3(Gv), dim3(FV_BLOCK), 0, ctx->stream, n, iter, (const double *)u, x_next, p->r.p,
                                   p->pvec.p, p->q.p, p->minv.p, (const double *)red, 1, p->scal.p, p->part_rz.p, p->part_rr.p);
            else
                hipLaunchKernelGGL(pcg_update_kernel<false>, dim3(Gv), dim3(FV_BLOCK), 0, ctx->stream, n, iter, (const double *)nullptr,
                                   x_next ? x_next : u, p->r.p, p->pvec.p, p->q.p, p->minv.p, (const double *)red, 1, p->scal.p,
                                   p->part_rz.p, p->part_rr.p);
            FV_PROF(3);
            if (defer_out && spec)
                continue; // summed and all-reduced with the next step's p.q, judged there
            const int nsums = spec ? 5 : 2;
            hipLaunchKernelGGL(final_sum_multi_kernel, dim3(nsums), dim3(FV_BLOCK), 0, ctx->stream, sums, Gv, red + 1);
            FV_LAUNCH_CHECK(ctx);
            FV_TRY(fv_comm_allreduce_sum(ctx, d, red + 1, nsums, ctx->stream));
            if (spec)
                hipLaunchKernelGGL(pcg_pupdate_kernel<true>, dim3(Gv), dim3(FV_BLOCK), 0, ctx->stream, n, iter, p->r.p, (const double *)p->minv.p,
                                   p->pvec.p, (const double *)(red + 1), (const double *)(red + 2), 1, p->scal.p, (double *)nullptr, (int64_t)0,
                                   (const double *)u, (const double *)x_next, (const double *)p->D.p, dt, chain_index,
                                   (chained && chain_index == g_chain_test_break) ? 1 : 0, zf ? (const double *)p->pnext.p : nullptr);
            else
                hipLaunchKernelGGL(pcg_pupdate_kernel<false>, dim3(Gv), dim3(FV_BLOCK), 0, ctx->stream, n, iter, p->r.p, (const double *)p->minv.p,
                                   p->pvec.p, (const double *)(red + 1), (const double *)(red + 2), 1, p->scal.p, (double *)nullptr, (int64_t)0,
                                   (const double *)nullptr, (const double *)nullptr, (const double *)nullptr, 0.0);
            FV_LAUNCH_CHECK(ctx);
        }
        it += m;
        if (chained) { // polled once per burst by the caller
            p->last_iters = 1;
            p->spec_valid = true;
            p->z_where = zf ? 2 : 0; // unless the chain stops on the device (the poll then says so)
            return FV_OK;
        }
        const int32_t iters_before = (p->profile && it > m) ? hs->iters : (int32_t)(resume ? resume_it : 0);
        FV_HIP(ctx, hipMemcpyAsync(hs, p->scal.p, sizeof(PcgScalars), hipMemcpyDeviceToHost, ctx->stream));
        FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (p->profile) { // only launches that did real work
            const int64_t live = (int64_t)hs->iters - iters_before;
            for (int64_t k = 0; k < m && k < live; k++)
                for (int c = 0; c < (p->profile_level == 1 ? 2 : 1); c++) {
                    float ms = 0.f;
                    FV_HIP(ctx, hipEventElapsedTime(&ms, p->prof_ev[(size_t)(6 * k + 2 * c)], p->prof_ev[(size_t)(6 * k + 2 * c + 1)]));
                    p->prof_ms[c] += ms;
                    p->prof_launches[c]++;
                }
        }
        if (hs->done)
            break;
        if (chunk < 32)
            chunk *= 2;
    }
#undef FV_PROF
    if (it == 0) {
        FV_HIP(ctx, hipMemcpyAsync(hs, p->scal.p, sizeof(PcgScalars), hipMemcpyDeviceToHost, ctx->stream));
        FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    p->last_iters = hs->iters;
    p->spec_valid = speculate && hs->done == 1 && hs->iters == 1;
    if (speculate && zf && hs->iters >= 1)
        p->z_where = p->spec_valid ? 2 : 0; // as in fv_pcg_solve
    if (use_spec && hs->done == 1 && hs->iters == 0) { // converged at its set-up: keep the prepared set-up (see fv_pcg_solve)
        p->pvec.swap(p->pnext);
        if (p->z_where == 1 || p->z_where == 2)
            p->z_where = 3 - p->z_where;
        p->spec_valid = true;
        p->last_iters = 1;
    }
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
    p->dist->fused_agreed = -1; // (the ranks agree on the fused step at this call's first burst: dist_step)
    p->dist->fused_bursts = 0;
    // ping-pong state + residual carry-over, as in fv_transient_run_fixed (identical decisions on every rank:
    // they depend only on the step index and on the all-reduced iteration count)
    const int64_t refresh = g_carry_refresh;
    const bool pingpong = refresh > 0 && nsteps >= 2 && fv_step_precond(p) != FV_PRECOND_AMG;
    if (pingpong && p->pingpong_slot < 0)
        rc = fv_slot_new(p, &p->pingpong_slot);
    double *u = p->slots[0];
    double *alt = (pingpong && rc == FV_OK) ? p->slots[(size_t)p->pingpong_slot] : nullptr;
    const double *prev = nullptr;
    int64_t s_base = 0; // go on where the previous call left off (fv_problem::resume, as fv_transient_run_fixed)
    {
        const fv_problem::FixedRunResume &rs = p->resume;
        if (g_resume_runs && rs.ok && pingpong && alt && rs.slot == 0 && rs.dt == dt && rs.rtol == rtol && rs.assemble_epoch == p->assemble_epoch &&
            rs.storage_epoch == p->storage_epoch && rs.refresh == (int)refresh && rs.speculate == g_carry_speculate && (rs.prev == u || rs.prev == alt)) {
            prev = rs.prev;
            s_base = rs.steps_since_refresh;
        }
        p->resume.ok = false;
    }
    for (int64_t s = 0; s < nsteps && rc == FV_OK; s++) {
        // bursts of unpolled one-iteration steps, as in fv_transient_run_fixed: with collectives in every step the host
        // needs tens of microseconds to enqueue one, which the device would otherwise spend idle after every poll
        if (pingpong && g_carry_speculate && g_chain_steps >= 2 && prev != nullptr && p->spec_valid && p->last_iters == 1) {
            int L = 0;
            while (L < g_chain_steps && L < 32 && s + L < nsteps && ((s_base + s + L) % refresh) != 0)
                L++;
            if (L >= 2) {
                double *snap_u[32], *snap_alt[32];
                for (int j = 0; j < L && rc == FV_OK; j++) {
                    snap_u[j] = u;
                    snap_alt[j] = alt;
                    rc = dist_step(p, u, dt, rtol, maxiter, &inf, alt, prev, true, true, j, 0, j == L - 1);
                    prev = u;
                    std::swap(u, alt);
                }
                int completed = 0;
                uint32_t zero_mask = 0;
                if (rc == FV_OK)
                    rc = fv_pcg_chain_poll(p, L, &completed, &inf, &zero_mask);
                if (rc != FV_OK)
                    break;
                for (int j = 0; j < completed && j < L; j++)
                    if (iters_per_step)
                        iters_per_step[s + j] = ((zero_mask >> j) & 1u) ? 0 : 1;
                if (completed < L) {
                    u = snap_u[completed];
                    alt = snap_alt[completed];
                    if ((L - 1 - completed) & 1) {
                        p->pvec.swap(p->pnext);
                    }
                    rc = dist_step(p, u, dt, rtol, maxiter, &inf, alt, nullptr, false, false, -1, 1);
                    if (iters_per_step)
                        iters_per_step[s + completed] = inf.iters;
                    prev = u;
                    if (rc == FV_OK && inf.iters > 0)
                        std::swap(u, alt);
                    s += completed;
                } else
                    s += L - 1;
                continue;
            }
        }
        const bool carry = prev != nullptr && ((s_base + s) % refresh) != 0;
        rc = dist_step(p, u, dt, rtol, maxiter, &inf, alt, carry ? prev : nullptr, pingpong && g_carry_speculate, carry);
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
        if (prev != nullptr) {
            fv_problem::FixedRunResume &rs = p->resume;
            rs.ok = true;
            rs.slot = 0;
            rs.dt = dt;
            rs.rtol = rtol;
            rs.assemble_epoch = p->assemble_epoch;
            rs.storage_epoch = p->storage_epoch;
            rs.prev = prev;
            rs.steps_since_refresh = (s_base + nsteps) % refresh;
            rs.refresh = (int)refresh;
            rs.speculate = g_carry_speculate;
        }
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

// ------------------------------------------------------------------ row blocks beyond the fixed-dt run (collective calls)
// solvediffusion on row blocks (FiniteVolume.jl:157-165): Jacobi-PCG on A x = b from x0 (null = 0), every rank its rows.
extern "C" int fv_dist_solve_steady(fv_problem *p, const double *x0_local, double rtol, int64_t maxiter, double *x_local, fv_solve_info *info)
{
    if (!p || !p->dist || !x_local)
        return FV_ERR_ARG;
    fv_ctx *ctx = p->ctx;
    FV_HIP(ctx, hipSetDevice(ctx->device));
    FV_TRY(fv_pcg_prepare(p));
    p->resume.ok = false;
    double *x = p->tmp.p; // n + nhalo + pad
    if (x0_local)
        FV_HIP(ctx, hipMemcpyAsync(x, x0_local, (size_t)p->n * sizeof(double), hipMemcpyDefault, ctx->stream));
    else
        FV_HIP(ctx, hipMemsetAsync(x, 0, (size_t)p->n * sizeof(double), ctx->stream));
    DistSystem ds;
    ds.explicit_system = true;
    ds.sigma = 0.0;
    ds.rhs = p->b.p;
    p->last_iters = 0;
    p->spec_valid = false;
    fv_solve_info inf = {};
    FV_TRY(dist_step(p, x, 1.0, rtol, maxiter, &inf, nullptr, nullptr, false, false, -1, 0, true, &ds));
    if (info)
        *info = inf;
    return fv_copy(ctx, x_local, x, (size_t)p->n * sizeof(double));
}

// One implicit step of the block's state (slot 0) with a caller's forcing: bhat_local = the rank's rows of the
// volume-scaled vector getb(t) of the reference (null = the assembled b), transient.jl:60-76,165-174.
extern "C" int fv_dist_step(fv_problem *p, double dt, const double *bhat_local, double rtol, int64_t maxiter, fv_solve_info *info)
{
    if (!p || !p->dist)
        return FV_ERR_ARG;
    fv_ctx *ctx = p->ctx;
    FV_HIP(ctx, hipSetDevice(ctx->device));
    if (!(dt > 0)) {
        fv_set_error(ctx, "time step must be positive");
        return FV_ERR_DT;
    }
    FV_TRY(fv_pcg_prepare(p));
    p->resume.ok = false;
    DistSystem ds;
    if (bhat_local) {
        FV_HIP(ctx, hipMemcpyAsync(p->rhs.p, bhat_local, (size_t)p->n * sizeof(double), hipMemcpyDefault, ctx->stream));
        ds.rhs = p->rhs.p;
    }
    p->spec_valid = false;
    fv_solve_info inf = {};
    FV_TRY(dist_step(p, p->slots[0], dt, rtol, maxiter, &inf, nullptr, nullptr, false, false, -1, 0, true, &ds));
    FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (info)
        *info = inf;
    return FV_OK;
}

// || a - b ||_2 over all ranks (the step-doubling error of transient.jl:81: one more all-reduce per trial)
static int dist_norm2_diff(fv_problem *p, const double *a, const double *b, double *out)
{
    fv_ctx *ctx = p->ctx;
    fv_dist *d = p->dist;
    int G = vec_grid(p->n);
    if (G > FV_VEC_PARTIALS - 1)
        G = FV_VEC_PARTIALS - 1;
    hipLaunchKernelGGL(diff2_kernel, dim3(G), dim3(FV_BLOCK), 0, ctx->stream, p->n, a, b, p->part_bb.p);
    hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(FV_BLOCK), 0, ctx->stream, (const double *)p->part_bb.p, G, d->red.p);
    FV_LAUNCH_CHECK(ctx);
    FV_TRY(fv_comm_allreduce_sum(ctx, d, d->red.p, 1, ctx->stream));
    double *h = reinterpret_cast<double *>(ctx->pinned);
    FV_HIP(ctx, hipMemcpyAsync(h, d->red.p, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *out = sqrt(*h);
    return FV_OK;
}

// The default stepper of backwardeulerintegrate (step doubling, transient.jl:78-121,136-154) with constant b on row blocks:
// the control flow of fv_transient_run_adaptive, every solve a dist_step, every error norm all-reduced, so all ranks take
// the same decisions.  The block's state (slot 0) is advanced to tfinal; ts_out as the reference's `ts`.
extern "C" int fv_dist_run_adaptive(fv_problem *p, double t0, double tfinal, double dt0, double atol, double rtol, int64_t maxiter,
                                    int64_t max_outer, double *ts_out, int64_t *n_outer, int64_t *n_solves, fv_solve_info *last_info)
{
    if (!p || !p->dist || !(tfinal >= t0) || max_outer < 0 || (max_outer > 0 && !ts_out))
        return FV_ERR_ARG;
    fv_ctx *ctx = p->ctx;
    FV_HIP(ctx, hipSetDevice(ctx->device));
    if (!(dt0 > 0)) {
        fv_set_error(ctx, "time step must be positive");
        return FV_ERR_DT;
    }
    FV_TRY(fv_pcg_prepare(p));
    p->resume.ok = false;
    int32_t scratch[4] = {-1, -1, -1, -1};
    for (int i = 0; i < 4; i++)
        FV_TRY(fv_slot_new(p, &scratch[i]));
    double *U = p->slots[0];
    double *E = p->slots[(size_t)scratch[0]], *S1 = p->slots[(size_t)scratch[1]], *S2 = p->slots[(size_t)scratch[2]], *S3 = p->slots[(size_t)scratch[3]];
    const size_t bytes = (size_t)p->n * sizeof(double);
    fv_solve_info inf = {};
    int64_t nout = 0, nsolves = 0;
    p->spec_valid = false;
    DistSystem plain; // the assembled b, no carried residual: every solve of a trial starts from its own state
    auto solve = [&](const double *src, double *dst, double dt) -> int {
        FV_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
        nsolves++;
        return dist_step(p, dst, dt, rtol, maxiter, &inf, nullptr, nullptr, false, false, -1, 0, true, &plain);
    };
    struct Trial {
        double *result;
        double last;
        bool increase;
    };
    auto twostep = [&](const double *uk, double dt, double *onestep, bool have_onestep, double *two1, double *two, Trial *out) -> int {
        if (!have_onestep)
            FV_TRY(solve(uk, onestep, dt));
        FV_TRY(solve(uk, two1, 0.5 * dt));
        FV_TRY(solve(two1, two, 0.5 * dt));
        double err = 0.0;
        FV_TRY(dist_norm2_diff(p, onestep, two, &err));
        if (err < atol)
            *out = Trial{two, dt, err < atol / 4};
        else
            *out = Trial{two1, 0.5 * dt, false};
        return FV_OK;
    };
    int rc = FV_OK;
    double t = t0;
    double dt = dt0 < tfinal - t0 ? dt0 : tfinal - t0;
    if (ts_out && max_outer > 0)
        ts_out[0] = t0;
    while (rc == FV_OK && t < tfinal && nout < max_outer) {
        Trial tr{};
        if ((rc = twostep(U, dt, S1, false, S2, S3, &tr)) != FV_OK)
            break;
        const double *unew = tr.result;
        if (tr.last < dt) { // rejected: cover dt with accepted sub-steps, transient.jl:93-120
            bool failed = true;
            double elapsed = 0.0, target = tr.last;
            if (hipMemcpyAsync(E, U, bytes, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) {
                rc = FV_ERR_HIP;
                break;
            }
            std::swap(S1, S2); // the half step just computed is the next trial's full step
            while (rc == FV_OK && elapsed < dt) {
                if ((rc = twostep(E, target, S1, failed, S2, S3, &tr)) != FV_OK)
                    break;
                if (tr.last == target) {
                    elapsed += tr.last;
                    if (hipMemcpyAsync(E, tr.result, bytes, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) {
                        rc = FV_ERR_HIP;
                        break;
                    }
                    if (tr.increase)
                        target = 2 * tr.last;
                    failed = false;
                } else if (tr.last < target) {
                    target = tr.last;
                    failed = true;
                    std::swap(S1, S2);
                } else {
                    fv_set_error(ctx, "Code is broken -- laststeptime should never be greater than targetdt");
                    rc = FV_ERR_STATE;
                    break;
                }
                if (dt - elapsed < target)
                    target = dt - elapsed;
            }
            unew = E;
        }
        if (rc != FV_OK)
            break;
        if (hipMemcpyAsync(U, unew, bytes, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) {
            rc = FV_ERR_HIP;
            break;
        }
        t += dt;
        nout++;
        if (ts_out)
            ts_out[nout] = t;
        const double remaining = tfinal - t;
        const double want = tr.increase ? 2 * tr.last : tr.last;
        dt = remaining < want ? remaining : want;
    }
    if (rc == FV_OK && hipStreamSynchronize(ctx->stream) != hipSuccess)
        rc = FV_ERR_HIP;
    if (rc == FV_ERR_HIP)
        fv_set_error(ctx, "fv_dist_run_adaptive: device copy failed: %s", hipGetErrorString(hipGetLastError()));
    if (rc == FV_OK && t < tfinal) {
        fv_set_error(ctx, "fv_dist_run_adaptive: %lld outer steps (max_outer) taken and t = %.17g < tfinal = %.17g; the state is u(t)", (long long)nout, t, tfinal);
        rc = FV_ERR_STATE;
    }
    for (int i = 0; i < 4; i++)
        p->slot_used[(size_t)scratch[i]] = 0;
    if (n_outer)
        *n_outer = nout;
    if (n_solves)
        *n_solves = nsolves;
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

FV_WARM_TU(pcg) // (fv_ctx_create loads every code object of the library up front: fv_warm_modules, fv_ctx.hip)
