// Trajectories kept in HBM and the adjoint sweep that reads them there (finished in round 4).
//
// The reference's adjoint workflow (/root/reference/examples/transientadjoint/ex.jl:100-123: per objective call nine forward +
// adjoint pairs) keeps every outer state of the forward run on the host (`us`, src/transient.jl:136-154), wraps it in a
// piecewise-linear interpolant (getcontinuoussolution, :176-180) and integrates the adjoint ODE
//     d gamma / dt = A' gamma + [dg/du (T - t)]',   gamma(0) = 0                       (adjointintegrate, :188-205)
// with the same stepper, evaluating the forcing dgdu(u_c, T - t) (src/transientadjointutils.jl:13-21) through that interpolant at
// the start time of every solve.  dg/du is non-zero on the observation rows only:
//     dgdu_i(t) = 2 sigma(i, t)^2 (u_i(t) - uobs_i(t)),  i in obsfreenodes.
// Here the states of a run stay where they were computed (fv_trajectory: one n-vector per knot in HBM, the knot times on the
// host), the observation series (uobs, sigma at the observation rows, piecewise linear in time) live on the device
// (fv_observation), a kernel over the observation rows writes the forcing of a solve into a dense vector that is zero elsewhere,
// and the stepper loop of fv_transient_run_adaptive (fv_stepper_run) runs the sweep without any host vector: per solve one
// launch over nobs rows instead of a host closure, an interpolation of two n-vectors on the host and an upload of n doubles.
// The sweep's own state is w = gamma / D (D = Ss * volumes): every step is then the forward step's SPD solve
// (D/dt + A) w+ = dgdu + D w/dt (transpose(D^-1 A) = A D^-1, src/transient.jl:193), the step-doubling error is measured on
// gamma = D w (a D-weighted norm), and the recorded states are gamma — returned as lambda(t) = gamma(T - t) like the reference.
// The objective G = int g dt (transientadjointutils.jl:46-49) and the time integral of dfdp' lambda (fv_param_gradient_integral)
// read the same trajectories.
#include "fv_internal.h"
#include "fv_device.h"

#include <cmath>

struct fv_trajectory {
    fv_problem *p = nullptr;
    std::vector<double *> knots; // device, n doubles each, the problem's internal numbering of the free cells
    std::vector<double> ts;
    std::vector<void *> blocks;  // allocations: BLOCK knots each
    size_t used_in_block = 0, block_knots = 0;
};

struct fv_observation {
    fv_problem *p = nullptr;
    int64_t nobs = 0, nt = 0;
    DevBuf<int32_t> idx; // internal free index of every observation row
    DevBuf<double> uobs, sigma; // nt x nobs, one row per knot
    bool has_sigma = false;
    std::vector<double> tobs;
};

namespace {

__global__ __launch_bounds__(FV_BLOCK) void traj_copy_kernel(int64_t n, const double *src, const double *__restrict__ scale,
                                                              double *dst) // (src may be dst: fv_adjoint_run scales a knot in place)
{
    for (int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x; i < n; i += vec_stride())
        dst[i] = scale ? scale[i] * src[i] : src[i];
}

// (1 - w) a + w b: the Gridded(Linear()) interpolant of getcontinuoussolution, as the host mirror and the oracle write it
__global__ __launch_bounds__(FV_BLOCK) void traj_lerp_kernel(int64_t n, const double *__restrict__ a, const double *__restrict__ b, double w,
                                                              double *__restrict__ out)
{
#pragma clang fp contract(off) // (two products and a sum, each rounded, like the host's and Julia's expression: no FMA)
    for (int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x; i < n; i += vec_stride())
        out[i] = (1.0 - w) * a[i] + w * b[i];
}

// dgdu of transientadjointutils.jl:13-21 at one time, on the observation rows (the rest of `out` is zero and stays zero)
__global__ __launch_bounds__(FV_BLOCK) void adjoint_forcing_kernel(int64_t nobs, const int32_t *__restrict__ idx, const double *__restrict__ ua,
                                                                    const double *__restrict__ ub, double w, const double *__restrict__ oa,
                                                                    const double *__restrict__ ob, const double *__restrict__ sa,
                                                                    const double *__restrict__ sb, double wo, double *__restrict__ out)
{
#pragma clang fp contract(off)
    const int64_t j = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (j >= nobs)
        return;
    const int32_t i = idx[j];
    const double u = (1.0 - w) * ua[i] + w * ub[i];
    const double uo = (1.0 - wo) * oa[j] + wo * ob[j];
    const double s = sa ? (1.0 - wo) * sa[j] + wo * sb[j] : 1.0;
    out[i] = 2.0 * (s * s) * (u - uo);
}

// g of transientadjointutils.jl:4-12 integrated over one piece [a, b] on which u and the observation series are linear: a
// 6-point Gauss-Legendre rule (exact: the integrand is a quartic), one block per piece, the observation rows in a fixed order
struct GPiece {
    double a, b;
    int32_t ku, ko;
};
__constant__ double kGLx[6] = {-0.93246951420315202781, -0.66120938646626451366, -0.23861918608319690863,
                               0.23861918608319690863,  0.66120938646626451366,  0.93246951420315202781};
__constant__ double kGLw[6] = {0.17132449237917034504, 0.36076157304813860757, 0.46791393457269104739,
                               0.46791393457269104739, 0.36076157304813860757, 0.17132449237917034504};
__global__ __launch_bounds__(FV_BLOCK) void observation_piece_kernel(int64_t nobs, const int32_t *__restrict__ idx, const GPiece *__restrict__ pieces,
                                                                      const double *const *__restrict__ knots, const double *__restrict__ tu,
                                                                      const double *__restrict__ uobs, const double *__restrict__ sigma,
                                                                      const double *__restrict__ to, double *__restrict__ out)
{
#pragma clang fp contract(off)
    __shared__ double smem[4];
    const GPiece pc = pieces[blockIdx.x];
    const double *ua = knots[pc.ku], *ub = knots[pc.ku + 1];
    const double *oa = uobs + (int64_t)pc.ko * nobs, *ob = oa + nobs;
    const double *sa = sigma ? sigma + (int64_t)pc.ko * nobs : nullptr, *sb = sa ? sa + nobs : nullptr;
    const double tu0 = tu[pc.ku], tu1 = tu[pc.ku + 1], to0 = to[pc.ko], to1 = to[pc.ko + 1];
    double acc = 0.0;
    for (int64_t j = threadIdx.x; j < nobs; j += FV_BLOCK) {
        const int32_t i = idx[j];
        const double a0 = ua[i], a1 = ub[i], o0 = oa[j], o1 = ob[j];
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < 6; q++) {
            const double t = 0.5 * (pc.a + pc.b) + 0.5 * (pc.b - pc.a) * kGLx[q];
            const double w = (t - tu0) / (tu1 - tu0), wo = (t - to0) / (to1 - to0);
            const double u = (1.0 - w) * a0 + w * a1, uo = (1.0 - wo) * o0 + wo * o1;
            const double sg = sa ? (1.0 - wo) * sa[j] + wo * sb[j] : 1.0;
            s += kGLw[q] * ((sg * sg) * ((u - uo) * (u - uo)));
        }
        acc += s;
    }
    const double tot = block_sum(acc, smem);
    if (threadIdx.x == 0)
        out[blockIdx.x] = 0.5 * (pc.b - pc.a) * tot;
}

__global__ __launch_bounds__(FV_BLOCK) void wdiff_kernel(int64_t n, const double *__restrict__ a, const double *__restrict__ b,
                                                          const double *__restrict__ w, double *__restrict__ part)
{
    __shared__ double smem[4];
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x; i < n; i += vec_stride()) {
        const double d = w[i] * a[i] - w[i] * b[i]; // the difference of the two states the caller would see (gamma = D w)
        acc += d * d;
    }
    const double s = block_sum(acc, smem);
    if (threadIdx.x == 0)
        part[blockIdx.x] = s;
}

// interval and weight of t in the sorted knots ts (as numpy.searchsorted(ts, t, "right") - 1, clipped: the host mirror's rule);
// a t outside the knots by more than rounding is the interpolant's BoundsError
int locate(fv_ctx *ctx, const std::vector<double> &ts, double t, const char *what, int64_t *k, double *w)
{
    const int64_t nt = (int64_t)ts.size();
    if (nt < 2) {
        fv_set_error(ctx, "%s: needs at least two knots", what);
        return FV_ERR_STATE;
    }
    const double span = ts[(size_t)nt - 1] - ts[0], slack = 1e-12 * (span > 0 ? span : 1.0);
    if (!(t >= ts[0] - slack && t <= ts[(size_t)nt - 1] + slack)) {
        fv_set_error(ctx, "%s: BoundsError: attempt to interpolate at t = %.17g outside [%.17g, %.17g]", what, t, ts[0], ts[(size_t)nt - 1]);
        return FV_ERR_ARG;
    }
    if (t < ts[0])
        t = ts[0];
    if (t > ts[(size_t)nt - 1])
        t = ts[(size_t)nt - 1];
    int64_t kk = (int64_t)(std::upper_bound(ts.begin(), ts.end(), t) - ts.begin()) - 1;
    if (kk < 0)
        kk = 0;
    if (kk > nt - 2)
        kk = nt - 2;
    *k = kk;
    *w = (t - ts[(size_t)kk]) / (ts[(size_t)kk + 1] - ts[(size_t)kk]);
    return FV_OK;
}

int traj_new_knot(fv_trajectory *tr, double **out)
{
    fv_problem *p = tr->p;
    const size_t nvec = (size_t)p->n + FV_VEC_PAD;
    if (tr->blocks.empty() || tr->used_in_block == tr->block_knots) {
        // blocks of 1, 2, 4, ... 16 GB (at least one knot, at most 256): a hipMalloc is a device-wide synchronisation of a millisecond
        // or two — one per three knots of a 10^7-cell run doubled the cost of a recorded step —, so few of them
        const size_t shift = tr->blocks.size() < 4 ? tr->blocks.size() : 4;
        size_t per = (((size_t)1 << 30) << shift) / (nvec * sizeof(double));
        per = per < 1 ? 1 : (per > 256 ? 256 : per);
        void *base = nullptr;
        const hipError_t e = hipMalloc(&base, per * nvec * sizeof(double));
        if (e != hipSuccess) {
            fv_set_error(p->ctx, "fv_trajectory: hipMalloc of %zu knots failed: %s", per, hipGetErrorString(e));
            return FV_ERR_NOMEM;
        }
        tr->blocks.push_back(base);
        tr->block_knots = per;
        tr->used_in_block = 0;
    }
    *out = static_cast<double *>(tr->blocks.back()) + tr->used_in_block * nvec;
    tr->used_in_block++;
    return FV_OK;
}

int check_traj(fv_trajectory *tr, const char *who)
{
    if (!tr || !tr->p)
        return FV_ERR_ARG;
    (void)who;
    return FV_OK;
}

} // namespace

int fv_norm2_diff_weighted_device(fv_problem *p, const double *a, const double *b, const double *w, double *out_host)
{
    fv_ctx *ctx = p->ctx;
    const int G = vec_grid(p->n);
    DevBuf<double> part;
    FV_TRY(part.alloc(ctx, (size_t)G));
    hipLaunchKernelGGL(wdiff_kernel, dim3(G), dim3(FV_BLOCK), 0, ctx->stream, p->n, a, b, w, part.p);
    FV_LAUNCH_CHECK(ctx);
    std::vector<double> h((size_t)G);
    FV_TRY(fv_copy(ctx, h.data(), part.p, (size_t)G * sizeof(double)));
    double s = 0.0;
    for (int i = 0; i < G; i++)
        s += h[(size_t)i];
    *out_host = std::sqrt(s);
    return FV_OK;
}

int fv_trajectory_push_device(fv_trajectory *tr, const double *state_dev, double t, const double *scale_dev)
{
    fv_problem *p = tr->p;
    fv_ctx *ctx = p->ctx;
    if (!tr->ts.empty() && !(t > tr->ts.back())) {
        fv_set_error(ctx, "fv_trajectory: knot-vectors must be unique and sorted in increasing order (t = %.17g after %.17g)", t, tr->ts.back());
        return FV_ERR_ARG;
    }
    double *k = nullptr;
    FV_TRY(traj_new_knot(tr, &k));
    hipLaunchKernelGGL(traj_copy_kernel, dim3(vec_grid(p->n)), dim3(FV_BLOCK), 0, ctx->stream, p->n, state_dev, scale_dev, k);
    FV_LAUNCH_CHECK(ctx);
    tr->knots.push_back(k);
    tr->ts.push_back(t);
    return FV_OK;
}

// ------------------------------------------------------------------ C ABI: trajectories
extern "C" int fv_trajectory_create(fv_problem *p, fv_trajectory **out)
{
    if (!p || !out)
        return FV_ERR_ARG;
    if (!p->transient_ready) {
        fv_set_error(p->ctx, "fv_trajectory_create: call fv_transient_begin first");
        return FV_ERR_STATE;
    }
    fv_trajectory *tr = new fv_trajectory;
    tr->p = p;
    p->trajectories.push_back(tr); // (fv_problem_destroy detaches what is still alive: fv_detach_dependents)
    *out = tr;
    return FV_OK;
}

// A trajectory / an observation series whose problem is gone (fv_problem_destroy came first — Julia does not order finalizers):
// its HBM has been released and its problem pointer cleared there; every entry point but destroy refuses it (check_traj).
void fv_detach_dependents(fv_problem *p)
{
    for (fv_trajectory *tr : p->trajectories) {
        for (void *b : tr->blocks)
            (void)hipFree(b);
        tr->blocks.clear();
        tr->knots.clear();
        tr->ts.clear();
        tr->used_in_block = tr->block_knots = 0;
        tr->p = nullptr;
    }
    p->trajectories.clear();
    for (fv_observation *o : p->observations) {
        o->idx.release();
        o->uobs.release();
        o->sigma.release();
        o->p = nullptr;
    }
    p->observations.clear();
    p->recording = nullptr;
}

extern "C" int fv_trajectory_clear(fv_trajectory *tr)
{
    FV_TRY(check_traj(tr, "fv_trajectory_clear"));
    (void)hipSetDevice(tr->p->ctx->device);
    (void)hipStreamSynchronize(tr->p->ctx->stream);
    for (void *b : tr->blocks)
        (void)hipFree(b);
    tr->blocks.clear();
    tr->knots.clear();
    tr->ts.clear();
    tr->used_in_block = tr->block_knots = 0;
    return FV_OK;
}

extern "C" int fv_trajectory_destroy(fv_trajectory *tr)
{
    if (!tr)
        return FV_OK;
    if (tr->p) {
        if (tr->p->recording == tr)
            tr->p->recording = nullptr;
        std::vector<fv_trajectory *> &v = tr->p->trajectories;
        v.erase(std::remove(v.begin(), v.end(), tr), v.end());
        fv_trajectory_clear(tr);
    }
    delete tr;
    return FV_OK;
}

extern "C" int fv_trajectory_push_state(fv_trajectory *tr, int32_t slot, double t)
{
    FV_TRY(check_traj(tr, "fv_trajectory_push_state"));
    fv_problem *p = tr->p;
    FV_HIP(p->ctx, hipSetDevice(p->ctx->device));
    if (slot < 0 || slot >= (int32_t)p->slots.size() || !p->slot_used[(size_t)slot]) {
        fv_set_error(p->ctx, "invalid state slot %d", (int)slot);
        return FV_ERR_ARG;
    }
    return fv_trajectory_push_device(tr, p->slots[(size_t)slot], t, nullptr);
}

extern "C" int fv_trajectory_push_free(fv_trajectory *tr, const double *u_free, double t)
{
    FV_TRY(check_traj(tr, "fv_trajectory_push_free"));
    if (!u_free)
        return FV_ERR_ARG;
    fv_problem *p = tr->p;
    fv_ctx *ctx = p->ctx;
    FV_HIP(ctx, hipSetDevice(ctx->device));
    FV_TRY(fv_pcg_prepare(p));
    FV_TRY(fv_free_in(p, p->tmp.p, u_free)); // (the caller's numbering of the free cells)
    return fv_trajectory_push_device(tr, p->tmp.p, t, nullptr);
}

extern "C" int fv_trajectory_size(fv_trajectory *tr, int64_t *nknots)
{
    FV_TRY(check_traj(tr, "fv_trajectory_size"));
    if (!nknots)
        return FV_ERR_ARG;
    *nknots = (int64_t)tr->ts.size();
    return FV_OK;
}

extern "C" int fv_trajectory_times(fv_trajectory *tr, double *ts, int64_t cap)
{
    FV_TRY(check_traj(tr, "fv_trajectory_times"));
    if (!ts || cap < (int64_t)tr->ts.size())
        return FV_ERR_ARG;
    std::copy(tr->ts.begin(), tr->ts.end(), ts);
    return FV_OK;
}

// lambda(t) = gamma(T - t): the knots in reverse order at the times T - t (what adjointintegrate returns, transient.jl:204)
extern "C" int fv_trajectory_reverse_time(fv_trajectory *tr, double T)
{
    FV_TRY(check_traj(tr, "fv_trajectory_reverse_time"));
    std::reverse(tr->knots.begin(), tr->knots.end());
    std::reverse(tr->ts.begin(), tr->ts.end());
    for (double &t : tr->ts)
        t = T - t;
    return FV_OK;
}

extern "C" int fv_trajectory_get_free(fv_trajectory *tr, int64_t k, double *u_free)
{
    FV_TRY(check_traj(tr, "fv_trajectory_get_free"));
    if (!u_free || k < 0 || k >= (int64_t)tr->knots.size())
        return FV_ERR_ARG;
    FV_HIP(tr->p->ctx, hipSetDevice(tr->p->ctx->device));
    return fv_free_out(tr->p, u_free, tr->knots[(size_t)k]);
}

extern "C" int fv_trajectory_get_nodes(fv_trajectory *tr, int64_t k, double *u_nodes)
{
    FV_TRY(check_traj(tr, "fv_trajectory_get_nodes"));
    if (!u_nodes || k < 0 || k >= (int64_t)tr->knots.size())
        return FV_ERR_ARG;
    fv_problem *p = tr->p;
    fv_ctx *ctx = p->ctx;
    FV_HIP(ctx, hipSetDevice(ctx->device));
    DevBuf<double> hd;
    FV_TRY(hd.alloc(ctx, (size_t)p->N));
    FV_TRY(fv_scatter_nodes(p, tr->knots[(size_t)k], hd.p)); // freenodes2nodes, transient.jl:172
    return fv_copy(ctx, u_nodes, hd.p, (size_t)p->N * sizeof(double));
}

// u_c(t) on the free cells (getcontinuoussolution, transient.jl:176-180), interpolated on the device
extern "C" int fv_trajectory_eval_free(fv_trajectory *tr, double t, double *u_free)
{
    FV_TRY(check_traj(tr, "fv_trajectory_eval_free"));
    if (!u_free)
        return FV_ERR_ARG;
    fv_problem *p = tr->p;
    fv_ctx *ctx = p->ctx;
    FV_HIP(ctx, hipSetDevice(ctx->device));
    int64_t k = 0;
    double w = 0.0;
    FV_TRY(locate(ctx, tr->ts, t, "fv_trajectory_eval_free", &k, &w));
    FV_TRY(fv_pcg_prepare(p));
    hipLaunchKernelGGL(traj_lerp_kernel, dim3(vec_grid(p->n)), dim3(FV_BLOCK), 0, ctx->stream, p->n, (const double *)tr->knots[(size_t)k],
                       (const double *)tr->knots[(size_t)k + 1], w, p->tmp.p);
    FV_LAUNCH_CHECK(ctx);
    return fv_free_out(p, u_free, p->tmp.p);
}

// While a trajectory is set, fv_transient_run_fixed and fv_transient_run_adaptive on this problem push the state of every outer
// step (the adaptive run also its initial state; a fixed-dt run counts its times from t0 in steps of dt).  NULL stops it.
extern "C" int fv_trajectory_record(fv_problem *p, fv_trajectory *tr, double t0)
{
    if (!p || (tr && tr->p != p))
        return FV_ERR_ARG;
    p->recording = tr;
    p->record_t = t0;
    p->resume.ok = false; // (recording runs are polled step by step: do not continue a burst regime's carried state blindly)
    return FV_OK;
}

// ------------------------------------------------------------------ C ABI: observation series
extern "C" int fv_observation_create(fv_problem *p, int64_t nobs, const int64_t *obs_free, int64_t nt, const double *tobs, const double *uobs,
                                     const double *sigma, fv_observation **out)
{
    if (!p || !out || nobs < 0 || nt < 2 || !tobs || !uobs || (nobs > 0 && !obs_free))
        return FV_ERR_ARG;
    fv_ctx *ctx = p->ctx;
    FV_HIP(ctx, hipSetDevice(ctx->device));
    for (int64_t k = 1; k < nt; k++)
        if (!(tobs[k] > tobs[k - 1])) {
            fv_set_error(ctx, "fv_observation_create: knot-vectors must be unique and sorted in increasing order");
            return FV_ERR_ARG;
        }
    std::vector<int32_t> idx((size_t)nobs);
    std::vector<int32_t> perm;
    if (p->reordered) { // perm[canonical free index] = internal row
        perm.resize((size_t)p->n);
        FV_TRY(fv_copy(ctx, perm.data(), p->perm.p, (size_t)p->n * sizeof(int32_t)));
    }
    for (int64_t j = 0; j < nobs; j++) {
        const int64_t f = obs_free[j];
        if (f < 1 || f > p->n) {
            fv_set_error(ctx, "fv_observation_create: observation row %lld is not a free-cell index in 1..%lld", (long long)f, (long long)p->n);
            return FV_ERR_ARG;
        }
        idx[(size_t)j] = p->reordered ? perm[(size_t)(f - 1)] : (int32_t)(f - 1);
    }
    fv_observation *o = new fv_observation;
    o->p = p;
    p->observations.push_back(o);
    o->nobs = nobs;
    o->nt = nt;
    o->tobs.assign(tobs, tobs + nt);
    o->has_sigma = sigma != nullptr;
    int rc = o->idx.alloc(ctx, (size_t)(nobs > 0 ? nobs : 1));
    if (rc == FV_OK)
        rc = o->uobs.alloc(ctx, (size_t)(nt * (nobs > 0 ? nobs : 1)));
    if (rc == FV_OK && sigma)
        rc = o->sigma.alloc(ctx, (size_t)(nt * (nobs > 0 ? nobs : 1)));
    if (rc == FV_OK && nobs > 0) {
        rc = fv_copy(ctx, o->idx.p, idx.data(), (size_t)nobs * sizeof(int32_t));
        if (rc == FV_OK)
            rc = fv_copy(ctx, o->uobs.p, uobs, (size_t)(nt * nobs) * sizeof(double));
        if (rc == FV_OK && sigma)
            rc = fv_copy(ctx, o->sigma.p, sigma, (size_t)(nt * nobs) * sizeof(double));
    }
    if (rc != FV_OK) {
        delete o;
        return rc;
    }
    *out = o;
    return FV_OK;
}

extern "C" int fv_observation_destroy(fv_observation *o)
{
    if (o) {
        if (o->p) {
            (void)hipSetDevice(o->p->ctx->device);
            std::vector<fv_observation *> &v = o->p->observations;
            v.erase(std::remove(v.begin(), v.end(), o), v.end());
        }
        delete o;
    }
    return FV_OK;
}

// G = int_t0^t1 g(u_c, t) dt with g = sum_i sigma(i, t)^2 (u_i(t) - uobs_i(t))^2 (transientadjointutils.jl:4-12, 46-49; QuadGK there):
// u_c and the series are piecewise linear, so between two consecutive knots of either the integrand is a quartic in t and a
// 6-point Gauss-Legendre rule per piece is exact.
extern "C" int fv_observation_integral(fv_trajectory *u, fv_observation *o, double t0, double t1, double *G)
{
    if (!u || !o || !G || !u->p || u->p != o->p || !(t1 >= t0))
        return FV_ERR_ARG;
    fv_problem *p = u->p;
    fv_ctx *ctx = p->ctx;
    FV_HIP(ctx, hipSetDevice(ctx->device));
    *G = 0.0;
    if (o->nobs == 0 || t1 == t0)
        return FV_OK;
    int64_t k = 0;
    double w = 0.0;
    FV_TRY(locate(ctx, u->ts, t0, "fv_observation_integral (solution)", &k, &w));
    FV_TRY(locate(ctx, u->ts, t1, "fv_observation_integral (solution)", &k, &w));
    FV_TRY(locate(ctx, o->tobs, t0, "fv_observation_integral (observations)", &k, &w));
    FV_TRY(locate(ctx, o->tobs, t1, "fv_observation_integral (observations)", &k, &w));
    std::vector<double> cuts;
    cuts.push_back(t0);
    for (double t : u->ts)
        if (t > t0 && t < t1)
            cuts.push_back(t);
    for (double t : o->tobs)
        if (t > t0 && t < t1)
            cuts.push_back(t);
    cuts.push_back(t1);
    std::sort(cuts.begin(), cuts.end());
    cuts.erase(std::unique(cuts.begin(), cuts.end()), cuts.end());
    std::vector<GPiece> pieces;
    for (size_t c = 0; c + 1 < cuts.size(); c++) {
        GPiece pc;
        pc.a = cuts[c];
        pc.b = cuts[c + 1];
        const double mid = 0.5 * (pc.a + pc.b);
        int64_t ku = 0, ko = 0;
        FV_TRY(locate(ctx, u->ts, mid, "fv_observation_integral", &ku, &w));
        FV_TRY(locate(ctx, o->tobs, mid, "fv_observation_integral", &ko, &w));
        pc.ku = (int32_t)ku;
        pc.ko = (int32_t)ko;
        pieces.push_back(pc);
    }
    const size_t np = pieces.size();
    DevBuf<GPiece> dp;
    DevBuf<const double *> dk;
    DevBuf<double> dtu, dto, dout;
    FV_TRY(dp.alloc(ctx, np));
    FV_TRY(dk.alloc(ctx, u->knots.size()));
    FV_TRY(dtu.alloc(ctx, u->ts.size()));
    FV_TRY(dto.alloc(ctx, o->tobs.size()));
    FV_TRY(dout.alloc(ctx, np));
    FV_TRY(fv_copy(ctx, dp.p, pieces.data(), np * sizeof(GPiece)));
    FV_TRY(fv_copy(ctx, dk.p, u->knots.data(), u->knots.size() * sizeof(double *)));
    FV_TRY(fv_copy(ctx, dtu.p, u->ts.data(), u->ts.size() * sizeof(double)));
    FV_TRY(fv_copy(ctx, dto.p, o->tobs.data(), o->tobs.size() * sizeof(double)));
    hipLaunchKernelGGL(observation_piece_kernel, dim3((unsigned)np), dim3(FV_BLOCK), 0, ctx->stream, o->nobs, (const int32_t *)o->idx.p,
                       (const GPiece *)dp.p, (const double *const *)dk.p, (const double *)dtu.p, (const double *)o->uobs.p,
                       o->has_sigma ? (const double *)o->sigma.p : nullptr, (const double *)dto.p, dout.p);
    FV_LAUNCH_CHECK(ctx);
    std::vector<double> h(np);
    FV_TRY(fv_copy(ctx, h.data(), dout.p, np * sizeof(double)));
    double s = 0.0;
    for (double v : h)
        s += v;
    *G = s;
    return FV_OK;
}

// ------------------------------------------------------------------ C ABI: the adjoint sweep
// adjointintegrate (transient.jl:188-205) with getdgdu = t -> dgdu(u_c, t) of getadjointfunctions: gamma(0) = 0, the default
// adaptive stepper (adaptive != 0) or the fixed one from dt0, the forcing of a solve that starts at gamma-time t evaluated at
// T - t from the trajectory `u` and the observation series `o`.  lambda_out receives the outer states as lambda(t) = gamma(T - t),
// knots ascending in t like the reference's reversed arrays.  T = tfinal (the reference passes tspan[2]); t0 is tspan[1].
extern "C" int fv_adjoint_run(fv_problem *p, fv_trajectory *u, fv_observation *o, double t0, double tfinal, double dt0, int adaptive, double atol,
                              double rtol, int64_t maxiter, int64_t max_outer, fv_trajectory *lambda_out, int64_t *n_outer, int64_t *n_solves,
                              fv_solve_info *last_info)
{
    if (!p || !u || !o || !lambda_out || u->p != p || o->p != p || lambda_out->p != p || !(tfinal >= t0) || max_outer < 0)
        return FV_ERR_ARG;
    fv_ctx *ctx = p->ctx;
    FV_HIP(ctx, hipSetDevice(ctx->device));
    if (!p->transient_ready) {
        fv_set_error(ctx, "fv_adjoint_run: call fv_transient_begin first");
        return FV_ERR_STATE;
    }
    if (!(dt0 > 0)) {
        fv_set_error(ctx, "time step must be positive");
        return FV_ERR_DT;
    }
    if (p->dist) {
        fv_set_error(ctx, "fv_adjoint_run: not on a row block");
        return FV_ERR_STATE;
    }
    if (!lambda_out->ts.empty()) {
        fv_set_error(ctx, "fv_adjoint_run: the output trajectory must be empty");
        return FV_ERR_ARG;
    }
    FV_TRY(fv_pcg_prepare(p));
    // the forcing vector: zero except on the observation rows, which every solve's kernel rewrites
    // (three of them: the solves of one step-doubling attempt may be enqueued together, fv_small_twostep)
    const size_t flen = (size_t)p->n + (size_t)p->nhalo + FV_VEC_PAD;
    DevBuf<double> forcing;
    FV_TRY(forcing.alloc(ctx, 3 * flen));
    FV_TRY(forcing.zero(ctx));
    int32_t slot = -1;
    FV_TRY(fv_slot_new(p, &slot));
    FV_HIP(ctx, hipMemsetAsync(p->slots[(size_t)slot], 0, ((size_t)p->n + (size_t)p->nhalo) * sizeof(double), ctx->stream)); // gamma0 = zeros, :201
    const double T = tfinal;
    fv_trajectory *gam = lambda_out;
    FvStepHooks h;
    h.mode = FV_STEP_W;
    h.norm_weight = p->D.p;
    h.forcing = [&](double t, int which, const double **rhs) -> int {
        double *const fbuf = forcing.p + (size_t)(which >= 0 && which < 3 ? which : 0) * flen;
        const double tau = T - t; // getdgdu(tspan[2] - t), :202
        int64_t k = 0, ko = 0;
        double w = 0.0, wo = 0.0;
        FV_TRY(locate(ctx, u->ts, tau, "fv_adjoint_run (solution)", &k, &w));
        FV_TRY(locate(ctx, o->tobs, tau, "fv_adjoint_run (observations)", &ko, &wo));
        if (o->nobs > 0) {
            const double *oa = o->uobs.p + ko * o->nobs, *sa = o->has_sigma ? o->sigma.p + ko * o->nobs : nullptr;
            hipLaunchKernelGGL(adjoint_forcing_kernel, dim3(fv_blocks(o->nobs)), dim3(FV_BLOCK), 0, ctx->stream, o->nobs, (const int32_t *)o->idx.p,
                               (const double *)u->knots[(size_t)k], (const double *)u->knots[(size_t)k + 1], w, oa, oa + o->nobs, sa,
                               sa ? sa + o->nobs : nullptr, wo, fbuf);
            FV_LAUNCH_CHECK(ctx);
        }
        *rhs = fbuf;
        return FV_OK;
    };
    h.record = [&](const double *state, double t) -> int { return fv_trajectory_push_device(gam, state, t, p->D.p); }; // gamma = D w
    std::vector<double> ts((size_t)max_outer + 2);
    p->resume.ok = false;
    int rc = fv_stepper_run(p, slot, t0, tfinal, dt0, adaptive == 0, atol, rtol, maxiter, max_outer, ts.data(), n_outer, n_solves, last_info, h);
    p->slot_used[(size_t)slot] = 0;
    p->resume.ok = false;
    if (rc != FV_OK)
        return rc;
    return fv_trajectory_reverse_time(gam, T);
}

// ------------------------------------------------------------------ C ABI: the gradient integral over two trajectories
// fv_param_gradient_integral with u and lambda taken from trajectories in HBM: the merged knots of both inside [t0, t1], both
// series interpolated there on the device (in passes of a bounded number of knots), the same per-face / per-row kernels.
int fv_param_gradient_integral_device(fv_problem *p, int64_t kc, const double *ts_dev, const double *X, const double *L, const double *D,
                                      int logtransform, int accumulate, double *gk, double *gd, double *gs); // fv_gradient.hip

extern "C" int fv_param_gradient_integral_traj(fv_problem *p, fv_trajectory *u, fv_trajectory *lam, double t0, double t1, int scale_by_storage,
                                               const double *lam_scale_free, int logtransform, double *face_k, double *face_dir, double *row_src)
{
    if (!p || !u || !lam || u->p != p || lam->p != p || !face_k || !face_dir || !row_src || !(t1 >= t0))
        return FV_ERR_ARG;
    fv_ctx *ctx = p->ctx;
    if (!p->assembled || p->from_csc || p->dist) {
        fv_set_error(ctx, "fv_param_gradient_integral_traj: needs a mesh problem after fv_assemble (not a CSC import or a row block)");
        return FV_ERR_STATE;
    }
    if (scale_by_storage && lam_scale_free) {
        fv_set_error(ctx, "fv_param_gradient_integral_traj: scale_by_storage and lam_scale_free exclude each other");
        return FV_ERR_ARG;
    }
    FV_HIP(ctx, hipSetDevice(ctx->device));
    int64_t k = 0;
    double w = 0.0;
    std::vector<double> knots;
    knots.push_back(t0);
    for (const fv_trajectory *tr : {u, lam}) {
        FV_TRY(locate(ctx, tr->ts, t0, "fv_param_gradient_integral_traj", &k, &w));
        FV_TRY(locate(ctx, tr->ts, t1, "fv_param_gradient_integral_traj", &k, &w));
        for (double t : tr->ts)
            if (t > t0 && t < t1)
                knots.push_back(t);
    }
    knots.push_back(t1);
    std::sort(knots.begin(), knots.end());
    knots.erase(std::unique(knots.begin(), knots.end()), knots.end());
    const int64_t nt = (int64_t)knots.size(), n = p->n, F = p->F;
    DevBuf<double> gk, gd, gs, scale;
    FV_TRY(gk.alloc(ctx, (size_t)(F > 0 ? F : 1)));
    FV_TRY(gd.alloc(ctx, (size_t)(F > 0 ? F : 1)));
    FV_TRY(gs.alloc(ctx, (size_t)(n > 0 ? n : 1)));
    FV_TRY(gk.zero(ctx));
    FV_TRY(gd.zero(ctx));
    FV_TRY(gs.zero(ctx));
    const double *D = scale_by_storage ? p->D.p : nullptr;
    if (lam_scale_free) { // lambda_f * scale_f with the caller's own factors (e.g. the reference's division by the FREE index's volume)
        FV_TRY(scale.alloc(ctx, (size_t)n));
        FV_TRY(fv_free_in(p, scale.p, lam_scale_free));
    }
    if (nt >= 2) {
        int64_t chunk = (int64_t)(1ull << 27) / (n > 0 ? n : 1);
        extern int g_gradient_knots_per_pass;
        if (g_gradient_knots_per_pass >= 2)
            chunk = g_gradient_knots_per_pass;
        chunk = chunk < 2 ? 2 : (chunk > nt ? nt : chunk);
        DevBuf<double> X, L, Tt;
        FV_TRY(X.alloc(ctx, (size_t)(chunk * n)));
        FV_TRY(L.alloc(ctx, (size_t)(chunk * n)));
        FV_TRY(Tt.alloc(ctx, (size_t)chunk));
        int accumulate = 0;
        for (int64_t k0 = 0; k0 + 1 < nt; k0 += chunk - 1) {
            const int64_t kc = nt - k0 < chunk ? nt - k0 : chunk;
            for (int64_t j = 0; j < kc; j++) {
                const double t = knots[(size_t)(k0 + j)];
                int64_t ku = 0, kl = 0;
                double wu = 0.0, wl = 0.0;
                FV_TRY(locate(ctx, u->ts, t, "fv_param_gradient_integral_traj", &ku, &wu));
                FV_TRY(locate(ctx, lam->ts, t, "fv_param_gradient_integral_traj", &kl, &wl));
                hipLaunchKernelGGL(traj_lerp_kernel, dim3(vec_grid(n)), dim3(FV_BLOCK), 0, ctx->stream, n, (const double *)u->knots[(size_t)ku],
                                   (const double *)u->knots[(size_t)ku + 1], wu, X.p + j * n);
                hipLaunchKernelGGL(traj_lerp_kernel, dim3(vec_grid(n)), dim3(FV_BLOCK), 0, ctx->stream, n, (const double *)lam->knots[(size_t)kl],
                                   (const double *)lam->knots[(size_t)kl + 1], wl, L.p + j * n);
                if (scale.p)
                    hipLaunchKernelGGL(traj_copy_kernel, dim3(vec_grid(n)), dim3(FV_BLOCK), 0, ctx->stream, n, (const double *)(L.p + j * n),
                                       (const double *)scale.p, L.p + j * n);
            }
            FV_LAUNCH_CHECK(ctx);
            FV_TRY(fv_copy(ctx, Tt.p, knots.data() + k0, (size_t)kc * sizeof(double)));
            FV_TRY(fv_param_gradient_integral_device(p, kc, Tt.p, X.p, L.p, D, logtransform, accumulate, gk.p, gd.p, gs.p));
            accumulate = 1;
        }
    }
    FV_TRY(fv_copy(ctx, face_k, gk.p, (size_t)F * sizeof(double)));
    FV_TRY(fv_copy(ctx, face_dir, gd.p, (size_t)F * sizeof(double)));
    return fv_free_out(p, row_src, gs.p);
}

FV_WARM_TU(trajectory) // (fv_ctx_create loads every code object of the library up front: fv_warm_modules, fv_ctx.hip)
