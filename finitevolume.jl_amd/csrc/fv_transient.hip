// Steady solve, implicit time stepping and device-resident state vectors.
// Reference: /root/reference/src/FiniteVolume.jl:157-165 (solvediffusion),
// src/transient.jl:7-22 (scalebyvolume!), :60-76 (backwardeuleronestep!),
// :130-154 (fixed stepper + outer loop), :188-205 (adjointintegrate).
//
// The reference row-scales A by 1/(Ss*V) and shifts the stored diagonal by
// +-1/dt around every solve.  Here A is never touched: each step solves the
// equivalent symmetric positive definite system
//        (D/dt + A) u+ = D u/dt + b ,     D = diag(Ss*V_free)
// with the shift applied inside the SpMV and the Jacobi diagonal.
#include "fv_internal.h"

static int need_assembled(fv_problem *p, const char *who)
{
    if (!p->assembled) {
        fv_set_error(p->ctx, "%s: call fv_assemble first", who);
        return FV_ERR_STATE;
    }
    return FV_OK;
}

static int slot_ptr(fv_problem *p, int32_t slot, double **out)
{
    if (slot < 0 || slot >= (int32_t)p->slots.size() || !p->slot_used[(size_t)slot]) {
        fv_set_error(p->ctx, "invalid state slot %d", (int)slot);
        return FV_ERR_ARG;
    }
    *out = p->slots[(size_t)slot];
    return FV_OK;
}

int fv_slot_new(fv_problem *p, int32_t *slot)
{
    for (size_t i = 0; i < p->slots.size(); i++)
        if (!p->slot_used[i]) {
            p->slot_used[i] = 1;
            *slot = (int32_t)i;
            return FV_OK;
        }
    void *base = nullptr;
    FV_TRY(fv_vec_alloc_raw(p, (size_t)p->n + (size_t)p->nhalo + FV_VEC_PAD, true, &base)); // (a state is written in every step: fv_place.hip)
    double *d = reinterpret_cast<double *>(static_cast<char *>(base) + fv_vec_skew(((size_t)p->n + (size_t)p->nhalo + FV_VEC_PAD) * sizeof(double)));
    p->slot_bases.push_back(base);
    p->slots.push_back(d);
    p->slot_used.push_back(1);
    *slot = (int32_t)p->slots.size() - 1;
    return FV_OK;
}

// ------------------------------------------------------------------ steady (a8)
extern "C" int fv_solve_steady(fv_problem *p, const double *x0_free, double rtol, int64_t maxiter, double *head_nodes,
                               double *result_free, double *resnorm, int64_t resnorm_cap, fv_solve_info *info)
{
    if (!p)
        return FV_ERR_ARG;
    fv_ctx *ctx = p->ctx;
    FV_HIP(ctx, hipSetDevice(ctx->device));
    FV_TRY(need_assembled(p, "fv_solve_steady"));
    FV_TRY(fv_pcg_prepare(p));
    if (resnorm && resnorm_cap > 0) {
        const int64_t want = resnorm_cap < maxiter ? resnorm_cap : maxiter;
        if (want > p->hist_cap) {
            FV_TRY(p->hist.alloc(ctx, (size_t)want));
            p->hist_cap = want;
        }
    }
    double *x = p->tmp.p;
    if (x0_free)
        FV_TRY(fv_free_in(p, x, x0_free));
    fv_solve_info local;
    double *saved_hist = p->hist.p;
    const int64_t saved_cap = p->hist_cap;
    if (!resnorm) { // no history wanted: do not write it
        p->hist.p = nullptr;
        p->hist_cap = 0;
    }
    PcgSystem sys;
    sys.rhs = p->b.p;
    sys.x0_zero = x0_free == nullptr;
    // FV_PRECOND_AUTO: like the reference's defaultlinearsolver (transient.jl:50-58: plain CG first, the AMG-preconditioned
    // one from the current iterate if that did not converge) — a quarter of the budget (at most 100 iterations) of
    // Jacobi-PCG, which settles easy problems without any set-up, then the V-cycle for the rest.
    const int kind = p->precond;
    int64_t first_budget = maxiter;
    if (kind == FV_PRECOND_AUTO) {
        p->precond = FV_PRECOND_JACOBI;
        first_budget = maxiter / 4 < 100 ? maxiter / 4 : 100;
    }
    int rc = FV_OK;
    local = fv_solve_info{};
    if (kind != FV_PRECOND_AUTO || first_budget > 0)
        rc = fv_pcg_solve(p, x, sys, rtol, first_budget, &local, true);
    const bool exhausted = local.iters >= first_budget; // stopped by the budget, not by a breakdown
    if (rc == FV_OK && kind == FV_PRECOND_AUTO && !local.converged && exhausted && maxiter > first_budget) {
        p->precond = FV_PRECOND_AMG;
        if (p->hist.p) { // the history continues behind the first phase
            p->hist.p += local.iters;
            p->hist_cap -= local.iters;
        }
        if (first_budget > 0)
            sys.x0_zero = false;
        fv_solve_info second = {};
        rc = fv_pcg_solve(p, x, sys, rtol, maxiter - first_budget, &second, true);
        second.iters += local.iters;
        second.solve_ms += local.solve_ms;
        local = second;
    }
    p->precond = kind;
    p->hist.p = saved_hist;
    p->hist_cap = saved_cap;
    FV_TRY(rc);
    if (resnorm && resnorm_cap > 0) {
        int64_t len = local.iters < resnorm_cap ? local.iters : resnorm_cap;
        if (len > p->hist_cap)
            len = p->hist_cap;
        FV_TRY(fv_copy(ctx, resnorm, p->hist.p, (size_t)len * sizeof(double)));
        local.resnorm_len = len;
    }
    if (result_free)
        FV_TRY(fv_free_out(p, result_free, x));
    if (head_nodes) {
        DevBuf<double> hd;
        FV_TRY(hd.alloc(ctx, (size_t)p->N));
        FV_TRY(fv_scatter_nodes(p, x, hd.p));
        FV_TRY(fv_copy(ctx, head_nodes, hd.p, (size_t)p->N * sizeof(double)));
    }
    if (info)
        *info = local;
    return FV_OK;
}

// ------------------------------------------------------------------ transient set-up (a9)
__global__ __launch_bounds__(FV_BLOCK) void storage_kernel(int64_t n, const int32_t *__restrict__ f2n, const double *__restrict__ vol,
                                                            double Ss, double *__restrict__ D)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i < n)
        D[i] = vol ? Ss * vol[f2n[i]] : Ss; // `Ss * volumes`, transient.jl:160,169
}

extern "C" int fv_transient_begin(fv_problem *p, double Ss, const double *volumes, const double *u0_nodes)
{
    if (!p)
        return FV_ERR_ARG;
    fv_ctx *ctx = p->ctx;
    FV_HIP(ctx, hipSetDevice(ctx->device));
    FV_TRY(need_assembled(p, "fv_transient_begin"));
    FV_TRY(fv_pcg_prepare(p));
    FV_TRY(p->D.alloc(ctx, (size_t)p->n + 2));
    p->Ss = Ss;
    p->minv_valid = false; // D changes: cached Jacobi diagonal and folded values are stale
    p->storage_epoch++;    // ... and so is the aggregated storage term of an AMG hierarchy
    p->shifted_epoch = -1;
    p->dia_epoch = -1; // ... and the solver's copies of a folded matrix (lane-major, symmetric), which are keyed on sigma only
    p->sym_epoch = -1;
    DevBuf<double> dvol;
    const double *vol = nullptr;
    if (volumes) {
        FV_TRY(dvol.alloc(ctx, (size_t)p->N));
        FV_HIP(ctx, hipMemcpyAsync(dvol.p, volumes, (size_t)p->N * sizeof(double), hipMemcpyDefault, ctx->stream));
        vol = dvol.p;
    } else if (p->from_grid)
        vol = p->gridvol.p;
    if (p->n > 0) {
        hipLaunchKernelGGL(storage_kernel, dim3(fv_blocks(p->n)), dim3(FV_BLOCK), 0, ctx->stream, p->n, p->f2n.p, vol, Ss, p->D.p);
        FV_LAUNCH_CHECK(ctx);
    }
    if (p->slots.empty()) {
        int32_t s0;
        FV_TRY(fv_slot_new(p, &s0));
    }
    p->slot_used[0] = 1;
    if (u0_nodes) {
        DevBuf<double> du;
        FV_TRY(du.alloc(ctx, (size_t)p->N));
        FV_HIP(ctx, hipMemcpyAsync(du.p, u0_nodes, (size_t)p->N * sizeof(double), hipMemcpyDefault, ctx->stream));
        FV_TRY(fv_gather_free(p, du.p, p->slots[0])); // u0[freenodes], transient.jl:170
        FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    } else
        FV_HIP(ctx, hipMemsetAsync(p->slots[0], 0, (size_t)p->n * sizeof(double), ctx->stream));
    FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    p->transient_ready = true;
    return FV_OK;
}

static int need_transient(fv_problem *p, const char *who)
{
    if (!p->transient_ready) {
        fv_set_error(p->ctx, "%s: call fv_transient_begin first", who);
        return FV_ERR_STATE;
    }
    return FV_OK;
}

// ------------------------------------------------------------------ state vectors
extern "C" int fv_state_alloc(fv_problem *p, int32_t *slot)
{
    if (!p || !slot)
        return FV_ERR_ARG;
    FV_HIP(p->ctx, hipSetDevice(p->ctx->device));
    FV_TRY(need_transient(p, "fv_state_alloc"));
    return fv_slot_new(p, slot);
}

extern "C" int fv_state_free(fv_problem *p, int32_t slot)
{
    if (!p)
        return FV_ERR_ARG;
    double *d;
    FV_TRY(slot_ptr(p, slot, &d));
    if (slot == 0) {
        fv_set_error(p->ctx, "slot 0 is owned by the problem");
        return FV_ERR_ARG;
    }
    p->slot_used[(size_t)slot] = 0;
    return FV_OK;
}

extern "C" int fv_state_set_nodes(fv_problem *p, int32_t slot, const double *u_nodes)
{
    if (!p || !u_nodes)
        return FV_ERR_ARG;
    fv_ctx *ctx = p->ctx;
    FV_HIP(ctx, hipSetDevice(ctx->device));
    double *d;
    p->resume.ok = false; // the state (or the vector it alternates with) changes under a fixed-dt run's feet
    FV_TRY(slot_ptr(p, slot, &d));
    DevBuf<double> du;
    FV_TRY(du.alloc(ctx, (size_t)p->N));
    FV_HIP(ctx, hipMemcpyAsync(du.p, u_nodes, (size_t)p->N * sizeof(double), hipMemcpyDefault, ctx->stream));
    FV_TRY(fv_gather_free(p, du.p, d));
    FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FV_OK;
}

extern "C" int fv_state_set_free(fv_problem *p, int32_t slot, const double *u_free)
{
    if (!p || !u_free)
        return FV_ERR_ARG;
    FV_HIP(p->ctx, hipSetDevice(p->ctx->device));
    double *d;
    p->resume.ok = false; // the state (or the vector it alternates with) changes under a fixed-dt run's feet
    FV_TRY(slot_ptr(p, slot, &d));
    FV_TRY(fv_free_in(p, d, u_free));
    FV_HIP(p->ctx, hipStreamSynchronize(p->ctx->stream));
    return FV_OK;
}

extern "C" int fv_state_get_nodes(fv_problem *p, int32_t slot, double *u_nodes)
{
    if (!p || !u_nodes)
        return FV_ERR_ARG;
    fv_ctx *ctx = p->ctx;
    FV_HIP(ctx, hipSetDevice(ctx->device));
    double *d;
    FV_TRY(slot_ptr(p, slot, &d));
    DevBuf<double> hd;
    FV_TRY(hd.alloc(ctx, (size_t)p->N));
    FV_TRY(fv_scatter_nodes(p, d, hd.p));
    return fv_copy(ctx, u_nodes, hd.p, (size_t)p->N * sizeof(double));
}

extern "C" int fv_state_get_free(fv_problem *p, int32_t slot, double *u_free)
{
    if (!p || !u_free)
        return FV_ERR_ARG;
    FV_HIP(p->ctx, hipSetDevice(p->ctx->device));
    double *d;
    FV_TRY(slot_ptr(p, slot, &d));
    return fv_free_out(p, u_free, d);
}

extern "C" int fv_state_copy(fv_problem *p, int32_t src, int32_t dst)
{
    if (!p)
        return FV_ERR_ARG;
    FV_HIP(p->ctx, hipSetDevice(p->ctx->device));
    double *a, *b;
    p->resume.ok = false; // the state (or the vector it alternates with) changes under a fixed-dt run's feet
    FV_TRY(slot_ptr(p, src, &a));
    FV_TRY(slot_ptr(p, dst, &b));
    if (a != b)
        FV_TRY(fv_copy(p->ctx, b, a, (size_t)p->n * sizeof(double)));
    return FV_OK;
}

extern "C" int fv_state_norm2_diff(fv_problem *p, int32_t a, int32_t b, double *out)
{
    if (!p || !out)
        return FV_ERR_ARG;
    FV_HIP(p->ctx, hipSetDevice(p->ctx->device));
    double *pa, *pb;
    FV_TRY(slot_ptr(p, a, &pa));
    FV_TRY(slot_ptr(p, b, &pb));
    return fv_norm2_diff_device(p, pa, pb, out);
}

// ------------------------------------------------------------------ one implicit step (a11)
// forward:  (D/dt + A) u+ = b' + D u/dt with b' = b (assembled) or D*bhat (bhat = the
//           volume-scaled b of the reference, transient.jl:71); initial guess u.
// adjoint:  the state is g = D w;  (D/dt + A) w+ = bhat + D w/dt, initial guess w = g/D,
//           afterwards g+ = D w+   (transpose(A_scaled) of transient.jl:193).
// Solved in place in the destination slot; no right-hand side vector is formed
// (see pcg_init_kernel<true>).
__global__ __launch_bounds__(FV_BLOCK) void scale_kernel(int64_t n, const double *__restrict__ D, double *__restrict__ x, int divide)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i < n)
        x[i] = divide ? x[i] / D[i] : x[i] * D[i];
}

int g_carry_refresh = 128; // fv_tune key 7: 0 = every step computes its residual with an SpMV (128: drift of the carried residual 4e-13 of the heads over 1 500 steps at 216^3, tools/carry_drift.py; 32 until round 2)
int g_carry_speculate = 1; // fv_tune key 8: the first K2 of a step also prepares the next step (pcg_update_spec_kernel)
int g_chain_steps = 8;     // fv_tune key 13: one-iteration steps enqueued per device poll (< 2: poll every step)
int g_resume_runs = 1;     // fv_tune key 33: a fixed-dt run goes on from the residual / set-up the previous call on the same slot left (fv_problem::resume)

static int step_impl(fv_problem *p, double *usrc, double *udst, double dt, const double *bhat_dev, int mode, double rtol,
                     int64_t maxiter, fv_solve_info *info, bool time_it, bool fold_shift = false, double *x_next = nullptr,
                     const double *carry_prev = nullptr, bool speculate = false, int chain_index = -1, int resume_it = 0,
                     bool chain_more = false, bool defer_flush = false)
{
    fv_ctx *ctx = p->ctx;
    if (!(dt > 0)) {
        fv_set_error(ctx, "time step must be positive"); // transient.jl:68-70
        return FV_ERR_DT;
    }
    PcgSystem sys;
    sys.sigma = 1.0 / dt;
    sys.dt = dt;
    sys.implicit_step = true;
    sys.fold_shift = fold_shift;
    sys.x_next = x_next;
    sys.carry_prev = carry_prev;
    sys.speculate = speculate;              // prepare the next step inside this step's first K2 ...
    sys.use_spec = carry_prev != nullptr;   // ... and start from such a set-up when the residual may be carried
    sys.chain_index = chain_index;
    sys.chain_more = chain_more;
    sys.resume_it = resume_it;
    sys.defer_flush = defer_flush;
    if (usrc != udst && resume_it == 0) {
        // small systems: the single-launch solver reads the state where it is and writes udst (a copy launch less per solve; not in
        // the adjoint's in-place scaling mode, which needs the state in udst first)
        if (mode != FV_STEP_ADJOINT && fv_step_precond(p) != FV_PRECOND_AMG && fv_pcg_small_takes(p, sys))
            sys.x0_src = usrc;
        else
            FV_HIP(ctx, hipMemcpyAsync(udst, usrc, (size_t)p->n * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    }
    if (mode == FV_STEP_FORWARD) {
        sys.rhs = bhat_dev ? bhat_dev : p->b.p;
        sys.b_times_D = bhat_dev != nullptr;
    } else {
        sys.rhs = bhat_dev; // may be null: zero forcing
        if (mode == FV_STEP_ADJOINT) { // (FV_STEP_W: the caller keeps w = g / D as its state — fv_adjoint_run — and no scaling pass is needed)
            hipLaunchKernelGGL(scale_kernel, dim3(fv_blocks(p->n)), dim3(FV_BLOCK), 0, ctx->stream, p->n, p->D.p, udst, 1);
            FV_LAUNCH_CHECK(ctx);
        }
    }
    if (fv_step_precond(p) == FV_PRECOND_AMG) { // the V-cycle path steps in place: no ping-pong, no carried residual
        sys.x_next = nullptr;
        sys.carry_prev = nullptr;
        sys.speculate = sys.use_spec = false;
    }
    FV_TRY(fv_pcg_solve(p, udst, sys, rtol, maxiter, info, time_it));
    if (p->precond == FV_PRECOND_AUTO && !p->auto_steps_amg && info && chain_index < 0 && info->iters > FV_AUTO_SWITCH_ITERS && !p->dist)
        p->auto_steps_amg = true; // this operator wants the V-cycle (it pays from ~30-50 Jacobi iterations per step)
    if (mode == FV_STEP_ADJOINT) {
        hipLaunchKernelGGL(scale_kernel, dim3(fv_blocks(p->n)), dim3(FV_BLOCK), 0, ctx->stream, p->n, p->D.p, udst, 0);
        FV_LAUNCH_CHECK(ctx);
    }
    return FV_OK;
}

extern "C" int fv_transient_step(fv_problem *p, int32_t src, int32_t dst, double dt, const double *bhat_free, int mode, double rtol,
                                 int64_t maxiter, fv_solve_info *info)
{
    if (!p || (mode != FV_STEP_FORWARD && mode != FV_STEP_ADJOINT))
        return FV_ERR_ARG;
    fv_ctx *ctx = p->ctx;
    FV_HIP(ctx, hipSetDevice(ctx->device));
    FV_TRY(need_transient(p, "fv_transient_step"));
    double *a, *b;
    FV_TRY(slot_ptr(p, src, &a));
    FV_TRY(slot_ptr(p, dst, &b));
    DevBuf<double> dbh;
    const double *bh = nullptr;
    if (bhat_free) {
        FV_TRY(dbh.alloc(ctx, (size_t)p->n));
        FV_TRY(fv_free_in(p, dbh.p, bhat_free));
        bh = dbh.p;
    }
    FV_TRY(step_impl(p, a, b, dt, bh, mode, rtol, maxiter, info, true));
    FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FV_OK;
}

extern "C" int fv_transient_run_fixed(fv_problem *p, int32_t slot, double dt, int64_t nsteps, double rtol, int64_t maxiter,
                                      int32_t *iters_per_step, fv_solve_info *last_info, double *total_ms)
{
    if (!p || nsteps < 0)
        return FV_ERR_ARG;
    fv_ctx *ctx = p->ctx;
    FV_HIP(ctx, hipSetDevice(ctx->device));
    FV_TRY(need_transient(p, "fv_transient_run_fixed"));
    double *u;
    FV_TRY(slot_ptr(p, slot, &u));
    hipEvent_t e0, e1;
    FV_HIP(ctx, hipEventCreate(&e0));
    FV_HIP(ctx, hipEventCreate(&e1));
    FV_HIP(ctx, hipEventRecord(e0, ctx->stream));
    fv_solve_info inf = {};
    int rc = FV_OK;
    // Fixed dt, constant b: the step's system differs from the previous one only by sigma D (u_new - u_old) on the
    // right-hand side, so the initial residual follows from the previous step's final residual without an SpMV
    // (PcgSystem::carry_prev).  The state ping-pongs between the caller's slot and a hidden one so that u_old stays
    // readable at no extra traffic; every g_carry_refresh steps the residual is recomputed from scratch (b' - A u), which
    // bounds the drift between the carried recurrence residual and the true one.
    const int64_t refresh = g_carry_refresh;
    const bool pingpong = refresh > 0 && nsteps >= 2 && fv_step_precond(p) != FV_PRECOND_AMG;
    double *alt = nullptr;
    if (pingpong) {
        if (p->pingpong_slot < 0)
            rc = fv_slot_new(p, &p->pingpong_slot);
        if (rc == FV_OK)
            rc = slot_ptr(p, slot, &u); // slot_new may have reallocated the table
        if (rc == FV_OK)
            alt = p->slots[(size_t)p->pingpong_slot];
    }
    const double *prev = nullptr; // state the last solve started from, while p->r holds that solve's final residual
    // go on where the previous call on this slot left off (see fv_problem::resume): its residual, set-up and refresh count
    int64_t s_base = 0;
    {
        const fv_problem::FixedRunResume &rs = p->resume;
        if (g_resume_runs && rs.ok && pingpong && alt && rs.slot == slot && rs.dt == dt && rs.rtol == rtol && rs.assemble_epoch == p->assemble_epoch &&
            rs.storage_epoch == p->storage_epoch && rs.refresh == (int)refresh && rs.speculate == g_carry_speculate &&
            (rs.prev == u || rs.prev == alt)) {
            prev = rs.prev;
            s_base = rs.steps_since_refresh;
        }
        p->resume.ok = false;
    }
    for (int64_t s = 0; s < nsteps && rc == FV_OK; s++) {
        // One-iteration regime: a burst of steps is enqueued without polling the device in between (each step is the
        // prepared set-up + K1 + K2S + K3; a step that does not converge in its iteration stops the chain on the device).
        if (pingpong && g_carry_speculate && g_chain_steps >= 2 && prev != nullptr && p->spec_valid && p->last_iters == 1 && !p->recording) {
            int L = 0;
            while (L < g_chain_steps && L < 32 && s + L < nsteps && ((s_base + s + L) % refresh) != 0)
                L++;
            if (L >= 2) {
                double *snap_u[32], *snap_alt[32];
                for (int j = 0; j < L && rc == FV_OK; j++) {
                    snap_u[j] = u;
                    snap_alt[j] = alt;
                    rc = step_impl(p, u, u, dt, nullptr, FV_STEP_FORWARD, rtol, maxiter, &inf, false, true, alt, prev, true, j, 0, j + 1 < L);
                    prev = u;
                    std::swap(u, alt);
                }
                int completed = 0;
                uint32_t zero_mask = 0; // steps of the burst that were already converged at their set-up
                if (rc == FV_OK)
                    rc = fv_pcg_chain_poll(p, L, &completed, &inf, &zero_mask);
                if (rc != FV_OK)
                    break;
                for (int j = 0; j < completed && j < L; j++)
                    if (iters_per_step)
                        iters_per_step[s + j] = ((zero_mask >> j) & 1u) ? 0 : 1;
                if (completed < L) { // step `completed` needs more iterations: back to its pointers, resume at iteration 1
                    u = snap_u[completed];
                    alt = snap_alt[completed];
                    if ((L - 1 - completed) & 1) { // direction vectors: undo the swaps of the no-op steps behind it
                        p->pvec.swap(p->pnext);
                    }
                    rc = step_impl(p, u, u, dt, nullptr, FV_STEP_FORWARD, rtol, maxiter, &inf, false, true, alt, nullptr, false, -1, 1);
                    if (iters_per_step)
                        iters_per_step[s + completed] = inf.iters;
                    prev = u;
                    if (rc == FV_OK && inf.iters > 0)
                        std::swap(u, alt);
                    s += completed; // + 1 by the loop
                } else
                    s += L - 1;
                continue;
            }
        }
        if (fv_step_precond(p) == FV_PRECOND_AMG) { // FV_PRECOND_AUTO switched over (or AMG was chosen): plain in-place steps
            rc = step_impl(p, u, u, dt, nullptr, FV_STEP_FORWARD, rtol, maxiter, &inf, false, nsteps >= 2);
            if (iters_per_step)
                iters_per_step[s] = inf.iters;
            prev = nullptr;
            if (rc == FV_OK && p->recording) {
                p->record_t += dt;
                rc = fv_trajectory_push_device(p->recording, u, p->record_t, nullptr);
            }
            continue;
        }
        const bool carry = prev != nullptr && ((s_base + s) % refresh) != 0;
        // (the step behind this one is a carried step of this same call: a loop of one-launch iterations may leave its last update to that step's set-up)
        const bool next_carried = pingpong && !p->recording && s + 1 < nsteps && ((s_base + s + 1) % refresh) != 0;
        rc = step_impl(p, u, u, dt, nullptr, FV_STEP_FORWARD, rtol, maxiter, &inf, false, nsteps >= 2, alt, carry ? prev : nullptr,
                       pingpong && g_carry_speculate, -1, 0, false, next_carried); // also on the last step: a 2-step warm-up then runs every kernel of the loop
        if (iters_per_step)
            iters_per_step[s] = inf.iters;
        if (pingpong && rc == FV_OK) {
            if (inf.iters > 0) { // the new state is in alt, u still holds the old one
                prev = u;
                double *t = u;
                u = alt;
                alt = t;
            } else
                prev = u; // converged on entry: state unchanged, zero increment
        }
        if (rc == FV_OK && p->recording) { // fv_trajectory_record: the state of every step stays in HBM (no bursts while recording)
            p->record_t += dt;
            rc = fv_trajectory_push_device(p->recording, u, p->record_t, nullptr);
        }
    }
    if (p->pl_pending.valid) { // (only after a failed step: the last step of a call never leaves its update pending)
        const int rcf = fv_ploop_flush_pending(p);
        if (rc == FV_OK)
            rc = rcf;
    }
    if (pingpong && alt) { // hand the buffers back: the caller's slot owns the current state — also after a failed step, where
        // u is the last state a step completed from (the failed step wrote, if anything, into alt)
        p->slots[(size_t)slot] = u;
        p->slots[(size_t)p->pingpong_slot] = alt;
        if (rc == FV_OK && prev != nullptr && fv_step_precond(p) != FV_PRECOND_AMG) { // the next call may go on from here
            fv_problem::FixedRunResume &rs = p->resume;
            rs.ok = true;
            rs.slot = slot;
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
        hipError_t e = hipEventRecord(e1, ctx->stream);
        if (e == hipSuccess)
            e = hipEventSynchronize(e1);
        float ms = 0.f;
        if (e == hipSuccess)
            e = hipEventElapsedTime(&ms, e0, e1);
        if (e != hipSuccess) {
            fv_set_error(ctx, "event timing failed: %s", hipGetErrorString(e));
            rc = FV_ERR_HIP;
        }
        if (total_ms)
            *total_ms = ms;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (last_info)
        *last_info = inf;
    if (nsteps >= 8)
        fv_vec_release_spares(p); // (every vector of the loop exists by now: the candidates nobody asked for go back — fv_place.hip)
    return rc;
}

// ------------------------------------------------------------------ the adaptive stepper on the device
// transient.jl:78-121,136-154 with constant b: step-doubling error control (one step of dt against two of dt/2, grow
// x2 when the difference is below atol/4, halve on failure and re-use the half step as the next trial's full step,
// sub-step without overshoot), everything resident; the host only sees scalars.  The state in `slot` is advanced to
// tfinal; ts_out receives the outer step times (ts[0] = t0), as the reference's `ts`.
namespace {
struct TwoStep {
    double *result; // twostep (accepted) or twostep1 (rejected)
    double last;    // dt taken: dt or dt/2
    bool increase;
};
} // namespace

// One solve of the stepper: the hooks' forcing at the solve's start time (transient.jl:60-62: getb(t)), the hooks' step mode.
static int hooked_step(fv_problem *p, const FvStepHooks &h, double *src, double *dst, double t, double dt, double rtol, int64_t maxiter, fv_solve_info *inf)
{
    const double *rhs = nullptr;
    if (h.forcing)
        FV_TRY(h.forcing(t, 0, &rhs));
    return fv_step_raw(p, src, dst, dt, rhs, h.mode, rtol, maxiter, inf);
}

static int adaptive_twostep(fv_problem *p, const FvStepHooks &h, double *uk, double t, double dt, double *onestep, bool have_onestep, double *two1, double *two,
                            double atol, double rtol, int64_t maxiter, fv_solve_info *inf, int64_t *nsolves, TwoStep *out)
{
    double err = 0.0;
    bool fused = false;
    if (fv_small_twostep_takes(p, h.mode, dt)) { // small systems: the three solves and the norm in one launch (fv_small.hip)
        const double *rhs[3] = {nullptr, nullptr, nullptr};
        const double tk[3] = {t, t, t + 0.5 * dt};
        const int k0 = have_onestep ? 1 : 0;
        int given = 0;
        for (int k = k0; k < 3; k++) {
            if (h.forcing)
                FV_TRY(h.forcing(tk[k], k, &rhs[k]));
            given += rhs[k] != nullptr;
        }
        // forward steps (step_impl): a given forcing is b' / D, none means the assembled b — one launch takes one convention
        const bool scaled = h.mode == FV_STEP_FORWARD && given > 0;
        if (!(scaled && given != 3 - k0)) {
            if (h.mode == FV_STEP_FORWARD && !scaled)
                for (int k = k0; k < 3; k++)
                    rhs[k] = p->b.p;
            FV_TRY(fv_small_twostep(p, h.mode, rhs, scaled, uk, dt, onestep, have_onestep, two1, two, h.norm_weight, rtol, maxiter, inf, &err, &fused));
        }
    }
    if (fused)
        *nsolves += have_onestep ? 2 : 3;
    else {
        if (!have_onestep) {
            FV_TRY(hooked_step(p, h, uk, onestep, t, dt, rtol, maxiter, inf));
            ++*nsolves;
        }
        FV_TRY(hooked_step(p, h, uk, two1, t, 0.5 * dt, rtol, maxiter, inf));
        FV_TRY(hooked_step(p, h, two1, two, t + 0.5 * dt, 0.5 * dt, rtol, maxiter, inf));
        *nsolves += 2;
        if (h.norm_weight)
            FV_TRY(fv_norm2_diff_weighted_device(p, onestep, two, h.norm_weight, &err)); // the state the caller sees is weight .* vector
        else
            FV_TRY(fv_norm2_diff_device(p, onestep, two, &err)); // norm(onestep - twostep), transient.jl:81
    }
    if (err < atol) {
        out->result = two;
        out->last = dt;
        out->increase = err < atol / 4;
    } else {
        out->result = two1;
        out->last = 0.5 * dt;
        out->increase = false;
    }
    return FV_OK;
}

int fv_step_raw(fv_problem *p, double *usrc, double *udst, double dt, const double *rhs_dev, int mode, double rtol, int64_t maxiter, fv_solve_info *info)
{
    return step_impl(p, usrc, udst, dt, rhs_dev, mode, rtol, maxiter, info, false);
}

// The loop itself, for the forward run (fv_transient_run_adaptive) and the device-resident adjoint sweep (fv_adjoint_run, fv_trajectory.hip):
// the hooks give the step mode, the forcing of a solve by its start time, the weight of the error norm and a recorder that sees
// the state of every outer step (the reference's `us`).  fixed: the fixed stepper (transient.jl:130-134) in the same outer loop.
int fv_stepper_run(fv_problem *p, int32_t slot, double t0, double tfinal, double dt0, bool fixed, double atol, double rtol, int64_t maxiter, int64_t max_outer,
                   double *ts_out, int64_t *n_outer, int64_t *n_solves, fv_solve_info *last_info, const FvStepHooks &h)
{
    fv_ctx *ctx = p->ctx;
    if (!(dt0 > 0)) {
        fv_set_error(ctx, "time step must be positive");
        return FV_ERR_DT;
    }
    int32_t scratch[4] = {-1, -1, -1, -1};
    int rc = FV_OK;
    for (int i = 0; i < 4 && rc == FV_OK; i++)
        rc = fv_slot_new(p, &scratch[i]);
    double *U = nullptr;
    if (rc == FV_OK)
        rc = slot_ptr(p, slot, &U); // after the allocations: the slot table may have moved
    fv_solve_info inf = {};
    int64_t nout = 0, nsolves = 0;
    if (rc == FV_OK) {
        double *E = p->slots[(size_t)scratch[0]], *S1 = p->slots[(size_t)scratch[1]], *S2 = p->slots[(size_t)scratch[2]],
               *S3 = p->slots[(size_t)scratch[3]];
        const size_t bytes = (size_t)p->n * sizeof(double);
        double *const U0 = U; // the caller's slot
        double t = t0;
        double dt = dt0 < tfinal - t0 ? dt0 : tfinal - t0;
        if (ts_out && max_outer > 0)
            ts_out[0] = t0;
        if (h.record)
            rc = h.record(U, t0);
        while (rc == FV_OK && t < tfinal && nout < max_outer) {
            TwoStep ts{};
            if (fixed) { // fixedbackwardeulerstep!: one solve, laststeptime = dt, never grows
                rc = hooked_step(p, h, U, S1, t, dt, rtol, maxiter, &inf);
                nsolves++;
                ts.result = S1;
                ts.last = dt;
                ts.increase = false;
            } else
                rc = adaptive_twostep(p, h, U, t, dt, S1, false, S2, S3, atol, rtol, maxiter, &inf, &nsolves, &ts);
            if (rc != FV_OK)
                break;
            const double *unew = ts.result;
            if (ts.last < dt) { // rejected: sub-step to t + dt, transient.jl:93-120
                bool failed = true;
                double elapsed = 0.0, target = ts.last;
                if (hipMemcpyAsync(E, U, bytes, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) {
                    rc = FV_ERR_HIP;
                    break;
                }
                std::swap(S1, S2); // the half step just computed is the next trial's full step
                while (rc == FV_OK && elapsed < dt) {
                    rc = adaptive_twostep(p, h, E, t + elapsed, target, S1, failed, S2, S3, atol, rtol, maxiter, &inf, &nsolves, &ts);
                    if (rc != FV_OK)
                        break;
                    if (ts.last == target) {
                        elapsed += ts.last;
                        if (hipMemcpyAsync(E, ts.result, bytes, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) {
                            rc = FV_ERR_HIP;
                            break;
                        }
                        if (ts.increase)
                            target = 2 * ts.last;
                        failed = false;
                    } else if (ts.last < target) {
                        target = ts.last;
                        failed = true;
                        std::swap(S1, S2);
                    } else {
                        fv_set_error(ctx, "Code is broken -- laststeptime should never be greater than targetdt"); // transient.jl:115
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
            // the accepted state becomes U by an exchange of buffers — the old state's buffer is scratch from here on — instead of a
            // copy per outer step; the caller's slot receives the final state once, below
            if (unew == S1)
                std::swap(U, S1);
            else if (unew == S2)
                std::swap(U, S2);
            else if (unew == S3)
                std::swap(U, S3);
            else if (unew == E)
                std::swap(U, E);
            else if (hipMemcpyAsync(U, unew, bytes, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) {
                rc = FV_ERR_HIP;
                break;
            }
            t += dt;
            nout++;
            if (ts_out)
                ts_out[nout] = t;
            if (h.record) {
                rc = h.record(U, t);
                if (rc != FV_OK)
                    break;
            }
            const double remaining = tfinal - t;
            const double want = ts.increase ? 2 * ts.last : ts.last;
            dt = remaining < want ? remaining : want;
        }
        if (U != U0 && rc != FV_ERR_HIP && hipMemcpyAsync(U0, U, bytes, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) // (also after an error: the slot holds u(t))
            rc = FV_ERR_HIP;
        if (rc == FV_OK && hipStreamSynchronize(ctx->stream) != hipSuccess)
            rc = FV_ERR_HIP;
        if (rc == FV_ERR_HIP)
            fv_set_error(ctx, "fv_stepper_run: device copy failed: %s", hipGetErrorString(hipGetLastError()));
        if (rc == FV_OK && t < tfinal) { // the reference always reaches tfinal (transient.jl:143-152): never hand u(t) back as u(tfinal)
            fv_set_error(ctx, "adaptive stepper (fv_transient_run_adaptive / fv_adjoint_run): %lld outer steps (max_outer) taken and t = %.17g < tfinal = %.17g; the state is u(t)",
                         (long long)nout, t, tfinal);
            rc = FV_ERR_STATE;
        }
    }
    for (int i = 0; i < 4; i++)
        if (scratch[i] >= 0)
            p->slot_used[(size_t)scratch[i]] = 0;
    if (n_outer)
        *n_outer = nout;
    if (n_solves)
        *n_solves = nsolves;
    if (last_info)
        *last_info = inf;
    return rc;
}

extern "C" int fv_transient_run_adaptive(fv_problem *p, int32_t slot, double t0, double tfinal, double dt0, double atol, double rtol,
                                         int64_t maxiter, int64_t max_outer, double *ts_out, int64_t *n_outer, int64_t *n_solves,
                                         fv_solve_info *last_info)
{
    if (!p || !(tfinal >= t0) || max_outer < 0 || (max_outer > 0 && !ts_out))
        return FV_ERR_ARG;
    fv_ctx *ctx = p->ctx;
    FV_HIP(ctx, hipSetDevice(ctx->device));
    FV_TRY(need_transient(p, "fv_transient_run_adaptive"));
    FvStepHooks h;
    if (p->recording) // fv_trajectory_record: every outer state stays in HBM (the reference's `us`)
        h.record = [p](const double *state, double t) { return fv_trajectory_push_device(p->recording, state, t, nullptr); };
    return fv_stepper_run(p, slot, t0, tfinal, dt0, false, atol, rtol, maxiter, max_outer, ts_out, n_outer, n_solves, last_info, h);
}

// ------------------------------------------------------------------ kernel-level entry points
extern "C" int fv_spmv(fv_problem *p, const double *x_free, double sigma, double *y_free)
{
    if (!p || !x_free || !y_free)
        return FV_ERR_ARG;
    fv_ctx *ctx = p->ctx;
    FV_HIP(ctx, hipSetDevice(ctx->device));
    FV_TRY(need_assembled(p, "fv_spmv"));
    if (sigma != 0.0)
        FV_TRY(need_transient(p, "fv_spmv with sigma != 0"));
    FV_TRY(fv_pcg_prepare(p));
    FV_TRY(fv_free_in(p, p->tmp.p, x_free));
    FV_TRY(fv_spmv_launch(p, p->tmp.p, p->rhs.p, sigma, nullptr));
    return fv_free_out(p, y_free, p->rhs.p);
}

extern "C" int fv_bench_spmv(fv_problem *p, double sigma, int32_t reps, double *avg_ms)
{
    if (!p || reps <= 0 || !avg_ms)
        return FV_ERR_ARG;
    fv_ctx *ctx = p->ctx;
    FV_HIP(ctx, hipSetDevice(ctx->device));
    FV_TRY(need_assembled(p, "fv_bench_spmv"));
    if (sigma != 0.0)
        FV_TRY(need_transient(p, "fv_bench_spmv with sigma != 0"));
    FV_TRY(fv_pcg_prepare(p));
    // the PCG's own kernel: SpMV with the p.q epilogue, on the resident search direction (a row block: its interior and
    // boundary passes without the exchange)
    auto once = [&]() -> int {
        if (p->dist)
            return fv_dist_local_spmv(p, p->pvec.p, p->q.p, sigma, true, true);
        return fv_spmv_launch(p, p->pvec.p, p->q.p, sigma, p->part_pq.p, true);
    };
    FV_TRY(once()); // warm (and fold the shift if enabled)
    FV_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    for (int32_t i = 0; i < reps; i++)
        FV_TRY(once());
    FV_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    FV_HIP(ctx, hipEventSynchronize(ctx->ev1));
    float ms = 0.f;
    FV_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    *avg_ms = (double)ms / reps;
    return FV_OK;
}

extern "C" int fv_dot(fv_problem *p, const double *a_free, const double *b_free, double *out)
{
    if (!p || !a_free || !b_free || !out)
        return FV_ERR_ARG;
    fv_ctx *ctx = p->ctx;
    FV_HIP(ctx, hipSetDevice(ctx->device));
    FV_TRY(fv_pcg_prepare(p));
    // (both vectors in the caller's numbering: a dot product does not care which)
    FV_HIP(ctx, hipMemcpyAsync(p->tmp.p, a_free, (size_t)p->n * sizeof(double), hipMemcpyDefault, ctx->stream));
    FV_HIP(ctx, hipMemcpyAsync(p->rhs.p, b_free, (size_t)p->n * sizeof(double), hipMemcpyDefault, ctx->stream));
    return fv_dot_device(p, p->tmp.p, p->rhs.p, out);
}

// Per-kernel timing of the PCG loop with HIP events on the launch stream.
extern "C" int fv_profile_enable(fv_problem *p, int on)
{
    if (!p)
        return FV_ERR_ARG;
    p->profile = on != 0;
    p->profile_level = on == 2 ? 2 : 1; // 2: the SpMV only
    for (int c = 0; c < 3; c++) {
        p->prof_ms[c] = 0;
        p->prof_launches[c] = 0;
    }
    return FV_OK;
}

// kernel 0: spmv_dot (K1), 1: update (K2), 2: pupdate (K3)
extern "C" int fv_profile_get(fv_problem *p, int kernel, double *total_ms, int64_t *launches)
{
    if (!p || kernel < 0 || kernel > 2)
        return FV_ERR_ARG;
    if (total_ms)
        *total_ms = p->prof_ms[kernel];
    if (launches)
        *launches = p->prof_launches[kernel];
    return FV_OK;
}

FV_WARM_TU(transient) // (fv_ctx_create loads every code object of the library up front: fv_warm_modules, fv_ctx.hip)
