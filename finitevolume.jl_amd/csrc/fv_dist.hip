// Row-block partition of an assembled problem across the ranks of the context's
// RCCL communicator (one process per GPU).  No reference counterpart: the
// reference is single-process; the plan mirrors
// finitevolume.jl_amd/partition.py, against which the GPU tests check it.
//
// Every rank builds the global structure (deterministic, so all ranks agree),
// then keeps only its contiguous range of free rows as a LOCAL problem whose
// columns are renumbered [local | halo] (halo = sorted remote columns) and whose
// vectors carry nhalo extra slots.  Per SpMV a rank sends each peer the rows that
// peer references, ascending, which is exactly the receiver's halo order.
#include "fv_internal.h"

__global__ __launch_bounds__(FV_BLOCK) void dist_mark_remote_kernel(int64_t e0, int64_t e1, const int32_t *__restrict__ colind, int32_t lo,
                                                                     int32_t hi, int32_t *__restrict__ flag)
{
    const int64_t k = e0 + (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (k >= e1)
        return;
    const int32_t c = colind[k];
    if (c < lo || c >= hi)
        flag[c] = 1;
}

// flags (over my local rows) of the rows that the entries [e0,e1) of a peer's row block reference
__global__ __launch_bounds__(FV_BLOCK) void dist_mark_wanted_kernel(int64_t e0, int64_t e1, const int32_t *__restrict__ colind, int32_t lo,
                                                                     int32_t hi, int32_t *__restrict__ flag)
{
    const int64_t k = e0 + (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (k >= e1)
        return;
    const int32_t c = colind[k];
    if (c >= lo && c < hi)
        flag[c - lo] = 1;
}

__global__ __launch_bounds__(FV_BLOCK) void dist_compact_kernel(int64_t n, const int32_t *__restrict__ flag, const int32_t *__restrict__ scan,
                                                                 int32_t *__restrict__ out, int32_t out_off)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i < n && flag[i])
        out[out_off + scan[i]] = (int32_t)i;
}

__global__ __launch_bounds__(FV_BLOCK) void dist_local_cols_kernel(int64_t nnz_loc, const int32_t *__restrict__ gcol, int32_t lo, int32_t hi,
                                                                    int32_t nloc, const int32_t *__restrict__ haloscan,
                                                                    int32_t *__restrict__ lcol)
{
    const int64_t k = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (k >= nnz_loc)
        return;
    const int32_t c = gcol[k];
    lcol[k] = (c >= lo && c < hi) ? c - lo : nloc + haloscan[c];
}

__global__ __launch_bounds__(FV_BLOCK) void dist_local_rowptr_kernel(int64_t nloc, const int32_t *__restrict__ grp, int32_t e0,
                                                                      int32_t *__restrict__ lrp)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i <= nloc)
        lrp[i] = grp[i] - e0;
}

__global__ __launch_bounds__(FV_BLOCK) void dist_local_diagpos_kernel(int64_t nloc, const int32_t *__restrict__ gdp, int32_t e0,
                                                                       int32_t *__restrict__ ldp)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i < nloc)
        ldp[i] = gdp[i] >= 0 ? gdp[i] - e0 : -1;
}

// a 64-row group is a boundary group when any of its rows references a halo slot
__global__ __launch_bounds__(FV_BLOCK) void dist_group_flags_kernel(int64_t nloc, const int32_t *__restrict__ lrp, const int32_t *__restrict__ lcol,
                                                                     int32_t *__restrict__ bnd, int32_t *__restrict__ inter)
{
    const int64_t g = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    const int64_t ngroups = (nloc + 63) >> 6;
    if (g >= ngroups)
        return;
    const int64_t r0 = g << 6;
    const int64_t r1 = (r0 + 64 < nloc) ? r0 + 64 : nloc;
    int32_t any = 0;
    for (int32_t k = lrp[r0]; k < lrp[r1]; k++)
        if (lcol[k] >= (int32_t)nloc) {
            any = 1;
            break;
        }
    bnd[g] = any;
    inter[g] = !any;
}

__global__ __launch_bounds__(FV_BLOCK) void iota32_kernel(int32_t *p, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i < n)
        p[i] = (int32_t)i;
}

// flags -> ascending index list appended at out[out_off...]; returns the count
static int compact_flags(fv_ctx *ctx, const int32_t *flag, int64_t n, int32_t *out, int64_t out_off, int64_t *count)
{
    DevBuf<int32_t> scan;
    FV_TRY(scan.alloc(ctx, (size_t)n + 1));
    FV_TRY(fv_exclusive_scan_i32(ctx, flag, scan.p, n, count));
    if (*count > 0 && out) {
        hipLaunchKernelGGL(dist_compact_kernel, dim3(fv_blocks(n)), dim3(FV_BLOCK), 0, ctx->stream, n, flag, scan.p, out, (int32_t)out_off);
        FV_LAUNCH_CHECK(ctx);
        FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return FV_OK;
}

static int copy_slice(fv_ctx *ctx, DevBuf<double> &dst, const double *src, int64_t off, int64_t cnt, int64_t pad)
{
    FV_TRY(dst.alloc(ctx, (size_t)(cnt + pad)));
    FV_TRY(dst.zero(ctx));
    if (cnt > 0)
        FV_HIP(ctx, hipMemcpyAsync(dst.p, src + off, (size_t)cnt * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    return FV_OK;
}

static int dist_setup_impl(fv_problem *pg, int nranks, int rank, const int64_t *bounds, fv_problem **out);

// Row blocks are ranges of the CALLER's free-cell numbering (rank among the free nodes).  A problem whose free cells were
// re-numbered for locality at creation (fv_reorder_free) keeps another numbering inside: for the time of the set-up its
// per-row arrays are replaced by copies in the canonical order — row i = internal row perm[i], columns mapped back through
// iperm and sorted, as fv_get_csc exports them — so a block cut from it equals the block cut from the un-numbered problem.
__global__ __launch_bounds__(FV_BLOCK) void canon_len32_kernel(int64_t n, const int32_t *__restrict__ perm, const int32_t *__restrict__ rowptr,
                                                                int32_t *__restrict__ len)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i < n)
        len[i] = rowptr[perm[i] + 1] - rowptr[perm[i]];
}
__global__ __launch_bounds__(FV_BLOCK) void canon_rows32_kernel(int64_t n, const int32_t *__restrict__ perm, const int32_t *__restrict__ iperm,
                                                                 const int32_t *__restrict__ rowptr, const int32_t *__restrict__ colind,
                                                                 const double *__restrict__ vals, const int32_t *__restrict__ start,
                                                                 int32_t *__restrict__ colind_out, double *__restrict__ vals_out,
                                                                 int32_t *__restrict__ diagpos_out)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i >= n)
        return;
    const int32_t r = perm[i], s0 = rowptr[r], len = rowptr[r + 1] - s0, o = start[i];
    int32_t dp = -1;
    for (int32_t k = 0; k < len; k++) { // insertion sort by canonical column (rows are short)
        const int32_t c = iperm[colind[s0 + k]];
        const double v = vals[s0 + k];
        int32_t j = k;
        while (j > 0 && colind_out[o + j - 1] > c) {
            colind_out[o + j] = colind_out[o + j - 1];
            vals_out[o + j] = vals_out[o + j - 1];
            j--;
        }
        colind_out[o + j] = c;
        vals_out[o + j] = v;
    }
    for (int32_t k = 0; k < len; k++)
        if (colind_out[o + k] == (int32_t)i)
            dp = o + k;
    diagpos_out[i] = dp;
}
__global__ __launch_bounds__(FV_BLOCK) void canon_vec_kernel(int64_t n, const int32_t *__restrict__ perm, const double *__restrict__ src,
                                                              double *__restrict__ dst)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i < n)
        dst[i] = src[perm[i]];
}

static int dist_setup_canonical(fv_problem *pg, int nranks, int rank, const int64_t *bounds, fv_problem **out)
{
    fv_ctx *ctx = pg->ctx;
    const int64_t n = pg->n, nnz = pg->nnz;
    DevBuf<int32_t> len, rowptr, colind, diagpos;
    DevBuf<double> vals, b, diagA, D, state;
    FV_TRY(len.alloc(ctx, (size_t)n));
    FV_TRY(rowptr.alloc(ctx, (size_t)n + 1));
    FV_TRY(colind.alloc(ctx, (size_t)nnz + 2));
    FV_TRY(colind.zero(ctx));
    FV_TRY(vals.alloc(ctx, (size_t)nnz + 2));
    FV_TRY(vals.zero(ctx));
    FV_TRY(diagpos.alloc(ctx, (size_t)n));
    hipLaunchKernelGGL(canon_len32_kernel, dim3(fv_blocks(n)), dim3(FV_BLOCK), 0, ctx->stream, n, (const int32_t *)pg->perm.p, (const int32_t *)pg->rowptr.p, len.p);
    FV_LAUNCH_CHECK(ctx);
    int64_t total = 0;
    FV_TRY(fv_exclusive_scan_i32(ctx, len.p, rowptr.p, n, &total));
    if (total != nnz) {
        fv_set_error(ctx, "internal: canonical view has %lld of %lld entries", (long long)total, (long long)nnz);
        return FV_ERR_STATE;
    }
    hipLaunchKernelGGL(canon_rows32_kernel, dim3(fv_blocks(n)), dim3(FV_BLOCK), 0, ctx->stream, n, (const int32_t *)pg->perm.p, (const int32_t *)pg->iperm.p,
                       (const int32_t *)pg->rowptr.p, (const int32_t *)pg->colind.p, (const double *)pg->vals.p, (const int32_t *)rowptr.p, colind.p, vals.p,
                       diagpos.p);
    FV_LAUNCH_CHECK(ctx);
    const size_t vn = (size_t)n + FV_VEC_PAD;
    for (auto pr : {std::make_pair(&b, pg->b.p), std::make_pair(&diagA, pg->diagA.p), std::make_pair(&D, pg->D.p), std::make_pair(&state, pg->slots[0])}) {
        FV_TRY(pr.first->alloc(ctx, vn));
        FV_TRY(pr.first->zero(ctx));
        hipLaunchKernelGGL(canon_vec_kernel, dim3(fv_blocks(n)), dim3(FV_BLOCK), 0, ctx->stream, n, (const int32_t *)pg->perm.p, (const double *)pr.second,
                           pr.first->p);
        FV_LAUNCH_CHECK(ctx);
    }
    FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    auto exchange = [&]() {
        pg->rowptr.swap(rowptr);
        pg->colind.swap(colind);
        pg->vals.swap(vals);
        pg->diagpos.swap(diagpos);
        pg->b.swap(b);
        pg->diagA.swap(diagA);
        pg->D.swap(D);
        std::swap(pg->slots[0], state.p);
    };
    // dist_setup_impl reads exactly these fields of the global problem: n, Ss, ns / slab_lo / slab_hi (slab problems are never
    // re-numbered), rowptr, colind, vals, diagpos, b, diagA, D and slots[0] — the eight arrays swapped here — and the flag
    // `reordered`.  None of the global problem's derived copies (lane-major / symmetric copies, storage and matrix codes, M^-1,
    // AMG hierarchy: all in the internal numbering) is consulted, and while the canonical arrays stand in, the epochs that key
    // those copies are parked on values no copy carries, so that any future read of one by this routine rebuilds instead of
    // silently mixing numberings.
    const int64_t assemble_epoch = pg->assemble_epoch, storage_epoch = pg->storage_epoch;
    pg->assemble_epoch = -7;
    pg->storage_epoch = -7;
    exchange();
    pg->reordered = false;
    const int rc = dist_setup_impl(pg, nranks, rank, bounds, out);
    pg->reordered = true;
    exchange();
    pg->assemble_epoch = assemble_epoch;
    pg->storage_epoch = storage_epoch;
    FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return rc;
}

extern "C" int fv_dist_setup(fv_problem *pg, int nranks, int rank, fv_problem **out)
{
    return dist_setup_impl(pg, nranks, rank, nullptr, out);
}

// The same with the caller's row ranges: rank r owns the free rows [bounds[r], bounds[r+1]) (bounds[0] = 0,
// bounds[nranks] = n, non-decreasing; every rank must pass the same array).  With a slab problem
// (fv_problem_create_regulargrid_slab) the rank's range must lie inside the slab's planes.
extern "C" int fv_dist_setup_bounds(fv_problem *pg, int nranks, int rank, const int64_t *bounds, fv_problem **out)
{
    if (!bounds)
        return FV_ERR_ARG;
    return dist_setup_impl(pg, nranks, rank, bounds, out);
}

static int dist_setup_impl(fv_problem *pg, int nranks, int rank, const int64_t *bounds, fv_problem **out)
{
    if (!pg || !out || nranks < 1 || rank < 0 || rank >= nranks)
        return FV_ERR_ARG;
    *out = nullptr;
    fv_ctx *ctx = pg->ctx;
    FV_HIP(ctx, hipSetDevice(ctx->device));
    if (!pg->assembled || !pg->transient_ready) {
        fv_set_error(ctx, "fv_dist_setup: call fv_assemble and fv_transient_begin on the global problem first");
        return FV_ERR_STATE;
    }
    // a lean global problem (FV_OPT_LEAN_SETUP) lends itself a CSR for the duration of this call: the block's rows are cut out of it, then it goes back
    struct CsrLoan {
        fv_problem *p;
        bool on = false;
        ~CsrLoan()
        {
            if (on) {
                p->rowptr.release();
                p->colind.release();
                p->vals.release();
                p->diagpos.release();
            }
        }
    } loan{pg};
    if (pg->lean) {
        loan.on = true;
        FV_TRY(fv_lean_csr32(pg, pg->rowptr, pg->colind, pg->vals, &pg->diagpos));
    }
    if (pg->dist || pg->nhalo) {
        fv_set_error(ctx, "fv_dist_setup: problem is already a row block");
        return FV_ERR_STATE;
    }
    if (pg->reordered) // row ranges are ranges of the caller's free-cell numbering: cut the block from a canonical view
        return dist_setup_canonical(pg, nranks, rank, bounds, out);
    const int64_t n = pg->n;
    fv_dist *d = new fv_dist();
    d->nranks = nranks;
    d->rank = rank;
    d->bounds.resize((size_t)nranks + 1);
    for (int r = 0; r <= nranks; r++)
        d->bounds[(size_t)r] = bounds ? bounds[r] : ((int64_t)r * n) / nranks; // default: partition.row_ranges
    bool bounds_ok = d->bounds[0] == 0 && d->bounds[(size_t)nranks] == n;
    for (int r = 0; r < nranks; r++)
        bounds_ok = bounds_ok && d->bounds[(size_t)r] <= d->bounds[(size_t)r + 1];
    if (!bounds_ok) {
        fv_set_error(ctx, "fv_dist_setup_bounds: bounds must run from 0 to n = %lld without decreasing", (long long)n);
        delete d;
        return FV_ERR_ARG;
    }
    if (pg->slab_lo >= 0) { // a slab problem only has the rows of its planes
        int64_t first = 0, last = 0;
        const int64_t plane = pg->ns[1] * pg->ns[2];
        int rc0 = fv_problem_free_rows_before(pg, pg->slab_lo * plane, &first);
        if (rc0 == FV_OK)
            rc0 = fv_problem_free_rows_before(pg, pg->slab_hi * plane, &last);
        if (rc0 != FV_OK || d->bounds[(size_t)rank] < first || d->bounds[(size_t)rank + 1] > last) {
            fv_set_error(ctx, "fv_dist_setup_bounds: rank %d's rows [%lld, %lld) are not inside the slab's rows [%lld, %lld)", rank,
                         (long long)d->bounds[(size_t)rank], (long long)d->bounds[(size_t)rank + 1], (long long)first, (long long)last);
            delete d;
            return rc0 != FV_OK ? rc0 : FV_ERR_ARG;
        }
    }
    const int64_t lo = d->bounds[(size_t)rank], hi = d->bounds[(size_t)rank + 1];
    const int64_t nloc = hi - lo;
    d->lo = lo;
    d->hi = hi;
    // entry offsets of every rank's row block
    std::vector<int32_t> eb((size_t)nranks + 1);
    for (int r = 0; r <= nranks; r++)
        FV_HIP(ctx, fv_memcpy_sync(ctx, &eb[(size_t)r], pg->rowptr.p + d->bounds[(size_t)r], sizeof(int32_t), hipMemcpyDeviceToHost));
    const int64_t e0 = eb[(size_t)rank], e1 = eb[(size_t)rank + 1];
    const int64_t nnz_loc = e1 - e0;
    d->entry_lo = e0;

    fv_problem *pl = new fv_problem();
    pl->ctx = ctx;
    pl->dist = d;
    pl->N = pl->n = nloc;
    pl->nnz = nnz_loc;
    pl->from_csc = true;
    // (the traversal order stays the interior / boundary lists; build_group_order only looks for the plane stride of a row block)
    int rc = FV_OK;
    do {
        // ---- halo: sorted remote columns referenced by my rows
        DevBuf<int32_t> flag, hscan;
        if ((rc = flag.alloc(ctx, (size_t)n)) || (rc = flag.zero(ctx)))
            break;
        if (nnz_loc > 0) {
            hipLaunchKernelGGL(dist_mark_remote_kernel, dim3(fv_blocks(nnz_loc)), dim3(FV_BLOCK), 0, ctx->stream, e0, e1, pg->colind.p,
                               (int32_t)lo, (int32_t)hi, flag.p);
        }
        if ((rc = hscan.alloc(ctx, (size_t)n + 1)))
            break;
        int64_t nhalo = 0;
        if ((rc = fv_exclusive_scan_i32(ctx, flag.p, hscan.p, n, &nhalo)))
            break;
        d->nhalo = nhalo;
        pl->nhalo = nhalo;
        if ((rc = d->halo_cols.alloc(ctx, (size_t)nhalo)))
            break;
        if (nhalo > 0) {
            hipLaunchKernelGGL(dist_compact_kernel, dim3(fv_blocks(n)), dim3(FV_BLOCK), 0, ctx->stream, n, flag.p, hscan.p, d->halo_cols.p, 0);
        }
        // halo slots per owner: contiguous because halo_cols is ascending
        d->recv_counts.assign((size_t)nranks, 0);
        {
            std::vector<int32_t> hb((size_t)nranks + 1);
            bool bad = false;
            for (int r = 0; r <= nranks && !bad; r++)
                bad = fv_memcpy_sync(ctx, &hb[(size_t)r], hscan.p + d->bounds[(size_t)r], sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess;
            if (bad) {
                fv_set_error(ctx, "fv_dist_setup: read-back of halo offsets failed");
                rc = FV_ERR_HIP;
                break;
            }
            for (int r = 0; r < nranks; r++)
                d->recv_counts[(size_t)r] = hb[(size_t)r + 1] - hb[(size_t)r];
        }
        // ---- local CSR
        if ((rc = pl->rowptr.alloc(ctx, (size_t)nloc + 1)) || (rc = pl->colind.alloc(ctx, (size_t)nnz_loc + 2)) ||
            (rc = pl->colind.zero(ctx)) || (rc = pl->diagpos.alloc(ctx, (size_t)nloc)))
            break;
        hipLaunchKernelGGL(dist_local_rowptr_kernel, dim3(fv_blocks(nloc + 1)), dim3(FV_BLOCK), 0, ctx->stream, nloc, pg->rowptr.p + lo,
                           (int32_t)e0, pl->rowptr.p);
        if (nnz_loc > 0)
            hipLaunchKernelGGL(dist_local_cols_kernel, dim3(fv_blocks(nnz_loc)), dim3(FV_BLOCK), 0, ctx->stream, nnz_loc, pg->colind.p + e0,
                               (int32_t)lo, (int32_t)hi, (int32_t)nloc, hscan.p, pl->colind.p);
        hipLaunchKernelGGL(dist_local_diagpos_kernel, dim3(fv_blocks(nloc)), dim3(FV_BLOCK), 0, ctx->stream, nloc, pg->diagpos.p + lo,
                           (int32_t)e0, pl->diagpos.p);
        if ((rc = copy_slice(ctx, pl->vals, pg->vals.p, e0, nnz_loc, 2)) || (rc = copy_slice(ctx, pl->b, pg->b.p, lo, nloc, 0)) ||
            (rc = copy_slice(ctx, pl->diagA, pg->diagA.p, lo, nloc, 0)) || (rc = copy_slice(ctx, pl->D, pg->D.p, lo, nloc, 2)))
            break;
        if ((rc = pl->nodemap.alloc(ctx, (size_t)nloc)) || (rc = pl->f2n.alloc(ctx, (size_t)nloc)) || (rc = pl->dheads.alloc(ctx, 1)))
            break;
        hipLaunchKernelGGL(iota32_kernel, dim3(fv_blocks(nloc)), dim3(FV_BLOCK), 0, ctx->stream, pl->nodemap.p, nloc);
        hipLaunchKernelGGL(iota32_kernel, dim3(fv_blocks(nloc)), dim3(FV_BLOCK), 0, ctx->stream, pl->f2n.p, nloc);
        if (hipStreamSynchronize(ctx->stream) != hipSuccess || hipGetLastError() != hipSuccess) {
            fv_set_error(ctx, "fv_dist_setup: local block kernels failed");
            rc = FV_ERR_HIP;
            break;
        }
        // ---- send lists: my rows referenced by each peer's row block, ascending
        d->send_counts.assign((size_t)nranks, 0);
        DevBuf<int32_t> want;
        if ((rc = want.alloc(ctx, (size_t)nloc)))
            break;
        std::vector<DevBuf<int32_t> *> lists;
        int64_t total = 0;
        std::vector<std::vector<int32_t>> host_lists((size_t)nranks);
        for (int q = 0; q < nranks && rc == FV_OK; q++) {
            if (q == rank)
                continue;
            const int64_t q0 = eb[(size_t)q], q1 = eb[(size_t)q + 1];
            if ((rc = want.zero(ctx)))
                break;
            if (q1 > q0)
                hipLaunchKernelGGL(dist_mark_wanted_kernel, dim3(fv_blocks(q1 - q0)), dim3(FV_BLOCK), 0, ctx->stream, q0, q1, pg->colind.p,
                                   (int32_t)lo, (int32_t)hi, want.p);
            DevBuf<int32_t> tmp;
            int64_t cnt = 0;
            if ((rc = tmp.alloc(ctx, (size_t)nloc)) || (rc = compact_flags(ctx, want.p, nloc, tmp.p, 0, &cnt)))
                break;
            d->send_counts[(size_t)q] = cnt;
            host_lists[(size_t)q].resize((size_t)cnt);
            if (cnt > 0 && fv_memcpy_sync(ctx, host_lists[(size_t)q].data(), tmp.p, (size_t)cnt * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess) {
                fv_set_error(ctx, "fv_dist_setup: read-back of a send list failed");
                rc = FV_ERR_HIP;
                break;
            }
            total += cnt;
        }
        if (rc)
            break;
        d->nsend = total;
        if ((rc = d->send_idx.alloc(ctx, (size_t)total)) || (rc = d->sendbuf.alloc(ctx, (size_t)total)))
            break;
        {
            std::vector<int32_t> all;
            all.reserve((size_t)total);
            for (int q = 0; q < nranks; q++)
                all.insert(all.end(), host_lists[(size_t)q].begin(), host_lists[(size_t)q].end());
            if (total > 0 && fv_memcpy_sync(ctx, d->send_idx.p, all.data(), (size_t)total * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) {
                rc = FV_ERR_HIP;
                break;
            }
        }
        // ---- interior / boundary group lists (interior SpMV overlaps the halo exchange)
        const int64_t ngroups = (nloc + 63) >> 6;
        DevBuf<int32_t> fb, fi;
        if ((rc = fb.alloc(ctx, (size_t)ngroups)) || (rc = fi.alloc(ctx, (size_t)ngroups)))
            break;
        hipLaunchKernelGGL(dist_group_flags_kernel, dim3(fv_blocks(ngroups)), dim3(FV_BLOCK), 0, ctx->stream, nloc, pl->rowptr.p, pl->colind.p,
                           fb.p, fi.p);
        if ((rc = d->groups_bnd.alloc(ctx, (size_t)ngroups)) || (rc = d->groups_int.alloc(ctx, (size_t)ngroups)) ||
            (rc = compact_flags(ctx, fb.p, ngroups, d->groups_bnd.p, 0, &d->n_bnd)) ||
            (rc = compact_flags(ctx, fi.p, ngroups, d->groups_int.p, 0, &d->n_int)))
            break;
        // ---- state: slot 0 of the local problem = my slice of the global slot 0
        pl->Ss = pg->Ss;
        pl->assembled = true;
        if ((rc = fv_pcg_prepare(pl)))
            break;
        double *s0 = nullptr;
        if (hipMalloc((void **)&s0, ((size_t)nloc + (size_t)nhalo + FV_VEC_PAD) * sizeof(double)) != hipSuccess) {
            rc = FV_ERR_NOMEM;
            break;
        }
        pl->slot_bases.push_back(s0);
        pl->slots.push_back(s0);
        pl->slot_used.push_back(1);
        if (hipMemsetAsync(s0, 0, ((size_t)nloc + (size_t)nhalo + FV_VEC_PAD) * sizeof(double), ctx->stream) != hipSuccess ||
            (nloc > 0 && hipMemcpyAsync(s0, pg->slots[0] + lo, (size_t)nloc * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) ||
            hipStreamSynchronize(ctx->stream) != hipSuccess) {
            rc = FV_ERR_HIP;
            break;
        }
        pl->transient_ready = true;
        if ((rc = d->red.alloc(ctx, 8)))
            break;
    } while (0);
    if (rc != FV_OK) {
        if (ctx->err.empty())
            fv_set_error(ctx, "fv_dist_setup failed (%d)", rc);
        delete pl; // also deletes d
        return rc;
    }
    *out = pl;
    return FV_OK;
}

extern "C" int fv_dist_plan_sizes(fv_problem *p, int64_t *lo, int64_t *hi, int64_t *nnz_loc, int64_t *nhalo, int64_t *nsend, int64_t *n_int,
                                  int64_t *n_bnd)
{
    if (!p || !p->dist)
        return FV_ERR_ARG;
    const fv_dist *d = p->dist;
    if (lo) *lo = d->lo;
    if (hi) *hi = d->hi;
    if (nnz_loc) *nnz_loc = p->nnz;
    if (nhalo) *nhalo = d->nhalo;
    if (nsend) *nsend = d->nsend;
    if (n_int) *n_int = d->n_int;
    if (n_bnd) *n_bnd = d->n_bnd;
    return FV_OK;
}

extern "C" int fv_dist_get_plan(fv_problem *p, int64_t *rowptr_loc, int64_t *colind_loc, int64_t *halo_cols, int64_t *recv_counts,
                                int64_t *send_counts, int64_t *send_idx, int64_t *groups_bnd)
{
    if (!p || !p->dist)
        return FV_ERR_ARG;
    fv_ctx *ctx = p->ctx;
    FV_HIP(ctx, hipSetDevice(ctx->device));
    fv_dist *d = p->dist;
    DevBuf<int64_t> w;
    auto out64 = [&](const int32_t *src, int64_t cnt, int64_t *dst) -> int {
        if (!dst || cnt <= 0)
            return FV_OK;
        FV_TRY(w.alloc(ctx, (size_t)cnt));
        FV_TRY(fv_widen_indices(ctx, src, w.p, cnt, 0));
        return fv_copy(ctx, dst, w.p, (size_t)cnt * sizeof(int64_t));
    };
    FV_TRY(out64(p->rowptr.p, p->n + 1, rowptr_loc));
    FV_TRY(out64(p->colind.p, p->nnz, colind_loc));
    FV_TRY(out64(d->halo_cols.p, d->nhalo, halo_cols));
    FV_TRY(out64(d->send_idx.p, d->nsend, send_idx));
    FV_TRY(out64(d->groups_bnd.p, d->n_bnd, groups_bnd));
    for (int r = 0; r < d->nranks; r++) {
        if (recv_counts)
            recv_counts[r] = d->recv_counts[(size_t)r];
        if (send_counts)
            send_counts[r] = d->send_counts[(size_t)r];
    }
    return FV_OK;
}

FV_WARM_TU(dist) // (fv_ctx_create loads every code object of the library up front: fv_warm_modules, fv_ctx.hip)
