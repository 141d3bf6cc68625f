// Lean regular-grid problems (FV_OPT_LEAN_SETUP): assembleA / assembleb and the set-up of the solver's storage forms without
// face arrays, incident lists or a CSR in HBM.  A regulargrid mesh is a closed form (fv_grid.hip; /root/reference/src/grid.jl:56-110),
// so row r of what assembleA would have built (/root/reference/src/FiniteVolume.jl:75-108) can be formed wherever it is needed
// (fv_lean.h) — here: by the kernels that fill b and the diagonal, classify the 64-row slices and fill the sliced-DIA and the
// symmetric copies.  Same doubles as the CSR route, bit for bit (tests/test_gpu_lean.py compares every array of the two).
// Compiled without FMA contraction, like fv_grid.hip and fv_assembly.hip.
#include "fv_lean.h"

GridRows fv_grid_rows(const fv_problem *p, double sigma)
{
    GridRows g{};
    g.n1 = p->ns[0];
    g.n2 = p->ns[1];
    g.n3 = p->ns[2];
    g.dx = p->lean_d[0];
    g.dy = p->lean_d[1];
    g.dz = p->lean_d[2];
    g.nodemap = p->nodemap.p;
    g.f2n = p->f2n.p;
    g.K = p->lean_K.p;
    g.meta = p->lean_meta.p;
    g.nK = p->lean_nK;
    g.logt = p->lean_logt;
    g.diagA = p->diagA.p;
    g.D = p->D.p;
    g.sigma = sigma;
    return g;
}

int fv_require_csr(fv_problem *p, const char *what)
{
    if (!p->lean)
        return FV_OK;
    fv_set_error(p->ctx, "%s needs the face arrays / the CSR, which a lean problem (FV_OPT_LEAN_SETUP) does not keep: create the problem with the option at 0",
                 what);
    return FV_ERR_STATE;
}

// ------------------------------------------------------------------ structure
__global__ __launch_bounds__(FV_BLOCK) void lean_count_kernel(GridRows g, int64_t n, unsigned long long *__restrict__ total)
{
    const int64_t r = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    int len = 0;
    if (r < n) {
        GridRow e;
        grid_row(g, r, e, false);
        len = e.len;
    }
    __shared__ int wsum[FV_BLOCK / 64];
    for (int off = 32; off > 0; off >>= 1)
        len += __shfl_xor(len, off, 64);
    if ((threadIdx.x & 63) == 0)
        wsum[threadIdx.x >> 6] = len;
    __syncthreads();
    if (threadIdx.x == 0) {
        int c = 0;
        for (int w = 0; w < FV_BLOCK / 64; w++)
            c += wsum[w];
        if (c)
            atomicAdd(total, (unsigned long long)c);
    }
}

// maps, the kept Dirichlet nodes, b / diagA / dheads, the spacing, nnz (what fv_problem_sizes reports): finish_problem without faces
int fv_lean_finish(fv_problem *p, const int64_t *dirichletnodes, const double mins[3], const double maxs[3])
{
    fv_ctx *ctx = p->ctx;
    std::vector<double> ax[3];
    if (fv_grid_axes(mins, maxs, p->ns, ax) != FV_OK) {
        fv_set_error(ctx, "regulargrid needs ns[d] >= 2 in every dimension");
        return FV_ERR_ARG;
    }
    for (int d = 0; d < 3; d++) {
        p->lean_d[d] = ax[d][1] - ax[d][0]; // grid.jl:65-67, as regulargrid_kernel forms it
        p->lean_mins[d] = mins[d];
        p->lean_maxs[d] = maxs[d];
    }
    FV_TRY(fv_build_maps(p, dirichletnodes));
    FV_TRY(p->dnodes0.alloc(ctx, (size_t)p->ndir));
    if (p->ndir > 0) {
        DevBuf<int64_t> w;
        FV_TRY(w.alloc(ctx, (size_t)p->ndir));
        FV_HIP(ctx, hipMemcpyAsync(w.p, dirichletnodes, (size_t)p->ndir * sizeof(int64_t), hipMemcpyDefault, ctx->stream));
        int bad = 0;
        FV_TRY(fv_narrow_indices(ctx, w.p, p->dnodes0.p, p->ndir, 1, p->N, &bad));
    }
    FV_TRY(p->b.alloc(ctx, (size_t)p->n));
    FV_TRY(p->diagA.alloc(ctx, (size_t)p->n));
    FV_TRY(p->dheads.alloc(ctx, (size_t)p->ndir));
    DevBuf<unsigned long long> cnt;
    FV_TRY(cnt.alloc(ctx, 1));
    FV_TRY(cnt.zero(ctx));
    if (p->n > 0) {
        hipLaunchKernelGGL(lean_count_kernel, dim3(fv_blocks(p->n)), dim3(FV_BLOCK), 0, ctx->stream, fv_grid_rows(p, 0.0), p->n, cnt.p);
        FV_LAUNCH_CHECK(ctx);
    }
    unsigned long long h = 0;
    FV_TRY(fv_copy(ctx, &h, cnt.p, sizeof h));
    p->nnz = (int64_t)h;
    return FV_OK;
}

// ------------------------------------------------------------------ assembleb and the diagonal of assembleA (assemble_rows_kernel without the lists)
// One thread per free row; its faces in face order — (-x, -y, -z) emitted by the lower neighbours, then its own (+x, +y, +z) —, the
// order sparse(I, J, V, n, n, +) and assembleb's `b[...] +=` statements fold contributions in (FiniteVolume.jl:96-103, :131-134).
__global__ __launch_bounds__(FV_BLOCK) void lean_assemble_kernel(GridRows g, int64_t n, const double *__restrict__ sources,
                                                                  const double *__restrict__ dheads, double *__restrict__ b, double *__restrict__ diagA)
{
    const int64_t r = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (r >= n)
        return;
    const int64_t n2 = g.n2, n3 = g.n3, plane = n2 * n3;
    const int64_t c = g.f2n[r];
    const int64_t i3 = c % n3, i2 = (c / n3) % n2, i1 = c / plane;
    double dacc = 0.0, bacc = sources[c]; // FiniteVolume.jl:113-120
    bool first = true;
    auto face = [&](bool exists, int64_t nb, double cond) {
        if (!exists)
            return;
        dacc = first ? cond : dacc + cond; // (row, row, +c): FiniteVolume.jl:96, 98, 101, 103
        first = false;
        const int32_t m = g.nodemap[nb];
        if (m < 0) // the other end is a Dirichlet cell: FiniteVolume.jl:131-134
            bacc += cond * dheads[-m - 1];
    };
    if (i1 > 0)
        face(true, c - plane, grid_face_cond(g, c - plane, i1 - 1, i2, i3, 0));
    if (i2 > 0)
        face(true, c - n3, grid_face_cond(g, c - n3, i1, i2 - 1, i3, 1));
    if (i3 > 0)
        face(true, c - 1, grid_face_cond(g, c - 1, i1, i2, i3 - 1, 2));
    if (i1 < g.n1 - 1)
        face(true, c + plane, grid_face_cond(g, c, i1, i2, i3, 0));
    if (i2 < n2 - 1)
        face(true, c + n3, grid_face_cond(g, c, i1, i2, i3, 1));
    if (i3 < n3 - 1)
        face(true, c + 1, grid_face_cond(g, c, i1, i2, i3, 2));
    diagA[r] = dacc;
    b[r] = bacc;
}

int fv_lean_assemble(fv_problem *p, const double *sources_dev)
{
    fv_ctx *ctx = p->ctx;
    if (p->n > 0) {
        hipLaunchKernelGGL(lean_assemble_kernel, dim3(fv_blocks(p->n)), dim3(FV_BLOCK), 0, ctx->stream, fv_grid_rows(p, 0.0), p->n, sources_dev,
                           (const double *)p->dheads.p, p->b.p, p->diagA.p);
        FV_LAUNCH_CHECK(ctx);
    }
    return FV_OK;
}

// ------------------------------------------------------------------ the plane stride (build_group_order's estimate and vote)
__global__ __launch_bounds__(FV_BLOCK) void lean_far_stride_kernel(GridRows g, int64_t n, int64_t stride, unsigned long long *__restrict__ count)
{
    const int64_t r = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    bool hit = false;
    if (r < n) {
        GridRow e;
        grid_row(g, r, e, false);
        hit = (int64_t)e.off[e.len - 1] == stride; // (the diagonal is always stored: len >= 1)
    }
    __shared__ int wcount[FV_BLOCK / 64];
    const unsigned long long m = __ballot(hit);
    if ((threadIdx.x & 63) == 0)
        wcount[threadIdx.x >> 6] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        int c = 0;
        for (int w = 0; w < FV_BLOCK / 64; w++)
            c += wcount[w];
        if (c)
            atomicAdd(count, (unsigned long long)c);
    }
}

__global__ void lean_last_offset_kernel(GridRows g, int64_t r, int64_t *__restrict__ out)
{
    GridRow e;
    grid_row(g, r, e, false);
    *out = e.off[e.len - 1];
}

// the last stored offset of the middle row (0 when that is its diagonal)
int fv_lean_plane_stride(fv_problem *p, int64_t *stride)
{
    fv_ctx *ctx = p->ctx;
    DevBuf<int64_t> out;
    FV_TRY(out.alloc(ctx, 1));
    hipLaunchKernelGGL(lean_last_offset_kernel, dim3(1), dim3(1), 0, ctx->stream, fv_grid_rows(p, 0.0), p->n / 2, out.p);
    FV_LAUNCH_CHECK(ctx);
    FV_TRY(fv_copy(ctx, stride, out.p, sizeof(int64_t)));
    return FV_OK;
}

int fv_lean_count_far_stride(fv_problem *p, int64_t stride, int64_t *agree)
{
    fv_ctx *ctx = p->ctx;
    DevBuf<unsigned long long> cnt;
    FV_TRY(cnt.alloc(ctx, 1));
    FV_TRY(cnt.zero(ctx));
    hipLaunchKernelGGL(lean_far_stride_kernel, dim3(fv_blocks(p->n)), dim3(FV_BLOCK), 0, ctx->stream, fv_grid_rows(p, 0.0), p->n, stride, cnt.p);
    FV_LAUNCH_CHECK(ctx);
    unsigned long long h = 0;
    FV_TRY(fv_copy(ctx, &h, cnt.p, sizeof h));
    *agree = (int64_t)h;
    return FV_OK;
}

// ------------------------------------------------------------------ sliced DIA: pattern and values (dia_pattern_kernel / dia_fill_kernel, fv_spmv.hip)
__global__ __launch_bounds__(FV_BLOCK) void lean_dia_pattern_kernel(GridRows g, int64_t n, uint8_t *__restrict__ sl_noff, int32_t *__restrict__ sl_off,
                                                                     int32_t *__restrict__ is_dia, int32_t *__restrict__ is_csr)
{
    constexpr int WPB = FV_BLOCK / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t nslices = (n + 63) >> 6;
    const int64_t sl = (int64_t)blockIdx.x * WPB + wave;
    if (sl >= nslices)
        return;
    const int64_t row = (sl << 6) + lane;
    int32_t o[DIA_K];
#pragma unroll
    for (int k = 0; k < DIA_K; k++)
        o[k] = 0x7fffffff;
    if (row < n) {
        GridRow e;
        grid_row(g, row, e, false);
#pragma unroll
        for (int k = 0; k < 7; k++)
            if (k < e.len)
                o[k] = e.off[k];
    }
    // the distinct offsets of the slice in ascending order (rows of a grid store at most seven entries: never too long for the form)
    int32_t last = -0x7fffffff - 1;
    int count = 0;
    bool ok = true;
    int32_t found[DIA_K];
    while (ok) {
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
        ok = false;
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

int fv_lean_dia_pattern(fv_problem *p, uint8_t *sl_noff, int32_t *sl_off, int32_t *is_dia, int32_t *is_csr)
{
    fv_ctx *ctx = p->ctx;
    const int64_t ns = (p->n + 63) >> 6;
    hipLaunchKernelGGL(lean_dia_pattern_kernel, dim3(fv_blocks(ns, FV_BLOCK / 64)), dim3(FV_BLOCK), 0, ctx->stream, fv_grid_rows(p, 0.0), p->n, sl_noff, sl_off,
                       is_dia, is_csr);
    FV_LAUNCH_CHECK(ctx);
    return FV_OK;
}

__global__ __launch_bounds__(FV_BLOCK) void lean_dia_fill_kernel(GridRows g, int64_t n, int64_t ndia, const int32_t *__restrict__ dia_list,
                                                                  const uint8_t *__restrict__ sl_noff, const int32_t *__restrict__ sl_off,
                                                                  const int32_t *__restrict__ dia_pos, double *__restrict__ sval)
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
    GridRow e;
    e.len = 0;
    if (row < n)
        grid_row(g, row, e, true);
    for (int k = 0; k < noff; k++) {
        const int32_t off = sl_off[sl * DIA_K + k];
        double v = 0.0;
#pragma unroll
        for (int j = 0; j < 7; j++)
            if (j < e.len && e.off[j] == off)
                v = e.val[j];
        sval[(base + k) * 64 + lane] = v;
    }
}

int fv_lean_dia_fill(fv_problem *p, double sigma, int64_t count, const int32_t *list)
{
    fv_ctx *ctx = p->ctx;
    if (count <= 0)
        return FV_OK;
    hipLaunchKernelGGL(lean_dia_fill_kernel, dim3(fv_blocks(count, FV_BLOCK / 64)), dim3(FV_BLOCK), 0, ctx->stream, fv_grid_rows(p, sigma), p->n, count, list,
                       (const uint8_t *)p->sl_noff.p, (const int32_t *)p->sl_off.p, (const int32_t *)p->dia_pos.p, p->dia_vals.p);
    FV_LAUNCH_CHECK(ctx);
    return FV_OK;
}

// ------------------------------------------------------------------ the symmetric copy (symdia_fill_kernel, fv_spmv.hip); a(i, i - d) is the bits of
// a(i - d, i) by construction — one face, one product —, so there is nothing for symdia_check_kernel to look at
__global__ __launch_bounds__(FV_BLOCK) void lean_symdia_fill_kernel(GridRows g, int64_t n, int32_t d1, int32_t d2, int32_t d3, double *__restrict__ dg,
                                                                     double *__restrict__ u1, double *__restrict__ u2, double *__restrict__ u3)
{
    const int64_t r = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (r >= n)
        return;
    GridRow e;
    grid_row(g, r, e, true);
#pragma unroll
    for (int j = 0; j < 7; j++)
        if (j < e.len) {
            const int32_t off = e.off[j];
            const double v = e.val[j];
            if (off == 0)
                dg[r] = v;
            else if (off == d1)
                u1[r] = v;
            else if (off == d2)
                u2[r] = v;
            else if (off == d3)
                u3[r] = v;
        }
}

int fv_lean_symdia_fill(fv_problem *p, double sigma, int32_t d1, int32_t d2, int32_t d3, double *dg, double *u1, double *u2, double *u3)
{
    fv_ctx *ctx = p->ctx;
    hipLaunchKernelGGL(lean_symdia_fill_kernel, dim3(fv_blocks(p->n)), dim3(FV_BLOCK), 0, ctx->stream, fv_grid_rows(p, sigma), p->n, d1, d2, d3, dg, u1, u2, u3);
    FV_LAUNCH_CHECK(ctx);
    return FV_OK;
}

// ------------------------------------------------------------------ the product over the 64-row groups the sliced-DIA form does not take (Dirichlet cells
// inside the box leave groups with more than eight distinct column offsets): the CSR route hands them to the wave-stream kernel; here each
// row is formed and applied on the spot — terms in ascending column order, the shift last, like that kernel.  A handful of groups per plane.
template <bool DOT>
__global__ __launch_bounds__(FV_BLOCK) void lean_rows_spmv_kernel(GridRows g, int64_t n, const double *__restrict__ x, double *__restrict__ y,
                                                                   const double *__restrict__ shift, double sigma, double *__restrict__ partials,
                                                                   const PcgScalars *__restrict__ scal, const int32_t *__restrict__ list, int64_t count)
{
    constexpr int WPB = FV_BLOCK / 64;
    __shared__ double smem[WPB];
    if (scal && scal->done)
        return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double dacc = 0.0;
    for (int64_t pos = (int64_t)blockIdx.x * WPB + wave; pos < count; pos += (int64_t)gridDim.x * WPB) {
        const int64_t row = ((list ? (int64_t)list[pos] : pos) << 6) + lane; // (no list: every group, in order)
        if (row < n) {
            GridRow e;
            grid_row(g, row, e, true);
            double sum = 0.0;
#pragma unroll
            for (int k = 0; k < 7; k++)
                if (k < e.len)
                    sum += e.val[k] * x[row + e.off[k]];
            const double xr = x[row];
            if (shift)
                sum += sigma * shift[row] * xr;
            y[row] = sum;
            if (DOT)
                dacc += xr * sum;
        }
    }
    if (DOT) {
        for (int off = 32; off > 0; off >>= 1)
            dacc += __shfl_xor(dacc, off, 64);
        if (lane == 0)
            smem[wave] = dacc;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (int w = 0; w < WPB; w++)
                t += smem[w];
            partials[blockIdx.x] = t;
        }
    }
}

int fv_lean_rows_spmv(fv_problem *p, double tag, const double *x, double *y, const double *shift, double sigma, bool dot, double *partials, const PcgScalars *scal,
                      const int32_t *list, int64_t count, int grid)
{
    fv_ctx *ctx = p->ctx;
    const GridRows g = fv_grid_rows(p, tag);
    if (dot)
        hipLaunchKernelGGL(lean_rows_spmv_kernel<true>, dim3(grid), dim3(FV_BLOCK), 0, ctx->stream, g, p->n, x, y, shift, sigma, partials, scal, list, count);
    else
        hipLaunchKernelGGL(lean_rows_spmv_kernel<false>, dim3(grid), dim3(FV_BLOCK), 0, ctx->stream, g, p->n, x, y, shift, sigma, partials, scal, list, count);
    FV_LAUNCH_CHECK(ctx);
    return FV_OK;
}

__global__ void lean_len_kernel(GridRows g, int64_t r0, int64_t count, int32_t *__restrict__ len);
// ------------------------------------------------------------------ the operator as a CSR with int32 offsets, for a set-up that needs one and gives it
// back (the AMG hierarchy's matching and Galerkin products read level 0 once; its cycles use the problem's own product): nnz < 2^31
__global__ __launch_bounds__(FV_BLOCK) void lean_csr32_kernel(GridRows g, int64_t n, const int32_t *__restrict__ rowptr, int32_t *__restrict__ colind,
                                                               double *__restrict__ vals, int32_t *__restrict__ diagpos)
{
    const int64_t r = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (r >= n)
        return;
    GridRow e;
    grid_row(g, r, e, true);
    const int32_t s = rowptr[r];
#pragma unroll
    for (int k = 0; k < 7; k++)
        if (k < e.len) {
            colind[s + k] = (int32_t)(r + (int64_t)e.off[k]);
            vals[s + k] = e.val[k];
            if (diagpos && e.off[k] == 0)
                diagpos[r] = s + k;
        }
}

int fv_lean_csr32(fv_problem *p, DevBuf<int32_t> &rowptr, DevBuf<int32_t> &colind, DevBuf<double> &vals, DevBuf<int32_t> *diagpos)
{
    fv_ctx *ctx = p->ctx;
    if (p->nnz >= 0x7fffffffLL - 2) {
        fv_set_error(ctx, "the operator's %lld entries do not fit a CSR with int32 offsets", (long long)p->nnz);
        return FV_ERR_TOO_LARGE;
    }
    const int64_t n = p->n;
    DevBuf<int32_t> len;
    FV_TRY(len.alloc(ctx, (size_t)n));
    FV_TRY(rowptr.alloc(ctx, (size_t)n + 1));
    const GridRows g = fv_grid_rows(p, 0.0);
    hipLaunchKernelGGL(lean_len_kernel, dim3(fv_blocks(n)), dim3(FV_BLOCK), 0, ctx->stream, g, (int64_t)0, n, len.p);
    FV_LAUNCH_CHECK(ctx);
    int64_t total = 0;
    FV_TRY(fv_exclusive_scan_i32(ctx, len.p, rowptr.p, n, &total));
    if (total != p->nnz) {
        fv_set_error(ctx, "internal: the rows of the lean problem hold %lld entries, %lld were counted at its creation", (long long)total, (long long)p->nnz);
        return FV_ERR_STATE;
    }
    FV_TRY(colind.alloc(ctx, (size_t)total + 2));
    FV_TRY(vals.alloc(ctx, (size_t)total + 2));
    FV_TRY(colind.zero(ctx));
    FV_TRY(vals.zero(ctx));
    if (diagpos)
        FV_TRY(diagpos->alloc(ctx, (size_t)n));
    hipLaunchKernelGGL(lean_csr32_kernel, dim3(fv_blocks(n)), dim3(FV_BLOCK), 0, ctx->stream, g, n, (const int32_t *)rowptr.p, colind.p, vals.p,
                       diagpos ? diagpos->p : (int32_t *)nullptr);
    FV_LAUNCH_CHECK(ctx);
    FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FV_OK;
}

// ------------------------------------------------------------------ fv_get_csc of a lean problem: assembleA's matrix written out from the rows, a
// window of rows at a time (the CSR never exists as a whole: 2^24 rows — at most 1.2e8 entries, 1.9 GB of scratch — per pass).  A is
// symmetric in pattern and bit for bit in value (one face, one product), so the rows are the columns.
__global__ __launch_bounds__(FV_BLOCK) void lean_len_kernel(GridRows g, int64_t r0, int64_t count, int32_t *__restrict__ len)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i >= count)
        return;
    GridRow e;
    grid_row(g, r0 + i, e, false);
    len[i] = e.len;
}

__global__ __launch_bounds__(FV_BLOCK) void lean_export_kernel(GridRows g, int64_t r0, int64_t count, const int32_t *__restrict__ start, int64_t base,
                                                                int64_t *__restrict__ colptr, int64_t *__restrict__ rowval, double *__restrict__ nzval,
                                                                int want_vals)
{
    const int64_t i = (int64_t)blockIdx.x * FV_BLOCK + threadIdx.x;
    if (i >= count)
        return;
    const int64_t r = r0 + i;
    GridRow e;
    grid_row(g, r, e, want_vals != 0);
    const int64_t s = start[i];
    if (colptr)
        colptr[i] = base + s + 1; // 1-based, like the reference's SparseMatrixCSC
#pragma unroll
    for (int k = 0; k < 7; k++)
        if (k < e.len) {
            if (rowval)
                rowval[s + k] = r + (int64_t)e.off[k] + 1;
            if (want_vals)
                nzval[s + k] = e.val[k];
        }
}

int fv_lean_get_csc(fv_problem *p, int64_t *colptr, int64_t *rowval, double *nzval)
{
    fv_ctx *ctx = p->ctx;
    const int64_t n = p->n, W = (int64_t)1 << 24;
    DevBuf<int32_t> len, start;
    DevBuf<int64_t> cp, rv;
    DevBuf<double> nv;
    FV_TRY(len.alloc(ctx, (size_t)W));
    FV_TRY(start.alloc(ctx, (size_t)W + 1));
    if (colptr)
        FV_TRY(cp.alloc(ctx, (size_t)W));
    if (rowval)
        FV_TRY(rv.alloc(ctx, (size_t)7 * W));
    if (nzval)
        FV_TRY(nv.alloc(ctx, (size_t)7 * W));
    const GridRows g = fv_grid_rows(p, 0.0);
    int64_t base = 0;
    for (int64_t r0 = 0; r0 < n; r0 += W) {
        const int64_t count = n - r0 < W ? n - r0 : W;
        hipLaunchKernelGGL(lean_len_kernel, dim3(fv_blocks(count)), dim3(FV_BLOCK), 0, ctx->stream, g, r0, count, len.p);
        FV_LAUNCH_CHECK(ctx);
        int64_t total = 0;
        FV_TRY(fv_exclusive_scan_i32(ctx, len.p, start.p, count, &total));
        hipLaunchKernelGGL(lean_export_kernel, dim3(fv_blocks(count)), dim3(FV_BLOCK), 0, ctx->stream, g, r0, count, (const int32_t *)start.p, base,
                           colptr ? cp.p : (int64_t *)nullptr, rowval ? rv.p : (int64_t *)nullptr, nzval ? nv.p : (double *)nullptr, nzval ? 1 : 0);
        FV_LAUNCH_CHECK(ctx);
        if (colptr)
            FV_TRY(fv_copy(ctx, colptr + r0, cp.p, (size_t)count * sizeof(int64_t)));
        if (rowval)
            FV_TRY(fv_copy(ctx, rowval + base, rv.p, (size_t)total * sizeof(int64_t)));
        if (nzval)
            FV_TRY(fv_copy(ctx, nzval + base, nv.p, (size_t)total * sizeof(double)));
        base += total;
    }
    if (colptr) {
        const int64_t last = base + 1;
        FV_TRY(fv_copy(ctx, colptr + n, &last, sizeof last));
    }
    if (base != p->nnz) {
        fv_set_error(ctx, "internal: the rows of the lean problem hold %lld entries, %lld were counted at its creation", (long long)base, (long long)p->nnz);
        return FV_ERR_STATE;
    }
    return FV_OK;
}

FV_WARM_TU(lean)
