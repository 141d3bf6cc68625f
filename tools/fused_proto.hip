// Prototype of the fused time step of the one-iteration regime: the vector update of step k (today's
// K2S: alpha, x_out, residual, the next step's set-up z') and the SpMV of step k + 1 (today's K1: q' = (A + sigma D) z',
// partial z'.q') in ONE pass over 2-D tiles of a plane, marching through the planes.  Standalone (synthetic symmetric
// 7-point operator, no library): it answers whether the fused access pattern reaches the streaming rate before the
// speculation / fall-back logic is carried into the library.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/fused_proto.hip -o tools/fused_proto && tools/fused_proto [P L nz reps]
// Bytes per row: x_in, q, z in (24) + storage code (1) + U1, U2, U3 (24) in; x_out, z', q' out (24) = 73
// (+ the stored diagonal on halo rows); the unfused pair moves 41 + 49 = 90.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include <cmath>

#define CK(x)                                                                                    \
    do {                                                                                         \
        hipError_t e_ = (x);                                                                     \
        if (e_ != hipSuccess) {                                                                  \
            printf("%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));             \
            exit(1);                                                                             \
        }                                                                                        \
    } while (0)

struct Tab {
    double v[16];
};

// one row of the update: everything today's K2S does, with sD = sigma * D of the row and d = the shifted diagonal
struct RowOut {
    double xn, zn, r, c, h, mv;
};
template <bool FASTRCP = false>
__device__ __forceinline__ RowOut update_row(double xin, double q, double z, double d, double sD, double alpha)
{
    RowOut o;
    const double mv = FASTRCP ? __builtin_amdgcn_rcp(d) : 1.0 / d;
    o.mv = mv;
    const double r0 = z * d;
    o.xn = xin + alpha * z;
    o.r = r0 - alpha * q;
    o.c = o.r + sD * (o.xn - xin);
    o.h = sD * o.xn;
    o.zn = mv * o.c;
    return o;
}

// one row of the v-form update: d = the shifted diagonal, sD = sigma D of the row
struct VRow {
    double xn, zn, r, c, h, mv;
};
__device__ __forceinline__ VRow vrow(double xin, double z, double v, double d, double sD, double alpha)
{
    VRow o;
    o.mv = 1.0 / d;
    o.xn = xin + alpha * z;
    o.zn = z + alpha * v;
    o.c = d * o.zn;                    // rho' = the next system's residual at x_out
    o.r = o.c - sD * (o.xn - xin);     // the finished step's residual
    o.h = sD * o.xn;
    return o;
}
__global__ void ref_update_v(int64_t n, const double *xin, const double *z, const double *v, double alpha, double *xout, double *zn)
{
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n)
        return;
    xout[r] = xin[r] + alpha * z[r];
    zn[r] = z[r] + alpha * v[r];
}
__global__ void ref_q_to_v(int64_t n, const double *qv, const double *zn, const double *dia, const uint8_t *code, Tab sD, double *v)
{
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n)
        return;
    v[r] = -((1.0 / dia[r]) * (qv[r] - sD.v[code[r]] * zn[r]));
}

// ------------------------------------------------------------------ naive reference (one thread per row)
__global__ void ref_diag(int64_t n, int32_t nz, int32_t d3, const double *u1, const double *u2, const double *u3, const uint8_t *code, Tab sD,
                         double *dia)
{
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n)
        return;
    double so = u3[r - d3] + u2[r - nz];
    so += u1[r - 1];
    so += u3[r];
    so += u2[r];
    so += u1[r];
    dia[r] = -so + sD.v[code[r]];
}
__global__ void ref_update(int64_t n, const double *xin, const double *q, const double *z, const double *dia, const uint8_t *code, Tab sD,
                           double alpha, double *xout, double *zn)
{
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n)
        return;
    const RowOut o = update_row(xin[r], q[r], z[r], dia[r], sD.v[code[r]], alpha);
    xout[r] = o.xn;
    zn[r] = o.zn;
}
template <bool ORDER2>
__global__ void ref_spmv(int64_t n, int32_t nz, int32_t d3, const double *u1, const double *u2, const double *u3, const double *dia,
                         const double *x, double *y)
{
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n)
        return;
    double s = ORDER2 ? u3[r - d3] * x[r - d3] + dia[r] * x[r] : 0.0;
    if (!ORDER2)
        s += u3[r - d3] * x[r - d3];
    s += u2[r - nz] * x[r - nz];
    s += u1[r - 1] * x[r - 1];
    if (!ORDER2)
        s += dia[r] * x[r];
    s += u1[r] * x[r + 1];
    s += u2[r] * x[r + nz];
    s += u3[r] * x[r + d3];
    y[r] = s;
}

// ------------------------------------------------------------------ today's K2S shape (49 B/row), for the box's streaming rate
__global__ __launch_bounds__(256) void k2s_like(int64_t n, const double *__restrict__ xin, const double *__restrict__ q,
                                                const double *__restrict__ z, const double *__restrict__ minv,
                                                const uint8_t *__restrict__ code, Tab sD, double alpha, double *__restrict__ xout,
                                                double *__restrict__ zn, double *__restrict__ part)
{
    __shared__ double tab[16];
    if (threadIdx.x < 16)
        tab[threadIdx.x] = sD.v[threadIdx.x];
    __syncthreads();
    double acc = 0.0;
    const int64_t n2 = n >> 1;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (int64_t)gridDim.x * 256) {
        const double2 xv = make_double2(__builtin_nontemporal_load(xin + 2 * i), __builtin_nontemporal_load(xin + 2 * i + 1));
        const double2 qv = make_double2(__builtin_nontemporal_load(q + 2 * i), __builtin_nontemporal_load(q + 2 * i + 1));
        const double2 mv = make_double2(__builtin_nontemporal_load(minv + 2 * i), __builtin_nontemporal_load(minv + 2 * i + 1));
        const double2 zv = reinterpret_cast<const double2 *>(z)[i];
        const unsigned c = reinterpret_cast<const uint16_t *>(code)[i];
        const double s0 = tab[c & 255u], s1 = tab[c >> 8];
        double r0 = zv.x / mv.x, r1 = zv.y / mv.y;
        const double x0 = xv.x + alpha * zv.x, x1 = xv.y + alpha * zv.y;
        r0 -= alpha * qv.x;
        r1 -= alpha * qv.y;
        const double c0 = r0 + s0 * (x0 - xv.x), c1 = r1 + s1 * (x1 - xv.y);
        __builtin_nontemporal_store(x0, xout + 2 * i);
        __builtin_nontemporal_store(x1, xout + 2 * i + 1);
        reinterpret_cast<double2 *>(zn)[i] = make_double2(mv.x * c0, mv.y * c1);
        acc += r0 * r0 + r1 * r1 + c0 * c0 + c1 * c1;
    }
    for (int off = 32; off > 0; off >>= 1)
        acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0)
        atomicAdd(part + (blockIdx.x & 1023), acc);
}

// ------------------------------------------------------------------ the fused kernel, 2-D tiles
// A block of TL * TW / 2 threads owns a tile of TL lines x TW columns of a plane (two consecutive columns per thread: one
// 16-byte access per stream) and marches through the planes [p0, p1) of a segment.  Per plane step:
//   L  loads of plane p + 2 (own rows: x_in, q, z, code, U1, U2, U3; halo rows: x_in, q, z, code, stored diagonal, U halo)
//   U  update of plane p + 1 from the streams loaded one step earlier: diagonal from the six arms (zero row sum) + sigma D,
//      M^-1 = 1 / d, x_out, z' -> global; z' of the tile's halo rows (one line above / below, one column left / right) is
//      formed again by the first 2 TW + 2 TL threads from the same inputs (their diagonal is streamed)
//   S  q'(p) = A z'(p): +-1 / +-line arms of z' and the -1 / -line matrix values from the LDS tiles of plane p, +-plane arms
//      from registers; q' -> global; partial z'.q'
//   W  the tiles of plane p + 1 (z') and p + 2 (U1, U2) into LDS
// Six partial sums per block: r.M^-1 r and r.r of the finished step, c.z', c.c, h.h of the next step's set-up, z'.q'.
constexpr int NSUM = 6;
template <int TL, int TW, int WPS>
__global__ __launch_bounds__(TL *TW / 2, WPS) void fused_step_kernel(int32_t nz, int32_t L, int32_t d3, int32_t nplanes, int32_t seglen, int32_t tilesC,
                                                                    int32_t tiles, int32_t nsegs, const double *__restrict__ u1,
                                                                    const double *__restrict__ u2, const double *__restrict__ u3,
                                                                    const double *__restrict__ dia, const uint8_t *__restrict__ code, Tab sDt,
                                                                    const double *__restrict__ xin, const double *__restrict__ q,
                                                                    const double *__restrict__ z, double alpha, double *__restrict__ xout,
                                                                    double *__restrict__ znext, double *__restrict__ qnext,
                                                                    double *__restrict__ partials)
{
    constexpr int NT = TL * TW / 2, HC = TW / 2; // threads, threads per tile line
    constexpr int ZS = TW + 4, U1S = TW + 2, U2S = TW; // LDS row strides: own column tc sits at 2 + tc (z', U1) so that pairs stay 16-byte aligned
    constexpr int NH = 2 * TW + 2 * TL;               // halo threads
    __shared__ __align__(16) double zs[(TL + 2) * ZS];
    __shared__ __align__(16) double u1s[TL * U1S];
    __shared__ __align__(16) double u2s[(TL + 1) * U2S];
    __shared__ double tab[16];
    __shared__ double red[NT / 64];
    const int tid = (int)threadIdx.x;
    if (tid < 16)
        tab[tid] = sDt.v[tid];
    const int tl = tid / HC, tc = 2 * (tid % HC);
    const int xcd = (int)(blockIdx.x & 7);
    const int64_t items = (int64_t)tiles * nsegs, per_xcd = (items + 7) / 8;
    double acc[NSUM] = {0, 0, 0, 0, 0, 0};
    for (int64_t j = (int64_t)(blockIdx.x >> 3); j < per_xcd; j += (int64_t)(gridDim.x >> 3)) {
        const int64_t item = (int64_t)xcd * per_xcd + j;
        if (item >= items)
            break;
        const int32_t seg = (int32_t)(item / tiles), tile = (int32_t)(item % tiles);
        const int32_t l0 = (tile / tilesC) * TL, c0 = (tile % tilesC) * TW;
        const int32_t p0 = 1 + seg * seglen, p1 = (p0 + seglen < nplanes) ? p0 + seglen : nplanes;
        if (p0 >= p1)
            continue;
        const bool own = (l0 + tl < L) && (c0 + tc < nz);
        const int32_t o = (l0 + tl) * nz + c0 + tc;
        // halo role of this thread
        bool hv = false;      // has a halo row that exists
        int32_t ho = 0;       // its in-plane offset
        int hz = 0;           // where its z' goes in zs
        int hu = -1;          // where its U value goes (u2s for the line above, u1s for the column to the left), -1: none
        bool hu_is_u2 = false;
        if (tid < TW) { // line above
            hv = l0 >= 1 && c0 + tid < nz;
            ho = (l0 - 1) * nz + c0 + tid;
            hz = 2 + tid;
            hu = tid;
            hu_is_u2 = true;
        } else if (tid < 2 * TW) { // line below
            const int t = tid - TW;
            hv = l0 + TL < L && c0 + t < nz;
            ho = (l0 + TL) * nz + c0 + t;
            hz = (TL + 1) * ZS + 2 + t;
        } else if (tid < 2 * TW + TL) { // column to the left
            const int t = tid - 2 * TW;
            hv = c0 >= 1 && l0 + t < L;
            ho = (l0 + t) * nz + c0 - 1;
            hz = (t + 1) * ZS + 1;
            hu = t * U1S + 1;
        } else if (tid < NH) { // column to the right
            const int t = tid - 2 * TW - TL;
            hv = c0 + TW < nz && l0 + t < L;
            ho = (l0 + t) * nz + c0 + TW;
            hz = (t + 1) * ZS + 2 + TW;
        }
        const bool ht = tid < NH;
        auto ld2 = [&](const double *a, int64_t row, bool pred) -> double2 {
            return pred ? *reinterpret_cast<const double2 *>(a + row) : make_double2(0.0, 0.0);
        };
        auto ld2nt = [&](const double *a, int64_t row, bool pred) -> double2 {
            return pred ? make_double2(__builtin_nontemporal_load(a + row), __builtin_nontemporal_load(a + row + 1)) : make_double2(0.0, 0.0);
        };
        auto ld1 = [&](const double *a, int64_t row, bool pred) -> double { return pred ? a[row] : 0.0; };
        auto sd2 = [&](int64_t row, bool pred) -> double2 {
            if (!pred)
                return make_double2(0.0, 0.0);
            const unsigned c = *reinterpret_cast<const uint16_t *>(code + row);
            return make_double2(tab[c & 255u], tab[c >> 8]);
        };
        __syncthreads(); // tab; the previous item's last LDS reads
        // ---------------- prologue: z'(p0 - 1) of the own rows, z'(p0) of own + halo rows (stored diagonals), the tiles
        int64_t r = (int64_t)p0 * d3 + o; // first own row in the centre plane
        double2 Zm, Zc, Dc;
        {
            const double2 xi = ld2(xin, r - d3, own), qq = ld2(q, r - d3, own), zz = ld2(z, r - d3, own), dd = ld2(dia, r - d3, own),
                          ss = sd2(r - d3, own);
            const RowOut a = update_row(xi.x, qq.x, zz.x, own ? dd.x : 1.0, ss.x, alpha), b = update_row(xi.y, qq.y, zz.y, own ? dd.y : 1.0, ss.y, alpha);
            Zm = make_double2(a.zn, b.zn);
        }
        {
            const double2 xi = ld2(xin, r, own), qq = ld2(q, r, own), zz = ld2(z, r, own), dd = ld2(dia, r, own), ss = sd2(r, own);
            const RowOut a = update_row(xi.x, qq.x, zz.x, own ? dd.x : 1.0, ss.x, alpha), b = update_row(xi.y, qq.y, zz.y, own ? dd.y : 1.0, ss.y, alpha);
            Zc = make_double2(a.zn, b.zn);
            Dc = dd;
            if (own) {
                *reinterpret_cast<double2 *>(xout + r) = make_double2(a.xn, b.xn);
                *reinterpret_cast<double2 *>(znext + r) = Zc;
                acc[0] += a.r * (a.mv * a.r) + b.r * (b.mv * b.r);
                acc[1] += a.r * a.r + b.r * b.r;
                acc[2] += a.c * a.zn + b.c * b.zn;
                acc[3] += a.c * a.c + b.c * b.c;
                acc[4] += a.h * a.h + b.h * b.h;
            }
        }
        double2 A3m = ld2nt(u3, r - d3, own), A3c = ld2nt(u3, r, own), V1c = ld2nt(u1, r, own), V2c = ld2nt(u2, r, own);
        *reinterpret_cast<double2 *>(zs + (tl + 1) * ZS + 2 + tc) = Zc;
        *reinterpret_cast<double2 *>(u1s + tl * U1S + 2 + tc) = V1c;
        *reinterpret_cast<double2 *>(u2s + (tl + 1) * U2S + tc) = V2c;
        if (ht) {
            const int64_t hr = (int64_t)p0 * d3 + ho;
            double zn = 0.0;
            if (hv) {
                const RowOut a = update_row(xin[hr], q[hr], z[hr], dia[hr], tab[code[hr]], alpha);
                zn = a.zn;
            }
            zs[hz] = zn;
            if (hu >= 0)
                (hu_is_u2 ? u2s : u1s)[hu] = hv ? (hu_is_u2 ? u2 : u1)[hr] : 0.0;
        }
        __syncthreads();
        double v1m0 = u1s[tl * U1S + 2 + tc - 1];
        double2 V2m = *reinterpret_cast<const double2 *>(u2s + tl * U2S + tc);
        // streams of plane p0 + 1 (set A), its matrix values
        int64_t rn = r + d3;
        double2 Xa = ld2nt(xin, rn, own), Qa = ld2nt(q, rn, own), Za = ld2(z, rn, own), Sa = sd2(rn, own);
        double2 V1n = ld2nt(u1, rn, own), V2n = ld2nt(u2, rn, own), A3n = ld2nt(u3, rn, own);
        double2 Da = (p0 + 1 >= p1) ? ld2(dia, rn, own) : make_double2(0.0, 0.0); // the plane after the segment: stored diagonal
        double hxa = 0.0, hqa = 0.0, hza = 0.0, hda = 1.0, hsa = 0.0, hua = 0.0;
        if (ht && hv) {
            const int64_t hr = (int64_t)(p0 + 1) * d3 + ho;
            hxa = xin[hr];
            hqa = q[hr];
            hza = z[hr];
            hda = dia[hr];
            hsa = tab[code[hr]];
            if (hu >= 0)
                hua = (hu_is_u2 ? u2 : u1)[hr];
        }
        __syncthreads(); // lower arms of plane p0 are in registers: the U tiles may move on
        *reinterpret_cast<double2 *>(u1s + tl * U1S + 2 + tc) = V1n;
        *reinterpret_cast<double2 *>(u2s + (tl + 1) * U2S + tc) = V2n;
        if (ht && hu >= 0)
            (hu_is_u2 ? u2s : u1s)[hu] = hua;
        __syncthreads();
        for (int32_t p = p0; p < p1; p++, r += d3) {
            // ---- L: loads of plane p + 2
            const bool more = p + 2 <= p1; // plane p + 2 is needed (z' of plane p1 is the +plane arm of the last plane)
            const int64_t r2 = r + 2 * (int64_t)d3;
            double2 Xb = make_double2(0.0, 0.0), Qb = Xb, Zb = Xb, Sb = Xb, V1b = Xb, V2b = Xb, A3b = Xb, Db = Xb;
            double hxb = 0.0, hqb = 0.0, hzb = 0.0, hdb = 1.0, hsb = 0.0, hub = 0.0;
            if (more) {
                Xb = ld2nt(xin, r2, own);
                Qb = ld2nt(q, r2, own);
                Zb = ld2(z, r2, own);
                Sb = sd2(r2, own);
                V1b = ld2nt(u1, r2, own);
                V2b = ld2nt(u2, r2, own);
                A3b = ld2nt(u3, r2, own);
                if (p + 2 >= p1)
                    Db = ld2(dia, r2, own);
                if (ht && hv && p + 2 < p1) { // (halo rows of the plane after the segment are not needed)
                    const int64_t hr = (int64_t)(p + 2) * d3 + ho;
                    hxb = xin[hr];
                    hqb = q[hr];
                    hzb = z[hr];
                    hdb = dia[hr];
                    hsb = tab[code[hr]];
                    if (hu >= 0)
                        hub = (hu_is_u2 ? u2 : u1)[hr];
                }
            }
            // ---- U: update of plane p + 1 (own rows; written only when the plane belongs to the segment)
            const bool inseg = p + 1 < p1;
            const double v1m0n = u1s[tl * U1S + 2 + tc - 1];
            const double2 V2mn = *reinterpret_cast<const double2 *>(u2s + tl * U2S + tc);
            double2 Dn;
            if (inseg) {
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
                Dn = Da;
            const RowOut ua = update_row(Xa.x, Qa.x, Za.x, own ? Dn.x : 1.0, Sa.x, alpha), ub = update_row(Xa.y, Qa.y, Za.y, own ? Dn.y : 1.0, Sa.y, alpha);
            const double2 Zn = make_double2(ua.zn, ub.zn);
            if (own && inseg) {
                __builtin_nontemporal_store(ua.xn, xout + rn);
                __builtin_nontemporal_store(ub.xn, xout + rn + 1);
                *reinterpret_cast<double2 *>(znext + rn) = Zn;
                acc[0] += ua.r * (ua.mv * ua.r) + ub.r * (ub.mv * ub.r);
                acc[1] += ua.r * ua.r + ub.r * ub.r;
                acc[2] += ua.c * ua.zn + ub.c * ub.zn;
                acc[3] += ua.c * ua.c + ub.c * ub.c;
                acc[4] += ua.h * ua.h + ub.h * ub.h;
            }
            double hzn = 0.0;
            if (ht && hv && inseg) {
                const RowOut a = update_row(hxa, hqa, hza, hda, hsa, alpha);
                hzn = a.zn;
            }
            // ---- S: q'(p)
            {
                const double *zrow = zs + (tl + 1) * ZS + 2 + tc;
                const double x1m0 = zrow[-1], x1p1 = zrow[2];
                const double2 x2m = *reinterpret_cast<const double2 *>(zrow - ZS), x2p = *reinterpret_cast<const double2 *>(zrow + ZS);
                double s0 = 0.0, s1 = 0.0;
                s0 += A3m.x * Zm.x;
                s1 += A3m.y * Zm.y;
                s0 += V2m.x * x2m.x;
                s1 += V2m.y * x2m.y;
                s0 += v1m0 * x1m0;
                s1 += V1c.x * Zc.x;
                s0 += Dc.x * Zc.x;
                s1 += Dc.y * Zc.y;
                s0 += V1c.x * Zc.y;
                s1 += V1c.y * x1p1;
                s0 += V2c.x * x2p.x;
                s1 += V2c.y * x2p.y;
                s0 += A3c.x * Zn.x;
                s1 += A3c.y * Zn.y;
                if (own) {
                    __builtin_nontemporal_store(s0, qnext + r);
                    __builtin_nontemporal_store(s1, qnext + r + 1);
                    acc[5] += Zc.x * s0 + Zc.y * s1;
                }
            }
            __syncthreads(); // everybody has read the z' tile of plane p and the U tiles of plane p + 1
            if (inseg) {
                *reinterpret_cast<double2 *>(zs + (tl + 1) * ZS + 2 + tc) = Zn;
                if (ht)
                    zs[hz] = hzn;
                *reinterpret_cast<double2 *>(u1s + tl * U1S + 2 + tc) = V1b;
                *reinterpret_cast<double2 *>(u2s + (tl + 1) * U2S + tc) = V2b;
                if (ht && hu >= 0)
                    (hu_is_u2 ? u2s : u1s)[hu] = hub;
            }
            __syncthreads();
            Zm = Zc;
            Zc = Zn;
            Dc = Dn;
            A3m = A3c;
            A3c = A3n;
            A3n = A3b;
            V1c = V1n;
            V1n = V1b;
            V2c = V2n;
            V2n = V2b;
            v1m0 = v1m0n;
            V2m = V2mn;
            Xa = Xb;
            Qa = Qb;
            Za = Zb;
            Sa = Sb;
            Da = Db;
            hxa = hxb;
            hqa = hqb;
            hza = hzb;
            hda = hdb;
            hsa = hsb;
            rn += d3;
        }
    }
    for (int k = 0; k < NSUM; k++) {
        double v = acc[k];
        for (int off = 32; off > 0; off >>= 1)
            v += __shfl_xor(v, off, 64);
        __syncthreads();
        if ((tid & 63) == 0)
            red[tid >> 6] = v;
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
            for (int w = 0; w < NT / 64; w++)
                t += red[w];
            partials[(size_t)k * gridDim.x + blockIdx.x] = t;
        }
    }
}


// ------------------------------------------------------------------ v2: matrix tiles in an LDS ring, one barrier per plane step
// As above, with what the first version kept in registers moved into LDS so that two blocks fit a CU: the z' tile is double
// buffered (the update of plane p + 1 writes the other half while the product of plane p reads this one), the U1 / U2 tiles
// live in a ring of three plane slots (p and p + 1 are read, p + 2 is written when its loads land), the halo rows' streams
// are loaded in the step that uses them.  Addresses are "plane base (uniform) + 32-bit in-plane offset".
template <int TL, int TW, int WPS, int DBG = 0> // DBG (wrong results): 1 no halo rows, 2 reciprocal by v_rcp_f64, 4 no q' store, 8 no x_out / z' store
__global__ __launch_bounds__(TL *TW / 2, WPS) void fused_step_v2(int32_t nz, int32_t L, int32_t d3, int32_t nplanes, int32_t seglen, int32_t tilesC,
                                                                  int32_t tiles, int32_t nsegs, const double *__restrict__ u1,
                                                                  const double *__restrict__ u2, const double *__restrict__ u3,
                                                                  const double *__restrict__ dia, const uint8_t *__restrict__ code, Tab sDt,
                                                                  const double *__restrict__ xin, const double *__restrict__ q,
                                                                  const double *__restrict__ z, double alpha, double *__restrict__ xout,
                                                                  double *__restrict__ znext, double *__restrict__ qnext,
                                                                  double *__restrict__ partials)
{
    constexpr int NT = TL * TW / 2, HC = TW / 2;
    constexpr int ZS = TW + 4, U1S = TW + 2, U2S = TW;
    constexpr int ZT = (TL + 2) * ZS, U1T = TL * U1S, U2T = (TL + 1) * U2S;
    constexpr int NH = 2 * TW + 2 * TL;
    __shared__ __align__(16) double zs[2 * ZT];
    __shared__ __align__(16) double u1s[3 * U1T];
    __shared__ __align__(16) double u2s[3 * U2T];
    __shared__ double tab[16];
    __shared__ double red[NT / 64];
    const int tid = (int)threadIdx.x;
    if (tid < 16)
        tab[tid] = sDt.v[tid];
    const int tl = tid / HC, tc = 2 * (tid % HC);
    const int xcd = (int)(blockIdx.x & 7);
    const int64_t items = (int64_t)tiles * nsegs, per_xcd = (items + 7) / 8;
    double acc[NSUM] = {0, 0, 0, 0, 0, 0};
    const int zo = (tl + 1) * ZS + 2 + tc, u1o = tl * U1S + 2 + tc, u2o = (tl + 1) * U2S + tc; // own positions in the tiles
    for (int64_t j = (int64_t)(blockIdx.x >> 3); j < per_xcd; j += (int64_t)(gridDim.x >> 3)) {
        const int64_t item = (int64_t)xcd * per_xcd + j;
        if (item >= items)
            break;
        const int32_t seg = (int32_t)(item / tiles), tile = (int32_t)(item % tiles);
        const int32_t l0 = (tile / tilesC) * TL, c0 = (tile % tilesC) * TW;
        const int32_t p0 = 1 + seg * seglen, p1 = (p0 + seglen < nplanes) ? p0 + seglen : nplanes;
        if (p0 >= p1)
            continue;
        const bool own = (l0 + tl < L) && (c0 + tc < nz);
        const uint32_t o = (uint32_t)((l0 + tl) * nz + c0 + tc);
        bool hv = false;
        uint32_t ho = 0;
        int hz = 0, hu = -1;
        bool hu_is_u2 = false;
        if (tid < TW) {
            hv = l0 >= 1 && c0 + tid < nz;
            ho = (uint32_t)((l0 - 1) * nz + c0 + tid);
            hz = 2 + tid;
            hu = tid;
            hu_is_u2 = true;
        } else if (tid < 2 * TW) {
            const int t = tid - TW;
            hv = l0 + TL < L && c0 + t < nz;
            ho = (uint32_t)((l0 + TL) * nz + c0 + t);
            hz = (TL + 1) * ZS + 2 + t;
        } else if (tid < 2 * TW + TL) {
            const int t = tid - 2 * TW;
            hv = c0 >= 1 && l0 + t < L;
            ho = (uint32_t)((l0 + t) * nz + c0 - 1);
            hz = (t + 1) * ZS + 1;
            hu = t * U1S + 1;
        } else if (tid < NH) {
            const int t = tid - 2 * TW - TL;
            hv = c0 + TW < nz && l0 + t < L;
            ho = (uint32_t)((l0 + t) * nz + c0 + TW);
            hz = (t + 1) * ZS + 2 + TW;
        }
        const bool ht = tid < NH;
        if (!hv)
            ho = o; // (never dereferenced; keeps the address in range)
        // one register for the halo role: z' slot | U slot << 12 | exists << 24 | has a U value << 25 | that value is U2's << 26
        const uint32_t hdesc = (uint32_t)hz | ((uint32_t)(hu >= 0 ? hu : 0) << 12) | ((uint32_t)hv << 24) | ((uint32_t)(hu >= 0) << 25) | ((uint32_t)hu_is_u2 << 26);
#define HZ ((int)(hdesc & 4095u))
#define HU ((int)((hdesc >> 12) & 4095u))
#define HV ((hdesc >> 24) & 1u)
#define HHU ((hdesc >> 25) & 1u)
#define HU2 ((hdesc >> 26) & 1u)
        // plane bases are uniform (scalar registers), the lane part is one 32-bit byte offset: "saddr + voffset" accesses
        uint32_t ob = (own ? o : 0u) * 8u, hb = ho * 8u; // rows outside the plane read row 0 of it: finite, never used
        auto PB = [&](const void *a, int32_t pl, int esz) -> const char * {
            return reinterpret_cast<const char *>(a) + (uint64_t)((int64_t)pl * d3) * (uint64_t)esz;
        };
        auto P2 = [&](const double *a, int32_t pl, bool) -> double2 { return *reinterpret_cast<const double2 *>(PB(a, pl, 8) + ob); };
        auto P2nt = [&](const double *a, int32_t pl, bool) -> double2 {
            const double *b = reinterpret_cast<const double *>(PB(a, pl, 8) + ob);
            return make_double2(__builtin_nontemporal_load(b), __builtin_nontemporal_load(b + 1));
        };
        auto C2 = [&](int32_t pl, bool) -> uint32_t { return (uint32_t) * reinterpret_cast<const uint16_t *>(PB(code, pl, 1) + (ob >> 3)); };
        auto H1 = [&](const double *a, int32_t pl) -> double { return *reinterpret_cast<const double *>(PB(a, pl, 8) + hb); };
        __syncthreads();
        // ---------------- prologue
        double2 Zm, Zc, Dc, Pc; // Pc: the -plane and diagonal terms of the centre plane's rows
        int zb = p0 & 1; // half of zs that holds the centre plane's z'
        int s0 = 0, s1 = 1, s2 = 2; // ring slots of planes p, p + 1, p + 2
        {
            const double2 xi = P2(xin, p0 - 1, own), qq = P2(q, p0 - 1, own), zz = P2(z, p0 - 1, own), dd = P2(dia, p0 - 1, own);
            const uint32_t cd = C2(p0 - 1, own);
            const RowOut a = update_row(xi.x, qq.x, zz.x, own ? dd.x : 1.0, tab[cd & 255u], alpha),
                         b = update_row(xi.y, qq.y, zz.y, own ? dd.y : 1.0, tab[cd >> 8], alpha);
            Zm = make_double2(a.zn, b.zn);
        }
        {
            const double2 xi = P2(xin, p0, own), qq = P2(q, p0, own), zz = P2(z, p0, own), dd = P2(dia, p0, own);
            const uint32_t cd = C2(p0, own);
            const RowOut a = update_row(xi.x, qq.x, zz.x, own ? dd.x : 1.0, tab[cd & 255u], alpha),
                         b = update_row(xi.y, qq.y, zz.y, own ? dd.y : 1.0, tab[cd >> 8], alpha);
            Zc = make_double2(a.zn, b.zn);
            Dc = dd;
            if (own) {
                *reinterpret_cast<double2 *>(const_cast<char *>(PB(xout, p0, 8)) + ob) = make_double2(a.xn, b.xn);
                *reinterpret_cast<double2 *>(const_cast<char *>(PB(znext, p0, 8)) + ob) = Zc;
                acc[0] += a.r * (a.mv * a.r) + b.r * (b.mv * b.r);
                acc[1] += a.r * a.r + b.r * b.r;
                acc[2] += a.c * a.zn + b.c * b.zn;
                acc[3] += a.c * a.c + b.c * b.c;
                acc[4] += a.h * a.h + b.h * b.h;
            }
        }
        double2 A3c = P2nt(u3, p0, own), A3n = P2nt(u3, p0 + 1, own);
        {
            const double2 A3m = P2nt(u3, p0 - 1, own);
            Pc = make_double2(A3m.x * Zm.x + Dc.x * Zc.x, A3m.y * Zm.y + Dc.y * Zc.y);
        }
        *reinterpret_cast<double2 *>(zs + zb * ZT + zo) = Zc;
        *reinterpret_cast<double2 *>(u1s + s0 * U1T + u1o) = P2nt(u1, p0, own);
        *reinterpret_cast<double2 *>(u2s + s0 * U2T + u2o) = P2nt(u2, p0, own);
        *reinterpret_cast<double2 *>(u1s + s1 * U1T + u1o) = P2nt(u1, p0 + 1, own);
        *reinterpret_cast<double2 *>(u2s + s1 * U2T + u2o) = P2nt(u2, p0 + 1, own);
        if (ht) {
            double zn = 0.0, ua = 0.0, ub = 0.0;
            if (hv) {
                const RowOut a = update_row(H1(xin, p0), H1(q, p0), H1(z, p0), H1(dia, p0), tab[*reinterpret_cast<const uint8_t *>(PB(code, p0, 1) + (hb >> 3))], alpha);
                zn = a.zn;
                if (hu >= 0) {
                    ua = H1(hu_is_u2 ? u2 : u1, p0);
                    ub = H1(hu_is_u2 ? u2 : u1, p0 + 1);
                }
            }
            zs[zb * ZT + hz] = zn;
            if (hu >= 0) {
                (hu_is_u2 ? u2s + s0 * U2T : u1s + s0 * U1T)[hu] = ua;
                (hu_is_u2 ? u2s + s1 * U2T : u1s + s1 * U1T)[hu] = ub;
            }
        }
        double2 Xa = P2nt(xin, p0 + 1, own), Qa = P2nt(q, p0 + 1, own), Za = P2(z, p0 + 1, own);
        uint32_t Ca = C2(p0 + 1, own);
        __syncthreads();
        for (int32_t p = p0; p < p1; p++) {
            // (opaque to the optimiser, or it keeps one 64-bit lane address per stream alive across the loop instead of
            // scalar plane base + this one offset)
            asm volatile("" : "+v"(ob), "+v"(hb));
            // ---- L: own streams of plane p + 2, halo streams of plane p + 1
            const bool more = p + 2 <= p1, inseg = p + 1 < p1;
            double2 Xb = make_double2(0.0, 0.0), Qb = Xb, Zb = Xb, V1b = Xb, V2b = Xb, A3b = Xb;
            uint32_t Cb = 0;
            double hx = 0.0, hq = 0.0, hzv = 0.0, hd = 1.0, hub = 0.0;
            uint32_t hc = 0;
            if (more) {
                Xb = P2nt(xin, p + 2, own);
                Qb = P2nt(q, p + 2, own);
                Zb = P2(z, p + 2, own);
                Cb = C2(p + 2, own);
                V1b = P2nt(u1, p + 2, own);
                V2b = P2nt(u2, p + 2, own);
                A3b = P2nt(u3, p + 2, own);
            }
            if (!(DBG & 1) && ht && HV) {
                if (inseg) {
                    hx = H1(xin, p + 1);
                    hq = H1(q, p + 1);
                    hzv = H1(z, p + 1);
                    hd = H1(dia, p + 1);
                    hc = *reinterpret_cast<const uint8_t *>(PB(code, p + 1, 1) + (hb >> 3));
                }
                if (more && HHU)
                    hub = H1(HU2 ? u2 : u1, p + 2);
            }
            // ---- U: update of plane p + 1
            const double *u1n = u1s + s1 * U1T, *u2n = u2s + s1 * U2T;
            double2 Dn;
            if (inseg) {
                const double2 V1n = *reinterpret_cast<const double2 *>(u1n + u1o), V2n = *reinterpret_cast<const double2 *>(u2n + u2o);
                const double v1m0n = u1n[u1o - 1];
                const double2 V2mn = *reinterpret_cast<const double2 *>(u2n + u2o - U2S);
                double so = A3c.x + V2mn.x;
                so += v1m0n;
                so += A3n.x;
                so += V2n.x;
                so += V1n.x;
                Dn.x = -so + tab[Ca & 255u];
                so = A3c.y + V2mn.y;
                so += V1n.x;
                so += A3n.y;
                so += V2n.y;
                so += V1n.y;
                Dn.y = -so + tab[Ca >> 8];
            } else
                Dn = P2(dia, p + 1, own);
            const RowOut ua = update_row<(DBG & 2) != 0>(Xa.x, Qa.x, Za.x, own ? Dn.x : 1.0, tab[Ca & 255u], alpha),
                         ub = update_row<(DBG & 2) != 0>(Xa.y, Qa.y, Za.y, own ? Dn.y : 1.0, tab[Ca >> 8], alpha);
            const double2 Zn = make_double2(ua.zn, ub.zn);
            const double2 Zc = *reinterpret_cast<const double2 *>(zs + zb * ZT + zo);
            const double2 Pn = make_double2(A3c.x * Zc.x + Dn.x * Zn.x, A3c.y * Zc.y + Dn.y * Zn.y);
            if (inseg) {
                if (own) {
                    if (!(DBG & 8)) {
                    double *xo = reinterpret_cast<double *>(const_cast<char *>(PB(xout, p + 1, 8)) + ob);
                    __builtin_nontemporal_store(ua.xn, xo);
                    __builtin_nontemporal_store(ub.xn, xo + 1);
                    *reinterpret_cast<double2 *>(const_cast<char *>(PB(znext, p + 1, 8)) + ob) = Zn;
                    }
                    acc[0] += ua.r * (ua.mv * ua.r) + ub.r * (ub.mv * ub.r);
                    acc[1] += ua.r * ua.r + ub.r * ub.r;
                    acc[2] += ua.c * ua.zn + ub.c * ub.zn;
                    acc[3] += ua.c * ua.c + ub.c * ub.c;
                    acc[4] += ua.h * ua.h + ub.h * ub.h;
                }
                *reinterpret_cast<double2 *>(zs + (zb ^ 1) * ZT + zo) = Zn;
            }
            // ---- S: q'(p)
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
                if (own && !(DBG & 4)) {
                    double *qo = reinterpret_cast<double *>(const_cast<char *>(PB(qnext, p, 8)) + ob);
                    __builtin_nontemporal_store(t0, qo);
                    __builtin_nontemporal_store(t1, qo + 1);
                    acc[5] += Zc.x * t0 + Zc.y * t1;
                }
            }
            // ---- W: halo z' of plane p + 1, U tiles of plane p + 2
            if (ht && inseg) {
                double hzn = 0.0;
                if (HV) {
                    const RowOut a = update_row(hx, hq, hzv, hd, tab[hc], alpha);
                    hzn = a.zn;
                }
                zs[(zb ^ 1) * ZT + HZ] = hzn;
            }
            if (more) {
                *reinterpret_cast<double2 *>(u1s + s2 * U1T + u1o) = V1b;
                *reinterpret_cast<double2 *>(u2s + s2 * U2T + u2o) = V2b;
                if (ht && HHU)
                    (HU2 ? u2s + s2 * U2T : u1s + s2 * U1T)[HU] = hub;
            }
            __syncthreads();
            Pc = Pn;
            A3c = A3n;
            A3n = A3b;
            Xa = Xb;
            Qa = Qb;
            Za = Zb;
            Ca = Cb;
            zb ^= 1;
            const int st = s0;
            s0 = s1;
            s1 = s2;
            s2 = st;
        }
    }
    for (int k = 0; k < NSUM; k++) {
        double v = acc[k];
        for (int off = 32; off > 0; off >>= 1)
            v += __shfl_xor(v, off, 64);
        __syncthreads();
        if ((tid & 63) == 0)
            red[tid >> 6] = v;
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
            for (int w = 0; w < NT / 64; w++)
                t += red[w];
            partials[(size_t)k * gridDim.x + blockIdx.x] = t;
        }
    }
}

#undef HZ
#undef HU
#undef HV
#undef HHU
#undef HU2
// ------------------------------------------------------------------ v3: the v-form
// The residual of an implicit step's system at its start state is rho(x) = b - A x whatever the time step, so across
// one-iteration steps  rho' = rho - alpha A z  with z = M^-1 rho: the next direction is  z' = z + alpha v  with
// v = -M^-1 (A z) = -M^-1 (q - sigma D z), which the launch that formed q = (A + sigma D) z can store INSTEAD of q.  A halo
// row then needs two streams (z, v) and one FMA, no diagonal, no storage code, no division.  Own rows: x, z, v in;
// x_out, z', v' out; diagonal from the arms; r = d z' - sigma D (x_out - x) for the convergence test of the finished step.
// As above, with what the first version kept in registers moved into LDS so that two blocks fit a CU: the z' tile is double
// buffered (the update of plane p + 1 writes the other half while the product of plane p reads this one), the U1 / U2 tiles
// live in a ring of three plane slots (p and p + 1 are read, p + 2 is written when its loads land), the halo rows' streams
// are loaded in the step that uses them.  Addresses are "plane base (uniform) + 32-bit in-plane offset".
template <int TL, int TW, int WPS, int DBG = 0> // DBG (wrong results): 1 no halo rows, 2 reciprocal by v_rcp_f64, 4 no q' store, 8 no x_out / z' store
__global__ __launch_bounds__(TL *TW / 2, WPS) void fused_step_v3(int32_t nz, int32_t L, int32_t d3, int32_t nplanes, int32_t seglen, int32_t tilesC,
                                                                  int32_t tiles, int32_t nsegs, const double *__restrict__ u1,
                                                                  const double *__restrict__ u2, const double *__restrict__ u3,
                                                                  const double *__restrict__ dia, const uint8_t *__restrict__ code, Tab sDt,
                                                                  const double *__restrict__ xin, const double *__restrict__ q,
                                                                  const double *__restrict__ z, double alpha, double *__restrict__ xout,
                                                                  double *__restrict__ znext, double *__restrict__ qnext,
                                                                  double *__restrict__ partials)
{
    constexpr int NT = TL * TW / 2, HC = TW / 2;
    constexpr int ZS = TW + 4, U1S = TW + 2, U2S = TW;
    constexpr int ZT = (TL + 2) * ZS, U1T = TL * U1S, U2T = (TL + 1) * U2S;
    constexpr int NH = 2 * TW + 2 * TL;
    __shared__ __align__(16) double zs[2 * ZT];
    __shared__ __align__(16) double u1s[3 * U1T];
    __shared__ __align__(16) double u2s[3 * U2T];
    __shared__ double tab[16];
    __shared__ double red[NT / 64];
    const int tid = (int)threadIdx.x;
    if (tid < 16)
        tab[tid] = sDt.v[tid];
    const int tl = tid / HC, tc = 2 * (tid % HC);
    const int xcd = (int)(blockIdx.x & 7);
    const int64_t items = (int64_t)tiles * nsegs, per_xcd = (items + 7) / 8;
    double acc[NSUM] = {0, 0, 0, 0, 0, 0};
    const int zo = (tl + 1) * ZS + 2 + tc, u1o = tl * U1S + 2 + tc, u2o = (tl + 1) * U2S + tc; // own positions in the tiles
    for (int64_t j = (int64_t)(blockIdx.x >> 3); j < per_xcd; j += (int64_t)(gridDim.x >> 3)) {
        const int64_t item = (int64_t)xcd * per_xcd + j;
        if (item >= items)
            break;
        const int32_t seg = (int32_t)(item / tiles), tile = (int32_t)(item % tiles);
        const int32_t l0 = (tile / tilesC) * TL, c0 = (tile % tilesC) * TW;
        const int32_t p0 = 1 + seg * seglen, p1 = (p0 + seglen < nplanes) ? p0 + seglen : nplanes;
        if (p0 >= p1)
            continue;
        const bool own = (l0 + tl < L) && (c0 + tc < nz);
        const uint32_t o = (uint32_t)((l0 + tl) * nz + c0 + tc);
        bool hv = false;
        uint32_t ho = 0;
        int hz = 0, hu = -1;
        bool hu_is_u2 = false;
        if (tid < TW) {
            hv = l0 >= 1 && c0 + tid < nz;
            ho = (uint32_t)((l0 - 1) * nz + c0 + tid);
            hz = 2 + tid;
            hu = tid;
            hu_is_u2 = true;
        } else if (tid < 2 * TW) {
            const int t = tid - TW;
            hv = l0 + TL < L && c0 + t < nz;
            ho = (uint32_t)((l0 + TL) * nz + c0 + t);
            hz = (TL + 1) * ZS + 2 + t;
        } else if (tid < 2 * TW + TL) {
            const int t = tid - 2 * TW;
            hv = c0 >= 1 && l0 + t < L;
            ho = (uint32_t)((l0 + t) * nz + c0 - 1);
            hz = (t + 1) * ZS + 1;
            hu = t * U1S + 1;
        } else if (tid < NH) {
            const int t = tid - 2 * TW - TL;
            hv = c0 + TW < nz && l0 + t < L;
            ho = (uint32_t)((l0 + t) * nz + c0 + TW);
            hz = (t + 1) * ZS + 2 + TW;
        }
        const bool ht = tid < NH;
        if (!hv)
            ho = o; // (never dereferenced; keeps the address in range)
        // one register for the halo role: z' slot | U slot << 12 | exists << 24 | has a U value << 25 | that value is U2's << 26
        const uint32_t hdesc = (uint32_t)hz | ((uint32_t)(hu >= 0 ? hu : 0) << 12) | ((uint32_t)hv << 24) | ((uint32_t)(hu >= 0) << 25) | ((uint32_t)hu_is_u2 << 26);
#define HZ ((int)(hdesc & 4095u))
#define HU ((int)((hdesc >> 12) & 4095u))
#define HV ((hdesc >> 24) & 1u)
#define HHU ((hdesc >> 25) & 1u)
#define HU2 ((hdesc >> 26) & 1u)
        // plane bases are uniform (scalar registers), the lane part is one 32-bit byte offset: "saddr + voffset" accesses
        uint32_t ob = (own ? o : 0u) * 8u, hb = ho * 8u; // rows outside the plane read row 0 of it: finite, never used
        auto PB = [&](const void *a, int32_t pl, int esz) -> const char * {
            return reinterpret_cast<const char *>(a) + (uint64_t)((int64_t)pl * d3) * (uint64_t)esz;
        };
        auto P2 = [&](const double *a, int32_t pl, bool) -> double2 { return *reinterpret_cast<const double2 *>(PB(a, pl, 8) + ob); };
        auto P2nt = [&](const double *a, int32_t pl, bool) -> double2 {
            const double *b = reinterpret_cast<const double *>(PB(a, pl, 8) + ob);
            return make_double2(__builtin_nontemporal_load(b), __builtin_nontemporal_load(b + 1));
        };
        auto C2 = [&](int32_t pl, bool) -> uint32_t { return (uint32_t) * reinterpret_cast<const uint16_t *>(PB(code, pl, 1) + (ob >> 3)); };
        auto H1 = [&](const double *a, int32_t pl) -> double { return *reinterpret_cast<const double *>(PB(a, pl, 8) + hb); };
        __syncthreads();
        // ---------------- prologue
        double2 Zm, Zc, Dc, Pc, Mc; // Mc: M^-1 of the centre plane's rows; Pc: the -plane and diagonal terms of the centre plane's rows
        uint32_t Cc = 0; // storage codes of the centre plane's rows
        int zb = p0 & 1; // half of zs that holds the centre plane's z'
        int s0 = 0, s1 = 1, s2 = 2; // ring slots of planes p, p + 1, p + 2
        {
            const double2 vv = P2(q, p0 - 1, own), zz = P2(z, p0 - 1, own);
            Zm = make_double2(zz.x + alpha * vv.x, zz.y + alpha * vv.y);
        }
        {
            const double2 xi = P2(xin, p0, own), qq = P2(q, p0, own), zz = P2(z, p0, own), dd = P2(dia, p0, own);
            const uint32_t cd = C2(p0, own);
            const VRow a = vrow(xi.x, zz.x, qq.x, own ? dd.x : 1.0, tab[cd & 255u], alpha), b = vrow(xi.y, zz.y, qq.y, own ? dd.y : 1.0, tab[cd >> 8], alpha);
            Zc = make_double2(a.zn, b.zn);
            Dc = dd;
            Mc = make_double2(a.mv, b.mv);
            Cc = cd;
            if (own) {
                *reinterpret_cast<double2 *>(const_cast<char *>(PB(xout, p0, 8)) + ob) = make_double2(a.xn, b.xn);
                *reinterpret_cast<double2 *>(const_cast<char *>(PB(znext, p0, 8)) + ob) = Zc;
                acc[0] += a.r * (a.mv * a.r) + b.r * (b.mv * b.r);
                acc[1] += a.r * a.r + b.r * b.r;
                acc[2] += a.c * a.zn + b.c * b.zn;
                acc[3] += a.c * a.c + b.c * b.c;
                acc[4] += a.h * a.h + b.h * b.h;
            }
        }
        double2 A3c = P2nt(u3, p0, own), A3n = P2nt(u3, p0 + 1, own);
        {
            const double2 A3m = P2nt(u3, p0 - 1, own);
            Pc = make_double2(A3m.x * Zm.x + Dc.x * Zc.x, A3m.y * Zm.y + Dc.y * Zc.y);
        }
        *reinterpret_cast<double2 *>(zs + zb * ZT + zo) = Zc;
        *reinterpret_cast<double2 *>(u1s + s0 * U1T + u1o) = P2nt(u1, p0, own);
        *reinterpret_cast<double2 *>(u2s + s0 * U2T + u2o) = P2nt(u2, p0, own);
        *reinterpret_cast<double2 *>(u1s + s1 * U1T + u1o) = P2nt(u1, p0 + 1, own);
        *reinterpret_cast<double2 *>(u2s + s1 * U2T + u2o) = P2nt(u2, p0 + 1, own);
        if (ht) {
            double zn = 0.0, ua = 0.0, ub = 0.0;
            if (hv) {
                zn = H1(z, p0) + alpha * H1(q, p0);
                if (hu >= 0) {
                    ua = H1(hu_is_u2 ? u2 : u1, p0);
                    ub = H1(hu_is_u2 ? u2 : u1, p0 + 1);
                }
            }
            zs[zb * ZT + hz] = zn;
            if (hu >= 0) {
                (hu_is_u2 ? u2s + s0 * U2T : u1s + s0 * U1T)[hu] = ua;
                (hu_is_u2 ? u2s + s1 * U2T : u1s + s1 * U1T)[hu] = ub;
            }
        }
        double2 Xa = P2nt(xin, p0 + 1, own), Qa = P2nt(q, p0 + 1, own), Za = P2(z, p0 + 1, own);
        uint32_t Ca = C2(p0 + 1, own);
        __syncthreads();
        for (int32_t p = p0; p < p1; p++) {
            // (opaque to the optimiser, or it keeps one 64-bit lane address per stream alive across the loop instead of
            // scalar plane base + this one offset)
            asm volatile("" : "+v"(ob), "+v"(hb));
            // ---- L: own streams of plane p + 2, halo streams of plane p + 1
            const bool more = p + 2 <= p1, inseg = p + 1 < p1;
            double2 Xb = make_double2(0.0, 0.0), Qb = Xb, Zb = Xb, V1b = Xb, V2b = Xb, A3b = Xb;
            uint32_t Cb = 0;
            double hq = 0.0, hzv = 0.0, hub = 0.0;
            if (more) {
                Xb = P2nt(xin, p + 2, own);
                Qb = P2nt(q, p + 2, own);
                Zb = P2(z, p + 2, own);
                Cb = C2(p + 2, own);
                V1b = P2nt(u1, p + 2, own);
                V2b = P2nt(u2, p + 2, own);
                A3b = P2nt(u3, p + 2, own);
            }
            if (!(DBG & 1) && ht && HV) {
                if (inseg) {
                    hq = H1(q, p + 1);
                    hzv = H1(z, p + 1);
                }
                if (more && HHU)
                    hub = H1(HU2 ? u2 : u1, p + 2);
            }
            // ---- U: update of plane p + 1
            const double *u1n = u1s + s1 * U1T, *u2n = u2s + s1 * U2T;
            double2 Dn;
            if (inseg) {
                const double2 V1n = *reinterpret_cast<const double2 *>(u1n + u1o), V2n = *reinterpret_cast<const double2 *>(u2n + u2o);
                const double v1m0n = u1n[u1o - 1];
                const double2 V2mn = *reinterpret_cast<const double2 *>(u2n + u2o - U2S);
                double so = A3c.x + V2mn.x;
                so += v1m0n;
                so += A3n.x;
                so += V2n.x;
                so += V1n.x;
                Dn.x = -so + tab[Ca & 255u];
                so = A3c.y + V2mn.y;
                so += V1n.x;
                so += A3n.y;
                so += V2n.y;
                so += V1n.y;
                Dn.y = -so + tab[Ca >> 8];
            } else
                Dn = P2(dia, p + 1, own);
            const VRow ua = vrow(Xa.x, Za.x, Qa.x, own ? Dn.x : 1.0, tab[Ca & 255u], alpha), ub = vrow(Xa.y, Za.y, Qa.y, own ? Dn.y : 1.0, tab[Ca >> 8], alpha);
            const double2 Zn = make_double2(ua.zn, ub.zn);
            const double2 Zc = *reinterpret_cast<const double2 *>(zs + zb * ZT + zo);
            const double2 Pn = make_double2(A3c.x * Zc.x + Dn.x * Zn.x, A3c.y * Zc.y + Dn.y * Zn.y);
            if (inseg) {
                if (own) {
                    if (!(DBG & 8)) {
                    double *xo = reinterpret_cast<double *>(const_cast<char *>(PB(xout, p + 1, 8)) + ob);
                    __builtin_nontemporal_store(ua.xn, xo);
                    __builtin_nontemporal_store(ub.xn, xo + 1);
                    *reinterpret_cast<double2 *>(const_cast<char *>(PB(znext, p + 1, 8)) + ob) = Zn;
                    }
                    acc[0] += ua.r * (ua.mv * ua.r) + ub.r * (ub.mv * ub.r);
                    acc[1] += ua.r * ua.r + ub.r * ub.r;
                    acc[2] += ua.c * ua.zn + ub.c * ub.zn;
                    acc[3] += ua.c * ua.c + ub.c * ub.c;
                    acc[4] += ua.h * ua.h + ub.h * ub.h;
                }
                *reinterpret_cast<double2 *>(zs + (zb ^ 1) * ZT + zo) = Zn;
            }
            // ---- S: q'(p)
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
                if (own && !(DBG & 4)) {
                    double *qo = reinterpret_cast<double *>(const_cast<char *>(PB(qnext, p, 8)) + ob);
                    *reinterpret_cast<double2 *>(qo) = make_double2(-(Mc.x * (t0 - tab[Cc & 255u] * Zc.x)), -(Mc.y * (t1 - tab[Cc >> 8] * Zc.y)));
                    acc[5] += Zc.x * t0 + Zc.y * t1;
                }
            }
            // ---- W: halo z' of plane p + 1, U tiles of plane p + 2
            if (ht && inseg) {
                double hzn = 0.0;
                if (HV)
                    hzn = hzv + alpha * hq;
                zs[(zb ^ 1) * ZT + HZ] = hzn;
            }
            if (more) {
                *reinterpret_cast<double2 *>(u1s + s2 * U1T + u1o) = V1b;
                *reinterpret_cast<double2 *>(u2s + s2 * U2T + u2o) = V2b;
                if (ht && HHU)
                    (HU2 ? u2s + s2 * U2T : u1s + s2 * U1T)[HU] = hub;
            }
            __syncthreads();
            Pc = Pn;
            Mc = make_double2(ua.mv, ub.mv);
            Cc = Ca;
            A3c = A3n;
            A3n = A3b;
            Xa = Xb;
            Qa = Qb;
            Za = Zb;
            Ca = Cb;
            zb ^= 1;
            const int st = s0;
            s0 = s1;
            s1 = s2;
            s2 = st;
        }
    }
    for (int k = 0; k < NSUM; k++) {
        double v = acc[k];
        for (int off = 32; off > 0; off >>= 1)
            v += __shfl_xor(v, off, 64);
        __syncthreads();
        if ((tid & 63) == 0)
            red[tid >> 6] = v;
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
            for (int w = 0; w < NT / 64; w++)
                t += red[w];
            partials[(size_t)k * gridDim.x + blockIdx.x] = t;
        }
    }
}

#undef HZ
#undef HU
#undef HV
#undef HHU
#undef HU2
// ------------------------------------------------------------------ host
static double *dalloc(size_t n, size_t front)
{
    double *p = nullptr;
    CK(hipMalloc(&p, (n + 2 * front) * sizeof(double)));
    CK(hipMemset(p, 0, (n + 2 * front) * sizeof(double)));
    return p + front;
}
__global__ void fill_kernel(int64_t n, int32_t nz, int32_t L, int32_t d3, int32_t P, double *u1, double *u2, double *u3, double *xin, double *q,
                            double *z, uint8_t *code, uint64_t seed)
{
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n)
        return;
    auto rnd = [&](uint64_t k) {
        uint64_t h = (uint64_t)r * 0x9E3779B97F4A7C15ull + k * 0xBF58476D1CE4E5B9ull + seed;
        h ^= h >> 31;
        h *= 0x94D049BB133111EBull;
        h ^= h >> 29;
        return (double)(h >> 11) * (1.0 / 9007199254740992.0);
    };
    const int32_t o = (int32_t)(r % d3), c = o % nz, l = o / nz;
    const int32_t p = (int32_t)(r / d3);
    u1[r] = c == nz - 1 ? 0.0 : -(0.5 + rnd(1));
    u2[r] = l == L - 1 ? 0.0 : -(0.5 + rnd(2));
    u3[r] = p == P - 1 ? 0.0 : -(50.0 + 100.0 * rnd(3));
    xin[r] = 1000.0 + rnd(4);
    q[r] = rnd(5) - 0.5;
    z[r] = 1e-3 * (rnd(6) - 0.5);
    code[r] = (uint8_t)((c == 0 || c == nz - 1) + (l == 0 || l == L - 1));
}

template <int TL, int TW, int WPS, int VAR = 1, int DBG = 0>
static void run_fused(const char *name, int P, int L, int nz, int reps, int blocks_per_cu, int nsegs_req, const double *u1, const double *u2,
                      const double *u3, const double *dia, const uint8_t *code, Tab sD, const double *xin, const double *q, const double *z,
                      double alpha, double *xout, double *zn, double *qn, const double *xout_ref, const double *zn_ref, const double *qn_ref)
{
    const int d3 = L * nz;
    const int64_t n = (int64_t)P * d3;
    const int tilesC = (nz + TW - 1) / TW, tilesL = (L + TL - 1) / TL, tiles = tilesC * tilesL;
    const int nplanes = P - 1; // planes [1, P - 1)
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int resident = prop.multiProcessorCount * blocks_per_cu;
    int best = 1;
    if (nsegs_req > 0)
        best = nsegs_req;
    else {
        double bestc = 1e300;
        for (int s = 1; s <= 64; s++) {
            const int seglen = (nplanes - 1 + s - 1) / s;
            const int64_t items = (int64_t)tiles * s;
            const int64_t rounds = (items + resident - 1) / resident;
            const double cost = (double)rounds * (seglen + 3);
            if (cost < bestc) {
                bestc = cost;
                best = s;
            }
        }
    }
    const int nsegs = best, seglen = (nplanes - 1 + nsegs - 1) / nsegs;
    const int grid = resident;
    double *partials;
    CK(hipMalloc(&partials, (size_t)NSUM * grid * sizeof(double)));
    CK(hipMemset(xout, 0, n * sizeof(double)));
    CK(hipMemset(zn, 0, n * sizeof(double)));
    CK(hipMemset(qn, 0, n * sizeof(double)));
    auto launch = [&]() {
        if (VAR == 3)
            hipLaunchKernelGGL((fused_step_v3<TL, TW, WPS, DBG>), dim3(grid), dim3(TL * TW / 2), 0, 0, nz, L, d3, nplanes, seglen, tilesC, tiles, nsegs, u1, u2, u3,
                               dia, code, sD, xin, q, z, alpha, xout, zn, qn, partials);
        else if (VAR == 2)
            hipLaunchKernelGGL((fused_step_v2<TL, TW, WPS, DBG>), dim3(grid), dim3(TL * TW / 2), 0, 0, nz, L, d3, nplanes, seglen, tilesC, tiles, nsegs, u1, u2, u3,
                               dia, code, sD, xin, q, z, alpha, xout, zn, qn, partials);
        else
            hipLaunchKernelGGL((fused_step_kernel<TL, TW, WPS>), dim3(grid), dim3(TL * TW / 2), 0, 0, nz, L, d3, nplanes, seglen, tilesC, tiles, nsegs, u1, u2, u3,
                               dia, code, sD, xin, q, z, alpha, xout, zn, qn, partials);
    };
    launch();
    CK(hipDeviceSynchronize());
    // check planes [1, P - 1)
    {
        const size_t m = (size_t)(nplanes - 1) * d3;
        std::vector<double> a(m), b(m);
        const char *names[3] = {"x_out", "z'", "q'"};
        const double *got[3] = {xout, zn, qn}, *ref[3] = {xout_ref, zn_ref, qn_ref};
        for (int k = 0; k < 3; k++) {
            CK(hipMemcpy(a.data(), got[k] + d3, m * sizeof(double), hipMemcpyDeviceToHost));
            CK(hipMemcpy(b.data(), ref[k] + d3, m * sizeof(double), hipMemcpyDeviceToHost));
            size_t bad = 0, first = 0;
            double maxrel = 0.0;
            for (size_t i = 0; i < m; i++)
                if (memcmp(&a[i], &b[i], 8) != 0) {
                    if (!bad)
                        first = i;
                    bad++;
                    const double rel = fabs(a[i] - b[i]) / (fabs(b[i]) + 1e-300);
                    if (rel > maxrel)
                        maxrel = rel;
                }
            printf("  %s %-6s: %zu of %zu differ (first at %zu, max rel %.2e)\n", name, names[k], bad, m, first, maxrel);
        }
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<float> ts;
    for (int rep = 0; rep < reps; rep++) {
        CK(hipEventRecord(e0, 0));
        launch();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    const double rows = (double)(nplanes - 1) * d3;
    printf("  %s: tiles %d x %d, %d segs of %d planes, grid %d: median %.4f ms (min %.4f) -> %.2f TB/s on 73 B/row, %.3e rows/s\n", name, tilesL, tilesC,
           nsegs, seglen, grid, ts[ts.size() / 2], ts[0], rows * 73 / ts[ts.size() / 2] / 1e9, rows / ts[ts.size() / 2] * 1e3);
    CK(hipFree(partials));
}

int main(int argc, char **argv)
{
    const int P = argc > 1 ? atoi(argv[1]) : 462, L = argc > 2 ? atoi(argv[2]) : 464, nz = argc > 3 ? atoi(argv[3]) : 464;
    const int reps = argc > 4 ? atoi(argv[4]) : 20;
    const int bpc = argc > 5 ? atoi(argv[5]) : 2, nsegs = argc > 6 ? atoi(argv[6]) : 0;
    const int d3 = L * nz;
    const int64_t n = (int64_t)P * d3;
    const size_t front = (size_t)d3 + 256;
    printf("fused step prototype: %d planes x %d lines x %d = %lld rows\n", P, L, nz, (long long)n);
    double *u1 = dalloc(n, front), *u2 = dalloc(n, front), *u3 = dalloc(n, front), *dia = dalloc(n, front);
    double *xin = dalloc(n, front), *q = dalloc(n, front), *z = dalloc(n, front);
    double *xout = dalloc(n, front), *zn = dalloc(n, front), *qn = dalloc(n, front);
    double *xout_ref = dalloc(n, front), *zn_ref = dalloc(n, front), *qn_ref = dalloc(n, front), *minv = dalloc(n, front);
    uint8_t *code;
    CK(hipMalloc(&code, n + 2 * front));
    CK(hipMemset(code, 0, n + 2 * front));
    code += front;
    Tab sD;
    for (int k = 0; k < 16; k++)
        sD.v[k] = 0.0;
    sD.v[0] = 1.0 / 60.0 * 0.1 * 0.8;
    sD.v[1] = sD.v[0] / 2;
    sD.v[2] = sD.v[0] / 4;
    const double alpha = 0.731;
    const int g = (int)((n + 255) / 256);
    hipLaunchKernelGGL(fill_kernel, dim3(g), dim3(256), 0, 0, n, nz, L, d3, P, u1, u2, u3, xin, q, z, code, 12345ull);
    hipLaunchKernelGGL(ref_diag, dim3(g), dim3(256), 0, 0, n, nz, d3, u1, u2, u3, code, sD, dia);
    hipLaunchKernelGGL(ref_update, dim3(g), dim3(256), 0, 0, n, xin, q, z, dia, code, sD, alpha, xout_ref, zn_ref);
    hipLaunchKernelGGL(ref_spmv<false>, dim3(g), dim3(256), 0, 0, n, nz, d3, u1, u2, u3, dia, zn_ref, qn_ref);
    double *qn_ref2 = dalloc(n, front);
    hipLaunchKernelGGL(ref_spmv<true>, dim3(g), dim3(256), 0, 0, n, nz, d3, u1, u2, u3, dia, zn_ref, qn_ref2);
    CK(hipDeviceSynchronize());
    // the box's streaming rate on today's K2S shape
    {
        double *part;
        CK(hipMalloc(&part, 1024 * sizeof(double)));
        CK(hipMemset(part, 0, 1024 * sizeof(double)));
        CK(hipMemset(minv - front, 0, 8)); // (touch)
        hipLaunchKernelGGL(ref_update, dim3(g), dim3(256), 0, 0, n, xin, q, z, dia, code, sD, 0.0, minv, minv); // minv := something positive
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        std::vector<float> ts;
        for (int rep = 0; rep < reps; rep++) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k2s_like, dim3(2048), dim3(256), 0, 0, n, xin, q, z, dia, code, sD, alpha, xout, zn, part);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            ts.push_back(ms);
        }
        std::sort(ts.begin(), ts.end());
        printf("  K2S-shaped stream (49 B/row): median %.4f ms -> %.2f TB/s\n", ts[ts.size() / 2], (double)n * 49 / ts[ts.size() / 2] / 1e9);
    }
    // v-form references: x_out, z' = z + alpha v (the q array plays v), v' = -M^-1 (q' - sigma D z')
    double *xout_v = dalloc(n, front), *zn_v = dalloc(n, front), *qtmp = dalloc(n, front), *vn_v = dalloc(n, front);
    hipLaunchKernelGGL(ref_update_v, dim3(g), dim3(256), 0, 0, n, xin, z, q, alpha, xout_v, zn_v);
    hipLaunchKernelGGL(ref_spmv<true>, dim3(g), dim3(256), 0, 0, n, nz, d3, u1, u2, u3, dia, zn_v, qtmp);
    hipLaunchKernelGGL(ref_q_to_v, dim3(g), dim3(256), 0, 0, n, qtmp, zn_v, dia, code, sD, vn_v);
    CK(hipDeviceSynchronize());
    const int mask = argc > 7 ? atoi(argv[7]) : 0xffff;
#define RUN(bit, ...)                                                                                                               \
    if (mask & (1 << (bit)))                                                                                                        \
    run_fused<__VA_ARGS__>
    RUN(0, 16, 64, 2)("v1 16x64 189 VGPRs, 1 block/CU", P, L, nz, reps, 1, nsegs, u1, u2, u3, dia, code, sD, xin, q, z, alpha, xout, zn, qn, xout_ref, zn_ref, qn_ref);
    RUN(1, 16, 64, 4, 2)("v2 16x64, 2 blocks/CU", P, L, nz, reps, 2, nsegs, u1, u2, u3, dia, code, sD, xin, q, z, alpha, xout, zn, qn, xout_ref, zn_ref, qn_ref2);
    RUN(2, 8, 128, 4, 2)("v2 8x128, 2 blocks/CU", P, L, nz, reps, 2, nsegs, u1, u2, u3, dia, code, sD, xin, q, z, alpha, xout, zn, qn, xout_ref, zn_ref, qn_ref2);
    RUN(3, 16, 64, 2, 2)("v2 16x64, 1 block/CU", P, L, nz, reps, 1, nsegs, u1, u2, u3, dia, code, sD, xin, q, z, alpha, xout, zn, qn, xout_ref, zn_ref, qn_ref2);
    RUN(4, 8, 128, 4, 2, 1)("v2 8x128 DBG no halo", P, L, nz, reps, 2, nsegs, u1, u2, u3, dia, code, sD, xin, q, z, alpha, xout, zn, qn, xout_ref, zn_ref, qn_ref2);
    RUN(5, 8, 128, 4, 2, 2)("v2 8x128 DBG v_rcp", P, L, nz, reps, 2, nsegs, u1, u2, u3, dia, code, sD, xin, q, z, alpha, xout, zn, qn, xout_ref, zn_ref, qn_ref2);
    RUN(6, 8, 128, 4, 2, 4)("v2 8x128 DBG no q' store", P, L, nz, reps, 2, nsegs, u1, u2, u3, dia, code, sD, xin, q, z, alpha, xout, zn, qn, xout_ref, zn_ref, qn_ref2);
    RUN(7, 8, 128, 4, 2, 12)("v2 8x128 DBG no stores at all", P, L, nz, reps, 2, nsegs, u1, u2, u3, dia, code, sD, xin, q, z, alpha, xout, zn, qn, xout_ref, zn_ref, qn_ref2);
    RUN(8, 8, 128, 4, 2, 15)("v2 8x128 DBG no halo, rcp, no stores", P, L, nz, reps, 2, nsegs, u1, u2, u3, dia, code, sD, xin, q, z, alpha, xout, zn, qn, xout_ref, zn_ref, qn_ref2);
    RUN(10, 8, 128, 4, 3)("v3 8x128, 2 blocks/CU", P, L, nz, reps, 2, nsegs, u1, u2, u3, dia, code, sD, xin, q, z, alpha, xout, zn, qn, xout_v, zn_v, vn_v);
    RUN(11, 16, 64, 4, 3)("v3 16x64, 2 blocks/CU", P, L, nz, reps, 2, nsegs, u1, u2, u3, dia, code, sD, xin, q, z, alpha, xout, zn, qn, xout_v, zn_v, vn_v);
    RUN(12, 8, 128, 4, 3, 1)("v3 8x128 DBG no halo", P, L, nz, reps, 2, nsegs, u1, u2, u3, dia, code, sD, xin, q, z, alpha, xout, zn, qn, xout_v, zn_v, vn_v);
    RUN(13, 8, 128, 2, 3)("v3 8x128, 1 block/CU", P, L, nz, reps, 1, nsegs, u1, u2, u3, dia, code, sD, xin, q, z, alpha, xout, zn, qn, xout_v, zn_v, vn_v);
    return 0;
}
