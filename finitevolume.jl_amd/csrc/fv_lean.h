// Rows of a regulargrid operator formed on the fly ("lean" problems, FV_OPT_LEAN_SETUP): what assembleA would have put into row r of
// the CSR — columns ascending, every value the double the assembly kernel computes — from the grid's closed-form face list
// (fv_grid.hip: a cell emits its x-, y-, z-face in that order, cells in node order; /root/reference/src/grid.jl:72-105), the
// conductivities as the caller handed them over and the node map.  The set-up kernels of the solver's storage forms (sliced DIA,
// symmetric copy) read rows through this instead of rowptr / colind / vals, so that neither the face arrays, nor the incident
// lists, nor the CSR ever exist in HBM: 8e8 cells on one GPU where the int32 CSR ends at 3e8.
// Every floating-point expression below is one operation per statement (or under `fp contract(off)`): the including translation
// unit's contraction setting cannot change a bit.
#pragma once
#include "fv_internal.h"

// the conductance of the face cell c = (i1, i2, i3) emits in direction dir (0: +x, 1: +y, 2: +z; the face must exist):
// K[metaindex(f)] * aol[f] or exp(K[...]) * aol[f] (FiniteVolume.jl:83, :96; conductance_kernel), aol as regulargrid_kernel forms it
__device__ inline double grid_face_cond(const GridRows &g, int64_t c, int64_t i1, int64_t i2, int64_t i3, int dir)
{
#pragma clang fp contract(off)
    const int64_t n1 = g.n1, n2 = g.n2, n3 = g.n3;
    double areadx = g.dx, aready = g.dy, areadz = g.dz;
    if (i1 == 0 || i1 == n1 - 1)
        areadx *= 0.5;
    if (i2 == 0 || i2 == n2 - 1)
        aready *= 0.5;
    if (i3 == 0 || i3 == n3 - 1)
        areadz *= 0.5;
    const int64_t X = (i1 < n1 - 1) ? c : (n1 - 1) * n2 * n3;
    const int64_t Y = i1 * (n2 - 1) * n3 + (i2 < n2 - 1 ? i2 * n3 + i3 : (n2 - 1) * n3);
    const int64_t Z = (i1 * n2 + i2) * (n3 - 1) + (i3 < n3 - 1 ? i3 : n3 - 1);
    int64_t f = X + Y + Z;
    double aol;
    if (dir == 0) {
        const double a = aready * areadz;
        aol = a / g.dx;
    } else if (dir == 1) {
        f += (i1 < n1 - 1) ? 1 : 0;
        const double a = areadx * areadz;
        aol = a / g.dy;
    } else {
        f += ((i1 < n1 - 1) ? 1 : 0) + ((i2 < n2 - 1) ? 1 : 0);
        const double a = areadx * aready;
        aol = a / g.dz;
    }
    const int64_t m = g.meta ? g.meta[f] - 1 : (g.nK == 1 ? 0 : f);
    const double k = g.K[m];
    if (g.logt) {
        const double e = exp(k);
        return e * aol;
    }
    return k * aol;
}

struct GridRow {
    int len;
    int32_t off[7]; // column - row, ascending
    double val[7];
};

// the stored entries of free row r: (-plane, -line, -1, diagonal, +1, +line, +plane) where the neighbour exists and is free
__device__ inline void grid_row(const GridRows &g, int64_t r, GridRow &e, bool want_vals)
{
#pragma clang fp contract(off)
    const int64_t n2 = g.n2, n3 = g.n3, plane = n2 * n3;
    const int64_t c = g.f2n[r];
    const int64_t i3 = c % n3, i2 = (c / n3) % n2, i1 = c / plane;
    int len = 0;
    auto lower = [&](bool exists, int64_t nb, int64_t j1, int64_t j2, int64_t j3, int dir) {
        if (!exists)
            return;
        const int32_t m = g.nodemap[nb];
        if (m < 0)
            return;
        e.off[len] = (int32_t)((int64_t)m - r);
        if (want_vals)
            e.val[len] = -grid_face_cond(g, nb, j1, j2, j3, dir);
        len++;
    };
    lower(i1 > 0, c - plane, i1 - 1, i2, i3, 0);
    lower(i2 > 0, c - n3, i1, i2 - 1, i3, 1);
    lower(i3 > 0, c - 1, i1, i2, i3 - 1, 2);
    e.off[len] = 0;
    if (want_vals) {
        double d = g.diagA[r];
        if (g.sigma != 0.0) { // the folded shift: product and sum rounded separately (fold_shift_kernel)
            const double s = g.sigma * g.D[r];
            d = d + s;
        }
        e.val[len] = d;
    }
    len++;
    auto upper = [&](bool exists, int64_t nb, int dir) {
        if (!exists)
            return;
        const int32_t m = g.nodemap[nb];
        if (m < 0)
            return;
        e.off[len] = (int32_t)((int64_t)m - r);
        if (want_vals)
            e.val[len] = -grid_face_cond(g, c, i1, i2, i3, dir);
        len++;
    };
    upper(i3 < n3 - 1, c + 1, 2);
    upper(i2 < n2 - 1, c + n3, 1);
    upper(i1 < g.n1 - 1, c + plane, 0);
    e.len = len;
}
