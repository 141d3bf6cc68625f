// Host-only helpers (no device code): the grid axes of regulargrid.
#include <cstddef>
#include <cstdint>
#include <vector>

#include "fvhip.h"

// Julia `range(a; stop=b, length=n)` (/root/reference/src/grid.jl:62-64) builds a
// twice-precision StepRangeLen: element k is the correctly rounded value of
// a + k (b-a)/(n-1).  Evaluated here in binary128 with a single final rounding.
// Only n1+n2+n3 values are needed (the device kernel indexes these tables), so
// this stays on the host.
int fv_grid_axes(const double mins[3], const double maxs[3], const int64_t ns[3], std::vector<double> ax[3])
{
    for (int d = 0; d < 3; d++) {
        const int64_t n = ns[d];
        if (n < 2)
            return FV_ERR_ARG; // grid.jl:65-67 index xs[2]
        ax[d].resize((size_t)n);
        for (int64_t k = 0; k < n; k++) {
            const __float128 v =
                ((__float128)mins[d] * (__float128)(n - 1 - k) + (__float128)maxs[d] * (__float128)k) / (__float128)(n - 1);
            ax[d][(size_t)k] = (double)v;
        }
        ax[d][0] = mins[d];
        ax[d][(size_t)n - 1] = maxs[d];
    }
    return FV_OK;
}

// ------------------------------------------------------------------ locality ordering of the free cells
// Reverse Cuthill-McKee over the graph whose edges are the faces between two free cells (ea[k] - eb[k], canonical free
// indices; repeated faces and self-loops allowed).  What the reference's users do by hand when a mesh arrives numbered at
// random (DFN meshes, examples/fractures): the assembled operator keeps its meaning under a symmetric permutation, but the
// SpMV's gather of x[col] only coalesces when a row's neighbours are numbered near it.  Breadth-first from a lowest-degree
// cell of every component, a cell's unvisited neighbours appended by ascending degree, the whole order reversed.
// perm[old] = new.  mean_before / mean_after: mean |i - j| over the edges, the quantity the caller decides on.
#include <algorithm>
#include <cmath>

int fv_host_locality_order(int64_t n, int64_t m, const int32_t *ea, const int32_t *eb, int32_t *perm, double *mean_before, double *mean_after)
{
    std::vector<int64_t> ptr((size_t)n + 1, 0);
    double sum = 0.0;
    int64_t used = 0;
    for (int64_t k = 0; k < m; k++) {
        const int32_t a = ea[k], b = eb[k];
        if (a == b)
            continue;
        ptr[(size_t)a + 1]++;
        ptr[(size_t)b + 1]++;
        sum += std::fabs((double)a - (double)b);
        used++;
    }
    *mean_before = used ? sum / (double)used : 0.0;
    for (int64_t i = 0; i < n; i++)
        ptr[(size_t)i + 1] += ptr[(size_t)i];
    std::vector<int32_t> adj((size_t)ptr[(size_t)n]);
    {
        std::vector<int64_t> fill(ptr.begin(), ptr.end() - 1);
        for (int64_t k = 0; k < m; k++) {
            const int32_t a = ea[k], b = eb[k];
            if (a == b)
                continue;
            adj[(size_t)fill[(size_t)a]++] = b;
            adj[(size_t)fill[(size_t)b]++] = a;
        }
    }
    auto degree = [&](int32_t v) { return (int64_t)(ptr[(size_t)v + 1] - ptr[(size_t)v]); };
    // cells by ascending degree (counting sort, ties by index): the start cells of the components
    int64_t maxdeg = 0;
    for (int64_t i = 0; i < n; i++)
        maxdeg = std::max(maxdeg, degree((int32_t)i));
    std::vector<int64_t> bucket((size_t)maxdeg + 2, 0);
    for (int64_t i = 0; i < n; i++)
        bucket[(size_t)degree((int32_t)i) + 1]++;
    for (int64_t d = 0; d <= maxdeg; d++)
        bucket[(size_t)d + 1] += bucket[(size_t)d];
    std::vector<int32_t> bydeg((size_t)n);
    for (int64_t i = 0; i < n; i++)
        bydeg[(size_t)bucket[(size_t)degree((int32_t)i)]++] = (int32_t)i;
    std::vector<int32_t> order;
    order.reserve((size_t)n);
    std::vector<uint8_t> seen((size_t)n, 0);
    std::vector<int32_t> nb;
    for (int64_t s = 0; s < n; s++) {
        const int32_t start = bydeg[(size_t)s];
        if (seen[(size_t)start])
            continue;
        seen[(size_t)start] = 1;
        size_t head = order.size();
        order.push_back(start);
        while (head < order.size()) {
            const int32_t u = order[head++];
            nb.clear();
            for (int64_t k = ptr[(size_t)u]; k < ptr[(size_t)u + 1]; k++) {
                const int32_t v = adj[(size_t)k];
                if (!seen[(size_t)v]) {
                    seen[(size_t)v] = 1;
                    nb.push_back(v);
                }
            }
            std::sort(nb.begin(), nb.end(), [&](int32_t x, int32_t y) {
                const int64_t dx = degree(x), dy = degree(y);
                return dx != dy ? dx < dy : x < y;
            });
            order.insert(order.end(), nb.begin(), nb.end());
        }
    }
    if ((int64_t)order.size() != n)
        return FV_ERR_STATE;
    for (int64_t k = 0; k < n; k++)
        perm[(size_t)order[(size_t)k]] = (int32_t)(n - 1 - k);
    sum = 0.0;
    for (int64_t k = 0; k < m; k++)
        if (ea[k] != eb[k])
            sum += std::fabs((double)perm[ea[k]] - (double)perm[eb[k]]);
    *mean_after = used ? sum / (double)used : 0.0;
    return FV_OK;
}
