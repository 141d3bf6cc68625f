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
