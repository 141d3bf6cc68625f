// Device-side helpers shared by the HIP translation units: wave64 / block reductions and the
// grid-stride conventions of the vector kernels.
#pragma once
#include "fv_internal.h"

__device__ inline double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        v += __shfl_xor(v, off, 64);
    return v;
}

// all threads of the 256-thread block get the sum; smem: 4 doubles
__device__ inline double block_sum(double v, double *smem)
{
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads(); // smem may still be read by a previous call
    if (lane == 0)
        smem[wave] = v;
    __syncthreads();
    return (smem[0] + smem[1]) + (smem[2] + smem[3]);
}

__device__ inline double reduce_partials(const double *__restrict__ part, int count, double *smem)
{
    double v = 0.0;
    for (int i = threadIdx.x; i < count; i += FV_BLOCK)
        v += part[i];
    return block_sum(v, smem);
}


__device__ inline int64_t vec_stride() { return (int64_t)gridDim.x * FV_BLOCK; }

// grid of the double2 vector kernels: at most FV_VEC_PARTIALS blocks, one partial sum per block
static inline int vec_grid(int64_t n)
{
    int64_t g = (n / 2 + FV_BLOCK - 1) / FV_BLOCK;
    if (g < 1)
        g = 1;
    if (g > FV_VEC_PARTIALS)
        g = FV_VEC_PARTIALS;
    return (int)g;
}
