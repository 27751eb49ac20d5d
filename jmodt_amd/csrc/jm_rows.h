// jm_rows.h — shared by rows_gemm.hip and rows_ops.hip (the training path's row-major kernels)
#pragma once
#include "jm_common.h"

#define JM_ROWS_CHUNKS 256      /* row chunks of the two-stage (deterministic) reductions */

namespace jm {

__device__ __forceinline__ int dev_count(int bound, const int* dev) { return dev ? min(bound, max(*dev, 0)) : bound; }

// out[i] (+)= sum over the chunks, in chunk order
static __global__ void rows_sum_partials_kernel(int n, int chunks, const float* __restrict__ partial, float* __restrict__ out, int accumulate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int k = 0; k < chunks; ++k) s += partial[(size_t)k * n + i];
    out[i] = accumulate ? out[i] + s : s;
}

// column sums of a row tensor, two stages: partial[s, c] over row chunk s (workgroup = (64 columns, chunk), 4 waves stride the
// chunk's rows), then rows_sum_partials_kernel
static __global__ void __launch_bounds__(256)
rows_colsum_part_kernel(int M, const int* __restrict__ m_dev, int N, int chunks, const float* __restrict__ x, int ldx, float* __restrict__ partial) {
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + lane;
    const int Mv = dev_count(M, m_dev);
    const int per = (Mv + chunks - 1) / chunks;
    const int r0 = min(Mv, (int)blockIdx.y * per), r1 = min(Mv, r0 + per);
    float s = 0.f;
    if (col < N)
        for (int r = r0 + wave; r < r1; r += 4) s += x[(size_t)r * ldx + col];
    part[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && col < N) partial[(size_t)blockIdx.y * N + col] = part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane];
}

static int grid_for(long long work, int block = 256, int cap = 65535 * 4) {
    long long g = (work + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

}  // namespace jm
