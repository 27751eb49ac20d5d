// jm_rows.h — shared by rows_gemm.hip and rows_ops.hip (the training path's row-major kernels)
#pragma once
#include "jm_common.h"

#ifndef JM_EGRID
#define JM_EGRID 8192
#endif
#define JM_ROWS_CHUNKS 128      /* most row chunks of a two-stage (deterministic) reduction; rows_chunks() picks fewer for short tensors */

namespace jm {

__device__ __forceinline__ int dev_count(int bound, const int* dev) { return dev ? min(bound, max(*dev, 0)) : bound; }

// out[i] (+)= sum over the chunks, in chunk order
static __global__ void rows_sum_partials_kernel(int n, int chunks, const float* __restrict__ partial, float* __restrict__ out, int accumulate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // four independent chains: the loads of consecutive chunks are in flight together (a single chain is one L2 round trip per chunk)
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = 0;
    for (; k + 4 <= chunks; k += 4) {
        s0 += partial[(size_t)k * n + i];       s1 += partial[(size_t)(k + 1) * n + i];
        s2 += partial[(size_t)(k + 2) * n + i]; s3 += partial[(size_t)(k + 3) * n + i];
    }
    for (; k < chunks; ++k) s0 += partial[(size_t)k * n + i];
    const float s = (s0 + s1) + (s2 + s3);
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

// row chunks of a two-stage reduction over (at most) m rows: about 1024 rows per chunk
static inline int rows_chunks(int m) {
    int c = (m + 1023) / 1024;
    return c < 1 ? 1 : (c > JM_ROWS_CHUNKS ? JM_ROWS_CHUNKS : c);
}

static int grid_for(long long work, int block = 256, int cap = 65535 * 4) {
    long long g = (work + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

}  // namespace jm
