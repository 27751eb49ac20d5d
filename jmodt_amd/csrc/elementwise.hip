// elementwise.hip — the one element-wise pass the image branch needs between its two convolutions.
//
// BasicBlock (jmodt/detection/modeling/backbone.py:16-32) is conv3x3 -> BatchNorm -> ReLU -> conv3x3/2.  In eval mode
// the BatchNorm scale folds into the first convolution's weights; what remains is y = relu(conv(x) + b[c]).  MIOpen's
// fused conv+bias+activation plans pick kernels 10-100x slower than its plain fp32 convolution here (measured,
// tools/conv_relu_probe.py), and BatchNorm + ReLU as two framework kernels are two full passes over a 1 GB tensor at
// the first level.  This is the single in-place pass: channels-last data, so the bias index is the fastest axis.
#include "jm_common.h"

namespace jm {

__global__ void __launch_bounds__(256)
bias_relu_cl_kernel(long long n4, int c4, float4* __restrict__ x, const float4* __restrict__ bias) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 v = x[i];
        const float4 b = bias[(int)(i % c4)];
        v.x = fmaxf(v.x + b.x, 0.f); v.y = fmaxf(v.y + b.y, 0.f); v.z = fmaxf(v.z + b.z, 0.f); v.w = fmaxf(v.w + b.w, 0.f);
        x[i] = v;
    }
}

}  // namespace jm

using namespace jm;

extern "C" int jm_bias_relu_channels_last(long long numel, int channels, float* x, const float* bias, jm_stream_t stream) {
    JM_REQUIRE(numel >= 0 && channels >= 4 && channels % 4 == 0 && numel % channels == 0,
               "bias_relu: channels %% 4 == 0 and numel %% channels == 0");
    if (numel == 0) return JM_OK;
    JM_REQUIRE(x && bias, "bias_relu: null pointer");
    JM_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(bias)) & 15u) == 0, "bias_relu: 16-byte alignment");
    const long long n4 = numel / 4;
    long long blocks = (n4 + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;          // grid-stride: 32 workgroups per CU
    hipLaunchKernelGGL(bias_relu_cl_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n4, channels / 4,
                       reinterpret_cast<float4*>(x), reinterpret_cast<const float4*>(bias));
    return check_launch("bias_relu");
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Small glue passes of the composed detector, each ONE launch where the framework needs four to eight (round 5: the step is a
// latency chain of ~300 launches, 174 of them element-wise / gather / copy glue).
namespace jm {

// inverse-distance weights of the three nearest neighbours (pointnet2_modules.py:148-150): dist = sqrt(dist2),
// r = 1 / (dist + 1e-8), w = r / (r0 + r1 + r2) — one thread per unknown point
__global__ void __launch_bounds__(256)
three_nn_weights_kernel(long long rows, const float* __restrict__ dist2, float* __restrict__ w) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    const float r0 = 1.f / (sqrtf(dist2[3 * i]) + 1e-8f), r1 = 1.f / (sqrtf(dist2[3 * i + 1]) + 1e-8f), r2 = 1.f / (sqrtf(dist2[3 * i + 2]) + 1e-8f);
    const float norm = (r0 + r1) + r2;
    w[3 * i] = r0 / norm; w[3 * i + 1] = r1 / norm; w[3 * i + 2] = r2 / norm;
}

// out[b, j, :] = src[b, idx[b, j], :] for rows of `width` floats (the pixel coordinates of the sampled centres, backbone.py:170-171)
__global__ void __launch_bounds__(256)
gather_point_rows_kernel(int n, int m, int width, const float* __restrict__ src, const int* __restrict__ idx, float* __restrict__ out, long long total) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int c = (int)(e % width);
    const long long row = e / width;
    const int b = (int)(row / m);
    out[e] = src[((size_t)b * n + idx[row]) * width + c];
}

// per-point RoI-pooling input [mask, depth, features] (point_rcnn.py:42-44, proposal_target_layer.py:26): mask = sigmoid(cls) >
// thresh, depth = |xyz| / 70 - 0.5, features transposed from (B, C, N) to point-major through a 32-point x 32-channel LDS tile
__global__ void __launch_bounds__(256)
pts_feature_kernel(int N, int C, const float* __restrict__ cls, long long bsc, int ldc, const float* __restrict__ xyz, const float* __restrict__ feats,
                   float thresh, float* __restrict__ out) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
    const int ld = 2 + C;
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k, p = p0 + tx;
        tile[k][tx] = (c < C && p < N) ? feats[((size_t)b * C + c) * N + p] : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int p = p0 + k, c = c0 + tx;
        if (p < N && c < C) out[((size_t)b * N + p) * ld + 2 + c] = tile[tx][k];
    }
    if (blockIdx.y == 0 && threadIdx.x < 32) {
        const int p = p0 + threadIdx.x;
        if (p < N) {
            const float x = cls[(size_t)b * bsc + (size_t)p * ldc];
            const float s = 1.f / (1.f + expf(-x));
            const float* q = xyz + ((size_t)b * N + p) * 3;
            const float d = sqrtf((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]);
            float* o = out + ((size_t)b * N + p) * ld;
            o[0] = s > thresh ? 1.f : 0.f;
            o[1] = d / 70.0f - 0.5f;
        }
    }
}

}  // namespace jm

extern "C" int jm_three_nn_weights(long long rows, const float* dist2, float* weight, jm_stream_t stream) {
    JM_REQUIRE(rows >= 0, "three_nn_weights: bad size");
    if (rows == 0) return JM_OK;
    JM_REQUIRE(dist2 && weight, "three_nn_weights: null pointer");
    hipLaunchKernelGGL(three_nn_weights_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rows, dist2, weight);
    return check_launch("three_nn_weights");
}

extern "C" int jm_gather_point_rows(int b, int n, int m, int width, const float* src, const int* idx, float* out, jm_stream_t stream) {
    JM_REQUIRE(b >= 0 && n >= 0 && m >= 0 && width >= 1, "gather_point_rows: bad sizes");
    const long long total = (long long)b * m * width;
    if (total == 0) return JM_OK;
    JM_REQUIRE(src && idx && out && n > 0, "gather_point_rows: null pointer");
    hipLaunchKernelGGL(gather_point_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, m, width, src, idx, out,
                       total);
    return check_launch("gather_point_rows");
}

extern "C" int jm_pts_feature(int b, int n, int c, const float* rpn_cls, long long batch_stride_cls, int ld_cls, const float* xyz, const float* feats, float score_thresh,
                              float* out, jm_stream_t stream) {
    JM_REQUIRE(b >= 0 && n >= 0 && c >= 0 && ld_cls >= 1, "pts_feature: bad sizes");
    if (b == 0 || n == 0) return JM_OK;
    JM_REQUIRE(rpn_cls && xyz && (c == 0 || feats) && out && b <= 65535, "pts_feature: null pointer / batch > 65535");
    hipLaunchKernelGGL(pts_feature_kernel, dim3((unsigned)divup(n, 32), (unsigned)imax(1, divup(c, 32)), (unsigned)b), dim3(256), 0,
                       (hipStream_t)stream, n, c, rpn_cls, batch_stride_cls, ld_cls, xyz, feats, score_thresh, out);
    return check_launch("pts_feature");
}
