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
