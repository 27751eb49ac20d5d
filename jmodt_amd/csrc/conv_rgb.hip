// conv_rgb.hip — the image branch's first convolution: 3x3, padding 1, stride 1, THREE input channels, with the folded
// BatchNorm bias and the ReLU in the same pass (gfx950).
//
// backbone.py:16-32 (BasicBlock.conv1 + bn1 + relu of Img_Block[0]) on the full-resolution image: K = 27, so the layer
// is 14 GFLOP per batch of 8 frames but WRITES 1 GB (8 x 64 x 384 x 1280 fp32).  As library calls it is a convolution
// (0.48 ms) followed by the bias + ReLU pass over the same 1 GB (0.30 ms read + write); here the output is written once,
// finished: 0.33 ms = 3.2 TB/s on the written bytes (measured, tools/conv_rgb_bench.py).
//   * lane = pixel (64 consecutive pixels of one image row per wave): the 27 input taps of a lane live in registers,
//     loaded from the NCHW image with coalesced rows (neighbouring lanes share them through L1);
//   * the weights are wave-uniform: tap-major [27][cout], read with scalar loads (s_load_dwordx4 = 4 output channels of
//     one tap) straight into the SGPR operand of v_fmac_f32 — no LDS, no vector loads in the inner loop;
//   * the finished 64 px x cout tile goes through a per-wave LDS transpose so that the channels-last output is written
//     as full 16-byte x 64-lane rows.
#include "jm_common.h"
#include <type_traits>

namespace jm {

typedef float f32x2 __attribute__((ext_vector_type(2)));

// cg = output channels per wave (a multiple of 4, <= 32), ng = cout / cg waves share one 64-pixel chunk (1, 2 or 4)
__global__ void __launch_bounds__(256)
conv3x3_rgb_kernel(int H, int W, int cout, int cg, int ng, const float* __restrict__ img, const float* __restrict__ wt,
                   const float* __restrict__ bias, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int chunk = wave / ng, grp = wave - chunk * ng;       // ng = 2: waves (0, 1) share chunk 0, (2, 3) chunk 1
    const int x0 = (blockIdx.x * (4 / ng) + chunk) * 64;
    if (x0 >= W) return;                                         // wave-uniform; no workgroup barrier below
    const int x = x0 + lane, y = blockIdx.y, b = blockIdx.z;
    float p[27];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int yy = y + dy - 1;
            const bool yok = yy >= 0 && yy < H;
            const float* row = img + (((size_t)b * 3 + c) * H + (yok ? yy : 0)) * W;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int xx = x + dx - 1;
                const bool ok = yok && xx >= 0 && xx < W;
                const float v = row[ok ? xx : 0];                // unconditional load on a clamped address
                p[(c * 3 + dy) * 3 + dx] = ok ? v : 0.f;
            }
        }
    const int ld = cg + 4, c0 = grp * cg;
    float* T = lds + (size_t)wave * 64 * ld;
    // tap-outer: all cg (<= 32) accumulators of the lane's pixel stay in registers; one tap's cg weights are cg
    // consecutive floats of the tap-major array = wide scalar loads (s_load_dwordx8 / x16) into the SGPR operands of
    // v_pk_fma_f32 (two channels per instruction); uniform addresses: weights and bias through the scalar cache
    auto run = [&](auto NQ_) __attribute__((always_inline)) {
        constexpr int NQ = decltype(NQ_)::value;                 // float4 groups of output channels: cg = 4 NQ
        f32x2 acc[2 * NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const float4 bv = *reinterpret_cast<const float4*>(bias + c0 + 4 * q);
            acc[2 * q] = (f32x2){bv.x, bv.y}; acc[2 * q + 1] = (f32x2){bv.z, bv.w};
        }
#pragma unroll
        for (int t = 0; t < 27; ++t) {
            const f32x2 pp = {p[t], p[t]};
            const float* wr = wt + (size_t)t * cout + c0;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const float4 w = *reinterpret_cast<const float4*>(wr + 4 * q);
                acc[2 * q] = __builtin_elementwise_fma(pp, (f32x2){w.x, w.y}, acc[2 * q]);
                acc[2 * q + 1] = __builtin_elementwise_fma(pp, (f32x2){w.z, w.w}, acc[2 * q + 1]);
            }
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            float4 a;
            a.x = fmaxf(acc[2 * q].x, 0.f); a.y = fmaxf(acc[2 * q].y, 0.f);
            a.z = fmaxf(acc[2 * q + 1].x, 0.f); a.w = fmaxf(acc[2 * q + 1].y, 0.f);
            *reinterpret_cast<float4*>(T + (size_t)lane * ld + 4 * q) = a;
        }
    };
    switch (cg >> 2) {
        case 8: run(std::integral_constant<int, 8>{}); break;
        case 4: run(std::integral_constant<int, 4>{}); break;
        default:
            for (int co = 0; co < cg; co += 4) {                 // other widths: four channels at a time
                const float4 bv = *reinterpret_cast<const float4*>(bias + c0 + co);
                f32x2 lo = {bv.x, bv.y}, hi = {bv.z, bv.w};
#pragma unroll
                for (int t = 0; t < 27; ++t) {
                    const float4 w = *reinterpret_cast<const float4*>(wt + (size_t)t * cout + c0 + co);
                    const f32x2 pp = {p[t], p[t]};
                    lo = __builtin_elementwise_fma(pp, (f32x2){w.x, w.y}, lo);
                    hi = __builtin_elementwise_fma(pp, (f32x2){w.z, w.w}, hi);
                }
                float4 a;
                a.x = fmaxf(lo.x, 0.f); a.y = fmaxf(lo.y, 0.f); a.z = fmaxf(hi.x, 0.f); a.w = fmaxf(hi.y, 0.f);
                *reinterpret_cast<float4*>(T + (size_t)lane * ld + co) = a;
            }
    }
    // the wave's own tile only: LDS operations of one wave complete in order, no barrier needed
    const int q4 = cg >> 2;                                      // float4 per pixel of this wave's channel group
    float* ob = out + (((size_t)b * H + y) * W + x0) * cout + c0;
    const int npx = min(64, W - x0);
    for (int i = lane; i < npx * q4; i += 64) {
        const int px = i / q4, q = i - px * q4;
        const float4 v = *reinterpret_cast<const float4*>(T + (size_t)px * ld + 4 * q);
        // streaming store: 1 GB that the next convolution reads once (measured 325 vs 347 us with the default policy)
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        __builtin_nontemporal_store((f32x4){v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4*>(ob + (size_t)px * cout + 4 * q));
    }
}

}  // namespace jm

using namespace jm;

extern "C" int jm_conv3x3_rgb_bias_relu(int b, int h, int w, int cout, const float* image, const float* weight_tap_major,
                                        const float* bias, float* out_channels_last, jm_stream_t stream) {
    JM_REQUIRE(b >= 0 && h >= 0 && w >= 0 && cout >= 4 && cout % 4 == 0 && (cout <= 32 || cout == 64 || cout == 128),
               "conv3x3_rgb: cout %% 4 == 0 and <= 32, or 64, or 128");
    if (b == 0 || h == 0 || w == 0) return JM_OK;
    JM_REQUIRE(image && weight_tap_major && bias && out_channels_last, "conv3x3_rgb: null pointer");
    JM_REQUIRE(((reinterpret_cast<uintptr_t>(weight_tap_major) | reinterpret_cast<uintptr_t>(bias) |
                 reinterpret_cast<uintptr_t>(out_channels_last)) & 15u) == 0, "conv3x3_rgb: 16-byte alignment");
    JM_REQUIRE(h <= 65535 && b <= 65535, "conv3x3_rgb: grid limits");
    const int cg = cout <= 32 ? cout : 32, ng = cout / cg;       // waves per 64-pixel chunk: 1, 2 or 4
    const size_t lds_bytes = (size_t)4 * 64 * (cg + 4) * sizeof(float);
    hipLaunchKernelGGL(conv3x3_rgb_kernel, dim3((unsigned)divup(w, 64 * (4 / ng)), (unsigned)h, (unsigned)b), dim3(256), lds_bytes,
                       (hipStream_t)stream, h, w, cout, cg, ng, image, weight_tap_major, bias, out_channels_last);
    return check_launch("conv3x3_rgb");
}
