// conv1d_stack.hip — a stack of 1x1 Conv1d (+ folded BatchNorm) (+ ReLU) layers on (B, C, n) tensors, one launch (gfx950).
//
// The per-point layers of the detector that are neither grouped (sa_mlp*.hip) nor gated (li_fusion.hip):
//   * the RPN heads, rpn.py:34-58 (Conv1d 128 -> 128 -> 1 and 128 -> 128 -> 76 on every point of the cloud),
//   * the feature-propagation MLPs, pointnet2_modules.py:139-153 (cat[interpolated, skip] -> SharedMLP),
//   * the hoisted first set-abstraction layer u = W_f . f + W_x . xyz^T (sa_mlp.hip, pre-projected form).
// As library calls each layer is a batched GEMM over 8 small problems + a bias broadcast copy + a ReLU pass (rocBLAS:
// 3-5 TFLOP/s on the 64..128-wide layers, 130 us for the K = 3 coordinate product alone), and the concatenation is a copy.
// Here the chain runs on 32-point tiles:
//   * a (B, C, n) tensor is k-major for a tile of consecutive points = the MFMA A-operand layout: operands whose width is
//     a multiple of 16 are read IN PLACE (row stride n, 128-byte coalesced rows per channel), others (and the point-major
//     xyz operand) are staged zero padded in LDS;
//   * the first layer takes up to two operands accumulating into the same columns (the concatenation is never built);
//   * hidden activations stay in LDS (k-major [column][36]); the last layer writes (B, out, n) as float4 runs of 4 points;
//   * four waves split the columns (blocks w, w + 4, ...), weights straight from L1/L2 in the packed layout of
//     jm_sa_mlp_pack, register double-buffered (jm_mfma.h: wide_ktiles).
// v_mfma_f32_32x32x2_f32: exact-f32 products, 1e-4 parity with the fp32 reference path.
#include "jm_mfma.h"

// (A/B switch of the weight prefetch depth, jm_mfma.h: -DJM_WK_DEEP=4 requests a group of four k-tiles ahead)
#ifndef JM_WK_DEEP
#define JM_WK_DEEP 0
#endif
#if JM_WK_DEEP
#define JM_WK2(...) wide_ktiles_deep<2, JM_WK_DEEP>(__VA_ARGS__)
#define JM_WK1(...) wide_ktiles_deep<1, JM_WK_DEEP>(__VA_ARGS__)
#else
#define JM_WK2(...) wide_ktiles<2>(__VA_ARGS__)
#define JM_WK1(...) wide_ktiles<1>(__VA_ARGS__)
#endif

namespace jm {

struct ConvStackParams {
    int n, tiles_per_frame;
    int c0, c1;                       // first-layer operand widths (c1 == 0: one operand)
    int c0p, c1p;                     // pad16
    int s0, s1;                       // operand staged in LDS (else read in place)
    int xyz1;                         // operand 1 is point-major xyz (B, n, 3): always staged
    const float *x0, *x1;             // (B, c, n)
    int L;                            // 1..3 layers
    int w0, w1, w2;                   // layer widths
    int np0, np1, np2;                // pad128
    const float *W0a, *W0b, *W1, *W2; // packed weights: layer 0 per operand, layers 1, 2
    const float *b0, *b1, *b2;        // packed biases
    int r0, r1, r2;                   // ReLU after layer l
    float* out;                       // (B, w_last, n), or (B, n, w_last) with out_pm
    int out_pm;
};

__global__ void __launch_bounds__(256)
conv1d_stack_kernel(ConvStackParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lk = lane >> 5;
    const int a_off = lk * SW_LD + lr;
    const int n = p.n;
    float* X0 = lds;                                              // [c0p][36] when staged
    float* X1 = X0 + (p.s0 ? (size_t)p.c0p * SW_LD : 0);          // [c1p][36] when staged
    float* T0 = X1 + (p.s1 ? (size_t)p.c1p * SW_LD : 0);          // [np0][36] layer-0 activations (L >= 2)
    float* T1 = T0 + (p.L >= 2 ? (size_t)p.np0 * SW_LD : 0);      // [np1][36] layer-1 activations (L == 3)
    const int bi_ = blockIdx.x / p.tiles_per_frame;
    const int row0 = (blockIdx.x % p.tiles_per_frame) * SW_BM;
    if (p.s0 || p.s1) {
        const int r = tid & 31, q0 = tid >> 5;
        if (p.s0) {
            const float* xb = p.x0 + (size_t)bi_ * p.c0 * n + row0 + r;
            for (int cb = q0; cb < p.c0p; cb += 64) {           // eight channel rows in flight per thread (see li_fusion.hip)
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int c = cb + 8 * u; v[u] = c < p.c0 ? xb[(size_t)c * n] : 0.f; }
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int c = cb + 8 * u; if (c < p.c0p) X0[c * SW_LD + r] = v[u]; }
            }
        }
        if (p.s1) {
            if (p.xyz1) {
                const float* xb = p.x1 + ((size_t)bi_ * n + row0 + r) * 3;
                for (int c = q0; c < p.c1p; c += 8) X1[c * SW_LD + r] = c < 3 ? xb[c] : 0.f;
            } else {
                const float* xb = p.x1 + (size_t)bi_ * p.c1 * n + row0 + r;
                for (int c = q0; c < p.c1p; c += 8) X1[c * SW_LD + r] = c < p.c1 ? xb[(size_t)c * n] : 0.f;
            }
        }
        lds_barrier();
    }
    const float* A0 = p.s0 ? X0 : p.x0 + (size_t)bi_ * p.c0 * n + row0;
    const size_t lda0 = p.s0 ? (size_t)SW_LD : (size_t)n;
    const int off0 = p.s0 ? a_off : lk * n + lr;
    const float* A1 = p.c1 == 0 ? nullptr : (p.s1 ? X1 : p.x1 + (size_t)bi_ * p.c1 * n + row0);
    const size_t lda1 = p.s1 ? (size_t)SW_LD : (size_t)n;
    const int off1 = p.s1 ? a_off : lk * n + lr;

    auto set_bias = [=](f32x16& a, const float* bias, int cb) __attribute__((always_inline)) {
        const float bv = bias[cb * 32 + lr];
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] = bv;
    };
    // one GEMM stage over one or two (A, W) operand pairs accumulating into the same columns; `fin(acc, cb)` consumes a
    // finished 32x32 block
    auto stage = [&](const float* Aa, int kap, size_t lda, int offa, const float* Wa, const float* Ab, int kbp, size_t ldb,
                     int offb, const float* Wb, int np, const float* bias, auto fin) __attribute__((always_inline)) {
        const int nb = np >> 7;
        const size_t st = (size_t)np * 16;
        for (int j0 = 0; j0 < nb; j0 += 2) {
            const int cb = wave + 4 * j0;
            const size_t off = ((size_t)cb * 32 + lr) * 16 + lk * 8;
            f32x16 acc[2];
            set_bias(acc[0], bias, cb);
            if (j0 + 1 < nb) {
                set_bias(acc[1], bias, cb + 4);
                JM_WK2(Aa, kap / 16, Wa + off, st, offa, acc, lda);
                if (Ab) JM_WK2(Ab, kbp / 16, Wb + off, st, offb, acc, ldb);
                fin(acc[0], cb); fin(acc[1], cb + 4);
            } else {
                JM_WK1(Aa, kap / 16, Wa + off, st, offa, acc, lda);
                if (Ab) JM_WK1(Ab, kbp / 16, Wb + off, st, offb, acc, ldb);
                fin(acc[0], cb);
            }
        }
    };
    // accumulator r = 4 rq + t  <->  row 8 rq + 4 lk + t, column cb * 32 + lr
    auto to_lds = [=](float* T, int relu) {
        return [=](const f32x16& a, int cb) __attribute__((always_inline)) {
            float* Tc = T + (size_t)(cb * 32 + lr) * SW_LD + 4 * lk;
            const float lo = relu ? 0.f : -__builtin_inff();
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                float4 v;
                v.x = fmaxf(a[4 * rq + 0], lo); v.y = fmaxf(a[4 * rq + 1], lo);
                v.z = fmaxf(a[4 * rq + 2], lo); v.w = fmaxf(a[4 * rq + 3], lo);
                *reinterpret_cast<float4*>(Tc + 8 * rq) = v;
            }
        };
    };
    float* const outp = p.out;
    const int out_pm = p.out_pm;
    auto to_out = [=](int oc, int relu) {
        float* outb = outp + (size_t)bi_ * oc * n + row0;
        float* outr = outp + ((size_t)bi_ * n + row0) * oc;          // point-major: row stride oc
        return [=](const f32x16& a, int cb) __attribute__((always_inline)) {
            const int col = cb * 32 + lr;
            if (col >= oc) return;
            const float lo = relu ? 0.f : -__builtin_inff();
            if (out_pm) {                                           // 32 lanes = 32 consecutive channels of one point
                float* o = outr + (size_t)(4 * lk) * oc + col;
#pragma unroll
                for (int r = 0; r < 16; ++r) o[(size_t)((r & 3) + 8 * (r >> 2)) * oc] = fmaxf(a[r], lo);
                return;
            }
            float* o = outb + (size_t)col * n + 4 * lk;
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                float4 v;
                v.x = fmaxf(a[4 * rq + 0], lo); v.y = fmaxf(a[4 * rq + 1], lo);
                v.z = fmaxf(a[4 * rq + 2], lo); v.w = fmaxf(a[4 * rq + 3], lo);
                *reinterpret_cast<float4*>(o + 8 * rq) = v;
            }
        };
    };
    if (p.L == 1) {
        stage(A0, p.c0p, lda0, off0, p.W0a, A1, p.c1p, lda1, off1, p.W0b, p.np0, p.b0, to_out(p.w0, p.r0));
        return;
    }
    stage(A0, p.c0p, lda0, off0, p.W0a, A1, p.c1p, lda1, off1, p.W0b, p.np0, p.b0, to_lds(T0, p.r0));
    lds_barrier();
    const int k1p = pad_to(p.w0, 16);
    if (p.L == 2) {
        stage(T0, k1p, SW_LD, a_off, p.W1, nullptr, 0, 0, 0, nullptr, p.np1, p.b1, to_out(p.w1, p.r1));
        return;
    }
    stage(T0, k1p, SW_LD, a_off, p.W1, nullptr, 0, 0, 0, nullptr, p.np1, p.b1, to_lds(T1, p.r1));
    lds_barrier();
    stage(T1, pad_to(p.w1, 16), SW_LD, a_off, p.W2, nullptr, 0, 0, 0, nullptr, p.np2, p.b2, to_out(p.w2, p.r2));
}

struct ConvStackPlan { int s0, s1; size_t lds_bytes; bool ok; };

static ConvStackPlan conv1d_stack_plan(int c0, int c1, int xyz1, int L, const int* w) {
    ConvStackPlan pl{};
    pl.s0 = (c0 % 16) != 0;
    pl.s1 = c1 > 0 && (xyz1 || (c1 % 16) != 0);
    size_t rows = (pl.s0 ? pad_to(c0, 16) : 0) + (pl.s1 ? pad_to(xyz1 ? 3 : c1, 16) : 0);
    for (int l = 0; l + 1 < L; ++l) rows += pad_to(w[l], 128);
    pl.lds_bytes = rows * SW_LD * sizeof(float);
    pl.ok = pl.lds_bytes <= 160 * 1024;
    return pl;
}

}  // namespace jm

using namespace jm;

extern "C" int jm_conv1d_stack_supported(int b, int n, int c0, int c1, int xyz1, int num_layers, const int* widths) {
    if (b < 0 || n < 1 || c0 < 1 || c1 < 0 || num_layers < 1 || num_layers > 3 || !widths) return 0;
    if (xyz1 && c1 != 3) return 0;
    if (n % 32 || (long long)b * (n / 32) >= (1LL << 31)) return 0;
    for (int l = 0; l < num_layers; ++l)
        if (widths[l] < 1) return 0;
    return conv1d_stack_plan(c0, c1, xyz1, num_layers, widths).ok ? 1 : 0;
}

extern "C" int jm_conv1d_stack_forward(int b, int n, int c0, const float* x0, int c1, const float* x1, int xyz1, int num_layers,
                                       const int* widths, const float* w0a, const float* w0b, const float* const* weights,
                                       const float* const* biases, const int* relu, int out_point_major, float* out,
                                       jm_stream_t stream) {
    JM_REQUIRE(b >= 0 && n >= 0, "conv1d_stack: bad sizes");
    if (b == 0 || n == 0) return JM_OK;
    JM_REQUIRE(jm_conv1d_stack_supported(b, n, c0, c1, xyz1, num_layers, widths),
               "conv1d_stack: unsupported shape (n %% 32 == 0, 1..3 layers, staged operands + hidden tiles within the 160 KB LDS)");
    JM_REQUIRE(x0 && w0a && weights && biases && relu && out && biases[0] && (c1 == 0 || (x1 && w0b)), "conv1d_stack: null pointer");
    for (int l = 1; l < num_layers; ++l) JM_REQUIRE(weights[l] && biases[l], "conv1d_stack: null layer pointer");
    JM_REQUIRE(((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(w0a) | reinterpret_cast<uintptr_t>(w0b)) & 15u) == 0,
               "conv1d_stack: 16-byte alignment");
    const ConvStackPlan pl = conv1d_stack_plan(c0, c1, xyz1, num_layers, widths);
    ConvStackParams p{};
    p.n = n; p.tiles_per_frame = n / 32;
    p.c0 = c0; p.c1 = c1; p.c0p = pad_to(c0, 16); p.c1p = c1 ? pad_to(c1, 16) : 0;
    p.s0 = pl.s0; p.s1 = pl.s1; p.xyz1 = xyz1;
    p.x0 = x0; p.x1 = x1;
    p.L = num_layers;
    p.w0 = widths[0]; p.np0 = pad_to(widths[0], 128); p.r0 = relu[0]; p.b0 = biases[0];
    p.W0a = w0a; p.W0b = w0b;
    if (num_layers > 1) { p.w1 = widths[1]; p.np1 = pad_to(widths[1], 128); p.r1 = relu[1]; p.b1 = biases[1]; p.W1 = weights[1]; }
    if (num_layers > 2) { p.w2 = widths[2]; p.np2 = pad_to(widths[2], 128); p.r2 = relu[2]; p.b2 = biases[2]; p.W2 = weights[2]; }
    p.out = out; p.out_pm = out_point_major ? 1 : 0;
    (void)hipFuncSetAttribute((const void*)conv1d_stack_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(conv1d_stack_kernel, dim3((unsigned)(b * (n / 32))), dim3(256), pl.lds_bytes, (hipStream_t)stream, p);
    return check_launch("conv1d_stack");
}
