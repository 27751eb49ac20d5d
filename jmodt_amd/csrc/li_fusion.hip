// li_fusion.hip — LI-Fusion attention block on the fp32 matrix cores (gfx950), one launch per pyramid level.
//
// Replaces AttentionFusion.forward / IALayer.forward (jmodt/detection/modeling/backbone.py:35-81) in eval mode:
//     ri = fc1(I^T); rp = fc2(P^T); att = sigmoid(fc3(tanh(ri + rp)))                 per point
//     img_new = relu(bn(conv1(I))) * att
//     out = relu(bn1(conv1(cat[P, img_new])))
// P (B, pc, n) point features, I (B, ic, n) image features gathered at the points, out (B, oc, n).
// The reference runs 2 transposes + 3 Linear + tanh + sigmoid + Conv1d + BN + ReLU + mul + cat + Conv1d + BN + ReLU:
// ~16 kernels and ~10 passes over (B, C, n) tensors per level; here the per-point chain is evaluated on 32-point
// tiles whose operands never leave LDS / registers:
//   * (B, C, n) tensors are k-major for a tile of consecutive points, i.e. already the MFMA A-operand layout:
//     the I and P tiles are staged once in LDS (coalesced 128-byte rows) and used by two GEMM stages each;
//   * stage A: T = tanh([W1 | W2] . [I ; P] + b1 + b2) -> LDS, att = sigmoid(w3 . T + b3) per point (32 threads);
//   * stage C: G = relu(Wi . I + bi) * att -> LDS (BatchNorm folded into Wi / bi by the caller);
//   * stage D: out = relu(WfP . P + WfG . G + bf): the concatenation is two GEMMs accumulating in the same registers;
//   * four waves split the columns of every stage (blocks w, w+4, ...), weights straight from L1/L2 in the packed
//     layout of jm_sa_mlp_pack, register double-buffered (jm_mfma.h: wide_ktiles).
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

struct LiFusionParams {
    int n, ic, pc, rc, oc;            // points per frame (multiple of 32), channel widths
    int icp, pcp;                     // pad16
    int np_a, np_c, np_d;             // pad128 of rc, pc, oc: packed columns of the three stages
    const float *I, *P;               // (B, ic, n), (B, pc, n)
    const float *W1, *W2, *Wi, *WfP, *WfG;   // packed: (rc x ic), (rc x pc), (pc x ic), (oc x pc), (oc x pc)
    const float *ba, *bi, *bf;        // packed biases: b1 + b2 (np_a), bi (np_c), bf (np_d)
    const float* w3;                  // (rc)
    float b3;
    float* out;                       // (B, oc, n)
    int tiles_per_frame;
    int p_in_lds;                     // 0: the point-feature tile does not fit next to I, T and G — read in place
};

__global__ void __launch_bounds__(256)
attention_fusion_kernel(LiFusionParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lk = lane >> 5;
    const int a_off = lk * SW_LD + lr;
    const int ic = p.ic, pc = p.pc, rc = p.rc, oc = p.oc, icp = p.icp, pcp = p.pcp, n = p.n;
    const bool p_lds = p.p_in_lds != 0;
    float* XI = lds;                                  // [icp][36]
    float* XP = XI + (size_t)icp * SW_LD;             // [pcp][36] (only when it fits)
    float* T = XP + (p_lds ? (size_t)pcp * SW_LD : 0);   // [np_a][36]  tanh(ri + rp)
    float* G = T + (size_t)p.np_a * SW_LD;            // [np_c][36]  gated image features
    float* att = G + (size_t)p.np_c * SW_LD;          // [32]
    const int bi_ = blockIdx.x / p.tiles_per_frame;
    const int row0 = (blockIdx.x % p.tiles_per_frame) * SW_BM;
    // ---- stage the two input tiles: thread -> (row = tid & 31, channels tid >> 5 + 8 j); zero rows up to pad16
    {
        const int r = tid & 31, c0 = tid >> 5;
        const float* Ib = p.I + (size_t)bi_ * ic * n + row0 + r;
        const float* Pb = p.P + (size_t)bi_ * pc * n + row0 + r;
        // eight channel rows in flight per thread: one load per iteration waits a whole global-load latency per channel row (round 5:
        // measured on conv1d_stack64.hip's staging, where that WAS the kernel's time)
        auto stage = [&](float* X, const float* Xb, int cw, int cwp) __attribute__((always_inline)) {
            for (int cb = c0; cb < cwp; cb += 64) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int c = cb + 8 * u;
                    v[u] = c < cw ? Xb[(size_t)c * n] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int c = cb + 8 * u;
                    if (c < cwp) X[c * SW_LD + r] = v[u];
                }
            }
        };
        stage(XI, Ib, ic, icp);
        if (p_lds) stage(XP, Pb, pc, pcp);
    }
    lds_barrier();
    // the P operand: the LDS tile, or — wide levels — the (pc, n) tensor itself: a k-major tile of consecutive points is
    // exactly the MFMA A-operand layout with row stride n (128-byte coalesced reads per k; host: pc % 16 == 0 then)
    const float* PA = p_lds ? XP : p.P + (size_t)bi_ * pc * n + row0;
    const size_t p_lda = p_lds ? (size_t)SW_LD : (size_t)n;
    const int p_off = p_lds ? a_off : lk * n + lr;

    auto set_bias = [=](f32x16& a, const float* bias, int cb) __attribute__((always_inline)) {
        const float bv = bias[cb * 32 + lr];
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] = bv;
    };
    // one GEMM stage over one or two (A, W) operand pairs accumulating into the same columns; `fin(acc, cb)` consumes
    // a finished 32x32 block
    auto stage = [&](const float* A0, int k0p, size_t lda0, int off0, const float* Wa, const float* A1, int k1p, size_t lda1,
                     int off1, const float* Wb, int np, const float* bias, auto fin) __attribute__((always_inline)) {
        const int nb = np >> 7;
        const size_t st = (size_t)np * 16;
        for (int j0 = 0; j0 < nb; j0 += 2) {
            const int cb = wave + 4 * j0;
            const size_t off = ((size_t)cb * 32 + lr) * 16 + lk * 8;
            f32x16 acc[2];
            set_bias(acc[0], bias, cb);
            if (j0 + 1 < nb) {
                set_bias(acc[1], bias, cb + 4);
                JM_WK2(A0, k0p / 16, Wa + off, st, off0, acc, lda0);
                if (A1) JM_WK2(A1, k1p / 16, Wb + off, st, off1, acc, lda1);
                fin(acc[0], cb); fin(acc[1], cb + 4);
            } else {
                JM_WK1(A0, k0p / 16, Wa + off, st, off0, acc, lda0);
                if (A1) JM_WK1(A1, k1p / 16, Wb + off, st, off1, acc, lda1);
                fin(acc[0], cb);
            }
        }
    };
    // accumulator r = 4 rq + t  <->  row 8 rq + 4 lk + t, column cb * 32 + lr
    // ---- stage A: T = tanh(W1 . I + W2 . P + b1 + b2)
    stage(XI, icp, SW_LD, a_off, p.W1, PA, pcp, p_lda, p_off, p.W2, p.np_a, p.ba, [=](const f32x16& a, int cb) __attribute__((always_inline)) {
        float* Tc = T + (size_t)(cb * 32 + lr) * SW_LD + 4 * lk;
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            float4 v;
            v.x = tanhf(a[4 * rq + 0]); v.y = tanhf(a[4 * rq + 1]); v.z = tanhf(a[4 * rq + 2]); v.w = tanhf(a[4 * rq + 3]);
            *reinterpret_cast<float4*>(Tc + 8 * rq) = v;
        }
    });
    lds_barrier();
    if (tid < 32) {                                   // att = sigmoid(fc3(T)) for the tile's 32 points
        float s = p.b3;
        for (int c = 0; c < rc; ++c) s = fmaf(p.w3[c], T[c * SW_LD + tid], s);
        att[tid] = 1.f / (1.f + expf(-s));
    }
    lds_barrier();
    // ---- stage C: G = relu(Wi . I + bi) * att
    stage(XI, icp, SW_LD, a_off, p.Wi, nullptr, 0, 0, 0, nullptr, p.np_c, p.bi, [=](const f32x16& a, int cb) __attribute__((always_inline)) {
        float* Gc = G + (size_t)(cb * 32 + lr) * SW_LD + 4 * lk;
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const float4 w = *reinterpret_cast<const float4*>(att + 8 * rq + 4 * lk);
            float4 v;
            v.x = fmaxf(a[4 * rq + 0], 0.f) * w.x; v.y = fmaxf(a[4 * rq + 1], 0.f) * w.y;
            v.z = fmaxf(a[4 * rq + 2], 0.f) * w.z; v.w = fmaxf(a[4 * rq + 3], 0.f) * w.w;
            *reinterpret_cast<float4*>(Gc + 8 * rq) = v;
        }
    });
    lds_barrier();
    // ---- stage D: out = relu(WfP . P + WfG . G + bf)
    float* outb = p.out + (size_t)bi_ * oc * n + row0;
    stage(PA, pcp, p_lda, p_off, p.WfP, G, pcp, SW_LD, a_off, p.WfG, p.np_d, p.bf, [=](const f32x16& a, int cb) __attribute__((always_inline)) {
        const int col = cb * 32 + lr;
        if (col >= oc) return;
        float* o = outb + (size_t)col * n + 4 * lk;
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            float4 v;
            v.x = fmaxf(a[4 * rq + 0], 0.f); v.y = fmaxf(a[4 * rq + 1], 0.f);
            v.z = fmaxf(a[4 * rq + 2], 0.f); v.w = fmaxf(a[4 * rq + 3], 0.f);
            *reinterpret_cast<float4*>(o + 8 * rq) = v;
        }
    });
}

static size_t li_fusion_lds_bytes(int ic, int pc, int rc, bool p_in_lds) {
    return ((size_t)(pad_to(ic, 16) + (p_in_lds ? pad_to(pc, 16) : 0) + pad_to(rc, 128) + pad_to(pc, 128)) * SW_LD + 32) * sizeof(float);
}
// 1: everything staged in LDS; 2: the point features are read in place (needs pc % 16 == 0); 0: does not fit
static int li_fusion_mode(int ic, int pc, int rc) {
    if (li_fusion_lds_bytes(ic, pc, rc, true) <= 160 * 1024) return 1;
    if (pc % 16 == 0 && li_fusion_lds_bytes(ic, pc, rc, false) <= 160 * 1024) return 2;
    return 0;
}

}  // namespace jm

using namespace jm;

extern "C" int jm_attention_fusion_supported(int b, int n, int ic, int pc, int rc, int oc) {
    if (b < 0 || n < 1 || ic < 1 || pc < 1 || rc < 1 || oc < 1) return 0;
    if (n % 32 || (long long)b * (n / 32) >= (1LL << 31)) return 0;
    return li_fusion_mode(ic, pc, rc) != 0 ? 1 : 0;
}

extern "C" int jm_attention_fusion_forward(int b, int n, int ic, int pc, int rc, int oc, const float* img_feats,
                                           const float* point_feats, const float* w_fc1, const float* w_fc2,
                                           const float* b_fc12, const float* w_fc3, float b_fc3, const float* w_img,
                                           const float* b_img, const float* w_fuse_point, const float* w_fuse_img,
                                           const float* b_fuse, float* out, jm_stream_t stream) {
    JM_REQUIRE(b >= 0 && n >= 0, "attention_fusion: bad sizes");
    if (b == 0 || n == 0) return JM_OK;
    JM_REQUIRE(jm_attention_fusion_supported(b, n, ic, pc, rc, oc), "attention_fusion: unsupported shape (n %% 32 == 0, "
               "pad16(ic) + pad128(rc) + pad128(pc) channels of a 32-point tile must fit the 160 KB LDS)");
    JM_REQUIRE(img_feats && point_feats && w_fc1 && w_fc2 && b_fc12 && w_fc3 && w_img && b_img && w_fuse_point && w_fuse_img &&
               b_fuse && out, "attention_fusion: null pointer");
    JM_REQUIRE(((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(w_fc1) | reinterpret_cast<uintptr_t>(w_fc2) |
                 reinterpret_cast<uintptr_t>(w_img) | reinterpret_cast<uintptr_t>(w_fuse_point) |
                 reinterpret_cast<uintptr_t>(w_fuse_img)) & 15u) == 0, "attention_fusion: 16-byte alignment");
    LiFusionParams p{};
    p.n = n; p.ic = ic; p.pc = pc; p.rc = rc; p.oc = oc;
    p.icp = pad_to(ic, 16); p.pcp = pad_to(pc, 16);
    p.np_a = pad_to(rc, 128); p.np_c = pad_to(pc, 128); p.np_d = pad_to(oc, 128);
    p.I = img_feats; p.P = point_feats;
    p.W1 = w_fc1; p.W2 = w_fc2; p.Wi = w_img; p.WfP = w_fuse_point; p.WfG = w_fuse_img;
    p.ba = b_fc12; p.bi = b_img; p.bf = b_fuse; p.w3 = w_fc3; p.b3 = b_fc3;
    p.out = out; p.tiles_per_frame = n / 32;
    p.p_in_lds = li_fusion_mode(ic, pc, rc) == 1;
    const size_t lds_bytes = li_fusion_lds_bytes(ic, pc, rc, p.p_in_lds != 0);
    (void)hipFuncSetAttribute((const void*)attention_fusion_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipLaunchKernelGGL(attention_fusion_kernel, dim3((unsigned)(b * (n / 32))), dim3(256), lds_bytes, (hipStream_t)stream, p);
    return check_launch("attention_fusion");
}
