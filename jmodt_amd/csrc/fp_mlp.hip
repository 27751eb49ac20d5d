// fp_mlp.hip — fused feature-propagation block on the fp32 matrix cores (gfx950):
//     inverse-distance weights of the 3 nearest coarse points  ->  3-tap interpolation  ->  skip concatenation
//     ->  [1x1 conv + BN(eval, folded) + ReLU] x 2
// Replaces the body of PointnetFPModule.forward after three_nn (jmodt/ops/pointnet2/pointnet2_modules.py:147-164):
// 4 element-wise kernels for the weights (:148-150), three_interpolate (a (B, C2, n) tensor — 171 MB at the finest RPN
// level), torch.cat with the skip features, and SharedMLP's cuDNN/MIOpen 1x1 convolutions + BatchNorm + ReLU on the
// (B, C2 + C1, n, 1) tensor.  Here one launch; neither the interpolated nor the concatenated tensor exists.
//
// One workgroup (4 waves) per tile of 32 consecutive fine points.  The first layer streams its C2 + C1 input channels
// in chunks of 128 through two LDS buffers: for an interpolated channel a thread reads its row's three taps of the
// coarse feature row (weights and indices of the row live in registers), for a skip channel one coalesced 128-byte
// row segment; chunk c+1 is gathered into registers under the MFMAs of chunk c.  The four waves split the columns of
// both layers (blocks w, w+4, ...; widths up to 512), weights from L1/L2 in the packed layout of jm_sa_mlp_pack.
// v_mfma_f32_32x32x2_f32: exact-f32 products.  Weights w_t = (1 / (sqrt(d2_t) + 1e-8)) / sum: the reference's float32
// expression order (pointnet2_utils.py:98, pointnet2_modules.py:148-150).
#include "jm_mfma.h"

namespace jm {

constexpr int FP_KC = 128;
constexpr int FP_XBUF = FP_KC * SW_LD;

struct FpMlpParams {
    int n, m, c2, c1;                 // fine / coarse points per frame, coarse (interpolated) / skip channels
    int k0p, np0, np1, kp1;           // pad16(c2 + c1), pad128(h1), pad128(h2), pad16(h1)
    const float* dist2;               // (B, n, 3) squared distances from jm_three_nn
    const int* idx;                   // (B, n, 3)
    const float* known;               // (B, c2, m)
    const float* skip;                // (B, c1, n) or null
    const float *W0, *W1, *b0, *b1;   // packed
    float* out;                       // (B, h2, n)
    int cout;
    int tiles_per_frame;
};

__global__ void __launch_bounds__(256)
fp_mlp_kernel(FpMlpParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lk = lane >> 5;
    const int a_off = lk * SW_LD + lr;
    const int np0 = p.np0, np1 = p.np1, K0 = p.k0p, n = p.n, m = p.m, c2 = p.c2, c1 = p.c1;
    float* X = lds;                                   // two chunk buffers
    float* H = lds + 2 * FP_XBUF;                     // layer-1 output, np0 columns
    const int bi = blockIdx.x / p.tiles_per_frame;
    const int row0 = (blockIdx.x % p.tiles_per_frame) * SW_BM;
    // ---- this thread's row: the three taps and their normalised inverse-distance weights
    const int r = tid & 31, gc = tid >> 5;
    const size_t rr = ((size_t)bi * n + row0 + r) * 3;
    const int i0 = p.idx[rr], i1 = p.idx[rr + 1], i2 = p.idx[rr + 2];
    float w0 = 1.f / (sqrtf(p.dist2[rr]) + 1e-8f), w1 = 1.f / (sqrtf(p.dist2[rr + 1]) + 1e-8f), w2 = 1.f / (sqrtf(p.dist2[rr + 2]) + 1e-8f);
    {
        const float norm = (w0 + w1) + w2;            // torch.sum over the 3 taps
        w0 = w0 / norm; w1 = w1 / norm; w2 = w2 / norm;
    }
    const float* kb = p.known + (size_t)bi * c2 * m;
    const float* sb = p.skip ? p.skip + (size_t)bi * c1 * n + row0 + r : kb;
    const size_t skip_stride = p.skip ? (size_t)n : 0;      // (no `p` inside the lambdas below: see sa_mlp_wide.hip)
    const int nchunks = (K0 + FP_KC - 1) / FP_KC;

    float g[16];
    auto issue = [=, &g](int c) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int k = c * FP_KC + gc + 8 * j;     // channel of the concatenation [interpolated (c2) | skip (c1)]
            const bool isi = k < c2, iss = (k >= c2) & (k < c2 + c1);
            // unconditional loads on always-valid addresses, selects afterwards
            const float* f = kb + (size_t)(isi ? k : 0) * m;
            const float t0 = f[i0], t1 = f[i1], t2 = f[i2];
            const float sv = sb[(size_t)(iss ? k - c2 : 0) * skip_stride];
            // interpolate_gpu.cu:96: w0 * p0 + w1 * p1 + w2 * p2 in float
            g[j] = isi ? (w0 * t0 + w1 * t1) + w2 * t2 : (iss ? sv : 0.f);
        }
    };
    auto park = [=, &g](float* Xb) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 16; ++j) Xb[(gc + 8 * j) * SW_LD + r] = g[j];
    };
    auto set_bias = [=](f32x16& a, const float* bias, int cb) __attribute__((always_inline)) {
        const float bv = bias[cb * 32 + lr];
#pragma unroll
        for (int q = 0; q < 16; ++q) a[q] = bv;
    };
    auto store_hidden = [=](const f32x16& a, int cb) __attribute__((always_inline)) {
        float* Hc = H + (size_t)(cb * 32 + lr) * SW_LD + 4 * lk;
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {              // accumulator q = 4 rq + t  <->  row 8 rq + 4 lk + t
            float4 v;
            v.x = fmaxf(a[4 * rq + 0], 0.f); v.y = fmaxf(a[4 * rq + 1], 0.f);
            v.z = fmaxf(a[4 * rq + 2], 0.f); v.w = fmaxf(a[4 * rq + 3], 0.f);
            *reinterpret_cast<float4*>(Hc + 8 * rq) = v;
        }
    };
    float* outb = p.out + (size_t)bi * p.cout * n + row0;
    const int cout = p.cout;
    auto store_out = [=](const f32x16& a, int cb) __attribute__((always_inline)) {
        const int col = cb * 32 + lr;
        if (col >= cout) return;
        float* o = outb + (size_t)col * n + 4 * lk;
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            float4 v;
            v.x = fmaxf(a[4 * rq + 0], 0.f); v.y = fmaxf(a[4 * rq + 1], 0.f);
            v.z = fmaxf(a[4 * rq + 2], 0.f); v.w = fmaxf(a[4 * rq + 3], 0.f);
            *reinterpret_cast<float4*>(o + 8 * rq) = v;
        }
    };

    // ---- layer 1 over the input chunks: up to 4 column blocks per wave (widths <= 512), accumulators persist
    f32x16 accA[2], accB[2];
    const int nb0 = np0 >> 7;
    set_bias(accA[0], p.b0, wave);
    if (nb0 > 1) set_bias(accA[1], p.b0, wave + 4);
    if (nb0 > 2) set_bias(accB[0], p.b0, wave + 8);
    if (nb0 > 3) set_bias(accB[1], p.b0, wave + 12);
    issue(0);
    park(X);
    lds_barrier();
    const size_t st0 = (size_t)np0 * 16;
    for (int c = 0; c < nchunks; ++c) {
        const bool more = c + 1 < nchunks;
        if (more) issue(c + 1);                       // loads in flight under this chunk's MFMAs
        const int kc = min(FP_KC, K0 - c * FP_KC);    // multiple of 16
        const float* bp = p.W0 + ((size_t)c * (FP_KC / 16) * np0 + wave * 32 + lr) * 16 + lk * 8;
        const float* A = X + (c & 1) * FP_XBUF;
        if (nb0 > 1) wide_ktiles<2>(A, kc / 16, bp, st0, a_off, accA); else wide_ktiles<1>(A, kc / 16, bp, st0, a_off, accA);
        if (nb0 > 3) wide_ktiles<2>(A, kc / 16, bp + 2 * 2048, st0, a_off, accB);
        else if (nb0 > 2) wide_ktiles<1>(A, kc / 16, bp + 2 * 2048, st0, a_off, accB);
        if (more) { park(X + ((c + 1) & 1) * FP_XBUF); lds_barrier(); }
    }
    store_hidden(accA[0], wave);
    if (nb0 > 1) store_hidden(accA[1], wave + 4);
    if (nb0 > 2) store_hidden(accB[0], wave + 8);
    if (nb0 > 3) store_hidden(accB[1], wave + 12);
    lds_barrier();
    // ---- layer 2, straight to the output
    const int nb1 = np1 >> 7;
    const size_t st1 = (size_t)np1 * 16;
    for (int j0 = 0; j0 < nb1; j0 += 2) {
        const int cb = wave + 4 * j0;
        const float* bp = p.W1 + ((size_t)cb * 32 + lr) * 16 + lk * 8;
        f32x16 acc[2];
        set_bias(acc[0], p.b1, cb);
        if (j0 + 1 < nb1) {
            set_bias(acc[1], p.b1, cb + 4);
            wide_ktiles<2>(H, p.kp1 / 16, bp, st1, a_off, acc);
            store_out(acc[0], cb); store_out(acc[1], cb + 4);
        } else {
            wide_ktiles<1>(H, p.kp1 / 16, bp, st1, a_off, acc);
            store_out(acc[0], cb);
        }
    }
}

static size_t fp_mlp_lds_bytes(int h1) { return (2 * (size_t)FP_XBUF + (size_t)pad_to(h1, 128) * SW_LD) * sizeof(float); }

}  // namespace jm

using namespace jm;

extern "C" int jm_fp_mlp_supported(int b, int n, int m, int c2, int c1, int h1, int h2) {
    if (b < 0 || n < 32 || n % 32 || m < 3 || c2 < 1 || c1 < 0 || h1 < 1 || h2 < 1) return 0;
    if (h1 > 512 || h2 > 512 || (long long)b * (n / 32) >= (1LL << 31)) return 0;
    return fp_mlp_lds_bytes(h1) <= 160 * 1024 ? 1 : 0;
}

extern "C" int jm_fp_mlp_forward(int b, int n, int m, int c2, int c1, int h1, int h2, const float* dist2, const int* idx,
                                 const float* known_feats, const float* skip_feats, const float* w0, const float* b0,
                                 const float* w1, const float* b1, float* out, jm_stream_t stream) {
    JM_REQUIRE(b >= 0, "fp_mlp: bad size");
    if (b == 0) return JM_OK;
    JM_REQUIRE(jm_fp_mlp_supported(b, n, m, c2, c1, h1, h2), "fp_mlp: unsupported shape (n %% 32 == 0, m >= 3, widths <= 512)");
    JM_REQUIRE(dist2 && idx && known_feats && (skip_feats || c1 == 0) && w0 && b0 && w1 && b1 && out, "fp_mlp: null pointer");
    JM_REQUIRE(((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(w0) | reinterpret_cast<uintptr_t>(w1)) & 15u) == 0,
               "fp_mlp: 16-byte alignment");
    FpMlpParams p{};
    p.n = n; p.m = m; p.c2 = c2; p.c1 = c1;
    p.k0p = pad_to(c2 + c1, 16); p.np0 = pad_to(h1, 128); p.np1 = pad_to(h2, 128); p.kp1 = pad_to(h1, 16);
    p.dist2 = dist2; p.idx = idx; p.known = known_feats; p.skip = c1 ? skip_feats : nullptr;
    p.W0 = w0; p.W1 = w1; p.b0 = b0; p.b1 = b1; p.out = out; p.cout = h2; p.tiles_per_frame = n / 32;
    const size_t lds_bytes = fp_mlp_lds_bytes(h1);
    (void)hipFuncSetAttribute((const void*)fp_mlp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipLaunchKernelGGL(fp_mlp_kernel, dim3((unsigned)((long long)b * (n / 32))), dim3(256), lds_bytes, (hipStream_t)stream, p);
    return check_launch("fp_mlp");
}
