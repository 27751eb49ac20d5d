// sa_mlp_wide.hip — the fused set-abstraction block for WIDE MLPs on few rows (gfx950, fp32 MFMA):
//     group (xyz - centre | features)  ->  [1x1 conv + BN(eval, folded) + ReLU] x L  ->  max over nsample
// Same contract as sa_mlp.hip (it replaces the per-scale body of _PointnetSAModuleBase.forward,
// jmodt/ops/pointnet2/pointnet2_modules.py:46-52, and GroupAll + SharedMLP + max_pool2d for the RCNN's last
// level, pointnet2_utils.py:267-290), for the levels that kernel does not take: hidden widths up to 512
// (RPN SA3 [259,128,196,256] x2, SA4 [515,256,256,512] / [515,256,384,512], RCNN SA3 [259,256,256,512],
// config.py:78-82,137-139).  Those levels have 8 k .. 64 k rows and 5 .. 21 GFLOP: the reference's materialised
// (B, 3+C, npoint, nsample) tensors + three cuDNN/MIOpen 1x1 convolutions + BN + ReLU + max cost ~10 launches and
// several passes over the grouped tensor each; here one launch per scale, nothing leaves the chip.
//
// One workgroup (4 waves, one per SIMD) per tile of 32 consecutive (centre, sample) rows = one 32x32 MFMA row
// block; the four waves split the COLUMNS of every layer (wave w owns 32-column blocks w, w+4, ...; up to 4 each,
// so layer widths up to 512), all read the same A operand from LDS:
//   * layer 1 streams its 3+C input channels in chunks of 128 through two LDS buffers: every thread gathers 16
//     elements of chunk c+1 (idx -> feature rows; xyz - centre) into registers before the MFMAs of chunk c and
//     parks them afterwards; accumulators (bias-initialised) persist over the chunks;
//   * hidden activations (ReLU in the epilogue) go to two further k-major LDS buffers sized pad32(width);
//   * weights come straight from L1/L2 in the packed layout of jm_sa_mlp_pack (one lane's B operand of a k-tile =
//     8 consecutive floats), register double-buffered one k-tile ahead;
//   * last layer: max over each centre's nsample rows out of the accumulator layout, ReLU, (B, cout, M) store.
// v_mfma_f32_32x32x2_f32: exact-f32 products.  idx == NULL means GroupAll (the group is the whole frame of
// N = nsample points, in order, and new_xyz == NULL: no centre subtraction, pointnet2_utils.py:278-283).
#include "jm_mfma.h"

namespace jm {

// (SW_BM = 32 rows per tile, SW_LD = 36: padded row stride of the k-major LDS tiles; wide_ktiles: jm_mfma.h)
constexpr int SW_KC = 128;                       // first-layer channels per gather chunk
constexpr int SW_XBUF = SW_KC * SW_LD;           // floats per input chunk buffer

struct SaWideParams {
    int N, M, C, ns;
    const float* xyz;        // (B,N,3)
    const float* new_xyz;    // (B,M,3) or null (GroupAll)
    const float* feat;       // (B,C,N) or null
    const int* idx;          // (B,M,ns) or null (GroupAll: point r of the frame)
    int L;                   // 2 or 3
    int kp[4];               // kp[l] = padded input channels of layer l (kp[0]: first-layer layout of sa_mlp_pack)
    int np[3];               // pad128(width_{l+1}): packed rows of layer l
    const float* W[3];
    const float* bias[3];
    float* out;              // (B, cout, M), frame stride obs
    int cout;
    size_t obs;
    int rows_per_frame;
};

// Column blocks: a layer's packed weights / biases are zero padded to np = pad128(width) columns, i.e. nb = np / 128
// blocks for EACH of the four waves (block w + 4 j of wave w); padding blocks produce zeros.  Blocks are processed
// two at a time (two independent accumulator chains), so one instantiation pair serves every width.
template <int L>
__global__ void __launch_bounds__(256)
sa_mlp_wide_kernel(SaWideParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lk = lane >> 5;
    const int a_off = lk * SW_LD + lr;
    const int np0 = p.np[0], np1 = p.np[1], np2 = p.np[2];
    const float *W0 = p.W[0], *W1 = p.W[1], *W2 = p.W[2], *bs0 = p.bias[0], *bs1 = p.bias[1], *bs2 = p.bias[2];
    const int kp1 = p.kp[1], kp2 = p.kp[2], K0 = p.kp[0];
    float* X = lds;                                               // two chunk buffers
    float* HA = lds + 2 * SW_XBUF;                                // layer-1 output, np0 columns
    float* HB = HA + (size_t)np0 * SW_LD;                         // layer-2 output, np1 columns (L == 3)

    // ---- this thread's gather row
    const unsigned R = blockIdx.x * SW_BM + (tid & 31);            // batch-global row (host: total rows < 2^31)
    const int gc = tid >> 5;                                      // channels gc + 8 j of a chunk
    const int bi = (int)(R / (unsigned)p.rows_per_frame);
    const int within = (int)(R % (unsigned)p.rows_per_frame);
    const int gidx = p.idx ? p.idx[R] : within % p.ns;
    float cx = 0.f, cy = 0.f, cz = 0.f;
    if (p.new_xyz) {
        const float* cp = p.new_xyz + ((size_t)bi * p.M + within / p.ns) * 3;
        cx = cp[0]; cy = cp[1]; cz = cp[2];
    }
    const int C = p.C, Cp = pad_to(C, 16), Nn = p.N;
    const float* feat_b = p.feat ? p.feat + (size_t)bi * C * Nn : nullptr;
    const float* pt = p.xyz + ((size_t)bi * Nn + gidx) * 3;
    const int nchunks = (K0 + SW_KC - 1) / SW_KC;

    // xyz slots Cp, Cp+1, Cp+2 (Cp % 16 == 0) belong to the threads with gc = 0, 1, 2: one component each, loaded once
    const float myrel = gc < 3 ? pt[gc] - (gc == 0 ? cx : (gc == 1 ? cy : cz)) : 0.f;
    const int kxyz = gc < 3 ? Cp + gc : -1;
    float g[16];
    // (captures by value, no select among captured variables inside: such a select becomes an indexed load from the
    // closure object, which then lives in scratch together with g)
    auto issue = [=, &g](int c) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int k = c * SW_KC + gc + 8 * j;                 // slot in the first layer's [features | pad | xyz | pad] order
            const bool isf = k < C;
            const float* src = isf ? feat_b + (size_t)k * Nn + gidx : pt;   // unconditional load, always-valid address
            const float v = *src;
            g[j] = isf ? v : (k == kxyz ? myrel : 0.f);
        }
    };
    auto park = [=, &g](float* Xb) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 16; ++j) Xb[(gc + 8 * j) * SW_LD + (tid & 31)] = g[j];
    };
    auto set_bias = [=](f32x16& a, const float* bias, int cb) __attribute__((always_inline)) {
        const float bv = bias[cb * 32 + lr];                      // zero padded to np
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] = bv;
    };
    auto store_hidden = [=](const f32x16& a, float* H, int cb) __attribute__((always_inline)) {
        float* Hc = H + (size_t)(cb * 32 + lr) * SW_LD + 4 * lk;
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {                          // accumulator r = 4 rq + t  <->  row 8 rq + 4 lk + t
            float4 v;
            v.x = fmaxf(a[4 * rq + 0], 0.f); v.y = fmaxf(a[4 * rq + 1], 0.f);
            v.z = fmaxf(a[4 * rq + 2], 0.f); v.w = fmaxf(a[4 * rq + 3], 0.f);
            *reinterpret_cast<float4*>(Hc + 8 * rq) = v;
        }
    };
    // one (pair of) column block(s) of a layer whose whole input sits in LDS
    const unsigned G0 = blockIdx.x * SW_BM / (unsigned)p.ns;      // first group (centre) of this tile, batch-global
    const int ns = p.ns, cout = p.cout;
    const unsigned Mu = (unsigned)p.M;
    float* outp = p.out;
    auto store_out = [=](const f32x16& a, int cb) __attribute__((always_inline)) {
        // max over each centre's nsample rows straight from the accumulator layout (rows 0-15 are r < 8), then ReLU
        const int col = cb * 32 + lr;
        float t0 = -INFINITY, t1 = -INFINITY;
#pragma unroll
        for (int r = 0; r < 8; ++r) { t0 = fmaxf(t0, a[r]); t1 = fmaxf(t1, a[r + 8]); }
        t0 = fmaxf(t0, __shfl_xor(t0, 32));                       // the other lane half holds the rows + 4
        t1 = fmaxf(t1, __shfl_xor(t1, 32));
        if (lk == 0 && col < cout) {
            if (ns == 32) {
                outp[(size_t)(G0 / Mu) * p.obs + (size_t)col * Mu + (G0 % Mu)] = fmaxf(fmaxf(t0, t1), 0.f);
            } else {
                const unsigned Gb = G0 + 1;
                outp[(size_t)(G0 / Mu) * p.obs + (size_t)col * Mu + (G0 % Mu)] = fmaxf(t0, 0.f);
                outp[(size_t)(Gb / Mu) * p.obs + (size_t)col * Mu + (Gb % Mu)] = fmaxf(t1, 0.f);
            }
        }
    };
    auto dense_layer = [&](const float* A, int kp, int np, const float* W, const float* bias, float* H, bool final_layer)
                           __attribute__((always_inline)) {
        const int nb = np >> 7;                                   // blocks per wave
        const size_t st = (size_t)np * 16;
        for (int j0 = 0; j0 < nb; j0 += 2) {
            const int cb = wave + 4 * j0;
            const float* bp = W + ((size_t)cb * 32 + lr) * 16 + lk * 8;
            f32x16 acc[2];
            set_bias(acc[0], bias, cb);
            if (j0 + 1 < nb) {
                set_bias(acc[1], bias, cb + 4);
                wide_ktiles<2>(A, kp / 16, bp, st, a_off, acc);
                if (final_layer) { store_out(acc[0], cb); store_out(acc[1], cb + 4); }
                else { store_hidden(acc[0], H, cb); store_hidden(acc[1], H, cb + 4); }
            } else {
                wide_ktiles<1>(A, kp / 16, bp, st, a_off, acc);
                if (final_layer) store_out(acc[0], cb); else store_hidden(acc[0], H, cb);
            }
        }
    };

    // ---- layer 1 over the input chunks (np0 <= 256: at most two blocks per wave, accumulators persist)
    f32x16 acc0[2];
    const int nb0 = np0 >> 7;
    set_bias(acc0[0], bs0, wave);
    if (nb0 > 1) set_bias(acc0[1], bs0, wave + 4);
    issue(0);
    park(X);
    lds_barrier();
    for (int c = 0; c < nchunks; ++c) {
        const bool more = c + 1 < nchunks;
        if (more) issue(c + 1);                                   // loads in flight under this chunk's MFMAs
        const int kc = min(SW_KC, K0 - c * SW_KC);                // multiple of 16
        const float* bp = W0 + ((size_t)c * (SW_KC / 16) * np0 + wave * 32 + lr) * 16 + lk * 8;
        if (nb0 > 1) wide_ktiles<2>(X + (c & 1) * SW_XBUF, kc / 16, bp, (size_t)np0 * 16, a_off, acc0);
        else wide_ktiles<1>(X + (c & 1) * SW_XBUF, kc / 16, bp, (size_t)np0 * 16, a_off, acc0);
        if (more) { park(X + ((c + 1) & 1) * SW_XBUF); lds_barrier(); }
    }
    store_hidden(acc0[0], HA, wave);
    if (nb0 > 1) store_hidden(acc0[1], HA, wave + 4);
    lds_barrier();
    if (L == 3) {
        dense_layer(HA, kp1, np1, W1, bs1, HB, false);
        lds_barrier();
        dense_layer(HB, kp2, np2, W2, bs2, nullptr, true);
    } else {
        dense_layer(HA, kp1, np1, W1, bs1, nullptr, true);
    }
}

size_t sa_wide_lds_bytes(int L, const int* widths) {
    size_t f = 2 * (size_t)SW_XBUF;
    for (int l = 1; l < L; ++l) f += (size_t)pad_to(widths[l], 128) * SW_LD;
    return f * sizeof(float);
}

// 0 when the shape fits this kernel, else a message
const char* sa_wide_unsupported(long long b, int n, int m, int c, int nsample, int group_all, int L, const int* widths) {
    if (L != 2 && L != 3) return "2 or 3 layers";
    if (nsample != 16 && nsample != 32) return "nsample in {16, 32}";
    if (((long long)b * m * nsample) % SW_BM) return "B*npoint*nsample % 32 == 0";
    if (group_all && (m != 1 || nsample != n)) return "GroupAll needs npoint == 1 and nsample == N";
    if (widths[1] < 1 || widths[1] > 256) return "first layer width <= 256";
    for (int l = 2; l <= L; ++l)
        if (widths[l] < 1 || widths[l] > 512) return "layer widths <= 512";
    if (sa_wide_lds_bytes(L, widths) > 160 * 1024) return "hidden activations exceed the 160 KB LDS";
    return nullptr;
}

int sa_mlp_wide_launch(int b, int n, int m, int c, int nsample, const float* xyz, const float* new_xyz,
                       const float* features, const int* idx, int L, const int* widths, const float* const* weights,
                       const float* const* biases, float* out, size_t obs, hipStream_t s) {
    const char* why = sa_wide_unsupported(b, n, m, c, nsample, idx == nullptr, L, widths);
    JM_REQUIRE(why == nullptr, "sa_mlp (wide): unsupported shape, needs %s", why);
    SaWideParams p{};
    p.N = n; p.M = m; p.C = c; p.ns = nsample;
    p.xyz = xyz; p.new_xyz = new_xyz; p.feat = features; p.idx = idx; p.L = L;
    p.kp[0] = pad_to(widths[0] - 3, 16) + 16;                      // == sa_first_kp(widths[0]) of sa_mlp.hip
    for (int l = 1; l <= L; ++l) p.kp[l] = pad_to(widths[l], 16);
    for (int l = 0; l < L; ++l) {
        JM_REQUIRE(weights[l] && biases[l], "sa_mlp: null layer %d", l);
        JM_REQUIRE((reinterpret_cast<uintptr_t>(weights[l]) & 15u) == 0, "sa_mlp: weights must be 16-byte aligned");
        p.W[l] = weights[l]; p.bias[l] = biases[l];
        p.np[l] = pad_to(widths[l + 1], 128);
    }
    p.out = out; p.cout = widths[L];
    p.obs = obs ? obs : (size_t)p.cout * (size_t)(idx ? m : 1);
    JM_REQUIRE((long long)b * m * nsample < (1LL << 31), "sa_mlp: too many rows");
    p.rows_per_frame = m * nsample;
    const long long tiles = (long long)b * m * nsample / SW_BM;
    const size_t lds_bytes = sa_wide_lds_bytes(L, widths);
    if (L == 3) {
        (void)hipFuncSetAttribute((const void*)sa_mlp_wide_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        hipLaunchKernelGGL(sa_mlp_wide_kernel<3>, dim3((unsigned)tiles), dim3(256), lds_bytes, s, p);
    } else {
        (void)hipFuncSetAttribute((const void*)sa_mlp_wide_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        hipLaunchKernelGGL(sa_mlp_wide_kernel<2>, dim3((unsigned)tiles), dim3(256), lds_bytes, s, p);
    }
    return check_launch("sa_mlp_wide");
}

}  // namespace jm
