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
// LISTED mode (round 4; sa_groups.hip plans it): ball_query back-fills a short neighbour list with its first hit
// (ball_query_gpu.cu:36-40), so a group with d < nsample distinct rows repeats rows whose maximum it already has.  Groups are
// binned by q = ceil(log2 d) into classes of 2^q rows (its first 2^q list entries: the d distinct ones + back-fill); a tile is
// 32 >> q groups of ONE class, the pool runs over 2^q rows, every group's output is written by exactly one tile at its own
// (frame, centre) position.  A row's value depends on (point, centre) only — never on the tile it sits in — and max does not
// depend on multiplicity: the result is bit-identical to the dense mode's, the rows executed drop from ns to 2^q per group.
// The grid is persistent in both modes (tiles strided over <= 512 workgroups; the listed tile count lives in device memory).
// v_mfma_f32_32x32x2_f32: exact-f32 products.  idx == NULL means GroupAll (the group is the whole frame of
// N = nsample points, in order, and new_xyz == NULL: no centre subtraction, pointnet2_utils.py:278-283).
#include "jm_mfma.h"

namespace jm {

// (SW_BM = 32 rows per tile, SW_LD = 36: padded row stride of the k-major LDS tiles; wide_ktiles: jm_mfma.h)
constexpr int SW_KC = 128;                       // first-layer channels per gather chunk
constexpr int SW_XBUF = SW_KC * SW_LD;           // floats per input chunk buffer
#ifndef JM_SW_ST
#define JM_SW_ST 4
#endif
constexpr int SW_ST = JM_SW_ST;                  // weight k-tiles requested per group, one group ahead of the MFMAs (jm_mfma.h: wide_ktiles_deep)

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
    int groups;              // B * M
    int qfull;               // log2(ns)
    int dense_tiles;         // dense mode: groups * ns / 32
    const int* cls_count;    // listed mode: [8] groups per class q (device memory, written by sg_plan_kernel), else null
    const int* glist;        // listed mode: class q's groups at glist[q * groups ...], batch-global group ids b * M + i
};

constexpr int SW_G_OFF = 2 * SW_XBUF;            // (in floats) 32 group ids + 32 output offsets (long long) behind the X buffers
constexpr int SW_G_FLOATS = 32 + 64 + 16;          // + the tile schedule TS[0..9] (padded: the hidden tiles stay 16-byte aligned)

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
    int* GL = reinterpret_cast<int*>(lds + SW_G_OFF);             // this tile's group ids (-1: padding slot)
    long long* GO = reinterpret_cast<long long*>(lds + SW_G_OFF + 32);   // their output offsets (frame * obs + centre)
    float* HA = lds + SW_G_OFF + SW_G_FLOATS;                     // layer-1 output, np0 columns
    float* HB = HA + (size_t)np0 * SW_LD;                         // layer-2 output, np1 columns (L == 3)
    const int C = p.C, Cp = pad_to(C, 16), Nn = p.N;
    const int nchunks = (K0 + SW_KC - 1) / SW_KC;
    const int ns = p.ns, cout = p.cout, Mi = p.M;
    float* outp = p.out;

    // ---- the tile schedule: dense = every group in the full class; listed = class q's tiles after those of the classes below
    // (TS[q] = first tile of class q, TS[8] = all tiles; kept in LDS: as scalars they would not fit next to the k-loops' state)
    int* TS = reinterpret_cast<int*>(lds + SW_G_OFF + 96);
    if (tid == 0) {
        int acc_t = 0;
        for (int c = 0; c < 8; ++c) {
            TS[c] = acc_t;
            if (p.cls_count && c <= p.qfull) acc_t += (int)((((long long)p.cls_count[c] << c) + 31) >> 5);
        }
        TS[8] = p.cls_count ? acc_t : p.dense_tiles;
    }
    lds_barrier();
    const int total = TS[8];
    const int gc = tid >> 5;                                      // channels gc + 8 j of a chunk

    for (int tile = blockIdx.x; tile < total; tile += gridDim.x) {
        if (tid < 32) {
            int q = p.qfull, tl = tile, cnt_q = p.groups;
            if (p.cls_count) {
                // the last class c with TS[c] <= tile (an empty class shares its successor's start and loses to it)
                q = 0;
                for (int c = 1; c <= p.qfull; ++c)
                    if (tile >= TS[c]) q = c;
                tl = tile - TS[q];
                cnt_q = p.cls_count[q];
            }
            const int gpt = 32 >> q;                              // groups per tile
            int g = -1;
            if (tid < gpt) {
                const int slot = tl * gpt + tid;
                if (slot < cnt_q) g = p.glist ? p.glist[(size_t)q * p.groups + slot] : slot;
            }
            GL[tid] = g;
            GO[tid] = g < 0 ? -1 : (long long)((size_t)(g / Mi) * p.obs + (size_t)(g % Mi));
            if (tid == 0) TS[9] = q;
        }
        lds_barrier();
        const int q = __builtin_amdgcn_readfirstlane(TS[9]);
        // ---- this thread's gather row: row r of the tile = sample r & (2^q - 1) of the tile's group r >> q
        const int r32 = tid & 31;
        const int gsl = GL[r32 >> q];
        const int g = gsl < 0 ? GL[0] : gsl;                      // padding rows compute a valid group's row; never stored
        const int smp = r32 & ((1 << q) - 1);
        const int bi = g / Mi;
        const int gidx = p.idx ? p.idx[(size_t)g * ns + smp] : smp;
        float cx = 0.f, cy = 0.f, cz = 0.f;
        if (p.new_xyz) {
            const float* cp = p.new_xyz + (size_t)g * 3;
            cx = cp[0]; cy = cp[1]; cz = cp[2];
        }
        const float* feat_b = p.feat ? p.feat + (size_t)bi * C * Nn : nullptr;
        const float* pt = p.xyz + ((size_t)bi * Nn + gidx) * 3;

        // xyz slots Cp, Cp+1, Cp+2 (Cp % 16 == 0) belong to the threads with gc = 0, 1, 2: one component each, loaded once
        const float myrel = gc < 3 ? pt[gc] - (gc == 0 ? cx : (gc == 1 ? cy : cz)) : 0.f;
        const int kxyz = gc < 3 ? Cp + gc : -1;
        float gg[16];
        // (captures by value, no select among captured variables inside: such a select becomes an indexed load from the
        // closure object, which then lives in scratch together with gg)
        auto issue = [=, &gg](int c) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int k = c * SW_KC + gc + 8 * j;             // slot in the first layer's [features | pad | xyz | pad] order
                const bool isf = k < C;
                const float* src = isf ? feat_b + (size_t)k * Nn + gidx : pt;   // unconditional load, always-valid address
                const float v = *src;
                gg[j] = isf ? v : (k == kxyz ? myrel : 0.f);
            }
        };
        auto park = [=, &gg](float* Xb) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < 16; ++j) Xb[(gc + 8 * j) * SW_LD + (tid & 31)] = gg[j];
        };
        auto set_bias = [=](f32x16& a, const float* bias, int cb) __attribute__((always_inline)) {
            const float bv = bias[cb * 32 + lr];                  // zero padded to np
#pragma unroll
            for (int r = 0; r < 16; ++r) a[r] = bv;
        };
        auto store_hidden = [=](const f32x16& a, float* H, int cb) __attribute__((always_inline)) {
            float* Hc = H + (size_t)(cb * 32 + lr) * SW_LD + 4 * lk;
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {                      // accumulator r = 4 rq + t  <->  row 8 rq + 4 lk + t
                float4 v;
                v.x = fmaxf(a[4 * rq + 0], 0.f); v.y = fmaxf(a[4 * rq + 1], 0.f);
                v.z = fmaxf(a[4 * rq + 2], 0.f); v.w = fmaxf(a[4 * rq + 3], 0.f);
                *reinterpret_cast<float4*>(Hc + 8 * rq) = v;
            }
        };
        // max over each group's 2^q rows straight from the accumulator layout (a[4 rq + t] = row 8 rq + 4 lk + t), ReLU, store
        auto put = [=](int slot, float v, int col) __attribute__((always_inline)) {
            const long long o = GO[slot];
            if (o >= 0) outp[(size_t)o + (size_t)col * (size_t)Mi] = fmaxf(v, 0.f);
        };
        auto store_out = [=](const f32x16& a, int cb) __attribute__((always_inline)) {
            const int col = cb * 32 + lr;
            if (col >= cout) return;
            if (q >= 3) {
                float h[4];                                       // rows 8 rq .. 8 rq + 7: this lane's quad and the other half's
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const float v = fmaxf(fmaxf(a[4 * rq], a[4 * rq + 1]), fmaxf(a[4 * rq + 2], a[4 * rq + 3]));
                    h[rq] = fmaxf(v, __shfl_xor(v, 32));
                }
                if (lk == 0) {
                    if (q == 3) { put(0, h[0], col); put(1, h[1], col); put(2, h[2], col); put(3, h[3], col); }
                    else if (q == 4) { put(0, fmaxf(h[0], h[1]), col); put(1, fmaxf(h[2], h[3]), col); }
                    else put(0, fmaxf(fmaxf(h[0], h[1]), fmaxf(h[2], h[3])), col);
                }
            } else if (q == 2) {
#pragma unroll
                for (int rq = 0; rq < 4; ++rq)                    // rows 8 rq + 4 lk .. + 3 = group 2 rq + lk
                    put(2 * rq + lk, fmaxf(fmaxf(a[4 * rq], a[4 * rq + 1]), fmaxf(a[4 * rq + 2], a[4 * rq + 3])), col);
            } else if (q == 1) {
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    put(4 * rq + 2 * lk, fmaxf(a[4 * rq], a[4 * rq + 1]), col);
                    put(4 * rq + 2 * lk + 1, fmaxf(a[4 * rq + 2], a[4 * rq + 3]), col);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) put(8 * (r >> 2) + 4 * lk + (r & 3), a[r], col);
            }
        };
        // one (pair of) column block(s) of a layer whose whole input sits in LDS
        auto dense_layer = [&](const float* A, int kp, int np, const float* W, const float* bias, float* H, bool final_layer)
                               __attribute__((always_inline)) {
            const int nb = np >> 7;                               // blocks per wave
            const size_t st = (size_t)np * 16;
            for (int j0 = 0; j0 < nb; j0 += 2) {
                const int cb = wave + 4 * j0;
                const float* bp = W + ((size_t)cb * 32 + lr) * 16 + lk * 8;
                f32x16 acc[2];
                set_bias(acc[0], bias, cb);
                if (j0 + 1 < nb) {
                    set_bias(acc[1], bias, cb + 4);
                    wide_ktiles_deep<2, SW_ST>(A, kp / 16, bp, st, a_off, acc);
                    if (final_layer) { store_out(acc[0], cb); store_out(acc[1], cb + 4); }
                    else { store_hidden(acc[0], H, cb); store_hidden(acc[1], H, cb + 4); }
                } else {
                    wide_ktiles_deep<1, SW_ST>(A, kp / 16, bp, st, a_off, acc);
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
            if (more) issue(c + 1);                               // loads in flight under this chunk's MFMAs
            const int kc = min(SW_KC, K0 - c * SW_KC);            // multiple of 16
            const float* bp = W0 + ((size_t)c * (SW_KC / 16) * np0 + wave * 32 + lr) * 16 + lk * 8;
            if (nb0 > 1) wide_ktiles_deep<2, SW_ST>(X + (c & 1) * SW_XBUF, kc / 16, bp, (size_t)np0 * 16, a_off, acc0);
            else wide_ktiles_deep<1, SW_ST>(X + (c & 1) * SW_XBUF, kc / 16, bp, (size_t)np0 * 16, a_off, acc0);
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
        lds_barrier();                                            // the next tile reuses GL / GO, X and the hidden tiles
    }
}

size_t sa_wide_lds_bytes(int L, const int* widths) {
    size_t f = 2 * (size_t)SW_XBUF + SW_G_FLOATS;
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

// cls_count / glist: the listed mode's plan (sa_groups.hip), both null for the dense mode
int sa_mlp_wide_launch(int b, int n, int m, int c, int nsample, const float* xyz, const float* new_xyz,
                       const float* features, const int* idx, int L, const int* widths, const float* const* weights,
                       const float* const* biases, float* out, size_t obs, hipStream_t s, const int* cls_count, const int* glist) {
    const char* why = sa_wide_unsupported(b, n, m, c, nsample, idx == nullptr, L, widths);
    JM_REQUIRE(why == nullptr, "sa_mlp (wide): unsupported shape, needs %s", why);
    JM_REQUIRE((cls_count == nullptr) == (glist == nullptr), "sa_mlp (wide): class counts and group list are both given or both NULL");
    JM_REQUIRE(!cls_count || idx, "sa_mlp (wide): the listed mode needs neighbour lists");
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
    const int mg = idx ? m : 1;                                    // GroupAll: one group per frame
    p.M = mg;
    p.obs = obs ? obs : (size_t)p.cout * (size_t)mg;
    JM_REQUIRE((long long)b * m * nsample < (1LL << 31), "sa_mlp: too many rows");
    p.groups = b * mg;
    p.qfull = nsample == 32 ? 5 : 4;
    p.dense_tiles = (int)((long long)p.groups * nsample / SW_BM);
    p.cls_count = cls_count; p.glist = glist;
    // persistent: the listed mode's tile count is data dependent (<= the dense count + one partial tile per class), and
    // thousands of large-LDS workgroups that only exit cost ~40 ns each to launch
    const long long bound = (long long)p.dense_tiles + (cls_count ? p.qfull + 1 : 0);
    const size_t lds_bytes = sa_wide_lds_bytes(L, widths);
    const int per_cu = lds_bytes <= 80 * 1024 ? 2 : 1;
    const int grid = (int)(bound < 256 * per_cu ? bound : 256 * per_cu);
    if (L == 3) {
        (void)hipFuncSetAttribute((const void*)sa_mlp_wide_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        hipLaunchKernelGGL(sa_mlp_wide_kernel<3>, dim3((unsigned)grid), dim3(256), lds_bytes, s, p);
    } else {
        (void)hipFuncSetAttribute((const void*)sa_mlp_wide_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        hipLaunchKernelGGL(sa_mlp_wide_kernel<2>, dim3((unsigned)grid), dim3(256), lds_bytes, s, p);
    }
    return check_launch("sa_mlp_wide");
}

}  // namespace jm
