// conv_wino.hip — 3x3 / stride 1 / padding 1 convolution + bias + ReLU on channels-last fp32 activations as ONE fused
// Winograd F(2x2, 3x3) kernel (gfx950).
//
// backbone.py:16-32: BasicBlock.conv1 + bn1 + relu of Img_Block[1..3] (64 -> 128 at 192 x 640, 128 -> 256 at 96 x 320,
// 256 -> 512 at 48 x 160; eval-mode BatchNorm folded into weight and bias by the caller): 3 x 145 GFLOP = 60 % of the image
// branch's arithmetic, which is half of the detect step.  As a direct convolution the fp32 MFMA pipe bounds each layer at
// 0.92 ms (157.3 TF); the library's implicit GEMM runs at 0.66-0.81 of that plus a bias/ReLU pass over the output.
// Winograd's minimal filtering needs 16 products per 2x2 output tile and (cin, cout) pair instead of 36: 64.4 GFLOP of
// MFMA work per layer, bound 0.41 ms.  Un-fused (transform kernels + 16 GEMMs + inverse transform) the transformed
// tensors move 6.7 GB per layer and lose to the direct form; here they never leave the CU:
//   * workgroup = 4 waves = 32 tiles (4 x 8: an 8 x 16 pixel output block, 10 x 18 input patch) x 64 output channels;
//     every wave owns ALL 16 transform positions of 32 tiles x 16 channels: 16 x 2 accumulator blocks of
//     v_mfma_f32_16x16x4_f32 (128 registers) — the inverse transform A^T M A is then lane-local arithmetic on the
//     accumulators, no exchange;
//   * the input transform B^T d B is done by the workgroup's 256 threads, one (tile, channel) each per 8-channel chunk:
//     16 buffer loads (out-of-image taps read 0 through the buffer descriptor's range check — no masks), 32 adds, 16 LDS
//     writes into a double-buffered V[16][8][32] (16 KB per buffer; bank = tile + 16 (k & 1) + 4 (k >> 1): writes and
//     MFMA operand reads both conflict-free), one barrier per chunk;
//   * the transformed weights U = G g G^T are packed once, in MFMA B-operand order ([16-channel block][k-step][position
//     group][lane][4]), and stream from L2 straight into registers as fully coalesced 1 KB wave loads, one k-step ahead:
//     no LDS for the weights (16 positions x 16 channels per wave have no reuse inside the workgroup).
// Per 2 k-steps a wave issues 64 MFMAs, 64 ds_read_b32, 8 global_load_dwordx4 and its share of the transform.
#include "jm_common.h"

namespace jm {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int WN_KC = 8;                     // input channels per chunk = two k-steps of the 16x16x4 MFMA
constexpr int WN_TILES = 32;                 // 4 x 8 output tiles of 2 x 2 pixels per workgroup
constexpr int WN_TN = 64;                    // output channels per workgroup (16 per wave)
constexpr int WN_VBUF = 16 * WN_KC * WN_TILES;   // floats per V buffer

__device__ __forceinline__ int wn_voff(int k, int tile) { return k * 32 + ((tile + 16 * (k & 1) + 4 * (k >> 1)) & 31); }

// U[p][q] = sum_ab G[p][a] g[a][b] G[q][b], G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]; one thread per packed float4
__global__ void __launch_bounds__(256)
wino_pack_kernel(int cin, int cout, const float* __restrict__ w, float* __restrict__ up) {
    const int ksteps = cin >> 2;
    const long long total = (long long)(cout >> 4) * ksteps * 4 * 64;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int lane = (int)(i & 63), pg = (int)((i >> 6) & 3);
    const long long r = i >> 8;
    const int kstep = (int)(r % ksteps), nblk = (int)(r / ksteps);
    const int ci = 4 * kstep + (lane >> 4), co = 16 * nblk + (lane & 15);
    const float* g = w + ((size_t)co * cin + ci) * 9;
    double gg[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) gg[a][b] = (double)g[a * 3 + b];
    const double G[4][3] = {{1, 0, 0}, {.5, .5, .5}, {.5, -.5, .5}, {0, 0, 1}};
    float o[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        double s = 0;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) s += G[pg][a] * gg[a][b] * G[q][b];
        o[q] = (float)s;
    }
    reinterpret_cast<f32x4*>(up)[i] = (f32x4){o[0], o[1], o[2], o[3]};
}

// EXP (tools build only, JM_WN_EXP): ablations for tools/conv_wino_bench.py — bit 0: no input path (raw loads + transform),
// bit 1: no weight stream, bit 2: no LDS operand reads.  0 in the product library.
template <int EXP>
__global__ void __launch_bounds__(256, 2)
conv3x3_wino_kernel(int H, int W, int cin, int cout, int patches_x, int patches_y, int npatches, int group, unsigned total_work, unsigned x_bytes,
                    const float* __restrict__ x, const float* __restrict__ up, const float* __restrict__ bias,
                    float* __restrict__ y, int relu) {
    __shared__ __attribute__((aligned(16))) float V[2 * WN_VBUF];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned work = blockIdx.x;
    if ((total_work & 7u) == 0) work = (work & 7u) * (total_work >> 3) + (work >> 3);   // an XCD walks a contiguous range
    // `group` patches share one 64-channel block of U before the next block starts: the workgroups resident on an XCD stream
    // the same 16 * cin * 64 floats (L2-resident) instead of cycling through all of U (8 MB at 256 -> 512: above the 4 MB L2)
    const int nblocks = cout / WN_TN;
    const unsigned per_group = (unsigned)group * (unsigned)nblocks;
    const unsigned grp = work / per_group, rem = work - grp * per_group;
    const unsigned gsize = min((unsigned)group, (unsigned)npatches - grp * (unsigned)group);
    const int nb = (int)(rem / gsize);
    int patch = (int)(grp * (unsigned)group + rem % gsize);
    const int px = patch % patches_x; patch /= patches_x;
    const int py = patch % patches_y;
    const int b = patch / patches_y;

    // ---- input-transform role: one (tile, channel of the chunk) per thread ----
    const int kk = tid & 7, ttile = tid >> 3;
    unsigned roff[16];
    {
        const int oy = (py * 4 + (ttile >> 3)) * 2 - 1, ox = (px * 8 + (ttile & 7)) * 2 - 1;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int iy = oy + i, ix = ox + j;
                const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
                roff[i * 4 + j] = ok ? (unsigned)((((size_t)b * H + iy) * W + ix) * cin + kk) * 4u : 0xFFFFFF00u;
            }
    }
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (int)x_bytes, 0x00020000);
    const int wslot = wn_voff(kk, ttile);

    // ---- MFMA role ----
    const int m = lane & 15, kq = lane >> 4;
    int aoff[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int a = 0; a < 2; ++a) aoff[ks][a] = wn_voff(4 * ks + kq, 16 * a + m);
    const int ksteps = cin >> 2, nch = cin / WN_KC;
    const f32x4* ub = reinterpret_cast<const f32x4*>(up) + ((size_t)(nb * 4 + wave) * ksteps) * 256 + lane;

    f32x4 acc[16][2];
#pragma unroll
    for (int p = 0; p < 16; ++p) acc[p][0] = acc[p][1] = (f32x4){0.f, 0.f, 0.f, 0.f};

    float d[16];
    auto load_raw = [&](int c) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (!(EXP & 1) && (!(EXP & 16) || c == 0)) d[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, roff[i], c * (WN_KC * 4), 0));
    };
    // B^T d B in two passes: columns (d -> t, 16 adds) and rows (t -> the 16 positions, 16 adds + 16 LDS writes); the row pass is
    // issued one row per MFMA stage so that the vector work sits in the shadow of the matrix pipe
    float t[16];
    auto transform_cols = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            t[0 + j] = d[0 + j] - d[8 + j];
            t[4 + j] = d[4 + j] + d[8 + j];
            t[8 + j] = d[8 + j] - d[4 + j];
            t[12 + j] = d[4 + j] - d[12 + j];
        }
    };
    auto transform_row = [&](float* Vb, int i) __attribute__((always_inline)) {
        if (EXP & 1) return;
        if ((EXP & 32) && i < 4) { if (i == 0) Vb[wslot] = t[0] + t[5] + t[10] + t[15]; return; }
        Vb[(i * 4 + 0) * 256 + wslot] = t[i * 4 + 0] - t[i * 4 + 2];
        Vb[(i * 4 + 1) * 256 + wslot] = t[i * 4 + 1] + t[i * 4 + 2];
        Vb[(i * 4 + 2) * 256 + wslot] = t[i * 4 + 2] - t[i * 4 + 1];
        Vb[(i * 4 + 3) * 256 + wslot] = t[i * 4 + 1] - t[i * 4 + 3];
    };
    auto load_b = [&](f32x4 (&bf)[4], int kstep) __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
            if (!(EXP & 2) || kstep == 0) bf[g] = (EXP & 8) ? __builtin_nontemporal_load(ub + (size_t)kstep * 256 + g * 64) : ub[(size_t)kstep * 256 + g * 64];
    };
    // one stage = 4 positions = 8 MFMAs; its A fragments (8 LDS reads) are fetched one stage ahead
    float af[2][8];
    auto load_a = [&](const float* Vb, int stage) __attribute__((always_inline)) {
        const int ks = stage >> 2, g = stage & 3;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (EXP & 4) { af[stage & 1][2 * q] = af[stage & 1][2 * q + 1] = (float)(q + stage); continue; }
            af[stage & 1][2 * q] = Vb[(g * 4 + q) * 256 + aoff[ks][0]];
            af[stage & 1][2 * q + 1] = Vb[(g * 4 + q) * 256 + aoff[ks][1]];
        }
    };
    auto mma = [&](int stage, const f32x4& bf) __attribute__((always_inline)) {
        const int g = stage & 3;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc[g * 4 + q][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[stage & 1][2 * q], bf[q], acc[g * 4 + q][0], 0, 0, 0);
            acc[g * 4 + q][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[stage & 1][2 * q + 1], bf[q], acc[g * 4 + q][1], 0, 0, 0);
        }
    };

    f32x4 b0[4], b1[4];
    load_raw(0);
    load_b(b0, 0);
    transform_cols();
#pragma unroll
    for (int i = 0; i < 4; ++i) transform_row(V, i);
    if (nch > 1) load_raw(1);
    __syncthreads();
    for (int c = 0; c < nch; ++c) {
        const float* Vc = V + (c & 1) * WN_VBUF;
        float* Vn = V + ((c + 1) & 1) * WN_VBUF;
        const bool more = c + 1 < nch;
        load_a(Vc, 0);
        load_b(b1, 2 * c + 1);
        if (more) transform_cols();
#pragma unroll
        for (int st = 0; st < 8; ++st) {
            if (st < 7) load_a(Vc, st + 1);
            if (st == 4 && more) load_b(b0, 2 * c + 2);
            mma(st, st < 4 ? b0[st & 3] : b1[st & 3]);
            if (st >= 4 && more) transform_row(Vn, st - 4);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (c + 2 < nch) load_raw(c + 2);
        __syncthreads();
    }

    // ---- inverse transform Y = A^T M A (A^T = [[1,1,1,0],[0,1,-1,-1]]), bias, ReLU, store ----
    const int co = nb * WN_TN + wave * 16 + m;
    const float bv = bias ? bias[co] : 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int tile = 16 * a + 4 * kq + i;
            const int oy = (py * 4 + (tile >> 3)) * 2, ox = (px * 8 + (tile & 7)) * 2;
            float r0[4], r1[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                r0[q] = acc[0 + q][a][i] + acc[4 + q][a][i] + acc[8 + q][a][i];
                r1[q] = acc[4 + q][a][i] - acc[8 + q][a][i] - acc[12 + q][a][i];
            }
            float o00 = r0[0] + r0[1] + r0[2] + bv, o01 = r0[1] - r0[2] - r0[3] + bv;
            float o10 = r1[0] + r1[1] + r1[2] + bv, o11 = r1[1] - r1[2] - r1[3] + bv;
            if (relu) { o00 = fmaxf(o00, 0.f); o01 = fmaxf(o01, 0.f); o10 = fmaxf(o10, 0.f); o11 = fmaxf(o11, 0.f); }
            if (oy < H && ox < W) {
                float* o = y + (((size_t)b * H + oy) * W + ox) * cout + co;
                o[0] = o00;
                if (ox + 1 < W) o[cout] = o01;
                if (oy + 1 < H) {
                    o[(size_t)W * cout] = o10;
                    if (ox + 1 < W) o[(size_t)W * cout + cout] = o11;
                }
            }
        }
}

}  // namespace jm

using namespace jm;

extern "C" size_t jm_conv3x3_wino_packed_elems(int cin, int cout) { return (size_t)16 * (size_t)cin * (size_t)cout; }

extern "C" int jm_conv3x3_wino_supported(int cin, int cout) { return cin >= WN_KC && cin % WN_KC == 0 && cout >= WN_TN && cout % WN_TN == 0; }

extern "C" int jm_conv3x3_wino_pack(int cin, int cout, const float* weight, float* packed, jm_stream_t stream) {
    JM_REQUIRE(jm_conv3x3_wino_supported(cin, cout), "conv3x3_wino_pack: cin %% 8 == 0 and cout %% 64 == 0");
    JM_REQUIRE(weight && packed, "conv3x3_wino_pack: null pointer");
    JM_REQUIRE((reinterpret_cast<uintptr_t>(packed) & 15u) == 0, "conv3x3_wino_pack: 16-byte alignment");
    const long long total = (long long)(cout >> 4) * (cin >> 2) * 256;
    hipLaunchKernelGGL(wino_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, cin, cout, weight, packed);
    return check_launch("conv3x3_wino_pack");
}

extern "C" int jm_conv3x3_wino_bias_relu(int b, int h, int w, int cin, int cout, const float* x_channels_last,
                                         const float* packed, const float* bias, int relu, float* out_channels_last,
                                         jm_stream_t stream) {
    JM_REQUIRE(b >= 0 && h >= 0 && w >= 0, "conv3x3_wino: negative size");
    JM_REQUIRE(jm_conv3x3_wino_supported(cin, cout), "conv3x3_wino: cin %% 8 == 0 and cout %% 64 == 0");
    if (b == 0 || h == 0 || w == 0) return JM_OK;
    JM_REQUIRE(x_channels_last && packed && out_channels_last, "conv3x3_wino: null pointer");
    JM_REQUIRE(((reinterpret_cast<uintptr_t>(packed) | reinterpret_cast<uintptr_t>(x_channels_last)) & 15u) == 0, "conv3x3_wino: 16-byte alignment");
    const unsigned long long xb = (unsigned long long)b * h * w * cin * 4ull;
    JM_REQUIRE(xb < 0xFFFFFF00ull, "conv3x3_wino: input above 4 GB");
    const int tx = divup(w, 2), ty = divup(h, 2), pxs = divup(tx, 8), pys = divup(ty, 4);
    const unsigned long long total = (unsigned long long)b * pxs * pys * (cout / WN_TN);
    JM_REQUIRE(total < 0x7FFFFFFFull, "conv3x3_wino: grid limit");
    const int npatches = b * pxs * pys, group = tune_env("JM_WN_G", 16);
#define WN_LAUNCH(E) hipLaunchKernelGGL(conv3x3_wino_kernel<E>, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream, h, w, cin, cout, pxs, pys, npatches, group, \
                                        (unsigned)total, (unsigned)xb, x_channels_last, packed, bias, out_channels_last, relu)
#ifdef JM_TOOLS_BUILD
    switch (tune_env("JM_WN_EXP", 0)) {
        case 1: WN_LAUNCH(1); break;
        case 2: WN_LAUNCH(2); break;
        case 3: WN_LAUNCH(3); break;
        case 4: WN_LAUNCH(4); break;
        case 7: WN_LAUNCH(7); break;
        case 8: WN_LAUNCH(8); break;
        case 16: WN_LAUNCH(16); break;
        case 32: WN_LAUNCH(32); break;
        default: WN_LAUNCH(0);
    }
#else
    WN_LAUNCH(0);
#endif
    return check_launch("conv3x3_wino");
}
