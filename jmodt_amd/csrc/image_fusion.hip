// image_fusion.hip — the final LI-Fusion image feature AT THE POINTS, without the full-resolution map (gfx950).
//
// Reference (jmodt/detection/modeling/backbone.py:187-195): every image pyramid level is up-sampled to the input
// resolution by a kernel==stride transposed convolution (16 channels each), the four results are concatenated
// (B, 64, 384, 1280), a 1x1 convolution + BatchNorm + ReLU makes the (B, 32, 384, 1280) "img_fusion" map, and
// feature_gather (bilinear, zeros padding) reads it at the 16384 points of each frame: 241 GFLOP of dense
// convolution and ~8 GB of traffic per batch of 8 for 4 x 16384 x 8 = 524288 bilinear taps — 13 % of the pixels.
//
// Both convolutions are linear and a kernel==stride transposed convolution writes every output pixel from exactly
// ONE input pixel, so the fused map at pixel (y, x) is
//     F[:, y, x] = relu(bias + sum_i Wc_i[:, :, y % k_i, x % k_i]^T . img_i[:, y / k_i, x / k_i])
// with Wc_i = the level's deconvolution weight composed with its slice of the (BatchNorm-folded) 1x1 convolution
// (C_i x 32 per sub-pixel phase; composed once on the host side of the caller).  Only the pixels under a bilinear tap
// are evaluated:
//   taps     one thread per point: 4 taps (pixel, weight, phase key = (y % 16, x % 16)), block-aggregated histogram
//   scan     256 buckets -> bucket / tile offsets (one workgroup)
//   scatter  taps sorted by phase (block-aggregated slots: one global atomic per block and phase)
//   gemm     one WAVE per tile of 32 taps of one phase: 32 taps x 960 channels x 32 outputs on v_mfma_f32_32x32x2_f32;
//            A operand = each tap's channels-last feature vectors (8 consecutive floats per lane and k-tile, straight
//            from L2 — the pyramid maps are 31..252 MB and mostly cache resident), B operand = the phase's packed
//            weights (every tap of the tile shares them — that is what the sort buys); no LDS, ~60 VGPRs, many waves
//            per SIMD hide the gather latency; epilogue: + bias, ReLU, x bilinear weight -> tap value row
//   combine  out[b, :, n] = sum of the point's 4 tap rows (zeros padding: invalid taps are skipped)
// Deterministic: every tap is computed independently and lands in its own row, whatever order the sort produced.
#include "jm_mfma.h"

namespace jm {

constexpr int IF_MAXLV = 4;

struct ImgFusionParams {
    int B, N, H, W, Q;                       // frames, points per frame, canvas, output channels (<= 32)
    int nlv, kmax;                           // pyramid levels, largest stride (phases = kmax^2)
    int C[IF_MAXLV], shift[IF_MAXLV], Hl[IF_MAXLV], Wl[IF_MAXLV];   // channels (multiple of 16), log2 stride, level size
    const float* map[IF_MAXLV];              // (B, Hl, Wl, C) channels-last
    const float* wp[IF_MAXLV];               // [k*k phases][C/16][32][2][8]
    const float* bias;                       // (32) zero padded
    const float* xy;                         // (B, N, 2)
    int *key, *pos, *sorted, *hist, *cursor, *bstart, *tstart;   // workspace
    float *tw, *tapval;
    float* out;                              // (B, Q, N)
};

// bilinear taps of grid_sample(align_corners=True, padding_mode='zeros') — the same arithmetic as feature_gather.hip
__global__ void __launch_bounds__(256)
if_taps_kernel(ImgFusionParams p) {
    __shared__ int lh[256];
    const int nkeys = p.kmax * p.kmax;
    for (int i = threadIdx.x; i < nkeys; i += 256) lh[i] = 0;
    __syncthreads();
    const long long pt = (long long)blockIdx.x * 256 + threadIdx.x;
    if (pt < (long long)p.B * p.N) {
        const float gx = p.xy[pt * 2], gy = p.xy[pt * 2 + 1];
        const float ix = (gx + 1.f) * 0.5f * (float)(p.W - 1), iy = (gy + 1.f) * 0.5f * (float)(p.H - 1);
        const float fx = floorf(ix), fy = floorf(iy);
        const int x0 = (int)fx, y0 = (int)fy;
        const float wx1 = ix - fx, wy1 = iy - fy, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int x = x0 + (j & 1), y = y0 + (j >> 1);
            const float w = ((j >> 1) ? wy1 : wy0) * ((j & 1) ? wx1 : wx0);
            const bool ok = x >= 0 && x < p.W && y >= 0 && y < p.H;
            const int k = ok ? (y % p.kmax) * p.kmax + (x % p.kmax) : -1;
            p.key[pt * 4 + j] = k;
            p.pos[pt * 4 + j] = ok ? ((y << 16) | x) : 0;
            p.tw[pt * 4 + j] = w;
            if (ok) atomicAdd(&lh[k], 1);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nkeys; i += 256)
        if (lh[i]) atomicAdd(&p.hist[i], lh[i]);
}

__global__ void __launch_bounds__(256)
if_scan_kernel(ImgFusionParams p) {   // one workgroup: exclusive scans over <= 256 buckets
    __shared__ int cnt[257], til[257];
    const int nkeys = p.kmax * p.kmax, t = threadIdx.x;
    const int c = t < nkeys ? p.hist[t] : 0;
    cnt[t] = c; til[t] = (c + 31) >> 5;
    __syncthreads();
    if (t == 0) {
        int a = 0, b = 0;
        for (int i = 0; i < nkeys; ++i) { const int ci = cnt[i], ti = til[i]; cnt[i] = a; til[i] = b; a += ci; b += ti; }
        cnt[nkeys] = a; til[nkeys] = b;
    }
    __syncthreads();
    if (t < nkeys) { p.bstart[t] = cnt[t]; p.tstart[t] = til[t]; p.cursor[t] = 0; }
    if (t == 0) { p.bstart[nkeys] = cnt[nkeys]; p.tstart[nkeys] = til[nkeys]; }
}

__global__ void __launch_bounds__(1024)
if_scatter_kernel(ImgFusionParams p) {
    __shared__ int lh[256], base[256];
    const int nkeys = p.kmax * p.kmax;
    for (int i = threadIdx.x; i < nkeys; i += 1024) lh[i] = 0;
    __syncthreads();
    const long long t = (long long)blockIdx.x * 1024 + threadIdx.x;
    const long long T = (long long)p.B * p.N * 4;
    const int k = t < T ? p.key[t] : -1;
    int local = 0;
    if (k >= 0) local = atomicAdd(&lh[k], 1);
    __syncthreads();
    for (int i = threadIdx.x; i < nkeys; i += 1024) base[i] = lh[i] ? atomicAdd(&p.cursor[i], lh[i]) : 0;
    __syncthreads();
    if (k >= 0) p.sorted[p.bstart[k] + base[k] + local] = (int)t;
}

__global__ void __launch_bounds__(64)
if_gemm_kernel(ImgFusionParams p) {
    const int lane = threadIdx.x, lr = lane & 31, lk = lane >> 5;
    const int nkeys = p.kmax * p.kmax;
    const int tile = blockIdx.x;
    if (tile >= p.tstart[nkeys]) return;
    int lo = 0, hi = nkeys;                       // largest k with tstart[k] <= tile
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (p.tstart[mid] <= tile) lo = mid; else hi = mid; }
    const int k = lo;
    const int first = p.bstart[k] + 32 * (tile - p.tstart[k]);
    const int cnt = min(32, p.bstart[k + 1] - first);
    const int tr = p.sorted[first + (lr < cnt ? lr : 0)];          // this lane's tap (row lr); rows >= cnt repeat row 0
    const int pos = p.pos[tr];
    const int b = tr / (4 * p.N), y = pos >> 16, x = pos & 0xffff;
    const int py = k / p.kmax, px = k % p.kmax;

    f32x16 acc;
    {
        const float bv = p.bias[lr];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = bv;
    }
#pragma unroll
    for (int lv = 0; lv < IF_MAXLV; ++lv) {      // unrolled: constant indices into the argument struct (no scratch copy)
        if (lv >= p.nlv) break;
        const int Cl = p.C[lv], sh = p.shift[lv], kl = 1 << sh;
        const float* vec = p.map[lv] + (((size_t)b * p.Hl[lv] + (y >> sh)) * p.Wl[lv] + (x >> sh)) * Cl + 8 * lk;
        const float* wq = p.wp[lv] + (size_t)((py & (kl - 1)) * kl + (px & (kl - 1))) * Cl * 32 + (size_t)lr * 16 + lk * 8;
        const int nkt = Cl >> 4;
        float4 a0 = *reinterpret_cast<const float4*>(vec), a1 = *reinterpret_cast<const float4*>(vec + 4);
        float4 b0 = *reinterpret_cast<const float4*>(wq), b1 = *reinterpret_cast<const float4*>(wq + 4);
        for (int kt = 0; kt < nkt; ++kt) {
            const int nx = kt + 1 < nkt ? kt + 1 : kt;              // unconditional prefetch on a clamped index
            const float4 na0 = *reinterpret_cast<const float4*>(vec + 16 * nx), na1 = *reinterpret_cast<const float4*>(vec + 16 * nx + 4);
            const float4 nb0 = *reinterpret_cast<const float4*>(wq + (size_t)nx * 512), nb1 = *reinterpret_cast<const float4*>(wq + (size_t)nx * 512 + 4);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b0.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b0.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b0.w, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b1.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b1.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b1.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b1.w, acc, 0, 0, 0);
            a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
        }
    }
    // accumulator r = 4 rq + t  <->  tap row 8 rq + 4 lk + t, output channel lr
    const float w_mine = p.tw[tr];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = 8 * (r >> 2) + 4 * lk + (r & 3);
        const int tid_row = __shfl(tr, row);
        const float w_row = __shfl(w_mine, row);
        if (row < cnt) p.tapval[(size_t)tid_row * 32 + lr] = fmaxf(acc[r], 0.f) * w_row;
    }
}

__global__ void __launch_bounds__(256)
if_combine_kernel(ImgFusionParams p) {
    const long long pt = (long long)blockIdx.x * 256 + threadIdx.x;
    if (pt >= (long long)p.B * p.N) return;
    const int b = (int)(pt / p.N), n = (int)(pt % p.N);
    float s[32];
#pragma unroll
    for (int q = 0; q < 32; ++q) s[q] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (p.key[pt * 4 + j] < 0) continue;                      // zeros padding: the tap lies outside the canvas
        const float4* row = reinterpret_cast<const float4*>(p.tapval + (size_t)(pt * 4 + j) * 32);
#pragma unroll
        for (int q4 = 0; q4 < 8; ++q4) {
            const float4 v = row[q4];
            s[4 * q4] += v.x; s[4 * q4 + 1] += v.y; s[4 * q4 + 2] += v.z; s[4 * q4 + 3] += v.w;
        }
    }
#pragma unroll
    for (int q = 0; q < 32; ++q)
        if (q < p.Q) p.out[((size_t)b * p.Q + q) * p.N + n] = s[q];
}

struct IfWs { int *key, *pos, *sorted, *hist, *cursor, *bstart, *tstart; float *tw, *tapval; size_t total; };
static IfWs if_carve(void* ws, long long taps) {
    IfWs w;
    size_t off = 0;
    auto take = [&](size_t bytes) { void* q = ws ? (char*)ws + off : nullptr; off += align_up(bytes, (size_t)256); return q; };
    w.key = (int*)take(taps * 4); w.pos = (int*)take(taps * 4); w.sorted = (int*)take(taps * 4);
    w.tw = (float*)take(taps * 4); w.tapval = (float*)take(taps * 32 * 4);
    w.hist = (int*)take(257 * 4); w.cursor = (int*)take(257 * 4); w.bstart = (int*)take(257 * 4); w.tstart = (int*)take(257 * 4);
    w.total = off;
    return w;
}

}  // namespace jm

using namespace jm;

extern "C" size_t jm_image_fusion_gather_workspace_bytes(int b, int n) {
    if (b < 1 || n < 1) return 0;
    return if_carve(nullptr, (long long)b * n * 4).total;
}

/* weights of one level: wc (cin, q, k, k) row-major [= deconvolution weight composed with the fusion convolution's
 * slice] -> wp [k*k][cin/16][32][2][8]: wp[ph][kt][qq][lk][kk] = wc[16 kt + 8 lk + kk][qq][ph / k][ph % k], zero for qq >= q */
__global__ void if_pack_kernel(int cin, int q, int k, const float* __restrict__ wc, float* __restrict__ wp) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)k * k * cin * 32;
    if (e >= total) return;
    const int kk = e & 7, lk = (e >> 3) & 1, qq = (e >> 4) & 31;
    const long long r = e >> 9;                      // ph * (cin/16) + kt
    const int kt = (int)(r % (cin / 16)), ph = (int)(r / (cin / 16));
    const int c = 16 * kt + 8 * lk + kk;
    wp[e] = qq < q ? wc[(((size_t)c * q + qq) * k + ph / k) * k + ph % k] : 0.f;
}

extern "C" size_t jm_image_fusion_packed_elems(int cin, int k) { return (cin < 16 || k < 1) ? 0 : (size_t)k * k * cin * 32; }

extern "C" int jm_image_fusion_pack(int cin, int q, int k, const float* wc, float* wp, jm_stream_t stream) {
    JM_REQUIRE(cin >= 16 && cin % 16 == 0 && q >= 1 && q <= 32 && k >= 1 && k <= 16, "image_fusion_pack: cin %% 16 == 0, q <= 32, k <= 16");
    JM_REQUIRE(wc && wp, "image_fusion_pack: null pointer");
    const long long total = (long long)k * k * cin * 32;
    hipLaunchKernelGGL(if_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, cin, q, k, wc, wp);
    return check_launch("image_fusion_pack");
}

extern "C" int jm_image_fusion_gather(int b, int n, int h, int w, int q, int num_levels, const int* channels, const int* strides,
                                      const float* const* maps, const float* const* packed_weights, const float* bias32,
                                      const float* xy, float* out, void* ws, size_t ws_bytes, jm_stream_t stream) {
    JM_REQUIRE(b >= 0 && n >= 0 && h >= 1 && w >= 1 && h < 32768 && w < 65536, "image_fusion_gather: bad sizes");
    if (b == 0 || n == 0) return JM_OK;
    JM_REQUIRE(q >= 1 && q <= 32 && num_levels >= 1 && num_levels <= IF_MAXLV, "image_fusion_gather: q <= 32, <= 4 levels");
    JM_REQUIRE(channels && strides && maps && packed_weights && bias32 && xy && out && ws, "image_fusion_gather: null pointer");
    const long long taps = (long long)b * n * 4;
    JM_REQUIRE(taps < (1LL << 31), "image_fusion_gather: too many taps");
    ImgFusionParams p{};
    p.B = b; p.N = n; p.H = h; p.W = w; p.Q = q; p.nlv = num_levels; p.kmax = 1;
    for (int i = 0; i < num_levels; ++i) {
        const int k = strides[i];
        JM_REQUIRE(k >= 1 && k <= 16 && (k & (k - 1)) == 0 && h % k == 0 && w % k == 0, "image_fusion_gather: stride %d", k);
        JM_REQUIRE(channels[i] >= 16 && channels[i] % 16 == 0, "image_fusion_gather: level channels %d not a multiple of 16", channels[i]);
        JM_REQUIRE(maps[i] && packed_weights[i], "image_fusion_gather: null level %d", i);
        JM_REQUIRE(((reinterpret_cast<uintptr_t>(maps[i]) | reinterpret_cast<uintptr_t>(packed_weights[i])) & 15u) == 0,
                   "image_fusion_gather: 16-byte alignment");
        int sh = 0;
        while ((1 << sh) < k) ++sh;
        p.C[i] = channels[i]; p.shift[i] = sh; p.Hl[i] = h / k; p.Wl[i] = w / k;
        p.map[i] = maps[i]; p.wp[i] = packed_weights[i];
        if (k > p.kmax) p.kmax = k;
    }
    const IfWs wsp = if_carve(ws, taps);
    if (ws_bytes < wsp.total) { set_error("image_fusion_gather: workspace %zu < %zu bytes", ws_bytes, wsp.total); return JM_EWORKSPACE; }
    JM_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 255u) == 0, "image_fusion_gather: workspace must be 256-byte aligned");
    p.key = wsp.key; p.pos = wsp.pos; p.sorted = wsp.sorted; p.tw = wsp.tw; p.tapval = wsp.tapval;
    p.hist = wsp.hist; p.cursor = wsp.cursor; p.bstart = wsp.bstart; p.tstart = wsp.tstart;
    p.bias = bias32; p.xy = xy; p.out = out;
    hipStream_t s = (hipStream_t)stream;
    (void)jm_zero_async(p.hist, 257 * sizeof(int), s);
    const long long pts = (long long)b * n;
    hipLaunchKernelGGL(if_taps_kernel, dim3((unsigned)((pts + 255) / 256)), dim3(256), 0, s, p);
    hipLaunchKernelGGL(if_scan_kernel, dim3(1), dim3(256), 0, s, p);
    hipLaunchKernelGGL(if_scatter_kernel, dim3((unsigned)((taps + 1023) / 1024)), dim3(1024), 0, s, p);
    const long long max_tiles = (taps + 31) / 32 + p.kmax * p.kmax;     // each phase adds at most one partial tile
    hipLaunchKernelGGL(if_gemm_kernel, dim3((unsigned)max_tiles), dim3(64), 0, s, p);
    hipLaunchKernelGGL(if_combine_kernel, dim3((unsigned)((pts + 255) / 256)), dim3(256), 0, s, p);
    return check_launch("image_fusion_gather");
}
