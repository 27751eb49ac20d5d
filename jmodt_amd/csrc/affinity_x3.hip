// affinity_x3.hip — EXPERIMENTAL, opt-in (never the default path): the batched link head of affinity.hip with every fp32
// product evaluated on the bf16 matrix pipe as a 3-term split
//     a = a1 + a2 + a3,  b = b1 + b2 + b3   (bf16 each: 8 + 8 + 8 significant bits, residuals exact in fp32)
//     a b ~ a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1)        six v_mfma_f32_32x32x16_bf16 per 16 k
// Every kept product is exact in the MFMA's fp32 accumulator; the three dropped terms are <= 2^-24 |a b| each — the size
// of ONE fp32 rounding of the product.  Measured (tools/split_bf16_probe.hip, K = 512): max error vs fp64 3.09e-6 against
// 3.15e-6 for the exact-fp32 MFMA kernel, at 2.5 x the matrix-pipe rate (356 vs 143 TF-equivalent bare).
//
// Structure = mlp_gemm_kernel (128 x 128 x 16 tiles, 4 waves as 2 x 2, double-buffered LDS), with
//   * operands held as three bf16 PLANES: weights are split once per call (split_planes_kernel), the pair rows |p_i - d_j| are
//     split while they are staged (layer 1), and layer 1's epilogue writes the hidden activations directly as planes, so
//     layer 2 stages plain 8-byte copies;
//   * A tiles in LDS, row-major [plane][row][16 k + 8 pad] bf16: a lane's eight k of a 32 x 32 x 16 operand are one
//     ds_read_b128; the WEIGHT planes are packed once per call in MFMA fragment order (pack_planes_frag_kernel) and read
//     straight from L2 into registers one k-tile ahead — no B tile in LDS (37 KB per workgroup instead of 72: four
//     workgroups per CU), no B staging instructions.
// The dual softmax and the start / end head are affinity.hip's.
#include "jm_common.h"

namespace jm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

constexpr int XM = 128, XN = 128, XK = 32, XLD = 40;     // rows, columns, k per tile (two MFMA k-steps per barrier); LDS row stride in bf16 (80 B)

__device__ __forceinline__ u16 bf16_rne(float x) {
    unsigned u = __float_as_uint(x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (u16)(u >> 16);
}
__device__ __forceinline__ float bf16_f(u16 h) { return __uint_as_float((unsigned)h << 16); }

__device__ __forceinline__ void split3(float v, u16& h1, u16& h2, u16& h3) {
    h1 = bf16_rne(v);
    const float r1 = v - bf16_f(h1);
    h2 = bf16_rne(r1);
    const float r2 = r1 - bf16_f(h2);
    h3 = bf16_rne(r2);
}

// fp32 (rows, cols) -> three bf16 planes (3, rows, cols)
__global__ void split_planes_kernel(long long total, const float* __restrict__ src, u16* __restrict__ dst) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    u16 a, b, c;
    split3(src[e], a, b, c);
    dst[e] = a; dst[total + e] = b; dst[2 * total + e] = c;
}

// weights W (N, K) fp32 -> three bf16 planes in fragment order: [plane][k-tile][32-column block][lane][8] with
// lane (lr, lk) holding W[32 cb + lr][16 kt + 8 lk .. + 7]: a wave's B operand of one MFMA is 1 KB contiguous
__global__ void pack_planes_frag_kernel(int N, int K, const float* __restrict__ W, u16* __restrict__ dst) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;       // one (n, k) element
    if (e >= (long long)N * K) return;
    const int n = (int)(e / K), k = (int)(e - (long long)n * K);
    u16 h[3];
    split3(W[e], h[0], h[1], h[2]);
    const int kt = k >> 4, lk = (k >> 3) & 1, t = k & 7, cb = n >> 5, lr = n & 31;
    const size_t plane = (size_t)N * K;
    const size_t at = ((((size_t)kt * (N >> 5) + cb) * 64) + lk * 32 + lr) * 8 + t;
#pragma unroll
    for (int q = 0; q < 3; ++q) dst[q * plane + at] = h[q];
}

struct X3Params {
    int M, N, K;
    const float *pf, *df;      // layer 1: pair rows (see GemmParams of affinity.hip): row m -> (m / D, m % D + (m / PD) * D)
    int D, PD;
    const u16* Ap;             // layer 2: A planes (3, M, K)
    const u16* Bp;             // weight planes in fragment order (pack_planes_frag_kernel)
    const float* bias;         // (N)
    u16* Hp;                   // layer 1 out: relu(acc + bias) as planes (3, M, N)
    const float* w3;           // layer 2: projection
    float* score;              // layer 2 out (M), pre-filled with b3
};

template <int LAYER>
__global__ void __launch_bounds__(256)
mlp_gemm_x3_kernel(X3Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];        // 2 x 3 x 128 x 40 bf16 = 60 KB
    typedef u16 (*Tile)[3][XM][XLD];
    Tile As = reinterpret_cast<Tile>(lds_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int ntn = (p.N + XN - 1) / XN;
    const int m0 = ((int)blockIdx.x / ntn) * XM, n0 = ((int)blockIdx.x % ntn) * XN;
    const size_t planeA = (size_t)p.M * p.K, planeB = (size_t)p.N * p.K;
    // staging: 128 rows x 32 k per k-tile = 1024 groups of 4 k: four per thread
    constexpr int NST = XM * XK / 4 / 256;
    int srow[NST], skq[NST];
    const float *a_ptr[NST], *a2_ptr[NST];
    const u16* ap_ptr[NST];
#pragma unroll
    for (int i = 0; i < NST; ++i) {
        const int f = tid + 256 * i;
        srow[i] = f / (XK / 4); skq[i] = (f % (XK / 4)) * 4;
        const int m = min(m0 + srow[i], p.M - 1);
        if (LAYER == 1) {
            const int pi = m / p.D, di = m - pi * p.D + (m / p.PD) * p.D;
            a_ptr[i] = p.pf + (size_t)pi * p.K + skq[i];
            a2_ptr[i] = p.df + (size_t)di * p.K + skq[i];
        } else {
            ap_ptr[i] = p.Ap + (size_t)m * p.K + skq[i];
        }
    }
    float4 rv[NST], ru[NST];
    uint2 ra[NST][3];
    auto g_load = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            if (LAYER == 1) {
                rv[i] = *reinterpret_cast<const float4*>(a_ptr[i] + k0);
                ru[i] = *reinterpret_cast<const float4*>(a2_ptr[i] + k0);
            } else {
#pragma unroll
                for (int q = 0; q < 3; ++q) ra[i][q] = *reinterpret_cast<const uint2*>(ap_ptr[i] + q * planeA + k0);
            }
        }
    };
    auto s_store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            if (LAYER == 1) {
                const float v[4] = {fabsf(rv[i].x - ru[i].x), fabsf(rv[i].y - ru[i].y), fabsf(rv[i].z - ru[i].z), fabsf(rv[i].w - ru[i].w)};
                u16 h[3][4];
#pragma unroll
                for (int t = 0; t < 4; ++t) split3(v[t], h[0][t], h[1][t], h[2][t]);
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    *reinterpret_cast<uint2*>(&As[buf][q][srow[i]][skq[i]]) =
                        make_uint2((unsigned)h[q][0] | ((unsigned)h[q][1] << 16), (unsigned)h[q][2] | ((unsigned)h[q][3] << 16));
            } else {
#pragma unroll
                for (int q = 0; q < 3; ++q) *reinterpret_cast<uint2*>(&As[buf][q][srow[i]][skq[i]]) = ra[i][q];
            }
        }
    };
    // this wave's B fragments of 16-k step ks: column blocks (n0 + wn * 64) / 32 + j, planes q (clamped block: loads stay unconditional)
    const int ncb = p.N >> 5;
    const int cb0 = min((n0 + wn * 64) >> 5, ncb - 1), cb1 = min(cb0 + 1, ncb - 1);
    auto b_load = [&](int ks, bf16x8 (&b)[2][3]) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const u16* base = p.Bp + q * planeB + ((size_t)ks * ncb * 64 + lane) * 8;
            b[0][q] = *reinterpret_cast<const bf16x8*>(base + (size_t)cb0 * 512);
            b[1][q] = *reinterpret_cast<const bf16x8*>(base + (size_t)cb1 * 512);
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int nkt = p.K / XK, nks = p.K / 16;
    const int lr = lane & 31, lk = lane >> 5;
    // six products per fp32 product, smallest terms first; the four accumulators interleaved (no back-to-back dependent MFMAs)
    auto mm = [&](const bf16x8 (&a)[2][3], const bf16x8 (&b)[2][3]) {
        constexpr int QA[6] = {0, 1, 2, 0, 1, 0}, QB[6] = {2, 1, 0, 1, 0, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][QA[t]], b[j][QB[t]], acc[i][j], 0, 0, 0);
    };
    auto a_load = [&](int buf, int half, bf16x8 (&a)[2][3]) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 3; ++q) a[i][q] = *reinterpret_cast<const bf16x8*>(&As[buf][q][wm * 64 + i * 32 + lr][half * 16 + lk * 8]);
    };
    bf16x8 b0[2][3], b1[2][3];
    g_load(0);
    b_load(0, b0);
    s_store(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        g_load(min(kt + 1, nkt - 1) * XK);
        b_load(min(2 * kt + 1, nks - 1), b1);
        bf16x8 a0[2][3], a1[2][3];
        a_load(buf, 0, a0);
        a_load(buf, 1, a1);
        __builtin_amdgcn_sched_barrier(0);
        mm(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        b_load(min(2 * kt + 2, nks - 1), b0);
        __builtin_amdgcn_sched_barrier(0);
        mm(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < nkt) s_store(buf ^ 1);
        __syncthreads();
    }
    // epilogue.  C layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const size_t planeH = (size_t)p.M * p.N;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float part[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) part[r] = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + lr;
            const bool cok = col < p.N;
            const float bv = cok ? p.bias[col] : 0.f;
            const float wv = (LAYER == 2 && cok) ? p.w3[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                const float h = fmaxf(acc[i][j][r] + bv, 0.f);
                if (LAYER == 1) {
                    if (cok && row < p.M) {
                        u16 h1, h2, h3;
                        split3(h, h1, h2, h3);
                        u16* o = p.Hp + (size_t)row * p.N + col;
                        o[0] = h1; o[planeH] = h2; o[2 * planeH] = h3;
                    }
                } else {
                    part[r] += h * wv;
                }
            }
        }
        if (LAYER == 2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = half_sum_f32_dpp(part[r]);
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (lr == 31 && row < p.M) unsafeAtomicAdd(p.score + row, v);
            }
        }
    }
}

__global__ void x3_fill_kernel(int n, const float* __restrict__ value, float* __restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = value[0];
}

}  // namespace jm

using namespace jm;

extern "C" size_t jm_affinity_x3_workspace_bytes(int nb, int p, int d, const jm_mlp3_t* link) {
    if (nb <= 0 || p <= 0 || d <= 0 || !link) return 0;
    const size_t pd = (size_t)nb * p * d;
    return align_up(3 * pd * link->h1 * sizeof(u16), 256) + align_up(3 * (size_t)link->h1 * link->c * sizeof(u16), 256) +
           align_up(3 * (size_t)link->h2 * link->h1 * sizeof(u16), 256);
}

/* EXPERIMENTAL: the raw link scores S (nb, P, D) of jm_affinity_forward_batched with split-bf16 products (link_raw must be
 * given; run the dual softmax with jm_affinity_forward_batched's machinery on top: ops/affinity.py) */
extern "C" int jm_affinity_link_scores_x3(int nb, int p, int d, const float* pred_feat, const float* det_feat, const jm_mlp3_t* link,
                                          float* link_raw, void* ws, size_t ws_bytes, jm_stream_t stream) {
    JM_REQUIRE(nb >= 0 && p >= 0 && d >= 0, "affinity_x3: bad sizes");
    if (nb == 0 || p == 0 || d == 0) return JM_OK;
    JM_REQUIRE(link && link->w1 && link->b1 && link->w2 && link->b2 && link->w3 && link->b3, "affinity_x3: null weights");
    JM_REQUIRE(link->c % 32 == 0 && link->h1 % 32 == 0 && link->h2 % 32 == 0 && link->c >= 32,
               "affinity_x3: channel sizes must be multiples of 32");
    JM_REQUIRE(pred_feat && det_feat && link_raw && ws, "affinity_x3: null pointer");
    JM_REQUIRE(((reinterpret_cast<uintptr_t>(pred_feat) | reinterpret_cast<uintptr_t>(det_feat) | reinterpret_cast<uintptr_t>(ws)) & 15u) == 0,
               "affinity_x3: 16-byte alignment");
    JM_REQUIRE((long long)nb * p * d < (1LL << 31), "affinity_x3: too many pairs");
    if (ws_bytes < jm_affinity_x3_workspace_bytes(nb, p, d, link)) { set_error("affinity_x3: workspace too small"); return JM_EWORKSPACE; }
    hipStream_t s = (hipStream_t)stream;
    const int M = nb * p * d;
    char* w = (char*)ws;
    u16* Hp = (u16*)w;  w += align_up(3 * (size_t)M * link->h1 * sizeof(u16), 256);
    u16* W1p = (u16*)w; w += align_up(3 * (size_t)link->h1 * link->c * sizeof(u16), 256);
    u16* W2p = (u16*)w;
    const long long t1 = (long long)link->h1 * link->c, t2 = (long long)link->h2 * link->h1;
    hipLaunchKernelGGL(pack_planes_frag_kernel, dim3((unsigned)((t1 + 255) / 256)), dim3(256), 0, s, link->h1, link->c, link->w1, W1p);
    hipLaunchKernelGGL(pack_planes_frag_kernel, dim3((unsigned)((t2 + 255) / 256)), dim3(256), 0, s, link->h2, link->h1, link->w2, W2p);
    X3Params a{};
    a.M = M; a.N = link->h1; a.K = link->c; a.pf = pred_feat; a.df = det_feat; a.D = d; a.PD = p * d;
    a.Bp = W1p; a.bias = link->b1; a.Hp = Hp;
    const size_t lds = sizeof(u16) * 2 * 3 * XM * XLD;
    (void)hipFuncSetAttribute((const void*)mlp_gemm_x3_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)mlp_gemm_x3_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((mlp_gemm_x3_kernel<1>), dim3((unsigned)(divup(M, XM) * divup(a.N, XN))), dim3(256), lds, s, a);
    hipLaunchKernelGGL(x3_fill_kernel, dim3(divup(M, 256)), dim3(256), 0, s, M, link->b3, link_raw);
    X3Params b{};
    b.M = M; b.N = link->h2; b.K = link->h1; b.Ap = Hp; b.Bp = W2p; b.bias = link->b2; b.w3 = link->w3; b.score = link_raw;
    hipLaunchKernelGGL((mlp_gemm_x3_kernel<2>), dim3((unsigned)(divup(M, XM) * divup(b.N, XN))), dim3(256), lds, s, b);
    return check_launch("affinity_x3");
}
