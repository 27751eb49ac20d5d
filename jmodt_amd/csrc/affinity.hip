// affinity.hip — pairwise link / start-end affinity head on the fp32 matrix cores (gfx950).
//
// Replaces the pure-PyTorch path of jmodt/tracking/tracker.py:81-112 (inference) and the
// link/se evaluation of jmodt/detection/modeling/rcnn.py:239-258,272-285:
//     cor = |p_i - d_j| (P,D,C) built with two repeat() copies      -> never materialised here
//     S   = link_layer(cor)  (Conv1d C->H1 +ReLU, H1->H2 +ReLU, H2->1)
//     A   = (softmax(S,1) + softmax(S,0)) / 2
//     start = se_layer(mean_i cor), end = se_layer(mean_j cor)
//
// Design:
//  * The two 512x512 layers are NT GEMMs on v_mfma_f32_32x32x2_f32 (exact f32 products, the only
//    matrix path that holds the 1e-4 score tolerance; no TF32 on gfx950).  128x128x16 tiles,
//    4 waves as 2x2, each wave 2x2 MFMA tiles (64 accumulator VGPRs), LDS tiles stored k-major
//    ([k][row], +4 pad) so an MFMA operand read is one conflict-free ds_read_b32 per lane,
//    register-staged double buffering (global loads for tile t+1 issued before the MFMAs of
//    tile t, written to the other LDS buffer after them; one barrier per k-tile).
//  * Layer 1 generates its A operand on the fly: |p_i - d_j| is formed in registers while the
//    tile is staged, so the (P*D, C) pair tensor (134 MB at 256^2) never exists.
//  * Layer 2's epilogue fuses bias + ReLU + the H2->1 projection: each wave reduces its 64x64
//    tile against w3 and adds one partial per row into the score vector — the second hidden
//    activation is never written either.
//  * Dual softmax and the start/end feature means are small bandwidth-trivial kernels.
#include "jm_common.h"

namespace jm {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 16, LDP = BM + 4;

struct GemmParams {
    int M, N, K;
    // A operand
    const float* A;   // plain rows (M,K)            [AMODE 0]
    const float* pf;  // (P,K) and (D,K), row m -> (m / D, m % D)   [AMODE 1]
    const float* df;
    int D;
    const float* W;     // (N,K) row-major (Conv1d weight (N,K,1))
    const float* bias;  // (N)
    float* H;           // (M,N) output, ReLU applied            [EMODE 0]
    const float* w3;    // (N)                                    [EMODE 1]
    float* score;       // (M) pre-filled with b3, accumulated    [EMODE 1]
};

template <int AMODE, int EMODE>
__global__ void __launch_bounds__(256)
mlp_gemm_kernel(GemmParams p) {
    __shared__ __attribute__((aligned(16))) float As[2][BK][LDP];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][LDP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

    // staging assignment: 2 float4 of A and 2 of B per thread per k-tile
    int srow[2], skq[2];
    const float *a_ptr[2], *a2_ptr[2], *b_ptr[2];
    bool a_ok[2], b_ok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int f = tid + 256 * i;
        srow[i] = f >> 2;
        skq[i] = (f & 3) * 4;
        const int m = m0 + srow[i], n = n0 + srow[i];
        a_ok[i] = m < p.M;
        b_ok[i] = n < p.N;
        if (AMODE == 0) {
            a_ptr[i] = p.A + (size_t)(a_ok[i] ? m : 0) * p.K + skq[i];
            a2_ptr[i] = nullptr;
        } else {
            const int mm = a_ok[i] ? m : 0;
            const int pi = mm / p.D, di = mm - pi * p.D;
            a_ptr[i] = p.pf + (size_t)pi * p.K + skq[i];
            a2_ptr[i] = p.df + (size_t)di * p.K + skq[i];
        }
        b_ptr[i] = p.W + (size_t)(b_ok[i] ? n : 0) * p.K + skq[i];
    }

    float4 ra[2], rb[2];
    auto g_load = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a_ok[i]) {
                v = *reinterpret_cast<const float4*>(a_ptr[i] + k0);
                if (AMODE == 1) {
                    const float4 u = *reinterpret_cast<const float4*>(a2_ptr[i] + k0);
                    v.x = fabsf(v.x - u.x); v.y = fabsf(v.y - u.y); v.z = fabsf(v.z - u.z); v.w = fabsf(v.w - u.w);
                }
            }
            ra[i] = v;
            rb[i] = b_ok[i] ? *reinterpret_cast<const float4*>(b_ptr[i] + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto s_store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            As[buf][skq[i] + 0][srow[i]] = ra[i].x; As[buf][skq[i] + 1][srow[i]] = ra[i].y;
            As[buf][skq[i] + 2][srow[i]] = ra[i].z; As[buf][skq[i] + 3][srow[i]] = ra[i].w;
            Bs[buf][skq[i] + 0][srow[i]] = rb[i].x; Bs[buf][skq[i] + 1][srow[i]] = rb[i].y;
            Bs[buf][skq[i] + 2][srow[i]] = rb[i].z; Bs[buf][skq[i] + 3][srow[i]] = rb[i].w;
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nkt = p.K / BK;
    g_load(0);
    s_store(0);
    __syncthreads();
    const int lr = lane & 31, lk = lane >> 5;
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nkt) g_load((kt + 1) * BK);
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            const int k2 = kk * 2 + lk;
            const float a0 = As[buf][k2][wm * 64 + lr], a1 = As[buf][k2][wm * 64 + 32 + lr];
            const float b0 = Bs[buf][k2][wn * 64 + lr], b1 = Bs[buf][k2][wn * 64 + 32 + lr];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (kt + 1 < nkt) s_store(buf ^ 1);
        __syncthreads();
    }

    // epilogue.  C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
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
            const float wv = (EMODE == 1 && cok) ? p.w3[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                const float h = fmaxf(acc[i][j][r] + bv, 0.f);
                if (EMODE == 0) {
                    if (cok && row < p.M) p.H[(size_t)row * p.N + col] = h;
                } else {
                    part[r] += h * wv;
                }
            }
        }
        if (EMODE == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = part[r];
                v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
                v += __shfl_xor(v, 8); v += __shfl_xor(v, 16);
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (lr == 0 && row < p.M) unsafeAtomicAdd(p.score + row, v);
            }
        }
    }
}

__global__ void fill_kernel(int n, const float* __restrict__ value, float* __restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = value[0];
}

// rows [0,D): mean_i |p_i - d_j| (start features, tracker.py:106 / rcnn.py:254);
// rows [D,D+P): mean_j |p_i - d_j| (end features).
__global__ void __launch_bounds__(256)
se_feature_kernel(int P, int D, int C, const float* __restrict__ pf, const float* __restrict__ df,
                  float* __restrict__ feat) {
    const int row = blockIdx.x;
    for (int k = threadIdx.x; k < C; k += blockDim.x) {
        float acc = 0.f;
        if (row < D) {
            const float dv = df[(size_t)row * C + k];
            for (int i = 0; i < P; ++i) acc += fabsf(pf[(size_t)i * C + k] - dv);
            acc = acc / (float)P;
        } else {
            const float pv = pf[(size_t)(row - D) * C + k];
            for (int j = 0; j < D; ++j) acc += fabsf(pv - df[(size_t)j * C + k]);
            acc = acc / (float)D;
        }
        feat[(size_t)row * C + k] = acc;
    }
}

// softmax statistics: blocks [0,P) rows (max, sum exp over D); blocks [P,P+D) columns
__global__ void __launch_bounds__(256)
softmax_stats_kernel(int P, int D, const float* __restrict__ S, float* __restrict__ stats) {
    __shared__ float red[4];
    const int blk = blockIdx.x;
    const bool is_row = blk < P;
    const int len = is_row ? D : P;
    const size_t base = is_row ? (size_t)blk * D : (size_t)(blk - P);
    const size_t stride = is_row ? 1 : (size_t)D;
    float mx = -INFINITY;
    for (int t = threadIdx.x; t < len; t += blockDim.x) mx = fmaxf(mx, S[base + t * stride]);
    mx = wave_max_f32(mx);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sm = 0.f;
    for (int t = threadIdx.x; t < len; t += blockDim.x) sm += expf(S[base + t * stride] - mx);
    sm = wave_sum_f32(sm);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sm;
    __syncthreads();
    if (threadIdx.x == 0) {
        stats[2 * blk + 0] = mx;
        stats[2 * blk + 1] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}

__global__ void __launch_bounds__(256)
dual_softmax_kernel(int P, int D, const float* __restrict__ S, const float* __restrict__ stats,
                    float* __restrict__ A) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= P * D) return;
    const int i = e / D, j = e - i * D;
    const float s = S[e];
    const float r = expf(s - stats[2 * i]) / stats[2 * i + 1];
    const float c = expf(s - stats[2 * (P + j)]) / stats[2 * (P + j) + 1];
    A[e] = (r + c) / 2;
}

static int check_mlp(const jm_mlp3_t* m, const char* who) {
    JM_REQUIRE(m && m->w1 && m->b1 && m->w2 && m->b2 && m->w3 && m->b3, "%s: null weights", who);
    JM_REQUIRE(m->c >= 16 && m->c % 16 == 0 && m->h1 >= 16 && m->h1 % 16 == 0 && m->h2 >= 1,
               "%s: channel sizes must be multiples of 16 (c=%d h1=%d h2=%d)", who, m->c, m->h1, m->h2);
    JM_REQUIRE(((reinterpret_cast<uintptr_t>(m->w1) | reinterpret_cast<uintptr_t>(m->w2)) & 15u) == 0,
               "%s: weights must be 16-byte aligned", who);
    return JM_OK;
}

// run the 3-layer MLP with either plain rows x (M,C) or the implicit pair rows
static int run_mlp(int M, const float* x, const float* pf, const float* df, int D, const jm_mlp3_t* mlp,
                   float* hidden /* (M,H1) */, float* y /* (M) */, hipStream_t s) {
    GemmParams g1{};
    g1.M = M; g1.N = mlp->h1; g1.K = mlp->c;
    g1.A = x; g1.pf = pf; g1.df = df; g1.D = D;
    g1.W = mlp->w1; g1.bias = mlp->b1; g1.H = hidden;
    dim3 grid1(divup(g1.N, BN), divup(M, BM));
    if (x) hipLaunchKernelGGL((mlp_gemm_kernel<0, 0>), grid1, dim3(256), 0, s, g1);
    else   hipLaunchKernelGGL((mlp_gemm_kernel<1, 0>), grid1, dim3(256), 0, s, g1);
    hipLaunchKernelGGL(fill_kernel, dim3(divup(M, 256)), dim3(256), 0, s, M, mlp->b3, y);
    GemmParams g2{};
    g2.M = M; g2.N = mlp->h2; g2.K = mlp->h1;
    g2.A = hidden; g2.W = mlp->w2; g2.bias = mlp->b2; g2.w3 = mlp->w3; g2.score = y;
    dim3 grid2(divup(g2.N, BN), divup(M, BM));
    hipLaunchKernelGGL((mlp_gemm_kernel<0, 1>), grid2, dim3(256), 0, s, g2);
    return check_launch("affinity mlp");
}

}  // namespace jm

using namespace jm;

extern "C" size_t jm_mlp3_workspace_bytes(int m, const jm_mlp3_t* mlp) {
    if (m <= 0 || !mlp) return 0;
    return align_up((size_t)m * mlp->h1 * sizeof(float), 256);
}

extern "C" int jm_mlp3_forward(int m, const float* x, const jm_mlp3_t* mlp, float* y, void* ws, size_t ws_bytes,
                               jm_stream_t stream) {
    JM_REQUIRE(m >= 0, "mlp3: bad size");
    if (m == 0) return JM_OK;
    int rc = check_mlp(mlp, "mlp3");
    if (rc) return rc;
    JM_REQUIRE(x && y && ws, "mlp3: null pointer");
    JM_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15u) == 0, "mlp3: x must be 16-byte aligned");
    if (ws_bytes < jm_mlp3_workspace_bytes(m, mlp)) { set_error("mlp3: workspace too small"); return JM_EWORKSPACE; }
    return run_mlp(m, x, nullptr, nullptr, 1, mlp, (float*)ws, y, (hipStream_t)stream);
}

// workspace: [hidden link (P*D,H1)] [S raw (P*D)] [se feat (D+P,C)] [se hidden (D+P,H1)] [se logit (D+P)] [stats 2(P+D)]
extern "C" size_t jm_affinity_workspace_bytes(int p, int d, const jm_mlp3_t* link, const jm_mlp3_t* se) {
    if (p <= 0 || d <= 0 || !link) return 0;
    const size_t pd = (size_t)p * d, r = (size_t)p + d;
    size_t b = align_up(pd * link->h1 * sizeof(float), 256) + align_up(pd * sizeof(float), 256) +
               align_up(2 * r * sizeof(float), 256);
    if (se) b += align_up(r * se->c * sizeof(float), 256) + align_up(r * se->h1 * sizeof(float), 256) +
                 align_up(r * sizeof(float), 256);
    return b;
}

extern "C" int jm_affinity_forward(int p, int d, const float* pred_feat, const float* det_feat,
                                   const jm_mlp3_t* link, const jm_mlp3_t* se, float* link_raw, float* link_out,
                                   float* start, float* end, void* ws, size_t ws_bytes, jm_stream_t stream) {
    JM_REQUIRE(p >= 0 && d >= 0, "affinity: bad sizes");
    if (p == 0 || d == 0) return JM_OK;
    int rc = check_mlp(link, "affinity link_layer");
    if (rc) return rc;
    if (se) { rc = check_mlp(se, "affinity se_layer"); if (rc) return rc; }
    JM_REQUIRE(pred_feat && det_feat && ws, "affinity: null pointer");
    JM_REQUIRE(!se || (se->c == link->c), "affinity: link/se input width differ");
    JM_REQUIRE(!se || (start && end), "affinity: se given but start/end null");
    JM_REQUIRE(((reinterpret_cast<uintptr_t>(pred_feat) | reinterpret_cast<uintptr_t>(det_feat)) & 15u) == 0,
               "affinity: features must be 16-byte aligned");
    JM_REQUIRE((long long)p * d < (1LL << 31), "affinity: too many pairs");
    if (ws_bytes < jm_affinity_workspace_bytes(p, d, link, se)) { set_error("affinity: workspace too small"); return JM_EWORKSPACE; }
    hipStream_t s = (hipStream_t)stream;
    const size_t pd = (size_t)p * d, r = (size_t)p + d;
    char* w = (char*)ws;
    float* hidden = (float*)w; w += align_up(pd * link->h1 * sizeof(float), 256);
    float* sraw = (float*)w;   w += align_up(pd * sizeof(float), 256);
    float* stats = (float*)w;  w += align_up(2 * r * sizeof(float), 256);
    float* S = link_raw ? link_raw : sraw;
    rc = run_mlp((int)pd, nullptr, pred_feat, det_feat, d, link, hidden, S, s);
    if (rc) return rc;
    if (link_out) {
        hipLaunchKernelGGL(softmax_stats_kernel, dim3((unsigned)r), dim3(256), 0, s, p, d, S, stats);
        hipLaunchKernelGGL(dual_softmax_kernel, dim3(divup((int)pd, 256)), dim3(256), 0, s, p, d, S, stats, link_out);
    }
    if (se) {
        float* feat = (float*)w;    w += align_up(r * se->c * sizeof(float), 256);
        float* sehid = (float*)w;   w += align_up(r * se->h1 * sizeof(float), 256);
        float* logit = (float*)w;
        hipLaunchKernelGGL(se_feature_kernel, dim3((unsigned)r), dim3(256), 0, s, p, d, se->c, pred_feat, det_feat, feat);
        rc = run_mlp((int)r, feat, nullptr, nullptr, 1, se, sehid, logit, s);
        if (rc) return rc;
        (void)hipMemcpyAsync(start, logit, (size_t)d * sizeof(float), hipMemcpyDeviceToDevice, s);
        (void)hipMemcpyAsync(end, logit + d, (size_t)p * sizeof(float), hipMemcpyDeviceToDevice, s);
    }
    return check_launch("affinity");
}
