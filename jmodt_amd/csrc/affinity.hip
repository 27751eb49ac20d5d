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
//
// Tried and rejected (MI355X, 128^2 / 256^2 pairs; this kernel: 216 / 759 us per affinity call):
//  * a wave-specialised persistent variant (4 MFMA waves + 4 loader waves forming |p - d| chunks, weights
//    straight from L2 in the packed layout of sa_mlp.hip): 236 / 728 us.  In-kernel cycle counters: the MFMA
//    stream itself runs at 87 % of the pipe, but a wave that shares a SIMD with an MFMA wave gets roughly
//    one VALU issue slot per MFMA — the loader needed 21 k cycles per 128-deep chunk (5.5 k alone) against
//    19 k of MFMA work, so the MFMA waves waited for it;
//  * the same tiling as here with packed weights from L2 and A staged 64 deep (1 barrier per 4 k-tiles):
//    270 / 893 us.
//  * (round 3) the sa_mlp_pm.hip recipe on this tiling — operand tiles ROW-major in LDS ([row][BK + 4]), one ds_write_b128
//    per staged float4 and one ds_read_b128 per four MFMA steps, with 8 waves (32 x 64 each, two per SIMD) or 4 waves
//    (64 x 64), BK 16 or 32: bit-correct, all four slower at 8 x 128^2 / 8 x 256^2 pairs: 1332 / 4926 us (8 waves, BK 16),
//    1530 / 5843 (4 waves), 1423 / 5368 and 1577 / 6016 (BK 32) against 1266 / 4630 us for this kernel.
//  * (round 3) the B operand (weights) NOT staged through LDS: packed once per call in MFMA operand order and streamed from L1 / L2
//    straight into operand registers half a k-tile ahead (the route conv_wino.hip takes): 112 registers, half the LDS traffic and
//    ds_reads, bit-identical — and 2-3 % slower (1270-1284 / 4647-4676 us against 1242-1261 / 4505-4521 us).
//  MfmaUtil of this kernel: 54-56 % (profiles/r01_ops_pmc_MfmaUtil.txt); 72-73 % in the batched form (r02).
#include <stdlib.h>


#include "jm_common.h"

namespace jm {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, LDP = BM + 4;

struct GemmParams {
    int M, N, K;
    // A operand
    const float* A;   // plain rows (M,K)            [AMODE 0]
    const float* pf;  // (P,K) and (D,K), row m -> (m / D, m % D)   [AMODE 1]
    const float* df;
    int D;
    int PD;             // 0: one (P, D) problem.  P*D: a BATCH of problems stacked on the row axis, pf (nb*P, K), df (nb*D, K):
                        // row m -> problem m / PD, pred row m / D (already batch-global), det row (m / PD) * D + m % D
    const float* W;     // (N,K) row-major (Conv1d weight (N,K,1))
    const float* bias;  // (N)
    float* H;           // (M,N) output, ReLU applied            [EMODE 0]
    const float* w3;    // (N)                                    [EMODE 1]
    float* score;       // (slots, M) projection partials, slot = 64-column group (single-wave kernel: 32)   [EMODE 1]
};

// Up to two independent problems per launch (grouped GEMM): the link head's P*D pair rows and
// the start/end head's P+D rows have different weights but the same shapes per layer, and the
// small problem would otherwise be a latency-bound 8-workgroup launch of its own.
struct GemmGroup {
    GemmParams p[2];
    int tiles0;   // workgroups of problem 0 (1-D grid; the rest belong to problem 1)
};

template <int EMODE, int BK, bool PIN>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))   // 144 -> 120 registers, no spills: 4 workgroups
mlp_gemm_kernel(GemmGroup grp) {                                                       // per CU instead of 3 (+2-3 %, measured)
    __shared__ __attribute__((aligned(16))) float As[2][BK][LDP];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][LDP];
    constexpr int NLD = BK / 8;   // float4 loads per operand per thread per k-tile (128 rows x BK / 256 threads / 4)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const bool second = (int)blockIdx.x >= grp.tiles0;
    const GemmParams& p = grp.p[second ? 1 : 0];
    int bid = second ? (int)blockIdx.x - grp.tiles0 : (int)blockIdx.x;
    // workgroups go round-robin to the 8 XCDs (blockIdx & 7), each with its own L2: hand every XCD a CONTIGUOUS range of
    // tiles, so the column tiles that re-read one row block of A (4 of them at N = 512) run on the same L2 one after the
    // other instead of on four different ones (counter traffic of the second layer: 1.1 GB fetched for a 268 MB operand)
    if (!second && (grp.tiles0 & 7) == 0) bid = (bid & 7) * (grp.tiles0 >> 3) + (bid >> 3);
    const int ntn = (p.N + BN - 1) / BN;
    const int m0 = (bid / ntn) * BM, n0 = (bid % ntn) * BN;
    const bool pair_mode = p.pf != nullptr;   // wave-uniform: A rows are |p_i - d_j| formed on the fly

    // staging assignment: 2 float4 of A and 2 of B per thread per k-tile
    int srow[NLD], skq[NLD];
    const float *a_ptr[NLD], *a2_ptr[NLD], *b_ptr[NLD];
    bool a_ok[NLD], b_ok[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int f = tid + 256 * i;
        srow[i] = f / (BK / 4);
        skq[i] = (f % (BK / 4)) * 4;
        const int m = m0 + srow[i], n = n0 + srow[i];
        a_ok[i] = m < p.M;
        b_ok[i] = n < p.N;
        if (!pair_mode) {
            a_ptr[i] = p.A + (size_t)(a_ok[i] ? m : 0) * p.K + skq[i];
            a2_ptr[i] = a_ptr[i];
        } else {
            const int mm = a_ok[i] ? m : 0;
            const int pi = mm / p.D;
            const int di = mm - pi * p.D + (p.PD ? (mm / p.PD) * p.D : 0);
            a_ptr[i] = p.pf + (size_t)pi * p.K + skq[i];
            a2_ptr[i] = p.df + (size_t)di * p.K + skq[i];
        }
        b_ptr[i] = p.W + (size_t)(b_ok[i] ? n : 0) * p.K + skq[i];
    }

    // Every staging load is UNCONDITIONAL on a clamped, always-valid pointer (rows >= M / N read
    // row 0 and are discarded by the epilogue guards): a `cond ? load : 0` makes hipcc branch
    // around each load and wait vmcnt(0) at every join, which serialises the prefetch.
    float4 rv[NLD], ru[NLD], rb[NLD];   // raw staged operands; |v - u| is formed at store time so the
                                  // loads are not waited for until after the MFMAs of this tile
    auto g_load = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            rv[i] = *reinterpret_cast<const float4*>(a_ptr[i] + k0);
            ru[i] = *reinterpret_cast<const float4*>(a2_ptr[i] + k0);
            rb[i] = *reinterpret_cast<const float4*>(b_ptr[i] + k0);
        }
    };
    auto s_store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const float4 v = rv[i], u = ru[i];
            float4 r;
            r.x = pair_mode ? fabsf(v.x - u.x) : v.x;
            r.y = pair_mode ? fabsf(v.y - u.y) : v.y;
            r.z = pair_mode ? fabsf(v.z - u.z) : v.z;
            r.w = pair_mode ? fabsf(v.w - u.w) : v.w;
            As[buf][skq[i] + 0][srow[i]] = r.x; As[buf][skq[i] + 1][srow[i]] = r.y;
            As[buf][skq[i] + 2][srow[i]] = r.z; As[buf][skq[i] + 3][srow[i]] = r.w;
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
        g_load(min(kt + 1, nkt - 1) * BK);   // unconditional (last tile re-read, unused): a branch here
                                             // makes hipcc copy the loaded registers and wait early
        if (PIN) __builtin_amdgcn_sched_barrier(0);   // ... and pin the loads ABOVE the MFMAs (hipcc sinks them otherwise)
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
        // keep the |v-u| arithmetic and the LDS writes of the prefetched tile BELOW the MFMAs:
        // without this fence hipcc hoists them above the loop and waits for the loads first
        if (PIN) __builtin_amdgcn_sched_barrier(0);
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
                const float v = half_sum_f32_dpp(part[r]);   // sum over the 32 columns of this lane half
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (lr == 31 && row < p.M) p.score[(size_t)((n0 >> 6) + wn) * p.M + row] = v;
            }
        }
    }
}

// Small-M variant: ONE wave per 32x32 output tile, operands straight from L2 into MFMA
// registers (no LDS, no barrier), 3-deep register prefetch.  A (P+D)-row start/end problem is
// (M/32)*(N/32) = 128 independent single-wave workgroups marching through K = 512 in ~10 us,
// instead of 8 four-wave workgroups doing 32 barrier-separated k-tiles (~55 us, latency bound).
// Lane (r = lane & 31, h = lane >> 5) loads 4 consecutive k of its row per 8-k step; MFMA step q
// pairs k = 8*kk + q (lanes 0-31) with k = 8*kk + 4 + q (lanes 32-63) on both operands, so every
// k is used exactly once (the summation order differs from the tiled kernel; tolerance 1e-4).
template <int EMODE>
__global__ void __launch_bounds__(64)
mlp_gemm_small_kernel(GemmParams p) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
    const bool pair_mode = p.pf != nullptr;
    const int row = min(m0 + r, p.M - 1), col = min(n0 + r, p.N - 1);
    const float *a_ptr, *a2_ptr;
    if (pair_mode) {
        const int pi = row / p.D;
        const int di = row - pi * p.D + (p.PD ? (row / p.PD) * p.D : 0);
        a_ptr = p.pf + (size_t)pi * p.K + 4 * h;
        a2_ptr = p.df + (size_t)di * p.K + 4 * h;
    } else {
        a_ptr = p.A + (size_t)row * p.K + 4 * h;
        a2_ptr = a_ptr;
    }
    const float* b_ptr = p.W + (size_t)col * p.K + 4 * h;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    constexpr int PD = 3;
    float4 ra[PD], ru[PD], rb[PD];
    const int nk = p.K / 8;
#pragma unroll
    for (int s = 0; s < PD; ++s) {
        const int kk = min(s, nk - 1);
        ra[s] = *reinterpret_cast<const float4*>(a_ptr + kk * 8);
        ru[s] = *reinterpret_cast<const float4*>(a2_ptr + kk * 8);
        rb[s] = *reinterpret_cast<const float4*>(b_ptr + kk * 8);
    }
    for (int kk = 0; kk < nk; kk += PD) {
#pragma unroll
        for (int s = 0; s < PD; ++s) {
            if (kk + s < nk) {   // uniform
                float4 a = ra[s];
                const float4 u = ru[s], b = rb[s];
                if (pair_mode) { a.x = fabsf(a.x - u.x); a.y = fabsf(a.y - u.y); a.z = fabsf(a.z - u.z); a.w = fabsf(a.w - u.w); }
                const int nx = min(kk + s + PD, nk - 1);   // refill this slot (clamped: unconditional load)
                ra[s] = *reinterpret_cast<const float4*>(a_ptr + nx * 8);
                ru[s] = *reinterpret_cast<const float4*>(a2_ptr + nx * 8);
                rb[s] = *reinterpret_cast<const float4*>(b_ptr + nx * 8);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
            }
        }
    }
    // epilogue: col = lane & 31, row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5)
    const int c = n0 + r;
    const bool cok = c < p.N;
    const float bv = cok ? p.bias[c] : 0.f;
    const float wv = (EMODE == 1 && cok) ? p.w3[c] : 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int orow = m0 + (i & 3) + 8 * (i >> 2) + 4 * h;
        const float hval = EMODE == 2 ? acc[i] + bv : fmaxf(acc[i] + bv, 0.f);   // EMODE 2: plain linear layer
        if (EMODE == 0 || EMODE == 2) {
            if (cok && orow < p.M) p.H[(size_t)orow * p.N + c] = hval;
        } else {
            const float v = half_sum_f32_dpp(hval * wv);
            if (r == 31 && orow < p.M) p.score[(size_t)blockIdx.x * p.M + orow] = v;
        }
    }
}

// y[m] = b3 + sum of the projection partials in slot order
__global__ void score_sum_kernel(int n, int slots, const float* __restrict__ part, const float* __restrict__ b3,
                                 float* __restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = b3[0];
    for (int k = 0; k < slots; ++k) v += part[(size_t)k * n + i];
    dst[i] = v;
}

// rows [0,D): mean_i |p_i - d_j| (start features, tracker.py:106 / rcnn.py:254);
// rows [D,D+P): mean_j |p_i - d_j| (end features).
__global__ void __launch_bounds__(256)
se_feature_kernel(int P, int D, int C, const float* __restrict__ pf, const float* __restrict__ df,
                  float* __restrict__ feat) {
    const int row = blockIdx.x;
    pf += (size_t)blockIdx.y * P * C; df += (size_t)blockIdx.y * D * C; feat += (size_t)blockIdx.y * (P + D) * C;   // problem
    for (int k = threadIdx.x; k < C; k += blockDim.x) {
        float acc = 0.f;
        if (row < D) {
            const float dv = df[(size_t)row * C + k];
#pragma unroll 8
            for (int i = 0; i < P; ++i) acc += fabsf(pf[(size_t)i * C + k] - dv);
            acc = acc / (float)P;
        } else {
            const float pv = pf[(size_t)(row - D) * C + k];
#pragma unroll 8
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
    S += (size_t)blockIdx.y * P * D; stats += (size_t)blockIdx.y * 2 * (P + D);   // problem
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
    S += (size_t)blockIdx.y * P * D; A += (size_t)blockIdx.y * P * D; stats += (size_t)blockIdx.y * 2 * (P + D);   // problem
    const int i = e / D, j = e - i * D;
    const float s = S[e];
    const float r = expf(s - stats[2 * i]) / stats[2 * i + 1];
    const float c = expf(s - stats[2 * (P + j)]) / stats[2 * (P + j) + 1];
    A[e] = (r + c) / 2;
}

static int check_mlp(const jm_mlp3_t* m, const char* who) {
    JM_REQUIRE(m && m->w1 && m->b1 && m->w2 && m->b2 && m->w3 && m->b3, "%s: null weights", who);
    JM_REQUIRE(m->c >= 32 && m->c % 32 == 0 && m->h1 >= 32 && m->h1 % 32 == 0 && m->h2 >= 1,
               "%s: channel sizes must be multiples of 32 (c=%d h1=%d h2=%d)", who, m->c, m->h1, m->h2);
    JM_REQUIRE(((reinterpret_cast<uintptr_t>(m->w1) | reinterpret_cast<uintptr_t>(m->w2)) & 15u) == 0,
               "%s: weights must be 16-byte aligned", who);
    return JM_OK;
}

// projection partials of the last layer: one slot per 32 output columns of layer 2 (the single-wave kernel's tile; the
// 128-column tiles use two slots each), summed in slot order by score_sum_kernel — no float atomics, bit-reproducible scores
// (the 128-column tile ALWAYS writes its two slots, also when h2 <= 64 leaves the second empty: for h2 <= 32 that is one slot
// more than divup(h2, 32) — the buffer is sized for whichever kernel takes more)
static int proj_slots(const jm_mlp3_t* mlp) { return imax(divup(mlp->h2, 32), 2 * divup(mlp->h2, BN)); }
static size_t hidden_bytes(size_t m, const jm_mlp3_t* mlp) {
    return align_up(m * mlp->h1 * sizeof(float), 256) + align_up(m * proj_slots(mlp) * sizeof(float), 256);
}
static float* proj_part(float* hidden, size_t m, const jm_mlp3_t* mlp) {
    return reinterpret_cast<float*>(reinterpret_cast<char*>(hidden) + align_up(m * mlp->h1 * sizeof(float), 256));
}

struct MlpJob {
    int M;                 // rows
    const float* x;        // plain rows (M,C), or nullptr for pair rows
    const float *pf, *df;  // pair mode
    int D;
    const jm_mlp3_t* mlp;
    float* hidden;         // hidden_bytes(M, mlp) of scratch: (M,H1) activations, then the projection partials
    float* y;              // (M) output
    int PD = 0;            // batched pair mode: P * D (see GemmParams)
};

static int tiles_of(int M, int N) { return divup(M, BM) * divup(N, BN); }

// run the 3-layer MLP for up to two jobs: 1 launch per layer (grouped) + 1 launch per job summing the projection partials
static int run_mlps(const MlpJob* jobs, int njobs, hipStream_t s) {
    GemmGroup g1{}, g2{};
    int t1 = 0, t2 = 0;
    for (int j = 0; j < njobs; ++j) {
        const MlpJob& jb = jobs[j];
        GemmParams& a = g1.p[j];
        a.M = jb.M; a.N = jb.mlp->h1; a.K = jb.mlp->c;
        a.A = jb.x; a.pf = jb.x ? nullptr : jb.pf; a.df = jb.df; a.D = jb.D; a.PD = jb.PD;
        a.W = jb.mlp->w1; a.bias = jb.mlp->b1; a.H = jb.hidden;
        GemmParams& b = g2.p[j];
        b.M = jb.M; b.N = jb.mlp->h2; b.K = jb.mlp->h1;
        b.A = jb.hidden; b.pf = nullptr; b.W = jb.mlp->w2; b.bias = jb.mlp->b2; b.w3 = jb.mlp->w3; b.score = proj_part(jb.hidden, (size_t)jb.M, jb.mlp);
        if (j == 0) { g1.tiles0 = tiles_of(a.M, a.N); g2.tiles0 = tiles_of(b.M, b.N); }
        t1 += tiles_of(a.M, a.N);
        t2 += tiles_of(b.M, b.N);
    }
    static const int small_m = tune_env("JM_GEMM_SMALL_M", 4096);
    if (njobs == 1 && jobs[0].M <= small_m) {
        const GemmParams& a = g1.p[0];
        const GemmParams& b = g2.p[0];
        hipLaunchKernelGGL((mlp_gemm_small_kernel<0>), dim3(divup(a.N, 32), divup(a.M, 32)), dim3(64), 0, s, a);
        hipLaunchKernelGGL((mlp_gemm_small_kernel<1>), dim3(divup(b.N, 32), divup(b.M, 32)), dim3(64), 0, s, b);
        hipLaunchKernelGGL(score_sum_kernel, dim3(divup(jobs[0].M, 256)), dim3(256), 0, s, jobs[0].M, divup(b.N, 32), b.score,
                           jobs[0].mlp->b3, jobs[0].y);
        return check_launch("affinity mlp (small)");
    }
    static const int bk = tune_env("JM_GEMM_BK", 16);
    static const int pin = tune_env("JM_GEMM_PIN", 1);
#define JM_GEMM_LAUNCH(E, T, G)                                                                          \
    do {                                                                                                 \
        if (bk == 32 && pin) hipLaunchKernelGGL((mlp_gemm_kernel<E, 32, true>), dim3(T), dim3(256), 0, s, G);        \
        else if (bk == 32) hipLaunchKernelGGL((mlp_gemm_kernel<E, 32, false>), dim3(T), dim3(256), 0, s, G);        \
        else if (pin) hipLaunchKernelGGL((mlp_gemm_kernel<E, 16, true>), dim3(T), dim3(256), 0, s, G);              \
        else hipLaunchKernelGGL((mlp_gemm_kernel<E, 16, false>), dim3(T), dim3(256), 0, s, G);                      \
    } while (0)
    JM_GEMM_LAUNCH(0, t1, g1);
    JM_GEMM_LAUNCH(1, t2, g2);
#undef JM_GEMM_LAUNCH
    for (int j = 0; j < njobs; ++j)
        hipLaunchKernelGGL(score_sum_kernel, dim3(divup(jobs[j].M, 256)), dim3(256), 0, s, jobs[j].M, 2 * divup(g2.p[j].N, BN),
                           g2.p[j].score, jobs[j].mlp->b3, jobs[j].y);
    return check_launch("affinity mlp");
}

}  // namespace jm

using namespace jm;

/* y (M, N) = act(x (M, K) W^T (N, K) + b): one dense layer on plain rows with the single-wave 32x32 MFMA tiles of the
 * start/end head (the RCNN's classification / regression heads: a few hundred..thousand rows x 512 channels, where a
 * library GEMM + bias + ReLU is three latency-bound launches) */
extern "C" int jm_linear_rows(int m, int k, int n, const float* x, const float* w, const float* b, float* y, int relu,
                              jm_stream_t stream) {
    JM_REQUIRE(m >= 0 && k >= 8 && k % 8 == 0 && n >= 1, "linear_rows: K must be a positive multiple of 8");
    if (m == 0) return JM_OK;
    JM_REQUIRE(x && w && b && y, "linear_rows: null pointer");
    JM_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w)) & 15u) == 0, "linear_rows: 16-byte alignment");
    JM_REQUIRE(divup(m, 32) <= 65535, "linear_rows: too many rows");
    GemmParams p{};
    p.M = m; p.N = n; p.K = k; p.A = x; p.pf = nullptr; p.df = nullptr; p.D = 1; p.PD = 0; p.W = w; p.bias = b; p.H = y;
    if (relu) hipLaunchKernelGGL((mlp_gemm_small_kernel<0>), dim3(divup(n, 32), divup(m, 32)), dim3(64), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((mlp_gemm_small_kernel<2>), dim3(divup(n, 32), divup(m, 32)), dim3(64), 0, (hipStream_t)stream, p);
    return check_launch("linear_rows");
}

extern "C" size_t jm_mlp3_workspace_bytes(int m, const jm_mlp3_t* mlp) {
    if (m <= 0 || !mlp) return 0;
    return hidden_bytes((size_t)m, mlp);
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
    const MlpJob job{m, x, nullptr, nullptr, 1, mlp, (float*)ws, y};
    return run_mlps(&job, 1, (hipStream_t)stream);
}

// workspace: [hidden link (P*D,H1)] [S raw (P*D)] [se feat (D+P,C)] [se hidden (D+P,H1)] [se logit (D+P)] [stats 2(P+D)]
extern "C" size_t jm_affinity_workspace_bytes(int p, int d, const jm_mlp3_t* link, const jm_mlp3_t* se) {
    if (p <= 0 || d <= 0 || !link) return 0;
    const size_t pd = (size_t)p * d, r = (size_t)p + d;
    size_t b = hidden_bytes(pd, link) + align_up(pd * sizeof(float), 256) +
               align_up(2 * r * sizeof(float), 256);
    if (se) b += align_up(r * se->c * sizeof(float), 256) + hidden_bytes(r, se) +
                 align_up(r * sizeof(float), 256);
    return b;
}

namespace jm {      // affinity_fused.hip
bool fused_link_supported(int c, int h1, int h2);
size_t fused_link_workspace_bytes(int c);
int fused_link_scores(int M, int D, int PD, int c, const float* pf, const float* df, const float* w1, const float* b1, const float* w2,
                      const float* b2, const float* w3, const float* b3, float* score, float* ws, hipStream_t s);
}
constexpr int JM_AFF_FUSED_DEFAULT = 1;      // (tools build: JM_AFF_FUSED=0 -> the two-launch chain, for A/B runs)
// scratch of the batched link head in front of the raw scores: the packed weights of the one-kernel form, or the hidden tensor
static bool link_fused(const jm_mlp3_t* link) {
    static const int fused = tune_env("JM_AFF_FUSED", JM_AFF_FUSED_DEFAULT);
    return fused && fused_link_supported(link->c, link->h1, link->h2);
}
static size_t link_scratch_bytes(size_t pd, const jm_mlp3_t* link) {
    return link_fused(link) ? align_up(fused_link_workspace_bytes(link->c), 256) : hidden_bytes(pd, link);
}

// ---------------------------------------------------------------- batched forms: nb independent (P, D) problems
// (the detector scores every frame of a batch against its predecessor: 8 x 128^2 pair rows become ONE GEMM chain of
// 131072 rows instead of 8 chains of 16384 — one launch per layer, full waves of workgroups, no per-call tails)
extern "C" size_t jm_affinity_batched_workspace_bytes(int nb, int p, int d, const jm_mlp3_t* link) {
    if (nb <= 0 || p <= 0 || d <= 0 || !link) return 0;
    const size_t pd = (size_t)nb * p * d, r = (size_t)nb * ((size_t)p + d);
    return link_scratch_bytes(pd, link) + align_up(pd * sizeof(float), 256) + align_up(2 * r * sizeof(float), 256);
}

extern "C" int jm_affinity_forward_batched(int nb, int p, int d, const float* pred_feat, const float* det_feat,
                                           const jm_mlp3_t* link, float* link_raw, float* link_out, void* ws, size_t ws_bytes,
                                           jm_stream_t stream) {
    JM_REQUIRE(nb >= 0 && p >= 0 && d >= 0, "affinity_batched: bad sizes");
    if (nb == 0 || p == 0 || d == 0) return JM_OK;
    int rc = check_mlp(link, "affinity link_layer");
    if (rc) return rc;
    JM_REQUIRE(pred_feat && det_feat && ws && (link_raw || link_out), "affinity_batched: null pointer");
    JM_REQUIRE(((reinterpret_cast<uintptr_t>(pred_feat) | reinterpret_cast<uintptr_t>(det_feat)) & 15u) == 0,
               "affinity: features must be 16-byte aligned");
    JM_REQUIRE((long long)nb * p * d < (1LL << 31) && nb <= 65535, "affinity_batched: too many pairs");
    if (ws_bytes < jm_affinity_batched_workspace_bytes(nb, p, d, link)) { set_error("affinity_batched: workspace too small"); return JM_EWORKSPACE; }
    hipStream_t s = (hipStream_t)stream;
    const size_t pd = (size_t)nb * p * d, r = (size_t)nb * ((size_t)p + d);
    char* w = (char*)ws;
    float* hidden = (float*)w; w += link_scratch_bytes(pd, link);
    float* sraw = (float*)w;   w += align_up(pd * sizeof(float), 256);
    float* stats = (float*)w;
    float* S = link_raw ? link_raw : sraw;
    if (link_fused(link)) {
        // both hidden layers in one kernel, the hidden activation in LDS (affinity_fused.hip); the scratch holds the packed weights
        rc = fused_link_scores((int)pd, d, p * d, link->c, pred_feat, det_feat, link->w1, link->b1, link->w2, link->b2, link->w3, link->b3,
                               S, hidden, s);
    } else {
        MlpJob lj{(int)pd, nullptr, pred_feat, det_feat, d, link, hidden, S};
        lj.PD = p * d;
        rc = run_mlps(&lj, 1, s);
    }
    if (rc) return rc;
    if (link_out) {
        hipLaunchKernelGGL(softmax_stats_kernel, dim3((unsigned)(p + d), (unsigned)nb), dim3(256), 0, s, p, d, S, stats);
        hipLaunchKernelGGL(dual_softmax_kernel, dim3(divup(p * d, 256), (unsigned)nb), dim3(256), 0, s, p, d, S, stats, link_out);
    }
    (void)r;
    return check_launch("affinity_batched");
}

/* the dual softmax alone: link (nb, P, D) = (softmax(S, dim 2) + softmax(S, dim 1)) / 2 of raw scores S; stats: 2 * nb * (P + D) floats */
extern "C" int jm_affinity_dual_softmax_batched(int nb, int p, int d, const float* link_raw, float* link_out, float* stats,
                                                jm_stream_t stream) {
    JM_REQUIRE(nb >= 0 && p >= 0 && d >= 0 && nb <= 65535, "affinity dual softmax: bad sizes");
    if (nb == 0 || p == 0 || d == 0) return JM_OK;
    JM_REQUIRE(link_raw && link_out && stats, "affinity dual softmax: null pointer");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(softmax_stats_kernel, dim3((unsigned)(p + d), (unsigned)nb), dim3(256), 0, s, p, d, link_raw, stats);
    hipLaunchKernelGGL(dual_softmax_kernel, dim3(divup(p * d, 256), (unsigned)nb), dim3(256), 0, s, p, d, link_raw, stats, link_out);
    return check_launch("affinity dual softmax");
}

extern "C" size_t jm_affinity_start_end_batched_workspace_bytes(int nb, int p, int d, const jm_mlp3_t* se) {
    if (nb <= 0 || p <= 0 || d <= 0 || !se) return 0;
    const size_t r = (size_t)nb * ((size_t)p + d);
    return align_up(r * se->c * sizeof(float), 256) + hidden_bytes(r, se);
}

/* se_out (nb, D + P): per problem [start logits (D) | end logits (P)] */
extern "C" int jm_affinity_start_end_batched(int nb, int p, int d, const float* pred_feat, const float* det_feat,
                                             const jm_mlp3_t* se, float* se_out, void* ws, size_t ws_bytes,
                                             jm_stream_t stream) {
    JM_REQUIRE(nb >= 0 && p >= 0 && d >= 0 && nb <= 65535, "affinity start/end batched: bad sizes");
    if (nb == 0 || p == 0 || d == 0) return JM_OK;
    int rc = check_mlp(se, "affinity se_layer");
    if (rc) return rc;
    JM_REQUIRE(pred_feat && det_feat && se_out && ws, "affinity start/end batched: null pointer");
    if (ws_bytes < jm_affinity_start_end_batched_workspace_bytes(nb, p, d, se)) { set_error("affinity start/end batched: workspace too small"); return JM_EWORKSPACE; }
    hipStream_t s = (hipStream_t)stream;
    const size_t r = (size_t)nb * ((size_t)p + d);
    JM_REQUIRE(r < (1ULL << 31), "affinity start/end batched: too many rows");
    char* w = (char*)ws;
    float* feat = (float*)w;  w += align_up(r * se->c * sizeof(float), 256);
    float* sehid = (float*)w;
    hipLaunchKernelGGL(se_feature_kernel, dim3((unsigned)(p + d), (unsigned)nb), dim3(256), 0, s, p, d, se->c, pred_feat, det_feat, feat);
    const MlpJob sj{(int)r, feat, nullptr, nullptr, 1, se, sehid, se_out};
    rc = run_mlps(&sj, 1, s);
    if (rc) return rc;
    return check_launch("affinity start/end batched");
}

// workspace: [se feat (D+P,C)] [se hidden (D+P,H1)] [se logit (D+P)]
extern "C" size_t jm_affinity_start_end_workspace_bytes(int p, int d, const jm_mlp3_t* se) {
    if (p <= 0 || d <= 0 || !se) return 0;
    const size_t r = (size_t)p + d;
    return align_up(r * se->c * sizeof(float), 256) + hidden_bytes(r, se) + align_up(r * sizeof(float), 256);
}

extern "C" int jm_affinity_start_end(int p, int d, const float* pred_feat, const float* det_feat, const jm_mlp3_t* se,
                                     float* start, float* end, void* ws, size_t ws_bytes, jm_stream_t stream) {
    JM_REQUIRE(p >= 0 && d >= 0, "affinity start/end: bad sizes");
    if (p == 0 || d == 0) return JM_OK;
    int rc = check_mlp(se, "affinity se_layer");
    if (rc) return rc;
    JM_REQUIRE(pred_feat && det_feat && start && end && ws, "affinity start/end: null pointer");
    JM_REQUIRE(((reinterpret_cast<uintptr_t>(pred_feat) | reinterpret_cast<uintptr_t>(det_feat)) & 15u) == 0,
               "affinity: features must be 16-byte aligned");
    if (ws_bytes < jm_affinity_start_end_workspace_bytes(p, d, se)) { set_error("affinity start/end: workspace too small"); return JM_EWORKSPACE; }
    hipStream_t s = (hipStream_t)stream;
    const size_t r = (size_t)p + d;
    char* w = (char*)ws;
    float* feat = (float*)w;    w += align_up(r * se->c * sizeof(float), 256);
    float* sehid = (float*)w;   w += hidden_bytes(r, se);
    float* logit = (float*)w;
    const bool contiguous_out = (end == start + d);   // caller gave one (D+P) buffer: write logits in place
    hipLaunchKernelGGL(se_feature_kernel, dim3((unsigned)r), dim3(256), 0, s, p, d, se->c, pred_feat, det_feat, feat);
    const MlpJob sj{(int)r, feat, nullptr, nullptr, 1, se, sehid, contiguous_out ? start : logit};
    rc = run_mlps(&sj, 1, s);
    if (rc) return rc;
    if (!contiguous_out) {
        (void)hipMemcpyAsync(start, logit, (size_t)d * sizeof(float), hipMemcpyDeviceToDevice, s);
        (void)hipMemcpyAsync(end, logit + d, (size_t)p * sizeof(float), hipMemcpyDeviceToDevice, s);
    }
    return check_launch("affinity start/end");
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
    float* hidden = (float*)w; w += hidden_bytes(pd, link);
    float* sraw = (float*)w;   w += align_up(pd * sizeof(float), 256);
    float* stats = (float*)w;  w += align_up(2 * r * sizeof(float), 256);
    float* S = link_raw ? link_raw : sraw;
    // The start/end head is a tiny, latency-bound problem ((P+D) rows: 8 workgroups marching through 32 k-tiles).
    // Grouping it into the link head's launches makes every launch as long as that slow chain (measured: +35 % per
    // GEMM), so it is its own short chain — here in front of the link head on the caller's stream; a caller that
    // wants the two to overlap runs jm_affinity_start_end on a second stream of its own (the library keeps no
    // streams or events: ops/affinity.py does exactly that).
    if (se) {
        rc = jm_affinity_start_end(p, d, pred_feat, det_feat, se, start, end, w, ws_bytes - (size_t)(w - (char*)ws), stream);
        if (rc) return rc;
    }
    static const int small_m = tune_env("JM_GEMM_SMALL_M", 4096);
    if (link_fused(link) && pd > (size_t)small_m && hidden_bytes(pd, link) >= fused_link_workspace_bytes(link->c)) {
        // above the single-wave kernels' range: the one-kernel form (affinity_fused.hip; 128 x 128 pairs = 256 tiles, one per CU);
        // the packed weights take the place of the hidden tensor
        rc = fused_link_scores((int)pd, d, 0, link->c, pred_feat, det_feat, link->w1, link->b1, link->w2, link->b2, link->w3, link->b3,
                               S, hidden, s);
    } else {
        const MlpJob lj{(int)pd, nullptr, pred_feat, det_feat, d, link, hidden, S};
        rc = run_mlps(&lj, 1, s);
    }
    if (rc) return rc;
    if (link_out) {
        hipLaunchKernelGGL(softmax_stats_kernel, dim3((unsigned)r), dim3(256), 0, s, p, d, S, stats);
        hipLaunchKernelGGL(dual_softmax_kernel, dim3(divup((int)pd, 256)), dim3(256), 0, s, p, d, S, stats, link_out);
    }
    return check_launch("affinity");
}
