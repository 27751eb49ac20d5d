// affinity_fused.hip — the link head's two hidden layers as ONE kernel: the first hidden activation never leaves the CU.
//
// affinity.hip runs  S = w3 . relu(W2 relu(W1 |p_i - d_j| + b1) + b2) + b3  (tracker.py:81-112, rcnn.py:239-258) as two GEMM
// launches with the (P*D, h1) hidden tensor in between: 268 MB written and read again per 8 x 128^2 problem (579 MB of counter
// traffic per call against 4.7 MB of operands).  Here a workgroup owns 64 pair rows from the pair features to the score:
//   * 16 waves; wave w owns output columns 32 w .. 32 w + 31 of BOTH layers (h1 = h2 = 512) for all 64 rows: two 32 x 32
//     accumulator blocks of v_mfma_f32_32x32x2_f32 (32 registers), four waves per SIMD;
//   * the A operand of a layer is a ROW-major LDS tile T[row][516] (129 KB): first the pair features |p_i - d_j| (staged once per
//     tile: 16 lanes copy 256 contiguous bytes of a row), then — in the same region — the first hidden activation, written from
//     the accumulators (a lane half holds 32 consecutive columns of a row: conflict-free ds_write_b32);
//   * a lane's k-steps of a 16-deep k-tile are k = 16 kt + 8 (lane >> 5) + s, s = 0..7 — eight CONSECUTIVE k, i.e. two
//     ds_read_b128 per row block and k-tile (row stride 516 floats: 8 lanes cover the 32 banks), and both operands agree, so every
//     k is used once; the summation order differs from affinity.hip's 2 s + (lane >> 5): tolerance 1e-4, not bit-identical;
//   * the weights never touch LDS: they are packed once per call in MFMA B-operand order ([column block][k-tile][half][lane][4]:
//     one fully coalesced 1 KB wave load per 4 k-steps) and stream from L2 one k-tile ahead of the MFMAs that use them;
//   * no barrier inside a layer (the LDS tile is read-only there): four per tile — staged / layer 1 read / hidden written /
//     projection partials;
//   * layer 2's epilogue = bias + ReLU + the h2 -> 1 projection: 32-column partial per wave (DPP), 16 partials per row summed in
//     wave order by the first 64 threads (+ b3): the raw score, bit-reproducible run to run;
//   * one workgroup per TILE (133 KB of LDS: one resident per CU).  A persistent grid is 1 % slower alone and costs the composed
//     step 2.6 % (757 against 778 frames/s): its workgroups never leave the CUs, so the detections' side stream (RCNN heads, box
//     decode, NMS — latency-bound launches meant to run UNDER this kernel) waits for the whole 1.2 ms.
// Per tile 67 MFLOP and 2 MB of weights from L2 (8 B per clock and CU at the matrix pipe's rate); HBM traffic per call = the
// operands and the scores.
//
// Measured (tools/aff_fused_ab.py, link head alone incl. the dual softmax, 20 calls, ReLU-ed random features; tools/aff_fused_step.sh
// in the step):
//   two launches (affinity.hip)            8 x 128^2 1153 us = 0.759 of the fp32 MFMA peak    8 x 256^2 4491 us = 0.779
//   this kernel                                      1054 us = 0.830                                   4162 us = 0.840
//   ... operands requested per whole k-tile (two register sets, 16 MFMAs per phase) instead of per half k-tile:
//                                                    1122 us = 0.779                                   4405 us = 0.794
//   ... and the LDS tile k-major T[k][68] (eight ds_read_b32 per row block and k-tile, ds_write_b128 of the hidden activation):
//                                                    1160 us = 0.754                                   4603 us = 0.760
//   ... 32-row tiles, 8 waves x 64 columns, two workgroups per CU (phases of one under the MFMA loops of the other, but every
//       weight register feeds ONE MFMA: twice the stream from L2):  1390 us = 0.630                    5454 us = 0.641
//   (timing-only experiments: every weight load an L1 hit -10 %, every LDS read the same k-tile -5 %; unrestricted Gaussian
//    features instead of ReLU-ed ones +4 %: the matrix pipe's clock follows the data)
// In the composed step (HIP events around the entry, which shares the machine with the next batch's image pyramid and the
// detections' stream): 1.19-1.20 ms = 0.73 against 1.25 ms = 0.70, headline +0.9 % (776 against 770 frames/s, tools build);
// 0.57 GB of HBM traffic per step no longer exists, nor 268 MB of workspace.
#include <algorithm>

#include "jm_common.h"

namespace jm {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// (timing experiments of the tools build, wrong results: every weight k-tile = the first one, i.e. an L1 hit instead of the stream
// from L2 / every LDS read = the first k-tile)
#ifdef JM_AFF_EXP_B0
#define AF_EXP_B(kc) ((kc) & 0)
#else
#define AF_EXP_B(kc) (kc)
#endif
#ifdef JM_AFF_EXP_A0
#define AF_EXP_A(kc) ((kc) & 0)
#else
#define AF_EXP_A(kc) (kc)
#endif
constexpr int AF_K = 512;                        // h1 = h2 (and the largest c)
constexpr int AF_ROWS = 64, AF_LD = AF_K + 4, AF_NW = 16, AF_NT = AF_NW * 64;
#ifndef AF_RING
#define AF_RING 4                                 // weight register sets in flight (half k-tiles); 8 measured the same
#endif
constexpr size_t AF_LDS = ((size_t)AF_ROWS * AF_LD + (size_t)AF_NW * AF_ROWS) * sizeof(float);

struct FusedLink {
    int M, D, PD, C;                // pair rows; row m -> pred row m / D, det row (PD ? (m / PD) * D : 0) + m % D
    const float *pf, *df;           // (.., C) features
    const float *w1p, *b1, *w2p, *b2, *w3, *b3;
    float* score;                   // (M)
    int ntiles;
};

// W (n, k) row-major -> B-operand order: [column block = n / 32][k-tile = k / 16][h = (k / 4) % 2][lane = n % 32 + 32 ((k / 8) % 2)][k % 4];
// both weight matrices of the head in one launch
__global__ void af_pack_kernel(int N, int K1, const float* __restrict__ W1, float* __restrict__ dst1, int K2, const float* __restrict__ W2,
                               float* __restrict__ dst2) {
    long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long n1 = (long long)N * K1;
    const bool second = e >= n1;
    if (second) e -= n1;
    const int K = second ? K2 : K1;
    if (e >= (long long)N * K) return;
    const float* W = second ? W2 : W1;
    float* dst = second ? dst2 : dst1;
    const int n = (int)(e / K), k = (int)(e - (long long)n * K);
    const int w = n >> 5, r = n & 31, kt = k >> 4, kk = (k >> 3) & 1, h = (k >> 2) & 1, t = k & 3;
    dst[((((size_t)w * (K >> 4) + kt) * 2 + h) * 64 + (r + 32 * kk)) * 4 + t] = W[e];
}

__global__ void __launch_bounds__(AF_NT) __attribute__((amdgpu_waves_per_eu(4, 4)))
affinity_fused_kernel(FusedLink p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* T = lds;                                           // [row][AF_LD]
    float* part = lds + (size_t)AF_ROWS * AF_LD;              // [wave][row]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lk = lane >> 5;
    const int col = 32 * wave + lr;
    const float bias1 = p.b1[col], bias2 = p.b2[col], w3 = p.w3[col];
    const float b3 = p.b3[0];

    // one layer over the LDS tile, in HALF k-tiles (four k-steps = eight MFMAs): a ring of AF_RING weight register sets (one 1 KB
    // wave load each) runs AF_RING - 1 halves ahead of the MFMAs (with the weights pinned to L1 the kernel is 10 % faster, but a ring of 8
    // instead of 4 changes nothing: it is not the latency of the stream) and the LDS operands (one ds_read_b128 per row block) one half
    // ahead.  Static ring indices under full unrolling + scheduling fences: hipcc otherwise rotates the loop into load ->
    // s_waitcnt vmcnt(0) -> use (the first version: one L2 round trip per eight MFMAs)
    auto layer = [&](const float* __restrict__ wp, int KT, f32x16& acc0, f32x16& acc1) __attribute__((always_inline)) {
        const float4* bp = reinterpret_cast<const float4*>(wp) + (size_t)wave * KT * 128 + lane;
        const float* ap = T + (size_t)lr * AF_LD + 8 * lk;
        const int NH = 2 * KT;                                 // (a multiple of 8: c % 64 == 0)
        float4 b[AF_RING], a[2][2];                            // a[buffer][row block]
#define AF_LOAD_B(H) bp[(size_t)AF_EXP_B(min((H), NH - 1)) * 64]          /* (unconditional: past the end the last half is re-read, unused) */
#define AF_LOAD_A(H, BUF)                                                                      \
        {                                                                                      \
            const int hc = AF_EXP_A(min((H), NH - 1));                                         \
            a[BUF][0] = *reinterpret_cast<const float4*>(ap + 16 * (hc >> 1) + 4 * (hc & 1));  \
            a[BUF][1] = *reinterpret_cast<const float4*>(ap + (size_t)32 * AF_LD + 16 * (hc >> 1) + 4 * (hc & 1)); \
        }
#define AF_STEP(BUF, E, BV)                                                                    \
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[BUF][0].E, BV, acc0, 0, 0, 0);       \
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[BUF][1].E, BV, acc1, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < AF_RING - 1; ++j) b[j] = AF_LOAD_B(j);
        AF_LOAD_A(0, 0)
        // (the LDS reads between the two MFMA quads of a phase, or no fence between loads and MFMAs at all, or a ring of 2 / 8:
        // all within the run-to-run noise of this form)
#define AF_PHASE(H0, J)                                                                         \
            {                                                                                  \
                b[((J) + AF_RING - 1) % AF_RING] = AF_LOAD_B((H0) + (J) + AF_RING - 1);        \
                AF_LOAD_A((H0) + (J) + 1, ((J) + 1) & 1)                                       \
                __builtin_amdgcn_sched_barrier(0);                                             \
                AF_STEP((J) & 1, x, b[(J) % AF_RING].x) AF_STEP((J) & 1, y, b[(J) % AF_RING].y) \
                AF_STEP((J) & 1, z, b[(J) % AF_RING].z) AF_STEP((J) & 1, w, b[(J) % AF_RING].w) \
                __builtin_amdgcn_sched_barrier(0);                                             \
            }
        // (all 64 halves of c = 512 unrolled — no loop header, where hipcc's wait insertion falls back to vmcnt(1) once per trip —
        // spills: 1.4 KB of scratch, 2 x slower)
        for (int h0 = 0; h0 < NH; h0 += AF_RING) {
#pragma unroll
            for (int j = 0; j < AF_RING; ++j) AF_PHASE(h0, j)
        }
#undef AF_PHASE
#undef AF_LOAD_B
#undef AF_LOAD_A
#undef AF_STEP
    };

    for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        const int m0 = tile * AF_ROWS;
        // ---- stage |p_i - d_j| of the tile's 64 pair rows: 16 lanes = 256 contiguous bytes of one row ----
        {
            const int k0 = 4 * (tid & 15);
            for (int rr = tid >> 4; rr < AF_ROWS; rr += AF_NT / 16) {
                const int m = m0 + rr;
                const float ok = m < p.M ? 1.f : 0.f;           // (a multiplier, not a branch around the loads)
                const int mm = m < p.M ? m : 0;
                const int pi = mm / p.D;
                const int di = mm - pi * p.D + (p.PD ? (mm / p.PD) * p.D : 0);
                const float* pp = p.pf + (size_t)pi * p.C;
                const float* dp = p.df + (size_t)di * p.C;
                for (int k = k0; k < p.C; k += 64) {
                    const float4 a = *reinterpret_cast<const float4*>(pp + k);
                    const float4 b = *reinterpret_cast<const float4*>(dp + k);
                    *reinterpret_cast<float4*>(T + (size_t)rr * AF_LD + k) =
                        make_float4(ok * fabsf(a.x - b.x), ok * fabsf(a.y - b.y), ok * fabsf(a.z - b.z), ok * fabsf(a.w - b.w));
                }
            }
        }
        __syncthreads();
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
        layer(p.w1p, p.C >> 4, acc0, acc1);
        __syncthreads();                                  // every wave has read the pair features
        // ---- hidden activation -> the same tile.  Accumulator register r = row (r & 3) + 8 (r >> 2) + 4 lk of the block, column lr ----
        {
            float* t = T + (size_t)(4 * lk) * AF_LD + col;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2);
                t[(size_t)row * AF_LD] = fmaxf(acc0[r] + bias1, 0.f);
                t[(size_t)(32 + row) * AF_LD] = fmaxf(acc1[r] + bias1, 0.f);
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
        layer(p.w2p, AF_K >> 4, acc0, acc1);
        // ---- bias + ReLU + projection: this wave's 32 columns of every row ----
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lk;
            const float s0 = half_sum_f32_dpp(fmaxf(acc0[r] + bias2, 0.f) * w3);
            const float s1 = half_sum_f32_dpp(fmaxf(acc1[r] + bias2, 0.f) * w3);
            if (lr == 31) { part[wave * AF_ROWS + row] = s0; part[wave * AF_ROWS + 32 + row] = s1; }
        }
        __syncthreads();                                  // partials complete; every wave is done with the hidden tile
        if (tid < AF_ROWS && m0 + tid < p.M) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < AF_NW; ++w) s += part[w * AF_ROWS + tid];
            p.score[m0 + tid] = s + b3;
        }
        // (a next tile's staging writes T only; `part` is rewritten three barriers from here)
    }
}

bool fused_link_supported(int c, int h1, int h2) {
    return h1 == AF_K && h2 == AF_K && c >= 64 && c <= AF_K && c % 64 == 0;
}

size_t fused_link_workspace_bytes(int c) { return ((size_t)AF_K * c + (size_t)AF_K * AF_K) * sizeof(float); }

// S (M) = the link head's raw scores of the pair rows.  ws: fused_link_workspace_bytes(c) for the packed weights
int fused_link_scores(int M, int D, int PD, int c, const float* pf, const float* df, const float* w1, const float* b1, const float* w2,
                      const float* b2, const float* w3, const float* b3, float* score, float* ws, hipStream_t s) {
    float* w1p = ws;
    float* w2p = ws + (size_t)AF_K * c;
    hipLaunchKernelGGL(af_pack_kernel, dim3(divup(AF_K * (c + AF_K), 256)), dim3(256), 0, s, AF_K, c, w1, w1p, AF_K, w2, w2p);
    FusedLink p{M, D, PD, c, pf, df, w1p, b1, w2p, b2, w3, b3, score, divup(M, AF_ROWS)};
    static const bool once = [] { (void)hipFuncSetAttribute((const void*)affinity_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)AF_LDS); return true; }();
    (void)once;
    static const int grid_env = tune_env("JM_AFF_GRID", -1);     // (tools build) -1: one workgroup per tile; n > 0: persistent on n workgroups
    const int grid = grid_env > 0 ? std::min(grid_env, p.ntiles) : p.ntiles;
    hipLaunchKernelGGL(affinity_fused_kernel, dim3(grid), dim3(AF_NT), AF_LDS, s, p);
    return check_launch("affinity fused link head");
}

}  // namespace jm
