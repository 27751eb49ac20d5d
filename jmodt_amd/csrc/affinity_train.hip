// affinity_train.hip — TRAINING-time pairwise affinity on the fp32 matrix cores: forward, the
// re-id losses and the BACKWARD of the link / start-end heads for a batch of (prev, next) frame pairs.
//
// Replaces, for the finetune step of BASELINE configs[3], what the reference does with a Python loop over frame pairs
// and torch autograd (jmodt/detection/modeling/rcnn.py:145-156,204-287; losses jmodt/detection/modeling/
// train_functions.py:282-329 with LOSS_LINK = LOSS_SE = 'L1'):
//     per pair: mean-pool the foreground RoI features per track id, cor = |prev_i - next_j|, link head + dual softmax,
//     start / end features = cor.mean(0) / cor.mean(1) through the se head, ground truth from track-id equality,
//     L = mean |link - gt| + mean |sigmoid(start) - gt_start| + mean |sigmoid(end) - gt_end|
//
// Static-shape form (no data-dependent sizes, no host synchronisation): every RoI slot stays in place and membership is
// carried by masks — `rep` marks the FIRST foreground RoI of each track id (one representative per unique id, exactly
// the rows of the reference's get_unique_tid_feature), the reference's P x D matrices are the (R, R) slot matrices
// restricted to rep_prev x rep_next, and sums over valid entries equal the reference's sums.
//
// Kernels:
//   train_pool_kernel / train_mask_kernel   pooled features, representatives, per-pair counts, gt_start / gt_end
//   train_gemm_kernel<AM, BM, EM>           128x128x16 fp32-MFMA tiles (v_mfma_f32_32x32x2_f32), double-buffered through
//                                           LDS, with the operand forms the backward needs next to the forward's:
//       forward   H1 = relu(|p_i - d_j| W1^T + b1)          A = pair rows formed on the fly, B = W rows
//                 H2 = relu(H1 W2^T + b2), s = H2 w3 + b3   epilogue stores H2 AND projects (the backward needs H2)
//       dH1 = (dH2 W2) .* (H1 > 0)                           B read k-major IN PLACE (W2 is (h2, h1): no transposed copy)
//       dW  = dY^T X, split over the M pair rows             A = dY and B = X both k-major in place (a contraction row m
//                                                            is a contiguous row of dY / of X), X = |p_i - d_j| REGENERATED
//                                                            per tile for the first layer — the (M, C) pair tensor exists
//                                                            neither in the forward nor in the backward
//   train_link_loss_kernel                   masked dual softmax + L1 loss + d(loss)/d(scores) per pair, one workgroup
//   train_se_feat_kernel / train_se_loss_kernel   masked start / end feature means; sigmoid + L1 + d(loss)/d(logits)
//   mlp_bwd_prep_kernel / colsum_kernel / sum_partials_kernel   dH2 in place of H2, bias / w3 gradients, fixed-order
//                                           reduction of the split-M partials (deterministic: no float atomics)
#include "jm_common.h"

namespace jm {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TBM = 128, TBN = 128, TBK = 16, TLDP = TBM + 4;

enum { A_ROWS = 0, A_PAIR = 1, A_KMAJOR = 2 };
enum { B_ROWS = 0, B_KMAJOR = 1, B_KMAJOR_PAIR = 2 };
enum { E_RELU = 0, E_RELU_PROJ = 1, E_MASK = 2, E_PARTIAL = 3 };

struct TGemm {
    int M, N, K;            // output rows, output columns, contraction length
    int kchunk;             // contraction elements per split (multiple of TBK); blockIdx.y = split
    const float* A; int lda;   // A_ROWS: (M, K) rows; A_KMAJOR: (K, M) — element (row r, contraction c) = A[c * lda + r]
    const float* B; int ldb;   // B_ROWS: (N, K) rows (a Conv1d weight); B_KMAJOR: (K, N)
    const float *pf, *df;      // pair rows |pf[pi] - df[di]|: A_PAIR (row m), B_KMAJOR_PAIR (contraction index m); width ldp
    int D, PD, ldp;            // m -> pi = m / D, di = m % D + (m / PD) * D
    const float* bias;         // E_RELU*: (N)
    const float* w3;           // E_RELU_PROJ: (N)
    float* score;              // E_RELU_PROJ: (slots, M) partial projections, slot = the 64- (small kernel: 32-) column group;
                               // train_score_sum_kernel adds them in slot order (no float atomics: bit-reproducible)
    const float* mask;         // E_MASK: (M, N), out = acc where mask > 0 else 0 (may alias out)
    float* out; int ldo;       // (M, N); E_PARTIAL: out + split * M * ldo
};

template <int AM, int BM_, int EM>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))   // <= 128 registers: 4 workgroups per CU
train_gemm_kernel(TGemm p) {
    __shared__ __attribute__((aligned(16))) float As[2][TBK][TLDP];
    __shared__ __attribute__((aligned(16))) float Bs[2][TBK][TLDP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int ntn = (p.N + TBN - 1) / TBN;
    const int m0 = ((int)blockIdx.x / ntn) * TBM, n0 = ((int)blockIdx.x % ntn) * TBN;
    const int k_begin = (int)blockIdx.y * p.kchunk;
    const int k_end = min(p.K, k_begin + p.kchunk);
    const int nkt = (k_end - k_begin + TBK - 1) / TBK;

    // staging: 128 x 16 floats per operand per k-tile = 512 float4 = 2 per thread
    //   transposing forms (ROWS / PAIR): thread reads 4 consecutive k of one row, scatters them to As[k..k+3][row]
    //   k-major forms: thread reads 4 consecutive rows of one contraction index, one ds_write_b128 to As[k][row..row+3]
    int t_row[2], t_kq[2];          // transposing: row within the tile, first k
    int d_k[2], d_r4[2];            // direct: contraction row within the k-tile, first of 4 tile rows
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int f = tid + 256 * i;
        t_row[i] = f >> 2; t_kq[i] = (f & 3) * 4;
        d_k[i] = f >> 5;   d_r4[i] = (f & 31) * 4;
    }
    const float *a_ptr[2], *a2_ptr[2], *b_ptr[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        // every staging load is unconditional on a clamped, always-valid address; out-of-range rows / columns are
        // discarded by the epilogue guards, out-of-range CONTRACTION indices are zeroed at store time
        if (AM == A_ROWS) {
            const int m = min(m0 + t_row[i], p.M - 1);
            a_ptr[i] = p.A + (size_t)m * p.lda + t_kq[i]; a2_ptr[i] = a_ptr[i];
        } else if (AM == A_PAIR) {
            const int m = min(m0 + t_row[i], p.M - 1);
            const int pi = m / p.D, di = m - pi * p.D + (m / p.PD) * p.D;
            a_ptr[i] = p.pf + (size_t)pi * p.ldp + t_kq[i]; a2_ptr[i] = p.df + (size_t)di * p.ldp + t_kq[i];
        } else {
            const int r = min(m0 + d_r4[i], p.M - 4);
            a_ptr[i] = p.A + r; a2_ptr[i] = a_ptr[i];
        }
        if (BM_ == B_ROWS) {
            const int n = min(n0 + t_row[i], p.N - 1);
            b_ptr[i] = p.B + (size_t)n * p.ldb + t_kq[i];
        } else {
            const int c = min(n0 + d_r4[i], p.N - 4);
            b_ptr[i] = (BM_ == B_KMAJOR ? p.B : p.pf) + c;       // B_KMAJOR_PAIR: column offset into the feature rows
        }
    }
    float4 ra[2], ru[2], rb[2], rbu[2];
    bool a_in[2], b_in[2];
    auto g_load = [&](int k0) {      // k0 = absolute contraction index of the k-tile
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (AM == A_KMAJOR) {
                const int kc = k0 + d_k[i];
                a_in[i] = kc < k_end;
                ra[i] = *reinterpret_cast<const float4*>(a_ptr[i] + (size_t)min(kc, p.K - 1) * p.lda);
            } else {
                a_in[i] = true;
                ra[i] = *reinterpret_cast<const float4*>(a_ptr[i] + k0);
                if (AM == A_PAIR) ru[i] = *reinterpret_cast<const float4*>(a2_ptr[i] + k0);
            }
            if (BM_ == B_ROWS) {
                b_in[i] = true;
                rb[i] = *reinterpret_cast<const float4*>(b_ptr[i] + k0);
            } else {
                const int kc = k0 + d_k[i];
                b_in[i] = kc < k_end;
                const int kk = min(kc, p.K - 1);
                if (BM_ == B_KMAJOR) {
                    rb[i] = *reinterpret_cast<const float4*>(b_ptr[i] + (size_t)kk * p.ldb);
                } else {
                    const int pi = kk / p.D, di = kk - pi * p.D + (kk / p.PD) * p.D;
                    const int c = (int)(b_ptr[i] - p.pf);
                    rb[i] = *reinterpret_cast<const float4*>(p.pf + (size_t)pi * p.ldp + c);
                    rbu[i] = *reinterpret_cast<const float4*>(p.df + (size_t)di * p.ldp + c);
                }
            }
        }
    };
    auto s_store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float4 a = ra[i];
            if (AM == A_PAIR) {
                const float4 u = ru[i];
                a.x = fabsf(a.x - u.x); a.y = fabsf(a.y - u.y); a.z = fabsf(a.z - u.z); a.w = fabsf(a.w - u.w);
            }
            if (AM == A_KMAJOR) {
                if (!a_in[i]) a = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(&As[buf][d_k[i]][d_r4[i]]) = a;
            } else {
                As[buf][t_kq[i] + 0][t_row[i]] = a.x; As[buf][t_kq[i] + 1][t_row[i]] = a.y;
                As[buf][t_kq[i] + 2][t_row[i]] = a.z; As[buf][t_kq[i] + 3][t_row[i]] = a.w;
            }
            float4 b = rb[i];
            if (BM_ == B_KMAJOR_PAIR) {
                const float4 u = rbu[i];
                b.x = fabsf(b.x - u.x); b.y = fabsf(b.y - u.y); b.z = fabsf(b.z - u.z); b.w = fabsf(b.w - u.w);
            }
            if (BM_ == B_ROWS) {
                Bs[buf][t_kq[i] + 0][t_row[i]] = b.x; Bs[buf][t_kq[i] + 1][t_row[i]] = b.y;
                Bs[buf][t_kq[i] + 2][t_row[i]] = b.z; Bs[buf][t_kq[i] + 3][t_row[i]] = b.w;
            } else {
                if (!b_in[i]) b = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(&Bs[buf][d_k[i]][d_r4[i]]) = b;
            }
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int lr = lane & 31, lk = lane >> 5;
    if (nkt > 0) {
        g_load(k_begin);
        s_store(0);
        __syncthreads();
        for (int kt = 0; kt < nkt; ++kt) {
            const int buf = kt & 1;
            g_load(k_begin + min(kt + 1, nkt - 1) * TBK);       // unconditional (last tile re-read, unused)
            __builtin_amdgcn_sched_barrier(0);                  // loads stay above the MFMAs
#pragma unroll
            for (int kk = 0; kk < TBK / 2; ++kk) {
                const int k2 = kk * 2 + lk;
                const float a0 = As[buf][k2][wm * 64 + lr], a1 = As[buf][k2][wm * 64 + 32 + lr];
                const float b0 = Bs[buf][k2][wn * 64 + lr], b1 = Bs[buf][k2][wn * 64 + 32 + lr];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);                  // |v - u| and the LDS writes of the next tile below them
            if (kt + 1 < nkt) s_store(buf ^ 1);
            __syncthreads();
        }
    }

    // epilogue.  C layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    float* out = p.out + (EM == E_PARTIAL ? (size_t)blockIdx.y * p.M * p.ldo : 0);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float part[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) part[r] = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + lr;
            const bool cok = col < p.N;
            const float bv = ((EM == E_RELU || EM == E_RELU_PROJ) && cok) ? p.bias[col] : 0.f;
            const float wv = (EM == E_RELU_PROJ && cok) ? p.w3[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                const bool ok = cok && row < p.M;
                float v = acc[i][j][r];
                if (EM == E_RELU || EM == E_RELU_PROJ) v = fmaxf(v + bv, 0.f);
                if (EM == E_MASK) v = (ok && p.mask[(size_t)row * p.ldo + col] > 0.f) ? v : 0.f;
                if (ok) out[(size_t)row * p.ldo + col] = v;
                if (EM == E_RELU_PROJ) part[r] += v * wv;
            }
        }
        if (EM == E_RELU_PROJ) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = half_sum_f32_dpp(part[r]);
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (lr == 31 && row < p.M) p.score[(size_t)((n0 >> 6) + wn) * p.M + row] = v;
            }
        }
    }
}

template <int AM, int BM_, int EM>
static void launch_tgemm(const TGemm& p, hipStream_t s) {
    const int splits = divup(p.K, p.kchunk);
    hipLaunchKernelGGL((train_gemm_kernel<AM, BM_, EM>), dim3((unsigned)(divup(p.M, TBM) * divup(p.N, TBN)), (unsigned)splits),
                       dim3(256), 0, s, p);
}

// Small-M variant (the start / end head: a few hundred rows): ONE wave per 32x32 output tile, operands straight from L2
// into the MFMA registers (no LDS, no barrier), 3-deep register prefetch — (M/32) * (N/32) independent single-wave
// workgroups instead of (M/128) * (N/128) four-wave workgroups marching through K behind a barrier per k-tile (8 of them
// at M = 256: 66-92 us per GEMM, measured).  Lane (r = lane & 31, h = lane >> 5) holds 4 consecutive k of its row per 8-k
// step; MFMA step q pairs k = 8 kk + q (lanes 0-31) with k = 8 kk + 4 + q (lanes 32-63) on both operands.
template <int BM_, int EM>
__global__ void __launch_bounds__(64)
train_gemm_small_kernel(TGemm p) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
    const int row = min(m0 + r, p.M - 1), col = min(n0 + r, p.N - 1);
    const float* a_ptr = p.A + (size_t)row * p.lda + 4 * h;
    const float* b_ptr = BM_ == B_ROWS ? p.B + (size_t)col * p.ldb + 4 * h : p.B + (size_t)(4 * h) * p.ldb + col;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    constexpr int PF = 3;
    float4 ra[PF], rb[PF];
    const int nk = p.K / 8;
    auto loadB = [&](int kk) {
        if (BM_ == B_ROWS) return *reinterpret_cast<const float4*>(b_ptr + kk * 8);
        const float* q = b_ptr + (size_t)(kk * 8) * p.ldb;       // 4 contraction rows, this lane's column
        return make_float4(q[0], q[p.ldb], q[2 * (size_t)p.ldb], q[3 * (size_t)p.ldb]);
    };
#pragma unroll
    for (int s = 0; s < PF; ++s) {
        const int kk = min(s, nk - 1);
        ra[s] = *reinterpret_cast<const float4*>(a_ptr + kk * 8);
        rb[s] = loadB(kk);
    }
    for (int kk = 0; kk < nk; kk += PF) {
#pragma unroll
        for (int s = 0; s < PF; ++s) {
            if (kk + s < nk) {   // uniform
                const float4 a = ra[s], b = rb[s];
                const int nx = min(kk + s + PF, nk - 1);   // refill this slot (clamped: unconditional load)
                ra[s] = *reinterpret_cast<const float4*>(a_ptr + nx * 8);
                rb[s] = loadB(nx);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
            }
        }
    }
    const int c = n0 + r;
    const bool cok = c < p.N;
    const float bv = (EM != E_MASK && cok) ? p.bias[c] : 0.f;
    const float wv = (EM == E_RELU_PROJ && cok) ? p.w3[c] : 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int orow = m0 + (i & 3) + 8 * (i >> 2) + 4 * h;
        const bool ok = cok && orow < p.M;
        float v = acc[i];
        if (EM == E_MASK) v = (ok && p.mask[(size_t)orow * p.ldo + c] > 0.f) ? v : 0.f;
        else v = fmaxf(v + bv, 0.f);
        if (ok) p.out[(size_t)orow * p.ldo + c] = v;
        if (EM == E_RELU_PROJ) {
            const float t = half_sum_f32_dpp(v * wv);
            if (r == 31 && orow < p.M) p.score[(size_t)blockIdx.x * p.M + orow] = t;
        }
    }
}

constexpr int SMALL_M = 2048;      // rows up to which the single-wave tiles win (128-row tiles: < 64 workgroups per 512 columns)

template <int BM_, int EM>
static void launch_tgemm_small(const TGemm& p, hipStream_t s) {
    hipLaunchKernelGGL((train_gemm_small_kernel<BM_, EM>), dim3((unsigned)divup(p.N, 32), (unsigned)divup(p.M, 32)), dim3(64), 0, s, p);
}

// y[m] = b3 + sum over the column-group slots of the projection partials, in slot order
__global__ void train_score_sum_kernel(int n, int slots, const float* __restrict__ part, const float* __restrict__ b3,
                                       float* __restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = b3[0];
    for (int k = 0; k < slots; ++k) v += part[(size_t)k * n + i];
    dst[i] = v;
}

// ---------------------------------------------------------------------------------------------------------------------
// pooled features + representatives.  One workgroup per RoI slot (frame f, slot i): get_unique_tid_feature (rcnn.py:145-156)
// without torch.unique — slot i carries the mean feature of the foreground RoIs sharing its track id, and is a
// representative iff it is the first such slot.  Frames are interleaved (prev, next, prev, next, ...) as rcnn.py:212-217
// de-interleaves them: even frames go to pooled_prev (F*R, C), odd frames to pooled_next.
__global__ void __launch_bounds__(128)
train_pool_kernel(int R, int C, const float* __restrict__ feats, const float* __restrict__ tids, float* __restrict__ pooled_prev,
                  float* __restrict__ pooled_next, int* __restrict__ rep) {
    __shared__ unsigned char same[256];
    __shared__ int cnt_s, earlier_s;
    const int f = blockIdx.x / R, i = blockIdx.x % R;
    const float* t = tids + (size_t)f * R;
    const float ti = t[i];
    const bool fg = ti > 0.f;
    if (threadIdx.x == 0) { cnt_s = 0; earlier_s = 0; }
    __syncthreads();
    for (int k = threadIdx.x; k < R; k += blockDim.x) {
        const bool sm = fg && t[k] > 0.f && t[k] == ti;
        same[k] = sm ? 1 : 0;
        if (sm) { atomicAdd(&cnt_s, 1); if (k < i) atomicOr(&earlier_s, 1); }
    }
    __syncthreads();
    const int cnt = cnt_s;
    float* dst = ((f & 1) ? pooled_next : pooled_prev) + ((size_t)(f >> 1) * R + i) * C;
    const float* src = feats + (size_t)f * R * C;
    const float inv = cnt > 0 ? 1.f / (float)cnt : 0.f;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float acc = 0.f;
        for (int k = 0; k < R; ++k)
            if (same[k]) acc += src[(size_t)k * C + c];
        dst[c] = acc * inv;
    }
    if (threadIdx.x == 0) rep[(size_t)f * R + i] = (fg && !earlier_s) ? 1 : 0;
}

// per pair: drop pairs without foreground on a side (rcnn.py:230), representatives' counts, the ground-truth start /
// end targets (1 - column / row sums of the tid-equality matrix over representatives, rcnn.py:247-252) and the three
// LOCAL element counts of the loss means (links, starts, ends) accumulated into counts[3] (exact: integers in float)
__global__ void __launch_bounds__(256)
train_mask_kernel(int R, const float* __restrict__ tids, const int* __restrict__ rep, int* __restrict__ rep_prev,
                  int* __restrict__ rep_next, int* __restrict__ n_pair, float* __restrict__ gt_starts,
                  float* __restrict__ gt_ends, float* __restrict__ counts) {
    __shared__ int np_s, nn_s;
    const int f = blockIdx.x;
    const int* rp = rep + (size_t)(2 * f) * R;
    const int* rn = rep + (size_t)(2 * f + 1) * R;
    const float* tp = tids + (size_t)(2 * f) * R;
    const float* tn = tids + (size_t)(2 * f + 1) * R;
    if (threadIdx.x == 0) { np_s = 0; nn_s = 0; }
    __syncthreads();
    for (int k = threadIdx.x; k < R; k += blockDim.x) {
        if (rp[k]) atomicAdd(&np_s, 1);
        if (rn[k]) atomicAdd(&nn_s, 1);
    }
    __syncthreads();
    const bool both = np_s > 0 && nn_s > 0;
    const int np = both ? np_s : 0, nn = both ? nn_s : 0;
    for (int k = threadIdx.x; k < R; k += blockDim.x) {
        const int a = both && rp[k], b = both && rn[k];
        rep_prev[(size_t)f * R + k] = a;
        rep_next[(size_t)f * R + k] = b;
        int m_end = 0, m_start = 0;
        for (int q = 0; q < R; ++q) {
            if (a && both && rn[q] && tn[q] == tp[k]) ++m_end;       // next representatives with prev slot k's id
            if (b && both && rp[q] && tp[q] == tn[k]) ++m_start;     // prev representatives with next slot k's id
        }
        gt_ends[(size_t)f * R + k] = a ? 1.f - (float)m_end : 0.f;
        gt_starts[(size_t)f * R + k] = b ? 1.f - (float)m_start : 0.f;
    }
    if (threadIdx.x == 0) {
        n_pair[2 * f] = np; n_pair[2 * f + 1] = nn;
        if (both) {
            atomicAdd(counts + 0, (float)(np * nn));
            atomicAdd(counts + 1, (float)nn);
            atomicAdd(counts + 2, (float)np);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// masked dual softmax + L1 link loss + its gradient w.r.t. the raw scores, one workgroup per pair.
//   link = (softmax over next representatives + softmax over prev representatives) / 2   (rcnn.py:242-244)
//   loss_part[f] = sum_valid |link - gt|;  dS = d(sum_valid |link - gt| / den) / dS
__global__ void __launch_bounds__(1024)
train_link_loss_kernel(int R, const float* __restrict__ S, const int* __restrict__ rep_prev, const int* __restrict__ rep_next,
                       const float* __restrict__ tids, const float* __restrict__ counts, float weight,
                       float* __restrict__ link_out, float* __restrict__ gt_out, float* __restrict__ dS,
                       float* __restrict__ loss_part) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Ss = lds;                 // R*R   raw scores
    float* Gs = Ss + R * R;          // R*R   g = d(loss)/d(link)
    float* rmax = Gs + R * R;        // R each below
    float* rsum = rmax + R;
    float* cmax = rsum + R;
    float* csum = cmax + R;
    float* gp = csum + R;
    float* gq = gp + R;
    int* rp = reinterpret_cast<int*>(gq + R);
    int* rn = rp + R;
    float* tp = reinterpret_cast<float*>(rn + R);
    float* tn = tp + R;
    __shared__ float red[16];
    const int f = blockIdx.x, tid = threadIdx.x;
    const float* Sf = S + (size_t)f * R * R;
    for (int e = tid; e < R * R; e += blockDim.x) Ss[e] = Sf[e];
    for (int k = tid; k < R; k += blockDim.x) {
        rp[k] = rep_prev[(size_t)f * R + k]; rn[k] = rep_next[(size_t)f * R + k];
        tp[k] = tids[(size_t)(2 * f) * R + k]; tn[k] = tids[(size_t)(2 * f + 1) * R + k];
    }
    __syncthreads();
    // every row / column ("line") is handled by 8 consecutive lanes: 2R lines x 8 lanes = the 1024 threads at R = 64
    const int sub = tid & 7;
    for (int k = tid >> 3; k < 2 * R; k += blockDim.x >> 3) {      // softmax statistics: rows over rep_next, columns over rep_prev
        const bool row = k < R;
        const int i = row ? k : k - R;
        float mx = -INFINITY;
        for (int q = sub; q < R; q += 8) {
            const bool in = row ? rn[q] : rp[q];
            const float v = row ? Ss[i * R + q] : Ss[q * R + i];
            if (in) mx = fmaxf(mx, v);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 1)); mx = fmaxf(mx, __shfl_xor(mx, 2)); mx = fmaxf(mx, __shfl_xor(mx, 4));
        float sm = 0.f;
        for (int q = sub; q < R; q += 8) {
            const bool in = row ? rn[q] : rp[q];
            const float v = row ? Ss[i * R + q] : Ss[q * R + i];
            if (in) sm += expf(v - mx);
        }
        sm += __shfl_xor(sm, 1); sm += __shfl_xor(sm, 2); sm += __shfl_xor(sm, 4);
        if (sub == 0) { if (row) { rmax[i] = mx; rsum[i] = sm; } else { cmax[i] = mx; csum[i] = sm; } }
    }
    __syncthreads();
    // P = row softmax, Q = column softmax of a valid entry (recomputed where needed: two R*R arrays fit the LDS at R = 128)
    auto softmaxes = [&](int i, int j, float& P, float& Q) {
        const float s = Ss[i * R + j];
        P = expf(s - rmax[i]) / rsum[i];
        Q = expf(s - cmax[j]) / csum[j];
    };
    const float den = fmaxf(counts[0], 1.f);
    float lsum = 0.f;
    for (int e = tid; e < R * R; e += blockDim.x) {
        const int i = e / R, j = e - i * R;
        float g = 0.f, link = 0.f, gt = 0.f;
        if (rp[i] && rn[j]) {
            float P, Q;
            softmaxes(i, j, P, Q);
            link = (P + Q) / 2;
            gt = tp[i] == tn[j] ? 1.f : 0.f;
            const float d = link - gt;
            lsum += fabsf(d);
            g = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * weight / den;
        }
        Gs[e] = g;
        if (link_out) link_out[(size_t)f * R * R + e] = link;
        if (gt_out) gt_out[(size_t)f * R * R + e] = gt;
    }
    __syncthreads();
    for (int k = tid >> 3; k < 2 * R; k += blockDim.x >> 3) {      // gp_i = sum_j g_ij P_ij, gq_j = sum_i g_ij Q_ij
        const bool row = k < R;
        const int i = row ? k : k - R;
        float a = 0.f;
        for (int q = sub; q < R; q += 8) {
            const int ii = row ? i : q, jj = row ? q : i;
            if (rp[ii] && rn[jj]) {
                float P, Q;
                softmaxes(ii, jj, P, Q);
                a += Gs[ii * R + jj] * (row ? P : Q);
            }
        }
        a += __shfl_xor(a, 1); a += __shfl_xor(a, 2); a += __shfl_xor(a, 4);
        if (sub == 0) { if (row) gp[i] = a; else gq[i] = a; }
    }
    __syncthreads();
    for (int e = tid; e < R * R; e += blockDim.x) {
        const int i = e / R, j = e - i * R;
        float d = 0.f;
        if (rp[i] && rn[j]) {
            float P, Q;
            softmaxes(i, j, P, Q);
            const float g = Gs[e];
            d = 0.5f * P * (g - gp[i]) + 0.5f * Q * (g - gq[j]);
        }
        dS[(size_t)f * R * R + e] = d;
    }
    lsum = wave_sum_f32(lsum);
    if ((tid & 63) == 0) red[tid >> 6] = lsum;
    __syncthreads();
    if (tid == 0) {
        float a = 0.f;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) a += red[w];
        loss_part[f] = a;
    }
}

// start / end features over the representatives (rcnn.py:254-255: cor.mean(dim=0) / cor.mean(dim=1) of the P x D tensor):
// rows [f][0, R): start feature of next slot j = mean over prev representatives i of |p_i - d_j|;
// rows [f][R, 2R): end feature of prev slot i = mean over next representatives j
__global__ void __launch_bounds__(256)
train_se_feat_kernel(int R, int C, const float* __restrict__ pp, const float* __restrict__ pn, const int* __restrict__ rep_prev,
                     const int* __restrict__ rep_next, const int* __restrict__ n_pair, float* __restrict__ feat) {
    __shared__ unsigned char in[256];
    const int f = blockIdx.x / (2 * R), q = blockIdx.x % (2 * R);
    const bool start = q < R;
    const int self = start ? q : q - R;
    const int* other_rep = (start ? rep_prev : rep_next) + (size_t)f * R;
    for (int k = threadIdx.x; k < R; k += blockDim.x) in[k] = other_rep[k] ? 1 : 0;
    __syncthreads();
    const int n_other = start ? n_pair[2 * f] : n_pair[2 * f + 1];
    const float inv = 1.f / (float)max(n_other, 1);
    const float* mine = (start ? pn : pp) + ((size_t)f * R + self) * C;
    const float* others = (start ? pp : pn) + (size_t)f * R * C;
    float* dst = feat + (size_t)blockIdx.x * C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float v = mine[c];
        float acc = 0.f;
        for (int k = 0; k < R; ++k)
            if (in[k]) acc += fabsf(others[(size_t)k * C + c] - v);
        dst[c] = acc * inv;
    }
}

// sigmoid + L1 start / end losses and d(loss)/d(logit) (train_functions.py:313-318), one workgroup per pair
__global__ void __launch_bounds__(256)
train_se_loss_kernel(int R, const float* __restrict__ z, const int* __restrict__ rep_prev, const int* __restrict__ rep_next,
                     const float* __restrict__ gt_starts, const float* __restrict__ gt_ends, const float* __restrict__ counts,
                     float weight, float* __restrict__ dz, float* __restrict__ loss_part) {
    __shared__ float red[2][4];
    const int f = blockIdx.x, tid = threadIdx.x;
    const float den_s = fmaxf(counts[1], 1.f), den_e = fmaxf(counts[2], 1.f);
    float ls = 0.f, le = 0.f;
    for (int q = tid; q < 2 * R; q += blockDim.x) {
        const bool start = q < R;
        const int k = start ? q : q - R;
        const bool valid = start ? rep_next[(size_t)f * R + k] : rep_prev[(size_t)f * R + k];
        const float gt = start ? gt_starts[(size_t)f * R + k] : gt_ends[(size_t)f * R + k];
        float d_out = 0.f;
        if (valid) {
            const float s = 1.f / (1.f + expf(-z[(size_t)f * 2 * R + q]));
            const float d = s - gt;
            if (start) ls += fabsf(d); else le += fabsf(d);
            d_out = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * s * (1.f - s) * weight / (start ? den_s : den_e);
        }
        dz[(size_t)f * 2 * R + q] = d_out;
    }
    ls = wave_sum_f32(ls); le = wave_sum_f32(le);
    if ((tid & 63) == 0) { red[0][tid >> 6] = ls; red[1][tid >> 6] = le; }
    __syncthreads();
    if (tid == 0) {
        loss_part[2 * f] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        loss_part[2 * f + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward helpers.  Row chunks of CH rows; partial column sums per chunk, reduced in fixed order afterwards.
constexpr int CH = 64;       // rows per partial
constexpr int CW = 128;      // columns per workgroup (one per thread): grid (row chunks, column groups)

// H2 (M, N) -> dH2 in place: dH2[m][n] = dy[m] * w3[n] * (H2[m][n] > 0); partials of dw3 = H2^T dy, db2 = colsum(dH2), db3 = sum dy.
// Rows are taken 8 at a time: the 8 loads are issued before the first dependent store (H2 is read and written in place, so
// the compiler cannot hoist a later row's load above an earlier row's store by itself).
__global__ void __launch_bounds__(CW)
mlp_bwd_prep_kernel(int M, int N, float* __restrict__ H2, const float* __restrict__ dy, const float* __restrict__ w3,
                    float* __restrict__ p_dw3, float* __restrict__ p_db2, float* __restrict__ p_db3) {
    const int chunk = blockIdx.x, r0 = chunk * CH, r1 = min(M, r0 + CH);
    const int n = blockIdx.y * CW + threadIdx.x;
    if (n < N) {
        const float w = w3[n];
        float a3 = 0.f, a2 = 0.f;
        for (int m = r0; m < r1; m += 8) {
            float h[8], d[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int mm = min(m + q, r1 - 1);
                h[q] = H2[(size_t)mm * N + n];
                d[q] = dy[mm];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (m + q < r1) {
                    a3 += d[q] * h[q];
                    const float g = h[q] > 0.f ? d[q] * w : 0.f;
                    H2[(size_t)(m + q) * N + n] = g;
                    a2 += g;
                }
            }
        }
        p_dw3[(size_t)chunk * N + n] = a3;
        p_db2[(size_t)chunk * N + n] = a2;
    }
    if (blockIdx.y == 0 && threadIdx.x == 0) {
        float a = 0.f;
        for (int m = r0; m < r1; ++m) a += dy[m];
        p_db3[chunk] = a;
    }
}

__global__ void __launch_bounds__(CW)
colsum_kernel(int M, int N, const float* __restrict__ A, float* __restrict__ part) {
    const int chunk = blockIdx.x, r0 = chunk * CH, r1 = min(M, r0 + CH);
    const int n = blockIdx.y * CW + threadIdx.x;
    if (n >= N) return;
    float a = 0.f;
    for (int m = r0; m < r1; m += 8) {
        float h[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) h[q] = A[(size_t)min(m + q, r1 - 1) * N + n];
#pragma unroll
        for (int q = 0; q < 8; ++q) a += (m + q < r1) ? h[q] : 0.f;
    }
    part[(size_t)chunk * N + n] = a;
}

// out[e] = sum_s in[s * len + e], s ascending: up to 6 jobs per launch
struct SumJobs {
    int njobs;
    int nparts[6], len[6], first_block[7];
    const float* in[6];
    float* out[6];
};

__global__ void __launch_bounds__(256)
sum_partials_kernel(SumJobs j) {
    int q = 0;
    while (q + 1 < j.njobs && (int)blockIdx.x >= j.first_block[q + 1]) ++q;
    const int e = ((int)blockIdx.x - j.first_block[q]) * 256 + threadIdx.x;
    if (e >= j.len[q]) return;
    const float* src = j.in[q] + e;
    float a = 0.f;
    for (int s = 0; s < j.nparts[q]; ++s) a += src[(size_t)s * j.len[q]];
    j.out[q][e] = a;
}

// loss = wl * sum(lp) / max(c0, 1) + wse * (sum(sp[:, 0]) / max(c1, 1) + sum(sp[:, 1]) / max(c2, 1)): one wave
__global__ void __launch_bounds__(64)
train_loss_value_kernel(int npairs, const float* __restrict__ lp, const float* __restrict__ sp, const float* __restrict__ counts,
                        float wl, float wse, float* __restrict__ out) {
    float a = 0.f, b = 0.f, c = 0.f;
    for (int f = threadIdx.x; f < npairs; f += 64) { a += lp[f]; b += sp[2 * f]; c += sp[2 * f + 1]; }
    a = wave_sum_f32(a); b = wave_sum_f32(b); c = wave_sum_f32(c);
    if (threadIdx.x == 0)
        out[0] = wl * a / fmaxf(counts[0], 1.f) + wse * (b / fmaxf(counts[1], 1.f) + c / fmaxf(counts[2], 1.f));
}

static int check_train_mlp(const jm_mlp3_t* m, const char* who) {
    JM_REQUIRE(m && m->w1 && m->b1 && m->w2 && m->b2 && m->w3 && m->b3, "%s: null weights", who);
    JM_REQUIRE(m->c >= 32 && m->c % 32 == 0 && m->h1 >= 32 && m->h1 % 32 == 0 && m->h2 >= 32 && m->h2 % 32 == 0,
               "%s: channel sizes must be multiples of 32 (c=%d h1=%d h2=%d)", who, m->c, m->h1, m->h2);
    JM_REQUIRE(((reinterpret_cast<uintptr_t>(m->w1) | reinterpret_cast<uintptr_t>(m->w2)) & 15u) == 0,
               "%s: weights must be 16-byte aligned", who);
    return JM_OK;
}

static int check_grads(const jm_mlp3_grad_t* g, const char* who) {
    JM_REQUIRE(g && g->dw1 && g->db1 && g->dw2 && g->db2 && g->dw3 && g->db3, "%s: null gradient pointer", who);
    return JM_OK;
}

// how the M contraction rows of dW = dY^T X are split over workgroups: >= ~256 workgroups of 16 output tiles each
static int split_rows(int M, int tiles) {
    int splits = max(1, min(64, 256 / max(tiles, 1)));
    int chunk = divup(divup(M, splits), TBK) * TBK;
    return max(chunk, TBK);
}

struct TrainWs {      // carved out of the caller's workspace
    float *h1, *h2, *y, *ypart, *dy, *p_dw3, *p_db2, *p_db3, *p_db1, *p_dw2, *p_dw1;
    int chunks, chunk1, chunk2, splits1, splits2;
    size_t bytes;
};

static TrainWs carve(int M, const jm_mlp3_t* mlp, void* ws) {
    TrainWs w{};
    char* p = (char*)ws;
    auto take = [&](size_t n) { float* q = (float*)p; p += align_up(n * sizeof(float), 256); return q; };
    w.chunks = divup(M, CH);
    w.chunk2 = split_rows(M, divup(mlp->h2, TBM) * divup(mlp->h1, TBN));
    w.chunk1 = split_rows(M, divup(mlp->h1, TBM) * divup(mlp->c, TBN));
    w.splits2 = divup(M, w.chunk2);
    w.splits1 = divup(M, w.chunk1);
    w.h1 = take((size_t)M * mlp->h1);
    w.h2 = take((size_t)M * mlp->h2);
    w.y = take(M);
    // one slot per 32 columns (single-wave kernel) or two per 128-column tile, written also when h2 <= 64 leaves the second
    // empty: h2 <= 32 needs two slots, not divup(h2, 32) = 1
    w.ypart = take((size_t)M * (size_t)imax(divup(mlp->h2, 32), 2 * divup(mlp->h2, TBN)));
    w.dy = take(M);
    w.p_dw3 = take((size_t)w.chunks * mlp->h2);
    w.p_db2 = take((size_t)w.chunks * mlp->h2);
    w.p_db3 = take(w.chunks);
    w.p_db1 = take((size_t)w.chunks * mlp->h1);
    w.p_dw2 = take((size_t)w.splits2 * mlp->h2 * mlp->h1);
    w.p_dw1 = take((size_t)w.splits1 * mlp->h1 * mlp->c);
    w.bytes = (size_t)(p - (char*)ws);
    return w;
}

// forward of the 3-layer head keeping both hidden activations: pair rows (x == nullptr) or plain rows x (M, C)
static int mlp_train_forward(int M, const float* x, const float* pf, const float* df, int D, int PD, const jm_mlp3_t* mlp,
                             const TrainWs& w, hipStream_t s) {
    TGemm a{};
    a.M = M; a.N = mlp->h1; a.K = mlp->c; a.kchunk = mlp->c;
    a.A = x; a.lda = mlp->c; a.pf = pf; a.df = df; a.D = D; a.PD = PD; a.ldp = mlp->c;
    a.B = mlp->w1; a.ldb = mlp->c; a.bias = mlp->b1; a.out = w.h1; a.ldo = mlp->h1;
    const bool small = x && M <= SMALL_M && mlp->c % 8 == 0 && mlp->h1 % 8 == 0 && mlp->h2 % 8 == 0;
    if (small) launch_tgemm_small<B_ROWS, E_RELU>(a, s);
    else if (x) launch_tgemm<A_ROWS, B_ROWS, E_RELU>(a, s);
    else launch_tgemm<A_PAIR, B_ROWS, E_RELU>(a, s);
    TGemm b{};
    b.M = M; b.N = mlp->h2; b.K = mlp->h1; b.kchunk = mlp->h1;
    b.A = w.h1; b.lda = mlp->h1; b.B = mlp->w2; b.ldb = mlp->h1; b.bias = mlp->b2; b.w3 = mlp->w3; b.score = w.ypart;
    b.out = w.h2; b.ldo = mlp->h2;
    if (small) launch_tgemm_small<B_ROWS, E_RELU_PROJ>(b, s); else launch_tgemm<A_ROWS, B_ROWS, E_RELU_PROJ>(b, s);
    const int slots = small ? divup(mlp->h2, 32) : 2 * divup(mlp->h2, TBN);
    hipLaunchKernelGGL(train_score_sum_kernel, dim3(divup(M, 256)), dim3(256), 0, s, M, slots, w.ypart, mlp->b3, w.y);
    return check_launch("affinity_train forward");
}

// backward given dy (M) in w.dy; destroys w.h1 / w.h2 (they become dH1 / dH2)
static int mlp_train_backward(int M, const float* x, const float* pf, const float* df, int D, int PD, const jm_mlp3_t* mlp,
                              const TrainWs& w, const jm_mlp3_grad_t* g, float* dx, hipStream_t s) {
    const int c = mlp->c, h1 = mlp->h1, h2 = mlp->h2;
    hipLaunchKernelGGL(mlp_bwd_prep_kernel, dim3(w.chunks, divup(h2, CW)), dim3(CW), 0, s, M, h2, w.h2, w.dy, mlp->w3, w.p_dw3, w.p_db2, w.p_db3);
    // dW2 (h2, h1) = dH2^T H1: contraction over the M rows, both operands k-major in place
    TGemm t2{};
    t2.M = h2; t2.N = h1; t2.K = M; t2.kchunk = w.chunk2;
    t2.A = w.h2; t2.lda = h2; t2.B = w.h1; t2.ldb = h1; t2.out = w.p_dw2; t2.ldo = h1;
    launch_tgemm<A_KMAJOR, B_KMAJOR, E_PARTIAL>(t2, s);
    // dH1 = (dH2 W2) .* (H1 > 0), written over H1
    TGemm n1{};
    n1.M = M; n1.N = h1; n1.K = h2; n1.kchunk = h2;
    n1.A = w.h2; n1.lda = h2; n1.B = mlp->w2; n1.ldb = h1; n1.mask = w.h1; n1.out = w.h1; n1.ldo = h1;
    if (x && M <= SMALL_M && h2 % 8 == 0) launch_tgemm_small<B_KMAJOR, E_MASK>(n1, s); else launch_tgemm<A_ROWS, B_KMAJOR, E_MASK>(n1, s);
    hipLaunchKernelGGL(colsum_kernel, dim3(w.chunks, divup(h1, CW)), dim3(CW), 0, s, M, h1, w.h1, w.p_db1);
    if (dx) {
        // d(loss)/d(input rows) (M, c) = dH1 W1: W1 (h1, c) is k-major for this product as it lies (joint training: the rows
        // are |p_i - d_j| or the start / end features; jm_affinity_train_feature_grad carries it on to the RoI features)
        TGemm nx{};
        nx.M = M; nx.N = c; nx.K = h1; nx.kchunk = h1;            // one split: E_PARTIAL stores the plain product
        nx.A = w.h1; nx.lda = h1; nx.B = mlp->w1; nx.ldb = c; nx.out = dx; nx.ldo = c;
        launch_tgemm<A_ROWS, B_KMAJOR, E_PARTIAL>(nx, s);
    }
    // dW1 (h1, c) = dH1^T X, X = plain rows or |p_i - d_j| regenerated per tile
    TGemm t1{};
    t1.M = h1; t1.N = c; t1.K = M; t1.kchunk = w.chunk1;
    t1.A = w.h1; t1.lda = h1; t1.B = x; t1.ldb = c; t1.pf = pf; t1.df = df; t1.D = D; t1.PD = PD; t1.ldp = c;
    t1.out = w.p_dw1; t1.ldo = c;
    if (x) launch_tgemm<A_KMAJOR, B_KMAJOR, E_PARTIAL>(t1, s); else launch_tgemm<A_KMAJOR, B_KMAJOR_PAIR, E_PARTIAL>(t1, s);
    SumJobs j{};
    j.njobs = 6;
    const int np[6] = {w.splits1, w.chunks, w.splits2, w.chunks, w.chunks, w.chunks};
    const int ln[6] = {h1 * c, h1, h2 * h1, h2, h2, 1};
    const float* in[6] = {w.p_dw1, w.p_db1, w.p_dw2, w.p_db2, w.p_dw3, w.p_db3};
    float* out[6] = {g->dw1, g->db1, g->dw2, g->db2, g->dw3, g->db3};
    int blocks = 0;
    for (int q = 0; q < 6; ++q) {
        j.nparts[q] = np[q]; j.len[q] = ln[q]; j.in[q] = in[q]; j.out[q] = out[q];
        j.first_block[q] = blocks;
        blocks += divup(ln[q], 256);
    }
    j.first_block[6] = blocks;
    hipLaunchKernelGGL(sum_partials_kernel, dim3(blocks), dim3(256), 0, s, j);
    return check_launch("affinity_train backward");
}

// ---------------------------------------------------------------------------------------------------------------------
// d(loss)/d(pooled features) from d(loss)/d(pair rows) of the link head (dxl: (F, R, R, C), zero on invalid pairs) and
// d(loss)/d(start / end features) of the se head (dxs: (F, 2R, C): rows [0, R) start feature of next slot j = mean over the
// prev representatives of |p_i - d_j|, rows [R, 2R) end feature of prev slot i = mean over the next representatives):
//   G[i][j][c] = sign(p_i - d_j) * (dxl[i][j][c] + [rep_prev[i] && rep_next[j]] * (dxs_start[j][c] / n_prev + dxs_end[i][c] / n_next))
//   dP[i] = sum_j G[i][j],  dD[j] = -sum_i G[i][j]       (|.|' = sign, 0 at 0 as torch.abs)
// One thread per (slot, channel) walks the other side in index order: deterministic.  blockIdx.y: 0 = prev slots, 1 = next slots.
__global__ void __launch_bounds__(256)
train_dpooled_kernel(int R, int C, const float* __restrict__ pp, const float* __restrict__ pn, const int* __restrict__ rep_prev,
                     const int* __restrict__ rep_next, const int* __restrict__ n_pair, const float* __restrict__ dxl,
                     const float* __restrict__ dxs, float* __restrict__ dpp, float* __restrict__ dpn) {
    const int f = blockIdx.x / R, k = blockIdx.x % R;
    const bool prev = blockIdx.y == 0;
    const int* rp = rep_prev + (size_t)f * R;
    const int* rn = rep_next + (size_t)f * R;
    const float inv_p = 1.f / (float)max(n_pair[2 * f], 1), inv_n = 1.f / (float)max(n_pair[2 * f + 1], 1);
    const float* P = pp + (size_t)f * R * C;
    const float* D = pn + (size_t)f * R * C;
    const float* XL = dxl + (size_t)f * R * R * C;
    const float* XS = dxs ? dxs + (size_t)f * 2 * R * C : nullptr;
    float* dst = (prev ? dpp : dpn) + ((size_t)f * R + k) * C;
    const bool me_rep = prev ? rp[k] != 0 : rn[k] != 0;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float acc = 0.f;
        if (me_rep) {                                        // non-representative slots take part in nothing
            const float mine = prev ? P[(size_t)k * C + c] : D[(size_t)k * C + c];
            for (int q = 0; q < R; ++q) {
                const bool other_rep = prev ? rn[q] != 0 : rp[q] != 0;
                if (!other_rep) continue;
                const int i = prev ? k : q, j = prev ? q : k;
                const float other = prev ? D[(size_t)q * C + c] : P[(size_t)q * C + c];
                const float diff = prev ? mine - other : other - mine;            // p_i - d_j
                const float sg = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
                float gsum = XL[((size_t)i * R + j) * C + c];
                if (XS) gsum += XS[(size_t)j * C + c] * inv_p + XS[(size_t)(R + i) * C + c] * inv_n;
                acc += sg * gsum;
            }
            if (!prev) acc = -acc;
        }
        dst[c] = acc;
    }
}

// ... and on to the RoI features through the mean pooling of get_unique_tid_feature: every foreground RoI k of a frame gets
// d(pooled)[representative of its track id] / (RoIs with that id); frames interleaved (prev, next, ...) as in train_pool_kernel
__global__ void __launch_bounds__(128)
train_dfeat_kernel(int R, int C, const float* __restrict__ tids, const float* __restrict__ dpp, const float* __restrict__ dpn,
                   float* __restrict__ dfeat) {
    __shared__ int rep_s, cnt_s;
    const int f = blockIdx.x / R, k = blockIdx.x % R;
    const float* t = tids + (size_t)f * R;
    const float tk = t[k];
    if (threadIdx.x == 0) {
        int rep = -1, cnt = 0;
        if (tk > 0.f)
            for (int q = 0; q < R; ++q)
                if (t[q] == tk) { if (rep < 0) rep = q; ++cnt; }
        rep_s = rep; cnt_s = cnt;
    }
    __syncthreads();
    const int rep = rep_s;
    const float inv = cnt_s > 0 ? 1.f / (float)cnt_s : 0.f;
    const float* src = ((f & 1) ? dpn : dpp) + ((size_t)(f >> 1) * R + max(rep, 0)) * C;
    float* dst = dfeat + ((size_t)f * R + k) * C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) dst[c] = rep >= 0 ? src[c] * inv : 0.f;
}

}  // namespace jm

using namespace jm;

/* ------------------------------------------------------------------ C ABI (include/jmodt_hip.h) */

extern "C" int jm_affinity_train_prepare(int npairs, int r, int c, const float* feats, const float* tids, float* pooled_prev,
                                         float* pooled_next, int* rep_ws, int* rep_prev, int* rep_next, int* n_pair,
                                         float* gt_starts, float* gt_ends, float* counts, jm_stream_t stream) {
    JM_REQUIRE(npairs >= 0 && r >= 1 && r <= 256 && c >= 1, "affinity_train_prepare: bad sizes (pairs=%d R=%d C=%d; R <= 256)", npairs, r, c);
    hipStream_t s = (hipStream_t)stream;
    JM_REQUIRE(counts, "affinity_train_prepare: null counts");
    (void)jm_zero_async(counts, 3 * sizeof(float), s);
    if (npairs == 0) return JM_OK;
    JM_REQUIRE(feats && tids && pooled_prev && pooled_next && rep_ws && rep_prev && rep_next && n_pair && gt_starts && gt_ends,
               "affinity_train_prepare: null pointer");
    hipLaunchKernelGGL(train_pool_kernel, dim3((unsigned)(2 * npairs * r)), dim3(128), 0, s, r, c, feats, tids, pooled_prev, pooled_next, rep_ws);
    hipLaunchKernelGGL(train_mask_kernel, dim3((unsigned)npairs), dim3(256), 0, s, r, tids, rep_ws, rep_prev, rep_next, n_pair,
                       gt_starts, gt_ends, counts);
    return check_launch("affinity_train_prepare");
}

extern "C" int jm_affinity_train_loss_value(int npairs, const float* link_loss_part, const float* se_loss_part, const float* counts,
                                            float link_weight, float se_weight, float* loss, jm_stream_t stream) {
    JM_REQUIRE(npairs >= 0 && counts && loss && (npairs == 0 || (link_loss_part && se_loss_part)), "affinity_train_loss_value: bad arguments");
    hipLaunchKernelGGL(train_loss_value_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, npairs, link_loss_part, se_loss_part, counts,
                       link_weight, se_weight, loss);
    return check_launch("affinity_train_loss_value");
}

extern "C" size_t jm_affinity_train_link_workspace_bytes(int npairs, int r, const jm_mlp3_t* link) {
    if (npairs <= 0 || r <= 0 || !link) return 0;
    return carve(npairs * r * r, link, nullptr).bytes;
}

extern "C" size_t jm_affinity_train_se_workspace_bytes(int npairs, int r, const jm_mlp3_t* se) {
    if (npairs <= 0 || r <= 0 || !se) return 0;
    return carve(npairs * 2 * r, se, nullptr).bytes + align_up((size_t)npairs * 2 * r * se->c * sizeof(float), 256);
}

extern "C" int jm_affinity_train_link_step(int npairs, int r, const float* pooled_prev, const float* pooled_next,
                                           const int* rep_prev, const int* rep_next, const float* tids, const float* counts,
                                           float loss_weight, const jm_mlp3_t* link, float* link_out, float* gt_links,
                                           float* loss_part, const jm_mlp3_grad_t* grads, float* dx, void* ws, size_t ws_bytes,
                                           jm_stream_t stream) {
    JM_REQUIRE(npairs >= 0 && r >= 1 && r <= 128, "affinity_train_link: bad sizes (pairs=%d R=%d; R <= 128)", npairs, r);
    if (npairs == 0) return JM_OK;
    int rc = check_train_mlp(link, "affinity_train link_layer");
    if (rc) return rc;
    rc = check_grads(grads, "affinity_train link_layer");
    if (rc) return rc;
    JM_REQUIRE(pooled_prev && pooled_next && rep_prev && rep_next && tids && counts && loss_part && ws, "affinity_train_link: null pointer");
    JM_REQUIRE(((reinterpret_cast<uintptr_t>(pooled_prev) | reinterpret_cast<uintptr_t>(pooled_next) | reinterpret_cast<uintptr_t>(ws)) & 15u) == 0,
               "affinity_train_link: 16-byte alignment");
    JM_REQUIRE((long long)npairs * r * r < (1LL << 30), "affinity_train_link: too many pair rows");
    const int M = npairs * r * r;
    const TrainWs w = carve(M, link, ws);
    if (ws_bytes < w.bytes) { set_error("affinity_train_link: workspace %zu < %zu bytes", ws_bytes, w.bytes); return JM_EWORKSPACE; }
    hipStream_t s = (hipStream_t)stream;
    rc = mlp_train_forward(M, nullptr, pooled_prev, pooled_next, r, r * r, link, w, s);
    if (rc) return rc;
    const size_t lds = ((size_t)2 * r * r + 10 * r) * sizeof(float);
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)train_link_loss_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(train_link_loss_kernel, dim3((unsigned)npairs), dim3(1024), lds, s, r, w.y, rep_prev, rep_next, tids, counts,
                       loss_weight, link_out, gt_links, w.dy, loss_part);
    return mlp_train_backward(M, nullptr, pooled_prev, pooled_next, r, r * r, link, w, grads, dx, s);
}

extern "C" int jm_affinity_train_se_step(int npairs, int r, const float* pooled_prev, const float* pooled_next,
                                         const int* rep_prev, const int* rep_next, const int* n_pair, const float* gt_starts,
                                         const float* gt_ends, const float* counts, float loss_weight, const jm_mlp3_t* se,
                                         float* se_logits, float* loss_part, const jm_mlp3_grad_t* grads, float* dx, void* ws,
                                         size_t ws_bytes, jm_stream_t stream) {
    JM_REQUIRE(npairs >= 0 && r >= 1 && r <= 256, "affinity_train_se: bad sizes (pairs=%d R=%d)", npairs, r);
    if (npairs == 0) return JM_OK;
    int rc = check_train_mlp(se, "affinity_train se_layer");
    if (rc) return rc;
    rc = check_grads(grads, "affinity_train se_layer");
    if (rc) return rc;
    JM_REQUIRE(pooled_prev && pooled_next && rep_prev && rep_next && n_pair && gt_starts && gt_ends && counts && loss_part && ws,
               "affinity_train_se: null pointer");
    JM_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 15u) == 0, "affinity_train_se: 16-byte alignment");
    const int M = npairs * 2 * r;
    if (ws_bytes < jm_affinity_train_se_workspace_bytes(npairs, r, se)) { set_error("affinity_train_se: workspace too small"); return JM_EWORKSPACE; }
    float* feat = (float*)ws;
    const TrainWs w = carve(M, se, (char*)ws + align_up((size_t)M * se->c * sizeof(float), 256));
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(train_se_feat_kernel, dim3((unsigned)M), dim3(256), 0, s, r, se->c, pooled_prev, pooled_next, rep_prev, rep_next,
                       n_pair, feat);
    rc = mlp_train_forward(M, feat, nullptr, nullptr, 1, 1, se, w, s);
    if (rc) return rc;
    if (se_logits) (void)hipMemcpyAsync(se_logits, w.y, (size_t)M * sizeof(float), hipMemcpyDeviceToDevice, s);
    hipLaunchKernelGGL(train_se_loss_kernel, dim3((unsigned)npairs), dim3(256), 0, s, r, w.y, rep_prev, rep_next, gt_starts, gt_ends,
                       counts, loss_weight, w.dy, loss_part);
    return mlp_train_backward(M, feat, nullptr, nullptr, 1, 1, se, w, grads, dx, s);
}

extern "C" int jm_affinity_train_feature_grad(int npairs, int r, int c, const float* tids, const float* pooled_prev,
                                              const float* pooled_next, const int* rep_prev, const int* rep_next, const int* n_pair,
                                              const float* dx_link, const float* dx_se, float* dpooled_ws, float* dfeat,
                                              jm_stream_t stream) {
    JM_REQUIRE(npairs >= 0 && r >= 1 && r <= 256 && c >= 1, "affinity_train_feature_grad: bad sizes");
    if (npairs == 0) return JM_OK;
    JM_REQUIRE(tids && pooled_prev && pooled_next && rep_prev && rep_next && n_pair && dx_link && dpooled_ws && dfeat,
               "affinity_train_feature_grad: null pointer");
    hipStream_t s = (hipStream_t)stream;
    float* dpp = dpooled_ws;
    float* dpn = dpooled_ws + (size_t)npairs * r * c;
    hipLaunchKernelGGL(train_dpooled_kernel, dim3((unsigned)(npairs * r), 2), dim3(256), 0, s, r, c, pooled_prev, pooled_next, rep_prev,
                       rep_next, n_pair, dx_link, dx_se, dpp, dpn);
    hipLaunchKernelGGL(train_dfeat_kernel, dim3((unsigned)(2 * npairs * r)), dim3(128), 0, s, r, c, tids, dpp, dpn, dfeat);
    return check_launch("affinity_train_feature_grad");
}
