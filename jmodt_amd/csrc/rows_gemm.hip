// rows_gemm.hip — dense layers of the TRAINING path on row-major ("point-major") activations: forward, data gradient and
// weight gradient of  Y = act(X W^T + b)  on the fp32 matrix cores (v_mfma_f32_32x32x2_f32), for the joint-mode step of
// BASELINE configs[3] (tools/train.py:96-107 without cfg.TRAIN.FINETUNE).
//
// What it replaces: torch autograd over the reference's module tree — pytorch_utils.py:6-33 (SharedMLP = Conv2d 1x1 + BN +
// ReLU on (B, C, npoint, nsample) tensors), pointnet2_modules.py:46-61 (set abstraction), :139-153 (feature propagation),
// backbone.py:35-81 (LI-Fusion attention block), rpn.py:34-58 / rcnn.py:43-89 (heads).  Every one of those is, per point or
// per (centre, neighbour) row, a chain of dense layers; in the training path all of them keep their activations as
// (rows, channels) row-major tensors, so ONE GEMM family serves them all and nothing is ever transposed:
//     forward   Y (M, N)  = act(X (M, K) W (N, K)^T + b)      X may be the concatenation [X1 | X2] of two row tensors
//                                                             (skip connections: never materialised)
//     dgrad     dX (M, K) = (dY (M, N) W (N, K)) .* (mask > 0)  W read k-major IN PLACE; mask = the layer input's own
//                                                             post-ReLU activation, so dX is already the gradient w.r.t. the
//                                                             previous layer's pre-activation
//     wgrad     dW (N, K) = dY^T X, split over the M rows       both operands k-major in place; split partials reduced in a
//                                                             fixed order by rows_wgrad_reduce_kernel (no float atomics)
// The row count may live in DEVICE memory (`m_dev`): the set-abstraction rows are compacted to the DISTINCT (centre, neighbour)
// pairs of every ball-query group on the device (rows_ops.hip: sa_rows_plan), and the host never learns how many there are.
// Grids are persistent over the tiles that exist.
//
// Tile: 128 x 128 x 16, four waves of 64 x 64 (2 x 2 MFMA blocks of 32 x 32), operands double-buffered through k-major LDS
// tiles (the layout of csrc/affinity_train.hip's train_gemm_kernel, whose operand forms these are).  Contraction tails
// (K % 16 != 0: the 196-wide hidden layer of RPN SA3, config.py:76) are zero-filled at LDS store time; every global load is
// unconditional on a clamped, valid address.  Requirements (checked by the entries): K % 4 == 0 and lda / ldb % 4 == 0 for
// row operands, N % 4 == 0 for k-major operands.
#include "jm_rows.h"
#ifdef JM_TOOLS_BUILD
#include "jmodt_hip_tools.h"
#endif

namespace jm {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int RBM = 128, RBN = 128, RBK = 32, RLDP = RBM + 4;
constexpr int RNLD = RBK / 8;
#ifndef JM_PGRID
#define JM_PGRID 2048
#endif       // float4 loads per thread and operand per k-tile (128 x RBK floats / 256 threads / 4)
enum { RM_FWD = 0, RM_DGRAD = 1, RM_WGRAD = 2 };

// A matrix stored PIXEL-SHUFFLED (round 5: the image branch's kernel == stride transposed convolutions as GEMMs, backbone.py:187-193):
// the (rows m = (b, y, x), columns n = ((dy k + dx) r + rr)) matrix is the channels-last map Y (B, h k, w k, ctot) with
// element (m, n) at Y[b][y k + dy][x k + dx][coff + rr] — the GEMM's output rows land where ConvTranspose2d(kernel = stride = k) puts
// them, and the backward's operand is read from there; r % 4 == 0, so a float4 along n stays inside one pixel
struct RShuffle { int on, k, r, h, w, ctot, coff; };

// the address is ADDITIVE in a row part and a column part: shuf(m, n) = row_part(m) + col_part(n).  Rows are walked incrementally
// (RowPos: one division per tile, not per element — the first form divided five times per stored element and was slower than the
// library convolution it replaces)
struct RowPos { int b, y, x; };
__device__ __forceinline__ RowPos row_pos(const RShuffle& s, int m) {
    const int t = m / s.w;
    return RowPos{t / s.h, t % s.h, m - t * s.w};
}
__device__ __forceinline__ void row_advance(const RShuffle& s, RowPos& p, int d) {
    p.x += d;
    while (p.x >= s.w) {
        p.x -= s.w;
        if (++p.y == s.h) { p.y = 0; ++p.b; }
    }
}
__device__ __forceinline__ size_t row_part(const RShuffle& s, const RowPos& p) {
    return (((size_t)(p.b * s.h + p.y) * s.k) * ((size_t)s.w * s.k) + (size_t)p.x * s.k) * s.ctot;
}
__device__ __forceinline__ size_t col_part(const RShuffle& s, int n) {
    const int q = n / s.r, rr = n - q * s.r, dy = q / s.k, dx = q - dy * s.k;
    return ((size_t)dy * ((size_t)s.w * s.k) + dx) * s.ctot + s.coff + rr;
}
__device__ __forceinline__ size_t shuf_addr(const RShuffle& s, int m, int n) { return row_part(s, row_pos(s, m)) + col_part(s, n); }

struct RGemm {
    RShuffle sh;                   // FWD: the OUTPUT is stored shuffled; DGRAD / WGRAD: the A operand (dY) is read shuffled
    int M, N, K;                   // output rows, output columns, contraction length
    const int* m_dev;              // FWD / DGRAD: valid rows = min(M, *m_dev); WGRAD: valid contraction = min(K, *m_dev)
    const float* A; int lda;       // FWD / DGRAD: (M, K) rows; WGRAD: (K, M) — element (row r, contraction c) = A[c * lda + r]
    const float* A2; int lda2;     // FWD: contraction indices K1 .. K - 1 are columns 0 .. of A2's rows
    int K1;
    const float* B; int ldb;       // FWD: (N, K) rows (the layer's weight); DGRAD / WGRAD: (K, N) k-major
    const float* bias; int act;    // FWD: act 0 none, 1 ReLU, 2 tanh
    const float* rowscale;         // FWD: out *= rowscale[row] after the activation (the attention gate)
    const float* mask; int ldm;    // DGRAD: out = mask[row, col] > 0 ? acc : 0
    int accumulate;                // DGRAD: out += acc
    float* out; int ldo;
    int splits;                    // WGRAD: blockIdx.y = split; splits > 1: partial s at out + s * M * ldo (reduced by
                                   // rows_wgrad_reduce_kernel), splits == 1: out IS dW (accumulate honoured)
    float* bias_out;               // WGRAD: row sums of A = column sums of dY (the bias gradient): (splits, M) partials, or the (M)
                                   // result itself when splits == 1; NULL: skipped
};

__host__ __device__ __forceinline__ int wgrad_live_splits(int rows, int splits) {
    const int want = (rows + 127) / 128;
    return want < 1 ? 1 : (want > splits ? splits : want);
}

template <int MODE>
__global__ void __launch_bounds__(256)
rows_gemm_kernel(RGemm p) {
    __shared__ __attribute__((aligned(16))) float As[2][RBK][RLDP];
    __shared__ __attribute__((aligned(16))) float Bs[2][RBK][RLDP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, lk = lane >> 5;
    const int valid = p.m_dev ? min(MODE == RM_WGRAD ? p.K : p.M, max(*p.m_dev, 0)) : (MODE == RM_WGRAD ? p.K : p.M);
    const int Mv = MODE == RM_WGRAD ? p.M : valid;
    const int Kv = MODE == RM_WGRAD ? valid : p.K;
    const int ntn = (p.N + RBN - 1) / RBN;
    const int ntiles = ((Mv + RBM - 1) / RBM) * ntn;
    int k_begin = 0, k_end = Kv;
    if (MODE == RM_WGRAD) {
        // the split count was chosen on the host for the row CAPACITY; with the count in device memory only the splits that have
        // at least 128 rows' worth of work exist (the others return: rows_wgrad_reduce_kernel applies the same rule and never
        // reads their partials) — a scale whose groups hold one distinct neighbour each runs 4 splits, not 128
        const int live = wgrad_live_splits(Kv, p.splits);
        if ((int)blockIdx.y >= live) return;
        const int kchunk = ((Kv + live - 1) / live + RBK - 1) / RBK * RBK;
        k_begin = min((int)blockIdx.y * kchunk, Kv);
        k_end = min(Kv, k_begin + kchunk);
    }
    const int nkt = (k_end - k_begin + RBK - 1) / RBK;

    // staging roles: 128 x RBK floats per operand per k-tile = RNLD float4 per thread
    //   row operands: 4 consecutive k of one row, scattered to S[k .. k + 3][row]
    //   k-major operands: 4 consecutive rows of one contraction index, one ds_write_b128 to S[k][row .. row + 3]
    // RBK = 32: a k-tile is 2048 MFMA cycles per wave (0.85 us), longer than a global-load round trip, so even ONE workgroup per CU
    // (the short grids of the small layers) keeps the matrix pipe fed; with RBK = 16 those launches ran at load latency per k-tile
    int t_row[RNLD], t_kq[RNLD], d_k[RNLD], d_r4[RNLD];
#pragma unroll
    for (int i = 0; i < RNLD; ++i) {
        const int f = tid + 256 * i;
        t_row[i] = f / (RBK / 4); t_kq[i] = (f % (RBK / 4)) * 4;
        d_k[i] = f >> 5;          d_r4[i] = (f & 31) * 4;
    }

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int m0 = (tile / ntn) * RBM, n0 = (tile % ntn) * RBN;
        float4 ra[RNLD], rb[RNLD];
        bool a_in[RNLD], b_in[RNLD];
        // shuffled A operand: the part of the address that is fixed for the tile (DGRAD: this thread's rows; WGRAD: its columns) and,
        // for WGRAD, the walking position of the contraction row of every load slot
        size_t sh_fix[RNLD];
        RowPos sh_pos[RNLD];
        int sh_at[RNLD];
        if (p.sh.on && MODE != RM_FWD) {
#pragma unroll
            for (int i = 0; i < RNLD; ++i) {
                if (MODE == RM_DGRAD) {
                    sh_fix[i] = row_part(p.sh, row_pos(p.sh, min(m0 + t_row[i], Mv - 1)));
                } else {
                    sh_fix[i] = col_part(p.sh, min(m0 + d_r4[i], p.M - 4));
                    sh_at[i] = min(k_begin + d_k[i], max(Kv - 1, 0));
                    sh_pos[i] = row_pos(p.sh, sh_at[i]);
                }
            }
        }
        auto g_load = [&](int k0) {          // k0 = absolute contraction index of the k-tile's first element
#pragma unroll
            for (int i = 0; i < RNLD; ++i) {
                if (MODE == RM_WGRAD) {
                    const int kc = k0 + d_k[i];
                    a_in[i] = kc < k_end; b_in[i] = a_in[i];
                    const size_t kk = (size_t)min(kc, max(Kv - 1, 0));
                    if (p.sh.on) {
                        if ((int)kk > sh_at[i]) { row_advance(p.sh, sh_pos[i], (int)kk - sh_at[i]); sh_at[i] = (int)kk; }   // (k-tiles ascend)
                        ra[i] = *reinterpret_cast<const float4*>(p.A + row_part(p.sh, sh_pos[i]) + sh_fix[i]);
                    } else {
                        ra[i] = *reinterpret_cast<const float4*>(p.A + kk * p.lda + min(m0 + d_r4[i], p.M - 4));
                    }
                    rb[i] = *reinterpret_cast<const float4*>(p.B + kk * p.ldb + min(n0 + d_r4[i], p.N - 4));
                } else {
                    const int m = min(m0 + t_row[i], Mv - 1);
                    const int k = k0 + t_kq[i];
                    a_in[i] = k < p.K;
                    const int kc = min(k, p.K - 4);
                    if (MODE == RM_FWD && p.A2 != nullptr && kc >= p.K1)
                        ra[i] = *reinterpret_cast<const float4*>(p.A2 + (size_t)m * p.lda2 + (kc - p.K1));
                    else if (MODE == RM_DGRAD && p.sh.on)
                        ra[i] = *reinterpret_cast<const float4*>(p.A + sh_fix[i] + col_part(p.sh, kc));
                    else
                        ra[i] = *reinterpret_cast<const float4*>(p.A + (size_t)m * p.lda + kc);
                    if (MODE == RM_FWD) {
                        b_in[i] = a_in[i];
                        const int n = min(n0 + t_row[i], p.N - 1);
                        rb[i] = *reinterpret_cast<const float4*>(p.B + (size_t)n * p.ldb + kc);
                    } else {
                        const int kd = k0 + d_k[i];
                        b_in[i] = kd < p.K;
                        rb[i] = *reinterpret_cast<const float4*>(p.B + (size_t)min(kd, p.K - 1) * p.ldb + min(n0 + d_r4[i], p.N - 4));
                    }
                }
            }
        };
        auto s_store = [&](int buf) {
#pragma unroll
            for (int i = 0; i < RNLD; ++i) {
                float4 a = a_in[i] ? ra[i] : make_float4(0.f, 0.f, 0.f, 0.f);
                float4 b = b_in[i] ? rb[i] : make_float4(0.f, 0.f, 0.f, 0.f);
                if (MODE == RM_WGRAD) {
                    *reinterpret_cast<float4*>(&As[buf][d_k[i]][d_r4[i]]) = a;
                } else {
                    As[buf][t_kq[i] + 0][t_row[i]] = a.x; As[buf][t_kq[i] + 1][t_row[i]] = a.y;
                    As[buf][t_kq[i] + 2][t_row[i]] = a.z; As[buf][t_kq[i] + 3][t_row[i]] = a.w;
                }
                if (MODE == RM_FWD) {
                    Bs[buf][t_kq[i] + 0][t_row[i]] = b.x; Bs[buf][t_kq[i] + 1][t_row[i]] = b.y;
                    Bs[buf][t_kq[i] + 2][t_row[i]] = b.z; Bs[buf][t_kq[i] + 3][t_row[i]] = b.w;
                } else {
                    *reinterpret_cast<float4*>(&Bs[buf][d_k[i]][d_r4[i]]) = b;
                }
            }
        };

        float bsum = 0.f;
        const bool do_bias = MODE == RM_WGRAD && p.bias_out != nullptr && n0 == 0 && tid < RBM;
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        if (nkt > 0) {
            g_load(k_begin);
            s_store(0);
            __syncthreads();
            for (int kt = 0; kt < nkt; ++kt) {
                const int buf = kt & 1;
                g_load(k_begin + min(kt + 1, nkt - 1) * RBK);       // unconditional (last tile re-read, unused)
                __builtin_amdgcn_sched_barrier(0);                  // loads stay above the MFMAs
#pragma unroll
                for (int kk = 0; kk < RBK / 2; ++kk) {
                    const int k2 = kk * 2 + lk;
                    const float a0 = As[buf][k2][wm * 64 + lr], a1 = As[buf][k2][wm * 64 + 32 + lr];
                    const float b0 = Bs[buf][k2][wn * 64 + lr], b1 = Bs[buf][k2][wn * 64 + 32 + lr];
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (do_bias) {          // the first column tile of every row block also sums its dY tile over the contraction
#pragma unroll
                    for (int kk = 0; kk < RBK; ++kk) bsum += As[buf][kk][tid];
                }
                if (kt + 1 < nkt) s_store(buf ^ 1);
                __syncthreads();
            }
        }

        // epilogue.  C layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
        float* out = p.out + ((MODE == RM_WGRAD && p.splits > 1) ? (size_t)blockIdx.y * p.M * p.ldo : 0);
        if (do_bias && m0 + tid < p.M) {
            float* bo = p.bias_out + (p.splits > 1 ? (size_t)blockIdx.y * p.M : 0) + m0 + tid;
            *bo = (p.splits == 1 && p.accumulate) ? *bo + bsum : bsum;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            // shuffled output: this thread's 16 rows of the block, walked from the block's first one
            size_t sh_row[16];
            if (MODE == RM_FWD && p.sh.on) {
                RowPos rp = row_pos(p.sh, min(m0 + wm * 64 + i * 32 + 4 * lk, max(Mv - 1, 0)));
                int at = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int off = (r & 3) + 8 * (r >> 2);
                    row_advance(p.sh, rp, off - at);
                    at = off;
                    sh_row[r] = row_part(p.sh, rp);
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = n0 + wn * 64 + j * 32 + lr;
                const bool cok = col < p.N;
                const float bv = (MODE == RM_FWD && p.bias != nullptr && cok) ? p.bias[col] : 0.f;
                const size_t sh_col = (MODE == RM_FWD && p.sh.on) ? col_part(p.sh, min(col, p.N - 1)) : 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                    const bool ok = cok && row < Mv;
                    float v = acc[i][j][r];
                    if (MODE == RM_FWD) {
                        v += bv;
                        if (p.act == 1) v = fmaxf(v, 0.f);
                        else if (p.act == 2) v = tanhf(v);
                        if (p.rowscale != nullptr && ok) v *= p.rowscale[row];
                    }
                    if (MODE == RM_DGRAD) {
                        if (p.mask != nullptr) v = (ok && p.mask[(size_t)row * p.ldm + col] > 0.f) ? v : 0.f;
                        if (p.accumulate && ok) v += out[(size_t)row * p.ldo + col];
                    }
                    if (MODE == RM_WGRAD && p.splits == 1 && p.accumulate && ok) v += out[(size_t)row * p.ldo + col];
                    if (ok) out[(MODE == RM_FWD && p.sh.on) ? sh_row[r] + sh_col : (size_t)row * p.ldo + col] = v;
                }
            }
        }
    }
}

// dW[n, k] (+)= sum over the split partials, and the same for the bias gradient's (splits, N) partials (items behind the N * K / 4
// weight quads).  A workgroup = 32 items x 8 split groups: group g sums splits g, g + 8, ... on four independent chains, the groups
// are added in group order through LDS — a fixed order whatever the launch, so the result is reproducible (no float atomics); the
// first form (one thread per quad walking ALL splits) took 15 - 30 us on 17 workgroups for a 128 x 128 weight
__global__ void __launch_bounds__(256)
rows_wgrad_reduce_kernel(int N, int K, int splits, int m, const int* __restrict__ m_dev, const float* __restrict__ part,
                         float* __restrict__ dW, int ldw, const float* __restrict__ bias_part, float* __restrict__ dbias, int accumulate) {
    __shared__ float4 sh[8][32];
    splits = wgrad_live_splits(dev_count(m, m_dev), splits);          // the splits the weight-gradient kernel actually wrote
    const int item = blockIdx.x * 32 + (threadIdx.x & 31), grp = threadIdx.x >> 5;
    const int quads = N * (K / 4);
    const bool is_w = item < quads, is_b = !is_w && dbias != nullptr && item - quads < N;
    const int n = is_w ? item / (K / 4) : item - quads, k4 = is_w ? (item % (K / 4)) * 4 : 0;
    const float* src = is_w ? part + (size_t)n * K + k4 : bias_part + n;
    const size_t step = is_w ? (size_t)N * K : (size_t)N;
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
    auto ld = [&](int t) {
        if (is_w) return *reinterpret_cast<const float4*>(src + (size_t)t * step);
        return make_float4(is_b ? src[(size_t)t * step] : 0.f, 0.f, 0.f, 0.f);
    };
    auto add = [](float4& a, const float4& v) { a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; };
    if (is_w || is_b) {
        int t = grp;
        for (; t + 24 < splits; t += 32) {
            const float4 v = ld(t), w = ld(t + 8), x = ld(t + 16), y = ld(t + 24);
            add(s0, v); add(s1, w); add(s2, x); add(s3, y);
        }
        for (; t < splits; t += 8) add(s0, ld(t));
        add(s0, s2); add(s1, s3); add(s0, s1);
    }
    sh[grp][threadIdx.x & 31] = s0;
    __syncthreads();
    if (grp != 0 || !(is_w || is_b)) return;
    float4 s = sh[0][threadIdx.x];
#pragma unroll
    for (int g = 1; g < 8; ++g) add(s, sh[g][threadIdx.x]);
    if (is_w) {
        float* d = dW + (size_t)n * ldw + k4;
        if (accumulate) { d[0] += s.x; d[1] += s.y; d[2] += s.z; d[3] += s.w; }
        else { d[0] = s.x; d[1] = s.y; d[2] = s.z; d[3] = s.w; }
    } else {
        dbias[n] = accumulate ? dbias[n] + s.x : s.x;
    }
}

// Weight gradient, DIRECT form (round 5): dW (n, k) = dY (rows, n)^T X (rows, k) with both operands read straight from their rows into
// the MFMA registers — no LDS tile, no barrier in the contraction.  One MFMA step contracts the row pair (r, r + 1): lane (lr, lk)
// supplies dY[r + lk][column of lr] and X[r + lk][column of lr], i.e. a half wave reads 32 BA (32 BB) CONSECUTIVE floats of one row
// (float / float2 loads, fully coalesced); a wave owns BA x BB blocks of 32 x 32 (block (a, b): output row BA lr' + a, column
// BB lr + b of its 32 BA x 32 BB patch — the interleaving is what makes the loads contiguous).  The workgroup's four waves cover a
// tile of up to 128 x 128 as WA x WB patches; when the tile needs fewer than four (narrow layers: most of the network), the spare
// waves take OTHER rows of the split (row groups) and the partial tiles are added in group order through LDS.  U row pairs are in
// flight per wave.  The tiled kernel above spends 64 MFMAs per wave and 32 rows on a 128 x 128 tile whatever the layer's widths and
// walks a split at one global-load latency per 32 rows: 17 us at best, 53 us for the usual 1-tile x 128-split launch (measured inside
// the joint step, tools/joint_timeline.py); this one does the work the widths ask for.
template <int BA, int BB>
__global__ void __launch_bounds__(256)
rows_wgrad_direct_kernel(RGemm p, int WA, int WB) {
    constexpr int U = 16, NV = BA * BB * 16 + BA;
    __shared__ float red[3][NV][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lk = lane >> 5;
    const int Kv = dev_count(p.K, p.m_dev);
    const int live = wgrad_live_splits(Kv, p.splits);
    if ((int)blockIdx.y >= live) return;
    const int chunk = ((Kv + live - 1) / live + 7) & ~7;
    const int k_begin = min((int)blockIdx.y * chunk, Kv), k_end = min(Kv, k_begin + chunk);
    const int W = WA * WB, RG = 4 / W;
    const int wsub = wave % W, g = wave / W, wa = wsub / WB, wb = wsub % WB;
    const int ntn = (p.N + 127) / 128;
    const int m0 = ((int)blockIdx.x / ntn) * 128, n0 = ((int)blockIdx.x % ntn) * 128;
    const int a_base = m0 + wa * 32 * BA, b_base = n0 + wb * 32 * BB;
    // columns beyond the matrix are read from a clamped (valid) address: they only reach output elements that are never stored
    const float* Ap = p.A + min(a_base + BA * lr, p.M - BA);
    const float* Bp = p.B + min(b_base + BB * lr, p.N - BB);
    const int npairs = (k_end - k_begin + 1) / 2;
    const int nt = npairs > g ? (npairs - g + RG - 1) / RG : 0;          // this wave's row pairs: g, g + RG, ...
    const bool do_bias = p.bias_out != nullptr && wb == 0 && n0 == 0;
    const int last = max(Kv - 1, 0);

    // raw loads; rows beyond the split are zeroed when the slot is CONSUMED (a select at load time makes the load's wait immediate)
    float av[U][BA], bv[U][BB];
    auto fetch = [&](int u, int t) __attribute__((always_inline)) {
        const size_t rc = (size_t)min(k_begin + 2 * (g + RG * t) + lk, last);
        if (BA == 2) {
            const float2 v = *reinterpret_cast<const float2*>(Ap + rc * p.lda);
            av[u][0] = v.x; av[u][BA - 1] = v.y;
        } else {
            av[u][0] = Ap[rc * p.lda];
        }
        if (BB == 2) {
            const float2 v = *reinterpret_cast<const float2*>(Bp + rc * p.ldb);
            bv[u][0] = v.x; bv[u][BB - 1] = v.y;
        } else {
            bv[u][0] = Bp[rc * p.ldb];
        }
    };
    f32x16 acc[BA][BB];
    float bs[BA];
#pragma unroll
    for (int a = 0; a < BA; ++a) {
        bs[a] = 0.f;
#pragma unroll
        for (int b = 0; b < BB; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) fetch(u, u);
    for (int t0 = 0; t0 < nt; t0 += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            // branch-free: steps beyond the wave's last pair contract zeros (with a branch around the loads hipcc's s_waitcnt
            // placement degrades to vmcnt(0..1) and the U steps in flight are lost)
            float a_[BA], b_[BB];
            const bool ok = k_begin + 2 * (g + RG * (t0 + u)) + lk < k_end;          // (implies t0 + u < nt)
#pragma unroll
            for (int a = 0; a < BA; ++a) a_[a] = ok ? av[u][a] : 0.f;
#pragma unroll
            for (int b = 0; b < BB; ++b) b_[b] = ok ? bv[u][b] : 0.f;
            fetch(u, t0 + u + U);                // refill the slot (beyond the split: a clamped load, never used)
            __builtin_amdgcn_sched_barrier(0);   // the refill stays HERE, U steps ahead of its use (hipcc sinks it next to the use otherwise)
#pragma unroll
            for (int a = 0; a < BA; ++a) {
                bs[a] += a_[a];
#pragma unroll
                for (int b = 0; b < BB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_[a], b_[b], acc[a][b], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int a = 0; a < BA; ++a) bs[a] += __shfl_xor(bs[a], 32);        // both row parities

    if (RG > 1) {               // add the row groups' partial tiles in group order
        if (g > 0) {
            float (*r)[64] = red[(g - 1) * W + wsub];
#pragma unroll
            for (int a = 0; a < BA; ++a) {
#pragma unroll
                for (int b = 0; b < BB; ++b)
#pragma unroll
                    for (int i = 0; i < 16; ++i) r[(a * BB + b) * 16 + i][lane] = acc[a][b][i];
                r[BA * BB * 16 + a][lane] = bs[a];
            }
        }
        __syncthreads();
        if (g > 0) return;
        for (int gg = 1; gg < RG; ++gg) {
            float (*r)[64] = red[(gg - 1) * W + wsub];
#pragma unroll
            for (int a = 0; a < BA; ++a) {
#pragma unroll
                for (int b = 0; b < BB; ++b)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[a][b][i] += r[(a * BB + b) * 16 + i][lane];
                bs[a] += r[BA * BB * 16 + a][lane];
            }
        }
    }

    const bool direct = p.splits == 1;
    float* out = p.out + (direct ? 0 : (size_t)blockIdx.y * p.M * p.ldo);
    const int col = b_base + BB * lr;
#pragma unroll
    for (int a = 0; a < BA; ++a) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = a_base + BA * ((i & 3) + 8 * (i >> 2) + 4 * lk) + a;
            if (row < p.M && col < p.N) {
                float* o = out + (size_t)row * p.ldo + col;
#pragma unroll
                for (int b = 0; b < BB; ++b) o[b] = (direct && p.accumulate) ? o[b] + acc[a][b][i] : acc[a][b][i];
            }
        }
        const int bc = a_base + BA * lr + a;
        if (do_bias && lk == 0 && bc < p.M) {
            float* bo = p.bias_out + (direct ? 0 : (size_t)blockIdx.y * p.M) + bc;
            *bo = (direct && p.accumulate) ? *bo + bs[a] : bs[a];
        }
    }
}

// Small-problem variant (FWD / DGRAD with a few hundred to a few thousand rows: the coarse backbone levels, the RoI heads, the
// set-abstraction levels whose distinct rows are a fraction of their capacity — 150 of the ~250 GEMMs of a joint-mode step): one
// WORKGROUP per 32 x 32 output tile, operands straight from L2 into the MFMA registers (no operand tile in LDS, no barrier in the
// contraction), the contraction split over the workgroup's four waves (contiguous quarters, SPF steps of 8 contraction elements in
// flight each), the four partial tiles added in wave order through LDS.  A 128 x 128-tile launch of such a problem is 8 - 32
// workgroups marching through the contraction at one global-load latency per k-tile (30 - 130 us measured); the first form of this
// kernel (one wave per tile, 4 steps in flight) still took 77 - 260 us on 1024 rows x 512 columns x 1024 contraction (tools/
// joint_timeline.py): a lone wave walks 128 steps at a quarter of a round trip each.  Lane (r = lane & 31, h = lane >> 5) holds 4
// consecutive k of its row per step; MFMA step q pairs k = 8 s + q (h = 0) with k = 8 s + 4 + q (h = 1) on both operands.  Same
// operand forms and epilogues as rows_gemm_kernel; with the row count in device memory the workgroups beyond it return at once.
constexpr int SPF = 8;

template <int MODE>
__global__ void __launch_bounds__(256)
rows_gemm_small_kernel(RGemm p) {
    __shared__ float red[3][16][64];
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
    const int Mv = dev_count(p.M, p.m_dev);
    if (m0 >= Mv) return;
    const int row = min(m0 + r, Mv - 1), col = min(n0 + r, p.N - 1);
    const int nk = (p.K + 7) / 8;
    const int per = (nk + 3) / 4, s_begin = wave * per, s_end = min(nk, s_begin + per);
    // raw loads from clamped addresses; a step's values are zeroed beyond the wave's quarter when it is CONSUMED (`live`)
    auto loadA = [&](int s) {
        const int kc = min(8 * s + 4 * h, p.K - 4);
        if (MODE == RM_FWD && p.A2 != nullptr && kc >= p.K1) return *reinterpret_cast<const float4*>(p.A2 + (size_t)row * p.lda2 + (kc - p.K1));
        return *reinterpret_cast<const float4*>(p.A + (size_t)row * p.lda + kc);
    };
    auto loadB = [&](int s) {
        const int kc = min(8 * s + 4 * h, p.K - 4);
        if (MODE == RM_FWD) return *reinterpret_cast<const float4*>(p.B + (size_t)col * p.ldb + kc);
        const float* q = p.B + (size_t)kc * p.ldb + col;       // 4 contraction rows of this lane's column
        return make_float4(q[0], q[p.ldb], q[2 * (size_t)p.ldb], q[3 * (size_t)p.ldb]);
    };
    auto live = [&](int s, const float4& v) { return (s < s_end && 8 * s + 4 * h < p.K) ? v : make_float4(0.f, 0.f, 0.f, 0.f); };
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float4 ra[SPF], rb[SPF];
#pragma unroll
    for (int s = 0; s < SPF; ++s) { ra[s] = loadA(s_begin + s); rb[s] = loadB(s_begin + s); }
    for (int s0 = s_begin; s0 < s_end; s0 += SPF) {
#pragma unroll
        for (int s = 0; s < SPF; ++s) {
            // branch-free: steps beyond the quarter load from a clamped address and contract zeros (a branch around the loads
            // costs the steps in flight: hipcc then waits with vmcnt(0..1))
            const float4 a = live(s0 + s, ra[s]), b = rb[s];
            ra[s] = loadA(s0 + s + SPF); rb[s] = loadB(s0 + s + SPF);
            __builtin_amdgcn_sched_barrier(0);       // the refill stays SPF steps ahead of its use
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) red[wave - 1][i][lane] = acc[i];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int w = 0; w < 3; ++w)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] += red[w][i][lane];
    const int c = n0 + r;
    const bool cok = c < p.N;
    const float bv = (MODE == RM_FWD && p.bias != nullptr && cok) ? p.bias[c] : 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int orow = m0 + (i & 3) + 8 * (i >> 2) + 4 * h;
        const bool ok = cok && orow < Mv;
        float v = acc[i];
        if (MODE == RM_FWD) {
            v += bv;
            if (p.act == 1) v = fmaxf(v, 0.f);
            else if (p.act == 2) v = tanhf(v);
            if (p.rowscale != nullptr && ok) v *= p.rowscale[orow];
        } else {
            if (p.mask != nullptr) v = (ok && p.mask[(size_t)orow * p.ldm + c] > 0.f) ? v : 0.f;
            if (p.accumulate && ok) v += p.out[(size_t)orow * p.ldo + c];
        }
        if (ok) p.out[(size_t)orow * p.ldo + c] = v;
    }
}

// which launch a FWD / DGRAD problem of m rows gets.  Row count on the host: few 128 x 128 tiles.  Row count in device memory (m =
// the capacity): the distinct rows of a set-abstraction level are a fraction of it (7 - 30 % on the benchmark clouds), so a capacity of
// up to 32768 rows takes the small tiles as well (beyond that the persistent tiles win: 30000 of 131072 rows, 64 -> 128: 20 vs 58 us) — the workgroups beyond the count return at once
static bool small_problem(int m, const int* m_dev, int n) {
    const long long wgs = (long long)divup(m, 32) * divup(n, 32);
    if (m_dev != nullptr) return m <= 32768 && wgs <= 16384;
    return (long long)divup(m, RBM) * divup(n, RBN) < 96 && wgs <= 16384;
}

static int persistent_grid(long long tiles) { return (int)(tiles < 1 ? 1 : (tiles > JM_PGRID ? JM_PGRID : tiles)); }

}  // namespace jm

using namespace jm;

extern "C" {

int jm_rows_linear_forward(int m, const int* m_dev, int k1, int k2, int n, const float* x1, int ldx1, const float* x2, int ldx2,
                           const float* w, int ldw, const float* bias, int act, const float* rowscale, float* y, int ldy,
                           jm_stream_t stream) {
    const int k = k1 + k2;
    JM_REQUIRE(m >= 0 && n > 0 && k1 > 0 && k2 >= 0 && x1 && w && y, "rows_linear_forward: bad arguments");
    JM_REQUIRE(k1 % 4 == 0 && k2 % 4 == 0 && ldx1 % 4 == 0 && ldw % 4 == 0 && (k2 == 0 || (x2 && ldx2 % 4 == 0)) && ldx1 >= k1 && ldw >= k,
               "rows_linear_forward: widths and leading dimensions must be multiples of 4 (k1 %d, k2 %d, ldx1 %d, ldx2 %d, ldw %d)", k1, k2, ldx1, ldx2, ldw);
    JM_REQUIRE(act >= 0 && act <= 2 && ldy >= n, "rows_linear_forward: act %d, ldy %d < n %d", act, ldy, n);
    if (m == 0) return JM_OK;
    RGemm p{};
    p.M = m; p.N = n; p.K = k; p.m_dev = m_dev; p.A = x1; p.lda = ldx1; p.A2 = k2 ? x2 : nullptr; p.lda2 = ldx2; p.K1 = k1;
    p.B = w; p.ldb = ldw; p.bias = bias; p.act = act; p.rowscale = rowscale; p.out = y; p.ldo = ldy; p.splits = 1;
    const long long tiles = (long long)divup(m, RBM) * divup(n, RBN);
    if (small_problem(m, m_dev, n))
        hipLaunchKernelGGL((rows_gemm_small_kernel<RM_FWD>), dim3((unsigned)divup(n, 32), (unsigned)divup(m, 32)), dim3(256), 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL((rows_gemm_kernel<RM_FWD>), dim3((unsigned)persistent_grid(tiles)), dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("rows_linear_forward");
}

int jm_rows_linear_dgrad(int m, const int* m_dev, int n, int k, const float* dy, int lddy, const float* w, int ldw,
                         const float* mask, int ldm, int accumulate, float* dx, int lddx, jm_stream_t stream) {
    // dx (m, k) = (dy (m, n) w (n, k)) .* (mask > 0): the GEMM's contraction is the layer's OUTPUT width n
    JM_REQUIRE(m >= 0 && n > 0 && k > 0 && dy && w && dx, "rows_linear_dgrad: bad arguments");
    JM_REQUIRE(n % 4 == 0 && k % 4 == 0 && lddy % 4 == 0 && ldw % 4 == 0 && lddy >= n && ldw >= k && lddx >= k && (!mask || ldm >= k),
               "rows_linear_dgrad: widths and leading dimensions must be multiples of 4 (n %d, k %d, lddy %d, ldw %d)", n, k, lddy, ldw);
    if (m == 0) return JM_OK;
    RGemm p{};
    p.M = m; p.N = k; p.K = n; p.m_dev = m_dev; p.A = dy; p.lda = lddy; p.B = w; p.ldb = ldw; p.mask = mask; p.ldm = ldm;
    p.accumulate = accumulate; p.out = dx; p.ldo = lddx; p.splits = 1;
    const long long tiles = (long long)divup(m, RBM) * divup(k, RBN);
    if (small_problem(m, m_dev, k))
        hipLaunchKernelGGL((rows_gemm_small_kernel<RM_DGRAD>), dim3((unsigned)divup(k, 32), (unsigned)divup(m, 32)), dim3(256), 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL((rows_gemm_kernel<RM_DGRAD>), dim3((unsigned)persistent_grid(tiles)), dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("rows_linear_dgrad");
}

#ifdef JM_TOOLS_BUILD   // (tools/csrc/jmodt_hip_tools.h: 5.7 ms against MIOpen's 3.5 ms per 4 frames, not in the product ABI)
// ---- kernel == stride transposed convolution (backbone.py:150-157 DeConv) as a GEMM with a pixel-shuffled output: x (m = B h w, c) rows of
// the channels-last input map, wt (k k r, c) with wt[(dy k + dx) r + rr][ci] = W[ci][rr][dy][dx]; y = the channels-last (B, h k, w k, ctot)
// map, this level in channels coff .. coff + r
static int deconv_check(int m, int c, int k, int r, int h, int w, int ctot, int coff) {
    JM_REQUIRE(m >= 0 && c >= 4 && c % 4 == 0 && k >= 1 && r >= 4 && r % 4 == 0 && h >= 1 && w >= 1 && m % (h * w) == 0 && ctot % 4 == 0 &&
                   coff % 4 == 0 && coff >= 0 && coff + r <= ctot,
               "rows_deconv: channels and the output slice in multiples of 4, m = B h w (m %d, c %d, k %d, r %d, h %d, w %d, ctot %d, coff %d)", m,
               c, k, r, h, w, ctot, coff);
    return JM_OK;
}

int jm_rows_deconv_forward(int m, int c, int k, int r, int h, int w, const float* x, int ldx, const float* wt, float* y, int ctot, int coff,
                           jm_stream_t stream) {
    if (int e = deconv_check(m, c, k, r, h, w, ctot, coff)) return e;
    JM_REQUIRE(x && wt && y && ldx >= c && ldx % 4 == 0, "rows_deconv_forward: bad arguments");
    if (m == 0) return JM_OK;
    RGemm p{};
    p.sh = RShuffle{1, k, r, h, w, ctot, coff};
    p.M = m; p.N = k * k * r; p.K = c; p.A = x; p.lda = ldx; p.K1 = c; p.B = wt; p.ldb = c; p.out = y; p.ldo = p.N; p.splits = 1;
    const long long tiles = (long long)divup(m, RBM) * divup(p.N, RBN);
    hipLaunchKernelGGL((rows_gemm_kernel<RM_FWD>), dim3((unsigned)persistent_grid(tiles)), dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("rows_deconv_forward");
}

int jm_rows_deconv_dgrad(int m, int c, int k, int r, int h, int w, const float* dy, int ctot, int coff, const float* wt, float* dx, int lddx,
                         jm_stream_t stream) {
    if (int e = deconv_check(m, c, k, r, h, w, ctot, coff)) return e;
    JM_REQUIRE(dy && wt && dx && lddx >= c, "rows_deconv_dgrad: bad arguments");
    if (m == 0) return JM_OK;
    RGemm p{};
    p.sh = RShuffle{1, k, r, h, w, ctot, coff};
    p.M = m; p.N = c; p.K = k * k * r; p.A = dy; p.lda = p.K; p.B = wt; p.ldb = c; p.out = dx; p.ldo = lddx; p.splits = 1;
    const long long tiles = (long long)divup(m, RBM) * divup(c, RBN);
    hipLaunchKernelGGL((rows_gemm_kernel<RM_DGRAD>), dim3((unsigned)persistent_grid(tiles)), dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("rows_deconv_dgrad");
}

int jm_rows_deconv_wgrad(int m, int c, int k, int r, int h, int w, const float* dy, int ctot, int coff, const float* x, int ldx, float* dwt,
                         void* ws, size_t ws_bytes, jm_stream_t stream) {
    if (int e = deconv_check(m, c, k, r, h, w, ctot, coff)) return e;
    JM_REQUIRE(dy && x && dwt && ldx >= c && ldx % 4 == 0, "rows_deconv_wgrad: bad arguments");
    const int n = k * k * r;
    const int splits = jm_rows_wgrad_splits(m, n, c);
    if (splits > 1 && (ws_bytes < jm_rows_wgrad_workspace_bytes(m, n, c) || !ws)) {
        set_error("rows_deconv_wgrad: workspace of %zu bytes, need %zu", ws_bytes, jm_rows_wgrad_workspace_bytes(m, n, c));
        return JM_EWORKSPACE;
    }
    JM_REQUIRE(m > 0, "rows_deconv_wgrad: no rows");
    RGemm p{};
    p.sh = RShuffle{1, k, r, h, w, ctot, coff};
    p.M = n; p.N = c; p.K = m; p.A = dy; p.lda = n; p.B = x; p.ldb = ldx; p.splits = splits;
    if (splits > 1) { p.out = (float*)ws; p.ldo = c; } else { p.out = dwt; p.ldo = c; }
    hipLaunchKernelGGL((rows_gemm_kernel<RM_WGRAD>), dim3((unsigned)(divup(n, RBM) * divup(c, RBN)), (unsigned)splits), dim3(256), 0,
                       (hipStream_t)stream, p);
    if (splits > 1) {
        const int work = n * (c / 4);
        hipLaunchKernelGGL(rows_wgrad_reduce_kernel, dim3((unsigned)divup(work, 32)), dim3(256), 0, (hipStream_t)stream, n, c, splits, m,
                           (const int*)nullptr, (const float*)ws, dwt, c, (const float*)nullptr, (float*)nullptr, 0);
    }
    return check_launch("rows_deconv_wgrad");
}

#endif  // JM_TOOLS_BUILD

int jm_rows_wgrad_splits(int m, int n, int k) {
    // enough (tile, split) workgroups to fill 256 CUs four times over with at least 256 contraction rows per split (a workgroup of the
    // direct kernel has up to 128 rows in flight); a short contraction runs un-split and writes dW / dbias straight from the
    // accumulators (one launch).  With the row count in device memory only the splits that hold >= 128 rows exist (wgrad_live_splits)
    const int tiles = divup(n, RBM) * divup(k, RBN);
    int s = divup(1024, tiles);
    const int cap = imax(1, m / 256);
    if (s > cap) s = cap;
    if (s > 256) s = 256;
    return imax(1, s);
}

size_t jm_rows_reduce_workspace_bytes(int n) { return (size_t)JM_ROWS_CHUNKS * (size_t)n * sizeof(float); }

size_t jm_rows_wgrad_workspace_bytes(int m, int n, int k) {
    // the split partials of dW, then the row-chunk partials of the bias gradient
    const size_t s = (size_t)jm_rows_wgrad_splits(m, n, k);
    return s > 1 ? s * ((size_t)n * (size_t)k + (size_t)n) * sizeof(float) : 0;
}

int jm_rows_colsum(int m, const int* m_dev, int n, const float* x, int ldx, float* out, int accumulate, void* ws, size_t ws_bytes,
                   jm_stream_t stream) {
    JM_REQUIRE(m >= 0 && n > 0 && x && out && ldx >= n, "rows_colsum: bad arguments");
    if (!ws || ws_bytes < jm_rows_reduce_workspace_bytes(n)) {
        set_error("rows_colsum: workspace of %zu bytes, need %zu", ws_bytes, jm_rows_reduce_workspace_bytes(n));
        return JM_EWORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    const int chunks = rows_chunks(m);
    hipLaunchKernelGGL(rows_colsum_part_kernel, dim3((unsigned)divup(n, 64), (unsigned)chunks), dim3(256), 0, s, m, m_dev, n, chunks, x, ldx,
                       (float*)ws);
    hipLaunchKernelGGL(rows_sum_partials_kernel, dim3((unsigned)divup(n, 256)), dim3(256), 0, s, n, chunks, (const float*)ws, out, accumulate);
    return check_launch("rows_colsum");
}


int jm_rows_linear_wgrad(int m, const int* m_dev, int n, int k, const float* dy, int lddy, const float* x, int ldx,
                         float* dw, int lddw, float* dbias, int accumulate, void* ws, size_t ws_bytes, jm_stream_t stream) {
    // dw (n, k) (+)= dy (m, n)^T x (m, k); dbias (n) (+)= column sums of dy (NULL: skipped)
    JM_REQUIRE(m >= 0 && n > 0 && k > 0 && dy && x && dw, "rows_linear_wgrad: bad arguments");
    JM_REQUIRE(n % 4 == 0 && k % 4 == 0 && lddy % 4 == 0 && ldx % 4 == 0 && lddy >= n && ldx >= k && lddw >= k,
               "rows_linear_wgrad: widths and leading dimensions must be multiples of 4 (n %d, k %d, lddy %d, ldx %d)", n, k, lddy, ldx);
    const int splits = jm_rows_wgrad_splits(m, n, k);
    if (splits > 1 && (ws_bytes < jm_rows_wgrad_workspace_bytes(m, n, k) || !ws)) {
        set_error("rows_linear_wgrad: workspace of %zu bytes, need %zu", ws_bytes, jm_rows_wgrad_workspace_bytes(m, n, k));
        return JM_EWORKSPACE;
    }
    if (m == 0) {           // no rows: the gradients are zero
        if (!accumulate) {
            for (int r = 0; r < n; ++r) (void)jm_zero_async(dw + (size_t)r * lddw, (size_t)k * sizeof(float), (hipStream_t)stream);
            if (dbias) (void)jm_zero_async(dbias, (size_t)n * sizeof(float), (hipStream_t)stream);
        }
        return check_launch("rows_linear_wgrad");
    }
    RGemm p{};
    p.M = n; p.N = k; p.K = m; p.m_dev = m_dev; p.A = dy; p.lda = lddy; p.B = x; p.ldb = ldx; p.splits = splits; p.accumulate = accumulate;
    float* bias_part = (float*)ws + (size_t)splits * n * k;
    if (splits > 1) { p.out = (float*)ws; p.ldo = k; p.bias_out = dbias ? bias_part : nullptr; }
    else { p.out = dw; p.ldo = lddw; p.bias_out = dbias; }
    // the tile's patches: one wave takes 32 or 64 columns of dY (BA) and of X (BB); up to 2 x 2 waves per 128 x 128 tile
    const int ba = n > 32 ? 2 : 1, bb = k > 32 ? 2 : 1;
    const int wa = n > 64 ? 2 : 1, wb = k > 64 ? 2 : 1;
    const dim3 grid((unsigned)(divup(n, RBM) * divup(k, RBN)), (unsigned)splits);
    if (ba == 1 && bb == 1) hipLaunchKernelGGL((rows_wgrad_direct_kernel<1, 1>), grid, dim3(256), 0, (hipStream_t)stream, p, wa, wb);
    else if (ba == 1) hipLaunchKernelGGL((rows_wgrad_direct_kernel<1, 2>), grid, dim3(256), 0, (hipStream_t)stream, p, wa, wb);
    else if (bb == 1) hipLaunchKernelGGL((rows_wgrad_direct_kernel<2, 1>), grid, dim3(256), 0, (hipStream_t)stream, p, wa, wb);
    else hipLaunchKernelGGL((rows_wgrad_direct_kernel<2, 2>), grid, dim3(256), 0, (hipStream_t)stream, p, wa, wb);
    if (splits > 1) {
        const int work = n * (k / 4) + (dbias ? n : 0);
        hipLaunchKernelGGL(rows_wgrad_reduce_kernel, dim3((unsigned)divup(work, 32)), dim3(256), 0, (hipStream_t)stream, n, k, splits, m, m_dev,
                           (const float*)ws, dw, lddw, (const float*)bias_part, dbias, accumulate);
    }
    return check_launch("rows_linear_wgrad");
}

}  // extern "C"
