// sa_mlp.hip — fused set-abstraction block on the fp32 matrix cores (gfx950):
//     group (xyz - centre | features)  ->  [1x1 conv + BN(eval, folded) + ReLU] x L  ->  max over nsample
//
// Replaces the per-scale body of _PointnetSAModuleBase.forward
// (jmodt/ops/pointnet2/pointnet2_modules.py:46-52): QueryAndGroup's two grouping_operation launches,
// the subtraction, the cat (pointnet2_utils.py:249-264), SharedMLP's cuDNN 1x1 convolutions on the
// materialised (B, 3+C, npoint, nsample) tensor (pytorch_utils.py:6-33) and F.max_pool2d.
// For the RCNN's first SA level that grouped tensor alone is 4.4 GB per 1024 RoIs;
// here neither it nor any hidden activation ever leaves the chip.
//
// One workgroup (4 waves, one per SIMD) = 128 consecutive (centre, sample) rows of one frame.
//   * Activations live in two 128(k) x 128(row) LDS buffers, k-major, used in ping-pong: a layer reads
//     its A operand straight from one and its epilogue (bias + ReLU) writes the other with 16-byte
//     stores in the MFMA accumulator's own row order.  Layer 1's input is gathered through idx into a
//     buffer in chunks of 128 channels (first 3 channels = xyz - centre).
//   * Weights never touch LDS: they are pre-packed (jm_sa_mlp_pack) so that the B operand of lane
//     (col, khalf) for one 16-deep k-tile is 8 consecutive floats -> two global_load_dwordx4 per
//     32-column block per k-tile, coalesced over the wave, served by L1/L2 (a layer is <= 64 KB and
//     every workgroup reads the same one), prefetched one k-tile ahead in registers.
//   * Hence the k-loops contain NO barrier: one __syncthreads() per layer.
//   * Last layer: any width (128-column tiles); its epilogue reduces max over each centre's nsample
//     rows straight out of the accumulator layout, then bias + ReLU (both commute with max).
// v_mfma_f32_32x32x2_f32 everywhere: exact-f32 products (1e-4 parity with the fp32 reference path).
// Constraints (the Python module falls back to the unfused path otherwise): nsample in {16,32,64},
// npoint*nsample % 128 == 0, hidden widths <= 128, eval-mode BatchNorm (folded by the caller).
#include "jm_common.h"

namespace jm {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int SM_BM = 128, SM_KC = 128, SM_LDP = SM_BM + 4;
constexpr int SM_BUF = SM_KC * SM_LDP;                 // floats per activation buffer
constexpr size_t SM_LDS_BYTES = 2 * (size_t)SM_BUF * sizeof(float);

__host__ __device__ inline int pad_to(int v, int m) { return (v + m - 1) / m * m; }

struct SaMlpParams {
    int N, M, C, ns;                 // points per frame, centres per frame, feature channels, nsample
    const float* xyz;                // (B,N,3)
    const float* new_xyz;            // (B,M,3)
    const float* feat;               // (B,C,N) or null
    const int* idx;                  // (B,M,ns)
    int L;                           // layers (1..4)
    int kp[5];                       // kp[l] = pad16(width_l), l = 0..L
    int np[4];                       // np[l] = pad128(width_{l+1}): packed rows of layer l
    const float* W[4];               // packed weights of layer l (see jm_sa_mlp_pack)
    const float* bias[4];            // (np[l]) zero padded
    float* out;                      // (B, cout, M)
    int cout;
};

// packed layout: Wp[kt][n][khalf][kk] = W[n][16 kt + 2 kk + khalf]   (kt < Kp/16, n < Np, khalf < 2, kk < 8)
__global__ void sa_mlp_pack_kernel(int cout, int cin, int Kp, int Np, const float* __restrict__ w,
                                   const float* __restrict__ b, float* __restrict__ wp, float* __restrict__ bp) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < Np) bp[e] = (e < cout && b) ? b[e] : 0.f;
    if (e >= Kp * Np) return;
    const int kk = e & 7, kh = (e >> 3) & 1, n = (e >> 4) % Np, kt = (e >> 4) / Np;
    const int k = 16 * kt + 2 * kk + kh;
    wp[e] = (n < cout && k < cin) ? w[(size_t)n * cin + k] : 0.f;
}

// acc += A(128 rows x 16 nkt, LDS k-major) x W-tile; this wave owns rows wm*64.., columns ncol0.. (+32 if two)
//   A : LDS buffer [k][SM_LDP], rows of this tile
//   bp: packed weights of this lane for k-tile 0 of the range, column block 0; kt_stride floats per k-tile
template <bool TWO>
__device__ __forceinline__ void mfma_ktiles(const float* __restrict__ A, int nkt, const float* __restrict__ bp,
                                            size_t kt_stride, int a_off, f32x16 (&acc)[2][2]) {
    float4 bc[4], bn[4];
    float ac[16], an[16];
    auto loadB = [&](float4 (&b)[4], int kt) {
        const float* q = bp + (size_t)kt * kt_stride;
        b[0] = *reinterpret_cast<const float4*>(q);
        b[1] = *reinterpret_cast<const float4*>(q + 4);
        if (TWO) {
            b[2] = *reinterpret_cast<const float4*>(q + 512);       // column + 32: (32 * 2) * 8 floats on
            b[3] = *reinterpret_cast<const float4*>(q + 516);
        }
    };
    auto loadA = [&](float (&a)[16], int kt) {
        const float* q = A + (size_t)kt * 16 * SM_LDP + a_off;       // a_off = khalf * SM_LDP + wm * 64 + lr
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            a[2 * kk] = q[(2 * kk) * SM_LDP];
            a[2 * kk + 1] = q[(2 * kk) * SM_LDP + 32];
        }
    };
    auto mm = [&](const float (&a)[16], const float4 (&b)[4]) {
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const float b0 = reinterpret_cast<const float*>(&b[0])[kk];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * kk], b0, acc[0][0], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * kk + 1], b0, acc[1][0], 0, 0, 0);
            if (TWO) {
                const float b1 = reinterpret_cast<const float*>(&b[2])[kk];
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * kk], b1, acc[0][1], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * kk + 1], b1, acc[1][1], 0, 0, 0);
            }
        }
    };
    // sched_barrier(0): keep the prefetches where they are written — left alone, the scheduler sinks
    // them to their first use and the L2 latency lands on the MFMA pipe
    loadB(bc, 0);
    loadA(ac, 0);
    int kt = 0;
    for (; kt + 2 <= nkt; kt += 2) {
        loadB(bn, kt + 1);
        loadA(an, kt + 1);
        __builtin_amdgcn_sched_barrier(0);
        mm(ac, bc);
        __builtin_amdgcn_sched_barrier(0);
        const int nx = min(kt + 2, nkt - 1);    // clamped: unconditional loads, no branchy waits
        loadB(bc, nx);
        loadA(ac, nx);
        __builtin_amdgcn_sched_barrier(0);
        mm(an, bn);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (kt < nkt) mm(ac, bc);
}

__global__ void __launch_bounds__(256)
sa_mlp_kernel(SaMlpParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, lk = lane >> 5;
    const int bi = blockIdx.y;
    const int row0 = blockIdx.x * SM_BM;            // first (centre, sample) row of this tile

    // gather identity: this thread fills row `grow`, channels gk0, gk0 + 2, ... of every chunk
    const int grow = tid & 127, gk0 = tid >> 7;
    const int gidx = p.idx[(size_t)bi * p.M * p.ns + row0 + grow];
    float cen[3];
    {
        const int m = (row0 + grow) / p.ns;
#pragma unroll
        for (int q = 0; q < 3; ++q) cen[q] = p.new_xyz[((size_t)bi * p.M + m) * 3 + q];
    }
    const float* xyz_b = p.xyz + (size_t)bi * p.N * 3;
    const float* feat_b = p.feat ? p.feat + (size_t)bi * p.C * p.N : xyz_b;   // never dereferenced when C == 0
    const int c_in = 3 + p.C;
    const int a_off = lk * SM_LDP + wm * 64 + lr;

    f32x16 acc[2][2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    };
    // lane's packed-weight pointer: layer l, k-tile kt0, column n
    auto wptr = [&](int l, int kt0, int n) {
        return p.W[l] + ((size_t)kt0 * p.np[l] + n) * 16 + lk * 8;
    };
    auto run = [&](const float* A, int nkt, int l, int kt0, int ncol0, int ncols) {
        // ncols: valid (padded-to-16) columns from ncol0 on; a wave without columns idles
        if (ncols <= 0) return;
        const float* bp = wptr(l, kt0, ncol0 + lr);
        const size_t st = (size_t)p.np[l] * 16;
        if (ncols > 32) mfma_ktiles<true>(A, nkt, bp, st, a_off, acc);
        else mfma_ktiles<false>(A, nkt, bp, st, a_off, acc);
    };

    int cur = 0;   // buffer the next consumer reads
    // ---------------- layer 1: gather chunks of <= 128 input channels, accumulate over chunks
    const bool single = (p.L == 1);
    const int K0 = p.kp[0];
    const int nchunks = (K0 + SM_KC - 1) / SM_KC;
    if (!single) zero_acc();
    for (int c = 0; c < nchunks; ++c) {
        const int kc = min(SM_KC, K0 - c * SM_KC);     // multiple of 16
        float* G = lds + (c & 1) * SM_BUF;
        if (c >= 2) __syncthreads();                   // the chunk two back has been consumed by every wave
        auto src = [&](int k) {                        // unconditional load on a clamped address
            const int kcl = min(k, c_in - 1);
            return kcl < 3 ? xyz_b + (size_t)gidx * 3 + kcl : feat_b + (size_t)(kcl - 3) * p.N + gidx;
        };
        auto fix = [&](float v, int k) {               // centre subtraction / zero padding
            if (k < 3) v = v - cen[k];
            return k >= c_in ? 0.f : v;
        };
        if (kc == SM_KC) {                             // full chunk: all 64 loads of this thread in flight
            float g[64];
#pragma unroll
            for (int i = 0; i < 64; ++i) g[i] = *src(c * SM_KC + gk0 + 2 * i);
#pragma unroll
            for (int i = 0; i < 64; ++i) G[(gk0 + 2 * i) * SM_LDP + grow] = fix(g[i], c * SM_KC + gk0 + 2 * i);
        } else {                                       // tail chunk: 16 channels (8 loads) per trip
            for (int kb = 0; kb < kc; kb += 16) {
                float g[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) g[i] = *src(c * SM_KC + kb + gk0 + 2 * i);
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    G[(kb + gk0 + 2 * i) * SM_LDP + grow] = fix(g[i], c * SM_KC + kb + gk0 + 2 * i);
            }
        }
        __syncthreads();
        if (!single) run(G, kc / 16, 0, c * (SM_KC / 16), wn * 64, p.kp[1] - wn * 64);
        cur = c & 1;
    }

    auto hidden_epilogue = [&](int l, float* Y) {
        const float* bl = p.bias[l];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = wn * 64 + j * 32 + lr;
            if (wn * 64 + j * 32 >= p.kp[l + 1]) continue;     // wave-uniform: columns the next layer never reads
            const float bv = bl[col];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    // accumulator r = 4 rq + t  <->  row 8 rq + 4 lk + t of the 32-row block: 4 consecutive rows
                    float4 v;
                    v.x = fmaxf(acc[i][j][4 * rq + 0] + bv, 0.f); v.y = fmaxf(acc[i][j][4 * rq + 1] + bv, 0.f);
                    v.z = fmaxf(acc[i][j][4 * rq + 2] + bv, 0.f); v.w = fmaxf(acc[i][j][4 * rq + 3] + bv, 0.f);
                    *reinterpret_cast<float4*>(Y + (size_t)col * SM_LDP + wm * 64 + i * 32 + 8 * rq + 4 * lk) = v;
                }
        }
    };

    int l = 0;
    if (!single) {
        // layer 1 epilogue -> the buffer the last chunk did not use (its readers finished before the
        // barrier that preceded the last chunk's MFMAs)
        float* Y = lds + ((cur ^ 1)) * SM_BUF;
        hidden_epilogue(0, Y);
        __syncthreads();
        cur ^= 1;
        // ---------------- hidden layers 2 .. L-1
        for (l = 1; l < p.L - 1; ++l) {
            zero_acc();
            run(lds + cur * SM_BUF, p.kp[l] / 16, l, 0, wn * 64, p.kp[l + 1] - wn * 64);
            hidden_epilogue(l, lds + (cur ^ 1) * SM_BUF);
            __syncthreads();
            cur ^= 1;
        }
    }
    // ---------------- last layer: 128-column tiles, max-pool epilogue  (l == L - 1)
    {
        const float* A = lds + cur * SM_BUF;
        const int nkt = single ? K0 / 16 : p.kp[l] / 16;   // single: host guarantees one chunk
        const float* bl = p.bias[l];
        for (int n0 = 0; n0 < p.np[l]; n0 += 128) {
            zero_acc();
            run(A, nkt, l, 0, n0 + wn * 64, p.kp[l + 1] - (n0 + wn * 64));
            // max over each centre's nsample rows, straight from the accumulator layout
            // row(i, r, lk) = 32 i + (r & 3) + 8 (r >> 2) + 4 lk  within this wave's 64 rows
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = n0 + wn * 64 + j * 32 + lr;
                if (n0 + wn * 64 + j * 32 >= p.cout) continue;   // wave-uniform
                const float bv = bl[col];
                float v[4];   // up to 4 centres per wave (nsample 16)
                if (p.ns == 64) {
                    float t = -INFINITY;
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) t = fmaxf(t, acc[i][j][r]);
                    v[0] = t; v[1] = v[2] = v[3] = -INFINITY;
                } else if (p.ns == 32) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        float t = -INFINITY;
#pragma unroll
                        for (int r = 0; r < 16; ++r) t = fmaxf(t, acc[i][j][r]);
                        v[i] = t;
                    }
                    v[2] = v[3] = -INFINITY;
                } else {   // 16: rows 0-15 of a 32-block are r < 8, rows 16-31 are r >= 8
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        float t0 = -INFINITY, t1 = -INFINITY;
#pragma unroll
                        for (int r = 0; r < 8; ++r) { t0 = fmaxf(t0, acc[i][j][r]); t1 = fmaxf(t1, acc[i][j][r + 8]); }
                        v[2 * i] = t0; v[2 * i + 1] = t1;
                    }
                }
                const int per_wave = 64 / p.ns;   // centres per wave-row-block
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (c < per_wave) {
                        float t = fmaxf(v[c], __shfl_xor(v[c], 32));   // the other lane half holds the rows + 4
                        const int m = (row0 + wm * 64) / p.ns + c;
                        if (lk == 0 && col < p.cout)
                            p.out[((size_t)bi * p.cout + col) * p.M + m] = fmaxf(t + bv, 0.f);
                    }
                }
            }
        }
    }
}

}  // namespace jm

using namespace jm;

extern "C" size_t jm_sa_mlp_packed_weight_elems(int cout, int cin) {
    return cout < 1 || cin < 1 ? 0 : (size_t)pad_to(cin, 16) * pad_to(cout, 128);
}

extern "C" size_t jm_sa_mlp_packed_bias_elems(int cout) { return cout < 1 ? 0 : (size_t)pad_to(cout, 128); }

extern "C" int jm_sa_mlp_pack(int cout, int cin, const float* w, const float* b, float* wp, float* bp,
                              jm_stream_t stream) {
    JM_REQUIRE(cout >= 1 && cin >= 1, "sa_mlp_pack: bad sizes");
    JM_REQUIRE(w && wp && bp, "sa_mlp_pack: null pointer");
    const int Kp = pad_to(cin, 16), Np = pad_to(cout, 128);
    const long long total = (long long)Kp * Np;
    hipLaunchKernelGGL(sa_mlp_pack_kernel, dim3((unsigned)divup(total, 256)), dim3(256), 0, (hipStream_t)stream, cout, cin,
                       Kp, Np, w, b, wp, bp);
    return check_launch("sa_mlp_pack");
}

/* layer widths: widths[0] = 3 + C (input), widths[1..L] = layer outputs.
 * weights[l] / biases[l]: packed by jm_sa_mlp_pack(widths[l+1], widths[l], ...). */
extern "C" int jm_sa_mlp_forward(int b, int n, int m, int c, int nsample, const float* xyz, const float* new_xyz,
                                 const float* features, const int* idx, int num_layers, const int* widths,
                                 const float* const* weights, const float* const* biases, float* out,
                                 jm_stream_t stream) {
    JM_REQUIRE(b >= 0 && n >= 1 && m >= 0 && c >= 0, "sa_mlp: bad sizes");
    if (b == 0 || m == 0) return JM_OK;
    JM_REQUIRE(xyz && new_xyz && idx && out && widths && weights && biases && (features || c == 0), "sa_mlp: null pointer");
    JM_REQUIRE(nsample == 16 || nsample == 32 || nsample == 64, "sa_mlp: nsample %d not in {16,32,64}", nsample);
    JM_REQUIRE(((long long)m * nsample) % SM_BM == 0, "sa_mlp: npoint*nsample = %lld is not a multiple of 128", (long long)m * nsample);
    JM_REQUIRE(num_layers >= 1 && num_layers <= 4, "sa_mlp: %d layers unsupported", num_layers);
    JM_REQUIRE(widths[0] == 3 + c, "sa_mlp: widths[0] = %d != 3 + C = %d", widths[0], 3 + c);
    JM_REQUIRE(b <= 65535, "sa_mlp: batch too large");
    JM_REQUIRE(num_layers > 1 || widths[0] <= SM_KC, "sa_mlp: a single layer needs 3 + C <= 128");
    SaMlpParams p{};
    p.N = n; p.M = m; p.C = c; p.ns = nsample;
    p.xyz = xyz; p.new_xyz = new_xyz; p.feat = features; p.idx = idx;
    p.L = num_layers;
    for (int l = 0; l <= num_layers; ++l) {
        JM_REQUIRE(widths[l] >= 1, "sa_mlp: bad width");
        JM_REQUIRE(l == 0 || l == num_layers || widths[l] <= 128, "sa_mlp: hidden width %d > 128", widths[l]);
        p.kp[l] = pad_to(widths[l], 16);
    }
    for (int l = 0; l < num_layers; ++l) {
        JM_REQUIRE(weights[l] && biases[l], "sa_mlp: null layer %d", l);
        JM_REQUIRE((reinterpret_cast<uintptr_t>(weights[l]) & 15u) == 0, "sa_mlp: weights must be 16-byte aligned");
        p.W[l] = weights[l]; p.bias[l] = biases[l];
        p.np[l] = pad_to(widths[l + 1], 128);
    }
    p.out = out; p.cout = widths[num_layers];
    (void)hipFuncSetAttribute((const void*)sa_mlp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SM_LDS_BYTES);
    hipLaunchKernelGGL(sa_mlp_kernel, dim3((unsigned)((long long)m * nsample / SM_BM), b), dim3(256), SM_LDS_BYTES,
                       (hipStream_t)stream, p);
    return check_launch("sa_mlp");
}
