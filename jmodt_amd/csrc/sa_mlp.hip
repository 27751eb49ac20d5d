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
// One workgroup = 128 consecutive (centre, sample) rows of one frame:
//   layer 1   A operand gathered on the fly through idx (first 3 channels = xyz - centre, then the
//             point features), W1 streamed through LDS in 16-deep k-tiles;
//   layer l   the previous activation tile (128 rows x <=128 channels) stays in LDS, k-major, and
//             is read directly as the MFMA A operand; only W_l is streamed;
//   last      any width (128-column tiles); its epilogue reduces max over each centre's nsample rows
//             straight out of the MFMA accumulator layout, then bias + ReLU (both commute with max).
// v_mfma_f32_32x32x2_f32 everywhere: exact-f32 products (1e-4 parity with the fp32 reference path).
// Constraints (the Python module falls back to the unfused path otherwise): nsample in {16,32,64},
// npoint*nsample % 128 == 0, hidden widths <= 128, eval-mode BatchNorm (folded by the caller).
#include "jm_common.h"

namespace jm {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int SM_BM = 128, SM_BK = 16, SM_LDP = SM_BM + 4;

struct SaMlpParams {
    int N, M, C, ns;                 // points per frame, centres per frame, feature channels, nsample
    const float* xyz;                // (B,N,3)
    const float* new_xyz;            // (B,M,3)
    const float* feat;               // (B,C,N) or null
    const int* idx;                  // (B,M,ns)
    int L;                           // layers (1..4)
    int kp[5];                       // padded widths: kp[0] = pad16(3+C); kp[l] = pad16(width_l) (<=128 for l<L); kp[L] = pad128(out)
    const float* W[4];               // W[l]: (kp[l+1], kp[l]) row-major, zero padded, BN folded
    const float* bias[4];            // (kp[l+1]) zero padded
    float* out;                      // (B, cout, M)
    int cout;
};

__global__ void __launch_bounds__(256)
sa_mlp_kernel(SaMlpParams p) {
    __shared__ __attribute__((aligned(16))) float As[2][SM_BK][SM_LDP];
    __shared__ __attribute__((aligned(16))) float Bs[2][SM_BK][SM_LDP];
    __shared__ __attribute__((aligned(16))) float Hs[SM_BM][SM_LDP];   // [channel k][row]: next layer's A operand
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, lk = lane >> 5;
    const int bi = blockIdx.y;
    const int row0 = blockIdx.x * SM_BM;            // first (centre, sample) row of this tile

    // staging assignment (as in affinity.hip): 2 x 4 consecutive k of one row per thread per k-tile
    int srow[2], skq[2], gidx[2];
    float cen[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int f = tid + 256 * i;
        srow[i] = f >> 2;
        skq[i] = (f & 3) * 4;
        const int row = row0 + srow[i];
        const int m = row / p.ns;
        gidx[i] = p.idx[(size_t)bi * p.M * p.ns + row];
#pragma unroll
        for (int q = 0; q < 3; ++q) cen[i][q] = p.new_xyz[((size_t)bi * p.M + m) * 3 + q];
    }
    const float* xyz_b = p.xyz + (size_t)bi * p.N * 3;
    const float* feat_b = p.feat ? p.feat + (size_t)bi * p.C * p.N : xyz_b;   // never dereferenced when C == 0
    const int c_in = 3 + p.C;

    f32x16 acc[2][2];
    float ra[2][4];
    float4 rb[2];

    // gathered A element of layer 1 (one unconditional load on a selected address)
    auto gather_load = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = k0 + skq[i] + q;
                const int kc = min(k, c_in - 1);
                const float* a = kc < 3 ? xyz_b + (size_t)gidx[i] * 3 + kc : feat_b + (size_t)(kc - 3) * p.N + gidx[i];
                ra[i][q] = *a;
            }
    };
    auto gather_fix = [&](int k0) {   // centre subtraction / zero padding, applied at store time
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = k0 + skq[i] + q;
                float v = ra[i][q];
                if (k < 3) v = v - cen[i][k];
                if (k >= c_in) v = 0.f;
                ra[i][q] = v;
            }
    };

    for (int l = 0; l < p.L; ++l) {
        const int K = p.kp[l];
        const int Nl = p.kp[l + 1];
        const bool last = (l == p.L - 1);
        const float* Wl = p.W[l];
        const int nkt = K / SM_BK;
        const int ntiles = last ? Nl / 128 : 1;    // hidden layers: single (<=128 wide) tile
        for (int nt = 0; nt < ntiles; ++nt) {
            const int n0 = nt * 128;
            const float* b_ptr[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) b_ptr[i] = Wl + (size_t)min(n0 + srow[i], Nl - 1) * K + skq[i];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

            auto g_load = [&](int kt) {
                const int k0 = kt * SM_BK;
                if (l == 0) gather_load(k0);
#pragma unroll
                for (int i = 0; i < 2; ++i) rb[i] = *reinterpret_cast<const float4*>(b_ptr[i] + k0);
            };
            auto s_store = [&](int kt, int buf) {
                if (l == 0) {
                    gather_fix(kt * SM_BK);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int q = 0; q < 4; ++q) As[buf][skq[i] + q][srow[i]] = ra[i][q];
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const bool ok = n0 + srow[i] < Nl;
                    Bs[buf][skq[i] + 0][srow[i]] = ok ? rb[i].x : 0.f; Bs[buf][skq[i] + 1][srow[i]] = ok ? rb[i].y : 0.f;
                    Bs[buf][skq[i] + 2][srow[i]] = ok ? rb[i].z : 0.f; Bs[buf][skq[i] + 3][srow[i]] = ok ? rb[i].w : 0.f;
                }
            };

            g_load(0);
            s_store(0, 0);
            __syncthreads();
            for (int kt = 0; kt < nkt; ++kt) {
                const int buf = kt & 1;
                g_load(min(kt + 1, nkt - 1));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kk = 0; kk < SM_BK / 2; ++kk) {
                    const int k2 = kk * 2 + lk;
                    float a0, a1;
                    if (l == 0) {
                        a0 = As[buf][k2][wm * 64 + lr]; a1 = As[buf][k2][wm * 64 + 32 + lr];
                    } else {
                        a0 = Hs[kt * SM_BK + k2][wm * 64 + lr]; a1 = Hs[kt * SM_BK + k2][wm * 64 + 32 + lr];
                    }
                    const float b0 = Bs[buf][k2][wn * 64 + lr], b1 = Bs[buf][k2][wn * 64 + 32 + lr];
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (kt + 1 < nkt) s_store(kt + 1, buf ^ 1);
                __syncthreads();
            }

            const float* bl = p.bias[l];
            if (!last) {
                // hidden activation -> Hs (k-major).  All waves are past the barrier that ended the
                // k-loop, i.e. nobody still reads the previous Hs.
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int col = wn * 64 + j * 32 + lr;
                        const float bv = bl[min(col, Nl - 1)];
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int row = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                            Hs[col][row] = col < Nl ? fmaxf(acc[i][j][r] + bv, 0.f) : 0.f;
                        }
                    }
                __syncthreads();
            } else {
                // max over each centre's nsample rows, straight from the accumulator layout
                // row(i, r, lk) = 32 i + (r & 3) + 8 (r >> 2) + 4 lk  within this wave's 64 rows
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int col = n0 + wn * 64 + j * 32 + lr;
                    const float bv = bl[min(col, Nl - 1)];
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
                __syncthreads();   // Bs / As are restaged for the next column tile
            }
        }
    }
}

}  // namespace jm

using namespace jm;

/* layer widths: widths[0] = 3 + C (input), widths[1..L] = layer outputs.
 * weights[l]: (pad16or128(widths[l+1]), pad16(widths[l])) zero padded, BN folded; bias likewise. */
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
    SaMlpParams p{};
    p.N = n; p.M = m; p.C = c; p.ns = nsample;
    p.xyz = xyz; p.new_xyz = new_xyz; p.feat = features; p.idx = idx;
    p.L = num_layers;
    for (int l = 0; l <= num_layers; ++l) {
        JM_REQUIRE(widths[l] >= 1, "sa_mlp: bad width");
        const bool lastw = (l == num_layers);
        JM_REQUIRE(l == 0 ? widths[l] <= 1024 + 3 : (lastw || widths[l] <= 128), "sa_mlp: hidden width %d > 128", widths[l]);
        p.kp[l] = lastw ? (widths[l] + 127) / 128 * 128 : (widths[l] + 15) / 16 * 16;
    }
    for (int l = 0; l < num_layers; ++l) {
        JM_REQUIRE(weights[l] && biases[l], "sa_mlp: null layer %d", l);
        JM_REQUIRE((reinterpret_cast<uintptr_t>(weights[l]) & 15u) == 0, "sa_mlp: weights must be 16-byte aligned");
        p.W[l] = weights[l]; p.bias[l] = biases[l];
    }
    p.out = out; p.cout = widths[num_layers];
    hipLaunchKernelGGL(sa_mlp_kernel, dim3((unsigned)((long long)m * nsample / SM_BM), b), dim3(256), 0,
                       (hipStream_t)stream, p);
    return check_launch("sa_mlp");
}
