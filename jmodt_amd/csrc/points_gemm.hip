// points_gemm.hip — dense layers on (B, C, n) per-point tensors with FEW points: the coarse end of the backbone (gfx950).
//
// Feature propagation level 4 (pointnet2_modules.py:139-153: cat[interpolated 1024, skip 512] -> 512 -> 512 on 8 x 256 points) and the
// LI-Fusion attention block of level 4 (backbone.py:35-81 on 8 x 64 points with 512 / 1024-wide operands) are a few hundred to two
// thousand rows against weight matrices of up to 1536 x 512: too few 32-point tiles for conv1d_stack.hip / li_fusion.hip (every tile
// streams the whole weight set; the attention tile does not fit the LDS at these widths), so until round 5 they went to rocBLAS —
// per layer a broadcast-bias GEMM (the Tensile rows of the profile, 260 us for the 1024-wide operand of FP4 alone) + element-wise passes.
// Here a layer is ONE launch: a workgroup of four waves per 32-point x 32-column output tile, each wave a quarter of the contraction
// (512 .. 2048 long here: split, the tile's latency chain is a quarter as long and the launch has four times the waves), both
// operands straight from L2 into the MFMA registers, eight steps of 8 contraction elements in flight, partial tiles summed through LDS
// in wave order — the small-problem form of rows_gemm.hip on the
// inference layout:
//   * A operand = the (B, C, n) tensor IN PLACE (k-major for a tile of consecutive points: 128-byte coalesced rows per channel),
//     optionally the concatenation of two tensors along C (never materialised);
//   * B operand = the layer's (N, K) weight row-major, float4 along K;
//   * epilogue: bias, none / ReLU / tanh / sigmoid, an optional per-point scale (the attention gate), output (B, N, n) or
//     point-major rows (B n, ld).
// v_mfma_f32_32x32x2_f32: exact-f32 products.
#include "jm_common.h"

namespace jm {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct PGemm {
    int n, N, K1, K2;
    const float *x1, *x2;          // (B, K1, n), (B, K2, n)
    const float* W; int ldw;       // (N, K1 + K2)
    const float* bias; int act;    // 0 none, 1 ReLU, 2 tanh, 3 sigmoid
    const float* rowscale; int rs_stride;     // out *= rowscale[(b n + p) * rs_stride]
    float* out; int out_rows, ldo; // (B, N, n), or rows (B n, ldo)
};

constexpr int PG_SPF = 8, PG_KS = 4;      // contraction steps in flight per wave; waves per tile (each takes a quarter of the contraction)

__global__ void __launch_bounds__(64 * PG_KS)
points_gemm_kernel(PGemm p) {
    __shared__ float red[PG_KS - 1][16][64];
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
    const int b = m0 / p.n, r0 = m0 - b * p.n;                 // n % 32 == 0: a tile never straddles two frames
    const int K = p.K1 + p.K2;
    const int col = min(n0 + r, p.N - 1);
    const int nk_all = (K + 7) / 8;
    const int per = (nk_all + PG_KS - 1) / PG_KS;
    const int s_lo = min(wave * per, nk_all), s_hi = min(s_lo + per, nk_all);      // this wave's steps of 8 contraction elements
    const int nk = s_hi - s_lo;
    const float* a1 = p.x1 + (size_t)b * p.K1 * p.n + r0 + r;
    const float* a2 = p.x2 ? p.x2 + (size_t)b * p.K2 * p.n + r0 + r : nullptr;
    auto loadA = [&](int s) {
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = 8 * s + 4 * h + q;
            const int kc = min(k, K - 1);
            const float x = kc < p.K1 ? a1[(size_t)kc * p.n] : a2[(size_t)(kc - p.K1) * p.n];
            v[q] = k < K ? x : 0.f;
        }
        return make_float4(v[0], v[1], v[2], v[3]);
    };
    auto loadB = [&](int s) {
        const int k = 8 * s + 4 * h;
        const float4 v = *reinterpret_cast<const float4*>(p.W + (size_t)col * p.ldw + min(k, K - 4));
        return k < K ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    if (nk > 0) {
        float4 ra[PG_SPF], rb[PG_SPF];
#pragma unroll
        for (int s = 0; s < PG_SPF; ++s) {
            const int ss = s_lo + min(s, nk - 1);
            ra[s] = loadA(ss); rb[s] = loadB(ss);
        }
        for (int s0 = 0; s0 < nk; s0 += PG_SPF) {
#pragma unroll
            for (int s = 0; s < PG_SPF; ++s) {
                if (s0 + s < nk) {           // wave-uniform
                    const float4 a = ra[s], w = rb[s];
                    const int nx = s_lo + min(s0 + s + PG_SPF, nk - 1);      // refill this slot (clamped: unconditional load)
                    ra[s] = loadA(nx); rb[s] = loadB(nx);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, w.x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, w.y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, w.z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, w.w, acc, 0, 0, 0);
                }
            }
        }
    }
    // the four partial tiles summed in wave order (deterministic), epilogue by wave 0
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) red[wave - 1][i][lane] = acc[i];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int w = 0; w < PG_KS - 1; ++w)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] += red[w][i][lane];
    const int c = n0 + r;
    if (c >= p.N) return;
    const float bv = p.bias ? p.bias[c] : 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * h;        // point inside the tile
        float v = acc[i] + bv;
        if (p.act == 1) v = fmaxf(v, 0.f);
        else if (p.act == 2) v = tanhf(v);
        else if (p.act == 3) v = 1.f / (1.f + expf(-v));
        if (p.rowscale) v *= p.rowscale[(size_t)(m0 + row) * p.rs_stride];
        if (p.out_rows) p.out[(size_t)(m0 + row) * p.ldo + c] = v;
        else p.out[((size_t)b * p.N + c) * p.n + r0 + row] = v;
    }
}

}  // namespace jm

using namespace jm;

extern "C" int jm_points_linear_supported(int b, int n, int k1, int k2, int n_out) {
    return b >= 1 && n >= 32 && n % 32 == 0 && k1 >= 4 && k2 >= 0 && (k1 + k2) % 4 == 0 && n_out >= 1 && (long long)b * n / 32 <= 65535;
}

extern "C" int jm_points_linear(int b, int n, int k1, const float* x1, int k2, const float* x2, int n_out, const float* w, int ldw,
                                const float* bias, int act, const float* rowscale, int rowscale_stride, int out_rows, int ldo, float* out,
                                jm_stream_t stream) {
    JM_REQUIRE(jm_points_linear_supported(b, n, k1, k2, n_out), "points_linear: n %% 32 == 0, (k1 + k2) %% 4 == 0 (b %d, n %d, k1 %d, k2 %d)", b, n,
               k1, k2);
    JM_REQUIRE(x1 && (k2 == 0 || x2) && w && out && ldw >= k1 + k2 && ldw % 4 == 0 && act >= 0 && act <= 3 && (!out_rows || ldo >= n_out) &&
                   (!rowscale || rowscale_stride >= 1),
               "points_linear: bad arguments");
    JM_REQUIRE((reinterpret_cast<uintptr_t>(w) & 15u) == 0, "points_linear: the weight must be 16-byte aligned");
    PGemm p{};
    p.n = n; p.N = n_out; p.K1 = k1; p.K2 = k2; p.x1 = x1; p.x2 = k2 ? x2 : nullptr; p.W = w; p.ldw = ldw; p.bias = bias; p.act = act;
    p.rowscale = rowscale; p.rs_stride = rowscale_stride; p.out = out; p.out_rows = out_rows ? 1 : 0; p.ldo = ldo;
    hipLaunchKernelGGL(points_gemm_kernel, dim3((unsigned)divup(n_out, 32), (unsigned)(b * (n / 32))), dim3(64 * PG_KS), 0, (hipStream_t)stream, p);
    return check_launch("points_linear");
}
