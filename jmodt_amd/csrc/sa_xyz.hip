// sa_xyz.hip — the xyz-only set-abstraction scale (first level of the RPN backbone, config.py:75-82:
// SharedMLP [3, 16, 16, 32] over 16 samples and [3, 32, 32, 64] over 32 samples of 4096 centres per frame) on the VECTOR
// pipe (gfx950).
//
// _PointnetSAModuleBase.forward (jmodt/ops/pointnet2/pointnet2_modules.py:41-52) with QueryAndGroup(use_xyz=True) and no
// input features: rows = (centre, sample), row input = xyz[idx] - centre (pointnet2_utils.py:259-262), three
// 1x1 conv + folded BN + ReLU layers, max over the samples.  7.5 GFLOP for 1.57 M rows at batch 8.
//
// Why not the matrix cores: fp32 MFMA and packed fp32 VALU (v_pk_fma_f32) have the SAME peak on this chip (157 TFLOP/s), and
// with contraction lengths of 3, 16 and 32 the MFMA route pays for its operand layout instead of computing: the general
// kernel (sa_mlp_kernel: gather waves -> LDS tiles -> 32x32x2 MFMAs with K padded to 16, 256 registers, 44 of them
// spilled) takes 0.19 ms per scale = 0.12 of the peak.  Here a LANE owns a row: the three layers are straight-line
// v_pk_fma_f32 on registers (two output channels per instruction, the input broadcast into both halves), the weights are
// wave-uniform and come through the SCALAR cache straight into the SGPR operand of the FMA (s_load_dwordx4.. of the k-major
// copy jm_sa_mlp_pack appends to small layers; as vector registers from LDS they push hipcc into thousands of spills: the
// weights of a fully unrolled layer do not fit next to its accumulators), the max over a centre's 16 / 32 samples is 4 / 5
// DPP max steps per channel
// inside the wave, and the (centre, channel) maxima leave through an LDS tile as 256-byte row segments of the (B, C, M)
// output.  No gather stage, no barrier in the row loop.
#include "jm_mfma.h"
#include <type_traits>

namespace jm {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
// LISTED: the weight tables through the CONSTANT address space.  The persistent pass loop puts the output stores of pass i in
// front of the weight loads of pass i + 1; from the global address space hipcc then cannot prove the tables unclobbered and
// fetches them with (uniform) VECTOR loads — 824 global_load_dwordx4 instead of scalar-cache loads into the FMA's SGPR
// operand: + 55 % on full lists.  Constant-address-space loads at a uniform address are always scalar.
typedef const float __attribute__((address_space(4))) * cptr4;
__device__ __forceinline__ f32x4v ld4(const float* q) {
    const float4 v = *reinterpret_cast<const float4*>(q);
    return (f32x4v){v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ f32x4v ld4(cptr4 q) { return *(const f32x4v __attribute__((address_space(4)))*)q; }

struct SaXyzParams {
    int N, M, ns;                  // points per frame, centres per frame, samples per centre (16 or 32)
    const float* xyz;              // (B, N, 3)
    const float* new_xyz;          // (B, M, 3)
    const int* idx;                // (B, M, ns)
    const float *w0, *w1, *w2;     // k-major tables [cin][cout] (jm_sa_mlp_pack appends them behind the MFMA layout)
    const float *b0, *b1, *b2;     // biases, zero padded
    float* out;                    // (B, C3, M), frame stride obs
    size_t obs;
    int groups;                    // LISTED: B * M
    const int* cls_count;          // LISTED: [8] groups per class q (device memory, sg_plan_kernel)
    const int* glist;              // LISTED: class q's group ids at glist[q * groups ...]
};

// max over the 16 lanes of each DPP row (result in every lane of the row) of FOUR values at once: the four independent
// chains fill each other's two wait states between a VALU write and a DPP read of the same register (one chain alone
// spends as many s_nops as max instructions).  NS == 32: ... and of each pair of rows, valid in the ODD rows (lanes 16-31,
// 48-63): row_bcast:15 hands lane 15 of the previous row to rows 1 and 3.
#define JM_DPP4(ctrl)                                               \
    "v_max_f32_dpp %0, %0, %0 " ctrl " row_mask:0xf bank_mask:0xf\n\t" \
    "v_max_f32_dpp %1, %1, %1 " ctrl " row_mask:0xf bank_mask:0xf\n\t" \
    "v_max_f32_dpp %2, %2, %2 " ctrl " row_mask:0xf bank_mask:0xf\n\t" \
    "v_max_f32_dpp %3, %3, %3 " ctrl " row_mask:0xf bank_mask:0xf\n\t"
template <int NS>
__device__ __forceinline__ void samples_max4(float& a, float& b, float& c, float& d) {
    asm volatile("s_nop 1\n\t" JM_DPP4("quad_perm:[1,0,3,2]") JM_DPP4("quad_perm:[2,3,0,1]") JM_DPP4("row_half_mirror")
                 JM_DPP4("row_mirror") "s_nop 0"
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    if (NS == 32)
        asm volatile("s_nop 1\n\t"
                     "v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                     "v_max_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                     "v_max_f32_dpp %2, %2, %2 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                     "v_max_f32_dpp %3, %3, %3 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                     "s_nop 1"
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
// LISTED: max over the aligned groups of 2^q lanes (q wave-uniform, <= log2 NS): the first q of the same five steps
template <int NS>
__device__ __forceinline__ void samples_max4_q(int q, float& a, float& b, float& c, float& d) {
    if (q >= 1) asm volatile("s_nop 1\n\t" JM_DPP4("quad_perm:[1,0,3,2]") "s_nop 0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    if (q >= 2) asm volatile("s_nop 1\n\t" JM_DPP4("quad_perm:[2,3,0,1]") "s_nop 0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    if (q >= 3) asm volatile("s_nop 1\n\t" JM_DPP4("row_half_mirror") "s_nop 0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    if (q >= 4) asm volatile("s_nop 1\n\t" JM_DPP4("row_mirror") "s_nop 0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    if (NS == 32 && q >= 5)
        asm volatile("s_nop 1\n\t"
                     "v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                     "v_max_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                     "v_max_f32_dpp %2, %2, %2 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                     "v_max_f32_dpp %3, %3, %3 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                     "s_nop 1"
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
#undef JM_DPP4

// ROWS rows per lane (rows tid, tid + T, ...: other centres of the same workgroup) share every weight load: the weights are
// the scalar cache's traffic — 12.7 KB per wave and pass at [32, 32, 64] —, and with one row per lane the kernel waits for
// them (0.225 ms); two rows halve that traffic per FMA (0.144 ms at 256 registers, two waves per SIMD).  Sharing the weights
// in the last layer only (two thirds of the FMAs) with the first two layers row by row, to fit 128 registers and four waves
// per SIMD: hipcc still wants 256 registers, and capped at 128 it spills 115 of them — 0.355 ms.
// LISTED (round 4, sa_groups.hip): a workgroup pass = 1024 rows of ONE class q = 1024 >> q groups of 2^q rows each (the first
// 2^q entries of the group's neighbour list: its distinct rows + back-fill), the maximum runs over the aligned 2^q lanes, and
// every group's channels go to its own (frame, centre) position: bit-identical to the dense mode (a row's value depends on
// its point and its centre only), persistent grid with the pass count read from the plan in device memory.
template <int H1, int H2, int C3, int NS, int ROWS, bool LISTED>
__global__ void __launch_bounds__(1024 / ROWS)
sa_xyz_valu_kernel(SaXyzParams p) {
    constexpr int T = 1024 / ROWS, CPW = 1024 / NS;       // threads, centres per workgroup (1024 rows)
    constexpr int QF = NS == 32 ? 5 : 4, TW = LISTED ? 64 : CPW;
    __shared__ __attribute__((aligned(16))) float tile[C3][TW + 4];
    __shared__ int TS[10];
    const int tid = threadIdx.x, lane = tid & 63;
    int total = 1;
    if (LISTED) {
        if (tid == 0) {
            int acc_t = 0;
            for (int c = 0; c <= QF; ++c) {
                TS[c] = acc_t;
                acc_t += (int)((((long long)p.cls_count[c] << c) + 1023) >> 10);
            }
            TS[QF + 1] = acc_t;
        }
        __syncthreads();
        total = TS[QF + 1];
    }
  for (int pass = LISTED ? (int)blockIdx.x : 0; pass < total; pass += LISTED ? (int)gridDim.x : 1) {
    // LISTED: the weights are loop invariant — left alone, hipcc hoists their scalar loads out of the pass loop and then spills
    // 254 SGPRs through v_writelane inside it (+60 % on full lists).  Pointers made opaque per pass keep the loads where they are
    // used; they stay __restrict__ (without it the loads may alias the output stores and leave the scalar cache: + 55 %)
    // (an opaque zero OFFSET, not an opaque pointer: that one would lose its address space and turn into flat vector loads)
    int zero = 0;
    if (LISTED) asm volatile("" : "+s"(zero));
    using wptr = cptr4;     // (the dense mode as well: from the global address space hipcc fetched 396 of its weight quads with
                            // vector loads and spilled; through the scalar cache only: 137 -> ~100 us at [32, 32, 64] x 32)
    const wptr W0t = (wptr)(p.w0 + zero), W1t = (wptr)(p.w1 + zero), W2t = (wptr)(p.w2 + zero);
    const wptr B0 = (wptr)(p.b0 + zero), B1 = (wptr)(p.b1 + zero), B2 = (wptr)(p.b2 + zero);
    int q = QF, slot0 = 0, cnt_q = 0;
    if (LISTED) {
        q = 0;
        for (int c = 1; c <= QF; ++c)
            if (pass >= TS[c]) q = c;                     // the last class whose first pass is <= pass (empty classes lose)
        q = __builtin_amdgcn_readfirstlane(q);
        slot0 = (pass - TS[q]) * (1024 >> q);
        cnt_q = p.cls_count[q];
    }
    long long goff[ROWS];                                 // LISTED: this row's group's output offset (frame * obs + centre), -1: padding
    f32x2 din[ROWS][2];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        long long row;
        int g;
        if (LISTED) {
            const int v = r * T + tid, slot = slot0 + (v >> q);
            const bool ok = slot < cnt_q;
            g = p.glist[(size_t)q * p.groups + (ok ? slot : slot0)];       // padding rows repeat the pass's first group
            row = (long long)g * NS + (v & ((1 << q) - 1));
            goff[r] = ok ? (long long)((size_t)(g / p.M) * p.obs + (size_t)(g % p.M)) : -1;
        } else {
            row = (long long)blockIdx.x * 1024 + r * T + tid;             // (frame, centre, sample) flattened; M * NS % 1024 == 0
            g = (int)(row / NS);
        }
        const int bi = g / p.M;
        const int k = p.idx[row];
        const float* qq = p.xyz + ((size_t)bi * p.N + k) * 3;
        const float* c = p.new_xyz + (size_t)g * 3;
        din[r][0] = (f32x2){qq[0] - c[0], qq[1] - c[1]};
        din[r][1] = (f32x2){qq[2] - c[2], 0.f};
    }

    // One layer for a chunk of 16 output channels: acc[r][8] (pairs) += in[r][kk] * Wt[kk][n0 .. n0 + 15] over kk < KIN; the
    // addresses are uniform, so the weights are scalar loads into SGPRs (the SGPR operand of v_pk_fma_f32)
    auto chunk16 = [&](auto KIN_, auto IN_, const f32x2 (*in)[decltype(IN_)::value], bool relu_in, wptr Wt, int ldw,
                       wptr bias, int n0, f32x2 (&acc)[ROWS][8]) __attribute__((always_inline)) {
        constexpr int KIN = decltype(KIN_)::value;
#pragma unroll
        for (int r = 0; r < ROWS; ++r)
#pragma unroll
            for (int n = 0; n < 8; ++n) acc[r][n] = (f32x2){bias[n0 + 2 * n], bias[n0 + 2 * n + 1]};
#pragma unroll
        for (int kk = 0; kk < KIN; ++kk) {
            f32x4v wc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) wc[j] = ld4(Wt + (kk * ldw + n0 + 4 * j));
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                float a = (kk & 1) ? in[r][kk >> 1].y : in[r][kk >> 1].x;
                if (relu_in) a = fmaxf(a, 0.f);
                const f32x2 pp = {a, a};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[r][2 * j] = __builtin_elementwise_fma(pp, (f32x2){wc[j][0], wc[j][1]}, acc[r][2 * j]);
                    acc[r][2 * j + 1] = __builtin_elementwise_fma(pp, (f32x2){wc[j][2], wc[j][3]}, acc[r][2 * j + 1]);
                }
            }
        }
    };
    f32x2 h1[ROWS][H1 / 2], h2[ROWS][H2 / 2];
#pragma unroll
    for (int ch = 0; ch < H1 / 16; ++ch) {
        f32x2 acc[ROWS][8];
        chunk16(std::integral_constant<int, 3>{}, std::integral_constant<int, 2>{}, din, false, W0t, H1, B0, ch * 16, acc);
#pragma unroll
        for (int r = 0; r < ROWS; ++r)
#pragma unroll
            for (int n = 0; n < 8; ++n) h1[r][ch * 8 + n] = acc[r][n];
    }
#pragma unroll
    for (int ch = 0; ch < H2 / 16; ++ch) {
        f32x2 acc[ROWS][8];
        chunk16(std::integral_constant<int, H1>{}, std::integral_constant<int, H1 / 2>{}, h1, true, W1t, H2, B1, ch * 16, acc);
#pragma unroll
        for (int r = 0; r < ROWS; ++r)
#pragma unroll
            for (int n = 0; n < 8; ++n) h2[r][ch * 8 + n] = acc[r][n];
    }
    // last layer chunk by chunk: the 16 channels of a chunk are reduced over the centre's samples (max and ReLU commute:
    // pointnet2_modules.py:48-52 applies the ReLU first) and parked in the workgroup's (channel, centre) tile at once
    // (the maximum over 2^q lanes is valid in every lane of the group up to 16 lanes, in the odd DPP rows for 32)
    const bool writer = LISTED ? (q == 5 ? (lane & 31) == 16 : (lane & ((1 << q) - 1)) == 0)
                               : (NS == 16 ? (lane & 15) == 0 : (lane & 31) == 16);
#pragma unroll
    for (int ch = 0; ch < C3 / 16; ++ch) {
        f32x2 acc[ROWS][8];
        chunk16(std::integral_constant<int, H2>{}, std::integral_constant<int, H2 / 2>{}, h2, true, W2t, C3, B2, ch * 16, acc);
        // all accumulator pairs are complete HERE: without this fence hipcc orders the layer by consumer — one accumulator
        // through all k, then its reduction, then the next — which needs every weight of the chunk live at once (the SGPRs
        // spill through v_writelane: 1276 registers)
#pragma unroll
        for (int r = 0; r < ROWS; ++r)
            asm volatile("" : "+v"(acc[r][0]), "+v"(acc[r][1]), "+v"(acc[r][2]), "+v"(acc[r][3]), "+v"(acc[r][4]), "+v"(acc[r][5]),
                              "+v"(acc[r][6]), "+v"(acc[r][7]));
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int cl = LISTED ? (r * T + tid) >> q : (r * T + tid) / NS;     // centre (group) within the workgroup's pass
#pragma unroll
            for (int n = 0; n < 8; n += 2) {                           // channels 2n .. 2n + 3 of the chunk
                float v0 = acc[r][n].x, v1 = acc[r][n].y, v2 = acc[r][n + 1].x, v3 = acc[r][n + 1].y;
                if (LISTED) samples_max4_q<NS>(q, v0, v1, v2, v3); else samples_max4<NS>(v0, v1, v2, v3);
                if (LISTED && q < 4) {
                    // 1024 >> q > 64 groups per pass: no tile; the >= 4 writer lanes of a 16-lane row hold groups that are
                    // neighbours in the class list, which the plan fills in runs of ascending group ids (coalescing stores)
                    if (writer && goff[r] >= 0) {
                        float* o = p.out + (size_t)goff[r] + (size_t)(ch * 16 + 2 * n) * (size_t)p.M;
                        o[0] = fmaxf(v0, 0.f); o[(size_t)p.M] = fmaxf(v1, 0.f);
                        o[2 * (size_t)p.M] = fmaxf(v2, 0.f); o[3 * (size_t)p.M] = fmaxf(v3, 0.f);
                    }
                } else if (writer) {
                    tile[ch * 16 + 2 * n + 0][cl] = fmaxf(v0, 0.f); tile[ch * 16 + 2 * n + 1][cl] = fmaxf(v1, 0.f);
                    tile[ch * 16 + 2 * n + 2][cl] = fmaxf(v2, 0.f); tile[ch * 16 + 2 * n + 3][cl] = fmaxf(v3, 0.f);
                }
            }
        }
    }
    if (LISTED) {
        if (q >= 4) {                                                  // 64 (q = 4) or 32 (q = 5) groups through the tile
            __syncthreads();
            const int gpt = 1024 >> q;
            for (int e = tid; e < C3 * gpt; e += T) {
                const int n = e / gpt, j = e - n * gpt;
                if (slot0 + j < cnt_q) {
                    const int g = p.glist[(size_t)q * p.groups + slot0 + j];
                    p.out[(size_t)(g / p.M) * p.obs + (size_t)n * p.M + (size_t)(g % p.M)] = tile[n][j];
                }
            }
            __syncthreads();                                           // the next pass reuses the tile
        }
    } else {
        __syncthreads();
        const long long crow0 = (long long)blockIdx.x * CPW;           // first (frame, centre) of the workgroup: one frame
        const int ob = (int)(crow0 / p.M), om = (int)(crow0 - (long long)ob * p.M);      // (CPW divides M)
        for (int e = tid; e < C3 * CPW; e += T) {
            const int n = e / CPW, j = e - n * CPW;
            p.out[(size_t)ob * p.obs + (size_t)n * p.M + om + j] = tile[n][j];
        }
    }
  }
}

// widths[1..3] of the two scales this file instantiates; anything else stays on the matrix-core kernels
bool sa_xyz_valu_supported(int m, int c, int nsample, int num_layers, const int* widths) {
    if (c != 0 || num_layers != 3) return false;
    const bool a = nsample == 16 && widths[1] == 16 && widths[2] == 16 && widths[3] == 32;
    const bool b = nsample == 32 && widths[1] == 32 && widths[2] == 32 && widths[3] == 64;
    return (a || b) && m % (1024 / nsample) == 0;
}

int sa_xyz_valu_launch(int b, int n, int m, int nsample, const float* xyz, const float* new_xyz, const int* idx,
                       const int* widths, const float* const* weights, const float* const* biases, float* out, size_t obs, hipStream_t s,
                       const int* cls_count, const int* glist) {
    SaXyzParams p{};
    p.N = n; p.M = m; p.ns = nsample; p.xyz = xyz; p.new_xyz = new_xyz; p.idx = idx;
    // the k-major copies behind the MFMA layouts (sa_mlp_pack_kernel): Kp x Np floats in
    p.w0 = weights[0] + (size_t)sa_first_kp(3) * pad_to(widths[1], 128);
    p.w1 = weights[1] + (size_t)pad_to(widths[1], 16) * pad_to(widths[2], 128);
    p.w2 = weights[2] + (size_t)pad_to(widths[2], 16) * pad_to(widths[3], 128);
    p.b0 = biases[0]; p.b1 = biases[1]; p.b2 = biases[2];
    p.out = out;
    p.obs = obs ? obs : (size_t)widths[3] * (size_t)m;
    const long long rows = (long long)b * m * nsample;
    JM_REQUIRE(rows / 1024 < (1LL << 31), "sa_xyz: too many rows");
    const dim3 grid((unsigned)(rows / 1024));
    if (cls_count) {
        // listed: passes <= the dense count + one partial pass per class; persistent over at most two workgroups per CU
        p.groups = b * m; p.cls_count = cls_count; p.glist = glist;
        const long long bound = rows / 1024 + (nsample == 16 ? 5 : 6);
        const dim3 lgrid((unsigned)(bound < 512 ? bound : 512));
        if (nsample == 16) hipLaunchKernelGGL((sa_xyz_valu_kernel<16, 16, 32, 16, 2, true>), lgrid, dim3(512), 0, s, p);
        else hipLaunchKernelGGL((sa_xyz_valu_kernel<32, 32, 64, 32, 2, true>), lgrid, dim3(512), 0, s, p);
        return check_launch("sa_xyz_valu (listed)");
    }
    if (nsample == 16) hipLaunchKernelGGL((sa_xyz_valu_kernel<16, 16, 32, 16, 2, false>), grid, dim3(512), 0, s, p);
    else hipLaunchKernelGGL((sa_xyz_valu_kernel<32, 32, 64, 32, 2, false>), grid, dim3(512), 0, s, p);
    return check_launch("sa_xyz_valu");
}

}  // namespace jm
