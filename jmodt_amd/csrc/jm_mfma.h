// jm_mfma.h — the fp32 MFMA inner block shared by the fused SA kernel (sa_mlp.hip) and the affinity GEMMs
// (affinity.hip): A operand k-major in LDS, B operand straight from L1/L2 in the packed layout of
// jm_sa_mlp_pack ( Wp[kt][n][khalf][kk] = W[n][16 kt + 2 kk + khalf] ), one k-tile of register prefetch,
// no barrier inside.
#pragma once
#include "jm_common.h"

namespace jm {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int SM_BM = 128, SM_LDP = SM_BM + 4;   // rows per tile, padded row stride of the k-major LDS tiles

__host__ __device__ inline int pad_to(int v, int m) { return (v + m - 1) / m * m; }
// padded contraction length of a FIRST set-abstraction layer in jm_sa_mlp_pack's layout: whole 16-channel groups of features,
// xyz in a group of its own (sa_mlp.hip)
__host__ __device__ inline int sa_first_kp(int cin) { return pad_to(cin - 3, 16) + 16; }
// layers small enough for the vector-pipe kernel of sa_xyz.hip get a second, k-major copy [cin][cout] behind the MFMA layout
__host__ __device__ inline int sa_kmajor_elems(int cout, int cin) { return (cout <= 64 && cin <= 64 && cout % 16 == 0) ? cin * cout : 0; }

// acc += A(128 rows x 16 nkt, LDS k-major) x W-tile; this wave owns rows wm*64.., columns of bp (+32 if TWO)
//   A      : LDS buffer [k][SM_LDP]
//   bp     : this lane's packed weights for k-tile 0 of the range; kt_stride floats per k-tile
//   bpre   : in  = k-tile 0's B operand, already loaded by the previous stage;
//            out = the first B operand of the NEXT stage (next_bp)
//   SUBV   : the A operand is relu(A - V): vt0 / vt1 = this lane's rows of a per-centre table in LDS for its two
//            32-row blocks, 16 floats per k-tile laid out [khalf][8] (see sa_mlp.hip, pre-projected first layer); the
//            subtraction and the ReLU are VALU work of THIS wave, issued under its own MFMAs
template <bool TWO, bool SUBV = false>
__device__ __forceinline__ void mfma_ktiles(const float* __restrict__ A, int nkt, const float* __restrict__ bp,
                                            size_t kt_stride, int a_off, f32x16 (&acc)[2][2], float4 (&bpre)[4],
                                            const float* __restrict__ next_bp, const float* __restrict__ vt0 = nullptr,
                                            const float* __restrict__ vt1 = nullptr) {
    float4 bc[4], bn[4];
    float ac[16], an[16];
    float4 vc[4], vn[4];      // SUBV: [block 0 lo, hi | block 1 lo, hi]
    auto loadB = [&](float4 (&b)[4], const float* q) {
        b[0] = *reinterpret_cast<const float4*>(q);
        b[1] = *reinterpret_cast<const float4*>(q + 4);
        b[2] = *reinterpret_cast<const float4*>(q + 512);       // column + 32: (32 * 2) * 8 floats on
        b[3] = *reinterpret_cast<const float4*>(q + 516);
    };
    auto loadA = [&](float (&a)[16], int kt) {
        const float* q = A + (size_t)kt * 16 * SM_LDP + a_off;       // a_off = khalf * SM_LDP + wm * 64 + lr
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            a[2 * kk] = q[(2 * kk) * SM_LDP];
            a[2 * kk + 1] = q[(2 * kk) * SM_LDP + 32];
        }
    };
    auto loadV = [&](float4 (&v)[4], int kt) {
        if (SUBV) {
            v[0] = *reinterpret_cast<const float4*>(vt0 + kt * 16); v[1] = *reinterpret_cast<const float4*>(vt0 + kt * 16 + 4);
            v[2] = *reinterpret_cast<const float4*>(vt1 + kt * 16); v[3] = *reinterpret_cast<const float4*>(vt1 + kt * 16 + 4);
        }
    };
    auto mm_raw = [&](const float (&a)[16], const float4 (&b)[4]) {
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
    auto mm = [&](const float (&a)[16], const float4 (&b)[4], const float4 (&v)[4]) {
        if (!SUBV) { mm_raw(a, b); return; }
        float t[16];
        const float v0[8] = {v[0].x, v[0].y, v[0].z, v[0].w, v[1].x, v[1].y, v[1].z, v[1].w};
        const float v1[8] = {v[2].x, v[2].y, v[2].z, v[2].w, v[3].x, v[3].y, v[3].z, v[3].w};
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            t[2 * kk] = fmaxf(a[2 * kk] - v0[kk], 0.f);
            t[2 * kk + 1] = fmaxf(a[2 * kk + 1] - v1[kk], 0.f);
        }
        mm_raw(t, b);
    };
    // sched_barrier(0): keep the prefetches where they are written — left alone, the scheduler sinks
    // them to their first use and the L2 latency lands on the MFMA pipe
#pragma unroll
    for (int q = 0; q < 4; ++q) bc[q] = bpre[q];
    loadA(ac, 0);
    loadV(vc, 0);
    int kt = 0;
    for (; kt + 2 <= nkt; kt += 2) {
        loadB(bn, bp + (size_t)(kt + 1) * kt_stride);
        loadA(an, kt + 1);
        loadV(vn, kt + 1);
        __builtin_amdgcn_sched_barrier(0);
        mm(ac, bc, vc);
        __builtin_amdgcn_sched_barrier(0);
        // the k-tile after next — or, on the last trip, the NEXT stage's first B operand;
        // unconditional loads on a selected address, no branchy waits
        const bool more = kt + 2 < nkt;
        loadB(bc, more ? bp + (size_t)(kt + 2) * kt_stride : next_bp);
        loadA(ac, more ? kt + 2 : kt);
        loadV(vc, more ? kt + 2 : kt);
        __builtin_amdgcn_sched_barrier(0);
        mm(an, bn, vn);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (kt < nkt) {            // odd tail: bc / ac hold k-tile nkt-1
        loadB(bpre, next_bp);
        __builtin_amdgcn_sched_barrier(0);
        mm(ac, bc, vc);
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) bpre[q] = bc[q];
    }
}

// ---------------------------------------------------------------------------------------------------------
// 32-row tiles (sa_mlp_wide.hip, li_fusion.hip): one 32x32 MFMA row block per workgroup, the four waves split the
// columns (wave w owns 32-column blocks w, w+4, ...)
constexpr int SW_BM = 32, SW_LD = SW_BM + 4;     // rows per tile, padded row stride of the k-major LDS tiles

// acc[j] += A(32 rows x 16 nkt, k-major with row stride lda: an LDS tile, or a (C, n) global tensor read in place)
//           x W-tile of this wave's column blocks cb0 + 4 j, j < NOWN
//   bp : this lane's packed weights for k-tile 0 of block cb0; kt_stride floats per k-tile;  a_off = khalf * lda + row
template <int NOWN>
__device__ __forceinline__ void wide_ktiles(const float* __restrict__ A, int nkt, const float* __restrict__ bp,
                                            size_t kt_stride, int a_off, f32x16 (&acc)[2], size_t lda = SW_LD) {
    float bc[NOWN][8], bn[NOWN][8];
    float ac[8], an[8];
    auto loadB = [&](float (&b)[NOWN][8], const float* q) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NOWN; ++j) {          // column block + 4: 4 blocks x 32 columns x 16 floats further on
            const float4 lo = *reinterpret_cast<const float4*>(q + j * 2048);
            const float4 hi = *reinterpret_cast<const float4*>(q + j * 2048 + 4);
            b[j][0] = lo.x; b[j][1] = lo.y; b[j][2] = lo.z; b[j][3] = lo.w;
            b[j][4] = hi.x; b[j][5] = hi.y; b[j][6] = hi.z; b[j][7] = hi.w;
        }
    };
    auto loadA = [&](float (&a)[8], int kt) __attribute__((always_inline)) {
        const float* q = A + (size_t)kt * 16 * lda + a_off;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) a[kk] = q[(2 * kk) * lda];
    };
    auto mm = [&](const float (&a)[8], const float (&b)[NOWN][8]) __attribute__((always_inline)) {
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
#pragma unroll
            for (int j = 0; j < NOWN; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], b[j][kk], acc[j], 0, 0, 0);
    };
    loadB(bc, bp);
    loadA(ac, 0);
    int kt = 0;
    for (; kt + 2 <= nkt; kt += 2) {
        loadB(bn, bp + (size_t)(kt + 1) * kt_stride);
        loadA(an, kt + 1);
        __builtin_amdgcn_sched_barrier(0);
        mm(ac, bc);
        __builtin_amdgcn_sched_barrier(0);
        const int nx = kt + 2 < nkt ? kt + 2 : kt;                    // unconditional loads on a clamped address
        loadB(bc, bp + (size_t)nx * kt_stride);
        loadA(ac, nx);
        __builtin_amdgcn_sched_barrier(0);
        mm(an, bn);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (kt < nkt) mm(ac, bc);
}

// The same product with the B operand (weights from L1 / L2) requested a whole GROUP of GK k-tiles ahead: two register sets of
// GK stages, the loads of group g + 1 issued in front of the MFMAs of group g, no branch inside a group (with wave-uniform
// branches around single k-tiles hipcc's s_waitcnt placement falls back to vmcnt(0) and the prefetch is lost).  For the kernels
// that run ONE wave per SIMD (sa_mlp_wide_kernel: its LDS tiles leave room for one workgroup per CU) nothing else hides the
// ~1 us L2 round trip of a weight tile; one k-tile ahead covers the 8 .. 16 MFMAs of one k-tile only.  The nkt % GK last
// k-tiles run one ahead.  Same k order, same accumulation chain as wide_ktiles: same bits.
template <int NOWN, int GK>
__device__ __forceinline__ void wide_ktiles_deep(const float* __restrict__ A, int nkt, const float* __restrict__ bp,
                                                 size_t kt_stride, int a_off, f32x16 (&acc)[2], size_t lda = SW_LD) {
    float b0[GK][NOWN][8], b1[GK][NOWN][8];
    float a0[8], a1[8];
    auto loadB = [&](float (&d)[NOWN][8], const float* q) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NOWN; ++j) {
            const float4 lo = *reinterpret_cast<const float4*>(q + j * 2048);
            const float4 hi = *reinterpret_cast<const float4*>(q + j * 2048 + 4);
            d[j][0] = lo.x; d[j][1] = lo.y; d[j][2] = lo.z; d[j][3] = lo.w;
            d[j][4] = hi.x; d[j][5] = hi.y; d[j][6] = hi.z; d[j][7] = hi.w;
        }
    };
    auto loadG = [&](float (&d)[GK][NOWN][8], int g) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < GK; ++s) loadB(d[s], bp + (size_t)(g * GK + s) * kt_stride);
    };
    auto loadA = [&](float (&a)[8], int kt) __attribute__((always_inline)) {
        const float* q = A + (size_t)kt * 16 * lda + a_off;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) a[kk] = q[(2 * kk) * lda];
    };
    auto mm = [&](const float (&a)[8], const float (&w)[NOWN][8]) __attribute__((always_inline)) {
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
#pragma unroll
            for (int j = 0; j < NOWN; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], w[j][kk], acc[j], 0, 0, 0);
    };
    // the MFMAs of group g (k-tiles g GK ..), the A operand one k-tile ahead (LDS: short latency); GK even: a0 holds the first
    auto group = [&](const float (&w)[GK][NOWN][8], int g, int nxt) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < GK; s += 2) {
            loadA(a1, g * GK + s + 1);
            mm(a0, w[s]);
            loadA(a0, s + 2 < GK ? g * GK + s + 2 : nxt);
            mm(a1, w[s + 1]);
        }
    };
    static_assert(GK % 2 == 0, "even group size");
    const int ngrp = nkt / GK, last = nkt - 1;
    if (ngrp > 0) {
        loadG(b0, 0);
        loadA(a0, 0);
        int g = 0;
        for (; g + 2 <= ngrp; g += 2) {
            loadG(b1, g + 1);
            __builtin_amdgcn_sched_barrier(0);
            group(b0, g, (g + 1) * GK);
            __builtin_amdgcn_sched_barrier(0);
            loadG(b0, g + 2 < ngrp ? g + 2 : g + 1);              // clamped: an unconditional (redundant at the end) load
            __builtin_amdgcn_sched_barrier(0);
            const int after = (g + 2) * GK;
            group(b1, g + 1, after < last ? after : last);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (g < ngrp) {                                            // odd count: b0 holds the last group
            const int after = (g + 1) * GK;
            group(b0, g, after < last ? after : last);
        }
    }
    // the remaining nkt % GK k-tiles, one ahead
    int kt = ngrp * GK;
    if (kt < nkt) {
        loadB(b0[0], bp + (size_t)kt * kt_stride);
        loadA(a0, kt);
        for (; kt < nkt; ++kt) {
            const int nx = kt + 1 < nkt ? kt + 1 : kt;
            loadB(b1[0], bp + (size_t)nx * kt_stride);
            loadA(a1, nx);
            __builtin_amdgcn_sched_barrier(0);
            mm(a0, b0[0]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NOWN; ++j)
#pragma unroll
                for (int e = 0; e < 8; ++e) b0[0][j][e] = b1[0][j][e];
#pragma unroll
            for (int e = 0; e < 8; ++e) a0[e] = a1[e];
        }
    }
}

}  // namespace jm
