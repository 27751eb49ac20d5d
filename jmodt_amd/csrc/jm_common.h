// jm_common.h — shared helpers for the gfx950 kernels (wave64 everywhere).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/jmodt_hip.h"

namespace jm {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return JM_ELAUNCH;
    }
    return JM_OK;
}

#define JM_REQUIRE(cond, ...)          \
    do {                               \
        if (!(cond)) {                 \
            jm::set_error(__VA_ARGS__); \
            return JM_EINVAL;          \
        }                              \
    } while (0)

// Experiment switches (JM_* environment variables) exist only in the TOOLS build of the library
// (python -m jmodt_amd.csrc.build --tools -> tools/bin/libjmodt_hip_tools.so, used by tools/*.py); the product
// library is compiled without JM_TOOLS_BUILD and every switch folds to its default.
#ifdef JM_TOOLS_BUILD
#include <stdlib.h>
static inline int tune_env(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
#else
static inline constexpr int tune_env(const char*, int dflt) { return dflt; }
#endif

// Zero-fill as a KERNEL, not hipMemsetAsync: a memset captured into a HIP graph is not reliably ordered against the kernel nodes
// around it when the graph is replayed on this stack (tools/graph_memset_probe.py; it showed up as accumulators that were not
// zero under a replayed backward pass) — and every entry of this library must behave the same eagerly and under capture.
static __global__ void jm_zero_words_kernel(unsigned* __restrict__ p, size_t nwords) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}
static inline hipError_t jm_zero_async(void* p, size_t bytes, hipStream_t s) {
    if (bytes == 0) return hipSuccess;
    if ((bytes & 3u) || (reinterpret_cast<uintptr_t>(p) & 3u)) return hipMemsetAsync(p, 0, bytes, s);   // (no such caller)
    const size_t nwords = bytes >> 2;
    const size_t blocks = (nwords + 255) / 256;
    hipLaunchKernelGGL(jm_zero_words_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, s, (unsigned*)p, nwords);
    return hipGetLastError();
}

static inline int divup(int a, int b) { return (a + b - 1) / b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- wave64 DPP reductions ---------------------------------------------------------------
// DPP control words (gfx9 encoding): quad_perm = 0x00..0xFF, row_mirror 0x140,
// row_half_mirror 0x141, row_bcast15 0x142, row_bcast31 0x143.
#define JM_DPP_XOR1 0xB1            /* quad_perm [1,0,3,2] */
#define JM_DPP_XOR2 0x4E            /* quad_perm [2,3,0,1] */
#define JM_DPP_HALF_MIRROR 0x141
#define JM_DPP_MIRROR 0x140
#define JM_DPP_BCAST15 0x142
#define JM_DPP_BCAST31 0x143

// max over the 64 lanes of a signed int, returned wave-uniform (SGPR via readlane 63).
// 4 butterfly steps make every 16-lane row uniform, then two row broadcasts fold the rows.
// Written as v_max_i32 with the DPP modifier on the source operand: one instruction per step
// (hipcc lowers the update_dpp builtin to v_mov_b32 + s_nop + v_mov_b32_dpp + v_max, 4 issue
// slots per step).  The s_nop 1 before each step is the VALU-write -> DPP-read hazard (2 wait
// states), which hipcc does not insert inside an asm statement.
__device__ __forceinline__ int wave_max_i32(int v) {
    asm volatile(
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v));
    return __builtin_amdgcn_readlane(v, 63);
}

// float sum over each 32-lane half of the wave (lanes 0-31 / 32-63); the total is valid in the UPPER
// 16 lanes of each half (16-31 / 48-63).  5 fused DPP adds; a __shfl_xor tree costs a ds_bpermute
// round trip per step.
__device__ __forceinline__ float half_sum_f32_dpp(float v) {
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v));
    return v;
}

// bitwise OR over the 64 lanes (same DPP pattern as wave_max_i32), wave-uniform result
__device__ __forceinline__ unsigned wave_or_u32(unsigned v) {
    asm volatile(
        "s_nop 1\n\t"
        "v_or_b32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_or_b32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_or_b32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_or_b32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_or_b32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_or_b32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// integer sum over the 64 lanes (same DPP pattern), wave-uniform result
__device__ __forceinline__ unsigned wave_sum_u32(unsigned v) {
    asm volatile(
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned long long wave_or_u64(unsigned long long v) {
    const unsigned lo = wave_or_u32((unsigned)v), hi = wave_or_u32((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

// v_min_f32 / v_max_f32 without the canonicalising v_max hipcc puts in front of fminf/fmaxf
// (inputs here are never signalling NaNs)
__device__ __forceinline__ float fast_min(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ float fast_max(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ float fast_max3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// workgroup barrier for LDS-only hand-offs: waits for this wave's LDS traffic, not for global
// memory (a __syncthreads() also drains vmcnt)
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ float wave_sum_f32(float v) {
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}
__device__ __forceinline__ float wave_max_f32(float v) {
    v = fmaxf(v, __shfl_xor(v, 1));
    v = fmaxf(v, __shfl_xor(v, 2));
    v = fmaxf(v, __shfl_xor(v, 4));
    v = fmaxf(v, __shfl_xor(v, 8));
    v = fmaxf(v, __shfl_xor(v, 16));
    v = fmaxf(v, __shfl_xor(v, 32));
    return v;
}

// inclusive prefix sum over the 64 lanes with DPP row shifts + row broadcasts (no LDS crossbar round trips: a __shfl_up
// tree is six dependent ds_bpermute latencies).  Lanes without a source read 0 (old = 0, bound_ctrl off).
__device__ __forceinline__ int wave_incl_scan_i32_dpp(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);     // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);     // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);     // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);     // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);     // row_bcast:15 into rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);     // row_bcast:31 into rows 2, 3
    return v;
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }

// number of set bits of `mask` strictly below this lane
__device__ __forceinline__ int mbcnt(unsigned long long mask) {
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

// squared distance with the contraction nvcc/clang apply to dx*dx + dy*dy + dz*dz
// (see oracle/jmodt_oracle.c header): fma(dz,dz, fma(dx,dx, dy*dy)).  The library is built
// with -ffp-contract=off, so these are the only fused operations.
__device__ __forceinline__ float sqdist3(float dx, float dy, float dz) {
    return __builtin_fmaf(dz, dz, __builtin_fmaf(dx, dx, dy * dy));
}

}  // namespace jm
