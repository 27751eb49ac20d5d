// jm_grid.h — per-frame hash grid shared by the radius search (ball_query_grid.hip) and the 3-NN search
// (three_nn_grid.hip): cell / bucket functions and the one-workgroup-per-frame build kernel.
#pragma once
#include "jm_common.h"

namespace jm {

constexpr int BG_T_MAX = 32768;          // buckets per frame <= this; the build kernel keeps the table in LDS (128 KB)
constexpr int BG_MIN_N = 2048;           // below: the brute-force scan is already a few microseconds
constexpr int BG_MAX_N = 131072;         // above: the per-wave bitmaps no longer fit the LDS

__device__ __forceinline__ int bg_cell(float x, float inv_h) {
    float f = floorf(x * inv_h);
    f = fminf(fmaxf(f, -1.0e9f), 1.0e9f);          // monotone; NaN -> -1e9 (never a hit anyway)
    return (int)f;
}

__device__ __forceinline__ unsigned bg_bucket(int ix, int iy, int iz, unsigned tmask) {
    return ((unsigned)ix * 73856093u ^ (unsigned)iy * 19349663u ^ (unsigned)iz * 83492791u) & tmask;
}

// ------------------------------------------------------------------ build: one workgroup per frame
// TPT = table entries per thread (T = 1024 TPT buckets); PPT = points per thread kept in registers between the histogram
// and the scatter pass (n <= 1024 PPT; 0: re-read).  Bucket ORDER in the sorted array is "thread-major": thread t owns
// buckets t, t + 1024, t + 2048, ... (conflict-free LDS columns); the table stores (start, end) per bucket, so the order
// is private to this kernel.
// hdr given without AUTO_H: 1 / cell edge = the inv_h ARGUMENT x hdr[frame].x (a second point set binned on the first one's
// grid, or on a coarser one: 0.5 = cells of twice the edge, exactly the union of 2 x 2 x 2 cells of the first grid).
// AUTO_H (needs PPT > 0): the cell edge is chosen from the frame's own bounding box — 1.5 x the point spacing of n points
// spread over the box's volume, its largest face or its longest edge, whichever is largest (flat and line-like clouds
// included) — and written with its reciprocal to hdr[frame] = {inv_h, h, 0, 0} for the query kernel.
template <int TPT, int PPT, bool AUTO_H = false>
__global__ void __launch_bounds__(1024)
bq_grid_build_kernel(int n, float inv_h, const float* __restrict__ xyz, uint2* __restrict__ tbl, float4* __restrict__ sorted,
                     float4* __restrict__ hdr = nullptr) {
    extern __shared__ __attribute__((aligned(16))) unsigned cnt[];       // T
    __shared__ unsigned wave_tot[16];
    constexpr int T = 1024 * TPT;
    constexpr unsigned tmask = (unsigned)T - 1u;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* pts = xyz + (size_t)blockIdx.x * n * 3;
    uint2* tb = tbl + (size_t)blockIdx.x * T;
    float4* so = sorted + (size_t)blockIdx.x * n;
    float px[PPT > 0 ? PPT : 1], py[PPT > 0 ? PPT : 1], pz[PPT > 0 ? PPT : 1];
    unsigned pb[PPT > 0 ? PPT : 1];
    if (!AUTO_H && hdr) inv_h *= hdr[blockIdx.x].x;               // (inv_h argument = scale of the header's value: 0.5 -> cells twice as wide)
    if (PPT > 0) {                                   // all loads in flight at once (clamped index: no branch around a load)
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const int k = min(tid + 1024 * j, n - 1);
            px[j] = pts[k * 3 + 0]; py[j] = pts[k * 3 + 1]; pz[j] = pts[k * 3 + 2];
        }
    }
#pragma unroll
    for (int e = 0; e < TPT; e += 4) *reinterpret_cast<uint4*>(cnt + (size_t)tid * 4 + 4096 * (e / 4)) = make_uint4(0u, 0u, 0u, 0u);
    if (AUTO_H && PPT > 0) {
        __shared__ float box[6][16];
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int j = 0; j < PPT; ++j) {          // (clamped duplicates of point n - 1 do not change the box)
            lo[0] = fminf(lo[0], px[j]); hi[0] = fmaxf(hi[0], px[j]);
            lo[1] = fminf(lo[1], py[j]); hi[1] = fmaxf(hi[1], py[j]);
            lo[2] = fminf(lo[2], pz[j]); hi[2] = fmaxf(hi[2], pz[j]);
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float l = -wave_max_f32(-lo[a]), h = wave_max_f32(hi[a]);
            if (lane == 0) { box[a][wave] = l; box[3 + a][wave] = h; }
        }
        __syncthreads();
        float ext[3], amax = 0.f;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float l = box[a][0], h = box[3 + a][0];
            for (int w = 1; w < 16; ++w) { l = fminf(l, box[a][w]); h = fmaxf(h, box[3 + a][w]); }
            ext[a] = fmaxf(h - l, 0.f);
            amax = fmaxf(amax, fmaxf(fabsf(l), fabsf(h)));
        }
        const float fn = (float)n;
        float s = cbrtf(ext[0] * ext[1] * ext[2] / fn);
        s = fmaxf(s, sqrtf(fmaxf(fmaxf(ext[0] * ext[1], ext[0] * ext[2]), ext[1] * ext[2]) / fn));
        s = fmaxf(s, fmaxf(fmaxf(ext[0], ext[1]), ext[2]) / fn);
        float h = (inv_h > 0.f ? inv_h : 1.5f) * s;          // (AUTO_H: the inv_h argument carries the spacing factor, default 1.5)
        if (!(h > 0.f) || !(h < 1e30f)) h = 1.f;             // all points identical / non-finite coordinates: any edge works
        h = fmaxf(h, amax * 1e-5f);                          // keep cell coordinates well inside the int range
        inv_h = 1.f / h;
        if (tid == 0) hdr[blockIdx.x] = make_float4(inv_h, h, 0.f, 0.f);
    }
    __syncthreads();
    if (PPT > 0) {
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            pb[j] = bg_bucket(bg_cell(px[j], inv_h), bg_cell(py[j], inv_h), bg_cell(pz[j], inv_h), tmask);
            if (tid + 1024 * j < n) atomicAdd(&cnt[pb[j]], 1u);
        }
    } else {
        for (int k = tid; k < n; k += 1024) {
            const float x = pts[k * 3 + 0], y = pts[k * 3 + 1], z = pts[k * 3 + 2];
            atomicAdd(&cnt[bg_bucket(bg_cell(x, inv_h), bg_cell(y, inv_h), bg_cell(z, inv_h), tmask)], 1u);
        }
    }
    __syncthreads();
    unsigned c[TPT];
    unsigned local = 0;
#pragma unroll
    for (int j = 0; j < TPT; ++j) { c[j] = cnt[tid + 1024 * j]; local += c[j]; }
    const int incl = wave_incl_scan_i32_dpp((int)local);
    if (lane == 63) wave_tot[wave] = (unsigned)incl;
    __syncthreads();
    unsigned base = 0;
    for (int w = 0; w < wave; ++w) base += wave_tot[w];
    unsigned run = base + (unsigned)incl - local;
#pragma unroll
    for (int j = 0; j < TPT; ++j) {
        tb[tid + 1024 * j] = make_uint2(run, run + c[j]);
        cnt[tid + 1024 * j] = run;                      // scatter cursor
        run += c[j];
    }
    __syncthreads();
    if (PPT > 0) {
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const int k = tid + 1024 * j;
            if (k < n) {
                const unsigned slot = atomicAdd(&cnt[pb[j]], 1u);
                so[slot] = make_float4(px[j], py[j], pz[j], __int_as_float(k));
            }
        }
    } else {
        for (int k = tid; k < n; k += 1024) {
            const float x = pts[k * 3 + 0], y = pts[k * 3 + 1], z = pts[k * 3 + 2];
            const unsigned slot = atomicAdd(&cnt[bg_bucket(bg_cell(x, inv_h), bg_cell(y, inv_h), bg_cell(z, inv_h), tmask)], 1u);
            so[slot] = make_float4(x, y, z, __int_as_float(k));
        }
    }
}


}  // namespace jm
