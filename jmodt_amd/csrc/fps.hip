// fps.hip — furthest point sampling for gfx950.
//
// Replaces farthest_point_sampling_kernel<BS> (jmodt/ops/pointnet2/src/sampling_gpu.cu:93-253).
//
// Design (MI355X-first, not a translation):
//  * One workgroup per cloud, the whole cloud REGISTER-RESIDENT: each thread keeps its
//    J = ceil(n/BS) points (x,y,z) and their running min-distance `temp` in VGPRs for the
//    full m-iteration loop.  The reference re-reads 20·n bytes from global memory every
//    iteration; here global memory is touched once on entry and once on exit.
//  * Bit-exact tie order.  The reference's block tree (ties keep the lower slot, strides
//    BS/2..1) makes the winner among exactly tied maxima the point minimising
//    (bitreverse_{log2 BS}(k mod BS), k)  (SURVEY.md A.1).  Thread T of the workgroup plays
//    the reference thread t = bitreverse(T), so "lowest thread id wins" IS that order:
//    a wave arg-max is one DPP max + ballot + find-first-set, no index comparisons.
//  * One barrier per iteration: every wave publishes its candidate {d2, k, x, y, z} to a
//    parity-double-buffered LDS slot; after the barrier every wave reduces the <=16
//    candidates redundantly (DPP again) and reads the winner's coordinates with one
//    broadcast ds_read_b128 — the next centre never goes through global memory.
//  * Distances use d = fma(dz,dz, fma(dx,dx, dy*dy)) (oracle convention; library is built
//    with -ffp-contract=off so nothing else fuses).
#include "jm_common.h"

namespace jm {

__device__ __forceinline__ unsigned bitrev_u(unsigned v, int bits) { return __brev(v) >> (32 - bits); }

constexpr int FPS_OUT_CHUNK = 4096;

struct __attribute__((aligned(16))) FpsCand {
    int val;  // float bits of the best min-distance (>= 0) or of -1.0f (no valid point): signed-int order == float order
    int k;
    float x, y, z;
    int pad[3];
};

template <int J>
__global__ void __launch_bounds__(1024)
fps_regs_kernel(int n, int m, int bs, int bs_log2, const float* __restrict__ dataset, float* __restrict__ temp,
                int* __restrict__ idxs) {
    __shared__ FpsCand cand[2][16];
    __shared__ int out_buf[FPS_OUT_CHUNK];  // picks are staged here: a global store inside the loop would
                                            // make every __syncthreads() wait for its write-ack (vmcnt(0))
    const int T = threadIdx.x;
    const int nwaves = (blockDim.x + 63) >> 6;
    const int wave = T >> 6;
    const int lane = T & 63;
    const float* ds = dataset + (size_t)blockIdx.x * n * 3;
    float* tp = temp + (size_t)blockIdx.x * n;
    int* out = idxs + (size_t)blockIdx.x * m;

    // reference thread id this thread impersonates (T >= bs only when bs < 64: idle lanes)
    const bool live = T < bs;
    const int t = live ? (int)bitrev_u((unsigned)T, bs_log2) : 0;

    float px[J], py[J], pz[J], tm[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int k = t + bs * j;
        const bool ok = live && k < n;
        // padding slots: coordinates +inf (d = inf) and temp -1, so min(d, temp) stays -1 and
        // can never beat `best = -1` under strict >
        px[j] = ok ? ds[k * 3 + 0] : INFINITY;
        py[j] = ok ? ds[k * 3 + 1] : INFINITY;
        pz[j] = ok ? ds[k * 3 + 2] : INFINITY;
        tm[j] = ok ? tp[k] : -1.f;
    }

    float x1 = ds[0], y1 = ds[1], z1 = ds[2];
    if (T == 0) out_buf[0] = 0;

    for (int it = 1; it < m; ++it) {
        if ((it & (FPS_OUT_CHUNK - 1)) == 0) {  // flush a full chunk of picks (uniform branch)
            __syncthreads();
            for (int e = T; e < FPS_OUT_CHUNK; e += blockDim.x) out[it - FPS_OUT_CHUNK + e] = out_buf[e];
            __syncthreads();
        }
        float best = -1.f;
        int bj = 0;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const float d = sqdist3(px[j] - x1, py[j] - y1, pz[j] - z1);
            const float d2 = fast_min(d, tm[j]);
            tm[j] = d2;
            const bool gt = d2 > best;
            bj = gt ? j : bj;
            best = gt ? d2 : best;
        }
        // wave arg-max: lanes are in tie-priority order, so the first lane holding the max wins
        const int bits = __float_as_int(best);
        const int wmax = wave_max_i32(bits);
        const unsigned long long eq = __ballot(bits == wmax);
        const int wl = (int)__ffsll((long long)eq) - 1;
        const int bj_u = __builtin_amdgcn_readlane(bj, wl);
        int sx = 0, sy = 0, sz = 0;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            if (bj_u == j) {  // wave-uniform branch
                sx = __builtin_amdgcn_readlane(__float_as_int(px[j]), wl);
                sy = __builtin_amdgcn_readlane(__float_as_int(py[j]), wl);
                sz = __builtin_amdgcn_readlane(__float_as_int(pz[j]), wl);
            }
        }
        const int t_w = (int)bitrev_u((unsigned)((wave << 6) | wl), bs_log2);
        const int k_w = t_w + bs * bj_u;
        int old;
        if (nwaves == 1) {
            old = k_w;
            x1 = __int_as_float(sx); y1 = __int_as_float(sy); z1 = __int_as_float(sz);
        } else {
            FpsCand* slot = cand[it & 1];
            if (lane == 0) {
                FpsCand c;
                c.val = wmax; c.k = k_w;
                c.x = __int_as_float(sx); c.y = __int_as_float(sy); c.z = __int_as_float(sz);
                slot[wave] = c;
            }
            __syncthreads();
            const int v = lane < nwaves ? slot[lane].val : (int)0x80000000;
            const int gmax = wave_max_i32(v);
            const unsigned long long weq = __ballot(v == gmax);
            const int ww = (int)__ffsll((long long)weq) - 1;  // lowest wave index among ties
            const FpsCand c = slot[ww];                        // uniform address: LDS broadcast
            old = c.k; x1 = c.x; y1 = c.y; z1 = c.z;
        }
        if (T == 0) out_buf[it & (FPS_OUT_CHUNK - 1)] = old;
    }
    __syncthreads();
    {
        const int done = ((m - 1) / FPS_OUT_CHUNK) * FPS_OUT_CHUNK;  // picks [done, m) are still staged
        for (int e = T; e < m - done; e += blockDim.x) out[done + e] = out_buf[e];
    }

    // leave `temp` as the reference kernel does (it updates it in place every iteration)
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int k = t + bs * j;
        if (live && k < n) tp[k] = tm[j];
    }
}

// Fallback for clouds that do not fit the register file (n > 16*1024): temp in LDS is not
// possible either (n*4 B), so temp and xyz stream through global/L2 like the reference, but
// keeping the same wave-level arg-max machinery and tie order.
__global__ void __launch_bounds__(1024)
fps_stream_kernel(int n, int m, int bs_log2, const float* __restrict__ dataset, float* __restrict__ temp,
                  int* __restrict__ idxs) {
    __shared__ FpsCand cand[2][16];
    const int T = threadIdx.x;
    const int bs = blockDim.x;  // == reference BS (1024 for every n >= 1024)
    const int nwaves = bs >> 6;
    const int wave = T >> 6, lane = T & 63;
    const float* ds = dataset + (size_t)blockIdx.x * n * 3;
    float* tp = temp + (size_t)blockIdx.x * n;
    int* out = idxs + (size_t)blockIdx.x * m;
    const int t = (int)bitrev_u((unsigned)T, bs_log2);
    float x1 = ds[0], y1 = ds[1], z1 = ds[2];
    if (T == 0) out[0] = 0;
    for (int it = 1; it < m; ++it) {
        float best = -1.f;
        int bk = 0;
        for (int k = t; k < n; k += bs) {
            const float d = sqdist3(ds[k * 3 + 0] - x1, ds[k * 3 + 1] - y1, ds[k * 3 + 2] - z1);
            const float d2 = fminf(d, tp[k]);
            tp[k] = d2;
            const bool gt = d2 > best;
            bk = gt ? k : bk;
            best = gt ? d2 : best;
        }
        const int bits = __float_as_int(best);
        const int wmax = wave_max_i32(bits);
        const unsigned long long eq = __ballot(bits == wmax);
        const int wl = (int)__ffsll((long long)eq) - 1;
        const int k_w = __builtin_amdgcn_readlane(bk, wl);
        FpsCand* slot = cand[it & 1];
        if (lane == 0) {
            FpsCand c;
            c.val = wmax; c.k = k_w;
            c.x = ds[k_w * 3 + 0]; c.y = ds[k_w * 3 + 1]; c.z = ds[k_w * 3 + 2];
            slot[wave] = c;
        }
        __syncthreads();
        const int v = lane < nwaves ? slot[lane].val : (int)0x80000000;
        const int gmax = wave_max_i32(v);
        const unsigned long long weq = __ballot(v == gmax);
        const int ww = (int)__ffsll((long long)weq) - 1;
        const FpsCand c = slot[ww];
        x1 = c.x; y1 = c.y; z1 = c.z;
        if (T == 0) out[it] = c.k;
    }
}

static int opt_n_threads(int work_size) {
    // cuda_utils.h:10-14: 2^floor(log2 n) clamped to [1,1024].  Integer form (exact).
    int p = 0;
    while ((2 << p) <= work_size && p < 30) ++p;
    int v = 1 << p;
    if (work_size < 1) v = 1;
    return v > 1024 ? 1024 : v;
}

}  // namespace jm

extern "C" int jm_furthest_point_sampling(int b, int n, int m, const float* xyz, float* temp, int* idx,
                                          jm_stream_t stream) {
    using namespace jm;
    JM_REQUIRE(b >= 0 && n >= 1 && m >= 0, "fps: bad sizes b=%d n=%d m=%d", b, n, m);
    if (b == 0 || m == 0) return JM_OK;
    JM_REQUIRE(xyz && temp && idx, "fps: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const int bs = opt_n_threads(n);
    int bs_log2 = 0;
    while ((1 << bs_log2) < bs) ++bs_log2;
    const int J = divup(n, bs);
    const int block = bs < 64 ? 64 : bs;
#define JM_FPS_LAUNCH(JJ) \
    hipLaunchKernelGGL(fps_regs_kernel<JJ>, dim3(b), dim3(block), 0, s, n, m, bs, bs_log2, xyz, temp, idx)
    if (J <= 1) JM_FPS_LAUNCH(1);
    else if (J <= 2) JM_FPS_LAUNCH(2);
    else if (J <= 4) JM_FPS_LAUNCH(4);
    else if (J <= 8) JM_FPS_LAUNCH(8);
    else if (J <= 16) JM_FPS_LAUNCH(16);
    else hipLaunchKernelGGL(fps_stream_kernel, dim3(b), dim3(bs), 0, s, n, m, bs_log2, xyz, temp, idx);
#undef JM_FPS_LAUNCH
    return check_launch("fps");
}
