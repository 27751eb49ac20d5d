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
#include <stdlib.h>

#include "jm_common.h"

namespace jm {

typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned bitrev_u(unsigned v, int bits) { return __brev(v) >> (32 - bits); }

constexpr int FPS_OUT_CHUNK = 4096;

struct __attribute__((aligned(16))) FpsCand {
    int val;  // float bits of the best min-distance (>= 0) or of -1.0f (no valid point): signed-int order == float order
    int k;
    float x, y, z;
    int pad[3];
};

// PTS points per thread, held in registers for the whole loop.  The workgroup has BT = blockDim.x
// threads and impersonates the reference's BS = bs threads: thread T covers the R = bs / BT
// consecutive *priority* indices P = T*R + r (reference thread t = bitreverse(P)), and for each
// of them the J = ceil(n / bs) points k = t + bs*j.  Register slot i = r*J + j, visited in
// ascending i with a strict `>`, so inside a thread — and, because lanes and waves are ordered by
// P, across the whole workgroup — "first maximum wins" is exactly the reference's tie order.
// Fewer, fatter waves (BT < bs) pay the per-iteration reduce/publish/barrier overhead once per
// wave instead of once per 16 points.
template <int PTS, int MAXBT>
__global__ void __launch_bounds__(MAXBT)
fps_regs_kernel(int n, int m, int bs, int bs_log2, int R, int J, int recipJ, const float* __restrict__ dataset,
                float* __restrict__ temp, int* __restrict__ idxs) {
    __shared__ FpsCand cand[2][16];
    __shared__ int out_buf[FPS_OUT_CHUNK];  // picks are staged here: a global store inside the loop would
                                            // make every __syncthreads() wait for its write-ack (vmcnt(0))
    const int T = threadIdx.x;
    const int nwaves = (blockDim.x + 63) >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(T >> 6);
    const int lane = T & 63;
    const float* ds = dataset + (size_t)blockIdx.x * n * 3;
    float* tp = temp ? temp + (size_t)blockIdx.x * n : nullptr;   // null: start from 1e10, final distances not stored
    int* out = idxs + (size_t)blockIdx.x * m;

    float px[PTS], py[PTS], pz[PTS], tm[PTS];
#pragma unroll
    for (int i = 0; i < PTS; ++i) {
        const int r = (i * recipJ) >> 16, j = i - r * J;   // i / J, i % J for i < 64
        const int P = T * R + r;
        const int k = (int)bitrev_u((unsigned)P, bs_log2) + bs * j;
        const bool ok = r < R && P < bs && k < n;
        // padding slots: coordinates +inf (d = inf) and temp -1, so min(d, temp) stays -1 and
        // can never beat `best = -1` under strict >
        px[i] = ok ? ds[k * 3 + 0] : INFINITY;
        py[i] = ok ? ds[k * 3 + 1] : INFINITY;
        pz[i] = ok ? ds[k * 3 + 2] : INFINITY;
        tm[i] = ok ? (tp ? tp[k] : 1e10f) : -1.f;
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
        int bi = 0;
#pragma unroll
        for (int i = 0; i < PTS; ++i) {
            const float d = sqdist3(px[i] - x1, py[i] - y1, pz[i] - z1);
            const float d2 = fast_min(d, tm[i]);
            tm[i] = d2;
            const bool gt = d2 > best;
            bi = gt ? i : bi;
            best = gt ? d2 : best;
        }
        // wave arg-max: lanes are in tie-priority order, so the first lane holding the max wins
        const int bits = __float_as_int(best);
        const int wmax = wave_max_i32(bits);
        const unsigned long long eq = __ballot(bits == wmax);
        const int wl = (int)__ffsll((long long)eq) - 1;
        const int bi_u = __builtin_amdgcn_readlane(bi, wl);
        // wave-uniform dynamic index into the register arrays (s_set_gpr_idx_on + v_mov)
        const int sx = __builtin_amdgcn_readlane(__float_as_int(px[bi_u]), wl);
        const int sy = __builtin_amdgcn_readlane(__float_as_int(py[bi_u]), wl);
        const int sz = __builtin_amdgcn_readlane(__float_as_int(pz[bi_u]), wl);
        const int r_w = (bi_u * recipJ) >> 16, j_w = bi_u - r_w * J;
        const int k_w = (int)bitrev_u((unsigned)(((wave << 6) | wl) * R + r_w), bs_log2) + bs * j_w;
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
    for (int i = 0; i < PTS; ++i) {
        const int r = (i * recipJ) >> 16, j = i - r * J;
        const int P = T * R + r;
        const int k = (int)bitrev_u((unsigned)P, bs_log2) + bs * j;
        if (tp && r < R && P < bs && k < n) tp[k] = tm[i];
    }
}

// Multi-wave variant with the candidate extraction done by ONE wave per iteration.
//
// In fps_regs_kernel every wave tracks (best, slot) per point and extracts its own candidate
// {k, x, y, z} before the barrier, and every wave repeats the cross-wave reduce after it: with 16
// waves that is ~110 overhead instructions x 4 waves per SIMD on top of the distance loop.  Here
//   phase 1 (all waves)   distance update with a value-only running max (8 VALU per point:
//                         3 sub, mul, 2 fma, min, max), wave max by DPP, publish ONE dword;
//   barrier A
//   phase 2 (all waves)   reduce the <=16 wave maxima -> winning wave (lowest index among ties);
//   phase 3 (winner only) find the lane (first = highest priority) and the register slot (first
//                         slot equal to the max), fetch its coordinates by wave-uniform register
//                         indexing, publish {k, x, y, z};
//   barrier B
//   all waves read the 16-byte winner record.
// Two barriers instead of one, but ~2.5x fewer issued instructions per iteration.
template <int PTS, int MAXBT>
__global__ void __launch_bounds__(MAXBT)
fps_regs2_kernel(int n, int m, int bs, int bs_log2, int R, int J, int recipJ, const float* __restrict__ dataset,
                 float* __restrict__ temp, int* __restrict__ idxs) {
    __shared__ int vals[2][16];
    __shared__ FpsCand win[2];
    __shared__ int out_buf[FPS_OUT_CHUNK];
    const int T = threadIdx.x;
    const int nwaves = (blockDim.x + 63) >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(T >> 6);
    const int lane = T & 63;
    const float* ds = dataset + (size_t)blockIdx.x * n * 3;
    float* tp = temp ? temp + (size_t)blockIdx.x * n : nullptr;   // null: start from 1e10, final distances not stored
    int* out = idxs + (size_t)blockIdx.x * m;

    float px[PTS], py[PTS], pz[PTS], tm[PTS];
#pragma unroll
    for (int i = 0; i < PTS; ++i) {
        const int r = (i * recipJ) >> 16, j = i - r * J;
        const int P = T * R + r;
        const int k = (int)bitrev_u((unsigned)P, bs_log2) + bs * j;
        const bool ok = r < R && P < bs && k < n;
        px[i] = ok ? ds[k * 3 + 0] : INFINITY;
        py[i] = ok ? ds[k * 3 + 1] : INFINITY;
        pz[i] = ok ? ds[k * 3 + 2] : INFINITY;
        tm[i] = ok ? (tp ? tp[k] : 1e10f) : -1.f;
    }
    float x1 = ds[0], y1 = ds[1], z1 = ds[2];
    if (T == 0) out_buf[0] = 0;
    __syncthreads();

    for (int it = 1; it < m; ++it) {
        if ((it & (FPS_OUT_CHUNK - 1)) == 0) {
            __syncthreads();
            for (int e = T; e < FPS_OUT_CHUNK; e += blockDim.x) out[it - FPS_OUT_CHUNK + e] = out_buf[e];
            __syncthreads();
        }
        float best = -1.f;
        if constexpr (PTS % 2 == 0) {
            // packed fp32 (v_pk_add/mul/fma_f32): two points per instruction.  Every component is
            // the same IEEE operation sequence as the scalar path (sub, mul, fma, fma), so results
            // are bit-identical; a wave64 packed op costs ~4 issue cycles against ~3 for a plain
            // VALU op, i.e. 1.5x the distance throughput.
            const f32x2 cx = {x1, x1}, cy = {y1, y1}, cz = {z1, z1};
#pragma unroll
            for (int i = 0; i < PTS; i += 2) {
                const f32x2 vx = {px[i], px[i + 1]}, vy = {py[i], py[i + 1]}, vz = {pz[i], pz[i + 1]};
                const f32x2 dx = vx - cx, dy = vy - cy, dz = vz - cz;
                const f32x2 d = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dx, dx, dy * dy));
                const float a = fast_min(d[0], tm[i]), b = fast_min(d[1], tm[i + 1]);
                tm[i] = a; tm[i + 1] = b;
                best = fast_max3(best, a, b);
            }
        } else {
#pragma unroll
            for (int i = 0; i < PTS; ++i) {
                const float d = sqdist3(px[i] - x1, py[i] - y1, pz[i] - z1);
                const float d2 = fast_min(d, tm[i]);
                tm[i] = d2;
                best = fast_max(best, d2);
            }
        }
        const int bits = __float_as_int(best);
        const int wmax = wave_max_i32(bits);
        if (lane == 0) vals[it & 1][wave] = wmax;
        lds_barrier();                                                        // A
        const int v = lane < nwaves ? vals[it & 1][lane] : (int)0x80000000;
        const int gmax = wave_max_i32(v);
        const unsigned long long weq = __ballot(v == gmax);
        const int ww = (int)__ffsll((long long)weq) - 1;
        if (wave == ww) {   // wave-uniform: exactly one wave extracts the winner
            const unsigned long long eq = __ballot(bits == gmax);
            const int wl = (int)__ffsll((long long)eq) - 1;
            // (a scalar-unit variant — one ballot per slot, then bit tests of the winner lane — measured
            //  8 % slower per iteration: 1.25 vs 1.16 us)
            int bi = 0;
#pragma unroll
            for (int i = PTS - 1; i >= 0; --i) bi = (__float_as_int(tm[i]) == gmax) ? i : bi;
            const int bi_u = __builtin_amdgcn_readlane(bi, wl);
            // (a scalar switch over the slot with statically indexed v_readlanes instead of the three
            //  s_set_gpr_idx windows measured slower: 1.23 vs 1.16 us per iteration)
            const int sx = __builtin_amdgcn_readlane(__float_as_int(px[bi_u]), wl);
            const int sy = __builtin_amdgcn_readlane(__float_as_int(py[bi_u]), wl);
            const int sz = __builtin_amdgcn_readlane(__float_as_int(pz[bi_u]), wl);
            // (publishing the coordinates first and working the index out after barrier B measured slower too:
            //  1.176 vs 1.160 us per iteration)
            const int r_w = (bi_u * recipJ) >> 16, j_w = bi_u - r_w * J;
            const int k_w = (int)bitrev_u((unsigned)(((wave << 6) | wl) * R + r_w), bs_log2) + bs * j_w;
            if (lane == 0) {
                FpsCand c;
                c.val = gmax; c.k = k_w;
                c.x = __int_as_float(sx); c.y = __int_as_float(sy); c.z = __int_as_float(sz);
                win[it & 1] = c;
                out_buf[it & (FPS_OUT_CHUNK - 1)] = k_w;
            }
        }
        lds_barrier();                                                        // B
        const FpsCand c = win[it & 1];
        x1 = c.x; y1 = c.y; z1 = c.z;
    }
    __syncthreads();
    {
        const int done = ((m - 1) / FPS_OUT_CHUNK) * FPS_OUT_CHUNK;
        for (int e = T; e < m - done; e += blockDim.x) out[done + e] = out_buf[e];
    }
#pragma unroll
    for (int i = 0; i < PTS; ++i) {
        const int r = (i * recipJ) >> 16, j = i - r * J;
        const int P = T * R + r;
        const int k = (int)bitrev_u((unsigned)P, bs_log2) + bs * j;
        if (tp && r < R && P < bs && k < n) tp[k] = tm[i];
    }
}

// ------------------------------------------------------------------------------------------
// Spatially pruned exact FPS (n = 4096 / 8192 / 16384, reference block size 1024).
//
// After a few hundred picks the coverage radius is small and a new sample changes the
// min-distance of only the points near it.  Points are therefore laid out so that each of the 16
// waves owns a spatially compact cluster (Morton order in x,z) and keeps that cluster's bounding
// box; before its distance loop a wave evaluates the SAME distance expression on the box's
// nearest corner offsets, which — rounding being monotone — is a lower bound LB of the computed
// distance of every point in the box.  If LB >= the wave's current maximum min-distance, no
// temp in the wave can change (min(d, temp) = temp for all of them), so the wave skips the loop
// and republishes its cached candidate.  The result is bit-identical to the full scan.
//
// The reference tie order (min (bitreverse(k mod 1024), k / 1024) =: pk among tied maxima) no longer
// follows from thread order, so it is carried by the layout: inside a wave the points are sorted
// by pk (lane-major, then slot), so "first lane, first slot" is still the pk minimum; between
// waves a tie on the value is resolved by comparing the published pk (rare, uniform branch).
//
// Setup (once per cloud, inside the kernel): two bitonic sorts of (key, index) pairs in LDS —
// first by Morton code, then by (cluster, pk) — ~40 us for 16384 points against milliseconds of
// sampling.
__device__ __forceinline__ unsigned part1by1(unsigned v) {
    v &= 0xFFFFu;
    v = (v | (v << 8)) & 0x00FF00FFu;
    v = (v | (v << 4)) & 0x0F0F0F0Fu;
    v = (v | (v << 2)) & 0x33333333u;
    v = (v | (v << 1)) & 0x55555555u;
    return v;
}

__device__ __forceinline__ void bitonic_sort_u64(unsigned long long* e, int n, int T, int NT) {
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = T; t < n / 2; t += NT) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int p = i | j;
                const unsigned long long a = e[i], b = e[p];
                const bool up = (i & k) == 0;
                if ((a > b) == up) { e[i] = b; e[p] = a; }
            }
            __syncthreads();
        }
    }
}

struct __attribute__((aligned(16))) FpsCandP {
    int val;   // float bits of the wave's max min-distance
    int pk;    // tie priority of that point
    int k;
    float x, y, z;
    int pad[2];
};

template <int PTS>
__global__ void __launch_bounds__(1024)
fps_pruned_kernel(int n, int m, const float* __restrict__ dataset, float* __restrict__ temp,
                  int* __restrict__ idxs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    unsigned long long* ent = reinterpret_cast<unsigned long long*>(lds_raw);   // [n] during setup
    constexpr int J = PTS;                  // n / 1024
    constexpr int JLOG = PTS == 16 ? 4 : (PTS == 8 ? 3 : 2);
    const int T = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(T >> 6);
    const int lane = T & 63;
    const float* ds = dataset + (size_t)blockIdx.x * n * 3;
    float* tp = temp ? temp + (size_t)blockIdx.x * n : nullptr;   // null: start from 1e10, final distances not stored
    int* out = idxs + (size_t)blockIdx.x * m;
    __shared__ float red[4][16];

    // ---- cloud extent in x, z
    float xmin = INFINITY, xmax = -INFINITY, zmin = INFINITY, zmax = -INFINITY;
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int k = T + 1024 * j;
        const float x = ds[k * 3 + 0], z = ds[k * 3 + 2];
        xmin = fminf(xmin, x); xmax = fmaxf(xmax, x); zmin = fminf(zmin, z); zmax = fmaxf(zmax, z);
    }
    xmin = -wave_max_f32(-xmin); xmax = wave_max_f32(xmax); zmin = -wave_max_f32(-zmin); zmax = wave_max_f32(zmax);
    if (lane == 0) { red[0][wave] = xmin; red[1][wave] = xmax; red[2][wave] = zmin; red[3][wave] = zmax; }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        xmin = fminf(xmin, red[0][w]); xmax = fmaxf(xmax, red[1][w]);
        zmin = fminf(zmin, red[2][w]); zmax = fmaxf(zmax, red[3][w]);
    }
    const float sx = 65535.f / fmaxf(xmax - xmin, 1e-20f), sz = 65535.f / fmaxf(zmax - zmin, 1e-20f);
    // ---- sort 1: Morton order (any key gives a valid permutation; exactness never depends on it)
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int k = T + 1024 * j;
        const float qx = fminf(fmaxf((ds[k * 3 + 0] - xmin) * sx, 0.f), 65535.f);
        const float qz = fminf(fmaxf((ds[k * 3 + 2] - zmin) * sz, 0.f), 65535.f);
        const unsigned key = part1by1((unsigned)qx) | (part1by1((unsigned)qz) << 1);
        ent[k] = ((unsigned long long)key << 32) | (unsigned)k;
    }
    __syncthreads();
    bitonic_sort_u64(ent, n, T, 1024);
    // ---- sort 2: (cluster = rank / (64*PTS), pk) -> lane-major pk order inside each wave
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int r = T + 1024 * j;
        const unsigned k = (unsigned)ent[r];
        const unsigned cluster = (unsigned)r / (64u * PTS);
        const unsigned pk = (bitrev_u(k & 1023u, 10) << JLOG) | (k >> 10);
        ent[r] = ((unsigned long long)((cluster << 20) | pk) << 32) | k;   // own entries only: no race
    }
    __syncthreads();
    bitonic_sort_u64(ent, n, T, 1024);

    float px[PTS], py[PTS], pz[PTS], tm[PTS];
    const int rank0 = (wave * 64 + lane) * PTS;   // this thread owns sorted ranks rank0 .. rank0 + PTS - 1
    {
        int kk[PTS];
#pragma unroll
        for (int i = 0; i < PTS; ++i) {
            const int k = (int)(unsigned)ent[rank0 + i];
            kk[i] = k;
            px[i] = ds[k * 3 + 0]; py[i] = ds[k * 3 + 1]; pz[i] = ds[k * 3 + 2];
            tm[i] = tp ? tp[k] : 1e10f;
        }
        __syncthreads();   // every entry has been read: compact the permutation to int32 in place
        int* kidx_w = reinterpret_cast<int*>(lds_raw);
#pragma unroll
        for (int i = 0; i < PTS; ++i) kidx_w[rank0 + i] = kk[i];
    }
    const int* kidx = reinterpret_cast<const int*>(lds_raw);                                   // [n] rank -> point index
    FpsCandP* cand = reinterpret_cast<FpsCandP*>(lds_raw + (size_t)n * 4);                     // [2][16]
    int* out_buf = reinterpret_cast<int*>(lds_raw + (size_t)n * 4 + 2 * 16 * sizeof(FpsCandP));   // [FPS_OUT_CHUNK]

    // ---- wave bounding box (all lanes hold the same values)
    float bx0 = INFINITY, bx1 = -INFINITY, by0 = INFINITY, by1 = -INFINITY, bz0 = INFINITY, bz1 = -INFINITY;
#pragma unroll
    for (int i = 0; i < PTS; ++i) {
        bx0 = fminf(bx0, px[i]); bx1 = fmaxf(bx1, px[i]);
        by0 = fminf(by0, py[i]); by1 = fmaxf(by1, py[i]);
        bz0 = fminf(bz0, pz[i]); bz1 = fmaxf(bz1, pz[i]);
    }
    bx0 = -wave_max_f32(-bx0); bx1 = wave_max_f32(bx1);
    by0 = -wave_max_f32(-by0); by1 = wave_max_f32(by1);
    bz0 = -wave_max_f32(-bz0); bz1 = wave_max_f32(bz1);

    float x1 = ds[0], y1 = ds[1], z1 = ds[2];
    if (T == 0) out_buf[0] = 0;
    // cached candidate of this wave (wave-uniform); cval = +inf forces the first evaluation
    float cval = INFINITY;
    int c_pk = 0, c_k = 0, c_x = 0, c_y = 0, c_z = 0;
    __syncthreads();

    for (int it = 1; it < m; ++it) {
        if ((it & (FPS_OUT_CHUNK - 1)) == 0) {
            __syncthreads();
            for (int e = T; e < FPS_OUT_CHUNK; e += 1024) out[it - FPS_OUT_CHUNK + e] = out_buf[e];
            __syncthreads();
        }
        // lower bound of the computed distance over the wave's box (same expression, monotone rounding)
        const float ddx = fmaxf(fmaxf(bx0 - x1, x1 - bx1), 0.f);
        const float ddy = fmaxf(fmaxf(by0 - y1, y1 - by1), 0.f);
        const float ddz = fmaxf(fmaxf(bz0 - z1, z1 - bz1), 0.f);
        const float lb = sqdist3(ddx, ddy, ddz);
        if (__builtin_amdgcn_readfirstlane(lb >= cval ? 0 : 1)) {   // wave-uniform: some temp may change
            float best = -1.f;
            int bi = 0;
#pragma unroll
            for (int i = 0; i < PTS; ++i) {
                const float d = sqdist3(px[i] - x1, py[i] - y1, pz[i] - z1);
                const float d2 = fast_min(d, tm[i]);
                tm[i] = d2;
                const bool gt = d2 > best;
                bi = gt ? i : bi;
                best = gt ? d2 : best;
            }
            const int bits = __float_as_int(best);
            const int wmax = wave_max_i32(bits);
            const unsigned long long eq = __ballot(bits == wmax);
            const int wl = (int)__ffsll((long long)eq) - 1;     // lanes are in pk order inside the wave
            const int bi_u = __builtin_amdgcn_readlane(bi, wl);
            c_x = __builtin_amdgcn_readlane(__float_as_int(px[bi_u]), wl);
            c_y = __builtin_amdgcn_readlane(__float_as_int(py[bi_u]), wl);
            c_z = __builtin_amdgcn_readlane(__float_as_int(pz[bi_u]), wl);
            c_k = kidx[(wave * 64 + wl) * PTS + bi_u];   // wave-uniform LDS read
            c_pk = (int)((bitrev_u((unsigned)c_k & 1023u, 10) << JLOG) | ((unsigned)c_k >> 10));
            cval = __int_as_float(wmax);
        }
        FpsCandP* slot = cand + (it & 1) * 16;
        if (lane == 0) {
            FpsCandP c;
            c.val = __float_as_int(cval); c.pk = c_pk; c.k = c_k;
            c.x = __int_as_float(c_x); c.y = __int_as_float(c_y); c.z = __int_as_float(c_z);
            slot[wave] = c;
        }
        lds_barrier();
        const int v = lane < 16 ? slot[lane & 15].val : (int)0x80000000;
        const int gmax = wave_max_i32(v);
        unsigned long long weq = __ballot(lane < 16 && v == gmax);
        int ww = (int)__ffsll((long long)weq) - 1;
        if (__popcll(weq) > 1) {   // value tie between waves: the smaller pk wins (uniform, rare)
            const int pkv = ((weq >> lane) & 1ULL) ? slot[lane & 15].pk : 0x7FFFFFFF;
            const int pmin = -wave_max_i32(-pkv);
            ww = (int)__ffsll((long long)__ballot(pkv == pmin)) - 1;
        }
        const FpsCandP c = slot[ww];
        x1 = c.x; y1 = c.y; z1 = c.z;
        if (T == 0) out_buf[it & (FPS_OUT_CHUNK - 1)] = c.k;
    }
    __syncthreads();
    {
        const int done = ((m - 1) / FPS_OUT_CHUNK) * FPS_OUT_CHUNK;
        for (int e = T; e < m - done; e += 1024) out[done + e] = out_buf[e];
    }
#pragma unroll
    for (int i = 0; i < PTS; ++i) if (tp) tp[kidx[rank0 + i]] = tm[i];
}

// ------------------------------------------------------------------------------------------
// Slot-clustered exact FPS (n = 8192 / 16384, reference block size 1024): pruning that shortens EVERY
// wave's loop instead of skipping whole waves.
//
// The wave-clustered kernel above skips 85 % of the wave-loops and is still slower, because each
// iteration waits for the one wave that owns the neighbourhood of the new sample and runs its full
// 16-slot loop alone.  Here the spatial clusters are the register SLOTS: slot i of all 1024 threads
// holds the 1024 points of Morton cluster i, so an iteration touches the same few slots in every wave
// and the work stays balanced.  Per iteration a wave
//   1. evaluates the distance expression on the nearest corner offsets of the 16 slot boxes (lane i does
//      box i) — by monotone rounding a lower bound LB_i of the computed distance of every point of the
//      slot — and keeps the slots with LB_i < G, where G is the maximum min-distance before this update
//      (the value the previous iteration just selected, known to everybody for free): a slot with
//      LB_i >= G >= temp cannot change any temp;
//   2. updates only those slots (2.2 of 16 on average on the bench cloud, 3.2 on a 1/z-dense one);
//   3. recomputes its per-lane maximum over the 16 temps and enters the usual reduce / publish chain.
// Tie order: slots no longer follow the reference's thread order, so every slot carries the key
// (pk << 4 | slot), pk = (bitreverse10(k mod 1024), k / 1024) being the reference's priority; the
// winner is the minimum key among the (lane, slot) pairs that hold the maximum — inside the winning
// wave by a min-chain + one DPP min, between waves (value ties: duplicates, grids) by comparing the
// published pk.  The (slot, thread) -> point index table lives in LDS and is only read by the winner.
//
// Measured (MI355X, B = 8, 16384 -> 4096): 1.15 us / iteration on the bench cloud, 1.17 on a 1/z-dense
// one — the SAME as the plain scan (1.16): the distance loop shrinks from 72 to ~40 instructions per wave,
// but the iteration is dominated by the reduce / winner-extraction / barrier chain (≈1900 of 2770 cycles
// are barrier waits for the slowest wave and the winner), and the uniform branching costs what the
// skipped arithmetic saves.  Off by default (JM_FPS_PRUNE=2); bit-exact, covered by the GPU tests.
template <int PTS>
__global__ void __launch_bounds__(1024)
fps_slotprune_kernel(int n, int m, const float* __restrict__ dataset, float* __restrict__ temp,
                     int* __restrict__ idxs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    unsigned long long* ent = reinterpret_cast<unsigned long long*>(lds_raw);   // [n] during setup
    constexpr int J = PTS;                  // n / 1024 = number of clusters = slots per thread
    constexpr int JLOG = PTS == 16 ? 4 : 3;
    static_assert(PTS == 16 || PTS == 8, "slot-clustered FPS: 8 or 16 slots");
    const int T = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(T >> 6);
    const int lane = T & 63;
    const float* ds = dataset + (size_t)blockIdx.x * n * 3;
    float* tp = temp ? temp + (size_t)blockIdx.x * n : nullptr;   // null: start from 1e10, final distances not stored
    int* out = idxs + (size_t)blockIdx.x * m;
    __shared__ float red[6][16];

    // ---- cloud extent in x, z
    float xmin = INFINITY, xmax = -INFINITY, zmin = INFINITY, zmax = -INFINITY;
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int k = T + 1024 * j;
        const float x = ds[k * 3 + 0], z = ds[k * 3 + 2];
        xmin = fminf(xmin, x); xmax = fmaxf(xmax, x); zmin = fminf(zmin, z); zmax = fmaxf(zmax, z);
    }
    xmin = -wave_max_f32(-xmin); xmax = wave_max_f32(xmax); zmin = -wave_max_f32(-zmin); zmax = wave_max_f32(zmax);
    if (lane == 0) { red[0][wave] = xmin; red[1][wave] = xmax; red[2][wave] = zmin; red[3][wave] = zmax; }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        xmin = fminf(xmin, red[0][w]); xmax = fmaxf(xmax, red[1][w]);
        zmin = fminf(zmin, red[2][w]); zmax = fmaxf(zmax, red[3][w]);
    }
    const float sx = 65535.f / fmaxf(xmax - xmin, 1e-20f), sz = 65535.f / fmaxf(zmax - zmin, 1e-20f);
    // ---- sort 1: Morton order (any key gives a valid permutation; exactness never depends on it)
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int k = T + 1024 * j;
        const float qx = fminf(fmaxf((ds[k * 3 + 0] - xmin) * sx, 0.f), 65535.f);
        const float qz = fminf(fmaxf((ds[k * 3 + 2] - zmin) * sz, 0.f), 65535.f);
        const unsigned key = part1by1((unsigned)qx) | (part1by1((unsigned)qz) << 1);
        ent[k] = ((unsigned long long)key << 32) | (unsigned)k;
    }
    __syncthreads();
    bitonic_sort_u64(ent, n, T, 1024);
    // ---- sort 2: (cluster = rank / 1024, pk): thread T of slot i gets the T-th priority of cluster i
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int r = T + 1024 * j;
        const unsigned k = (unsigned)ent[r];
        const unsigned cluster = (unsigned)r >> 10;
        const unsigned pk = (bitrev_u(k & 1023u, 10) << JLOG) | (k >> 10);
        ent[r] = ((unsigned long long)((cluster << 20) | pk) << 32) | k;   // own entries only: no race
    }
    __syncthreads();
    bitonic_sort_u64(ent, n, T, 1024);

    float px[PTS], py[PTS], pz[PTS], tm[PTS];
    int* kidx = reinterpret_cast<int*>(lds_raw);            // [PTS][1024]: point index of (slot, thread)
    {
        int kk[PTS];
#pragma unroll
        for (int i = 0; i < PTS; ++i) {
            kk[i] = (int)(unsigned)ent[i * 1024 + T];
            px[i] = ds[kk[i] * 3 + 0]; py[i] = ds[kk[i] * 3 + 1]; pz[i] = ds[kk[i] * 3 + 2];
            tm[i] = tp ? tp[kk[i]] : 1e10f;
        }
        __syncthreads();   // every sorted entry has been read: compact the permutation to int32 in place
#pragma unroll
        for (int i = 0; i < PTS; ++i) kidx[i * 1024 + T] = kk[i];
    }
    unsigned char* after = lds_raw + (size_t)n * 4;

    // ---- slot bounding boxes: lane i (< PTS) of every wave ends up with box i; other lanes get an
    // empty box (LB = +inf, never active)
    float* bred = reinterpret_cast<float*>(after);          // [6][PTS][16 waves]
#pragma unroll
    for (int i = 0; i < PTS; ++i) {
        const float a0 = -wave_max_f32(-px[i]), a1 = wave_max_f32(px[i]);
        const float b0 = -wave_max_f32(-py[i]), b1 = wave_max_f32(py[i]);
        const float c0 = -wave_max_f32(-pz[i]), c1 = wave_max_f32(pz[i]);
        if (lane == 0) {
            bred[(0 * PTS + i) * 16 + wave] = a0; bred[(1 * PTS + i) * 16 + wave] = a1;
            bred[(2 * PTS + i) * 16 + wave] = b0; bred[(3 * PTS + i) * 16 + wave] = b1;
            bred[(4 * PTS + i) * 16 + wave] = c0; bred[(5 * PTS + i) * 16 + wave] = c1;
        }
    }
    __syncthreads();
    float blx = INFINITY, bhx = -INFINITY, bly = INFINITY, bhy = -INFINITY, blz = INFINITY, bhz = -INFINITY;
    if (lane < PTS) {
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            blx = fminf(blx, bred[(0 * PTS + lane) * 16 + w]); bhx = fmaxf(bhx, bred[(1 * PTS + lane) * 16 + w]);
            bly = fminf(bly, bred[(2 * PTS + lane) * 16 + w]); bhy = fmaxf(bhy, bred[(3 * PTS + lane) * 16 + w]);
            blz = fminf(blz, bred[(4 * PTS + lane) * 16 + w]); bhz = fmaxf(bhz, bred[(5 * PTS + lane) * 16 + w]);
        }
    }
    __syncthreads();
    int* vals = reinterpret_cast<int*>(after);                                     // [2][16]
    FpsCandP* cand = reinterpret_cast<FpsCandP*>(after + 256);                     // [2][16]
    int* out_buf = reinterpret_cast<int*>(after + 256 + 2 * 16 * sizeof(FpsCandP));   // [FPS_OUT_CHUNK]
    auto pk_of = [&](int k) { return (int)((bitrev_u((unsigned)k & 1023u, 10) << JLOG) | ((unsigned)k >> 10)); };

    float x1 = ds[0], y1 = ds[1], z1 = ds[2];
    float G = INFINITY;          // max min-distance before the coming update (all temps start at 1e10)
    float best = INFINITY;       // per-lane max over the slots; recomputed whenever a slot is touched
    if (T == 0) out_buf[0] = 0;
    __syncthreads();

    for (int it = 1; it < m; ++it) {
        if ((it & (FPS_OUT_CHUNK - 1)) == 0) {
            __syncthreads();
            for (int e = T; e < FPS_OUT_CHUNK; e += 1024) out[it - FPS_OUT_CHUNK + e] = out_buf[e];
            __syncthreads();
        }
        // 1. which slots can change?  (same expression as the distance, on the box's nearest offsets)
        const float ddx = fast_max3(blx - x1, x1 - bhx, 0.f);
        const float ddy = fast_max3(bly - y1, y1 - bhy, 0.f);
        const float ddz = fast_max3(blz - z1, z1 - bhz, 0.f);
        const float lb = sqdist3(ddx, ddy, ddz);
        const unsigned act = (unsigned)__ballot(lb < G) & ((1u << PTS) - 1u);   // wave-uniform
        // 2. update those slots only
        if (act) {
#pragma unroll
            for (int i = 0; i < PTS; ++i) {
                if (act & (1u << i)) {   // uniform branch, static registers
                    const float d = sqdist3(px[i] - x1, py[i] - y1, pz[i] - z1);
                    tm[i] = fast_min(d, tm[i]);
                }
            }
            // 3. per-lane maximum over all slots
            float b = tm[0];
#pragma unroll
            for (int i = 1; i + 1 < PTS; i += 2) b = fast_max3(b, tm[i], tm[i + 1]);
            best = fast_max(b, tm[PTS - 1]);
        }
        const int bits = __float_as_int(best);
        const int wmax = wave_max_i32(bits);
        if (lane == 0) vals[(it & 1) * 16 + wave] = wmax;
        lds_barrier();                                                        // A
        const int v = lane < 16 ? vals[(it & 1) * 16 + lane] : (int)0x80000000;
        const int gmax = wave_max_i32(v);
        const unsigned long long weq = __ballot(v == gmax);
        FpsCandP* slot_c = cand + (it & 1) * 16;
        if ((weq >> wave) & 1ULL) {   // wave-uniform: normally exactly one wave holds the maximum
            const unsigned long long eq = __ballot(bits == gmax);
            int wl = (int)__ffsll((long long)eq) - 1;
            // first matching slot and number of matching slots of every lane
            int bi = 0, cnt = 0;
#pragma unroll
            for (int i = PTS - 1; i >= 0; --i) {
                const bool mt = __float_as_int(tm[i]) == gmax;
                bi = mt ? i : bi;
                cnt += mt ? 1 : 0;
            }
            int si = __builtin_amdgcn_readlane(bi, wl);
            if (__popcll(eq) > 1 || __builtin_amdgcn_readlane(cnt, wl) > 1) {
                // several (lane, slot) pairs hold the maximum (duplicated or grid-aligned points): the
                // reference's priority decides — smallest pk, read from the index table
                int lmin = 0x7FFFFFFF;
#pragma unroll
                for (int i = 0; i < PTS; ++i)
                    if (__float_as_int(tm[i]) == gmax) lmin = min(lmin, (pk_of(kidx[i * 1024 + T]) << 4) | i);
                const int kmin = -wave_max_i32(-lmin);
                wl = (int)__ffsll((long long)__ballot(lmin == kmin)) - 1;
                si = kmin & 15;
            }
            const int k_w = kidx[si * 1024 + (wave << 6) + wl];   // wave-uniform LDS read
            const int cx = __builtin_amdgcn_readlane(__float_as_int(px[si]), wl);
            const int cy = __builtin_amdgcn_readlane(__float_as_int(py[si]), wl);
            const int cz = __builtin_amdgcn_readlane(__float_as_int(pz[si]), wl);
            if (lane == 0) {
                FpsCandP c;
                c.val = gmax; c.pk = pk_of(k_w); c.k = k_w;
                c.x = __int_as_float(cx); c.y = __int_as_float(cy); c.z = __int_as_float(cz);
                slot_c[wave] = c;
            }
        }
        lds_barrier();                                                        // B
        int ww = (int)__ffsll((long long)weq) - 1;
        if (__popcll(weq) > 1) {   // value tie between waves: the smaller pk wins (uniform, rare)
            const int pkv = ((weq >> lane) & 1ULL) ? slot_c[lane & 15].pk : 0x7FFFFFFF;
            const int pmin = -wave_max_i32(-pkv);
            ww = (int)__ffsll((long long)__ballot(pkv == pmin)) - 1;
        }
        const FpsCandP c = slot_c[ww];
        x1 = c.x; y1 = c.y; z1 = c.z;
        G = __int_as_float(c.val);
        if (T == 0) out_buf[it & (FPS_OUT_CHUNK - 1)] = c.k;
    }
    __syncthreads();
    {
        const int done = ((m - 1) / FPS_OUT_CHUNK) * FPS_OUT_CHUNK;
        for (int e = T; e < m - done; e += 1024) out[done + e] = out_buf[e];
    }
#pragma unroll
    for (int i = 0; i < PTS; ++i) if (tp) tp[kidx[i * 1024 + T]] = tm[i];
}

// ------------------------------------------------------------------------------------------
// Cooperative FPS for clouds larger than one register file (16384 < n <= 131072; config 5's
// 65536 points): the cloud is split over G = ceil(n / 16384) workgroups, each keeping its 16 slots x
// 1024 threads register-resident exactly like fps_regs2_kernel, and the G local candidates are
// exchanged through global memory once per iteration.
//
//   * Workgroup g owns the reference threads' slots j in [16 g, 16 g + 16), i.e. points
//     k in [16384 g, 16384 (g + 1)); inside it "lowest thread, lowest slot" is still the reference
//     tie order; between workgroups a tie on the value is resolved by the smaller
//     pk = (bitreverse10(k mod 1024), k / 1024).
//   * Exchange record = 5 self-validating 64-bit words {iteration | payload}: a reader issues the 5
//     agent-scope loads of a peer's record at once and accepts them when all carry the current
//     iteration number — one L2 round trip, no flag-then-data dependency, no fences.  Records are
//     double-buffered by iteration parity (a peer cannot get two iterations ahead: it needs this
//     workgroup's record of the next iteration first).  The buffer is zeroed before each launch.
//   * The G workgroups of a cloud get block ids that differ by multiples of 8, i.e. (as dispatched
//     today) the same XCD and the same L2; correctness does not depend on that.
//   * All workgroups of a launch must be co-resident (they spin on each other): the host launches at
//     most (#CUs / G) clouds per kernel, and callers must not overlap two cooperative FPS launches on
//     one device (the Python wrapper serialises them with an event).
// Measured: 65536 -> 4096, B = 8: see DESIGN.md §4 (the streaming fallback below takes 410 ms).
struct __attribute__((aligned(64))) FpsXchg { unsigned long long w[8]; };

__device__ __forceinline__ unsigned fps_pk(int k) {
    return (bitrev_u((unsigned)k & 1023u, 10) << 8) | ((unsigned)k >> 10);   // k / 1024 < 256
}

__global__ void __launch_bounds__(1024)
fps_coop_kernel(int n, int m, int G, int nclouds, const float* __restrict__ dataset, float* __restrict__ temp,
                int* __restrict__ idxs, FpsXchg* __restrict__ xchg) {
    constexpr int PTS = 16;
    __shared__ int vals[2][16];
    __shared__ FpsCand win[2];
    __shared__ int out_buf[FPS_OUT_CHUNK];
    const int T = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(T >> 6);
    const int lane = T & 63;
    // block -> (cloud, part): parts of one cloud are 8 block ids apart
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int cloud = xcd + 8 * (slot / G), g = slot % G;
    if (cloud >= nclouds) return;
    const float* ds = dataset + (size_t)cloud * n * 3;
    float* tp = temp ? temp + (size_t)cloud * n : nullptr;
    int* out = idxs + (size_t)cloud * m;
    FpsXchg* xc = xchg + (size_t)cloud * 2 * G;          // [parity][part]

    const int kT = (int)bitrev_u((unsigned)T, 10) + 1024 * PTS * g;   // this thread's first point
    float px[PTS], py[PTS], pz[PTS], tm[PTS];
#pragma unroll
    for (int i = 0; i < PTS; ++i) {
        const int k = kT + 1024 * i;
        const bool ok = k < n;
        px[i] = ok ? ds[k * 3 + 0] : INFINITY;
        py[i] = ok ? ds[k * 3 + 1] : INFINITY;
        pz[i] = ok ? ds[k * 3 + 2] : INFINITY;
        tm[i] = ok ? (tp ? tp[k] : 1e10f) : -1.f;
    }
    float x1 = ds[0], y1 = ds[1], z1 = ds[2];
    if (T == 0) out_buf[0] = 0;
    __syncthreads();

    for (int it = 1; it < m; ++it) {
        if (g == 0 && (it & (FPS_OUT_CHUNK - 1)) == 0) {
            __syncthreads();
            for (int e = T; e < FPS_OUT_CHUNK; e += 1024) out[it - FPS_OUT_CHUNK + e] = out_buf[e];
            __syncthreads();
        }
        float best = -1.f;
        const f32x2 cx = {x1, x1}, cy = {y1, y1}, cz = {z1, z1};
#pragma unroll
        for (int i = 0; i < PTS; i += 2) {
            const f32x2 vx = {px[i], px[i + 1]}, vy = {py[i], py[i + 1]}, vz = {pz[i], pz[i + 1]};
            const f32x2 dx = vx - cx, dy = vy - cy, dz = vz - cz;
            const f32x2 d = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dx, dx, dy * dy));
            const float a = fast_min(d[0], tm[i]), b = fast_min(d[1], tm[i + 1]);
            tm[i] = a; tm[i + 1] = b;
            best = fast_max3(best, a, b);
        }
        const int bits = __float_as_int(best);
        const int wmax = wave_max_i32(bits);
        if (lane == 0) vals[it & 1][wave] = wmax;
        lds_barrier();                                                        // A
        const int v = lane < 16 ? vals[it & 1][lane] : (int)0x80000000;
        const int lmax = wave_max_i32(v);
        const int ww = (int)__ffsll((long long)__ballot(v == lmax)) - 1;
        FpsXchg* mine = xc + (it & 1) * G + g;
        if (wave == ww) {   // wave-uniform: this workgroup's candidate goes straight to the exchange record
            const int wl = (int)__ffsll((long long)__ballot(bits == lmax)) - 1;
            int bi = 0;
#pragma unroll
            for (int i = PTS - 1; i >= 0; --i) bi = (__float_as_int(tm[i]) == lmax) ? i : bi;
            const int bi_u = __builtin_amdgcn_readlane(bi, wl);
            const unsigned sx = (unsigned)__builtin_amdgcn_readlane(__float_as_int(px[bi_u]), wl);
            const unsigned sy = (unsigned)__builtin_amdgcn_readlane(__float_as_int(py[bi_u]), wl);
            const unsigned sz = (unsigned)__builtin_amdgcn_readlane(__float_as_int(pz[bi_u]), wl);
            const unsigned k_w = bitrev_u((unsigned)((wave << 6) | wl), 10) + 1024u * (unsigned)(PTS * g + bi_u);
            if (lane < 5) {   // five lanes, one 64-bit word each: {iteration | payload}
                const unsigned pay = lane == 0 ? (unsigned)lmax : lane == 1 ? k_w : lane == 2 ? sx : lane == 3 ? sy : sz;
                __hip_atomic_store(&mine->w[lane], ((unsigned long long)(unsigned)it << 32) | pay, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (wave == 0) {    // poll the G records of this iteration (lane q reads part q), pick the global winner
            const int q = lane < G ? lane : 0;
            const FpsXchg* rec = xc + (it & 1) * G + q;
            unsigned long long w0, w1, w2, w3, w4;
            int spins = 0;
            for (;;) {
                w0 = __hip_atomic_load(&rec->w[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                w1 = __hip_atomic_load(&rec->w[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                w2 = __hip_atomic_load(&rec->w[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                w3 = __hip_atomic_load(&rec->w[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                w4 = __hip_atomic_load(&rec->w[4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned tag = (unsigned)it;
                const bool okk = (unsigned)(w0 >> 32) == tag && (unsigned)(w1 >> 32) == tag && (unsigned)(w2 >> 32) == tag &&
                                 (unsigned)(w3 >> 32) == tag && (unsigned)(w4 >> 32) == tag;
                if (__all(okk) || ++spins > (1 << 21)) break;   // bounded (~seconds): a lost peer must not hang the GPU
            }
            const int cv = lane < G ? (int)(unsigned)w0 : (int)0x80000000;
            const int ck = (int)(unsigned)w1;
            const int gmax = wave_max_i32(cv);
            const int npk = (lane < G && cv == gmax) ? -(int)fps_pk(ck) : (int)0x80000000;   // smaller pk wins
            const int pbest = wave_max_i32(npk);
            const int wq = (int)__ffsll((long long)__ballot(npk == pbest)) - 1;
            const int k_g = __builtin_amdgcn_readlane(ck, wq);
            const int gx = __builtin_amdgcn_readlane((int)(unsigned)w2, wq);
            const int gy = __builtin_amdgcn_readlane((int)(unsigned)w3, wq);
            const int gz = __builtin_amdgcn_readlane((int)(unsigned)w4, wq);
            if (lane == 0) {
                FpsCand c;
                c.val = gmax; c.k = k_g;
                c.x = __int_as_float(gx); c.y = __int_as_float(gy); c.z = __int_as_float(gz);
                win[it & 1] = c;
                out_buf[it & (FPS_OUT_CHUNK - 1)] = k_g;
            }
        }
        lds_barrier();                                                        // B
        const FpsCand c = win[it & 1];
        x1 = c.x; y1 = c.y; z1 = c.z;
    }
    __syncthreads();
    if (g == 0) {
        const int done = ((m - 1) / FPS_OUT_CHUNK) * FPS_OUT_CHUNK;
        for (int e = T; e < m - done; e += 1024) out[done + e] = out_buf[e];
    }
#pragma unroll
    for (int i = 0; i < PTS; ++i) {
        const int k = kT + 1024 * i;
        if (tp && k < n) tp[k] = tm[i];
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
    float* tp = temp ? temp + (size_t)blockIdx.x * n : nullptr;   // null: start from 1e10, final distances not stored
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

// new_xyz[b, j, :] = xyz[b, idx[b, j], :]: the centres every caller gathers right after sampling
// (pointnet2_modules.py:36-39 does it with two transposes around gather_operation)
__global__ void fps_gather_xyz_kernel(int n, int m, long long total, const float* __restrict__ xyz,
                                      const int* __restrict__ idx, float* __restrict__ new_xyz) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const long long b = e / m;
    const float* q = xyz + ((size_t)b * n + idx[e]) * 3;
    float* o = new_xyz + e * 3;
    o[0] = q[0]; o[1] = q[1]; o[2] = q[2];
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

constexpr int FPS_COOP_MAX_N = 131072;

extern "C" size_t jm_fps_workspace_bytes(int b, int n) {
    if (b < 1 || n <= 16 * 1024 || n > FPS_COOP_MAX_N) return 0;
    const int G = (n + 16383) / 16384;
    return (size_t)b * 2 * G * sizeof(jm::FpsXchg);
}

static int fps_impl(int b, int n, int m, const float* xyz, float* temp, int* idx, void* ws, size_t ws_bytes,
                    jm_stream_t stream);

extern "C" int jm_furthest_point_sampling(int b, int n, int m, const float* xyz, float* temp, int* idx,
                                          jm_stream_t stream) {
    return fps_impl(b, n, m, xyz, temp, idx, nullptr, 0, stream);
}

extern "C" int jm_furthest_point_sampling_ws(int b, int n, int m, const float* xyz, float* temp, int* idx, void* ws,
                                             size_t ws_bytes, jm_stream_t stream) {
    return fps_impl(b, n, m, xyz, temp, idx, ws, ws_bytes, stream);
}

extern "C" int jm_furthest_point_sampling_xyz(int b, int n, int m, const float* xyz, float* temp, int* idx,
                                              float* new_xyz, void* ws, size_t ws_bytes, jm_stream_t stream) {
    int rc = fps_impl(b, n, m, xyz, temp, idx, ws, ws_bytes, stream);
    if (rc || b == 0 || m == 0) return rc;
    JM_REQUIRE(new_xyz, "fps_xyz: null new_xyz");
    const long long total = (long long)b * m;
    hipLaunchKernelGGL(jm::fps_gather_xyz_kernel, dim3((unsigned)jm::divup(total, 256LL)), dim3(256), 0, (hipStream_t)stream,
                       n, m, total, xyz, idx, new_xyz);
    return jm::check_launch("fps_xyz(gather)");
}

static int fps_impl(int b, int n, int m, const float* xyz, float* temp, int* idx, void* ws, size_t ws_bytes,
                    jm_stream_t stream) {
    using namespace jm;
    JM_REQUIRE(b >= 0 && n >= 1 && m >= 0, "fps: bad sizes b=%d n=%d m=%d", b, n, m);
    if (b == 0 || m == 0) return JM_OK;
    JM_REQUIRE(xyz && idx, "fps: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const int bs = opt_n_threads(n);
    int bs_log2 = 0;
    while ((1 << bs_log2) < bs) ++bs_log2;
    const int J = divup(n, bs);
    if (J > 32 || (J > 16 && bs > 512)) {  // does not fit one workgroup's register file
        const size_t need = jm_fps_workspace_bytes(b, n);
        static const int no_coop = getenv("JM_FPS_NO_COOP") ? atoi(getenv("JM_FPS_NO_COOP")) : 0;
        if (ws && need && !no_coop) {
            // cooperative: G workgroups per cloud, exchange records in the caller's workspace
            if (ws_bytes < need) { set_error("fps: workspace %zu < %zu bytes", ws_bytes, need); return JM_EWORKSPACE; }
            JM_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 63u) == 0, "fps: workspace must be 64-byte aligned");
            const int G = (n + 16383) / 16384;
            int dev = 0, cus = 256;
            (void)hipGetDevice(&dev);
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8 * G) cus = 256;
            const int per_launch = 8 * (cus / 8 / G);          // clouds per launch: all workgroups co-resident
            JM_REQUIRE(per_launch >= 8, "fps: device too small for the cooperative kernel");
            (void)hipMemsetAsync(ws, 0, need, s);
            for (int c0 = 0; c0 < b; c0 += per_launch) {
                const int nc = b - c0 < per_launch ? b - c0 : per_launch;
                const int grid = 8 * ((nc + 7) / 8) * G;
                hipLaunchKernelGGL(fps_coop_kernel, dim3(grid), dim3(1024), 0, s, n, m, G, nc, xyz + (size_t)c0 * n * 3,
                                   temp ? temp + (size_t)c0 * n : nullptr, idx + (size_t)c0 * m,
                                   reinterpret_cast<FpsXchg*>(ws) + (size_t)c0 * 2 * G);
            }
            return check_launch("fps(coop)");
        }
        // no workspace (legacy 7-argument entry) or n beyond the cooperative limit: stream from L2
        JM_REQUIRE(temp, "fps: the streaming kernel needs the temp buffer");
        hipLaunchKernelGGL(fps_stream_kernel, dim3(b), dim3(bs), 0, s, n, m, bs_log2, xyz, temp, idx);
        return check_launch("fps(stream)");
    }
    // Spatially pruned kernel: OFF by default.  Measured on MI355X (B=8, 16384 -> 4096): the skip
    // test removes 85 % of the wave-level distance loops (76.6 k of 524 k wave-iterations stay
    // active), yet the kernel takes 5.68 ms against 4.74 ms for the full scan: an iteration is bound
    // by the argmax / publish / barrier latency chain, and the few active waves run their loop
    // latency-bound (one wave per SIMD) in about the time four interleaved waves need for the full
    // scan.  Kept (bit-exact, covered by the GPU tests through JM_FPS_PRUNE=1) as the starting
    // point for a slot-interleaved layout; see DESIGN.md §7.
    static const int prune = getenv("JM_FPS_PRUNE") ? atoi(getenv("JM_FPS_PRUNE")) : 0;
    if (prune == 2 && bs == 1024 && n % 1024 == 0 && (J == 8 || J == 16) && m > 1) {
        const size_t lds = (size_t)n * 8;   // sort entries; later: index table (n * 4) + boxes / candidates / picks (< n * 4)
        if (J == 16) {
            (void)hipFuncSetAttribute((const void*)fps_slotprune_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL((fps_slotprune_kernel<16>), dim3(b), dim3(1024), lds, s, n, m, xyz, temp, idx);
        } else {
            (void)hipFuncSetAttribute((const void*)fps_slotprune_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL((fps_slotprune_kernel<8>), dim3(b), dim3(1024), lds, s, n, m, xyz, temp, idx);
        }
        return check_launch("fps(slot-pruned)");
    }
    if (prune == 1 && bs == 1024 && n % 1024 == 0 && (J == 4 || J == 8 || J == 16) && m > 1) {
        const size_t need = (size_t)n * 8;   // sort buffer; the loop uses n*4 (rank -> index) + candidates + staged picks
        const size_t loop_lds = (size_t)n * 4 + 2 * 16 * sizeof(FpsCandP) + FPS_OUT_CHUNK * sizeof(int);
        const size_t lds = need > loop_lds ? need : loop_lds;
#define JM_FPS_PRUNED(P)                                                                                    \
    do {                                                                                                    \
        if (lds > 64 * 1024)                                                                                \
            (void)hipFuncSetAttribute((const void*)fps_pruned_kernel<P>,                                     \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                \
        hipLaunchKernelGGL((fps_pruned_kernel<P>), dim3(b), dim3(1024), lds, s, n, m, xyz, temp, idx);        \
    } while (0)
        if (J == 16) JM_FPS_PRUNED(16);
        else if (J == 8) JM_FPS_PRUNED(8);
        else JM_FPS_PRUNED(4);
#undef JM_FPS_PRUNED
        return check_launch("fps(pruned)");
    }
    // threads per workgroup: points-per-thread target from a measured table,
    // overridable for experiments with JM_FPS_PTS
    // measured on MI355X (tools/fps_sweep.py, us per iteration):
    //   n = 16384: 16 waves x 16 pts 1.25 | 8 x 32: 1.32        n = 4096: 16 waves x 4 pts 0.68 | 4 x 16: 0.81
    //   n = 1024 : 16 waves x 1 pt 0.52 | 1 wave x 16: 0.56     n = 256 : 4 waves x 1 pt 0.51 | 1 wave x 4: 0.41
    //   1024 clouds of n = 512: 8 waves x 1 pt 0.77 | 1 wave x 8 pts 0.54
    // -> a full 1024-thread workgroup (one reference thread per thread) when the reference block
    //    is 1024 wide, otherwise ONE wave per cloud (no barrier, no LDS).
    int want_pts = bs >= 1024 ? J : 16;
    if (const char* e = getenv("JM_FPS_PTS")) want_pts = atoi(e);
    if (want_pts < J) want_pts = J;
    if (want_pts > 32) want_pts = 32;
    int R = 1;
    while (R * 2 * J <= want_pts && bs / (R * 2) >= 64) R *= 2;
    int block = bs / R;
    if (block < 64) block = 64;
    const int pts = R * J;
    const int recipJ = 65536 / J + 1;
    // multi-wave workgroups use the two-barrier kernel (one wave extracts the winner)
    static const int force_v1 = getenv("JM_FPS_V1") ? atoi(getenv("JM_FPS_V1")) : 0;
    const bool v2 = block > 64 && !force_v1;
#define JM_FPS_LAUNCH(P, MB)                                                                                    \
    do {                                                                                                        \
        if (v2) hipLaunchKernelGGL((fps_regs2_kernel<P, MB>), dim3(b), dim3(block), 0, s, n, m, bs, bs_log2, R, \
                                   J, recipJ, xyz, temp, idx);                                                  \
        else hipLaunchKernelGGL((fps_regs_kernel<P, MB>), dim3(b), dim3(block), 0, s, n, m, bs, bs_log2, R, J,  \
                                recipJ, xyz, temp, idx);                                                        \
    } while (0)
    if (pts <= 1) JM_FPS_LAUNCH(1, 1024);
    else if (pts <= 2) JM_FPS_LAUNCH(2, 1024);
    else if (pts <= 4) JM_FPS_LAUNCH(4, 1024);
    else if (pts <= 8) JM_FPS_LAUNCH(8, 1024);
    else if (pts <= 16) JM_FPS_LAUNCH(16, 1024);
    else if (block <= 512) JM_FPS_LAUNCH(32, 512);
    else { set_error("fps: internal geometry error (pts=%d block=%d)", pts, block); return JM_EINVAL; }
#undef JM_FPS_LAUNCH
    return check_launch("fps");
}
