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

#include "fps_common.h"

namespace jm {

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
//   phase 2 (all waves)   reduce the <=16 wave maxima -> winning wave (lowest index among ties): one ds_max_u64 per wave on a
//                         (value, 15 - wave) key before barrier A and one broadcast read after it (ATOM; the 16-word
//                         exchange + a second DPP reduction in every wave measured 0.5 - 4 % slower per iteration);
//   phase 3 (winner only) find the lane (first = highest priority) and the register slot (first
//                         slot equal to the max), fetch its coordinates by wave-uniform register
//                         indexing, publish {k, x, y, z};
//   barrier B
//   all waves read the 16-byte winner record.
// Two barriers instead of one, but ~2.5x fewer issued instructions per iteration.
template <int PTS, int MAXBT, bool ATOM = false>
__global__ void __launch_bounds__(MAXBT)
fps_regs2_kernel(int n, int m, int bs, int bs_log2, int R, int J, int recipJ, const float* __restrict__ dataset,
                 float* __restrict__ temp, int* __restrict__ idxs) {
    __shared__ int vals[2][16];
    __shared__ unsigned long long amax[2];   // ATOM: max over the waves of (value bits << 32 | 15 - wave) by ds_max_u64
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
    if (T == 0) { out_buf[0] = 0; amax[0] = 0ULL; amax[1] = 0ULL; }
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
        int gmax, ww;
        if constexpr (ATOM) {
            // one LDS atomic per wave instead of 16 words + a second DPP reduction in every wave: the key orders by value,
            // then by LOWER wave index (ties); values are >= 0 or -1.0f, whose bit patterns order like unsigned after the flip
            const unsigned key_hi = (unsigned)wmax ^ 0x80000000u;             // signed-int order -> unsigned order
            if (lane == 0) atomicMax(&amax[it & 1], ((unsigned long long)key_hi << 32) | (unsigned)(15 - wave));
            lds_barrier();                                                    // A
            const unsigned long long k = amax[it & 1];
            gmax = (int)((unsigned)(k >> 32) ^ 0x80000000u);
            ww = 15 - (int)(k & 15ULL);
        } else {
            if (lane == 0) vals[it & 1][wave] = wmax;
            lds_barrier();                                                    // A
            const int v = lane < nwaves ? vals[it & 1][lane] : (int)0x80000000;
            gmax = wave_max_i32(v);
            const unsigned long long weq = __ballot(v == gmax);
            ww = (int)__ffsll((long long)weq) - 1;
        }
        if (wave == ww) {   // wave-uniform: exactly one wave extracts the winner
            if (ATOM && lane == 0) amax[(it + 1) & 1] = 0ULL;                 // every wave has read it (before barrier B of it - 1)
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
    __shared__ int failed;
    const int T = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(T >> 6);
    const int lane = T & 63;
    if (T == 0) failed = 0;
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
                if (__all(okk)) break;
                // bounded (~seconds): a peer that never publishes (the host sizes every launch to be co-resident, so
                // this means a broken device partition, or a caller that overlapped the launch with work that keeps the
                // peers off the machine for seconds) must neither hang the GPU, nor kill the HIP context, nor be papered
                // over with stale records: every workgroup of the cloud gives up (its peers time out the same way) and
                // the cloud's whole index row is written as -1 — a valid row always starts with index 0
                if (++spins > (1 << 21)) { if (lane == 0) failed = 1; break; }
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
        if (failed) break;                                                    // uniform: read behind the barrier
        const FpsCand c = win[it & 1];
        x1 = c.x; y1 = c.y; z1 = c.z;
    }
    __syncthreads();
    if (failed) {                                  // exchange timed out: poison the row (jm_furthest_point_sampling*: idx[b][0] == -1)
        if (g == 0)
            for (int e = T; e < m; e += 1024) out[e] = -1;
        return;
    }
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
    const int i = idx[e];
    float* o = new_xyz + e * 3;
    if (i < 0) { o[0] = o[1] = o[2] = __builtin_nanf(""); return; }   // a row the cooperative kernel gave up on (-1)
    const float* q = xyz + ((size_t)b * n + i) * 3;
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
        static const int no_coop = tune_env("JM_FPS_NO_COOP", 0);
        if (ws && need && !no_coop) {
            // cooperative: G workgroups per cloud, exchange records in the caller's workspace
            if (ws_bytes < need) { set_error("fps: workspace %zu < %zu bytes", ws_bytes, need); return JM_EWORKSPACE; }
            JM_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 63u) == 0, "fps: workspace must be 64-byte aligned");
            const int G = (n + 16383) / 16384;
            // every workgroup of a launch must be resident at once (they wait for each other): size the launch from
            // the CU count the runtime reports — a partitioned / masked device with fewer than 8 G CUs cannot host
            // even one round of 8 clouds and takes the streaming kernel instead of spinning on peers that never run
            int dev = 0, cus = 0;
            (void)hipGetDevice(&dev);
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
            const int per_launch = 8 * (cus / 8 / G);          // clouds per launch: all workgroups co-resident
            if (per_launch < 8) {
                JM_REQUIRE(temp, "fps: %d CUs cannot host the cooperative kernel (needs %d) and the streaming kernel needs "
                                 "the temp buffer", cus, 8 * G);
                hipLaunchKernelGGL(fps_stream_kernel, dim3(b), dim3(bs), 0, s, n, m, bs_log2, xyz, temp, idx);
                return check_launch("fps(stream)");
            }
            (void)jm_zero_async(ws, need, s);
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
#ifdef JM_TOOLS_BUILD
    // exact spatially pruned variants (tools/csrc/fps_pruned.hip, tools build only): measured no faster
    static const int prune = tune_env("JM_FPS_PRUNE", 0);
    if (prune && bs == 1024 && temp && launch_fps_pruned(prune, b, n, m, xyz, temp, idx, s)) return check_launch("fps(pruned)");
#endif
    // threads per workgroup: points-per-thread target from a measured table,
    // overridable for experiments with JM_FPS_PTS
    // measured on MI355X (tools/fps_sweep.py, us per iteration):
    //   n = 16384: 16 waves x 16 pts 1.25 | 8 x 32: 1.32        n = 4096: 16 waves x 4 pts 0.68 | 4 x 16: 0.81
    //   n = 1024 : 16 waves x 1 pt 0.52 | 1 wave x 16: 0.56     n = 256 : 4 waves x 1 pt 0.51 | 1 wave x 4: 0.41
    //   1024 clouds of n = 512: 8 waves x 1 pt 0.77 | 1 wave x 8 pts 0.54
    // -> a full 1024-thread workgroup (one reference thread per thread) when the reference block
    //    is 1024 wide, otherwise ONE wave per cloud (no barrier, no LDS).
    int want_pts = bs >= 1024 ? J : 16;
    want_pts = tune_env("JM_FPS_PTS", want_pts);
    if (want_pts < J) want_pts = J;
    if (want_pts > 32) want_pts = 32;
    int R = 1;
    while (R * 2 * J <= want_pts && bs / (R * 2) >= 64) R *= 2;
    int block = bs / R;
    if (block < 64) block = 64;
    const int pts = R * J;
    const int recipJ = 65536 / J + 1;
    // multi-wave workgroups use the two-barrier kernel (one wave extracts the winner)
    static const int force_v1 = tune_env("JM_FPS_V1", 0);
    const bool v2 = block > 64 && !force_v1;
    static const int atom = tune_env("JM_FPS_ATOM", 1);   // cross-wave arg-max through ds_max_u64 (-0.5 .. -4 % per iteration, bit-exact)
#define JM_FPS_LAUNCH(P, MB)                                                                                    \
    do {                                                                                                        \
        if (v2 && atom) hipLaunchKernelGGL((fps_regs2_kernel<P, MB, true>), dim3(b), dim3(block), 0, s, n, m, bs, bs_log2, R, \
                                   J, recipJ, xyz, temp, idx);                                                  \
        else if (v2) hipLaunchKernelGGL((fps_regs2_kernel<P, MB>), dim3(b), dim3(block), 0, s, n, m, bs, bs_log2, R, \
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
