// fps_pruned.hip — exact, spatially pruned furthest point sampling: two experiments kept for reference.
//
// Both are bit-exact against the plain scan (tests/test_gpu_parity.py::test_fps_pruned_variants_bit_exact)
// and neither is faster on MI355X, because an FPS iteration is bound by the reduce / winner-extraction /
// barrier chain rather than by distance arithmetic (numbers in each kernel's header and in DESIGN.md §4).
// They are off by default: JM_FPS_PRUNE=1 (wave clusters) or =2 (slot clusters).
#include "fps_common.h"

namespace jm {

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
//
// Measured (MI355X, B = 8, 16384 -> 4096): the skip test removes 85 % of the wave-level distance loops
// (76.6 k of 524 k wave-iterations stay active), yet the kernel takes 5.68 ms against 4.74 ms for the full
// scan: every iteration still waits for the one wave that owns the neighbourhood of the new sample and runs
// its full loop alone, latency-bound, in about the time four interleaved waves need for theirs.
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

bool launch_fps_pruned(int variant, int b, int n, int m, const float* xyz, float* temp, int* idx, hipStream_t s) {
    if (n % 1024 != 0 || m <= 1) return false;
    const int J = n / 1024;
    const size_t lds = (size_t)n * 8;   // sort entries; later: index table (n * 4) + boxes / candidates / picks
    if (variant == 2 && (J == 8 || J == 16)) {
        if (J == 16) {
            (void)hipFuncSetAttribute((const void*)fps_slotprune_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL((fps_slotprune_kernel<16>), dim3(b), dim3(1024), lds, s, n, m, xyz, temp, idx);
        } else {
            (void)hipFuncSetAttribute((const void*)fps_slotprune_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL((fps_slotprune_kernel<8>), dim3(b), dim3(1024), lds, s, n, m, xyz, temp, idx);
        }
        return true;
    }
    if (variant == 1 && (J == 4 || J == 8 || J == 16)) {
        const size_t need = (size_t)n * 8;   // sort buffer; the loop uses n*4 (rank -> index) + candidates + staged picks
        const size_t loop_lds = (size_t)n * 4 + 2 * 16 * sizeof(FpsCandP) + FPS_OUT_CHUNK * sizeof(int);
        const size_t lds1 = need > loop_lds ? need : loop_lds;
#define JM_FPS_PRUNED(P)                                                                                      \
    do {                                                                                                     \
        (void)hipFuncSetAttribute((const void*)fps_pruned_kernel<P>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1); \
        hipLaunchKernelGGL((fps_pruned_kernel<P>), dim3(b), dim3(1024), lds1, s, n, m, xyz, temp, idx);       \
    } while (0)
        if (J == 16) JM_FPS_PRUNED(16);
        else if (J == 8) JM_FPS_PRUNED(8);
        else JM_FPS_PRUNED(4);
#undef JM_FPS_PRUNED
        return true;
    }
    return false;
}

}  // namespace jm
