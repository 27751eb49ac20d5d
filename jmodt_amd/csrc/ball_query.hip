// ball_query.hip — radius neighbour search for gfx950.
//
// Replaces ball_query_kernel_fast (jmodt/ops/pointnet2/src/ball_query_gpu.cu:9-67).
//
// Design:
//  * Lane = centre.  The candidate point is wave-UNIFORM, so its coordinates are fetched with
//    scalar loads (s_load_dwordx4, 4 points per 48-byte fetch through the scalar cache) and fed
//    to the VALU as SGPR operands: no LDS staging, no per-lane point traffic at all.  The
//    reference makes every thread stream the whole cloud with stride-3 scalar loads per lane.
//  * The scan of one cloud is split across the NW waves of a workgroup (wave w scans the w-th
//    contiguous slice of point indices), which multiplies the resident waves by NW — a level
//    with 8x4096 centres would otherwise run on 512 waves for 16384 sequential steps each.
//    Each wave appends its hits, in ascending index, to an LDS list [slice][slot][lane]
//    (conflict-free: lane is the fastest axis).  After one barrier the slices are concatenated
//    in slice order, truncated to nsample and back-filled with the first hit — exactly the
//    reference's "first nsample in index order".
//  * Up to NR = 2 radii in the same pass (the two MSG scales of an SA level share centres and
//    points; pointnet2_modules.py:46-47), sharing the distance evaluation.
//  * The result rows of the 64 centres of a workgroup are contiguous in idx, so the final
//    write is a linear, coalesced copy out of LDS.
#include "jm_common.h"

namespace jm {

struct BqParams {
    int n, m;
    float r2[2];
    int ns[2];
    int* idx[2];
    int lds_hits_off[2];  // int offsets into dynamic LDS of the per-radius hit lists
    int lds_cnt_off[2];   // per-radius counts [NW][64]
    int chunk;            // points per wave slice (multiple of 8)
    int b, gx;            // frames, centre blocks per frame; gx > 0: 1-D grid with whole frames per XCD (see the kernel)
};

// HT = unsigned short when n <= 65536 (halves the LDS per wave -> twice the waves per CU)
template <int NR, typename HT>
__device__ __forceinline__ void bq_hit(int k, float d2, const BqParams& p, int (&cnt)[NR], HT* lds, int wave,
                                       int lane) {
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        if (d2 < p.r2[r] && cnt[r] < p.ns[r]) {
            lds[p.lds_hits_off[r] + (wave * p.ns[r] + cnt[r]) * 64 + lane] = (HT)k;
            ++cnt[r];
        }
    }
}

// 8 consecutive points = 24 floats = 6 aligned float4, fetched through the scalar cache
struct Pts8 { float4 q[6]; };

__device__ __forceinline__ Pts8 load_pts8(const float* __restrict__ base) {  // wave-uniform address
    Pts8 g;
    const float4* q = reinterpret_cast<const float4*>(base);
#pragma unroll
    for (int i = 0; i < 6; ++i) g.q[i] = q[i];
    return g;
}

template <int NR, typename HT>
__device__ __forceinline__ void bq_group8(int k, const Pts8& g, float cx, float cy, float cz, float r2max,
                                          const BqParams& p, int (&cnt)[NR], HT* lds, int wave, int lane) {
    // (new_x - x)^2 + ... with the oracle's contraction, for the 8 points of the group
    const float f[24] = {g.q[0].x, g.q[0].y, g.q[0].z, g.q[0].w, g.q[1].x, g.q[1].y, g.q[1].z, g.q[1].w,
                         g.q[2].x, g.q[2].y, g.q[2].z, g.q[2].w, g.q[3].x, g.q[3].y, g.q[3].z, g.q[3].w,
                         g.q[4].x, g.q[4].y, g.q[4].z, g.q[4].w, g.q[5].x, g.q[5].y, g.q[5].z, g.q[5].w};
    float d2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) d2[i] = sqdist3(cx - f[3 * i], cy - f[3 * i + 1], cz - f[3 * i + 2]);
    // fast path: nobody in the wave has a hit among these 8 points (the common case: hits are a
    // handful per centre per cloud) -> one min tree + one compare per 8 points
    const float dmin = fminf(fminf(fminf(d2[0], d2[1]), fminf(d2[2], d2[3])), fminf(fminf(d2[4], d2[5]), fminf(d2[6], d2[7])));
    bool open = false;
#pragma unroll
    for (int r = 0; r < NR; ++r) open = open || (cnt[r] < p.ns[r]);
    if (__any(open && dmin < r2max)) {
#pragma unroll
        for (int i = 0; i < 8; ++i) bq_hit<NR, HT>(k + i, d2[i], p, cnt, lds, wave, lane);
    }
}

template <int NR, typename HT>
__global__ void __launch_bounds__(1024)
ball_query_kernel(BqParams p, const float* __restrict__ new_xyz, const float* __restrict__ xyz) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    HT* lds = reinterpret_cast<HT*>(lds_raw);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // provably wave-uniform -> scalar loads
    const int nw = blockDim.x >> 6;
    // Workgroups are dealt to the 8 XCDs round-robin and every XCD has its own L2: with the frame in blockIdx.y the 64+
    // workgroups of ONE frame land on all eight and each L2 pulls every cloud (32 MB of HBM fetches for 8.3 MB of
    // compulsory bytes at B = 8 x 16384, round-1 counters).  1-D grid: workgroup g runs on XCD g % 8 and takes frame
    // (g % 8) + 8 * (slot / gx), so a frame's cloud is streamed through one L2 only.
    int bi = blockIdx.y, cblk = blockIdx.x;
    if (p.gx > 0) {
        const int slot = blockIdx.x >> 3;
        bi = (blockIdx.x & 7) + 8 * (slot / p.gx);
        cblk = slot % p.gx;
        if (bi >= p.b) return;                                   // workgroup-uniform
    }
    const int c0 = cblk * 64;
    const int ci = c0 + lane;
    const bool active = ci < p.m;
    const float* cptr = new_xyz + ((size_t)bi * p.m + (active ? ci : 0)) * 3;
    const float cx = cptr[0], cy = cptr[1], cz = cptr[2];
    const float* pts = xyz + (size_t)bi * p.n * 3;
    const float r2max = NR == 2 ? fmaxf(p.r2[0], p.r2[1]) : p.r2[0];

    int cnt[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) cnt[r] = active ? 0 : p.ns[r];  // inactive lanes look "full"

    const int k_begin = wave * p.chunk;                 // multiple of 8
    const int k_end = min(p.n, k_begin + p.chunk);
    int k = k_begin;
    // 16-byte aligned fast path: groups of 8 points, the next group's scalar loads are issued
    // before the current group is processed (software prefetch through the scalar cache)
    const bool aligned16 = ((reinterpret_cast<uintptr_t>(pts) & 15u) == 0);
    if (aligned16 && k + 16 <= k_end) {
        // two named buffers: while one group of 8 is being evaluated the other one's scalar loads
        // are in flight (prefetch distance = a full group of work per wave)
        Pts8 ga = load_pts8(pts + (size_t)k * 3);
        Pts8 gb = load_pts8(pts + (size_t)(k + 8) * 3);
        for (; k + 32 <= k_end; k += 16) {
            bq_group8<NR, HT>(k, ga, cx, cy, cz, r2max, p, cnt, lds, wave, lane);
            ga = load_pts8(pts + (size_t)(k + 16) * 3);
            bq_group8<NR, HT>(k + 8, gb, cx, cy, cz, r2max, p, cnt, lds, wave, lane);
            gb = load_pts8(pts + (size_t)(k + 24) * 3);
            bool full = true;
#pragma unroll
            for (int r = 0; r < NR; ++r) full = full && (cnt[r] >= p.ns[r]);
            if (__all(full)) { k = k_end - 16; break; }   // every list of this wave is full: done
        }
        bq_group8<NR, HT>(k, ga, cx, cy, cz, r2max, p, cnt, lds, wave, lane);
        bq_group8<NR, HT>(k + 8, gb, cx, cy, cz, r2max, p, cnt, lds, wave, lane);
        k += 16;
    }
    for (; k < k_end; ++k) {
        bool full = true;
#pragma unroll
        for (int r = 0; r < NR; ++r) full = full && (cnt[r] >= p.ns[r]);
        if (__all(full)) break;
        const float d2 = sqdist3(cx - pts[k * 3 + 0], cy - pts[k * 3 + 1], cz - pts[k * 3 + 2]);
        bq_hit<NR, HT>(k, d2, p, cnt, lds, wave, lane);
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) lds[p.lds_cnt_off[r] + wave * 64 + lane] = (HT)(active ? cnt[r] : 0);
    __syncthreads();

    // merge: out row of centre l = concat over slices, truncated to ns, back-filled with the first
    // hit.  Element e of the block's contiguous output [64][ns] -> centre e / ns, slot e % ns.
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int ns = p.ns[r];
        const HT* hits = lds + p.lds_hits_off[r];
        const HT* cnts = lds + p.lds_cnt_off[r];
        int* out = p.idx[r] + ((size_t)bi * p.m + c0) * ns;
        const int valid = min(64, p.m - c0) * ns;
        for (int e = threadIdx.x; e < valid; e += blockDim.x) {
            const int l = e / ns, s = e - l * ns;
            int total = 0, val = -1, first = -1;
            for (int w = 0; w < nw; ++w) {
                const int cw = cnts[w * 64 + l];
                if (first < 0 && cw > 0) first = hits[(w * ns) * 64 + l];
                if (val < 0 && s < total + cw) val = hits[(w * ns + (s - total)) * 64 + l];
                total += cw;
            }
            out[e] = total > 0 ? ((s < total) ? val : first) : 0;  // no hit: 0, what the reference's caller pre-fills (pointnet2_utils.py:218)
        }
    }
}

static int launch_ball_query(int b, int n, int m, int nr, const float* radius, const int* nsample,
                             const float* new_xyz, const float* xyz, int* const* idx, hipStream_t s) {
    JM_REQUIRE(b >= 0 && n >= 0 && m >= 0, "ball_query: bad sizes b=%d n=%d m=%d", b, n, m);
    for (int r = 0; r < nr; ++r) JM_REQUIRE(nsample[r] >= 1 && nsample[r] <= 1024, "ball_query: nsample=%d unsupported", nsample[r]);
    if (b == 0 || m == 0 || n == 0) return JM_OK;
    JM_REQUIRE(new_xyz && xyz && idx[0] && (nr == 1 || idx[1]), "ball_query: null pointer");
    int ns_sum = 0;
    for (int r = 0; r < nr; ++r) ns_sum += nsample[r];
    // waves per workgroup: enough to put >= ~4096 waves on the chip, bounded by LDS (hit lists)
    const int groups = b * divup(m, 64);
    int nw = 1;
    while (nw < 16 && groups * nw < 4096 && (n / (nw * 2)) >= 256) nw *= 2;
    const bool small = n <= 65536;   // indices (and counts <= 1024) fit 16 bits
    const int esz = small ? 2 : 4;
    const int lds_budget = 52 * 1024;  // <= 3 workgroups per CU
    while (nw > 1 && (nw * 64 * (ns_sum + nr)) * esz > lds_budget) nw /= 2;
    BqParams p;
    p.n = n; p.m = m;
    int off = 0;
    for (int r = 0; r < 2; ++r) {
        const int rr = r < nr ? r : 0;
        p.r2[r] = radius[rr] * radius[rr];  // float product, as ball_query_gpu.cu:24
        p.ns[r] = nsample[rr];
        p.idx[r] = idx[rr];
        p.lds_hits_off[r] = off;
        if (r < nr) off += nw * nsample[rr] * 64;
        p.lds_cnt_off[r] = off;
        if (r < nr) off += nw * 64;
    }
    p.chunk = (divup(n, nw) + 7) / 8 * 8;
    const size_t lds_bytes = (size_t)off * esz;
    JM_REQUIRE(lds_bytes <= 160 * 1024, "ball_query: nsample too large for LDS (%zu B)", lds_bytes);
    p.b = b; p.gx = 0;
    dim3 grid(divup(m, 64), b), block(64 * nw);
    if (b >= 8 && (long long)divup(b, 8) * 8 * divup(m, 64) < (1LL << 31)) {   // whole frames per XCD
        p.gx = divup(m, 64);
        grid = dim3((unsigned)(divup(b, 8) * 8 * p.gx), 1);
    }
#define JM_BQ_LAUNCH(NRV, HTV)                                                                              \
    do {                                                                                                   \
        if (lds_bytes > 64 * 1024)                                                                         \
            (void)hipFuncSetAttribute((const void*)ball_query_kernel<NRV, HTV>,                             \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);         \
        hipLaunchKernelGGL((ball_query_kernel<NRV, HTV>), grid, block, lds_bytes, s, p, new_xyz, xyz);      \
    } while (0)
    if (nr == 1) { if (small) JM_BQ_LAUNCH(1, unsigned short); else JM_BQ_LAUNCH(1, int); }
    else         { if (small) JM_BQ_LAUNCH(2, unsigned short); else JM_BQ_LAUNCH(2, int); }
#undef JM_BQ_LAUNCH
    return check_launch("ball_query");
}

}  // namespace jm

extern "C" int jm_ball_query(int b, int n, int m, float radius, int nsample, const float* new_xyz, const float* xyz,
                             int* idx, jm_stream_t stream) {
    int* idxs[2] = {idx, nullptr};
    return jm::launch_ball_query(b, n, m, 1, &radius, &nsample, new_xyz, xyz, idxs, (hipStream_t)stream);
}

extern "C" int jm_ball_query_dual(int b, int n, int m, float radius0, int nsample0, float radius1, int nsample1,
                                  const float* new_xyz, const float* xyz, int* idx0, int* idx1, jm_stream_t stream) {
    const float rad[2] = {radius0, radius1};
    const int ns[2] = {nsample0, nsample1};
    int* idxs[2] = {idx0, idx1};
    return jm::launch_ball_query(b, n, m, 2, rad, ns, new_xyz, xyz, idxs, (hipStream_t)stream);
}
