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
    int chunk;            // points per wave slice (multiple of 4)
};

template <int NR>
__device__ __forceinline__ void bq_visit(int k, float px, float py, float pz, float cx, float cy, float cz,
                                         const BqParams& p, int (&cnt)[NR], int* lds, int wave, int lane) {
    // (new_x - x)^2 + ... with the oracle's contraction
    const float d2 = sqdist3(cx - px, cy - py, cz - pz);
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        if (d2 < p.r2[r] && cnt[r] < p.ns[r]) {
            lds[p.lds_hits_off[r] + (wave * p.ns[r] + cnt[r]) * 64 + lane] = k;
            ++cnt[r];
        }
    }
}

template <int NR>
__global__ void __launch_bounds__(1024)
ball_query_kernel(BqParams p, const float* __restrict__ new_xyz, const float* __restrict__ xyz) {
    extern __shared__ __attribute__((aligned(16))) int lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // provably wave-uniform -> scalar loads
    const int nw = blockDim.x >> 6;
    const int bi = blockIdx.y;
    const int c0 = blockIdx.x * 64;
    const int ci = c0 + lane;
    const bool active = ci < p.m;
    const float* cptr = new_xyz + ((size_t)bi * p.m + (active ? ci : 0)) * 3;
    const float cx = cptr[0], cy = cptr[1], cz = cptr[2];
    const float* pts = xyz + (size_t)bi * p.n * 3;

    int cnt[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) cnt[r] = active ? 0 : p.ns[r];  // inactive lanes look "full"

    const int k_begin = wave * p.chunk;
    const int k_end = min(p.n, k_begin + p.chunk);
    int k = k_begin;
    // 16-byte aligned fast path: 4 points = 3 x s_load_dwordx4
    const bool aligned16 = ((reinterpret_cast<uintptr_t>(pts) & 15u) == 0);
    if (aligned16) {
        for (; k + 4 <= k_end; k += 4) {
            bool full = true;
#pragma unroll
            for (int r = 0; r < NR; ++r) full = full && (cnt[r] >= p.ns[r]);
            if (__all(full)) { k = k_end; break; }
            const float4* q = reinterpret_cast<const float4*>(pts + (size_t)k * 3);  // wave-uniform address
            const float4 a = q[0], b = q[1], c = q[2];
            bq_visit<NR>(k + 0, a.x, a.y, a.z, cx, cy, cz, p, cnt, lds, wave, lane);
            bq_visit<NR>(k + 1, a.w, b.x, b.y, cx, cy, cz, p, cnt, lds, wave, lane);
            bq_visit<NR>(k + 2, b.z, b.w, c.x, cx, cy, cz, p, cnt, lds, wave, lane);
            bq_visit<NR>(k + 3, c.y, c.z, c.w, cx, cy, cz, p, cnt, lds, wave, lane);
        }
    }
    for (; k < k_end; ++k) {
        bool full = true;
#pragma unroll
        for (int r = 0; r < NR; ++r) full = full && (cnt[r] >= p.ns[r]);
        if (__all(full)) break;
        bq_visit<NR>(k, pts[k * 3 + 0], pts[k * 3 + 1], pts[k * 3 + 2], cx, cy, cz, p, cnt, lds, wave, lane);
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) lds[p.lds_cnt_off[r] + wave * 64 + lane] = active ? cnt[r] : 0;
    __syncthreads();

    // merge: out row of centre l = concat over slices, truncated to ns, back-filled with the first
    // hit.  Element e of the block's contiguous output [64][ns] -> centre e / ns, slot e % ns.
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int ns = p.ns[r];
        const int* hits = lds + p.lds_hits_off[r];
        const int* cnts = lds + p.lds_cnt_off[r];
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
            if (total > 0) out[e] = (s < total) ? val : first;  // no hit: keep the caller's fill
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
    const int lds_budget = 64 * 1024;
    while (nw > 1 && (nw * 64 * (ns_sum + nr)) * (int)sizeof(int) > lds_budget) nw /= 2;
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
    p.chunk = (divup(n, nw) + 3) / 4 * 4;
    const size_t lds_bytes = (size_t)off * sizeof(int);
    JM_REQUIRE(lds_bytes <= 160 * 1024, "ball_query: nsample too large for LDS (%zu B)", lds_bytes);
    dim3 grid(divup(m, 64), b), block(64 * nw);
    if (nr == 1) {
        if (lds_bytes > 64 * 1024)
            (void)hipFuncSetAttribute((const void*)ball_query_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        hipLaunchKernelGGL(ball_query_kernel<1>, grid, block, lds_bytes, s, p, new_xyz, xyz);
    } else {
        if (lds_bytes > 64 * 1024)
            (void)hipFuncSetAttribute((const void*)ball_query_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        hipLaunchKernelGGL(ball_query_kernel<2>, grid, block, lds_bytes, s, p, new_xyz, xyz);
    }
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
