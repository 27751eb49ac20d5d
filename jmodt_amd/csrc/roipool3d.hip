// roipool3d.hip — RoI point pooling for gfx950.
//
// Replaces assign_pts_to_box3d + get_pooled_idx + roipool3d_forward
// (jmodt/ops/roipool3d/src/roipool3d_kernel.cu:97-237) and the host CPU entry points
// (roipool3d.cpp:82-195).
//
// Design:
//  * One workgroup per (frame, box).  Phase A compacts the indices of the first S in-box points
//    in ascending order straight into LDS with wave ballot + prefix popcount (one barrier per
//    256-point tile, parity-buffered wave totals); the reference materialises a B*N*M int
//    assignment tensor in HBM (67 MB at B=8,N=16384,M=128) and then scans it with one thread
//    per box.  Early exit once S points are found.
//  * Phase B writes the box's (S, 3+C) output slab as ONE flat, fully coalesced stream (16-byte
//    stores when the slab is 16-byte aligned); the source rows (xyz | feature) are contiguous,
//    so reads are coalesced too.  The reference copies with a thread-serial inner loop over
//    3+C floats (uncoalesced across threads).
//  * No temp allocation, no host sync.  With zero_empty=1 the kernel also writes the zeros of
//    empty boxes, so the caller can skip the 279 MB memset of the reference wrapper
//    (roipool3d_utils.py:22-24).
//  * In-box test: same mixed float/double expressions as pt_in_box3d
//    (roipool3d_kernel.cu:14-28); sin/cos via jm_detmath.h, once per box.
#include "../../include/jm_detmath.h"
#include "jm_common.h"

namespace jm {

// The reference writes the box test with double literals (`h / 2.0`, `-l / 2.0`), i.e. float values
// promoted to double and compared against a double threshold.  Halving a float is exact, so each
// threshold IS a float and `(double)f > (double)t` == `f > t`: the per-point test below is pure
// float arithmetic with bit-identical decisions (the oracle keeps the literal double form and the
// GPU tests compare against it).  Only cy = (float)(bottom_y - h / 2.0) is evaluated in double,
// once per box, exactly as written.
struct BoxTest {
    float cx, cz, cy, cosa, sina;
    float half_h, half_l, half_w;
};

__host__ __device__ __forceinline__ BoxTest make_box_test(const float* bx) {
    BoxTest t;
    const float bottom_y = bx[1], h = bx[3], w = bx[4], l = bx[5];
    t.cx = bx[0]; t.cz = bx[2];
    t.cy = (float)(bottom_y - h / 2.0);
    t.half_h = h * 0.5f; t.half_l = l * 0.5f; t.half_w = w * 0.5f;   // exact
    jm_sincosf(bx[6], &t.sina, &t.cosa);
    return t;
}

__host__ __device__ __forceinline__ int pt_in_box(const BoxTest& t, float x, float y, float z) {
    if ((fabsf(x - t.cx) > 10.0f) || (fabsf(y - t.cy) > t.half_h) || (fabsf(z - t.cz) > 10.0f)) return 0;
    const float x_rot = (x - t.cx) * t.cosa + (z - t.cz) * (-t.sina);
    const float z_rot = (x - t.cx) * t.sina + (z - t.cz) * t.cosa;
    return (x_rot >= -t.half_l) & (x_rot <= t.half_l) & (z_rot >= -t.half_w) & (z_rot <= t.half_w);
}

// 16-byte load from a 4-byte aligned address (one global_load_dwordx4; gfx950 handles the
// misalignment in hardware)
struct __attribute__((packed, aligned(4))) F4u { float x, y, z, w; };

constexpr int RP_T = 512;          // threads per box: 8 waves -> 32 waves per CU with 4 boxes resident
constexpr int RP_W = RP_T / 64;
constexpr int RP_TILE = RP_T * 4;  // points per compaction tile
#ifndef RP_U
#define RP_U 4
#endif
constexpr int RP_UNROLL = RP_U;    // independent 16-byte outputs in flight per thread per copy trip

// CANON: boxes3d are the ORIGINAL RoIs; the kernel enlarges them itself (h, w, l += e2; y += e1 —
// kitti_utils.py:152-162 in float32) for the in-box test, and writes the pooled coordinates in the
// RoI's canonical frame: (x, y, z) - roi centre, then (x, z) rotated by ry
// (proposal_target_layer.py:106-112; kitti_utils.py:46-64: [x z] @ R^T, R = [[cos, -sin], [sin, cos]]).
// Empty RoIs get the transform of the zero row the reference pre-fills, features 0.
template <bool VEC4, bool CANON>
__global__ void __launch_bounds__(RP_T)
roipool3d_kernel(int N, int M, int C, int S, const float* __restrict__ xyz, const float* __restrict__ boxes3d,
                 const float* __restrict__ pts_feature, float* __restrict__ pooled, int* __restrict__ empty_flag,
                 int zero_empty, float e1, float e2, int* __restrict__ pooled_cnt) {
    extern __shared__ __attribute__((aligned(16))) int lds[];  // [S] indices, then [2][RP_W] wave totals
    int* sel = lds;
    int* wtot = lds + ((S + 3) & ~3);
    const int mi = blockIdx.x, bi = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* pts = xyz + (size_t)bi * N * 3;
    const float* braw = boxes3d + ((size_t)bi * M + mi) * 7;
    float bx[7];
#pragma unroll
    for (int q = 0; q < 7; ++q) bx[q] = braw[q];
    const float ccx = bx[0], ccy = bx[1], ccz = bx[2];      // canonical frame origin: the un-enlarged RoI
    if (CANON) { bx[1] += e1; bx[3] += e2; bx[4] += e2; bx[5] += e2; }
    const BoxTest bt = make_box_test(bx);

    // ---- phase A: ordered compaction of in-box point indices.  Tile = 1024 points, thread t owns
    // the 4 consecutive points base + 4t .. + 3 (three 16-byte loads), so the order inside a tile
    // is thread-major: position = cnt + hits of earlier waves + hits of earlier lanes + own
    // earlier hits.  One barrier per 1024 points (parity-buffered wave totals).
    int cnt = 0;  // identical in every thread
    int parity = 0;
    const bool vec_ok = ((reinterpret_cast<uintptr_t>(pts) & 15u) == 0);
    auto load_tile = [&](int base, float (&p)[12]) {   // unconditional, clamped: stays in flight across the barrier
        const int k0 = base + tid * 4;
        if (vec_ok && k0 + 4 <= N) {
            const float4* q = reinterpret_cast<const float4*>(pts + (size_t)k0 * 3);
            const float4 a = q[0], b = q[1], c = q[2];
            p[0] = a.x; p[1] = a.y; p[2] = a.z; p[3] = a.w; p[4] = b.x; p[5] = b.y;
            p[6] = b.z; p[7] = b.w; p[8] = c.x; p[9] = c.y; p[10] = c.z; p[11] = c.w;
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int kk = min(max(k0 + q, 0), N - 1);
                p[3 * q] = pts[kk * 3]; p[3 * q + 1] = pts[kk * 3 + 1]; p[3 * q + 2] = pts[kk * 3 + 2];
            }
        }
    };
    float pa[12], pb[12];
    if (N > 0) load_tile(0, pa);   // (no points at all: nothing to read, every box takes the empty path below)
    for (int base = 0; base < N && cnt < S; base += RP_TILE, parity ^= 1) {
        load_tile(min(base + RP_TILE, max(N - 4, 0) & ~3), pb);   // prefetch the next tile (clamped, 4-aligned, when past the end)
        const int k0 = base + tid * 4;
        int in[4];
        int own = 0, before_lane = 0, wave_total = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            in[q] = (k0 + q < N) ? pt_in_box(bt, pa[3 * q], pa[3 * q + 1], pa[3 * q + 2]) : 0;
            const unsigned long long bal = __ballot(in[q]);
            before_lane += mbcnt(bal);
            wave_total += (int)__popcll(bal);
        }
        if (lane == 0) wtot[parity * RP_W + wave] = wave_total;
        lds_barrier();
        int before_wave = 0, all = 0;
#pragma unroll
        for (int w = 0; w < RP_W; ++w) {
            const int tw = wtot[parity * RP_W + w];
            before_wave += (w < wave) ? tw : 0;
            all += tw;
        }
        const int pos = cnt + before_wave + before_lane;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (in[q]) {
                if (pos + own < S) sel[pos + own] = k0 + q;
                ++own;
            }
        }
        cnt += all;
#pragma unroll
        for (int q = 0; q < 12; ++q) pa[q] = pb[q];
    }
    __syncthreads();
    if (cnt > S) cnt = S;
    // rows cnt .. S-1 of the slab are cyclic copies of rows 0 .. cnt-1 (row s = row s % cnt): callers that want to skip
    // the redundant rows (ops/pointnet2/fused.py, sa_dedupe.hip) read the count here
    if (pooled_cnt && tid == 0) pooled_cnt[(size_t)bi * M + mi] = cnt;

    float* dst = pooled + ((size_t)bi * M + mi) * S * (3 + C);
    const int RC = 3 + C;
    const int total = S * RC;
    if (cnt == 0) {
        if (tid == 0) empty_flag[(size_t)bi * M + mi] = 1;
        if (CANON) {
            const float x = 0.f - ccx, y = 0.f - ccy, z = 0.f - ccz;
            const float t0 = x * bt.cosa + z * (-bt.sina), t2 = x * bt.sina + z * bt.cosa;
            for (int e = tid; e < total; e += RP_T) {
                const int j = e % RC;
                dst[e] = j == 0 ? t0 : (j == 1 ? y : (j == 2 ? t2 : 0.f));
            }
        } else if (zero_empty) {
            if (VEC4) {
                float4* d4 = reinterpret_cast<float4*>(dst);
                for (int e = tid; e < total / 4; e += RP_T) d4[e] = make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                for (int e = tid; e < total; e += RP_T) dst[e] = 0.f;
            }
        }
        return;
    }
    if ((zero_empty || CANON) && tid == 0) empty_flag[(size_t)bi * M + mi] = 0;
    // expand the cyclic padding once (sel[s] = sel[s % cnt]) so the copy loop never divides
    for (int s2 = cnt + tid; s2 < S; s2 += RP_T) sel[s2] = sel[s2 % cnt];
    __syncthreads();

    // ---- phase B: flat coalesced copy; row s comes from point sel[s].  The loop is latency
    // bound (LDS index -> dependent global read -> store), so each thread keeps 4 independent
    // 16-byte outputs (16 gathers) in flight per trip.
    const float* feat = pts_feature + (size_t)bi * N * C;
    const unsigned rc_magic = (unsigned)(0x100000000ULL / (unsigned)RC) + 1u;  // e / RC = umulhi(e, magic), e < 2^31 / RC
    auto fetch = [&](int s, int j) -> float {
        // select the ADDRESS, then one unconditional load (a `cond ? load_a : load_b` becomes two
        // branch-guarded loads with a vmcnt(0) at every join, which serialises the 16 gathers)
        const int src = sel[s];
        const float* a = j < 3 ? pts + (src * 3 + j) : feat + ((size_t)src * C + (j - 3));
        const float v = *a;
        if (!CANON) return v;
        // canonical coordinates need x and z together: two more (always valid) loads, then a select
        const float x = pts[src * 3] - ccx, z = pts[src * 3 + 2] - ccz;
        const float r0 = x * bt.cosa + z * (-bt.sina), r2 = x * bt.sina + z * bt.cosa;
        return j == 0 ? r0 : (j == 1 ? v - ccy : (j == 2 ? r2 : v));
    };
    if (VEC4) {
        const int nvec = total / 4;
        for (int f0 = tid; f0 < nvec; f0 += RP_UNROLL * RP_T) {
            float v[RP_UNROLL][4];
#pragma unroll
            for (int u = 0; u < RP_UNROLL; ++u) {
                const int f = min(f0 + u * RP_T, nvec - 1);   // clamped: loads stay unconditional
                const int e = f * 4;
                int s = (int)__umulhi((unsigned)e, rc_magic), j = e - s * RC;
                if (j >= 3 && j + 3 < RC) {
                    // the 4 outputs lie inside one feature row: ONE 16-byte gather instead of four
                    // lane-strided dword gathers (the copy was bound by the texture addresser)
                    const F4u q = *reinterpret_cast<const F4u*>(feat + ((size_t)sel[s] * C + (j - 3)));
                    v[u][0] = q.x; v[u][1] = q.y; v[u][2] = q.z; v[u][3] = q.w;
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        v[u][q] = fetch(s, j);
                        if (++j == RC) { j = 0; ++s; }
                    }
                }
            }
            float4* d4 = reinterpret_cast<float4*>(dst);
#pragma unroll
            for (int u = 0; u < RP_UNROLL; ++u)
                if (f0 + u * RP_T < nvec) d4[f0 + u * RP_T] = make_float4(v[u][0], v[u][1], v[u][2], v[u][3]);
        }
    } else {
        for (int e0 = tid; e0 < total; e0 += 4 * RP_T) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = min(e0 + u * RP_T, total - 1);
                const int s = (int)__umulhi((unsigned)e, rc_magic), j = e - s * RC;
                v[u] = fetch(s, j);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (e0 + u * RP_T < total) dst[e0 + u * RP_T] = v[u];
        }
    }
}

}  // namespace jm

using namespace jm;

template <bool CANON>
static int roipool3d_launch(int batch_size, int pts_num, int boxes_num, int feature_in_len, int sampled_pts_num,
                            const float* xyz, const float* boxes3d, const float* pts_feature, float* pooled_features,
                            int* pooled_empty_flag, int zero_empty, float e1, float e2, jm_stream_t stream,
                            int* pooled_cnt = nullptr) {
    JM_REQUIRE(batch_size >= 0 && pts_num >= 0 && boxes_num >= 0 && feature_in_len >= 0 && sampled_pts_num >= 1,
               "roipool3d: bad sizes");
    if (batch_size == 0 || boxes_num == 0) return JM_OK;
    JM_REQUIRE(xyz && boxes3d && pooled_features && pooled_empty_flag && (pts_feature || feature_in_len == 0),
               "roipool3d: null pointer");
    JM_REQUIRE(batch_size <= 65535, "roipool3d: batch %d > 65535", batch_size);
    JM_REQUIRE(sampled_pts_num <= 32768, "roipool3d: sampled_pts_num %d > 32768", sampled_pts_num);
    const size_t lds = (size_t)(((sampled_pts_num + 3) & ~3) + 2 * RP_W) * sizeof(int);
    const long long slab = (long long)sampled_pts_num * (3 + feature_in_len);
    JM_REQUIRE(slab < (1LL << 31), "roipool3d: slab too large");
    // the copy loop divides flat offsets e <= S*(3+C) by (3+C) with umulhi(e, 2^32/(3+C) + 1): exact while e*(3+C) < 2^32
    JM_REQUIRE(slab * (3 + feature_in_len) < (1LL << 32), "roipool3d: sampled_pts_num * (3 + C)^2 must stay below 2^32");
    const bool vec = (slab % 4 == 0) && ((reinterpret_cast<uintptr_t>(pooled_features) & 15u) == 0);
    dim3 grid(boxes_num, batch_size), block(RP_T);
    if (vec) {
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)roipool3d_kernel<true, CANON>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((roipool3d_kernel<true, CANON>), grid, block, lds, (hipStream_t)stream, pts_num, boxes_num,
                           feature_in_len, sampled_pts_num, xyz, boxes3d, pts_feature, pooled_features,
                           pooled_empty_flag, zero_empty, e1, e2, pooled_cnt);
    } else {
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)roipool3d_kernel<false, CANON>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((roipool3d_kernel<false, CANON>), grid, block, lds, (hipStream_t)stream, pts_num, boxes_num,
                           feature_in_len, sampled_pts_num, xyz, boxes3d, pts_feature, pooled_features,
                           pooled_empty_flag, zero_empty, e1, e2, pooled_cnt);
    }
    return check_launch("roipool3d");
}

extern "C" int jm_roipool3d_forward(int batch_size, int pts_num, int boxes_num, int feature_in_len,
                                    int sampled_pts_num, const float* xyz, const float* boxes3d,
                                    const float* pts_feature, float* pooled_features, int* pooled_empty_flag,
                                    int zero_empty, jm_stream_t stream) {
    return roipool3d_launch<false>(batch_size, pts_num, boxes_num, feature_in_len, sampled_pts_num, xyz, boxes3d,
                                   pts_feature, pooled_features, pooled_empty_flag, zero_empty, 0.f, 0.f, stream);
}

extern "C" int jm_roipool3d_canonical(int batch_size, int pts_num, int boxes_num, int feature_in_len,
                                      int sampled_pts_num, const float* xyz, const float* rois, float extra_width,
                                      const float* pts_feature, float* pooled_features, int* pooled_empty_flag,
                                      jm_stream_t stream) {
    // the reference enlarges with `+= extra_width * 2` / `+= extra_width` on float32 tensors: the Python
    // scalars are rounded to float32 first
    const float e1 = extra_width, e2 = (float)((double)extra_width * 2.0);
    return roipool3d_launch<true>(batch_size, pts_num, boxes_num, feature_in_len, sampled_pts_num, xyz, rois,
                                  pts_feature, pooled_features, pooled_empty_flag, 1, e1, e2, stream);
}

/* jm_roipool3d_canonical + pooled_cnt (B, M) i32: the number of DISTINCT source points of every slab (0 = empty RoI,
 * else rows cnt .. S-1 repeat rows 0 .. cnt-1 cyclically, roipool3d_kernel.cu:123-160) */
extern "C" int jm_roipool3d_canonical_cnt(int batch_size, int pts_num, int boxes_num, int feature_in_len,
                                          int sampled_pts_num, const float* xyz, const float* rois, float extra_width,
                                          const float* pts_feature, float* pooled_features, int* pooled_empty_flag,
                                          int* pooled_cnt, jm_stream_t stream) {
    JM_REQUIRE(pooled_cnt || batch_size == 0 || boxes_num == 0, "roipool3d: null pooled_cnt");
    const float e1 = extra_width, e2 = (float)((double)extra_width * 2.0);
    return roipool3d_launch<true>(batch_size, pts_num, boxes_num, feature_in_len, sampled_pts_num, xyz, rois,
                                  pts_feature, pooled_features, pooled_empty_flag, 1, e1, e2, stream, pooled_cnt);
}

// ---- host CPU entry points of the reference API (roipool3d.cpp:97-195); synchronous ----------
extern "C" int jm_pts_in_boxes3d_cpu(int boxes_num, int pts_num, const float* pts, const float* boxes3d,
                                     int64_t* pts_flag) {
    JM_REQUIRE(boxes_num >= 0 && pts_num >= 0, "pts_in_boxes3d_cpu: bad sizes");
    if (boxes_num == 0 || pts_num == 0) return JM_OK;
    JM_REQUIRE(pts && boxes3d && pts_flag, "pts_in_boxes3d_cpu: null pointer");
    for (int i = 0; i < boxes_num; ++i) {
        const BoxTest bt = make_box_test(boxes3d + (size_t)i * 7);
        for (int j = 0; j < pts_num; ++j)
            pts_flag[(size_t)i * pts_num + j] = pt_in_box(bt, pts[j * 3], pts[j * 3 + 1], pts[j * 3 + 2]);
    }
    return JM_OK;
}

extern "C" int jm_roipool3d_cpu(int pts_num, int boxes_num, int feature_len, int sampled_pts_num, const float* pts,
                                const float* boxes3d, const float* pts_feature, float* pooled_pts,
                                float* pooled_features, int64_t* pooled_empty_flag) {
    JM_REQUIRE(pts_num >= 0 && boxes_num >= 0 && feature_len >= 0 && sampled_pts_num >= 1, "roipool3d_cpu: bad sizes");
    if (boxes_num == 0) return JM_OK;
    JM_REQUIRE(pts && boxes3d && pooled_pts && pooled_features && pooled_empty_flag, "roipool3d_cpu: null pointer");
    const int S = sampled_pts_num, C = feature_len;
    for (int i = 0; i < boxes_num; ++i) {
        pooled_empty_flag[i] = 0;
        const BoxTest bt = make_box_test(boxes3d + (size_t)i * 7);
        int cnt = 0;
        for (int j = 0; j < pts_num && cnt < S; ++j) {
            if (!pt_in_box(bt, pts[j * 3], pts[j * 3 + 1], pts[j * 3 + 2])) continue;
            for (int k = 0; k < 3; ++k) pooled_pts[((size_t)i * S + cnt) * 3 + k] = pts[(size_t)j * 3 + k];
            for (int k = 0; k < C; ++k) pooled_features[((size_t)i * S + cnt) * C + k] = pts_feature[(size_t)j * C + k];
            ++cnt;
        }
        if (cnt == 0) { pooled_empty_flag[i] = 1; continue; }
        for (int j = cnt; j < S; ++j) {
            for (int k = 0; k < 3; ++k) pooled_pts[((size_t)i * S + j) * 3 + k] = pooled_pts[((size_t)i * S + j % cnt) * 3 + k];
            for (int k = 0; k < C; ++k) pooled_features[((size_t)i * S + j) * C + k] = pooled_features[((size_t)i * S + j % cnt) * C + k];
        }
    }
    return JM_OK;
}
