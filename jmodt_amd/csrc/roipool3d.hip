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

struct BoxTest {
    float cx, cz, cy, cosa, sina;
    double half_h, half_l, half_w;
};

__host__ __device__ __forceinline__ BoxTest make_box_test(const float* bx) {
    BoxTest t;
    const float bottom_y = bx[1], h = bx[3], w = bx[4], l = bx[5];
    t.cx = bx[0]; t.cz = bx[2];
    t.cy = (float)(bottom_y - h / 2.0);
    t.half_h = h / 2.0; t.half_l = l / 2.0; t.half_w = w / 2.0;
    jm_sincosf(bx[6], &t.sina, &t.cosa);
    return t;
}

__host__ __device__ __forceinline__ int pt_in_box(const BoxTest& t, float x, float y, float z) {
    if ((fabsf(x - t.cx) > 10.0f) || ((double)fabsf(y - t.cy) > t.half_h) || (fabsf(z - t.cz) > 10.0f)) return 0;
    const float x_rot = (x - t.cx) * t.cosa + (z - t.cz) * (-t.sina);
    const float z_rot = (x - t.cx) * t.sina + (z - t.cz) * t.cosa;
    return ((double)x_rot >= -t.half_l) & ((double)x_rot <= t.half_l) & ((double)z_rot >= -t.half_w) &
           ((double)z_rot <= t.half_w);
}

template <bool VEC4>
__global__ void __launch_bounds__(256)
roipool3d_kernel(int N, int M, int C, int S, const float* __restrict__ xyz, const float* __restrict__ boxes3d,
                 const float* __restrict__ pts_feature, float* __restrict__ pooled, int* __restrict__ empty_flag,
                 int zero_empty) {
    extern __shared__ __attribute__((aligned(16))) int lds[];  // [S] indices, then [2][4] wave totals
    int* sel = lds;
    int* wtot = lds + ((S + 3) & ~3);
    const int mi = blockIdx.x, bi = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* pts = xyz + (size_t)bi * N * 3;
    const BoxTest bt = make_box_test(boxes3d + ((size_t)bi * M + mi) * 7);

    // ---- phase A: ordered compaction of in-box point indices
    int cnt = 0;  // identical in every thread
    int parity = 0;
    for (int base = 0; base < N && cnt < S; base += 256, parity ^= 1) {
        const int k = base + tid;
        int in = 0;
        if (k < N) in = pt_in_box(bt, pts[k * 3 + 0], pts[k * 3 + 1], pts[k * 3 + 2]);
        const unsigned long long bal = __ballot(in);
        if (lane == 0) wtot[parity * 4 + wave] = __popcll(bal);
        __syncthreads();
        const int t0 = wtot[parity * 4 + 0], t1 = wtot[parity * 4 + 1], t2 = wtot[parity * 4 + 2],
                  t3 = wtot[parity * 4 + 3];
        const int before = (wave > 0 ? t0 : 0) + (wave > 1 ? t1 : 0) + (wave > 2 ? t2 : 0);
        const int pos = cnt + before + mbcnt(bal);
        if (in && pos < S) sel[pos] = k;
        cnt += t0 + t1 + t2 + t3;
    }
    __syncthreads();
    if (cnt > S) cnt = S;

    float* dst = pooled + ((size_t)bi * M + mi) * S * (3 + C);
    const int RC = 3 + C;
    const int total = S * RC;
    if (cnt == 0) {
        if (tid == 0) empty_flag[(size_t)bi * M + mi] = 1;
        if (zero_empty) {
            if (VEC4) {
                float4* d4 = reinterpret_cast<float4*>(dst);
                for (int e = tid; e < total / 4; e += 256) d4[e] = make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                for (int e = tid; e < total; e += 256) dst[e] = 0.f;
            }
        }
        return;
    }
    if (zero_empty && tid == 0) empty_flag[(size_t)bi * M + mi] = 0;

    // ---- phase B: flat coalesced copy; row s comes from point sel[s % cnt]
    const float* feat = pts_feature + (size_t)bi * N * C;
    auto fetch = [&](int s, int j) -> float {
        const int src = sel[s < cnt ? s : s % cnt];
        return j < 3 ? pts[src * 3 + j] : feat[(size_t)src * C + (j - 3)];
    };
    if (VEC4) {
        // element index e4*4 .. e4*4+3 ; track (s, j) incrementally: step = 1024 elements
        const int step_s = 1024 / RC, step_j = 1024 % RC;
        int e = tid * 4;
        int s = e / RC, j = e - s * RC;
        float4* d4 = reinterpret_cast<float4*>(dst);
        for (; e < total; e += 1024) {
            float v[4];
            int ss = s, jj = j;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                v[q] = fetch(ss, jj);
                if (++jj == RC) { jj = 0; ++ss; }
            }
            d4[e >> 2] = make_float4(v[0], v[1], v[2], v[3]);
            s += step_s; j += step_j;
            if (j >= RC) { j -= RC; ++s; }
        }
    } else {
        const int step_s = 256 / RC, step_j = 256 % RC;
        int e = tid;
        int s = e / RC, j = e - s * RC;
        for (; e < total; e += 256) {
            dst[e] = fetch(s, j);
            s += step_s; j += step_j;
            if (j >= RC) { j -= RC; ++s; }
        }
    }
}

}  // namespace jm

using namespace jm;

extern "C" int jm_roipool3d_forward(int batch_size, int pts_num, int boxes_num, int feature_in_len,
                                    int sampled_pts_num, const float* xyz, const float* boxes3d,
                                    const float* pts_feature, float* pooled_features, int* pooled_empty_flag,
                                    int zero_empty, jm_stream_t stream) {
    JM_REQUIRE(batch_size >= 0 && pts_num >= 0 && boxes_num >= 0 && feature_in_len >= 0 && sampled_pts_num >= 1,
               "roipool3d: bad sizes");
    if (batch_size == 0 || boxes_num == 0) return JM_OK;
    JM_REQUIRE(xyz && boxes3d && pooled_features && pooled_empty_flag && (pts_feature || feature_in_len == 0),
               "roipool3d: null pointer");
    JM_REQUIRE(batch_size <= 65535, "roipool3d: batch %d > 65535", batch_size);
    JM_REQUIRE(sampled_pts_num <= 32768, "roipool3d: sampled_pts_num %d > 32768", sampled_pts_num);
    const size_t lds = (size_t)(((sampled_pts_num + 3) & ~3) + 8) * sizeof(int);
    const long long slab = (long long)sampled_pts_num * (3 + feature_in_len);
    JM_REQUIRE(slab < (1LL << 31), "roipool3d: slab too large");
    const bool vec = (slab % 4 == 0) && ((reinterpret_cast<uintptr_t>(pooled_features) & 15u) == 0);
    dim3 grid(boxes_num, batch_size), block(256);
    if (vec) {
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)roipool3d_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(roipool3d_kernel<true>, grid, block, lds, (hipStream_t)stream, pts_num, boxes_num,
                           feature_in_len, sampled_pts_num, xyz, boxes3d, pts_feature, pooled_features,
                           pooled_empty_flag, zero_empty);
    } else {
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)roipool3d_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(roipool3d_kernel<false>, grid, block, lds, (hipStream_t)stream, pts_num, boxes_num,
                           feature_in_len, sampled_pts_num, xyz, boxes3d, pts_feature, pooled_features,
                           pooled_empty_flag, zero_empty);
    }
    return check_launch("roipool3d");
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
