// detections.hip — the glue around the per-frame detection NMS as TWO launches (gfx950).
//
// tools/eval.py:171-193 per frame: sigmoid, score threshold, sort by score, rotated BEV NMS, gather of the survivors' boxes / scores /
// 512-d features.  Batched over the frames that was ~25 element-wise / sort / gather launches around jm_nms_batched
// (ops/detections.select_detections, round 2); here
//   detections_sort_kernel    one workgroup per frame: sigmoid + threshold, the STABLE descending order of the raw scores with the
//                             rejected slots behind the accepted ones (rank by counting: slot i goes to position #{j : key_j > key_i or
//                             (key_j == key_i and j < i)}, the order torch.sort(stable=True, descending=True) produces), the accepted
//                             count, and the sorted boxes in the BEV form jm_nms_batched reads (kitti_utils.py:136-149)
//   detections_gather_kernel  one workgroup per (frame, output slot): boxes, scores, logits, RoI slot and the feature row of the
//                             k-th survivor, zeros behind the last one
// Nothing is compared with a tolerance here: the outputs are selections of the inputs; the sigmoid is 1 / (1 + expf(-x)), the
// expression torch.sigmoid evaluates in float.
#include "jm_common.h"

namespace jm {

constexpr int DET_MAX_M = 32768;      // slots per frame: the keys of one frame in dynamic LDS (128 KB of the CU's 160 KB at the maximum)

__global__ void __launch_bounds__(256)
detections_sort_kernel(int M, const float* __restrict__ boxes, const float* __restrict__ raw, float thresh, long long* __restrict__ order,
                       int* __restrict__ counts, float* __restrict__ bev) {
    extern __shared__ float key[];          // M floats (the launcher sizes it)
    __shared__ int cnt;
    const int b = blockIdx.x;
    const float* rs = raw + (size_t)b * M;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    int mine = 0;
    for (int i = threadIdx.x; i < M; i += blockDim.x) {
        const float x = rs[i];
        const float s = 1.f / (1.f + expf(-x));
        const bool ok = s > thresh;
        key[i] = ok ? x : -__builtin_inff();
        mine += ok ? 1 : 0;
    }
    if (mine) atomicAdd(&cnt, mine);
    __syncthreads();
    if (threadIdx.x == 0) counts[b] = cnt;
    for (int i = threadIdx.x; i < M; i += blockDim.x) {
        const float k = key[i];
        int rank = 0;
        for (int j = 0; j < M; ++j) {
            const float o = key[j];
            rank += (o > k || (o == k && j < i)) ? 1 : 0;
        }
        order[(size_t)b * M + rank] = i;
        const float* bx = boxes + ((size_t)b * M + i) * 7;
        const float cu = bx[0], cv = bx[2], half_l = bx[5] / 2, half_w = bx[4] / 2;
        float* o = bev + ((size_t)b * M + rank) * 5;
        o[0] = cu - half_l; o[1] = cv - half_w; o[2] = cu + half_l; o[3] = cv + half_w; o[4] = bx[6];
    }
}

__global__ void __launch_bounds__(128)
detections_gather_kernel(int M, int C, const float* __restrict__ boxes, const float* __restrict__ raw, const float* __restrict__ feats,
                         const long long* __restrict__ order, const long long* __restrict__ keep, const int* __restrict__ num_keep,
                         float* __restrict__ out_boxes, float* __restrict__ out_scores, float* __restrict__ out_raw,
                         float* __restrict__ out_feats, int* __restrict__ out_count, long long* __restrict__ out_slot) {
    const int b = blockIdx.y, k = blockIdx.x;
    const int n = num_keep[b];
    const bool live = k < n;
    const size_t o = (size_t)b * M + k;
    long long src = 0;
    if (live) src = order[(size_t)b * M + keep[o]];
    if (threadIdx.x == 0) {
        const float x = live ? raw[(size_t)b * M + src] : 0.f;
        out_raw[o] = x;
        out_scores[o] = live ? 1.f / (1.f + expf(-x)) : 0.f;
        out_slot[o] = src;
        if (k == 0) out_count[b] = n;
    }
    if (threadIdx.x < 7) out_boxes[o * 7 + threadIdx.x] = live ? boxes[((size_t)b * M + src) * 7 + threadIdx.x] : 0.f;
    const float* f = feats + ((size_t)b * M + src) * C;
    float* of = out_feats + o * C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) of[c] = live ? f[c] : 0.f;
}

}  // namespace jm

using namespace jm;

extern "C" int jm_detections_sort(int frames, int slots, const float* boxes, const float* raw_scores, float score_thresh, long long* order,
                                  int* counts, float* bev, jm_stream_t stream) {
    JM_REQUIRE(frames >= 0 && slots >= 0 && slots <= DET_MAX_M, "detections_sort: at most %d slots per frame (got %d)", DET_MAX_M, slots);
    if (frames == 0) return JM_OK;
    JM_REQUIRE(counts && (slots == 0 || (boxes && raw_scores && order && bev)), "detections_sort: null pointer");
    // (the reference's eval keeps at most config.py:213 RPN_POST_NMS_TOP_N = 100 slots; 1024 slots = 4 KB; above 64 KB of dynamic LDS the
    // attribute must say so, once)
    const size_t lds = (size_t)slots * sizeof(float);
    if (lds > (48u << 10)) {
        static const bool once = [] { (void)hipFuncSetAttribute((const void*)detections_sort_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                                (int)(DET_MAX_M * sizeof(float))); return true; }();
        (void)once;
    }
    hipLaunchKernelGGL(detections_sort_kernel, dim3((unsigned)frames), dim3(256), lds, (hipStream_t)stream, slots, boxes, raw_scores, score_thresh,
                       order, counts, bev);
    return check_launch("detections_sort");
}

extern "C" int jm_detections_gather(int frames, int slots, int channels, const float* boxes, const float* raw_scores, const float* feats,
                                    const long long* order, const long long* keep, const int* num_keep, float* out_boxes, float* out_scores,
                                    float* out_raw, float* out_feats, int* out_count, long long* out_slot, jm_stream_t stream) {
    JM_REQUIRE(frames >= 0 && slots >= 0 && channels >= 0, "detections_gather: bad sizes");
    if (frames == 0 || slots == 0) return JM_OK;
    JM_REQUIRE(boxes && raw_scores && (channels == 0 || feats) && order && keep && num_keep && out_boxes && out_scores && out_raw &&
                   (channels == 0 || out_feats) && out_count && out_slot,
               "detections_gather: null pointer");
    hipLaunchKernelGGL(detections_gather_kernel, dim3((unsigned)slots, (unsigned)frames), dim3(128), 0, (hipStream_t)stream, slots, channels,
                       boxes, raw_scores, feats, order, keep, num_keep, out_boxes, out_scores, out_raw, out_feats, out_count, out_slot);
    return check_launch("detections_gather");
}
