// proposal.hip — RPN proposal selection for a whole batch.
//
// Replaces the per-frame Python loop of ProposalLayer.forward after the box decode
// (jmodt/detection/layers/proposal_layer.py:34-144: score order -> depth bands with their pre-NMS
// budgets -> BEV NMS per band -> post-NMS budgets -> zero-padded result).  The reference runs, per
// frame and band, ~10 indexing ops, one NMS with a device->host copy of the whole mask and a CPU loop;
// here the batch needs FOUR launches after the score sort and nothing touches the host:
//   proposal_compact_kernel  one workgroup per frame: walks the score order once, ranks the members of
//                            each band with wave ballots + prefix popcounts, writes the BEV boxes of the
//                            selected ones straight into the padded NMS problem buffers
//   nms_first_k              (iou3d.hip) all 2 B problems at once, counts read on the device; only the first post-NMS-budget
//                            survivors of a band are ever used, so the greedy rule is evaluated lazily against the kept
//                            boxes only (axis-aligned IoU; the rotated form keeps the full mask + reduce)
//   proposal_stitch_kernel   kept boxes of band 0 then band 1, post budgets, zero padding
#include "jm_common.h"
#include "../../include/jm_detmath.h"

namespace jm {

int launch_nms_batched(int nprob, int nmax, const int* counts, const float* boxes, float thresh, int normal,
                       int64_t* keep, int* num_keep, void* mask_ws, hipStream_t s);   // iou3d.hip
int launch_nms_first_k(int nprob, int nmax, const int* counts, const float* boxes, float thresh, int group, int cap0, int cap1,
                       int64_t* keep, int* num_keep, unsigned long long* evals, hipStream_t s);   // iou3d.hip
int nms_first_k_capacity();

constexpr int PC_T = 1024, PC_W = PC_T / 64;

// mode 0: distance based (two bands (r0, r1] and (r1, r2] on z; an empty far band takes near members
//         pre1 .. pre1+pre2-1 instead, proposal_layer.py:88-98);  mode 1: score based (one problem)
__global__ void __launch_bounds__(PC_T)
proposal_compact_kernel(int n, int mode, int pre1, int pre2, float r0, float r1, float r2, int pmax,
                        const float* __restrict__ proposals, const long long* __restrict__ order,
                        float* __restrict__ bev, int* __restrict__ src, int* __restrict__ counts) {
    __shared__ int wtot[2][2][PC_W];   // [parity][band][wave]
    __shared__ int red[PC_W];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = mode == 0 ? 2 : 1;
    const float* pr = proposals + (size_t)b * n * 7;
    const long long* ord = order + (size_t)b * n;
    float* bev_b = bev + (size_t)b * K * pmax * 5;
    int* src_b = src + (size_t)b * K * pmax;

    bool far_empty = false;
    if (mode == 0) {   // pass 1: is the far band populated at all?
        int any = 0;
        for (int p = tid; p < n; p += PC_T) {
            const float z = pr[(size_t)ord[p] * 7 + 2];
            any |= (z > r1 && z <= r2) ? 1 : 0;
        }
        const unsigned long long bal = __ballot(any);
        if (lane == 0) red[wave] = bal != 0ULL;
        __syncthreads();
        int tot = 0;
#pragma unroll
        for (int w = 0; w < PC_W; ++w) tot |= red[w];
        far_empty = tot == 0;
    }
    auto emit = [&](int prob, int slot, int k) {
        const float* q = pr + (size_t)k * 7;
        const float cu = q[0], cv = q[2], half_w = q[4] / 2, half_l = q[5] / 2;   // kitti_utils.py:136-149
        float* o = bev_b + ((size_t)prob * pmax + slot) * 5;
        o[0] = cu - half_l; o[1] = cv - half_w; o[2] = cu + half_l; o[3] = cv + half_w; o[4] = q[6];
        src_b[(size_t)prob * pmax + slot] = k;
    };
    const int near_budget = pre1 + ((mode == 0 && far_empty) ? pre2 : 0);
    int c1 = 0, c2 = 0, parity = 0;   // members seen so far (identical in every thread)
    for (int base = 0; base < n; base += PC_T, parity ^= 1) {
        const int p = base + tid;
        int k = 0;
        bool m1 = false, m2 = false;
        if (p < n) {
            k = (int)ord[p];
            if (mode == 0) {
                const float z = pr[(size_t)k * 7 + 2];
                m1 = z > r0 && z <= r1;
                m2 = z > r1 && z <= r2;
            } else {
                m1 = true;
            }
        }
        const unsigned long long b1 = __ballot(m1), b2 = __ballot(m2);
        if (lane == 0) { wtot[parity][0][wave] = (int)__popcll(b1); wtot[parity][1][wave] = (int)__popcll(b2); }
        lds_barrier();
        int before1 = 0, before2 = 0, all1 = 0, all2 = 0;
#pragma unroll
        for (int w = 0; w < PC_W; ++w) {
            const int t1 = wtot[parity][0][w], t2 = wtot[parity][1][w];
            before1 += w < wave ? t1 : 0; before2 += w < wave ? t2 : 0;
            all1 += t1; all2 += t2;
        }
        if (m1) {
            const int rank = c1 + before1 + mbcnt(b1);
            if (rank < pre1) emit(0, rank, k);
            else if (far_empty && rank < pre1 + pre2) emit(1, rank - pre1, k);
        }
        if (m2) {
            const int rank = c2 + before2 + mbcnt(b2);
            if (rank < pre2) emit(1, rank, k);
        }
        c1 += all1; c2 += all2;
        if (c1 >= near_budget && (mode == 1 || far_empty || c2 >= pre2)) break;   // uniform: budgets are full
    }
    if (mode == 0 && !far_empty) {
        // the loop may have stopped before the populations were fully counted; counts are clamped anyway
    }
    if (tid == 0) {
        counts[b * K] = min(c1, pre1);
        if (K == 2) counts[b * K + 1] = far_empty ? max(0, min(c1 - pre1, pre2)) : min(c2, pre2);
    }
}

__global__ void proposal_stitch_kernel(int n, int K, int pmax, int post1, int post2, const float* __restrict__ scores,
                                       const float* __restrict__ proposals, const int* __restrict__ src,
                                       const long long* __restrict__ keep, const int* __restrict__ num_keep,
                                       float* __restrict__ out_boxes, float* __restrict__ out_scores) {
    const int b = blockIdx.y, t = blockIdx.x * blockDim.x + threadIdx.x;
    const int post = post1 + post2;
    if (t >= post) return;
    const int nk1 = min(num_keep[b * K], post1);
    const int nk2 = K == 2 ? min(num_keep[b * K + 1], post2) : 0;
    float v[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, s = 0.f;
    if (t < nk1 + nk2) {
        const int band = t < nk1 ? 0 : 1, j = t - (band ? nk1 : 0);
        const int slot = (int)keep[((size_t)b * K + band) * pmax + j];
        const int k = src[((size_t)b * K + band) * pmax + slot];
        const float* q = proposals + ((size_t)b * n + k) * 7;
#pragma unroll
        for (int c = 0; c < 7; ++c) v[c] = q[c];
        s = scores[(size_t)b * n + k];
    }
    float* o = out_boxes + ((size_t)b * post + t) * 7;
#pragma unroll
    for (int c = 0; c < 7; ++c) o[c] = v[c];
    out_scores[(size_t)b * post + t] = s;
}

struct ProposalWs {
    float* bev; int* src; int* counts; int64_t* keep; int* num_keep; unsigned long long* evals; void* mask; size_t total;
    size_t evals_off;
};

static ProposalWs carve(void* ws, int b, int K, int pmax) {
    ProposalWs w;
    size_t off = 0;
    auto take = [&](size_t bytes) { void* p = ws ? (char*)ws + off : nullptr; off += align_up(bytes, (size_t)256); return p; };
    const size_t P = (size_t)b * K;
    w.bev = (float*)take(P * pmax * 5 * sizeof(float));
    w.src = (int*)take(P * pmax * sizeof(int));
    w.counts = (int*)take(P * sizeof(int));
    w.keep = (int64_t*)take(P * pmax * sizeof(int64_t));
    w.num_keep = (int*)take(P * sizeof(int));
    w.evals_off = off;
    w.evals = (unsigned long long*)take(P * sizeof(unsigned long long));   // IoU evaluations of the lazy NMS, per problem
    w.mask = take(P * jm_nms_workspace_bytes(pmax));
    w.total = off;
    return w;
}

}  // namespace jm

using namespace jm;

extern "C" size_t jm_proposal_select_workspace_bytes(int b, int distance_based, int pre_nms_top_n) {
    if (b < 1 || pre_nms_top_n < 1) return 0;
    const int pre1 = distance_based ? (int)(pre_nms_top_n * 0.7) : pre_nms_top_n;
    const int pre2 = distance_based ? pre_nms_top_n - pre1 : 0;
    const int pmax = pre1 > pre2 ? pre1 : pre2;
    return carve(nullptr, b, distance_based ? 2 : 1, pmax < 1 ? 1 : pmax).total;
}

extern "C" size_t jm_proposal_select_evals_offset(int b, int distance_based, int pre_nms_top_n) {
    if (b < 1 || pre_nms_top_n < 1) return 0;
    const int pre1 = distance_based ? (int)(pre_nms_top_n * 0.7) : pre_nms_top_n;
    const int pre2 = distance_based ? pre_nms_top_n - pre1 : 0;
    const int pmax = pre1 > pre2 ? pre1 : pre2;
    return carve(nullptr, b, distance_based ? 2 : 1, pmax < 1 ? 1 : pmax).evals_off;
}

extern "C" int jm_proposal_select(int b, int n, const float* scores, const float* proposals, const int64_t* order,
                                  int distance_based, int pre_nms_top_n, int post_nms_top_n, float nms_thresh,
                                  int nms_normal, float* out_boxes, float* out_scores, void* ws, size_t ws_bytes,
                                  jm_stream_t stream) {
    JM_REQUIRE(b >= 0 && n >= 0 && pre_nms_top_n >= 1 && post_nms_top_n >= 1, "proposal_select: bad sizes");
    if (b == 0) return JM_OK;
    JM_REQUIRE(scores && proposals && order && out_boxes && out_scores && ws, "proposal_select: null pointer");
    JM_REQUIRE(b <= 32767, "proposal_select: batch too large");
    hipStream_t s = (hipStream_t)stream;
    // proposal_layer.py:63-69: int(total * 0.7) and the rest
    const int pre1 = distance_based ? (int)(pre_nms_top_n * 0.7) : pre_nms_top_n;
    const int pre2 = distance_based ? pre_nms_top_n - pre1 : 0;
    const int post1 = distance_based ? (int)(post_nms_top_n * 0.7) : post_nms_top_n;
    const int post2 = distance_based ? post_nms_top_n - post1 : 0;
    const int K = distance_based ? 2 : 1;
    int pmax = pre1 > pre2 ? pre1 : pre2;
    if (pmax < 1) pmax = 1;
    const ProposalWs w = carve(ws, b, K, pmax);
    if (ws_bytes < w.total) { set_error("proposal_select: workspace %zu < %zu bytes", ws_bytes, w.total); return JM_EWORKSPACE; }
    JM_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 255u) == 0, "proposal_select: workspace must be 256-byte aligned");
    hipLaunchKernelGGL(proposal_compact_kernel, dim3(b), dim3(PC_T), 0, s, n, distance_based ? 0 : 1, pre1, pre2, 0.0f,
                       40.0f, 80.0f, pmax, proposals, (const long long*)order, w.bev, w.src, w.counts);
    int rc = check_launch("proposal_compact");
    if (rc) return rc;
    // measured (tools/nms_first_k_bench.py, 16 problems of 6300 / 2700 boxes): budgets of 89 -> 25-95 us, 358 -> 65-335 us
    // across many-/few-survivor box clouds, against 322-343 us for the pair mask + reduce; beyond ~500 the n * K walk loses
    if (nms_normal && post1 <= 512 && post2 <= 512 && post1 <= nms_first_k_capacity() && post2 <= nms_first_k_capacity())
        rc = launch_nms_first_k(b * K, pmax, w.counts, w.bev, nms_thresh, K, post1, post2, w.keep, w.num_keep, w.evals, s);
    else
        rc = launch_nms_batched(b * K, pmax, w.counts, w.bev, nms_thresh, nms_normal, w.keep, w.num_keep, w.mask, s);
    if (rc) return rc;
    const int post = post1 + post2;
    hipLaunchKernelGGL(proposal_stitch_kernel, dim3(divup(post, 128), b), dim3(128), 0, s, n, K, pmax, post1, post2, scores,
                       proposals, w.src, (const long long*)w.keep, w.num_keep, out_boxes, out_scores);
    return check_launch("proposal_stitch");
}

// ------------------------------------------------------------------------------------------------
// RPN box decode (the first half of ProposalLayer.forward, proposal_layer.py:24-34 ->
// decode_bbox_target, jmodt/utils/bbox_transform.py:27-260) for the RPN's configuration:
// roi = the point itself (N,3), get_xz_fine = True, get_y_by_bin = False, get_ry_fine = False,
// RY_WITH_BIN = False.  One thread per point, the 4*nb + 1 + 2*nh + 3 regression channels of a point read
// as 16-byte vectors.  avg_by_bin selects cfg.*.BBOX_AVG_BY_BIN (config.py:197,207,216 default True):
//   1: x = sum_i softmax(bin)_i * (centre_i + res_i * bin_size)        (:74-103)
//   0: x = centre_argmax + res_argmax * bin_size                        (:52-72)
// heading: argmax bin, ry = (bin * 2pi/nh + res * pi/nh) mod 2pi, wrapped to (-pi, pi]  (:127-145)
// size: res * anchor + anchor (:237-242); y = point y + offset, then += h / 2 (proposal_layer.py:33).
// NOTE parity: the reference function cannot be imported in the authoring container (jmodt.config needs
// easydict), so this kernel is checked against the oracle's restatement only ("parity unpinned").
namespace jm {

// reg element (point p of frame b, channel i) = reg[b * bstride + p * pstride + i * cstride]: point-major rows (bstride = n C, pstride = C,
// cstride = 1: the reference's (B, N, C) layout) or the RPN head's own (B, C, N) output (pstride = 1, cstride = N: a lane = a point,
// every channel read is one coalesced 256-byte row — the point-major form walks 304-byte rows with 4-byte loads, 4 x the bytes)
__global__ void __launch_bounds__(256)
decode_rpn_kernel(long long total, int n, long long bstride, long long pstride, long long cstride, int nb, int nh, float loc_scope,
                  float bin_size, float a_h, float a_w, float a_l, int avg_by_bin, const float* __restrict__ xyz,
                  const float* __restrict__ reg, float* __restrict__ out) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= total) return;
    const long long fb = p / n;
    const float* rbase = reg + fb * bstride + (p - fb * n) * pstride;
    auto R = [&](int i) { return rbase[(long long)i * cstride]; };
    auto axis = [&](int bin_off, int res_off) {
        if (avg_by_bin) {
            float mx = -INFINITY;
            for (int i = 0; i < nb; ++i) mx = fmaxf(mx, R(bin_off + i));
            float den = 0.f, num = 0.f;
            for (int i = 0; i < nb; ++i) {
                const float e = expf(R(bin_off + i) - mx);
                const float centre = (float)i * bin_size + bin_size / 2 - loc_scope;
                den += e;
                num += e * (centre + R(res_off + i) * bin_size);
            }
            return num / den;
        }
        int best = 0;
        float bv = R(bin_off);
        for (int i = 1; i < nb; ++i) { const float v = R(bin_off + i); if (v > bv) { bv = v; best = i; } }   // first maximum
        return (float)best * bin_size + bin_size / 2 - loc_scope + R(res_off + best) * bin_size;
    };
    const float pos_x = axis(0, 2 * nb) + xyz[p * 3 + 0];
    const float pos_z = axis(nb, 3 * nb) + xyz[p * 3 + 2];
    int off = 4 * nb;
    float pos_y = xyz[p * 3 + 1] + R(off);
    off += 1;
    int rb = 0;
    float rv = R(off);
    for (int i = 1; i < nh; ++i) { const float v = R(off + i); if (v > rv) { rv = v; rb = i; } }
    const float two_pi = 6.283185307179586f, pi = 3.141592653589793f;
    const float apc = two_pi / (float)nh;
    float ry = fmodf((float)rb * apc + R(off + nh + rb) * (apc / 2), two_pi);
    if (ry < 0.f) ry += two_pi;          // python % is non-negative
    if (ry > pi) ry -= two_pi;
    off += 2 * nh;
    const float h = R(off) * a_h + a_h, w = R(off + 1) * a_w + a_w, l = R(off + 2) * a_l + a_l;
    pos_y += h / 2;                      // proposal_layer.py:33: y becomes the bottom centre
    float* o = out + p * 7;
    o[0] = pos_x; o[1] = pos_y; o[2] = pos_z; o[3] = h; o[4] = w; o[5] = l; o[6] = ry;
}

}  // namespace jm

static int decode_rpn_launch(long long num_points, int n, long long bstride, long long pstride, long long cstride, int reg_channels,
                             const float* xyz, const float* rpn_reg, float loc_scope, float loc_bin_size, int num_head_bin,
                             const float* anchor_hwl, int avg_by_bin, float* proposals, jm_stream_t stream) {
    JM_REQUIRE(num_points >= 0 && loc_bin_size > 0.f && num_head_bin >= 1 && anchor_hwl, "decode_rpn: bad arguments");
    if (num_points == 0) return JM_OK;
    JM_REQUIRE(xyz && rpn_reg && proposals, "decode_rpn: null pointer");
    const int nb = (int)(loc_scope / loc_bin_size) * 2;          // per_loc_bin_num (bbox_transform.py:45)
    JM_REQUIRE(nb >= 1 && reg_channels == 4 * nb + 1 + 2 * num_head_bin + 3,
               "decode_rpn: %d regression channels, expected 4*%d + 1 + 2*%d + 3", reg_channels, nb, num_head_bin);
    hipLaunchKernelGGL(jm::decode_rpn_kernel, dim3((unsigned)jm::divup(num_points, 256LL)), dim3(256), 0, (hipStream_t)stream, num_points, n,
                       bstride, pstride, cstride, nb, num_head_bin, loc_scope, loc_bin_size, anchor_hwl[0], anchor_hwl[1], anchor_hwl[2],
                       avg_by_bin ? 1 : 0, xyz, rpn_reg, proposals);
    return jm::check_launch("decode_rpn");
}

extern "C" int jm_decode_rpn_proposals(long long num_points, int reg_channels, const float* xyz, const float* rpn_reg,
                                       float loc_scope, float loc_bin_size, int num_head_bin, const float* anchor_hwl,
                                       int avg_by_bin, float* proposals, jm_stream_t stream) {
    const int n = (int)(num_points < 1 ? 1 : (num_points > 0x7fffffffLL ? 0x7fffffffLL : num_points));     // one "frame" of rows
    JM_REQUIRE(num_points <= 0x7fffffffLL, "decode_rpn: too many points");
    return decode_rpn_launch(num_points, n, 0, reg_channels, 1, reg_channels, xyz, rpn_reg, loc_scope, loc_bin_size, num_head_bin, anchor_hwl,
                             avg_by_bin, proposals, stream);
}

extern "C" int jm_decode_rpn_proposals_strided(int b, int n, int reg_channels, const float* xyz, const float* rpn_reg, long long batch_stride,
                                               long long point_stride, long long channel_stride, float loc_scope, float loc_bin_size,
                                               int num_head_bin, const float* anchor_hwl, int avg_by_bin, float* proposals,
                                               jm_stream_t stream) {
    JM_REQUIRE(b >= 0 && n >= 0 && batch_stride >= 0 && point_stride >= 1 && channel_stride >= 1, "decode_rpn_strided: bad arguments");
    if (b == 0 || n == 0) return JM_OK;
    return decode_rpn_launch((long long)b * n, n, batch_stride, point_stride, channel_stride, reg_channels, xyz, rpn_reg, loc_scope,
                             loc_bin_size, num_head_bin, anchor_hwl, avg_by_bin, proposals, stream);
}

// ------------------------------------------------------------------------------------------------
// RCNN box decode = decode_bbox_target as the detection post-processing calls it (tools/eval.py:108-116;
// bbox_transform.py:27-260 with roi_box3d (P,7), get_xz_fine = True, get_y_by_bin = False, get_ry_fine = True,
// RY_WITH_BIN = False): the offsets are predicted in the RoI's canonical frame, so the decoded centre is
// rotated back by the RoI heading (rotate_pc_along_y_torch(box, -roi_ry), bbox_transform.py:8-24,251-256)
// and shifted by the RoI centre; heading bins cover (-pi/4, pi/4) around the RoI heading (:131-135).
// y is NOT moved to the bottom face here (that `+= h/2` belongs to the RPN's ProposalLayer only).
namespace jm {

__global__ void __launch_bounds__(256)
decode_rcnn_kernel(long long total, int C, int nb, int nh, float loc_scope, float bin_size, float a_h, float a_w,
                   float a_l, int avg_by_bin, const float* __restrict__ rois, const float* __restrict__ reg,
                   float* __restrict__ out) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= total) return;
    const float* r = reg + p * C;
    const float* roi = rois + p * 7;
    auto axis = [&](int bin_off, int res_off) {
        if (avg_by_bin) {
            float mx = -INFINITY;
            for (int i = 0; i < nb; ++i) mx = fmaxf(mx, r[bin_off + i]);
            float den = 0.f, num = 0.f;
            for (int i = 0; i < nb; ++i) {
                const float e = expf(r[bin_off + i] - mx);
                const float centre = (float)i * bin_size + bin_size / 2 - loc_scope;
                den += e;
                num += e * (centre + r[res_off + i] * bin_size);
            }
            return num / den;
        }
        int best = 0;
        float bv = r[bin_off];
        for (int i = 1; i < nb; ++i) if (r[bin_off + i] > bv) { bv = r[bin_off + i]; best = i; }
        return (float)best * bin_size + bin_size / 2 - loc_scope + r[res_off + best] * bin_size;
    };
    const float pos_x = axis(0, 2 * nb);
    const float pos_z = axis(nb, 3 * nb);
    int off = 4 * nb;
    const float pos_y = roi[1] + r[off];
    off += 1;
    int rb = 0;
    float rv = r[off];
    for (int i = 1; i < nh; ++i) if (r[off + i] > rv) { rv = r[off + i]; rb = i; }
    const float half_pi = 1.5707963267948966f, quarter_pi = 0.7853981633974483f;
    const float apc = half_pi / (float)nh;
    float ry = ((float)rb * apc + apc / 2) + r[off + nh + rb] * (apc / 2) - quarter_pi;
    off += 2 * nh;
    const float h = r[off] * a_h + a_h, w = r[off + 1] * a_w + a_w, l = r[off + 2] * a_l + a_l;
    // rotate (x, z) by -roi_ry: [x, z] @ [[cos, -sin], [sin, cos]]^T with the angle -roi_ry
    const float roi_ry = roi[6];
    float sn, cs;
    jm_sincosf(-roi_ry, &sn, &cs);
    const float x = pos_x * cs + pos_z * (-sn);
    const float z = pos_x * sn + pos_z * cs;
    ry += roi_ry;
    float* o = out + p * 7;
    o[0] = x + roi[0]; o[1] = pos_y; o[2] = z + roi[2]; o[3] = h; o[4] = w; o[5] = l; o[6] = ry;
}

}  // namespace jm

extern "C" int jm_decode_rcnn_boxes(long long num_rois, int reg_channels, const float* rois, const float* rcnn_reg,
                                    float loc_scope, float loc_bin_size, int num_head_bin, const float* anchor_hwl,
                                    int avg_by_bin, float* boxes, jm_stream_t stream) {
    JM_REQUIRE(num_rois >= 0 && loc_bin_size > 0.f && num_head_bin >= 1 && anchor_hwl, "decode_rcnn: bad arguments");
    if (num_rois == 0) return JM_OK;
    JM_REQUIRE(rois && rcnn_reg && boxes, "decode_rcnn: null pointer");
    const int nb = (int)(loc_scope / loc_bin_size) * 2;
    JM_REQUIRE(nb >= 1 && reg_channels == 4 * nb + 1 + 2 * num_head_bin + 3,
               "decode_rcnn: %d regression channels, expected 4*%d + 1 + 2*%d + 3", reg_channels, nb, num_head_bin);
    hipLaunchKernelGGL(decode_rcnn_kernel, dim3((unsigned)divup(num_rois, 256LL)), dim3(256), 0, (hipStream_t)stream,
                       num_rois, reg_channels, nb, num_head_bin, loc_scope, loc_bin_size, anchor_hwl[0], anchor_hwl[1],
                       anchor_hwl[2], avg_by_bin ? 1 : 0, rois, rcnn_reg, boxes);
    return check_launch("decode_rcnn");
}
