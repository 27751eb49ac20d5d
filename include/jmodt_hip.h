/*
 * jmodt_hip.h — C ABI of libjmodt_hip.so: the MI355X (gfx950) implementation of JMODT's
 * detection + association hot path.  Plain pointers and sizes only (no torch types).
 *
 * Every entry point replaces one function of the reference's three pybind extension modules
 * (file:line cited per function, paths relative to the reference repo) or one of the two
 * pure-PyTorch pieces named by the north star (LI-Fusion gather, affinity head).
 *
 * Conventions
 *   - All tensor arguments are DEVICE pointers to contiguous float32 / int32 / int64 data
 *     unless the name ends in _cpu (host pointers; the reference's CPU entry points).
 *   - The caller owns every buffer, including scratch (`ws`); nothing is allocated or freed
 *     and no host synchronisation happens inside a call (the reference cudaMalloc/cudaMemcpy's
 *     inside nms_gpu and roipool3d: iou3d.cpp:87-95, roipool3d_kernel.cu:214-232).
 *   - `stream` is a hipStream_t (NULL = the legacy default stream, which is what the reference
 *     launches on).  Calls are re-entrant; the library keeps no global mutable state except a
 *     thread-local error string.
 *   - Return value: 0 on success, a JM_E* code otherwise (the reference returns a meaningless 1
 *     and exit()s the process on kernel failure: ball_query_gpu.cu:62-66, iou3d.cpp:13-21).
 *     jm_last_error() describes the most recent failure on the calling thread.
 *   - No environment variables are read: experiment switches exist only in the tools build (JM_TOOLS_BUILD).
 */
#ifndef JMODT_HIP_H
#define JMODT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default) /* the library is built with -fvisibility=hidden */

#define JM_OK 0
#define JM_EINVAL 1   /* bad argument (negative size, NULL pointer, unsupported shape) */
#define JM_ELAUNCH 2  /* hipGetLastError() after a launch */
#define JM_EWORKSPACE 3 /* workspace too small */

typedef void* jm_stream_t; /* hipStream_t */

int jm_version(void);
const char* jm_last_error(void);

/* ------------------------------------------------------------------ pointnet2_cuda -------- */

/* farthest_point_sampling_wrapper (pointnet2/src/sampling.cpp:36-46, sampling_gpu.cu:93-253).
 * xyz (B,N,3); temp (B,N) pre-filled by the caller (1e10) and updated in place; idx (B,M) i32.
 * Bit-exact with the reference's block-tree arg-max tie order. */
int jm_furthest_point_sampling(int b, int n, int m, const float* xyz, float* temp, int* idx, jm_stream_t stream);
/* Same, with a caller-provided workspace: clouds of 16384 < n <= 131072 points (config 5: 65536) are then
 * split over ceil(n/16384) co-operating workgroups instead of being streamed from L2 by one.
 * ws: >= jm_fps_workspace_bytes(b, n) bytes (0 when no workspace is needed), 64-byte aligned.  Two such
 * launches must not run concurrently on one device (their workgroups wait for each other).  A cloud whose
 * workgroups could not exchange records for ~2 s (peers kept off the device) is not sampled: its whole index
 * row is -1 (a valid row starts with index 0) and its new_xyz row NaN; the HIP context stays usable. */
size_t jm_fps_workspace_bytes(int b, int n);
/* sampling + the gather of the sampled coordinates that always follows it (pointnet2_modules.py:35-39):
 * additionally writes new_xyz (B, m, 3) = xyz[b, idx[b, j], :].  ws may be NULL when
 * jm_fps_workspace_bytes(b, n) == 0.  temp may be NULL for n <= 131072: the kernels then start from the
 * reference's 1e10 fill themselves and do not store the final distances (nobody reads them after the
 * call in the reference: pointnet2_utils.py:25-27 allocates the buffer per call). */
int jm_furthest_point_sampling_xyz(int b, int n, int m, const float* xyz, float* temp, int* idx, float* new_xyz,
                                   void* ws, size_t ws_bytes, jm_stream_t stream);
int jm_furthest_point_sampling_ws(int b, int n, int m, const float* xyz, float* temp, int* idx, void* ws,
                                  size_t ws_bytes, jm_stream_t stream);

/* gather_points_wrapper / gather_points_grad_wrapper (sampling.cpp:11-33, sampling_gpu.cu:8-83).
 * points (B,C,N), idx (B,M) -> out (B,C,M);  grad_points (B,C,N) pre-zeroed, accumulated. */
int jm_gather_points(int b, int c, int n, int npoints, const float* points, const int* idx, float* out,
                     jm_stream_t stream);
int jm_gather_points_grad(int b, int c, int n, int npoints, const float* grad_out, const int* idx,
                          float* grad_points, jm_stream_t stream);

/* ball_query_wrapper (ball_query.cpp:14-25, ball_query_gpu.cu:9-67).  new_xyz (B,M,3),
 * xyz (B,N,3) -> idx (B,M,nsample) i32, pre-zeroed by the caller; centres without a hit are
 * left untouched.  First `nsample` hits in ascending point index, back-filled with the first. */
int jm_ball_query(int b, int n, int m, float radius, int nsample, const float* new_xyz, const float* xyz,
                  int* idx, jm_stream_t stream);
/* MI355X-native extension: both MSG radii of one SA level in a single pass over xyz
 * (pointnet2_modules.py:46-47 calls ball_query once per radius on the same centres). */
int jm_ball_query_dual(int b, int n, int m, float radius0, int nsample0, float radius1, int nsample1,
                       const float* new_xyz, const float* xyz, int* idx0, int* idx1, jm_stream_t stream);
/* The same two searches through a per-frame hash grid (csrc/ball_query_grid.hip): identical output (the first nsample
 * indices in ascending order, back-fill, untouched rows without a hit), but only the points of the <= 64 grid cells a
 * ball can touch are evaluated instead of all n.  ws: >= jm_ball_query_workspace_bytes(b, n) bytes, 16-byte aligned
 * (bucket table + the points sorted by bucket); the function returns 0 where the brute-force scan is used anyway
 * (n < 2048 or n > 131072: the *_ws entries then forward to the entries above, as they do for ws == NULL). */
size_t jm_ball_query_workspace_bytes(int b, int n);
/* byte offset inside the workspace of 32 uint64 slots whose SUM is the number of (centre, candidate) distance evaluations of
 * the last call that used it (0: no grid form) — the op's real work, for profiles */
size_t jm_ball_query_evals_offset(int b, int n);
int jm_ball_query_ws(int b, int n, int m, float radius, int nsample, const float* new_xyz, const float* xyz, int* idx, void* ws,
                     size_t ws_bytes, jm_stream_t stream);
int jm_ball_query_dual_ws(int b, int n, int m, float radius0, int nsample0, float radius1, int nsample1, const float* new_xyz,
                          const float* xyz, int* idx0, int* idx1, void* ws, size_t ws_bytes, jm_stream_t stream);
/* The two steps of the *_ws entries on their own.  The BUILD depends on the points and the cell radius only — a caller can run
 * it early / on another stream (ops/pointnet2/pyramid.py: every level's grid is built on the FPS side stream) and leave only
 * the query on its critical path.  cell_radius: the largest radius the grid will be searched with (any positive value gives
 * correct results; the same value must be passed to the query).  radius1 <= 0 or idx1 == NULL: one radius. */
int jm_ball_query_grid_build(int b, int n, float cell_radius, const float* xyz, void* ws, size_t ws_bytes, jm_stream_t stream);
int jm_ball_query_grid_query(int b, int n, int m, float cell_radius, float radius0, int nsample0, float radius1, int nsample1,
                             const float* new_xyz, int* idx0, int* idx1, void* ws, size_t ws_bytes, jm_stream_t stream);

/* group_points_wrapper / group_points_grad_wrapper (group_points.cpp:11-36,
 * group_points_gpu.cu:8-86).  points (B,C,N), idx (B,P,S) -> out (B,C,P,S). */
int jm_group_points(int b, int c, int n, int npoints, int nsample, const float* points, const int* idx,
                    float* out, jm_stream_t stream);
int jm_group_points_grad(int b, int c, int n, int npoints, int nsample, const float* grad_out, const int* idx,
                         float* grad_points, jm_stream_t stream);

/* three_nn_wrapper (interpolate.cpp:14-23, interpolate_gpu.cu:9-74).  unknown (B,N,3),
 * known (B,M,3) -> dist2 (B,N,3) SQUARED distances, idx (B,N,3) i32. */
int jm_three_nn(int b, int n, int m, const float* unknown, const float* known, float* dist2, int* idx,
                jm_stream_t stream);
/* The same search through a hash grid over the known points (csrc/three_nn_grid.hip): identical output (the three smallest
 * (distance, index) pairs), a few dozen distance evaluations per unknown point instead of m.  ws: >=
 * jm_three_nn_grid_workspace_bytes(b, n, m) bytes, 16-byte aligned (0: m outside [1024, 16384], no grid form).
 * jm_three_nn_workspace_bytes is the POLICY: the same size where the walk is faster than the scan (n * m >= 2^27, measured),
 * else 0; jm_three_nn_ws scans when ws == NULL and walks the grid whenever it is handed a workspace. */
size_t jm_three_nn_workspace_bytes(int b, int n, int m);
size_t jm_three_nn_grid_workspace_bytes(int b, int n, int m);
int jm_three_nn_ws(int b, int n, int m, const float* unknown, const float* known, float* dist2, int* idx, void* ws,
                   size_t ws_bytes, jm_stream_t stream);

/* three_interpolate_wrapper / _grad_wrapper (interpolate.cpp:26-54, interpolate_gpu.cu:77-161).
 * points (B,C,M), idx/weight (B,N,3) -> out (B,C,N);  grad_points (B,C,M) pre-zeroed. */
int jm_three_interpolate(int b, int c, int m, int n, const float* points, const int* idx, const float* weight,
                         float* out, jm_stream_t stream);
int jm_three_interpolate_grad(int b, int c, int n, int m, const float* grad_out, const int* idx,
                              const float* weight, float* grad_points, jm_stream_t stream);

/* Fused set-abstraction block (MI355X-native; replaces the per-scale body of
 * _PointnetSAModuleBase.forward, pointnet2_modules.py:46-52 = QueryAndGroup + SharedMLP +
 * max_pool2d): out[b, :, m] = max_s relu(W_L ... relu(W_1 [xyz[idx]-new_xyz | feat[idx]] + b_1) ... + b_L).
 * xyz (B,N,3), new_xyz (B,M,3), features (B,C,N) or NULL (C = 0), idx (B,M,nsample) -> out (B, widths[L], M).
 * widths[0] = 3 + C, widths[1..L] = layer outputs (hidden <= 128).  weights[l] / biases[l] are the 1x1
 * conv weight (widths[l+1], widths[l]) and bias with eval-mode BatchNorm folded in, in the device layout
 * produced by jm_sa_mlp_pack (k-tile-major, so that one lane's MFMA B operand for a 16-deep k-tile is 8
 * consecutive floats; zero padded to pad16(widths[l]) x pad128(widths[l+1])).
 * Three kernels behind one entry (jm_sa_mlp_supported tells which; returns 3: the xyz-only scales [3,16,16,32] x 16 samples
 * and [3,32,32,64] x 32 samples of the first backbone level, config.py:75-82, on the vector pipe — sa_xyz.hip):
 * the persistent wave-specialised one
 * (returns 1: nsample in {16,32,64}, M*nsample % 128 == 0, hidden widths <= 128, <= 4 layers) and the wide one
 * (returns 2: 2-3 layers of up to 512 channels, nsample in {16,32}, B*M*nsample % 32 == 0), which also covers
 * GroupAll (pointnet2_utils.py:267-290): idx == NULL and new_xyz == NULL, M == 1, nsample == N — the group is the
 * frame's N points in order, xyz not re-centred.  0: unsupported (callers use the un-fused operators). */
int jm_sa_mlp_supported(int b, int n, int m, int c, int nsample, int group_all, int num_layers, const int* widths);
size_t jm_sa_mlp_packed_weight_elems(int cout, int cin, int first_layer);
size_t jm_sa_mlp_packed_bias_elems(int cout);
/* w (cout, cin) row-major and b (cout) or NULL, both on the device -> wp, bp (sizes above).
 * first_layer != 0 for layer 0, whose cin = 3 + C input channels are QueryAndGroup's [xyz | features]
 * (pointnet2_utils.py:258-262): the kernel consumes them as [features | xyz], each part padded to 16. */
int jm_sa_mlp_pack(int cout, int cin, int first_layer, const float* w, const float* b, float* wp, float* bp,
                   jm_stream_t stream);
int jm_sa_mlp_forward(int b, int n, int m, int c, int nsample, const float* xyz, const float* new_xyz,
                      const float* features, const int* idx, int num_layers, const int* widths,
                      const float* const* weights, const float* const* biases, float* out, jm_stream_t stream);
/* Pre-projected form of the same block.  The first layer is linear in its input, so
 *   W_1 [xyz_j - c_i | f_j] + b_1 = u_j - W_1x c_i,      u = W_1 [xyz | f] + b_1 per POINT (B, C, N), W_1x = W_1[:, 0:3]
 * (C = the first layer's width, <= 128, multiple of 16): one small GEMM by the caller replaces the first layer's work
 * on every (centre, sample) ROW — 31 % of the RCNN SA1 flops.  The kernel gathers u through idx, forms
 * relu(u_j - W_1x c_i) while parking the tile in LDS (w1x: (C, 4) row-major, 4th column unused; new_xyz (B, M, 3)) and
 * runs layers 2..L: widths[0] = C, widths[1..num_layers] = the remaining layer outputs, weights packed with
 * first_layer = 0.  Same shape limits as the persistent kernel of jm_sa_mlp_forward. */
int jm_sa_mlp_forward_pre(int b, int n, int m, int c, int nsample, const float* u, const float* w1x, const float* new_xyz,
                          const int* idx, int num_layers, const int* widths, const float* const* weights,
                          const float* const* biases, float* out, jm_stream_t stream);

/* Duplicate-aware (LISTED) form of the fused set-abstraction block, exact (round 4; csrc/sa_groups.hip).
 * ball_query back-fills a list with cnt < nsample hits with copies of its first hit (ball_query_gpu.cu:36-40), and the
 * max-pool of _PointnetSAModuleBase.forward (pointnet2_modules.py:50-52) is idempotent: only a group's first d = cnt rows
 * matter.  jm_sa_group_plan bins the groups (frame, centre) by q = max(qmin, ceil(log2 d)) — d = 1 + the last slot of the list
 * that differs from its first entry, so the first 2^q entries contain every distinct entry of ANY list — into per-class lists in
 * device memory; jm_sa_mlp_forward_listed runs jm_sa_mlp_forward_into's kernel on tiles of one class each, pooling over 2^q
 * rows, and writes every group's output at its own position: bit-identical to jm_sa_mlp_forward_into (a row's value depends
 * on its point and centre only, max on neither order nor multiplicity), rows executed 2^q instead of nsample per group.
 * plan = cls_count (8 ints: groups per class) + glist (jm_sa_group_list_elems(groups, nsample) = (log2 nsample + 1) x groups
 * ints: class q's group ids at glist[q * groups ...]), both in device memory; no host decision and no host sync anywhere: the
 * plan and the consumer are always launched.  jm_sa_group_plan_dual plans the two scales of a multi-scale level (same centres,
 * two neighbour lists) in one launch: cls_count = 16 ints, [0..7] scale 0, [8..15] scale 1.
 * jm_sa_mlp_listed_supported: 0 = no listed kernel for the shape, else the kernel (as jm_sa_mlp_supported) and
 * jm_sa_mlp_listed_qmin its smallest class. */
size_t jm_sa_group_list_elems(int groups, int nsample);
int jm_sa_group_plan(int groups, int nsample, const int* idx, int qmin, int* cls_count, int* glist, jm_stream_t stream);
int jm_sa_group_plan_dev(int groups, int nsample, const int* idx, int qmin, const int* groups_dev, int* cls_count, int* glist,
                         jm_stream_t stream);   /* valid groups = min(groups_dev[0], groups), read on the device */
int jm_sa_group_plan_dual(int groups, int nsample0, const int* idx0, int qmin0, int nsample1, const int* idx1, int qmin1,
                          int* cls_count, int* glist0, int* glist1, jm_stream_t stream);
int jm_sa_mlp_listed_supported(int b, int n, int m, int c, int nsample, int num_layers, const int* widths);
int jm_sa_mlp_listed_qmin(int kind);
int jm_sa_mlp_forward_listed(int b, int n, int m, int c, int nsample, const float* xyz, const float* new_xyz,
                             const float* features, const int* idx, int num_layers, const int* widths,
                             const float* const* weights, const float* const* biases, const int* cls_count, const int* glist,
                             float* out, size_t out_frame_stride, jm_stream_t stream);

/* the pre-projected two-layer block (jm_sa_mlp_pm_forward_into) in the listed form: plan from jm_sa_group_plan with
 * qmin = jm_sa_mlp_pm_listed_qmin(c, hidden, cout): 2 (sa_mlp_pm_kernel's accumulator layout pools four consecutive rows in a
 * lane), 3 where the per-quad tables do not fit the LDS next to the tiles (C = hidden = 128: the RCNN scales, through
 * fused.sa_scale_pm_dedupe), -1: no listed form */
int jm_sa_mlp_pm_listed_qmin(int c, int hidden, int cout);
int jm_sa_mlp_pm_listed_supported(int b, int n, int m, int c, int nsample, int hidden, int cout);
int jm_sa_mlp_pm_forward_listed(int b, int n, int m, int c, int nsample, int hidden, int cout, const float* u_point_major,
                                const float* w1x, const float* new_xyz, const int* idx, const float* w_hidden,
                                const float* b_hidden, const float* w_out, const float* b_out, const int* cls_count,
                                const int* glist, float* out, size_t out_frame_stride, jm_stream_t stream);

/* ------------------------------------------------------------------ roipool3d_cuda -------- */

/* forward / forward_slow (roipool3d/src/roipool3d.cpp:16-79, roipool3d_kernel.cu:31-237).
 * xyz (B,N,3), boxes3d (B,M,7) ALREADY ENLARGED, pts_feature (B,N,C) ->
 * pooled_features (B,M,S,3+C), pooled_empty_flag (B,M) i32.
 * zero_empty = 0: reference contract — both outputs pre-zeroed by the caller, rows of empty
 *                 boxes are not written.
 * zero_empty = 1: the kernel writes zeros for empty boxes and 0/1 into every flag, so the
 *                 caller may pass uninitialised buffers (saves one full pass over the output). */
int jm_roipool3d_forward(int batch_size, int pts_num, int boxes_num, int feature_in_len, int sampled_pts_num,
                         const float* xyz, const float* boxes3d, const float* pts_feature,
                         float* pooled_features, int* pooled_empty_flag, int zero_empty, jm_stream_t stream);

/* roipool3d + the canonical transformation that always follows it (SURVEY.md §8f row 3;
 * proposal_target_layer.py:100-112, :45-69): rois (B,M,7) are the ORIGINAL boxes; the kernel enlarges them
 * by extra_width (kitti_utils.py:152-162) for the in-box test and writes pooled xyz relative to the RoI:
 * minus rois[...,0:3], then (x, z) rotated by rois[...,6] (kitti_utils.py:46-64).  Every row and flag is
 * written (RoIs without points get the transform of the reference's pre-filled zero row). */
int jm_roipool3d_canonical(int batch_size, int pts_num, int boxes_num, int feature_in_len, int sampled_pts_num,
                           const float* xyz, const float* rois, float extra_width, const float* pts_feature,
                           float* pooled_features, int* pooled_empty_flag, jm_stream_t stream);

/* pts_in_boxes3d_cpu / roipool3d_cpu (roipool3d.cpp:97-195): HOST pointers, synchronous. */
int jm_pts_in_boxes3d_cpu(int boxes_num, int pts_num, const float* pts, const float* boxes3d, int64_t* pts_flag);
int jm_roipool3d_cpu(int pts_num, int boxes_num, int feature_len, int sampled_pts_num, const float* pts,
                     const float* boxes3d, const float* pts_feature, float* pooled_pts, float* pooled_features,
                     int64_t* pooled_empty_flag);

/* ------------------------------------------------------------------ iou3d_cuda ------------ */

/* boxes_overlap_bev_gpu / boxes_iou_bev_gpu (iou3d/src/iou3d.cpp:31-71,
 * iou3d_kernel.cu:108-248,354-371).  boxes (N,5) [x1,y1,x2,y2,ry] -> (Na,Nb). */
int jm_boxes_overlap_bev(int num_a, const float* boxes_a, int num_b, const float* boxes_b, float* ans_overlap,
                         jm_stream_t stream);
int jm_boxes_iou_bev(int num_a, const float* boxes_a, int num_b, const float* boxes_b, float* ans_iou,
                     jm_stream_t stream);

/* Tracker association cost (jmodt/tracking/data_association.py:10-28,42-45,117-119):
 *   cost = link_scores * w_app + boxes_iou3d_gpu(pred, det) * w_iou + boxes_dist_gpu(pred, det) * w_dis
 * pred_boxes (P,7), det_boxes (D,7) [x,y,z,h,w,l,ry]; link_scores (P,D) or NULL; outputs (P,D), any of
 * cost / iou3d_out / dist_out may be NULL.  One launch instead of ~25 torch kernels. */
int jm_association_cost(int num_pred, const float* pred_boxes, int num_det, const float* det_boxes,
                        const float* link_scores, float w_app, float w_iou, float w_dis, float* cost,
                        float* iou3d_out, float* dist_out, jm_stream_t stream);

/* nms_gpu / nms_normal_gpu (iou3d.cpp:73-166, iou3d_kernel.cu:250-348,374-387).
 * boxes (N,5) score-sorted.  The suppression bit-mask AND the greedy reduce both run on the
 * device: keep (N) int64 and num_keep (1) int32 are DEVICE buffers; ws holds the mask
 * (jm_nms_workspace_bytes(N)).  normal = 1 selects the axis-aligned IoU (nms_normal_gpu). */
size_t jm_nms_workspace_bytes(int boxes_num);
int jm_nms(int boxes_num, const float* boxes, float nms_overlap_thresh, int normal, int64_t* keep,
           int* num_keep, void* ws, size_t ws_bytes, jm_stream_t stream);
/* Batched form (the RPN runs 2 distance bands x B frames of NMS per batch,
 * proposal_layer.py:41-117, one after the other with a host sync each in the reference): problem p
 * has counts[p] (device int, <= max_boxes) score-sorted boxes at boxes + p*max_boxes*5; keep is
 * (P, max_boxes) int64, num_keep (P) int32; ws >= P * jm_nms_workspace_bytes(max_boxes).
 * All problems run concurrently and nothing touches the host. */
int jm_nms_batched(int num_problems, int max_boxes, const int* counts, const float* boxes,
                   float nms_overlap_thresh, int normal, int64_t* keep, int* num_keep, void* ws, size_t ws_bytes,
                   jm_stream_t stream);
/* The first `first_k` entries of jm_nms_batched's keep lists with the axis-aligned IoU (nms_normal_gpu), without the pair
 * mask: callers that truncate the keep list (proposal_layer.py:103-117 `keep_idx[:post_nms_top_n]`) only need each box's
 * IoU with the boxes KEPT before it.  keep (P, max_boxes) int64 — entries [0, num_keep[p]) are written —, num_keep (P)
 * int32 = min(first_k, survivors); first_k <= 2048; no workspace.  Bit-identical to the truncated jm_nms_batched result. */
int jm_nms_normal_first_k_batched(int num_problems, int max_boxes, const int* counts, const float* boxes,
                                  float nms_overlap_thresh, int first_k, int64_t* keep, int* num_keep,
                                  jm_stream_t stream);
/* RPN proposal selection for a whole batch (ProposalLayer.forward after the decode,
 * proposal_layer.py:34-144): scores (B,N), proposals (B,N,7) [x, y_bottom, z, h, w, l, ry], order (B,N) =
 * indices of the scores in descending order.  distance_based != 0: depth bands (0,40] / (40,80] with the
 * 70 % / 30 % budgets, empty far band -> next slice of the near band (:57-117), NMS variant nms_normal;
 * distance_based == 0: one problem per frame (:119-144; the reference uses rotated NMS there).
 * out_boxes (B, post_nms_top_n, 7), out_scores (B, post_nms_top_n), zero padded.  No host round trip. */
size_t jm_proposal_select_workspace_bytes(int b, int distance_based, int pre_nms_top_n);
/* byte offset inside that workspace of (K * b) uint64 counters: IoU evaluations the lazy NMS of the last jm_proposal_select
 * call did per (frame, band) problem (the pair mask would do pre^2 / 2) */
size_t jm_proposal_select_evals_offset(int b, int distance_based, int pre_nms_top_n);
int jm_proposal_select(int b, int n, const float* scores, const float* proposals, const int64_t* order,
                       int distance_based, int pre_nms_top_n, int post_nms_top_n, float nms_thresh, int nms_normal,
                       float* out_boxes, float* out_scores, void* ws, size_t ws_bytes, jm_stream_t stream);
/* RPN box decode = decode_bbox_target as ProposalLayer calls it (proposal_layer.py:24-34;
 * bbox_transform.py:27-260 with get_xz_fine=True, get_y_by_bin=False, get_ry_fine=False, RY_WITH_BIN=False):
 * xyz (P,3), rpn_reg (P, 4*nb + 1 + 2*num_head_bin + 3) with nb = int(loc_scope / loc_bin_size) * 2,
 * anchor_hwl = HOST pointer to cfg.CLS_MEAN_SIZE[0] (3 floats), avg_by_bin = cfg.*.BBOX_AVG_BY_BIN.
 * proposals (P,7) = [x, y_bottom, z, h, w, l, ry] (the `+= h/2` of proposal_layer.py:33 included). */
int jm_decode_rpn_proposals(long long num_points, int reg_channels, const float* xyz, const float* rpn_reg,
                            float loc_scope, float loc_bin_size, int num_head_bin, const float* anchor_hwl,
                            int avg_by_bin, float* proposals, jm_stream_t stream);
/* jm_decode_rpn_proposals on a STRIDED regression tensor: element (frame b, point p, channel i) at
 * rpn_reg[b * batch_stride + p * point_stride + i * channel_stride] — e.g. the RPN head's own (B, C, N) output read in place
 * (point_stride 1, channel_stride N: coalesced, no transposed copy).  xyz (B,n,3), proposals (B,n,7). */
int jm_decode_rpn_proposals_strided(int b, int n, int reg_channels, const float* xyz, const float* rpn_reg, long long batch_stride,
                                    long long point_stride, long long channel_stride, float loc_scope, float loc_bin_size,
                                    int num_head_bin, const float* anchor_hwl, int avg_by_bin, float* proposals, jm_stream_t stream);
/* RCNN box decode = decode_bbox_target as the detection post-processing calls it (tools/eval.py:108-116;
 * bbox_transform.py:27-260 with roi_box3d (P,7), get_xz_fine=True, get_y_by_bin=False, get_ry_fine=True,
 * RY_WITH_BIN=False): offsets are in the RoI's canonical frame; the result is rotated back by the RoI heading
 * and shifted by the RoI centre (bbox_transform.py:8-24,251-258).  rois (P,7), rcnn_reg (P, 4*nb+1+2*nh+3),
 * anchor_hwl = HOST pointer (3 floats) -> boxes (P,7) [x, y, z, h, w, l, ry]. */
int jm_decode_rcnn_boxes(long long num_rois, int reg_channels, const float* rois, const float* rcnn_reg,
                         float loc_scope, float loc_bin_size, int num_head_bin, const float* anchor_hwl,
                         int avg_by_bin, float* boxes, jm_stream_t stream);
/* Detection post-processing around the per-frame NMS (tools/eval.py:171-193), batched over the frames,
 * as two launches around jm_nms_batched.  jm_detections_sort: boxes (B,M,7), raw_scores (B,M) logits -> order (B,M) int64 = the
 * stable descending order of the logits with the slots whose sigmoid score is <= score_thresh behind the accepted ones
 * (scores.sort(descending) of eval.py:181 on the thresholded set), counts (B) int32 accepted slots, bev (B,M,5) = the sorted boxes as
 * [x - l/2, z - w/2, x + l/2, z + w/2, ry] (kitti_utils.py:136-149).  M <= 1024.
 * jm_detections_gather: + feats (B,M,C), keep (B,M) int64 / num_keep (B) of jm_nms_batched (positions in sorted order) ->
 * out_boxes (B,M,7), out_scores (B,M) sigmoid, out_raw (B,M), out_feats (B,M,C), out_count (B) int32, out_slot (B,M) int64 = the RoI
 * slot of the k-th survivor; everything behind the last survivor is zero. */
int jm_detections_sort(int frames, int slots, const float* boxes, const float* raw_scores, float score_thresh, long long* order,
                       int* counts, float* bev, jm_stream_t stream);
int jm_detections_gather(int frames, int slots, int channels, const float* boxes, const float* raw_scores, const float* feats,
                         const long long* order, const long long* keep, const int* num_keep, float* out_boxes, float* out_scores,
                         float* out_raw, float* out_feats, int* out_count, long long* out_slot, jm_stream_t stream);
/* boxes_iou3d_gpu (iou3d_utils.py:25-54) for a whole batch (the RoI sampler's per-frame
 * Python loop, proposal_target_layer.py:137-151,288): boxes_a (B,Na,7), boxes_b (B,Nb,7) [x,y,z,h,w,l,ry] with
 * y = box bottom -> iou3d (B,Na,Nb).  counts_b (B) device int32 or NULL: valid boxes per frame in boxes_b
 * (zero-padded ground-truth lists, :141-145); columns beyond it are written as 0. */
int jm_boxes_iou3d_batched(int batch, int num_a, const float* boxes_a, int num_b, const float* boxes_b,
                           const int* counts_b, float* iou3d, jm_stream_t stream);
/* mask only (N, ceil(N/64)) uint64; tiles with col_block < row_block are never consumed by the
 * reduce (iou3d.cpp:108) and are left unwritten. */
int jm_nms_mask(int boxes_num, const float* boxes, float nms_overlap_thresh, int normal,
                unsigned long long* mask, jm_stream_t stream);

/* ------------------------------------------------------------------ LI-Fusion gather ------ */

/* feature_gather (jmodt/detection/modeling/backbone.py:79-89) =
 * F.grid_sample(map, xy[B,1,N,2], bilinear, zeros padding, align_corners=True).
 * fmap (B,C,H,W) with ELEMENT strides (sb,sc,sh,sw) — NCHW or channels-last, no copy;
 * xy (B,N,2) in [-1,1] -> out (B,C,N). */
int jm_feature_gather(int b, int c, int h, int w, int n, const float* fmap, int64_t sb, int64_t sc, int64_t sh,
                      int64_t sw, const float* xy, float* out, jm_stream_t stream);
/* gradient w.r.t. the feature map (pre-zeroed, same strides), accumulated with atomics */
int jm_feature_gather_grad(int b, int c, int h, int w, int n, const float* grad_out, const float* xy,
                           float* grad_fmap, int64_t sb, int64_t sc, int64_t sh, int64_t sw, jm_stream_t stream);

/* The RCNN stage's per-point input MLP (jmodt/detection/modeling/rcnn.py:176-184): pts (R, S, K + C) pooled RoI
 * points [K geometric channels = xyz, mask, depth | C RPN feature channels] ->
 *   h1 = relu(W_up1 x_K + b);  h2 = relu(W_up2 h1 + b);  m = relu(W_merge [h2 | rpn] + b)          (xyz_up + merge_down)
 * out (R, h_m, S).  With h_out > 0 the first set-abstraction layer (linear part, hoisted in front of its gather, see
 * jm_sa_mlp_forward_pre) follows in the same launch: out = u = w_out_m m + w_out_x x_K + b_out, (R, h_out, S).
 * One launch on 32-point tiles instead of 2 transposes + cat + 3 convolutions over (R, 128..256, S) tensors.
 * All matrices / biases in the layout of jm_sa_mlp_pack(cout, cin, 0); w_merge split column-wise into its h2 and C
 * parts; widths <= 128, S % 32 == 0, 3 <= K <= 16. */
/* out_point_major != 0: out is (R, S, h) instead of (R, h, S) — the layout jm_sa_mlp_pm_forward gathers. */
int jm_rcnn_lift_supported(int s, int k, int c, int h1, int h2, int hm, int ho);
int jm_rcnn_lift_forward(int r, int s, int k, int c, int h1, int h2, int hm, int ho, const float* pts, const float* w_up1,
                         const float* b_up1, const float* w_up2, const float* b_up2, const float* w_merge_h,
                         const float* w_merge_f, const float* b_merge, const float* w_out_m, const float* w_out_x,
                         const float* b_out, int out_point_major, float* out, jm_stream_t stream);
/* the same on slabs whose rows count[r] .. S-1 are cyclic copies of rows 0 .. count[r]-1 (jm_roipool3d_canonical_cnt): 32-point
 * tiles holding only copies are skipped and their output rows are NOT written — for consumers that read canonical rows only
 * (the duplicate-compacted set abstraction, jm_sa_dedupe_plan).  work: (1 + r * s / 32) i32 device scratch (list of live tiles) */
int jm_rcnn_lift_forward_cnt(int r, int s, int k, int c, int h1, int h2, int hm, int ho, const float* pts, const float* w_up1,
                             const float* b_up1, const float* w_up2, const float* b_up2, const float* w_merge_h,
                             const float* w_merge_f, const float* b_merge, const float* w_out_m, const float* w_out_x,
                             const float* b_out, int out_point_major, float* out, const int* count, int* work,
                             jm_stream_t stream);

/* The pre-projected set-abstraction block (jm_sa_mlp_forward_pre) for exactly TWO layers after the hoisted one, on a
 * POINT-major u (B, N, C) (C = 32, 64 or 128; the producers jm_conv1d_stack_forward / jm_rcnn_lift_forward write that
 * layout on request): two MFMA waves per SIMD, row-major activation tiles read with 16-byte LDS loads, hidden layer
 * computed transposed (csrc/sa_mlp_pm.hip).  w_hidden (hidden x C) and w_out (cout x hidden) are packed by
 * jm_sa_mlp_pack(cout, cin, 0) AFTER the caller permuted their columns within every 16-block to
 * [0, 8, 1, 9, 2, 10, ...] (source column 16 kt + 8 (j & 1) + (j >> 1) at position 16 kt + j), which makes a lane's
 * eight k-values contiguous in the activation row; biases zero padded as jm_sa_mlp_pack writes them.
 * out (B, cout, M).  hidden <= 128, cout <= 256, nsample in {16, 32, 64}, M * nsample % 128 == 0. */
int jm_sa_mlp_pm_supported(int b, int n, int m, int c, int nsample, int hidden, int cout);
int jm_sa_mlp_pm_forward(int b, int n, int m, int c, int nsample, int hidden, int cout, const float* u_point_major,
                         const float* w1x, const float* new_xyz, const int* idx, const float* w_hidden, const float* b_hidden,
                         const float* w_out, const float* b_out, float* out, jm_stream_t stream);

/* jm_sa_mlp_forward / _forward_pre / _pm_forward writing frame b's (cout, M) block at out + b * out_frame_stride floats
 * (0 = cout * M; otherwise >= cout * M): a channel slice of a wider (B, Ctot, M) tensor, i.e. the concatenation of an MSG
 * module's scales (pointnet2_modules.py:54: torch.cat(new_features_list, dim=1)) without the copy. */
int jm_sa_mlp_forward_into(int b, int n, int m, int c, int nsample, const float* xyz, const float* new_xyz,
                      const float* features, const int* idx, int num_layers, const int* widths,
                      const float* const* weights, const float* const* biases, float* out, size_t out_frame_stride, jm_stream_t stream);
int jm_sa_mlp_forward_pre_into(int b, int n, int m, int c, int nsample, const float* u, const float* w1x, const float* new_xyz,
                          const int* idx, int num_layers, const int* widths, const float* const* weights,
                          const float* const* biases, float* out, size_t out_frame_stride, jm_stream_t stream);
int jm_sa_mlp_pm_forward_into(int b, int n, int m, int c, int nsample, int hidden, int cout, const float* u_point_major,
                         const float* w1x, const float* new_xyz, const int* idx, const float* w_hidden, const float* b_hidden,
                         const float* w_out, const float* b_out, float* out, size_t out_frame_stride, jm_stream_t stream);

/* A stack of 1..3 kernel-size-1 Conv1d layers (BatchNorm folded by the caller, optional ReLU each) on (B, C, n) tensors in
 * one launch: the RPN heads (rpn.py:34-58), the feature-propagation SharedMLPs on cat[interpolated, skip]
 * (pointnet2_modules.py:139-153) and the hoisted first set-abstraction layer u = W_f f + W_x xyz^T.  The first layer
 * takes one or two operands that accumulate into the same output columns (x0 (B, c0, n) and x1 (B, c1, n), i.e. the
 * channel concatenation without building it); xyz1 != 0: x1 is point-major coordinates (B, n, 3), c1 == 3.
 * widths[l] = output channels of layer l; weights in the layout of jm_sa_mlp_pack(cout, cin, 0): w0a (widths[0] x c0),
 * w0b (widths[0] x c1) or NULL, weights[1..] (weights[0] ignored), biases[0..]; relu[l] != 0 applies ReLU after layer l.
 * out (B, widths[num_layers-1], n), or (B, n, widths[num_layers-1]) when out_point_major != 0 (the layout
 * jm_sa_mlp_pm_forward gathers).  n % 32 == 0; operands whose width is not a multiple of 16 and the hidden
 * activations of a 32-point tile must fit the 160 KB LDS (jm_conv1d_stack_supported). */
int jm_conv1d_stack_supported(int b, int n, int c0, int c1, int xyz1, int num_layers, const int* widths);
int jm_conv1d_stack_forward(int b, int n, int c0, const float* x0, int c1, const float* x1, int xyz1, int num_layers,
                            const int* widths, const float* w0a, const float* w0b, const float* const* weights,
                            const float* const* biases, const int* relu, int out_point_major, float* out,
                            jm_stream_t stream);

/* The image branch's first layer in one pass: out = relu(conv3x3(image, padding 1, stride 1) + bias) for a THREE-channel
 * image (backbone.py:16-32, Img_Block[0].conv1 + bn1 + relu with the eval-mode BatchNorm folded by the caller).
 * image (B, 3, H, W) NCHW; weight_tap_major (27, cout) = weight (cout, 3, 3, 3) permuted to [c][dy][dx] x cout; bias (cout);
 * out (B, H, W, cout) = a channels-last (B, cout, H, W) tensor.  The layer writes 1 GB at 384 x 1280 x 8 frames and has
 * K = 27: one HBM-bound pass instead of convolution + bias/ReLU pass.  cout % 4 == 0 and <= 32, or 64, or 128. */
int jm_conv3x3_rgb_bias_relu(int b, int h, int w, int cout, const float* image, const float* weight_tap_major,
                             const float* bias, float* out_channels_last, jm_stream_t stream);

/* order (b, n) int64 = the indices that sort each row of scores (b, n) in DESCENDING order, equal scores in index order:
 * torch.sort(scores, dim=1, descending=True, stable=True)[1] of proposal_layer.py:45 (the score order in front of the distance
 * bands and the NMS walk) as one workgroup per row in LDS instead of a 14-kernel merge sort.  n <= 16384
 * (jm_argsort_desc_supported); -0.0 orders as +0.0; NaNs by bit pattern (+NaN above +inf, -NaN below -inf), as torch.sort on this platform. */
int jm_argsort_desc_supported(int n);
int jm_argsort_desc_stable(int b, int n, const float* scores, long long* order, jm_stream_t stream);

/* The image branch's stride-1 3x3 convolutions with their folded BatchNorm bias and ReLU in one kernel, as a fused Winograd
 * F(2x2, 3x3) (backbone.py:16-32: BasicBlock.conv1 + bn1 + relu of Img_Block[1..3]; 2.25x fewer multiplications than the
 * direct form, fp32 throughout, transformed tensors never leave the CU — csrc/conv_wino.hip).
 * x (B, H, W, cin) and out (B, H, W, cout) = channels-last (B, C, H, W) tensors; padding 1, stride 1; cin % 16 == 0,
 * cout % 64 == 0 (jm_conv3x3_wino_supported).  `packed` = jm_conv3x3_wino_pack of the (cout, cin, 3, 3) weight
 * (16 * cin * cout floats: U = G g G^T in MFMA operand order, made once per weight); bias (cout) or NULL; relu != 0 applies it. */
size_t jm_conv3x3_wino_packed_elems(int cin, int cout);
int jm_conv3x3_wino_supported(int cin, int cout);
int jm_conv3x3_wino_pack(int cin, int cout, const float* weight, float* packed, jm_stream_t stream);
int jm_conv3x3_wino_bias_relu(int b, int h, int w, int cin, int cout, const float* x_channels_last, const float* packed,
                              const float* bias, int relu, float* out_channels_last, jm_stream_t stream);

/* x = relu(x + bias[c]) in place on CHANNELS-LAST data (numel = pixels * channels, channels % 4 == 0): the one
 * element-wise pass of the image branch's BasicBlock (backbone.py:16-32) once its eval-mode BatchNorm is folded into
 * the first convolution (replaces a BatchNorm pass + a ReLU pass over up to 1 GB). */
int jm_bias_relu_channels_last(long long numel, int channels, float* x, const float* bias, jm_stream_t stream);

/* jm_conv1d_stack_forward on 64-POINT tiles (csrc/conv1d_stack64.hip, round 5): the same stacks (rpn.py:34-58 heads,
 * pointnet2_modules.py:139-153 feature-propagation MLPs, the hoisted first set-abstraction layer) with the A operand row-major in LDS
 * and the weights streamed in MFMA B-operand order, each weight register feeding two matrix instructions.  Weights are packed per layer
 * with jm_conv1d_stack64_pack: layer 0 from the (n_out, c0 + c1) matrix on the concatenated input, layer l > 0 from (n_out, widths[l-1]);
 * packed holds jm_conv1d_stack64_packed_elems(n_out, k) floats, bias_padded ceil(n_out / 32) * 32 floats (bias NULL: zeros).
 * n % 64 == 0; _supported says whether the tiles fit the LDS (else: jm_conv1d_stack_forward). */
int jm_conv1d_stack64_supported(int b, int n, int c0, int c1, int xyz1, int num_layers, const int* widths);
size_t jm_conv1d_stack64_packed_elems(int n_out, int k);
int jm_conv1d_stack64_pack(int n_out, int k, const float* w, int ldw, const float* bias, float* packed, float* bias_padded,
                           jm_stream_t stream);
int jm_conv1d_stack64_forward(int b, int n, int c0, const float* x0, int c1, const float* x1, int xyz1, int num_layers,
                              const int* widths, const float* const* packed, const float* const* biases_padded, const int* relu,
                              int out_point_major, float* out, jm_stream_t stream);

/* A dense layer on a (B, C, n) per-point tensor with FEW points (the coarse end of the backbone: feature propagation level 4,
 * pointnet2_modules.py:139-153, and the LI-Fusion attention block of level 4, backbone.py:35-81), one launch of independent waves
 * (csrc/points_gemm.hip): out = act(W [x1 ; x2] + bias) (* rowscale per point).  x1 (B,k1,n), x2 (B,k2,n) or NULL (k2 = 0): the
 * channel concatenation is never built; w (n_out, >= k1 + k2) row-major with leading dimension ldw; act 0 none, 1 ReLU, 2 tanh,
 * 3 sigmoid; rowscale[(b n + p) * rowscale_stride] or NULL; out (B,n_out,n), or point-major rows (B n, ldo) with out_rows.
 * n % 32 == 0, (k1 + k2) % 4 == 0, ldw % 4 == 0. */
int jm_points_linear_supported(int b, int n, int k1, int k2, int n_out);
int jm_points_linear(int b, int n, int k1, const float* x1, int k2, const float* x2, int n_out, const float* w, int ldw,
                     const float* bias, int act, const float* rowscale, int rowscale_stride, int out_rows, int ldo, float* out,
                     jm_stream_t stream);

/* Glue passes of the composed detector as single launches (round 5).
 * jm_three_nn_weights: dist2 (rows, 3) squared distances of jm_three_nn -> weight (rows, 3) = the normalised inverse distances of
 *   PointnetFPModule.forward (pointnet2_modules.py:148-150: dist = sqrt(dist2), 1 / (dist + 1e-8), divided by their sum).
 * jm_gather_point_rows: out (B,m,width) = src (B,n,width) rows at idx (B,m) int32 — `torch.gather(l_xy, 1, li_index...)` of
 *   backbone.py:170-171 on the int32 FPS indices.
 * jm_pts_feature: the RoI-pooling input of the RCNN stage (point_rcnn.py:42-44, proposal_target_layer.py:26): rpn_cls (B,N) logits
 *   at rpn_cls[b * batch_stride_cls + p * ld_cls] (the heads' own channel-major output is read in place), xyz (B,N,3), feats (B,C,N) -> out (B,N,2+C) = [sigmoid(cls) > score_thresh, |xyz| / 70 - 0.5,
 *   features point-major]. */
int jm_three_nn_weights(long long rows, const float* dist2, float* weight, jm_stream_t stream);
int jm_gather_point_rows(int b, int n, int m, int width, const float* src, const int* idx, float* out, jm_stream_t stream);
int jm_pts_feature(int b, int n, int c, const float* rpn_cls, long long batch_stride_cls, int ld_cls, const float* xyz, const float* feats,
                   float score_thresh, float* out, jm_stream_t stream);

/* The final LI-Fusion image feature AT THE POINTS (jmodt/detection/modeling/backbone.py:187-195):
 *   feature_gather(relu(bn(conv1x1(cat_i deconv_i(img_i)))), xy)
 * without the (B, q, H, W) map: only the pixels under a bilinear tap are evaluated (sorted by sub-pixel phase, one
 * fp32-MFMA tile of 32 taps per wave).  maps[i] (B, H/k_i, W/k_i, C_i) CHANNELS-LAST, C_i % 16 == 0, strides k_i
 * (powers of two <= 16, kernel == stride of the level's ConvTranspose2d); packed_weights[i] from
 * jm_image_fusion_pack(C_i, q, k_i, wc_i) with wc_i (C_i, q, k_i, k_i) = the deconvolution weight composed with the
 * level's slice of the BatchNorm-folded 1x1 fusion convolution; bias32 = the folded bias (deconvolution biases
 * included), zero padded to 32; xy (B, N, 2) in [-1, 1] on the (h, w) canvas -> out (B, q, N), q <= 32. */
size_t jm_image_fusion_gather_workspace_bytes(int b, int n);
size_t jm_image_fusion_packed_elems(int cin, int k);
int jm_image_fusion_pack(int cin, int q, int k, const float* wc, float* wp, jm_stream_t stream);
int jm_image_fusion_gather(int b, int n, int h, int w, int q, int num_levels, const int* channels, const int* strides,
                           const float* const* maps, const float* const* packed_weights, const float* bias32,
                           const float* xy, float* out, void* ws, size_t ws_bytes, jm_stream_t stream);

/* LI-Fusion attention block (jmodt/detection/modeling/backbone.py:35-81, AttentionFusion + IALayer, eval mode):
 *   att = sigmoid(fc3(tanh(fc1(I^T) + fc2(P^T))));  G = relu(bn(conv1(I))) * att;  out = relu(bn1(conv1(cat[P, G])))
 * img_feats I (B, ic, n), point_feats P (B, pc, n) -> out (B, oc, n), one launch (the reference: ~16 kernels).
 * All matrices in the device layout of jm_sa_mlp_pack(cout, cin, first_layer = 0): w_fc1 (rc x ic), w_fc2 (rc x pc),
 * w_img (pc x ic, BatchNorm folded), w_fuse_point / w_fuse_img = the two column halves of the fusion convolution
 * (oc x pc each, BatchNorm folded); b_fc12 = b_fc1 + b_fc2, b_img, b_fuse = packed biases (pad128); w_fc3 (rc) plain,
 * b_fc3 by value.  n % 32 == 0 and the 32-point tile of [I | P | T | G] must fit the LDS (jm_attention_fusion_supported). */
int jm_attention_fusion_supported(int b, int n, int ic, int pc, int rc, int oc);
int jm_attention_fusion_forward(int b, int n, int ic, int pc, int rc, int oc, const float* img_feats,
                                const float* point_feats, const float* w_fc1, const float* w_fc2, const float* b_fc12,
                                const float* w_fc3, float b_fc3, const float* w_img, const float* b_img,
                                const float* w_fuse_point, const float* w_fuse_img, const float* b_fuse, float* out,
                                jm_stream_t stream);

/* ------------------------------------------------------------------ affinity head --------- */

/* link_layer / se_layer MLP (jmodt/detection/modeling/rcnn.py:91-111):
 *   Conv1d(C,H1)+ReLU -> Conv1d(H1,H2)+ReLU -> Conv1d(H2,1), bias, kernel size 1.
 * W1 (H1,C), b1 (H1), W2 (H2,H1), b2 (H2), w3 (H2), b3 (1): all device pointers. */
typedef struct {
    int c, h1, h2;
    const float *w1, *b1, *w2, *b2, *w3, *b3;
} jm_mlp3_t;

/* Inference-time pairwise affinity (jmodt/tracking/tracker.py:81-112; training form
 * rcnn.py:239-258).  pred_feat (P,C), det_feat (D,C) ->
 *   link_raw (P,D)  raw link scores S              (may be NULL)
 *   link     (P,D)  (softmax(S,dim=1)+softmax(S,dim=0))/2
 *   start    (D)    se(mean_i |p_i-d_j|)  raw logit (tracker applies w_se*sigmoid)
 *   end      (P)    se(mean_j |p_i-d_j|)  raw logit
 * The (P*D,C) pair tensor is never materialised.  fp32 MFMA, exact-f32 products.
 * Everything runs on `stream`.  The start/end head is a short, latency-bound chain of its own: a caller that wants
 * it to overlap the link head calls jm_affinity_start_end on a second stream it owns (se = NULL here) — the library
 * itself keeps no streams, events or other per-device state. */
size_t jm_affinity_workspace_bytes(int p, int d, const jm_mlp3_t* link, const jm_mlp3_t* se);
size_t jm_affinity_start_end_workspace_bytes(int p, int d, const jm_mlp3_t* se);
int jm_affinity_start_end(int p, int d, const float* pred_feat, const float* det_feat, const jm_mlp3_t* se,
                          float* start, float* end, void* ws, size_t ws_bytes, jm_stream_t stream);
int jm_affinity_forward(int p, int d, const float* pred_feat, const float* det_feat, const jm_mlp3_t* link,
                        const jm_mlp3_t* se, float* link_raw, float* link_out, float* start, float* end, void* ws,
                        size_t ws_bytes, jm_stream_t stream);

/* Batched forms: nb independent (P, D) problems stacked on the leading axis — pred_feat (nb, P, C), det_feat (nb, D, C) ->
 * link_raw / link (nb, P, D), se_out (nb, D + P) = per problem [start logits (D) | end logits (P)].  The detector scores
 * every frame of a batch against its predecessor (tracker.py:81-112 per frame pair): one GEMM chain over nb*P*D pair
 * rows instead of nb chains. */
size_t jm_affinity_batched_workspace_bytes(int nb, int p, int d, const jm_mlp3_t* link);
int jm_affinity_forward_batched(int nb, int p, int d, const float* pred_feat, const float* det_feat, const jm_mlp3_t* link,
                                float* link_raw, float* link_out, void* ws, size_t ws_bytes, jm_stream_t stream);
size_t jm_affinity_start_end_batched_workspace_bytes(int nb, int p, int d, const jm_mlp3_t* se);
int jm_affinity_start_end_batched(int nb, int p, int d, const float* pred_feat, const float* det_feat, const jm_mlp3_t* se,
                                  float* se_out, void* ws, size_t ws_bytes, jm_stream_t stream);

/* the dual softmax of jm_affinity_forward_batched on its own: link = (softmax(S, dim 2) + softmax(S, dim 1)) / 2;
 * stats: 2 * nb * (P + D) floats of scratch */
int jm_affinity_dual_softmax_batched(int nb, int p, int d, const float* link_raw, float* link_out, float* stats,
                                     jm_stream_t stream);

/* EXPERIMENTAL, opt-in (csrc/affinity_x3.hip): the raw link scores S (nb, P, D) with every fp32 product evaluated on the
 * bf16 matrix pipe as a 3-term split (six bf16 products per fp32 product, fp32 accumulate): same error against fp64 as the
 * exact-fp32 kernels (tests assert it), 2.5 x the matrix-pipe rate.  Not used by any default path. */
size_t jm_affinity_x3_workspace_bytes(int nb, int p, int d, const jm_mlp3_t* link);
int jm_affinity_link_scores_x3(int nb, int p, int d, const float* pred_feat, const float* det_feat, const jm_mlp3_t* link,
                               float* link_raw, void* ws, size_t ws_bytes, jm_stream_t stream);

/* One dense layer on plain rows: y (M, N) = act(x (M, K) w^T + b), w (N, K) row-major = a Conv1d(k=1) / Linear weight,
 * relu != 0 applies ReLU.  For the small-M heads (RCNN cls_layer / reg_layer, rcnn.py:43-89: 1024 RoIs x 512): single-wave
 * 32x32 fp32-MFMA tiles, one launch per layer instead of GEMM + bias + ReLU.  K % 8 == 0. */
int jm_linear_rows(int m, int k, int n, const float* x, const float* w, const float* b, float* y, int relu,
                   jm_stream_t stream);

/* The same MLP on plain rows x (M,C) -> y (M): used for the start/end features and exposed for
 * callers that hold a materialised feature matrix (rcnn.py:272-285). */
size_t jm_mlp3_workspace_bytes(int m, const jm_mlp3_t* mlp);
int jm_mlp3_forward(int m, const float* x, const jm_mlp3_t* mlp, float* y, void* ws, size_t ws_bytes,
                    jm_stream_t stream);

/* ------------------------------------------------------------------ duplicate-aware set abstraction (exact) ---- */

/* jm_roipool3d_canonical + pooled_cnt (B, M) i32: the number of distinct source points of every pooled slab (0: empty RoI;
 * else rows cnt .. S-1 repeat rows 0 .. cnt-1 cyclically, roipool3d_kernel.cu:123-160). */
int jm_roipool3d_canonical_cnt(int batch_size, int pts_num, int boxes_num, int feature_in_len, int sampled_pts_num,
                               const float* xyz, const float* rois, float extra_width, const float* pts_feature,
                               float* pooled_features, int* pooled_empty_flag, int* pooled_cnt, jm_stream_t stream);

/* sa_mlp_pm_kernel (jm_sa_mlp_pm_forward) on a row set whose size lives in DEVICE memory: one frame of n points, capacity m
 * (virtual) centres, tiles_dev[0] = number of 128-row tiles to run. */
int jm_sa_mlp_pm_forward_dyn(int n, int m, int c, int nsample, int hidden, int cout, const float* u_point_major,
                             const float* w1x, const float* new_xyz, const int* idx, const float* w_hidden, const float* b_hidden,
                             const float* w_out, const float* b_out, float* out, const int* tiles_dev, jm_stream_t stream);

/* Exact removal of repeated rows from a set-abstraction scale over r point sets of n points (csrc/sa_dedupe.hip): rows
 * whose neighbour is a copy of an earlier neighbour of the same centre (cyclic roipool padding, ball-query back-fill) and
 * centres that are copies of an earlier centre contribute nothing to the max-pool.
 *   canon (r, n) i32        canon[k] = first point of which point k is an exact copy (canon_from_cnt: k % max(cnt, 1))
 *   fps_idx (r, m), nb (r, m, nsample), new_xyz (r, m, 3)   the scale's sampling, neighbour lists and centres
 * plan ->  rep (r, m)       first centre with the same canonical point (the next level's canon)
 *          seg_start / seg_cnt (r, m)   the representative's segments in the virtual-centre arrays
 *          vidx (cap, 16) i32 GLOBAL point indices (set * n + point), vxyz (cap, 3): a 16-sample problem over all sets;
 *          cap = jm_sa_dedupe_capacity(r, m, nsample)
 *          counters (4) i32, device: [virtual centres, tiles (-> jm_sa_mlp_pm_forward_dyn), virtual centres, -]
 * combine: out (r, cout, m) = per centre the max over its representative's segments of out_virtual (cout, cap). */
int jm_sa_dedupe_canon_from_cnt(int r, int n, const int* cnt, int* canon, jm_stream_t stream);
long long jm_sa_dedupe_capacity(int r, int m, int nsample);
int jm_sa_dedupe_plan(int r, int n, int m, int nsample, const int* canon, const int* fps_idx, const int* nb, const float* new_xyz,
                      int* rep, int* seg_start, int* seg_cnt, int* vidx, float* vxyz, int* counters, jm_stream_t stream);
int jm_sa_dedupe_combine(int r, int m, int cout, long long vmax, const float* out_virtual, const int* rep, const int* seg_start,
                         const int* seg_cnt, float* out, jm_stream_t stream);

/* ------------------------------------------------------------------ training-time affinity ---- */

/* Gradients of one 3-layer head, same shapes as jm_mlp3_t's tensors: dw1 (h1, c), db1 (h1), dw2 (h2, h1), db2 (h2),
 * dw3 (h2), db3 (1).  Written (not accumulated). */
typedef struct {
    float *dw1, *db1, *dw2, *db2, *dw3, *db3;
} jm_mlp3_grad_t;

/* Training-time pairwise affinity for `npairs` (prev, next) frame pairs of r RoI slots each, static shapes, no host
 * synchronisation (jmodt/detection/modeling/rcnn.py:145-156,204-287; losses train_functions.py:282-329, L1 form).
 *
 * jm_affinity_train_prepare: feats (2*npairs, r, c) and tids (2*npairs, r) with frames interleaved prev, next, prev, ...
 *   (rcnn.py:212-217; tid <= 0 = background) ->
 *   pooled_prev / pooled_next (npairs*r, c)  mean feature of the foreground RoIs sharing the slot's track id
 *                                            (get_unique_tid_feature, rcnn.py:145-156), zero for background slots
 *   rep_prev / rep_next (npairs, r) i32      1 = the slot is the first foreground RoI of its id (one row of the reference's
 *                                            unique-id tensors) and the pair has foreground on both sides (rcnn.py:230)
 *   n_pair (npairs, 2) i32                   representatives per pair [prev, next]
 *   gt_starts / gt_ends (npairs, r)          1 - column / row sums of the tid-equality matrix (rcnn.py:247-252)
 *   counts (3) f32, DEVICE                   LOCAL element counts of the three loss means [links, starts, ends]; a
 *                                            data-parallel caller all-reduces them before the step calls (the reference
 *                                            takes its means over the gathered batch)
 *   rep_ws: (2*npairs, r) i32 scratch.
 * jm_affinity_train_link_step: link head forward on all npairs*r*r slot pairs (|prev_i - next_j| formed on the fly),
 *   masked dual softmax, loss_part[f] = sum over valid entries |link - gt| (divide by counts[0] for the mean), and the
 *   gradients of loss_weight * sum(loss_part) / max(counts[0], 1) w.r.t. the six tensors of `link`.
 *   link_out / gt_links (npairs, r, r) optional (zero outside rep_prev x rep_next).
 * jm_affinity_train_se_step: masked start / end feature means, se head, sigmoid + L1; se_logits (npairs, 2r) = per pair
 *   [start logits of the next slots | end logits of the prev slots] (optional); loss_part (npairs, 2) = [sum |sigmoid(start) -
 *   gt|, sum |sigmoid(end) - gt|]; gradients of loss_weight * (sum_start / max(counts[1], 1) + sum_end / max(counts[2], 1)).
 * The two step calls are independent of each other (a caller may run them on two streams).
 * Gradients w.r.t. the RoI features (joint training; the finetune step of tools/train.py:96-107 freezes the detector and
 * passes dx = NULL): dx (rows, c) of a step receives d(loss)/d(its input rows) — the (npairs * r * r) pair rows |p_i - d_j|
 * of the link head, the (npairs * 2r) start / end feature rows of the se head —, and jm_affinity_train_feature_grad carries
 * both through |.|, the masked means and the per-track-id mean pooling to dfeat (2 npairs, r, c) = d(loss)/d(feats);
 * dpooled_ws: 2 * npairs * r * c floats of scratch; dx_se may be NULL (link loss only). */
int jm_affinity_train_prepare(int npairs, int r, int c, const float* feats, const float* tids, float* pooled_prev,
                              float* pooled_next, int* rep_ws, int* rep_prev, int* rep_next, int* n_pair, float* gt_starts,
                              float* gt_ends, float* counts, jm_stream_t stream);
/* loss[0] = link_weight * sum(link_loss_part) / max(counts[0], 1) + se_weight * (sum(se_loss_part[:, 0]) / max(counts[1], 1) +
 * sum(se_loss_part[:, 1]) / max(counts[2], 1))   (train_functions.py:282-329: the three L1 means, weighted) */
int jm_affinity_train_loss_value(int npairs, const float* link_loss_part, const float* se_loss_part, const float* counts,
                                 float link_weight, float se_weight, float* loss, jm_stream_t stream);
size_t jm_affinity_train_link_workspace_bytes(int npairs, int r, const jm_mlp3_t* link);
int jm_affinity_train_link_step(int npairs, int r, const float* pooled_prev, const float* pooled_next, const int* rep_prev,
                                const int* rep_next, const float* tids, const float* counts, float loss_weight,
                                const jm_mlp3_t* link, float* link_out, float* gt_links, float* loss_part,
                                const jm_mlp3_grad_t* grads, float* dx, void* ws, size_t ws_bytes, jm_stream_t stream);
size_t jm_affinity_train_se_workspace_bytes(int npairs, int r, const jm_mlp3_t* se);
int jm_affinity_train_se_step(int npairs, int r, const float* pooled_prev, const float* pooled_next, const int* rep_prev,
                              const int* rep_next, const int* n_pair, const float* gt_starts, const float* gt_ends,
                              const float* counts, float loss_weight, const jm_mlp3_t* se, float* se_logits, float* loss_part,
                              const jm_mlp3_grad_t* grads, float* dx, void* ws, size_t ws_bytes, jm_stream_t stream);
int jm_affinity_train_feature_grad(int npairs, int r, int c, const float* tids, const float* pooled_prev,
                                   const float* pooled_next, const int* rep_prev, const int* rep_next, const int* n_pair,
                                   const float* dx_link, const float* dx_se, float* dpooled_ws, float* dfeat,
                                   jm_stream_t stream);

/* ------------------------------------------------------------------ training path on rows ---- */
/* The joint-mode training step (tools/train.py:96-107 without cfg.TRAIN.FINETUNE; point_rcnn.py:24-70 in TRAIN mode) keeps every
 * per-point / per-(centre, neighbour) activation as a ROW-major (rows, channels) float32 tensor.  These entries are the forward
 * AND backward of what the reference runs through torch autograd there: SharedMLP / Conv1d stacks (pytorch_utils.py:6-33),
 * set abstraction (pointnet2_modules.py:46-61 with pointnet2_utils.py:156-197,231-290), feature propagation (:139-153 with
 * pointnet2_utils.py:105-150), the LI-Fusion gather and attention block (backbone.py:35-89).  `*_dev` row counts live in device
 * memory (NULL: the host bound is the count); `ld*` = floats between consecutive rows; widths and ld* are multiples of 4. */

/* y (m, n) = act(x1 (m, k1) w[:, :k1]^T + x2 (m, k2) w[:, k1:]^T + bias) * rowscale[row];  w (n, k1 + k2) rows ldw apart
 * (an nn.Conv1d / Conv2d 1x1 / Linear weight as it is); x2 / bias / rowscale may be NULL (k2 = 0); act 0 none, 1 ReLU, 2 tanh.
 * The two-operand form is the concatenation of pointnet2_modules.py:157-160 / backbone.py:78 without the copy. */
int jm_rows_linear_forward(int m, const int* m_dev, int k1, int k2, int n, const float* x1, int ldx1, const float* x2, int ldx2,
                           const float* w, int ldw, const float* bias, int act, const float* rowscale, float* y, int ldy,
                           jm_stream_t stream);
/* dx (m, k) = (dy (m, n) w (n, k)) .* (mask (m, k) > 0), optionally added to dx: what autograd computes for conv + ReLU of
 * the layer below (mask = that layer's output); mask may be NULL; w + k1 with the same ldw addresses the second operand */
int jm_rows_linear_dgrad(int m, const int* m_dev, int n, int k, const float* dy, int lddy, const float* w, int ldw,
                         const float* mask, int ldm, int accumulate, float* dx, int lddx, jm_stream_t stream);
/* dw (n, k) (+)= dy (m, n)^T x (m, k), dbias (n) (+)= column sums of dy (NULL: skipped).  The m rows are split over
 * jm_rows_wgrad_splits(m, n, k) partials in ws (with m_dev: only the splits that hold >= 128 of the counted rows exist), reduced in a
 * FIXED order (eight groups of every eighth split, the groups added in group order): deterministic, no float atomics; a short
 * contraction (one split) writes dw / dbias straight from the accumulators and needs no workspace.  Direct form since round 5:
 * the operands go from their rows into the MFMA registers, a wave's patch is 32 or 64 columns of dy and of x (csrc/rows_gemm.hip) */
int jm_rows_wgrad_splits(int m, int n, int k);
size_t jm_rows_wgrad_workspace_bytes(int m, int n, int k);
int jm_rows_linear_wgrad(int m, const int* m_dev, int n, int k, const float* dy, int lddy, const float* x, int ldx,
                         float* dw, int lddw, float* dbias, int accumulate, void* ws, size_t ws_bytes, jm_stream_t stream);
/* out (n) (+)= column sums of x (m, n); ws >= jm_rows_reduce_workspace_bytes(n) */
size_t jm_rows_reduce_workspace_bytes(int n);
int jm_rows_colsum(int m, const int* m_dev, int n, const float* x, int ldx, float* out, int accumulate, void* ws, size_t ws_bytes,
                   jm_stream_t stream);
/* out = y > 0 ? dy : 0 (the ReLU of a block's last layer, whose gradient arrives from outside the block); out may be dy */
int jm_rows_relu_mask(int m, const int* m_dev, int n, const float* dy, int ldd, const float* y, int ldy, float* out, int ldo, jm_stream_t stream);

/* Set-abstraction rows: the DISTINCT (centre, neighbour) pairs of every group.  idx (groups, ns) int32 = ball_query's lists
 * (ns <= 64), entries local to their point set (frame / RoI) of n_per_set points; group g belongs to set g / groups_per_set;
 * canon (sets, n_per_set) int32 or NULL maps a point to the first point it is an exact copy of (cyclic roipool3d padding).
 * d (groups): rows per group; offsets (groups + 1): exclusive scan, offsets[groups] = the row count R (device memory);
 * row_point (R) = set * n_per_set + canonical entry, row_group (R) = g; buffers sized groups * ns.  Three launches. */
int jm_sa_rows_plan(int groups, int ns, const int* idx, const int* canon, int n_per_set, int groups_per_set, int* d, int* offsets,
                    int* row_point, int* row_group, jm_stream_t stream);
/* h1[r, :] = relu((u ? u[row_point[r], :] : b1) + w1x (h, 3) (xyz[row_point[r]] - ctr[row_group[r]])): the first SharedMLP layer
 * on [xyz_j - c_i ; f_j] (pointnet2_utils.py:259-269) with its feature part u = W1f f + b1 computed per point; ctr NULL = GroupAll.
 * delta (rows, 4) or NULL: also stores [xyz_j - c_i, 0] per row — the operand of the backward's d(w1x) = dh1^T delta
 * (jm_rows_linear_wgrad with k = 4, whose bias output is d(b1)) */
int jm_sa_rows_h1(int rows, const int* rows_dev, int h, const float* u, int ldu, const float* b1, const float* w1x, const float* xyz,
                  const float* ctr, const int* row_point, const int* row_group, float* h1, int ldh, float* delta, jm_stream_t stream);
/* out (groups, c) = max over each group's rows (F.max_pool2d, pointnet2_modules.py:50-55), argrow (groups, c) = the first row holding it */
int jm_sa_rows_pool(int groups, int c, const float* h, int ldh, const int* offsets, float* out, int ldo, int* argrow, jm_stream_t stream);
/* dh (rows, c) = d(out) routed to the arg-max rows where the pooled (post-ReLU) value is positive, 0 elsewhere */
int jm_sa_rows_pool_grad(int rows, const int* rows_dev, int c, const float* dout, int lddo, const float* out, int ldo, const int* argrow,
                         const int* row_group, float* dh, int ldd, jm_stream_t stream);
/* du[row_point[r], :] += dh1[r, :] (the grouping_operation backward of group_points_gpu.cu:48-86 on the compacted rows) */
int jm_sa_rows_scatter_add(int rows, const int* rows_dev, int h, const float* dh1, int ldd, const int* row_point, float* du, int ldu,
                           jm_stream_t stream);
/* dw1x (h, 3) (+)= sum_r dh1[r, :]^T (xyz[row_point[r]] - ctr[row_group[r]]); ws >= jm_rows_reduce_workspace_bytes(3 h) */
int jm_sa_rows_xyz_wgrad(int rows, const int* rows_dev, int h, const float* dh1, int ldd, const float* xyz, const float* ctr,
                         const int* row_point, const int* row_group, float* dw1x, int accumulate, void* ws, size_t ws_bytes, jm_stream_t stream);

/* three_interpolate (interpolate_gpu.cu:77-161) on rows: known (b m, c) -> out (b n, c); the gradient adds into dknown (pre-zeroed) */
int jm_three_interpolate_rows(int b, int n, int m, int c, const float* known, int ldk, const int* idx, const float* w, float* out, int ldo,
                              jm_stream_t stream);
int jm_three_interpolate_rows_grad(int b, int n, int m, int c, const float* dout, int ldo, const int* idx, const float* w, float* dknown, int ldk,
                                   jm_stream_t stream);
/* feature_gather (backbone.py:79-89) from a channels-last map (b, h, w, c) to rows (b n, c); the gradient adds into a
 * channels-last map (pre-zeroed) */
int jm_feature_gather_rows(int b, int c, int h, int w, int n, const float* fmap_cl, const float* xy, float* out, int ldo, jm_stream_t stream);
int jm_feature_gather_rows_grad(int b, int c, int h, int w, int n, const float* grad_out, int ldo, const float* xy, float* grad_fmap_cl,
                                jm_stream_t stream);
/* IA_Layer gate (backbone.py:54-62): g[r] = sigmoid(z[r * ldz]); and the backward through J = relu(conv1(img)) * g:
 * dj (m, pc) <- gradient w.r.t. conv1's pre-activation (in place), dz (m, 4) = [d(gate logit), 0, 0, 0], dt (m, rc) = gradient
 * w.r.t. the tanh's argument; j = the gated output, t = the tanh output, w3 (rc) = fc3's weight */
int jm_rows_sigmoid(int m, const float* z, int ldz, float* g, jm_stream_t stream);
int jm_rows_gate_backward(int m, int pc, int rc, float* dj, int ldj, const float* j, int ldjj, const float* g, const float* t, int ldt,
                          const float* w3, float* dz, float* dt, int lddt, jm_stream_t stream);

/* Whole chains of the row kernels behind ONE call (csrc/rows_chain.hip): the host-side layer loops of a dense stack and of one
 * set-abstraction scale, forward and backward, so that the caller's interpreter is not in the launch path.  Same kernels, same
 * bits as the single-layer entries above; the caller provides every buffer. */
#define JM_ROWS_MAX_LAYERS 6
typedef struct {
    int nl, m;                       /* layers, rows */
    const int* m_dev;                /* row count in device memory, or NULL */
    int k1, k2;                      /* widths of the two input operands (k2 = 0: one) */
    const float* x1; int ldx1; const float* x2; int ldx2;
    int widths[JM_ROWS_MAX_LAYERS];  /* output width of layer l */
    int acts[JM_ROWS_MAX_LAYERS];    /* 0 none, 1 ReLU, 2 tanh */
    const float* w[JM_ROWS_MAX_LAYERS]; int ldw[JM_ROWS_MAX_LAYERS]; const float* b[JM_ROWS_MAX_LAYERS];
    float* y[JM_ROWS_MAX_LAYERS];    /* (m, widths[l]) outputs, kept for the backward */
} jm_rows_mlp_t;
typedef struct {
    const float* dout; int lddout;   /* gradient w.r.t. y[nl - 1] */
    float* dw[JM_ROWS_MAX_LAYERS]; int lddw[JM_ROWS_MAX_LAYERS]; float* db[JM_ROWS_MAX_LAYERS];   /* db[l] NULL: layer without bias */
    float* dx1; float* dx2;          /* (m, k1), (m, k2) or NULL */
    float* scratch[2];               /* two (m, max width) row buffers */
    void* ws; size_t ws_bytes;       /* >= the largest jm_rows_wgrad_workspace_bytes of the stack */
} jm_rows_mlp_grad_t;
int jm_rows_mlp_forward(const jm_rows_mlp_t* d, jm_stream_t stream);
int jm_rows_mlp_backward(const jm_rows_mlp_t* d, const jm_rows_mlp_grad_t* g, jm_stream_t stream);
int jm_rows_tanh_grad(int m, const int* m_dev, int n, const float* dy, int ldd, const float* y, int ldy, float* out, int ldo, jm_stream_t stream);
typedef struct {
    int nl, groups, max_rows;        /* layers (>= 2), groups, row capacity (= groups * nsample) */
    const int* rows_dev; const int* offsets; const int* row_point; const int* row_group;   /* jm_sa_rows_plan's outputs */
    int points, c;                   /* rows of f, feature width (0: xyz only) */
    const float* f; int ldf; const float* xyz; const float* ctr;
    int widths[JM_ROWS_MAX_LAYERS];  /* output width of layer l (widths[0] = H1) */
    const float* w1x; const float* w1f; const float* b1;       /* layer 0: (H1, 3) packed, (H1, c), (H1) */
    const float* w[JM_ROWS_MAX_LAYERS]; const float* b[JM_ROWS_MAX_LAYERS];   /* layers 1 .. nl - 1, contiguous (out, in) */
    float* u; float* delta; float* h[JM_ROWS_MAX_LAYERS];      /* (points, H1); (max_rows, 4); (max_rows, widths[l]) — kept for the backward */
    float* out; int ldo; int* argrow;                          /* pooled (groups, widths[nl - 1]) rows ldo apart; (groups, widths[nl - 1]) */
} jm_sa_scale_t;
typedef struct {
    const float* dout; int lddout;
    float* dw1; float* db1; float* dw4;    /* (H1, 3 + c) on [xyz ; f]; (H1); (H1, 4) scratch */
    float* dw[JM_ROWS_MAX_LAYERS]; float* db[JM_ROWS_MAX_LAYERS];
    float* du; float* df; int df_accumulate;                   /* (points, H1) scratch; (points, c) or NULL */
    float* scratch[2]; void* ws; size_t ws_bytes;
} jm_sa_scale_grad_t;
int jm_sa_scale_forward(const jm_sa_scale_t* d, jm_stream_t stream);
int jm_sa_scale_backward(const jm_sa_scale_t* d, const jm_sa_scale_grad_t* g, jm_stream_t stream);

/* Eval-mode BatchNorm folded into the preceding convolution, for all n (convolution, BatchNorm) pairs of a network in one launch:
 * wf[l] (rows_l, cols_l) = w[l] * s[soff[l] + row] with s = gamma / sqrt(running_var + eps) (pytorch_utils.py:21-33 / backbone.py
 * conv + bn chains at inference statistics); the pointer arrays are HOST arrays of device pointers.  The backward: dw[l] = dwf[l] * s,
 * ds[soff[l] + row] = sum_col dwf[l] * w[l] (ds rows of layers not listed keep their value: zero it first) */
int jm_fold_bn_multi(int n, const float* const* w, float* const* wf, const int* rows, const int* cols, const int* soff, const float* s,
                     jm_stream_t stream);
int jm_fold_bn_multi_grad(int n, const float* const* dwf, const float* const* w, float* const* dw, const int* rows, const int* cols,
                          const int* soff, const float* s, float* ds, jm_stream_t stream);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* JMODT_HIP_H */
