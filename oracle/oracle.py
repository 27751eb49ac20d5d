"""numpy/ctypes front-end of the CPU oracle (oracle/jmodt_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under jmodt_amd/ may import this module.

All functions take/return numpy arrays (float32 / int32 unless stated) and mirror the argument
meaning of the reference's extension wrappers.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libjmodt_oracle.so")
_lib = None

_f = ctypes.POINTER(ctypes.c_float)
_i = ctypes.POINTER(ctypes.c_int)
_l = ctypes.POINTER(ctypes.c_int64)
_u = ctypes.POINTER(ctypes.c_uint64)


def build(force: bool = False) -> None:
    src = os.path.join(_HERE, "jmodt_oracle.c")
    hdr = os.path.join(_HERE, "..", "include", "jm_detmath.h")
    stale = (not os.path.isfile(_SO)) or os.path.getmtime(_SO) < max(os.path.getmtime(src), os.path.getmtime(hdr))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libjmodt_oracle.so"], stdout=subprocess.DEVNULL)


def _cpu_has_fma() -> bool:
    try:
        with open("/proc/cpuinfo") as fh:
            return " fma " in fh.read()
    except OSError:
        return True


def lib():
    global _lib
    if _lib is None:
        build()
        if not _cpu_has_fma():  # .so may have been built with -mfma on another host
            build(force=True)
        _lib = ctypes.CDLL(_SO)
        _lib.orc_nms.restype = ctypes.c_int
        _lib.orc_opt_n_threads.restype = ctypes.c_int
    return _lib


def _p(a, t):
    return a.ctypes.data_as(t)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


# ---------------------------------------------------------------- pointnet2
def opt_n_threads(n):
    return lib().orc_opt_n_threads(int(n))


def furthest_point_sample(xyz, npoint, return_temp=False):
    """idx (B, npoint) int32 [, temp (B, N): the running min-distance buffer as the kernel leaves it]"""
    xyz = _f32(xyz)
    B, N, _ = xyz.shape
    temp = np.full((B, N), 1e10, dtype=np.float32)
    idx = np.zeros((B, npoint), dtype=np.int32)
    lib().orc_furthest_point_sampling(B, N, int(npoint), _p(xyz, _f), _p(temp, _f), _p(idx, _i))
    return (idx, temp) if return_temp else idx


def gather_operation(features, idx):
    features, idx = _f32(features), _i32(idx)
    B, C, N = features.shape
    M = idx.shape[1]
    out = np.empty((B, C, M), dtype=np.float32)
    lib().orc_gather_points(B, C, N, M, _p(features, _f), _p(idx, _i), _p(out, _f))
    return out


def gather_operation_grad(grad_out, idx, N):
    grad_out, idx = _f32(grad_out), _i32(idx)
    B, C, M = grad_out.shape
    g = np.zeros((B, C, N), dtype=np.float32)
    lib().orc_gather_points_grad(B, C, N, M, _p(grad_out, _f), _p(idx, _i), _p(g, _f))
    return g


def ball_query(radius, nsample, xyz, new_xyz):
    xyz, new_xyz = _f32(xyz), _f32(new_xyz)
    B, N, _ = xyz.shape
    M = new_xyz.shape[1]
    idx = np.zeros((B, M, nsample), dtype=np.int32)
    lib().orc_ball_query(B, N, M, ctypes.c_float(radius), int(nsample), _p(new_xyz, _f), _p(xyz, _f), _p(idx, _i))
    return idx


def grouping_operation(features, idx):
    features, idx = _f32(features), _i32(idx)
    B, C, N = features.shape
    _, P, S = idx.shape
    out = np.empty((B, C, P, S), dtype=np.float32)
    lib().orc_group_points(B, C, N, P, S, _p(features, _f), _p(idx, _i), _p(out, _f))
    return out


def grouping_operation_grad(grad_out, idx, N):
    grad_out, idx = _f32(grad_out), _i32(idx)
    B, C, P, S = grad_out.shape
    g = np.zeros((B, C, N), dtype=np.float32)
    lib().orc_group_points_grad(B, C, N, P, S, _p(grad_out, _f), _p(idx, _i), _p(g, _f))
    return g


def three_nn(unknown, known):
    """returns (dist2, idx): SQUARED distances as the kernel writes them (the sqrt is applied by
    the Python wrapper, pointnet2_utils.py:98)"""
    unknown, known = _f32(unknown), _f32(known)
    B, N, _ = unknown.shape
    M = known.shape[1]
    d2 = np.empty((B, N, 3), dtype=np.float32)
    idx = np.empty((B, N, 3), dtype=np.int32)
    lib().orc_three_nn(B, N, M, _p(unknown, _f), _p(known, _f), _p(d2, _f), _p(idx, _i))
    return d2, idx


def three_interpolate(features, idx, weight):
    features, idx, weight = _f32(features), _i32(idx), _f32(weight)
    B, C, M = features.shape
    N = idx.shape[1]
    out = np.empty((B, C, N), dtype=np.float32)
    lib().orc_three_interpolate(B, C, M, N, _p(features, _f), _p(idx, _i), _p(weight, _f), _p(out, _f))
    return out


def three_interpolate_grad(grad_out, idx, weight, M):
    grad_out, idx, weight = _f32(grad_out), _i32(idx), _f32(weight)
    B, C, N = grad_out.shape
    g = np.zeros((B, C, M), dtype=np.float32)
    lib().orc_three_interpolate_grad(B, C, N, M, _p(grad_out, _f), _p(idx, _i), _p(weight, _f), _p(g, _f))
    return g


# ---------------------------------------------------------------- roipool3d
def enlarge_box3d(boxes3d, extra_width):
    """jmodt/utils/kitti_utils.py:152-162"""
    b = np.array(boxes3d, dtype=np.float32, copy=True)
    b[..., 3:6] += np.float32(extra_width * 2)
    b[..., 1] += np.float32(extra_width)
    return b


def roipool3d(pts, pts_feature, boxes3d_enlarged, sampled_pt_num=512, return_idx=False):
    """GPU-layout semantic of roipool3d_cuda.forward on ALREADY ENLARGED boxes."""
    pts, pts_feature, boxes = _f32(pts), _f32(pts_feature), _f32(boxes3d_enlarged)
    B, N, _ = pts.shape
    M = boxes.shape[1]
    C = pts_feature.shape[2]
    S = int(sampled_pt_num)
    pooled = np.zeros((B, M, S, 3 + C), dtype=np.float32)
    empty = np.zeros((B, M), dtype=np.int32)
    pidx = np.zeros((B, M, S), dtype=np.int32)
    lib().orc_roipool3d(B, N, M, C, S, _p(pts, _f), _p(boxes, _f), _p(pts_feature, _f), _p(pooled, _f),
                        _p(empty, _i), _p(pidx, _i))
    return (pooled, empty, pidx) if return_idx else (pooled, empty)


def roipool3d_canonical(pts, pts_feature, boxes3d, extra_width, sampled_pt_num=512):
    """roipool3d_gpu + canonical transformation, as the RCNN stage's inference branch does it
    (proposal_target_layer.py:100-112; rotate_pc_along_y_torch kitti_utils.py:46-64), float32 throughout.
    pts (B,N,3), pts_feature (B,N,C), boxes3d (B,M,7) un-enlarged -> pooled (B,M,S,3+C), empty (B,M)"""
    pts, pts_feature, boxes3d = _f32(pts), _f32(pts_feature), _f32(boxes3d)
    B, M = boxes3d.shape[:2]
    enlarged = np.stack([enlarge_box3d(boxes3d[b], extra_width) for b in range(B)])
    pooled, empty = roipool3d(pts, pts_feature, enlarged, sampled_pt_num)
    pooled = pooled.copy()
    pooled[..., 0:3] -= boxes3d[:, :, None, 0:3]
    cosa, sina = np.cos(boxes3d[..., 6]), np.sin(boxes3d[..., 6])           # float32
    x, z = pooled[..., 0].copy(), pooled[..., 2].copy()
    pooled[..., 0] = x * cosa[..., None] + z * (-sina[..., None])
    pooled[..., 2] = x * sina[..., None] + z * cosa[..., None]
    return pooled, empty


def pts_in_boxes3d(pts, boxes3d):
    pts, boxes = _f32(pts), _f32(boxes3d)
    M, N = boxes.shape[0], pts.shape[0]
    flags = np.zeros((M, N), dtype=np.int64)
    lib().orc_pts_in_boxes3d(M, N, _p(pts, _f), _p(boxes, _f), _p(flags, _l))
    return flags


def roipool3d_cpu_layout(pts, boxes3d, pts_feature, sampled_pt_num):
    pts, boxes, feat = _f32(pts), _f32(boxes3d), _f32(pts_feature)
    N, M, C, S = pts.shape[0], boxes.shape[0], feat.shape[1], int(sampled_pt_num)
    pp = np.zeros((M, S, 3), dtype=np.float32)
    pf = np.zeros((M, S, C), dtype=np.float32)
    ef = np.zeros((M,), dtype=np.int64)
    lib().orc_roipool3d_cpu(N, M, C, S, _p(pts, _f), _p(boxes, _f), _p(feat, _f), _p(pp, _f), _p(pf, _f), _p(ef, _l))
    return pp, pf, ef


# ---------------------------------------------------------------- iou3d
def boxes3d_to_bev(boxes3d):
    """jmodt/utils/kitti_utils.py:136-149"""
    b = _f32(boxes3d)
    out = np.empty((b.shape[0], 5), dtype=np.float32)
    half_l, half_w = b[:, 5] / np.float32(2), b[:, 4] / np.float32(2)
    out[:, 0], out[:, 1] = b[:, 0] - half_l, b[:, 2] - half_w
    out[:, 2], out[:, 3] = b[:, 0] + half_l, b[:, 2] + half_w
    out[:, 4] = b[:, 6]
    return out


def boxes_overlap_bev(boxes_a, boxes_b):
    a, b = _f32(boxes_a), _f32(boxes_b)
    out = np.zeros((a.shape[0], b.shape[0]), dtype=np.float32)
    lib().orc_boxes_overlap_bev(a.shape[0], _p(a, _f), b.shape[0], _p(b, _f), _p(out, _f))
    return out


def boxes_iou_bev(boxes_a, boxes_b):
    a, b = _f32(boxes_a), _f32(boxes_b)
    out = np.zeros((a.shape[0], b.shape[0]), dtype=np.float32)
    lib().orc_boxes_iou_bev(a.shape[0], _p(a, _f), b.shape[0], _p(b, _f), _p(out, _f))
    return out


def boxes_iou3d(boxes_a, boxes_b):
    """jmodt/ops/iou3d/iou3d_utils.py:25-54 with the torch math restated in float32 numpy."""
    a, b = _f32(boxes_a), _f32(boxes_b)
    ov = boxes_overlap_bev(boxes3d_to_bev(a), boxes3d_to_bev(b))
    a_min, a_max = (a[:, 1] - a[:, 3]).reshape(-1, 1), a[:, 1].reshape(-1, 1)
    b_min, b_max = (b[:, 1] - b[:, 3]).reshape(1, -1), b[:, 1].reshape(1, -1)
    oh = np.clip(np.minimum(a_max, b_max) - np.maximum(a_min, b_min), 0, None).astype(np.float32)
    o3 = ov * oh
    va = (a[:, 3] * a[:, 4] * a[:, 5]).reshape(-1, 1)
    vb = (b[:, 3] * b[:, 4] * b[:, 5]).reshape(1, -1)
    return (o3 / np.clip(va + vb - o3, np.float32(1e-7), None)).astype(np.float32)


def boxes_dist(boxes_a, boxes_b):
    """data_association.py:10-28: 1 - centre distance / farthest corner-pair distance, (M, N)"""
    a, b = _f32(boxes_a), _f32(boxes_b)
    out = np.zeros((a.shape[0], b.shape[0]), dtype=np.float32)
    lib().orc_boxes_dist(a.shape[0], _p(a, _f), b.shape[0], _p(b, _f), _p(out, _f))
    return out


def association_cost(pred_boxes, det_boxes, link_scores, w_app, w_iou, w_dis):
    """data_association.py:42-45: link * w_app + iou3d * w_iou + dist * w_dis"""
    return (_f32(link_scores) * np.float32(w_app) + boxes_iou3d(pred_boxes, det_boxes) * np.float32(w_iou)
            + boxes_dist(pred_boxes, det_boxes) * np.float32(w_dis)).astype(np.float32)


def nms_mask(boxes_sorted, thresh, normal):
    b = _f32(boxes_sorted)
    n = b.shape[0]
    mask = np.zeros((n, (n + 63) // 64), dtype=np.uint64)
    lib().orc_nms_mask(n, _p(b, _f), ctypes.c_float(thresh), int(bool(normal)), _p(mask, _u))
    return mask


def nms_sorted(boxes_sorted, thresh, normal):
    """extension-level semantic: boxes already score-sorted; returns kept row indices (int64)."""
    b = _f32(boxes_sorted)
    n = b.shape[0]
    keep = np.zeros((max(n, 1),), dtype=np.int64)
    num = lib().orc_nms(n, _p(b, _f), ctypes.c_float(thresh), int(bool(normal)), _p(keep, _l))
    return keep[:num].copy()


def nms(boxes, scores, thresh, normal):
    """iou3d_utils.nms_gpu / nms_normal_gpu (iou3d_utils.py:57-88) with a STABLE descending sort
    (the reference's torch sort is not declared stable; fixtures avoid score ties)."""
    order = np.argsort(-np.asarray(scores, dtype=np.float32), kind="stable")
    keep = nms_sorted(_f32(boxes)[order], thresh, normal)
    return order[keep].astype(np.int64)


def decode_rpn_proposals(xyz, rpn_reg, loc_scope=3.0, loc_bin_size=0.5, num_head_bin=12,
                         anchor_size=(1.52563191462, 1.62856739989, 3.88311640418), avg_by_bin=True):
    """decode_bbox_target as ProposalLayer calls it + `y += h / 2` (proposal_layer.py:24-34;
    bbox_transform.py:27-260 with roi = xyz (N,3), get_xz_fine=True, get_y_by_bin=False, get_ry_fine=False,
    RY_WITH_BIN=False), float32.  PARITY UNPINNED: restated from reading the reference; the reference
    function itself is not importable here (jmodt.config -> easydict)."""
    xyz = _f32(xyz).reshape(-1, 3)
    reg = _f32(rpn_reg).reshape(xyz.shape[0], -1)
    f = np.float32
    nb = int(loc_scope / loc_bin_size) * 2
    bs, sc = f(loc_bin_size), f(loc_scope)

    def axis(bin_off, res_off):
        if avg_by_bin:                                              # :74-103
            z = reg[:, bin_off:bin_off + nb]
            e = np.exp(z - z.max(1, keepdims=True)).astype(f)
            pbin = e / e.sum(1, keepdims=True, dtype=f)
            centre = (np.arange(nb, dtype=f) * bs + bs / f(2) - sc).astype(f)
            absx = centre[None] + reg[:, res_off:res_off + nb] * bs
            return (absx * pbin).sum(1, dtype=f)
        b = np.argmax(reg[:, bin_off:bin_off + nb], 1)              # :52-72
        res = np.take_along_axis(reg[:, res_off:res_off + nb], b[:, None], 1)[:, 0]
        return (b.astype(f) * bs + bs / f(2) - sc + res * bs).astype(f)

    pos_x = axis(0, 2 * nb) + xyz[:, 0]
    pos_z = axis(nb, 3 * nb) + xyz[:, 2]
    off = 4 * nb
    pos_y = xyz[:, 1] + reg[:, off]
    off += 1
    rb = np.argmax(reg[:, off:off + num_head_bin], 1)
    rres = np.take_along_axis(reg[:, off + num_head_bin:off + 2 * num_head_bin], rb[:, None], 1)[:, 0]
    apc = f(2 * np.pi / num_head_bin)
    ry = np.mod(rb.astype(f) * apc + rres * (apc / f(2)), f(2 * np.pi)).astype(f)   # :137-145
    ry = np.where(ry > f(np.pi), ry - f(2 * np.pi), ry).astype(f)
    off += 2 * num_head_bin
    anchor = np.asarray(anchor_size, dtype=f)
    hwl = reg[:, off:off + 3] * anchor + anchor
    out = np.concatenate([pos_x[:, None], (pos_y + hwl[:, 0] / f(2))[:, None], pos_z[:, None], hwl, ry[:, None]], 1)
    return out.astype(f).reshape(np.shape(rpn_reg)[:-1] + (7,))


def decode_rcnn_boxes(rois, rcnn_reg, loc_scope=1.5, loc_bin_size=0.5, num_head_bin=9,
                      anchor_size=(1.52563191462, 1.62856739989, 3.88311640418), avg_by_bin=True):
    """decode_bbox_target as the detection post-processing calls it (tools/eval.py:108-116; bbox_transform.py:27-260
    with roi (N,7), get_xz_fine=True, get_y_by_bin=False, get_ry_fine=True, RY_WITH_BIN=False), float32; the
    rotation by -roi_ry (bbox_transform.py:8-24,251-256) uses the deterministic sin/cos of include/jm_detmath.h."""
    rois = _f32(rois).reshape(-1, 7)
    reg = _f32(rcnn_reg).reshape(rois.shape[0], -1)
    f = np.float32
    nb = int(loc_scope / loc_bin_size) * 2
    bs, sc = f(loc_bin_size), f(loc_scope)

    def axis(bin_off, res_off):
        if avg_by_bin:
            z = reg[:, bin_off:bin_off + nb]
            e = np.exp(z - z.max(1, keepdims=True)).astype(f)
            pbin = e / e.sum(1, keepdims=True, dtype=f)
            centre = (np.arange(nb, dtype=f) * bs + bs / f(2) - sc).astype(f)
            return ((centre[None] + reg[:, res_off:res_off + nb] * bs) * pbin).sum(1, dtype=f)
        b = np.argmax(reg[:, bin_off:bin_off + nb], 1)
        res = np.take_along_axis(reg[:, res_off:res_off + nb], b[:, None], 1)[:, 0]
        return (b.astype(f) * bs + bs / f(2) - sc + res * bs).astype(f)

    pos_x, pos_z = axis(0, 2 * nb), axis(nb, 3 * nb)
    off = 4 * nb
    pos_y = rois[:, 1] + reg[:, off]
    off += 1
    rb = np.argmax(reg[:, off:off + num_head_bin], 1)
    rres = np.take_along_axis(reg[:, off + num_head_bin:off + 2 * num_head_bin], rb[:, None], 1)[:, 0]
    apc = f(f(np.pi / 2) / f(num_head_bin))
    ry = ((rb.astype(f) * apc + apc / f(2)) + rres * (apc / f(2)) - f(np.pi / 4)).astype(f)   # :131-135
    off += 2 * num_head_bin
    anchor = np.asarray(anchor_size, dtype=f)
    hwl = reg[:, off:off + 3] * anchor + anchor
    sn, cs = detmath_sincos(-rois[:, 6])
    x = pos_x * cs + pos_z * (-sn)
    z = pos_x * sn + pos_z * cs
    out = np.concatenate([(x + rois[:, 0])[:, None], pos_y[:, None], (z + rois[:, 2])[:, None], hwl,
                          (ry + rois[:, 6])[:, None]], 1)
    return out.astype(f)


def select_detections(pred_boxes3d, raw_scores, score_thresh=0.2, nms_thresh=0.1):
    """tools/eval.py:171-193 per frame: sigmoid(score) > thresh, rotated BEV NMS by raw score (stable order);
    returns a list of index arrays (RoI slots kept, in keep order), one per frame."""
    boxes = _f32(pred_boxes3d)
    raw = _f32(raw_scores)
    out = []
    for k in range(boxes.shape[0]):
        norm = (1.0 / (1.0 + np.exp(-raw[k].astype(np.float64)))).astype(np.float32)
        inds = np.nonzero(norm > np.float32(score_thresh))[0]
        if len(inds) == 0:
            out.append(np.zeros(0, np.int64))
            continue
        keep = nms(boxes3d_to_bev(boxes[k][inds]), raw[k][inds], nms_thresh, 0)
        out.append(inds[keep].astype(np.int64))
    return out


def proposal_select(scores, proposals, pre_nms_top_n, post_nms_top_n, nms_thresh, nms_type="normal",
                    distance_based=True):
    """ProposalLayer.forward after the decode, frame by frame (proposal_layer.py:34-55) with
    distance_based_proposal (:57-117) or score_based_proposal (:119-144).  Stable score sorts."""
    scores = _f32(scores)
    proposals = _f32(proposals)
    B = scores.shape[0]
    ret_boxes = np.zeros((B, post_nms_top_n, 7), np.float32)
    ret_scores = np.zeros((B, post_nms_top_n), np.float32)
    normal = {"normal": 1, "rotate": 0}[nms_type]
    for k in range(B):
        order = np.argsort(-scores[k], kind="stable")
        so, po = scores[k][order], proposals[k][order]
        out_s, out_p = [], []
        if distance_based:
            ranges = [0.0, 40.0, 80.0]
            pre = [0, int(pre_nms_top_n * 0.7), pre_nms_top_n - int(pre_nms_top_n * 0.7)]
            post = [0, int(post_nms_top_n * 0.7), post_nms_top_n - int(post_nms_top_n * 0.7)]
            dist = po[:, 2]
            first = (dist > ranges[0]) & (dist <= ranges[1])
            for i in (1, 2):
                m = (dist > ranges[i - 1]) & (dist <= ranges[i])
                if m.sum() != 0:
                    cs, cp = so[m][:pre[i]], po[m][:pre[i]]
                else:
                    if i == 1:
                        continue
                    cs, cp = so[first][pre[i - 1]:][:pre[i]], po[first][pre[i - 1]:][:pre[i]]
                keep = nms(boxes3d_to_bev(cp), cs, nms_thresh, normal)[:post[i]] if len(cs) else np.zeros(0, np.int64)
                out_s.append(cs[keep])
                out_p.append(cp[keep])
        else:
            cs, cp = so[:pre_nms_top_n], po[:pre_nms_top_n]
            keep = nms(boxes3d_to_bev(cp), cs, nms_thresh, 0)[:post_nms_top_n]
            out_s.append(cs[keep])
            out_p.append(cp[keep])
        s1 = np.concatenate(out_s) if out_s else np.zeros(0, np.float32)
        p1 = np.concatenate(out_p) if out_p else np.zeros((0, 7), np.float32)
        ret_boxes[k, :len(s1)] = p1
        ret_scores[k, :len(s1)] = s1
    return ret_boxes, ret_scores


# ---------------------------------------------------------------- LI-Fusion gather
def feature_gather(feature_map, xy):
    """feature_map (B,C,H,W) any strides, xy (B,N,2) in [-1,1] -> (B,C,N)"""
    fm = np.asarray(feature_map, dtype=np.float32)
    xy = _f32(xy)
    B, C, H, W = fm.shape
    N = xy.shape[1]
    out = np.empty((B, C, N), dtype=np.float32)
    sb, sc, sh, sw = (s // 4 for s in fm.strides)
    lib().orc_feature_gather(B, C, H, W, N, _p(fm, _f), ctypes.c_int64(sb), ctypes.c_int64(sc), ctypes.c_int64(sh),
                             ctypes.c_int64(sw), _p(xy, _f), _p(out, _f))
    return out


# ---------------------------------------------------------------- affinity
def _mlp_args(w):
    W1, b1, W2, b2, w3, b3 = w
    W1, b1, W2, b2, w3 = _f32(W1), _f32(b1), _f32(W2), _f32(b2), _f32(np.reshape(w3, -1))
    H1, C = W1.shape
    H2 = W2.shape[0]
    return C, H1, H2, (W1, b1, W2, b2, w3), ctypes.c_float(float(np.reshape(b3, -1)[0]))


def link_scores(pred_feat, det_feat, link_w):
    """raw link scores S (P,D); link_w = (W1(H1,C), b1, W2(H2,H1), b2, w3(H2), b3)"""
    pf, df = _f32(pred_feat), _f32(det_feat)
    C, H1, H2, (W1, b1, W2, b2, w3), b3 = _mlp_args(link_w)
    P, D = pf.shape[0], df.shape[0]
    S = np.empty((P, D), dtype=np.float32)
    lib().orc_link_scores(P, D, C, H1, H2, _p(pf, _f), _p(df, _f), _p(W1, _f), _p(b1, _f), _p(W2, _f), _p(b2, _f),
                          _p(w3, _f), b3, _p(S, _f))
    return S


def dual_softmax(S):
    S = _f32(S)
    A = np.empty_like(S)
    lib().orc_dual_softmax(S.shape[0], S.shape[1], _p(S, _f), _p(A, _f))
    return A


def start_end_logits(pred_feat, det_feat, se_w):
    pf, df = _f32(pred_feat), _f32(det_feat)
    C, H1, H2, (W1, b1, W2, b2, w3), b3 = _mlp_args(se_w)
    P, D = pf.shape[0], df.shape[0]
    start = np.empty((D,), dtype=np.float32)
    end = np.empty((P,), dtype=np.float32)
    lib().orc_start_end_logits(P, D, C, H1, H2, _p(pf, _f), _p(df, _f), _p(W1, _f), _p(b1, _f), _p(W2, _f),
                               _p(b2, _f), _p(w3, _f), b3, _p(start, _f), _p(end, _f))
    return start, end


def affinity(pred_feat, det_feat, link_w, se_w):
    """tracker.py:81-112: returns (A (P,D), start_logit (D), end_logit (P)); the tracker applies
    w_se * sigmoid to the logits, the training path (rcnn.py) uses them raw."""
    A = dual_softmax(link_scores(pred_feat, det_feat, link_w))
    s, e = start_end_logits(pred_feat, det_feat, se_w)
    return A, s, e


# ---------------------------------------------------------------- detmath passthrough
def detmath_sincos(a):
    a = _f32(a).ravel()
    s, c = np.empty_like(a), np.empty_like(a)
    lib().orc_detmath_sincos(a.size, _p(a, _f), _p(s, _f), _p(c, _f))
    return s, c


def detmath_atan2(y, x):
    y, x = _f32(y).ravel(), _f32(x).ravel()
    r = np.empty_like(y)
    lib().orc_detmath_atan2(y.size, _p(y, _f), _p(x, _f), _p(r, _f))
    return r
