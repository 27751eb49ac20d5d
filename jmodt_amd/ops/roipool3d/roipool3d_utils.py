"""RoI point pooling on the gfx950 kernel.

Mirror of jmodt/ops/roipool3d/roipool3d_utils.py: roipool3d_gpu (:8-29), pts_in_boxes3d_cpu
(:32-51), roipool_pc_cpu (:54-72), roipool3d_cpu (:75-109).
"""
import numpy as np
import torch

from ...ext import roipool3d_cuda


def enlarge_box3d(boxes3d, extra_width):
    """(N, 7) [x, y, z, h, w, l, ry]: h, w, l += 2*extra, y += extra  (jmodt/utils/kitti_utils.py:152-162)"""
    large = boxes3d.copy() if isinstance(boxes3d, np.ndarray) else boxes3d.clone()
    large[:, 3:6] += extra_width * 2
    large[:, 1] += extra_width
    return large


def rotate_pc_along_y(pc, rot_angle):
    """(N, 3+) rotate x,z by rot_angle about the y axis, in place  (jmodt/utils/kitti_utils.py:31-43)"""
    cosval, sinval = np.cos(rot_angle), np.sin(rot_angle)
    rotmat = np.array([[cosval, -sinval], [sinval, cosval]])
    pc[:, [0, 2]] = np.dot(pc[:, [0, 2]], np.transpose(rotmat))
    return pc


def roipool3d_gpu(pts, pts_feature, boxes3d, pool_extra_width, sampled_pt_num=512):
    """pts (B, N, 3), pts_feature (B, N, C), boxes3d (B, M, 7) ->
    pooled_features (B, M, sampled_pt_num, 3 + C), pooled_empty_flag (B, M) int32"""
    batch_size, boxes_num, feature_len = pts.shape[0], boxes3d.shape[1], pts_feature.shape[2]
    pooled_boxes3d = enlarge_box3d(boxes3d.view(-1, 7), pool_extra_width).view(batch_size, -1, 7)
    # uninitialised outputs: the kernel writes every row, zeros for empty boxes included
    pooled_features = torch.empty((batch_size, boxes_num, sampled_pt_num, 3 + feature_len), dtype=torch.float32,
                                  device=pts.device)
    pooled_empty_flag = torch.empty((batch_size, boxes_num), dtype=torch.int32, device=pts.device)
    roipool3d_cuda.forward(pts.contiguous(), pooled_boxes3d.contiguous(), pts_feature.contiguous(), pooled_features,
                           pooled_empty_flag, zero_empty=1)
    return pooled_features, pooled_empty_flag


def roipool3d_canonical_gpu(pts, pts_feature, boxes3d, pool_extra_width, sampled_pt_num=512, return_count=False):
    """roipool3d_gpu followed by the canonical transformation the RCNN stage always applies
    (jmodt/detection/layers/proposal_target_layer.py:100-112): pooled xyz minus the RoI centre, then rotated by
    the RoI heading — in ONE kernel instead of the pooling + two full passes over the (B, M, S, 3 + C) tensor
    and a per-frame Python loop.  Returns (pooled (B, M, S, 3 + C), pooled_empty_flag (B, M) int32) [+ pooled_count (B, M)
    int32 with return_count: the number of distinct source points per slab — rows count .. S-1 are cyclic copies]."""
    batch_size, boxes_num, feature_len = pts.shape[0], boxes3d.shape[1], pts_feature.shape[2]
    pooled = torch.empty((batch_size, boxes_num, sampled_pt_num, 3 + feature_len), dtype=torch.float32,
                         device=pts.device)
    empty = torch.empty((batch_size, boxes_num), dtype=torch.int32, device=pts.device)
    if return_count:
        count = torch.empty((batch_size, boxes_num), dtype=torch.int32, device=pts.device)
        roipool3d_cuda.forward_canonical(pts.contiguous(), boxes3d.contiguous().float(), pool_extra_width,
                                         pts_feature.contiguous(), pooled, empty, count)
        return pooled, empty, count
    roipool3d_cuda.forward_canonical(pts.contiguous(), boxes3d.contiguous().float(), pool_extra_width,
                                     pts_feature.contiguous(), pooled, empty)
    return pooled, empty


def _host_f32(t):
    """a contiguous float32 CPU tensor (the CPU entry points take host pointers)"""
    return t.detach().to(device="cpu", dtype=torch.float32).contiguous()


def pts_in_boxes3d_cpu(pts, boxes3d):
    """pts (N, 3), boxes3d (M, 7) on the CPU -> list of M boolean masks (N)"""
    if pts.is_cuda:
        raise NotImplementedError   # as the reference: this entry point is CPU only
    p, b = _host_f32(pts), _host_f32(boxes3d)
    flags = torch.zeros((b.shape[0], p.shape[0]), dtype=torch.int64)
    roipool3d_cuda.pts_in_boxes3d_cpu(flags, p, b)
    return list((flags > 0).unbind(0))


def roipool_pc_cpu(pts, pts_feature, boxes3d, sampled_pt_num):
    """pts (N, 3), pts_feature (N, C), boxes3d (M, 7) -> pooled_pts (M, S, 3), pooled_features (M, S, C),
    pooled_empty_flag (M) int64"""
    p, f, b = _host_f32(pts), _host_f32(pts_feature), _host_f32(boxes3d)
    assert p.shape[0] == f.shape[0] and p.shape[1] == 3, "%s %s" % (p.shape, f.shape)
    m = b.shape[0]
    out_xyz = torch.zeros((m, sampled_pt_num, 3), dtype=torch.float32)
    out_feat = torch.zeros((m, sampled_pt_num, f.shape[1]), dtype=torch.float32)
    empty = torch.zeros(m, dtype=torch.int64)
    roipool3d_cuda.roipool3d_cpu(p, b, f, out_xyz, out_feat, empty)
    return out_xyz, out_feat, empty


def roipool3d_cpu(boxes3d, pts, pts_feature, pts_extra_input, pool_extra_width, sampled_pt_num=512,
                  canonical_transform=True):
    """numpy front end: boxes3d (M, 7), pts (N, 3), pts_feature (N, C), pts_extra_input (N, C2) ->
    (sampled_pts_input (M, S, 3 + C2), sampled_pts_feature (M, S, C)[, pooled_empty_flag when not canonical])"""
    n_extra = pts_extra_input.shape[1]
    stacked = np.concatenate((pts_extra_input, pts_feature), axis=1)        # extra channels first, as the reference
    xyz, feat, empty = roipool_pc_cpu(torch.from_numpy(pts), torch.from_numpy(stacked),
                                      torch.from_numpy(enlarge_box3d(boxes3d, pool_extra_width)), sampled_pt_num)
    pts_input = np.concatenate((xyz.numpy(), feat[:, :, :n_extra].numpy()), axis=2)
    pts_feat = feat[:, :, n_extra:].numpy()
    if not canonical_transform:
        return pts_input, pts_feat, empty.numpy()
    # canonical frame of each RoI: origin at the box (x, y_bottom, z), heading along +x
    pts_input[:, :, 0:3] -= boxes3d[:, np.newaxis, 0:3]
    headings = boxes3d[:, 6] % (2 * np.pi)
    for box_i, angle in enumerate(headings):
        rotate_pc_along_y(pts_input[box_i], angle)                          # in place
    return pts_input, pts_feat
