"""`roipool3d_cuda` extension-module shim (roipool3d.cpp:198-203) over the C ABI."""
import torch

from .. import _lib as L

f32, i32, i64 = torch.float32, torch.int32, torch.int64


def forward(xyz, boxes3d, pts_feature, pooled_features, pooled_empty_flag, zero_empty=0):
    lib = L.load()
    B, N = xyz.size(0), xyz.size(1)
    M, C, S = boxes3d.size(1), pts_feature.size(2), pooled_features.size(2)
    L.check(lib.jm_roipool3d_forward(B, N, M, C, S, L.dev(xyz, f32, "xyz"), L.dev(boxes3d, f32, "boxes3d"),
                                     L.dev(pts_feature, f32, "pts_feature"),
                                     L.dev(pooled_features, f32, "pooled_features"),
                                     L.dev(pooled_empty_flag, i32, "pooled_empty_flag"), int(zero_empty),
                                     L.stream_ptr()), "roipool3d.forward")
    return 1


def forward_canonical(xyz, rois, extra_width, pts_feature, pooled_features, pooled_empty_flag, pooled_count=None):
    """MI355X-native: pooling + the canonical transformation of proposal_target_layer.py:106-112 in one pass;
    `rois` are the un-enlarged boxes; pooled_count (B, M) int32 (optional) receives the distinct points per slab"""
    lib = L.load()
    B, N = xyz.size(0), xyz.size(1)
    M, C, S = rois.size(1), pts_feature.size(2), pooled_features.size(2)
    if pooled_count is not None:
        L.check(lib.jm_roipool3d_canonical_cnt(B, N, M, C, S, L.dev(xyz, f32, "xyz"), L.dev(rois, f32, "rois"),
                                               float(extra_width), L.dev(pts_feature, f32, "pts_feature"),
                                               L.dev(pooled_features, f32, "pooled_features"),
                                               L.dev(pooled_empty_flag, i32, "pooled_empty_flag"),
                                               L.dev(pooled_count, i32, "pooled_count"), L.stream_ptr()),
                "roipool3d.forward_canonical")
        return 1
    L.check(lib.jm_roipool3d_canonical(B, N, M, C, S, L.dev(xyz, f32, "xyz"), L.dev(rois, f32, "rois"),
                                       float(extra_width), L.dev(pts_feature, f32, "pts_feature"),
                                       L.dev(pooled_features, f32, "pooled_features"),
                                       L.dev(pooled_empty_flag, i32, "pooled_empty_flag"), L.stream_ptr()),
            "roipool3d.forward_canonical")
    return 1


# the reference exposes a second, slower kernel under this name that computes the same result
# (roipool3d_kernel.cu:31-94); it is never called from Python.  One implementation serves both.
forward_slow = forward


def pts_in_boxes3d_cpu(pts_flag, pts, boxes3d):
    lib = L.load()
    L.check(lib.jm_pts_in_boxes3d_cpu(boxes3d.size(0), pts.size(0), L.host(pts, f32, "pts"),
                                      L.host(boxes3d, f32, "boxes3d"), L.host(pts_flag, i64, "pts_flag")),
            "pts_in_boxes3d_cpu")
    return 1


def roipool3d_cpu(pts, boxes3d, pts_feature, pooled_pts, pooled_features, pooled_empty_flag):
    lib = L.load()
    L.check(lib.jm_roipool3d_cpu(pts.size(0), boxes3d.size(0), pts_feature.size(1), pooled_pts.size(1),
                                 L.host(pts, f32, "pts"), L.host(boxes3d, f32, "boxes3d"),
                                 L.host(pts_feature, f32, "pts_feature"), L.host(pooled_pts, f32, "pooled_pts"),
                                 L.host(pooled_features, f32, "pooled_features"),
                                 L.host(pooled_empty_flag, i64, "pooled_empty_flag")), "roipool3d_cpu")
    return 1
