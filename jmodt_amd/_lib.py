"""ctypes binding of libjmodt_hip.so (the C ABI declared in include/jmodt_hip.h).

The HIP library is the ONLY compute path of this package: there is no CPU / eager-PyTorch
fallback.  Importing this module never needs a GPU (the driver's build check and the CPU test
tier load the library and check its symbols), but calling any op without the library or with
non-GPU tensors raises immediately.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libjmodt_hip.so")

_P = ctypes.c_void_p
_I = ctypes.c_int
_F = ctypes.c_float
_L = ctypes.c_int64
_Z = ctypes.c_size_t


class Mlp3(ctypes.Structure):
    """jm_mlp3_t"""
    _fields_ = [("c", _I), ("h1", _I), ("h2", _I), ("w1", _P), ("b1", _P), ("w2", _P), ("b2", _P), ("w3", _P),
                ("b3", _P)]


class Mlp3Grad(ctypes.Structure):
    """jm_mlp3_grad_t"""
    _fields_ = [("dw1", _P), ("db1", _P), ("dw2", _P), ("db2", _P), ("dw3", _P), ("db3", _P)]


ROWS_MAX_LAYERS = 6


class RowsMlp(ctypes.Structure):
    """jm_rows_mlp_t"""
    _fields_ = [("nl", _I), ("m", _I), ("m_dev", _P), ("k1", _I), ("k2", _I), ("x1", _P), ("ldx1", _I), ("x2", _P), ("ldx2", _I),
                ("widths", _I * ROWS_MAX_LAYERS), ("acts", _I * ROWS_MAX_LAYERS), ("w", _P * ROWS_MAX_LAYERS), ("ldw", _I * ROWS_MAX_LAYERS),
                ("b", _P * ROWS_MAX_LAYERS), ("y", _P * ROWS_MAX_LAYERS)]


class RowsMlpGrad(ctypes.Structure):
    """jm_rows_mlp_grad_t"""
    _fields_ = [("dout", _P), ("lddout", _I), ("dw", _P * ROWS_MAX_LAYERS), ("lddw", _I * ROWS_MAX_LAYERS), ("db", _P * ROWS_MAX_LAYERS),
                ("dx1", _P), ("dx2", _P), ("scratch", _P * 2), ("ws", _P), ("ws_bytes", _Z)]


class SaScale(ctypes.Structure):
    """jm_sa_scale_t"""
    _fields_ = [("nl", _I), ("groups", _I), ("max_rows", _I), ("rows_dev", _P), ("offsets", _P), ("row_point", _P), ("row_group", _P),
                ("points", _I), ("c", _I), ("f", _P), ("ldf", _I), ("xyz", _P), ("ctr", _P), ("widths", _I * ROWS_MAX_LAYERS),
                ("w1x", _P), ("w1f", _P), ("b1", _P), ("w", _P * ROWS_MAX_LAYERS), ("b", _P * ROWS_MAX_LAYERS),
                ("u", _P), ("delta", _P), ("h", _P * ROWS_MAX_LAYERS), ("out", _P), ("ldo", _I), ("argrow", _P)]


class SaScaleGrad(ctypes.Structure):
    """jm_sa_scale_grad_t"""
    _fields_ = [("dout", _P), ("lddout", _I), ("dw1", _P), ("db1", _P), ("dw4", _P), ("dw", _P * ROWS_MAX_LAYERS), ("db", _P * ROWS_MAX_LAYERS),
                ("du", _P), ("df", _P), ("df_accumulate", _I), ("scratch", _P * 2), ("ws", _P), ("ws_bytes", _Z)]


# name -> (restype, argtypes); mirrors include/jmodt_hip.h one to one
SIGNATURES = {
    "jm_version": (_I, []),
    "jm_last_error": (ctypes.c_char_p, []),
    "jm_furthest_point_sampling": (_I, [_I, _I, _I, _P, _P, _P, _P]),
    "jm_fps_workspace_bytes": (_Z, [_I, _I]),
    "jm_furthest_point_sampling_xyz": (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _Z, _P]),
    "jm_furthest_point_sampling_ws": (_I, [_I, _I, _I, _P, _P, _P, _P, _Z, _P]),
    "jm_gather_points": (_I, [_I, _I, _I, _I, _P, _P, _P, _P]),
    "jm_gather_points_grad": (_I, [_I, _I, _I, _I, _P, _P, _P, _P]),
    "jm_ball_query": (_I, [_I, _I, _I, _F, _I, _P, _P, _P, _P]),
    "jm_ball_query_dual": (_I, [_I, _I, _I, _F, _I, _F, _I, _P, _P, _P, _P, _P]),
    "jm_ball_query_workspace_bytes": (_Z, [_I, _I]),
    "jm_ball_query_evals_offset": (_Z, [_I, _I]),
    "jm_ball_query_ws": (_I, [_I, _I, _I, _F, _I, _P, _P, _P, _P, _Z, _P]),
    "jm_ball_query_dual_ws": (_I, [_I, _I, _I, _F, _I, _F, _I, _P, _P, _P, _P, _P, _Z, _P]),
    "jm_ball_query_grid_build": (_I, [_I, _I, _F, _P, _P, _Z, _P]),
    "jm_ball_query_grid_query": (_I, [_I, _I, _I, _F, _F, _I, _F, _I, _P, _P, _P, _P, _Z, _P]),
    "jm_group_points": (_I, [_I, _I, _I, _I, _I, _P, _P, _P, _P]),
    "jm_group_points_grad": (_I, [_I, _I, _I, _I, _I, _P, _P, _P, _P]),
    "jm_three_nn": (_I, [_I, _I, _I, _P, _P, _P, _P, _P]),
    "jm_three_nn_workspace_bytes": (_Z, [_I, _I, _I]),
    "jm_three_nn_grid_workspace_bytes": (_Z, [_I, _I, _I]),
    "jm_three_nn_ws": (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _Z, _P]),
    "jm_three_interpolate": (_I, [_I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "jm_three_interpolate_grad": (_I, [_I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "jm_sa_mlp_supported": (_I, [_I, _I, _I, _I, _I, _I, _I, ctypes.POINTER(_I)]),
    "jm_sa_mlp_packed_weight_elems": (_Z, [_I, _I, _I]),
    "jm_sa_mlp_packed_bias_elems": (_Z, [_I]),
    "jm_sa_mlp_pack": (_I, [_I, _I, _I, _P, _P, _P, _P, _P]),
    "jm_sa_mlp_forward": (_I, [_I, _I, _I, _I, _I, _P, _P, _P, _P, _I, ctypes.POINTER(_I), ctypes.POINTER(_P),
                               ctypes.POINTER(_P), _P, _P]),
    "jm_sa_mlp_forward_pre": (_I, [_I, _I, _I, _I, _I, _P, _P, _P, _P, _I, ctypes.POINTER(_I), ctypes.POINTER(_P),
                                   ctypes.POINTER(_P), _P, _P]),
    "jm_sa_mlp_forward_into": (_I, [_I, _I, _I, _I, _I, _P, _P, _P, _P, _I, ctypes.POINTER(_I), ctypes.POINTER(_P),
                                    ctypes.POINTER(_P), _P, _Z, _P]),
    "jm_sa_mlp_forward_pre_into": (_I, [_I, _I, _I, _I, _I, _P, _P, _P, _P, _I, ctypes.POINTER(_I), ctypes.POINTER(_P),
                                        ctypes.POINTER(_P), _P, _Z, _P]),
    "jm_sa_group_list_elems": (_Z, [_I, _I]),
    "jm_sa_group_plan": (_I, [_I, _I, _P, _I, _P, _P, _P]),
    "jm_sa_group_plan_dev": (_I, [_I, _I, _P, _I, _P, _P, _P, _P]),
    "jm_sa_group_plan_dual": (_I, [_I, _I, _P, _I, _I, _P, _I, _P, _P, _P, _P]),
    "jm_sa_mlp_listed_supported": (_I, [_I, _I, _I, _I, _I, _I, ctypes.POINTER(_I)]),
    "jm_sa_mlp_listed_qmin": (_I, [_I]),
    "jm_sa_mlp_forward_listed": (_I, [_I, _I, _I, _I, _I, _P, _P, _P, _P, _I, ctypes.POINTER(_I), ctypes.POINTER(_P),
                                      ctypes.POINTER(_P), _P, _P, _P, _Z, _P]),
    "jm_sa_mlp_pm_listed_qmin": (_I, [_I, _I, _I]),
    "jm_sa_mlp_pm_listed_supported": (_I, [_I, _I, _I, _I, _I, _I, _I]),
    "jm_sa_mlp_pm_forward_listed": (_I, [_I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "jm_sa_mlp_pm_forward_into": (_I, [_I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "jm_roipool3d_forward": (_I, [_I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _I, _P]),
    "jm_roipool3d_canonical": (_I, [_I, _I, _I, _I, _I, _P, _P, _F, _P, _P, _P, _P]),
    "jm_pts_in_boxes3d_cpu": (_I, [_I, _I, _P, _P, _P]),
    "jm_roipool3d_cpu": (_I, [_I, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "jm_boxes_overlap_bev": (_I, [_I, _P, _I, _P, _P, _P]),
    "jm_boxes_iou_bev": (_I, [_I, _P, _I, _P, _P, _P]),
    "jm_association_cost": (_I, [_I, _P, _I, _P, _P, _F, _F, _F, _P, _P, _P, _P]),
    "jm_nms_workspace_bytes": (_Z, [_I]),
    "jm_nms": (_I, [_I, _P, _F, _I, _P, _P, _P, _Z, _P]),
    "jm_nms_batched": (_I, [_I, _I, _P, _P, _F, _I, _P, _P, _P, _Z, _P]),
    "jm_nms_normal_first_k_batched": (_I, [_I, _I, _P, _P, _F, _I, _P, _P, _P]),
    "jm_proposal_select_workspace_bytes": (_Z, [_I, _I, _I]),
    "jm_proposal_select_evals_offset": (_Z, [_I, _I, _I]),
    "jm_proposal_select": (_I, [_I, _I, _P, _P, _P, _I, _I, _I, _F, _I, _P, _P, _P, _Z, _P]),
    "jm_decode_rpn_proposals": (_I, [ctypes.c_longlong, _I, _P, _P, _F, _F, _I, ctypes.POINTER(_F), _I, _P, _P]),
    "jm_conv1d_stack64_supported": (_I, [_I, _I, _I, _I, _I, _I, _P]),
    "jm_conv1d_stack64_packed_elems": (_Z, [_I, _I]),
    "jm_conv1d_stack64_pack": (_I, [_I, _I, _P, _I, _P, _P, _P, _P]),
    "jm_conv1d_stack64_forward": (_I, [_I, _I, _I, _P, _I, _P, _I, _I, _P, _P, _P, _P, _I, _P, _P]),
    "jm_points_linear_supported": (_I, [_I, _I, _I, _I, _I]),
    "jm_points_linear": (_I, [_I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _P, _P]),
    "jm_three_nn_weights": (_I, [ctypes.c_longlong, _P, _P, _P]),
    "jm_gather_point_rows": (_I, [_I, _I, _I, _I, _P, _P, _P, _P]),
    "jm_pts_feature": (_I, [_I, _I, _I, _P, ctypes.c_longlong, _I, _P, _P, _F, _P, _P]),
    "jm_decode_rpn_proposals_strided": (_I, [_I, _I, _I, _P, _P, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_longlong, _F, _F, _I,
                                            ctypes.POINTER(_F), _I, _P, _P]),
    "jm_detections_sort": (_I, [_I, _I, _P, _P, _F, _P, _P, _P, _P]),
    "jm_detections_gather": (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "jm_decode_rcnn_boxes": (_I, [ctypes.c_longlong, _I, _P, _P, _F, _F, _I, ctypes.POINTER(_F), _I, _P, _P]),
    "jm_boxes_iou3d_batched": (_I, [_I, _I, _P, _I, _P, _P, _P, _P]),
    "jm_nms_mask": (_I, [_I, _P, _F, _I, _P, _P]),
    "jm_feature_gather": (_I, [_I, _I, _I, _I, _I, _P, _L, _L, _L, _L, _P, _P, _P]),
    "jm_feature_gather_grad": (_I, [_I, _I, _I, _I, _I, _P, _P, _P, _L, _L, _L, _L, _P]),
    "jm_rcnn_lift_supported": (_I, [_I] * 7),
    "jm_rcnn_lift_forward": (_I, [_I] * 8 + [_P] * 11 + [_I] + [_P] * 2),
    "jm_rcnn_lift_forward_cnt": (_I, [_I] * 8 + [_P] * 11 + [_I] + [_P] * 4),
    "jm_bias_relu_channels_last": (_I, [ctypes.c_longlong, _I, _P, _P, _P]),
    "jm_image_fusion_gather_workspace_bytes": (_Z, [_I, _I]),
    "jm_image_fusion_packed_elems": (_Z, [_I, _I]),
    "jm_image_fusion_pack": (_I, [_I, _I, _I, _P, _P, _P]),
    "jm_image_fusion_gather": (_I, [_I, _I, _I, _I, _I, _I, ctypes.POINTER(_I), ctypes.POINTER(_I), ctypes.POINTER(_P),
                                    ctypes.POINTER(_P), _P, _P, _P, _P, _Z, _P]),
    "jm_attention_fusion_supported": (_I, [_I, _I, _I, _I, _I, _I]),
    "jm_attention_fusion_forward": (_I, [_I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _P, _P]),
    "jm_affinity_workspace_bytes": (_Z, [_I, _I, ctypes.POINTER(Mlp3), ctypes.POINTER(Mlp3)]),
    "jm_affinity_forward": (_I, [_I, _I, _P, _P, ctypes.POINTER(Mlp3), ctypes.POINTER(Mlp3), _P, _P, _P, _P, _P, _Z,
                                 _P]),
    "jm_affinity_start_end_workspace_bytes": (_Z, [_I, _I, ctypes.POINTER(Mlp3)]),
    "jm_affinity_start_end": (_I, [_I, _I, _P, _P, ctypes.POINTER(Mlp3), _P, _P, _P, _Z, _P]),
    "jm_affinity_batched_workspace_bytes": (_Z, [_I, _I, _I, ctypes.POINTER(Mlp3)]),
    "jm_affinity_forward_batched": (_I, [_I, _I, _I, _P, _P, ctypes.POINTER(Mlp3), _P, _P, _P, _Z, _P]),
    "jm_affinity_start_end_batched_workspace_bytes": (_Z, [_I, _I, _I, ctypes.POINTER(Mlp3)]),
    "jm_affinity_start_end_batched": (_I, [_I, _I, _I, _P, _P, ctypes.POINTER(Mlp3), _P, _P, _Z, _P]),
    "jm_conv1d_stack_supported": (_I, [_I, _I, _I, _I, _I, _I, _P]),
    "jm_conv1d_stack_forward": (_I, [_I, _I, _I, _P, _I, _P, _I, _I, _P, _P, _P, _P, _P, _P, _I, _P, _P]),
    "jm_conv3x3_rgb_bias_relu": (_I, [_I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "jm_argsort_desc_supported": (_I, [_I]),
    "jm_argsort_desc_stable": (_I, [_I, _I, _P, _P, _P]),
    "jm_conv3x3_wino_packed_elems": (_Z, [_I, _I]),
    "jm_conv3x3_wino_supported": (_I, [_I, _I]),
    "jm_conv3x3_wino_pack": (_I, [_I, _I, _P, _P, _P]),
    "jm_conv3x3_wino_bias_relu": (_I, [_I, _I, _I, _I, _I, _P, _P, _P, _I, _P, _P]),
    "jm_sa_mlp_pm_supported": (_I, [_I, _I, _I, _I, _I, _I, _I]),
    "jm_sa_mlp_pm_forward": (_I, [_I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "jm_affinity_dual_softmax_batched": (_I, [_I, _I, _I, _P, _P, _P, _P]),
    "jm_affinity_x3_workspace_bytes": (_Z, [_I, _I, _I, ctypes.POINTER(Mlp3)]),
    "jm_affinity_link_scores_x3": (_I, [_I, _I, _I, _P, _P, ctypes.POINTER(Mlp3), _P, _P, _Z, _P]),
    "jm_linear_rows": (_I, [_I, _I, _I, _P, _P, _P, _P, _I, _P]),
    "jm_mlp3_workspace_bytes": (_Z, [_I, ctypes.POINTER(Mlp3)]),
    "jm_mlp3_forward": (_I, [_I, _P, ctypes.POINTER(Mlp3), _P, _P, _Z, _P]),
    "jm_roipool3d_canonical_cnt": (_I, [_I, _I, _I, _I, _I, _P, _P, _F, _P, _P, _P, _P, _P]),
    "jm_sa_mlp_pm_forward_dyn": (_I, [_I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "jm_sa_dedupe_canon_from_cnt": (_I, [_I, _I, _P, _P, _P]),
    "jm_sa_dedupe_capacity": (_L, [_I, _I, _I]),
    "jm_sa_dedupe_plan": (_I, [_I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "jm_sa_dedupe_combine": (_I, [_I, _I, _I, _L, _P, _P, _P, _P, _P, _P]),
    "jm_affinity_train_prepare": (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "jm_affinity_train_loss_value": (_I, [_I, _P, _P, _P, _F, _F, _P, _P]),
    "jm_affinity_train_link_workspace_bytes": (_Z, [_I, _I, ctypes.POINTER(Mlp3)]),
    "jm_affinity_train_link_step": (_I, [_I, _I, _P, _P, _P, _P, _P, _P, _F, ctypes.POINTER(Mlp3), _P, _P, _P,
                                         ctypes.POINTER(Mlp3Grad), _P, _P, _Z, _P]),
    "jm_affinity_train_se_workspace_bytes": (_Z, [_I, _I, ctypes.POINTER(Mlp3)]),
    "jm_affinity_train_se_step": (_I, [_I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _F, ctypes.POINTER(Mlp3), _P, _P,
                                       ctypes.POINTER(Mlp3Grad), _P, _P, _Z, _P]),
    "jm_affinity_train_feature_grad": (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "jm_rows_linear_forward": (_I, [_I, _P, _I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _P, _I, _P]),
    "jm_rows_linear_dgrad": (_I, [_I, _P, _I, _I, _P, _I, _P, _I, _P, _I, _I, _P, _I, _P]),
    "jm_rows_wgrad_splits": (_I, [_I, _I, _I]),
    "jm_rows_wgrad_workspace_bytes": (_Z, [_I, _I, _I]),
    "jm_rows_linear_wgrad": (_I, [_I, _P, _I, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _Z, _P]),
    "jm_rows_reduce_workspace_bytes": (_Z, [_I]),
    "jm_rows_colsum": (_I, [_I, _P, _I, _P, _I, _P, _I, _P, _Z, _P]),
    "jm_rows_relu_mask": (_I, [_I, _P, _I, _P, _I, _P, _I, _P, _I, _P]),
    "jm_sa_rows_plan": (_I, [_I, _I, _P, _P, _I, _I, _P, _P, _P, _P, _P]),
    "jm_sa_rows_h1": (_I, [_I, _P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P]),
    "jm_sa_rows_pool": (_I, [_I, _I, _P, _I, _P, _P, _I, _P, _P]),
    "jm_sa_rows_pool_grad": (_I, [_I, _P, _I, _P, _I, _P, _I, _P, _P, _P, _I, _P]),
    "jm_sa_rows_scatter_add": (_I, [_I, _P, _I, _P, _I, _P, _P, _I, _P]),
    "jm_sa_rows_xyz_wgrad": (_I, [_I, _P, _I, _P, _I, _P, _P, _P, _P, _P, _I, _P, _Z, _P]),
    "jm_three_interpolate_rows": (_I, [_I, _I, _I, _I, _P, _I, _P, _P, _P, _I, _P]),
    "jm_three_interpolate_rows_grad": (_I, [_I, _I, _I, _I, _P, _I, _P, _P, _P, _I, _P]),
    "jm_feature_gather_rows": (_I, [_I, _I, _I, _I, _I, _P, _P, _P, _I, _P]),
    "jm_feature_gather_rows_grad": (_I, [_I, _I, _I, _I, _I, _P, _I, _P, _P, _P]),
    "jm_rows_mlp_forward": (_I, [ctypes.POINTER(RowsMlp), _P]),
    "jm_rows_mlp_backward": (_I, [ctypes.POINTER(RowsMlp), ctypes.POINTER(RowsMlpGrad), _P]),
    "jm_rows_tanh_grad": (_I, [_I, _P, _I, _P, _I, _P, _I, _P, _I, _P]),
    "jm_sa_scale_forward": (_I, [ctypes.POINTER(SaScale), _P]),
    "jm_sa_scale_backward": (_I, [ctypes.POINTER(SaScale), ctypes.POINTER(SaScaleGrad), _P]),
    "jm_fold_bn_multi": (_I, [_I, _P, _P, _P, _P, _P, _P, _P]),
    "jm_fold_bn_multi_grad": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "jm_rows_sigmoid": (_I, [_I, _P, _I, _P, _P]),
    "jm_rows_gate_backward": (_I, [_I, _I, _I, _P, _I, _P, _I, _P, _P, _I, _P, _P, _P, _I, _P]),
}

_lib = None


def load():
    """dlopen the library and bind every symbol of the ABI; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise RuntimeError(
            f"jmodt_amd: {LIB_PATH} is missing. Build it with `python -m jmodt_amd.csrc.build` "
            "(hipcc, gfx950). There is no CPU fallback for the jmodt ops.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    from .profile import LibProxy
    _lib = LibProxy(lib)     # plain forwarding unless jmodt_amd.profile.prof is enabled (bench.py)
    return _lib


SYNC_DEBUG = False       # debugging aid: device synchronisation behind every library call, so that an asynchronous device fault
                         # aborts inside the call that caused it (python -X faulthandler then names it); never on in product runs


def check(rc, what):
    if rc != 0:
        msg = load().jm_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"jmodt_amd.{what} failed (code {rc}): {msg}")
    if SYNC_DEBUG:
        torch.cuda.synchronize()


# the raw queries: torch.cuda.current_stream() builds a Stream object and re-checks the lazy initialisation, ~8 us x 80 launches
# per step.  They are private symbols that CPU-only torch builds do not define (and no stable API): resolved with getattr, the
# public calls stand in where they are missing
_current_device = getattr(torch._C, "_cuda_getDevice", None) or (lambda: torch.cuda.current_device())
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_current_raw_stream = _raw_stream or (lambda device: torch.cuda.current_stream(device).cuda_stream)


def stream_ptr():
    """the hipStream_t torch is currently launching on (never the legacy default stream implicitly)"""
    return ctypes.c_void_p(_current_raw_stream(_current_device()))


def dev(t: torch.Tensor, dtype, name: str):
    """validate a device tensor argument and return its raw pointer"""
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a GPU tensor (jmodt_amd has no CPU path); got device {t.device}")
    if t.device.index != _current_device():
        # the launch goes to the CURRENT device's current stream (stream_ptr); a tensor living elsewhere would be
        # dereferenced on the wrong GPU.  Callers switch with `torch.cuda.device(t.device)` (one process per GPU
        # never needs to).
        raise RuntimeError(f"{name} lives on {t.device} but the current device is cuda:{torch.cuda.current_device()}; "
                           "wrap the call in `with torch.cuda.device(tensor.device):`")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
    return ctypes.c_void_p(t.data_ptr())


def host(t: torch.Tensor, dtype, name: str):
    if t.is_cuda:
        raise RuntimeError(f"{name} must be a CPU tensor")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
    return ctypes.c_void_p(t.data_ptr())
