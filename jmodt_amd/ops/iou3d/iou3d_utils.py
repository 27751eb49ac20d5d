"""BEV / 3D IoU and NMS on the gfx950 kernels.

Mirror of jmodt/ops/iou3d/iou3d_utils.py: boxes_iou_bev (:7-22), boxes_iou3d_gpu (:25-54),
nms_gpu (:57-71), nms_normal_gpu (:74-88) — same arguments, same returns (kept indices are an
int64 tensor on the boxes' device, in descending-score order).  The score sort is STABLE here
(the reference's `scores.sort` leaves the order of equal scores unspecified).
"""
import torch

from ...ext import iou3d_cuda


def boxes3d_to_bev_torch(boxes3d: torch.Tensor) -> torch.Tensor:
    """(N, 7) [x, y, z, h, w, l, ry] -> (N, 5) [x1, y1, x2, y2, ry]  (jmodt/utils/kitti_utils.py:136-149)"""
    cu, cv = boxes3d[:, 0], boxes3d[:, 2]
    half_l, half_w = boxes3d[:, 5] / 2, boxes3d[:, 4] / 2
    return torch.stack((cu - half_l, cv - half_w, cu + half_l, cv + half_w, boxes3d[:, 6]), dim=1)


def boxes_iou_bev(boxes_a: torch.Tensor, boxes_b: torch.Tensor) -> torch.Tensor:
    """boxes_a (M, 5), boxes_b (N, 5) -> (M, N) rotated BEV IoU"""
    ans_iou = torch.empty((boxes_a.shape[0], boxes_b.shape[0]), dtype=torch.float32, device=boxes_a.device)
    iou3d_cuda.boxes_iou_bev_gpu(boxes_a.contiguous(), boxes_b.contiguous(), ans_iou)
    return ans_iou


def boxes_iou3d_gpu(boxes_a: torch.Tensor, boxes_b: torch.Tensor) -> torch.Tensor:
    """boxes_a (N, 7), boxes_b (M, 7) [x, y, z, h, w, l, ry] (y = box bottom) -> (N, M) 3D IoU"""
    bev_a = boxes3d_to_bev_torch(boxes_a).float().contiguous()
    bev_b = boxes3d_to_bev_torch(boxes_b).float().contiguous()
    overlaps_bev = torch.empty((boxes_a.shape[0], boxes_b.shape[0]), dtype=torch.float32, device=boxes_a.device)
    iou3d_cuda.boxes_overlap_bev_gpu(bev_a, bev_b, overlaps_bev)

    a_top, a_bot = (boxes_a[:, 1] - boxes_a[:, 3]).view(-1, 1), boxes_a[:, 1].view(-1, 1)
    b_top, b_bot = (boxes_b[:, 1] - boxes_b[:, 3]).view(1, -1), boxes_b[:, 1].view(1, -1)
    overlaps_h = torch.clamp(torch.min(a_bot, b_bot) - torch.max(a_top, b_top), min=0)
    overlaps_3d = overlaps_bev * overlaps_h
    vol_a = (boxes_a[:, 3] * boxes_a[:, 4] * boxes_a[:, 5]).view(-1, 1)
    vol_b = (boxes_b[:, 3] * boxes_b[:, 4] * boxes_b[:, 5]).view(1, -1)
    return overlaps_3d / torch.clamp(vol_a + vol_b - overlaps_3d, min=1e-7)


def _nms(boxes: torch.Tensor, scores: torch.Tensor, thresh: float, normal: int) -> torch.Tensor:
    order = torch.sort(scores, dim=0, descending=True, stable=True)[1]
    sorted_boxes = boxes[order].contiguous()
    keep, num = iou3d_cuda.nms_device(sorted_boxes, thresh, normal)
    # the result length is data dependent, so one 4-byte read-back is inherent to this API; the
    # mask build AND the greedy reduce already ran on the device (the reference copies the whole
    # N x N/64 mask to the host and reduces it there).
    return order[keep[: int(num.item())]].contiguous()


def nms_gpu(boxes: torch.Tensor, scores: torch.Tensor, thresh: float) -> torch.Tensor:
    """rotated-IoU NMS.  boxes (N, 5) [x1, y1, x2, y2, ry], scores (N) -> kept indices"""
    return _nms(boxes, scores, thresh, 0)


def nms_normal_gpu(boxes: torch.Tensor, scores: torch.Tensor, thresh: float) -> torch.Tensor:
    """axis-aligned-IoU NMS (ry ignored), the RPN's default (config.py:94)"""
    return _nms(boxes, scores, thresh, 1)
