"""Tracker association cost terms on the gfx950 kernel.

Mirrors jmodt/tracking/data_association.py: `boxes_dist_gpu` (:10-28) and the cost matrix both
solvers build (:42-45 ortools_solve, :117-119 hungarian_match):
    link_matrix = link_scores * w_app + boxes_iou3d_gpu(pred, det) * w_iou + boxes_dist_gpu(pred, det) * w_dis
One launch; the result stays on the device (the reference copies it to the host for the CBC /
Hungarian solver, which is out of scope here)."""
import ctypes
from typing import Optional, Tuple

import torch

from .. import _lib as L

_f32 = torch.float32


def _run(pred_boxes, det_boxes, link_scores, w_app, w_iou, w_dis, want_cost, want_parts):
    lib = L.load()
    a, b = pred_boxes.to(_f32).contiguous(), det_boxes.to(_f32).contiguous()
    P, D = a.shape[0], b.shape[0]
    dev = a.device
    cost = torch.empty((P, D), dtype=_f32, device=dev) if want_cost else None
    iou = torch.empty((P, D), dtype=_f32, device=dev) if want_parts else None
    dist = torch.empty((P, D), dtype=_f32, device=dev) if want_parts else None
    link = link_scores.to(_f32).contiguous() if link_scores is not None else None

    def ptr(t):
        return ctypes.c_void_p(t.data_ptr()) if t is not None else None
    L.check(lib.jm_association_cost(P, L.dev(a, _f32, "pred_boxes"), D, L.dev(b, _f32, "det_boxes"),
                                    L.dev(link, _f32, "link_scores") if link is not None else None,
                                    float(w_app), float(w_iou), float(w_dis), ptr(cost), ptr(iou), ptr(dist),
                                    L.stream_ptr()), "association_cost")
    return cost, iou, dist


def boxes_dist_gpu(boxes_a: torch.Tensor, boxes_b: torch.Tensor) -> torch.Tensor:
    """boxes (M,7), (N,7) [x,y,z,h,w,l,ry] -> (M,N): 1 - centre distance / farthest corner-pair distance"""
    return _run(boxes_a, boxes_b, None, 0.0, 0.0, 0.0, False, True)[2]


def association_cost(pred_boxes: torch.Tensor, det_boxes: torch.Tensor, link_scores: Optional[torch.Tensor],
                     w_app: float, w_iou: float, w_dis: float, return_parts: bool = False
                     ) -> Tuple[torch.Tensor, ...]:
    """(P,D) cost matrix handed to the assignment solver; with return_parts also (iou3d, dist)"""
    cost, iou, dist = _run(pred_boxes, det_boxes, link_scores, w_app, w_iou, w_dis, True, return_parts)
    return (cost, iou, dist) if return_parts else cost
