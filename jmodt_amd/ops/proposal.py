"""RPN proposal selection for a whole batch on the device, with no host round trips.

Mirror of the selection half of jmodt/detection/layers/proposal_layer.py (ProposalLayer.forward
:34-55 after the decode, distance_based_proposal :57-117, score_based_proposal :119-144): score
sort, split into the (0, 40] m and (40, 80] m depth bands with the 70 % / 30 % pre-NMS budgets
(empty far band -> the next slice of the near band), BEV NMS per band, post-NMS budgets,
zero-padded (B, POST_TOP_N, 7) result.

The reference walks the frames in Python and runs 2 B NMS calls one after the other, each with
a device->host copy of the whole suppression mask.  Here the score sort is one torch call and the
rest is `jm_proposal_select` (csrc/proposal.hip): one compaction kernel, the 2 B NMS problems as
ONE batched launch pair with device-side counts, one stitch kernel — the only sync is whatever
the caller does with the result.

The bin-based box decode (bbox_transform.py:28-132) is not part of this module: it takes decoded
proposals (B, N, 7) [x, y_bottom, z, h, w, l, ry].
"""
import ctypes
from typing import Tuple

import torch

from .. import _lib as L

_f32 = torch.float32


def _select(scores, proposals, distance_based, pre, post, thresh, normal):
    lib = L.load()
    B, N = scores.shape
    scores = scores.contiguous().to(_f32)
    proposals = proposals.contiguous().to(_f32)
    # stable, so equal scores keep their index order (the reference's torch.sort leaves that unspecified)
    order = torch.sort(scores, dim=1, descending=True, stable=True)[1].contiguous()
    out_boxes = torch.empty((B, post, 7), dtype=_f32, device=scores.device)
    out_scores = torch.empty((B, post), dtype=_f32, device=scores.device)
    ws_bytes = lib.jm_proposal_select_workspace_bytes(B, int(distance_based), pre)
    ws = torch.empty((ws_bytes + 256,), dtype=torch.uint8, device=scores.device)
    base = (ws.data_ptr() + 255) // 256 * 256
    L.check(lib.jm_proposal_select(B, N, L.dev(scores, _f32, "scores"), L.dev(proposals, _f32, "proposals"),
                                   L.dev(order, torch.int64, "order"), int(distance_based), int(pre), int(post),
                                   float(thresh), int(normal), ctypes.c_void_p(out_boxes.data_ptr()),
                                   ctypes.c_void_p(out_scores.data_ptr()), ctypes.c_void_p(base), ws_bytes,
                                   L.stream_ptr()), "proposal_select")
    return out_boxes, out_scores


@torch.no_grad()
def distance_based_proposal(scores: torch.Tensor, proposals: torch.Tensor, pre_nms_top_n: int, post_nms_top_n: int,
                            nms_thresh: float, nms_type: str = "normal") -> Tuple[torch.Tensor, torch.Tensor]:
    """scores (B, N), proposals (B, N, 7) -> ret_bbox3d (B, post_nms_top_n, 7), ret_scores (B, post_nms_top_n)
    (proposal_layer.py:34-117 for every frame of the batch at once)"""
    if nms_type not in ("normal", "rotate"):
        raise NotImplementedError(nms_type)  # proposal_layer.py:107-108
    return _select(scores, proposals, True, pre_nms_top_n, post_nms_top_n, nms_thresh, nms_type == "normal")


@torch.no_grad()
def score_based_proposal(scores: torch.Tensor, proposals: torch.Tensor, pre_nms_top_n: int, post_nms_top_n: int,
                         nms_thresh: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """proposal_layer.py:119-144 (always rotated NMS) for every frame of the batch at once"""
    return _select(scores, proposals, False, pre_nms_top_n, post_nms_top_n, nms_thresh, False)
