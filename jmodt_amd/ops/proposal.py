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

`decode_rpn_proposals` is the bin-based box decode in front of it (bbox_transform.py:27-260 in the RPN's
configuration); `proposal_layer` chains the two.  The decode's parity is UNPINNED: the reference function
cannot be imported where the fixtures are made (jmodt.config needs easydict), so it is tested against the
oracle's restatement and an independent torch restatement only.
"""
import ctypes
from typing import Tuple

import torch

from .. import _lib as L

_f32 = torch.float32


NMS_EVALS = []    # (workspace, counter offset, problems, pre-NMS pair count of the full mask) while the profiler is on


def nms_evals():
    """(IoU evaluations the lazy first-K NMS did, evaluations of the full pair masks, calls) of the recorded
    jm_proposal_select calls; clears the record (synchronises)"""
    torch.cuda.synchronize()
    done = full = 0
    for ws, off, nprob, tri in NMS_EVALS:
        done += int(ws[off:off + 8 * nprob].view(torch.int64).sum().item())
        full += tri
    calls = len(NMS_EVALS)
    NMS_EVALS.clear()
    return done, full, calls


def argsort_desc_stable(scores: torch.Tensor) -> torch.Tensor:
    """(B, N) float32 -> (B, N) int64 = torch.sort(scores, dim=1, descending=True, stable=True)[1]: one workgroup per row in LDS
    (csrc/sort.hip) for N <= 16384, the library sort above that"""
    lib = L.load()
    B, N = scores.shape
    if not (scores.is_cuda and scores.dtype == _f32 and scores.is_contiguous() and lib.jm_argsort_desc_supported(N)):
        return torch.sort(scores, dim=1, descending=True, stable=True)[1].contiguous()
    order = torch.empty((B, N), dtype=torch.int64, device=scores.device)
    L.check(lib.jm_argsort_desc_stable(B, N, L.dev(scores, _f32, "scores"), ctypes.c_void_p(order.data_ptr()), L.stream_ptr()), "argsort_desc")
    return order


def _select(scores, proposals, distance_based, pre, post, thresh, normal):
    lib = L.load()
    B, N = scores.shape
    scores = scores.contiguous().to(_f32)
    proposals = proposals.contiguous().to(_f32)
    # stable, so equal scores keep their index order (the reference's torch.sort leaves that unspecified)
    order = argsort_desc_stable(scores)
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
    from ..profile import prof
    if prof.enabled and prof.only is None and normal:
        off = base - ws.data_ptr() + lib.jm_proposal_select_evals_offset(B, int(distance_based), pre)
        pre1 = int(pre * 0.7) if distance_based else pre
        tri = B * (pre1 * (pre1 - 1) // 2 + (pre - pre1) * (pre - pre1 - 1) // 2)      # upper bound: bands filled to their budgets
        NMS_EVALS.append((ws, off, B * (2 if distance_based else 1), tri))
    return out_boxes, out_scores


CLS_MEAN_SIZE = (1.52563191462, 1.62856739989, 3.88311640418)   # cfg.CLS_MEAN_SIZE[0] (config.py:38): h, w, l


@torch.no_grad()
def decode_rpn_proposals(xyz: torch.Tensor, rpn_reg: torch.Tensor, loc_scope: float = 3.0, loc_bin_size: float = 0.5,
                         num_head_bin: int = 12, anchor_size=CLS_MEAN_SIZE, avg_by_bin: bool = True) -> torch.Tensor:
    """xyz (B, N, 3), rpn_reg (B, N, C) -> proposals (B, N, 7) [x, y_bottom, z, h, w, l, ry]
    = decode_bbox_target(xyz, rpn_reg, ...) followed by `proposals[:, 1] += proposals[:, 3] / 2`
    (proposal_layer.py:24-34; defaults = cfg.RPN.* of config.py:65-68, BBOX_AVG_BY_BIN of config.py:207)"""
    lib = L.load()
    B, N, C = rpn_reg.shape
    xyz = xyz.contiguous().to(_f32)
    out = torch.empty((B, N, 7), dtype=_f32, device=xyz.device)
    anchor = (ctypes.c_float * 3)(*[float(a) for a in anchor_size])
    if rpn_reg.is_cuda and rpn_reg.dtype == _f32 and not rpn_reg.is_contiguous() and min(rpn_reg.stride()) >= 1:
        # a strided view — the heads' own (B, C, N) output seen as (B, N, C): read in place, one coalesced row per channel
        L.check(lib.jm_decode_rpn_proposals_strided(B, N, C, L.dev(xyz, _f32, "xyz"), ctypes.c_void_p(rpn_reg.data_ptr()), rpn_reg.stride(0),
                                                    rpn_reg.stride(1), rpn_reg.stride(2), float(loc_scope), float(loc_bin_size),
                                                    int(num_head_bin), anchor, int(bool(avg_by_bin)), ctypes.c_void_p(out.data_ptr()),
                                                    L.stream_ptr()), "decode_rpn_proposals")
        return out
    rpn_reg = rpn_reg.contiguous().to(_f32)
    L.check(lib.jm_decode_rpn_proposals(B * N, C, L.dev(xyz, _f32, "xyz"), L.dev(rpn_reg, _f32, "rpn_reg"),
                                        float(loc_scope), float(loc_bin_size), int(num_head_bin), anchor,
                                        int(bool(avg_by_bin)), ctypes.c_void_p(out.data_ptr()), L.stream_ptr()),
            "decode_rpn_proposals")
    return out


@torch.no_grad()
def proposal_layer(rpn_scores: torch.Tensor, rpn_reg: torch.Tensor, xyz: torch.Tensor, pre_nms_top_n: int = 9000,
                   post_nms_top_n: int = 100, nms_thresh: float = 0.8, nms_type: str = "normal",
                   distance_based: bool = True, **decode_kw) -> Tuple[torch.Tensor, torch.Tensor]:
    """ProposalLayer.forward (proposal_layer.py:16-55) for a whole batch: decode + selection, 6 launches.
    Defaults = the TEST configuration (config.py:226-230)."""
    proposals = decode_rpn_proposals(xyz, rpn_reg, **decode_kw)
    if distance_based:
        return distance_based_proposal(rpn_scores, proposals, pre_nms_top_n, post_nms_top_n, nms_thresh, nms_type)
    return score_based_proposal(rpn_scores, proposals, pre_nms_top_n, post_nms_top_n, nms_thresh)


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
