"""Detection post-processing and the device-resident detection -> tracking hand-off.

The reference finishes detection per frame on the host: decode (tools/eval.py:108-116), score threshold + rotated
NMS with `.cpu().numpy()` of boxes, scores and the (K, 512) RoI features (:176-193), KITTI txt + `feat/%06d.npy`
on disk (:245-274); tracking reads both back per frame (:358-364) and uploads the features again
(tracker.py:46-48).  Here the whole batch is finished on the device and STAYS there:

    boxes = decode_rcnn_boxes(rois, rcnn_reg, ...)                       one launch
    cache = select_detections(boxes, rcnn_cls, rcnn_feat, 0.2, 0.1)      sort + ONE batched rotated NMS + gathers
    cost, link, start, end = cache.associate(prev, cur, link_head, se_head, w_app, w_iou, w_dis)

`DetectionCache` keeps zero-padded (B, M, ...) tensors and a device-side count per frame; nothing is copied to
the host until the caller asks for the assignment problem (`to_host`): one D2H of the count vector per batch
and one of each (P, D) cost matrix, instead of three `.cpu()` calls per frame plus the disk round trip.
(The 2-D image-box validity filter of save_kitti_detection_format, eval.py:247-256, needs the camera calibration
and belongs to the KITTI writer — out of scope)
"""
import ctypes
from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from .. import _lib as L
from ..ext import iou3d_cuda
from .affinity import pairwise_affinity
from .association import association_cost
from .proposal import CLS_MEAN_SIZE

_f32 = torch.float32


@torch.no_grad()
def decode_rcnn_boxes(rois: torch.Tensor, rcnn_reg: torch.Tensor, loc_scope: float = 1.5, loc_bin_size: float = 0.5,
                      num_head_bin: int = 9, anchor_size=CLS_MEAN_SIZE, avg_by_bin: bool = True) -> torch.Tensor:
    """rois (P, 7), rcnn_reg (P, C) -> pred_boxes3d (P, 7) = decode_bbox_target(rois, rcnn_reg, get_xz_fine=True,
    get_y_by_bin=False, get_ry_fine=True) (tools/eval.py:108-116; defaults = cfg.RCNN.* of config.py:118-124)"""
    lib = L.load()
    rois = rois.contiguous().to(_f32)
    rcnn_reg = rcnn_reg.contiguous().to(_f32)
    P, C = rcnn_reg.shape
    out = torch.empty((P, 7), dtype=_f32, device=rois.device)
    anchor = (ctypes.c_float * 3)(*[float(a) for a in anchor_size])
    L.check(lib.jm_decode_rcnn_boxes(P, C, L.dev(rois, _f32, "rois"), L.dev(rcnn_reg, _f32, "rcnn_reg"),
                                     float(loc_scope), float(loc_bin_size), int(num_head_bin), anchor,
                                     int(bool(avg_by_bin)), ctypes.c_void_p(out.data_ptr()), L.stream_ptr()),
            "decode_rcnn_boxes")
    return out


@torch.no_grad()
def boxes_iou3d_batched(boxes_a: torch.Tensor, boxes_b: torch.Tensor, counts_b: Optional[torch.Tensor] = None) -> torch.Tensor:
    """boxes_a (B, Na, 7), boxes_b (B, Nb, 7) -> (B, Na, Nb): `boxes_iou3d_gpu` for every frame in ONE launch (the
    RoI sampler's per-frame loop, proposal_target_layer.py:137-151).  counts_b (B) int32: valid boxes per frame in
    boxes_b (trailing all-zero ground-truth rows are padding, :141-145); columns beyond it come back as 0."""
    lib = L.load()
    a, b = boxes_a.contiguous().to(_f32), boxes_b.contiguous().to(_f32)
    B, Na, Nb = a.shape[0], a.shape[1], b.shape[1]
    out = torch.empty((B, Na, Nb), dtype=_f32, device=a.device)
    cb = L.dev(counts_b.contiguous(), torch.int32, "counts_b") if counts_b is not None else None
    L.check(lib.jm_boxes_iou3d_batched(B, Na, L.dev(a, _f32, "boxes_a"), Nb, L.dev(b, _f32, "boxes_b"), cb,
                                       ctypes.c_void_p(out.data_ptr()), L.stream_ptr()), "boxes_iou3d_batched")
    return out


@dataclass
class DetectionCache:
    """detections of B frames, resident on the device, zero padded to M slots per frame"""
    boxes: torch.Tensor        # (B, M, 7) [x, y, z, h, w, l, ry]
    scores: torch.Tensor       # (B, M) sigmoid scores (what the tracker reads from the txt, eval.py:270)
    raw_scores: torch.Tensor   # (B, M) logits
    feats: torch.Tensor        # (B, M, C) RoI features (the reference's feat/%06d.npy rows, eval.py:273-274)
    count: torch.Tensor        # (B) int32 valid detections per frame
    roi_index: torch.Tensor    # (B, M) int64 RoI slot each detection came from
    _count_host: Optional[list] = None

    def counts_host(self) -> list:
        """the per-frame counts on the host: ONE small D2H per batch, cached"""
        if self._count_host is None:
            self._count_host = [int(v) for v in self.count.cpu().tolist()]
            from .pointnet2.pointnet2_utils import check_fps_failures
            check_fps_failures()          # (the host has just waited for the device: the sampling flags are there too)
        return self._count_host

    def frame(self, b: int) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """(boxes (K, 7), scores (K), feats (K, C)) of frame b — device views, no copies"""
        k = self.counts_host()[b]
        return self.boxes[b, :k], self.scores[b, :k], self.feats[b, :k]

    @torch.no_grad()
    def associate(self, prev: int, cur: int, link_head, se_head, w_app: float, w_iou: float, w_dis: float,
                  pred_boxes: Optional[torch.Tensor] = None, pred_feats: Optional[torch.Tensor] = None):
        """affinity + association cost between frame `prev`'s detections (or explicit Kalman-predicted track boxes /
        features) and frame `cur`'s detections, all on the device (tracker.py:81-112 + data_association.py:42-45):
        returns (cost (P, D), link (P, D), start logits (D), end logits (P)); None when either side is empty."""
        pb, _, pf = self.frame(prev)
        if pred_boxes is not None:
            pb, pf = pred_boxes, pred_feats
        db, _, df = self.frame(cur)
        if pb.shape[0] == 0 or db.shape[0] == 0:
            return None
        link, start, end = pairwise_affinity(pf, df, link_head, se_head)
        cost = association_cost(pb, db, link, w_app, w_iou, w_dis)
        return cost, link, start, end

    def to_host(self, b: int):
        """numpy (boxes, scores, feats) of one frame — the arrays the reference writes to disk"""
        bx, sc, ft = self.frame(b)
        return bx.cpu().numpy(), sc.cpu().numpy(), ft.cpu().numpy()


@torch.no_grad()
def select_detections(pred_boxes3d: torch.Tensor, raw_scores: torch.Tensor, feats: torch.Tensor,
                      score_thresh: float = 0.2, nms_thresh: float = 0.1) -> DetectionCache:
    """pred_boxes3d (B, M, 7), raw_scores (B, M) logits, feats (B, M, C) -> DetectionCache.
    tools/eval.py:171-193 for every frame at once: keep sigmoid(score) > score_thresh, rotated BEV NMS in
    descending raw-score order (stable), gather the survivors — no host round trip, three launches (csrc/detections.hip:
    sort + BEV form, jm_nms_batched, gather)."""
    B, M = raw_scores.shape
    dev = raw_scores.device
    lib = L.load()
    C = feats.shape[2]
    boxes_c, raw_c, feats_c = pred_boxes3d.contiguous(), raw_scores.contiguous(), feats.contiguous()
    boxes_in = L.dev(boxes_c, _f32, "pred_boxes3d")
    order = torch.empty((B, M), dtype=torch.int64, device=dev)
    counts = torch.empty((B,), dtype=torch.int32, device=dev)
    bev = torch.empty((B, M, 5), dtype=_f32, device=dev)
    L.check(lib.jm_detections_sort(B, M, boxes_in, L.dev(raw_c, _f32, "raw_scores"), float(score_thresh), ctypes.c_void_p(order.data_ptr()),
                                   ctypes.c_void_p(counts.data_ptr()), ctypes.c_void_p(bev.data_ptr()), L.stream_ptr()), "detections_sort")
    keep, num_keep = iou3d_cuda.nms_batched_device(bev, counts, nms_thresh, 0)           # positions in sorted order
    boxes = torch.empty((B, M, 7), dtype=_f32, device=dev)
    scores = torch.empty((B, M), dtype=_f32, device=dev)
    raw_out = torch.empty((B, M), dtype=_f32, device=dev)
    out_feats = torch.empty((B, M, C), dtype=_f32, device=dev)
    count = torch.empty((B,), dtype=torch.int32, device=dev)
    slot = torch.empty((B, M), dtype=torch.int64, device=dev)
    L.check(lib.jm_detections_gather(B, M, C, boxes_in, L.dev(raw_c, _f32, "raw_scores"), L.dev(feats_c, _f32, "feats"),
                                     ctypes.c_void_p(order.data_ptr()), ctypes.c_void_p(keep.data_ptr()), L.dev(num_keep, torch.int32, "num_keep"),
                                     ctypes.c_void_p(boxes.data_ptr()), ctypes.c_void_p(scores.data_ptr()), ctypes.c_void_p(raw_out.data_ptr()),
                                     ctypes.c_void_p(out_feats.data_ptr()), ctypes.c_void_p(count.data_ptr()), ctypes.c_void_p(slot.data_ptr()),
                                     L.stream_ptr()), "detections_gather")
    return DetectionCache(boxes=boxes, scores=scores, raw_scores=raw_out, feats=out_feats, count=count, roi_index=slot)
