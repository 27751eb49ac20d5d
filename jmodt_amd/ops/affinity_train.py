"""Training-time pairwise affinity and the finetune DP step.

Mirrors jmodt/detection/modeling/rcnn.py:145-156,204-287 (per (prev, next) frame pair:
mean-pool foreground RoI features per track id, `cor = |prev_i - next_j|`, link head + dual
softmax, start/end features = cor.mean(0) / cor.mean(1) through the se head, ground-truth link
matrix from track-id equality) and the re-id losses of
jmodt/detection/modeling/train_functions.py:282-329 (L1 on links, L1 on sigmoid(start/end)).

Forward+backward here go through torch autograd on the GPU (hipBLASLt GEMMs): the fused fp32-MFMA
kernels of jmodt_amd/csrc/affinity.hip are forward-only so far (backward = DESIGN.md §7 "next").
The *inference* affinity (tracker.py:81-112) never comes through this module.
"""
from typing import Dict, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import dist as jdist


def get_unique_tid_feature(fg_tid: torch.Tensor, fg_feat: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """mean feature per distinct track id, ids ascending (rcnn.py:145-156)"""
    uniq, inv = torch.unique(fg_tid, return_inverse=True)   # sorted
    onehot = F.one_hot(inv, num_classes=uniq.numel()).to(fg_feat.dtype).t()      # (U, n)
    onehot = onehot / onehot.sum(dim=1, keepdim=True)
    return uniq, onehot @ fg_feat


def training_affinity(roi_features: torch.Tensor, gt_tids: torch.Tensor, link_layer: nn.Module,
                      se_layer: nn.Module) -> Dict[str, torch.Tensor]:
    """roi_features (2F, R, C) RoI features of F interleaved (prev, next) frame pairs, gt_tids
    (2F, R) track id per RoI (<= 0: background).  Returns rcnn_link (sum n_p*n_n, 1), rcnn_start,
    rcnn_end (raw logits, (., 1)) and gt_links / gt_starts / gt_ends exactly as rcnn.py:204-287."""
    num_frames = gt_tids.shape[0]
    prev_tids, next_tids = gt_tids[0::2], gt_tids[1::2]
    prev_feats, next_feats = roi_features[0::2], roi_features[1::2]
    rcnn_link, start_feats, end_feats, gt_links, gt_starts, gt_ends = [], [], [], [], [], []
    for i in range(num_frames // 2):
        pm, nm = prev_tids[i] > 0, next_tids[i] > 0
        if pm.sum() == 0 or nm.sum() == 0:
            continue
        p_tid, p_feat = get_unique_tid_feature(prev_tids[i][pm], prev_feats[i][pm])
        n_tid, n_feat = get_unique_tid_feature(next_tids[i][nm], next_feats[i][nm])
        link_gt = (p_tid.unsqueeze(1) == n_tid).float()
        cor = torch.abs(p_feat.unsqueeze(1) - n_feat.unsqueeze(0))              # (P, D, C), broadcast not repeat
        P, D, C = cor.shape
        scores = link_layer(cor.reshape(P * D, C, 1)).view(P, D)
        scores = (torch.softmax(scores, dim=1) + torch.softmax(scores, dim=0)) / 2
        rcnn_link.append(scores.reshape(P * D, 1))
        gt_links.append(link_gt.reshape(-1))
        gt_starts.append(1 - link_gt.sum(0))
        gt_ends.append(1 - link_gt.sum(1))
        start_feats.append(cor.mean(dim=0))
        end_feats.append(cor.mean(dim=1))
    dev, dt = roi_features.device, roi_features.dtype
    if not gt_links:
        empty = torch.zeros(0, device=dev, dtype=dt)
        return dict(rcnn_link=empty.view(0, 1), rcnn_start=empty.view(0, 1), rcnn_end=empty.view(0, 1),
                    gt_links=empty, gt_starts=empty, gt_ends=empty)
    return dict(rcnn_link=torch.cat(rcnn_link), gt_links=torch.cat(gt_links),
                rcnn_start=se_layer(torch.cat(start_feats).unsqueeze(-1)).squeeze(-1), gt_starts=torch.cat(gt_starts),
                rcnn_end=se_layer(torch.cat(end_feats).unsqueeze(-1)).squeeze(-1), gt_ends=torch.cat(gt_ends))


def reid_loss(out: Dict[str, torch.Tensor], link_weight: float = 1.0, se_weight: float = 1.0) -> torch.Tensor:
    """train_functions.py:282-329: mean L1 on links, mean L1 on sigmoid(start) and sigmoid(end)"""
    loss = out["rcnn_link"].new_zeros(())
    if out["gt_links"].numel():
        loss = loss + link_weight * F.l1_loss(out["rcnn_link"].view(-1), out["gt_links"], reduction="mean")
        loss = loss + se_weight * F.l1_loss(torch.sigmoid(out["rcnn_start"].view(-1)), out["gt_starts"], reduction="mean")
        loss = loss + se_weight * F.l1_loss(torch.sigmoid(out["rcnn_end"].view(-1)), out["gt_ends"], reduction="mean")
    return loss


def finetune_step(roi_features: torch.Tensor, gt_tids: torch.Tensor, link_layer: nn.Module, se_layer: nn.Module,
                  optimizer: torch.optim.Optimizer, world: int = 1) -> float:
    """One data-parallel finetune step (tools/train.py:96-107 trains only the link/se heads): local
    forward/backward on this rank's frame pairs, then ONE bucketed gradient all-reduce over RCCL.

    The reference computes the loss AFTER DataParallel has gathered every replica's outputs, i.e.
    each term is a mean over the links / starts / ends of the WHOLE batch.  To reproduce that
    gradient exactly, each rank back-propagates its local SUMS divided by the GLOBAL element counts
    (one 3-float all-reduce), and the gradient all-reduce is a plain SUM."""
    import torch.distributed as tdist
    optimizer.zero_grad(set_to_none=True)
    out = training_affinity(roi_features, gt_tids, link_layer, se_layer)
    counts = torch.tensor([out["gt_links"].numel(), out["gt_starts"].numel(), out["gt_ends"].numel()],
                          dtype=torch.float64, device=roi_features.device)
    if world > 1:
        tdist.all_reduce(counts, op=tdist.ReduceOp.SUM)
    loss = roi_features.new_zeros(())
    if out["gt_links"].numel():
        loss = loss + (out["rcnn_link"].view(-1) - out["gt_links"]).abs().sum() / counts[0].item()
        loss = loss + (torch.sigmoid(out["rcnn_start"].view(-1)) - out["gt_starts"]).abs().sum() / counts[1].item()
        loss = loss + (torch.sigmoid(out["rcnn_end"].view(-1)) - out["gt_ends"]).abs().sum() / counts[2].item()
        loss.backward()
    params = list(link_layer.parameters()) + list(se_layer.parameters())
    jdist.allreduce_gradients(params, world=world, average=False)
    optimizer.step()
    total = loss.detach().to(torch.float64).reshape(1)
    if world > 1:
        tdist.all_reduce(total, op=tdist.ReduceOp.SUM)
    return float(total.item())   # == reid_loss of the whole batch
