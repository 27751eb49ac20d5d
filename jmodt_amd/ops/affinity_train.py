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


# ---------------------------------------------------------------------------------------------------------
# Static-shape, sync-free form (MI355X-native): the same losses and gradients without a Python loop over frame pairs,
# without torch.unique's data-dependent shapes and without a single device -> host read.  Every RoI slot stays in
# place; membership is carried by masks:
#   pooled_i   = mean feature of the foreground RoIs sharing RoI i's track id           (get_unique_tid_feature, rcnn.py:145-156)
#   rep_i      = RoI i is the FIRST foreground RoI of its track id                      (one representative per unique id)
#   valid[i,j] = rep_prev[i] & rep_next[j]                                              (the reference's P x D matrix entries)
# and the reference's per-pair tensors are these (R, R) matrices restricted to `valid`.  Sums over valid entries equal
# the reference's sums whatever the order, so the L1 means and their gradients are identical.  Costs R^2 = 4096 pair
# rows per frame pair instead of <= ~150, i.e. ~9 GFLOP forward+backward per GPU and step — 0.3 ms of GEMMs instead
# of ~4 ms of launch- and sync-bound Python (the finetune step of BASELINE configs[3] becomes GPU-bound).
# ---------------------------------------------------------------------------------------------------------

def training_affinity_static(roi_features: torch.Tensor, gt_tids: torch.Tensor, link_layer: nn.Module,
                             se_layer: nn.Module) -> Dict[str, torch.Tensor]:
    """roi_features (2F, R, C), gt_tids (2F, R) as `training_affinity`.  Returns (F, R, R) / (F, R) tensors:
    link / gt_links / valid, start / gt_starts / start_valid (per NEXT RoI), end / gt_ends / end_valid (per PREV RoI);
    start / end are raw logits (sigmoid in the loss, train_functions.py:313,317)."""
    prev_t, next_t = gt_tids[0::2], gt_tids[1::2]                     # (F, R)
    prev_f, next_f = roi_features[0::2], roi_features[1::2]           # (F, R, C)
    F_, R, C = prev_f.shape

    def pool(tid, feat):
        fg = tid > 0
        same = (tid.unsqueeze(2) == tid.unsqueeze(1)) & fg.unsqueeze(2) & fg.unsqueeze(1)       # (F, R, R)
        w = same.to(feat.dtype)
        pooled = torch.bmm(w / w.sum(dim=2, keepdim=True).clamp_min(1.0), feat)                 # mean over the track's RoIs
        earlier = torch.tril(same, diagonal=-1).any(dim=2)
        return pooled, fg & ~earlier

    pp, rep_p = pool(prev_t, prev_f)
    pn, rep_n = pool(next_t, next_f)
    both = (rep_p.any(dim=1) & rep_n.any(dim=1)).view(F_, 1)          # rcnn.py:230: pairs without foreground on a side are skipped
    rep_p, rep_n = rep_p & both, rep_n & both
    valid = rep_p.unsqueeze(2) & rep_n.unsqueeze(1)                   # (F, R, R)
    gt_links = ((prev_t.unsqueeze(2) == next_t.unsqueeze(1)) & valid).to(pp.dtype)
    cor = torch.abs(pp.unsqueeze(2) - pn.unsqueeze(1))                # (F, R, R, C)
    scores = link_layer(cor.reshape(F_ * R * R, C, 1)).view(F_, R, R)
    neg = torch.finfo(scores.dtype).min / 4                           # finite: fully masked rows stay finite (and are masked out)
    over_next = torch.softmax(scores.masked_fill(~rep_n.unsqueeze(1), neg), dim=2)
    over_prev = torch.softmax(scores.masked_fill(~rep_p.unsqueeze(2), neg), dim=1)
    link = (over_next + over_prev) / 2
    vf = valid.to(cor.dtype).unsqueeze(-1)
    n_prev = rep_p.sum(dim=1).clamp_min(1).view(F_, 1, 1).to(cor.dtype)
    n_next = rep_n.sum(dim=1).clamp_min(1).view(F_, 1, 1).to(cor.dtype)
    start_feat = (cor * vf).sum(dim=1) / n_prev                       # mean over the prev representatives -> (F, R_next, C)
    end_feat = (cor * vf).sum(dim=2) / n_next                         # mean over the next representatives -> (F, R_prev, C)
    start = se_layer(start_feat.reshape(F_ * R, C, 1)).view(F_, R)
    end = se_layer(end_feat.reshape(F_ * R, C, 1)).view(F_, R)
    return dict(link=link, gt_links=gt_links, valid=valid, start=start, gt_starts=1 - gt_links.sum(dim=1), start_valid=rep_n,
                end=end, gt_ends=1 - gt_links.sum(dim=2), end_valid=rep_p)


def reid_loss_static(out: Dict[str, torch.Tensor], counts: torch.Tensor = None, link_weight: float = 1.0,
                     se_weight: float = 1.0) -> Tuple[torch.Tensor, torch.Tensor]:
    """(loss, local counts (3,)): train_functions.py:282-329 on the static form; `counts` (3,) overrides the
    denominators (the GLOBAL element counts of a data-parallel step)"""
    v, sv, ev = out["valid"], out["start_valid"], out["end_valid"]
    local = torch.stack([v.sum(), sv.sum(), ev.sum()]).to(out["link"].dtype)
    den = (local if counts is None else counts.to(local.dtype)).clamp_min(1.0)
    l_link = ((out["link"] - out["gt_links"]).abs() * v).sum() / den[0]
    l_start = ((torch.sigmoid(out["start"]) - out["gt_starts"]).abs() * sv).sum() / den[1]
    l_end = ((torch.sigmoid(out["end"]) - out["gt_ends"]).abs() * ev).sum() / den[2]
    return link_weight * l_link + se_weight * (l_start + l_end), local


def finetune_step_static(roi_features: torch.Tensor, gt_tids: torch.Tensor, link_layer: nn.Module, se_layer: nn.Module,
                         optimizer: torch.optim.Optimizer, world: int = 1) -> torch.Tensor:
    """`finetune_step` without host synchronisation: static-shape forward/backward, the three global element counts
    and the gradients all-reduced on the device (RCCL), Adam; returns the whole-batch loss as a DEVICE scalar"""
    import torch.distributed as tdist
    optimizer.zero_grad(set_to_none=True)
    out = training_affinity_static(roi_features, gt_tids, link_layer, se_layer)
    with torch.no_grad():
        counts = torch.stack([out["valid"].sum(), out["start_valid"].sum(), out["end_valid"].sum()]).to(torch.float32)
        if world > 1:
            tdist.all_reduce(counts, op=tdist.ReduceOp.SUM)
    loss, _ = reid_loss_static(out, counts)
    loss.backward()
    params = list(link_layer.parameters()) + list(se_layer.parameters())
    jdist.allreduce_gradients(params, world=world, average=False)
    optimizer.step()
    total = loss.detach().clone()
    if world > 1:
        tdist.all_reduce(total, op=tdist.ReduceOp.SUM)
    return total
