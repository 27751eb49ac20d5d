"""Training-time pairwise affinity and the finetune DP step.

Mirrors jmodt/detection/modeling/rcnn.py:145-156,204-287 (per (prev, next) frame pair:
mean-pool foreground RoI features per track id, `cor = |prev_i - next_j|`, link head + dual
softmax, start/end features = cor.mean(0) / cor.mean(1) through the se head, ground-truth link
matrix from track-id equality) and the re-id losses of
jmodt/detection/modeling/train_functions.py:282-329 (L1 on links, L1 on sigmoid(start/end)).

Three forms of the same computation:
  * `training_affinity` / `reid_loss` / `finetune_step`: the reference's op sequence (Python loop over frame pairs,
    torch.unique, data-dependent shapes) — the readable restatement, used by the tests as the bridge to the float64
    statement-by-statement copy of rcnn.py;
  * `training_affinity_static` / `reid_loss_static`: the static-shape, sync-free form in plain torch (masks instead of
    torch.unique) — runs on CPU tensors, which is what the world-size-2 gloo tests of the data-parallel step use;
  * `AffinityTrainState` / `affinity_train_loss` (GPU tensors): the static-shape form on the HAND-WRITTEN kernels of
    jmodt_amd/csrc/affinity_train.hip — forward, losses and the backward of both heads as fp32-MFMA GEMM chains that never
    materialise the |prev_i - next_j| pair tensor.  `finetune_step_static` takes this path for GPU tensors (no torch fallback
    on the GPU: a missing library raises).
The *inference* affinity (tracker.py:81-112) never comes through this module.
"""
import ctypes
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib as L
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
                  optimizer: torch.optim.Optimizer, world: Optional[int] = None, local: bool = False) -> float:
    """One data-parallel finetune step (tools/train.py:96-107 trains only the link/se heads): local
    forward/backward on this rank's frame pairs, then ONE bucketed gradient all-reduce over RCCL.

    The reference computes the loss AFTER DataParallel has gathered every replica's outputs, i.e.
    each term is a mean over the links / starts / ends of the WHOLE batch.  To reproduce that
    gradient exactly, each rank back-propagates its local SUMS divided by the GLOBAL element counts
    (one 3-float all-reduce), and the gradient all-reduce is a plain SUM.

    world: None (default) = the process group's size when one exists, else a single-process step; an int is a declaration
    that must equal the group's size (dist.group_world).  local=True: no collectives even inside a group."""
    import torch.distributed as tdist
    optimizer.zero_grad(set_to_none=True)
    out = training_affinity(roi_features, gt_tids, link_layer, se_layer)
    counts = torch.tensor([out["gt_links"].numel(), out["gt_starts"].numel(), out["gt_ends"].numel()],
                          dtype=torch.float64, device=roi_features.device)
    collective = jdist.collective_path(world, local)
    if collective:
        tdist.all_reduce(counts, op=tdist.ReduceOp.SUM)
    loss = roi_features.new_zeros(())
    if out["gt_links"].numel():
        loss = loss + (out["rcnn_link"].view(-1) - out["gt_links"]).abs().sum() / counts[0].item()
        loss = loss + (torch.sigmoid(out["rcnn_start"].view(-1)) - out["gt_starts"]).abs().sum() / counts[1].item()
        loss = loss + (torch.sigmoid(out["rcnn_end"].view(-1)) - out["gt_ends"]).abs().sum() / counts[2].item()
        loss.backward()
    params = list(link_layer.parameters()) + list(se_layer.parameters())
    jdist.allreduce_gradients(params, world=world, average=False, local=local)
    optimizer.step()
    total = loss.detach().to(torch.float64).reshape(1)
    if collective:
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


# ---------------------------------------------------------------------------------------------------------
# the static-shape form on the HIP kernels (csrc/affinity_train.hip)
# ---------------------------------------------------------------------------------------------------------
_f32, _i32 = torch.float32, torch.int32


def _head_tensors(head: nn.Sequential):
    """the six parameter tensors of a link / se head in jm_mlp3_t order (w1, b1, w2, b2, w3, b3)"""
    convs = [m for m in head.modules() if isinstance(m, nn.Conv1d)]
    if len(convs) != 3 or any(isinstance(m, nn.BatchNorm1d) for m in head.modules()) or any(c.bias is None for c in convs):
        raise NotImplementedError("affinity_train kernels support the reference head: 3 Conv1d(k=1) with bias, no BN")
    for m in head.modules():
        if isinstance(m, nn.Dropout) and m.p != 0 and head.training:
            raise NotImplementedError("affinity_train kernels: dropout p > 0 in training mode")     # config.py:166-169: DP_RATIO 0
    if convs[2].out_channels != 1 or any(c.kernel_size != (1,) for c in convs):
        raise NotImplementedError("affinity_train kernels: kernel_size 1, last layer 1-wide")
    return [convs[0].weight, convs[0].bias, convs[1].weight, convs[1].bias, convs[2].weight, convs[2].bias]


def _mlp3(tensors):
    ts = [t.detach() for t in tensors]
    for t in ts:
        L.dev(t, _f32, "affinity head parameter")
    c, h1, h2 = ts[0].shape[1], ts[0].shape[0], ts[2].shape[0]
    return L.Mlp3(c, h1, h2, *[ctypes.c_void_p(t.data_ptr()) for t in ts]), ts


class AffinityTrainState:
    """jm_affinity_train_prepare: pooled per-id features, representatives, targets and LOCAL loss-mean element counts of a
    batch of interleaved (prev, next) frames: roi_features (2F, R, C), gt_tids (2F, R).  `counts` (3,) lives on the device;
    a data-parallel step all-reduces it (SUM) before `affinity_train_loss`."""

    def __init__(self, roi_features: torch.Tensor, gt_tids: torch.Tensor):
        feats = roi_features.detach().to(_f32).contiguous()
        tids = gt_tids.detach().to(_f32).contiguous()
        if feats.dim() != 3 or tids.shape != feats.shape[:2] or feats.shape[0] % 2:
            raise ValueError(f"roi_features (2F, R, C) and gt_tids (2F, R) expected, got {tuple(feats.shape)} / {tuple(tids.shape)}")
        self.F, self.R, self.C = feats.shape[0] // 2, feats.shape[1], feats.shape[2]
        dev, F_, R, C = feats.device, self.F, self.R, self.C
        self.roi_features = roi_features      # as given: when it requires grad, affinity_train_loss sends d(loss)/d(features) back
        self.tids = tids
        self.pooled_prev = torch.empty((F_ * R, C), dtype=_f32, device=dev)
        self.pooled_next = torch.empty((F_ * R, C), dtype=_f32, device=dev)
        self.rep_prev = torch.empty((F_, R), dtype=_i32, device=dev)
        self.rep_next = torch.empty((F_, R), dtype=_i32, device=dev)
        self.n_pair = torch.empty((F_, 2), dtype=_i32, device=dev)
        self.gt_starts = torch.empty((F_, R), dtype=_f32, device=dev)
        self.gt_ends = torch.empty((F_, R), dtype=_f32, device=dev)
        self.counts = torch.empty((3,), dtype=_f32, device=dev)
        rep_ws = torch.empty((2 * F_, R), dtype=_i32, device=dev)
        L.check(L.load().jm_affinity_train_prepare(
            F_, R, C, L.dev(feats, _f32, "roi_features"), L.dev(tids, _f32, "gt_tids"), L.dev(self.pooled_prev, _f32, "pooled_prev"),
            L.dev(self.pooled_next, _f32, "pooled_next"), L.dev(rep_ws, _i32, "rep_ws"), L.dev(self.rep_prev, _i32, "rep_prev"),
            L.dev(self.rep_next, _i32, "rep_next"), L.dev(self.n_pair, _i32, "n_pair"), L.dev(self.gt_starts, _f32, "gt_starts"),
            L.dev(self.gt_ends, _f32, "gt_ends"), L.dev(self.counts, _f32, "counts"), L.stream_ptr()), "affinity_train_prepare")


OVERLAP_SE = False


def _train_steps(st: AffinityTrainState, counts: torch.Tensor, link_t, se_t, link_weight: float, se_weight: float,
                 want_outputs: bool, want_dfeat: bool = False):
    """both heads' forward + loss + backward: returns (loss parts link (F,), se (F, 2), gradient tensors [6 link, 6 se],
    outputs dict or None); want_dfeat: + d(loss)/d(RoI features) (2F, R, C) as a fifth value (joint training)"""
    from .pointnet2.pyramid import side_stream
    lib = L.load()
    dev = st.pooled_prev.device
    F_, R = st.F, st.R
    link, keep_l = _mlp3(link_t)
    se, keep_s = _mlp3(se_t)
    g_link = [torch.empty_like(t) for t in keep_l]
    g_se = [torch.empty_like(t) for t in keep_s]
    gl = L.Mlp3Grad(*[ctypes.c_void_p(t.data_ptr()) for t in g_link])
    gs = L.Mlp3Grad(*[ctypes.c_void_p(t.data_ptr()) for t in g_se])
    counts = counts.to(_f32).contiguous()
    lp = torch.empty((F_,), dtype=_f32, device=dev)
    sp = torch.empty((F_, 2), dtype=_f32, device=dev)
    link_out = torch.empty((F_, R, R), dtype=_f32, device=dev) if want_outputs else None
    gt_links = torch.empty((F_, R, R), dtype=_f32, device=dev) if want_outputs else None
    se_logits = torch.empty((F_, 2 * R), dtype=_f32, device=dev) if want_outputs else None
    C = st.pooled_prev.shape[-1]
    dx_link = torch.empty((F_ * R * R, C), dtype=_f32, device=dev) if want_dfeat else None
    dx_se = torch.empty((F_ * 2 * R, C), dtype=_f32, device=dev) if want_dfeat else None
    main = torch.cuda.current_stream(dev)
    # OVERLAP_SE: the start / end head on a side stream under the link head's chain.  Off: measured inside the training step,
    # the fork / join (two stream waits + ~20 record_stream marks per step) costs more than the 0.25 ms chain it hides as soon as
    # the streams have hardware queues of their own (GPU_MAX_HW_QUEUES = 8, needed next to RCCL: 320 vs ~500 frames/s)
    side = side_stream(dev, 2) if OVERLAP_SE else main
    se_bytes = lib.jm_affinity_train_se_workspace_bytes(F_, R, ctypes.byref(se))
    se_ws = torch.empty((max(se_bytes, 16),), dtype=torch.uint8, device=dev)
    if side is not main:
        side.wait_stream(main)
        touched = [st.pooled_prev, st.pooled_next, st.rep_prev, st.rep_next, st.n_pair, st.gt_starts, st.gt_ends, counts, sp, se_ws,
                   *keep_s, *g_se] + ([se_logits] if want_outputs else [])
        for t in touched:
            t.record_stream(side)
    with torch.cuda.stream(side):
        L.check(lib.jm_affinity_train_se_step(
            F_, R, L.dev(st.pooled_prev, _f32, "pooled_prev"), L.dev(st.pooled_next, _f32, "pooled_next"),
            L.dev(st.rep_prev, _i32, "rep_prev"), L.dev(st.rep_next, _i32, "rep_next"), L.dev(st.n_pair, _i32, "n_pair"),
            L.dev(st.gt_starts, _f32, "gt_starts"), L.dev(st.gt_ends, _f32, "gt_ends"), L.dev(counts, _f32, "counts"), float(se_weight),
            ctypes.byref(se), L.dev(se_logits, _f32, "se_logits") if want_outputs else None, L.dev(sp, _f32, "loss_part"),
            ctypes.byref(gs), L.dev(dx_se, _f32, "dx_se") if want_dfeat else None, ctypes.c_void_p(se_ws.data_ptr()), se_bytes,
            L.stream_ptr()), "affinity_train_se_step")
    ws_bytes = lib.jm_affinity_train_link_workspace_bytes(F_, R, ctypes.byref(link))
    ws = torch.empty((max(ws_bytes, 16),), dtype=torch.uint8, device=dev)
    L.check(lib.jm_affinity_train_link_step(
        F_, R, L.dev(st.pooled_prev, _f32, "pooled_prev"), L.dev(st.pooled_next, _f32, "pooled_next"),
        L.dev(st.rep_prev, _i32, "rep_prev"), L.dev(st.rep_next, _i32, "rep_next"), L.dev(st.tids, _f32, "gt_tids"),
        L.dev(counts, _f32, "counts"), float(link_weight), ctypes.byref(link),
        L.dev(link_out, _f32, "link_out") if want_outputs else None, L.dev(gt_links, _f32, "gt_links") if want_outputs else None,
        L.dev(lp, _f32, "loss_part"), ctypes.byref(gl), L.dev(dx_link, _f32, "dx_link") if want_dfeat else None,
        ctypes.c_void_p(ws.data_ptr()), ws_bytes, L.stream_ptr()), "affinity_train_link_step")
    if side is not main:
        main.wait_stream(side)
    outputs = None
    if want_outputs:
        rp, rn = st.rep_prev.bool(), st.rep_next.bool()
        outputs = dict(link=link_out, gt_links=gt_links, valid=rp.unsqueeze(2) & rn.unsqueeze(1), start=se_logits[:, :R],
                       gt_starts=st.gt_starts, start_valid=rn, end=se_logits[:, R:], gt_ends=st.gt_ends, end_valid=rp)
    if not want_dfeat:
        return lp, sp, g_link + g_se, outputs
    # through |p - d|, the masked start / end means and the per-track-id mean pooling, back to the RoI features
    dfeat = torch.empty((2 * F_, R, C), dtype=_f32, device=dev)
    dpooled = torch.empty((2, F_, R, C), dtype=_f32, device=dev)
    L.check(lib.jm_affinity_train_feature_grad(
        F_, R, C, L.dev(st.tids, _f32, "gt_tids"), L.dev(st.pooled_prev, _f32, "pooled_prev"), L.dev(st.pooled_next, _f32, "pooled_next"),
        L.dev(st.rep_prev, _i32, "rep_prev"), L.dev(st.rep_next, _i32, "rep_next"), L.dev(st.n_pair, _i32, "n_pair"),
        L.dev(dx_link, _f32, "dx_link"), L.dev(dx_se, _f32, "dx_se"), ctypes.c_void_p(dpooled.data_ptr()),
        ctypes.c_void_p(dfeat.data_ptr()), L.stream_ptr()), "affinity_train_feature_grad")
    return lp, sp, g_link + g_se, outputs, dfeat


def _loss_from_parts(lp, sp, counts, link_weight, se_weight):
    """the weighted sum of the three L1 means as a device scalar (one launch)"""
    out = torch.empty((), dtype=_f32, device=lp.device)
    counts = counts.to(_f32).contiguous()
    L.check(L.load().jm_affinity_train_loss_value(lp.shape[0], L.dev(lp, _f32, "link_loss_part"), L.dev(sp, _f32, "se_loss_part"),
                                                  L.dev(counts, _f32, "counts"), float(link_weight), float(se_weight),
                                                  ctypes.c_void_p(out.data_ptr()), L.stream_ptr()), "affinity_train_loss_value")
    return out


class _AffinityTrainLoss(torch.autograd.Function):
    """loss = link_weight * sum|link - gt| / counts[0] + se_weight * (sum|sig(start) - gt| / counts[1] + sum|sig(end) - gt| / counts[2])
    with the gradients of the twelve head tensors computed by the kernels in the SAME pass (the backward GEMM chains run
    right behind the forward ones); autograd's backward only scales them by the incoming gradient"""

    @staticmethod
    def forward(ctx, st, counts, link_weight, se_weight, feats, *params):
        want_dfeat = feats is not None and feats.requires_grad
        res = _train_steps(st, counts, params[:6], params[6:], link_weight, se_weight, False, want_dfeat)
        ctx.grads = res[2]
        ctx.dfeat = res[4].view_as(feats) if want_dfeat else None
        return _loss_from_parts(res[0], res[1], counts, link_weight, se_weight)

    @staticmethod
    def backward(ctx, g):
        # every stored gradient times the incoming scalar as ONE multi-tensor launch (13 element-wise launches per step otherwise)
        ts = list(ctx.grads) + ([ctx.dfeat] if ctx.dfeat is not None else [])
        scaled = list(torch._foreach_mul(ts, g)) if g.dim() == 0 else [g * t for t in ts]
        dfeat = scaled.pop() if ctx.dfeat is not None else None
        return (None, None, None, None, dfeat) + tuple(scaled)


def affinity_train_loss(st: AffinityTrainState, link_layer: nn.Module, se_layer: nn.Module, counts: Optional[torch.Tensor] = None,
                        link_weight: float = 1.0, se_weight: float = 1.0) -> torch.Tensor:
    """the re-id loss of train_functions.py:282-329 (L1 forms) of the prepared batch as a differentiable DEVICE scalar w.r.t.
    the parameters of the two heads and — when the `roi_features` the state was built from require grad (joint training; the
    finetune step of tools/train.py:96-107 trains the heads only) — w.r.t. those features.  `counts` overrides the
    denominators (global element counts of a data-parallel step)."""
    params = _head_tensors(link_layer) + _head_tensors(se_layer)
    return _AffinityTrainLoss.apply(st, st.counts if counts is None else counts, float(link_weight), float(se_weight),
                                    st.roi_features, *params)


def training_affinity_hip(roi_features: torch.Tensor, gt_tids: torch.Tensor, link_layer: nn.Module, se_layer: nn.Module):
    """`training_affinity_static`'s dictionary from the HIP kernels (+ 'loss', the local re-id loss), no gradients"""
    st = AffinityTrainState(roi_features, gt_tids)
    lp, sp, _, out = _train_steps(st, st.counts, _head_tensors(link_layer), _head_tensors(se_layer), 1.0, 1.0, True)
    out["loss"] = _loss_from_parts(lp, sp, st.counts, 1.0, 1.0)
    out["counts"] = st.counts
    return out


LAST_GRAD_COLLECTIVES = 0      # gradient collectives the last _finetune_step_hip issued (bench.py reports it)


def _finetune_step_hip(roi_features, gt_tids, link_layer, se_layer, optimizer, world, local=False):
    import torch.distributed as tdist
    st = AffinityTrainState(roi_features, gt_tids)
    counts = st.counts
    collective = jdist.collective_path(world, local)       # a process group exists (a one-rank group included: same path at every size)
    if collective:
        counts = counts.clone()
        tdist.all_reduce(counts, op=tdist.ReduceOp.SUM)
    params = _head_tensors(link_layer) + _head_tensors(se_layer)
    lp, sp, grads, _ = _train_steps(st, counts, params[:6], params[6:], 1.0, 1.0, False)
    optimizer.zero_grad(set_to_none=True)
    for p, g in zip(params, grads):             # the kernels wrote d(loss)/d(param) of THIS rank's sums over the GLOBAL counts
        p.grad = g
    # ONE flat fp32 all-reduce of the twelve head tensors (4.2 MB at 512-wide heads) over RCCL; timed on this stream when the
    # profiler is on (BASELINE.md §3 config 4 asks for the all-reduce time next to frames/s)
    from ..profile import prof
    global LAST_GRAD_COLLECTIVES
    LAST_GRAD_COLLECTIVES = prof.region("grad_allreduce(RCCL)", lambda: jdist.allreduce_gradients(params, world=world, average=False, local=local),
                                        algo_bytes=sum(p.numel() for p in params) * 4)
    optimizer.step()
    total = _loss_from_parts(lp, sp, counts, 1.0, 1.0)
    if collective:
        tdist.all_reduce(total, op=tdist.ReduceOp.SUM)
    return total


def finetune_step_static(roi_features: torch.Tensor, gt_tids: torch.Tensor, link_layer: nn.Module, se_layer: nn.Module,
                         optimizer: torch.optim.Optimizer, world: Optional[int] = None, local: bool = False) -> torch.Tensor:
    """`finetune_step` without host synchronisation: static-shape forward/backward, the three global element counts
    and the gradients all-reduced on the device (RCCL), Adam; returns the whole-batch loss as a DEVICE scalar.
    GPU tensors: the hand-written kernels (csrc/affinity_train.hip); CPU tensors (the gloo tests of the data-parallel
    logic): the plain-torch static form."""
    import torch.distributed as tdist
    if roi_features.is_cuda:
        return _finetune_step_hip(roi_features, gt_tids, link_layer, se_layer, optimizer, world, local)
    optimizer.zero_grad(set_to_none=True)
    out = training_affinity_static(roi_features, gt_tids, link_layer, se_layer)
    with torch.no_grad():
        counts = torch.stack([out["valid"].sum(), out["start_valid"].sum(), out["end_valid"].sum()]).to(torch.float32)
        collective = jdist.collective_path(world, local)
        if collective:
            tdist.all_reduce(counts, op=tdist.ReduceOp.SUM)
    loss, _ = reid_loss_static(out, counts)
    loss.backward()
    params = list(link_layer.parameters()) + list(se_layer.parameters())
    jdist.allreduce_gradients(params, world=world, average=False, local=local)
    optimizer.step()
    total = loss.detach().clone()
    if collective:
        tdist.all_reduce(total, op=tdist.ReduceOp.SUM)
    return total
