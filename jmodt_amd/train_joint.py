"""Joint-mode training step of the detector + affinity heads (tools/train.py:96-107 without cfg.TRAIN.FINETUNE; BASELINE
configs[3]'s second message size: the gradient of ALL 16.7 M parameters = 66.9 MB per step).

The inference engine (detector.py) runs fused, folded, no-grad kernels; this module is the DIFFERENTIABLE composition of the
same parameter containers — what the reference's `PointRCNN.forward` does in TRAIN mode (point_rcnn.py:24-70) with
  * backbone.py:159-196 (`backbone_forward`): four set-abstraction levels through the un-fused operator route of
    ops/pointnet2 (QueryAndGroup + SharedMLP + max-pool; grouping / interpolation / LI-Fusion gather backward = the jm_*_grad
    kernels), the image blocks and the deconvolution pyramid on MIOpen's autograd, the attention fusion modules in plain torch;
  * rpn.py:71-87 heads; ProposalLayer and roipool3d WITHOUT gradient (the reference's are not differentiable either:
    proposal_layer.py / roipool3d_utils.py define no backward), so the backbone learns from the RPN heads and the RCNN from
    its own;
  * rcnn.py:158-202 (`rcnn_forward_train`) and the training affinity of rcnn.py:204-287 on the kernels of
    csrc/affinity_train.hip (ops/affinity_train.affinity_train_loss, differentiable w.r.t. the RoI features).
The reference's loss terms (train_functions.py) are the caller's business and out of scope;
`thin_loss` is the smallest functional that sends a gradient into every parameter, which is what the data-parallel exchange
needs: sums over frames, so that the gradient of a batch is the SUM of its shards' gradients and the all-reduce is a plain
SUM (no count bookkeeping), + the re-id loss with its global-count weighting.
"""
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from . import dist as jdist
from .ops.fusion import feature_gather
from .ops.roipool3d.roipool3d_utils import roipool3d_canonical_gpu
from .profile import prof


def backbone_forward(net, xyz: torch.Tensor, image: torch.Tensor, pts_xy: torch.Tensor) -> torch.Tensor:
    """PointNet2MSG.forward (backbone.py:159-196) on the module containers, differentiable: (B, N, 3), (B, 3, H, W),
    (B, N, 2) -> point features (B, C, N)"""
    l_xyz, l_feats, l_xy, img = [xyz], [None], [pts_xy], [image]
    for i, sa in enumerate(net.SA_modules):
        new_xyz, feats, idx = sa(l_xyz[i], l_feats[i])
        xy_i = torch.gather(l_xy[i], 1, idx.long().unsqueeze(-1).expand(-1, -1, 2))
        im = net.Img_Block[i](img[i])
        feats = net.Fusion_Conv[i](feats, feature_gather(im, xy_i))
        l_xyz.append(new_xyz); l_feats.append(feats); l_xy.append(xy_i); img.append(im)
    for i in range(-1, -(len(net.FP_modules) + 1), -1):
        l_feats[i - 1] = net.FP_modules[i](l_xyz[i - 1], l_xyz[i], l_feats[i - 1], l_feats[i])
    de = torch.cat([net.DeConv[i](img[i + 1]) for i in range(len(net.DeConv))], dim=1)
    fused_img = F.relu(net.image_fusion_bn(net.image_fusion_conv(de)))
    return net.final_fusion_img_point(l_feats[0], feature_gather(fused_img, pts_xy))


def rcnn_forward_train(net, pts_input: torch.Tensor) -> Dict[str, torch.Tensor]:
    """RCNN.forward (rcnn.py:176-202) on pooled RoI points (R, S, 5 + C) -> rcnn_cls (R, 1), rcnn_reg (R, 46), rcnn_feat (R, 512)"""
    k = net.rcnn_input_channel
    xyz = pts_input[..., 0:3].contiguous()
    xyz_feature = net.xyz_up_layer(pts_input[..., 0:k].transpose(1, 2).contiguous().unsqueeze(3))
    rpn_feature = pts_input[..., k:].transpose(1, 2).contiguous().unsqueeze(3)
    merged = net.merge_down_layer(torch.cat((xyz_feature, rpn_feature), dim=1)).squeeze(3)
    l_xyz, l_feat = xyz, merged
    for sa in net.SA_modules:
        l_xyz, l_feat, _ = sa(l_xyz, l_feat)
    return dict(rcnn_cls=net.cls_layer(l_feat).squeeze(-1), rcnn_reg=net.reg_layer(l_feat).squeeze(-1), rcnn_feat=l_feat.squeeze(-1))


def joint_forward(engine, xyz, image, pts_xy, rois_per_frame: int = 64) -> Dict[str, torch.Tensor]:
    """the detector in TRAIN composition (point_rcnn.py:24-70): backbone + RPN heads with gradient, proposals and RoI pooling
    without (the first `rois_per_frame` proposals of every frame stand in for ProposalTargetLayer's sampled RoIs,
    config.py:153), RCNN with gradient on the pooled points"""
    rpn, cfg = engine.rpn, engine.cfg
    feats = backbone_forward(rpn.backbone_net, xyz, image, pts_xy)
    rpn_cls = rpn.rpn_cls_layer(feats).transpose(1, 2).contiguous()          # (B, N, 1)
    rpn_reg = rpn.rpn_reg_layer(feats).transpose(1, 2).contiguous()          # (B, N, C)
    with torch.no_grad():
        det = dict(rpn_cls=rpn_cls.detach(), rpn_reg=rpn_reg.detach(), backbone_xyz=xyz, backbone_features=feats.detach())
        rois, _ = engine.proposals(det)
        rois = rois[:, :rois_per_frame].contiguous()
        pf = engine.pts_feature(det)
        pooled, _ = roipool3d_canonical_gpu(xyz, pf, rois, cfg.pool_extra_width, cfg.rcnn_num_points)
        pts_input = pooled.view(-1, cfg.rcnn_num_points, pooled.shape[-1])
    out = rcnn_forward_train(engine.rcnn_net, pts_input)
    out.update(rpn_cls=rpn_cls, rpn_reg=rpn_reg, backbone_features=feats, rois=rois)
    return out


def thin_loss(engine, out: Dict[str, torch.Tensor], gt_tids: torch.Tensor, counts: Optional[torch.Tensor] = None) -> torch.Tensor:
    """a functional of every head output (sums over frames: shard gradients ADD) + the re-id loss of rcnn.py:204-287 /
    train_functions.py:282-329 on the RoI features with the (global) element counts"""
    from .ops.affinity_train import AffinityTrainState, affinity_train_loss
    B = gt_tids.shape[0]
    feats = out["rcnn_feat"].view(B, -1, out["rcnn_feat"].shape[-1])
    st = AffinityTrainState(feats, gt_tids)
    reid = affinity_train_loss(st, engine.rcnn_net.link_layer, engine.rcnn_net.se_layer, counts=counts)
    n = float(out["rpn_cls"].shape[1])
    return (out["rpn_cls"].sum() + out["rpn_reg"].sum()) / n + out["rcnn_cls"].sum() + out["rcnn_reg"].sum() + reid


CONV_FIND = False    # see joint_step (bench.py switches it on: seconds of MIOpen search per convolution shape at first use)

_bn_lists = {}       # id(engine) -> (registration epoch, [BatchNorm modules], [parameters])


def _engine_lists(engine):
    """(BatchNorm modules, parameters) of the engine, re-collected only after a registration anywhere in the process: three module
    walks per step are 3 ms of host time on a step the host must keep ahead of"""
    from ._registry import EPOCH
    hit = _bn_lists.get(id(engine))
    if hit is None or hit[0] != EPOCH[0] or hit[3]() is not engine:
        import weakref
        hit = _bn_lists[id(engine)] = (EPOCH[0], [m for m in engine.modules() if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d))],
                                       list(engine.parameters()), weakref.ref(engine))
    return hit[1], hit[2]


def frozen_bn(engine) -> bool:
    """is every BatchNorm of the detector in eval mode (running statistics: cfg.RPN.FIXED-style, point_rcnn.py:29-30)?"""
    return not any(m.training for m in _engine_lists(engine)[0])


def freeze_bn(engine) -> None:
    """train mode for everything but the BatchNorms (frozen statistics, trainable gamma / beta)"""
    engine.train()
    for m in engine.modules():
        if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
            m.eval()


def prepare_rows(engine) -> None:
    """once, before the optimizer is built: BatchNorms frozen, the image blocks' 3x3 kernels in channels-last memory (what
    MIOpen's fp32 kernels take without a transposition pass; the folded copies and the gradients then have that layout too)"""
    freeze_bn(engine)
    for blk in engine.rpn.backbone_net.Img_Block:
        blk.to(memory_format=torch.channels_last)


def joint_step(engine, xyz, image, pts_xy, gt_tids, optimizer, world: Optional[int] = None, rois_per_frame: int = 64,
               bucket_bytes: int = 64 << 20, route: str = "auto", next_xyz=None, local: bool = False) -> torch.Tensor:
    """one data-parallel joint-mode step on this rank's frames: differentiable forward, thin loss, backward through the whole
    detector, the gradient of every parameter all-reduced in 64 MiB buckets (66.9 MB at the reference widths: ONE
    collective), optimizer step.  Returns the local loss (device scalar, detached).

    route: "rows" = forward and backward on the hand-written row kernels (train_rows.py; BatchNorm frozen: eval-mode statistics),
    "operators" = the un-fused operator route above (torch autograd over (B, C, npoint, nsample) tensors; any BatchNorm mode),
    "auto" = rows whenever the BatchNorms are frozen (the faster of the two: DESIGN.md section 6; the HIP-graph form of the rows route
    measured slower than eager and lives under tools/quarantine/).  next_xyz: the next batch's cloud — its FPS pyramid / neighbour search
    starts on the side stream under this step (rows route).  world / local: see dist.group_world."""
    import torch.distributed as tdist
    from .ops.affinity_train import AffinityTrainState
    params = [p for p in _engine_lists(engine)[1] if p.requires_grad]
    if CONV_FIND and xyz.is_cuda and not torch.backends.cudnn.benchmark:
        # MIOpen's find mode for the image branch's convolutions, forward AND backward (a process-wide flag: the autograd engine's
        # thread sees it too): the kernels it measures at first use instead of the immediate-mode heuristic's
        torch.backends.cudnn.benchmark = True
    optimizer.zero_grad(set_to_none=True)
    if route == "auto":
        route = "rows" if frozen_bn(engine) and xyz.is_cuda else "operators"
    if route not in ("rows", "operators"):
        raise ValueError(f"joint_step: route {route!r} (rows | operators | auto)")
    if route == "rows":
        if not frozen_bn(engine):
            raise RuntimeError("joint_step(route='rows') folds the BatchNorms: call train_joint.freeze_bn(engine) (or engine.eval()) first")
        loss = prof.region("joint_forward+backward(span)", lambda: _rows_forward_backward(engine, xyz, image, pts_xy, gt_tids, world, local,
                                                                                        rois_per_frame, next_xyz))
    else:
        out = prof.region("joint_forward(span)", lambda: joint_forward(engine, xyz, image, pts_xy, rois_per_frame))
        counts = None
        if jdist.collective_path(world, local):      # the re-id means run over the GLOBAL element counts (as in the finetune step)
            B = gt_tids.shape[0]
            with torch.no_grad():
                counts = AffinityTrainState(out["rcnn_feat"].detach().view(B, -1, out["rcnn_feat"].shape[-1]), gt_tids).counts.clone()
            tdist.all_reduce(counts, op=tdist.ReduceOp.SUM)
        loss = thin_loss(engine, out, gt_tids, counts)
        prof.region("joint_backward(span)", lambda: loss.backward())
    global LAST_GRAD_COLLECTIVES
    LAST_GRAD_COLLECTIVES = prof.region("grad_allreduce(RCCL)", lambda: jdist.allreduce_gradients(params, world=world, bucket_bytes=bucket_bytes,
                                                                                                  average=False, local=local),
                                        algo_bytes=sum(p.numel() for p in params) * 4)
    optimizer.step()
    return loss.detach()


def _rows_forward_backward(engine, xyz, image, pts_xy, gt_tids, world, local, rois_per_frame, next_xyz):
    """forward and backward of the rows route, issued in the order that keeps three streams busy: backbone + RPN heads forward
    (main / image streams) -> proposals + RoI pooling (no gradient) -> the BACKWARD of the RPN part of the loss (main / image
    streams: the loss is a sum, its parts back-propagate independently) -> RCNN forward, its loss and backward on the RCNN stream,
    under the backbone's backward.  Same gradients as one backward over the summed loss: the two graphs share no node (the
    pooled RoI points carry no gradient)."""
    import torch.distributed as tdist
    from .ops.affinity_train import AffinityTrainState, affinity_train_loss
    from .train_rows import BnFold, pooled_rois, rcnn_branch_rows, rpn_forward_rows
    pyr = engine._take_prefetched(xyz)
    if next_xyz is not None:
        engine.prefetch(next_xyz, None)
    fold = BnFold(engine.rpn)                   # (the RCNN branch folds its own BatchNorms, on its stream, into its own graph)
    out = rpn_forward_rows(engine, xyz, image, pts_xy, fold, pyr)
    rois, pts_input, count = pooled_rois(engine, xyz, out, rois_per_frame)
    pooled_ev = torch.cuda.Event()
    pooled_ev.record()
    n = float(out["rpn_cls"].shape[1])
    rpn_loss = (out["rpn_cls"].sum() + out["rpn_reg"].sum()) / n
    main = torch.cuda.current_stream(xyz.device)
    B = gt_tids.shape[0]

    def rcnn_half():
        rc = rcnn_branch_rows(engine, pts_input, count, ready=pooled_ev)
        side = rc.pop("_stream")
        with torch.cuda.stream(side):
            feats = rc["rcnn_feat"].view(B, -1, rc["rcnn_feat"].shape[-1])
            st = AffinityTrainState(feats, gt_tids)
            counts = None
            if jdist.collective_path(world, local):      # the re-id means run over the GLOBAL element counts (as in the finetune step)
                counts = st.counts.clone()
                tdist.all_reduce(counts, op=tdist.ReduceOp.SUM)
            reid = affinity_train_loss(st, engine.rcnn_net.link_layer, engine.rcnn_net.se_layer, counts=counts)
            rcnn_loss = rc["rcnn_cls"].sum() + rc["rcnn_reg"].sum() + reid
            rcnn_loss.backward()
        return side, rcnn_loss.detach()

    # (which half the host issues first does not matter — 192.1 / 192.7 frames/s RPN half first, 192.3 / 192.3 RCNN half first;
    # issuing the RPN half from a helper thread next to the RCNN half's forward: 191.2 / 190.7 against 191.2 / 190.4 — the step is
    # the host's 17.3 ms of interpreter-bound enqueue plus a 3.3 ms tail either way)
    rpn_loss.backward()
    side, rcnn_loss = rcnn_half()
    if side is not main:
        gt_tids.record_stream(side)
        main.wait_stream(side)               # every gradient is in place before the all-reduce / optimizer on the main stream
        rcnn_loss.record_stream(main)
    # (the sum on the MAIN stream, behind the join: the RCNN stream only ever waited for the pooling event, not for rpn_loss)
    total = rcnn_loss + rpn_loss.detach()
    img = _image_stream(engine, xyz.device)
    if img is not None:
        main.wait_stream(img)
    return total


_rcnn_lists = {}      # id(engine) -> (registration epoch, weakref, RCNN parameters, RPN parameters)


def _rcnn_split(engine):
    """(RCNN parameters, RPN parameters), re-collected only after a module / parameter registration anywhere in the process (two
    module walks per step are ~0.5 ms of host time on a step whose host side is 5.6 ms)"""
    from ._registry import EPOCH
    hit = _rcnn_lists.get(id(engine))
    if hit is None or hit[0] != EPOCH[0] or hit[1]() is not engine:
        import weakref
        hit = _rcnn_lists[id(engine)] = (EPOCH[0], weakref.ref(engine), list(engine.rcnn_net.parameters()), list(engine.rpn.parameters()))
    return hit[2], hit[3]


def rcnn_parameters(engine):
    """what tools/train.py:104 ends up updating with the shipped configuration (config.py:57 RPN.FIXED = True, FINETUNE off): the
    optimizer holds model.parameters(), the RPN runs under no_grad (point_rcnn.py:28-31) — only the RCNN's tensors (set abstraction,
    heads, link / start-end heads) ever receive a gradient"""
    return [p for p in _rcnn_split(engine)[0] if p.requires_grad]


def prepare_rcnn(engine) -> None:
    """once, before the optimizer is built: the RPN in eval mode without gradient (point_rcnn.py:29-30 `self.rpn.eval()` under
    set_grad_enabled(False)), the RCNN in train mode with gradient"""
    engine.eval()
    engine.rcnn_net.train()
    for p in engine.rpn.parameters():
        p.requires_grad_(False)
    for p in engine.rcnn_net.parameters():
        p.requires_grad_(True)
    bad = [n for n, m in engine.rcnn_net.named_modules() if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d))]
    if bad:      # config.py:107 RCNN.USE_BN = False; a train-mode BatchNorm (batch statistics) has no rows form
        raise NotImplementedError(f"rcnn_step: the RCNN carries BatchNorm layers ({bad[:3]} ...); the rows route folds eval-mode statistics only")


_AHEAD_SLOT = 7          # side-stream slot of the frozen half that runs one batch ahead (pyramid.side_stream: 0-6 are the engine's)


def _frozen_half(engine, xyz, image, pts_xy, rois_per_frame, next_xyz, next_image):
    """what the RPN-fixed step computes WITHOUT gradient (point_rcnn.py:28-47 under cfg.RPN.FIXED): the fused engine's RPN forward,
    the proposal layer, the per-point [mask, depth, features] rows and the RoI pooling of the first `rois_per_frame` proposals"""
    cfg = engine.cfg
    with torch.no_grad():
        rpn_out = engine.rpn_forward(xyz, image, pts_xy, next_xyz, next_image)
        rois, _ = engine.proposals(rpn_out)
        rois = rois[:, :rois_per_frame].contiguous()
        pf = engine.pts_feature(rpn_out)
        pooled, _, count = roipool3d_canonical_gpu(xyz, pf, rois, cfg.pool_extra_width, cfg.rcnn_num_points, return_count=True)
    return dict(rois=rois, pts_input=pooled.view(-1, cfg.rcnn_num_points, pooled.shape[-1]), count=count.view(-1),
                rpn_cls=rpn_out["rpn_cls"], rpn_reg=rpn_out["rpn_reg"])


def _batch_key(xyz, image, pts_xy, rois_per_frame):
    return tuple((id(t), t.data_ptr(), t._version) for t in (xyz, image, pts_xy)) + (int(rois_per_frame),)


def drop_ahead(engine) -> None:
    """forget a frozen half computed ahead (after the RPN's weights were changed by hand: rcnn_step itself never changes them)"""
    engine.__dict__.pop("_rcnn_ahead", None)


def rcnn_forward_backward(engine, xyz, image, pts_xy, gt_tids, world=None, local=False, rois_per_frame: int = 64,
                          next_xyz=None, next_image=None, next_batch=None):
    """forward + loss + backward of the RPN-fixed step; returns (local loss, outputs).  The frozen half is the fused inference
    engine as it stands (detector.py: rpn_forward, proposals, pts_feature: no autograd graph, the next batch's FPS pyramid and image
    pyramid started on their side streams under this batch's RCNN); the trainable half is train_rows.rcnn_forward_rows on the
    CURRENT stream (forward and backward = csrc/rows_*.hip), the re-id loss on csrc/affinity_train.hip.

    next_batch = (xyz, image, pts_xy) of the NEXT step: its whole frozen half is then issued NOW, on a stream of its own, and runs
    UNDER this batch's RCNN forward / backward / optimizer — the RPN is frozen, so nothing of it depends on the update this step
    makes (the reference evaluates it under no_grad for the same reason, point_rcnn.py:28-31).  The next call must pass the same
    tensor objects (unchanged); anything else is computed in line.  With next_batch, next_xyz / next_image announce the batch AFTER
    it (its FPS pyramid / image pyramid start under the next batch's frozen half)."""
    import torch.distributed as tdist
    from .ops.affinity_train import AffinityTrainState, affinity_train_loss
    from .ops.pointnet2.pyramid import side_stream
    from .train_rows import BnFold, rcnn_forward_rows
    dev = xyz.device
    main = torch.cuda.current_stream(dev)
    ahead = engine.__dict__.pop("_rcnn_ahead", None)
    if ahead is not None and ahead["key"] == _batch_key(xyz, image, pts_xy, rois_per_frame):
        main.wait_event(ahead["event"])
        fh = ahead["out"]
        for t in fh.values():                   # allocated on the other stream, consumed (and later freed) under this one
            t.record_stream(main)
    else:
        if ahead is not None:                   # another batch than the announced one: its frozen half is of no use, but it may still
            main.wait_event(ahead["event"])     # be running on its stream — the engine's cached scratch buffers serve one call at a time
        fh = _frozen_half(engine, xyz, image, pts_xy, rois_per_frame, None if next_batch is not None else next_xyz,
                          None if next_batch is not None else next_image)
    if next_batch is not None and engine.overlap:
        nx, ni, npxy = next_batch
        side = side_stream(dev, _AHEAD_SLOT)
        side.wait_stream(main)                  # the batch is resident; what this stream's last round allocated is consumed
        with torch.cuda.stream(side):
            out_next = _frozen_half(engine, nx, ni, npxy, rois_per_frame, next_xyz, next_image)
            ev = torch.cuda.Event()
            ev.record(side)
        for t in (nx, ni, npxy):
            t.record_stream(side)
        engine._rcnn_ahead = dict(key=_batch_key(nx, ni, npxy, rois_per_frame), event=ev, out=out_next)
    rois, pts_input, count = fh["rois"], fh["pts_input"], fh["count"]
    fold = BnFold(engine.rcnn_net)                      # (config.py:107 ships none; eval-mode ones would fold here)
    out = prof.region("rcnn_forward(span)", lambda: rcnn_forward_rows(engine, pts_input, fold, count))
    B = gt_tids.shape[0]
    feats = out["rcnn_feat"].view(B, -1, out["rcnn_feat"].shape[-1])
    st = AffinityTrainState(feats, gt_tids)
    counts = None
    if jdist.collective_path(world, local):             # the re-id means run over the GLOBAL element counts (as in the finetune step)
        counts = st.counts.clone()
        tdist.all_reduce(counts, op=tdist.ReduceOp.SUM)
    reid = affinity_train_loss(st, engine.rcnn_net.link_layer, engine.rcnn_net.se_layer, counts=counts)
    loss = out["rcnn_cls"].sum() + out["rcnn_reg"].sum() + reid
    prof.region("rcnn_backward(span)", lambda: loss.backward())
    out.update(rois=rois, rpn_cls=fh["rpn_cls"], rpn_reg=fh["rpn_reg"])
    return loss.detach(), out


def rcnn_step(engine, xyz, image, pts_xy, gt_tids, optimizer, world: Optional[int] = None, rois_per_frame: int = 64,
              bucket_bytes: int = 64 << 20, next_xyz=None, next_image=None, local: bool = False, next_batch=None) -> torch.Tensor:
    """one data-parallel step of the reference's DEFAULT training mode — tools/train.py:86-107 with the shipped config.py:57
    (`RPN.FIXED = True`) and FINETUNE off: the RPN is evaluated without gradient (point_rcnn.py:28-31), the RCNN and the re-id heads
    train (rcnn.py:158-287).  This rank's frames: frozen fused detector forward -> proposals -> RoI pooling -> RCNN forward / loss /
    backward on the row kernels -> ONE bucketed all-reduce of the RCNN's gradients (head sums + re-id loss with global counts: shard
    gradients ADD, dist.py) -> optimizer step.  Returns the local loss (device scalar).  Call prepare_rcnn(engine) once before
    building the optimizer.  next_batch: see rcnn_forward_backward (the next step's frozen half under this step's RCNN)."""
    params = rcnn_parameters(engine)
    if engine.rpn.training or any(p.requires_grad for p in _rcnn_split(engine)[1]):
        raise RuntimeError("rcnn_step runs the RPN frozen: call train_joint.prepare_rcnn(engine) first (point_rcnn.py:28-31)")
    optimizer.zero_grad(set_to_none=True)
    loss, _ = rcnn_forward_backward(engine, xyz, image, pts_xy, gt_tids, world, local, rois_per_frame, next_xyz, next_image, next_batch)
    global LAST_GRAD_COLLECTIVES
    LAST_GRAD_COLLECTIVES = prof.region("grad_allreduce(RCCL)", lambda: jdist.allreduce_gradients(params, world=world, bucket_bytes=bucket_bytes,
                                                                                                  average=False, local=local),
                                        algo_bytes=sum(p.numel() for p in params) * 4)
    optimizer.step()
    return loss


def _image_stream(engine, device):
    from .ops.pointnet2.pyramid import side_stream
    return side_stream(device, 1) if engine.overlap else None


LAST_GRAD_COLLECTIVES = 0
