"""Chained CPU oracle of the composed detect + affinity forward (TEST INFRASTRUCTURE ONLY — imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; nothing under jmodt_amd/ may import it).

A literal, un-fused restatement of the reference's inference forward, in the reference's op order:
  PointNet2MSG.forward           jmodt/detection/modeling/backbone.py:159-196 (+ BasicBlock :16-32, IALayer :35-63,
                                 AttentionFusion :66-81, feature_gather :79-89)
  _PointnetSAModuleBase.forward  jmodt/ops/pointnet2/pointnet2_modules.py:20-63; QueryAndGroup / GroupAll
                                 pointnet2_utils.py:231-290; PointnetFPModule.forward :135-164
  RPN.forward                    jmodt/detection/modeling/rpn.py:71-87
  PointRCNN.forward              jmodt/detection/modeling/point_rcnn.py:24-70 (EVAL)
  ProposalLayer / ProposalTargetLayer (EVAL)  layers/proposal_layer.py:16-117, proposal_target_layer.py:16-34,99-115
  RCNN.forward (EVAL)            jmodt/detection/modeling/rcnn.py:158-202,288-289
  detection post-processing      tools/eval.py:108-193
  inference affinity             jmodt/tracking/tracker.py:81-112

The jmodt.ops operators come from oracle/oracle.py (the C restatement of the reference's CUDA kernels); everything
the reference does with PyTorch (convolutions, BatchNorm, Linear, grid_sample, softmax) is done with the same
PyTorch CPU operators here, un-fused and un-folded, in `dtype` (float32 = the reference's arithmetic and the
cpu_baseline leg; float64 = a tighter checker for the GPU path).  Weights come from a state_dict with the
reference's parameter names.

PINNED against the reference itself: tests/golden/forward_ref.npz holds the outputs of the reference's own
PointRCNN.forward (TEST mode) executed in the authoring container over oracle.py's extension entry points
(tests/golden/make_golden_forward.py); tests/test_oracle_cpu.py::test_chained_oracle_matches_the_references_complete_forward
checks backbone + RPN heads, proposal layer, RoI pooling + canonical transform and the RCNN of this file against it, and
glue_ref.npz does the same for sa_module / fp_module alone.  The detection post-processing and the inference affinity are
pinned by decode_ref.npz / affinity_ref.npz.  What no fixture can pin here are the CUDA kernels inside oracle.py's
restatements (the reference has no CPU code for them): "parity unpinned" for those, as oracle/jmodt_oracle.c says.
"""
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import oracle as orc

BN_EPS = 1e-5


def _np(t):
    return t.detach().cpu().numpy()


class Chain:
    def __init__(self, state_dict: Dict[str, torch.Tensor], cfg, dtype=torch.float32):
        self.sd = {k: (v.detach().cpu().to(dtype) if v.is_floating_point() else v.detach().cpu()) for k, v in state_dict.items()}
        self.cfg = cfg
        self.dtype = dtype

    # ---- layer helpers (eval mode) ---------------------------------------------------------------
    def _bn(self, x, prefix):
        shape = [1, -1] + [1] * (x.dim() - 2)
        w, b = self.sd[prefix + ".weight"], self.sd[prefix + ".bias"]
        m, v = self.sd[prefix + ".running_mean"], self.sd[prefix + ".running_var"]
        return (x - m.view(shape)) / torch.sqrt(v.view(shape) + BN_EPS) * w.view(shape) + b.view(shape)

    def _conv_unit(self, x, prefix, relu=True):
        """pytorch_utils Conv1d / Conv2d unit: conv [+ bn.bn] [+ relu]"""
        w = self.sd[prefix + ".conv.weight"]
        b = self.sd.get(prefix + ".conv.bias")
        x = F.conv2d(x, w, b) if w.dim() == 4 else F.conv1d(x, w, b)
        if prefix + ".bn.bn.weight" in self.sd:
            x = self._bn(x, prefix + ".bn.bn")
        return torch.relu(x) if relu else x

    def _shared_mlp(self, x, prefix):
        i = 0
        while f"{prefix}.layer{i}.conv.weight" in self.sd:
            x = self._conv_unit(x, f"{prefix}.layer{i}")
            i += 1
        return x

    def _head(self, x, prefix):
        """Sequential of Conv1d units with a Dropout at index 1 (identity in eval); last unit has no activation"""
        idxs = sorted({int(k[len(prefix) + 1:].split(".")[0]) for k in self.sd if k.startswith(prefix + ".") and ".conv.weight" in k})
        for n, i in enumerate(idxs):
            x = self._conv_unit(x, f"{prefix}.{i}", relu=n + 1 < len(idxs))
        return x

    # ---- pointnet2 modules ------------------------------------------------------------------------
    def _group(self, feats, nb):
        """feats (B, C, N) torch, nb (B, m, ns) int numpy -> (B, C, m, ns)   (grouping_operation)"""
        idx = torch.from_numpy(nb.astype(np.int64))
        B, C, _ = feats.shape
        m, ns = idx.shape[1:]
        return torch.gather(feats, 2, idx.view(B, 1, m * ns).expand(-1, C, -1)).view(B, C, m, ns)

    def sa_module(self, prefix, xyz, feats, npoint, radii, nsamples):
        """xyz (B, N, 3) float32 numpy, feats (B, C, N) torch or None -> new_xyz numpy, new feats torch, fps idx"""
        B = xyz.shape[0]
        if npoint is not None:
            idx = orc.furthest_point_sample(xyz, npoint)
            new_xyz = np.take_along_axis(xyz, idx[..., None].astype(np.int64), axis=1)
        else:
            idx, new_xyz = None, None
        outs = []
        xyz_t = torch.from_numpy(xyz).to(self.dtype).transpose(1, 2).contiguous()              # (B, 3, N)
        for k, (r, ns) in enumerate(zip(radii, nsamples)):
            if npoint is not None:
                nb = orc.ball_query(r, ns, xyz, new_xyz)
                g_xyz = self._group(xyz_t, nb) - torch.from_numpy(new_xyz).to(self.dtype).transpose(1, 2).unsqueeze(-1)
                grouped = g_xyz if feats is None else torch.cat([g_xyz, self._group(feats, nb)], dim=1)
            else:   # GroupAll (pointnet2_utils.py:267-290)
                g_xyz = xyz_t.unsqueeze(2)
                grouped = g_xyz if feats is None else torch.cat([g_xyz, feats.unsqueeze(2)], dim=1)
            y = self._shared_mlp(grouped, f"{prefix}.mlps.{k}")
            outs.append(y.max(dim=3)[0])
        return new_xyz, torch.cat(outs, dim=1), idx

    def fp_module(self, prefix, unknown, known, unknown_feats, known_feats):
        d2, idx = orc.three_nn(unknown, known)
        dist = torch.sqrt(torch.from_numpy(d2).to(self.dtype))
        inv = 1.0 / (dist + 1e-8)
        w = inv / inv.sum(dim=2, keepdim=True)                                                  # (B, n, 3)
        B, C, m = known_feats.shape
        n = idx.shape[1]
        gi = torch.from_numpy(idx.astype(np.int64)).view(B, 1, n * 3).expand(-1, C, -1)
        taps = torch.gather(known_feats, 2, gi).view(B, C, n, 3)
        interp = (taps * w.unsqueeze(1)).sum(dim=3)
        x = interp if unknown_feats is None else torch.cat([interp, unknown_feats], dim=1)
        return self._shared_mlp(x.unsqueeze(-1), f"{prefix}.mlp").squeeze(-1)

    # ---- LI-Fusion -------------------------------------------------------------------------------------
    def _attention_fusion(self, prefix, point_feats, img_feats):
        sd = self.sd
        B = img_feats.shape[0]
        ic, pc = img_feats.shape[1], point_feats.shape[1]
        img_f = img_feats.transpose(1, 2).contiguous().view(-1, ic)
        pt_f = point_feats.transpose(1, 2).contiguous().view(-1, pc)
        ia = prefix + ".IA_Layer"
        ri = F.linear(img_f, sd[ia + ".fc1.weight"], sd[ia + ".fc1.bias"])
        rp = F.linear(pt_f, sd[ia + ".fc2.weight"], sd[ia + ".fc2.bias"])
        att = torch.sigmoid(F.linear(torch.tanh(ri + rp), sd[ia + ".fc3.weight"], sd[ia + ".fc3.bias"])).squeeze(1).view(B, 1, -1)
        img_new = torch.relu(self._bn(F.conv1d(img_feats, sd[ia + ".conv1.0.weight"], sd[ia + ".conv1.0.bias"]), ia + ".conv1.1"))
        fused = torch.cat([point_feats, img_new * att], dim=1)
        return torch.relu(self._bn(F.conv1d(fused, sd[prefix + ".conv1.weight"], sd[prefix + ".conv1.bias"]), prefix + ".bn1"))

    def _gather(self, fmap, xy):
        return F.grid_sample(fmap, xy.unsqueeze(1), align_corners=True).squeeze(2)

    # ---- stages ------------------------------------------------------------------------------------------
    def backbone(self, xyz: np.ndarray, image: np.ndarray, pts_xy: np.ndarray):
        cfg, sd, dt = self.cfg, self.sd, self.dtype
        bb = "rpn.backbone_net"
        l_xyz, l_feats = [xyz], [None]
        l_xy = [torch.from_numpy(pts_xy).to(dt)]
        img = [torch.from_numpy(image).to(dt)]
        fps_idx = []
        for i, npoint in enumerate(cfg.sa_npoints):
            new_xyz, feats, idx = self.sa_module(f"{bb}.SA_modules.{i}", l_xyz[i], l_feats[i], npoint, cfg.sa_radius[i], cfg.sa_nsample[i])
            fps_idx.append(idx)
            gi = torch.from_numpy(idx.astype(np.int64)).unsqueeze(-1).repeat(1, 1, 2)
            xy_i = torch.gather(l_xy[i], 1, gi)
            blk = f"{bb}.Img_Block.{i}"
            x = F.conv2d(img[i], sd[blk + ".conv1.weight"], None, stride=1, padding=1)
            x = torch.relu(self._bn(x, blk + ".bn1"))
            x = F.conv2d(x, sd[blk + ".conv2.weight"], None, stride=2, padding=1)
            feats = self._attention_fusion(f"{bb}.Fusion_Conv.{i}", feats, self._gather(x, xy_i))
            l_xy.append(xy_i); img.append(x); l_xyz.append(new_xyz); l_feats.append(feats)
        nfp = len(cfg.fp_mlps)
        for i in range(-1, -(nfp + 1), -1):
            l_feats[i - 1] = self.fp_module(f"{bb}.FP_modules.{nfp + i}", l_xyz[i - 1], l_xyz[i], l_feats[i - 1], l_feats[i])
        de = []
        for i, k in enumerate(cfg.deconv_kernels):
            de.append(F.conv_transpose2d(img[i + 1], sd[f"{bb}.DeConv.{i}.weight"], sd[f"{bb}.DeConv.{i}.bias"], stride=k))
        cat = torch.cat(de, dim=1)
        fused_map = torch.relu(self._bn(F.conv2d(cat, sd[f"{bb}.image_fusion_conv.weight"], sd[f"{bb}.image_fusion_conv.bias"]),
                                        f"{bb}.image_fusion_bn"))
        out = self._attention_fusion(f"{bb}.final_fusion_img_point", l_feats[0], self._gather(fused_map, l_xy[0]))
        self.last = dict(fps_idx=fps_idx, l_xyz=l_xyz, img=img, fused_map=fused_map)
        return out

    def rpn(self, xyz, image, pts_xy):
        feats = self.backbone(xyz, image, pts_xy)
        rpn_cls = self._head(feats, "rpn.rpn_cls_layer").transpose(1, 2).contiguous()
        rpn_reg = self._head(feats, "rpn.rpn_reg_layer").transpose(1, 2).contiguous()
        return dict(rpn_cls=rpn_cls, rpn_reg=rpn_reg, backbone_xyz=xyz, backbone_features=feats)

    def proposals(self, rpn_cls: np.ndarray, rpn_reg: np.ndarray, xyz: np.ndarray):
        cfg = self.cfg
        props = orc.decode_rpn_proposals(xyz, rpn_reg, cfg.rpn_loc_scope, cfg.rpn_loc_bin_size, cfg.rpn_num_head_bin,
                                         cfg.mean_size, True)
        return orc.proposal_select(rpn_cls[:, :, 0], props, cfg.rpn_pre_nms_top_n, cfg.rpn_post_nms_top_n,
                                   cfg.rpn_nms_thresh, cfg.rpn_nms_type, True)

    def roi_pool(self, xyz: np.ndarray, rpn_cls: np.ndarray, feats: np.ndarray, rois: np.ndarray):
        """-> pts_input (B*M, S, 5 + C) float32 numpy"""
        cfg = self.cfg
        score = 1.0 / (1.0 + np.exp(-rpn_cls[:, :, 0].astype(np.float64)))
        seg_mask = (score.astype(np.float32) > np.float32(cfg.rpn_score_thresh)).astype(np.float32)
        depth = (np.sqrt((xyz.astype(np.float32) ** 2).sum(-1, dtype=np.float32)) / np.float32(70.0) - np.float32(0.5)).astype(np.float32)
        pts_feature = np.concatenate([seg_mask[..., None], depth[..., None], feats.transpose(0, 2, 1)], axis=2).astype(np.float32)
        pooled, _ = orc.roipool3d_canonical(xyz, pts_feature, rois, cfg.pool_extra_width, cfg.rcnn_num_points)
        return pooled.reshape(-1, pooled.shape[2], pooled.shape[3]), pts_feature

    def rcnn(self, pts_input: np.ndarray):
        cfg, dt = self.cfg, self.dtype
        p = torch.from_numpy(pts_input).to(dt)
        k = 5
        xyz = np.ascontiguousarray(pts_input[..., 0:3])
        xyz_input = p[..., 0:k].transpose(1, 2).contiguous().unsqueeze(3)
        xyz_feature = self._shared_mlp(xyz_input, "rcnn_net.xyz_up_layer")
        rpn_feature = p[..., k:].transpose(1, 2).contiguous().unsqueeze(3)
        merged = self._shared_mlp(torch.cat((xyz_feature, rpn_feature), dim=1), "rcnn_net.merge_down_layer")
        l_xyz, l_feats = xyz, merged.squeeze(3)
        for i, npoint in enumerate(cfg.rcnn_sa_npoints):
            l_xyz, l_feats, _ = self.sa_module(f"rcnn_net.SA_modules.{i}", l_xyz, l_feats, npoint if npoint != -1 else None,
                                               [cfg.rcnn_sa_radius[i]], [cfg.rcnn_sa_nsample[i]])
        rcnn_cls = self._head(l_feats, "rcnn_net.cls_layer").squeeze(-1)
        rcnn_reg = self._head(l_feats, "rcnn_net.reg_layer").squeeze(-1)
        return dict(rcnn_cls=rcnn_cls, rcnn_reg=rcnn_reg, rcnn_feat=l_feats)

    def affinity(self, pred_feats: torch.Tensor, det_feats: torch.Tensor):
        """tracker.py:81-112 with torch CPU ops: (link (P, D), start logits (D), end logits (P))"""
        P, D = pred_feats.shape[0], det_feats.shape[0]
        cor = torch.abs(pred_feats.unsqueeze(1).repeat(1, D, 1) - det_feats.unsqueeze(0).repeat(P, 1, 1))
        s = self._head(cor.view(P * D, -1, 1), "rcnn_net.link_layer").view(P, D)
        link = (torch.softmax(s, dim=1) + torch.softmax(s, dim=0)) / 2
        start = self._head(cor.mean(dim=0).unsqueeze(-1), "rcnn_net.se_layer").flatten()
        end = self._head(cor.mean(dim=1).unsqueeze(-1), "rcnn_net.se_layer").flatten()
        return link, start, end

    # ---- whole path (free running: every stage consumes the oracle's own previous stage) ---------------
    def forward(self, xyz: np.ndarray, image: np.ndarray, pts_xy: np.ndarray):
        import time
        cfg = self.cfg
        t = [time.perf_counter()]
        r = self.rpn(xyz, image, pts_xy)
        t.append(time.perf_counter())
        rpn_cls, rpn_reg = _np(r["rpn_cls"]).astype(np.float32), _np(r["rpn_reg"]).astype(np.float32)
        rois, roi_scores = self.proposals(rpn_cls, rpn_reg, xyz)
        t.append(time.perf_counter())
        pts_input, _ = self.roi_pool(xyz, rpn_cls, _np(r["backbone_features"]).astype(np.float32), rois)
        t.append(time.perf_counter())
        out = self.rcnn(pts_input)
        t.append(time.perf_counter())
        B, M = rois.shape[:2]
        boxes = orc.decode_rcnn_boxes(rois.reshape(-1, 7), _np(out["rcnn_reg"]).astype(np.float32), cfg.rcnn_loc_scope,
                                      cfg.rcnn_loc_bin_size, cfg.rcnn_num_head_bin, cfg.mean_size).reshape(B, M, 7)
        raw = _np(out["rcnn_cls"]).astype(np.float32).reshape(B, M)
        keep = orc.select_detections(boxes, raw, cfg.rcnn_score_thresh, cfg.rcnn_nms_thresh)
        t.append(time.perf_counter())
        feats = out["rcnn_feat"].view(B, M, -1)
        aff = [self.affinity(feats[b - 1], feats[b]) for b in range(B)]
        t.append(time.perf_counter())
        self.stage_seconds = dict(zip(("backbone+rpn_heads", "proposal_layer", "roipool3d+canonical", "rcnn", "decode+detection_nms",
                                       "pairwise_affinity"), (round(b - a, 4) for a, b in zip(t[:-1], t[1:]))))
        return dict(r, rois=rois, roi_scores_raw=roi_scores, pts_input=pts_input, pred_boxes3d=boxes, keep=keep, affinity=aff, **out)
