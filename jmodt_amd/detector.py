"""The composed detect + affinity forward: the CALLER of the jmodt.ops hot path (SURVEY.md §2 rows 9-12,
BASELINE.json configs[2] + the pairwise affinity of configs[0]).

What the reference spreads over jmodt/detection/modeling/{backbone,rpn,rcnn,point_rcnn}.py,
layers/{proposal_layer,proposal_target_layer}.py and tracking/tracker.py:81-112 is ONE inference
pipeline here, laid out for a single MI355X:

    stream F  FPS pyramid 16384 -> 4096 -> 1024 -> 256 -> 64 (coordinates only; one workgroup / frame)
    stream I  image branch: 4 x (conv3x3 + BN + ReLU + conv3x3/2), channels-last, then the fused image map
    stream M  per level: fused SA scale kernels (ball query + group + MLP + max) -> LI-Fusion gather
              -> attention fusion; 4 x feature propagation; RPN heads; proposal layer (decode + banded NMS,
              whole batch, no host sync); roipool3d + canonical transform; RCNN (xyz lift + 3 SA levels
              + heads); box decode; per-frame detection NMS on the device; pairwise affinity of
              consecutive frames straight from the resident RoI features.

The parameter containers keep the reference's attribute names (`rpn.backbone_net.SA_modules.0.mlps.1.layer2.conv`,
`rpn.backbone_net.Fusion_Conv.2.IA_Layer.fc1`, `rcnn_net.link_layer.3.conv`, ...), so a JMODT checkpoint's
`model_state` loads with `load_state_dict`; the forward code is this package's own.  All dense arithmetic is
float32 (the reference's arithmetic; `dtype` of bench.py).  Eval mode only: BatchNorm is folded into the
neighbouring 1x1 convolution where that removes a pass over a tensor.
"""
import contextlib
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .ops import proposal as proposal_ops
from .ops.affinity import linear_rows, make_affinity_mlp, pairwise_affinity, pairwise_affinity_batched
from .ops.detections import DetectionCache, decode_rcnn_boxes, select_detections
from .ops.fusion import (PackedAttentionFusion, PackedImageFusion, bias_relu_, conv3x3_rgb_bias_relu, conv3x3_wino_bias_relu,
                         pack_wino_weight, wino_supported, feature_gather,
                         pack_rgb_weight)
from .ops.pointnet2 import fused, pointnet2_utils
from .ops.pointnet2 import pytorch_utils as pt_utils
from .ops.pointnet2.pointnet2_modules import PointnetFPModule, PointnetSAModule, PointnetSAModuleMSG
from .ops.pointnet2.pyramid import FpsPyramid, side_stream

_RCNN_KEYS = ("xyz_up.", "merge_down", "rcnn_")   # keys of `_folded` made from rcnn_net's parameters (see _refresh)
_FPS_SLOTS = (0, 4, 5, 6)      # side-stream slots of the FPS pyramids in flight (1: image branch, 2: start/end head, 3: detections)
from .profile import prof
from .ops.rcnn_lift import PackedRcnnLift
from .ops.roipool3d.roipool3d_utils import roipool3d_canonical_gpu


@dataclass
class DetectorConfig:
    """the values of jmodt/config.py the forward depends on (line numbers of that file)"""
    # RPN backbone (config.py:71-82)
    sa_npoints: Sequence[int] = (4096, 1024, 256, 64)
    sa_radius: Sequence[Sequence[float]] = ((0.1, 0.5), (0.5, 1.0), (1.0, 2.0), (2.0, 4.0))
    sa_nsample: Sequence[Sequence[int]] = ((16, 32), (16, 32), (16, 32), (16, 32))
    sa_mlps: Sequence[Sequence[Sequence[int]]] = (((16, 16, 32), (32, 32, 64)), ((64, 64, 128), (64, 96, 128)),
                                                  ((128, 196, 256), (128, 196, 256)), ((256, 256, 512), (256, 384, 512)))
    fp_mlps: Sequence[Sequence[int]] = ((128, 128), (256, 256), (512, 512), (512, 512))
    rpn_cls_fc: Sequence[int] = (128,)
    rpn_reg_fc: Sequence[int] = (128,)
    rpn_dp_ratio: float = 0.5
    rpn_loc_scope: float = 3.0            # :65-68
    rpn_loc_bin_size: float = 0.5
    rpn_num_head_bin: int = 12
    rpn_score_thresh: float = 0.2         # :97
    # LI-Fusion (config.py:43-52)
    img_channels: Sequence[int] = (3, 64, 128, 256, 512)
    point_channels: Sequence[int] = (96, 256, 512, 1024)
    deconv_reduce: Sequence[int] = (16, 16, 16, 16)
    deconv_kernels: Sequence[int] = (2, 4, 8, 16)
    img_features_channel: int = 128
    # proposal layer, TEST mode (config.py:204,213,224-230): 100 RoIs per frame as the reference's EVAL / TEST configuration;
    # SURVEY.md §8 / BASELINE.json benchmark 128 per frame: DetectorConfig.survey()
    rpn_pre_nms_top_n: int = 9000
    rpn_post_nms_top_n: int = 100
    rpn_nms_thresh: float = 0.8
    rpn_nms_type: str = "normal"          # :94
    # RCNN (config.py:100-139)
    pool_extra_width: float = 0.2         # :115
    rcnn_num_points: int = 512            # :132
    rcnn_xyz_up: Sequence[int] = (128, 128)
    rcnn_sa_npoints: Sequence[int] = (128, 32, -1)
    rcnn_sa_radius: Sequence[float] = (0.2, 0.4, 100.0)
    rcnn_sa_nsample: Sequence[int] = (64, 64, 64)
    rcnn_sa_mlps: Sequence[Sequence[int]] = ((128, 128, 128), (128, 128, 256), (256, 256, 512))
    rcnn_cls_fc: Sequence[int] = (512, 512)
    rcnn_reg_fc: Sequence[int] = (512, 512)
    rcnn_loc_scope: float = 1.5           # :118-124
    rcnn_loc_bin_size: float = 0.5
    rcnn_num_head_bin: int = 9
    rcnn_score_thresh: float = 0.2        # :159-160
    rcnn_nms_thresh: float = 0.1
    # re-id heads (config.py:163-169)
    link_fc: Sequence[int] = (512, 512)
    se_fc: Sequence[int] = (512, 512)
    mean_size: Sequence[float] = proposal_ops.CLS_MEAN_SIZE    # :38

    @property
    def rpn_reg_channels(self) -> int:   # rpn.py:31-37 with LOC_XZ_FINE
        return int(self.rpn_loc_scope / self.rpn_loc_bin_size) * 2 * 4 + self.rpn_num_head_bin * 2 + 3 + 1

    @property
    def rcnn_reg_channels(self) -> int:  # rcnn.py:73-77 with LOC_Y_BY_BIN = False
        return int(self.rcnn_loc_scope / self.rcnn_loc_bin_size) * 2 * 4 + self.rcnn_num_head_bin * 2 + 3 + 1

    @staticmethod
    def survey() -> "DetectorConfig":
        """the benchmarked configuration (BASELINE.json: 128 proposals per frame); everything else as
        jmodt/config.py"""
        return DetectorConfig(rpn_post_nms_top_n=128)

    @staticmethod
    def tiny() -> "DetectorConfig":
        """same topology, small widths / point counts: the chained-oracle tests and smoke runs"""
        return DetectorConfig(
            sa_npoints=(256, 128, 64, 32), sa_radius=((0.6, 1.5), (1.5, 3.0), (3.0, 6.0), (6.0, 12.0)),
            sa_mlps=(((16, 16, 16), (16, 16, 32)), ((16, 16, 32), (16, 32, 32)), ((32, 32, 64), (32, 48, 64)),
                     ((64, 64, 64), (64, 80, 64))),
            fp_mlps=((32, 32), (32, 32), (64, 64), (64, 64)), rpn_cls_fc=(32,), rpn_reg_fc=(32,),
            img_channels=(3, 16, 16, 16, 32), point_channels=(48, 64, 128, 128), deconv_reduce=(4, 4, 4, 4),
            img_features_channel=32, rpn_pre_nms_top_n=300, rpn_post_nms_top_n=16, rcnn_num_points=64,
            rcnn_xyz_up=(32, 32), rcnn_sa_npoints=(32, 8, -1), rcnn_sa_radius=(0.8, 1.6, 100.0),
            rcnn_sa_nsample=(16, 16, 16), rcnn_sa_mlps=((32, 32, 32), (32, 32, 64), (64, 64, 64)),
            rcnn_cls_fc=(64, 64), rcnn_reg_fc=(64, 64), link_fc=(64, 64), se_fc=(64, 64))


# ---------------------------------------------------------------------------------------------------------
# parameter containers (reference attribute names; forward code below is the engine's)
# ---------------------------------------------------------------------------------------------------------

class ImageBlock(nn.Module):
    """conv3x3 + BN + ReLU + conv3x3 stride 2 (BasicBlock of backbone.py:16-32 with stride=1)"""

    def __init__(self, cin: int, cout: int):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride=1, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, stride=2, padding=1, bias=False)

    def forward(self, x):
        return self.conv2(F.relu(self.bn1(self.conv1(x)), inplace=True))


class IALayer(nn.Module):
    """point-wise attention of the image feature by the point feature (backbone.py:35-63)"""

    def __init__(self, ic: int, pc: int):
        super().__init__()
        rc = pc // 4
        self.conv1 = nn.Sequential(nn.Conv1d(ic, pc, 1), nn.BatchNorm1d(pc), nn.ReLU())
        self.fc1, self.fc2, self.fc3 = nn.Linear(ic, rc), nn.Linear(pc, rc), nn.Linear(rc, 1)

    def forward(self, img_feats, point_feats):
        # (B, C, n) operands; the Linear layers act on the channel axis
        gate = torch.sigmoid(self.fc3(torch.tanh(self.fc1(img_feats.transpose(1, 2)) + self.fc2(point_feats.transpose(1, 2)))))
        return self.conv1(img_feats) * gate.transpose(1, 2)


class AttentionFusion(nn.Module):
    """backbone.py:66-81"""

    def __init__(self, ic: int, pc: int, oc: int):
        super().__init__()
        self.IA_Layer = IALayer(ic, pc)
        self.conv1 = nn.Conv1d(pc + pc, oc, 1)
        self.bn1 = nn.BatchNorm1d(oc)

    def forward(self, point_feats, img_feats):
        return F.relu(self.bn1(self.conv1(torch.cat((point_feats, self.IA_Layer(img_feats, point_feats)), dim=1))))


class PointNet2MSG(nn.Module):
    """parameters of the LI-Fusion backbone (backbone.py:92-157)"""

    def __init__(self, cfg: DetectorConfig, input_channels: int = 0):
        super().__init__()
        self.SA_modules = nn.ModuleList()
        cin, skip = input_channels, [input_channels]
        for k, npoint in enumerate(cfg.sa_npoints):
            specs = [[cin] + list(m) for m in cfg.sa_mlps[k]]
            self.SA_modules.append(PointnetSAModuleMSG(npoint=npoint, radii=list(cfg.sa_radius[k]),
                                                       nsamples=list(cfg.sa_nsample[k]), mlps=specs, use_xyz=True, bn=True))
            cin = sum(m[-1] for m in cfg.sa_mlps[k])
            skip.append(cin)
        self.Img_Block, self.Fusion_Conv, self.DeConv = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        for i in range(len(cfg.img_channels) - 1):
            self.Img_Block.append(ImageBlock(cfg.img_channels[i], cfg.img_channels[i + 1]))
            self.Fusion_Conv.append(AttentionFusion(cfg.img_channels[i + 1], cfg.point_channels[i], cfg.point_channels[i]))
            self.DeConv.append(nn.ConvTranspose2d(cfg.img_channels[i + 1], cfg.deconv_reduce[i],
                                                  kernel_size=cfg.deconv_kernels[i], stride=cfg.deconv_kernels[i]))
        q = cfg.img_features_channel // 4
        self.image_fusion_conv = nn.Conv2d(sum(cfg.deconv_reduce), q, kernel_size=1)
        self.image_fusion_bn = nn.BatchNorm2d(q)
        self.final_fusion_img_point = AttentionFusion(q, cfg.img_features_channel, cfg.img_features_channel)
        self.FP_modules = nn.ModuleList()
        for k in range(len(cfg.fp_mlps)):
            pre = cfg.fp_mlps[k + 1][-1] if k + 1 < len(cfg.fp_mlps) else cin
            self.FP_modules.append(PointnetFPModule(mlp=[pre + skip[k]] + list(cfg.fp_mlps[k])))


def _head(cin: int, fc: Sequence[int], cout: int, bn: bool, dp_ratio: float) -> nn.Sequential:
    """Conv1d(+BN)+ReLU per hidden width, Dropout after the first one when dp_ratio >= 0, plain Conv1d last
    (rpn.py:20-46, rcnn.py:43-89)"""
    layers: List[nn.Module] = []
    pre = cin
    for width in fc:
        layers.append(pt_utils.Conv1d(pre, width, bn=bn))
        pre = width
    layers.append(pt_utils.Conv1d(pre, cout, activation=None))
    if dp_ratio >= 0:
        layers.insert(1, nn.Dropout(dp_ratio))
    return nn.Sequential(*layers)


class RPN(nn.Module):
    def __init__(self, cfg: DetectorConfig):
        super().__init__()
        self.backbone_net = PointNet2MSG(cfg)
        width = cfg.fp_mlps[0][-1]
        self.rpn_cls_layer = _head(width, cfg.rpn_cls_fc, 1, True, cfg.rpn_dp_ratio)
        self.rpn_reg_layer = _head(width, cfg.rpn_reg_fc, cfg.rpn_reg_channels, True, cfg.rpn_dp_ratio)
        # rpn.py:64-69: focal-loss prior on the objectness bias, small regression weights
        nn.init.constant_(self.rpn_cls_layer[2].conv.bias, -4.59511985013459)
        nn.init.normal_(self.rpn_reg_layer[-1].conv.weight, mean=0, std=0.001)


class RCNN(nn.Module):
    def __init__(self, cfg: DetectorConfig, input_channels: int):
        super().__init__()
        self.rcnn_input_channel = 5          # xyz + mask + depth (rcnn.py:21)
        self.xyz_up_layer = pt_utils.SharedMLP([self.rcnn_input_channel] + list(cfg.rcnn_xyz_up), bn=False)
        c_out = cfg.rcnn_xyz_up[-1]
        self.merge_down_layer = pt_utils.SharedMLP([c_out * 2, c_out], bn=False)
        self.SA_modules = nn.ModuleList()
        cin = input_channels
        for k, npoint in enumerate(cfg.rcnn_sa_npoints):
            self.SA_modules.append(PointnetSAModule(npoint=npoint if npoint != -1 else None, radius=cfg.rcnn_sa_radius[k],
                                                    nsample=cfg.rcnn_sa_nsample[k], mlp=[cin] + list(cfg.rcnn_sa_mlps[k]),
                                                    use_xyz=True, bn=False))
            cin = cfg.rcnn_sa_mlps[k][-1]
        self.cls_layer = _head(cin, cfg.rcnn_cls_fc, 1, False, 0.0)
        self.reg_layer = _head(cin, cfg.rcnn_reg_fc, cfg.rcnn_reg_channels, False, 0.0)
        self.link_layer = make_affinity_mlp(cin, tuple(cfg.link_fc))
        self.se_layer = make_affinity_mlp(cin, tuple(cfg.se_fc))
        for m in self.modules():             # rcnn.py:116-134
            if isinstance(m, (nn.Conv1d, nn.Conv2d)):
                nn.init.xavier_normal_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
        nn.init.normal_(self.reg_layer[-1].conv.weight, mean=0, std=0.001)


# ---------------------------------------------------------------------------------------------------------
# the engine
# ---------------------------------------------------------------------------------------------------------

def _fold_conv_bn(conv: nn.Module, bn: Optional[nn.Module]) -> Tuple[torch.Tensor, torch.Tensor]:
    """(W (out, in), b (out)) of a 1x1 convolution / Linear followed by an eval-mode BatchNorm"""
    W = conv.weight.detach().reshape(conv.weight.shape[0], -1)
    b = conv.bias.detach() if conv.bias is not None else W.new_zeros(W.shape[0])
    if bn is not None:
        scale = bn.weight.detach() / torch.sqrt(bn.running_var.detach() + bn.eps)
        W = W * scale[:, None]
        b = (b - bn.running_mean.detach()) * scale + bn.bias.detach()
    return W.contiguous(), b.contiguous()


def _unit_wb(unit: nn.Module) -> Tuple[torch.Tensor, torch.Tensor]:
    """folded (W, b) of a pytorch_utils Conv1d/Conv2d unit (conv [+ bn.bn])"""
    bn = unit.bn.bn if hasattr(unit, "bn") else None
    return _fold_conv_bn(unit.conv, bn)


from ._registry import OPT_GEN as _OPT_GEN
from ._registry import EPOCH as _REGISTRATION_EPOCH      # bumped whenever ANY nn.Module of the process registers a parameter / buffer


class DetectAffinityEngine(nn.Module):
    """PointRCNN-with-affinity inference (point_rcnn.py:24-70 in EVAL mode + tools/eval.py:84-190 post-processing +
    tracker.py:81-112 affinity), batch-level, device-resident end to end."""

    def __init__(self, cfg: Optional[DetectorConfig] = None):
        super().__init__()
        self.cfg = cfg or DetectorConfig()
        self.rpn = RPN(self.cfg)
        self.rcnn_net = RCNN(self.cfg, input_channels=self.cfg.fp_mlps[0][-1])
        self.eval()
        self._folded: Dict[str, Tuple[torch.Tensor, torch.Tensor]] = {}
        self._folded_sig = None            # (data_ptr, _version) of every parameter / buffer the folded entries were made from
        self._sig_tensors: Optional[List[torch.Tensor]] = None
        self._sig_epoch = -1
        self.overlap = True                # FPS pyramid + image branch on side streams
        self.last_fps_idx: List[torch.Tensor] = []
        self.sparse_image_fusion = True    # final image feature only under the bilinear taps (else dense deconvolutions)
        self.fuse_rgb_conv = True          # image branch's first (3-channel) convolution + bias + ReLU as one pass
        self.wino_conv = True              # its other stride-1 convolutions + bias + ReLU as a fused Winograd F(2x2, 3x3) (conv_wino.hip)
        self.conv_find = True              # MIOpen picks each image convolution's kernel by measurement on first use (find
                                           # mode, scoped to those calls): 6.03 vs 6.34 ms over the seven 3x3 convolutions
                                           # (tools/miopen_find_probe.py); costs 1-3 s per new shape, once per process
        self.fuse_small_heads = True       # RCNN cls / reg heads: one MFMA launch per dense layer
        # ... or, where conv1d_stack takes the shape, one launch per HEAD.  Off: measured SLOWER (tools/rcnn_heads_bench.py,
        # HIP-graph replays, 1024 RoIs: 185 us against 141 us for the six jm_linear_rows launches) — a head is 32 tiles of 32 RoIs,
        # each a serial chain of three K = 512 / 256 layers on one CU, whereas a layer per launch spreads its 32 x 16 output tiles
        # over the machine; neither is on the critical path (the detections' side stream)
        self.fuse_head_stacks = False
        self.fuse_rcnn_lift = True         # xyz_up + merge_down (+ hoisted first SA layer) as one kernel
        self.fuse_attention = True         # LI-Fusion attention block as one kernel where it fits (else rocBLAS GEMMs)
        self.affinity_split_bf16 = False   # EXPERIMENTAL (csrc/affinity_x3.hip): link-head products as 3-term bf16 splits
        self.dedupe_rcnn = True            # RCNN SA1 / SA2: skip (centre, sample) rows that are exact copies (bit-identical output)
        self._prefetched = []              # FIFO of (xyz, FpsPyramid): the announced upcoming batches, next one first
        self._prefetched_img = None
        self.prefetch_depth = 1            # how many upcoming batches may have their FPS pyramid in flight (prefetch([x1, x2, ...]))
        self._fps_launches = 0
        self.prefetch_image = True         # prefetch(xyz, image) also starts the next batch's image pyramid
        self.prefetch_image_late = False   # ... after this batch's backbone (under proposals / RCNN) instead of under it

    # -- helpers -------------------------------------------------------------------------------------
    @staticmethod
    def _t(name: str, algo_bytes: int, fn, flops: int = 0):
        """caller-side (torch / MIOpen / rocBLAS) span, timed when jmodt_amd.profile.prof is enabled; the jm_*
        entry points time themselves"""
        return prof.region(name, fn, algo_bytes, flops)

    def _wb(self, key: str, make):
        hit = self._folded.get(key)
        if hit is None:
            hit = self._folded[key] = make()
        return hit

    def invalidate(self):
        """drop folded weights (not needed after load_state_dict / optimizer steps / .to(): `_refresh` sees those)"""
        self._folded.clear()
        self._folded_sig = None

    def _apply(self, fn, *args, **kwargs):
        # .to() / .cuda() / .float(): the parameters' storage is replaced, folded copies live on the old device
        self._folded.clear()
        self._folded_sig = self._sig_tensors = None
        return super()._apply(fn, *args, **kwargs)

    def _refresh(self):
        """drop every folded / packed weight when ANY parameter or buffer of the engine changed since it was made: torch
        bumps `_version` on every in-place update (load_state_dict, manual copy_, the for-loop / foreach optimizers), the FUSED
        optimizers (which do not) are counted by _registry's optimizer-step hook, `.to()` changes the storage.  Same rule as the SA / FP module caches (ops/pointnet2/fused.py:_packed_layers); called at every public
        entry (a few hundred attribute reads, ~0.1 ms of host time per batch)."""
        # (the link / start-end heads are never folded: ops/affinity.py hands their live tensors to the kernels on every
        # call, and the finetune step updates them every iteration — tools/train.py:96-107)
        # The tensors are walked afresh on every call: a parameter that was REPLACED (load_state_dict(assign=True),
        # `module.weight = nn.Parameter(...)`, parametrizations) is a new object, which a list cached at the first call would
        # never see — its id is part of the signature
        # ... but walking 300 modules costs the host ~1 ms, three times per step (the 4-frame training step is host-bound): the
        # list is rebuilt only when some module of the process (re-)registered a parameter or buffer since it was made — torch's
        # global registration hooks bump _REGISTRATION_EPOCH on every `register_parameter` / `register_buffer`, which is what an
        # attribute assignment of a Parameter, load_state_dict(assign=True) and parametrizations go through
        # Two groups, each with its own signature: the RPN (backbone, image branch, RPN heads) and the RCNN.  The reference's default
        # training mode (config.py:57 RPN.FIXED, point_rcnn.py:28-31) updates the RCNN every step under a FROZEN RPN — the frozen
        # half's ~100 packed weights must survive those optimizer steps (train_joint.rcnn_step)
        if self._sig_tensors is None or self._sig_epoch != _REGISTRATION_EPOCH[0]:
            skip = {id(t) for head in (self.rcnn_net.link_layer, self.rcnn_net.se_layer) for t in head.parameters()}
            rcnn = [t for t in list(self.rcnn_net.parameters()) + list(self.rcnn_net.buffers()) if id(t) not in skip]
            mine = {id(t) for t in rcnn} | skip
            self._sig_tensors = ([t for t in list(self.parameters()) + list(self.buffers()) if id(t) not in mine], rcnn)
            self._sig_epoch = _REGISTRATION_EPOCH[0]
        gen = _OPT_GEN.get                       # (inlined _registry.tensor_sig: ~500 tensors per call, two calls per step)
        sig = tuple(tuple([(id(t), t.data_ptr(), t._version, gen(id(t), 0)) for t in grp]) for grp in self._sig_tensors)
        old = self._folded_sig
        if sig != old:
            if old is None or sig[0] != old[0]:
                self._folded.clear()
            else:
                for k in [k for k in self._folded if k.startswith(_RCNN_KEYS)]:
                    del self._folded[k]
            self._folded_sig = sig

    def _attention_fusion(self, tag: str, mod: AttentionFusion, point_feats: torch.Tensor, img_feats: torch.Tensor):
        """AttentionFusion.forward on (B, C, n) operands with the BatchNorms folded (backbone.py:44-81)"""
        W_i, b_i = self._wb(tag + ".ia", lambda: _fold_conv_bn(mod.IA_Layer.conv1[0], mod.IA_Layer.conv1[1]))
        W_f, b_f = self._wb(tag + ".fuse", lambda: _fold_conv_bn(mod.conv1, mod.bn1))
        ia = mod.IA_Layer
        if point_feats.is_cuda and point_feats.dtype == torch.float32 and self.fuse_attention:
            # the whole block as ONE fp32-MFMA kernel on 32-point tiles (csrc/li_fusion.hip) where the tile fits the LDS
            packed = self._wb(tag + ".packed", lambda: PackedAttentionFusion(
                ia.fc1.weight, ia.fc1.bias, ia.fc2.weight, ia.fc2.bias, ia.fc3.weight, ia.fc3.bias, W_i, b_i, W_f, b_f))
            if packed.supported(point_feats.shape[0], point_feats.shape[2]):
                return packed(point_feats, img_feats)
        if point_feats.is_cuda and point_feats.dtype == torch.float32 and self.fuse_attention:
            # the coarse level (8 x 64 points, 512 / 1024-wide operands: the tile of li_fusion.hip does not fit the LDS): four launches
            # of csrc/points_gemm.hip — fc1 / fc2 as one two-operand layer with tanh, fc3 with the sigmoid (point-major: the gate),
            # the image convolution with ReLU and the gate as a per-point scale, the fusion convolution on [point ; image]
            from .ops.conv1d import points_linear, points_linear_supported
            Bn, pc, n = point_feats.shape
            icn = img_feats.shape[1]
            rc = ia.fc1.weight.shape[0]
            if (points_linear_supported(Bn, n, icn, pc, rc) and rc % 4 == 0 and points_linear_supported(Bn, n, pc, W_i.shape[0], W_f.shape[0])):
                def make():
                    w12 = torch.cat([ia.fc1.weight.detach(), ia.fc2.weight.detach()], dim=1).contiguous()
                    b12 = (ia.fc1.bias.detach() + ia.fc2.bias.detach()).contiguous()
                    w3 = torch.zeros((4, rc), dtype=torch.float32, device=w12.device)
                    w3[0] = ia.fc3.weight.detach().reshape(-1)
                    b3 = torch.zeros((4,), dtype=torch.float32, device=w12.device)
                    b3[0] = ia.fc3.bias.detach().reshape(-1)[0]
                    return w12, b12, w3, b3, W_i.contiguous(), b_i.contiguous(), W_f.contiguous(), b_f.contiguous()
                w12, b12, w3, b3, Wi_c, bi_c, Wf_c, bf_c = self._wb(tag + ".points", make)
                P, I = point_feats.contiguous(), img_feats.contiguous()
                t = points_linear(I, w12, b12, 2, x2=P)                                              # (B, rc, n)
                gate = points_linear(t, w3, b3, 3, out_rows=4)                                       # (B n, 4): column 0
                img_new = points_linear(I, Wi_c, bi_c, 1, rowscale=gate, rowscale_stride=4)          # (B, pc', n)
                return points_linear(P, Wf_c, bf_c, 1, x2=img_new)
        it, pt = img_feats.transpose(1, 2), point_feats.transpose(1, 2)                    # (B, n, C) views
        gate = torch.sigmoid(ia.fc3(torch.tanh(ia.fc1(it) + ia.fc2(pt))))                  # (B, n, 1)
        img_new = torch.relu(torch.baddbmm(b_i[None, :, None], W_i.expand(img_feats.shape[0], -1, -1), img_feats))
        img_new = img_new * gate.transpose(1, 2)
        pc = point_feats.shape[1]
        # conv1 on the concatenation = two GEMMs accumulating into one output (no cat tensor)
        out = torch.baddbmm(b_f[None, :, None], W_f[:, :pc].expand(point_feats.shape[0], -1, -1), point_feats)
        out = torch.baddbmm(out, W_f[:, pc:].expand(point_feats.shape[0], -1, -1), img_new)
        return torch.relu_(out)

    def _head_forward(self, tag: str, head: nn.Sequential, x: torch.Tensor) -> torch.Tensor:
        """a Conv1d head on (B, C, n): folded 1x1 convolutions as batched GEMMs"""
        units = [m for m in head if not isinstance(m, nn.Dropout)]
        if x.is_cuda and x.dtype == torch.float32 and x.shape[2] == 1 and x.shape[1] % 8 == 0 and self.fuse_small_heads:
            # (R, C, 1) inputs = plain rows: one MFMA launch per layer (jm_linear_rows) instead of GEMM + bias + ReLU
            rows = x[:, :, 0]
            for li, unit in enumerate(units):
                W, b = self._wb(f"{tag}.{li}", lambda u=unit: _unit_wb(u))
                if rows.shape[1] % 8:
                    break
                rows = linear_rows(rows, W, b, getattr(unit, "activation", None) is not None)
            else:
                return rows.unsqueeze(-1)
        for li, unit in enumerate(units):
            W, b = self._wb(f"{tag}.{li}", lambda u=unit: _unit_wb(u))
            x = torch.baddbmm(b[None, :, None], W.expand(x.shape[0], -1, -1), x)
            if getattr(unit, "activation", None) is not None:
                x = torch.relu_(x)
        return x

    def _rpn_heads_stack(self, feats: torch.Tensor) -> Optional[torch.Tensor]:
        """both RPN heads (rpn.py:34-58: Conv1d + BN + ReLU -> Conv1d each) as ONE two-layer stack: the first layers
        stacked row-wise, the second layers block-diagonal (the zero blocks contribute exact zeros) -> (B, 1 + C, N)"""
        if not (self.fuse_small_heads and feats.is_cuda and feats.dtype == torch.float32):
            return None

        def make():
            heads = []
            for tag, head in (("rpn_cls", self.rpn.rpn_cls_layer), ("rpn_reg", self.rpn.rpn_reg_layer)):
                units = [m for m in head if not isinstance(m, nn.Dropout)]
                if len(units) != 2 or getattr(units[0], "activation", None) is None or getattr(units[1], "activation", None) is not None:
                    return False
                heads.append([_unit_wb(u) for u in units])
            (Wc0, bc0), (Wc1, bc1) = heads[0]
            (Wr0, br0), (Wr1, br1) = heads[1]
            hc, hr = Wc0.shape[0], Wr0.shape[0]
            W1 = torch.zeros((Wc1.shape[0] + Wr1.shape[0], hc + hr), dtype=Wc0.dtype, device=Wc0.device)
            W1[:Wc1.shape[0], :hc] = Wc1
            W1[Wc1.shape[0]:, hc:] = Wr1
            from .ops.conv1d import PackedConv1dStack
            return PackedConv1dStack([(torch.cat([Wc0, Wr0], 0), torch.cat([bc0, br0], 0), True),
                                      (W1, torch.cat([bc1, br1], 0), False)], Wc0.shape[1])
        st = self._wb("rpn_heads.stack", make)
        if st is False or not st.supported(feats.shape[0], feats.shape[2]):
            return None
        return st(feats)

    # -- stage 1: backbone + RPN heads -----------------------------------------------------------------
    @torch.no_grad()
    def prefetch(self, xyz, image: Optional[torch.Tensor] = None) -> None:
        """announce the NEXT batch: its FPS pyramid (coordinates only, one workgroup per frame, ~6 ms of
        latency-bound sampling) starts now on the side stream, under the current batch's set abstraction /
        RCNN work, instead of at the head of the next call's critical path; with `image`, so does its image pyramid
        (the four convolution blocks depend on the image alone, backbone.py:162-168), which then no longer holds the
        point branch of the next call at every LI-Fusion level.  The next call must pass the same tensor objects.

        `xyz` may be a LIST of the upcoming clouds, next one first: up to `prefetch_depth` of them are kept in flight, each
        chain on a side stream of its own.  A sampling chain is serial and occupies one CU per frame: where a step is shorter
        than the chain (4 frames per step: 6.0 ms of chain against 5 ms of everything else), announcing only the next batch
        makes the chain the step; two in flight take it off the critical path again.  Clouds that are already in flight (same
        objects, same order) are left alone, so announcing [k+1, k+2] at step k and [k+2, k+3] at step k+1 starts one pyramid
        per step."""
        self._refresh()
        if not self.overlap:
            return
        upcoming = list(xyz) if isinstance(xyz, (list, tuple)) else [xyz]
        fifo = self._prefetched
        for i, x in enumerate(upcoming[:max(1, int(self.prefetch_depth))]):
            if i < len(fifo) and fifo[i][0] is x:
                continue
            for _, stale in fifo[i:]:
                stale.release()
            del fifo[i:]
            slot = _FPS_SLOTS[self._fps_launches % min(len(_FPS_SLOTS), max(1, int(self.prefetch_depth)))]
            self._fps_launches += 1
            fifo.append((x, FpsPyramid(x, list(self.cfg.sa_npoints), overlap=True, with_interp=True,
                                       grid_radii=self._grid_radii(), slot=slot)))
        if image is not None and self.prefetch_image:
            self._prefetched_img = (image, self._launch_image_branch(image))

    def _grid_radii(self):
        """largest ball-query radius per RPN SA level: the pyramid builds the levels' neighbour-search grids on its side stream"""
        return [max(r) for r in self.cfg.sa_radius]

    def _drop_kept(self):
        kept, self._kept = getattr(self, "_kept", None), None
        if kept is not None:
            ib, pyr = kept
            pyr.release()                                       # side stream ordered after this one, then dropped
            cur = torch.cuda.current_stream(ib["maps"][0].device)
            if ib["stream"] != cur:
                ib["stream"].wait_stream(cur)                   # same for the image pyramid's blocks

    def _take_prefetched(self, xyz: torch.Tensor):
        fifo = self._prefetched
        if fifo and fifo[0][0] is xyz:
            return fifo.pop(0)[1]
        for _, stale in fifo:          # another cloud than the announced one: what is in flight is of no use
            stale.release()
        del fifo[:]
        return None

    def _take_prefetched_image(self, image: torch.Tensor):
        hit, self._prefetched_img = self._prefetched_img, None
        return hit[1] if hit is not None and hit[0] is image else None

    def _launch_image_branch(self, image: torch.Tensor, pts_xy: Optional[torch.Tensor] = None) -> dict:
        """stream I: the image pyramid, channels-last, on its side stream (ordered after the current stream's position:
        the image is ready, and nothing this stream has queued so far still needs buffers the branch will recycle)"""
        net = self.rpn.backbone_net
        dev = image.device
        main = torch.cuda.current_stream(dev)
        img_stream = side_stream(dev, 1) if self.overlap else main
        img_stream.wait_stream(main)
        if img_stream is not main:
            image.record_stream(img_stream)
        img_maps, img_events = [], []
        with torch.cuda.stream(img_stream):
            cur = image if self.fuse_rgb_conv and image.is_cuda else image.contiguous(memory_format=torch.channels_last)
            for i, blk in enumerate(net.Img_Block):
                # flops of the block's MIOpen convolutions (3x3 stride 1 cin -> cout at (h, w), 3x3 stride 2 cout -> cout);
                # the 3-channel first layer of block 1 runs on the vector units (csrc/conv_rgb.hip) and is listed on its own
                h, w, cin, cout = cur.shape[2], cur.shape[3], blk.conv1.in_channels, blk.conv1.out_channels
                rgb = self.fuse_rgb_conv and image.is_cuda and cin == 3
                own = rgb or self._wino_ok(blk.conv1, cur)        # conv1 on this library's kernels: listed under its own name
                fl = 2 * 9 * cur.shape[0] * ((0 if own else cin * cout * h * w) + cout * cout * ((h + 1) // 2) * ((w + 1) // 2))
                cur = self._t(f"image_block_{i + 1}(MIOpen)", 0, lambda k=i, c=cur: self._image_block(k, c), flops=fl)
                ev = torch.cuda.Event()
                ev.record(img_stream)
                img_maps.append(cur)
                img_events.append(ev)
            H, W = image.shape[2], image.shape[3]
            sparse = None
            if self.sparse_image_fusion and image.is_cuda:
                sparse = self._wb("img_fusion.packed", lambda: PackedImageFusion(
                    *self._composed_image_fusion(), [dc.kernel_size[0] for dc in net.DeConv]))
                if not sparse.supported(img_maps, H, W):
                    sparse = None
            fused_map = None
            if sparse is None:      # dense fall-back: the full-resolution fused map, gathered afterwards
                fused_map = self._t("image_deconv+fusion_conv(MIOpen)", 0, lambda: self._image_fusion_map(img_maps))
            final = None
            if sparse is not None and pts_xy is not None:
                # the fused image feature under the points' bilinear taps needs the image pyramid and the pixel coordinates
                # only: evaluated here, behind the pyramid, while the other stream is still in its last LI-Fusion / FP levels
                if img_stream is not main:
                    pts_xy.record_stream(img_stream)
                with prof.scope("li_fusion_final"):
                    final = sparse(img_maps, pts_xy, H, W)
            fused_ev = torch.cuda.Event()
            fused_ev.record(img_stream)
        return dict(maps=img_maps, events=img_events, sparse=sparse, fused_map=fused_map, fused_ev=fused_ev, H=H, W=W,
                    stream=img_stream, final=final)

    @torch.no_grad()
    def backbone(self, xyz: torch.Tensor, image: torch.Tensor, pts_xy: torch.Tensor,
                 next_xyz=None, next_image: Optional[torch.Tensor] = None) -> torch.Tensor:
        """xyz (B, N, 3), image (B, 3, H, W), pts_xy (B, N, 2) in [-1, 1] -> point features (B, 128, N)
        (PointNet2MSG.forward, backbone.py:159-196).  next_xyz: the next batch's cloud, or the list of the upcoming ones
        (`prefetch`); next_image: the next batch's image"""
        cfg, net = self.cfg, self.rpn.backbone_net
        dev = xyz.device
        main = torch.cuda.current_stream(dev)
        B, N, _ = xyz.shape
        self._refresh()
        self._drop_kept()
        # --- stream F: the whole FPS chain (coordinates only); already running if this batch was announced ---
        pyr = self._take_prefetched(xyz) or FpsPyramid(xyz, list(cfg.sa_npoints), overlap=self.overlap, with_interp=self.overlap,
                                                       grid_radii=self._grid_radii())
        # --- stream I: image pyramid; already running (or done) if this batch's image was announced ---
        ib = self._take_prefetched_image(image) or self._launch_image_branch(image, pts_xy)
        if next_xyz is not None:
            self.prefetch(next_xyz, None if self.prefetch_image_late else next_image)
        img_maps, img_events, sparse, fused_map, fused_ev = ib["maps"], ib["events"], ib["sparse"], ib["fused_map"], ib["fused_ev"]
        H, W, img_stream = ib["H"], ib["W"], ib["stream"]
        # (allocated on the image stream, read by this one: kept referenced until the next call's _drop_kept / the next
        # _launch_image_branch, which orders the image stream after this one — no record_stream, see pyramid.py)
        # --- stream M: set abstraction + LI-Fusion per level ---
        l_xyz, l_feats, l_xy = [xyz], [None], [pts_xy]
        self.last_fps_idx = []
        for i, sa in enumerate(net.SA_modules):
            idx, new_xyz = pyr.level(i)
            self.last_fps_idx.append(idx)
            with prof.scope(f"rpn_sa{i + 1}"):
                _, feats, _ = sa(l_xyz[i], l_feats[i], new_xyz=new_xyz, grid=pyr.grid(i))
            xy_i = (pointnet2_utils.gather_point_rows(l_xy[i], idx) if idx.is_cuda else
                    torch.gather(l_xy[i], 1, idx.long().unsqueeze(-1).expand(-1, -1, 2)))       # backbone.py:170-171
            prof.stall(f"image_exposed_wait_L{i + 1}", lambda e=img_events[i]: main.wait_event(e))
            with prof.scope(f"li_fusion{i + 1}"):
                gathered = feature_gather(img_maps[i], xy_i)
                feats = self._t("attention_fusion(span)", 0, lambda f=feats, g=gathered, k=i: self._attention_fusion(
                    f"fusion{k}", net.Fusion_Conv[k], f, g))
            l_xyz.append(new_xyz); l_feats.append(feats); l_xy.append(xy_i)
        # --- feature propagation, coarse to fine (backbone.py:182-185) ---
        for i in range(-1, -(len(net.FP_modules) + 1), -1):
            with prof.scope(f"fp{len(net.FP_modules) + 1 + i}"):
                l_feats[i - 1] = net.FP_modules[i](l_xyz[i - 1], l_xyz[i], l_feats[i - 1], l_feats[i],
                                                   interp=pyr.interp(len(net.FP_modules) + i))
        # --- final image fusion on the full cloud (backbone.py:187-195) ---
        prof.stall("image_exposed_wait_final", lambda: main.wait_event(fused_ev))
        with prof.scope("li_fusion_final"):
            # sparse: the fused image feature evaluated only under the points' bilinear taps (csrc/image_fusion.hip)
            if ib["final"] is not None:
                gathered = ib["final"]
            else:
                gathered = sparse(img_maps, pts_xy, H, W) if sparse is not None else feature_gather(fused_map, pts_xy)
            out = self._t("attention_fusion(span)", 0, lambda: self._attention_fusion(
                "fusion_final", net.final_fusion_img_point, l_feats[0], gathered))
        # the previous batch's pyramids are dropped at the START of the next call: their blocks were allocated on the side
        # streams and record_stream-ed here, so freeing them makes the allocator record one event per block on THIS stream —
        # a ~100 us train of markers in front of the RPN heads when done at this point
        self._kept = (ib, pyr)
        if next_image is not None and self.prefetch_image_late and self.prefetch_image and self.overlap:
            self._prefetched_img = (next_image, self._launch_image_branch(next_image))    # ordered after this batch's backbone
        return out

    @contextlib.contextmanager
    def _miopen_find(self):
        """torch.backends.cudnn.benchmark (= MIOpen find mode on ROCm) for the convolutions issued inside, nothing else"""
        prev = torch.backends.cudnn.benchmark
        torch.backends.cudnn.benchmark = bool(self.conv_find) or prev
        try:
            yield
        finally:
            torch.backends.cudnn.benchmark = prev

    def _wino_ok(self, conv: nn.Conv2d, x: torch.Tensor) -> bool:
        """3x3 / stride 1 / padding 1 on a channels-last fp32 device tensor with cin % 16 == 0, cout % 64 == 0"""
        return bool(self.wino_conv and x.is_cuda and x.dtype == torch.float32 and tuple(conv.kernel_size) == (3, 3)
                    and tuple(conv.stride) == (1, 1) and tuple(conv.padding) == (1, 1) and conv.groups == 1
                    and tuple(conv.dilation) == (1, 1) and x.shape[1] == conv.in_channels
                    and wino_supported(conv.in_channels, conv.out_channels) and x.is_contiguous(memory_format=torch.channels_last))

    def _image_block(self, i: int, x: torch.Tensor) -> torch.Tensor:
        """BasicBlock (backbone.py:16-32) with the eval-mode BatchNorm folded into conv1: conv3x3 (MIOpen) ->
        + bias, ReLU in one in-place pass -> conv3x3 stride 2 (MIOpen)"""
        blk = self.rpn.backbone_net.Img_Block[i]

        def make():
            bn = blk.bn1
            scale = bn.weight.detach() / torch.sqrt(bn.running_var.detach() + bn.eps)
            # both weights in the activations' channels-last layout: an NCHW weight is re-laid-out by every call
            # (0.25 ms per step over the four blocks)
            W = (blk.conv1.weight.detach() * scale[:, None, None, None]).contiguous(memory_format=torch.channels_last)
            W2 = blk.conv2.weight.detach().contiguous(memory_format=torch.channels_last)
            b2 = blk.conv2.bias.detach() if blk.conv2.bias is not None else None
            return W, (bn.bias.detach() - bn.running_mean.detach() * scale).contiguous(), W2, b2
        W, b, W2, b2 = self._wb(f"img_block{i}", make)
        if (self.fuse_rgb_conv and x.is_cuda and x.dtype == torch.float32 and x.shape[1] == 3 and W.shape[0] % 4 == 0
                and (W.shape[0] <= 32 or W.shape[0] in (64, 128)) and tuple(blk.conv1.kernel_size) == (3, 3) and tuple(blk.conv1.stride) == (1, 1)
                and tuple(blk.conv1.padding) == (1, 1)):
            # the 3-channel first layer writes 1 GB and has K = 27: convolution + bias + ReLU as one HBM-bound pass
            wt = self._wb(f"img_block{i}.rgb", lambda: pack_rgb_weight(W))
            y = conv3x3_rgb_bias_relu(x, W, b, wt)
        elif self._wino_ok(blk.conv1, x):
            # 2.25x fewer multiplications than the direct form and no separate bias / ReLU pass
            y = conv3x3_wino_bias_relu(x, self._wb(f"img_block{i}.wino", lambda: pack_wino_weight(W)), b, W.shape[0])
        else:
            with self._miopen_find():
                y = F.conv2d(x, W, None, stride=1, padding=1)
            y = bias_relu_(y, b)
        with self._miopen_find():
            return F.conv2d(y, W2, b2, stride=blk.conv2.stride, padding=blk.conv2.padding)

    def _image_fusion_map(self, img_maps: List[torch.Tensor]) -> torch.Tensor:
        """relu(bn(conv1x1(cat_i deconv_i(img_i)))) (backbone.py:187-193).  The 1x1 fusion convolution is linear,
        so its slice for pyramid level i is composed with that level's kernel==stride transposed convolution into
        ONE transposed convolution per level writing the 32-channel result directly: the 64-channel concatenation
        (1 GB at 384x1280, batch 8) is never materialised."""
        net = self.rpn.backbone_net

        def make():
            Wf, bf = _fold_conv_bn(net.image_fusion_conv, net.image_fusion_bn)     # (q, sum reduce)
            ws, off = [], 0
            bias = bf.clone()
            for dc in net.DeConv:
                r = dc.out_channels
                Wi = dc.weight.detach()                                           # (cin, r, k, k)
                Ws = Wf[:, off:off + r]                                           # (q, r)
                ws.append(torch.einsum("crhw,qr->cqhw", Wi, Ws).contiguous())
                bias += Ws @ dc.bias.detach()
                off += r
            return ws, bias
        ws, bias = self._wb("img_fusion", make)
        acc = None
        for i, (m, w) in enumerate(zip(img_maps, ws)):
            k = net.DeConv[i].kernel_size[0]
            y = F.conv_transpose2d(m, w, bias if i == 0 else None, stride=k)
            acc = y if acc is None else acc.add_(y)
        return torch.relu_(acc)

    def _composed_image_fusion(self):
        """([wc_i (C_i, q, k_i, k_i)], bias (q)): each level's transposed convolution composed with its slice of the
        BatchNorm-folded 1x1 fusion convolution (linear o linear), deconvolution biases folded into the bias"""
        net = self.rpn.backbone_net

        def make():
            Wf, bf = _fold_conv_bn(net.image_fusion_conv, net.image_fusion_bn)     # (q, sum reduce)
            ws, off = [], 0
            bias = bf.clone()
            for dc in net.DeConv:
                r = dc.out_channels
                Wi = dc.weight.detach()                                           # (cin, r, k, k)
                Ws = Wf[:, off:off + r]                                           # (q, r)
                ws.append(torch.einsum("crhw,qr->cqhw", Wi, Ws).contiguous())
                bias += Ws @ dc.bias.detach()
                off += r
            return ws, bias
        return self._wb("img_fusion", make)

    @torch.no_grad()
    def rpn_forward(self, xyz, image, pts_xy, next_xyz=None, next_image=None) -> Dict[str, torch.Tensor]:
        """RPN.forward (rpn.py:71-87): backbone features + objectness / box regression per point.
        rpn_cls (B, N, 1) and rpn_reg (B, N, C) are STRIDED views of one channel-major (B, 1 + C, N) buffer whenever the two heads run
        as one stack (the engine's own consumers read it in place): same values and shapes as the reference's tensors, `.reshape`,
        indexing and every operator work; a caller that wants `.view(-1, C)` (the reference's loss code does, on the TRAIN route's
        outputs, which are contiguous) calls `.contiguous()` first."""
        feats = self.backbone(xyz, image, pts_xy, next_xyz, next_image)
        def heads():
            both = self._rpn_heads_stack(feats)
            if both is not None:                                   # (B, 1 + C, N): one launch for the two heads
                # (B, N, 1) / (B, N, C) VIEWS of the stack's channel-major output: the decode, the score sort's input copy and the
                # RoI-pooling input read it in place (two transposed copies per step gone, the decode's loads coalesced)
                return both[:, :1].transpose(1, 2), both[:, 1:].transpose(1, 2)
            cls = self._head_forward("rpn_cls", self.rpn.rpn_cls_layer, feats).transpose(1, 2).contiguous()   # (B, N, 1)
            reg = self._head_forward("rpn_reg", self.rpn.rpn_reg_layer, feats).transpose(1, 2).contiguous()   # (B, N, C)
            return cls, reg
        rpn_cls, rpn_reg = self._t("rpn_heads(span)", 0, heads)
        return dict(rpn_cls=rpn_cls, rpn_reg=rpn_reg, backbone_xyz=xyz, backbone_features=feats)

    # -- stage 2: proposals + RoI pooling + RCNN ---------------------------------------------------------
    @torch.no_grad()
    def proposals(self, rpn_out: Dict[str, torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
        """ProposalLayer.forward for the batch (proposal_layer.py:16-55; point_rcnn.py:47)"""
        cfg = self.cfg
        B, N = rpn_out["rpn_cls"].shape[:2]
        with prof.scope("proposal_layer"):
            return proposal_ops.proposal_layer(
                rpn_out["rpn_cls"][:, :, 0], rpn_out["rpn_reg"], rpn_out["backbone_xyz"],
                pre_nms_top_n=cfg.rpn_pre_nms_top_n, post_nms_top_n=cfg.rpn_post_nms_top_n,
                nms_thresh=cfg.rpn_nms_thresh, nms_type=cfg.rpn_nms_type, loc_scope=cfg.rpn_loc_scope,
                loc_bin_size=cfg.rpn_loc_bin_size, num_head_bin=cfg.rpn_num_head_bin, anchor_size=cfg.mean_size)

    @torch.no_grad()
    def pts_feature(self, rpn_out: Dict[str, torch.Tensor]) -> torch.Tensor:
        """per-point [mask, depth, rpn features] (B, N, 2 + C) of ProposalTargetLayer.forward in EVAL mode
        (point_rcnn.py:42-44, proposal_target_layer.py:26): independent of the proposals"""
        cfg = self.cfg
        xyz, feats = rpn_out["backbone_xyz"], rpn_out["backbone_features"]
        B, N, _ = xyz.shape
        C = feats.shape[1]
        pf = torch.empty((B, N, 2 + C), dtype=torch.float32, device=xyz.device)
        cls = rpn_out["rpn_cls"]
        if (xyz.is_cuda and feats.is_contiguous() and xyz.is_contiguous() and feats.dtype == torch.float32 and cls.dtype == torch.float32
                and cls.stride(0) >= 0 and cls.stride(1) >= 1):
            # mask + depth + transposed features as ONE launch (csrc/elementwise.hip) instead of seven element-wise passes
            import ctypes
            from . import _lib as L
            L.check(L.load().jm_pts_feature(B, N, C, ctypes.c_void_p(cls.data_ptr()), int(cls.stride(0)), int(cls.stride(1)), L.dev(xyz, torch.float32, "xyz"),
                                            L.dev(feats, torch.float32, "features"), float(cfg.rpn_score_thresh),
                                            ctypes.c_void_p(pf.data_ptr()), L.stream_ptr()), "pts_feature")
            return pf
        pf[:, :, 0] = (torch.sigmoid(rpn_out["rpn_cls"][:, :, 0]) > cfg.rpn_score_thresh).float()   # point_rcnn.py:42-43
        pf[:, :, 1] = torch.norm(xyz, p=2, dim=2) / 70.0 - 0.5                                       # :44; ptl.py:26
        pf[:, :, 2:] = feats.transpose(1, 2)
        return pf

    def roi_pool(self, rpn_out: Dict[str, torch.Tensor], rois: torch.Tensor, pts_feature: Optional[torch.Tensor] = None) -> torch.Tensor:
        """ProposalTargetLayer.forward in EVAL mode (proposal_target_layer.py:16-34,99-115): per-point
        [mask, depth, rpn features] -> pooled (B*M, S, 3 + 2 + C) in each RoI's canonical frame"""
        cfg = self.cfg
        xyz = rpn_out["backbone_xyz"]
        B = xyz.shape[0]
        if pts_feature is None:
            pts_feature = self.pts_feature(rpn_out)
        C = pts_feature.shape[2] - 2
        M, S = rois.shape[1], cfg.rcnn_num_points
        if xyz.is_cuda and self.dedupe_rcnn:
            # + the number of distinct points per RoI slab: rows count .. S-1 are cyclic copies (roipool3d_kernel.cu:123-160)
            pooled, _, count = roipool3d_canonical_gpu(xyz, pts_feature, rois, cfg.pool_extra_width, S, return_count=True)
            # keyed by the pooled tensor's identity AND version: rcnn_forward compacts only the very tensor these counts describe
            self._roi_count = ((pooled.data_ptr(), tuple(pooled.view(B * M, S, 5 + C).shape), pooled._version), count.view(B * M))
        else:
            pooled, _ = roipool3d_canonical_gpu(xyz, pts_feature, rois, cfg.pool_extra_width, S)
        return pooled.view(B * M, S, 5 + C)

    @torch.no_grad()
    def rcnn_forward(self, pts_input: torch.Tensor, heads: bool = True) -> Dict[str, torch.Tensor]:
        """RCNN.forward in EVAL mode (rcnn.py:158-202,288-289): pts_input (R, S, 5 + C) ->
        rcnn_cls (R, 1), rcnn_reg (R, 46), rcnn_feat (R, 512, 1)"""
        net = self.rcnn_net
        R, S, Cin = pts_input.shape
        self._refresh()
        fused.DedupeStats.last.clear()
        k = net.rcnn_input_channel
        rows = pts_input.view(R * S, Cin)
        up = [self._wb(f"xyz_up.{i}", lambda u=u: _unit_wb(u)) for i, u in enumerate(net.xyz_up_layer)]
        Wm, bm = self._wb("merge_down", lambda: _unit_wb(net.merge_down_layer[0]))
        c_up = up[-1][0].shape[0]

        def lift():
            h = rows[:, :k]                                   # strided view (row stride Cin): no copy, BLAS lda = Cin
            for W, b in up:
                h = torch.relu_(torch.addmm(b, h, W.t()))
            m = torch.addmm(bm, h, Wm[:, :c_up].t())
            m = torch.addmm(m, rows[:, k:], Wm[:, c_up:].t())  # merge_down on [xyz_feature | rpn_feature] without the cat
            return torch.relu_(m).view(R, S, -1).transpose(1, 2).contiguous()       # (R, C, S) for the SA kernels
        xyz = pts_input[:, :, 0:3].contiguous()
        sa1 = net.SA_modules[0]
        lifted = None
        if self.fuse_rcnn_lift and pts_input.is_cuda and len(up) == 2 and sa1.fuse:
            # one kernel for xyz_up + merge_down (+ the first SA layer hoisted in front of its gather)
            def make():
                g0 = sa1.groupers[0]
                hoist = None
                if isinstance(g0, pointnet2_utils.QueryAndGroup) and len(sa1.groupers) == 1 and g0.use_xyz and sa1.npoint:
                    hoist = fused.hoistable_first_layer(sa1.mlps[0], sa1.npoint, g0.nsample, pts_input.device)
                return PackedRcnnLift(up, (Wm, bm), hoist)
            packed = self._wb("rcnn_lift", make)
            if packed.supported(S):
                lifted = packed
        l_xyz, first = xyz, 0
        if lifted is not None and lifted.ho:
            g0 = sa1.groupers[0]
            pm = fused.pm_plan(sa1.mlps[0], pts_input.device, R, S, sa1.npoint, g0.nsample) is not None
            kept = getattr(self, "_roi_count", None)
            count = (kept[1] if kept is not None and kept[0] == (pts_input.data_ptr(), tuple(pts_input.shape), pts_input._version)
                     and kept[1].numel() == R else None)       # a slice or a modified copy of the pooled points: dense kernels
            dedupe = (pm and self.dedupe_rcnn and count is not None
                      and fused.dedupe_applies(sa1.mlps[0], pts_input.device, R, S, lifted.ho, sa1.npoint, g0.nsample))
            # (with the compaction, only canonical rows of u are ever gathered: the lift skips 32-point tiles of pure copies)
            u = lifted(pts_input, point_major=pm, count=count if dedupe else None)     # (R, H1, S), or (R, S, H1) for sa_mlp_pm
            res = None
            if dedupe:
                # the pooled sets are full of exact copies (cyclic padding -> copied centres -> back-filled neighbour lists):
                # only distinct rows go through the MFMA kernel (csrc/sa_dedupe.hip), the result is bit-identical
                with prof.scope("rcnn_sa1"):
                    res = fused.sa_scale_pm_dedupe(xyz, u, sa1.mlps[0], sa1.npoint, g0.radius, g0.nsample,
                                                   fused.canon_from_count(count, S), "rcnn_sa1")
            if res is not None:
                l_xyz, l_feats, rep = res
                first = 1
                sa2 = net.SA_modules[1] if len(net.SA_modules) > 1 else None
                g2 = sa2.groupers[0] if sa2 is not None and len(sa2.groupers) == 1 else None
                if (g2 is not None and sa2.fuse and sa2.npoint and isinstance(g2, pointnet2_utils.QueryAndGroup) and g2.use_xyz
                        and fused.hoistable_first_layer(sa2.mlps[0], sa2.npoint, g2.nsample, pts_input.device) is not None):
                    with prof.scope("rcnn_sa2"):
                        u2 = fused.hoisted_u_point_major(l_xyz, l_feats, sa2.mlps[0])
                        res2 = None if u2 is None else fused.sa_scale_pm_dedupe(l_xyz, u2, sa2.mlps[0], sa2.npoint, g2.radius,
                                                                                g2.nsample, rep, "rcnn_sa2")
                    if res2 is not None:
                        l_xyz, l_feats, _ = res2
                        first = 2
            else:
                if dedupe:                      # (cannot happen: dedupe_applies mirrors sa_scale_pm_dedupe) — rows were skipped
                    u = lifted(pts_input, point_major=pm)
                with prof.scope("rcnn_sa1"):
                    _, new_xyz = pointnet2_utils.farthest_point_sample_xyz(xyz, sa1.npoint)
                    nb = pointnet2_utils.ball_query(g0.radius, g0.nsample, xyz, new_xyz)
                    l_feats = fused.sa_mlp_pre_from_u(u, new_xyz, nb, sa1.mlps[0], point_major=pm)
                l_xyz, first = new_xyz, 1
        elif lifted is not None:
            l_feats = lifted(pts_input)
        else:
            flops = 2 * R * S * (sum(W.numel() for W, _ in up) + Wm.numel())
            l_feats = self._t("rcnn_xyz_lift+merge(rocBLAS)", 0, lift, flops=flops)
        for i, sa in enumerate(net.SA_modules):
            if i < first:
                continue
            with prof.scope(f"rcnn_sa{i + 1}"):
                l_xyz, l_feats, _ = sa(l_xyz, l_feats)
        out = dict(rcnn_feat=l_feats)
        if heads:
            out.update(self.rcnn_heads(l_feats))
        return out

    @torch.no_grad()
    def rcnn_heads(self, l_feats: torch.Tensor) -> Dict[str, torch.Tensor]:
        """classification / regression heads on the RoI features (rcnn.py:186-200).  Only the detections consume them — the
        affinity head takes the features themselves — so `forward` runs them on the detections' side stream"""
        net = self.rcnn_net

        def heads():
            st = self._rcnn_head_stacks(l_feats)
            if st is not None:
                # ONE launch per head (csrc/conv1d_stack.hip: the three dense layers chained on-chip on 32-RoI tiles) on the
                # transposed rows, shared by both heads; outputs point-major = the (R, C) rows the decode reads
                x = l_feats[:, :, 0].t().contiguous().unsqueeze(0)                 # (1, C, R)
                return st[0](x, point_major=True)[0], st[1](x, point_major=True)[0]
            return (self._head_forward("rcnn_cls", net.cls_layer, l_feats).squeeze(-1),
                    self._head_forward("rcnn_reg", net.reg_layer, l_feats).squeeze(-1))
        rcnn_cls, rcnn_reg = self._t("rcnn_heads(span)", 0, heads)
        return dict(rcnn_cls=rcnn_cls, rcnn_reg=rcnn_reg)

    def _rcnn_head_stacks(self, l_feats: torch.Tensor):
        """(cls stack, reg stack) when both heads (rcnn.py:57-89: Conv1d (+ BN) + ReLU x 2 -> Conv1d) run as one conv1d_stack
        launch each on (1, C, R), else None"""
        if not (self.fuse_head_stacks and self.fuse_small_heads and l_feats.is_cuda and l_feats.dtype == torch.float32
                and l_feats.dim() == 3 and l_feats.shape[2] == 1 and l_feats.shape[0] % 32 == 0):
            return None

        def make():
            from .ops.conv1d import PackedConv1dStack
            out = []
            for head in (self.rcnn_net.cls_layer, self.rcnn_net.reg_layer):
                units = [m for m in head if not isinstance(m, nn.Dropout)]
                if not 1 <= len(units) <= 3:
                    return False
                layers = [(*_unit_wb(u), getattr(u, "activation", None) is not None) for u in units]
                out.append(PackedConv1dStack(layers, layers[0][0].shape[1]))
            return tuple(out)
        st = self._wb("rcnn_heads.stacks", make)
        if st is False or not all(t.supported(1, l_feats.shape[0]) for t in st):
            return None
        return st

    # -- the whole path ----------------------------------------------------------------------------------
    @torch.no_grad()
    def _trunk(self, xyz, image, pts_xy, next_xyz=None, next_image=None, heads: bool = True):
        rpn_out = self.rpn_forward(xyz, image, pts_xy, next_xyz, next_image)
        pf = None
        if self.overlap and xyz.is_cuda:
            # the roipool input (mask, depth, transposed features: half a dozen element-wise passes) does not depend on the
            # proposals: built on a side stream under the proposal layer's sort / decode / NMS chain
            main, side = torch.cuda.current_stream(xyz.device), side_stream(xyz.device, 3)
            side.wait_stream(main)
            # (inputs allocated on this stream, read by the side stream: no record_stream — this stream waits for the side
            # stream below, before any of them can be freed, so their blocks cannot be recycled under the side stream's reads;
            # a record_stream makes the allocator record and poll one event per tensor when it is freed)
            with torch.cuda.stream(side):
                pf = self.pts_feature(rpn_out)
            pf.record_stream(main)
        rois, roi_scores = self.proposals(rpn_out)
        if pf is not None:
            main.wait_stream(side)
        pts_input = self.roi_pool(rpn_out, rois, pf)
        out = self.rcnn_forward(pts_input, heads=heads)
        return rpn_out, rois, roi_scores, pts_input, out

    @torch.no_grad()
    def _detections(self, rois, out) -> Tuple[DetectionCache, torch.Tensor]:
        cfg = self.cfg
        B, M = rois.shape[:2]
        with prof.scope("detections"):
            boxes = decode_rcnn_boxes(rois.view(-1, 7), out["rcnn_reg"], cfg.rcnn_loc_scope, cfg.rcnn_loc_bin_size,
                                      cfg.rcnn_num_head_bin, cfg.mean_size).view(B, M, 7)
            feats = out["rcnn_feat"].view(B, M, -1)
            cache = select_detections(boxes, out["rcnn_cls"].view(B, M), feats, cfg.rcnn_score_thresh, cfg.rcnn_nms_thresh)
        return cache, boxes

    @torch.no_grad()
    def detect(self, xyz, image, pts_xy, next_xyz=None, next_image=None) -> Tuple[DetectionCache, Dict[str, torch.Tensor]]:
        """frames -> device-resident detections (boxes, scores, 512-d features, per-frame counts)"""
        rpn_out, rois, roi_scores, pts_input, out = self._trunk(xyz, image, pts_xy, next_xyz, next_image)
        cache, boxes = self._detections(rois, out)
        inter = dict(rpn_out, rois=rois, roi_scores_raw=roi_scores, pts_input=pts_input, pred_boxes3d=boxes, **out)
        return cache, inter

    @torch.no_grad()
    def forward(self, xyz, image, pts_xy, next_xyz=None, next_image=None):
        """detect + affinity of every frame against its predecessor in the batch (frame 0 against the last):
        returns (DetectionCache, [(A (M, M), start (M), end (M)) per frame]) with all M RoI slots as the
        affinity operands (fixed work per frame: P = D = M)."""
        overlap = self.overlap and xyz.is_cuda
        rpn_out, rois, roi_scores, pts_input, out = self._trunk(xyz, image, pts_xy, next_xyz, next_image, heads=not overlap)
        B, M = rois.shape[:2]
        dev = rois.device
        main = torch.cuda.current_stream(dev) if rois.is_cuda else None
        # the RCNN's classification / regression heads, box decode + score filter + per-frame NMS + gathers (two dozen
        # latency-bound launches) do not feed the affinity head (it takes the features of all M RoI slots): they run on a side
        # stream under the affinity GEMMs
        side = side_stream(dev, 3) if overlap else None
        if side is not None:
            side.wait_stream(main)                      # (inputs: no record_stream, see _trunk — joined below before they can be freed)
            with torch.cuda.stream(side):
                out.update(self.rcnn_heads(out["rcnn_feat"]))
                cache, boxes = self._detections(rois, out)
        else:
            cache, boxes = self._detections(rois, out)
        feats = out["rcnn_feat"].view(B, M, -1)
        C = feats.shape[2]
        link, se = self.rcnn_net.link_layer, self.rcnn_net.se_layer
        with prof.scope(f"affinity_{B}x{M}x{M}"):
            # every frame against its predecessor, all B problems as one GEMM chain (jm_affinity_forward_batched)
            A, start, end = pairwise_affinity_batched(torch.roll(feats, 1, 0), feats, link, se, split_bf16=self.affinity_split_bf16)
            aff = [(A[b], start[b], end[b]) for b in range(B)]
        if side is not None:
            main.wait_stream(side)
            for t in (boxes, cache.boxes, cache.scores, cache.raw_scores, cache.feats, cache.count, cache.roi_index,
                      out["rcnn_cls"], out["rcnn_reg"]):
                t.record_stream(main)
        inter = dict(rpn_out, rois=rois, roi_scores_raw=roi_scores, pts_input=pts_input, pred_boxes3d=boxes, **out)
        return cache, aff, inter
