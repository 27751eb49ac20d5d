"""Golden vectors that need the reference's CONFIG and MODEL CLASSES (tests/golden/{decode,fusion,tid_feature}_ref.npz,
model_keys_ref.json).  Run in the authoring container, where /root/reference exists:

    python tests/golden/make_golden_model.py

What is executed is the reference's own Python, imported from /root/reference:
  * jmodt/config.py as is (all values come from it);
  * jmodt/utils/bbox_transform.py::decode_bbox_target — the RPN variant as ProposalLayer calls it
    (proposal_layer.py:24-34) and the RCNN variant as the evaluation calls it (tools/eval.py:108-116);
  * jmodt/detection/modeling/point_rcnn.py::PointRCNN constructed in TEST mode: parameter / buffer names and shapes;
  * jmodt/detection/modeling/backbone.py::{BasicBlock, AttentionFusion (IALayer), the DeConv / image_fusion_conv /
    image_fusion_bn chain of PointNet2MSG} forward on CPU for seeded inputs (with a REDUCED channel configuration
    written into the reference's cfg before construction, to keep the fixture small);
  * jmodt/detection/modeling/rcnn.py::RCNN.get_unique_tid_feature (a static method).
Three things the image lacks are bridged WITHOUT touching reference code:
  * `easydict` (a third-party package jmodt/config.py imports) is absent: a 9-line attribute dict is registered under
    that name so that config.py executes unchanged;
  * the three CUDA extension modules (`pointnet2_cuda`, `iou3d_cuda`, `roipool3d_cuda`) do not exist: empty modules
    are registered under their names — none of their functions is called by anything generated here;
  * ProposalLayer.__init__ calls `.cuda()` on a constant (proposal_layer.py:14): Tensor.cuda is a no-op while the
    model is constructed.
Only data is stored: seeded inputs, expected outputs, parameter names / shapes.  No reference source text.
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REFERENCE = "/root/reference"


class _AttrDict(dict):
    """stand-in for the absent `easydict` package: a dict with attribute access, nested dicts converted"""
    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v
    __getattr__ = dict.__getitem__

    def __setattr__(self, k, v):
        self[k] = v

    def __setitem__(self, k, v):
        super().__setitem__(k, _AttrDict(v) if isinstance(v, dict) and not isinstance(v, _AttrDict) else v)


def import_reference():
    if not os.path.isdir(os.path.join(REFERENCE, "jmodt")):
        raise SystemExit("needs /root/reference")
    sys.modules["easydict"] = types.SimpleNamespace(EasyDict=_AttrDict)
    for name in ("jmodt.ops.pointnet2.pointnet2_cuda", "jmodt.ops.iou3d.iou3d_cuda", "jmodt.ops.roipool3d.roipool3d_cuda"):
        sys.modules[name] = types.ModuleType(name)
    sys.path.insert(0, REFERENCE)
    import jmodt.ops.iou3d as a, jmodt.ops.pointnet2 as b, jmodt.ops.roipool3d as c   # namespace packages
    a.iou3d_cuda = sys.modules["jmodt.ops.iou3d.iou3d_cuda"]
    b.pointnet2_cuda = sys.modules["jmodt.ops.pointnet2.pointnet2_cuda"]
    c.roipool3d_cuda = sys.modules["jmodt.ops.roipool3d.roipool3d_cuda"]


def save(name, **kw):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **kw)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def main():
    import_reference()
    from jmodt.config import cfg
    from jmodt.utils.bbox_transform import decode_bbox_target
    sys.path.insert(0, ROOT)
    from jmodt_amd import synth

    # ---------------- decode_bbox_target, both call forms, both BBOX_AVG_BY_BIN settings
    rng = np.random.default_rng(31)
    out = {}
    mean_size = torch.from_numpy(cfg.CLS_MEAN_SIZE[0])
    tensor_get_device = torch.Tensor.get_device
    torch.Tensor.get_device = lambda self: "cpu"     # bbox_transform.py:44 `.to(roi_box3d.get_device())` is -1 on CPU tensors
    try:
        for avg in (True, False):
            cfg.TRAIN.BBOX_AVG_BY_BIN = cfg.EVAL.BBOX_AVG_BY_BIN = avg
            xyz = synth.cloud(1, 700, seed=32)[0]
            reg = rng.normal(0, 1.5, (700, 76)).astype(np.float32)
            with torch.no_grad():     # proposal_layer.py:24-34
                p = decode_bbox_target(torch.from_numpy(xyz).view(-1, 3), torch.from_numpy(reg), anchor_size=mean_size,
                                       loc_scope=cfg.RPN.LOC_SCOPE, loc_bin_size=cfg.RPN.LOC_BIN_SIZE,
                                       num_head_bin=cfg.RPN.NUM_HEAD_BIN, get_xz_fine=cfg.RPN.LOC_XZ_FINE, get_y_by_bin=False,
                                       get_ry_fine=False)
                p[:, 1] += p[:, 3] / 2
            rois = synth.proposals(synth.cloud(1, 2048, 33), 500, 34)[0]
            rreg = rng.normal(0, 1.2, (500, 46)).astype(np.float32)
            with torch.no_grad():     # tools/eval.py:108-116
                q = decode_bbox_target(torch.from_numpy(rois.copy()), torch.from_numpy(rreg), anchor_size=mean_size,
                                       loc_scope=cfg.RCNN.LOC_SCOPE, loc_bin_size=cfg.RCNN.LOC_BIN_SIZE,
                                       num_head_bin=cfg.RCNN.NUM_HEAD_BIN, get_xz_fine=True, get_y_by_bin=cfg.RCNN.LOC_Y_BY_BIN,
                                       loc_y_scope=cfg.RCNN.LOC_Y_SCOPE, loc_y_bin_size=cfg.RCNN.LOC_Y_BIN_SIZE, get_ry_fine=True)
            tag = "avg" if avg else "argmax"
            out.update({f"{tag}_xyz": xyz, f"{tag}_rpn_reg": reg, f"{tag}_proposals": p.numpy(), f"{tag}_rois": rois,
                        f"{tag}_rcnn_reg": rreg, f"{tag}_boxes": q.numpy()})
    finally:
        torch.Tensor.get_device = tensor_get_device
        cfg.TRAIN.BBOX_AVG_BY_BIN = cfg.EVAL.BBOX_AVG_BY_BIN = True
    out["rpn_params"] = np.array([cfg.RPN.LOC_SCOPE, cfg.RPN.LOC_BIN_SIZE, cfg.RPN.NUM_HEAD_BIN], np.float64)
    out["rcnn_params"] = np.array([cfg.RCNN.LOC_SCOPE, cfg.RCNN.LOC_BIN_SIZE, cfg.RCNN.NUM_HEAD_BIN], np.float64)
    out["mean_size"] = cfg.CLS_MEAN_SIZE[0]
    save("decode_ref.npz", source="reference", **out)

    # ---------------- PointRCNN (TEST mode): names and shapes of every parameter / buffer
    from jmodt.detection.modeling.point_rcnn import PointRCNN
    tensor_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        torch.manual_seed(0)
        model = PointRCNN(num_classes=2, use_xyz=True, mode="TEST")
    finally:
        torch.Tensor.cuda = tensor_cuda
    keys = {k: list(v.shape) for k, v in model.state_dict().items()}
    json.dump(dict(source="reference", num_parameters=int(sum(p.numel() for p in model.parameters())), state_dict=keys,
                   config=dict(RPN_POST_NMS_TOP_N=cfg.TEST.RPN_POST_NMS_TOP_N, RPN_PRE_NMS_TOP_N=cfg.TEST.RPN_PRE_NMS_TOP_N,
                               RPN_NMS_THRESH=cfg.TEST.RPN_NMS_THRESH, RCNN_SCORE_THRESH=cfg.RCNN.SCORE_THRESH,
                               RCNN_NMS_THRESH=cfg.RCNN.NMS_THRESH, RPN_SCORE_THRESH=cfg.RPN.SCORE_THRESH,
                               POOL_EXTRA_WIDTH=cfg.RCNN.POOL_EXTRA_WIDTH, RCNN_NUM_POINTS=cfg.RCNN.NUM_POINTS)),
              open(os.path.join(HERE, "model_keys_ref.json"), "w"), indent=0)
    print("model_keys_ref.json:", len(keys), "entries")

    # ---------------- LI-Fusion blocks forward (reduced channel configuration written into the reference's cfg)
    from jmodt.detection.modeling import backbone as ref_bb
    cfg.LI_FUSION.IMG_CHANNELS = [3, 16, 16, 16, 32]
    cfg.LI_FUSION.POINT_CHANNELS = [48, 64, 128, 128]
    cfg.LI_FUSION.DeConv_Reduce = [4, 4, 4, 4]
    cfg.LI_FUSION.IMG_FEATURES_CHANNEL = 32
    cfg.RPN.SA_CONFIG.NPOINTS = [256, 128, 64, 32]
    cfg.RPN.SA_CONFIG.MLPS = [[[16, 16, 16], [16, 16, 32]], [[16, 16, 32], [16, 32, 32]], [[32, 32, 64], [32, 48, 64]],
                              [[64, 64, 64], [64, 80, 64]]]
    cfg.RPN.FP_MLPS = [[32, 32], [32, 32], [64, 64], [64, 64]]
    torch.manual_seed(1)
    net = ref_bb.PointNet2MSG(input_channels=0, use_xyz=True).eval()
    g = torch.Generator().manual_seed(2)
    for m in net.modules():
        if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
            with torch.no_grad():
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.2)
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.3)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
    out = {f"sd.{k}": v.numpy() for k, v in net.state_dict().items() if not k.startswith("SA_modules") and not k.startswith("FP_modules")}
    with torch.no_grad():
        image = torch.randn(2, 3, 32, 64, generator=g)
        img = [image]
        for i in range(4):
            img.append(net.Img_Block[i](img[i]))
        for i in range(4):
            out[f"img{i + 1}"] = img[i + 1].numpy()
        de = torch.cat([net.DeConv[i](img[i + 1]) for i in range(4)], dim=1)            # backbone.py:187-193
        fused_map = torch.nn.functional.relu(net.image_fusion_bn(net.image_fusion_conv(de)))
        out["image"], out["fused_map"] = image.numpy(), fused_map.numpy()
        for i in range(4):
            pc, ic = cfg.LI_FUSION.POINT_CHANNELS[i], cfg.LI_FUSION.IMG_CHANNELS[i + 1]
            P, I = torch.randn(2, pc, 29, generator=g), torch.randn(2, ic, 29, generator=g)
            out[f"fusion{i}_point"], out[f"fusion{i}_img"] = P.numpy(), I.numpy()
            out[f"fusion{i}_out"] = net.Fusion_Conv[i](P, I).numpy()
        P, I = torch.randn(2, 32, 41, generator=g), torch.randn(2, 8, 41, generator=g)
        out["final_point"], out["final_img"], out["final_out"] = P.numpy(), I.numpy(), net.final_fusion_img_point(P, I).numpy()
        xy = torch.rand(2, 41, 2, generator=g) * 2.2 - 1.1
        out["xy"], out["gathered"] = xy.numpy(), ref_bb.feature_gather(fused_map, xy).numpy()
    save("fusion_ref.npz", source="reference", **out)

    # ---------------- RCNN.get_unique_tid_feature (rcnn.py:145-156)
    from jmodt.detection.modeling.rcnn import RCNN
    tid = torch.tensor([7., 3., 7., 12., 3., 3., 9.])
    feat = torch.randn(7, 16, generator=g)
    u, f = RCNN.get_unique_tid_feature(tid, feat)
    save("tid_feature_ref.npz", source="reference", tid=tid.numpy(), feat=feat.numpy(), unique_tid=u.numpy(), unique_feat=f.numpy())


if __name__ == "__main__":
    main()
