"""Golden vectors of the reference's COMPLETE detector forward (tests/golden/forward_ref.npz).  Run in the authoring
container, where /root/reference exists:

    python tests/golden/make_golden_forward.py

What is executed is the reference's own model code, imported from /root/reference, in TEST mode on CPU:
jmodt/detection/modeling/point_rcnn.py::PointRCNN.forward = rpn.py (backbone.py: PointNet++ MSG backbone with the LI-Fusion image
branch, RPN heads) -> layers/proposal_layer.py::ProposalLayer (bbox_transform.decode_bbox_target, distance-based pre-NMS
budgets, iou3d_utils.nms_normal_gpu, post-NMS budgets) -> rcnn.py::RCNN.forward (layers/proposal_target_layer.py in TEST
mode: roipool3d_utils.roipool3d_gpu + canonical transform; xyz_up / merge_down, three set-abstraction levels, cls / reg
heads, rcnn_feat) — every line of Python of the inference path between the input dict and the network outputs.  The CUDA
extension entry points are bound to this repository's CPU oracle and the hard-coded device constructors produce CPU tensors,
exactly as in make_golden_glue.py (see its header).  The configuration is REDUCED (written into the reference's cfg before the
model is constructed; same topology, small widths and point counts) so that the fixture stays small; two constraints of the
reference's constructors are kept: RPN.FP_MLPS[0][-1] = LI_FUSION.IMG_FEATURES_CHANNEL = 128 (point_rcnn.py:19 hard-codes the
RCNN's input width, backbone.py sizes the final fusion by IMG_FEATURES_CHANNEL) and
RCNN.XYZ_UP_LAYER[-1] = 128 (rcnn.py:24-25 merges it with the 128 RPN channels).
Stored: the reduced configuration, the seeded input frames, the weights' seed (synth.seeded_state) with the parameter names /
shapes, and the outputs (backbone features, RPN
heads, RoIs and their scores, the pooled canonical RoI points, RCNN heads and features).  No reference source text.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from jmodt_amd import synth  # noqa: E402
import make_golden_glue as glue  # noqa: E402
import make_golden_model as mgm  # noqa: E402

MINI = dict(
    sa_npoints=(256, 128, 64, 32), sa_radius=((0.6, 1.5), (1.5, 3.0), (3.0, 6.0), (6.0, 12.0)),
    sa_nsample=((16, 32), (16, 32), (16, 32), (16, 32)),
    sa_mlps=(((16, 16, 16), (16, 16, 32)), ((16, 16, 32), (16, 32, 32)), ((32, 32, 64), (32, 48, 64)), ((64, 64, 64), (64, 80, 64))),
    fp_mlps=((128, 128), (32, 32), (64, 64), (64, 64)), rpn_cls_fc=(32,), rpn_reg_fc=(32,),
    img_channels=(3, 16, 16, 16, 32), point_channels=(48, 64, 128, 128), deconv_reduce=(4, 4, 4, 4), img_features_channel=128,
    rpn_pre_nms_top_n=300, rpn_post_nms_top_n=8, rcnn_num_points=64, rcnn_xyz_up=(32, 128), rcnn_sa_npoints=(32, 8, -1),
    rcnn_sa_radius=(0.8, 1.6, 100.0), rcnn_sa_nsample=(16, 16, 16), rcnn_sa_mlps=((32, 32, 32), (32, 32, 64), (64, 64, 64)),
    rcnn_cls_fc=(64, 64), rcnn_reg_fc=(64, 64), link_fc=(64, 64), se_fc=(64, 64))


def L(x):
    return [L(v) for v in x] if isinstance(x, (tuple, list)) else x


def apply_mini(cfg):
    """write the reduced configuration into the reference's cfg (before any model is constructed)"""
    m = MINI
    cfg.RPN.SA_CONFIG.NPOINTS, cfg.RPN.SA_CONFIG.RADIUS = L(m["sa_npoints"]), L(m["sa_radius"])
    cfg.RPN.SA_CONFIG.NSAMPLE, cfg.RPN.SA_CONFIG.MLPS = L(m["sa_nsample"]), L(m["sa_mlps"])
    cfg.RPN.FP_MLPS, cfg.RPN.CLS_FC, cfg.RPN.REG_FC = L(m["fp_mlps"]), L(m["rpn_cls_fc"]), L(m["rpn_reg_fc"])
    cfg.LI_FUSION.IMG_CHANNELS, cfg.LI_FUSION.POINT_CHANNELS = L(m["img_channels"]), L(m["point_channels"])
    cfg.LI_FUSION.DeConv_Reduce, cfg.LI_FUSION.IMG_FEATURES_CHANNEL = L(m["deconv_reduce"]), m["img_features_channel"]
    cfg.TEST.RPN_PRE_NMS_TOP_N, cfg.TEST.RPN_POST_NMS_TOP_N = m["rpn_pre_nms_top_n"], m["rpn_post_nms_top_n"]
    cfg.RCNN.NUM_POINTS, cfg.RCNN.XYZ_UP_LAYER = m["rcnn_num_points"], L(m["rcnn_xyz_up"])
    cfg.RCNN.SA_CONFIG.NPOINTS, cfg.RCNN.SA_CONFIG.RADIUS = L(m["rcnn_sa_npoints"]), L(m["rcnn_sa_radius"])
    cfg.RCNN.SA_CONFIG.NSAMPLE, cfg.RCNN.SA_CONFIG.MLPS = L(m["rcnn_sa_nsample"]), L(m["rcnn_sa_mlps"])
    cfg.RCNN.CLS_FC, cfg.RCNN.REG_FC = L(m["rcnn_cls_fc"]), L(m["rcnn_reg_fc"])
    cfg.REID.LINK_FC, cfg.REID.SE_FC = L(m["link_fc"]), L(m["se_fc"])


def main():
    glue.import_reference_with_oracle_extensions()
    from jmodt.config import cfg
    m = MINI
    apply_mini(cfg)
    from jmodt.detection.modeling.point_rcnn import PointRCNN

    model = PointRCNN(num_classes=2, use_xyz=True, mode="TEST").eval()
    # weights: synth.seeded_state — a function of the sorted parameter names and shapes, so the tests rebuild them from the
    # seed (the state dict itself would be 1.4 MB of the fixture)
    SEED = 91
    ref_sd = model.state_dict()
    filled = synth.seeded_state({k: tuple(v.shape) for k, v in ref_sd.items()}, SEED)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in filled.items()}, strict=False)
    _, img, _ = synth.frames(2, 512, 93, H=64, W=192, native=(62, 186))
    # a DENSE patch of road (16 m x 16 m: ~2 points per square metre), so that car-sized proposals pool a dozen distinct points
    rng = np.random.default_rng(94)
    xyz = np.stack([rng.uniform(-8, 8, (2, 512)), rng.uniform(-1, 3, (2, 512)), rng.uniform(6, 22, (2, 512))], axis=-1).astype(np.float32)
    xy = synth.pts_xy(xyz)
    inp = dict(pts_input=torch.from_numpy(xyz), img=torch.from_numpy(img), pts_xy=torch.from_numpy(xy))
    with torch.no_grad():
        # small regression outputs: proposals = anchor-sized boxes near their points (random residuals give boxes with negative
        # extents that hold a point or two), so that the RoIs pool dozens of distinct points
        [p_ for p_ in model.rpn.rpn_reg_layer.parameters() if p_.dim() > 1][-1].mul_(0.02)
        [p_ for p_ in model.rpn.rpn_reg_layer.parameters() if p_.dim() == 1][-1].mul_(0.02)
        # centre the RPN scores on the segmentation threshold, so that the mask channel of the RoI points is mixed
        raw = model.rpn(inp)["rpn_cls"]
        last = [p_ for p_ in model.rpn.rpn_cls_layer.parameters() if p_.dim() == 1][-1]
        last.sub_(raw.median() - float(np.log(cfg.RPN.SCORE_THRESH / (1 - cfg.RPN.SCORE_THRESH))))
    # bbox_transform.py:44 `.to(roi_box3d.get_device())` is -1 on CPU tensors
    get_device = torch.Tensor.get_device
    torch.Tensor.get_device = lambda self: "cpu"
    try:
        with torch.no_grad():
            out = model(inp)
            # the pooled canonical RoI points the RCNN consumed (proposal_target_layer.py:102-118), recomputed by the same call
            seg_mask = (torch.sigmoid(out["rpn_cls"][:, :, 0]) > cfg.RPN.SCORE_THRESH).float()
            info = dict(rpn_xyz=out["backbone_xyz"], rpn_features=out["backbone_features"].permute(0, 2, 1), seg_mask=seg_mask,
                        roi_boxes3d=out["rois"], pts_depth=torch.norm(out["backbone_xyz"], p=2, dim=2))
            pts_input, _ = model.rcnn_net.proposal_target_layer(info)
    finally:
        torch.Tensor.get_device = get_device
    keep = ("backbone_xyz", "backbone_features", "rpn_cls", "rpn_reg", "rois", "roi_scores_raw", "seg_result", "rcnn_cls", "rcnn_reg",
            "rcnn_feat")
    res = {f"out.{k}": out[k].detach().numpy() for k in keep}
    res["out.pts_input_geom"] = pts_input.numpy()[:, :, :5].copy()   # xyz (canonical), mask, depth; the other 128 columns are
    res["out.pts_input_sum"] = pts_input.numpy().astype(np.float64).sum(axis=(1, 2))   # copies of backbone_features rows
    # the few tensors that differ from the seeded fill (the shifted score bias)
    sd = {f"sd.{k}": v.numpy() for k, v in model.state_dict().items()
          if v.dtype.is_floating_point and not np.array_equal(v.numpy(), filled[k])}
    assert len(sd) == 3, list(sd)
    print({k: v.shape for k, v in res.items()})
    print("rois filled:", int((np.abs(res["out.rois"]).sum(-1) > 0).sum()), "of", res["out.rois"].shape[0] * res["out.rois"].shape[1],
          "| seg points:", int(res["out.seg_result"].sum()), "| max |rcnn_feat|", float(np.abs(res["out.rcnn_feat"]).max()),
          "| pooled mask mean", float(res["out.pts_input_geom"][..., 3].mean()),
          "| distinct pooled points per RoI", [len(np.unique(r[:, :3], axis=0)) for r in res["out.pts_input_geom"]])
    mgm.save("forward_ref.npz", source="reference PointRCNN.forward (TEST mode) over the CPU oracle's extension entry points",
             config=np.array(json.dumps(MINI)), seed=SEED, keys=np.array(json.dumps({k: list(v.shape) for k, v in ref_sd.items()})),
             xyz=xyz, img=img, pts_xy=xy, **res, **sd)


if __name__ == "__main__":
    main()
