"""Golden vectors of the reference's TRAINING-time affinity and re-id loss (tests/golden/train_ref.npz).  Run in the
authoring container, where /root/reference exists:

    python tests/golden/make_golden_train.py

What is executed is the reference's own code, imported from /root/reference, on CPU:
  * PointRCNN.forward in TRAIN mode (point_rcnn.py:24-70): the fixed RPN, ProposalLayer with the TRAIN budgets,
    ProposalTargetLayer's TRAIN path (proposal_target_layer.py:36-97: RoI sampling against the ground-truth boxes through
    iou3d_utils.boxes_iou3d_gpu, roipool, canonical transform, labels, `gt_tids` of the sampled RoIs), the RCNN, and the
    training affinity of rcnn.py:204-287 (per frame pair: foreground RoIs, get_unique_tid_feature, |p - d|, link_layer + dual
    softmax, start / end features + se_layer, the ground-truth link / start / end vectors);
  * the re-id part of get_rcnn_loss (train_functions.py:170-333 with cfg.TRAIN.FINETUNE, the reference's default training mode:
    the detection losses are skipped there, :182-183) — the function is a closure of model_joint_fn_decorator and is taken out
    of model_fn_train's cells; `.backward()` on its return value gives the reference's own gradients of the twelve head tensors.
The CUDA extension entry points are bound to the CPU oracle and the device constructors produce CPU tensors as in
make_golden_glue.py; configuration = make_golden_forward.py's reduced one (+ 32 sampled RoIs per frame), weights =
synth.seeded_state + the same two head adjustments; numpy / torch RNGs seeded (the RoI sampling draws from them).
Stored: the RoI features the heads saw (output of the last RCNN set-abstraction level) with the sampled RoIs' track ids — the
INPUTS of the affinity —, the reference's link / start / end outputs and ground-truth vectors, the three loss terms, the loss
and its gradients (head tensors and RoI features).  No reference source text.
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
import make_golden_forward as fwd  # noqa: E402
import make_golden_glue as glue  # noqa: E402
import make_golden_model as mgm  # noqa: E402


def main():
    glue.import_reference_with_oracle_extensions()
    from jmodt.config import cfg
    fwd.apply_mini(cfg)
    R = 32
    cfg.RCNN.ROI_PER_IMAGE = R
    cfg.TRAIN.RPN_PRE_NMS_TOP_N, cfg.TRAIN.RPN_POST_NMS_TOP_N = 300, 48
    assert cfg.TRAIN.FINETUNE and cfg.REID.ENABLED and not cfg.AUG_DATA
    from jmodt.detection.modeling.point_rcnn import PointRCNN
    from jmodt.detection.modeling.train_functions import model_joint_fn_decorator

    model = PointRCNN(num_classes=2, use_xyz=True, mode="TRAIN").eval()      # eval(): BatchNorm statistics fixed (RPN.FIXED does
    SEED = 91                                                                  # that for the RPN anyway, point_rcnn.py:29-30)
    ref_sd = model.state_dict()
    filled = synth.seeded_state({k: tuple(v.shape) for k, v in ref_sd.items()}, SEED)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in filled.items()}, strict=False)
    B = 4
    rng = np.random.default_rng(95)
    _, img, _ = synth.frames(B, 512, 96, H=64, W=192, native=(62, 186))
    xyz = np.stack([rng.uniform(-8, 8, (B, 512)), rng.uniform(-1, 3, (B, 512)), rng.uniform(6, 22, (B, 512))], axis=-1).astype(np.float32)
    xy = synth.pts_xy(xyz)
    inp = dict(pts_input=torch.from_numpy(xyz), img=torch.from_numpy(img), pts_xy=torch.from_numpy(xy))
    get_device = torch.Tensor.get_device
    torch.Tensor.get_device = lambda self: "cpu"
    try:
        with torch.no_grad():
            [p_ for p_ in model.rpn.rpn_reg_layer.parameters() if p_.dim() > 1][-1].mul_(0.02)
            [p_ for p_ in model.rpn.rpn_reg_layer.parameters() if p_.dim() == 1][-1].mul_(0.02)
            rpn_out = model.rpn(inp)
            rois, roi_scores = model.rpn.proposal_layer(rpn_out["rpn_cls"][:, :, 0], rpn_out["rpn_reg"], rpn_out["backbone_xyz"])
        # ProposalLayer with the TRAIN budgets (config.py:189-205: NMS threshold 0.85, 300 -> 48 here): frame 0's inputs and result
        prop = dict(prop_xyz=xyz[:1], prop_cls=rpn_out["rpn_cls"][:1, :, 0].numpy(), prop_reg=rpn_out["rpn_reg"][:1].numpy(),
                    prop_rois=rois[:1].numpy(), prop_scores=roi_scores[:1].numpy(),
                    prop_params=np.array([cfg.TRAIN.RPN_PRE_NMS_TOP_N, cfg.TRAIN.RPN_POST_NMS_TOP_N, cfg.TRAIN.RPN_NMS_THRESH], np.float64))
        with torch.no_grad():
            pass
        # ground truth: a few of the proposals themselves (IoU 1 with at least one RoI), track ids shared between the frames of a pair
        G = 6
        gt_boxes = np.zeros((B, G, 7), np.float32)
        gt_tids = np.zeros((B, G), np.float32)
        pick = [[0, 5, 9, 14, 20], [1, 4, 8, 13], [2, 6, 11, 17, 23], [0, 3, 7, 12, 19]]
        tids = [[11, 12, 13, 14, 15], [12, 13, 15, 16], [21, 22, 23, 24, 25], [25, 21, 26, 27, 22]]
        for b in range(B):
            for j, (ri, t) in enumerate(zip(pick[b], tids[b])):
                gt_boxes[b, j], gt_tids[b, j] = rois[b, ri].numpy(), t
        inp["gt_boxes3d"], inp["gt_tids"] = torch.from_numpy(gt_boxes), torch.from_numpy(gt_tids)
        feats = {}
        model.rcnn_net.SA_modules[-1].register_forward_hook(lambda mod, args, out: feats.__setitem__("f", out[1]))
        np.random.seed(97)
        torch.manual_seed(98)
        ret = model(inp)
        model_fn = model_joint_fn_decorator()
        cells = dict(zip(model_fn.__code__.co_freevars, (c.cell_contents for c in model_fn.__closure__)))
        tb = {}
        loss = cells["get_rcnn_loss"](model, ret, tb)
        heads = {f"rcnn_net.{h}.{k}": v for h in ("link_layer", "se_layer") for k, v in getattr(model.rcnn_net, h).named_parameters()}
        grads = torch.autograd.grad(loss, list(heads.values()) + [feats["f"]])
        grads, g_feat = grads[:-1], grads[-1]
    finally:
        torch.Tensor.get_device = get_device
    roi_feat = feats["f"].detach().squeeze(-1).view(B, R, -1).numpy()
    out = dict(roi_feat=roi_feat, gt_tids=ret["gt_tids"].detach().numpy(),
               rcnn_link=ret["rcnn_link"].detach().numpy(), rcnn_start=ret["rcnn_start"].detach().numpy(),
               rcnn_end=ret["rcnn_end"].detach().numpy(), gt_links=ret["gt_links"].numpy(), gt_starts=ret["gt_starts"].numpy(),
               gt_ends=ret["gt_ends"].numpy(), loss=np.float64(loss.item()),
               loss_terms=np.array([tb.get("rcnn_loss_link_mean", 0.0), tb.get("rcnn_loss_start_mean", 0.0), tb.get("rcnn_loss_end_mean", 0.0)]),
               weights=np.array([cfg.TRAIN.LINK_TRAIN_WEIGHT, cfg.TRAIN.SE_TRAIN_WEIGHT], np.float64))
    out.update(prop)
    out.update({f"grad.{k}": g.numpy() for k, g in zip(heads, grads)})
    out["grad.roi_feat"] = g_feat.squeeze(-1).view(B, R, -1).numpy()          # d(loss) / d(RoI features): what joint training sends back
    fg = (out["gt_tids"] > 0).sum(axis=1)
    print("foreground RoIs per frame", fg.tolist(), "| link entries", out["rcnn_link"].shape, "| loss", out["loss"], out["loss_terms"],
          "| gt_links positives", int(out["gt_links"].sum()))
    mgm.save("train_ref.npz", source="reference PointRCNN.forward (TRAIN mode) + get_rcnn_loss (FINETUNE) over the CPU oracle's entry points",
             config=np.array(json.dumps(fwd.MINI)), seed=SEED, keys=np.array(json.dumps({k: list(v.shape) for k, v in ref_sd.items()})), **out)


if __name__ == "__main__":
    main()
