"""Golden vectors of the reference's BACKWARD through the whole detector (tests/golden/backward_ref.npz).  Run in the authoring
container, where /root/reference exists:

    python tests/golden/make_golden_backward.py

Same reduced configuration, weights and frames as forward_ref.npz (make_golden_forward.py).  What is executed is the reference's
own code with autograd ON: `model.rpn(input)` — backbone.py's set abstraction / feature propagation through the reference's
autograd Functions (pointnet2_utils.py: GroupingOperation.backward, ThreeInterpolate.backward, GatherOperation.backward, bound to
this repository's CPU oracle as in make_golden_glue.py), the image blocks, the LI-Fusion gathers (F.grid_sample) and attention
blocks, the RPN heads — and `model(input)` for the RCNN (rcnn.py:158-202 on the pooled RoI points; roipool3d and the proposal
layer carry no gradient in the reference either).  BatchNorm in eval mode (running statistics), so that the numbers do not depend
on a batch.  Losses: (sum rpn_cls + sum rpn_reg) / N for the RPN side, sum rcnn_cls + sum rcnn_reg for the RCNN side — the
"thin loss" of jmodt_amd/train_joint.py without its re-id term (pinned separately by train_ref.npz).
Stored per parameter: L2 norm, sum, max |g| and 64 entries at fixed positions of the gradient — a 1 MB state-dict-sized dump would
pin nothing more.  No reference source text.
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

import make_golden_glue as glue  # noqa: E402
import make_golden_forward as mgf  # noqa: E402
import make_golden_model as mgm  # noqa: E402

SAMPLES = 64


def digest(g: torch.Tensor):
    flat = g.detach().double().reshape(-1)
    pos = np.floor(np.linspace(0, flat.numel() - 1, SAMPLES)).astype(np.int64)
    return [float(flat.norm()), float(flat.sum()), float(flat.abs().max())], flat[torch.from_numpy(pos)].float().numpy()


def main():
    glue.import_reference_with_oracle_extensions()
    from jmodt.config import cfg
    mgf.apply_mini(cfg)
    from jmodt.detection.modeling.point_rcnn import PointRCNN
    from tests.test_oracle_cpu import reference_forward_fixture
    _, sd, g = reference_forward_fixture()
    model = PointRCNN(num_classes=2, use_xyz=True, mode="TEST").eval()
    missing = model.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys and all(k.endswith("num_batches_tracked") for k in missing.missing_keys), missing
    inp = dict(pts_input=torch.from_numpy(g["xyz"]), img=torch.from_numpy(g["img"]), pts_xy=torch.from_numpy(g["pts_xy"]))
    N = g["xyz"].shape[1]
    for p in model.parameters():
        p.requires_grad_(True)
    # ---- RPN side: backbone + heads with autograd on
    with torch.enable_grad():
        r = model.rpn(inp)
        assert np.abs(r["rpn_cls"].detach().numpy() - g["out.rpn_cls"]).max() < 1e-5          # the fixture's forward, reproduced
        loss_rpn = (r["rpn_cls"].sum() + r["rpn_reg"].sum()) / N
    loss_rpn.backward()
    grads = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    assert all(k.startswith("rpn.") for k in grads) and len(grads) > 100, len(grads)
    model.zero_grad(set_to_none=True)
    # ---- RCNN side: point_rcnn.py runs the RPN without grad in eval mode and the RCNN with
    get_device = torch.Tensor.get_device
    torch.Tensor.get_device = lambda self: "cpu"
    try:
        with torch.enable_grad():
            out = model(inp)
            loss_rcnn = out["rcnn_cls"].sum() + out["rcnn_reg"].sum()
        loss_rcnn.backward()
    finally:
        torch.Tensor.get_device = get_device
    assert np.abs(out["rcnn_cls"].detach().numpy() - g["out.rcnn_cls"]).max() < 1e-5
    rc = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    assert all(k.startswith("rcnn_net.") for k in rc) and not any("link_layer" in k or "se_layer" in k for k in rc), sorted(rc)[:5]
    grads.update(rc)
    names = sorted(grads)
    stats, samples = zip(*(digest(grads[k]) for k in names))
    print(len(names), "gradient tensors;", "loss_rpn", float(loss_rpn), "loss_rcnn", float(loss_rcnn))
    print("largest:", sorted(((s[2], k) for s, k in zip(stats, names)), reverse=True)[:3])
    mgm.save("backward_ref.npz", source="reference rpn(input) / PointRCNN.forward with autograd over the CPU oracle's extension entry "
             "points, eval-mode BatchNorm; weights / frames = forward_ref.npz",
             names=np.array(json.dumps(names)), shapes=np.array(json.dumps({k: list(grads[k].shape) for k in names})),
             stats=np.asarray(stats, np.float64), samples=np.asarray(samples, np.float32),
             loss_rpn=np.float64(float(loss_rpn)), loss_rcnn=np.float64(float(loss_rcnn)))


if __name__ == "__main__":
    main()
