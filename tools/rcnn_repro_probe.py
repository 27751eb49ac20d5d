"""GPU box: is the RCNN half (operator route, rows route) reproducible call to call on FIXED pooled points?  (yes: 1e-6 / 1e-7)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from jmodt_amd import synth, train_joint, _lib as L
from jmodt_amd.detector import DetectorConfig
from jmodt_amd.ops.pointnet2 import pointnet2_utils as PU
from jmodt_amd.train_rows import BnFold, joint_forward_rows, pooled_rois, rcnn_forward_rows, rpn_forward_rows
from tests.test_gpu_detector import make_engine
DEV = "cuda:0"; K = 64
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
kind = sys.argv[1] if len(sys.argv) > 1 else "uniform"
eng = make_engine(seed=5, cfg=DetectorConfig.survey()).to(DEV).eval()
for p in eng.parameters(): p.requires_grad_(True)
xyz_h, img_h, xy_h = synth.frames(2, 16384, 4321, kind=kind, H=96, W=320, native=(94, 310))
xy_h = np.random.default_rng(5).uniform(-0.98, 0.98, size=xy_h.shape).astype(np.float32)
xyz, img, xy = T(xyz_h), T(img_h), T(xy_h)
tids = torch.randint(0, 6, (2, K), generator=torch.Generator().manual_seed(4)).float().to(DEV)
def grads(prefix=""):
    g = {k: v.grad.detach().clone() for k, v in eng.named_parameters() if v.grad is not None and k.startswith(prefix)}
    eng.zero_grad(set_to_none=True); return g
def cmp(a, b, tag):
    gmax = max(float(w.abs().max()) for w in b.values())
    errs = sorted(((float((a[k] - w).abs().max()) / max(float(w.abs().max()), 1e-4 * gmax), k) for k, w in b.items()), reverse=True)
    print("   ", tag, [(f"{e:.2e}", k) for e, k in errs[:3]], flush=True)
d = lambda a, b: float((a.double() - b.double()).abs().max())
with torch.no_grad():
    got = joint_forward_rows(eng, xyz, img, xy, rois_per_frame=K)
    N, C = xyz.shape[1], got["backbone_features"].shape[1]
    frows = got["backbone_features"].transpose(1, 2).reshape(2 * N, C).contiguous()
    rois, pts_input, count = pooled_rois(eng, xyz, dict(rpn_cls=got["rpn_cls"], rpn_reg=got["rpn_reg"], feature_rows=frows), K)
torch.cuda.synchronize()
print("distinct points per RoI: min", int(count.min()), "median", int(count.float().median()), "max", int(count.max()))
# 1. FPS on the RoI sets
rx = pts_input[:, :, :3].contiguous()
picks = [PU.farthest_point_sample(rx, 128).clone() for _ in range(4)]
print("FPS picks equal across 4 runs:", [bool(torch.equal(p, picks[0])) for p in picks])
# 2. operators RCNN
outs, gs = [], []
for i in range(4):
    o = train_joint.rcnn_forward_train(eng.rcnn_net, pts_input)
    (o["rcnn_cls"].sum() + o["rcnn_reg"].sum()).backward(); torch.cuda.synchronize()
    outs.append({k: v.detach().clone() for k, v in o.items()}); gs.append(grads("rcnn_net."))
for i in range(1, 4):
    print(f"operators RCNN run{i+1} vs run1: outputs", {k: d(outs[i][k], outs[0][k]) for k in outs[0]})
    cmp(gs[i], gs[0], "grads (head sums only)")
# 3. operators RCNN with re-id loss
gs2 = []
for i in range(4):
    o = train_joint.rcnn_forward_train(eng.rcnn_net, pts_input)
    o.update(rpn_cls=torch.zeros(2, 4, 1, device=DEV), rpn_reg=torch.zeros(2, 4, 1, device=DEV))
    loss = train_joint.thin_loss(eng, o, tids); loss.backward(); torch.cuda.synchronize()
    gs2.append((float(loss), grads("rcnn_net.")))
for i in range(1, 4):
    print(f"operators RCNN + re-id run{i+1} vs run1: loss", gs2[i][0], gs2[0][0]); cmp(gs2[i][1], gs2[0][1], "grads")
# 4. rows RCNN
outs, gs = [], []
for i in range(4):
    o = rcnn_forward_rows(eng, pts_input, BnFold(eng.rcnn_net), count)
    (o["rcnn_cls"].sum() + o["rcnn_reg"].sum()).backward(); torch.cuda.synchronize()
    outs.append({k: v.detach().clone() for k, v in o.items()}); gs.append(grads("rcnn_net."))
for i in range(1, 4):
    print(f"rows RCNN run{i+1} vs run1: outputs", {k: d(outs[i][k], outs[0][k]) for k in outs[0]})
    cmp(gs[i], gs[0], "grads (head sums only)")
cmp(gs[0], gs2[0][1], "rows vs operators (head sums; operators incl. re-id: expect differences in feat path)")
# 5. RPN operators route
outs, gs = [], []
for i in range(3):
    feats = train_joint.backbone_forward(eng.rpn.backbone_net, xyz, img, xy)
    cls, reg = eng.rpn.rpn_cls_layer(feats), eng.rpn.rpn_reg_layer(feats)
    ((cls.sum() + reg.sum()) / N).backward(); torch.cuda.synchronize()
    outs.append(dict(feats=feats.detach().clone(), cls=cls.detach().clone(), reg=reg.detach().clone())); gs.append(grads("rpn."))
for i in range(1, 3):
    print(f"operators RPN run{i+1} vs run1: outputs", {k: d(outs[i][k], outs[0][k]) for k in outs[0]})
    cmp(gs[i], gs[0], "grads")
# 6. RPN rows route (asynchronous)
outs2, gsr = [], []
for i in range(3):
    o = rpn_forward_rows(eng, xyz, img, xy, BnFold(eng.rpn))
    ((o["rpn_cls"].sum() + o["rpn_reg"].sum()) / N).backward(); torch.cuda.synchronize()
    outs2.append(dict(feats=o["backbone_features"].detach().clone(), cls=o["rpn_cls"].detach().clone(), reg=o["rpn_reg"].detach().clone())); gsr.append(grads("rpn."))
for i in range(1, 3):
    print(f"rows RPN run{i+1} vs run1: outputs", {k: d(outs2[i][k], outs2[0][k]) for k in outs2[0]})
    cmp(gsr[i], gsr[0], "grads")
print("rows RPN vs operators RPN: feats", d(outs2[0]["feats"], outs[0]["feats"]))
cmp(gsr[0], gs[0], "rows vs operators RPN grads")
