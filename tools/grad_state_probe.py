"""GPU box: gradients of the rows route and of the operator route at the benchmarked widths, each called several times, asynchronous and
with a device synchronisation behind every library call (and with JM_POISON_EMPTY-style poisoning: POISON=1): which calls agree with which
(tests/test_gpu_train_full.py docstring).  usage: grad_state_probe.py {uniform|kitti|packed} [small]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from jmodt_amd import synth, train_joint, _lib as L
from jmodt_amd.detector import DetectorConfig
from jmodt_amd.train_rows import joint_forward_rows, pooled_rois
from tests.test_gpu_detector import make_engine
DEV = "cuda:0"; K = 64
if os.environ.get("POISON"):
    import tests.conftest as _c
    _c._poison_empty()
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
kind = sys.argv[1] if len(sys.argv) > 1 else "uniform"
small = len(sys.argv) > 2 and sys.argv[2] == "small"
eng = make_engine(seed=5, cfg=DetectorConfig.survey()).to(DEV).eval()
for p in eng.parameters(): p.requires_grad_(True)
if small:
    xyz_h, img_h, xy_h = synth.frames(2, 16384, 4321, kind=kind, H=96, W=320, native=(94, 310))
    xy_h = np.random.default_rng(5).uniform(-0.98, 0.98, size=xy_h.shape).astype(np.float32)
else:
    xyz_h, img_h, xy_h = synth.frames(2, 16384, 4321, kind=kind)
xyz, img, xy = T(xyz_h), T(img_h), T(xy_h)
tids = torch.randint(0, 6, (2, K), generator=torch.Generator().manual_seed(4)).float().to(DEV)
def grads():
    g = {k: (None if v.grad is None else v.grad.detach().clone()) for k, v in eng.named_parameters()}
    eng.zero_grad(set_to_none=True); return g
def nan_report(g, tag):
    bad = [k for k, v in g.items() if v is not None and not bool(torch.isfinite(v).all())]
    rb = g["rcnn_net.reg_layer.2.conv.bias"]
    print("   ", tag, "non-finite tensors:", len(bad), bad[:6], "| reg bias grad min/max", float(rb.min()), float(rb.max()), flush=True)
def cmp(a, b, tag):
    nan_report(a, tag)
    gmax = max(float(w.abs().max()) for w in b.values() if w is not None)
    errs = []
    for k, w in b.items():
        if w is None: continue
        scale = max(float(w.abs().max()), 1e-4 * gmax)
        errs.append((float((a[k] - w).abs().max()) / scale, k))
    errs.sort(reverse=True)
    print(tag, [(f"{e:.2e}", k) for e, k in errs[:4]], flush=True)
def rows(sync=False):
    L.SYNC_DEBUG = sync
    try:
        got = joint_forward_rows(eng, xyz, img, xy, rois_per_frame=K)
        train_joint.thin_loss(eng, got, tids).backward()
        torch.cuda.synchronize()
    finally:
        L.SYNC_DEBUG = False
    return got, grads()
t = time.time(); got, m1 = rows(); print("rows first call s", time.time() - t, flush=True)
N, C = xyz.shape[1], got["backbone_features"].shape[1]
frows = got["backbone_features"].detach().transpose(1, 2).reshape(2 * N, C).contiguous()
rois, pts_input, count = pooled_rois(eng, xyz, dict(rpn_cls=got["rpn_cls"], rpn_reg=got["rpn_reg"], feature_rows=frows), K)
del got
def ops():
    feats = train_joint.backbone_forward(eng.rpn.backbone_net, xyz, img, xy)
    ref = train_joint.rcnn_forward_train(eng.rcnn_net, pts_input)
    ref.update(rpn_cls=eng.rpn.rpn_cls_layer(feats).transpose(1, 2), rpn_reg=eng.rpn.rpn_reg_layer(feats).transpose(1, 2))
    train_joint.thin_loss(eng, ref, tids).backward()
    torch.cuda.synchronize()
    return grads()
t = time.time(); w1 = ops(); print("operators first call s", time.time() - t, flush=True)
t = time.time(); w2 = ops(); print("operators second call s", time.time() - t, flush=True)
cmp(w2, w1, "operators run2 vs run1")
cmp(m1, w1, "rows(async, first) vs operators")
for i in range(3):
    _, mi = rows()
    cmp(mi, m1, f"rows async run{i + 2} vs run1")
    cmp(mi, w1, f"rows async run{i + 2} vs operators")
_, ms = rows(sync=True)
cmp(ms, w1, "rows SYNC vs operators")
cmp(ms, m1, "rows SYNC vs rows async run1")
