import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from jmodt_amd import synth, train_joint
from jmodt_amd.detector import DetectorConfig
from jmodt_amd.train_rows import BnFold, rcnn_forward_rows
from tests.test_gpu_detector import make_engine
DEV = "cuda:0"
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
eng = make_engine(seed=3, cfg=DetectorConfig.tiny()).to(DEV)
xyz, img, xy = synth.frames(2, 2048, 77, H=96, W=320, native=(94, 310))
xy = np.random.default_rng(5).uniform(-0.98, 0.98, size=xy.shape).astype(np.float32)
xyz, img, xy = T(xyz), T(img), T(xy)
K = min(64, eng.cfg.rpn_post_nms_top_n)
tids = torch.randint(0, 6, (2, K), generator=torch.Generator().manual_seed(4)).float().to(DEV)
train_joint.prepare_rcnn(eng)
for m in eng.modules():
    if isinstance(m, torch.nn.Dropout):
        m.eval()
loss, out = train_joint.rcnn_forward_backward(eng, xyz, img, xy, tids, rois_per_frame=K)
with torch.no_grad():
    rpn_out = eng.rpn_forward(xyz, img, xy)
    rois, _ = eng.proposals(rpn_out)
    print("rois equal", torch.equal(rois[:, :K], out["rois"]), K, rois.shape)
    pts = eng.roi_pool(rpn_out, rois[:, :K].contiguous())
    ref = eng.rcnn_forward(pts)
    op = train_joint.rcnn_forward_train(eng.rcnn_net, pts)
    count = eng._roi_count[1]
    r1 = rcnn_forward_rows(eng, pts, BnFold(eng.rcnn_net), count)
    r0 = rcnn_forward_rows(eng, pts, BnFold(eng.rcnn_net), None)
d = lambda a, b: float((a.reshape(-1) - b.reshape(-1)).abs().max())
for k in ("rcnn_cls", "rcnn_reg", "rcnn_feat"):
    print(k, "step-vs-fused", d(out[k], ref[k]), "op-vs-fused", d(op[k], ref[k]), "rows(count)-vs-op", d(r1[k], op[k]), "rows(nocount)-vs-op", d(r0[k], op[k]),
          "step-vs-rows(count)", d(out[k], r1[k]))
eng.eval()
with torch.no_grad():
    ref2 = eng.rcnn_forward(pts)
for k in ("rcnn_cls", "rcnn_reg", "rcnn_feat"):
    print(k, "fused(eval)-vs-fused(train)", d(ref2[k], ref[k]), "op-vs-fused(eval)", d(op[k], ref2[k]))
