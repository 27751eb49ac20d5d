"""which piece of the forward alternates between two states from call to call?  (see tests/test_gpu_train_full.py docstring)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from jmodt_amd import synth, train_joint
from jmodt_amd.detector import DetectorConfig
from jmodt_amd.train_rows import BnFold, rpn_forward_rows, _image_pyramid
from tests.test_gpu_detector import make_engine
DEV = "cuda:0"
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
eng = make_engine(seed=5, cfg=DetectorConfig.survey()).to(DEV).eval()
xyz_h, img_h, xy_h = synth.frames(2, 16384, 4321, H=96, W=320, native=(94, 310))
xy_h = np.random.default_rng(5).uniform(-0.98, 0.98, size=xy_h.shape).astype(np.float32)
xyz, img, xy = T(xyz_h), T(img_h), T(xy_h)
net = eng.rpn.backbone_net
d = lambda a, b: float((a.double() - b.double()).abs().max())
with torch.no_grad():
    # 1. the module image blocks (operator route's forward): MIOpen NCHW
    runs = []
    for _ in range(6):
        x, maps = img, []
        for blk in net.Img_Block:
            x = blk(x); maps.append(x.clone())
        runs.append(maps)
    print("module image blocks, max |diff| vs call 1 per call:", [[f"{d(m, m0):.1e}" for m, m0 in zip(r, runs[0])] for r in runs[1:]])
    # 2. single MIOpen calls
    x = torch.randn(2, 64, 96, 320, device=DEV)
    w = torch.randn(64, 64, 3, 3, device=DEV) * 0.05
    ys = [F.conv2d(x, w, None, stride=2, padding=1) for _ in range(6)]
    print("F.conv2d stride 2 NCHW:", [f"{d(y, ys[0]):.1e}" for y in ys[1:]])
    xc, wc = x.contiguous(memory_format=torch.channels_last), w.contiguous(memory_format=torch.channels_last)
    ys = [F.conv2d(xc, wc, None, stride=2, padding=1) for _ in range(6)]
    print("F.conv2d stride 2 NHWC:", [f"{d(y, ys[0]):.1e}" for y in ys[1:]])
    wd = torch.randn(64, 16, 4, 4, device=DEV) * 0.05
    ys = [F.conv_transpose2d(x, wd, None, stride=4) for _ in range(6)]
    print("F.conv_transpose2d k4 NCHW:", [f"{d(y, ys[0]):.1e}" for y in ys[1:]])
    # 3. whole forwards
    fo = [train_joint.backbone_forward(net, xyz, img, xy).clone() for _ in range(6)]
    print("operator-route backbone features:", [f"{d(f, fo[0]):.1e}" for f in fo[1:]])
    fr = [rpn_forward_rows(eng, xyz, img, xy, BnFold(eng.rpn))["backbone_features"].clone() for _ in range(6)]
    print("rows-route backbone features:", [f"{d(f, fr[0]):.1e}" for f in fr[1:]])
    torch.cuda.synchronize()
