"""does the COMPOSED engine run BASELINE configs[4] (65536 points per frame, 256 proposals, 256^2 affinity) as is?
Usage: PYTHONPATH=. python tools/dense_engine_probe.py [frames]"""
import dataclasses
import sys
import time

import torch

from jmodt_amd import synth
from jmodt_amd.detector import DetectAffinityEngine, DetectorConfig

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
torch.manual_seed(5)
cfg = dataclasses.replace(DetectorConfig.survey(), rpn_post_nms_top_n=256)
eng = DetectAffinityEngine(cfg).to(dev)
xyz, img, xy = synth.frames(B, 65536, 1238)
st = dict(xyz=torch.from_numpy(xyz).to(dev), image=torch.from_numpy(img).to(dev), pts_xy=torch.from_numpy(xy).to(dev))
for i in range(3):
    cache, aff, inter = eng(st["xyz"], st["image"], st["pts_xy"], next_xyz=st["xyz"])
torch.cuda.synchronize()
print("rois", tuple(inter["rois"].shape), "affinity", tuple(aff[0][0].shape), "detections per frame", cache.count.tolist())
t0 = time.perf_counter()
n = 10
for i in range(n):
    eng(st["xyz"], st["image"], st["pts_xy"], next_xyz=st["xyz"])
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
print(f"{B} frames x 65536 points, 256 RoIs: {ms:.2f} ms per step = {B / ms * 1e3:.1f} frames/s")
