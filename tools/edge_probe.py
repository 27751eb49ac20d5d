"""odd shapes through the composed engine: the 100 RoIs per frame of the reference configuration, one frame, point counts
that are no power of two; optimised (side streams, duplicate compaction) against plain.
Usage: PYTHONPATH=. python tools/edge_probe.py"""
import numpy as np, torch, dataclasses
from jmodt_amd import synth
from jmodt_amd.detector import DetectAffinityEngine, DetectorConfig
from oracle.pipeline import Chain
dev='cuda:0'
torch.manual_seed(3)
def T(a): return torch.from_numpy(np.ascontiguousarray(a)).to(dev)
for cfg, B, N in ((DetectorConfig(), 3, 16384), (DetectorConfig(), 1, 16000), (dataclasses.replace(DetectorConfig(), rpn_post_nms_top_n=37), 2, 20000)):
    eng = DetectAffinityEngine(cfg).to(dev).eval()
    xyz, img, xy = synth.frames(B, N, 77)
    with torch.no_grad():
        cache, aff, inter = eng(T(xyz), T(img), T(xy))
        torch.cuda.synchronize()
        eng.overlap = False; eng.dedupe_rcnn = False
        c2, a2, i2 = eng(T(xyz), T(img), T(xy))
    ok = {k: float((inter[k] - i2[k]).abs().max()) for k in ("backbone_features", "rois", "pts_input", "rcnn_feat")}
    ok["A"] = float((aff[0][0] - a2[0][0]).abs().max())
    print(cfg.rpn_post_nms_top_n, B, N, "rois", tuple(inter["rois"].shape), "A", tuple(aff[0][0].shape), "det", cache.count.tolist(), "max |optimised - plain| (MIOpen convolutions differ run to run at 1e-8)", ok, "finite", bool(torch.isfinite(aff[-1][0]).all()))
