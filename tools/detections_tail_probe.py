"""is the detections' side stream (RCNN heads, box decode, score filter, NMS, gathers: ~35 small launches under the affinity GEMMs) the
tail of the step?  The same steps with those launches replaced by their cached results = the most a shorter chain could buy.
    gpurun -- 'python tools/detections_tail_probe.py'"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda:0")
st = bench.make_detect_state(8, 1236, dev)
eng = st["engine"]


def run(tag, n=40):
    for _ in range(6):
        bench.detect_step(st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        bench.detect_step(st)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    print(f"{tag:60s} {ms:7.3f} ms/step  {8 / ms * 1e3:7.1f} frames/s", flush=True)


run("as shipped")
real_det, real_heads = eng._detections, eng.rcnn_heads
with torch.no_grad():
    _, _, inter = eng(st["xyz"], st["image"], st["pts_xy"])
    heads_out = real_heads(inter["rcnn_feat"])
    det_out = real_det(inter["rois"], dict(inter, **heads_out))
torch.cuda.synchronize()
eng._detections = lambda rois, out: det_out
run("decode / filter / NMS / gathers cached (heads still run)")
eng.rcnn_heads = lambda feat: heads_out
run("... and the RCNN heads cached")
eng._detections, eng.rcnn_heads = real_det, real_heads
run("as shipped")
