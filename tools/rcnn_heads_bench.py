"""RCNN cls / reg heads (rcnn.py:57-89: 512 -> 256 -> 256 -> 1 / 46 on 1024 RoIs): six jm_linear_rows launches vs one conv1d_stack
launch per head (+ the shared transpose), GPU time from HIP-graph replays.    gpurun -- 'python tools/rcnn_heads_bench.py'"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
from jmodt_amd.detector import DetectAffinityEngine, DetectorConfig
from rpn_listed_bench_timeit import timeit

eng = DetectAffinityEngine(DetectorConfig.survey()).to("cuda:0").eval()
for R in (1024, 2048, 800):
    x = torch.relu(torch.randn(R, 512, 1, device="cuda:0"))
    with torch.no_grad():
        eng.fuse_head_stacks = True
        a = eng.rcnn_heads(x)
        t1 = timeit(lambda: eng.rcnn_heads(x))
        eng.fuse_head_stacks = False
        b = eng.rcnn_heads(x)
        t0 = timeit(lambda: eng.rcnn_heads(x))
    err = max((a[k] - b[k]).abs().max().item() for k in a)
    print(f"R = {R}: linear_rows x 6 {t0:7.1f} us   stacks {t1:7.1f} us   max |diff| {err:.2e}", flush=True)
