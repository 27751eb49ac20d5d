"""9 composed steps with the per-entry profiler OFF (no event records besides the engine's own stream dependencies):
the workload tools/timeline.sh traces"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
st = bench.make_detect_state(8, 1236, torch.device("cuda:0"))
st["engine"].prefetch_image = False
for _ in range(9):
    bench.detect_step(st)
torch.cuda.synchronize()
