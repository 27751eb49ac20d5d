"""is the host ahead of the GPU?  per step: host time at which the step's enqueue started / ended vs the GPU time at which
the step started / ended (events on the main stream), all relative to the first step"""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
st = bench.make_detect_state(8, 1236, torch.device("cuda:0"))
st["engine"].prefetch_image = False
for _ in range(5): bench.detect_step(st)
torch.cuda.synchronize(); gc.disable()
K = 10
evs = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
host = []
t0 = time.perf_counter()
for k in range(K):
    evs[k].record()
    a = time.perf_counter()
    bench.detect_step(st)
    host.append((a - t0, time.perf_counter() - t0))
evs[K].record()
torch.cuda.synchronize()
for k in range(K):
    print(f"step {k}: host enqueue {host[k][0] * 1e3:7.2f} -> {host[k][1] * 1e3:7.2f} ms   GPU {evs[0].elapsed_time(evs[k]):7.2f} -> {evs[0].elapsed_time(evs[k + 1]):7.2f} ms")
