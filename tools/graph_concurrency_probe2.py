"""a LONG-running small graph (one workgroup spinning through a dependent chain, like the FPS pyramid) on stream F, then a short
graph / eager kernels on stream I launched right behind it: when does the second one finish?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device("cuda:0")
from jmodt_amd.ops.pointnet2 import pointnet2_utils
xyz = torch.rand(4, 16384, 3, device=dev) * 40
y = torch.zeros(1 << 20, device=dev)
def long_fn():
    pointnet2_utils.farthest_point_sample_xyz(xyz, 4096)          # ~5 ms on 4 workgroups
def short_fn():
    for _ in range(20):
        y.mul_(1.0001).add_(1.0)
cap = torch.cuda.Stream()
def capture(fn):
    cap.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cap):
        fn(); fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=cap):
        fn()
    torch.cuda.synchronize()
    return g
gl, gs = capture(long_fn), capture(short_fn)
F, I = torch.cuda.Stream(), torch.cuda.Stream()
def run(first, second):
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    F.wait_stream(torch.cuda.current_stream()); I.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(F):
        first(); e1.record(F)
    with torch.cuda.stream(I):
        second(); e2.record(I)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1), e0.elapsed_time(e2)
for name, a, b in (("graph long, graph short", gl.replay, gs.replay), ("graph long, eager short", gl.replay, short_fn),
                   ("eager long, graph short", long_fn, gs.replay), ("eager long, eager short", long_fn, short_fn)):
    run(a, b)
    t = [run(a, b) for _ in range(5)]
    print(f"{name:26s}: long done at {sum(x[0] for x in t) / 5:.2f} ms, short done at {sum(x[1] for x in t) / 5:.2f} ms", flush=True)
