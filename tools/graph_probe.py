"""does capturing the whole composed step in a HIP graph (torch.cuda.graph) shorten it?  eager vs graph replay"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda:0")
st = bench.make_detect_state(8, 1236, dev)
eng = st["engine"]
xyz, img, xy = st["xyz"], st["image"], st["pts_xy"]

def run(n, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

eager_pf = lambda: eng(xyz, img, xy, next_xyz=xyz, next_image=None)
eager = lambda: eng(xyz, img, xy)
for _ in range(4): eager_pf()
print(f"eager, FPS prefetch      {run(12, eager_pf):.3f} ms/step", flush=True)
for _ in range(3): eager()
print(f"eager, no prefetch       {run(12, eager):.3f} ms/step", flush=True)
for overlap in (True, False):
    eng.overlap = overlap
    for _ in range(3): eager()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    try:
        with torch.cuda.stream(s):
            for _ in range(2): eager()
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                out = eager()
        torch.cuda.synchronize()
        g.replay(); torch.cuda.synchronize()
        print(f"graph replay, overlap={overlap}  {run(12, g.replay):.3f} ms/step", flush=True)
    except Exception as e:
        print(f"capture failed (overlap={overlap}): {type(e).__name__}: {str(e)[:300]}", flush=True)
        torch.cuda.synchronize()
