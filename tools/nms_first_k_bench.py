"""jm_nms_normal_first_k_batched (lazy greedy, stops at K survivors) vs jm_nms_batched (pair mask + reduce) on 16 problems of
6300 / 2700 boxes (the RPN's two depth bands x 8 frames), for clouds of boxes with many / few survivors.
Usage: PYTHONPATH=. python tools/nms_first_k_bench.py"""
import numpy as np
import torch

from jmodt_amd import synth
from jmodt_amd.ext import iou3d_cuda

dev = torch.device("cuda:0")


def timeit(f, reps=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for kind in ("clustered", "piled40", "piled400", "sparse"):
    counts = [6300, 2700] * 8
    nmax = 6300
    boxes = np.zeros((16, nmax, 5), np.float32)
    rng = np.random.default_rng(5)
    for p, c in enumerate(counts):
        if kind == "clustered":
            b, s = synth.bev_boxes(c, 70 + p)
        elif kind.startswith("piled"):
            per = int(kind[5:])
            b, s = synth.bev_boxes(c, 80 + p, jitter_clusters=False)
            b = b[rng.integers(0, max(1, c // per), c)] + rng.normal(0, 0.02, (c, 5)).astype(np.float32)
        else:
            b, s = synth.bev_boxes(c, 90 + p, extent=2000.0)
        boxes[p, :c] = b[np.argsort(-s, kind="stable")][:c]
    tb = torch.from_numpy(boxes).to(dev)
    tc = torch.tensor(counts, dtype=torch.int32, device=dev)
    for thr in (0.8, 0.85):
        full = timeit(lambda: iou3d_cuda.nms_batched_device(tb, tc, thr, 1))
        _, fn = iou3d_cuda.nms_batched_device(tb, tc, thr, 1)
        for k in (89, 358):
            t = timeit(lambda: iou3d_cuda.nms_normal_first_k_device(tb, tc, thr, k))
            print(f"{kind:10s} thr {thr}: survivors/problem {fn.float().mean().item():7.1f} | mask+reduce {full:7.1f} us | first {k:3d}: {t:7.1f} us")
