"""rcnn_lift with / without the copy-tile skipping at the detector's shape (1024 RoIs x 512 points x 133 channels), HIP-event
timing of the bare entry:   python tools/lift_skip_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jmodt_amd.ops.rcnn_lift import PackedRcnnLift  # noqa: E402

dev = "cuda:0"
g = torch.Generator().manual_seed(0)
mk = lambda o, i: (torch.randn(o, i, generator=g) * (2.0 / i) ** 0.5).to(dev)   # noqa: E731
bias = lambda o: (torch.randn(o, generator=g) * 0.1).to(dev)                    # noqa: E731
lift = PackedRcnnLift([(mk(128, 5), bias(128)), (mk(128, 128), bias(128))], (mk(128, 256), bias(128)), (mk(128, 131), bias(128)))
x = torch.randn(1024, 512, 133, generator=g).to(dev)


def bench(count):
    for _ in range(3):
        lift(x, point_major=True, count=count)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        lift(x, point_major=True, count=count)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / 20 * 1e3


print("no count        : %.1f us" % bench(None))
for c in (512, 128, 33, 11, 1):
    print("count = %-4d    : %.1f us" % (c, bench(torch.full((1024,), c, dtype=torch.int32, device=dev))))
