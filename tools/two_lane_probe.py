"""would TWO batches in flight (two engines sharing nothing but the GPU, each on a main stream and side streams of its own) raise the
throughput of the composed step?  The 4- vs 8-frame steps (6.6 / 10.0 ms) say a step has ~3 ms that does not scale with the batch: small
launches that leave the machine idle.     gpurun -- 'python tools/two_lane_probe.py [frames] [lanes]'"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import jmodt_amd.detector as det
from jmodt_amd.ops.pointnet2 import pyramid
from jmodt_amd.ops import affinity

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 8
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda:0")
LANE = [0]
real = pyramid.side_stream


def side(device, slot=0):
    return real(device, slot + 8 * LANE[0])


for mod in (pyramid, det, affinity):
    mod.side_stream = side
states = [bench.make_detect_state(frames, 1236, dev) for _ in range(lanes)]
mains = [torch.cuda.Stream(device=dev) for _ in range(lanes)]


def steps(n, use):
    for i in range(n):
        k = i % use
        LANE[0] = k
        with torch.cuda.stream(mains[k]):
            bench.detect_step(states[k])


for use in (1, lanes, 1, lanes):
    steps(6 * use, use)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 40
    steps(n, use)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    print(f"{use} lane(s): {ms:7.3f} ms per batch of {frames}  {frames / ms * 1e3:7.1f} frames/s", flush=True)
