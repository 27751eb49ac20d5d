"""what does a second FPS pyramid in flight do to the 4-frame (train) / 8-frame (detect) step?  ms per step over 30 plain steps with
parts of the pyramid's side-stream work switched off (probe variants only, not product settings).
    gpurun -- 'python tools/prefetch_depth_probe.py [frames]'"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import jmodt_amd.detector as det
from jmodt_amd.ops.pointnet2 import pyramid

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 4
mode = sys.argv[2] if len(sys.argv) > 2 else "detect"          # detect | train (bench.train_step: eng.detect + the finetune step)
dev = torch.device("cuda:0")
st = bench.make_train_state(frames, 1237, dev) if mode == "train" else bench.make_detect_state(frames, 1236, dev)
eng = st["engine"]
step = (lambda: bench.train_step(st, 1)) if mode == "train" else (lambda: bench.detect_step(st))
REAL = pyramid.FpsPyramid


def run(tag, depth, n=30, **patch):
    class Patched(REAL):
        def __init__(self, xyz, npoints, overlap=True, with_interp=False, grid_radii=None, slot=0):
            if patch.get("no_interp"):
                with_interp = False
            if patch.get("no_grid"):
                grid_radii = None
            if patch.get("slots"):
                slot = patch["slots"][Patched.n % len(patch["slots"])]
                Patched.n += 1
            super().__init__(xyz, npoints, overlap=overlap, with_interp=with_interp, grid_radii=grid_radii, slot=slot)

        def release(self):
            if patch.get("no_release_wait"):
                self._levels, self._interp, self._xyz, self._grids = [], [], None, []
            else:
                super().release()
    Patched.n = 0
    det.FpsPyramid = Patched
    eng.prefetch_depth = depth
    for _ in range(6):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    print(f"{tag:44s} depth {depth}: {ms:7.3f} ms/step  {frames / ms * 1e3:7.1f} frames/s", flush=True)
    with torch.no_grad():
        eng(st["xyz"], st["image"], st["pts_xy"])      # drain the announcements
        eng(st["xyz"], st["image"], st["pts_xy"])
    torch.cuda.synchronize()
    det.FpsPyramid = REAL


run("baseline", 1)
run("two in flight", 2)
run("two in flight, no 3-NN on the side stream", 2, no_interp=True)
run("two in flight, no grid build on the side stream", 2, no_grid=True)
run("two in flight, neither", 2, no_interp=True, no_grid=True)
run("two in flight, release without the stream wait", 2, no_release_wait=True)
run("two in flight, slots 4 / 5", 2, slots=(4, 5))
run("three in flight", 3)
run("baseline again", 1)
