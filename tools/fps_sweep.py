"""FPS geometry sweep (points per thread) on the GPU box: JM_FPS_PTS is read per call."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jmodt_amd import synth, _lib
from jmodt_amd.csrc import build as _hip_build
_lib.LIB_PATH = _hip_build.TOOLS_LIB   # the JM_* switches exist only in the tools build (python -m jmodt_amd.csrc.build --tools)
from jmodt_amd.ops.pointnet2.pointnet2_utils import farthest_point_sample


def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


for B, n, m in [(8, 16384, 4096), (8, 4096, 1024), (8, 1024, 256), (8, 256, 64), (1024, 512, 128), (1024, 128, 32), (256, 16384, 4096)]:
    xyz = torch.from_numpy(synth.cloud(B, n, seed=3)).cuda()
    ref = None
    row = []
    for pts in (1, 2, 4, 8, 16, 32, 64):
        os.environ["JM_FPS_PTS"] = str(pts)
        try:
            ms = timeit(lambda: farthest_point_sample(xyz, m))
            out = farthest_point_sample(xyz, m)
            if ref is None:
                ref = out
            ok = bool(torch.equal(out, ref))
            row.append(f"pts{pts}: {ms:8.3f} ms ({ms * 1e3 / (m - 1):6.3f} us/it){'' if ok else ' MISMATCH'}")
        except Exception as ex:
            row.append(f"pts{pts}: ERR {str(ex)[:40]}")
    print(f"B={B} n={n} m={m}\n   " + "\n   ".join(row))
