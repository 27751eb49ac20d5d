"""Which stage of the full-size backbone differs between two forwards of the same engine on the same resident batch?
Wraps the stage-level Python entry points, records every output of two consecutive forwards and prints the first call whose
outputs are not bit-identical.    python tools/determinism_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jmodt_amd import synth  # noqa: E402
from jmodt_amd.detector import DetectAffinityEngine, DetectorConfig  # noqa: E402
from jmodt_amd.ops import fusion  # noqa: E402
from jmodt_amd.ops.pointnet2 import fused, pointnet2_modules, pointnet2_utils  # noqa: E402
import jmodt_amd.detector as det  # noqa: E402

dev = "cuda:0"
torch.manual_seed(5)
eng = DetectAffinityEngine(DetectorConfig.survey()).to(dev)
xyz, img, xy = synth.frames(2, 16384, 99)
a = [torch.from_numpy(t).to(dev) for t in (xyz, img, xy)]
log = []


def wrap(mod, name):
    fn = getattr(mod, name)

    def inner(*args, **kw):
        out = fn(*args, **kw)
        outs = out if isinstance(out, (tuple, list)) else (out,)
        log.append((f"{getattr(mod, '__name__', type(mod).__name__)}.{name}", [o.detach().clone() for o in outs if isinstance(o, torch.Tensor)]))
        return out
    setattr(mod, name, inner)


wrap(pointnet2_utils, "ball_query_dual")
wrap(pointnet2_utils, "ball_query")
wrap(pointnet2_utils, "three_interpolate")
wrap(fused, "sa_mlp_fused")
wrap(fused, "shared_mlp_points")
wrap(det, "feature_gather")
wrap(eng, "_attention_fusion")
wrap(eng, "_image_block")
wrap(eng, "_rpn_heads_stack")
with torch.no_grad():
    eng(*a)
    torch.cuda.synchronize()
    log.clear()
    eng(*a)
    torch.cuda.synchronize()
    first = list(log)
    log.clear()
    eng(*a)
    torch.cuda.synchronize()
    second = list(log)
print(len(first), len(second))
for i, ((n0, o0), (n1, o1)) in enumerate(zip(first, second)):
    same = n0 == n1 and len(o0) == len(o1) and all(torch.equal(x, y) for x, y in zip(o0, o1))
    if not same:
        d = max(float((x.float() - y.float()).abs().max()) for x, y in zip(o0, o1)) if n0 == n1 else -1
        print("DIFF", i, n0, n1, d)
print("done")
