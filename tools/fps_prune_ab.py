"""JM_FPS_PRUNE=0/1/2 timing on the bench cloud and a 1/z-dense one (one subprocess per setting)"""
import os, subprocess, sys
CHILD = r'''
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from jmodt_amd import synth, _lib
from jmodt_amd.csrc import build as _hip_build
_lib.LIB_PATH = _hip_build.TOOLS_LIB   # the JM_* switches exist only in the tools build (python -m jmodt_amd.csrc.build --tools)
from jmodt_amd.ops.pointnet2.pointnet2_utils import farthest_point_sample
def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
out = []
for name, xyz in (("uniform", synth.cloud(8, 16384, seed=1235)), ("dense", synth.dense_cloud(8, 16384, 5)), ("dup10%", synth.cloud(8, 16384, seed=7, dup_frac=0.1)), ("n8192", synth.cloud(8, 8192, seed=3))):
    t = torch.from_numpy(xyz).cuda()
    m = t.shape[1] // 4
    ms = timeit(lambda: farthest_point_sample(t, m))
    out.append(f"{name}: {ms:6.3f} ms ({ms / m * 1e3:.3f} us/it)")
print("JM_FPS_PRUNE=" + os.environ.get("JM_FPS_PRUNE", "0"), " | ".join(out))
'''
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for v in ("0", "1", "2", "0", "2"):
    subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, JM_FPS_PRUNE=v), cwd=root)
