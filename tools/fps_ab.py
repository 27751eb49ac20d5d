"""A/B two builds of the library on FPS: python tools/fps_ab.py libA.so libB.so  (one subprocess per build)"""
import os, subprocess, sys
CHILD = r'''
import os, sys, shutil
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from jmodt_amd import synth, _lib
_lib.LIB_PATH = os.environ["JM_LIB"]
from jmodt_amd.ops.pointnet2.pointnet2_utils import farthest_point_sample
def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
out = []
for B, n, m in [(8, 16384, 4096), (8, 4096, 1024), (8, 1024, 256), (8, 65536, 4096)]:
    xyz = torch.from_numpy(synth.cloud(B, n, seed=3)).cuda()
    ms = timeit(lambda: farthest_point_sample(xyz, m))
    out.append(f"{n}->{m}: {ms:7.3f} ms ({ms / m * 1e3:.3f} us/it)")
print(os.path.basename(os.environ["JM_LIB"]), " | ".join(out))
'''
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _ in range(2):
    for lib in sys.argv[1:]:
        subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, JM_LIB=os.path.abspath(lib)), cwd=root)
