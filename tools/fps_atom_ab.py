"""fps_regs2_kernel with the cross-wave arg-max through one LDS atomic per wave (JM_FPS_ATOM=1, tools build) against the 16-word
exchange + second DPP reduction: time per iteration at the pyramid's shapes and bit-exactness against the oracle.
Usage: python tools/fps_atom_ab.py"""
import os
import subprocess
import sys

CHILD = r'''
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from jmodt_amd import synth, _lib
from jmodt_amd.csrc import build as _hip_build
_lib.LIB_PATH = _hip_build.TOOLS_LIB
from jmodt_amd.ops.pointnet2.pointnet2_utils import farthest_point_sample
from oracle import oracle
def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
out = []
for B, n, m in [(8, 16384, 4096), (8, 4096, 1024), (8, 1024, 256)]:
    xyz = torch.from_numpy(synth.cloud(B, n, seed=3)).cuda()
    ms = timeit(lambda: farthest_point_sample(xyz, m))
    out.append(f"{n}->{m}: {ms:7.3f} ms ({ms / m * 1e3:.3f} us/it)")
ok = []
for kind in ("cloud", "dup", "grid"):
    x = synth.cloud(2, 4096, seed=5)
    if kind == "dup": x[:, 2048:] = x[:, :2048]
    if kind == "grid": x = np.round(x * 2) / 2
    ok.append(bool(np.array_equal(farthest_point_sample(torch.from_numpy(x).cuda(), 512).cpu().numpy(), oracle.furthest_point_sample(x, 512))))
print("JM_FPS_ATOM=" + os.environ.get("JM_FPS_ATOM", "0"), " | ".join(out), "| bit-exact vs oracle (random / duplicated / lattice):", ok)
'''
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _ in range(1):
    for v, extra in (("0", {}), ("1", {}), ("0", {"JM_FPS_V1": "1"})):
        print(extra or "", end=" ", flush=True)
        subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, JM_FPS_ATOM=v, **extra), cwd=root)
