"""affinity GEMM variant timing (env vars are read once per process -> one subprocess per variant)"""
import os
import subprocess
import sys

CHILD = r'''
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(sys.argv[0]))) if False else os.getcwd())
from jmodt_amd import synth, _lib
from jmodt_amd.csrc import build as _hip_build
_lib.LIB_PATH = _hip_build.TOOLS_LIB   # the JM_* switches exist only in the tools build (python -m jmodt_amd.csrc.build --tools)
from jmodt_amd.ops.affinity import make_affinity_mlp, pairwise_affinity, mlp3_forward
torch.manual_seed(0)
link, se = make_affinity_mlp().cuda().eval(), make_affinity_mlp().cuda().eval()
def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
for P in (64, 128, 256):
    pf = torch.from_numpy(synth.roi_features(P, 512, 1)).cuda(); df = torch.from_numpy(synth.roi_features(P, 512, 2)).cuda()
    fl = P * P * (2 * 512 * 512 * 2 + 2 * 512)
    t_full = timeit(lambda: pairwise_affinity(pf, df, link, se))
    t_link = timeit(lambda: pairwise_affinity(pf, df, link, None))
    x = torch.randn(P * P, 512, device="cuda")
    t_mlp = timeit(lambda: mlp3_forward(x, link))
    print(f"  P=D={P}: full {t_full*1e3:7.1f} us ({fl/t_full/1e9:6.1f} TF)  link-only {t_link*1e3:7.1f} us ({fl/t_link/1e9:6.1f} TF)  plain-rows mlp {t_mlp*1e3:7.1f} us ({fl/t_mlp/1e9:6.1f} TF)")
'''
for small in (2048, 4096, 16384, 0):
    for pin in (1,):
        env = dict(os.environ, JM_GEMM_SMALL_M=str(small))
        print(f"SMALL_M={small}", flush=True)
        subprocess.run([sys.executable, "-c", CHILD], env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
