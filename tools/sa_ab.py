"""A/B two library builds on the fused SA kernel (RCNN SA1 shape): python tools/sa_ab.py libA.so libB.so"""
import os, subprocess, sys
CHILD = r'''
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from jmodt_amd import _lib
_lib.LIB_PATH = os.environ["JM_LIB"]
from jmodt_amd.ops.pointnet2.pointnet2_modules import PointnetSAModule
from jmodt_amd.ops.pointnet2 import pointnet2_utils as pu, fused
torch.manual_seed(0)
R = 1024
xyz = ((torch.rand(R, 512, 3) - 0.5) * torch.tensor([4.0, 2.0, 2.0])).cuda()
feat = torch.randn(R, 128, 512).cuda()
sa = PointnetSAModule(mlp=[128, 128, 128, 128], npoint=128, radius=0.2, nsample=64).cuda().eval()
with torch.no_grad():
    idx, new_xyz = pu.farthest_point_sample_xyz(xyz, 128)
    nb = pu.ball_query(0.2, 64, xyz, new_xyz)
    fn = lambda: fused.sa_mlp_fused(xyz, new_xyz, feat, nb, sa.mlps[0])
    ref = fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): fn()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
fl = R * 128 * 64 * 2 * (131 * 128 + 128 * 128 + 128 * 128)
print(f"{os.path.basename(os.environ['JM_LIB']):18s} {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TF   checksum {ref.double().sum().item():.6f}")
'''
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _ in range(2):
    for lib in sys.argv[1:]:
        subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, JM_LIB=os.path.abspath(lib)), cwd=root)
