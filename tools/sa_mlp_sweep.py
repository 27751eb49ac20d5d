"""fused SA kernel timing on the RCNN SA1 shape (1024 RoIs x 512 pts x 128 ch, 128 centres, ns 64)
    R=<rois> python tools/sa_mlp_sweep.py"""
import os
import subprocess
import sys

CHILD = r'''
import os, sys, torch
sys.path.insert(0, os.getcwd())
from jmodt_amd.ops.pointnet2.pointnet2_modules import PointnetSAModule
from jmodt_amd.ops.pointnet2 import pointnet2_utils as pu, fused
torch.manual_seed(0)
R = int(os.environ.get("R", 1024))
xyz = ((torch.rand(R, 512, 3) - 0.5) * torch.tensor([4.0, 2.0, 2.0])).cuda()
feat = torch.randn(R, 128, 512).cuda()
sa = PointnetSAModule(mlp=[128, 128, 128, 128], npoint=128, radius=0.2, nsample=64).cuda().eval()
with torch.no_grad():
    idx = pu.farthest_point_sample(xyz, 128)
    new_xyz = pu.gather_operation(xyz.transpose(1, 2).contiguous(), idx).transpose(1, 2).contiguous()
    nb = pu.ball_query(0.2, 64, xyz, new_xyz)
    fn = lambda: fused.sa_mlp_fused(xyz, new_xyz, feat, nb, sa.mlps[0])
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): fn()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
fl = R * 128 * 64 * 2 * (131 * 128 + 128 * 128 + 128 * 128)
print(f"  R={R}: {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TF  ({fl / ms / 1e9 / 157.3 * 100:.1f}% of fp32 MFMA peak)")
'''
subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ), cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
