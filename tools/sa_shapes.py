"""fused SA kernel on every eligible SA level of the reference config (RPN levels 1-2, RCNN levels 1-2)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jmodt_amd.ops.pointnet2.pointnet2_modules import PointnetSAModuleMSG
from jmodt_amd.ops.pointnet2 import pointnet2_utils as pu, fused
torch.manual_seed(0)
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
cases = [  # name, B, N, C, npoint, radii, nsamples, mlps (config.py:75-82, 134-139)
    ("RPN SA1", 8, 16384, 0, 4096, [0.1, 0.5], [16, 32], [[0, 16, 16, 32], [0, 32, 32, 64]]),
    ("RPN SA2", 8, 4096, 96, 1024, [0.5, 1.0], [16, 32], [[96, 64, 64, 128], [96, 64, 96, 128]]),
    ("RPN SA3", 8, 1024, 256, 256, [1.0, 2.0], [16, 32], [[256, 128, 196, 256], [256, 128, 196, 256]]),
    ("RPN SA4", 8, 256, 512, 64, [2.0, 4.0], [16, 32], [[512, 256, 256, 512], [512, 256, 384, 512]]),
    ("RCNN SA1", 1024, 512, 128, 128, [0.2], [64], [[128, 128, 128, 128]]),
    ("RCNN SA2", 1024, 128, 128, 32, [0.4], [64], [[128, 128, 128, 256]]),
]
for name, B, N, C, npoint, radii, nsamples, mlps in cases:
    sa = PointnetSAModuleMSG(npoint=npoint, radii=radii, nsamples=nsamples, mlps=[list(m) for m in mlps]).cuda().eval()
    xyz = (torch.rand(B, N, 3) * torch.tensor([4.0, 2.0, 2.0]) * (8.0 if B == 8 else 1.0)).cuda()
    feat = torch.randn(B, C, N).cuda() if C else None
    with torch.no_grad():
        idx = pu.farthest_point_sample(xyz, npoint)
        new_xyz = pu.gather_operation(xyz.transpose(1, 2).contiguous(), idx).transpose(1, 2).contiguous()
        for r, ns, spec, mlp in zip(radii, nsamples, mlps, sa.mlps):
            nb = pu.ball_query(r, ns, xyz, new_xyz)
            w = [3 + C] + spec[1:]
            fl = B * npoint * ns * 2 * sum(a * b for a, b in zip(w[:-1], w[1:]))
            ms = timeit(lambda: fused.sa_mlp_fused(xyz, new_xyz, feat, nb, mlp))
            grouped = B * (3 + C) * npoint * ns * 4
            print(f"{name:9s} r={r:<4} ns={ns:<3} widths={w}: {ms:7.3f} ms  {fl / ms / 1e9:6.1f} TF  (grouped tensor it replaces: {grouped / 1e6:7.1f} MB)")
