"""is the fused SA kernel gather-bound?  RCNN SA1 shape (1024 RoIs x 512 pts -> 128 centres x 64 samples, 128->128->128
after the hoisted first layer) with (a) real ball-query neighbour lists, (b) contiguous index runs, (c) one index"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jmodt_amd import _lib
from jmodt_amd.csrc import build as _hip_build
_lib.LIB_PATH = _hip_build.TOOLS_LIB
from jmodt_amd.ops.pointnet2 import fused, pointnet2_utils as pu
from jmodt_amd.ops.pointnet2.pointnet2_modules import PointnetSAModule

def timeit(fn, iters=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

torch.manual_seed(0)
R, N, M, ns = 1024, 512, 128, 64
sa = PointnetSAModule(mlp=[128, 128, 128, 128], npoint=M, radius=0.2, nsample=ns, bn=False).cuda().eval()
xyz = (torch.rand(R, N, 3, device="cuda") - 0.5) * torch.tensor([4.0, 2.0, 2.0], device="cuda")
u = torch.randn(R, 128, N, device="cuda")
_, new_xyz = pu.farthest_point_sample_xyz(xyz, M)
real = pu.ball_query(0.2, ns, xyz, new_xyz)
ar = torch.arange(M * ns, device="cuda", dtype=torch.int32).view(1, M, ns)
runs = ((ar // ns * 4 + ar % ns) % N).expand(R, -1, -1).contiguous()          # 64 consecutive points per centre
one = torch.zeros_like(real)
rand = torch.randint(0, N, (R, M, ns), device="cuda", dtype=torch.int32)
fl = 2 * R * M * ns * (128 * 128 * 2)

for name, idx in (("ball_query", real), ("single", one)):
    ms = timeit(lambda: fused.sa_mlp_pre_from_u(u, new_xyz, idx, sa.mlps[0]))
    print(f"{name:12s} {ms:.3f} ms  executed {fl / ms / 1e9:.1f} TF", flush=True)
feats = torch.randn(R, 128, N, device="cuda")
fused.PRE_PROJECT = False
ms = timeit(lambda: fused.sa_mlp_fused(xyz, new_xyz, feats, real, sa.mlps[0]))
print(f"row-wise first layer (no hoist): {ms:.3f} ms  {2 * R * M * ns * (131 * 128 + 2 * 128 * 128) / ms / 1e9:.1f} TF")
