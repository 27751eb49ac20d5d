"""time the RCNN SA1 shape of the fused SA kernel (hoisted first layer) with one tools-library variant
(JM_TOOLS_LIB=tools/bin/libjmodt_hip_tools_exp<N>.so, see tools/sa_exp.sh): which operand feed costs what"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jmodt_amd import _lib
from jmodt_amd.csrc import build as _hip_build
_lib.LIB_PATH = os.environ.get("JM_TOOLS_LIB", _hip_build.TOOLS_LIB)
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
feats = torch.randn(R, 128, N, device="cuda")
_, new_xyz = pu.farthest_point_sample_xyz(xyz, M)
idx = pu.ball_query(0.2, ns, xyz, new_xyz)
tag = os.path.basename(_lib.LIB_PATH)
ms = timeit(lambda: fused.sa_mlp_pre_from_u(u, new_xyz, idx, sa.mlps[0]))
mf = R * M * ns / 128 / 256 * 512          # MFMAs per wave
print(f"{tag:36s} hoisted {ms:.3f} ms = {ms * 1e-3 * 2.4e9 / mf:.1f} cycles@2.4GHz per MFMA (incl. epilogues)", flush=True)
u_pm = u.transpose(1, 2).contiguous()
ms = timeit(lambda: fused.sa_mlp_pre_from_u(u_pm, new_xyz, idx, sa.mlps[0], point_major=True))
print(f"{tag:36s} point-major kernel {ms:.3f} ms = {ms * 1e-3 * 2.4e9 / mf:.1f} cycles@2.4GHz per MFMA-slot  ({2 * R * M * ns * 2 * 128 * 128 / ms / 1e9:.1f} TF executed)", flush=True)
a = fused.sa_mlp_pre_from_u(u_pm, new_xyz, idx, sa.mlps[0], point_major=True); b = fused.sa_mlp_pre_from_u(u, new_xyz, idx, sa.mlps[0])
print("   max |pm - k-major| =", (a - b).abs().max().item(), " max |out| =", b.abs().max().item())
fused.PRE_PROJECT = False
ms = timeit(lambda: fused.sa_mlp_fused(xyz, new_xyz, feats, idx, sa.mlps[0]))
print(f"{tag:36s} row-wise {ms:.3f} ms = {ms * 1e-3 * 2.4e9 / (mf / 512 * 800):.1f} cycles@2.4GHz per MFMA", flush=True)
