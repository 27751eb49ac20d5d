"""determinism soak: the kernels with cross-workgroup or cross-wave hand-offs must return the same bits every time"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jmodt_amd import synth
from jmodt_amd.ops.pointnet2 import pointnet2_utils as pu, fused
from jmodt_amd.ops.pointnet2.pointnet2_modules import PointnetSAModule
from jmodt_amd.ops.proposal import distance_based_proposal
from jmodt_amd.ops.iou3d.iou3d_utils import nms_gpu
torch.manual_seed(0)
bad = 0
t = torch.from_numpy(synth.cloud(8, 65536, seed=3, dup_frac=0.1)).cuda()
ref = pu.farthest_point_sample(t, 2048)
for i in range(300):
    bad += int(not torch.equal(pu.farthest_point_sample(t, 2048), ref))
print("coop fps mismatches:", bad)
t2 = torch.from_numpy(synth.cloud(64, 16384, seed=4, dup_frac=0.1)).cuda()
ref = pu.farthest_point_sample(t2, 1024); b2 = 0
for i in range(100):
    b2 += int(not torch.equal(pu.farthest_point_sample(t2, 1024), ref))
print("fps 16384 mismatches:", b2)
R = 256
xyz = ((torch.rand(R, 512, 3) - 0.5) * torch.tensor([4.0, 2.0, 2.0])).cuda(); feat = torch.randn(R, 128, 512).cuda()
sa = PointnetSAModule(mlp=[128, 128, 128, 128], npoint=128, radius=0.2, nsample=64).cuda().eval()
with torch.no_grad():
    idx, nx = pu.farthest_point_sample_xyz(xyz, 128); nb = pu.ball_query(0.2, 64, xyz, nx)
    ref = fused.sa_mlp_fused(xyz, nx, feat, nb, sa.mlps[0]); b3 = 0
    for i in range(200):
        b3 += int(not torch.equal(fused.sa_mlp_fused(xyz, nx, feat, nb, sa.mlps[0]), ref))
print("fused SA mismatches:", b3)
rs, rp = synth.rpn_output(8, 16384, 9)
s, p = torch.from_numpy(rs).cuda(), torch.from_numpy(rp).cuda()
ref = distance_based_proposal(s, p, 9000, 100, 0.8); b4 = 0
for i in range(200):
    o = distance_based_proposal(s, p, 9000, 100, 0.8)
    b4 += int(not (torch.equal(o[0], ref[0]) and torch.equal(o[1], ref[1])))
print("proposal selection mismatches:", b4)
bb, ss = synth.bev_boxes(6300, 5)
bb, ss = torch.from_numpy(bb).cuda(), torch.from_numpy(ss).cuda()
ref = nms_gpu(bb, ss, 0.7); b5 = 0
for i in range(100):
    b5 += int(not torch.equal(nms_gpu(bb, ss, 0.7), ref))
print("rotated nms mismatches:", b5)
print("SOAK_OK" if bad + b2 + b3 + b4 + b5 == 0 else "SOAK_FAIL")
