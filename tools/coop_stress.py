"""co-operative FPS on a side stream while a persistent, CU-filling kernel runs on the main stream:
the FPS workgroups may be scheduled late or piecemeal; results must still be exact and nothing may hang"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from jmodt_amd import synth
from jmodt_amd.ops.pointnet2 import pointnet2_utils as pu, fused
from jmodt_amd.ops.pointnet2.pointnet2_modules import PointnetSAModule
from oracle import oracle
torch.manual_seed(0)
R = 1024
rx = ((torch.rand(R, 512, 3) - 0.5) * torch.tensor([4.0, 2.0, 2.0])).cuda()
rf = torch.randn(R, 128, 512).cuda()
sa = PointnetSAModule(mlp=[128, 128, 128, 128], npoint=128, radius=0.2, nsample=64).cuda().eval()
with torch.no_grad():
    ci = pu.farthest_point_sample(rx, 128)
    cx = pu.gather_operation(rx.transpose(1, 2).contiguous(), ci).transpose(1, 2).contiguous()
    nb = pu.ball_query(0.2, 64, rx, cx)
xyz = synth.cloud(8, 65536, seed=5)
want = oracle.furthest_point_sample(xyz, 512)
t = torch.from_numpy(xyz).cuda()
side = torch.cuda.Stream()
ok = True
for rep in range(6):
    with torch.no_grad():
        for _ in range(3): fused.sa_mlp_fused(rx, cx, rf, nb, sa.mlps[0])       # ~24 ms of a grid that fills every CU
        side.wait_stream(torch.cuda.current_stream()) if rep % 2 else None
        with torch.cuda.stream(side):
            got = pu.farthest_point_sample(t, 512)
        for _ in range(3): fused.sa_mlp_fused(rx, cx, rf, nb, sa.mlps[0])
    torch.cuda.synchronize()
    ok &= bool(np.array_equal(got.cpu().numpy(), want))
print("COOP_STRESS_OK" if ok else "COOP_STRESS_MISMATCH")
