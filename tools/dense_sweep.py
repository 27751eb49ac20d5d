"""config 5 (dense scene: 65536 pts/frame, 256 proposals, 256^2 affinity) op timings, B frames per GPU"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jmodt_amd import synth
from jmodt_amd.ops.pointnet2 import pointnet2_utils as pu
from jmodt_amd.ops.roipool3d.roipool3d_utils import roipool3d_gpu
from jmodt_amd.ops.affinity import make_affinity_mlp, pairwise_affinity

B = int(os.environ.get("B", 8))
N = 65536
def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
xyz_np = synth.cloud(B, N, seed=7)
xyz = torch.from_numpy(xyz_np).cuda()
for m in (4096, 16384):
    print(f"fps {N}->{m}: {timeit(lambda: pu.farthest_point_sample(xyz, m), 2):9.3f} ms")
idx = pu.farthest_point_sample(xyz, 4096)
new_xyz = pu.gather_operation(xyz.transpose(1, 2).contiguous(), idx).transpose(1, 2).contiguous()
print(f"ball_query_dual L1: {timeit(lambda: pu.ball_query_dual(0.1, 16, 0.5, 32, xyz, new_xyz)):9.3f} ms")
print(f"three_nn {N} x 4096: {timeit(lambda: pu.three_nn(xyz, new_xyz)):9.3f} ms")
feat = torch.randn(B, N, 130, device="cuda")
boxes = torch.from_numpy(synth.proposals(xyz_np, 256, 9)).cuda()
ms = timeit(lambda: roipool3d_gpu(xyz, feat, boxes, 0.2, 512))
print(f"roipool3d N={N} M=256: {ms:9.3f} ms  ({(B*(12*N+28*256+4*130*N)+B*256*512*133*4)/ms/1e6:7.1f} GB/s)")
torch.manual_seed(0)
link, se = make_affinity_mlp().cuda().eval(), make_affinity_mlp().cuda().eval()
pf = torch.from_numpy(synth.roi_features(256, 512, 1)).cuda(); df = torch.from_numpy(synth.roi_features(256, 512, 2)).cuda()
ms = timeit(lambda: pairwise_affinity(pf, df, link, se))
print(f"affinity 256^2: {ms:9.3f} ms  ({256*256*(2*512*512*2+2*512)/ms/1e9:6.1f} TF)")
