"""where does a tile of the fused SA kernel spend its cycles?  (tools build: shader-clock stamps of workgroup 0's first
MFMA wave; RCNN SA1 shape, hoisted first layer).  Prints median cycles per tile between the stamps."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jmodt_amd import _lib
from jmodt_amd.csrc import build as _hip_build
_lib.LIB_PATH = os.environ.get("JM_TOOLS_LIB", _hip_build.TOOLS_LIB)
from jmodt_amd.ops.pointnet2 import fused, pointnet2_utils as pu
from jmodt_amd.ops.pointnet2.pointnet2_modules import PointnetSAModule

torch.manual_seed(0)
R, N, M, ns = 1024, 512, 128, 64
sa = PointnetSAModule(mlp=[128, 128, 128, 128], npoint=M, radius=0.2, nsample=ns, bn=False).cuda().eval()
xyz = (torch.rand(R, N, 3, device="cuda") - 0.5) * torch.tensor([4.0, 2.0, 2.0], device="cuda")
u = torch.randn(R, 128, N, device="cuda")
feats = torch.randn(R, 128, N, device="cuda")
_, new_xyz = pu.farthest_point_sample_xyz(xyz, M)
idx = pu.ball_query(0.2, ns, xyz, new_xyz)
lib = _lib.load()
raw = ctypes.CDLL(_lib.LIB_PATH)
trace = torch.zeros(64 * 8, dtype=torch.int64, device="cuda")
raw.jm_tools_set_sa_trace.argtypes = [ctypes.c_void_p]
names = ["layer A (MFMA)", "hidden epilogue + init", "barrier 1 wait", "last layer (MFMA)", "max epilogue + init", "tile-end barrier wait", "-> next tile start"]
for label, fn in (("hoisted (pre) 128->128->128", lambda: fused.sa_mlp_pre_from_u(u, new_xyz, idx, sa.mlps[0])),):
    fn(); torch.cuda.synchronize()
    raw.jm_tools_set_sa_trace(ctypes.c_void_p(trace.data_ptr()))
    fn(); torch.cuda.synchronize()
    raw.jm_tools_set_sa_trace(None)
    t = trace.cpu().numpy().reshape(64, 8)[4:60]            # skip warm-up tiles
    d = np.diff(t[:, :7], axis=1)
    per_tile = np.diff(t[:, 0])
    print(label, "median cycles per tile:", int(np.median(per_tile)))
    for k in range(6):
        print(f"   {names[k]:28s} {int(np.median(d[:, k])):7d}")
    print(f"   {'(stamp 6 -> next stamp 0)':28s} {int(np.median(t[1:, 0] - t[:-1, 6])):7d}")

# ---- the point-major kernel (csrc/sa_mlp_pm.hip)
raw.jm_tools_set_pm_trace.argtypes = [ctypes.c_void_p]
u_pm = u.transpose(1, 2).contiguous()
fn = lambda: fused.sa_mlp_pre_from_u(u_pm, new_xyz, idx, sa.mlps[0], point_major=True)
fn(); torch.cuda.synchronize()
trace.zero_()
raw.jm_tools_set_pm_trace(ctypes.c_void_p(trace.data_ptr()))
fn(); torch.cuda.synchronize()
raw.jm_tools_set_pm_trace(None)
t = trace.cpu().numpy().reshape(64, 8)[4:60]
d = np.diff(t[:, :7], axis=1)
print("point-major kernel, median cycles per tile:", int(np.median(np.diff(t[:, 0]))))
for k, nm in enumerate(["hidden layer k-loop (incl. bias init)", "hidden epilogue (b128 stores)", "barrier 1 wait", "last layer k-loop", "max-pool partials", "barrier 2 wait"]):
    print(f"   {nm:40s} {int(np.median(d[:, k])):7d}")
print(f"   {'output phase + next tile set-up':40s} {int(np.median(t[1:, 0] - t[:-1, 6])):7d}")
