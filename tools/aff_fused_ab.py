"""affinity link head: the two-launch GEMM chain (affinity.hip) against the one-kernel form that keeps the hidden activation in LDS
(affinity_fused.hip, the default; tools build: JM_AFF_FUSED=0 selects the two-launch chain, JM_AFF_GRID=256 a persistent grid).  Checks the scores against float64 and times both forms of the SAME process
image by running itself twice.      gpurun -- 'python tools/aff_fused_ab.py'"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) == 1:
    for flag, grid in (("0", "-1"), ("1", "-1"), ("1", "256")):
        env = dict(os.environ, JM_AFF_FUSED=flag, JM_AFF_GRID=grid)
        subprocess.run([sys.executable, os.path.abspath(__file__), flag], env=env, check=False)
    sys.exit(0)
import torch
from jmodt_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "libjmodt_hip_tools.so")
from jmodt_amd.ops.affinity import pairwise_affinity_batched
import torch.nn as nn

torch.manual_seed(0)
dev = torch.device("cuda:0")
C = 512
link = nn.Sequential(nn.Conv1d(C, 512, 1), nn.ReLU(), nn.Conv1d(512, 512, 1), nn.ReLU(), nn.Conv1d(512, 1, 1)).to(dev)
print(f"JM_AFF_FUSED={os.environ.get('JM_AFF_FUSED')} JM_AFF_GRID={os.environ.get('JM_AFF_GRID')} (-1: a workgroup per tile, n: persistent on n workgroups)")
for nb, P, D in ((2, 37, 50), (1, 64, 64), (3, 128, 128), (8, 128, 128), (8, 256, 256)):
    pf = torch.randn(nb, P, C, device=dev)
    df = torch.randn(nb, D, C, device=dev)
    A, raw = pairwise_affinity_batched(pf, df, link, None, return_raw=True)
    if nb * P * D <= 3 * 128 * 128:
        cor = (pf[:, :, None, :] - df[:, None, :, :]).abs().double().reshape(nb, P * D, C).transpose(1, 2)
        want = link.double()(cor).reshape(nb, P, D)
        link.float()
        err = (raw.double() - want).abs().max().item() / max(1.0, want.abs().max().item())
        sm = (torch.softmax(want, 2) + torch.softmax(want, 1)) / 2
        print(f"  {nb} x {P} x {D}: raw score error vs float64 {err:.2e}, link {((A.double() - sm).abs().max().item()):.2e}")
    print(f"  {nb} x {P} x {D}: checksum of the raw scores {raw.double().sum().item():.10f} (differs between the forms: other summation order)")
    for _ in range(3):
        pairwise_affinity_batched(pf, df, link, None)
    torch.cuda.synchronize()
    n = 20
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        pairwise_affinity_batched(pf, df, link, None)
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))
    fl = 2.0 * nb * P * D * (C * 512 + 512 * 512 + 512)
    print(f"  {nb} x {P} x {D}: {ms[n // 2] * 1e3:8.1f} us median ({ms[0] * 1e3:.1f} min)  {fl / (ms[n // 2] * 1e-3) / 1e12:6.1f} TF  {fl / (ms[n // 2] * 1e-3) / 1e12 / 157.3:.3f} of peak")
