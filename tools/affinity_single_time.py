import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jmodt_amd.ops.affinity import pairwise_affinity, make_affinity_mlp
dev = torch.device("cuda:0")
link = make_affinity_mlp(512, (512, 512)).to(dev).eval(); se = make_affinity_mlp(512, (512, 512)).to(dev).eval()
for P in (64, 100, 128, 256):
    pf = torch.relu(torch.randn(P, 512, device=dev)); df = torch.relu(torch.randn(P, 512, device=dev))
    for _ in range(3): pairwise_affinity(pf, df, link, se)
    torch.cuda.synchronize(); n = 20
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]; ev[0].record()
    for i in range(n):
        pairwise_affinity(pf, df, link, se); ev[i + 1].record()
    torch.cuda.synchronize()
    ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))
    print(f"pairwise_affinity {P} x {P}: {ms[n // 2] * 1e3:.1f} us median")
