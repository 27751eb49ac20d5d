import os, sys, torch
sys.path.insert(0, "/root/repo")
from jmodt_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from jmodt_amd.ops.affinity import pairwise_affinity_batched, make_affinity_mlp
dev = torch.device("cuda:0")
link = make_affinity_mlp(512, (512, 512)).to(dev).eval()
pf = torch.randn(8, 128, 512, device=dev); df = torch.randn(8, 128, 512, device=dev)
for _ in range(3): pairwise_affinity_batched(pf, df, link, None)
torch.cuda.synchronize()
n = 20
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
ev[0].record()
for i in range(n):
    pairwise_affinity_batched(pf, df, link, None); ev[i + 1].record()
torch.cuda.synchronize()
ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))
print(sys.argv[1].split("/")[-1], f"{ms[n // 2] * 1e3:.1f} us median, {ms[0] * 1e3:.1f} min")
