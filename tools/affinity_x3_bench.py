"""EXPERIMENTAL: the batched link head at the detector's shape (8 x 128 x 128 pairs x 512 channels) with exact-fp32 MFMA
products (affinity.hip) and with split-bf16 products (affinity_x3.hip): time and error vs float64 side by side."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jmodt_amd.ops.affinity import make_affinity_mlp, pairwise_affinity_batched  # noqa: E402

dev = "cuda:0"
torch.manual_seed(0)
link = make_affinity_mlp().to(dev).eval()
g = torch.Generator().manual_seed(1)
pf = torch.relu(torch.randn(8, 128, 512, generator=g)).to(dev)
df = torch.relu(torch.randn(8, 128, 512, generator=g)).to(dev)


def run(split):
    for _ in range(3):
        pairwise_affinity_batched(pf, df, link, None, split_bf16=split)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        pairwise_affinity_batched(pf, df, link, None, split_bf16=split)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / 20


for split in (False, True):
    ms = run(split)
    print(f"split_bf16={split}: {ms:.3f} ms per call = {137.57 / ms:.1f} TF-equivalent")
