"""Isolated times of the training path's row GEMMs (forward / data gradient) on shapes of the joint-mode step, HIP-graph replays of
20 back-to-back calls:   python tools/rows_gemm_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jmodt_amd.ops import rows as R        # noqa: E402

dev = "cuda:0"


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        g.capture_begin()
        for _ in range(20):
            fn()
        g.capture_end()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / 20


def bench(M, k, n, mv=None):
    x, w, b = torch.randn(M, k, device=dev), torch.randn(n, k, device=dev) * 0.1, torch.randn(n, device=dev)
    dy, mask = torch.randn(M, n, device=dev), torch.randn(M, k, device=dev)
    m_dev = torch.tensor([mv], dtype=torch.int32, device=dev) if mv is not None else None
    rows = mv if mv is not None else M
    f = timed(lambda: R.linear_forward(x, w, b, 1, None, m_dev=m_dev))
    d = timed(lambda: R.linear_dgrad(dy, w, 0, k, mask=mask, m_dev=m_dev))
    gf = 2e-6 * rows * k * n
    print(f"M {M:7d} rows {rows:7d} k {k:5d} n {n:5d}: forward {f:7.1f} us ({gf / f:6.2f} TF)   dgrad {d:7.1f} us ({gf / d:6.2f} TF)")


for M, k, n, mv in [(1024, 1024, 512, None), (1024, 512, 512, None), (256, 1536, 512, None), (4096, 768, 512, None), (16384, 256, 128, None),
                    (4096, 256, 256, None), (256, 512, 1024, None), (8192, 256, 512, 2000), (4096, 256, 256, 1200), (32768, 128, 196, 8000),
                    (16384, 128, 128, 5000), (131072, 64, 128, 30000), (65536, 64, 64, 15000), (524288, 32, 64, 40000), (16384, 128, 128, None)]:
    bench(M, k, n, mv)
