"""Isolated time of the training path's weight gradient + its reduction (csrc/rows_gemm.hip: rows_wgrad_direct_kernel, rows_wgrad_reduce_kernel)
on shapes of the joint-mode step, HIP-graph replays of 20 back-to-back calls:   python tools/rows_wgrad_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jmodt_amd.ops import rows as R
dev='cuda:0'
def bench(M,n,k,mv=None):
    dy=torch.randn(M,n,device=dev); x=torch.randn(M,k,device=dev)
    m_dev = torch.tensor([mv],dtype=torch.int32,device=dev) if mv is not None else None
    for _ in range(3): R.linear_wgrad(dy,[x],m_dev=m_dev)
    torch.cuda.synchronize()
    g=torch.cuda.CUDAGraph()
    s=torch.cuda.Stream()
    with torch.cuda.stream(s):
        R.linear_wgrad(dy,[x],m_dev=m_dev)
        g.capture_begin()
        for _ in range(20): R.linear_wgrad(dy,[x],m_dev=m_dev)
        g.capture_end()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)*1e3/20
    rows = mv if mv is not None else M
    print(f"M {M:7d} rows {rows:7d} n {n:4d} k {k:4d}: {us:7.1f} us  ({rows*(n+k)*4/us/1e3:7.1f} GB/s, {2*rows*n*k/us/1e6:6.2f} TF)")
for M,n,k,mv in [(65536,128,128,None),(65536,64,64,None),(65536,32,32,None),(16384,128,128,None),(524288,64,32,40000),(524288,64,32,None),(262144,16,16,20000),(131072,128,64,30000),
                 (32768,256,196,8000),(8192,512,256,2000),(16384,256,256,None),(2097152,128,128,150000)]:
    bench(M,n,k,mv)
