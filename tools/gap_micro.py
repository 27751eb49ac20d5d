"""isolate the ~100 us bubble in front of the RPN-heads launch: producer kernel -> conv1d_stack heads, back to back"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jmodt_amd.ops.conv1d import PackedConv1dStack
dev = "cuda"
B, N = 8, 16384
W0 = torch.randn(256, 128, device=dev) * 0.1; b0 = torch.randn(256, device=dev) * 0.1
W1 = torch.randn(77, 256, device=dev) * 0.1; b1 = torch.randn(77, device=dev) * 0.1
heads = PackedConv1dStack([(W0, b0, True), (W1, b1, False)], 128)
Wp = torch.randn(128, 128, device=dev) * 0.1; bp = torch.randn(128, device=dev) * 0.1
prod = PackedConv1dStack([(Wp, bp, True)], 128)            # a producer of the same size class as the final fusion: (8,128,16384)
x = torch.randn(B, 128, N, device=dev)
def gap(producer, consumer, n=30):
    for _ in range(3): consumer(producer())
    torch.cuda.synchronize()
    tot = 0.0
    for _ in range(n):
        e0, e1, e2, e3 = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        e0.record(); f = producer(); e1.record(); e2.record(); consumer(f); e3.record()
        torch.cuda.synchronize()
        tot += e1.elapsed_time(e2)
    return tot / n * 1e3
print(f"conv1d producer -> heads           : gap {gap(lambda: prod(x), heads):7.1f} us")
print(f"fill producer -> heads             : gap {gap(lambda: x.mul_(1.0), heads):7.1f} us")
print(f"conv1d producer -> relu (torch)    : gap {gap(lambda: prod(x), torch.relu):7.1f} us")
print(f"conv1d producer -> same producer   : gap {gap(lambda: prod(x), prod):7.1f} us")
# host-side cost of the heads call (python + ctypes), GPU idle
import time
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): heads(x)
t1 = time.perf_counter(); torch.cuda.synchronize()
print(f"host time per heads() call {1e6 * (t1 - t0) / 200:.1f} us")
t0 = time.perf_counter()
for _ in range(200): heads.supported(B, N)
print(f"host time per supported() {1e6 * (time.perf_counter() - t0) / 200:.1f} us")
