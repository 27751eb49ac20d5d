"""pure-write ceiling on this GPU: time a fill of tensors of several sizes (hipMemset-class kernel)"""
import torch
def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
for mb in (50, 100, 150, 279, 600, 2000):
    x = torch.empty(mb * 1000 * 1000 // 4, device="cuda")
    y = torch.empty_like(x)
    t = timeit(lambda: x.fill_(1.0))
    tc = timeit(lambda: y.copy_(x))
    print(f"{mb:5d} MB  fill {t*1e3:8.1f} us = {mb/1e3/t:6.2f} TB/s   copy {tc*1e3:8.1f} us = {2*mb/1e3/tc:6.2f} TB/s (read+write)")
