import torch, time, sys, os
sys.path.insert(0, os.getcwd())
from jmodt_amd.ops.proposal import argsort_desc_stable
def timeit(fn, it=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/it*1e3
for B,N in ((8,16384),(8,4096),(8,128)):
    x=torch.randn(B,N,device="cuda")
    print(B,N,"lds %.1f us"%timeit(lambda: argsort_desc_stable(x)),"torch %.1f us"%timeit(lambda: torch.sort(x,dim=1,descending=True,stable=True)))
