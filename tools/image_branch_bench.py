"""image branch (4 x conv3x3+BN+ReLU+conv3x3/2, fp32, batch 8, 384x1280) under MIOpen: memory format x find mode"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jmodt_amd.detector import DetectAffinityEngine

def timeit(fn, iters=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

eng = DetectAffinityEngine().cuda()
blocks = eng.rpn.backbone_net.Img_Block
x = torch.randn(8, 3, 384, 1280, device="cuda")
for bench in (False, True):
    torch.backends.cudnn.benchmark = bench
    for fmt_name, fmt in (("nchw", torch.contiguous_format), ("nhwc", torch.channels_last)):
        def run():
            cur = x.contiguous(memory_format=fmt)
            outs = []
            with torch.no_grad():
                for b in blocks:
                    cur = b(cur)
                    outs.append(cur)
            return outs
        t0 = time.time()
        ms = timeit(run)
        per = []
        cur = x.contiguous(memory_format=fmt)
        with torch.no_grad():
            for b in blocks:
                per.append(timeit(lambda b=b, c=cur: b(c), 3)); cur = b(cur)
        print(f"benchmark={bench} {fmt_name}: total {ms:.2f} ms  blocks {['%.2f' % p for p in per]}  (setup {time.time()-t0:.1f}s)", flush=True)
