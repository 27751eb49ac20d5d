"""does a hipGraph help the 128x128 affinity call (8 launches on two streams)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jmodt_amd import synth
from jmodt_amd.ops.affinity import make_affinity_mlp, pairwise_affinity
torch.manual_seed(0)
link, se = make_affinity_mlp().cuda().eval(), make_affinity_mlp().cuda().eval()
def timeit(fn, iters=50):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
for P in (32, 64, 128):
    pf = torch.from_numpy(synth.roi_features(P, 512, 1)).cuda(); df = torch.from_numpy(synth.roi_features(P, 512, 2)).cuda()
    eager = timeit(lambda: pairwise_affinity(pf, df, link, se))
    ref = [t.clone() for t in pairwise_affinity(pf, df, link, se)]
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): pairwise_affinity(pf, df, link, se)
    torch.cuda.current_stream().wait_stream(side)
    try:
        with torch.cuda.graph(g):
            out = pairwise_affinity(pf, df, link, se)
        graph = timeit(lambda: g.replay())
        ok = all(torch.allclose(a, b, atol=1e-6) for a, b in zip(out, ref))
        print(f"P=D={P}: eager {eager*1e3:7.1f} us   graph replay {graph*1e3:7.1f} us   same={ok}")
    except Exception as ex:
        print(f"P=D={P}: eager {eager*1e3:7.1f} us   capture failed: {type(ex).__name__}: {str(ex)[:200]}")
