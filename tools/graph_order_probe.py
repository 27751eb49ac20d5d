"""is work enqueued on a stream AFTER a graph replay ordered behind the graph's kernels (and the graph behind what was enqueued
before it)?  A graph of a long dependent chain writes a buffer; an eager kernel right behind the replay reads / modifies it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device("cuda:0")
n = 1 << 22
a = torch.zeros(n, device=dev)
out = torch.zeros(n, device=dev)
def chain():
    t = a
    for _ in range(200):
        t = t * 1.0001 + 1.0
    out.copy_(t)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    chain(); chain()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    chain()
torch.cuda.synchronize()
chain(); torch.cuda.synchronize()
want = out.clone()
bad_after = bad_before = 0
for it in range(20):
    out.zero_()
    a.fill_(float(it))                     # eager work BEFORE the replay that the graph must see
    g.replay()
    got = out.clone()                      # eager work right BEHIND the replay
    out.add_(1.0)                          # ... and a modification behind it
    torch.cuda.synchronize()
    a.fill_(float(it)); chain(); torch.cuda.synchronize()
    ref = out.clone()
    bad_after += int(not torch.equal(got, ref))
    g.replay(); torch.cuda.synchronize()
    bad_before += int(not torch.equal(out, ref))
print("reads behind a replay that saw stale data:", bad_after, "of 20; replays that missed the eager fill before them:", bad_before)
# on a non-default stream
s2 = torch.cuda.Stream()
bad = 0
for it in range(20):
    with torch.cuda.stream(s2):
        out.zero_(); a.fill_(float(it)); g.replay(); got = out.clone(); out.add_(1.0)
    torch.cuda.synchronize()
    a.fill_(float(it)); chain(); torch.cuda.synchronize()
    bad += int(not torch.equal(got, out))
print("side stream: stale reads", bad, "of 20")
