"""untraced: GPU-side gap between consecutive jm entries of the main chain (HIP events of jmodt_amd.profile), to tell
launch-latency / host gaps from real work.  Prints the largest end->start gaps of one steady-state step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from jmodt_amd.profile import prof
dev = torch.device("cuda:0")
st = bench.make_detect_state(8, 1236, dev)
st["engine"].prefetch_image = False
for _ in range(6): bench.detect_step(st)
torch.cuda.synchronize()
prof.reset(); prof.enabled = True
for _ in range(3): bench.detect_step(st)
torch.cuda.synchronize(); prof.enabled = False
ev = []
for name, recs in prof.records.items():
    n = len(recs) // 3
    for r in recs[2 * n:]:                      # the last of the three steps
        ev.append((name, r[0], r[1]))
base = min(ev, key=lambda e: 0)[1]
t = [(name, base.elapsed_time(s) * 1e3, base.elapsed_time(e) * 1e3) for name, s, e in ev]
t.sort(key=lambda x: x[1])
t0 = t[0][1]
# union busy
busy, gaps, cs, ce, last = 0.0, [], t[0][1], t[0][2], t[0][0]
for name, s, e in t[1:]:
    if s > ce:
        gaps.append((s - ce, last, name, ce - t0)); busy += ce - cs; cs, ce, last = s, e, name
    elif e > ce:
        ce, last = e, name
busy += ce - cs
print(f"step window {(ce - t0) / 1e3:.3f} ms, covered by timed entries/spans {busy / 1e3:.3f} ms, gaps {sum(g[0] for g in gaps) / 1e3:.3f} ms in {len(gaps)}")
for g in sorted(gaps, reverse=True)[:25]:
    print(f"  {g[0]:8.1f} us at +{g[3] / 1e3:7.3f} ms  after {g[1]:50s} before {g[2]}")
