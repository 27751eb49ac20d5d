"""which torch ops (copies, cats, elementwise) does one composed step still launch, and from where?
torch.profiler over 3 steps of the default bench workload, grouped by op + input shapes + python call site"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench

dev = torch.device("cuda:0")
st = bench.make_detect_state(8, 1236, dev)
for _ in range(3):
    bench.detect_step(st)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as p:
    for _ in range(3):
        bench.detect_step(st)
    torch.cuda.synchronize()
rows = []
for e in p.key_averages(group_by_input_shape=True, group_by_stack_n=6):
    dt = getattr(e, "self_device_time_total", None)
    if dt is None:
        dt = e.self_cuda_time_total
    if dt <= 0 or not e.key.startswith("aten::"):
        continue
    stack = [s for s in e.stack if "jmodt_amd" in s or "bench.py" in s][:2]
    rows.append((dt / 3.0, e.count / 3.0, e.key, str(e.input_shapes)[:90], " <- ".join(s.split("/")[-1][:60] for s in stack)))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"aten ops with device time: {tot / 1e3:.3f} ms per step")
from collections import defaultdict
by = defaultdict(lambda: [0.0, 0.0])
for r in rows:
    by[r[2]][0] += r[0]; by[r[2]][1] += r[1]
for k, v in sorted(by.items(), key=lambda kv: -kv[1][0]):
    print(f"   {k:32s} {v[0]:9.1f} us  x{v[1]:5.1f}")
for r in rows[:110]:
    print(f"{r[0]:9.1f} us  x{r[1]:4.1f}  {r[2]:28s} {r[3]:90s} {r[4]}")
