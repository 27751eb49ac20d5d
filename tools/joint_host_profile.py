"""GPU box: where the HOST time of a training step goes (the joint step is host-bound: 17.8 of 21.2 ms per 4-frame step are enqueue).
cProfile of the calling thread (forward, loss, optimizer) + torch.profiler's CPU-side operator table (which also sees the autograd
engine's device thread, where every custom Function's backward runs).
usage: joint_host_profile.py {joint|rcnn} [steps]"""
import cProfile
import io
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                        # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "joint"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
st = (bench.make_joint_state if mode == "joint" else bench.make_rcnn_state)(4, 1234, dev)
for _ in range(4):
    bench.train_step(st, 1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    bench.train_step(st, 1)
t_host = (time.perf_counter() - t0) / steps
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) / steps
print(f"{mode}: host enqueue {t_host * 1e3:.2f} ms / step, with final sync {t_all * 1e3:.2f} ms / step", flush=True)

pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    bench.train_step(st, 1)
pr.disable()
torch.cuda.synchronize()
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
    print(f"==== cProfile (calling thread), {steps} steps, by {key}")
    print("\n".join(ln[:200] for ln in s.getvalue().splitlines()[:70]), flush=True)

from torch.profiler import ProfilerActivity, profile   # noqa: E402
with profile(activities=[ProfilerActivity.CPU], record_shapes=False) as p:
    for _ in range(steps):
        bench.train_step(st, 1)
    torch.cuda.synchronize()
print(f"==== torch.profiler CPU side, {steps} steps, by self CPU time")
print(p.key_averages().table(sort_by="self_cpu_time_total", row_limit=60, max_name_column_width=70))
print(f"==== by total CPU time")
print(p.key_averages().table(sort_by="cpu_time_total", row_limit=40, max_name_column_width=70))
