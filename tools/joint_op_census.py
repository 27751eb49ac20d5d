"""GPU probe (run through gpurun): which torch operators the joint-mode step launches, by count — the host-side names behind the
`at::native::*` / `__amd_rocclr_copyBuffer` rows of the rocprofv3 table (torch.profiler over 3 steps)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile
import bench

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
st = bench.make_joint_state(4, 1234, dev)
for _ in range(3):
    bench.train_step(st, None)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    for _ in range(3):
        bench.train_step(st, None)
    torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda e: -e.count)
print(f"{'op':60s} {'count/step':>10s} {'cpu us/step':>12s} {'gpu us/step':>12s}")
for e in rows[:70]:
    gpu = getattr(e, "device_time_total", getattr(e, "cuda_time_total", 0))
    print(f"{e.key[:60]:60s} {e.count / 3:10.1f} {e.self_cpu_time_total / 3:12.1f} {gpu / 3:12.1f}")
print()
print("by host time (self CPU, us per step):")
rows = sorted(prof.key_averages(), key=lambda e: -e.self_cpu_time_total)
for e in rows[:45]:
    print(f"{e.key[:70]:70s} {e.count / 3:8.1f} {e.self_cpu_time_total / 3:12.1f}")
