"""GPU box: which Python lines issue the ~320 at::native launches of a joint-mode step?  torch.profiler with stacks, CPU side, grouped by
(operator, innermost jmodt_amd / torch.autograd frame)."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                        # noqa: E402
from torch.profiler import ProfilerActivity, profile   # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "joint"
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
st = (bench.make_joint_state if mode == "joint" else bench.make_rcnn_state)(4, 1234, dev)
for _ in range(4):
    bench.train_step(st, 1)
torch.cuda.synchronize()
steps = 3
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as p:
    for _ in range(steps):
        bench.train_step(st, 1)
    torch.cuda.synchronize()
LAUNCHING = ("aten::copy_", "aten::mul", "aten::fill_", "aten::add", "aten::add_", "aten::cat", "aten::sum", "aten::sigmoid", "aten::sub", "aten::div",
             "aten::threshold_backward", "aten::zero_", "aten::clone", "aten::rsqrt", "aten::neg", "aten::tanh", "aten::where", "aten::gt", "aten::mul_",
             "aten::sigmoid_backward", "aten::tanh_backward", "aten::_foreach_add_", "aten::zeros", "aten::index_select", "aten::gather")
agg = collections.Counter()
tim = collections.Counter()
for ev in p.events():
    if ev.name not in LAUNCHING:
        continue
    top, par = "(forward / optimizer: no enclosing operator)", ev.cpu_parent
    chain = []
    while par is not None:
        chain.append(par.name)
        par = par.cpu_parent
    if chain:
        top = " < ".join(c.replace("autograd::engine::evaluate_function: ", "bwd:") for c in chain[-2:][::-1])[:110]
    agg[(ev.name, top)] += 1
    tim[(ev.name, top)] += ev.cpu_time_total
print(f"{mode}: launching aten ops per step, by enclosing operator (count / step, host us / step)")
for (name, top), n in agg.most_common(70):
    print(f"{n / steps:7.1f} {tim[(name, top)] / steps:8.1f}  {name:24s} {top}")
print("total per step:", sum(agg.values()) / steps)
