"""time the batched proposal selection and list its kernels (run under rocprofv3 --kernel-trace --stats)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jmodt_amd import synth
from jmodt_amd.ops.proposal import distance_based_proposal
rs, rp = synth.rpn_output(8, 16384, 77)
s, p = torch.from_numpy(rs).cuda(), torch.from_numpy(rp).cuda()
for _ in range(3): distance_based_proposal(s, p, 9000, 100, 0.8, "normal")
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): distance_based_proposal(s, p, 9000, 100, 0.8, "normal")
e1.record(); torch.cuda.synchronize()
print(f"proposal selection B=8: {e0.elapsed_time(e1) / 10:.3f} ms per call")
