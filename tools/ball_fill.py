"""how full are the ball-query neighbourhoods of the composed workload?  (slots after the last found neighbour repeat
the first index: pointnet2 ball_query semantics — duplicate rows that the max-pool cannot see)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from jmodt_amd.ops.pointnet2 import pointnet2_utils as pu

dev = torch.device("cuda:0")
st = bench.make_detect_state(8, 1236, dev)
stats = []
_bq, _bqd = pu.ball_query, getattr(pu, "ball_query_dual", None)
def rec(tag, idx):
    ns = idx.shape[-1]
    cnt = 1 + (idx[..., 1:] != idx[..., :1]).sum(-1)
    stats.append((tag, tuple(idx.shape), ns, cnt.float().mean().item(), (cnt <= 16).float().mean().item(), (cnt <= 32).float().mean().item(),
                  (((cnt + 15) // 16) * 16).float().mean().item() / ns, (((cnt + 31) // 32) * 32).float().mean().item() / ns))
def bq(radius, nsample, xyz, new_xyz, *a, **k):
    out = _bq(radius, nsample, xyz, new_xyz, *a, **k); rec(f"r={radius}", out); return out
pu.ball_query = bq
if _bqd is not None:
    def bqd(r0, n0, r1, n1, xyz, new_xyz, *a, **k):
        o = _bqd(r0, n0, r1, n1, xyz, new_xyz, *a, **k); rec(f"r={r0}", o[0]); rec(f"r={r1}", o[1]); return o
    pu.ball_query_dual = bqd
with torch.no_grad():
    bench.detect_step(st)
torch.cuda.synchronize()
print(f"{'query':10s} {'idx shape':22s} ns  mean_cnt  P(cnt<=16) P(cnt<=32)  rows kept @16-granularity  @32-granularity")
for s in stats:
    print(f"{s[0]:10s} {str(s[1]):22s} {s[2]:3d} {s[3]:8.1f} {s[4]:10.3f} {s[5]:10.3f} {s[6]:14.3f} {s[7]:20.3f}")
